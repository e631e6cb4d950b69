-- tfhe_gates.lua -- text to append to ao-tfhe/tfhe.lua (same pass-through style as :4-53).
-- Not executed in this repository (no Lua interpreter in the image); integration/node/tfhe.js is its tested twin.
function Tfhe.generateGateKey(lambda, seed) return Tfhe.backend.generateGateKey(lambda, seed) end
function Tfhe.encryptBit(bit, key)          return Tfhe.backend.encryptBit(bit, key) end
function Tfhe.decryptBit(ct, key)           return Tfhe.backend.decryptBit(ct, key) end
function Tfhe.nand(a, b, pk)                return Tfhe.backend.gateNAND(a, b, pk) end
function Tfhe.xor(a, b, pk)                 return Tfhe.backend.gateXOR(a, b, pk) end
function Tfhe.mux(a, b, c, pk)              return Tfhe.backend.gateMUX(a, b, c, pk) end
-- … and/or/nor/xnor/not likewise

-- 8-bit ripple-carry adder over bit-sliced ciphertext tables (LSB first): 2 XOR + 2 AND + 1 OR per bit
function Tfhe.addBits(A, B, pk)
  local S, c = {}, nil
  for i = 1, #A do
    local p = Tfhe.xor(A[i], B[i], pk)
    local g = Tfhe.backend.gateAND(A[i], B[i], pk)
    if c then
      S[i] = Tfhe.xor(p, c, pk)
      c = Tfhe.backend.gateOR(g, Tfhe.backend.gateAND(p, c, pk), pk)
    else S[i], c = p, g end
  end
  S[#A + 1] = c
  return S
end
