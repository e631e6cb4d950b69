// tfhe.js -- the JS twin of ao-tfhe/tfhe.lua: `Tfhe.*` pass-throughs to `Tfhe.backend.*` (tfhe.lua:1-53),
// plus the Boolean gate operations and small circuits this repo adds behind the same surface.
'use strict';
const path = require('path');
const Tfhe = {};
Tfhe.backend = require(path.join(__dirname, 'eoc_tfhe.node'));
const B = Tfhe.backend;

// ---- the reference's 11 functions, same names and argument order (ao-tfhe/tfhe.lua:4-53) ----
Tfhe.info = () => B.info();
Tfhe.testJWT = () => B.testJWT();
Tfhe.generateSecretKey = (jwtToken, jwksBase64) => B.generateSecretKey(jwtToken, jwksBase64);
Tfhe.generatePublicKey = () => B.generatePublicKey();
Tfhe.encryptInteger = (value, key) => B.encryptInteger(value, key);
Tfhe.encryptInteger_dummy = (value, key) => B.encryptInteger_dummy(value, key);
Tfhe.decryptInteger = (value, key, jwtToken, jwksBase64) => B.decryptInteger(value, key, jwtToken, jwksBase64);
Tfhe.addCiphertexts = (c1, c2, pk) => B.addCiphertexts(c1, c2, pk);
// tfhe.lua:41-43 forwards subtract to backend.addCiphertexts (tests/tfhe.test.js:185 pins 50 "-" 8 = 58)
Tfhe.subtractCiphertexts = (c1, c2, pk) => B.addCiphertexts(c1, c2, pk);
Tfhe.encryptASCIIString = (value, length, key) => B.encryptASCIIString(value, length, key);
Tfhe.decryptASCIIString = (value, length, key, jwtToken, jwksBase64) =>
  B.decryptASCIIString(value, length, key, jwtToken, jwksBase64);

// ---- Boolean path (bootstrapped gates on the GPU engine) ----
Tfhe.generateGateKey = (lambda, seed) => B.generateGateKey(lambda, seed);
Tfhe.encryptBit = (bit, key) => B.encryptBit(bit, key);
Tfhe.constantBit = bit => B.constantBit(bit);  // bootsCONSTANT: noiseless trivial sample
Tfhe.decryptBit = (ct, key) => B.decryptBit(ct, key);
Tfhe.nand = (a, b, pk) => B.gateNAND(a, b, pk);
Tfhe.and = (a, b, pk) => B.gateAND(a, b, pk);
Tfhe.or = (a, b, pk) => B.gateOR(a, b, pk);
Tfhe.nor = (a, b, pk) => B.gateNOR(a, b, pk);
Tfhe.xor = (a, b, pk) => B.gateXOR(a, b, pk);
Tfhe.xnor = (a, b, pk) => B.gateXNOR(a, b, pk);
Tfhe.not = (a, pk) => B.gateNOT(a, pk);
Tfhe.mux = (a, b, c, pk) => B.gateMUX(a, b, c, pk);

// ripple-carry adder over bit-sliced ciphertext arrays (LSB first): 2 XOR + 2 AND + 1 OR per bit
Tfhe.addBits = (A, Bs, pk) => {
  const S = [];
  let c = null;
  for (let i = 0; i < A.length; i++) {
    const p = Tfhe.xor(A[i], Bs[i], pk), g = Tfhe.and(A[i], Bs[i], pk);
    if (c === null) { S.push(p); c = g; } else { S.push(Tfhe.xor(p, c, pk)); c = Tfhe.or(g, Tfhe.and(p, c, pk), pk); }
  }
  S.push(c);
  return S;
};
// ASCII string equality: XOR per bit, OR tree, NOT -- batched over the bits with the raw-buffer API
Tfhe.equalBits = (X, Y) => {  // X, Y: Buffers of int32 samples [nbits][n+1]
  const w = B.sampleInts() * 4;
  let level = B.gateBatch(4 /* XOR */, X, Y, null);
  let n = level.length / w;
  while (n > 1) {
    const half = n >> 1;
    const a = level.slice(0, half * w), b = level.slice(half * w, 2 * half * w);
    const next = B.gateBatch(2 /* OR */, Buffer.from(a), Buffer.from(b), null);
    level = (n & 1) ? Buffer.concat([next, level.slice(2 * half * w)]) : next;
    n = level.length / w;
  }
  return B.gateBatch(11 /* NOT */, level, null, null);
};
// string helpers of the gate path: a string travels as 8 bit-ciphertexts per byte, LSB first (raw Buffer of samples)
Tfhe.encryptStringBits = str => B.encryptBits(Buffer.from([...Buffer.from(str)].flatMap(c => [...Array(8).keys()].map(k => (c >> k) & 1))));
Tfhe.equalStrings = (X, Y) => Tfhe.equalBits(X, Y);   // one ciphertext: 1 iff the two encrypted strings are equal
// unsigned comparison and min/max over bit-sliced ciphertext arrays (LSB first), string API:
// lt_0 = (not a_0) and b_0 -- written NOT + AND here; the batch/circuit layer uses bootsANDNY directly --
// lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i)
Tfhe.lessThanBits = (A, Bs, pk) => {
  let lt = Tfhe.and(Tfhe.not(A[0], pk), Bs[0], pk);
  for (let i = 1; i < A.length; i++) lt = Tfhe.mux(Tfhe.xnor(A[i], Bs[i], pk), lt, Bs[i], pk);
  return lt;
};
Tfhe.minMaxBits = (A, Bs, pk) => {
  const lt = Tfhe.lessThanBits(A, Bs, pk);
  return { min: A.map((a, i) => Tfhe.mux(lt, a, Bs[i], pk)), max: A.map((a, i) => Tfhe.mux(lt, Bs[i], a, pk)) };
};
module.exports = Tfhe;
