-- tfhe_gates.lua -- text to append to ao-tfhe/tfhe.lua (same pass-through style as :4-53; needs Lua 5.3 string.pack).
-- Never run by a Lua VM (none in the build image).  Executed by tests/lua_double/minilua.py -- an interpreter for the subset of
-- Lua 5.3 used below -- in tests/test_lua_facade.py: every netlist builder on plaintext bits over all input pairs, the batch
-- functions' wire packing, and facade -> binding C -> library on the GPU.  integration/node/tfhe.js is its twin with the
-- same functions and netlists: tests/test_binding_surfaces.py compares the two name by name (`and`, `or`, `not` are Lua
-- keywords: band / bor / bnot here; base64Decode / base64Encode exist only here: Node has Buffer for that).
function Tfhe.generateGateKey(lambda, seed) return Tfhe.backend.generateGateKey(lambda, seed) end
function Tfhe.resetGateKey()                return Tfhe.backend.resetGateKey() end
function Tfhe.deviceCount()                 return Tfhe.backend.deviceCount() end     -- GPUs the process can see
function Tfhe.engineCount()                 return Tfhe.backend.engineCount() end     -- engines behind the global key
function Tfhe.setDevices(...)               return Tfhe.backend.setDevices(...) end   -- GPUs behind the next gate key
function Tfhe.encryptBit(bit, key)          return Tfhe.backend.encryptBit(bit, key) end
function Tfhe.constantBit(bit)              return Tfhe.backend.constantBit(bit) end
function Tfhe.decryptBit(ct, key)           return Tfhe.backend.decryptBit(ct, key) end
function Tfhe.nand(a, b, pk)                return Tfhe.backend.gateNAND(a, b, pk) end
function Tfhe.band(a, b, pk)                return Tfhe.backend.gateAND(a, b, pk) end   -- `and`, `or`, `not` are Lua keywords
function Tfhe.bor(a, b, pk)                 return Tfhe.backend.gateOR(a, b, pk) end
function Tfhe.nor(a, b, pk)                 return Tfhe.backend.gateNOR(a, b, pk) end
function Tfhe.xor(a, b, pk)                 return Tfhe.backend.gateXOR(a, b, pk) end
function Tfhe.xnor(a, b, pk)                return Tfhe.backend.gateXNOR(a, b, pk) end
function Tfhe.bnot(a, pk)                   return Tfhe.backend.gateNOT(a, pk) end
function Tfhe.mux(a, b, c, pk)              return Tfhe.backend.gateMUX(a, b, c, pk) end
function Tfhe.maj(a, b, c, pk)              return Tfhe.backend.gateMAJ(a, b, c, pk) end   -- extension gates (one bootstrap):
function Tfhe.xor3(a, b, c, pk)             return Tfhe.backend.gateXOR3(a, b, c, pk) end  -- a full adder's carry and sum
function Tfhe.exportSecretKey()             return Tfhe.backend.exportSecretKey() end
function Tfhe.importSecretKey(k)            return Tfhe.backend.importSecretKey(k) end
-- the cloud ("public") key: exported by the client (= generatePublicKey), all a server installs
function Tfhe.exportCloudKey()              return Tfhe.backend.exportCloudKey() end
function Tfhe.importCloudKey(k)             return Tfhe.backend.importCloudKey(k) end
function Tfhe.exportCloudKeyToFile(path)    return Tfhe.backend.exportCloudKeyToFile(path) end
function Tfhe.importCloudKeyFromFile(path)  return Tfhe.backend.importCloudKeyFromFile(path) end
function Tfhe.keyMode()                     return Tfhe.backend.keyMode() end          -- 0 none, 1 secret + cloud, 2 cloud only

-- ---- circuit layer: netlists evaluated by ONE backend call (circuitRun), batched over instances ----
-- wires travel as one binary string [nWires][instances][n+1] of int32 samples; a netlist is packed as 5 int32 per gate
Tfhe.OP = { NAND = 0, AND = 1, OR = 2, NOR = 3, XOR = 4, XNOR = 5, ANDNY = 6, ANDYN = 7, ORNY = 8, ORYN = 9, MUX = 10,
            NOT = 11, COPY = 12, CONST0 = 13, CONST1 = 14,
            MAJ = 15, XOR3 = 16 }   -- extension gates: majority / three-input parity, ONE bootstrap each
local OP = Tfhe.OP
local function newNetlist()
  local nl = { gates = {}, nWires = 0 }
  function nl.wire(n) local w = nl.nWires; nl.nWires = nl.nWires + (n or 1); return w end
  function nl.gate(op, in0, in1, in2)
    local out = nl.wire()
    nl.gates[#nl.gates + 1] = string.pack("<i4i4i4i4i4", op, in0, in1 or -1, in2 or -1, out)
    return out
  end
  function nl.packed() return table.concat(nl.gates) end
  return nl
end
-- ripple-carry adder, LSB first: half adder at bit 0, then 2 XOR + 2 AND + 1 OR per bit (5 nbits - 3 bootstraps);
-- carryInZero = true: a full adder at bit 0 as well, its carry-in bootsCONSTANT(0) -- the uniform 5 gates per bit
-- (40 per 8-bit pair) BASELINE.md counts
function Tfhe.adderNetlist(nbits, carryInZero)
  local nl = newNetlist()
  local a, b, sum, c = nl.wire(nbits), nl.wire(nbits), {}, nil
  if carryInZero then c = nl.gate(OP.CONST0, -1) end
  for i = 0, nbits - 1 do
    local p, g = nl.gate(OP.XOR, a + i, b + i), nl.gate(OP.AND, a + i, b + i)
    if c then
      sum[#sum + 1] = nl.gate(OP.XOR, p, c)
      c = nl.gate(OP.OR, g, nl.gate(OP.AND, p, c))
    else sum[1], c = p, g end
  end
  sum[#sum + 1] = c
  return nl, a, b, sum
end
-- equality of two nbits-wide values: XOR per bit, OR tree, NOT (free)
function Tfhe.equalNetlist(nbits)
  local nl = newNetlist()
  local x, y, level = nl.wire(nbits), nl.wire(nbits), {}
  for i = 0, nbits - 1 do level[#level + 1] = nl.gate(OP.XOR, x + i, y + i) end
  while #level > 1 do
    local nxt = {}
    for i = 1, #level - 1, 2 do nxt[#nxt + 1] = nl.gate(OP.OR, level[i], level[i + 1]) end
    if #level % 2 == 1 then nxt[#nxt + 1] = level[#level] end
    level = nxt
  end
  return nl, x, y, nl.gate(OP.NOT, level[1])
end
-- unsigned a < b (LSB first) and min / max: lt_0 = ANDNY(a_0, b_0); lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i)
function Tfhe.minMaxNetlist(nbits)
  local nl = newNetlist()
  local a, b = nl.wire(nbits), nl.wire(nbits)
  local lt = nl.gate(OP.ANDNY, a, b)
  for i = 1, nbits - 1 do lt = nl.gate(OP.MUX, nl.gate(OP.XNOR, a + i, b + i), lt, b + i) end
  local mn, mx = {}, {}
  for i = 0, nbits - 1 do
    mn[#mn + 1] = nl.gate(OP.MUX, lt, a + i, b + i)
    mx[#mx + 1] = nl.gate(OP.MUX, lt, b + i, a + i)
  end
  return nl, a, b, lt, mn, mx
end
-- a - b mod 2^nbits and the final borrow (= a < b), LSB first: d_i = a_i ^ b_i ^ br_i, br_{i+1} = MUX(a_i ^ b_i, b_i, br_i)
function Tfhe.subtractorNetlist(nbits)
  local nl = newNetlist()
  local a, b, diff = nl.wire(nbits), nl.wire(nbits), {}
  diff[1] = nl.gate(OP.XOR, a, b)
  local br = nl.gate(OP.ANDNY, a, b)
  for i = 1, nbits - 1 do
    local p = nl.gate(OP.XOR, a + i, b + i)
    diff[#diff + 1] = nl.gate(OP.XOR, p, br)
    br = nl.gate(OP.MUX, p, b + i, br)
  end
  return nl, a, b, diff, br
end
-- ---- forms picked by instance count (fewest bootstraps for wide batches, fewest levels for small ones) ----
-- carry as ONE gate per bit: c_{i+1} = MUX(a_i ^ b_i, c_i, a_i); 2 + 4 (nbits - 1) bootstraps on nbits levels (30 / 8 for 8 bits)
function Tfhe.muxAdderNetlist(nbits)
  local nl = newNetlist()
  local a, b, sum = nl.wire(nbits), nl.wire(nbits), {}
  sum[1] = nl.gate(OP.XOR, a, b)
  local c = nl.gate(OP.AND, a, b)
  for i = 1, nbits - 1 do
    local p = nl.gate(OP.XOR, a + i, b + i)
    sum[#sum + 1] = nl.gate(OP.XOR, p, c)
    c = nl.gate(OP.MUX, p, c, a + i)
  end
  sum[#sum + 1] = c
  return nl, a, b, sum
end
-- with the extension gates a full adder is XOR3(a, b, c) + MAJ(a, b, c), one bootstrap each on the level of c:
-- 2 nbits bootstraps on nbits levels (16 / 8 at 8 bits)
function Tfhe.majAdderNetlist(nbits)
  local nl = newNetlist()
  local a, b, sum = nl.wire(nbits), nl.wire(nbits), {}
  sum[1] = nl.gate(OP.XOR, a, b)
  local c = nl.gate(OP.AND, a, b)
  for i = 1, nbits - 1 do
    sum[#sum + 1] = nl.gate(OP.XOR3, a + i, b + i, c)
    c = nl.gate(OP.MAJ, a + i, b + i, c)
  end
  sum[#sum + 1] = c
  return nl, a, b, sum
end
-- a - b and the final borrow: d_i = XOR3(a_i, b_i, br_i), br_{i+1} = MAJ(NOT a_i, b_i, br_i) (NOT is free); 16 / 8 at 8 bits
function Tfhe.majSubtractorNetlist(nbits)
  local nl = newNetlist()
  local a, b, diff = nl.wire(nbits), nl.wire(nbits), {}
  diff[1] = nl.gate(OP.XOR, a, b)
  local br = nl.gate(OP.ANDNY, a, b)
  for i = 1, nbits - 1 do
    local na = nl.gate(OP.NOT, a + i)
    diff[#diff + 1] = nl.gate(OP.XOR3, a + i, b + i, br)
    br = nl.gate(OP.MAJ, na, b + i, br)
  end
  return nl, a, b, diff, br
end
-- unsigned a < b = that borrow alone: ONE bootstrap per bit (8 / 8 at 8 bits)
function Tfhe.majLessThanNetlist(nbits)
  local nl = newNetlist()
  local a, b = nl.wire(nbits), nl.wire(nbits)
  local lt = nl.gate(OP.ANDNY, a, b)
  for i = 1, nbits - 1 do lt = nl.gate(OP.MAJ, nl.gate(OP.NOT, a + i), b + i, lt) end
  return nl, a, b, lt
end
-- logarithmic depth: Sklansky prefix network over (generate, propagate); cell = MUX(P_hi, G_lo, G_hi) + AND(P_hi, P_lo);
-- 48 bootstraps on 5 levels for 8 bits.  sub = true: the same network over (a borrow arises, a borrow passes) =
-- (ANDNY(a, b), XNOR(a, b)) computes a - b and the final borrow.  -> nl, a, b, out bits, carry / borrow out
local function prefixCells(nl, a, b, sub)                 -- a, b: 0-based arrays of wires, LSB first -> out bits (1-based), top
  local nbits = #a + 1
  local out = {}
  local pOp, gOp, oOp = OP.XOR, OP.AND, OP.XOR
  if sub then pOp, gOp, oOp = OP.XNOR, OP.ANDNY, OP.XNOR end
  out[1] = nl.gate(OP.XOR, a[0], b[0])
  if nbits == 1 then return out, nl.gate(gOp, a[0], b[0]) end
  local P, G, pbit, single = {}, {}, {}, {}
  for i = 1, nbits - 1 do P[i] = nl.gate(pOp, a[i], b[i]); pbit[i] = P[i] end
  for i = 0, nbits - 1 do
    single[i] = true
    if i == 0 or (i % 2 == 0 and i + 1 < nbits) then G[i] = nl.gate(gOp, a[i], b[i]) end
  end
  local k = 0
  while (1 << k) < nbits do
    local newG, newP, newS = {}, {}, {}
    for i = 0, nbits - 1 do newG[i] = G[i]; newP[i] = P[i]; newS[i] = single[i] end
    for i = 0, nbits - 1 do
      if (i >> k) & 1 == 1 then
        local j = ((i >> k) << k) - 1
        local ghi = G[i]
        if single[i] then
          -- (the borrow as NOT a_i, a free gate: the optimizer then turns the cell into MAJ(NOT a_i, b_i, G_lo))
          if sub then ghi = nl.gate(OP.NOT, a[i]) else ghi = a[i] end
        end
        newG[i] = nl.gate(OP.MUX, P[i], G[j], ghi)
        if i < (1 << (k + 1)) then newP[i] = nil else newP[i] = nl.gate(OP.AND, P[i], P[j]) end
        newS[i] = false
      end
    end
    G, P, single = newG, newP, newS
    k = k + 1
  end
  for i = 1, nbits - 1 do out[#out + 1] = nl.gate(oOp, pbit[i], G[i - 1]) end
  return out, G[nbits - 1]
end
local function prefixNetwork(nbits, sub)
  local nl = newNetlist()
  local a, b, A, B = nl.wire(nbits), nl.wire(nbits), {}, {}
  for i = 0, nbits - 1 do A[i] = a + i; B[i] = b + i end
  local out, top = prefixCells(nl, A, B, sub)
  return nl, a, b, out, top
end
function Tfhe.prefixAdderNetlist(nbits)
  local nl, a, b, sum, carry = prefixNetwork(nbits, false)
  sum[#sum + 1] = carry
  return nl, a, b, sum
end
function Tfhe.prefixSubtractorNetlist(nbits)              -- -> nl, a, b, difference bits, borrow (= a < b)
  return prefixNetwork(nbits, true)
end
-- unsigned a < b alone, ripple form: lt_0 = ANDNY(a_0, b_0); lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i); 1 + 3 (nbits - 1) bootstraps
function Tfhe.lessThanNetlist(nbits)
  local nl = newNetlist()
  local a, b = nl.wire(nbits), nl.wire(nbits)
  local lt = nl.gate(OP.ANDNY, a, b)
  for i = 1, nbits - 1 do lt = nl.gate(OP.MUX, nl.gate(OP.XNOR, a + i, b + i), lt, b + i) end
  return nl, a, b, lt
end
-- unsigned a < b in logarithmic depth: tree over (LT, EQ) of bit ranges, LT = MUX(EQ_hi, LT_lo, LT_hi); 29 bootstraps on 4
-- levels for 8 bits (lessThanNetlist: 22 on 8)
function Tfhe.lessThanTreeNetlist(nbits)
  local nl = newNetlist()
  local a, b = nl.wire(nbits), nl.wire(nbits)
  local function build(lo, hi, needLt, needEq)
    if hi - lo == 1 then
      local eq, lt = nil, nil
      if needEq then eq = nl.gate(OP.XNOR, a + lo, b + lo) end
      if needLt then lt = nl.gate(OP.ANDNY, a + lo, b + lo) end
      return lt, eq
    end
    local mid = lo + (hi - lo + 1) // 2
    local upSingle = (hi - mid == 1)
    local ltLo, eqLo = build(lo, mid, true, needEq)
    local ltHi, eqHi = build(mid, hi, not upSingle, true)
    local lt, eq = nil, nil
    if needLt then
      local hiv = ltHi
      -- a single bit as the upper operand: where a_i ~= b_i NOT a_i equals b_i; NOT is free, and written this way the
      -- optimizer turns the cell into MAJ(NOT a_i, b_i, LT_lo): one bootstrap
      if upSingle then hiv = nl.gate(OP.NOT, a + mid) end
      lt = nl.gate(OP.MUX, eqHi, ltLo, hiv)
    end
    if needEq then eq = nl.gate(OP.AND, eqHi, eqLo) end
    return lt, eq
  end
  local lt = build(0, nbits, true, false)
  return nl, a, b, lt
end
-- the form of lowest estimated cost for this many instances (backend.netlistCost: below a quarter of the resident set a
-- level costs the same whatever its width, so depth decides for small batches and bootstraps for wide ones)
-- outPos: positions of the builder's output wires (lists or single wires) in what it returns -- the candidates are priced
-- AFTER backend.netlistOptimize, the way runNetlist runs them (the row-by-row multiplier shrinks from 320 to 176 bootstraps,
-- the column form to 230, the prefix adder from 48 to 40, the tree comparator from 29 to 24)
local function cheapest(builders, nbits, instances, outPos)
  local best, bestCost
  for i = 1, #builders do
    local r = { builders[i](nbits) }
    local gates, o = r[1].packed(), {}
    for _, pos in ipairs(outPos) do
      local part = r[pos]
      if type(part) == "table" then
        for k = 1, #part do o[#o + 1] = string.pack("<i4", part[k]) end
      else o[#o + 1] = string.pack("<i4", part) end
    end
    gates = Tfhe.backend.netlistOptimize(gates, table.concat(o)) or gates
    local cost = Tfhe.backend.netlistCost(gates, instances)
    if not best or (cost >= 0 and cost < bestCost) then best, bestCost = r, cost end
  end
  return table.unpack(best)
end
function Tfhe.adderNetlistFor(nbits, instances)
  return cheapest({ Tfhe.majAdderNetlist, Tfhe.prefixAdderNetlist }, nbits, instances, { 4 })
end
function Tfhe.lessThanNetlistFor(nbits, instances)
  return cheapest({ Tfhe.majLessThanNetlist, Tfhe.lessThanTreeNetlist }, nbits, instances, { 4 })
end
function Tfhe.multiplierNetlistFor(nbits, instances)
  return cheapest({ Tfhe.multiplierNetlist, Tfhe.wallaceMultiplierNetlist }, nbits, instances, { 4 })
end
function Tfhe.subtractorNetlistFor(nbits, instances)
  return cheapest({ Tfhe.majSubtractorNetlist, Tfhe.prefixSubtractorNetlist }, nbits, instances, { 4, 5 })
end
-- (min, max) behind a comparator: min_i = MUX(lt, a_i, b_i) and max_i = MUX(lt, b_i, a_i) -- or, xor3Select,
-- max_i = XOR3(a_i, b_i, min_i) (min_i XOR max_i = a_i XOR b_i): ONE bootstrap instead of the MUX's two, one level later
function Tfhe.minMaxNetlistOn(nbits, xor3Select, nl, a, b, lt)
  local mn, mx = {}, {}
  for i = 0, nbits - 1 do
    mn[#mn + 1] = nl.gate(OP.MUX, lt, a + i, b + i)
    if xor3Select then mx[#mx + 1] = nl.gate(OP.XOR3, a + i, b + i, mn[#mn])
    else mx[#mx + 1] = nl.gate(OP.MUX, lt, b + i, a + i) end
  end
  return nl, a, b, lt, mn, mx
end
-- every comparator form with both ways of selecting the maximum, the cheapest for this many instances: tree comparator + two
-- MUXes per bit for small batches (8 bits: 56 bootstraps on 5 levels after the optimizer), MAJ chain + MUX + XOR3 for wide
-- ones (32 on 10; with two MUXes 40 on 9)
function Tfhe.minMaxNetlistFor(nbits, instances)
  local builders = {}
  for _, lt in ipairs({ Tfhe.majLessThanNetlist, Tfhe.lessThanTreeNetlist }) do
    for _, x3 in ipairs({ false, true }) do
      builders[#builders + 1] = function(n) return Tfhe.minMaxNetlistOn(n, x3, lt(n)) end
    end
  end
  return cheapest(builders, nbits, instances, { 5, 6 })
end
-- a * b -> 2 nbits bits, LSB first: nbits^2 AND partial products, nbits - 1 shifted ripple-carry rows
function Tfhe.multiplierNetlist(nbits)
  local nl = newNetlist()
  local a, b, pp = nl.wire(nbits), nl.wire(nbits), {}
  for r = 0, nbits - 1 do
    pp[r] = {}
    for j = 0, nbits - 1 do pp[r][j] = nl.gate(OP.AND, a + j, b + r) end
  end
  local prod, acc, top = { pp[0][0] }, {}, nil          -- acc[0 .. nbits - 2]: the row above the current one, shifted
  for j = 1, nbits - 1 do acc[j - 1] = pp[0][j] end
  for r = 1, nbits - 1 do
    local nxt, carry = {}, nil
    for j = 0, nbits - 1 do
      local x, y = top, pp[r][j]
      if j < nbits - 1 then x = acc[j] end
      if x == nil and carry == nil then nxt[j] = y
      elseif x == nil or carry == nil then
        local z = x
        if z == nil then z = carry end
        nxt[j] = nl.gate(OP.XOR, z, y); carry = nl.gate(OP.AND, z, y)
      else
        local p, g = nl.gate(OP.XOR, x, y), nl.gate(OP.AND, x, y)
        nxt[j] = nl.gate(OP.XOR, p, carry); carry = nl.gate(OP.OR, g, nl.gate(OP.AND, p, carry))
      end
    end
    prod[#prod + 1] = nxt[0]
    for j = 1, nbits - 1 do acc[j - 1] = nxt[j] end
    top = carry
  end
  for j = 0, nbits - 2 do prod[#prod + 1] = acc[j] end
  if top == nil then top = nl.gate(OP.CONST0, -1) end
  prod[#prod + 1] = top
  return nl, a, b, prod
end
-- a * b in logarithmic depth: partial products in columns by weight, Dadda column compression by full adders (XOR3 + MAJ: the
-- extension gates, one level) and half adders until no column holds more than two wires, then ONE parallel-prefix addition of
-- the two remaining rows; 8 bits: 244 bootstraps on 11 levels against the row-by-row form's 320 on 40
function Tfhe.wallaceMultiplierNetlist(nbits)
  local nl = newNetlist()
  local a, b = nl.wire(nbits), nl.wire(nbits)
  if nbits == 1 then return nl, a, b, { nl.gate(OP.AND, a, b), nl.gate(OP.CONST0, -1) } end
  local ncol = 2 * nbits
  local cols = {}
  for c = 0, ncol - 1 do cols[c] = {} end
  for r = 0, nbits - 1 do
    for j = 0, nbits - 1 do
      local col = cols[r + j]
      col[#col + 1] = { 1, nl.gate(OP.AND, a + j, b + r) }
    end
  end
  local function sorted(col)                            -- by (level, wire): insertion sort (columns hold a handful of wires)
    local t = {}
    for i = 1, #col do
      local e, k = col[i], i - 1
      while k >= 1 and (t[k][1] > e[1] or (t[k][1] == e[1] and t[k][2] > e[2])) do t[k + 1] = t[k]; k = k - 1 end
      t[k + 1] = e
    end
    return t
  end
  -- Dadda's schedule: column heights come down through 9, 6, 4, 3, 2; in a layer every column is reduced to the target with
  -- as few adders as possible (a full adder = XOR3 + MAJ removes two wires, a half adder = XOR + AND one), counting the
  -- carries the column below sends up in the same layer
  local targets = { 2 }
  while targets[#targets] * 3 // 2 < nbits do targets[#targets + 1] = targets[#targets] * 3 // 2 end
  for ti = #targets, 1, -1 do
    local target = targets[ti]
    local new = {}
    for c = 0, ncol - 1 do new[c] = {} end
    for c = 0, ncol - 1 do
      local col = sorted(cols[c])
      local i = 1
      while #col - i + 1 + #new[c] > target do
        local sm, cy
        if #col - i + 1 + #new[c] >= target + 2 and #col - i + 1 >= 3 then
          local x, y, z = col[i], col[i + 1], col[i + 2]
          local lv = x[1]
          if y[1] > lv then lv = y[1] end
          if z[1] > lv then lv = z[1] end
          lv = lv + 1
          sm = { lv, nl.gate(OP.XOR3, x[2], y[2], z[2]) }
          cy = { lv, nl.gate(OP.MAJ, x[2], y[2], z[2]) }
          i = i + 3
        else
          local x, y = col[i], col[i + 1]
          local lv = x[1]
          if y[1] > lv then lv = y[1] end
          lv = lv + 1
          sm = { lv, nl.gate(OP.XOR, x[2], y[2]) }
          cy = { lv, nl.gate(OP.AND, x[2], y[2]) }
          i = i + 2
        end
        local nc, nc1 = new[c], new[c + 1]
        nc[#nc + 1] = sm
        nc1[#nc1 + 1] = cy
      end
      local nc = new[c]
      for k = i, #col do nc[#nc + 1] = col[k] end
    end
    cols = new
  end
  local prod, c0 = {}, 0
  while c0 < ncol and #cols[c0] <= 1 do                  -- low columns that are already final
    if #cols[c0] == 1 then prod[#prod + 1] = cols[c0][1][2] else prod[#prod + 1] = -1 end
    c0 = c0 + 1
  end
  local hi = ncol - 1
  while hi >= c0 and #cols[hi] == 0 do hi = hi - 1 end
  if hi >= c0 then
    local zero, xs, ys = nil, {}, {}
    for c = c0, hi do
      local col = sorted(cols[c])
      xs[c - c0] = col[1][2]
      if #col > 1 then ys[c - c0] = col[2][2]
      else
        if zero == nil then zero = nl.gate(OP.CONST0, -1) end
        ys[c - c0] = zero
      end
    end
    local out, top = prefixCells(nl, xs, ys, false)
    for i = 1, #out do prod[#prod + 1] = out[i] end
    prod[#prod + 1] = top
  end
  local res = {}
  for k = 1, ncol do
    local w = prod[k]
    if w == nil or w == -1 then w = nl.gate(OP.CONST0, -1) end
    res[k] = w
  end
  return nl, a, b, res
end
-- run a netlist over `instances` instances; inputs = { [firstWire] = samples [k][instances][n+1] };
-- outputs (optional): the wires the caller reads afterwards -- the netlist is then rewritten first (NOT folding, MUX fusion)
function Tfhe.runNetlist(nl, inputs, instances, outputs)
  local plane = instances * Tfhe.backend.sampleInts() * 4
  local parts, w = {}, 0
  while w < nl.nWires do                                  -- assemble the wire array plane by plane
    local buf = inputs[w]
    if buf then parts[#parts + 1] = buf; w = w + #buf // plane
    else parts[#parts + 1] = string.rep("\0", plane); w = w + 1 end
  end
  local gates = nl.packed()
  if outputs then
    local o = {}
    for i = 1, #outputs do o[i] = string.pack("<i4", outputs[i]) end
    gates = Tfhe.backend.netlistOptimize(gates, table.concat(o)) or gates
  end
  return Tfhe.backend.circuitRun(gates, table.concat(parts), nl.nWires, instances)
end
local function planes(wires, first, count, instances)
  local plane = instances * Tfhe.backend.sampleInts() * 4
  return wires:sub(first * plane + 1, (first + count) * plane)
end
-- base64 ciphertext strings <-> raw samples (wire format of export_lweSample_toStream: a[n] | b | f64 variance,
-- eoc-tfhe-run.cpp:293-295); Lua 5.3 has no base64 in its standard library
local B64 = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"
local B64DEC = {}
for i = 1, 64 do B64DEC[B64:byte(i)] = i - 1 end
local function b64decode(s)                               -- stops at the first non-alphabet byte, like the reference decoder
  local out, acc, bits = {}, 0, 0
  for i = 1, #s do
    local v = B64DEC[s:byte(i)]
    if not v then break end
    acc = ((acc << 6) | v) & 0xFFFFFF
    bits = bits + 6
    if bits >= 8 then
      bits = bits - 8
      out[#out + 1] = string.char((acc >> bits) & 0xFF)
    end
  end
  return table.concat(out)
end
local function b64char(v) return B64:sub(v + 1, v + 1) end
local function b64encode(s)
  local out = {}
  for i = 1, #s, 3 do
    local a, b, c = s:byte(i, i + 2)
    local v = (a << 16) | ((b or 0) << 8) | (c or 0)
    out[#out + 1] = b64char(v >> 18) .. b64char((v >> 12) & 63) .. (b and b64char((v >> 6) & 63) or "=") .. (c and b64char(v & 63) or "=")
  end
  return table.concat(out)
end
Tfhe.base64Decode = b64decode
Tfhe.base64Encode = b64encode
local function strToSample(s) return b64decode(s):sub(1, Tfhe.backend.sampleInts() * 4) end
local function sampleToStr(buf) return b64encode(buf .. string.rep("\0", 8)) end
local function stack(arr)
  local parts = {}
  for i = 1, #arr do parts[i] = strToSample(arr[i]) end
  return table.concat(parts)
end
local function joined(...)                                -- wire lists end to end
  local out = {}
  for _, part in ipairs({ ... }) do
    for i = 1, #part do out[#out + 1] = part[i] end
  end
  return out
end
local function pick(wires, ws)                            -- wires of ONE instance -> array of base64 ciphertext strings
  local out = {}
  for i = 1, #ws do out[i] = sampleToStr(planes(wires, ws[i], 1, 1)) end
  return out
end
-- string-API circuits: arrays of base64 bit ciphertexts (LSB first) in, arrays out -- ONE backend call per circuit
function Tfhe.addBits(A, B)                               -- -> #A + 1 ciphertexts (one instance: the log-depth form)
  local nl, a, b, sum = Tfhe.adderNetlistFor(#A, 1)
  local wires = Tfhe.runNetlist(nl, { [a] = stack(A), [b] = stack(B) }, 1, sum)
  return wires and pick(wires, sum)
end
function Tfhe.lessThanBits(A, B)                          -- -> one ciphertext: 1 iff A < B (unsigned; the log-depth form)
  local nl, a, b, lt = Tfhe.lessThanNetlistFor(#A, 1)
  local wires = Tfhe.runNetlist(nl, { [a] = stack(A), [b] = stack(B) }, 1, { lt })
  return wires and sampleToStr(planes(wires, lt, 1, 1))
end
function Tfhe.minMaxBits(A, B)                            -- -> min, max (arrays of #A ciphertexts)
  local nl, a, b, lt, mn, mx = Tfhe.minMaxNetlistFor(#A, 1)
  local wires = Tfhe.runNetlist(nl, { [a] = stack(A), [b] = stack(B) }, 1, joined(mn, mx))
  if not wires then return nil end
  return pick(wires, mn), pick(wires, mx)
end
-- raw-buffer circuits over many instances: operands are samples [nbits][instances][n+1]
function Tfhe.addBitsBatch(A, B, nbits, instances)      -- the form is picked by the instance count (adderNetlistFor)
  local nl, a, b, sum = Tfhe.adderNetlistFor(nbits, instances)
  local wires = Tfhe.runNetlist(nl, { [a] = A, [b] = B }, instances, sum)
  if not wires then return nil end
  local out = {}
  for i = 1, #sum do out[i] = planes(wires, sum[i], 1, instances) end
  return table.concat(out)                                -- [nbits + 1][instances][n+1]
end
function Tfhe.lessThanBitsBatch(A, B, nbits, instances)  -- -> [instances][n+1]: 1 iff A < B (unsigned); the form by instance count
  local nl, a, b, lt = Tfhe.lessThanNetlistFor(nbits, instances)
  local wires = Tfhe.runNetlist(nl, { [a] = A, [b] = B }, instances, { lt })
  return wires and planes(wires, lt, 1, instances)
end
function Tfhe.subtractBitsBatch(A, B, nbits, instances)  -- -> [nbits + 1][instances][n+1]: difference bits, then the borrow
  local nl, a, b, diff, borrow = Tfhe.subtractorNetlistFor(nbits, instances)
  local wires = Tfhe.runNetlist(nl, { [a] = A, [b] = B }, instances, joined(diff, { borrow }))
  if not wires then return nil end
  local out = {}
  for i = 1, #diff do out[i] = planes(wires, diff[i], 1, instances) end
  out[#out + 1] = planes(wires, borrow, 1, instances)
  return table.concat(out)
end
function Tfhe.multiplyBitsBatch(A, B, nbits, instances)  -- -> [2 nbits][instances][n+1]
  local nl, a, b, prod = Tfhe.multiplierNetlistFor(nbits, instances)
  local wires = Tfhe.runNetlist(nl, { [a] = A, [b] = B }, instances, prod)    -- through netlistOptimize (carry rewrite, constants)
  if not wires then return nil end
  local out = {}
  for i = 1, #prod do out[i] = planes(wires, prod[i], 1, instances) end
  return table.concat(out)
end
function Tfhe.equalBits(X, Y)                             -- X, Y: samples [nbits][n+1]; one ciphertext: 1 iff equal
  local nbits = #X // (Tfhe.backend.sampleInts() * 4)
  local nl, x, y, out = Tfhe.equalNetlist(nbits)
  local wires = Tfhe.runNetlist(nl, { [x] = X, [y] = Y }, 1)
  return wires and planes(wires, out, 1, 1)
end
-- a string travels as 8 bit-ciphertexts per byte, LSB first
function Tfhe.encryptStringBits(str)
  local bits = {}
  for i = 1, #str do
    local c = str:byte(i)
    for k = 0, 7 do bits[#bits + 1] = string.char((c >> k) & 1) end
  end
  return Tfhe.backend.encryptBits(table.concat(bits))
end
function Tfhe.equalStrings(X, Y) return Tfhe.equalBits(X, Y) end
function Tfhe.minMaxBitsBatch(A, B, nbits, instances)
  local nl, a, b, lt, mn, mx = Tfhe.minMaxNetlistFor(nbits, instances)
  local wires = Tfhe.runNetlist(nl, { [a] = A, [b] = B }, instances, joined(mn, mx, { lt }))
  if not wires then return nil end
  local lo, hi = {}, {}
  for i = 1, nbits do lo[i] = planes(wires, mn[i], 1, instances); hi[i] = planes(wires, mx[i], 1, instances) end
  return table.concat(lo), table.concat(hi), planes(wires, lt, 1, instances)
end

-- ---- deferred gates: the reference's call style (one ciphertext operation per Lua call, tfhe.lua:4-53), ONE backend call ----
-- A gate call on the string API costs a whole blind rotation's n sequential steps (1.8 ms) however little it computes.  A
-- deferred circuit records the same calls on wire handles and evaluates them together: `run` sends the recorded netlist
-- through backend.netlistOptimize and ONE backend.circuitRun, where every LEVEL costs those 1.8 ms -- a hundred gates on ten
-- levels take 18 ms instead of 177.
--   local c = Tfhe.newCircuit()
--   local x, y = c.input(ctX), c.input(ctY)            -- base64 ciphertext strings in (or c.inputSamples(buf) for batches)
--   local s, k = c.xor(x, y), c.band(x, y)             -- handles out: nothing runs yet
--   local outs = c.run({ s, k })                       -- array of base64 ciphertext strings, one backend call
function Tfhe.newCircuit()
  local nl, inputs, c = newNetlist(), {}, { instances = nil }
  local function addInput(buf, instances)
    if c.instances and c.instances ~= instances then return nil end   -- every input of a circuit has the same instance count
    c.instances = instances
    local w = nl.wire(1)
    inputs[w] = buf
    return w
  end
  function c.input(ct) return addInput(strToSample(ct), 1) end         -- one base64 ciphertext string
  function c.inputSamples(buf)                                          -- raw samples [instances][n+1]
    return addInput(buf, #buf // (Tfhe.backend.sampleInts() * 4))
  end
  function c.constant(bit)
    if bit == 0 then return nl.gate(OP.CONST0, -1) end
    return nl.gate(OP.CONST1, -1)
  end
  function c.nand(a, b) return nl.gate(OP.NAND, a, b) end
  function c.band(a, b) return nl.gate(OP.AND, a, b) end
  function c.bor(a, b) return nl.gate(OP.OR, a, b) end
  function c.nor(a, b) return nl.gate(OP.NOR, a, b) end
  function c.xor(a, b) return nl.gate(OP.XOR, a, b) end
  function c.xnor(a, b) return nl.gate(OP.XNOR, a, b) end
  function c.bnot(a) return nl.gate(OP.NOT, a) end
  function c.mux(a, b, d) return nl.gate(OP.MUX, a, b, d) end
  function c.maj(a, b, d) return nl.gate(OP.MAJ, a, b, d) end
  function c.xor3(a, b, d) return nl.gate(OP.XOR3, a, b, d) end
  function c.gateCount() return #nl.gates end
  function c.netlist() return nl end
  -- outs: array of handles -> array of base64 strings (instances == 1) or of raw sample buffers [instances][n+1]
  function c.run(outs)
    local instances = c.instances or 1
    local wires = Tfhe.runNetlist(nl, inputs, instances, outs)
    if not wires then return nil end
    local res = {}
    for i = 1, #outs do
      local buf = planes(wires, outs[i], 1, instances)
      if instances == 1 then res[i] = sampleToStr(buf) else res[i] = buf end
    end
    return res
  end
  return c
end

