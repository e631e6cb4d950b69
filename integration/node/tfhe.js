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
Tfhe.maj = (a, b, c, pk) => B.gateMAJ(a, b, c, pk);     // extension gates (one bootstrap): a full adder's carry ...
Tfhe.xor3 = (a, b, c, pk) => B.gateXOR3(a, b, c, pk);   // ... and its sum
// ---- keys: the secret key stays with the client, the cloud ("public") key is all a server installs ----
Tfhe.resetGateKey = () => B.resetGateKey();
Tfhe.setDevices = devices => B.setDevices(Int32Array.from(devices));   // GPUs behind the next gate key / cloud key
Tfhe.exportSecretKey = () => B.exportSecretKey();
Tfhe.importSecretKey = k => B.importSecretKey(k);
Tfhe.exportCloudKey = () => B.exportCloudKey();                 // = generatePublicKey(): base64 of the EOCCK1 blob
Tfhe.importCloudKey = k => B.importCloudKey(k);                 // server: cloud-key-only context (no encrypt / decrypt)
Tfhe.exportCloudKeyToFile = path => B.exportCloudKeyToFile(path);
Tfhe.importCloudKeyFromFile = path => B.importCloudKeyFromFile(path);
Tfhe.keyMode = () => B.keyMode();                               // 0 none, 1 secret + cloud, 2 cloud only
Tfhe.deviceCount = () => B.deviceCount();
Tfhe.engineCount = () => B.engineCount();

// ---- circuit layer: netlists evaluated by ONE backend call (eoc_global_circuit_run), batched over instances --------
// A netlist is a list of gates {op, in0, in1, in2, out} over numbered wires; wires travel as one Buffer
// [nWires][instances][n+1] of int32 samples.  The builders mirror eoc_tfhe_amd/circuits.py.
const OP = { NAND: 0, AND: 1, OR: 2, NOR: 3, XOR: 4, XNOR: 5, ANDNY: 6, ANDYN: 7, ORNY: 8, ORYN: 9, MUX: 10, NOT: 11, COPY: 12,
             CONST0: 13, CONST1: 14,
             MAJ: 15, XOR3: 16 };   // extension gates: majority / three-input parity, ONE bootstrap each
Tfhe.OP = OP;
class Netlist {
  constructor() { this.gates = []; this.nWires = 0; }
  wire(n = 1) { const w = this.nWires; this.nWires += n; return w; }
  gate(op, in0, in1 = -1, in2 = -1) { const out = this.wire(); this.gates.push([op, in0, in1, in2, out]); return out; }
  packed() { return Int32Array.from(this.gates.flat()); }
}
Tfhe.Netlist = Netlist;
// ripple-carry adder, LSB first: half adder at bit 0, then 2 XOR + 2 AND + 1 OR per bit (5 nbits - 3 bootstraps)
// carryInZero: a full adder at bit 0 as well, its carry-in bootsCONSTANT(0) -- the uniform 5 gates per bit (40 per 8-bit
// pair) BASELINE.md counts
Tfhe.adderNetlist = (nbits, carryInZero) => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), sum = [];
  let c = carryInZero ? nl.gate(OP.CONST0, -1) : -1;
  for (let i = 0; i < nbits; i++) {
    const p = nl.gate(OP.XOR, a + i, b + i), g = nl.gate(OP.AND, a + i, b + i);
    if (c < 0) { sum.push(p); c = g; } else { sum.push(nl.gate(OP.XOR, p, c)); c = nl.gate(OP.OR, g, nl.gate(OP.AND, p, c)); }
  }
  sum.push(c);
  return { nl, a, b, sum };
};
// equality of two nbits-wide values: XOR per bit, OR tree, NOT (free)
Tfhe.equalNetlist = nbits => {
  const nl = new Netlist(), x = nl.wire(nbits), y = nl.wire(nbits);
  let level = [...Array(nbits).keys()].map(i => nl.gate(OP.XOR, x + i, y + i));
  while (level.length > 1) {
    const next = [];
    for (let i = 0; i + 1 < level.length; i += 2) next.push(nl.gate(OP.OR, level[i], level[i + 1]));
    if (level.length & 1) next.push(level[level.length - 1]);
    level = next;
  }
  return { nl, x, y, out: nl.gate(OP.NOT, level[0]) };
};
// unsigned a < b, LSB first: lt_0 = ANDNY(a_0, b_0); lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i); min / max by MUX
Tfhe.minMaxNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  let lt = nl.gate(OP.ANDNY, a, b);
  for (let i = 1; i < nbits; i++) lt = nl.gate(OP.MUX, nl.gate(OP.XNOR, a + i, b + i), lt, b + i);
  const min = [], max = [];
  for (let i = 0; i < nbits; i++) { min.push(nl.gate(OP.MUX, lt, a + i, b + i)); max.push(nl.gate(OP.MUX, lt, b + i, a + i)); }
  return { nl, a, b, lt, min, max };
};
// a - b mod 2^nbits and the final borrow (= a < b), LSB first: d_i = a_i ^ b_i ^ br_i, br_{i+1} = MUX(a_i ^ b_i, b_i, br_i)
Tfhe.subtractorNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), diff = [];
  diff.push(nl.gate(OP.XOR, a, b));
  let br = nl.gate(OP.ANDNY, a, b);
  for (let i = 1; i < nbits; i++) {
    const p = nl.gate(OP.XOR, a + i, b + i);
    diff.push(nl.gate(OP.XOR, p, br));
    br = nl.gate(OP.MUX, p, b + i, br);
  }
  return { nl, a, b, diff, borrow: br };
};
// a * b -> 2 nbits bits, LSB first: nbits^2 AND partial products, nbits - 1 shifted ripple-carry rows
Tfhe.multiplierNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  const pp = [...Array(nbits).keys()].map(r => [...Array(nbits).keys()].map(j => nl.gate(OP.AND, a + j, b + r)));
  const prod = [pp[0][0]];
  let acc = pp[0].slice(1), top = null;
  for (let r = 1; r < nbits; r++) {
    const next = [];
    let carry = null;
    for (let j = 0; j < nbits; j++) {
      const x = j < acc.length ? acc[j] : top, y = pp[r][j];
      if (x === null && carry === null) { next.push(y); carry = null; continue; }
      if (x === null || carry === null) {
        const z = x === null ? carry : x;
        next.push(nl.gate(OP.XOR, z, y)); carry = nl.gate(OP.AND, z, y);
      } else {
        const p = nl.gate(OP.XOR, x, y), g = nl.gate(OP.AND, x, y);
        next.push(nl.gate(OP.XOR, p, carry)); carry = nl.gate(OP.OR, g, nl.gate(OP.AND, p, carry));
      }
    }
    prod.push(next[0]); acc = next.slice(1); top = carry;
  }
  prod.push(...acc);
  prod.push(top === null ? nl.gate(OP.CONST0, -1) : top);
  return { nl, a, b, prod };
};
// ---- forms picked by instance count (fewest bootstraps for wide batches, fewest levels for small ones) ----
// carry as ONE gate per bit: c_{i+1} = MUX(a_i ^ b_i, c_i, a_i); 2 + 4 (nbits - 1) bootstraps on nbits levels (30 / 8 for 8 bits)
Tfhe.muxAdderNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), sum = [];
  sum.push(nl.gate(OP.XOR, a, b));
  let c = nl.gate(OP.AND, a, b);
  for (let i = 1; i < nbits; i++) {
    const p = nl.gate(OP.XOR, a + i, b + i);
    sum.push(nl.gate(OP.XOR, p, c));
    c = nl.gate(OP.MUX, p, c, a + i);
  }
  sum.push(c);
  return { nl, a, b, sum };
};
// with the extension gates a full adder is XOR3(a, b, c) + MAJ(a, b, c), one bootstrap each on the level of c:
// 2 nbits bootstraps on nbits levels (16 / 8 at 8 bits)
Tfhe.majAdderNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), sum = [];
  sum.push(nl.gate(OP.XOR, a, b));
  let c = nl.gate(OP.AND, a, b);
  for (let i = 1; i < nbits; i++) {
    sum.push(nl.gate(OP.XOR3, a + i, b + i, c));
    c = nl.gate(OP.MAJ, a + i, b + i, c);
  }
  sum.push(c);
  return { nl, a, b, sum };
};
// a - b and the final borrow: d_i = XOR3(a_i, b_i, br_i), br_{i+1} = MAJ(NOT a_i, b_i, br_i) (NOT is free); 16 / 8 at 8 bits
Tfhe.majSubtractorNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), diff = [];
  diff.push(nl.gate(OP.XOR, a, b));
  let br = nl.gate(OP.ANDNY, a, b);
  for (let i = 1; i < nbits; i++) {
    const na = nl.gate(OP.NOT, a + i);
    diff.push(nl.gate(OP.XOR3, a + i, b + i, br));
    br = nl.gate(OP.MAJ, na, b + i, br);
  }
  return { nl, a, b, diff, borrow: br };
};
// unsigned a < b = that borrow alone: ONE bootstrap per bit (8 / 8 at 8 bits)
Tfhe.majLessThanNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  let lt = nl.gate(OP.ANDNY, a, b);
  for (let i = 1; i < nbits; i++) lt = nl.gate(OP.MAJ, nl.gate(OP.NOT, a + i), b + i, lt);
  return { nl, a, b, lt };
};
// logarithmic depth: Sklansky prefix network over (generate, propagate); cell = MUX(P_hi, G_lo, G_hi) + AND(P_hi, P_lo);
// 48 bootstraps on 5 levels for 8 bits.  sub: the same network over (a borrow arises, a borrow passes) =
// (ANDNY(a, b), XNOR(a, b)) computes a - b and the final borrow
const prefixCells = (nl, a, b, sub) => {               // a, b: arrays of wires, LSB first -> { out, top }
  const nbits = a.length;
  const [pOp, gOp, oOp] = sub ? [OP.XNOR, OP.ANDNY, OP.XNOR] : [OP.XOR, OP.AND, OP.XOR];
  const out = [nl.gate(OP.XOR, a[0], b[0])];
  if (nbits === 1) return { out, top: nl.gate(gOp, a[0], b[0]) };
  let P = [null];
  for (let i = 1; i < nbits; i++) P.push(nl.gate(pOp, a[i], b[i]));
  const pbit = P.slice();
  let G = [];
  for (let i = 0; i < nbits; i++) G.push((i === 0 || (i % 2 === 0 && i + 1 < nbits)) ? nl.gate(gOp, a[i], b[i]) : null);
  let single = Array(nbits).fill(true);
  for (let k = 0; (1 << k) < nbits; k++) {
    const newG = G.slice(), newP = P.slice(), newS = single.slice();
    for (let i = 0; i < nbits; i++) {
      if (!((i >> k) & 1)) continue;
      const j = ((i >> k) << k) - 1;
      // (the borrow as NOT a_i, a free gate: the optimizer then turns the cell into MAJ(NOT a_i, b_i, G_lo))
      const gHi = single[i] ? (sub ? nl.gate(OP.NOT, a[i]) : a[i]) : G[i];
      newG[i] = nl.gate(OP.MUX, P[i], G[j], gHi);
      newP[i] = i < (1 << (k + 1)) ? null : nl.gate(OP.AND, P[i], P[j]);
      newS[i] = false;
    }
    G = newG; P = newP; single = newS;
  }
  for (let i = 1; i < nbits; i++) out.push(nl.gate(oOp, pbit[i], G[i - 1]));
  return { out, top: G[nbits - 1] };
};
const prefixNetwork = (nbits, sub) => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits), idx = [...Array(nbits).keys()];
  const { out, top } = prefixCells(nl, idx.map(i => a + i), idx.map(i => b + i), sub);
  return { nl, a, b, out, top };
};
Tfhe.prefixAdderNetlist = nbits => { const { nl, a, b, out, top } = prefixNetwork(nbits, false); return { nl, a, b, sum: [...out, top] }; };
Tfhe.prefixSubtractorNetlist = nbits => { const { nl, a, b, out, top } = prefixNetwork(nbits, true); return { nl, a, b, diff: out, borrow: top }; };
// unsigned a < b alone, ripple form: lt_0 = ANDNY(a_0, b_0); lt_i = MUX(a_i XNOR b_i, lt_{i-1}, b_i); 1 + 3 (nbits - 1) bootstraps
Tfhe.lessThanNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  let lt = nl.gate(OP.ANDNY, a, b);
  for (let i = 1; i < nbits; i++) lt = nl.gate(OP.MUX, nl.gate(OP.XNOR, a + i, b + i), lt, b + i);
  return { nl, a, b, lt };
};
// unsigned a < b in logarithmic depth: tree over (LT, EQ) of bit ranges, LT = MUX(EQ_hi, LT_lo, LT_hi); 29 bootstraps on 4
// levels for 8 bits (lessThanNetlist: 22 on 8)
Tfhe.lessThanTreeNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  const build = (lo, hi, needLt, needEq) => {
    if (hi - lo === 1) {
      const eq = needEq ? nl.gate(OP.XNOR, a + lo, b + lo) : null;
      const lt = needLt ? nl.gate(OP.ANDNY, a + lo, b + lo) : null;
      return [lt, eq];
    }
    const mid = lo + ((hi - lo + 1) >> 1), upSingle = hi - mid === 1;
    const [ltLo, eqLo] = build(lo, mid, true, needEq);
    const [ltHi, eqHi] = build(mid, hi, !upSingle, true);
    // a single bit as the upper operand: where a_i != b_i NOT a_i equals b_i; NOT is free, and written this way the optimizer
    // turns the cell into MAJ(NOT a_i, b_i, LT_lo): one bootstrap
    const lt = needLt ? nl.gate(OP.MUX, eqHi, ltLo, upSingle ? nl.gate(OP.NOT, a + mid) : ltHi) : null;
    const eq = needEq ? nl.gate(OP.AND, eqHi, eqLo) : null;
    return [lt, eq];
  };
  const lt = build(0, nbits, true, false)[0];
  return { nl, a, b, lt };
};
// the form of lowest estimated cost for this many instances (B.netlistCost: below a quarter of the resident set a level
// costs the same whatever its width, so depth decides for small batches and bootstraps for wide ones)
// outKeys: the builder's output wires (lists or single wires) -- the candidates are priced AFTER B.netlistOptimize, the way
// runNetlist runs them (the row-by-row multiplier shrinks from 320 to 176 bootstraps, the column form to 230, the prefix
// adder from 48 to 40, the tree comparator from 29 to 24)
const outputsOf = (r, outKeys) => outKeys.flatMap(k => r[k]);
const cheapest = (builders, nbits, instances, outKeys) => {
  let best = null, bestCost = 0;
  for (const build of builders) {
    const r = build(nbits);
    let gates = r.nl.packed();
    gates = B.netlistOptimize(gates, Int32Array.from(outputsOf(r, outKeys))) || gates;
    const cost = B.netlistCost(gates, instances);
    if (best === null || (cost >= 0 && cost < bestCost)) { best = r; bestCost = cost; }
  }
  return best;
};
Tfhe.adderNetlistFor = (nbits, instances) => cheapest([Tfhe.majAdderNetlist, Tfhe.prefixAdderNetlist], nbits, instances, ['sum']);
Tfhe.lessThanNetlistFor = (nbits, instances) => cheapest([Tfhe.majLessThanNetlist, Tfhe.lessThanTreeNetlist], nbits, instances, ['lt']);
Tfhe.multiplierNetlistFor = (nbits, instances) => cheapest([Tfhe.multiplierNetlist, Tfhe.wallaceMultiplierNetlist], nbits, instances, ['prod']);
Tfhe.subtractorNetlistFor = (nbits, instances) => cheapest([Tfhe.majSubtractorNetlist, Tfhe.prefixSubtractorNetlist], nbits, instances, ['diff', 'borrow']);
// (min, max) behind a comparator: min_i = MUX(lt, a_i, b_i) and max_i = MUX(lt, b_i, a_i) -- or, xor3Select,
// max_i = XOR3(a_i, b_i, min_i) (min_i XOR max_i = a_i XOR b_i): ONE bootstrap instead of the MUX's two, one level later
Tfhe.minMaxNetlistOn = ({ nl, a, b, lt }, nbits, xor3Select) => {
  const min = [], max = [];
  for (let i = 0; i < nbits; i++) {
    min.push(nl.gate(OP.MUX, lt, a + i, b + i));
    max.push(xor3Select ? nl.gate(OP.XOR3, a + i, b + i, min[i]) : nl.gate(OP.MUX, lt, b + i, a + i));
  }
  return { nl, a, b, lt, min, max };
};
// every comparator form with both ways of selecting the maximum, the cheapest for this many instances: tree comparator + two
// MUXes per bit for small batches (8 bits: 56 bootstraps on 5 levels after the optimizer), MAJ chain + MUX + XOR3 for wide
// ones (32 on 10; with two MUXes 40 on 9)
Tfhe.minMaxNetlistFor = (nbits, instances) => cheapest(
  [Tfhe.majLessThanNetlist, Tfhe.lessThanTreeNetlist].flatMap(lt => [false, true].map(x3 => n => Tfhe.minMaxNetlistOn(lt(n), n, x3))),
  nbits, instances, ['min', 'max']);
// a * b in logarithmic depth: partial products in columns by weight, Dadda column compression by full adders (XOR3 + MAJ: the
// extension gates, one level) and half adders until no column holds more than two wires, then ONE parallel-prefix addition of
// the two remaining rows; 8 bits: 244 bootstraps on 11 levels against the row-by-row form's 320 on 40
Tfhe.wallaceMultiplierNetlist = nbits => {
  const nl = new Netlist(), a = nl.wire(nbits), b = nl.wire(nbits);
  if (nbits === 1) return { nl, a, b, prod: [nl.gate(OP.AND, a, b), nl.gate(OP.CONST0, -1)] };
  const ncol = 2 * nbits, byLevel = (p, q) => p[0] - q[0] || p[1] - q[1];
  let cols = [...Array(ncol)].map(() => []);
  for (let r = 0; r < nbits; r++) for (let j = 0; j < nbits; j++) cols[r + j].push([1, nl.gate(OP.AND, a + j, b + r)]);
  // Dadda's schedule: column heights come down through 9, 6, 4, 3, 2; in a layer every column is reduced to the target with
  // as few adders as possible (a full adder = XOR3 + MAJ removes two wires, a half adder = XOR + AND one), counting the
  // carries the column below sends up in the same layer
  const targets = [2];
  while (Math.floor(targets[targets.length - 1] * 3 / 2) < nbits) targets.push(Math.floor(targets[targets.length - 1] * 3 / 2));
  for (const target of targets.slice().reverse()) {
    const next = [...Array(ncol)].map(() => []);
    for (let c = 0; c < ncol; c++) {
      const col = cols[c].slice().sort(byLevel);
      let i = 0;
      while (col.length - i + next[c].length > target) {
        let sm, cy;
        if (col.length - i + next[c].length >= target + 2 && col.length - i >= 3) {
          const [x, y, z] = [col[i], col[i + 1], col[i + 2]], lv = Math.max(x[0], y[0], z[0]) + 1;
          sm = [lv, nl.gate(OP.XOR3, x[1], y[1], z[1])];
          cy = [lv, nl.gate(OP.MAJ, x[1], y[1], z[1])];
          i += 3;
        } else {
          const [x, y] = [col[i], col[i + 1]], lv = Math.max(x[0], y[0]) + 1;
          sm = [lv, nl.gate(OP.XOR, x[1], y[1])];
          cy = [lv, nl.gate(OP.AND, x[1], y[1])];
          i += 2;
        }
        next[c].push(sm);
        next[c + 1].push(cy);
      }
      next[c].push(...col.slice(i));
    }
    cols = next;
  }
  const prod = [];
  let c0 = 0;
  for (; c0 < ncol && cols[c0].length <= 1; c0++) prod.push(cols[c0].length ? cols[c0][0][1] : null);   // already final
  let hi = ncol - 1;
  while (hi >= c0 && cols[hi].length === 0) hi--;
  if (hi >= c0) {
    let zero = null;
    const xs = [], ys = [];
    for (let c = c0; c <= hi; c++) {
      const col = cols[c].slice().sort(byLevel);
      xs.push(col[0][1]);
      if (col.length > 1) ys.push(col[1][1]);
      else { if (zero === null) zero = nl.gate(OP.CONST0, -1); ys.push(zero); }
    }
    const { out, top } = prefixCells(nl, xs, ys, false);
    prod.push(...out, top);
  }
  prod.length = ncol;
  for (let k = 0; k < ncol; k++) if (prod[k] === null || prod[k] === undefined) prod[k] = nl.gate(OP.CONST0, -1);
  return { nl, a, b, prod };
};
// run a netlist over `instances` instances: inputs = {firstWire: Buffer [k][instances][n+1]}; returns the wire Buffer
Tfhe.runNetlist = (nl, inputs, instances, outputs) => {
  const w = B.sampleInts() * 4, plane = instances * w;
  const wires = Buffer.alloc(nl.nWires * plane);
  for (const [first, buf] of Object.entries(inputs)) buf.copy(wires, Number(first) * plane);
  let gates = nl.packed();
  if (outputs) gates = B.netlistOptimize(gates, Int32Array.from(outputs)) || gates;  // NOT folding, MUX fusion
  return B.circuitRun(gates, wires, nl.nWires, instances) ? wires : null;
};
const planes = (wires, first, count, instances) => {
  const plane = instances * B.sampleInts() * 4;
  return wires.slice(first * plane, (first + count) * plane);
};
// base64 ciphertext strings <-> raw samples (wire format of export_lweSample_toStream: a[n] | b | f64 variance)
const strToSample = s => Buffer.from(s, 'base64').slice(0, B.sampleInts() * 4);
const sampleToStr = buf => Buffer.concat([buf, Buffer.alloc(8)]).toString('base64');
const stack = arr => Buffer.concat(arr.map(strToSample));
const unstack = (buf, k) => { const w = B.sampleInts() * 4; return [...Array(k).keys()].map(i => sampleToStr(buf.slice(i * w, (i + 1) * w))); };

// string-API circuits: arrays of base64 bit ciphertexts in, arrays out -- ONE backend call per circuit
Tfhe.addBits = (A, Bs) => {   // one instance: the log-depth form
  const { nl, a, b, sum } = Tfhe.adderNetlistFor(A.length, 1);
  const wires = Tfhe.runNetlist(nl, { [a]: stack(A), [b]: stack(Bs) }, 1, sum);
  return wires && sum.map(wi => unstack(planes(wires, wi, 1, 1), 1)[0]);
};
Tfhe.lessThanBits = (A, Bs) => {   // one instance: the log-depth form
  const { nl, a, b, lt } = Tfhe.lessThanNetlistFor(A.length, 1);
  const wires = Tfhe.runNetlist(nl, { [a]: stack(A), [b]: stack(Bs) }, 1, [lt]);
  return wires && unstack(planes(wires, lt, 1, 1), 1)[0];
};
Tfhe.minMaxBits = (A, Bs) => {
  const { nl, a, b, min, max } = Tfhe.minMaxNetlistFor(A.length, 1);
  const wires = Tfhe.runNetlist(nl, { [a]: stack(A), [b]: stack(Bs) }, 1, [...min, ...max]);
  const pick = ws => ws.map(wi => unstack(planes(wires, wi, 1, 1), 1)[0]);
  return wires && { min: pick(min), max: pick(max) };
};
// raw-buffer circuits over many instances: operands are Buffers [nbits][instances][n+1]
Tfhe.addBitsBatch = (Abuf, Bbuf, nbits, instances) => {   // the form is picked by the instance count
  const { nl, a, b, sum } = Tfhe.adderNetlistFor(nbits, instances);
  const wires = Tfhe.runNetlist(nl, { [a]: Abuf, [b]: Bbuf }, instances, sum);
  return wires && Buffer.concat(sum.map(wi => planes(wires, wi, 1, instances)));   // [nbits + 1][instances][n+1]
};
Tfhe.lessThanBitsBatch = (Abuf, Bbuf, nbits, instances) => {   // -> [instances][n+1]: 1 iff A < B (unsigned); the form by instance count
  const { nl, a, b, lt } = Tfhe.lessThanNetlistFor(nbits, instances);
  const wires = Tfhe.runNetlist(nl, { [a]: Abuf, [b]: Bbuf }, instances, [lt]);
  return wires && planes(wires, lt, 1, instances);
};
Tfhe.subtractBitsBatch = (Abuf, Bbuf, nbits, instances) => {   // -> [nbits + 1][instances][n+1]: difference bits, then the borrow
  const { nl, a, b, diff, borrow } = Tfhe.subtractorNetlistFor(nbits, instances);
  const wires = Tfhe.runNetlist(nl, { [a]: Abuf, [b]: Bbuf }, instances, [...diff, borrow]);
  return wires && Buffer.concat([...diff, borrow].map(wi => planes(wires, wi, 1, instances)));
};
Tfhe.multiplyBitsBatch = (Abuf, Bbuf, nbits, instances) => {   // -> [2 nbits][instances][n+1]
  const { nl, a, b, prod } = Tfhe.multiplierNetlistFor(nbits, instances);
  const wires = Tfhe.runNetlist(nl, { [a]: Abuf, [b]: Bbuf }, instances, prod);   // through netlistOptimize (carry rewrite, constants)
  return wires && Buffer.concat(prod.map(wi => planes(wires, wi, 1, instances)));
};
Tfhe.minMaxBitsBatch = (Abuf, Bbuf, nbits, instances) => {     // -> { min, max: [nbits][instances][n+1], lt: [instances][n+1] }
  const { nl, a, b, lt, min, max } = Tfhe.minMaxNetlistFor(nbits, instances);
  const wires = Tfhe.runNetlist(nl, { [a]: Abuf, [b]: Bbuf }, instances, [...min, ...max, lt]);
  const pick = ws => Buffer.concat(ws.map(wi => planes(wires, wi, 1, instances)));
  return wires && { min: pick(min), max: pick(max), lt: planes(wires, lt, 1, instances) };
};
Tfhe.equalBits = (X, Y) => {  // X, Y: Buffers of int32 samples [nbits][n+1] (one instance)
  const nbits = X.length / (B.sampleInts() * 4);
  const { nl, x, y, out } = Tfhe.equalNetlist(nbits);
  const wires = Tfhe.runNetlist(nl, { [x]: X, [y]: Y }, 1);
  return wires && planes(wires, out, 1, 1);
};
// string helpers of the gate path: a string travels as 8 bit-ciphertexts per byte, LSB first (raw Buffer of samples)
Tfhe.encryptStringBits = str => B.encryptBits(Buffer.from([...Buffer.from(str)].flatMap(c => [...Array(8).keys()].map(k => (c >> k) & 1))));
Tfhe.equalStrings = (X, Y) => Tfhe.equalBits(X, Y);   // one ciphertext: 1 iff the two encrypted strings are equal
// ---- deferred gates: the reference's call style (one ciphertext operation per call, tfhe.lua:4-53), ONE backend call ----
// A gate call on the string API costs a whole blind rotation's n sequential steps (1.8 ms) however little it computes.  A
// deferred circuit records the same calls on wire handles and evaluates them together: run() sends the recorded netlist
// through B.netlistOptimize and ONE B.circuitRun, where every LEVEL costs those 1.8 ms.
//   const c = Tfhe.newCircuit(), x = c.input(ctX), y = c.input(ctY);
//   const [s, k] = c.run([c.xor(x, y), c.and(x, y)]);      // base64 ciphertext strings, one backend call
Tfhe.newCircuit = () => {
  const nl = new Netlist(), inputs = {}, c = { instances: null };
  const addInput = (buf, instances) => {
    if (c.instances !== null && c.instances !== instances) return null;   // every input has the same instance count
    c.instances = instances;
    const w = nl.wire(1);
    inputs[w] = buf;
    return w;
  };
  c.input = ct => addInput(strToSample(ct), 1);                          // one base64 ciphertext string
  c.inputSamples = buf => addInput(buf, buf.length / (B.sampleInts() * 4)); // raw samples [instances][n+1]
  c.constant = bit => nl.gate(bit ? OP.CONST1 : OP.CONST0, -1);
  c.nand = (a, b) => nl.gate(OP.NAND, a, b);
  c.and = (a, b) => nl.gate(OP.AND, a, b);
  c.or = (a, b) => nl.gate(OP.OR, a, b);
  c.nor = (a, b) => nl.gate(OP.NOR, a, b);
  c.xor = (a, b) => nl.gate(OP.XOR, a, b);
  c.xnor = (a, b) => nl.gate(OP.XNOR, a, b);
  c.not = a => nl.gate(OP.NOT, a);
  c.mux = (a, b, d) => nl.gate(OP.MUX, a, b, d);
  c.maj = (a, b, d) => nl.gate(OP.MAJ, a, b, d);
  c.xor3 = (a, b, d) => nl.gate(OP.XOR3, a, b, d);
  c.gateCount = () => nl.gates.length;
  c.netlist = () => nl;
  // outs: array of handles -> array of base64 strings (instances == 1) or of raw sample Buffers [instances][n+1]
  c.run = outs => {
    const instances = c.instances === null ? 1 : c.instances;
    const wires = Tfhe.runNetlist(nl, inputs, instances, outs);
    if (!wires) return null;
    return outs.map(w => { const buf = planes(wires, w, 1, instances); return instances === 1 ? sampleToStr(buf) : buf; });
  };
  return c;
};
module.exports = Tfhe;
