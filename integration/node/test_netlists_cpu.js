// The circuit layer of tfhe.js on plaintext (no GPU, no key): every netlist builder against integer arithmetic for all
// input pairs at 4 bits and random pairs at 8, and the forms picked by instance count (backend.netlistCost =
// eoc_netlist_cost, backend.netlistDepth = eoc_netlist_levels' bootstrap depth).
'use strict';
const assert = require('assert');
const tfhe = require('./tfhe.js');
const B = tfhe.backend, OP = tfhe.OP;

const sem = (op, a, b, c) => {
  switch (op) {
    case OP.NAND: return 1 - (a & b); case OP.AND: return a & b; case OP.OR: return a | b; case OP.NOR: return 1 - (a | b);
    case OP.XOR: return a ^ b; case OP.XNOR: return 1 - (a ^ b); case OP.ANDNY: return (1 - a) & b; case OP.ANDYN: return a & (1 - b);
    case OP.ORNY: return (1 - a) | b; case OP.ORYN: return a | (1 - b); case OP.MUX: return a ? b : c;
    case OP.NOT: return 1 - a; case OP.COPY: return a; case OP.CONST0: return 0;
    case OP.MAJ: return a + b + c >= 2 ? 1 : 0; case OP.XOR3: return a ^ b ^ c; default: return 1;
  }
};
const run = (nl, inputs) => {            // inputs: { firstWire: [bit, bit, ...] }
  const w = new Array(nl.nWires).fill(0);
  for (const [first, bits] of Object.entries(inputs)) bits.forEach((v, i) => { w[Number(first) + i] = v; });
  for (const [op, i0, i1, i2, out] of nl.gates) w[out] = sem(op, i0 >= 0 ? w[i0] : 0, i1 >= 0 ? w[i1] : 0, i2 >= 0 ? w[i2] : 0);
  return w;
};
const bitsOf = (v, n) => [...Array(n).keys()].map(i => (v >> i) & 1);
const valueOf = (w, wires) => wires.reduce((acc, wi, i) => acc + w[wi] * 2 ** i, 0);
const shape = nl => [B.circuitBootstraps(nl.packed()), B.netlistDepth(nl.packed())];
const WALLACE8 = [244, 11];

for (const nbits of [1, 2, 3, 4]) {
  for (let x = 0; x < 1 << nbits; x++) for (let y = 0; y < 1 << nbits; y++) {
    for (const build of [tfhe.adderNetlist, tfhe.muxAdderNetlist, tfhe.majAdderNetlist, tfhe.prefixAdderNetlist]) {
      const { nl, a, b, sum } = build(nbits);
      assert.strictEqual(valueOf(run(nl, { [a]: bitsOf(x, nbits), [b]: bitsOf(y, nbits) }), sum), x + y);
    }
    for (const build of [tfhe.multiplierNetlist, tfhe.wallaceMultiplierNetlist]) {
      const { nl, a, b, prod } = build(nbits);
      assert.strictEqual(valueOf(run(nl, { [a]: bitsOf(x, nbits), [b]: bitsOf(y, nbits) }), prod), x * y);
    }
    for (const build of [tfhe.subtractorNetlist, tfhe.majSubtractorNetlist, tfhe.prefixSubtractorNetlist]) {
      const { nl, a, b, diff, borrow } = build(nbits);
      const w = run(nl, { [a]: bitsOf(x, nbits), [b]: bitsOf(y, nbits) });
      assert.deepStrictEqual([valueOf(w, diff), w[borrow]], [(x - y + (1 << nbits)) % (1 << nbits), x < y ? 1 : 0]);
    }
    for (const build of [tfhe.lessThanNetlist, tfhe.majLessThanNetlist, tfhe.lessThanTreeNetlist, tfhe.minMaxNetlist]) {
      const { nl, a, b, lt } = build(nbits);
      assert.strictEqual(run(nl, { [a]: bitsOf(x, nbits), [b]: bitsOf(y, nbits) })[lt], x < y ? 1 : 0);
    }
  }
}
let seed = 12345;
const rnd = () => { seed = (seed * 1103515245 + 12345) & 0x7fffffff; return (seed >> 8) & 255; };
for (let t = 0; t < 300; t++) {
  const x = rnd(), y = rnd();
  for (const build of [tfhe.muxAdderNetlist, tfhe.majAdderNetlist, tfhe.prefixAdderNetlist]) {
    const { nl, a, b, sum } = build(8);
    assert.strictEqual(valueOf(run(nl, { [a]: bitsOf(x, 8), [b]: bitsOf(y, 8) }), sum), x + y);
  }
  const { nl, a, b, lt } = tfhe.lessThanTreeNetlist(8);
  assert.strictEqual(run(nl, { [a]: bitsOf(x, 8), [b]: bitsOf(y, 8) })[lt], x < y ? 1 : 0);
  const wm = tfhe.wallaceMultiplierNetlist(8);
  assert.strictEqual(valueOf(run(wm.nl, { [wm.a]: bitsOf(x, 8), [wm.b]: bitsOf(y, 8) }), wm.prod), x * y);
}
for (const inst of [1, 5000]) {
  for (let x = 0; x < 8; x++) for (let y = 0; y < 8; y++) {
    const { nl, a, b, lt, min, max } = tfhe.minMaxNetlistFor(3, inst);
    const w = run(nl, { [a]: bitsOf(x, 3), [b]: bitsOf(y, 3) });
    assert.deepStrictEqual([w[lt], valueOf(w, min), valueOf(w, max)], [x < y ? 1 : 0, Math.min(x, y), Math.max(x, y)]);
  }
}
assert.deepStrictEqual(shape(tfhe.minMaxNetlistFor(8, 1).nl), [29 + 32, 5]);
assert.deepStrictEqual(shape(tfhe.minMaxNetlistFor(8, 4096).nl), [8 + 16 + 8, 10]);   // MAJ chain, min by MUX, max = XOR3(a, b, min)
// bootstraps / dependent levels of the 8-bit forms (eoc_tfhe_amd/circuits.py states the same numbers)
assert.deepStrictEqual(shape(tfhe.adderNetlist(8).nl), [37, 15]);
assert.deepStrictEqual(shape(tfhe.adderNetlist(8, true).nl), [40, 17]);
assert.deepStrictEqual(shape(tfhe.muxAdderNetlist(8).nl), [30, 8]);
assert.deepStrictEqual(shape(tfhe.majAdderNetlist(8).nl), [16, 8]);
assert.deepStrictEqual(shape(tfhe.majSubtractorNetlist(8).nl), [16, 8]);
assert.deepStrictEqual(shape(tfhe.majLessThanNetlist(8).nl), [8, 8]);
assert.deepStrictEqual(shape(tfhe.prefixAdderNetlist(8).nl), [48, 5]);
assert.deepStrictEqual(shape(tfhe.lessThanNetlist(8).nl), [22, 8]);
assert.deepStrictEqual(shape(tfhe.multiplierNetlist(8).nl), [320, 40]);
assert.deepStrictEqual(shape(tfhe.wallaceMultiplierNetlist(8).nl), WALLACE8);
assert.deepStrictEqual(shape(tfhe.multiplierNetlistFor(8, 2).nl), WALLACE8);
assert.deepStrictEqual(shape(tfhe.multiplierNetlistFor(8, 4096).nl), [320, 40]);
const wopt = B.netlistOptimize(tfhe.wallaceMultiplierNetlist(8).nl.packed(), Int32Array.from(tfhe.wallaceMultiplierNetlist(8).prod));
assert.deepStrictEqual([B.circuitBootstraps(wopt), B.netlistDepth(wopt)], [230, 11]);
assert.deepStrictEqual(shape(tfhe.subtractorNetlist(8).nl), [30, 8]);
assert.deepStrictEqual(shape(tfhe.prefixSubtractorNetlist(8).nl), [48, 5]);
assert.deepStrictEqual(shape(tfhe.subtractorNetlistFor(8, 3).nl), [48, 5]);
assert.deepStrictEqual(shape(tfhe.subtractorNetlistFor(8, 4096).nl), [16, 8]);
assert.deepStrictEqual(shape(tfhe.lessThanTreeNetlist(8).nl), [29, 4]);
// the literal adder through the optimizer: a textbook full adder becomes XOR3 + MAJ (the extension gates), constants fold
const lit = tfhe.adderNetlist(8, true);
const opt = B.netlistOptimize(lit.nl.packed(), Int32Array.from(lit.sum));
assert.deepStrictEqual([B.circuitBootstraps(opt), B.netlistDepth(opt)], [16, 8]);
// picked by instance count: depth for small batches, bootstraps for wide ones
for (const inst of [1, 8, 64]) {
  assert.deepStrictEqual(shape(tfhe.adderNetlistFor(8, inst).nl), [48, 5]);
  assert.deepStrictEqual(shape(tfhe.lessThanNetlistFor(8, inst).nl), [29, 4]);
}
for (const inst of [1024, 4096]) {
  assert.deepStrictEqual(shape(tfhe.adderNetlistFor(8, inst).nl), [16, 8]);
  assert.deepStrictEqual(shape(tfhe.lessThanNetlistFor(8, inst).nl), [8, 8]);
}
assert.strictEqual(B.netlistCost(tfhe.prefixAdderNetlist(8).nl.packed(), 8), 5 * 18);
assert.strictEqual(B.netlistCost(tfhe.muxAdderNetlist(8).nl.packed(), 4096), 30 * 4 * 30);
assert.strictEqual(B.netlistCost(tfhe.majAdderNetlist(8).nl.packed(), 4096), 16 * 4 * 30);
assert.strictEqual(B.netlistCost(Int32Array.from([1, 2, 3]), 1), -1);
console.log('node netlist cpu tests OK');
