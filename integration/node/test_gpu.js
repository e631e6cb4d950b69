// Boolean gates through the Node binding on a GPU box.  Usage: node integration/node/test_gpu.js
'use strict';
const assert = require('assert');
const tfhe = require('./tfhe.js');
assert.ok(tfhe.backend.deviceCount() >= 1, 'needs a GPU');
// two engines behind the one global key (both on device 0 here: the one-GPU rehearsal of a multi-GPU host)
assert.strictEqual(tfhe.backend.setDevices(Int32Array.from([0, 0])), 0);
assert.ok(tfhe.generateGateKey(80, 3));
assert.strictEqual(tfhe.backend.engineCount(), 2);
const e = [tfhe.encryptBit(0, ''), tfhe.encryptBit(1, '')];
for (const x of [0, 1]) for (const y of [0, 1]) {
  assert.strictEqual(tfhe.decryptBit(tfhe.nand(e[x], e[y], ''), ''), 1 - (x & y));
  assert.strictEqual(tfhe.decryptBit(tfhe.xor(e[x], e[y], ''), ''), x ^ y);
  assert.strictEqual(tfhe.decryptBit(tfhe.mux(e[x], e[y], e[1 - y], ''), ''), x ? y : 1 - y);
  assert.strictEqual(tfhe.decryptBit(tfhe.maj(e[x], e[y], e[1 - y], ''), ''), x);          // MAJ(x, y, NOT y) = x
  assert.strictEqual(tfhe.decryptBit(tfhe.xor3(e[x], e[y], e[1], ''), ''), x ^ y ^ 1);
}
assert.strictEqual(tfhe.decryptBit(tfhe.not(e[1], ''), ''), 0);
assert.strictEqual(tfhe.decryptBit(tfhe.and(tfhe.constantBit(1), e[1], ''), ''), 1);
assert.strictEqual(tfhe.decryptBit(tfhe.constantBit(0), ''), 0);
// 4-bit adder with the string API: 9 + 5 = 14
const enc = (v, n) => [...Array(n).keys()].map(i => tfhe.encryptBit((v >> i) & 1, ''));
const S = tfhe.addBits(enc(9, 4), enc(5, 4), '');
assert.strictEqual(S.reduce((acc, c, i) => acc | (tfhe.decryptBit(c, '') << i), 0), 14);
// 3-bit comparison / min / max with the string API
const val = cs => cs.reduce((acc, c, i) => acc | (tfhe.decryptBit(c, '') << i), 0);
for (const [x, y] of [[5, 3], [2, 6], [4, 4]]) {
  assert.strictEqual(tfhe.decryptBit(tfhe.lessThanBits(enc(x, 3), enc(y, 3), ''), ''), x < y ? 1 : 0);
}
const mm = tfhe.minMaxBits(enc(5, 3), enc(3, 3), '');
assert.strictEqual(val(mm.min), 3);
assert.strictEqual(val(mm.max), 5);
// deferred gates: the reference's call style (one operation per call) recorded on handles, ONE backend call per run():
// a 3-bit adder written gate by gate the textbook way; then a batch of raw samples through maj / xor3 / not / constant
{
  const c = tfhe.newCircuit(), xs = enc(5, 3).map(c.input), ys = enc(6, 3).map(c.input), outs = [];
  let carry = null;
  for (let i = 0; i < 3; i++) {
    const p = c.xor(xs[i], ys[i]), g = c.and(xs[i], ys[i]);
    if (carry === null) { outs.push(p); carry = g; } else { outs.push(c.xor(p, carry)); carry = c.or(g, c.and(p, carry)); }
  }
  assert.strictEqual(c.gateCount(), 12);
  assert.strictEqual(val(c.run([...outs, carry])), 11);
  const c2 = tfhe.newCircuit(), v = Buffer.from([1, 0, 1, 1, 0]), nv = Buffer.from([0, 1, 0, 0, 1]);
  const h = [v, nv, v].map(b => c2.inputSamples(tfhe.backend.encryptBits(b)));
  assert.strictEqual(c2.inputSamples(tfhe.backend.encryptBits(Buffer.from([1, 1]))), null);   // another instance count: refused
  const [m, q] = c2.run([c2.maj(h[0], h[1], h[2]), c2.xor3(h[0], c2.not(h[1]), c2.constant(1))]);
  assert.deepStrictEqual([...tfhe.backend.decryptBits(m)], [...v]);
  assert.deepStrictEqual([...tfhe.backend.decryptBits(q)], [...v].map(b => b ^ b ^ 1));
}
// batched: 2048 NANDs and a 16-byte string equality through raw buffers
const N = 2048, bits0 = Buffer.alloc(N), bits1 = Buffer.alloc(N);
for (let i = 0; i < N; i++) { bits0[i] = (i * 7 + 3) & 1; bits1[i] = (i >> 3) & 1; }
const out = tfhe.backend.decryptBits(tfhe.backend.gateBatch(0, tfhe.backend.encryptBits(bits0), tfhe.backend.encryptBits(bits1), null));
for (let i = 0; i < N; i++) assert.strictEqual(out[i], 1 - (bits0[i] & bits1[i]));
const toBits = s => Buffer.from([...Buffer.from(s)].flatMap(c => [...Array(8).keys()].map(k => (c >> k) & 1)));
const eq = (s, t) => tfhe.backend.decryptBits(tfhe.equalBits(tfhe.backend.encryptBits(toBits(s)), tfhe.backend.encryptBits(toBits(t))))[0];
assert.strictEqual(eq('sixteen byte str', 'sixteen byte str'), 1);
assert.strictEqual(eq('sixteen byte str', 'sixteen byte stR'), 0);
assert.strictEqual(tfhe.backend.decryptBits(tfhe.equalStrings(tfhe.encryptStringBits('abc'), tfhe.encryptStringBits('abc')))[0], 1);
assert.strictEqual(tfhe.backend.decryptBits(tfhe.equalStrings(tfhe.encryptStringBits('abc'), tfhe.encryptStringBits('abd')))[0], 0);
// 8-bit adder over 4096 input pairs through ONE circuitRun call (BASELINE configs[2] through the Node host)
{
  const S = 4096, nb = 8, B = tfhe.backend;
  const A = [...Array(S).keys()].map(i => (i * 37 + 11) & 255), Bv = [...Array(S).keys()].map(i => (i * 101 + 7) & 255);
  const planeBits = (vals, k) => Buffer.from(vals.map(v => (v >> k) & 1));
  const enc = vals => Buffer.concat([...Array(nb).keys()].map(k => B.encryptBits(planeBits(vals, k))));
  const t0 = Date.now();
  const sums = tfhe.addBitsBatch(enc(A), enc(Bv), nb, S);
  const dt = (Date.now() - t0) / 1000;
  assert.ok(sums, 'circuitRun failed');
  const w = B.sampleInts() * 4, plane = S * w;
  const tot = new Array(S).fill(0);
  for (let k = 0; k <= nb; k++) {
    const bits = B.decryptBits(sums.slice(k * plane, (k + 1) * plane));
    for (let i = 0; i < S; i++) tot[i] |= bits[i] << k;
  }
  for (let i = 0; i < S; i++) assert.strictEqual(tot[i], A[i] + Bv[i]);
  // comparison and min / max over the same 4096 pairs: the wide-batch forms (MAJ chain; the maximum as XOR3(a, b, min))
  const ltBits = B.decryptBits(tfhe.lessThanBitsBatch(enc(A), enc(Bv), nb, S));
  for (let i = 0; i < S; i++) assert.strictEqual(ltBits[i], A[i] < Bv[i] ? 1 : 0);
  const mm = tfhe.minMaxBitsBatch(enc(A), enc(Bv), nb, S);
  const lo = new Array(S).fill(0), hi = new Array(S).fill(0);
  for (let k = 0; k < nb; k++) {
    const bl = B.decryptBits(mm.min.slice(k * plane, (k + 1) * plane)), bh = B.decryptBits(mm.max.slice(k * plane, (k + 1) * plane));
    for (let i = 0; i < S; i++) { lo[i] |= bl[i] << k; hi[i] |= bh[i] << k; }
  }
  for (let i = 0; i < S; i++) { assert.strictEqual(lo[i], Math.min(A[i], Bv[i])); assert.strictEqual(hi[i], Math.max(A[i], Bv[i])); }
  const boots = B.circuitBootstraps(tfhe.adderNetlist(nb).nl.packed()) * S;
  console.log(`node adder8 x ${S}: ${boots} bootstraps in ${dt.toFixed(2)} s (${Math.round(boots / dt)} /s incl. host copies)`);
  // netlistOptimize: NOT(x) feeding an AND becomes one ANDNY
  const opt = B.netlistOptimize(Int32Array.from([11, 0, -1, -1, 2, 1, 2, 1, -1, 3]), Int32Array.from([3]));
  assert.deepStrictEqual(Array.from(opt), [6, 0, 1, -1, 3]);
  // operand checks: a short second operand is refused instead of being read out of bounds
  const one = B.encryptBits(Buffer.from([1, 0]));
  assert.strictEqual(B.gateBatch(0, one, one.slice(0, w), null), null);
  assert.strictEqual(B.decryptBits(B.gateBatch(14, null, null, null, null, 3)).join(''), '111');   // bootsCONSTANT
}
// word-level circuits added in round 3: 4-bit subtraction and multiplication over 64 instances, one circuitRun each
{
  const B = tfhe.backend, S = 64, nb = 4, w = B.sampleInts() * 4, plane = S * w;
  const A = [...Array(S).keys()].map(i => (i * 7 + 3) & 15), Bv = [...Array(S).keys()].map(i => (i * 5 + 9) & 15);
  const enc = vals => Buffer.concat([...Array(nb).keys()].map(k => B.encryptBits(Buffer.from(vals.map(v => (v >> k) & 1)))));
  const dec = (buf, nplanes) => { const t = new Array(S).fill(0); for (let k = 0; k < nplanes; k++) { const bits = B.decryptBits(buf.slice(k * plane, (k + 1) * plane)); for (let i = 0; i < S; i++) t[i] |= bits[i] << k; } return t; };
  const d = dec(tfhe.subtractBitsBatch(enc(A), enc(Bv), nb, S), nb + 1);
  const p = dec(tfhe.multiplyBitsBatch(enc(A), enc(Bv), nb, S), 2 * nb);
  for (let i = 0; i < S; i++) {
    assert.strictEqual(d[i] & 15, (A[i] - Bv[i]) & 15);
    assert.strictEqual(d[i] >> 4, A[i] < Bv[i] ? 1 : 0);
    assert.strictEqual(p[i], A[i] * Bv[i]);
  }
}
// asynchronous batches on pinned Buffers: four batches kept two deep in flight equal the synchronous call's bytes
{
  const B = tfhe.backend, w = B.sampleInts() * 4, n = 600;
  const mk = seed => { const bits = Buffer.alloc(n); for (let i = 0; i < n; i++) bits[i] = (i * seed + (i >> 2)) & 1; return B.encryptBits(bits); };
  const jobs = [3, 5, 7, 11].map(seed => {
    const a = mk(seed), b = mk(seed + 1), pa = B.hostAlloc(n * w), pb = B.hostAlloc(n * w), po = B.hostAlloc(n * w);
    assert.ok(pa && pb && po, 'hostAlloc failed');
    a.copy(pa); b.copy(pb);
    return { a, b, pa, pb, po, op: seed % 2 ? tfhe.OP.XOR : tfhe.OP.NAND };
  });
  const tickets = [];
  jobs.forEach((j, k) => {
    if (k >= 2) assert.strictEqual(B.gateBatchWait(tickets[k - 2]), 0);
    const t = B.gateBatchSubmit(j.op, j.pa, j.pb, null, null, j.po);
    assert.ok(t !== null, 'gateBatchSubmit failed');
    tickets.push(t);
  });
  tickets.forEach(t => assert.strictEqual(B.gateBatchWait(t), 0));
  jobs.forEach((j, k) => assert.ok(Buffer.from(j.po).equals(B.gateBatch(j.op, j.a, j.b, null)), `async batch ${k}`));
  assert.strictEqual(B.gateBatchSubmit(0, jobs[0].a, jobs[0].b, null, null, jobs[0].po), null);   // pageable operands are refused
}
assert.strictEqual(tfhe.backend.engineCount(), 2);
tfhe.backend.resetGateKey();
console.log('node gpu tests OK');
