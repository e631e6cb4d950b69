// Boolean gates through the Node binding on a GPU box.  Usage: node integration/node/test_gpu.js
'use strict';
const assert = require('assert');
const tfhe = require('./tfhe.js');
assert.ok(tfhe.backend.deviceCount() >= 1, 'needs a GPU');
assert.ok(tfhe.generateGateKey(80, 3));
const e = [tfhe.encryptBit(0, ''), tfhe.encryptBit(1, '')];
for (const x of [0, 1]) for (const y of [0, 1]) {
  assert.strictEqual(tfhe.decryptBit(tfhe.nand(e[x], e[y], ''), ''), 1 - (x & y));
  assert.strictEqual(tfhe.decryptBit(tfhe.xor(e[x], e[y], ''), ''), x ^ y);
  assert.strictEqual(tfhe.decryptBit(tfhe.mux(e[x], e[y], e[1 - y], ''), ''), x ? y : 1 - y);
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
tfhe.backend.resetGateKey();
console.log('node gpu tests OK');
