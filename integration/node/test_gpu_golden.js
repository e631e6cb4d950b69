// The Node binding's raw batch call against the COMMITTED golden bytes (tests/golden/golden_arrays.npz, exported to
// flat little-endian int32 files by tests/test_node_binding.py): key seed 1 on Set A is the fixture's key, so
// gateBatch(NAND / MUX) on the fixture's inputs must return the fixture's outputs bit for bit -- no decryption involved.
// Usage: EOC_GOLDEN_DIR=<dir with A_c0.bin ...> node integration/node/test_gpu_golden.js
'use strict';
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const tfhe = require('./tfhe.js');
const dir = process.env.EOC_GOLDEN_DIR;
assert.ok(dir, 'EOC_GOLDEN_DIR is not set');
assert.ok(tfhe.backend.deviceCount() >= 1, 'needs a GPU');
const rd = name => fs.readFileSync(path.join(dir, name + '.bin'));
assert.ok(tfhe.generateGateKey(80, 1));               // lambda <= 80 selects Set A (n = 500); seed 1 = the fixture's key
const B = tfhe.backend;
assert.strictEqual(B.sampleInts(), 501);
const c0 = rd('A_c0'), c1 = rd('A_c1'), c2 = rd('A_c2');
assert.strictEqual(c0.length, 4 * 501 * 4);
const nand = B.gateBatch(tfhe.OP.NAND, c0, c1, null);
assert.ok(nand && nand.equals(rd('A_NAND_out')), 'gateBatch(NAND) differs from the golden bytes');
const mux = B.gateBatch(tfhe.OP.MUX, c0, c1, c2);
assert.ok(mux && mux.equals(rd('A_MUX_out')), 'gateBatch(MUX) differs from the golden bytes');
assert.strictEqual(B.decryptBits(nand).join(''), '1110');
B.resetGateKey();
console.log('node gpu golden OK');
