// The reference's six tests (tests/tfhe.test.js:51-186) against the native library, in the reference's own
// host language.  Plain asserts: Node 12 has no node:test.  Usage: node integration/node/test_cpu.js
'use strict';
const assert = require('assert');
const tfhe = require('./tfhe.js');
const tkn = 'eyJhbGciOiJSUzI1NiJ9.eyJvd25lciI6InRlc3QifQ';
const jwks = 'ewogICJrZXlzIjogW10KfQ';

// "TFHE info function returns library information" (:56-76)
tfhe.info(); tfhe.testJWT();
assert.strictEqual(tfhe.encryptInteger(1, ''), null, 'no key yet -> nil');
// "TFHE key generation and integer encryption/decryption" (:78-104)
const key = tfhe.generateSecretKey(tkn, jwks);
assert.ok(key && key.length > 1000);
assert.strictEqual(tfhe.generateSecretKey(tkn, jwks), null, 'already generated -> nil');
let enc = tfhe.encryptInteger(42, '');
assert.strictEqual(String(tfhe.decryptInteger(enc, '', tkn, jwks)), '42');
// "TFHE string encryption/decryption" (:106-128)
const text = 'Hello TFHE!';
const es = tfhe.encryptASCIIString(text, text.length, '');
assert.strictEqual(tfhe.decryptASCIIString(es, text.length, '', tkn, jwks), 'Hello TFHE!');
// "TFHE homomorphic addition" (:130-157)
let a = tfhe.encryptInteger(15, ''), b = tfhe.encryptInteger(27, '');
assert.strictEqual(String(tfhe.decryptInteger(tfhe.addCiphertexts(a, b, ''), '', tkn, jwks)), '42');
// "TFHE homomorphic subtraction" (:159-186): the facade forwards to add, the reference's test pins "58"
a = tfhe.encryptInteger(50, ''); b = tfhe.encryptInteger(8, '');
assert.strictEqual(String(tfhe.decryptInteger(tfhe.subtractCiphertexts(a, b, ''), '', tkn, jwks)), '58');
assert.strictEqual(tfhe.decryptInteger(tfhe.backend.subtractCiphertexts(a, b, ''), '', tkn, jwks), 42);
// error conventions
assert.strictEqual(tfhe.decryptInteger(enc, '', 'no-dot', jwks), -1);
assert.strictEqual(tfhe.addCiphertexts('AAAA', enc, ''), null);
// raw-buffer client-side calls work without a GPU; gates do not
const bits = Buffer.from([0, 1, 1, 0, 1]);
const cts = tfhe.backend.encryptBits(bits);
assert.strictEqual(cts.length, 5 * tfhe.backend.sampleInts() * 4);
assert.deepStrictEqual([...tfhe.backend.decryptBits(cts)], [0, 1, 1, 0, 1]);
if (tfhe.backend.deviceCount() === 0) {
  assert.strictEqual(tfhe.backend.gateBatch(0, cts, cts, null), null, 'no GPU -> gates fail, no CPU fallback');
  assert.strictEqual(tfhe.nand(tfhe.encryptBit(1, ''), tfhe.encryptBit(1, ''), ''), null);
}
console.log('node cpu tests OK');
