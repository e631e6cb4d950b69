// Client / server split of the key material through the Node binding, on CPU (the GPU half is test_gpu_cloud.js):
//   node test_cpu_cloud.js client <dir>   secret key in this process; writes <dir>/cloud.key and two ciphertexts
//   node test_cpu_cloud.js server <dir>   installs ONLY the cloud key; adds the ciphertexts; cannot encrypt or decrypt
//   node test_cpu_cloud.js verify <dir>   the client again: decrypts what the server produced
// Reference anchors: globalPublicKey = the cloud key set (ao-tfhe/eoc-tfhe-run.cpp:232-234), the ops check only that
// key (:427-470), generatePublicKey is declared and empty (eoc-tfhe-run.h:10, eoc-tfhe-bindings.c:51-57).
'use strict';
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const tfhe = require('./tfhe.js');
const [mode, dir] = process.argv.slice(2);
const tkn = 'eyJhbGciOiJSUzI1NiJ9.eyJvd25lciI6InRlc3QifQ', jwks = 'ewogICJrZXlzIjogW10KfQ';
const f = name => path.join(dir, name);
if (mode === 'client') {
  assert.strictEqual(tfhe.keyMode(), 0);
  assert.strictEqual(tfhe.exportCloudKey(), null, 'no key yet -> nil');
  const sk = tfhe.generateSecretKey(tkn, jwks);
  assert.ok(sk);
  assert.strictEqual(tfhe.keyMode(), 1);
  assert.strictEqual(tfhe.exportCloudKeyToFile(f('cloud.key')), 0);
  assert.strictEqual(fs.readFileSync(f('cloud.key')).slice(0, 6).toString(), 'EOCCK1');
  fs.writeFileSync(f('client_secret.b64'), tfhe.exportSecretKey());   // stays with the client
  fs.writeFileSync(f('a.ct'), tfhe.encryptInteger(15, ''));
  fs.writeFileSync(f('b.ct'), tfhe.encryptInteger(27, ''));
  fs.writeFileSync(f('x.bin'), tfhe.backend.encryptBits(Buffer.from([0, 0, 1, 1, 1, 0])));   // raw samples for gateBatch
  fs.writeFileSync(f('y.bin'), tfhe.backend.encryptBits(Buffer.from([0, 1, 0, 1, 1, 1])));
} else if (mode === 'server') {
  assert.strictEqual(tfhe.importCloudKey('AAAA'), -1);
  assert.strictEqual(tfhe.importCloudKeyFromFile(f('cloud.key')), 0);
  assert.strictEqual(tfhe.keyMode(), 2);
  assert.strictEqual(tfhe.importCloudKeyFromFile(f('cloud.key')), -1, 'one key per process');
  assert.strictEqual(tfhe.encryptBit(1, ''), null);
  assert.strictEqual(tfhe.encryptInteger(1, ''), null);
  assert.strictEqual(tfhe.exportSecretKey(), null);
  assert.strictEqual(tfhe.decryptInteger(fs.readFileSync(f('a.ct'), 'utf8'), '', tkn, jwks), -1);
  assert.strictEqual(tfhe.backend.encryptBits(Buffer.from([1])), null);
  assert.ok(tfhe.backend.sampleInts() > 0);
  const sum = tfhe.addCiphertexts(fs.readFileSync(f('a.ct'), 'utf8'), fs.readFileSync(f('b.ct'), 'utf8'), '');
  assert.ok(sum);
  fs.writeFileSync(f('sum.ct'), sum);
  const one = tfhe.constantBit(1);
  fs.writeFileSync(f('one.ct'), one);
  if (tfhe.backend.deviceCount() === 0) {                      // no CPU fallback for the gates
    assert.strictEqual(tfhe.nand(one, one, ''), null);
  } else {                                                     // the secret-free server evaluates gates on the GPU
    const nand = tfhe.nand(one, one, '');
    assert.ok(nand);
    fs.writeFileSync(f('nand.ct'), nand);
    const out = tfhe.backend.gateBatch(tfhe.OP.XOR, fs.readFileSync(f('x.bin')), fs.readFileSync(f('y.bin')), null);
    assert.ok(out);
    fs.writeFileSync(f('xor.bin'), out);
    tfhe.resetGateKey();
    assert.strictEqual(tfhe.keyMode(), 0);
  }
} else if (mode === 'verify') {
  assert.strictEqual(tfhe.importSecretKey(fs.readFileSync(f('client_secret.b64'), 'utf8')), 0);
  assert.strictEqual(tfhe.decryptInteger(fs.readFileSync(f('sum.ct'), 'utf8'), '', tkn, jwks), 42);
  assert.strictEqual(tfhe.decryptBit(fs.readFileSync(f('one.ct'), 'utf8'), ''), 1);
  if (fs.existsSync(f('xor.bin'))) {                           // the GPU leg ran on the server
    assert.strictEqual(tfhe.decryptBit(fs.readFileSync(f('nand.ct'), 'utf8'), ''), 0);
    assert.deepStrictEqual([...tfhe.backend.decryptBits(fs.readFileSync(f('xor.bin')))], [0, 1, 1, 0, 0, 1]);
    console.log('node cloud gpu leg verified');
  }
} else {
  throw new Error('usage: test_cpu_cloud.js client|server|verify <dir>');
}
console.log('node cloud ' + mode + ' OK');
