"""SURVEY.md row f2, server side, on CPU: the cloud ("public") key leaves a client process as an EOCCK1 blob and a second
process installs it as a cloud-key-ONLY global context.  The reference aliases the cloud key set as its public key
(ao-tfhe/eoc-tfhe-run.cpp:232-234), checks only that key in its homomorphic ops (:427-470) and leaves generatePublicKey
undefined (ao-tfhe/eoc-tfhe-run.h:10, ao-tfhe/eoc-tfhe-bindings.c:51-57); here the pair is real.

No GPU is needed: the linear ops and constantBit run on the CPU as in the reference; the gate calls must FAIL loudly on
a box without a device (no CPU fallback).  The GPU side of the same story is tests/test_gpu_cloud_server.py.
"""
import base64
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TKN = "eyJhbGciOiJSUzI1NiJ9.eyJvd25lciI6InRlc3QifQ"
JWKS = "ewogICJrZXlzIjogW10KfQ"


def run_child(body, cwd=None, env=None):
    import json
    code = textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        import eoc_tfhe_amd as eoc
        from eoc_tfhe_amd import Tfhe
        tkn, jwks = %r, %r
        out = {}
    """ % (ROOT, TKN, JWKS)) + textwrap.dedent(body) + "\nprint('RESULT' + json.dumps(out))\n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=cwd, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):]), r.stdout, r.stderr


@pytest.fixture(scope="module")
def small_secret_blob(built_lib):
    """a seeded Set-A-shaped key with n = 16 (keygen in milliseconds, 1.1 MB cloud key), as base64(EOCSK1)"""
    import eoc_tfhe_amd as eoc
    p = eoc.default_params(0)
    p.n = 16
    return base64.b64encode(eoc.SecretKey(p, 77, with_cloud_key=False).export_bytes()).decode()


def test_client_exports_server_imports(small_secret_blob, tmp_path):
    ck_path = str(tmp_path / "cloud.key")
    client, _, _ = run_child("""
        out['none_yet'] = Tfhe.exportCloudKey()
        out['mode0'] = Tfhe.keyMode()
        out['imp'] = Tfhe.importSecretKey(%r)
        out['mode1'] = Tfhe.keyMode()
        ck = Tfhe.exportCloudKey()
        out['ck_head'] = ck[:8]
        out['ck_len'] = len(ck)
        out['pub_same'] = Tfhe.generatePublicKey() == ck          # generatePublicKey IS the cloud key export
        out['file'] = Tfhe.exportCloudKeyToFile(%r)
        out['bad_file'] = Tfhe.exportCloudKeyToFile('/nonexistent-dir/x.key')
        out['raw_same'] = eoc.global_cloud_key_export().tobytes() == open(%r, 'rb').read()
        out['a'], out['b'] = Tfhe.encryptInteger(15, ''), Tfhe.encryptInteger(27, '')
        out['second_import'] = Tfhe.importCloudKey(ck)            # one key per process
    """ % (small_secret_blob, ck_path, ck_path))
    assert client["none_yet"] is None and client["mode0"] == 0 and client["imp"] == 0 and client["mode1"] == 1
    assert base64.b64decode(client["ck_head"])[:6] == b"EOCCK1"
    blob = open(ck_path, "rb").read()
    assert blob[:6] == b"EOCCK1" and client["ck_len"] == (len(blob) + 2) // 3 * 4
    assert client["pub_same"] and client["raw_same"] and client["file"] == 0 and client["bad_file"] == -1
    assert client["second_import"] == -1
    assert b"EOCSK" not in blob                                    # nothing of the secret blob travels

    server, _, stderr = run_child("""
        out['bad'] = Tfhe.importCloudKey('AAAA')
        out['secret_refused'] = Tfhe.importCloudKey(%r)           # a SECRET key blob is not a cloud key
        out['missing_file'] = Tfhe.importCloudKeyFromFile('/nonexistent.key')
        out['imp'] = Tfhe.importCloudKeyFromFile(%r)
        out['mode'] = Tfhe.keyMode()
        out['again'] = Tfhe.importCloudKeyFromFile(%r)
        out['gen_refused'] = Tfhe.generateSecretKey(tkn, jwks)
        out['impsk_refused'] = Tfhe.importSecretKey(%r)
        # everything that needs the secret key answers as if there were none
        out['enc_bit'] = Tfhe.encryptBit(1, '')
        out['dec_bit'] = Tfhe.decryptBit('AAAA', '')
        out['enc_int'] = Tfhe.encryptInteger(5, '')
        out['dec_int'] = Tfhe.decryptInteger(%r, '', tkn, jwks)
        out['enc_str'] = Tfhe.encryptASCIIString('hi', 2, '')
        out['exp_sk'] = Tfhe.exportSecretKey()
        try:
            eoc.global_encrypt_bits([0, 1]); out['enc_bits'] = 'ok'
        except eoc.EocError: out['enc_bits'] = 'refused'
        # what needs only the public key works
        out['n'] = eoc.global_params().n
        out['const'] = Tfhe.constantBit(1)
        out['sum'] = Tfhe.addCiphertexts(%r, %r, '')
        out['ck_round_trip'] = Tfhe.exportCloudKey() is not None and eoc.global_cloud_key_export().tobytes() == open(%r, 'rb').read()
        out['devices'] = eoc.lib().eoc_device_count()
        out['gate'] = Tfhe.nand(out['const'], out['const'], '')   # needs the GPU engine: NULL on a box without one
        Tfhe.resetGateKey()
        out['mode_after_reset'] = Tfhe.keyMode()
        out['imp_after_reset'] = Tfhe.importCloudKeyFromFile(%r)
    """ % (small_secret_blob, ck_path, ck_path, small_secret_blob, client["a"], client["a"], client["b"], ck_path, ck_path))
    assert server["bad"] == -1 and server["secret_refused"] == -1 and server["missing_file"] == -1
    assert "SECRET key blob" in stderr
    assert server["imp"] == 0 and server["mode"] == 2 and server["again"] == -1
    assert server["gen_refused"] is None and server["impsk_refused"] == -1
    for k in ("enc_bit", "enc_int", "enc_str", "exp_sk"):
        assert server[k] is None, k
    assert server["dec_bit"] == -1 and server["dec_int"] == -1 and server["enc_bits"] == "refused"
    assert "Secret key not initialized. Generate the secret key first." in stderr
    assert server["n"] == 16 and server["const"] is not None and server["sum"] is not None and server["ck_round_trip"]
    if server["devices"] == 0:                                     # no CPU fallback for gates: fail loudly
        assert server["gate"] is None
    assert server["mode_after_reset"] == 0 and server["imp_after_reset"] == 0

    # the client decrypts what the secret-free server computed
    back, _, _ = run_child("""
        Tfhe.importSecretKey(%r)
        out['sum'] = Tfhe.decryptInteger(%r, '', tkn, jwks)
        out['const'] = Tfhe.decryptBit(%r, '')
    """ % (small_secret_blob, server["sum"], server["const"]))
    assert back == {"sum": 42, "const": 1}


def test_cloud_blob_argument_checks(built_lib):
    import ctypes as C
    import numpy as np
    import eoc_tfhe_amd as eoc
    L = eoc.lib()
    assert L.eoc_global_key_mode() == 0
    assert L.eoc_global_cloud_key_export(None, 0) == 0
    junk = np.zeros(64, np.uint8)
    assert L.eoc_global_import_cloud_key_blob(junk.ctypes.data, junk.size) == -1     # EOC_ERR_ARG
    assert L.eoc_global_import_cloud_key_blob(None, 0) == -1
    assert L.eoc_global_key_mode() == 0
    p = eoc.default_params(0)
    p.n = 8
    sk = eoc.SecretKey(p, 5)
    ck = sk.export_cloud_key()
    assert L.eoc_global_import_cloud_key_blob(ck.ctypes.data, ck.size - 4) == -1     # truncated
    assert L.eoc_global_import_cloud_key_blob(ck.ctypes.data, ck.size) == 0
    try:
        assert L.eoc_global_key_mode() == 2
        q = eoc.global_params()
        assert q.n == 8 and q.l == p.l
        assert np.array_equal(eoc.global_cloud_key_export(), ck)
    finally:
        L.resetGateKey()
    assert L.eoc_global_key_mode() == 0
