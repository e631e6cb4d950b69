"""SURVEY.md row f2 on the GPU: cloud-key-only evaluation on the drop-in surface, as two (three) OS processes.

  client  (CPU only)  secret key, encrypts, exports the cloud key            -> wire/ (cloud.key, ciphertexts)
  server  (the GPU)   importCloudKeyFromFile: a cloud-key-ONLY global context; mixed NAND/XOR/MUX batch, the 8-bit
                      adder as one circuit call, a few string-API gates; cannot encrypt, decrypt or export a secret
  client  again       decrypts what came back
  this process        the oracle: the server's bytes must equal the oracle's bytes

The reference's model is exactly this split -- every homomorphic op takes a base64PublicKey and checks only
globalPublicKey (ao-tfhe/eoc-tfhe-run.cpp:427-470), which aliases the cloud key set of the secret key (:232-234);
generatePublicKey is declared (ao-tfhe/eoc-tfhe-run.h:10) and its binding is an empty stub
(ao-tfhe/eoc-tfhe-bindings.c:51-57), and the bindings never forward the key arguments (:63-110).
"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEY_SEED = 1            # Set A, key seed 1: the oracle regenerates the same key material from the seed (PRNG v1)
N_GATES, N_ADD = 96, 8


def child(body, cwd, timeout=900):
    code = textwrap.dedent("""
        import json, sys, os
        import numpy as np
        sys.path.insert(0, %r)
        import eoc_tfhe_amd as eoc
        from eoc_tfhe_amd import Tfhe
        out = {}
    """ % ROOT) + textwrap.dedent(body) + "\nprint('RESULT' + json.dumps(out))\n"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=cwd)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):]), r.stdout, r.stderr


def test_client_server_split_bit_exact(built_lib, tmp_path):
    from eoc_tfhe_amd import circuits, Gate
    wire, vault = tmp_path / "wire", tmp_path / "client_vault"
    wire.mkdir()
    vault.mkdir()
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(8)
    rng = np.random.default_rng(11)
    ops = rng.choice([ol.OPS["NAND"], ol.OPS["XOR"], ol.OPS["MUX"]], N_GATES).astype(np.uint8)
    bits = rng.integers(0, 2, (3, N_GATES)).astype(np.uint8)
    A, B = rng.integers(0, 256, N_ADD), rng.integers(0, 256, N_ADD)
    np.save(vault / "bits.npy", bits)
    np.save(vault / "ab.npy", np.stack([A, B]))

    # ---- client: CPU only (no GPU call is made: importSecretKey + encryption + export) -------------------------------
    client, _, _ = child("""
        import base64
        p = eoc.default_params(0)
        blob = eoc.SecretKey(p, %d, with_cloud_key=False).export_bytes()
        open(%r, 'wb').write(blob)                                  # the secret key never leaves the vault
        assert Tfhe.importSecretKey(base64.b64encode(blob).decode()) == 0
        out['mode'] = Tfhe.keyMode()
        assert Tfhe.exportCloudKeyToFile('cloud.key') == 0
        open('cloud.b64', 'w').write(Tfhe.exportCloudKey())
        bits = np.load(%r)
        for k in range(3):
            np.save('in%%d.npy' %% k, eoc.global_encrypt_bits(bits[k]))
        A, B = np.load(%r)
        for name, v in (('a', A), ('b', B)):
            planes = np.stack([eoc.global_encrypt_bits((v >> i) & 1) for i in range(8)])   # [8][S][n+1], LSB first
            np.save('adder_%%s.npy' %% name, planes)
        json.dump([Tfhe.encryptBit(0), Tfhe.encryptBit(1)], open('str_in.json', 'w'))
        out['engines'] = eoc.gpu_engine_count()                     # the client never brought a GPU engine up
    """ % (KEY_SEED, str(vault / "secret.key"), str(vault / "bits.npy"), str(vault / "ab.npy")), cwd=str(wire))
    assert client == {"mode": 1, "engines": 0}
    np.save(wire / "ops.npy", ops)

    # nothing secret is on the wire: no EOCSK magic, raw or base64, in anything the server will be able to read
    for f in os.listdir(wire):
        data = open(wire / f, "rb").read()
        assert b"EOCSK" not in data and b"RU9DU0s" not in data, f
    assert open(vault / "secret.key", "rb").read()[:6] == b"EOCSK1"

    # ---- server: the GPU box's process; its whole world is wire/ -----------------------------------------------------
    gate_list = [[g.op, g.in0, g.in1, g.in2, g.out] for g in gates]
    server, _, server_err = child("""
        assert Tfhe.importCloudKeyFromFile('cloud.key') == 0
        out['mode'] = Tfhe.keyMode()
        # the secret-key half of the surface answers NULL / -1 / EOC_ERR_NO_KEY
        out['enc'], out['exp'] = Tfhe.encryptBit(1), Tfhe.exportSecretKey()
        s0, s1 = json.load(open('str_in.json'))
        out['dec'] = Tfhe.decryptBit(s1)
        try:
            eoc.global_decrypt_bits(np.load('in0.npy')); out['dec_bits'] = 'ok'
        except eoc.EocError: out['dec_bits'] = 'refused'
        # the public-key half runs on the GPU
        ops = np.load('ops.npy')
        res = eoc.global_gate_batch(0, np.load('in0.npy'), np.load('in1.npy'), np.load('in2.npy'), ops=ops)
        np.save('out.npy', res)
        gates = [eoc.Gate(*g) for g in %r]
        n_wires, S = %d, %d
        wires = np.zeros((n_wires, S, eoc.global_params().n + 1), np.int32)
        wires[%d:%d] = np.load('adder_a.npy'); wires[%d:%d] = np.load('adder_b.npy')
        np.save('adder_wires.npy', eoc.global_circuit_run(gates, wires, S))
        strs = dict(nand=Tfhe.nand(s1, s1), xor=Tfhe.xor(s0, s1), mux=Tfhe.mux(s1, s0, s1), not_=Tfhe.not_(s0),
                    one=Tfhe.constantBit(1))
        json.dump(strs, open('str_out.json', 'w'))
        st = eoc.stats()
        out['bootstraps'] = int(st['bootstraps'])
        out['engines'] = eoc.gpu_engine_count()
        # the base64 string form of the import at full size (110 MB of text), after a reset: same results
        Tfhe.resetGateKey()
        out['mode_reset'] = Tfhe.keyMode()
        assert Tfhe.importCloudKey(open('cloud.b64').read()) == 0
        out['mode2'] = Tfhe.keyMode()
        again = eoc.global_gate_batch(0, np.load('in0.npy')[:8], np.load('in1.npy')[:8], np.load('in2.npy')[:8], ops=ops[:8])
        out['string_form_same'] = bool(np.array_equal(again, res[:8]))
        Tfhe.resetGateKey()
    """ % (gate_list, n_wires, N_ADD, aw[0], aw[0] + 8, bw[0], bw[0] + 8), cwd=str(wire))
    assert server["mode"] == 2 and server["mode2"] == 2 and server["mode_reset"] == 0
    assert server["enc"] is None and server["exp"] is None and server["dec"] == -1 and server["dec_bits"] == "refused"
    assert "Secret key not initialized. Generate the secret key first." in server_err
    n_mux = int((ops == ol.OPS["MUX"]).sum())
    assert server["bootstraps"] == N_GATES + n_mux + 37 * N_ADD + 4 and server["engines"] == 1
    assert server["string_form_same"]

    # ---- the oracle (this process): the server's bytes are the oracle's bytes ------------------------------------------
    orc = ol.Oracle(0, KEY_SEED)
    in0, in1, in2 = (np.load(wire / ("in%d.npy" % k)) for k in range(3))
    got = np.load(wire / "out.npy")
    assert np.array_equal(got, orc.gate_batch(0, in0, in1, in2, ops=ops))
    wires_in = np.zeros((n_wires, N_ADD, orc.n + 1), np.int32)
    wires_in[aw[0]: aw[0] + 8] = np.load(wire / "adder_a.npy")
    wires_in[bw[0]: bw[0] + 8] = np.load(wire / "adder_b.npy")
    want = wires_in.copy()
    for g in gates:
        want[g.out] = orc.gate_batch(g.op, want[g.in0], None if g.in1 < 0 else want[g.in1], None if g.in2 < 0 else want[g.in2])
    got_wires = np.load(wire / "adder_wires.npy")
    for g in gates:
        assert np.array_equal(got_wires[g.out], want[g.out]), f"adder wire {g.out}"

    # ---- client again: decrypts -----------------------------------------------------------------------------------------
    back, _, _ = child("""
        import base64
        assert Tfhe.importSecretKey(base64.b64encode(open(%r, 'rb').read()).decode()) == 0
        out['gates'] = eoc.global_decrypt_bits(np.load('out.npy')).tolist()
        w = np.load('adder_wires.npy')
        out['sums'] = [int(sum(int(eoc.global_decrypt_bits(w[%d + i][s:s + 1])[0]) << i for i in range(9))) for s in range(%d)]
        strs = json.load(open('str_out.json'))
        out['strs'] = {k: Tfhe.decryptBit(v) for k, v in strs.items()}
    """ % (str(vault / "secret.key"), sw[0], N_ADD), cwd=str(wire))
    b0, b1, b2 = bits.astype(np.int64)
    expect = np.where(ops == ol.OPS["NAND"], 1 - (b0 & b1), np.where(ops == ol.OPS["XOR"], b0 ^ b1, np.where(b0 == 1, b1, b2)))
    assert back["gates"] == expect.tolist()
    assert back["sums"] == (A + B).tolist()
    assert back["strs"] == {"nand": 0, "xor": 1, "mux": 0, "not_": 1, "one": 1}


def test_cloud_only_server_on_several_engines(built_lib, tmp_path):
    """the same secret-free context in front of THREE engines (EOC_TFHE_DEVICES=0,0,0: the one-GPU rehearsal of a
    multi-GPU server): the imported cloud key is built once on engine 0 and replicated, the batch is cut into blocks,
    every engine evaluates its block with its own replica -- bytes equal the oracle's"""
    import eoc_tfhe_amd as eoc
    p = eoc.default_params(1)                      # Set B shape, small LWE dimension: l = 3, Bgbit = 7
    p.n = 48
    sk = eoc.SecretKey(p, 21)
    orc = ol.Oracle(1, 21, n_override=48)
    rng = np.random.default_rng(3)
    total = 41
    ops = rng.choice(np.array([ol.OPS["NAND"], ol.OPS["XOR"], ol.OPS["MUX"], ol.OPS["ORNY"]], np.uint8), total)
    bits = rng.integers(0, 2, (3, total)).astype(np.uint8)
    c = [sk.encrypt_bits(bits[k], 70 + k, 0) for k in range(3)]
    sk.export_cloud_key().tofile(str(tmp_path / "cloud.key"))
    for k in range(3):
        np.save(tmp_path / ("in%d.npy" % k), c[k])
    np.save(tmp_path / "ops.npy", ops)
    env = dict(os.environ, EOC_TFHE_DEVICES="0,0,0")
    code = textwrap.dedent("""
        import sys, json
        import numpy as np
        sys.path.insert(0, %r)
        import eoc_tfhe_amd as eoc
        from eoc_tfhe_amd import Tfhe
        assert Tfhe.importCloudKeyFromFile('cloud.key') == 0 and Tfhe.keyMode() == 2
        out = eoc.global_gate_batch(0, np.load('in0.npy'), np.load('in1.npy'), np.load('in2.npy'), ops=np.load('ops.npy'))
        np.save('out.npy', out)
        st = eoc.stats_multi()
        print('RESULT' + json.dumps(dict(engines=len(st['engines']), per=[e['bootstraps'] for e in st['engines']],
                                         method=st['key_broadcast_method'], enc=Tfhe.encryptBit(1))))
        Tfhe.resetGateKey()
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][len("RESULT"):])
    assert res["engines"] == 3 and all(x > 0 for x in res["per"]) and res["method"] == "peer-copy" and res["enc"] is None
    got = np.load(tmp_path / "out.npy")
    assert np.array_equal(got, orc.gate_batch(0, c[0], c[1], c[2], ops=ops))
    b0, b1, b2 = bits.astype(np.int64)
    want = np.where(ops == ol.OPS["NAND"], 1 - (b0 & b1), np.where(ops == ol.OPS["XOR"], b0 ^ b1,
                    np.where(ops == ol.OPS["MUX"], np.where(b0 == 1, b1, b2), (1 - b0) | b1)))
    assert np.array_equal(sk.decrypt_bits(got), want)
