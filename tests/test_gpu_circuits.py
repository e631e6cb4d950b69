"""GPU tests of the circuit executor and the string API (BASELINE configs 3 and 5 at test sizes)."""
import numpy as np
import pytest

import oracle_lib as ol
from gpu_util import sync, to_dev, torch_cuda

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


def _oracle_run(orc, gates, wires):
    """gate-by-gate evaluation of the netlist by the oracle on [n_wires][S][n+1]"""
    w = wires.copy()
    for g in gates:
        i1 = None if g.in1 < 0 else w[g.in1]
        i2 = None if g.in2 < 0 else w[g.in2]
        i0 = w[g.in0] if g.in0 >= 0 else np.zeros_like(w[g.out])   # bootsCONSTANT has no input
        w[g.out] = orc.gate_batch(g.op, i0, i1, i2)
    return w


def _setup(eoc, pset, seed, n_override):
    p = eoc.default_params(pset)
    if n_override:
        p.n = n_override
    sk = eoc.SecretKey(p, seed)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    orc = ol.Oracle(pset, seed, n_override=n_override)
    return p, sk, eng, orc


def test_adder_4bit_small_params_bit_exact(eoc):
    from eoc_tfhe_amd import circuits
    p, sk, eng, orc = _setup(eoc, 0, 3, 20)
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(4)
    S = 5
    rng = np.random.default_rng(1)
    A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for i in range(4):
        wires[aw[i]] = sk.encrypt_bits((A >> i) & 1, 50 + i, 0)
        wires[bw[i]] = sk.encrypt_bits((B >> i) & 1, 70 + i, 0)
    d = to_dev(wires)
    eng.circuit_run_device(gates, d.data_ptr(), n_wires, S)
    sync()
    got = d.cpu().numpy()
    want = _oracle_run(orc, gates, wires)
    for w in sw:
        assert np.array_equal(got[w], want[w]), w
    total = sum(sk.decrypt_bits(got[sw[i]]).astype(np.int64) << i for i in range(5))
    assert np.array_equal(total, A + B)


def test_adder_8bit_set_a_decrypts(eoc):
    """BASELINE config 3 shape (8-bit ripple-carry) on full Set A, 64 pairs; sums checked by decryption,
    the first pair's sum bits bit-exact vs the oracle."""
    from eoc_tfhe_amd import circuits
    p, sk, eng, orc = _setup(eoc, 0, 1, None)
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(8)
    assert eoc.circuit_bootstraps(gates) == 37
    S = 64
    rng = np.random.default_rng(3)
    A, B = rng.integers(0, 256, S), rng.integers(0, 256, S)
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for i in range(8):
        wires[aw[i]] = sk.encrypt_bits((A >> i) & 1, 100 + i, 0)
        wires[bw[i]] = sk.encrypt_bits((B >> i) & 1, 200 + i, 0)
    d = to_dev(wires)
    eng.circuit_run_device(gates, d.data_ptr(), n_wires, S)
    sync()
    got = d.cpu().numpy()
    total = sum(sk.decrypt_bits(got[sw[i]]).astype(np.int64) << i for i in range(9))
    assert np.array_equal(total, A + B)
    want = _oracle_run(orc, gates, wires[:, :1].copy())
    for w in sw:
        assert np.array_equal(got[w][:1], want[w]), w
    st = eng.stats()
    assert st["bootstraps"] == 37 * S


def test_string_equal_small(eoc):
    """BASELINE config 5 shape at 2 bytes: 16 XOR + 15 OR + NOT, half of the pairs equal."""
    from eoc_tfhe_amd import circuits
    p, sk, eng, orc = _setup(eoc, 1, 4, 16)
    gates, n_wires, xw, yw, out = circuits.string_equal(2)
    assert eoc.circuit_bootstraps(gates) == 31
    S = 6
    rng = np.random.default_rng(5)
    X = rng.integers(32, 127, (S, 2))
    Y = X.copy()
    Y[::2, 1] ^= 1 << rng.integers(0, 7, len(Y[::2]))  # every other pair differs in one bit
    bitsx = np.unpackbits(X.astype(np.uint8), axis=1, bitorder="little")
    bitsy = np.unpackbits(Y.astype(np.uint8), axis=1, bitorder="little")
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for i in range(16):
        wires[xw[i]] = sk.encrypt_bits(bitsx[:, i], 300 + i, 0)
        wires[yw[i]] = sk.encrypt_bits(bitsy[:, i], 400 + i, 0)
    got = eoc_run(eoc, eng, gates, wires, n_wires, S)
    want = _oracle_run(orc, gates, wires)
    assert np.array_equal(got[out], want[out])
    assert np.array_equal(sk.decrypt_bits(got[out]), (X == Y).all(axis=1).astype(np.uint8))


def eoc_run(eoc, eng, gates, wires, n_wires, S):
    d = to_dev(wires)
    eng.circuit_run_device(gates, d.data_ptr(), n_wires, S)
    sync()
    return d.cpu().numpy()


def test_in_place_and_hazards(eoc):
    """netlists that reuse wires (WAR / WAW) are ordered correctly by the leveliser"""
    p, sk, eng, orc = _setup(eoc, 0, 6, 16)
    G, O = eoc.Gate, eoc.OPS
    gates = [G(O["XOR"], 0, 1, -1, 2), G(O["AND"], 2, 0, -1, 0),   # overwrites input 0 after it was read
             G(O["NOT"], 0, -1, -1, 0), G(O["OR"], 0, 2, -1, 2), G(O["MUX"], 2, 0, 1, 3)]
    S = 3
    wires = np.zeros((4, S, p.n + 1), np.int32)
    wires[0] = sk.encrypt_bits([0, 1, 1], 9, 0)
    wires[1] = sk.encrypt_bits([1, 1, 0], 10, 0)
    got = eoc_run(eoc, eng, gates, wires, 4, S)
    want = _oracle_run(orc, gates, wires)
    assert np.array_equal(got, want)


def test_string_api_and_host_batch_api(eoc):
    """reference-style string API: global key, base64 strings, truth tables through the GPU"""
    T = eoc.Tfhe
    tok = T.generateGateKey(80, 11)
    try:
        assert tok is not None
        assert T.generateGateKey(80, 11) is None          # "already generated" (eoc-tfhe-run.cpp:245-249)
        import base64
        c0, c1 = T.encryptBit(0), T.encryptBit(1)
        raw = base64.b64decode(c1)
        assert len(raw) == 4 * 501 + 8                   # a[n] | b | f64 variance
        assert T.decryptBit(c0) == 0 and T.decryptBit(c1) == 1
        assert T.decryptBit(T.nand(c1, c1)) == 0 and T.decryptBit(T.nand(c0, c1)) == 1
        assert T.decryptBit(T.xor(c0, c1)) == 1 and T.decryptBit(T.and_(c1, c1)) == 1
        assert T.decryptBit(T.not_(c1)) == 0
        k0, k1 = T.constantBit(0), T.constantBit(1)       # bootsCONSTANT: trivial samples, variance field 0
        assert T.decryptBit(k0) == 0 and T.decryptBit(k1) == 1
        assert base64.b64decode(k1)[: 4 * 500] == bytes(4 * 500) and base64.b64decode(k1)[-8:] == bytes(8)
        assert T.decryptBit(T.and_(k1, c1)) == 1 and T.decryptBit(T.or_(k0, c0)) == 0
        assert T.decryptBit(T.mux(c1, c0, c1)) == 0 and T.decryptBit(T.mux(c0, c0, c1)) == 1
        # the extension gates through the string API: majority and parity of three bits, one bootstrap each
        for k in range(8):
            x, y, z = (c1 if (k >> j) & 1 else c0 for j in range(3))
            assert T.decryptBit(T.maj(x, y, z)) == int(bin(k).count("1") >= 2) and T.decryptBit(T.xor3(x, y, z)) == bin(k).count("1") & 1
        assert T.maj(c1, "not base64!", c0) is None and T.xor3(c1, c0, "") is None
        assert T.nand("not base64!", c1) is None          # malformed input -> NULL
        # host-buffer batch API on the same global engine
        raw0 = np.frombuffer(base64.b64decode(c0)[: 4 * 501], np.int32)
        raw1 = np.frombuffer(raw[: 4 * 501], np.int32)
        out = eoc.gate_batch(eoc.OPS["OR"], np.stack([raw0, raw1]), np.stack([raw0, raw0]))
        assert out.shape == (2, 501)
        # word-level circuits of the facade (round 6): one instance takes the log-depth forms -- one backend call each
        for av, bv in ((200, 100), (13, 250), (77, 77)):
            A = [T.encryptBit((av >> i) & 1) for i in range(8)]
            B = [T.encryptBit((bv >> i) & 1) for i in range(8)]
            st0 = eoc.stats()
            ssum = T.addBits(A, B)
            st1 = eoc.stats()
            assert len(ssum) == 9 and sum(T.decryptBit(x) << i for i, x in enumerate(ssum)) == av + bv
            assert st1["bootstraps"] - st0["bootstraps"] == 40 and st1["batches"] - st0["batches"] == 5   # prefix adder, optimized
            assert T.decryptBit(T.lessThanBits(A, B)) == int(av < bv)
            assert eoc.stats()["bootstraps"] - st1["bootstraps"] == 24                                  # tree comparator, optimized
        # deferred gates (round 6): the reference's call style -- one operation per call -- recorded on handles and run by ONE
        # backend call: a 4-bit adder written gate by gate the textbook way (17 bootstrapped gate calls, 7 dependent levels as
        # written) costs 8 bootstraps on 4 levels after the rewrite, and equals the gate-by-gate string API's answer
        av, bv = 11, 6
        A = [T.encryptBit((av >> i) & 1) for i in range(4)]
        B = [T.encryptBit((bv >> i) & 1) for i in range(4)]
        c = eoc.Circuit()
        xs, ys = [c.input(x) for x in A], [c.input(y) for y in B]
        carry, outs = None, []
        for i in range(4):
            p_, g_ = c.xor(xs[i], ys[i]), c.and_(xs[i], ys[i])
            if carry is None:
                outs.append(p_)
                carry = g_
            else:
                outs.append(c.xor(p_, carry))
                carry = c.or_(g_, c.and_(p_, carry))
        st0 = eoc.stats()
        res = c.run(outs + [carry])
        st1 = eoc.stats()
        assert sum(T.decryptBit(x) << i for i, x in enumerate(res)) == av + bv
        assert st1["bootstraps"] - st0["bootstraps"] == 8 and st1["batches"] - st0["batches"] == 4, (st0, st1)
        c2 = eoc.Circuit()                                     # batches of raw samples through the same builder
        vals = np.array([1, 0, 1, 1, 0], np.uint8)
        h = [c2.input_samples(eoc.global_encrypt_bits(v)) for v in (vals, 1 - vals, vals)]
        m, q = c2.run([c2.maj(*h), c2.xor3(h[0], c2.not_(h[1]), c2.constant(1))])
        assert np.array_equal(eoc.global_decrypt_bits(m), vals) and np.array_equal(eoc.global_decrypt_bits(q), vals ^ vals ^ 1)
        planes = lambda vals: np.stack([eoc.global_encrypt_bits(((vals >> i) & 1).astype(np.uint8)) for i in range(8)])
        rng = np.random.default_rng(12)
        Av, Bv = rng.integers(0, 256, 1200), rng.integers(0, 256, 1200)
        st0 = eoc.stats()
        sums = T.addBitsBatch(planes(Av), planes(Bv))
        assert eoc.stats()["bootstraps"] - st0["bootstraps"] == 16 * 1200                               # XOR3 / MAJ adder
        assert np.array_equal(sum(eoc.global_decrypt_bits(sums[i]).astype(np.int64) << i for i in range(9)), Av + Bv)
        assert np.array_equal(eoc.global_decrypt_bits(T.lessThanBitsBatch(planes(Av), planes(Bv))), (Av < Bv).astype(np.uint8))
        word = lambda w: sum(eoc.global_decrypt_bits(w[i]).astype(np.int64) << i for i in range(w.shape[0]))
        diff = T.subtractBitsBatch(planes(Av[:9]), planes(Bv[:9]))                                   # 9 instances: prefix form
        assert np.array_equal(word(diff[:8]), (Av[:9] - Bv[:9]) % 256) and np.array_equal(word(diff[8:]), (Av[:9] < Bv[:9]))
        pl4 = lambda vals: np.stack([eoc.global_encrypt_bits(((vals >> i) & 1).astype(np.uint8)) for i in range(4)])
        assert np.array_equal(word(T.multiplyBitsBatch(pl4(Av[:5] & 15), pl4(Bv[:5] & 15))), (Av[:5] & 15) * (Bv[:5] & 15))
        mn, mx = T.minMaxBitsBatch(planes(Av[:9]), planes(Bv[:9]))
        assert np.array_equal(word(mn), np.minimum(Av[:9], Bv[:9])) and np.array_equal(word(mx), np.maximum(Av[:9], Bv[:9]))
        # wide batches take the other forms (MAJ chains; the maximum as XOR3(a, b, min)), and 16-bit words the same builders:
        # every word-level call at 700 pairs of 8 bits and at 3 and 300 pairs of 16 bits, with ties and the extremes
        mn, mx = T.minMaxBitsBatch(planes(Av[:700]), planes(Bv[:700]))
        assert np.array_equal(word(mn), np.minimum(Av[:700], Bv[:700])) and np.array_equal(word(mx), np.maximum(Av[:700], Bv[:700]))
        diff = T.subtractBitsBatch(planes(Av[:700]), planes(Bv[:700]))
        assert np.array_equal(word(diff[:8]), (Av[:700] - Bv[:700]) % 256) and np.array_equal(word(diff[8:]), (Av[:700] < Bv[:700]))
        pl16 = lambda vals: np.stack([eoc.global_encrypt_bits(((vals >> i) & 1).astype(np.uint8)) for i in range(16)])
        for S in (3, 300):
            A16, B16 = rng.integers(0, 1 << 16, S), rng.integers(0, 1 << 16, S)
            A16[0], B16[0] = 0xFFFF, 0xFFFF                                                        # a tie at the maximum
            A16[1], B16[1] = 0, 0xFFFF
            pa, pb = pl16(A16), pl16(B16)
            assert np.array_equal(word(T.addBitsBatch(pa, pb)), A16 + B16)
            d16 = T.subtractBitsBatch(pa, pb)
            assert np.array_equal(word(d16[:16]), (A16 - B16) % 65536) and np.array_equal(word(d16[16:]), (A16 < B16))
            assert np.array_equal(eoc.global_decrypt_bits(T.lessThanBitsBatch(pa, pb)), (A16 < B16).astype(np.uint8))
            mn, mx = T.minMaxBitsBatch(pa, pb)
            assert np.array_equal(word(mn), np.minimum(A16, B16)) and np.array_equal(word(mx), np.maximum(A16, B16))
        assert np.array_equal(word(T.multiplyBitsBatch(planes(Av[:3]), planes(Bv[:3]))), Av[:3] * Bv[:3])     # 8 x 8 -> 16 bits
    finally:
        T.resetGateKey()


def test_string_api_on_a_device_list(eoc):
    """eoc_gpu_set_devices: the reference-style global key in front of two engines (both on device 0 on this box);
    the string calls and the host-buffer batch call shard over them and give the single-engine bits"""
    import ctypes as C
    T, L = eoc.Tfhe, eoc.lib()
    devs = (C.c_int * 2)(0, 0)
    assert L.eoc_gpu_set_devices(devs, 2) == 0
    assert L.eoc_gpu_set_devices(devs, -1) < 0 and L.eoc_gpu_set_devices(None, 2) < 0
    try:
        assert T.generateGateKey(80, 12) is not None
        assert eoc.gpu_engine_count() == 2
        c0, c1 = T.encryptBit(0), T.encryptBit(1)
        assert T.decryptBit(T.nand(c1, c1)) == 0 and T.decryptBit(T.mux(c0, c0, c1)) == 1
        import base64
        rows = np.stack([np.frombuffer(base64.b64decode(c)[: 4 * 501], np.int32) for c in (c0, c1, c1, c0, c1)])
        two = eoc.gate_batch(eoc.OPS["XOR"], rows, rows[::-1].copy())
        T.resetGateKey()
        assert L.eoc_gpu_set_devices(None, 0) == 0                   # forget the list: one engine on device 0 again
        assert T.generateGateKey(80, 12) is not None
        assert eoc.gpu_engine_count() == 1
        one = eoc.gate_batch(eoc.OPS["XOR"], rows, rows[::-1].copy())
        assert np.array_equal(one, two)
    finally:
        L.eoc_gpu_set_devices(None, 0)
        T.resetGateKey()


def test_engine_from_cloud_key_blob(eoc):
    """f2: a server holding only the EOCCK1 blob evaluates gates bit-exactly"""
    p, sk, eng, orc = _setup(eoc, 1, 8, 14)
    srv = eoc.Engine.from_cloud_key_blob(sk.export_cloud_key())
    assert (srv.params.n, srv.params.l, srv.params.Bgbit) == (14, 3, 7)
    rng = np.random.default_rng(2)
    b0, b1 = rng.integers(0, 2, 9), rng.integers(0, 2, 9)
    c0, c1 = sk.encrypt_bits(b0, 5, 0), sk.encrypt_bits(b1, 6, 0)
    torch = torch_cuda()
    d0, d1 = to_dev(c0), to_dev(c1)
    out = torch.empty_like(d0)
    srv.gate_batch_device(eoc.OPS["XOR"], d0.data_ptr(), d1.data_ptr(), None, out.data_ptr(), 9)
    sync()
    got = out.cpu().numpy()
    assert np.array_equal(got, orc.gate_batch(ol.OPS["XOR"], c0, c1))
    assert np.array_equal(sk.decrypt_bits(got), b0 ^ b1)
    srv.close()


def test_legacy_key_then_gates(eoc):
    """f1 + hot path in one process: the reference's generateSecretKey (lambda 128 -> Set B) followed by
    Boolean gates on the same global key; the GPU engine comes up lazily at the first gate."""
    T = eoc.Tfhe
    tkn, jwks = "eyJhbGciOiJSUzI1NiJ9.eyJvd25lciI6InRlc3QifQ", "e30"
    try:
        key = T.generateSecretKey(tkn, jwks)
        assert key is not None
        assert T.decryptInteger(T.addCiphertexts(T.encryptInteger(15), T.encryptInteger(27)), "", tkn, jwks) == 42
        c0, c1 = T.encryptBit(0), T.encryptBit(1)
        assert T.decryptBit(T.nand(c1, c1)) == 0 and T.decryptBit(T.or_(c0, c1)) == 1
        assert T.decryptBit(T.mux(c1, c0, c1)) == 0
    finally:
        T.resetGateKey()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_netlists_bit_exact(eoc, seed):
    """random netlists with every opcode, wire re-use (WAR/WAW hazards) and fan-out: the levelised, batched GPU
    evaluation equals the oracle's gate-by-gate evaluation bit for bit"""
    p, sk, eng, orc = _setup(eoc, seed % 2, 20 + seed, 12 + seed)
    rng = np.random.default_rng(seed)
    n_in, n_wires, n_gates, S = 5, 11, 40, 3
    names = ["NAND", "AND", "OR", "NOR", "XOR", "XNOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "NOT", "COPY",
             "CONST0", "CONST1"]
    gates = []
    for _ in range(n_gates):
        op = eoc.OPS[names[rng.integers(0, len(names))]]
        a, b, c = (int(x) for x in rng.integers(0, n_wires, 3))
        out = int(rng.integers(n_in, n_wires))     # may overwrite a wire that earlier gates read or wrote
        if op in (eoc.OPS["CONST0"], eoc.OPS["CONST1"]):
            gates.append(eoc.Gate(op, -1, -1, -1, out))
        elif op in (eoc.OPS["NOT"], eoc.OPS["COPY"]):
            gates.append(eoc.Gate(op, a, -1, -1, out))
        elif op == eoc.OPS["MUX"]:
            gates.append(eoc.Gate(op, a, b, c, out))
        else:
            gates.append(eoc.Gate(op, a, b, -1, out))
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for w in range(n_wires):                        # every wire starts as a valid ciphertext
        wires[w] = sk.encrypt_bits(rng.integers(0, 2, S), 700 + w, 0)
    got = eoc_run(eoc, eng, gates, wires, n_wires, S)
    want = _oracle_run(orc, gates, wires)
    assert np.array_equal(got, want)
    # and the host-buffer entry point gives the same
    eoc.gpu_init(p)
    try:
        eoc.upload_cloud_key(sk)
        assert np.array_equal(eoc.circuit_run(gates, wires.copy(), S), want)
        st = eoc.stats()
        assert st["bootstraps"] == eoc.circuit_bootstraps(gates) * S and st["batches"] >= 1
    finally:
        eoc.gpu_shutdown()


def test_min_max_4bit_bit_exact_and_optimised_select(eoc):
    """8f3: comparator + word select.  min_max (XNOR/MUX chain) equals the oracle bit for bit; the same select
    written the long way (NOT, AND, AND, OR per bit) and rewritten by circuits.optimize (-> one MUX per bit)
    decrypts to the same words with a third fewer bootstraps."""
    from eoc_tfhe_amd import circuits
    p, sk, eng, orc = _setup(eoc, 0, 9, 18)
    nb, S = 4, 8
    rng = np.random.default_rng(8)
    A, B = rng.integers(0, 16, S), rng.integers(0, 16, S)
    A[0], B[0] = 7, 7                                         # a tie
    gates, n_wires, aw, bw, mn, mx = circuits.min_max(nb)
    wires = np.zeros((n_wires, S, p.n + 1), np.int32)
    for i in range(nb):
        wires[aw[i]] = sk.encrypt_bits((A >> i) & 1, 500 + i, 0)
        wires[bw[i]] = sk.encrypt_bits((B >> i) & 1, 520 + i, 0)
    got = eoc_run(eoc, eng, gates, wires, n_wires, S)
    want = _oracle_run(orc, gates, wires)
    for w in mn + mx:
        assert np.array_equal(got[w], want[w]), w
    val = lambda ws: sum(sk.decrypt_bits(got[w]).astype(np.int64) << i for i, w in enumerate(ws))
    assert np.array_equal(val(mn), np.minimum(A, B)) and np.array_equal(val(mx), np.maximum(A, B))

    # long-form select on the comparator's output, before and after rewriting
    lt_gates, nxt, a2, b2, lt = circuits.less_than(nb)
    ns = nxt; nxt += 1
    long_form, outs = list(lt_gates) + [eoc.Gate(eoc.OPS["NOT"], lt, -1, -1, ns)], []
    for i in range(nb):
        t0, t1, o = nxt, nxt + 1, nxt + 2
        nxt += 3
        long_form += [eoc.Gate(eoc.OPS["AND"], lt, a2[i], -1, t0), eoc.Gate(eoc.OPS["AND"], ns, b2[i], -1, t1),
                      eoc.Gate(eoc.OPS["OR"], t0, t1, -1, o)]
        outs.append(o)
    opt = circuits.optimize(long_form, outs)
    assert eoc.circuit_bootstraps(opt) == nb + 2 * nb      # the comparator chain as MAJ(NOT a, b, lt) per bit, one MUX per selected bit
    assert eoc.circuit_bootstraps(circuits.optimize(long_form, outs, extension_gates=False)) == eoc.circuit_bootstraps(lt_gates) + 2 * nb
    assert eoc.circuit_bootstraps(long_form) == eoc.circuit_bootstraps(lt_gates) + 3 * nb
    wires2 = np.zeros((nxt, S, p.n + 1), np.int32)
    wires2[: 2 * nb] = wires[: 2 * nb]
    g1 = eoc_run(eoc, eng, long_form, wires2, nxt, S)
    g2 = eoc_run(eoc, eng, opt, wires2, nxt, S)
    for g in (g1, g2):
        v = sum(sk.decrypt_bits(g[w]).astype(np.int64) << i for i, w in enumerate(outs))
        assert np.array_equal(v, np.minimum(A, B))


def test_reserved_engine_is_graph_capturable(eoc):
    """launch-path hygiene: after eoc_engine_reserve a gate batch and a netlist are captured into a hipGraph (no
    allocation, no synchronisation, descriptors from engine-owned pinned memory) and replayed on new operand values"""
    torch = torch_cuda()
    from eoc_tfhe_amd import circuits
    p = eoc.default_params(0)
    p.n = 40
    sk = eoc.SecretKey(p, 21)
    orc = ol.Oracle(0, 21, n_override=40)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    L = eoc.lib()
    S = 48
    gates, n_wires, aw, bw, sw = circuits.ripple_carry_adder(3)
    assert L.eoc_engine_reserve(eng.h, 4 * S, 256, 0) == 0
    rng = np.random.default_rng(2)

    def enc(bits, seed):
        return to_dev(sk.encrypt_bits(bits.astype(np.uint8), seed, 0))

    b0, b1 = rng.integers(0, 2, S), rng.integers(0, 2, S)
    d0, d1 = enc(b0, 1), enc(b1, 2)
    out = torch.empty_like(d0)
    wires = torch.zeros((n_wires, S, p.n + 1), dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):   # warm-up outside the capture (first launches load code objects)
        eng.gate_batch_device(eoc.OPS["XOR"], d0.data_ptr(), d1.data_ptr(), None, out.data_ptr(), S, stream=st.cuda_stream)
        eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S, stream=st.cuda_stream)
    st.synchronize()
    grows = L.eoc_engine_workspace_grows(eng.h)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        eng.gate_batch_device(eoc.OPS["XOR"], d0.data_ptr(), d1.data_ptr(), None, out.data_ptr(), S, stream=st.cuda_stream)
        eng.circuit_run_device(gates, wires.data_ptr(), n_wires, S, stream=st.cuda_stream)
    assert L.eoc_engine_workspace_grows(eng.h) == grows
    for rnd in range(2):          # replay on fresh operand values written into the captured buffers
        b0, b1 = rng.integers(0, 2, S), rng.integers(0, 2, S)
        A, B = rng.integers(0, 8, S), rng.integers(0, 8, S)
        d0.copy_(enc(b0, 10 + rnd))
        d1.copy_(enc(b1, 20 + rnd))
        for i in range(3):
            wires[aw[0] + i].copy_(enc((A >> i) & 1, 100 + 10 * rnd + i))
            wires[bw[0] + i].copy_(enc((B >> i) & 1, 200 + 10 * rnd + i))
        # a few other calls in between re-use the engine's descriptor ring: the captured graph must not care
        tmp = torch.empty_like(d0)
        for _ in range(3):
            eng.gate_batch_device(eoc.OPS["AND"], d0.data_ptr(), d1.data_ptr(), None, tmp.data_ptr(), S)
        sync()
        g.replay()
        sync()
        got = out.cpu().numpy()
        assert np.array_equal(got, orc.gate_batch(ol.OPS["XOR"], d0.cpu().numpy(), d1.cpu().numpy()))
        tot = sum(sk.decrypt_bits(wires[sw[0] + i].cpu().numpy()).astype(np.int64) << i for i in range(4))
        assert np.array_equal(tot, A + B)
    eng.close()


def test_mixed_batch_capture_and_replay(eoc):
    """a mixed batch that takes the gather path (more than 15 opcode runs) is captured with its permutation
    in the never-re-used arena: a replay after OTHER mixed batches (which rewrite the engine's pinned permutation
    buffer) still scatters in the captured order, and the next un-captured mixed batch works (perm_ev was not captured).
    A call that would have to grow the workspace under capture is refused."""
    torch = torch_cuda()
    p = eoc.default_params(0)
    p.n = 36
    sk = eoc.SecretKey(p, 23)
    orc = ol.Oracle(0, 23, n_override=36)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    L = eoc.lib()
    S = 90
    assert L.eoc_engine_reserve(eng.h, 2 * S, 256, S) == 0
    rng = np.random.default_rng(8)
    ops = rng.choice(np.array([0, 4, 10, 11, 2, 13], np.uint8), S)       # ~75 opcode runs: gather path
    assert (np.diff(ops.astype(int)) != 0).sum() + 1 > 15
    other = ops[::-1].copy()

    def enc(seed):
        bits = rng.integers(0, 2, S).astype(np.uint8)
        return sk.encrypt_bits(bits, seed, 0)

    c = [enc(k) for k in (1, 2, 3)]
    d = [to_dev(x) for x in c]
    out = torch.empty_like(d[0])
    tmp = torch.empty_like(d[0])
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        eng.gate_batch_device(0, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), S, ops=ops, stream=st.cuda_stream)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        eng.gate_batch_device(0, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), S, ops=ops, stream=st.cuda_stream)
    for rnd in range(2):
        c = [enc(10 * rnd + k) for k in (4, 5, 6)]
        for k in range(3):
            d[k].copy_(to_dev(c[k]))
        # another mixed batch in between rewrites the engine's own permutation buffer and records perm_ev
        eng.gate_batch_device(0, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), tmp.data_ptr(), S, ops=other)
        sync()
        assert np.array_equal(tmp.cpu().numpy(), orc.gate_batch(0, c[0], c[1], c[2], ops=other))
        out.zero_()
        g.replay()
        sync()
        assert np.array_equal(out.cpu().numpy(), orc.gate_batch(0, c[0], c[1], c[2], ops=ops)), f"replay {rnd}"
    # growth under capture is refused, loudly, and leaves the engine usable
    wide = 13 * S                         # 1170 rows: beyond the 1024-job minimum the reserve call allocated
    big = [to_dev(np.tile(x, (13, 1))) for x in c]
    bout = torch.empty_like(big[0])
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(eoc.EocError, match="graph capture"):
        with torch.cuda.graph(g2, stream=st):
            eng.gate_batch_device(0, big[0].data_ptr(), big[1].data_ptr(), None, bout.data_ptr(), wide, stream=st.cuda_stream)
    sync()
    eng.gate_batch_device(0, big[0].data_ptr(), big[1].data_ptr(), None, bout.data_ptr(), wide)
    sync()
    assert np.array_equal(bout.cpu().numpy()[:S], orc.gate_batch(0, c[0], c[1]))
    eng.close()


def test_subtractor_and_multiplier_bit_exact(eoc):
    """round 3 word-level circuits on the engine: 5-bit subtractor and 3-bit multiplier (30 bootstraps, CONST-free) over
    21 instances, every written wire against the oracle, results against plaintext"""
    from eoc_tfhe_amd import circuits
    p, sk, eng, orc = _setup(eoc, 0, 6, 28)
    S = 21
    rng = np.random.default_rng(12)
    w4 = circuits.MULTIPLIER_FORMS["wallace"](4)[0]
    c4 = (eoc.circuit_bootstraps(w4), circuits.bootstrap_depth(w4))
    for build, nbits, expect in ((circuits.subtractor, 5, lambda a, b: (a - b) % 32), (circuits.multiplier, 3, lambda a, b: a * b),
                                 (circuits.prefix_subtractor, 5, lambda a, b: (a - b) % 32),        # round 6: log-depth form
                                 (circuits.prefix_subtractor, 8, lambda a, b: (a - b) % 256),
                                 (circuits.maj_subtractor, 8, lambda a, b: (a - b) % 256),         # XOR3 + MAJ(NOT a, b, borrow)
                                 (circuits.wallace_multiplier, 4, lambda a, b: a * b),             # column compression + prefix addition
                                 (lambda n: circuits.multiplier_for(n, S), 4, lambda a, b: a * b)): # ... after eoc_netlist_optimize
        res = build(nbits)
        gates, n_wires, aw, bw, outw = res[0], res[1], res[2], res[3], res[4]
        A, B = rng.integers(0, 1 << nbits, S), rng.integers(0, 1 << nbits, S)
        wires = np.zeros((n_wires, S, p.n + 1), np.int32)
        for i in range(nbits):
            wires[aw[i]] = sk.encrypt_bits((A >> i) & 1, 300 + i, 0)
            wires[bw[i]] = sk.encrypt_bits((B >> i) & 1, 400 + i, 0)
        got = eoc_run(eoc, eng, gates, wires, n_wires, S)
        want = _oracle_run(orc, gates, wires)
        for g in gates:
            assert np.array_equal(got[g.out], want[g.out]), (build.__name__, g.out)
        if build.__name__ == "<lambda>":
            assert eoc.circuit_bootstraps(gates) == c4[0] and eoc.netlist_levels(gates)[2] == c4[1]  # the column form, optimized
        val = sum(sk.decrypt_bits(got[w]).astype(np.int64) << i for i, w in enumerate(outw))
        assert np.array_equal(val, expect(A, B)), build.__name__
        if "subtractor" in build.__name__:
            assert np.array_equal(sk.decrypt_bits(got[res[5]]), (A < B).astype(np.uint8)), build.__name__   # the final borrow
