"""CPU tests of the oracle: committed golden vectors, properties that do not need upstream libtfhe
(SURVEY.md 8c "what pins results instead").  No GPU needed."""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
NH = 512
N = 1024


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    g = json.load(open(os.path.join(HERE, "golden", "golden.json")))
    a = np.load(os.path.join(HERE, "golden", "golden_arrays.npz"), allow_pickle=False)
    return g, a


@pytest.fixture(scope="module")
def orc_a():
    return ol.Oracle(0, 1)


def test_prng_and_modswitch_kats(golden):
    g, _ = golden
    L = ol.lib()
    k = L.orc_stream_key(1, 3, 5)
    assert str(k) == g["prng"]["stream_key(1,3,5)"]
    assert [str(L.orc_rng_u64(k, i)) for i in range(4)] == g["prng"]["u64(key(1,3,5),0..3)"]
    k2 = L.orc_stream_key(9, 5, 0)
    assert [int(L.orc_gaussian32(k2, 10 + 2 * i, 0, 2.44e-5)) for i in range(4)] == \
        g["prng"]["gaussian32(key(9,5,0),ctr=10,mu=0,sigma=2.44e-5)x4"]
    m = g["modswitch"]
    assert L.orc_modswitch_to_torus32(1, 8) == m["to(1,8)"] == 2**29
    assert L.orc_modswitch_to_torus32(-1, 8) == m["to(-1,8)"] == -2**29
    assert L.orc_modswitch_to_torus32(1, 4) == m["to(1,4)"] == 2**30
    # the reference's own use: Msize = 2^31-1 round trip (eoc-tfhe-run.cpp:290,412; tests/tfhe.test.js:103 pins 42)
    assert m["from(to(42,M),M)"] == 42
    assert L.orc_modswitch_from_torus32(L.orc_modswitch_to_torus32(42, 2**31 - 1), 2**31 - 1) == 42
    assert L.orc_modswitch_from_torus32(-1, 2048) == m["from(-1,2048)"] == 0
    assert L.orc_modswitch_from_torus32(2**20 - 1, 2048) == 0
    assert L.orc_modswitch_from_torus32(2**20, 2048) == 1
    for phase in [0, 1, -1, 2**31 - 1, -2**31, 123456789, -987654321, 2**20, 2**21 + 2**20 - 1]:
        want = ((phase + 2**20) >> 21) & 2047
        assert L.orc_modswitch_from_torus32(phase, 2048) == want


def test_transform_golden_and_properties(golden):
    g, a = golden
    fs = ol.fft_fwd(a["fft_small_in"])
    fb = ol.fft_fwd(a["fft_big_in"])
    assert np.array_equal(fs, a["fft_small_spec"])
    assert sha(fs) == g["fft"]["small_spec_sha"] and sha(fb) == g["fft"]["big_spec_sha"]
    prod = (fs.view(np.complex128) * fb.view(np.complex128)).view(np.float64)
    assert np.array_equal(ol.fft_inv(prod), a["fft_prod_inv"])
    # round trip on full-range torus polynomials: the conversion TRUNCATES (Torus32(int64(x))), so a
    # value that the FP64 transform returns a hair below the integer comes back one unit closer to zero -- never more
    rng = np.random.default_rng(3)
    for _ in range(4):
        p = rng.integers(-2**31 + 1, 2**31, N).astype(np.int32)
        d = ol.fft_inv(ol.fft_fwd(p)).astype(np.int64) - p
        assert set(np.unique(d * np.sign(p))) <= {0, -1}
    # small integers (digit range) come back exactly or one unit towards zero as well; zero stays zero
    z = np.zeros(N, np.int32)
    assert np.array_equal(ol.fft_inv(ol.fft_fwd(z)), z)
    # linearity of the forward map on small integers is exact
    x = rng.integers(-512, 512, N).astype(np.int32)
    y = rng.integers(-512, 512, N).astype(np.int32)
    assert np.allclose(ol.fft_fwd(x) + ol.fft_fwd(y), ol.fft_fwd(x + y), rtol=0, atol=1e-6)
    # X * p is a negacyclic shift: spectrum of X^1 times spectrum of p
    X1 = np.zeros(N, np.int32); X1[1] = 1
    sp = (ol.fft_fwd(X1).view(np.complex128) * ol.fft_fwd(x).view(np.complex128)).view(np.float64)
    want = np.concatenate([[-x[-1]], x[:-1]]).astype(np.int32)
    d = ol.fft_inv(sp).astype(np.int64) - want
    assert set(np.unique(d * np.sign(want))) <= {0, -1}  # truncation toward zero


def test_forward_transform_is_the_evaluation_map_by_direct_summation():
    """An anchor that shares nothing with the implementation (no twiddle table, no butterfly graph): SURVEY.md A.7
    defines the forward map as P_m = p(zeta^(4m+1)), zeta = e^(i pi / N).  Every one of the 512 bins of the oracle's
    spectrum -- bin e holds the root index m = bitrev9(e) and is stored at sigma(e) = (e & 7) * 64 + (e >> 3) -- is
    compared with the polynomial evaluated by plain summation in extended precision, for a full-range torus polynomial
    and a digit-range one; and the inverse map returns the coefficients (truncated toward zero)."""
    rng = np.random.default_rng(12)
    j = np.arange(N, dtype=np.longdouble)
    pi = np.longdouble("3.14159265358979323846264338327950288")
    for lo, hi in ((-2**31, 2**31), (-512, 512)):
        poly = rng.integers(lo, hi, N).astype(np.int32)
        spec = ol.fft_fwd(poly).view(np.complex128)
        pl = poly.astype(np.longdouble)
        scale = float(np.abs(pl).sum())
        worst = 0.0
        for e in range(NH):
            m = int(format(e, "09b")[::-1], 2)
            ang = pi * (4 * m + 1) / np.longdouble(N)
            want = complex(float((pl * np.cos(ang * j)).sum()), float((pl * np.sin(ang * j)).sum()))
            worst = max(worst, abs(complex(spec[(e & 7) * 64 + (e >> 3)]) - want) / scale)
        assert worst < 1e-13, worst                      # binary64 round-off of a 1024-term sum, nothing structural
        back = ol.fft_inv(spec.view(np.float64)).astype(np.int64) - poly
        assert set(np.unique(back * np.sign(poly))) <= {0, -1}


def test_keyswitch_and_extract_by_a_numpy_restatement():
    """A second, independently written statement of SURVEY.md A.5 / A.6 (numpy, vectorised over the 8192 (i, j) pairs; it
    shares no code with oracle/tfhe_oracle.c): lweKeySwitch is  res = (0, b') - sum over (i, j) with digit d != 0 of
    KSK[i][j][d],  digit d = ((a'_i + 2^(31 - t basebit)) >> (32 - (j + 1) basebit)) & (base - 1);  and the key-switched
    sample decrypts, under the LWE key, to the phase the extracted sample has under the extracted TLWE key."""
    o = ol.Oracle(0, 5, n_override=24)
    p = o.p
    t, bb = p.ks_t, p.ks_basebit
    base = 1 << bb
    rng = np.random.default_rng(4)
    for _ in range(3):
        u = rng.integers(-2**31, 2**31, N + 1).astype(np.int32)
        abar = (u[:N].astype(np.int64) + (1 << (31 - t * bb))) & 0xFFFFFFFF
        jj = np.arange(t)
        dig = (abar[:, None] >> (32 - (jj[None, :] + 1) * bb)) & (base - 1)               # [N][t]
        ksk = o.ksk.reshape(N, t, base - 1, p.n + 1).astype(np.int64)                     # rows d = 1 .. base-1
        res = np.zeros(p.n + 1, np.int64)
        res[p.n] = int(u[N])
        ii, jx = np.nonzero(dig)
        res -= ksk[ii, jx, dig[ii, jx] - 1].sum(axis=0)
        want = (res & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        got = o.keyswitch(u)
        assert np.array_equal(got, want)
        # phases: <a', s'> under the extracted key (s'_j = tlwe_key[j]) against the key-switched sample under the LWE key
        ph_in = (int(u[N]) - int((u[:N].astype(np.int64) * o.tlwe_key).sum())) & 0xFFFFFFFF
        ph_out = (int(got[p.n]) - int((got[:p.n].astype(np.int64) * o.lwe_key).sum())) & 0xFFFFFFFF
        d = (ph_out - ph_in + 2**31) % 2**32 - 2**31
        assert abs(d) < 2**27, d      # rounding of the t-digit decomposition (2^15 per index) + key-switch noise, far below 1/8


def _schoolbook(a, b):
    full = np.zeros(2 * N, dtype=object)
    bo = b.astype(object)
    for m in range(N):
        if a[m]:
            full[m:m + N] += int(a[m]) * bo
    neg = full[:N] - full[N:]
    return np.array([int(v) % 2**32 for v in neg], dtype=np.uint64).astype(np.uint32).astype(np.int32)


def test_fft_product_vs_exact_schoolbook():
    """T2: the FP64 transform path equals the exact negacyclic product mod 2^32 up to a few LSB at
    the magnitudes of an external product (digits x full-range torus)."""
    rng = np.random.default_rng(8)
    worst = 0
    for Bg in (1024, 128):
        d = rng.integers(-Bg // 2, Bg // 2, N).astype(np.int32)
        t = rng.integers(-2**31, 2**31, N).astype(np.int32)
        got = ol.fft_inv((ol.fft_fwd(d).view(np.complex128) * ol.fft_fwd(t).view(np.complex128)).view(np.float64))
        dev = (got.astype(np.int64) - _schoolbook(d, t).astype(np.int64) + 2**31) % 2**32 - 2**31
        worst = max(worst, int(np.abs(dev).max()))
    assert worst <= 4, worst


def test_keys_match_golden(golden, orc_a):
    g, _ = golden
    o = orc_a
    e = g["A"]
    assert (o.n, o.l) == (e["n"], e["l"]) == (500, 2)
    assert sha(o.lwe_key) == e["lwe_key_sha"] and sha(o.tlwe_key) == e["tlwe_key_sha"]
    assert sha(o.bk) == e["bk_sha"] and sha(o.ksk) == e["ksk_sha"]
    assert sha(o.bkfft + 0.0) == e["bkfft_sha"]
    # TGSW structure: phase of every BK row is small noise + s_i * gadget on the diagonal
    i, row = 3, 1
    a, b = o.bk[i, row, 0].astype(np.int64), o.bk[i, row, 1].astype(np.int64)
    s = o.tlwe_key.astype(object)
    full = np.zeros(2 * N, dtype=object)
    for m in range(N):
        if s[m]:
            full[m:m + N] += a.astype(object)
    sa = np.array([int(v) for v in (full[:N] - full[N:])], dtype=object)
    phase = np.array([(int(b[j]) - int(sa[j]) + 2**31) % 2**32 - 2**31 for j in range(N)], dtype=np.int64)
    q, pp = row // o.l, row % o.l + 1
    h = int(o.lwe_key[i]) << (32 - pp * o.p.Bgbit)
    if q == 1:
        phase[0] -= h                                   # message on the body: + s_i * h
    else:
        phase += h * o.tlwe_key.astype(np.int64)        # message on the mask:  - s_i * h * s(X)
    assert np.abs(phase).max() < 2**32 * 7.18e-9 * 8  # 8 sigma


def test_gates_match_golden_set_a(golden, orc_a):
    g, a = golden
    o = orc_a
    c0, c1, c2 = a["A_c0"], a["A_c1"], a["A_c2"]
    assert np.array_equal(o.encrypt_bits([0, 0, 1, 1], 2, 0), c0)
    assert np.array_equal(o.encrypt_bits([0, 1, 0, 1], 2, 100), c1)
    b0, b1, b2 = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1]), np.array([1, 0, 0, 1])
    tt = dict(NAND=1 - (b0 & b1), AND=b0 & b1, OR=b0 | b1, NOR=1 - (b0 | b1), XOR=b0 ^ b1, XNOR=1 - (b0 ^ b1),
              ANDNY=(1 - b0) & b1, ANDYN=b0 & (1 - b1), ORNY=(1 - b0) | b1, ORYN=b0 | (1 - b1),
              MUX=np.where(b0 == 1, b1, b2), NOT=1 - b0, MAJ=(b0 + b1 + b2 >= 2) * 1, XOR3=b0 ^ b1 ^ b2)
    for name, want in tt.items():
        out = o.gate_batch(ol.OPS[name], c0, None if name == "NOT" else c1, c2 if name in ("MUX", "MAJ", "XOR3") else None)
        assert sha(out) == g["A"]["gates"][name]["sha"], name
        assert o.decrypt_bits(out).tolist() == want.tolist() == g["A"]["gates"][name]["bits"], name
    assert np.array_equal(o.gate_batch(ol.OPS["NAND"], c0, c1), a["A_NAND_out"])
    t = o.gate_linear(ol.OPS["NAND"], c0[3], c1[3])
    assert np.array_equal(o.blind_rotate_extract(t), a["A_nand_11_u"])


def test_gates_match_golden_set_b(golden):
    g, a = golden
    o = ol.Oracle(1, 1)
    e = g["B"]
    assert (o.n, o.l) == (630, 3)
    assert sha(o.bk) == e["bk_sha"] and sha(o.ksk) == e["ksk_sha"]
    out = o.gate_batch(ol.OPS["NAND"], a["B_c0"], a["B_c1"])
    assert np.array_equal(out, a["B_NAND_out"])
    assert o.decrypt_bits(out).tolist() == [1, 1, 1, 0]
    assert np.array_equal(o.gate_batch(ol.OPS["MUX"], a["B_c0"], a["B_c1"], a["B_c2"]), a["B_MUX_out"])
    for opn in ("MAJ", "XOR3"):
        assert np.array_equal(o.gate_batch(ol.OPS[opn], a["B_c0"], a["B_c1"], a["B_c2"]), a[f"B_{opn}_out"]), opn


def test_fft_step_close_to_exact_step(orc_a):
    """one CMux step: FP64 path vs exact integer external product differ by at most a few LSB"""
    import ctypes as C
    o = orc_a
    rng = np.random.default_rng(4)
    acc = rng.integers(-2**31, 2**31, 2 * N).astype(np.int32)
    a1, a2 = acc.copy(), acc.copy()
    i, abar = 5, 777
    bkfft_i = np.ascontiguousarray(o.bkfft[i])
    bk_i = np.ascontiguousarray(o.bk[i])
    o.L.orc_blind_rotate_step(C.byref(o.p), bkfft_i.ctypes.data, None, abar, a1, 1)
    o.L.orc_blind_rotate_step(C.byref(o.p), None, bk_i.ctypes.data, abar, a2, 0)
    dev = (a1.astype(np.int64) - a2.astype(np.int64) + 2**31) % 2**32 - 2**31
    assert np.abs(dev).max() <= 8, np.abs(dev).max()
    # a = 0 is the identity on the accumulator (what skipping the step amounts to)
    a3 = acc.copy()
    o.L.orc_blind_rotate_step(C.byref(o.p), bkfft_i.ctypes.data, None, 0, a3, 1)
    assert np.array_equal(a3, acc)


def test_gate_linear_stage_and_modswitch_by_a_numpy_restatement():
    """SURVEY.md 8a a1 / A.1 restated in numpy: the ten two-input gates' linear stage  t = (0, c) + s0 ca + s1 cb  with the
    constants of the published boots* family (1/8 = 2^29), and modSwitchFromTorus32(., 2N) as round(phi 2N / 2^32).
    Bit for bit against the oracle; and the sign of the phase of t IS the gate (what the bootstrap then refreshes)."""
    import ctypes as C
    o = ol.Oracle(0, 3, n_override=64, with_bk=False)
    table = {  # name: (constant in eighths of the torus, s0, s1)
        "NAND": (1, -1, -1), "AND": (-1, 1, 1), "OR": (1, 1, 1), "NOR": (-1, -1, -1), "XOR": (2, 2, 2), "XNOR": (-2, -2, -2),
        "ANDNY": (-1, -1, 1), "ANDYN": (-1, 1, -1), "ORNY": (1, -1, 1), "ORYN": (1, 1, -1)}
    sem = {"NAND": lambda a, b: 1 - (a & b), "AND": lambda a, b: a & b, "OR": lambda a, b: a | b, "NOR": lambda a, b: 1 - (a | b),
           "XOR": lambda a, b: a ^ b, "XNOR": lambda a, b: 1 - (a ^ b), "ANDNY": lambda a, b: (1 - a) & b,
           "ANDYN": lambda a, b: a & (1 - b), "ORNY": lambda a, b: (1 - a) | b, "ORYN": lambda a, b: a | (1 - b)}
    for a in (0, 1):
        for b in (0, 1):
            ca, cb = o.encrypt_bits([a], 40 + a, 0)[0], o.encrypt_bits([b], 50 + b, 0)[0]
            for name, (c8, s0, s1) in table.items():
                want = (s0 * ca.astype(np.int64) + s1 * cb.astype(np.int64))
                want[o.n] += c8 * (1 << 29)
                want = (want & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
                got = o.gate_linear(ol.OPS[name], ca, cb)
                assert np.array_equal(got, want), name
                phase = (int(got[o.n]) - int((got[:o.n].astype(np.int64) * o.lwe_key).sum()) + 2**31) % 2**32 - 2**31
                assert (phase > 0) == bool(sem[name](a, b)), (name, a, b)
                # mod-switch of the whole sample to [0, 2N): round(phi * 2N / 2^32)
                bara = np.zeros(o.n, np.int32)
                barb = np.zeros(1, np.int32)
                o.L.orc_modswitch_sample(C.byref(o.p), got, bara, barb)
                ms = ((got.astype(np.int64) & 0xFFFFFFFF) + (1 << 20) >> 21) & 2047
                assert np.array_equal(bara, ms[:o.n]) and barb[0] == ms[o.n], name


def test_extension_gates_maj_and_xor3_linear_stage_and_truth(orc_a):
    """round 6: the extension gates are NOT libtfhe functions -- they are libtfhe's primitives composed the way bootsXOR is
    (lweAddTo / lweAddMulTo over the inputs, then tfhe_bootstrap_FFT with mu = 1/8): MAJ  t = a + b + c,  XOR3  t = -2 (a + b + c).
    Restated in numpy from that one line each: for all eight input combinations the phase of t has the sign of the majority /
    the parity with the margins the header states (|phase| within noise of 1/8 or 3/8, resp. 1/4), the oracle's output is its
    own bootstrap of exactly that t (bit for bit), and it decrypts; then at Set A's full size through a carry chain."""
    o = ol.Oracle(0, 3, n_override=64)
    for k in range(8):
        bits = [(k >> j) & 1 for j in range(3)]
        cts = [o.encrypt_bits([bits[j]], 60 + 3 * k + j, 0)[0] for j in range(3)]
        ssum = sum(x.astype(np.int64) for x in cts)
        for name, scale, truth in (("MAJ", 1, int(sum(bits) >= 2)), ("XOR3", -2, bits[0] ^ bits[1] ^ bits[2])):
            t = ((scale * ssum) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
            phase = ((int(t[o.n]) - int((t[:o.n].astype(np.int64) * o.lwe_key).sum()) + 2**31) % 2**32 - 2**31) / 2**32
            assert (phase > 0) == bool(truth), (name, bits, phase)
            want_abs = 0.25 if name == "XOR3" else (0.375 if sum(bits) in (0, 3) else 0.125)
            assert abs(abs(phase) - want_abs) < 1e-3, (name, bits, phase)
            u = o.blind_rotate_extract(t)
            assert np.array_equal(o.gate_batch(ol.OPS[name], cts[0][None], cts[1][None], cts[2][None])[0], o.keyswitch(u)), name
    a = orc_a
    rng = np.random.default_rng(5)
    b = [rng.integers(0, 2, 48).astype(np.uint8) for _ in range(3)]
    c = [a.encrypt_bits(b[j], 70 + j, 0) for j in range(3)]
    carry, ssum = a.gate_batch(ol.OPS["MAJ"], *c), a.gate_batch(ol.OPS["XOR3"], *c)
    assert np.array_equal(a.decrypt_bits(carry), (b[0] + b[1] + b[2] >= 2)) and np.array_equal(a.decrypt_bits(ssum), b[0] ^ b[1] ^ b[2])
    nxt = a.gate_batch(ol.OPS["MAJ"], np.roll(c[0], 1, axis=0), np.roll(c[1], 1, axis=0), carry)   # a carry chain step
    assert np.array_equal(a.decrypt_bits(nxt), (np.roll(b[0], 1) + np.roll(b[1], 1) + a.decrypt_bits(carry) >= 2))
    ph = a.phases(carry) / 2**32
    assert np.abs(ph - np.sign(ph) * 0.125).max() < 1 / 16


def test_blind_rotate_frame_by_a_numpy_restatement(orc_a):
    """SURVEY.md A.4 / A.5 around the CMux steps, restated in numpy: ACC_0 = (0, X^(2N - barb) * (mu, ..., mu)), the loop
    over i with rotation amount bara[i] (a zero amount is the identity), and tLweExtractLweSample at index 0:
    a'[0] = ACC_0[0], a'[j] = -ACC_0[N - j], b' = ACC_1[0].  The steps themselves are the oracle's FP64 steps; the frame
    around them must reproduce orc_blind_rotate_extract bit for bit."""
    import ctypes as C
    o = orc_a
    mu = 1 << 29
    t = o.gate_linear(ol.OPS["NAND"], o.encrypt_bits([1], 61, 0)[0], o.encrypt_bits([0], 62, 0)[0])
    want = o.blind_rotate_extract(t, mu)
    ms = ((t.astype(np.int64) & 0xFFFFFFFF) + (1 << 20) >> 21) & 2047            # modSwitchFromTorus32(., 2N)
    bara, barb = ms[:o.n], int(ms[o.n])
    tv = np.full(N, mu, np.int64)
    a = (2 * N - barb) % (2 * N)
    sign = 1
    if a >= N:
        a, sign = a - N, -1
    body = sign * np.concatenate([-tv[N - a:], tv[:N - a]]) if a else sign * tv
    acc = np.concatenate([np.zeros(N, np.int64), body]).astype(np.int32)
    for i in range(o.n):
        if bara[i] == 0:
            continue
        o.L.orc_blind_rotate_step(C.byref(o.p), np.ascontiguousarray(o.bkfft[i]).ctypes.data, None, int(bara[i]), acc, 1)
    u = np.empty(N + 1, np.int32)
    u[0] = acc[0]
    u[1:N] = (-acc[N - 1:0:-1].astype(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    u[N] = acc[N]
    assert np.array_equal(u, want)
    # and the refreshed sample carries NAND(1, 0) = 1: phase under the extracted key close to +1/8
    ph = (int(u[N]) - int((u[:N].astype(np.int64) * o.tlwe_key).sum()) + 2**31) % 2**32 - 2**31
    assert abs(ph - mu) < 2**26, ph


def test_exact_cmux_step_by_a_numpy_restatement(orc_a):
    """SURVEY.md A.3 / A.4 written a second time, in numpy, sharing no code with the oracle's C: negacyclic rotation by
    X^a (a in [0, 2N)), the signed gadget decomposition with its offset, the row order (all p for q = 0, then q = 1) and
    the external product as exact integer convolutions mod 2^32 -- equal BIT FOR BIT to the oracle's exact-integer step,
    which in turn bounds the FP64 step (test above) that the GPU reproduces bit for bit."""
    import ctypes as C
    o = orc_a
    l, Bgbit = o.l, o.p.Bgbit
    Bg, half = 1 << Bgbit, 1 << (Bgbit - 1)
    offset = sum(half << (32 - pp * Bgbit) for pp in range(1, l + 1))
    rng = np.random.default_rng(14)

    def mul_xai(poly, a):                      # X^a * poly in Z[X]/(X^N + 1), int64 in, int64 out
        a %= 2 * N
        sign = 1
        if a >= N:
            a, sign = a - N, -1
        return sign * np.concatenate([-poly[N - a:], poly[:N - a]])

    def negacyclic(d, b):                      # exact: |d| <= 2^9, |b| <= 2^31, 1024 terms: < 2^51, fits int64
        full = np.convolve(d, b)
        return full[:N] - np.concatenate([full[N:], [0]])

    for i, abar in ((5, 777), (0, 1), (17, 1024), (499, 2047), (250, 1500)):
        acc = rng.integers(-2**31, 2**31, 2 * N).astype(np.int32)
        got = acc.copy()
        o.L.orc_blind_rotate_step(C.byref(o.p), None, np.ascontiguousarray(o.bk[i]).ctypes.data, abar, got, 0)
        A = acc.astype(np.int64).reshape(2, N)
        out = A.copy()
        for q in range(2):
            x = (mul_xai(A[q], abar) - A[q]) & 0xFFFFFFFF            # (X^a - 1) * ACC_q as uint32
            u = (x + offset) & 0xFFFFFFFF
            for pp in range(1, l + 1):
                digit = ((u >> (32 - pp * Bgbit)) & (Bg - 1)) - half    # in [-Bg/2, Bg/2)
                row = o.bk[i][q * l + (pp - 1)].astype(np.int64)         # TLWE row (q, p): two polynomials
                for c in range(2):
                    out[c] += negacyclic(digit, row[c])
        want = (out & 0xFFFFFFFF).astype(np.uint32).view(np.int32).reshape(-1)
        assert np.array_equal(got, want), (i, abar)


def test_bootstrap_noise_and_truth_many(orc_a):
    """64 random NANDs: every output decrypts correctly and sits within 1/16 of +-1/8; the noise has the stdev
    eoc_tfhe_amd/noise.py predicts for this key (4.1e-3 of the torus; 64 samples resolve it to +-9 % -- the tight
    measured-vs-predicted anchor, before and after the key switch, is tests/test_noise_cpu.py)."""
    from eoc_tfhe_amd import noise
    o = orc_a
    rng = np.random.default_rng(77)
    b0, b1 = rng.integers(0, 2, 64), rng.integers(0, 2, 64)
    c0, c1 = o.encrypt_bits(b0, 11, 0), o.encrypt_bits(b1, 12, 0)
    out = o.gate_batch(ol.OPS["NAND"], c0, c1)
    assert np.array_equal(o.decrypt_bits(out), 1 - (b0 & b1))
    ph = o.phases(out) / 2**32
    err = ph - np.sign(ph) * 0.125
    pred = noise.predict(o.p, o.lwe_key, o.tlwe_key, o.ksk)
    assert np.abs(err).max() < 1 / 16 and 0.65 < err.std() / np.sqrt(pred["total_var"]) < 1.45, (err.std(), pred)
    # a second level of gates on bootstrapped outputs still decrypts (noise does not accumulate)
    out2 = o.gate_batch(ol.OPS["XOR"], out, np.roll(out, 1, axis=0))
    w = 1 - (b0 & b1)
    assert np.array_equal(o.decrypt_bits(out2), w ^ np.roll(w, 1))


def test_conversion_range_stays_below_2_pow_51(orc_a):
    """a10: the HIP kernel converts with trunc + (1.5 * 2^52) and reads the low dword, which equals upstream's
    Torus32(int64(x)) for |x| < 2^51.  The oracle keeps the largest magnitude it ever converted: real bootstraps stay
    near 2^45, six binary orders below the limit."""
    o = orc_a
    o.L.orc_dbg_max_conv(1)
    rng = np.random.default_rng(5)
    b0, b1 = rng.integers(0, 2, 16), rng.integers(0, 2, 16)
    out = o.gate_batch(ol.OPS["NAND"], o.encrypt_bits(b0, 21, 0), o.encrypt_bits(b1, 22, 0))
    assert np.array_equal(o.decrypt_bits(out), 1 - (b0 & b1))
    mx = o.L.orc_dbg_max_conv(0)
    assert 2.0**38 < mx < 2.0**48, mx
    ob = ol.Oracle(1, 1)
    ob.L.orc_dbg_max_conv(1)
    ob.gate_batch(ol.OPS["NAND"], ob.encrypt_bits(b0[:4], 21, 0), ob.encrypt_bits(b1[:4], 22, 0))
    assert ob.L.orc_dbg_max_conv(0) < 2.0**48


def test_oracle_under_asan_ubsan(tmp_path):
    """the oracle's C code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only)"""
    import subprocess
    root = os.path.dirname(HERE)
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", "canon_twiddles.h"])  # the generated table copy
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-mavx2", "-mfma", "-ffp-contract=off", "-fopenmp", "-I" + os.path.join(root, "oracle"),
                           os.path.join(root, "tests", "c", "oracle_sanitize.c"), os.path.join(root, "oracle", "tfhe_oracle.c"),
                           "-o", exe, "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "oracle_sanitize OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_constant_gate_is_the_noiseless_trivial_sample(orc_a):
    """bootsCONSTANT: (0, ..., 0, +-1/8); decrypts to the bit under any key"""
    z = np.zeros((3, orc_a.n + 1), np.int32)
    for name, bit in (("CONST0", 0), ("CONST1", 1)):
        out = orc_a.gate_batch(ol.OPS[name], z)
        assert not out[:, :-1].any() and np.all(out[:, -1] == (1 << 29) * (2 * bit - 1))
        assert np.array_equal(orc_a.decrypt_bits(out), np.full(3, bit))
