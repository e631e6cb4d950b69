"""Noise anchor on the HIP path (VERDICT r4 task 1): 16 384 fresh encryptions per set through
eoc_blind_rotate_device and eoc_keyswitch_device; the measured mean and variance of the phase error, before and after
the key switch, against the per-key CGGI prediction of eoc_tfhe_amd/noise.py (derivation: DESIGN.md 2.3).
CPU twin on the oracle: tests/test_noise_cpu.py.  This replaces the old `err.std() < max_stdev` checks, which a wrong
gadget offset or precision offset would have passed."""
import numpy as np
import pytest

from eoc_tfhe_amd import noise
from gpu_util import dev_empty, sync, to_dev, torch_cuda

pytestmark = pytest.mark.gpu
N = 1024
COUNT = 16384         # variance estimate +-1.1 % (1 sigma)


@pytest.fixture(scope="module")
def eoc(built_lib):
    torch_cuda()
    import eoc_tfhe_amd
    return eoc_tfhe_amd


def run_noise(eoc, pset, count=COUNT, seed=1):
    """-> noise.compare() dict for `count` NAND inputs on fresh encryptions (also used by bench.py's secondary leg)"""
    torch = torch_cuda()
    p = eoc.default_params(pset)
    sk = eoc.SecretKey(p, seed)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    rng = np.random.default_rng(100 + pset)
    b0, b1 = rng.integers(0, 2, count), rng.integers(0, 2, count)
    c0, c1 = sk.encrypt_bits(b0, 4001 + pset), sk.encrypt_bits(b1, 4101 + pset)
    # bootsNAND's linear stage: t = (0, 1/8) - c0 - c1, wrapping
    t = (-(c0.astype(np.int64) + c1.astype(np.int64)))
    t[:, -1] += 1 << 29
    t = (t & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    d_t = to_dev(t)
    d_u = dev_empty((count, N + 1), torch.int32)
    d_o = dev_empty((count, p.n + 1), torch.int32)
    eng.blind_rotate_device(d_t.data_ptr(), d_u.data_ptr(), count)
    eng.keyswitch_device(d_u.data_ptr(), d_o.data_ptr(), count)
    sync()
    u, out = d_u.cpu().numpy(), d_o.cpu().numpy()
    assert np.array_equal(sk.decrypt_bits(out), 1 - (b0 & b1))
    # the two-call path is the gate: same bytes as eoc_gate_batch_device
    d_g = dev_empty((count, p.n + 1), torch.int32)
    d_c0, d_c1 = to_dev(c0), to_dev(c1)
    eng.gate_batch_device(eoc.OPS["NAND"], d_c0.data_ptr(), d_c1.data_ptr(), None, d_g.data_ptr(), count)
    sync()
    assert np.array_equal(d_g.cpu().numpy(), out)
    pred = noise.predict(p, sk.lwe_key, sk.tlwe_key, sk.ksk)
    e_br, e_ks, e_tot = noise.measure(u, out, sk.lwe_key, sk.tlwe_key)
    r = noise.compare(pred, e_br, e_ks, e_tot)
    cm = noise.br_conditional_mean(p, sk.lwe_key, sk.tlwe_key, t)
    r.update(noise.regress(e_br, cm, pred))
    r.update(noise.residual_mean(e_br, cm, noise.br_early_term(p, sk.lwe_key, sk.tlwe_key, sk.bk, t)))
    return r


@pytest.mark.parametrize("pset,seed", [(0, 1), (1, 1), (0, 5), (0, 77), (1, 77)],
                         ids=["setA", "setB", "setA-key5", "setA-holdout-key77", "setB-holdout-key77"])
def test_gpu_noise_matches_prediction(eoc, pset, seed):
    """(a second key for Set A: the key switch's bias and variance are properties of the KEY's rows -- predicted per key,
    not fitted.  Key 77 is the HOLD-OUT: the model's refinements were chosen while looking at keys 1, 2, 3, 4, 5, 9
    (profiles/r05_noise_262144.txt); key 77 was first measured after eoc_tfhe_amd/noise.py was frozen, and is held to a
    window of six standard errors of the variance estimate instead of [0.8, 1.25])"""
    r = run_noise(eoc, pset, seed=seed)
    print({k: (f"{v:.4e}" if isinstance(v, float) else v) for k, v in r.items()})
    if seed == 77:
        se = np.sqrt(2.0 / r["count"])
        assert abs(r["br_ratio"] - 1) < 6 * se and abs(r["ks_ratio"] - 1) < 6 * se, (se, r)
    assert 0.8 < r["br_ratio"] < 1.25, r
    assert 0.8 < r["ks_ratio"] < 1.25, r
    assert abs(r["br_mean_z"]) < 5 and abs(r["ks_mean_z"]) < 5, r
    # where a predicted mean is resolvably non-zero at this sample size (the blind rotation's always is; the key switch's is
    # a property of the key: +1.1e-3 for key 1, +2e-5 for key 5) so is the measured one: a sign slip fails, not just a scale slip
    for part in ("br", "ks"):
        se = np.sqrt(r[part + "_var"] / r["count"])
        if abs(r[part + "_mean_pred"]) > 8 * se:
            assert abs(r[part + "_mean"]) > 4 * se and r[part + "_mean"] * r[part + "_mean_pred"] > 0, (part, r)
    assert r["max_abs_err"] < 1 / 16
    # sample by sample: the error regresses on the truncation model's conditional mean (known from the public rotation
    # amounts and the key) with slope 1 and the predicted correlation -- the order in which the steps' rotations accumulate,
    # sign and size of the remainder; the opposite order gives slope 0 (tests/test_noise_cpu.py; standard error 0.011 / 0.014)
    assert 0.93 < r["br_cm_slope"] < 1.07, r
    assert abs(r["br_cm_corr"] - r["br_cm_corr_pred"]) < 0.04, r
    # ... and what the per-sample model (truncation + step 0's fixed row noise) leaves has zero mean
    assert abs(r["br_resid_z"]) < 5, r
    # the nearest neighbours are excluded: rounding decomposition (textbook formula) 1.53x (A) / 1.33x (B),
    # average-over-keys key switch 0.75x
    assert r["br_ratio_textbook"] > 1.2 and r["ks_ratio_textbook"] < 0.82, r


def test_gpu_mux_noise_is_two_rotations_and_one_key_switch(eoc):
    """bootsMUX: two blind rotations summed in the extracted domain, ONE key switch -- its output error has
    variance 2 V_BR + V_KS and mean 2 M_BR + M_KS (the two rotations see independent inputs), Set A, 16 384 samples"""
    torch = torch_cuda()
    p = eoc.default_params(0)
    sk = eoc.SecretKey(p, 1)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    rng = np.random.default_rng(300)
    b = [rng.integers(0, 2, COUNT) for _ in range(3)]
    d = [to_dev(sk.encrypt_bits(b[k], 4301 + k)) for k in range(3)]
    out = dev_empty((COUNT, p.n + 1), torch.int32)
    eng.gate_batch_device(eoc.OPS["MUX"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), COUNT)
    sync()
    o = out.cpu().numpy().astype(np.int64)
    want = np.where(b[0] == 1, b[1], b[2])
    assert np.array_equal(sk.decrypt_bits(out.cpu().numpy()), want)
    ph = ((o[:, -1] - o[:, :-1] @ sk.lwe_key.astype(np.int64)) + 2**31) % 2**32 - 2**31
    err = (ph - (2 * want - 1) * 2**29) / 2.0**32
    pred = noise.predict(p, sk.lwe_key, sk.tlwe_key, sk.ksk)
    var_pred, mean_pred = 2 * pred["br_var"] + pred["ks_var"], 2 * pred["br_mean"] + pred["ks_mean"]
    assert 0.8 < err.var() / var_pred < 1.25, (err.var(), var_pred)
    assert abs(err.mean() - mean_pred) < 5 * err.std() / np.sqrt(COUNT), (err.mean(), mean_pred)
    # and it is NOT one rotation's worth: the single-bootstrap prediction is excluded
    assert err.var() / pred["total_var"] > 1.5


def test_gpu_extension_gates_see_the_predicted_input_noise(eoc):
    """round 6: circuits.noise_margin prices the phase an extension gate's blind rotation sees -- XOR3: -2 (a + b + c), variance
    4 (V_a + V_b + V_c); MAJ: a + b + c, variance V_a + V_b + V_c (+ the mod-switch rounding, which the phase below does not
    contain).  Measured on the WORST input the gate set allows: three bootsMUX outputs (2 V_BR + V_KS each) from the GPU,
    Set A, 16 384 samples: the variance of t's phase error is within [0.8, 1.25] of the prediction, every sample stays inside
    its decision margin (1/4 for XOR3, 1/8 for MAJ) with room to spare, and both gates then decrypt correctly."""
    torch = torch_cuda()
    p = eoc.default_params(0)
    sk = eoc.SecretKey(p, 1)
    eng = eoc.Engine(p)
    eng.load_cloud_key(sk)
    rng = np.random.default_rng(310)
    pred = noise.predict(p, sk.lwe_key, sk.tlwe_key, sk.ksk)
    v_mux = 2 * pred["br_var"] + pred["ks_var"]
    s = sk.lwe_key.astype(np.int64)
    mux_out, mux_bits = [], []
    for k in range(3):
        b = [rng.integers(0, 2, COUNT) for _ in range(3)]
        d = [to_dev(sk.encrypt_bits(b[j], 4400 + 10 * k + j)) for j in range(3)]
        out = dev_empty((COUNT, p.n + 1), torch.int32)
        eng.gate_batch_device(eoc.OPS["MUX"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), COUNT)
        sync()
        mux_out.append(out)
        mux_bits.append(np.where(b[0] == 1, b[1], b[2]))
    ssum = sum(o.cpu().numpy().astype(np.int64) for o in mux_out)
    ones = mux_bits[0] + mux_bits[1] + mux_bits[2]
    for name, scale, var_pred, margin, truth in (("XOR3", -2, 12 * v_mux, 0.25, ones & 1), ("MAJ", 1, 3 * v_mux, 0.125, (ones >= 2) * 1)):
        t = scale * ssum
        ph = ((t[:, -1] - t[:, :-1] @ s) + 2**31) % 2**32 - 2**31
        ideal = scale * (2 * ones - 3) * 2**29                                   # the noiseless phase: scale * sum of +-1/8
        err = (((ph - ideal) + 2**31) % 2**32 - 2**31) / 2.0**32
        assert 0.8 < err.var() / var_pred < 1.25, (name, err.var(), var_pred)
        assert np.abs(err).max() < 0.6 * margin, (name, np.abs(err).max())        # 16 384 samples stay far inside the margin
        out = dev_empty((COUNT, p.n + 1), torch.int32)
        eng.gate_batch_device(eoc.OPS[name], mux_out[0].data_ptr(), mux_out[1].data_ptr(), mux_out[2].data_ptr(), out.data_ptr(), COUNT)
        sync()
        assert np.array_equal(sk.decrypt_bits(out.cpu().numpy()), truth), name
    eng.close()
