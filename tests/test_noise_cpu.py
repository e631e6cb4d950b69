"""Noise anchor, CPU twin (VERDICT r4 task 1): the oracle's blind rotation and key switch,
measured over fresh encryptions, against the per-key CGGI prediction of eoc_tfhe_amd/noise.py -- mean and variance,
before and after the key switch, Set A and Set B.

Why this pins something bit-exactness cannot: GPU == oracle says the two agree, not that either follows TFHE as
published.  The predictions come from SURVEY.md Appendix A alone (gadget offset that centres the digits but TRUNCATES
the remainder, key-switch precision offset that rounds, subtraction of rows): with a rounding decomposition the
blind rotation's variance would be 0.65x (A) / 0.75x (B) of what is asserted here, with a missing key-switch
precision offset the key switch's mean would move by ~ |s'| 2^-18, with a wrong gadget base or row scaling every
number moves by factors.  Each of those still decrypts every gate.

The GPU twin (tests/test_gpu_noise.py) runs the same measurement over 16 384 encryptions per set on the HIP path.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_lib as ol
from eoc_tfhe_amd import noise

N = 1024
COUNT = 1536          # variance estimate +-3.6 % (1 sigma): the asserted window is [0.8, 1.25]


def _measure(o, count, seed):
    rng = np.random.default_rng(seed)
    b0, b1 = rng.integers(0, 2, count), rng.integers(0, 2, count)
    c0, c1 = o.encrypt_bits(b0, 1000 + seed), o.encrypt_bits(b1, 2000 + seed)
    t = [o.gate_linear(ol.OPS["NAND"], c0[i], c1[i]) for i in range(count)]
    with ThreadPoolExecutor(max(1, min(8, ol.lib().orc_max_threads()))) as ex:   # ctypes calls release the GIL
        u = np.stack(list(ex.map(o.blind_rotate_extract, t)))
        out = np.stack(list(ex.map(o.keyswitch, u)))
    assert np.array_equal(o.decrypt_bits(out), 1 - (b0 & b1))
    return noise.measure(u, out, o.lwe_key, o.tlwe_key), np.stack(t)


@pytest.mark.parametrize("pset,key", [(0, 1), (1, 1), (0, 77)], ids=["setA", "setB", "setA-holdout-key77"])
def test_oracle_noise_matches_prediction(pset, key):
    """key 77 is the hold-out: never measured while eoc_tfhe_amd/noise.py's refinements were chosen (its docstring)"""
    o = ol.Oracle(pset, key)
    pred = noise.predict(o.p, o.lwe_key, o.tlwe_key, o.ksk)
    (e_br, e_ks, e_tot), t = _measure(o, COUNT, 7 + pset)
    r = noise.compare(pred, e_br, e_ks, e_tot)
    cm = noise.br_conditional_mean(o.p, o.lwe_key, o.tlwe_key, t)
    r.update(noise.regress(e_br, cm, pred))
    r.update(noise.residual_mean(e_br, cm, noise.br_early_term(o.p, o.lwe_key, o.tlwe_key, o.bk, t)))
    print({k: (f"{v:.4e}" if isinstance(v, float) else v) for k, v in r.items()})
    assert 0.8 < r["br_ratio"] < 1.25, r          # blind rotation: rows + remainder + truncation-bias terms
    assert 0.8 < r["ks_ratio"] < 1.25, r          # key switch, this key's rows
    assert abs(r["br_mean_z"]) < 5 and abs(r["ks_mean_z"]) < 5, r     # both means are predicted, not just "small"
    assert abs(r["ks_mean"]) > 4 * np.sqrt(r["ks_var"] / r["count"])  # ... and the key switch's is resolvably non-zero
    assert r["max_abs_err"] < 1 / 16
    # sample by sample (noise.br_conditional_mean): slope 1 on the truncation model's conditional mean, standard error
    # 0.036 (A) / 0.044 (B) at this sample size
    assert 0.78 < r["br_cm_slope"] < 1.22, r
    assert abs(r["br_cm_corr"] - r["br_cm_corr_pred"]) < 0.1, r
    assert abs(r["br_resid_z"]) < 5, r            # what the per-sample model leaves has zero mean
    # negative control: the steps taken in the opposite order (rho_i from the EARLIER steps) -- slope 0.  (Running the
    # rotation backwards is no control: M is antisymmetric about N/2 to leading order, (X^-rho M)[0] ~ (X^rho M)[0].)
    t64 = t.astype(np.int64)
    body = t64[:, :-1][:, ::-1]
    flip = noise.regress(e_br, noise.br_conditional_mean(o.p, o.lwe_key[::-1], o.tlwe_key, np.concatenate([body, t64[:, -1:]], 1)))
    assert abs(flip["br_cm_slope"]) < 0.25, flip
    # the measurement separates this algorithm from its nearest neighbours: against the average-case textbook formula
    # (a ROUNDING decomposition) the blind rotation is 1.53x (A) / 1.33x (B) too noisy, against the average-over-keys
    # key-switch formula this key's rows are 0.75x
    assert r["br_ratio_textbook"] > 1.15 and r["ks_ratio_textbook"] < 0.87, r


def test_prediction_terms_setA_by_hand():
    """the closed forms, evaluated by hand for Set A with an idealised key (|s| = n/2, |s'| = N/2)"""
    class P:
        n, l, Bgbit, ks_t, ks_basebit, ks_stdev, bk_stdev = 500, 2, 10, 8, 2, 2.44e-5, 7.18e-9
    s = np.zeros(500, np.int64); s[::2] = 1
    s1 = np.zeros(N, np.int64); s1[::2] = 1
    p = noise.predict(P, s, s1)
    # rows: sigma^2 corrected for gaussian32's truncation toward zero (30.8 units: -2.6 %), and step 0, which
    # sees the trivial accumulator, counted at 0.5 * 256^2 / (4 * E[d^2]) = 9.4 % of a regular step
    u = 2.0**-32
    sig2 = 7.18e-9**2 - 7.18e-9 * np.sqrt(2 / np.pi) * u + u * u / 3
    assert sig2 / 7.18e-9**2 == pytest.approx(0.9745, abs=2e-4)
    ed2 = (1024**2 + 2) / 12
    n_eff = (500 - 1 * (1 - 0.5 * 256**2 / (4 * ed2))) * (1 - 1 / 2048)
    assert p["br_var_rows"] == pytest.approx(n_eff * 4 * 1024 * ed2 * sig2, rel=1e-12)                     # 8.97e-6
    assert p["br_var_remainder"] == pytest.approx(249 * 513 * 2.0**-40 / 12, rel=1e-12)                   # 9.7e-9
    # alternating key: (J*(1-s'))[r] = 1 - (r//2 + 1) + (511 - r//2) = 511 - 2 (r//2): mean square ~ N^2/12
    jv = 511 - 2 * (np.arange(N) // 2)
    # 250 active steps: step 0 (trivial accumulator, zero remainder) and the last one (rotated by nothing: the constant
    # M_BR) carry no truncation VARIANCE
    assert p["br_var_truncation_bias"] == pytest.approx(248 * 2.0**-42 * (jv.astype(float)**2).mean(), rel=1e-12)
    assert 4.5e-6 < p["br_var_truncation_bias"] < 5.5e-6
    assert p["br_mean"] == pytest.approx(-(2.0**-21) * 511, rel=1e-12)        # 1 + |s'| - 2 s'_0, s'_0 = 1
    assert p["ks_var_textbook"] == pytest.approx(1024 * 8 * 0.75 * 2.44e-5**2 + 512 * 2.0**-34 / 3, rel=1e-12)


def test_every_circuit_form_stays_inside_the_noise_budget():
    """round 6: the netlist compiler chains MUX outputs (2 V_BR + V_KS) into selectors and XORs more than the textbook
    forms do.  With the per-key predicted V_BR / V_KS of both default sets every blind rotation of every shipped form --
    as written and after eoc_netlist_optimize, with and without the extension gates -- keeps at least 12 standard deviations
    to its decision boundary (failure probability below 1e-32 per gate); the worst input the gate set allows at all, an XOR3
    of three MUX outputs, still has 13"""
    from eoc_tfhe_amd import circuits as c
    for pset in (0, 1):
        o = ol.Oracle(pset, 1)
        pred = noise.predict(o.p, o.lwe_key, o.tlwe_key, o.ksk)
        v_ms = (1 + int(np.asarray(o.lwe_key).sum())) / (48.0 * N * N)
        fresh = float(o.p.ks_stdev) ** 2
        forms = [c.ripple_carry_adder(8, carry_in_zero=True), c.mux_carry_adder(8), c.prefix_adder(8), c.prefix_adder(16),
                 c.subtractor(8), c.prefix_subtractor(8), c.less_than(8), c.less_than_tree(8), c.min_max_for(8, 1),
                 c.multiplier(4), c.wallace_multiplier(8), c.wallace_multiplier(8, False), c.string_equal(4),
                 c.maj_adder(8), c.maj_subtractor(8), c.maj_less_than(8), c.min_max_for(8, 64), c.min_max_for(8, 4096), c.multiplier(8)]
        worst = []
        for built in forms:
            gates = built[0]
            outs = built[-1] if isinstance(built[-1], list) else [built[-1]]
            for nl in (gates, c.optimize(gates, outs), c.optimize(gates, outs, extension_gates=False)):
                m, k = c.noise_margin(nl, fresh, pred["br_var"], pred["ks_var"], v_ms)
                worst.append(m)
                assert m > 12.0, (pset, len(nl), m, k)
        # the worst inputs the gate set allows: an XOR of two MUX outputs, and an XOR3 of three (no shipped form has the latter)
        g = [Gate_(10, 0, 1, 2, 6), Gate_(10, 3, 4, 5, 7), Gate_(4, 6, 7, -1, 8)]
        m2, k = c.noise_margin(g, fresh, pred["br_var"], pred["ks_var"], v_ms)
        g = [Gate_(10, 0, 1, 2, 6), Gate_(10, 3, 4, 5, 7), Gate_(10, 0, 3, 5, 8), Gate_(16, 6, 7, 8, 9)]
        m3, k3 = c.noise_margin(g, fresh, pred["br_var"], pred["ks_var"], v_ms)
        assert k == 2 and k3 == 3 and 12.0 < m3 < m2 and m3 <= min(worst) + 1e-9, (m2, m3, min(worst))
        print(f"set {'AB'[pset]}: smallest margin over the shipped forms {min(worst):.1f} sigma; XOR of two MUX outputs {m2:.1f}, "
              f"XOR3 of three MUX outputs {m3:.1f} sigma")


def Gate_(op, i0, i1, i2, out):
    from eoc_tfhe_amd import Gate
    return Gate(op, i0, i1, i2, out)

