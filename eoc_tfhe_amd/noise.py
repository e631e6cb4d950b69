"""Predicted output noise of one gate bootstrap, for a GIVEN key (pure numpy; no GPU, no oracle).

This is the quantitative anchor of the parity statement: upstream libtfhe is absent from the
reference tree, so what ties this path to TFHE as published is (a) bit-exact agreement with the CPU restatement and
(b) agreement of the measured output noise -- mean AND variance, before and after the key switch -- with what the
CGGI analysis predicts for the conventions of SURVEY.md Appendix A.  A wrong gadget offset, a rounding instead of
upstream's truncating decomposition, a wrong key-switch precision offset or a mis-scaled key row moves one of the four
numbers below by tens of per cent while every gate still decrypts.

All quantities are in torus units (int32 / 2^32).  N = 1024, k = 1.

Blind rotation (tfhe_blindRotateAndExtract_FFT), error of the extracted sample under the
extracted key, over many input ciphertexts and ONE key:

  V_BR = n_eff * 2l * N * E[d^2] * sigma_eff^2           (the TLWE-zero rows of BK_i, every step; E[d^2] = (Bg^2 + 2)/12:
                                                          digits uniform on [-Bg/2, Bg/2), mean -1/2.  As `predict` computes
                                                          it: sigma_eff^2 = sigma_bk^2 - sigma_bk sqrt(2/pi) 2^-32 + 2^-64/3
                                                          -- gaussian32 truncates toward zero -- and n_eff = (n - 1 + f0)
                                                          (1 - 1/2N): step 0 works on the trivial accumulator and counts as
                                                          the fraction f0 of a regular step; a step whose amount is 0 adds
                                                          nothing)
       + w_eff * (1 + |s'|) * q^2 / 12                    (w_eff = w - s_0, w = Hamming weight of the LWE key: step 0's
                                                          remainder is exactly zero; |s'| = weight of the TLWE key; the
                                                          decomposition's remainder, q = Bg^-l)
       + (w_eff - 1) * (q/2)^2 * G(s')                    (the remainder is a TRUNCATION: upstream's offset
                                                          sum_p (Bg/2) 2^(32 - p Bgbit) centres the digits, not the
                                                          remainder, which is uniform on [0, q) with mean q/2; the LAST
                                                          active step contributes the constant M_BR below, not variance)

The third term is what CGGI's worst-case bound hides and the average-case textbook formula omits.  At a step with
s_i = 1 the update adds  -(low_b - low_a * s')  to the accumulator, low_x[j] in [0, q).  Its mean is the polynomial
-(q/2) J*(1 - s'), J = 1 + X + ... + X^(N-1); later steps multiply it by X^rho with rho uniform on [0, 2N) (the
remaining rotation), so coefficient 0 of the result picks +-((q/2) (J*(1 - s'))[r]) for a uniform r:
G(s') = mean_r ((J*(1-s'))[r])^2, about N^2/12 -- N/2 times the variance term next to it.  For the LAST step with
s_i = 1 nothing rotates any more (rho = 0): a deterministic bias

  M_BR = -(q/2) * (1 + |s'| - 2 s'_0).

(At 262 144 samples a third, input-dependent mean term resolves: step 0 multiplies constant band digits by the FIXED noise
of its rows -- about 50 (q/2), zero over uniform rotations.  tools/noise_mean_diag.py simulates
it from the key and matches the measured means per input class; `predict` keeps the input-independent M_BR.)

Key switch (lweKeySwitch; A.6), added error for ONE key whose rows have noises e[i][j][d]
(d = 1..base-1; d = 0 is "no row", e = 0), digits uniform:

  V_KS = sum_{i,j} ( mean_d e^2 - (mean_d e)^2 )  +  |s'| * 2^(-2 (t basebit + 1)) / 3
  M_KS = - sum_{i,j} mean_d e[i][j][d]            (res = (0, b') - sum of rows: the phase loses msg + e per row)

The textbook average over keys, kN t (1 - 1/base) sigma_ks^2, is ~ 4/3 of V_KS for a fixed key: a quarter of the
rows' second moment is the per-key constant M_KS, not spread.

How much of this is fitted: the three refinements (sigma_eff, n_eff / f0, w_eff) were added in round 5 while looking at
twelve 262 144-sample cases -- keys 1, 2, 3, 4, 5, 9 on both parameter sets (profiles/r05_noise_262144.txt) -- so the
0.9986 +- 0.0029 agreement on THOSE keys is partly tuned (each refinement is derived, none has a free parameter, but which
ones to derive was chosen by looking).  Key 77 was never looked at while the model was written: it is the asserted
HOLD-OUT of tests/test_gpu_noise.py and tests/test_noise_cpu.py.  Measured once the model was frozen, at 262 144
samples (profiles/r06_noise_holdout_262144.txt, keys 77 and 78, both sets): V_BR ratios 0.9999, 1.0011, 0.9951, 1.0013, V_KS
ratios 1.0006, 0.9972, 0.9990, 0.9960 (standard error 0.0028), all predicted means within 1.5 standard errors.
"""
import numpy as np

N = 1024


def _wrap32(x):
    return ((np.asarray(x, np.int64) + 2**31) % 2**32) - 2**31


def ksk_noise(ksk, lwe_key, tlwe_key, ks_t, ks_basebit):
    """e[i*t + j][d-1]: noise of key-switch row (i, j, d) in torus units (phase minus message)."""
    base = 1 << ks_basebit
    ksk = np.asarray(ksk, np.int64).reshape(N * ks_t * (base - 1), -1)
    s = np.asarray(lwe_key, np.int64)
    s1 = np.asarray(tlwe_key, np.int64)
    ph = _wrap32(ksk[:, -1] - ksk[:, :-1] @ s)
    r = np.arange(ksk.shape[0])
    d = r % (base - 1) + 1
    j = (r // (base - 1)) % ks_t
    i = r // ((base - 1) * ks_t)
    msg = _wrap32(s1[i] * (d << (32 - (j + 1) * ks_basebit)))
    return (_wrap32(ph - msg) / 2.0**32).reshape(N * ks_t, base - 1)


def predict(params, lwe_key, tlwe_key, ksk=None):
    """dict of predicted means / variances (torus units) for this key; `ksk` = [N t (base-1)][n+1] torus rows
    (None: the key-switch part uses the average-key textbook formula and a zero mean)."""
    n, l, Bgbit = int(params.n), int(params.l), int(params.Bgbit)
    t, bb = int(params.ks_t), int(params.ks_basebit)
    Bg, base = 1 << Bgbit, 1 << bb
    s = np.asarray(lwe_key, np.int64)
    s1 = np.asarray(tlwe_key, np.int64)
    w, hw = int(s.sum()), int(s1.sum())
    q = 2.0 ** (-l * Bgbit)
    # J * (1 - s') in Z[X]/(X^N + 1): coefficient r = sum_{m <= r} v_m - sum_{m > r} v_m
    v = -s1.astype(np.float64)
    v[0] += 1.0
    pre = np.cumsum(v)
    Jv = 2.0 * pre - pre[-1]
    out = {}
    # (a) gaussian32 converts sigma * z to Torus32 by TRUNCATION toward zero (dtot32 = int32(int64(frac * 2^32)),
    #     upstream's conversion): the stored noise is sign(x) floor(|x|), variance sigma^2 - sigma sqrt(2/pi) u + u^2/3 with
    #     u = 2^-32 -- 2.6 % below sigma^2 for Set A's sigma_bk (30.8 units), 0.6 % for Set B's (128 units); measured on the
    #     keys' rows: 0.9736 / 0.9924
    u = 2.0**-32
    sig = float(params.bk_stdev)
    sig2 = sig * sig - sig * np.sqrt(2.0 / np.pi) * u + u * u / 3.0
    # (b) STEP 0 sees the trivial accumulator (0, X^-b testvector): its mask polynomial decomposes to all-zero digits and its
    #     body to ~N/2 digits of 2 mu / h_1, so that step adds about (N/2) (2 mu / h_1)^2 / (2l N E[d^2]) of a regular step's
    #     row noise (9 % for Set A), and its remainder is exactly zero (no truncation term from it when s_0 = 1).  From step 1
    #     on the accumulator's mask is the pseudo-random sum of key-row masks -- a regular step whether s_0 was 0 or 1
    Ed2 = (Bg * Bg + 2) / 12.0
    d_triv = (1 << 30) >> (32 - Bgbit) if Bgbit <= 30 else 0          # first digit of +-2 mu = 2^30
    triv_frac = min(1.0, 0.5 * d_triv * d_triv / (2 * l * Ed2))
    n_eff = (n - (1.0 - triv_frac)) * (1.0 - 1.0 / (2 * N))          # step 0 only: after it the accumulator's MASK is
    w_eff = max(w - int(s[0]), 0)                                     # pseudo-random whatever s_0 is, and every later step regular
    out["br_var_rows"] = n_eff * 2 * l * N * Ed2 * sig2
    out["br_var_remainder"] = w_eff * (1 + hw) * q * q / 12.0
    # the LAST active step is rotated by nothing: its term is the constant M_BR (the mean below), not variance
    out["br_var_truncation_bias"] = max(w_eff - 1, 0) * (q / 2) ** 2 * float((Jv**2).mean())
    out["br_var"] = out["br_var_rows"] + out["br_var_remainder"] + out["br_var_truncation_bias"]
    out["br_mean"] = -(q / 2) * float(Jv[0])
    out["br_var_textbook"] = (n * 2 * l * N * (Bg * Bg / 12.0) * float(params.bk_stdev) ** 2
                              + n * (1 + N / 2) * q * q / 12.0)        # the average-case formula WITHOUT the truncation term
    ks_round = hw * 2.0 ** (-2 * (t * bb + 1)) / 3.0
    out["ks_var_textbook"] = N * t * (1 - 1.0 / base) * float(params.ks_stdev) ** 2 + ks_round
    if ksk is not None:
        e = ksk_noise(ksk, s, s1, t, bb)
        m1 = e.sum(1) / base
        m2 = (e**2).sum(1) / base
        out["ks_var"] = float((m2 - m1**2).sum()) + ks_round
        # res = (0, b') - sum of rows: phase(res) = b' - sum (msg + e) = phase(u) + rounding - sum e
        out["ks_mean"] = -float(m1.sum())
    else:
        out["ks_var"] = out["ks_var_textbook"]
        out["ks_mean"] = 0.0
    out["total_var"] = out["br_var"] + out["ks_var"]
    out["total_mean"] = out["br_mean"] + out["ks_mean"]
    return out


def rotation_amounts(t_rows):
    """modSwitchFromTorus32(., 2N) of every word of the LWE samples `t_rows` [count][n+1]: the blind
    rotation's abar_0..abar_{n-1} and barb (last column)."""
    t = np.asarray(t_rows, np.int64) & 0xFFFFFFFF
    return ((t + (1 << 20)) >> 21) & (2 * N - 1)


def br_conditional_mean(params, lwe_key, tlwe_key, t_rows):
    """What the truncating decomposition adds to EACH sample, given only public data and the key: sum over the steps with
    s_i = 1 (step 0 excepted: its accumulator is the trivial one, its remainder exactly zero) of coefficient 0 of
    X^rho_i * M, M = -(q/2) J*(1 - s'), rho_i = sum of the LATER active steps' rotation amounts (mod 2N).  [count] torus units.

    The measured error of a sample regresses on this with slope 1 and correlation sqrt(V_truncation / V_BR) (0.58 for
    Set A): a sample-by-sample check of the order in which the steps' rotations accumulate and of the remainder's sign
    and size -- which the variance alone does not see (steps taken in the opposite order give slope 0, a remainder of the
    other sign slope -1; the rotation's DIRECTION is not resolved: M is antisymmetric about N/2 to leading order)."""
    n, l, Bgbit = int(params.n), int(params.l), int(params.Bgbit)
    s = np.asarray(lwe_key, np.int64)
    s1 = np.asarray(tlwe_key, np.int64)
    q = 2.0 ** (-l * Bgbit)
    v = -s1.astype(np.float64)
    v[0] += 1.0
    pre = np.cumsum(v)
    M = -(q / 2) * (2.0 * pre - pre[-1])
    act = np.flatnonzero(s)
    A = rotation_amounts(t_rows)[:, :n][:, act]
    rc = np.cumsum(A[:, ::-1], axis=1)[:, ::-1]
    rho = (rc - A) % (2 * N)
    # (X^rho M)[0]: rho = 0 -> M[0]; 0 < rho < N -> -M[N - rho]; rho = N -> -M[0]; N < rho < 2N -> +M[2N - rho]
    idx = np.where(rho % N == 0, 0, np.where(rho < N, N - rho, 2 * N - rho))
    sgn = np.where(rho == 0, 1.0, np.where(rho <= N, -1.0, 1.0))
    contrib = sgn * M[idx]
    if len(act) and act[0] == 0:
        contrib[:, :1] = 0.0          # step 0 works on the trivial accumulator: its remainder is exactly zero
    return contrib.sum(1)


def br_early_term(params, lwe_key, tlwe_key, bk, t_rows, mu=1 << 29, chunk=16384):
    """STEP 0 sees the trivial accumulator (0, X^-barb testvector): the digits of (X^abar_0 - 1) ACC are the CONSTANT
    +-2 mu / h_1 on a band (first digit of the body polynomial only), so that step adds  (2 mu / h_1) * sum_band +-e_0  with the
    FIXED noise e_0 of row (q = 1, p = 1) of BK_0, rotated by the active steps behind it.  Deterministic given the key and the
    public rotation amounts; zero over uniform rotations, of the order of 50 (q/2) over NAND's three phase classes
    (DESIGN.md 2.3).  (After step 0 the accumulator's mask is a pseudo-random sum of key-row masks whether s_0 is 0 or 1:
    every later step is a regular one.)  `bk` = the torus rows [n][2l][2][N] (SecretKey.bk).  [count] torus units."""
    n, l, Bgbit = int(params.n), int(params.l), int(params.Bgbit)
    s = np.asarray(lwe_key, np.int64)
    s1 = np.asarray(tlwe_key, np.int64)
    bk = np.asarray(bk).reshape(n, 2 * l, 2, N)
    ones = np.flatnonzero(s)
    bara = rotation_amounts(t_rows)
    out = np.zeros(bara.shape[0])
    if not len(ones):
        return out
    i0 = 0          # step 0 only
    dig = float((2 * mu) >> (32 - Bgbit))
    S = np.zeros((N, N))                                   # a @ S = a * s' in Z[X]/(X^N + 1)
    for m_ in range(N):
        col = np.roll(s1.astype(np.float64), m_)
        col[:m_] *= -1
        S[m_, :] = col
    ext = []
    for i in range(i0 + 1):
        a_, b_ = bk[i, l, 0].astype(np.float64), bk[i, l, 1].astype(np.int64)   # row (q = 1, p = 1)
        ph = b_ - np.rint(a_ @ S).astype(np.int64)
        if s[i]:
            ph[0] -= 1 << (32 - Bgbit)                      # the row's message: s_i h_1 on the body's constant coefficient
        ei = _wrap32(ph) / 2.0**32
        ext.append(np.concatenate(([ei[0]], -ei[:0:-1], [-ei[0]], ei[:0:-1])))   # (X^k e_i)[0], k in [0, 2N)
    jj = np.arange(N)[None, :]
    act_after = [ones[ones > i] for i in range(i0 + 1)]
    for lo in range(0, bara.shape[0], chunk):
        B = bara[lo:lo + chunk]
        bb = B[:, n][:, None]

        def sign_of(j):                                     # coefficient j of X^(2N - barb) testvector, 2N-periodic extension
            return np.where(((j - (2 * N - bb)) % (2 * N)) < N, 1, -1)
        for i in range(i0 + 1):
            ab = B[:, i][:, None]
            k = (sign_of(jj - ab) - sign_of(jj)) // 2
            rho = (B[:, act_after[i]].sum(1) % (2 * N))[:, None]
            out[lo:lo + chunk] += dig * (k * ext[i][(jj + rho) % (2 * N)]).sum(1)
    return out


def regress(e_br, cond_mean, pred=None):
    """slope / correlation of the measured blind-rotation error on its per-sample conditional mean (+ the correlation the
    prediction implies and the slope's standard error)"""
    x = cond_mean - cond_mean.mean()
    y = e_br - e_br.mean()
    r = float((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum()))
    out = {"br_cm_slope": float((x * y).sum() / (x * x).sum()), "br_cm_corr": r,
           "br_cm_slope_se": float(np.sqrt(max(1.0 - r * r, 0.0) / max(r * r, 1e-30) / len(x)))}
    if pred is not None:
        out["br_cm_corr_pred"] = float(np.sqrt(pred["br_var_truncation_bias"] / pred["br_var"]))
    return out


def residual_mean(e_br, cond_mean, early=None):
    """mean of what the per-sample model leaves of the blind rotation's error (measured - conditional mean - early term), and
    its z: the sharpest mean test there is -- the model removes a third of the variance and all of the known bias"""
    res = e_br - cond_mean - (0.0 if early is None else early)
    return {"br_resid_mean": float(res.mean()), "br_resid_z": float(res.mean() / (res.std() / np.sqrt(len(res)))),
            "br_model_mean": float((cond_mean + (0.0 if early is None else early)).mean())}


def measure(u, out, lwe_key, tlwe_key):
    """errors of blind-rotation outputs `u` [count][N+1] under the extracted key and of the key-switched
    samples `out` [count][n+1] under the LWE key, against the nearest of +-1/8.  Returns (e_br, e_ks, e_total) in
    torus units; e_ks = e_total - e_br is the key switch's own contribution, sample by sample."""
    s = np.asarray(lwe_key, np.int64)
    s1 = np.asarray(tlwe_key, np.int64)
    u = np.asarray(u, np.int64).reshape(-1, N + 1)
    out = np.asarray(out, np.int64).reshape(u.shape[0], -1)
    phu = _wrap32(u[:, N] - u[:, :N] @ s1)
    pho = _wrap32(out[:, -1] - out[:, :-1] @ s)
    sign = np.where(phu > 0, 1, -1)
    e_br = (phu - sign * 2**29) / 2.0**32
    e_tot = (pho - sign * 2**29) / 2.0**32
    return e_br, e_tot - e_br, e_tot


def compare(pred, e_br, e_ks, e_tot):
    """measured / predicted summary (what the tests assert on and bench.py prints)"""
    cnt = len(e_br)
    return {
        "count": int(cnt),
        "br_var": float(e_br.var()), "br_var_pred": pred["br_var"], "br_ratio": float(e_br.var() / pred["br_var"]),
        "br_ratio_textbook": float(e_br.var() / pred["br_var_textbook"]),
        "br_mean": float(e_br.mean()), "br_mean_pred": pred["br_mean"],
        "br_mean_z": float((e_br.mean() - pred["br_mean"]) / (e_br.std() / np.sqrt(cnt))),
        "ks_var": float(e_ks.var()), "ks_var_pred": pred["ks_var"], "ks_ratio": float(e_ks.var() / pred["ks_var"]),
        "ks_ratio_textbook": float(e_ks.var() / pred["ks_var_textbook"]),
        "ks_mean": float(e_ks.mean()), "ks_mean_pred": pred["ks_mean"],
        "ks_mean_z": float((e_ks.mean() - pred["ks_mean"]) / (e_ks.std() / np.sqrt(cnt))),
        "total_std": float(e_tot.std()), "total_std_pred": float(np.sqrt(pred["total_var"])),
        "max_abs_err": float(np.abs(e_tot).max()),
    }
