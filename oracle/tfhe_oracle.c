/*
 * tfhe_oracle.c -- CPU oracle (plain C) for the TFHE gate-bootstrapping hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see tfhe_oracle.h for the full statement.
 *
 * Every function cites what it follows: SURVEY.md Appendix A (the recalled conventions of the
 * absent upstream module tfhe/tfhe@bc71bfae) and the nearest call site in /root/reference.
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp  (see oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the canonical transform names every fused multiply-add
 * explicitly (__builtin_fma) and every other * and + must round separately, so that the
 * HIP engine (compiled with the same flag) matches bit for bit.
 */
#define _GNU_SOURCE
#include "tfhe_oracle.h"
#include "canon_twiddles.h"
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define N ORC_N
#define NH ORC_NH
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

/* ------------------------------------------------------------------------------------------
 * parameters -- SURVEY.md 0.3 (wasm forensics of new_default_gate_bootstrapping_parameters,
 * called at ao-tfhe/eoc-tfhe-run.cpp:230)
 * ---------------------------------------------------------------------------------------- */
int orc_default_params(int set, orc_params *o)
{
    if (set == 0) { /* Set A: pinned library's "80-bit" set, BASELINE.json's numbers */
        o->n = 500; o->l = 2; o->Bgbit = 10; o->ks_t = 8; o->ks_basebit = 2;
        o->ks_stdev = 2.44e-5; o->bk_stdev = 7.18e-9;
        return 0;
    }
    if (set == 1) { /* Set B: what minimum_lambda = 128 (eoc-tfhe-run.cpp:34) selects */
        o->n = 630; o->l = 3; o->Bgbit = 7; o->ks_t = 8; o->ks_basebit = 2;
        o->ks_stdev = 1.0 / 32768.0; o->bk_stdev = 1.0 / 33554432.0;
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * PRNG -- this repo's own (DESIGN.md "PRNG").  The reference seeds libtfhe's generator from an
 * unseeded lrand48() (eoc-tfhe-run.cpp:226-228); nothing about its stream is pinned.
 * splitmix64 finaliser used as a counter-based generator: random access, order independent.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
    return z ^ (z >> 31);
}
#define GOLD UINT64_C(0x9E3779B97F4A7C15)

uint64_t orc_stream_key(uint64_t seed, uint32_t tag, uint64_t idx)
{
    uint64_t a = mix64(seed + GOLD * ((uint64_t)tag + 1));
    return mix64(a ^ (idx * UINT64_C(0xD1342543DE82EF95) + UINT64_C(0x632BE59BD9B4E019)));
}
uint64_t orc_rng_u64(uint64_t key, uint64_t ctr) { return mix64(key + GOLD * (ctr + 1)); }

enum { TAG_LWEKEY = 1, TAG_TLWEKEY = 2, TAG_BK = 3, TAG_KSK = 4, TAG_ENC = 5 };

static inline int32_t rng_torus32(uint64_t key, uint64_t ctr)
{
    return (int32_t)(uint32_t)(orc_rng_u64(key, ctr) >> 32);
}

/* gaussian32(mu, sigma) = mu + dtot32(N(0, sigma)) -- SURVEY.md A.1.  Box-Muller (cosine branch),
 * consumes counters ctr and ctr+1. */
int32_t orc_gaussian32(uint64_t key, uint64_t ctr, int32_t mu, double sigma)
{
    const double two53 = 1.0 / 9007199254740992.0;
    double u1 = (double)((orc_rng_u64(key, ctr) >> 11) + 1) * two53;     /* (0,1] */
    double u2 = (double)(orc_rng_u64(key, ctr + 1) >> 11) * two53;       /* [0,1) */
    double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    double d = sigma * z;
    double frac = d - (double)(int64_t)d;
    int64_t v = (int64_t)(frac * 4294967296.0);
    return (int32_t)((uint32_t)mu + (uint32_t)(uint64_t)v);
}

/* ------------------------------------------------------------------------------------------
 * scalar maps -- SURVEY.md A.1; modSwitchToTorus32 / modSwitchFromTorus32 are called by the
 * reference at eoc-tfhe-run.cpp:145,148,162,260,290,412 (with Msize = 2^31-1 there).
 * ---------------------------------------------------------------------------------------- */
int32_t orc_modswitch_to_torus32(int32_t mu, int32_t Msize)
{
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    uint64_t phase64 = (uint64_t)(int64_t)mu * interv;
    return (int32_t)(uint32_t)(phase64 >> 32);
}
int32_t orc_modswitch_from_torus32(int32_t phase, int32_t Msize)
{
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t)Msize) * 2;
    uint64_t half = interv / 2;
    uint64_t phase64 = ((uint64_t)(uint32_t)phase << 32) + half;
    return (int32_t)(phase64 / interv);
}

/* ------------------------------------------------------------------------------------------
 * key generation -- SURVEY.md A.2, A.3, A.6 (what new_random_gate_bootstrapping_secret_keyset,
 * eoc-tfhe-run.cpp:231, builds: lweKeyGen, tGswKeyGen, lweCreateKeySwitchKey, n x tGswSymEncryptInt)
 * ---------------------------------------------------------------------------------------- */
void orc_keygen_secret(const orc_params *p, uint64_t seed, int32_t *lwe_key, int32_t *tlwe_key)
{
    uint64_t k1 = orc_stream_key(seed, TAG_LWEKEY, 0);
    uint64_t k2 = orc_stream_key(seed, TAG_TLWEKEY, 0);
    for (int i = 0; i < p->n; i++) lwe_key[i] = (int32_t)(orc_rng_u64(k1, (uint64_t)i) >> 63);
    for (int j = 0; j < N; j++) tlwe_key[j] = (int32_t)(orc_rng_u64(k2, (uint64_t)j) >> 63);
}

/* b = gaussian32(mu, sigma) + <a, s>, a uniform -- lweSymEncrypt (eoc-tfhe-run.cpp:149,261,291) */
static void lwe_encrypt_stream(int n, const int32_t *s, uint64_t key, int32_t mu, double sigma,
                               int32_t *ct)
{
    uint32_t b = (uint32_t)orc_gaussian32(key, (uint64_t)n, mu, sigma);
    for (int m = 0; m < n; m++) {
        int32_t a = rng_torus32(key, (uint64_t)m);
        ct[m] = a;
        if (s[m]) b += (uint32_t)a;
    }
    ct[n] = (int32_t)b;
}

/* KSK[i][j][d], d = 1..base-1: LWE_n encryption of s'_i * d * 2^(32-(j+1)*basebit).
 * d = 0 rows (trivial zeros upstream) are not stored.  Upstream's centring of the noise samples
 * is a statistical nicety and is not reproduced. */
void orc_keygen_ksk(const orc_params *p, uint64_t seed, const int32_t *lwe_key,
                    const int32_t *tlwe_key, int32_t *ksk)
{
    const int n = p->n, t = p->ks_t, bb = p->ks_basebit, base = 1 << bb;
    const size_t rows = (size_t)N * t * (base - 1);
#pragma omp parallel for schedule(static) num_threads(orc_max_threads())
    for (size_t r = 0; r < rows; r++) {
        int d = (int)(r % (base - 1)) + 1;
        int j = (int)((r / (base - 1)) % t);
        int i = (int)(r / ((size_t)(base - 1) * t));
        uint32_t msg = tlwe_key[i] ? ((uint32_t)d << (32 - (j + 1) * bb)) : 0u;
        lwe_encrypt_stream(n, lwe_key, orc_stream_key(seed, TAG_KSK, r), (int32_t)msg,
                           p->ks_stdev, ksk + r * (size_t)(n + 1));
    }
}

/* BK_i = TGSW(s_i): kpl TLWE encryptions of 0, row (q,p) gets s_i * 2^(32-p*Bgbit) on the constant
 * coefficient of polynomial q.  bk[i][row][c][N], c=0: mask a, c=1: body b. */
void orc_keygen_bk(const orc_params *p, uint64_t seed, const int32_t *lwe_key,
                   const int32_t *tlwe_key, int32_t *bk)
{
    const int n = p->n, l = p->l, kpl = 2 * l;
#pragma omp parallel for schedule(dynamic, 8) num_threads(orc_max_threads())
    for (int ir = 0; ir < n * kpl; ir++) {
        int i = ir / kpl, row = ir % kpl;
        uint64_t key = orc_stream_key(seed, TAG_BK, (uint64_t)ir);
        int32_t *a = bk + ((size_t)ir * 2 + 0) * N;
        int32_t *b = bk + ((size_t)ir * 2 + 1) * N;
        for (int j = 0; j < N; j++) {
            a[j] = rng_torus32(key, (uint64_t)j);
            b[j] = orc_gaussian32(key, (uint64_t)N + 2 * (uint64_t)j, 0, p->bk_stdev);
        }
        /* b += s * a in Z[X]/(X^N+1), s binary */
        for (int m = 0; m < N; m++) {
            if (!tlwe_key[m]) continue;
            for (int j = 0; j < m; j++) b[j] = (int32_t)((uint32_t)b[j] - (uint32_t)a[j - m + N]);
            for (int j = m; j < N; j++) b[j] = (int32_t)((uint32_t)b[j] + (uint32_t)a[j - m]);
        }
        if (lwe_key[i]) {
            int q = row / l, pp = row % l + 1;
            uint32_t h = 1u << (32 - pp * p->Bgbit);
            int32_t *tgt = q ? b : a;
            tgt[0] = (int32_t)((uint32_t)tgt[0] + h);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * LWE encrypt / decrypt
 * ---------------------------------------------------------------------------------------- */
void orc_lwe_encrypt(const orc_params *p, const int32_t *lwe_key, uint64_t enc_seed, uint64_t idx,
                     int32_t mu, double sigma, int32_t *ct)
{
    lwe_encrypt_stream(p->n, lwe_key, orc_stream_key(enc_seed, TAG_ENC, idx), mu, sigma, ct);
}
/* lwePhase = b - <a,s>  (eoc-tfhe-run.cpp:161) */
int32_t orc_lwe_phase(const orc_params *p, const int32_t *lwe_key, const int32_t *ct)
{
    uint32_t ph = (uint32_t)ct[p->n];
    for (int m = 0; m < p->n; m++)
        if (lwe_key[m]) ph -= (uint32_t)ct[m];
    return (int32_t)ph;
}
/* bootsSymEncrypt: mu = +-1/8, sigma = in_out alpha_min = ks_stdev */
void orc_encrypt_bit(const orc_params *p, const int32_t *lwe_key, uint64_t enc_seed, uint64_t idx,
                     int bit, int32_t *ct)
{
    int32_t mu = orc_modswitch_to_torus32(bit ? 1 : -1, 8);
    orc_lwe_encrypt(p, lwe_key, enc_seed, idx, mu, p->ks_stdev, ct);
}
int orc_decrypt_bit(const orc_params *p, const int32_t *lwe_key, const int32_t *ct)
{
    return orc_lwe_phase(p, lwe_key, ct) > 0;
}

/* ------------------------------------------------------------------------------------------
 * Canonical transform v3 (SURVEY.md A.7 gives the maths).
 *
 * Replaces upstream IntPolynomial_ifft / TorusPolynomial_ifft / TorusPolynomial_fft (SURVEY.md
 * 8a a8,a10; the nayuki-portable FFT the reference selects in config.yml:22-26).  The upstream
 * butterfly order is unknowable here, so this repo owns the data-flow graph.  E[k] = exp(2 pi i k / 2048)
 * (canon_twiddles.h, correctly rounded, E[k+512] = i E[k] exactly), zeta = E[1], W[k] = E[4k].
 * Every butterfly multiplies BEFORE it adds, which allows the 6-operation fused form (Linzer-Feig):
 *
 *     a' = u + w v :  a'.re = fma(-w.im, v.im, fma(w.re, v.re, u.re))
 *                     a'.im = fma( w.re, v.im, fma(w.im, v.re, u.im))
 *     b' = u - w v =  2u - a' :  b' = fma(2, u, -a')           (component-wise)
 *
 *   forward (integer polynomial p -> 512 complex values p(zeta^(4m+1))): NO twist pass.  With
 *             c_j = p_j + i p_{j+512} the wanted values are the evaluations of sum_j c_j X^j at the roots of
 *             X^512 - i, so the evaluation tree of X^512 - i is walked directly (natural order in,
 *             bit-reversed root index out):  9 stages s = 0..8, h = 256 >> s, block b = i >> (9 - s),
 *               rho(s, b) = E[(256 + 1024 bitrev_s(b)) >> s],  (x_i, x_{i+h}) = (u + rho v, u - rho v).
 *             Stage 0 has the single root rho = e^{i pi/4} = (c, c), c = E[256].re, and integer inputs
 *             u = (a, b), v = (p, q); it is written on the exact integer sums dm = p - q, dp = p + q:
 *               x_j = (fma(c, dm, a), fma(c, dp, b)),  x_{j+256} = (fma(-c, dm, a), fma(-c, dp, b)).
 *             Stages 1..8 use the fused form for every root.
 *             Output bin e (root zeta^(4 bitrev9(e) + 1)) is stored at sigma(e) = (e & 7)*64 + (e >> 3).
 *   inverse:  stages s = 8..0 (decimation in time, bit-reversed in, natural out), h = 256 >> s:
 *               (u, v) = (x_i, x_{i+h}),  w = conj(W[(i mod h) << s]),  (x_i, x_{i+h}) = (u + w v, u - w v);
 *             in the three register-constant stages 8,7,6 w = 1 and w = conj(i) are exact moves:
 *               w = 1 -> (u + v, u - v);  w = conj(i) -> t = (v.im, -v.re), (u + t, u - t);
 *             y_j = x_j * conj(E[j]) / 512:  re = fma(x.re, t.re, x.im*t.im),  im = fma(x.im, t.re, -(x.re*t.im))
 *             with t = E[j]/512 (exact scaling);  p_j = Torus32(int64(Re y_j)), p_{j+512} = Torus32(int64(Im y_j)):
 *             TRUNCATION toward zero, then wrap -- the conversion of upstream's TorusPolynomial_fft
 *             (SURVEY.md 8a a10 / A.7).  v2 of this repo rounded to nearest; ORC_ROUND_NEAREST=1 in the
 *             environment selects that again (named, non-default mode, for A/B noise measurements only).
 * ---------------------------------------------------------------------------------------- */
static inline int sigma_of(int e) { return ((e & 7) << 6) | (e >> 3); }
static inline int bitrev_n(int b, int bits)
{
    int r = 0;
    for (int k = 0; k < bits; k++) r |= ((b >> k) & 1) << (bits - 1 - k);
    return r;
}

/* (u, v) -> (u + w v, u - w v) in the fused form, w = (wr, wi) */
static inline __attribute__((always_inline)) void butterfly_w(double *ur, double *ui, double *vr, double *vi, double wr, double wi)
{
    double ar = FMA(-wi, *vi, FMA(wr, *vr, *ur));
    double ai = FMA(wr, *vi, FMA(wi, *vr, *ui));
    *vr = FMA(2.0, *ur, -ar);
    *vi = FMA(2.0, *ui, -ai);
    *ur = ar; *ui = ai;
}

/* inverse butterfly with w = conj(W[k]); `moves`: stages 8,7,6 apply w = 1 and w = conj(i) as exact moves */
static inline __attribute__((always_inline)) void butterfly_inv(double *ur, double *ui, double *vr, double *vi, int k, int moves)
{
    double ar, ai;
    if (moves && k == 0) {
        ar = *ur + *vr; ai = *ui + *vi;
        *vr = *ur - *vr; *vi = *ui - *vi;
        *ur = ar; *ui = ai;
        return;
    }
    if (moves && k == 128) { /* w = conj(i) */
        double tr = *vi, ti = -*vr;
        ar = *ur + tr; ai = *ui + ti;
        *vr = *ur - tr; *vi = *ui - ti;
        *ur = ar; *ui = ai;
        return;
    }
    butterfly_w(ur, ui, vr, vi, EOC_E2048[4 * k][0], -EOC_E2048[4 * k][1]);
}

void orc_fft_fwd(const int32_t *poly, double *spec)
{
    double xr[NH], xi[NH];
    const double c = EOC_E2048[256][0];
    for (int j = 0; j < 256; j++) { /* stage 0 on exact integer sums */
        double a = (double)poly[j], b = (double)poly[j + NH];
        double pp = (double)poly[j + 256], q = (double)poly[j + 256 + NH];
        double dm = pp - q, dp = pp + q; /* exact: |.| <= 2^32 */
        xr[j] = FMA(c, dm, a);
        xi[j] = FMA(c, dp, b);
        xr[j + 256] = FMA(-c, dm, a);
        xi[j + 256] = FMA(-c, dp, b);
    }
    for (int s = 1; s < 9; s++) {
        int h = 256 >> s;
        for (int blk = 0; blk < (1 << s); blk++) {
            int k = (256 + 1024 * bitrev_n(blk, s)) >> s, base = blk * 2 * h;
            double wr = EOC_E2048[k][0], wi = EOC_E2048[k][1];
            for (int i = base; i < base + h; i++) butterfly_w(&xr[i], &xi[i], &xr[i + h], &xi[i + h], wr, wi);
        }
    }
    for (int e = 0; e < NH; e++) {
        spec[2 * sigma_of(e)] = xr[e];
        spec[2 * sigma_of(e) + 1] = xi[e];
    }
}

static int g_round_nearest = -1; /* ORC_ROUND_NEAREST=1: v2's conversion (non-default, diagnostics) */
static double g_max_abs_conv = 0.0; /* largest |value| ever converted (tests check it stays < 2^51) */
double orc_dbg_max_conv(int reset)
{
    double v = g_max_abs_conv;
    if (reset) g_max_abs_conv = 0.0;
    return v;
}
static inline int32_t wrap_convert(double x)
{
    if (g_round_nearest) x = rint(x);
    return (int32_t)(uint32_t)(uint64_t)(int64_t)x; /* C cast: truncation toward zero */
}

/* the inverse transform up to, not including, the conversion: vals[j] = Re y_j, vals[j + NH] = Im y_j */
void orc_fft_inv_raw(const double *spec, double *vals)
{
    double xr[NH], xi[NH];
    for (int e = 0; e < NH; e++) {
        xr[e] = spec[2 * sigma_of(e)];
        xi[e] = spec[2 * sigma_of(e) + 1];
    }
    for (int s = 8; s >= 0; s--) {
        int h = 256 >> s;
        for (int base = 0; base < NH; base += 2 * h)
            for (int j = 0; j < h; j++)
                butterfly_inv(&xr[base + j], &xi[base + j], &xr[base + j + h], &xi[base + j + h], j << s, s >= 6);
    }
    for (int j = 0; j < NH; j++) {
        double tc = EOC_E2048[j][0] * 0.001953125, ts = EOC_E2048[j][1] * 0.001953125;
        vals[j] = FMA(xr[j], tc, xi[j] * ts);
        vals[j + NH] = FMA(xi[j], tc, -(xr[j] * ts));
    }
}

/* tLweFromFFTConvert / TorusPolynomial_fft: Torus32(int64(x)), truncation then wrap */
void orc_fft_inv(const double *spec, int32_t *poly)
{
    double vals[N];
    if (g_round_nearest < 0) {
        const char *m = getenv("ORC_ROUND_NEAREST");
        g_round_nearest = (m && m[0] == '1') ? 1 : 0;
    }
    orc_fft_inv_raw(spec, vals);
    double mx = 0.0;
    for (int j = 0; j < N; j++) {
        if (fabs(vals[j]) > mx) mx = fabs(vals[j]);
        poly[j] = wrap_convert(vals[j]);
    }
    if (mx > g_max_abs_conv) g_max_abs_conv = mx; /* benign race between threads: a diagnostic high-water mark */
}

/* tGswToFFTConvert (keygen side): every BK polynomial through the forward map */
void orc_bk_to_fft(const orc_params *p, const int32_t *bk, double *bkfft)
{
    const int npoly = p->n * 2 * p->l * 2;
#pragma omp parallel for schedule(static) num_threads(orc_max_threads())
    for (int i = 0; i < npoly; i++) orc_fft_fwd(bk + (size_t)i * N, bkfft + (size_t)i * N);
}

/* ------------------------------------------------------------------------------------------
 * hot path
 * ---------------------------------------------------------------------------------------- */
typedef struct { int32_t cst8; int s0, s1; } gate_lin; /* cst in eighths of the torus */

/* bootsNAND & friends: t = (0, cst) + s0*ca + s1*cb   */
static int gate_table(int op, gate_lin *g)
{
    switch (op) {
    case ORC_NAND:  *g = (gate_lin){ 1, -1, -1}; return 0;
    case ORC_AND:   *g = (gate_lin){-1,  1,  1}; return 0;
    case ORC_OR:    *g = (gate_lin){ 1,  1,  1}; return 0;
    case ORC_NOR:   *g = (gate_lin){-1, -1, -1}; return 0;
    case ORC_XOR:   *g = (gate_lin){ 2,  2,  2}; return 0;
    case ORC_XNOR:  *g = (gate_lin){-2, -2, -2}; return 0;
    case ORC_ANDNY: *g = (gate_lin){-1, -1,  1}; return 0;
    case ORC_ANDYN: *g = (gate_lin){-1,  1, -1}; return 0;
    case ORC_ORNY:  *g = (gate_lin){ 1, -1,  1}; return 0;
    case ORC_ORYN:  *g = (gate_lin){ 1,  1, -1}; return 0;
    default: return -1;
    }
}

int orc_gate_linear(const orc_params *p, int op, const int32_t *ca, const int32_t *cb, int32_t *t)
{
    gate_lin g;
    if (gate_table(op, &g)) return -1;
    const int n = p->n;
    for (int m = 0; m <= n; m++)
        t[m] = (int32_t)((uint32_t)g.s0 * (uint32_t)ca[m] + (uint32_t)g.s1 * (uint32_t)cb[m]);
    t[n] = (int32_t)((uint32_t)t[n] + (uint32_t)orc_modswitch_to_torus32(g.cst8, 8));
    return 0;
}

/* tfhe_bootstrap_woKS_FFT preamble: bara_i, barb = modSwitchFromTorus32(., 2N) */
void orc_modswitch_sample(const orc_params *p, const int32_t *t, int32_t *bara, int32_t *barb)
{
    for (int m = 0; m < p->n; m++) bara[m] = orc_modswitch_from_torus32(t[m], 2 * N);
    *barb = orc_modswitch_from_torus32(t[p->n], 2 * N);
}

/* coefficient j of X^a * in, a in [0, 2N)  (torusPolynomialMulByXai) */
static inline int32_t rot_coef(const int32_t *in, int a, int j)
{
    int idx = (j - a) & (2 * N - 1);
    int32_t v = in[idx & (N - 1)];
    return (idx & N) ? (int32_t)(0u - (uint32_t)v) : v;
}

/* tGswTorus32PolynomialDecompH digit p (1..l) of x */
static inline int32_t decomp_digit(uint32_t x, uint32_t offset, int p, int Bgbit)
{
    uint32_t u = x + offset;
    uint32_t Bg = 1u << Bgbit;
    return (int32_t)((u >> (32 - p * Bgbit)) & (Bg - 1)) - (int32_t)(Bg >> 1);
}
static inline uint32_t decomp_offset(int l, int Bgbit)
{
    uint32_t off = 0;
    for (int pp = 1; pp <= l; pp++) off += (1u << (Bgbit - 1)) << (32 - pp * Bgbit);
    return off;
}

/* tfhe_MuxRotate_FFT + tLweAddTo: acc += BK_i (x) ((X^a - 1) acc).
 * FFT path: tGswFFTExternMulToTLwe with this repo's canonical accumulation order (v3): output polynomial c is ONE
 * chain over the 2l (q_in, p) terms, the digits of the OTHER input polynomial (q_in = 1 - c, p = 1..l) first, then
 * the own ones (q_in = c); first term a plain product (2 mul + 2 fma), every later term 4 fused multiply-adds.
 * (This is the order in which a wave pair of the HIP kernel produces it: the partner's partial chain arrives
 * through LDS and the own terms are accumulated onto it.)
 * Exact path (use_fft = 0): the same external product as a schoolbook negacyclic
 * convolution mod 2^32 -- the mathematical definition the FFT path approximates. */
void orc_blind_rotate_step(const orc_params *p, const double *bkfft_i, const int32_t *bk_i,
                           int a, int32_t *acc, int use_fft)
{
    const int l = p->l, Bgbit = p->Bgbit;
    const uint32_t off = decomp_offset(l, Bgbit);
    int32_t diff[2][N];
    for (int q = 0; q < 2; q++)
        for (int j = 0; j < N; j++)
            diff[q][j] = (int32_t)((uint32_t)rot_coef(acc + q * N, a, j) - (uint32_t)acc[q * N + j]);

    if (use_fft) {
        static _Thread_local double D[2][4][N]; /* [q_in][p-1][512 complex]: spectra of the digit polynomials */
        int32_t dec[N];
        for (int q = 0; q < 2; q++)
            for (int pp = 1; pp <= l; pp++) {
                for (int j = 0; j < N; j++) dec[j] = decomp_digit((uint32_t)diff[q][j], off, pp, Bgbit);
                orc_fft_fwd(dec, D[q][pp - 1]);
            }
        for (int c = 0; c < 2; c++) {
            double S[N];
            int32_t r[N];
            int first = 1;
            for (int qq = 0; qq < 2; qq++) {
                int q = qq == 0 ? 1 - c : c; /* the other input polynomial's digits first, then the own ones */
                for (int pp = 1; pp <= l; pp++) {
                    const double *B = bkfft_i + ((size_t)(q * l + (pp - 1)) * 2 + c) * N;
                    const double *X = D[q][pp - 1];
                    for (int e = 0; e < NH; e++) {
                        double dr = X[2 * e], di = X[2 * e + 1], br = B[2 * e], bi = B[2 * e + 1];
                        if (first) {
                            S[2 * e] = FMA(-di, bi, dr * br);
                            S[2 * e + 1] = FMA(di, br, dr * bi);
                        } else {
                            double re = FMA(dr, br, S[2 * e]);
                            S[2 * e] = FMA(-di, bi, re);
                            double im = FMA(dr, bi, S[2 * e + 1]);
                            S[2 * e + 1] = FMA(di, br, im);
                        }
                    }
                    first = 0;
                }
            }
            orc_fft_inv(S, r);
            for (int j = 0; j < N; j++) acc[c * N + j] = (int32_t)((uint32_t)acc[c * N + j] + (uint32_t)r[j]);
        }
    } else {
        uint32_t r[2][N];
        memset(r, 0, sizeof r);
        int32_t dec[N];
        for (int q = 0; q < 2; q++)
            for (int pp = 1; pp <= l; pp++) {
                int row = q * l + (pp - 1);
                for (int j = 0; j < N; j++) dec[j] = decomp_digit((uint32_t)diff[q][j], off, pp, Bgbit);
                for (int c = 0; c < 2; c++) {
                    const int32_t *B = bk_i + ((size_t)row * 2 + c) * N;
                    for (int m = 0; m < N; m++) {
                        uint32_t d = (uint32_t)dec[m];
                        if (!d) continue;
                        for (int j = 0; j < N - m; j++) r[c][j + m] += d * (uint32_t)B[j];
                        for (int j = N - m; j < N; j++) r[c][j + m - N] -= d * (uint32_t)B[j];
                    }
                }
            }
        for (int c = 0; c < 2; c++)
            for (int j = 0; j < N; j++) acc[c * N + j] = (int32_t)((uint32_t)acc[c * N + j] + r[c][j]);
    }
}

/* tfhe_blindRotateAndExtract_FFT: test vector = N copies of mu */
void orc_blind_rotate_extract(const orc_params *p, const double *bkfft, const int32_t *t,
                              int32_t mu, int32_t *u)
{
    const int n = p->n, kpl = 2 * p->l;
    int32_t *bara = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t barb;
    int32_t acc[2 * N], tv[N];
    orc_modswitch_sample(p, t, bara, &barb);
    for (int j = 0; j < N; j++) tv[j] = mu;
    for (int j = 0; j < N; j++) {
        acc[j] = 0;
        acc[N + j] = rot_coef(tv, (2 * N - barb) & (2 * N - 1), j);
    }
    for (int i = 0; i < n; i++) {
        if (bara[i] == 0) continue;
        orc_blind_rotate_step(p, bkfft + (size_t)i * kpl * 2 * N, NULL, bara[i], acc, 1);
    }
    /* tLweExtractLweSample, index 0 */
    u[0] = acc[0];
    for (int j = 1; j < N; j++) u[j] = (int32_t)(0u - (uint32_t)acc[N - j]);
    u[N] = acc[N];
    free(bara);
}

/* lweKeySwitch */
void orc_keyswitch(const orc_params *p, const int32_t *ksk, const int32_t *u, int32_t *out)
{
    const int n = p->n, t = p->ks_t, bb = p->ks_basebit, base = 1 << bb;
    const uint32_t prec_offset = 1u << (32 - (1 + bb * t));
    const uint32_t mask = (uint32_t)base - 1;
    for (int m = 0; m < n; m++) out[m] = 0;
    out[n] = u[N];
    for (int i = 0; i < N; i++) {
        uint32_t ai = (uint32_t)u[i] + prec_offset;
        for (int j = 0; j < t; j++) {
            uint32_t d = (ai >> (32 - (j + 1) * bb)) & mask;
            if (!d) continue;
            const int32_t *row = ksk + (((size_t)i * t + j) * (base - 1) + (d - 1)) * (size_t)(n + 1);
            for (int m = 0; m <= n; m++) out[m] = (int32_t)((uint32_t)out[m] - (uint32_t)row[m]);
        }
    }
}

/* tfhe_bootstrap_FFT */
void orc_bootstrap(const orc_params *p, const double *bkfft, const int32_t *ksk, const int32_t *t,
                   int32_t mu, int32_t *out)
{
    int32_t u[N + 1];
    orc_blind_rotate_extract(p, bkfft, t, mu, u);
    orc_keyswitch(p, ksk, u, out);
}

int orc_gate(const orc_params *p, const double *bkfft, const int32_t *ksk, int op,
             const int32_t *ca, const int32_t *cb, const int32_t *cc, int32_t *out)
{
    const int n = p->n;
    const int32_t mu = orc_modswitch_to_torus32(1, 8);
    if (op == ORC_NOT) {
        for (int m = 0; m <= n; m++) out[m] = (int32_t)(0u - (uint32_t)ca[m]);
        return 0;
    }
    if (op == ORC_COPY) {
        memcpy(out, ca, sizeof(int32_t) * (size_t)(n + 1));
        return 0;
    }
    if (op == ORC_CONST0 || op == ORC_CONST1) {
        /* bootsCONSTANT: lweNoiselessTrivial(result, value ? +1/8 : -1/8)   */
        memset(out, 0, sizeof(int32_t) * (size_t)n);
        out[n] = op == ORC_CONST1 ? mu : (int32_t)(0u - (uint32_t)mu);
        return 0;
    }
    int32_t *t = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int rc = 0;
    if (op == ORC_MUX) {
        /* bootsMUX(a,b,c) = a ? b : c   */
        int32_t u1[N + 1], u2[N + 1];
        orc_gate_linear(p, ORC_AND, ca, cb, t);
        orc_blind_rotate_extract(p, bkfft, t, mu, u1);
        orc_gate_linear(p, ORC_ANDNY, ca, cc, t);
        orc_blind_rotate_extract(p, bkfft, t, mu, u2);
        for (int j = 0; j <= N; j++) u1[j] = (int32_t)((uint32_t)u1[j] + (uint32_t)u2[j]);
        u1[N] = (int32_t)((uint32_t)u1[N] + (uint32_t)mu);
        orc_keyswitch(p, ksk, u1, out);
    } else if (op == ORC_MAJ || op == ORC_XOR3) {
        /* extension gates: lweAddTo / lweAddMulTo of THREE samples (the linear ops of SURVEY.md 8a a13), no constant, then
         * tfhe_bootstrap_FFT exactly as bootsAND ... bootsXOR do.  MAJ: a + b + c; XOR3: -2 (a + b + c) */
        const uint32_t s = op == ORC_MAJ ? 1u : (uint32_t)-2;
        for (int m = 0; m <= n; m++) t[m] = (int32_t)(s * ((uint32_t)ca[m] + (uint32_t)cb[m] + (uint32_t)cc[m]));
        orc_bootstrap(p, bkfft, ksk, t, mu, out);
    } else if (orc_gate_linear(p, op, ca, cb, t) == 0) {
        orc_bootstrap(p, bkfft, ksk, t, mu, out);
    } else {
        rc = -1;
    }
    free(t);
    return rc;
}

/* usable worker threads: min(cores, affinity mask, cgroup v2 quota) -- a container on a GPU box gets a
 * share of the host's cores and an oversubscribed OpenMP team spins instead of working */
int orc_max_threads(void)
{
#ifdef _OPENMP
    static int cached = 0;
    if (cached) return cached;
    int n = omp_get_num_procs();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            long long q = (quota + period - 1) / period;
            if (q < n) n = (int)q;
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) { /* cgroup v1 */
        long long quota = 0, period = 100000;
        if (fscanf(f, "%lld", &quota) != 1) quota = 0;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r"))) {
            if (fscanf(f, "%lld", &period) != 1) period = 100000;
            fclose(f);
        }
        if (quota > 0 && period > 0 && (quota + period - 1) / period < n) n = (int)((quota + period - 1) / period);
    }
    cached = n < 1 ? 1 : n;
    return cached;
#else
    return 1;
#endif
}

int orc_gate_batch(const orc_params *p, const double *bkfft, const int32_t *ksk, int op,
                   const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                   int32_t *out, size_t count, int nthreads)
{
    const size_t st = (size_t)p->n + 1;
    int bad = 0;
    if (nthreads <= 0 || nthreads > orc_max_threads()) nthreads = orc_max_threads();
#pragma omp parallel for schedule(dynamic, 1) reduction(| : bad) num_threads(nthreads)
    for (size_t g = 0; g < count; g++) {
        int o = ops ? (int)ops[g] : op;
        if (orc_gate(p, bkfft, ksk, o, in0 + g * st, in1 ? in1 + g * st : NULL,
                     in2 ? in2 + g * st : NULL, out + g * st))
            bad = 1;
    }
    return bad ? -1 : 0;
}
