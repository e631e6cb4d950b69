// kernels.hip.h -- hand-written HIP kernels (gfx950 / CDNA4) for the TFHE gate-bootstrapping path.
//
// Replaces, on the device, the upstream libtfhe CPU functions listed in SURVEY.md 8a:
//   k_prepare        bootsNAND... linear stage + modSwitchFromTorus32 loop of tfhe_bootstrap_woKS_FFT
//   k_blind_rotate   tfhe_blindRotateAndExtract_FFT / tfhe_blindRotate_FFT / tfhe_MuxRotate_FFT /
//                    tLweMulByXaiMinusOne / tGswTorus32PolynomialDecompH / IntPolynomial_ifft /
//                    tLweFFTAddMulRTo / tLweFromFFTConvert / tLweAddTo / tLweExtractLweSample
//   k_keyswitch      lweKeySwitch
//   k_fft_fwd_polys  tGswToFFTConvert (key load)
//
// Arithmetic contract (DESIGN.md "Canonical transform v3"): every floating-point operation below is a
// separately rounded IEEE-754 binary64 +, -, * or an explicit fma; the file MUST be compiled with
// -ffp-contract=off.  The data-flow graph is the oracle's radix-2 graph:
//   forward  evaluation tree of X^512 - i on c_j = p_j + i p_{j+512} (no twist pass; natural order in,
//            bit-reversed root index out); stage 0 is written on exact integer sums (4 fma per butterfly),
//            every later butterfly is (u + w v, u - w v) in the 6-operation fused form
//            a' = u + w v by 4 fma, b' = fma(2, u, -a');
//   inverse  decimation in time with conjugate twiddles, three register-constant stages as exact moves, then
//            the un-twist by conj(E[j]) and the conversion Torus32(int64(.)) (truncation, as upstream).
// Three radix-2 stages are executed per register pass (8 points per lane, one 512-point transform per wave64).
// The twiddles of a pass are 1 + 2 + 4 per lane, but the second of every sibling pair is i times the first
// (E[k + 512] = i E[k] holds exactly in the table), so a pass LOADS four and applies the other three through
// operand swaps and sign modifiers: bit-identical, 4 instead of 7 ds_read_b128 and 16 instead of 28 registers.
//
// Wave layouts of the 512 complex points (e = 9-bit index):
//   L0: reg r = e[8:6], lane = e[5:0]           (input of forward / output of inverse)
//   L1: reg r = e[5:3], lane = e[8:6]*8 + e[2:0]
//   L2: reg r = e[2:0], lane = e[8:3]           (spectrum layout; stored at sigma(e) = r*64 + lane)
// The two transposes go through a per-wave LDS scratch with conflict-free address maps f01 / f12.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace eoc {

typedef double d2 __attribute__((ext_vector_type(2))); // (re, im)

constexpr int kN = 1024;
constexpr int kNH = 512;
constexpr int kScr = 568;      // d2 elements of per-wave transpose scratch (f01 needs 72*7+64)
// twiddle tables in LDS (d2 entries), FOUR per lane and pass {A, B0, C0, C2}: the stage twiddles are
// A | B0, i B0 | C0, i C0, C2, i C2
constexpr int kTwF1 = 0;                    // forward pass 1 [4][8]  (by lane >> 3)
constexpr int kTwF2 = 32;                   // forward pass 2 [4][64] (by lane)
constexpr int kTwI1 = 32 + 256;             // inverse middle pass [4][8] (by lane & 7)
constexpr int kTwI0 = 32 + 256 + 32;        // inverse last pass [4][64] (by lane)
constexpr int kTwEntries = 32 + 256 + 32 + 256;

#define EOC_FMA(a, b, c) __builtin_fma((a), (b), (c))

// register constants of the forward transform's first pass: E[256] = (c, c), E[128], E[64], E[320]
// (checked against canon_twiddles.h when an engine is created)
#define EOC_SQRT_HALF 0x1.6a09e667f3bcdp-1
#define EOC_E128_RE 0x1.d906bcf328d46p-1
#define EOC_E128_IM 0x1.87de2a6aea963p-2
#define EOC_E64_RE 0x1.f6297cff75cb0p-1
#define EOC_E64_IM 0x1.8f8b83c69a60bp-3
#define EOC_E320_RE 0x1.1c73b39ae68c8p-1
#define EOC_E320_IM 0x1.a9b66290ea1a3p-1

__device__ __forceinline__ d2 cmulc(d2 a, d2 w)
{ // a * conj(w)
    d2 r;
    r.x = EOC_FMA(a.x, w.x, a.y * w.y);
    r.y = EOC_FMA(a.y, w.x, -(a.x * w.y));
    return r;
}

// --- radix-2 butterflies (u, v) -> (u + t v, u - t v), fused form ---------------------------------
#define EOC_BFLY_TAIL()                      \
    a = n;                                   \
    b.x = EOC_FMA(2.0, u.x, -n.x);           \
    b.y = EOC_FMA(2.0, u.y, -n.y)
__device__ __forceinline__ void ct_w(d2 &a, d2 &b, d2 w)
{ // t = w
    d2 u = a, n;
    n.x = EOC_FMA(-w.y, b.y, EOC_FMA(w.x, b.x, u.x));
    n.y = EOC_FMA(w.x, b.y, EOC_FMA(w.y, b.x, u.y));
    EOC_BFLY_TAIL();
}
__device__ __forceinline__ void ct_iw(d2 &a, d2 &b, d2 w)
{ // t = i w = (-w.im, w.re)
    d2 u = a, n;
    n.x = EOC_FMA(-w.x, b.y, EOC_FMA(-w.y, b.x, u.x));
    n.y = EOC_FMA(-w.y, b.y, EOC_FMA(w.x, b.x, u.y));
    EOC_BFLY_TAIL();
}
__device__ __forceinline__ void ct_wc(d2 &a, d2 &b, d2 w)
{ // t = conj(w)
    d2 u = a, n;
    n.x = EOC_FMA(w.y, b.y, EOC_FMA(w.x, b.x, u.x));
    n.y = EOC_FMA(w.x, b.y, EOC_FMA(-w.y, b.x, u.y));
    EOC_BFLY_TAIL();
}
__device__ __forceinline__ void ct_iwc(d2 &a, d2 &b, d2 w)
{ // t = conj(i w) = (-w.im, -w.re)
    d2 u = a, n;
    n.x = EOC_FMA(w.x, b.y, EOC_FMA(-w.y, b.x, u.x));
    n.y = EOC_FMA(-w.y, b.y, EOC_FMA(-w.x, b.x, u.y));
    EOC_BFLY_TAIL();
}
__device__ __forceinline__ void ct_1(d2 &a, d2 &b)
{ // t = 1 (register-constant stages of the inverse only)
    d2 u = a, v = b;
    a = u + v;
    b = u - v;
}
__device__ __forceinline__ void ct_ic(d2 &a, d2 &b)
{ // t = conj(i): t v = (v.y, -v.x)
    d2 u = a, t;
    t.x = b.y;
    t.y = -b.x;
    a = u + t;
    b = u - t;
}

// --- LDS address maps ---------------------------------------------------------------------------
// f12: XOR swizzle, conflict-free for ds_read_b128 (16-lane groups, 64 banks) AND ds_write_b128 (8-lane
// groups, 32 banks) in both the L1 and the L2 access pattern (brute-forced against the bank model of
// MI355X_MICROARCH.md)
__device__ __forceinline__ int f12(int e) { return e ^ (((e >> 3) & 7) | (((e >> 6) & 1) << 3)); }

// compiler-level ordering between a wave's own LDS writes and cross-lane reads (the hardware
// executes one wave's LDS operations in order, so no s_barrier is needed inside a wave)
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- forward transform, in pieces ---------------------------------------------------------------
// stage 0 on exact integer-valued doubles: u = (a, b), v = (p, q), dm = p - q, dp = p + q
__device__ __forceinline__ void fwd_stage0(d2 &lo, d2 &hi, double a, double b, double dm, double dp)
{
    lo.x = EOC_FMA(EOC_SQRT_HALF, dm, a);
    lo.y = EOC_FMA(EOC_SQRT_HALF, dp, b);
    hi.x = EOC_FMA(-EOC_SQRT_HALF, dm, a);
    hi.y = EOC_FMA(-EOC_SQRT_HALF, dp, b);
}
__device__ __forceinline__ void fwd_pass0_tail(d2 (&x)[8])
{ // stages 1, 2 (bits 7, 6): rho(1,0) = E[128], rho(1,1) = i E[128]; rho(2,.) = E[64], i E[64], E[320], i E[320]
    const d2 w1 = {EOC_E128_RE, EOC_E128_IM}, w2 = {EOC_E64_RE, EOC_E64_IM}, w3 = {EOC_E320_RE, EOC_E320_IM};
    ct_w(x[0], x[2], w1);
    ct_w(x[1], x[3], w1);
    ct_iw(x[4], x[6], w1);
    ct_iw(x[5], x[7], w1);
    ct_w(x[0], x[1], w2);
    ct_iw(x[2], x[3], w2);
    ct_w(x[4], x[5], w3);
    ct_iw(x[6], x[7], w3);
}
// the four loaded twiddles of a table-driven pass; `stride` = 8 (pass 1, q = table + (lane >> 3)) or 64 (pass 2)
__device__ __forceinline__ void tw_load(d2 (&t)[4], const d2 *q, int stride)
{
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = q[k * stride];
}
__device__ __forceinline__ void fwd_pass12(d2 (&x)[8], const d2 (&t)[4])
{ // three radix-2 stages on the 8 register points: twiddles A | B0, i B0 | C0, i C0, C2, i C2
    ct_w(x[0], x[4], t[0]);
    ct_w(x[1], x[5], t[0]);
    ct_w(x[2], x[6], t[0]);
    ct_w(x[3], x[7], t[0]);
    ct_w(x[0], x[2], t[1]);
    ct_w(x[1], x[3], t[1]);
    ct_iw(x[4], x[6], t[1]);
    ct_iw(x[5], x[7], t[1]);
    ct_w(x[0], x[1], t[2]);
    ct_iw(x[2], x[3], t[2]);
    ct_w(x[4], x[5], t[3]);
    ct_iw(x[6], x[7], t[3]);
}
// transposes: write in the source layout, read in the destination layout.  One wave's LDS operations
// execute in issue order, so a later write to the same scratch cannot overtake an earlier read.
__device__ __forceinline__ void t01_write(const d2 (&x)[8], d2 *scr, int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) scr[72 * r + lane] = x[r];
    wave_lds_fence();
}
__device__ __forceinline__ void t01_read(d2 (&x)[8], const d2 *scr, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = scr[72 * hi + 8 * r + lo];
    wave_lds_fence();
}
__device__ __forceinline__ void t12_write(const d2 (&x)[8], d2 *scr, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; r++) scr[f12(hi * 64 + r * 8 + lo)] = x[r];
    wave_lds_fence();
}
__device__ __forceinline__ void t12_read(d2 (&x)[8], const d2 *scr, int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = scr[f12(lane * 8 + r)];
    wave_lds_fence();
}

// one forward transform after its first pass: x[] in L0 (stages 0-2 done) -> x[] in L2.  The second twiddle set is
// requested under the second transpose (register-lean: this form runs next to the live spectra of a finished pair)
__device__ __forceinline__ void fft_fwd_rest(d2 (&x)[8], const d2 *tw, d2 *scr, int lane)
{
    d2 t1[4], t2[4];
    tw_load(t1, tw + kTwF1 + (lane >> 3), 8);
    t01_write(x, scr, lane);
    t01_read(x, scr, lane);
    fwd_pass12(x, t1);
    t12_write(x, scr, lane);
    tw_load(t2, tw + kTwF2 + lane, 64);
    t12_read(x, scr, lane);
    fwd_pass12(x, t2);
}
// EOC_SB: scheduling barriers that pin the phase order of the skewed pair below (the machine scheduler otherwise sinks
// the early twiddle reads under the transposes and waits for BOTH transposes before the first register pass).
// EOC_SGB: the stores of one transform's transpose are interleaved one by one with the other transform's register pass
// (sched_group_barrier pipelines): a ds_write_b128 holds the CU's LDS store path for about 13 cycles, and a wave that
// issues eight of them back to back is issue-blocked for all of them (SQ_WAIT_INST_LDS was 22 % of the wave cycles)
#define EOC_SB() __builtin_amdgcn_sched_barrier(0)
#define EOC_SGB(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
#define EOC_M_VALU 0x002
#define EOC_M_DSR 0x100
#define EOC_M_DSW 0x200
// T2 != nullptr: the last pass's four twiddles are resident in registers (loop-invariant, kept by the caller)
template <class MakeB>
__device__ __forceinline__ void fft_fwd_rest_x2(d2 (&xa)[8], d2 (&xb)[8], MakeB make_b, const d2 *tw, d2 *scr, int lane,
                                                const d2 (*T2)[4] = nullptr)
{
    d2 t1[4], t2[4];
    EOC_SB();
    // region B: first pass of b, a's stores spread through it; then a's reads
    tw_load(t1, tw + kTwF1 + (lane >> 3), 8);
    wave_lds_fence();
    t01_write(xa, scr, lane);
    make_b();
    t01_read(xa, scr, lane);
    EOC_SGB(EOC_M_DSR, 4);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 10);
    }
    EOC_SGB(EOC_M_VALU, 64);
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SB();
    // region C: second pass of a; b's stores spread through its first half, then b's reads and the last twiddle set
    t01_write(xb, scr, lane);
    t01_read(xb, scr, lane);
    if (T2) {
#pragma unroll
        for (int k = 0; k < 4; k++) t2[k] = (*T2)[k];
    } else {
        tw_load(t2, tw + kTwF2 + lane, 64);
    }
    wave_lds_fence();
    fwd_pass12(xa, t1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    if (T2) EOC_SGB(EOC_M_DSR, 8);
    else EOC_SGB(EOC_M_DSR, 12);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    // region D: second pass of b under a's second transpose
    t12_write(xa, scr, lane);
    t12_read(xa, scr, lane);
    fwd_pass12(xb, t1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    // region E: third pass of a under b's second transpose
    t12_write(xb, scr, lane);
    t12_read(xb, scr, lane);
    fwd_pass12(xa, t2);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    fwd_pass12(xb, t2);
}

// ---- inverse transform, in pieces: x[] in L2 -> x[] in L0 (before the un-twist); decimation in time,
// conjugate twiddles ---------------------------------------------------------------------------------
__device__ __forceinline__ void inv_pass2(d2 (&x)[8])
{ // stages 8,7,6 (bits 0,1,2): register constants; w = 1 and w = conj(i) are exact moves
    const d2 wc = {EOC_SQRT_HALF, EOC_SQRT_HALF}; // W[64]; W[192] = i W[64]
    ct_1(x[0], x[1]);
    ct_1(x[2], x[3]);
    ct_1(x[4], x[5]);
    ct_1(x[6], x[7]);
    ct_1(x[0], x[2]);
    ct_ic(x[1], x[3]);
    ct_1(x[4], x[6]);
    ct_ic(x[5], x[7]);
    ct_1(x[0], x[4]);
    ct_wc(x[1], x[5], wc);
    ct_ic(x[2], x[6]);
    ct_iwc(x[3], x[7], wc);
}
__device__ __forceinline__ void inv_pass10(d2 (&x)[8], const d2 (&t)[4])
{ // three DIT stages, twiddles conj of: m6 | m4, i m4 | m0, m1, i m0, i m1   (t = {m6, m4, m0, m1})
    ct_wc(x[0], x[1], t[0]);
    ct_wc(x[2], x[3], t[0]);
    ct_wc(x[4], x[5], t[0]);
    ct_wc(x[6], x[7], t[0]);
    ct_wc(x[0], x[2], t[1]);
    ct_iwc(x[1], x[3], t[1]);
    ct_wc(x[4], x[6], t[1]);
    ct_iwc(x[5], x[7], t[1]);
    ct_wc(x[0], x[4], t[2]);
    ct_wc(x[1], x[5], t[3]);
    ct_iwc(x[2], x[6], t[2]);
    ct_iwc(x[3], x[7], t[3]);
}
__device__ __forceinline__ void t21_read(d2 (&x)[8], const d2 *scr, int lane)
{
    const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = scr[f12(hi * 64 + r * 8 + lo)];
    wave_lds_fence();
}
__device__ __forceinline__ void t10_read(d2 (&x)[8], const d2 *scr, int lane)
{
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = scr[72 * r + lane];
    wave_lds_fence();
}
// inverse transform + un-twist factors: the twiddles of both table passes and the 8 un-twist factors are requested
// under the first transpose, so that no table read sits between a transpose read and its use.  The eight stores of each
// transpose leave in the order the last stage of the register pass completes its butterflies -- (0,4) (1,5) (2,6) (3,7) --
// and are spread through that stage (sched_group_barrier), as the forward pair does
__device__ __forceinline__ void fft_inv_wave(d2 (&x)[8], d2 (&ut)[8], const d2 *tw, const d2 *s_twist, d2 *scr, int lane,
                                             const d2 (*T1)[4] = nullptr)
{
    d2 t1[4], t0[4];
    EOC_SB();
    if (T1) {
#pragma unroll
        for (int k = 0; k < 4; k++) t1[k] = (*T1)[k];
    } else {
        tw_load(t1, tw + kTwI1 + (lane & 7), 8);
    }
    inv_pass2(x);
    {
        constexpr int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
#pragma unroll
        for (int q = 0; q < 8; q++) scr[f12(lane * 8 + ord[q])] = x[ord[q]];
        wave_lds_fence();
    }
    t21_read(x, scr, lane);
    EOC_SGB(EOC_M_VALU, 32);                                     // stages 8 and 7
    EOC_SGB(EOC_M_VALU, 4);  EOC_SGB(EOC_M_DSW, 2);              // (0,4): w = 1
    EOC_SGB(EOC_M_VALU, 6);  EOC_SGB(EOC_M_DSW, 2);              // (1,5)
    EOC_SGB(EOC_M_VALU, 4);  EOC_SGB(EOC_M_DSW, 2);              // (2,6): w = conj(i)
    EOC_SGB(EOC_M_VALU, 6);  EOC_SGB(EOC_M_DSW, 2);              // (3,7)
    EOC_SGB(EOC_M_DSR, 8);
    tw_load(t0, tw + kTwI0 + lane, 64);
#pragma unroll
    for (int r = 0; r < 8; r++) ut[r] = s_twist[lane + 64 * r];
    wave_lds_fence();
    EOC_SB();
    inv_pass10(x, t1);
    {
        constexpr int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int q = 0; q < 8; q++) scr[72 * hi + 8 * ord[q] + lo] = x[ord[q]];
        wave_lds_fence();
    }
    t10_read(x, scr, lane);
    EOC_SGB(EOC_M_VALU, 48);                                     // first two stages of the pass
#pragma unroll
    for (int k = 0; k < 4; k++) {
        EOC_SGB(EOC_M_VALU, 6);
        EOC_SGB(EOC_M_DSW, 2);
    }
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SB();
    inv_pass10(x, t0);
}

// Torus32(int64(v)) for |v| < 2^51: t = trunc(v) (exact), t + 1.5 * 2^52 is exact and carries t mod 2^32 in its
// low dword.  Two operations.  (An external product is bounded by 2 l N (Bg/2) 2^31 <= 2^52 in exact arithmetic and
// is statistically near 2^45; tests/ hold the oracle's high-water mark below 2^51.)
__device__ __forceinline__ uint32_t wrap_trunc(double v)
{
    double m = __builtin_trunc(v) + 6755399441055744.0;
    return (uint32_t)__double2loint(m);
}

// copy the two constant tables into LDS (called by all threads of the workgroup, followed by __syncthreads)
__device__ __forceinline__ void load_tables(d2 *s_tw, d2 *s_twist, const d2 *g_tw, const d2 *g_twist, int tid, int nthreads)
{
    for (int i = tid; i < kTwEntries; i += nthreads) s_tw[i] = g_tw[i];
    for (int i = tid; i < kNH; i += nthreads) s_twist[i] = g_twist[i];
}

// =================================================================================================
// K4 / debug: forward transform of `count` integer polynomials, one wave each
// =================================================================================================
// `scale` must be a power of two (exact): 1 for the plain transform, 2^-9 for the key image so that the
// inverse transform's 1/512 is already in the products (bit-identical to scaling at the end)
__global__ __launch_bounds__(256) void k_fft_fwd_polys(const int32_t *__restrict__ polys,
                                                        double *__restrict__ specs, size_t count,
                                                        const d2 *__restrict__ g_tw,
                                                        const d2 *__restrict__ g_twist, double scale)
{
    __shared__ d2 s_tw[kTwEntries];
    __shared__ d2 s_scr[4][kScr];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < kTwEntries; i += 256) s_tw[i] = g_tw[i];
    (void)g_twist;
    __syncthreads();
    size_t poly = (size_t)blockIdx.x * 4 + w;
    if (poly >= count) return;
    const int32_t *p = polys + poly * kN;
    d2 x[8];
#pragma unroll
    for (int r = 0; r < 4; r++) { // full-range torus inputs: the integer sums are formed in binary64 (exact, < 2^33)
        int j = lane + 64 * r;
        double a = (double)p[j], b = (double)p[j + kNH], pp = (double)p[j + 256], q = (double)p[j + 256 + kNH];
        fwd_stage0(x[r], x[r + 4], a, b, pp - q, pp + q);
    }
    fwd_pass0_tail(x);
    fft_fwd_rest(x, s_tw, s_scr[w], lane);
    d2 *o = reinterpret_cast<d2 *>(specs) + poly * kNH;
#pragma unroll
    for (int r = 0; r < 8; r++) o[r * 64 + lane] = x[r] * scale;
}

__global__ __launch_bounds__(256) void k_fft_inv_polys(const double *__restrict__ specs,
                                                        int32_t *__restrict__ polys, size_t count,
                                                        const d2 *__restrict__ g_tw,
                                                        const d2 *__restrict__ g_twist)
{
    __shared__ d2 s_tw[kTwEntries];
    __shared__ d2 s_twist[kNH];
    __shared__ d2 s_scr[4][kScr];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    load_tables(s_tw, s_twist, g_tw, g_twist, tid, 256);
    __syncthreads();
    size_t poly = (size_t)blockIdx.x * 4 + w;
    if (poly >= count) return;
    const d2 *in = reinterpret_cast<const d2 *>(specs) + poly * kNH;
    d2 x[8];
#pragma unroll
    for (int r = 0; r < 8; r++) x[r] = in[r * 64 + lane];
    d2 ut[8];
    fft_inv_wave(x, ut, s_tw, s_twist, s_scr[w], lane);
    int32_t *p = polys + poly * kN;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int j = lane + 64 * r;
        d2 y = cmulc(x[r], ut[r] * 0x1p-9);
        p[j] = (int32_t)wrap_trunc(y.x);
        p[j + kNH] = (int32_t)wrap_trunc(y.y);
    }
}

// =================================================================================================
// K1: gate linear stage + mod-switch.   t = (0,cst) + s0*in0 + s1*in1  ->  bara/barb in [0, 2N)
// =================================================================================================
struct GateDesc {
    int32_t op;        // eoc_op, or OP_RAW
    uint32_t job_base; // first blind-rotate job of this gate (jobs are [variant][instance])
    const int32_t *in0, *in1, *in2;
    int32_t *out;
};
constexpr int OP_MUX = 10, OP_NOT = 11, OP_COPY = 12, OP_CONST0 = 13, OP_CONST1 = 14, OP_RAW = 100;
// extension gates (include/eoc_tfhe_gpu.h): one bootstrap behind a linear stage over THREE operands (in0, in1, in2).  Their
// rotation amounts come from k_prepare (a launch of its own): the blind rotation's folded prologue stays the two-operand
// code it was, so the hot kernels' ISA does not change for them
constexpr int OP_MAJ = 15, OP_XOR3 = 16;
__device__ __forceinline__ int gate_lin3(int op) { return op == OP_MAJ ? 1 : (op == OP_XOR3 ? -2 : 0); } // sign of ALL three
// OP_MULTI (internal, mixed batches): one gate descriptor over rows of DIFFERENT two-input opcodes (0..9) -- only the
// linear stage differs between them, so the whole opcode-sorted block runs as one level (full launches instead of one
// partly filled launch per opcode).  `in2` then points at one 32-bit word per row whose top four bits are the row's
// opcode (the gather permutation: original index | op << 28).
constexpr int OP_MULTI = 101;
constexpr uint32_t kPermIndexMask = 0x0FFFFFFFu;
__device__ __forceinline__ int desc_op(const GateDesc &d, uint32_t inst)
{
    return d.op == OP_MULTI ? (int)(reinterpret_cast<const uint32_t *>(d.in2)[inst] >> 28) : d.op;
}

__device__ __forceinline__ void gate_lin(int op, int &cst8, int &s0, int &s1)
{
    // (cst in eighths, s0, s1) -- SURVEY.md 8a a1
    switch (op) {
    case 0: cst8 = 1; s0 = -1; s1 = -1; break;  // NAND
    case 1: cst8 = -1; s0 = 1; s1 = 1; break;   // AND
    case 2: cst8 = 1; s0 = 1; s1 = 1; break;    // OR
    case 3: cst8 = -1; s0 = -1; s1 = -1; break; // NOR
    case 4: cst8 = 2; s0 = 2; s1 = 2; break;    // XOR
    case 5: cst8 = -2; s0 = -2; s1 = -2; break; // XNOR
    case 6: cst8 = -1; s0 = -1; s1 = 1; break;  // ANDNY
    case 7: cst8 = -1; s0 = 1; s1 = -1; break;  // ANDYN
    case 8: cst8 = 1; s0 = -1; s1 = 1; break;   // ORNY
    case 9: cst8 = 1; s0 = 1; s1 = -1; break;   // ORYN
    default: cst8 = 0; s0 = 1; s1 = 0; break;   // RAW: t = in0
    }
}

// grid: x = jobs of the widest gate (S or 2S; the large dimension goes on x), y = ceil((n+1)/256),
//       z = gates of this level
__global__ __launch_bounds__(256) void k_prepare(const GateDesc *__restrict__ descs, int n, uint32_t S,
                                                 uint16_t *__restrict__ bara, int bara_stride)
{
    const GateDesc d = descs[blockIdx.z];
    const uint32_t y = blockIdx.x;
    const uint32_t variant = y / S, s = y - variant * S;
    if (variant >= (d.op == OP_MUX ? 2u : 1u)) return;
    const int m = blockIdx.y * 256 + threadIdx.x;
    if (m > n) return;
    int op = desc_op(d, s);
    const int32_t *a = d.in0, *b = d.in1;
    if (d.op == OP_MUX) { // u1 = AND(a,b), u2 = ANDNY(a,c)
        op = variant ? 6 : 1;
        b = variant ? d.in2 : d.in1;
    }
    int cst8, s0, s1;
    gate_lin(op, cst8, s0, s1);
    const size_t off = (size_t)s * (n + 1) + m;
    uint32_t t;
    if (const int s3 = gate_lin3(op)) // MAJ: a + b + c; XOR3: -2 (a + b + c); no constant
        t = (uint32_t)s3 * ((uint32_t)a[off] + (uint32_t)b[off] + (uint32_t)d.in2[off]);
    else {
        t = (uint32_t)s0 * (uint32_t)a[off];
        if (s1) t += (uint32_t)s1 * (uint32_t)b[off];
        if (m == n) t += (uint32_t)cst8 << 29;
    }
    // modSwitchFromTorus32(t, 2N), N = 1024: round(t * 2048 / 2^32) mod 2048
    bara[(size_t)(d.job_base + y) * bara_stride + m] = (uint16_t)(((t + (1u << 20)) >> 21) & 2047u);
}

// The folded prologue writes the job's row of rotation amounts with vector stores; the step loop reads it back.
// SHIPPED FORM (SABAR = false): behind the workgroup barrier the row is copied ONCE into LDS (vector loads: writer and
// reader share the CU's vector L1 and the barrier carries a workgroup-scope release / acquire -- inside the formal memory
// model) and every step reads its amount with one ds_read_u16 of a wave-uniform address + v_readfirstlane.  VERDICT r5 asked
// for vector loads + v_readfirstlane in the loop (measured in round 5: +0.3 % Set A / +0.6 % Set B); that form keeps the
// loaded word and its address in VGPRs across a step, which the 256-register kernels do not have (the run-time-base gadget-
// length-3 instance spilled 16 registers, the wide kernel up to 40); the LDS copy needs neither (register counts equal to
// the scalar form's) and puts one extra LDS read per step on a pipe that serves ~250.
// SCALAR FORM (SABAR = true, EOC_TFHE_SCALAR_ABAR=1): scalar loads through the constant address space.  The scalar cache is
// not coherent with vector stores: the vector L1 is write-through, a store is complete -- vmcnt decremented -- when this
// XCD's L2 has it; the scalar cache fills from that same L2 and is emptied by s_dcache_inv behind the barrier.  So:
// compiler-level release fence, an explicit s_waitcnt vmcnt(0) (a workgroup-scope fence alone emits none outside
// threadgroup-split mode), the workgroup barrier, s_dcache_inv, and s_waitcnt lgkmcnt(0) so that no later s_load can be
// issued past the invalidate (scalar memory operations may complete out of order).  Rows are bara_stride =
// a multiple of 32 entries = 64 bytes apart, so two jobs never share a scalar-cache line.  An agent-scope release would
// also be correct but emits buffer_wbl2 -- an L2 write-back per workgroup (+18 us per 1024-job launch).  This form relies
// on cache behaviour outside the memory model; it is kept as a measured alternative, not the default.
// EOC_TFHE_NO_FOLD=1 restores the separate k_prepare launch (a kernel boundary orders everything) for either form.
__device__ __forceinline__ void eoc_row_stores_to_l2()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// scalar form only, behind the workgroup barrier: drop the scalar cache's (possibly stale) lines and let nothing pass
__device__ __forceinline__ void eoc_scalar_cache_acquire()
{
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// the same for ONE job inside the blind rotation (levels without MUX): thread `t` of `nt` writes entries t, t + nt, ... of
// the job's row; the caller orders the stores before its loads of the row (the workgroup barrier; the scalar form adds
// eoc_row_stores_to_l2 in front of it and eoc_scalar_cache_acquire behind it)
__device__ __forceinline__ void prepare_row(const GateDesc &d, uint32_t inst, int n, uint16_t *row, int t, int nt)
{
    int cst8, s0, s1;
    gate_lin(desc_op(d, inst), cst8, s0, s1);
    typedef const __attribute__((address_space(1))) int32_t *gi32p; // operand rows are global memory (device or mapped host)
    const gi32p a = (gi32p)(uintptr_t)(d.in0 + (size_t)inst * (n + 1));
    const gi32p b = s1 ? (gi32p)(uintptr_t)(d.in1 + (size_t)inst * (n + 1)) : a; // one-operand forms carry no second row
    for (int m = t; m <= n; m += nt) {
        uint32_t v = (uint32_t)s0 * (uint32_t)a[m];
        if (s1) v += (uint32_t)s1 * (uint32_t)b[m];
        if (m == n) v += (uint32_t)cst8 << 29;
        row[m] = (uint16_t)(((v + (1u << 20)) >> 21) & 2047u);
    }
}

// NOT / COPY: no bootstrap.  grid: x = ceil(S*(n+1)/256), y = gates
__global__ __launch_bounds__(256) void k_free_gates(const GateDesc *__restrict__ descs, size_t total, int rowlen, int32_t mu)
{
    const GateDesc d = descs[blockIdx.y];
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (d.op >= OP_CONST0) { // bootsCONSTANT: (0, ..., 0, +-mu), no input
        const bool is_b = (int)(i % (size_t)rowlen) == rowlen - 1;
        d.out[i] = is_b ? (d.op == OP_CONST1 ? mu : (int32_t)(0u - (uint32_t)mu)) : 0;
        return;
    }
    uint32_t v = (uint32_t)d.in0[i];
    d.out[i] = (int32_t)(d.op == OP_NOT ? 0u - v : v);
}


// mixed batches in arbitrary opcode order: rows are gathered into opcode-sorted order, evaluated run by
// run, and scattered back.  perm[i] = original index of the i-th gate in sorted order.
// grid: x = rows, y = ceil(rowlen / 256)
__global__ __launch_bounds__(256) void k_gather_rows(const int32_t *__restrict__ src, int32_t *__restrict__ dst,
                                                     const uint32_t *__restrict__ perm, int rowlen, int scatter)
{
    const int m = blockIdx.y * 256 + threadIdx.x;
    if (m >= rowlen) return;
    const size_t i = blockIdx.x, j = perm[i] & kPermIndexMask; // the top four bits carry the row's opcode (OP_MULTI)
    if (scatter) dst[j * rowlen + m] = src[i * rowlen + m];
    else dst[i * rowlen + m] = src[j * rowlen + m];
}


// Wave-priority alternation between the two waves sharing a SIMD (see the step loop): one step per phase, the
// later-placed wave holds the high priority kPrioDuty sixteenths of the steps.  The host enables it per launch
// (BRArgs::prio_duty).  Everything that was measured and rejected on this kernel is recorded in DESIGN.md 5.1; the
// timing-only ablations are patches under tools/patches/ (tools/variants.sh applies them to a scratch copy).
constexpr int kPrioDuty = 11;

// In-kernel stamps (diagnostic build only, -DEOC_STAMPS): per-wave cycle shares of the step's
// segments.  Never enabled in the shipped library; the values leave through a buffer of their own.
#ifdef EOC_STAMPS
#define EOC_STAMP(k)                                                                      \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long _t;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                \
        st_acc[k] += _t - st_prev;                                                        \
        st_prev = _t;                                                                     \
    } while (0)
#else
#define EOC_STAMP(k) do { } while (0)
#endif

// =================================================================================================
// K2: blind rotate + sample extract.  One workgroup = ONE ciphertext = a wave pair (h = 0,1), four workgroups per
// CU (35.6 KB of LDS each), so that s_barrier couples only the two waves that exchange data (with two ciphertexts per
// workgroup the barriers cost 6 % of the launch).  Wave h owns accumulator polynomial h, decomposes it, runs the l forward
// transforms of its digits, multiplies them by rows (h, p) of BK_i for the OTHER output polynomial and hands
// that partial chain to its partner through LDS; it then continues the chain it received from the partner with
// its own digits (rows (h, p), output polynomial h), runs the inverse transform of the sum and updates ACC_h.
// Accumulation order of output polynomial c (canonical, oracle/tfhe_oracle.c): terms (q_in = 1 - c, p = 1..l)
// first, then (q_in = c, p = 1..l); first term a product, every later one four fused multiply-adds.
// =================================================================================================
struct BRArgs {
    const double *bkfft;  // [n][2l][2][512] complex, bin order sigma, scaled by 2^-9
    uint16_t *bara;       // [jobs][stride], entry n = barb (written by k_prepare, or by this kernel's prologue: `prep`)
    int32_t *u;           // [jobs][N+1]
    uint32_t njobs;
    int n, Bgbit, bara_stride;
    int32_t mu;
    unsigned long long *stamps; // diagnostic build only: [waves][16] cycle sums per segment
    int prio_duty;              // < 0: leave wave priorities alone; else see the loop (single-round launches)
    // folded key-switch set-up (levels without MUX): the epilogue writes the key-switch operand and the output row
    // itself, k_ks_init is not launched and the extracted sample never goes to memory
    const GateDesc *ks_descs;   // non-null = fold; gate of job j is j / ks_S (jobs are [gate][instance])
    GateDesc desc0;             // single-gate levels (the plain batch call): the descriptor travels as a kernel argument,
    int inline_desc;            // `ks_descs` is then a non-null dummy and no descriptor copy precedes the launch
    int prep;                   // fold only: the prologue derives the job's rotation amounts from the gate's operand rows
                                // itself (bootsNAND... linear stage + modSwitchFromTorus32), k_prepare is not launched
    uint32_t *ubar;             // [jobs][N], row-major: the epilogue's stores are lane-contiguous (256 B per instruction)
    uint32_t ks_S, ks_prec_offset, job0; // job0: first job of this launch within the level
    // step range of this launch: the n steps of a blind rotation may be cut into consecutive launches; every boundary
    // brings all workgroups of the chip back to the same step (the accumulator travels through `acc_state`)
    int step_begin, step_end;
    int32_t *acc_state;         // [jobs][2][N], only used when the range is a proper part of [0, n)
};

// LDS: the two tables + one 9 KB scratch per wave.  The accumulator lives in registers (racc[16]); the scratch holds it
// as a signed 2N-periodic image ext[k] = ACC[k], ext[k + N] = -ACC[k] (8 KB) only for the sample extraction after the
// last step, which reads it by index.
// (X^abar - 1) * ACC without an accumulator image: coefficient lane + 64 r of ACC lives in racc[r];
// entry k of the signed 2N-periodic extension, k = lane' + 64 r' (r' < 32), is +racc[r'] (r' < 16) or -racc[r' - 16] of
// lane lane'.  Entry (lane + 64 r) - abar with abar = 64 Q + s is lane (lane - s) mod 64, register r - Q - (lane < s):
// one ds_bpermute_b32 per register moves the lanes, the register shift Q is wave-uniform and selects one of 32
// compile-time renamings of the block below.
template <int Q>
__device__ __forceinline__ void rot_digits(const uint32_t (&t)[16], const uint32_t (&racc)[16], bool borrow, uint32_t offset,
                                           uint32_t (&d)[16])
{
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int k0 = (r - Q) & 31, k1 = (r - Q - 1) & 31;
        const uint32_t a0 = t[k0 & 15], a1 = t[k1 & 15];
        const uint32_t A = offset - racc[r];
        const bool n0 = k0 >= 16, n1 = k1 >= 16;
        if (n0 == n1) {
            const uint32_t w = borrow ? a1 : a0;
            d[r] = n0 ? A - w : A + w;
        } else {
            const uint32_t d0 = n0 ? A - a0 : A + a0, d1 = n1 ? A - a1 : A + a1;
            d[r] = borrow ? d1 : d0;
        }
    }
}
constexpr int kAbarLds = 2048;                                  // one job's rotation amounts (n + 1 <= 1024 entries of 16 bits)
constexpr int kBRLds = (kTwEntries + kNH + 2 * kScr) * 16 + kAbarLds; // 37 632 bytes: four workgroups per CU

// BGBIT > 0: gadget base known at compile time (digit extraction becomes one bit-field extract); 0: run time.
// Register budget (hipcc 7.2, -Rpass-analysis=kernel-resource-usage): <2,10> 250 VGPRs, <3,7> 256, <1> 162, <2> 254, <3> 256,
// none of them spills.  Gadget length 4 (no default parameter set uses it) is a SLOW CORRECTNESS PATH: four live spectra
// exceed the 256 registers of a wave at two waves per SIMD, its kernel spills (324 VGPRs to scratch) and runs several
// times slower per transform; it is bit-exact (tests/test_gpu_parity.py) and nothing else is claimed for it.
template <int L, int BGBIT = 0, bool SABAR = false>
__global__ __launch_bounds__(128, 2) void k_blind_rotate(BRArgs A, const d2 *__restrict__ g_tw,
                                                         const d2 *__restrict__ g_twist)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    d2 *s_tw = reinterpret_cast<d2 *>(smem);
    d2 *s_twist = s_tw + kTwEntries;
    d2 *s_scr_all = s_twist + kNH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    d2 *scr = s_scr_all + h * kScr;
    d2 *scr_partner = s_scr_all + (h ^ 1) * kScr;
    int32_t *ext = reinterpret_cast<int32_t *>(scr); // [2N] signed periodic image of ACC_h (between steps)

    const uint32_t job = blockIdx.x; // grid = number of jobs
    // the rotation amounts of this job are wave-uniform and constant during the kernel: read as dwords (rows are 64-byte
    // aligned: bara_stride is a multiple of 32 entries) by vector loads of a uniform address, or (SABAR) by scalar loads
    // through the constant address space
    typedef const __attribute__((address_space(4))) uint32_t *cu32p;
    unsigned long long bara_addr = (unsigned long long)(uintptr_t)(A.bara + (size_t)job * A.bara_stride);

    load_tables(s_tw, s_twist, g_tw, g_twist, tid, 128);
    if (A.prep && A.step_begin == 0) { // folded k_prepare: this workgroup's row of rotation amounts
        const uint32_t gjob = A.job0 + job, g = gjob / A.ks_S;
        prepare_row(A.inline_desc ? A.desc0 : A.ks_descs[g], gjob - g * A.ks_S, A.n, A.bara + (size_t)job * A.bara_stride, tid, 128);
        if constexpr (SABAR) eoc_row_stores_to_l2();
        __syncthreads();
        if constexpr (SABAR) eoc_scalar_cache_acquire();
    }
    // loads through the constant address space may be moved freely by the compiler (the memory is assumed invariant): the
    // row's address is made opaque HERE, behind the prologue that may just have written the row, so that no load of it can
    // be scheduled above this point
    asm volatile("" : "+s"(bara_addr));
    const cu32p bara32 = (cu32p)bara_addr;
    // shipped form: the row is copied into LDS once (vector loads behind the barrier above: the workgroup's own stores,
    // or an earlier kernel's) and every step reads its amount from there
    uint16_t *s_abar = reinterpret_cast<uint16_t *>(s_scr_all + 2 * kScr);
    if constexpr (!SABAR) {
        const uint32_t *bara_v = reinterpret_cast<const uint32_t *>(A.bara + (size_t)job * A.bara_stride);
        for (int m = tid; m < (A.n + 2) / 2; m += 128) reinterpret_cast<uint32_t *>(s_abar)[m] = bara_v[m];
        __syncthreads();
    }
    auto load_abar = [&](int idx) __attribute__((always_inline)) {
        if constexpr (SABAR) return (int)((bara32[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu);
        else return (int)s_abar[idx];
    };

    // ACC = (0, X^(2N - barb) * testvect), testvect = (mu, ..., mu)
    uint32_t racc[16]; // register copy of ACC_h: coefficient lane + 64 r in racc[r] (r < 8), lane + 64 r + 512 in racc[8 + r]
    {
        const int barb = load_abar(A.n);
        const int rot = (2 * kN - barb) & (2 * kN - 1);
        const int32_t *st = A.acc_state + ((size_t)job * 2 + h) * kN;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int j = lane + 64 * (r & 7) + (r >> 3) * kNH;
            int idx = (j - rot) & (2 * kN - 1);
            int32_t v = (idx & kN) ? -A.mu : A.mu;
            v = h ? v : 0;
            if (A.step_begin > 0) v = st[j]; // continue a blind rotation started by an earlier launch
            racc[r] = (uint32_t)v;
        }
    }
    __syncthreads();

    const int Bgbit = BGBIT > 0 ? BGBIT : A.Bgbit;
    const uint32_t Bg = 1u << Bgbit, maskBg = Bg - 1, halfBg = Bg >> 1;
    uint32_t offset = 0;
#pragma unroll
    for (int p = 1; p <= L; p++) offset += halfBg << (32 - p * Bgbit);
    constexpr int KPL = 2 * L;
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.bkfft), 0, (int)((size_t)A.n * KPL * 2 * kNH * 16), 0x00020000);
    // digits arrive biased, u = digit + Bg/2 in [0, Bg); as_double(2^52 | u) - (2^52 + Bg/2) is the digit, exactly;
    // the stage-0 sums p - q and p + q are formed on the biased integers and converted the same way
    const double bias1 = 4503599627370496.0 + (double)halfBg, bias2 = 4503599627370496.0 + (double)Bg;

#ifdef EOC_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
    st_acc[12] = st_prev;                                          // loop entry time
    st_acc[14] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID: CU / SE / SIMD / wave slot
    st_acc[13] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
#endif
    const int prio_slot = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)); // HW_ID.WAVE_ID: slot on the SIMD
    // gadget length 2 has registers to spare: the forward transform's last four twiddles stay resident
    constexpr bool kResT2 = L == 2;
    d2 res_t2[4];
    if constexpr (kResT2) tw_load(res_t2, s_tw + kTwF2 + lane, 64);
    constexpr bool kResT1 = L == 2; // ... and the inverse transform's first table pass
    d2 res_t1[4];
    if constexpr (kResT1) tw_load(res_t1, s_tw + kTwI1 + (lane & 7), 8);
    int abar_next = load_abar(A.step_begin);
    for (int i = A.step_begin; i < A.step_end; i++) {
        EOC_STAMP(15);
        // The two waves that share a SIMD belong to different workgroups, and the issue arbiter favours the older
        // one: in a launch that exactly fills the chip (1024 gates = 512 workgroups) the first half of the grid
        // finishes at 0.82x and the second half at 1.19x of the mean, and the launch lasts as long as its slowest
        // workgroup.  The two waves occupy different wave slots, so the slot parity tells them apart: the
        // later-placed (odd) one holds the high priority A.prio_duty sixteenths of the steps, the other one the
        // rest.  12/16 makes both halves finish together (-10 % on the launch).  Launches of several rounds are
        // faster WITHOUT it (the arbiter's run-to-completion bias suits them: +4 %), so the host passes a
        // negative duty there.
        if (A.prio_duty >= 0) {
            const bool first_part = (i & 15) < A.prio_duty;
            if (first_part == ((prio_slot & 1) != 0))
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        } else if (A.prio_duty <= -2) {
            // launches of several rounds: the co-resident waves are at unrelated steps, so the alternation is taken
            // from the shader clock both of them read (phase length 2^(-prio_duty) cycles), even shares
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if ((int)((now >> (-A.prio_duty)) & 1) == (prio_slot & 1))
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        }
        const int abar = __builtin_amdgcn_readfirstlane(abar_next);
        // one step ahead (entry n is barb: always in bounds): an LDS read that retires with the rotation's ds_bpermutes, or
        // (SABAR) an s_load that retires with the key rows
        abar_next = load_abar(i + 1);
        // (X^abar - 1) * ACC_h.  abar == 0 gives an all-zero polynomial, all-zero digits and an exact
        // zero update, which is what skipping the step (as libtfhe does) amounts to.
        // rows (h, p), p = 1..L, of BK_i by buffer loads: the row's byte offset is wave-uniform (SGPR), the lane part one
        // loop-invariant VGPR
        auto load_row = [&](int p, int c, d2 (&b)[8]) __attribute__((always_inline)) {
            const uint32_t row_off = (uint32_t)((((size_t)i * KPL + h * L) * 2 + (size_t)(p - 1) * 2 + c) * kNH * 16);
#pragma unroll
            for (int r = 0; r < 8; r++)
                b[r] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane * 16, (int)(row_off + r * 1024), 0));
        };
        d2 ra[8], rb[8];
        uint32_t dlo[8], dhi[8];
        {
            const int s = abar & 63;
            const int src = ((lane - s) & 63) << 2;
            uint32_t t[16], d[16];
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)racc[r]);
            const bool borrow = lane < s;
#define EOC_RQ(q) case q: rot_digits<q>(t, racc, borrow, offset, d); break;
            switch (abar >> 6) {
                EOC_RQ(0) EOC_RQ(1) EOC_RQ(2) EOC_RQ(3) EOC_RQ(4) EOC_RQ(5) EOC_RQ(6) EOC_RQ(7)
                EOC_RQ(8) EOC_RQ(9) EOC_RQ(10) EOC_RQ(11) EOC_RQ(12) EOC_RQ(13) EOC_RQ(14) EOC_RQ(15)
                EOC_RQ(16) EOC_RQ(17) EOC_RQ(18) EOC_RQ(19) EOC_RQ(20) EOC_RQ(21) EOC_RQ(22) EOC_RQ(23)
                EOC_RQ(24) EOC_RQ(25) EOC_RQ(26) EOC_RQ(27) EOC_RQ(28) EOC_RQ(29) EOC_RQ(30)
                default: rot_digits<31>(t, racc, borrow, offset, d); break;
            }
#undef EOC_RQ
#pragma unroll
            for (int r = 0; r < 8; r++) {
                dlo[r] = d[r];
                dhi[r] = d[8 + r];
            }
        }
        EOC_STAMP(0);
        // digit p of the 16 coefficients of this lane, first pass of its forward transform (stages 0-2)
        auto make_x0 = [&](int p, d2 (&x)[8]) __attribute__((always_inline)) {
            const int shift = 32 - p * Bgbit;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ua = (dlo[r] >> shift) & maskBg, ub = (dhi[r] >> shift) & maskBg;
                const uint32_t up = (dlo[r + 4] >> shift) & maskBg, uq = (dhi[r + 4] >> shift) & maskBg;
                const double a = __hiloint2double(0x43300000, (int)ua) - bias1;
                const double b = __hiloint2double(0x43300000, (int)ub) - bias1;
                const double dm = __hiloint2double(0x43300000, (int)(up - uq + Bg)) - bias2;
                const double dp = __hiloint2double(0x43300000, (int)(up + uq)) - bias2;
                fwd_stage0(x[r], x[r + 4], a, b, dm, dp);
            }
            fwd_pass0_tail(x);
        };
        auto mac = [&](bool first, const d2 (&x)[8], const d2 (&b)[8], d2 (&acc_)[8]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                if (first) {
                    acc_[r].x = EOC_FMA(-x[r].y, b[r].y, x[r].x * b[r].x);
                    acc_[r].y = EOC_FMA(x[r].y, b[r].x, x[r].x * b[r].y);
                } else {
                    acc_[r].x = EOC_FMA(-x[r].y, b[r].y, EOC_FMA(x[r].x, b[r].x, acc_[r].x));
                    acc_[r].y = EOC_FMA(x[r].y, b[r].x, EOC_FMA(x[r].x, b[r].y, acc_[r].y));
                }
            }
        };
        // forward transforms of the l digits (two at a time, skewed on the one scratch; an odd last one alone) and
        // the chain for the partner's output polynomial.  The spectra stay in registers for the own chain below.
        d2 xs[L][8], S[8];
#pragma unroll
        for (int p0 = 0; p0 + 1 < L; p0 += 2) {
            load_row(p0 + 1, 1 - h, ra);
            load_row(p0 + 2, 1 - h, rb);
            make_x0(p0 + 1, xs[p0]);
            EOC_STAMP(1);
            fft_fwd_rest_x2(xs[p0], xs[p0 + 1], [&]() __attribute__((always_inline)) { make_x0(p0 + 2, xs[p0 + 1]); },
                            s_tw, scr, lane, kResT2 ? &res_t2 : nullptr);
            EOC_STAMP(2);
            mac(p0 == 0, xs[p0], ra, S);
            mac(false, xs[p0 + 1], rb, S);
            EOC_STAMP(3);
        }
        if constexpr ((L & 1) != 0) {
            load_row(L, 1 - h, ra);
            make_x0(L, xs[L - 1]);
            EOC_STAMP(1);
            fft_fwd_rest(xs[L - 1], s_tw, scr, lane);
            EOC_STAMP(2);
            mac(L == 1, xs[L - 1], ra, S);
            EOC_STAMP(3);
        }
        // own rows: the first two are requested before the exchange (requesting the first one a register pass
        // earlier into a third buffer, or the second one only after the exchange, changes nothing: measured)
        load_row(1, h, ra);
        if constexpr (L >= 2) load_row(2, h, rb);
#pragma unroll
        for (int r = 0; r < 8; r++) scr[r * 64 + lane] = S[r];
        EOC_STAMP(4);
        __syncthreads();
        EOC_STAMP(5);
#pragma unroll
        for (int r = 0; r < 8; r++) S[r] = scr_partner[r * 64 + lane]; // the chain of the other input polynomial
        mac(false, xs[0], ra, S);
        if constexpr (L >= 3) load_row(3, h, ra);
        if constexpr (L >= 2) mac(false, xs[1], rb, S);
        if constexpr (L >= 4) load_row(4, h, rb);
        if constexpr (L >= 3) mac(false, xs[2], ra, S);
        if constexpr (L >= 4) mac(false, xs[3], rb, S);
        EOC_STAMP(6);
        __syncthreads(); // the partner has read this wave's scratch before the inverse transform overwrites it
        EOC_STAMP(7);
        d2 ut[8];
        fft_inv_wave(S, ut, s_tw, s_twist, scr, lane, kResT1 ? &res_t1 : nullptr);
        EOC_STAMP(8);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            d2 y = cmulc(S[r], ut[r]); // 1/512 is in the key image
            racc[r] += wrap_trunc(y.x);
            racc[8 + r] += wrap_trunc(y.y);
        }
        wave_lds_fence();
        EOC_STAMP(9);
    }
    if (A.step_end >= A.n) { // the sample extraction below reads the image: written once, after the last step
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = lane + 64 * r;
            ext[j] = (int32_t)racc[r];
            ext[j + kN] = (int32_t)(0u - racc[r]);
        }
        wave_lds_fence();
    }
#ifdef EOC_STAMPS
    st_acc[11] = __builtin_amdgcn_s_memtime(); // loop exit time
    if (A.stamps && lane == 0)
        for (int k = 0; k < 16; k++) A.stamps[((size_t)blockIdx.x * 2 + h) * 16 + k] = st_acc[k];
#endif

    // the lane index again, from the hardware (not from threadIdx): nothing lane-derived then has to stay live across
    // the step loop just for these stores
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (A.step_end < A.n) { // not the last part: park the accumulator for the next launch
        int32_t *st = A.acc_state + ((size_t)job * 2 + h) * kN;
#pragma unroll
        for (int r = 0; r < 16; r++) st[lane_e + 64 * (r & 7) + (r >> 3) * kNH] = (int32_t)racc[r];
        return;
    }
    // tLweExtractLweSample, index 0: u_0 = ACC_0[0], u_j = -ACC_0[N - j] = ext[2N - j]; b = ACC_1[0]
    if (A.ks_descs) { // + lweKeySwitch set-up: ubar_j = u_j + 2^(31 - t basebit), out = (0, ..., 0, b)
        const uint32_t gjob = A.job0 + job;
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int j = lane + 64 * r;
                A.ubar[(size_t)gjob * kN + j] = (uint32_t)ext[(2 * kN - j) & (2 * kN - 1)] + A.ks_prec_offset;
            }
        } else {
            const uint32_t g = gjob / A.ks_S, si = gjob - g * A.ks_S;
            int32_t *o = (A.inline_desc ? A.desc0.out : A.ks_descs[g].out) + (size_t)si * (A.n + 1);
            for (int m = lane; m < A.n; m += 64) o[m] = 0;
            if (lane == 0) o[A.n] = ext[0];
        }
    } else {
        int32_t *u = A.u + (size_t)job * (kN + 1);
        if (h == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int j = lane + 64 * r;
                u[j] = ext[(2 * kN - j) & (2 * kN - 1)];
            }
        } else if (lane == 0) {
            u[kN] = ext[0];
        }
    }
}

// =================================================================================================
// K2w: blind rotate + sample extract, WIDE form (gadget length 2): ONE WAVE = ONE CIPHERTEXT.  For launches of at
// least twice the pair kernel's resident set, two waves per SIMD can come from two ciphertexts instead of one wave
// pair: the wave owns both accumulator polynomials, runs the four forward transforms of a step as two skewed pairs and
// the two inverse transforms as a third, and needs no partner -- no exchange of partial chains through LDS (16 of 112
// ds_write_b128 and 16 of 144 ds_read_b128 per ciphertext-step), no s_barrier in the step loop, no wait for a partner on
// another SIMD (timing-only ablation on the pair kernel, tools/patches/abl_noexchange.patch: -14 %).
// Results are bit-identical to k_blind_rotate: the same transforms, and the accumulation of output polynomial c runs
// in the canonical order (oracle/tfhe_oracle.c) -- terms (q_in = 1 - c, p = 1, 2) first, then (q_in = c, p = 1, 2) --
// which needs all four spectra live when the chains start (128 registers): the key rows are therefore streamed bin
// block by bin block (r = 0..7: eight 16-byte loads, one per row, then the eight chain steps of that block), so that
// a chain value replaces two spectrum values as the loop advances.  Gadget length 3 would need six live spectra and
// stays on the pair kernel.
// Workgroup = two independent ciphertexts (128 threads, the pair kernel's 35.6 KB of LDS: tables + one scratch per wave).
// =================================================================================================
// two inverse transforms skewed on one scratch (a's transposes under b's register passes and vice versa); the un-twist
// factors are read once for both.  Arithmetic per transform identical to fft_inv_wave.
__device__ __forceinline__ void fft_inv_x2(d2 (&xa)[8], d2 (&xb)[8], d2 (&ut)[8], const d2 *tw, const d2 *s_twist, d2 *scr,
                                           int lane)
{
    d2 t1[4], t0[4];
    const int hi = lane >> 3, lo = lane & 7;
    EOC_SB();
    tw_load(t1, tw + kTwI1 + (lane & 7), 8);
    inv_pass2(xa);
    EOC_SB();
    // region B: register-constant pass of b, a's first transpose spread through it
#pragma unroll
    for (int r = 0; r < 8; r++) scr[f12(lane * 8 + r)] = xa[r];
    wave_lds_fence();
    inv_pass2(xb);
    t21_read(xa, scr, lane);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 5);
    }
    EOC_SGB(EOC_M_VALU, 16);
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SB();
    // region C: middle pass of a, b's first transpose spread through it; then b's reads and the last twiddle set
#pragma unroll
    for (int r = 0; r < 8; r++) scr[f12(lane * 8 + r)] = xb[r];
    wave_lds_fence();
    t21_read(xb, scr, lane);
    tw_load(t0, tw + kTwI0 + lane, 64);
    wave_lds_fence();
    inv_pass10(xa, t1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    EOC_SGB(EOC_M_DSR, 12);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    // region D: middle pass of b under a's second transpose
#pragma unroll
    for (int r = 0; r < 8; r++) scr[72 * hi + 8 * r + lo] = xa[r];
    wave_lds_fence();
    t10_read(xa, scr, lane);
    inv_pass10(xb, t1);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    EOC_SGB(EOC_M_DSR, 8);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    // region E: last pass of a under b's second transpose; the un-twist factors
#pragma unroll
    for (int r = 0; r < 8; r++) scr[72 * hi + 8 * r + lo] = xb[r];
    wave_lds_fence();
    t10_read(xb, scr, lane);
#pragma unroll
    for (int r = 0; r < 8; r++) ut[r] = s_twist[lane + 64 * r];
    wave_lds_fence();
    inv_pass10(xa, t0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        EOC_SGB(EOC_M_DSW, 1);
        EOC_SGB(EOC_M_VALU, 4);
    }
    EOC_SGB(EOC_M_DSR, 16);
    EOC_SGB(EOC_M_VALU, 40);
    EOC_SB();
    inv_pass10(xb, t0);
}

constexpr int kBRWideJobsPerWG = 2;
constexpr int kBRWideLds = (kTwEntries + kNH + kBRWideJobsPerWG * kScr) * 16 + kBRWideJobsPerWG * kAbarLds;

template <int BGBIT = 0, bool SABAR = false>
__global__ __launch_bounds__(64 * kBRWideJobsPerWG, 2) void k_blind_rotate_wide(BRArgs A, const d2 *__restrict__ g_tw,
                                                              const d2 *__restrict__ g_twist)
{
    constexpr int L = 2, KPL = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    d2 *s_tw = reinterpret_cast<d2 *>(smem);
    d2 *s_twist = s_tw + kTwEntries;
    d2 *s_scr_all = s_twist + kNH;

    const int tid = threadIdx.x, lane = tid & 63;
    const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
    d2 *scr = s_scr_all + h * kScr;
    int32_t *ext = reinterpret_cast<int32_t *>(scr);

    const uint32_t job = blockIdx.x * kBRWideJobsPerWG + (uint32_t)h; // grid = ceil(jobs / waves per workgroup)
    load_tables(s_tw, s_twist, g_tw, g_twist, tid, 64 * kBRWideJobsPerWG);
    if (A.prep && A.step_begin == 0 && job < A.njobs) { // folded k_prepare: this wave's row of rotation amounts
        const uint32_t gjob = A.job0 + job, g = gjob / A.ks_S;
        prepare_row(A.inline_desc ? A.desc0 : A.ks_descs[g], gjob - g * A.ks_S, A.n, A.bara + (size_t)job * A.bara_stride, lane, 64);
        if constexpr (SABAR) eoc_row_stores_to_l2();
    }
    __syncthreads();
    if constexpr (SABAR) eoc_scalar_cache_acquire();
    if (job >= A.njobs) return; // the idle wave of an odd last workgroup (the only barrier is behind it)

    typedef const __attribute__((address_space(4))) uint32_t *cu32p;
    unsigned long long bara_addr = (unsigned long long)(uintptr_t)(A.bara + (size_t)job * A.bara_stride);
    asm volatile("" : "+s"(bara_addr)); // opaque behind the prologue: see k_blind_rotate
    const cu32p bara32 = (cu32p)bara_addr;
    // shipped form: the wave copies its row into LDS once and every step reads its amount from there (a wave's LDS
    // operations execute in order: no barrier between the copy and the reads)
    uint16_t *s_abar = reinterpret_cast<uint16_t *>(s_scr_all + kBRWideJobsPerWG * kScr) + h * (kAbarLds / 2);
    if constexpr (!SABAR) {
        const uint32_t *bara_v = reinterpret_cast<const uint32_t *>(A.bara + (size_t)job * A.bara_stride);
        for (int m = lane; m < (A.n + 2) / 2; m += 64) reinterpret_cast<uint32_t *>(s_abar)[m] = bara_v[m];
        wave_lds_fence();
    }
    auto load_abar = [&](int idx) __attribute__((always_inline)) {
        if constexpr (SABAR) return (int)((bara32[idx >> 1] >> ((idx & 1) * 16)) & 0xffffu);
        else return (int)s_abar[idx];
    };

    // ACC = (0, X^(2N - barb) * testvect): coefficient lane + 64 r of polynomial q in racc_q[r] (r < 8), + 512 in racc_q[8 + r]
    uint32_t racc0[16], racc1[16];
    {
        const int barb = load_abar(A.n);
        const int rot = (2 * kN - barb) & (2 * kN - 1);
        const int32_t *st = A.acc_state + (size_t)job * 2 * kN;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = lane + 64 * (r & 7) + (r >> 3) * kNH;
            const int idx = (j - rot) & (2 * kN - 1);
            int32_t v0 = 0, v1 = (idx & kN) ? -A.mu : A.mu;
            if (A.step_begin > 0) { // continue a blind rotation started by an earlier launch
                v0 = st[j];
                v1 = st[kN + j];
            }
            racc0[r] = (uint32_t)v0;
            racc1[r] = (uint32_t)v1;
        }
    }

    const int Bgbit = BGBIT > 0 ? BGBIT : A.Bgbit;
    const uint32_t Bg = 1u << Bgbit, maskBg = Bg - 1, halfBg = Bg >> 1;
    uint32_t offset = 0;
#pragma unroll
    for (int p = 1; p <= L; p++) offset += halfBg << (32 - p * Bgbit);
    const __amdgpu_buffer_rsrc_t bk_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(A.bkfft), 0, (int)((size_t)A.n * KPL * 2 * kNH * 16), 0x00020000);
    const double bias1 = 4503599627370496.0 + (double)halfBg, bias2 = 4503599627370496.0 + (double)Bg;
    const int prio_slot = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (3 << 11)); // HW_ID.WAVE_ID: slot on the SIMD

#ifdef EOC_STAMPS
    unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
    st_acc[12] = st_prev;
    st_acc[14] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    st_acc[13] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
#endif
    int abar_next = load_abar(A.step_begin);
    for (int i = A.step_begin; i < A.step_end; i++) {
        EOC_STAMP(15);
        if (A.prio_duty >= 0) { // see k_blind_rotate: the two waves of a SIMD alternate the issue priority
            const bool first_part = (i & 15) < A.prio_duty;
            if (first_part == ((prio_slot & 1) != 0))
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        }
        const int abar = __builtin_amdgcn_readfirstlane(abar_next);
        abar_next = load_abar(i + 1); // one step ahead: an LDS read, or (SABAR) an s_load
        // (X^abar - 1) * ACC_q, q = 0, 1, as biased digit words (see k_blind_rotate)
        uint32_t d0[16], d1[16];
        {
            const int s = abar & 63;
            const int src = ((lane - s) & 63) << 2;
            uint32_t t0[16], t1[16];
#pragma unroll
            for (int r = 0; r < 16; r++) t0[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)racc0[r]);
#pragma unroll
            for (int r = 0; r < 16; r++) t1[r] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)racc1[r]);
            const bool borrow = lane < s;
#define EOC_RQ(q) case q: rot_digits<q>(t0, racc0, borrow, offset, d0); rot_digits<q>(t1, racc1, borrow, offset, d1); break;
            switch (abar >> 6) {
                EOC_RQ(0) EOC_RQ(1) EOC_RQ(2) EOC_RQ(3) EOC_RQ(4) EOC_RQ(5) EOC_RQ(6) EOC_RQ(7)
                EOC_RQ(8) EOC_RQ(9) EOC_RQ(10) EOC_RQ(11) EOC_RQ(12) EOC_RQ(13) EOC_RQ(14) EOC_RQ(15)
                EOC_RQ(16) EOC_RQ(17) EOC_RQ(18) EOC_RQ(19) EOC_RQ(20) EOC_RQ(21) EOC_RQ(22) EOC_RQ(23)
                EOC_RQ(24) EOC_RQ(25) EOC_RQ(26) EOC_RQ(27) EOC_RQ(28) EOC_RQ(29) EOC_RQ(30)
                default: rot_digits<31>(t0, racc0, borrow, offset, d0); rot_digits<31>(t1, racc1, borrow, offset, d1); break;
            }
#undef EOC_RQ
        }
        // digit p of the 16 coefficients of this lane (digit words d[0..7] = low half, d[8..15] = high half), first pass
        auto make_x0 = [&](const uint32_t (&d)[16], int p, d2 (&x)[8]) __attribute__((always_inline)) {
            const int shift = 32 - p * Bgbit;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t ua = (d[r] >> shift) & maskBg, ub = (d[8 + r] >> shift) & maskBg;
                const uint32_t up = (d[r + 4] >> shift) & maskBg, uq = (d[12 + r] >> shift) & maskBg;
                const double a = __hiloint2double(0x43300000, (int)ua) - bias1;
                const double b = __hiloint2double(0x43300000, (int)ub) - bias1;
                const double dm = __hiloint2double(0x43300000, (int)(up - uq + Bg)) - bias2;
                const double dp = __hiloint2double(0x43300000, (int)(up + uq)) - bias2;
                fwd_stage0(x[r], x[r + 4], a, b, dm, dp);
            }
            fwd_pass0_tail(x);
        };
        EOC_STAMP(0);
        d2 xs0[2][8], xs1[2][8];
        make_x0(d0, 1, xs0[0]);
        fft_fwd_rest_x2(xs0[0], xs0[1], [&]() __attribute__((always_inline)) { make_x0(d0, 2, xs0[1]); }, s_tw, scr, lane);
        EOC_STAMP(1);
        make_x0(d1, 1, xs1[0]);
        fft_fwd_rest_x2(xs1[0], xs1[1], [&]() __attribute__((always_inline)) { make_x0(d1, 2, xs1[1]); }, s_tw, scr, lane);
        EOC_STAMP(2);

        // the two chains, bin block by bin block; row (q, p, c) of BK_i sits at ((i KPL + q L + p - 1) 2 + c) * 8 KiB
        d2 S0[8], S1[8];
        const uint32_t step_off = (uint32_t)((size_t)i * KPL * 2 * kNH * 16);
        auto ld = [&](int q, int p, int c, int r) __attribute__((always_inline)) {
            const uint32_t off = step_off + (uint32_t)((((q * L) + (p - 1)) * 2 + c) * kNH * 16 + r * 1024);
            return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(bk_rsrc, lane * 16, (int)off, 0));
        };
        auto mul0 = [](d2 x, d2 b) __attribute__((always_inline)) {
            d2 o;
            o.x = EOC_FMA(-x.y, b.y, x.x * b.x);
            o.y = EOC_FMA(x.y, b.x, x.x * b.y);
            return o;
        };
        auto mac1 = [](d2 x, d2 b, d2 a) __attribute__((always_inline)) {
            d2 o;
            o.x = EOC_FMA(-x.y, b.y, EOC_FMA(x.x, b.x, a.x));
            o.y = EOC_FMA(x.y, b.x, EOC_FMA(x.x, b.y, a.y));
            return o;
        };
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const d2 b110 = ld(1, 1, 0, r), b120 = ld(1, 2, 0, r), b010 = ld(0, 1, 0, r), b020 = ld(0, 2, 0, r);
            const d2 b011 = ld(0, 1, 1, r), b021 = ld(0, 2, 1, r), b111 = ld(1, 1, 1, r), b121 = ld(1, 2, 1, r);
            d2 a0 = mul0(xs1[0][r], b110);
            a0 = mac1(xs1[1][r], b120, a0);
            a0 = mac1(xs0[0][r], b010, a0);
            a0 = mac1(xs0[1][r], b020, a0);
            d2 a1 = mul0(xs0[0][r], b011);
            a1 = mac1(xs0[1][r], b021, a1);
            a1 = mac1(xs1[0][r], b111, a1);
            a1 = mac1(xs1[1][r], b121, a1);
            S0[r] = a0;
            S1[r] = a1;
        }
        EOC_STAMP(3);
        d2 ut[8];
        fft_inv_x2(S0, S1, ut, s_tw, s_twist, scr, lane);
        EOC_STAMP(8);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const d2 y0 = cmulc(S0[r], ut[r]), y1 = cmulc(S1[r], ut[r]); // 1/512 is in the key image
            racc0[r] += wrap_trunc(y0.x);
            racc0[8 + r] += wrap_trunc(y0.y);
            racc1[r] += wrap_trunc(y1.x);
            racc1[8 + r] += wrap_trunc(y1.y);
        }
        wave_lds_fence();
        EOC_STAMP(9);
    }
#ifdef EOC_STAMPS
    st_acc[11] = __builtin_amdgcn_s_memtime();
    if (A.stamps && lane == 0)
        for (int k = 0; k < 16; k++) A.stamps[((size_t)blockIdx.x * kBRWideJobsPerWG + h) * 16 + k] = st_acc[k];
#endif

    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (A.step_end < A.n) { // not the last part: park the accumulators for the next launch
        int32_t *st = A.acc_state + (size_t)job * 2 * kN;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            st[lane_e + 64 * (r & 7) + (r >> 3) * kNH] = (int32_t)racc0[r];
            st[kN + lane_e + 64 * (r & 7) + (r >> 3) * kNH] = (int32_t)racc1[r];
        }
        return;
    }
    // tLweExtractLweSample, index 0: u_0 = ACC_0[0], u_j = -ACC_0[N - j]; b = ACC_1[0].  The signed periodic image of
    // ACC_0 goes through the scratch once (read by index); register r holds coefficient lane + 64 (r & 7) + 512 (r >> 3)
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int j = lane_e + 64 * r;
        ext[j] = (int32_t)racc0[r];
        ext[j + kN] = (int32_t)(0u - racc0[r]);
    }
    wave_lds_fence();
    const int32_t bval = (int32_t)__builtin_amdgcn_readfirstlane((int)racc1[0]); // ACC_1[0]: lane 0, register 0
    if (A.ks_descs) { // + lweKeySwitch set-up: ubar_j = u_j + 2^(31 - t basebit), out = (0, ..., 0, b)
        const uint32_t gjob = A.job0 + job;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = lane_e + 64 * r;
            A.ubar[(size_t)gjob * kN + j] = (uint32_t)ext[(2 * kN - j) & (2 * kN - 1)] + A.ks_prec_offset;
        }
        const uint32_t g = gjob / A.ks_S, si = gjob - g * A.ks_S;
        int32_t *o = (A.inline_desc ? A.desc0.out : A.ks_descs[g].out) + (size_t)si * (A.n + 1);
        for (int m = lane_e; m < A.n; m += 64) o[m] = 0;
        if (lane_e == 0) o[A.n] = bval;
    } else {
        int32_t *u = A.u + (size_t)job * (kN + 1);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = lane_e + 64 * r;
            u[j] = ext[(2 * kN - j) & (2 * kN - 1)];
        }
        if (lane_e == 0) u[kN] = bval;
    }
}

// =================================================================================================
// K3: key switch (lweKeySwitch), two launches.
//   k_ks_init      per ciphertext: ubar[job][i] = u_i + 2^(31 - t*basebit)  (sum of the two extracted
//                  samples + (0, mu) for MUX), out = (0, ..., 0, b')
//   k_keyswitch_waves   basebit 2, even t: 64 ciphertexts per workgroup (one per lane), two digits per LDS look-up
//                  through a table of row sums, partial sums of the i-slices through LDS and a few integer atomics
//   k_keyswitch_generic any other (basebit, t): the plain one-thread-per-output-word form
// KSK device image: [N*t][base-1][n1p], rows padded with zeros to n1p (multiple of 256).
// =================================================================================================
constexpr int KS_GT = 64; // ciphertexts per workgroup (one per lane)

struct KSArgs {
    const int32_t *ksk;
    const int32_t *u;   // [jobs][N+1]
    uint32_t *ubar;     // [jobs][N] row-major
    int n, n1p, t, basebit;
    uint32_t S;
    GateDesc desc0;     // single-gate levels: the descriptor as a kernel argument (see BRArgs)
    int inline_desc;
    int32_t mu;
};

// grid: x = S, y = gates
__global__ __launch_bounds__(256) void k_ks_init(const GateDesc *__restrict__ descs, KSArgs A)
{
    const GateDesc d = A.inline_desc ? A.desc0 : descs[blockIdx.y];
    const uint32_t s = blockIdx.x;
    const uint32_t job = d.job_base + s;
    const int32_t *u1 = A.u + (size_t)job * (kN + 1);
    const bool mux = d.op == OP_MUX;
    const int32_t *u2 = u1 + (size_t)A.S * (kN + 1);
    const uint32_t prec_offset = 1u << (32 - (1 + A.basebit * A.t));
    for (int j = threadIdx.x; j < kN; j += 256) {
        uint32_t v = (uint32_t)u1[j];
        if (mux) v += (uint32_t)u2[j];
        A.ubar[(size_t)job * kN + j] = v + prec_offset;
    }
    int32_t *o = d.out + (size_t)s * (A.n + 1);
    for (int m = threadIdx.x; m < A.n; m += 256) o[m] = 0;
    if (threadIdx.x == 0) {
        uint32_t b = (uint32_t)u1[kN];
        if (mux) b += (uint32_t)u2[kN] + (uint32_t)A.mu;
        o[A.n] = (int32_t)b;
    }
}

typedef int i4 __attribute__((ext_vector_type(4)));

// ---- k_keyswitch_waves<T, NWV, IW>: basebit 2, even t (both default sets) ------------------------------------------
// One lane = one ciphertext.  A workgroup is (tile of 64 ciphertexts) x (ONE block of 64 key columns) x (a slice of
// NWV * IW indices i); its NWV waves split the SLICE: wave w walks IW indices of its own.
//   * Two digits per look-up.  For digits j, j + 1 of index i the wave first combines the 3 + 3 candidate key rows into
//     the table of their 16 sums, S[d1][d2] = row_j[d1] + row_{j+1}[d2] (64 columns, 4.3 KB, double-buffered in LDS;
//     d1 = d2 = 0 is the zero row), built by its own lanes with lane = COLUMN: six coalesced dword loads (exactly the
//     wave's 256-byte strips of the six rows), nine additions, and the fifteen non-zero rows leave through
//     ds_write_addtid_b32 -- the LDS store whose address is M0 + offset + 4 * lane, no address register: a 256-byte row
//     in 2 cycles of the CU's store path against 13.6 for a ds_write_b128 (tools/ubench_addtid.hip pins its addressing:
//     lane within the wave, no 16-bit wrap of the sum, one wait state after the SALU write of M0).
//     Every lane then selects with the 4-bit nibble of its operand word that holds
//     both digits of ITS ciphertext and does ONE look-up and ONE subtraction per column for the pair:
//     acc -= S[nibble]  ==  acc -= row_j[d1]; acc -= row_{j+1}[d2]  in wrapping int32 arithmetic -- bit-identical, at half
//     the subtractions and half the LDS reads per ciphertext of the one-digit form (rows are skewed by 16 B so that the 16
//     table rows cover the 64 banks: lanes with equal nibbles share an address, others never share a bank).
//   * No barrier in the stage loop: the table is the wave's own (a wave's LDS operations execute in order).
//   * The NWV partial sums of a workgroup are added through LDS (tree: the upper half of the waves parks its
//     accumulators, the lower half adds), so only the kN / (NWV IW) = 2...8 slices meet in the output row through integer
//     atomics (exact, order independent), 256 contiguous bytes per atomic instruction.
// History (Set A, 1024 gates): round 3's kernel -- one digit per look-up, the candidate rows of 4 digits staged per
// workgroup barrier, one i-slice x all columns per workgroup, 32 atomics per output word -- took 0.212 ms: integer issue
// and the LDS pipe were loaded exactly alike (4 bytes from LDS per subtraction) and 16.8 M atomics arrived together at the
// end of the kernel (33 us, timing-only ablation).  Pair tables alone: 0.180 ms; with one i-range per wave: 0.133 ms; tables
// built through ds_write_addtid_b32 instead of ds_write_b128: 0.122 ms (Set B: 0.363 -> 0.304 -> 0.196 -> 0.178 ms).  An XCD-aware workgroup order (all tiles of a slice on one XCD) changed nothing:
// the 50 MB key image is served by the Infinity Cache either way.
template <int T, int NWV, int IW>
struct KS3Cfg {
    static constexpr int NT = 64 * NWV;
    static constexpr int ROWI = 64 + 4;                  // table row stride in ints (16-B skew: 16 rows cover the 64 banks)
    static constexpr int TABI = 16 * ROWI;               // one table, ints
    static constexpr int WAVE_I = 2 * TABI;              // two tables per wave
    static constexpr int RED_I = 64 * ROWI;              // one wave's partial sums [64 ciphertexts][64 columns + 4]
    static constexpr int TAB_BYTES = NWV * WAVE_I * 4, RED_BYTES = (NWV > 1 ? NWV / 2 : 1) * RED_I * 4;
    static constexpr int LDS_BYTES = TAB_BYTES > RED_BYTES ? TAB_BYTES : RED_BYTES;
    static constexpr int NPAIR = T / 2;
    static constexpr int NSTAGE = IW * NPAIR;
    static constexpr int SLICE_I = NWV * IW;             // indices i per workgroup
    static constexpr int NS = kN / SLICE_I;              // slices (= atomics per output word)
};

// grid: x = ntiles * (n1p / 64) * NS, y = gates; block = 64 NWV
template <int T, int NWV, int IW>
__global__ __launch_bounds__(64 * NWV, 4) void k_keyswitch_waves(const GateDesc *__restrict__ descs, KSArgs A)
{
    typedef KS3Cfg<T, NWV, IW> C;
    static_assert(T % 2 == 0 && kN % C::SLICE_I == 0 && (NWV & (NWV - 1)) == 0, "shape");
    static_assert((NWV * C::WAVE_I - C::TABI) * 4 < 65536, "ds_write_addtid_b32 takes its base from M0[15:0]");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *s_all = reinterpret_cast<int *>(smem);
    // ds_write_addtid_b32 adds M0[15:0]: the static_assert above bounds the largest base ONLY IF the dynamic block starts at
    // LDS offset 0, i.e. the kernel has no static __shared__ object -- trap instead of silently wrapping if that ever changes
    if ((uint32_t)(uintptr_t)smem != 0u) __builtin_trap();

    const GateDesc d = A.inline_desc ? A.desc0 : descs[blockIdx.y];
    const uint32_t ntiles = (A.S + 63) / 64, ncb = (uint32_t)A.n1p / 64u;
    const uint32_t tile = blockIdx.x % ntiles, rest = blockIdx.x / ntiles;
    const uint32_t cb = rest % ncb, slice = rest / ncb;
    const uint32_t s0 = tile * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool valid = s0 + lane < A.S;
    const uint32_t job = d.job_base + (valid ? s0 + lane : A.S - 1);
    const int i0 = (int)slice * C::SLICE_I + w * IW; // this wave's indices: [i0, i0 + IW)

    int *s_tab = s_all + w * C::WAVE_I; // this wave's two tables [2][16][ROWI]
    const size_t n1p = (size_t)A.n1p;
    // Table building, lane = column.  Stage st covers i = i0 + st / NPAIR, digits j0 = 2 (st % NPAIR) and j0 + 1: six
    // adjacent key rows, of which this wave needs the 256-byte strip of its column block.  Everything in the stage loop is
    // unconditional (the last stage re-loads itself): one basic block, so that look-ups and subtractions stay in the order
    // written.  M0 is saved and restored around the add-TID stores (the compiler does not expect it to change).
    int kr[6];
    for (int k = lane; k < 2 * C::ROWI; k += 64) s_tab[(k / C::ROWI) * C::TABI + k % C::ROWI] = 0; // the two zero rows
    const uint32_t tab_lds = (uint32_t)(uintptr_t)s_tab;
    auto stage_load = [&](int st) {
        const int i = i0 + st / C::NPAIR, j0 = 2 * (st % C::NPAIR);
        const int *src = A.ksk + ((size_t)i * T + j0) * 3 * n1p + cb * 64 + lane;
#pragma unroll
        for (int k = 0; k < 6; k++) kr[k] = src[k * n1p];
    };
#define EOC_KS_ROW(BUF, D1, D2) ((BUF) * C::TABI * 4 + ((D1) * 4 + (D2)) * C::ROWI * 4)
    auto stage_store = [&](int buf) {
        const int a1 = kr[0], a2 = kr[1], a3 = kr[2], b1 = kr[3], b2 = kr[4], b3 = kr[5];
        const uint32_t base = tab_lds + (uint32_t)buf * (C::TABI * 4);
        uint32_t m0_saved;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %1\n\t"
                     "s_nop 1\n\t" /* an SALU write of M0 needs a wait state before an add-TID LDS instruction reads it */
                     "ds_write_addtid_b32 %2 offset:%17\n\t"
                     "ds_write_addtid_b32 %3 offset:%18\n\t"
                     "ds_write_addtid_b32 %4 offset:%19\n\t"
                     "ds_write_addtid_b32 %5 offset:%20\n\t"
                     "ds_write_addtid_b32 %6 offset:%21\n\t"
                     "ds_write_addtid_b32 %7 offset:%22\n\t"
                     "ds_write_addtid_b32 %8 offset:%23\n\t"
                     "ds_write_addtid_b32 %9 offset:%24\n\t"
                     "ds_write_addtid_b32 %10 offset:%25\n\t"
                     "ds_write_addtid_b32 %11 offset:%26\n\t"
                     "ds_write_addtid_b32 %12 offset:%27\n\t"
                     "ds_write_addtid_b32 %13 offset:%28\n\t"
                     "ds_write_addtid_b32 %14 offset:%29\n\t"
                     "ds_write_addtid_b32 %15 offset:%30\n\t"
                     "ds_write_addtid_b32 %16 offset:%31\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(m0_saved)
                     : "s"(base), "v"(b1), "v"(b2), "v"(b3), "v"(a1), "v"(a1 + b1), "v"(a1 + b2), "v"(a1 + b3), "v"(a2),
                       "v"(a2 + b1), "v"(a2 + b2), "v"(a2 + b3), "v"(a3), "v"(a3 + b1), "v"(a3 + b2), "v"(a3 + b3),
                       "n"(EOC_KS_ROW(0, 0, 1)), "n"(EOC_KS_ROW(0, 0, 2)), "n"(EOC_KS_ROW(0, 0, 3)), "n"(EOC_KS_ROW(0, 1, 0)),
                       "n"(EOC_KS_ROW(0, 1, 1)), "n"(EOC_KS_ROW(0, 1, 2)), "n"(EOC_KS_ROW(0, 1, 3)), "n"(EOC_KS_ROW(0, 2, 0)),
                       "n"(EOC_KS_ROW(0, 2, 1)), "n"(EOC_KS_ROW(0, 2, 2)), "n"(EOC_KS_ROW(0, 2, 3)), "n"(EOC_KS_ROW(0, 3, 0)),
                       "n"(EOC_KS_ROW(0, 3, 1)), "n"(EOC_KS_ROW(0, 3, 2)), "n"(EOC_KS_ROW(0, 3, 3))
                     : "memory"); // (an "m0" clobber is refused -- "reserved register" -- so M0 is saved and restored by
                                  // hand inside the block; the trailing s_mov m0 is followed by the block's own consumers
                                  // only after a wave_lds_fence, never by an M0-reading instruction of the compiler's)
    };
#undef EOC_KS_ROW

    i4 acc[16];
#pragma unroll
    for (int c = 0; c < 16; c++) acc[c] = (i4){0, 0, 0, 0};

    stage_load(0);
    stage_store(0);
    wave_lds_fence();
    // this lane's IW operand words are IW * 4 contiguous bytes of its ciphertext's row (the blind rotation's epilogue
    // writes rows, coalesced): read 16 bytes every fourth index -- the tile is transposed on load, one lane per row
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 *ubp = reinterpret_cast<const u4 *>(A.ubar + (size_t)job * kN + i0);
    u4 ub4 = ubp[0];
#pragma unroll 1
    for (int ii = 0; ii < IW; ii++) {
        const int k4 = ii & 3;
        const uint32_t ub = k4 == 0 ? ub4.x : (k4 == 1 ? ub4.y : (k4 == 2 ? ub4.z : ub4.w));
        u4 ub4n = ub4;
        if (k4 == 3 && ii + 1 < IW) ub4n = ubp[(ii + 1) >> 2];
#pragma unroll 1
        for (int q = 0; q < C::NPAIR; q++) {
            const int st = ii * C::NPAIR + q;
            const int buf = st & 1;
            stage_load(st + 1 < C::NSTAGE ? st + 1 : st);
            const uint32_t nib = (ub >> (28 - 4 * q)) & 15u; // digits 2q and 2q + 1 of this lane's ciphertext
            const i4 *row = reinterpret_cast<const i4 *>(s_tab + buf * C::TABI + (int)nib * C::ROWI);
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += 8) {
                i4 v[8];
#pragma unroll
                for (int c = 0; c < 8; c++) v[c] = row[c0 + c];
#pragma unroll
                for (int c = 0; c < 8; c++) acc[c0 + c] -= v[c];
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_store(buf ^ 1);
            wave_lds_fence();
        }
        ub4 = ub4n;
    }

    // the workgroup's NWV partial sums -> one: the upper half of the waves parks its accumulators in LDS, the lower half adds
    __syncthreads(); // every wave is done with its tables
#pragma unroll
    for (int h = NWV / 2; h >= 1; h /= 2) {
        if (w >= h && w < 2 * h) {
            int *r = s_all + (w - h) * C::RED_I + lane * C::ROWI;
#pragma unroll
            for (int c = 0; c < 16; c++) *reinterpret_cast<i4 *>(r + 4 * c) = acc[c];
        }
        __syncthreads();
        if (w < h) {
            const int *r = s_all + w * C::RED_I + lane * C::ROWI;
#pragma unroll
            for (int c = 0; c < 16; c++) acc[c] += *reinterpret_cast<const i4 *>(r + 4 * c);
        }
        __syncthreads();
    }
    if (w == 0) {
        int *r = s_all + lane * C::ROWI;
#pragma unroll
        for (int c = 0; c < 16; c++) *reinterpret_cast<i4 *>(r + 4 * c) = acc[c];
    }
    __syncthreads();
    // output: wave w takes 64 / NWV ciphertexts, lane = column (256 contiguous bytes per atomic instruction)
    const int col = (int)cb * 64 + lane;
#pragma unroll 4
    for (int k = 0; k < 64 / NWV; k++) {
        const int ct = w * (64 / NWV) + k;
        const int v = s_all[ct * C::ROWI + lane];
        if (v != 0 && s0 + ct < A.S && col <= A.n) atomicAdd(d.out + (size_t)(s0 + ct) * (A.n + 1) + col, v);
    }
}

// Key switch for any (basebit <= 4, t) the kernel above is not instantiated for (both default parameter sets use
// basebit 2, t 8 and never come here): one thread per output word, rows read straight from the key image.  A plain,
// slow, exact form -- lweKeySwitchTranslate_fromArray as written.  The output row holds (0, ..., 0, b')
// when it starts (k_ks_init or the blind rotate's epilogue wrote it).  grid: x = S, y = gates; block = 256
__global__ __launch_bounds__(256) void k_keyswitch_generic(const GateDesc *__restrict__ descs, KSArgs A)
{
    const GateDesc d = A.inline_desc ? A.desc0 : descs[blockIdx.y];
    const uint32_t s = blockIdx.x, job = d.job_base + s;
    const int base1 = (1 << A.basebit) - 1;
    int32_t *o = d.out + (size_t)s * (A.n + 1);
    for (int m = threadIdx.x; m <= A.n; m += 256) {
        uint32_t acc = (uint32_t)o[m];
        for (int i = 0; i < kN; i++) {
            const uint32_t ub = A.ubar[(size_t)job * kN + i];
            for (int j = 0; j < A.t; j++) {
                const uint32_t dg = (ub >> (32 - (j + 1) * A.basebit)) & (uint32_t)base1;
                if (dg) acc -= (uint32_t)A.ksk[(((size_t)i * A.t + j) * base1 + (dg - 1)) * A.n1p + m];
            }
        }
        o[m] = (int32_t)acc;
    }
}

} // namespace eoc
