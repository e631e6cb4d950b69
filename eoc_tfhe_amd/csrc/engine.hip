// engine.hip -- device engine behind the C ABI of include/eoc_tfhe_gpu.h (layers 1 and 2).
//
// Mirrors the role libtfhe's LweBootstrappingKeyFFT + boots* functions play under
// ao-tfhe/eoc-tfhe-run.cpp (the reference builds the key at :231 and never calls a gate,
// SURVEY.md 0.1); here the cloud key lives in HBM and every gate batch is three kernel launches.
// No CPU fallback exists: if HIP reports no device the constructors fail.
#include "kernels.hip.h"
#include "canon_twiddles.h"
#include "common.h"
#include "../../include/eoc_tfhe_gpu.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using namespace eoc;

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            eoc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                        \
            return EOC_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

struct eoc_engine {
    int device = 0;
    eoc_params p{};
    int kpl = 0;
    size_t n1p = 0;
    d2 *d_tw = nullptr, *d_twist = nullptr;
    // cloud key images
    double *d_bkfft = nullptr;
    int32_t *d_ksk = nullptr;
    bool own_keys = false;
    const double *bkfft = nullptr; // in use (owned or adopted)
    const int32_t *ksk = nullptr;
    // workspace: device buffers AND a pinned host ring for the gate descriptors / opcode permutations that every launch
    // sends ahead of its kernels: nothing on the launch path reads pageable memory asynchronously, allocates or
    // synchronises once the workspace has its size (eoc_engine_reserve).  One set per engine: all kernels of an engine
    // run on one stream at a time (the host-buffer path's chunks share the kernel stream, multi.hip).
    struct Workspace {
        uint16_t *d_bara = nullptr;
        int32_t *d_u = nullptr;
        uint32_t *d_ubar = nullptr;  // [ws_jobs][N] key-switch operand u + 2^(31 - t basebit), row-major
        int32_t *d_acc_state = nullptr; // [resident jobs][2][N]: accumulators between the parts of a cut blind rotation
        size_t ws_jobs = 0;
        GateDesc *d_descs = nullptr, *h_descs = nullptr; // device ring + pinned host ring, same capacity
        size_t ws_descs = 0, desc_pos = 0;
        // recorded behind the consumers of the ring's last quarter (ring_mark): what the wrap-around waits for
        hipEvent_t ring_ev = nullptr;
        bool ring_ev_set = false;
        bool ring_pushed_since_mark = false; // descriptors were queued behind the last recorded mark (or no mark exists)
        // descriptors sent while the stream is being captured into a hipGraph live in blocks that are never re-used:
        // the captured copy node reads its pinned source again at every replay
        // (allocated with the ring -- nothing may be allocated while a stream captures -- four times its size)
        GateDesc *d_persist = nullptr, *h_persist = nullptr;
        size_t persist_cap = 0, persist_pos = 0;
        int32_t *d_mixed = nullptr; // gather/scatter space of mixed batches: 4 row arrays + perm
        uint32_t *h_perm = nullptr; // pinned
        hipEvent_t perm_ev = nullptr; // the last copy out of h_perm (awaited before h_perm is rewritten)
        size_t ws_mixed = 0;
    } ws;
    unsigned long long *d_stamps = nullptr; // diagnostic build (-DEOC_STAMPS) only
    int num_cus = 256;
    int prio_duty_override = INT32_MIN;     // EOC_TFHE_PRIO_DUTY in the environment (tuning / diagnostics)
    int prio_multi = -1;                    // duty code of launches of several rounds (EOC_TFHE_PRIO_MULTI)
    int br_slice = 0;                       // jobs per blind-rotate launch: 0 = resident set, < 0 = unlimited (EOC_TFHE_BR_SLICE)
    bool no_fold = false;                   // EOC_TFHE_NO_FOLD: keep k_prepare and k_ks_init as launches of their own
    bool scalar_abar = false;               // EOC_TFHE_SCALAR_ABAR: rotation amounts read back by scalar loads (kernels.hip.h)
    bool no_pool = false;                   // EOC_TFHE_NO_POOL: every opcode run of a mixed batch is a level of its own
    int br_parts = 0;                       // consecutive launches per blind rotation (EOC_TFHE_BR_PARTS); 0 = by key-row size
    int br_wide = -1;                       // one-wave-per-ciphertext kernel: -1 = by launch width, 0 = never, 1 = whenever l = 2 (EOC_TFHE_BR_WIDE)
    int bara_stride = 0;
    bool ks_waves_ok = true;                // the > 64 KiB dynamic-LDS attribute of k_keyswitch_waves was granted
    uint64_t stats[3] = {0, 0, 0};
    uint64_t ws_grows = 0; // times a workspace had to grow inside a call (0 after eoc_engine_reserve)
    uint64_t br_launches = 0; // k_blind_rotate kernel launches (a wide level is several, a cut blind rotation too)
    uint64_t br_wide_launches = 0; // ... of which k_blind_rotate_wide
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline)
    bool profiling = false;
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    double kernel_ms[3] = {0, 0, 0};
    uint64_t kernel_launches[3] = {0, 0, 0};
    std::mutex mu;
};

enum { KIND_PREPARE = 0, KIND_BLIND_ROTATE = 1, KIND_KEYSWITCH = 2 };

struct SpanGuard { // records start/stop events around one kernel launch when profiling is on
    eoc_engine *e;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    SpanGuard(eoc_engine *e_, hipStream_t st_, int kind_) : e(e_), st(st_), kind(kind_)
    {
        if (!e->profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
            a = b = nullptr;
            return;
        }
        hipEventRecord(a, st);
    }
    ~SpanGuard()
    {
        if (!a) return;
        hipEventRecord(b, st);
        e->spans.push_back({a, b, kind});
    }
};

static int valid_params(const eoc_params *p)
{
    if (!p) return 0;
    if (p->n < 1 || p->n > 1023) return 0;
    if (p->l < 1 || p->l > 4 || p->Bgbit < 1 || p->l * p->Bgbit > 32) return 0;
    // An external-product coefficient is bounded by l * Bg * 2^41 (2 l N digits of magnitude <= Bg / 2 times key
    // coefficients below 2^31).  The FP64 transform and its conversion are specified for |v| < 2^51 (include/eoc_tfhe_gpu.h:
    // unconditional for l * Bg < 1024, overwhelmingly probable up to l * Bg = 8192, where the typical magnitude is still
    // below 2^48); beyond that binary64 no longer holds the sums and neither this engine nor upstream's FFT path computes
    // the exact product.  Such shapes are refused rather than evaluated approximately.
    if (((int64_t)p->l << p->Bgbit) > 8192) return 0;
    if (p->ks_t < 1 || p->ks_basebit < 1 || p->ks_basebit > 4 || p->ks_t * p->ks_basebit > 31) return 0;
    return 1;
}

extern "C" size_t eoc_bkfft_bytes(const eoc_params *p) { return (size_t)p->n * 2 * p->l * 2 * kNH * 16; }
extern "C" size_t eoc_ksk_row_stride(const eoc_params *p) { return ((size_t)p->n + 1 + 255) / 256 * 256; }
extern "C" size_t eoc_ksk_dev_bytes(const eoc_params *p)
{
    return (size_t)kN * p->ks_t * (((size_t)1 << p->ks_basebit) - 1) * eoc_ksk_row_stride(p) * 4;
}

extern "C" int eoc_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

static int upload_tables(eoc_engine *e)
{
    // twiddle tables of kernels.hip.h (canonical transform v3) and the un-twist table E2048[j].
    // forward: rho(s, b) = E[(256 + 1024 bitrev_s(b)) >> s]; inverse: W[k] = E[4k].  Four entries per lane and pass;
    // the other three twiddles of a pass are i times these (E[k + 512] = i E[k] exactly).
    if (EOC_E2048[256][0] != EOC_SQRT_HALF || EOC_E2048[256][1] != EOC_SQRT_HALF || EOC_E2048[128][0] != EOC_E128_RE ||
        EOC_E2048[128][1] != EOC_E128_IM || EOC_E2048[64][0] != EOC_E64_RE || EOC_E2048[64][1] != EOC_E64_IM ||
        EOC_E2048[320][0] != EOC_E320_RE || EOC_E2048[320][1] != EOC_E320_IM) {
        eoc_set_error("kernel register constants differ from canon_twiddles.h");
        return EOC_ERR_STATE;
    }
    for (int k = 0; k < 512; k++)
        if (EOC_E2048[k + 512][0] != -EOC_E2048[k][1] || EOC_E2048[k + 512][1] != EOC_E2048[k][0]) {
            eoc_set_error("canon_twiddles.h: E[k + 512] != i E[k] at k = %d", k);
            return EOC_ERR_STATE;
        }
    std::vector<double> tw((size_t)kTwEntries * 2), twist((size_t)kNH * 2);
    auto put = [&](int idx2048, int entry) {
        tw[(size_t)entry * 2] = EOC_E2048[idx2048][0];
        tw[(size_t)entry * 2 + 1] = EOC_E2048[idx2048][1];
    };
    auto rho = [](int s, int b) { // index into E of the root of stage s, block b
        int r = 0;
        for (int k = 0; k < s; k++) r |= ((b >> k) & 1) << (s - 1 - k);
        return (256 + 1024 * r) >> s;
    };
    // forward pass 1 (stages 3,4,5): block = (hi << s') | c, hi = lane >> 3; entries A, B0, C0, C2
    for (int hi = 0; hi < 8; hi++) {
        put(rho(3, hi), kTwF1 + 0 * 8 + hi);
        put(rho(4, 2 * hi), kTwF1 + 1 * 8 + hi);
        put(rho(5, 4 * hi), kTwF1 + 2 * 8 + hi);
        put(rho(5, 4 * hi + 2), kTwF1 + 3 * 8 + hi);
    }
    // forward pass 2 (stages 6,7,8): block = (lane << s') | c
    for (int lane = 0; lane < 64; lane++) {
        put(rho(6, lane), kTwF2 + 0 * 64 + lane);
        put(rho(7, 2 * lane), kTwF2 + 1 * 64 + lane);
        put(rho(8, 4 * lane), kTwF2 + 2 * 64 + lane);
        put(rho(8, 4 * lane + 2), kTwF2 + 3 * 64 + lane);
    }
    // inverse middle pass (stages 5,4,3): i mod 64 = 8 r' + lo, lo = lane & 7; entries m6, m4, m0, m1
    for (int lo = 0; lo < 8; lo++) {
        put(4 * (lo * 32), kTwI1 + 0 * 8 + lo);
        put(4 * (lo * 16), kTwI1 + 1 * 8 + lo);
        put(4 * (lo * 8), kTwI1 + 2 * 8 + lo);
        put(4 * ((8 + lo) * 8), kTwI1 + 3 * 8 + lo);
    }
    // inverse last pass (stages 2,1,0): W[(i mod h) << s], i = lane + 64 r; entries n6, n4, n0, n1
    for (int lane = 0; lane < 64; lane++) {
        put(4 * (4 * lane), kTwI0 + 0 * 64 + lane);
        put(4 * (2 * lane), kTwI0 + 1 * 64 + lane);
        put(4 * lane, kTwI0 + 2 * 64 + lane);
        put(4 * (lane + 64), kTwI0 + 3 * 64 + lane);
    }
    for (int j = 0; j < kNH; j++) {
        twist[2 * j] = EOC_E2048[j][0];
        twist[2 * j + 1] = EOC_E2048[j][1];
    }
    HIP_TRY(hipMalloc(&e->d_tw, tw.size() * 8));
    HIP_TRY(hipMalloc(&e->d_twist, twist.size() * 8));
    HIP_TRY(hipMemcpy(e->d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->d_twist, twist.data(), twist.size() * 8, hipMemcpyHostToDevice));
    // NULL-stream audit: the tables are read by kernels on caller streams, which may be hipStreamNonBlocking
    // and would not wait for NULL-stream work: drain once, here, at engine creation
    HIP_TRY(hipDeviceSynchronize());
    return EOC_OK;
}

extern "C" int eoc_engine_create(int device, const eoc_params *p, eoc_engine **out)
{
    if (!out || !valid_params(p)) {
        eoc_set_error("eoc_engine_create: bad arguments (need 1 <= n <= 1023, 1 <= l <= 4, l * Bgbit <= 32, l * 2^Bgbit <= 8192 "
                      "-- the FP64 external product is not exact beyond that --, ks_t * ks_basebit <= 31, ks_basebit <= 4)");
        return EOC_ERR_ARG;
    }
    int cnt = eoc_device_count();
    if (cnt <= 0 || device < 0 || device >= cnt) {
        eoc_set_error("eoc_engine_create: no usable HIP device (count=%d, asked %d); the gate path has no CPU fallback",
                      cnt, device);
        return EOC_ERR_NO_DEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    eoc_engine *e = new (std::nothrow) eoc_engine();
    if (!e) return EOC_ERR_ALLOC;
    e->device = device;
    e->p = *p;
    e->kpl = 2 * p->l;
    e->n1p = eoc_ksk_row_stride(p);
    e->bara_stride = (p->n + 1 + 31) / 32 * 32; // rows 64 bytes apart: two jobs never share a (scalar) cache line
    int rc = upload_tables(e);
    if (rc) {
        delete e;
        return rc;
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) e->num_cus = cus;
        if (const char *s = getenv("EOC_TFHE_PRIO_DUTY")) e->prio_duty_override = atoi(s);
        if (const char *s = getenv("EOC_TFHE_PRIO_MULTI")) e->prio_multi = atoi(s);
        if (const char *s = getenv("EOC_TFHE_BR_SLICE")) e->br_slice = atoi(s);
        if (getenv("EOC_TFHE_NO_FOLD")) e->no_fold = true;
        if (getenv("EOC_TFHE_NO_POOL")) e->no_pool = true;
        if (const char *s = getenv("EOC_TFHE_SCALAR_ABAR")) e->scalar_abar = atoi(s) != 0;
        if (const char *s = getenv("EOC_TFHE_BR_PARTS")) e->br_parts = atoi(s);
        if (const char *s = getenv("EOC_TFHE_BR_WIDE")) e->br_wide = atoi(s);
    }
    // the key-switch kernel uses > 64 KiB of dynamic LDS: raise the limit once, here, not on the launch path
    // (a device that refuses it -- 64 KiB of LDS per workgroup -- sends every shape to k_keyswitch_generic: slow, exact)
#define EOC_KS_ATTR(TT, NWV, IWV)                                                                              \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_keyswitch_waves<TT, NWV, IWV>),                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, KS3Cfg<TT, NWV, IWV>::LDS_BYTES) != hipSuccess) \
        e->ks_waves_ok = false
    EOC_KS_ATTR(8, 8, 16);
    EOC_KS_ATTR(8, 8, 32);
    EOC_KS_ATTR(8, 4, 64);
    EOC_KS_ATTR(8, 8, 64);
#undef EOC_KS_ATTR
    if (!e->ks_waves_ok) (void)hipGetLastError();
    *out = e;
    return EOC_OK;
}

static void free_ws(eoc_engine::Workspace &W)
{
    hipFree(W.d_bara);
    hipFree(W.d_u);
    hipFree(W.d_ubar);
    hipFree(W.d_acc_state);
    hipFree(W.d_descs);
    hipFree(W.d_mixed);
    if (W.h_descs) hipHostFree(W.h_descs);
    if (W.h_perm) hipHostFree(W.h_perm);
    if (W.perm_ev) hipEventDestroy(W.perm_ev);
    if (W.ring_ev) hipEventDestroy(W.ring_ev);
    hipFree(W.d_persist);
    if (W.h_persist) hipHostFree(W.h_persist);
    W = eoc_engine::Workspace();
}

extern "C" void eoc_engine_destroy(eoc_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    hipFree(e->d_tw);
    hipFree(e->d_twist);
    if (e->own_keys) {
        hipFree(e->d_bkfft);
        hipFree(e->d_ksk);
    }
    free_ws(e->ws);
    hipFree(e->d_stamps);
    delete e;
}

// Grow a workspace set (outside the kernels: the device is synchronised and buffers are re-allocated; callers that
// must not stall or want hipGraph capture size the sets once with eoc_engine_reserve).  jobs = blind rotations of the
// widest level, descs = gate descriptors in flight between two wrap-arounds of the ring, mixed = rows of a mixed batch.
static bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
}

static int ensure_ws(eoc_engine *e, eoc_engine::Workspace &W, size_t jobs, size_t descs, size_t mixed, hipStream_t st = nullptr)
{
    if (jobs <= W.ws_jobs && descs <= W.ws_descs && mixed <= W.ws_mixed) return EOC_OK;
    // growing frees and re-allocates buffers that captured nodes may refer to, and nothing may be allocated while a
    // stream captures: a capture needs eoc_engine_reserve first
    if (stream_is_capturing(st)) {
        eoc_set_error("graph capture: the workspace would have to grow (jobs %zu > %zu, descriptors %zu > %zu or mixed rows "
                      "%zu > %zu); call eoc_engine_reserve with the sizes of the captured work before capturing",
                      jobs, W.ws_jobs, descs, W.ws_descs, mixed, W.ws_mixed);
        return EOC_ERR_STATE;
    }
    // Growth synchronises the DEVICE.  `st` is the only stream this call can test for capture; a capture in progress on
    // another stream of the process would be invalidated by the synchronise: eoc_engine_reserve before capturing ANYWHERE
    // in the process (include/eoc_tfhe_gpu.h, "Capture rules").
    HIP_TRY(hipDeviceSynchronize()); // nothing in flight may still use the buffers that are about to be replaced
    if (jobs > W.ws_jobs) {
        hipFree(W.d_bara);
        hipFree(W.d_u);
        hipFree(W.d_ubar);
        W.d_bara = nullptr;
        W.d_u = nullptr;
        W.d_ubar = nullptr;
        W.ws_jobs = 0;
        size_t cap = (std::max<size_t>(jobs, 1024) + 63) / 64 * 64;
        HIP_TRY(hipMalloc(&W.d_bara, cap * e->bara_stride * sizeof(uint16_t)));
        HIP_TRY(hipMalloc(&W.d_u, cap * (kN + 1) * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&W.d_ubar, (cap + KS_GT) * (size_t)kN * sizeof(uint32_t)));
        HIP_TRY(hipMemset(W.d_ubar, 0, (cap + KS_GT) * (size_t)kN * sizeof(uint32_t)));
        // NULL-stream audit: ordered by the hipDeviceSynchronize two lines below.
        // the memset runs on the NULL stream and callers may launch on hipStreamNonBlocking streams, which do not wait for
        // it: without this the fill could land on key-switch operands the first batch has already written (seen as 7
        // mismatches in 5 000 host-path cases of tools/soak_parity.py once the persistent workers removed the thread
        // start-up delay that used to hide it)
        HIP_TRY(hipDeviceSynchronize());
        if (!W.d_acc_state) HIP_TRY(hipMalloc(&W.d_acc_state, (size_t)4 * e->num_cus * 2 * kN * sizeof(int32_t) * 4));
        W.ws_jobs = cap;
        e->ws_grows++;
    }
    if (descs > W.ws_descs) {
        hipFree(W.d_descs);
        if (W.h_descs) hipHostFree(W.h_descs);
        W.d_descs = W.h_descs = nullptr;
        W.ws_descs = 0;
        W.desc_pos = 0;
        size_t cap = std::max<size_t>(descs, 1024);
        HIP_TRY(hipMalloc(&W.d_descs, cap * sizeof(GateDesc)));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&W.h_descs), cap * sizeof(GateDesc), hipHostMallocDefault));
        W.ws_descs = cap;
        if (!W.ring_ev) HIP_TRY(hipEventCreateWithFlags(&W.ring_ev, hipEventDisableTiming));
        W.ring_ev_set = false;
        if (W.persist_pos == 0) { // no captured graph refers to the arena yet: it may grow with the ring
            hipFree(W.d_persist);
            if (W.h_persist) hipHostFree(W.h_persist);
            W.d_persist = W.h_persist = nullptr;
            W.persist_cap = 0;
            HIP_TRY(hipMalloc(&W.d_persist, 4 * cap * sizeof(GateDesc)));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&W.h_persist), 4 * cap * sizeof(GateDesc), hipHostMallocDefault));
            W.persist_cap = 4 * cap;
        }
        e->ws_grows++;
    }
    if (mixed > W.ws_mixed) {
        hipFree(W.d_mixed);
        if (W.h_perm) hipHostFree(W.h_perm);
        W.d_mixed = nullptr;
        W.h_perm = nullptr;
        W.ws_mixed = 0;
        const size_t rows_bytes = mixed * ((size_t)e->p.n + 1) * 4;
        HIP_TRY(hipMalloc(&W.d_mixed, 4 * rows_bytes + mixed * 4));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&W.h_perm), mixed * 4, hipHostMallocDefault));
        if (!W.perm_ev) HIP_TRY(hipEventCreateWithFlags(&W.perm_ev, hipEventDisableTiming));
        W.ws_mixed = mixed;
        e->ws_grows++;
    }
    return EOC_OK;
}

extern "C" int eoc_engine_reserve(eoc_engine *e, size_t max_jobs, size_t max_descs, size_t max_mixed_rows)
{
    if (!e) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    int rc = ensure_ws(e, e->ws, max_jobs, max_descs, max_mixed_rows);
    if (rc) return rc;
    e->ws_grows = 0;
    return EOC_OK;
}

// ---- raw buffers -----------------------------------------------------------------------------
extern "C" int eoc_device_alloc(eoc_engine *e, size_t bytes, void **d_ptr)
{
    if (!e || !d_ptr) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMalloc(d_ptr, bytes));
    return EOC_OK;
}
extern "C" int eoc_device_free(eoc_engine *e, void *d_ptr)
{
    if (!e) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipFree(d_ptr));
    return EOC_OK;
}
extern "C" int eoc_host_to_device(eoc_engine *e, void *d_dst, const void *src, size_t bytes)
{
    if (!e || !d_dst || !src) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return EOC_OK;
}
extern "C" int eoc_device_to_host(eoc_engine *e, void *dst, const void *d_src, size_t bytes)
{
    if (!e || !dst || !d_src) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return EOC_OK;
}
extern "C" int eoc_engine_synchronize(eoc_engine *e)
{
    if (!e) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    return EOC_OK;
}

// ---- transforms (debug / key load) ----------------------------------------------------------
static int launch_fft_fwd(eoc_engine *e, const int32_t *d_polys, double *d_specs, size_t count, double scale,
                          hipStream_t st)
{
    dim3 grid((unsigned)((count + 3) / 4));
    hipLaunchKernelGGL(k_fft_fwd_polys, grid, dim3(256), 0, st, d_polys, d_specs, count, e->d_tw, e->d_twist, scale);
    HIP_TRY(hipGetLastError());
    return EOC_OK;
}

extern "C" int eoc_dbg_fft_fwd_device(eoc_engine *e, const int32_t *d_polys, double *d_specs, size_t count,
                                      void *hip_stream)
{
    if (!e || !d_polys || !d_specs) return EOC_ERR_ARG;
    if (!count) return EOC_OK;
    HIP_TRY(hipSetDevice(e->device));
    return launch_fft_fwd(e, d_polys, d_specs, count, 1.0, (hipStream_t)hip_stream);
}
extern "C" int eoc_dbg_fft_inv_device(eoc_engine *e, const double *d_specs, int32_t *d_polys, size_t count,
                                      void *hip_stream)
{
    if (!e || !d_polys || !d_specs) return EOC_ERR_ARG;
    if (!count) return EOC_OK;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    dim3 grid((unsigned)((count + 3) / 4));
    hipLaunchKernelGGL(k_fft_inv_polys, grid, dim3(256), 0, st, d_specs, d_polys, count, e->d_tw, e->d_twist);
    HIP_TRY(hipGetLastError());
    return EOC_OK;
}

// ---- cloud key ------------------------------------------------------------------------------
static int build_cloud_key_images(eoc_engine *e, const int32_t *bk, const int32_t *ksk, double *d_bkfft,
                                  int32_t *d_ksk)
{
    const eoc_params &p = e->p;
    // BK: upload the torus form, transform on the GPU (tGswToFFTConvert)
    const size_t npoly = (size_t)p.n * e->kpl * 2;
    int32_t *d_bk = nullptr;
    HIP_TRY(hipMalloc(&d_bk, npoly * kN * 4));
    if (hipMemcpy(d_bk, bk, npoly * kN * 4, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(d_bk);
        eoc_set_error("cloud key upload failed");
        return EOC_ERR_HIP;
    }
    // the key image carries the inverse transform's 1/512 (exact power-of-two scaling)
    // NULL-stream audit: the blocking hipMemcpy above has completed when it returns; the transform below
    // runs on the NULL stream and is drained by the hipDeviceSynchronize that follows it, before any caller stream
    // (blocking or hipStreamNonBlocking) can be handed the image.
    int rc = launch_fft_fwd(e, d_bk, d_bkfft, npoly, 0x1p-9, nullptr);
    hipError_t se = hipDeviceSynchronize();
    hipFree(d_bk);
    if (rc) return rc;
    HIP_TRY(se);
    // KSK: [N*t*(base-1)][n+1]  ->  same rows padded with zeros to n1p
    const size_t rows = (size_t)kN * p.ks_t * (((size_t)1 << p.ks_basebit) - 1);
    // NULL-stream audit: fill and copy are both on the NULL stream (ordered among themselves), drained by the
    // hipDeviceSynchronize below -- a non-blocking stream cannot see a half-written key image.
    HIP_TRY(hipMemset(d_ksk, 0, eoc_ksk_dev_bytes(&p)));
    HIP_TRY(hipMemcpy2D(d_ksk, e->n1p * 4, ksk, (size_t)(p.n + 1) * 4, (size_t)(p.n + 1) * 4, rows,
                        hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    return EOC_OK;
}

extern "C" int eoc_engine_load_cloud_key(eoc_engine *e, const int32_t *bk, const int32_t *ksk)
{
    if (!e || !bk || !ksk) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    if (!e->own_keys) {
        HIP_TRY(hipMalloc(&e->d_bkfft, eoc_bkfft_bytes(&e->p)));
        HIP_TRY(hipMalloc(&e->d_ksk, eoc_ksk_dev_bytes(&e->p)));
        e->own_keys = true;
    }
    int rc = build_cloud_key_images(e, bk, ksk, e->d_bkfft, e->d_ksk);
    if (rc) return rc;
    e->bkfft = e->d_bkfft;
    e->ksk = e->d_ksk;
    return EOC_OK;
}

// same, into caller-owned device buffers (which the engine then uses); lets the host keep the
// images in its own allocations, e.g. torch tensors that an RCCL broadcast reads or fills
extern "C" int eoc_engine_build_cloud_key_device(eoc_engine *e, const int32_t *bk, const int32_t *ksk,
                                                 void *d_bkfft, void *d_ksk)
{
    if (!e || !bk || !ksk || !d_bkfft || !d_ksk) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    int rc = build_cloud_key_images(e, bk, ksk, static_cast<double *>(d_bkfft), static_cast<int32_t *>(d_ksk));
    if (rc) return rc;
    e->bkfft = static_cast<const double *>(d_bkfft);
    e->ksk = static_cast<const int32_t *>(d_ksk);
    return EOC_OK;
}

extern "C" int eoc_engine_set_cloud_key_device(eoc_engine *e, const void *d_bkfft, const void *d_ksk)
{
    if (!e || !d_bkfft || !d_ksk) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    e->bkfft = static_cast<const double *>(d_bkfft);
    e->ksk = static_cast<const int32_t *>(d_ksk);
    return EOC_OK;
}
// take ownership of device images the engine's allocator produced (eoc_device_alloc): used for key replicas
extern "C" int eoc_engine_adopt_cloud_key_device(eoc_engine *e, void *d_bkfft, void *d_ksk)
{
    if (!e || !d_bkfft || !d_ksk) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    if (e->own_keys) {
        hipFree(e->d_bkfft);
        hipFree(e->d_ksk);
    }
    e->d_bkfft = static_cast<double *>(d_bkfft);
    e->d_ksk = static_cast<int32_t *>(d_ksk);
    e->own_keys = true;
    e->bkfft = e->d_bkfft;
    e->ksk = e->d_ksk;
    return EOC_OK;
}
extern "C" int eoc_engine_cloud_key_device(eoc_engine *e, const void **d_bkfft, const void **d_ksk)
{
    if (!e || !e->bkfft || !e->ksk) return EOC_ERR_NO_KEY;
    if (d_bkfft) *d_bkfft = e->bkfft;
    if (d_ksk) *d_ksk = e->ksk;
    return EOC_OK;
}

// ---- launch helpers -------------------------------------------------------------------------
typedef eoc_engine::Workspace WS;

// One launch holds at most `br_slice` jobs (default: what is resident at once, four workgroups per CU): a launch in
// which every workgroup is resident from the start runs with the wave-priority alternation and finishes all its
// workgroups within half a per cent of each other (3.0 ms per 1024 jobs), while a launch of several rounds settles at
// a 10 % lower rate (the arbiter's age bias), so wide levels are cut into back-to-back single-round launches.
static int launch_blind_rotate(eoc_engine *e, WS &W, uint32_t njobs_total, hipStream_t st,
                               const GateDesc *fold_descs = nullptr, uint32_t fold_S = 0, const GateDesc *inline_desc = nullptr,
                               bool fold_prep = true)
{
    SpanGuard span(e, st, KIND_BLIND_ROTATE);
    // Two kernel shapes (kernels.hip.h).  The pair kernel (one ciphertext = one wave pair) fills the chip with 4 x CUs
    // ciphertexts; the wide kernel (one ciphertext = one wave, gadget length 2 only) with 8 x CUs, and saves the partial-chain
    // exchange and its barriers.  A level wider than the pair kernel's resident set runs as full wide launches plus a
    // remainder: on the pair kernel when it fits its resident set (a half-empty wide launch has one wave per SIMD and runs
    // at 0.7 of the pair kernel's rate), on the wide kernel otherwise (tools/wide_sweep.py).
    const uint32_t resident_pair = 4u * (uint32_t)e->num_cus, resident_wide = 8u * (uint32_t)e->num_cus;
    const bool can_wide = e->p.l == 2 && e->br_wide != 0;
    const bool force_wide = can_wide && e->br_wide > 0;
    struct Seg { uint32_t off, njobs; bool wide; };
    std::vector<Seg> segs;
    segs.reserve(njobs_total / resident_pair + 2);
    const int auto_parts = e->kpl * 2 * kNH * 16 > 80 * 1024 ? 2 : 1;
    const int parts = std::max(1, std::min(e->br_parts > 0 ? e->br_parts : auto_parts, e->p.n));
    {
        auto even = [&](uint32_t off, uint32_t total, uint32_t slice, bool wide) { // even slices: 1536 jobs as 768 + 768
            if (parts > 1 && slice > 16u * (uint32_t)e->num_cus) slice = 16u * (uint32_t)e->num_cus; // acc_state capacity
            const uint32_t nsl = (total + slice - 1) / slice;
            slice = (total + nsl - 1) / nsl;
            for (uint32_t o = 0; o < total; o += slice) segs.push_back({off + o, std::min(slice, total - o), wide});
        };
        if (e->br_slice != 0) // diagnostics: fixed slice (> 0) or one launch (< 0), one kernel shape
            even(0, njobs_total, e->br_slice > 0 ? (uint32_t)e->br_slice : njobs_total, force_wide);
        else if (force_wide)
            even(0, njobs_total, resident_wide, true);
        else if (!can_wide || njobs_total <= resident_pair)
            even(0, njobs_total, resident_pair, false);
        else {
            const uint32_t full = njobs_total / resident_wide * resident_wide, rem = njobs_total - full;
            if (full) even(0, full, resident_wide, true);
            if (rem) even(full, rem, rem <= resident_pair ? resident_pair : resident_wide, rem > resident_pair);
        }
    }
    for (const Seg &sg : segs) {
        const uint32_t off = sg.off, njobs = sg.njobs;
        const bool wide = sg.wide;
        const uint32_t resident = wide ? resident_wide : resident_pair;
        for (int part = 0; part < parts; part++) {
            BRArgs a{};
            a.bkfft = e->bkfft;
            a.bara = W.d_bara + (size_t)off * e->bara_stride;
            a.u = W.d_u + (size_t)off * (kN + 1);
            a.njobs = njobs;
            a.n = e->p.n;
            a.Bgbit = e->p.Bgbit;
            a.bara_stride = e->bara_stride;
            a.mu = (int32_t)(1u << 29);
            a.stamps = e->d_stamps;
            a.ks_descs = fold_descs;
            a.prep = fold_descs != nullptr && fold_prep; // false: k_prepare wrote the rotation amounts (three-operand gates)
            a.inline_desc = inline_desc != nullptr;
            if (inline_desc) {
                a.desc0 = *inline_desc;
                a.ks_descs = reinterpret_cast<const GateDesc *>(W.d_descs); // non-null = fold; never dereferenced
                a.prep = 1;
            }
            a.ubar = W.d_ubar;
            a.ks_S = fold_S ? fold_S : 1;
            a.ks_prec_offset = 1u << (32 - (1 + e->p.ks_basebit * e->p.ks_t));
            a.job0 = off;
            a.step_begin = (int)((long long)e->p.n * part / parts);
            a.step_end = (int)((long long)e->p.n * (part + 1) / parts);
            a.acc_state = W.d_acc_state;
            // priority alternation pays only when every workgroup is resident from the start (four per CU)
            a.prio_duty = (e->prio_duty_override != INT32_MIN) ? e->prio_duty_override
                          : (njobs <= resident ? kPrioDuty : e->prio_multi);
            dim3 grid(njobs), block(128);
            // SABAR: the rotation amounts read back by scalar loads (EOC_TFHE_SCALAR_ABAR=1) instead of vector loads
#define EOC_BR_LAUNCH(KERNEL_, LDS_, ...)                                                                          \
    do {                                                                                                          \
        if (e->scalar_abar) hipLaunchKernelGGL((KERNEL_<__VA_ARGS__, true>), grid, block, LDS_, st, a, e->d_tw, e->d_twist);  \
        else hipLaunchKernelGGL((KERNEL_<__VA_ARGS__, false>), grid, block, LDS_, st, a, e->d_tw, e->d_twist);     \
    } while (0)
            if (wide) {
                grid = dim3((njobs + kBRWideJobsPerWG - 1) / kBRWideJobsPerWG);
                block = dim3(64 * kBRWideJobsPerWG);
                if (e->p.Bgbit == 10) EOC_BR_LAUNCH(k_blind_rotate_wide, kBRWideLds, 10); // Set A
                else EOC_BR_LAUNCH(k_blind_rotate_wide, kBRWideLds, 0);
                e->br_wide_launches++;
            } else if (e->p.l == 2 && e->p.Bgbit == 10) EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 2, 10); // Set A
            else if (e->p.l == 3 && e->p.Bgbit == 7) EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 3, 7);     // Set B
            else
                switch (e->p.l) {
                case 1: EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 1, 0); break;
                case 2: EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 2, 0); break;
                case 3: EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 3, 0); break;
                case 4: EOC_BR_LAUNCH(k_blind_rotate, kBRLds, 4, 0); break;
                default: return EOC_ERR_ARG;
                }
#undef EOC_BR_LAUNCH
            HIP_TRY(hipGetLastError());
            e->br_launches++;
        }
    }
    return EOC_OK;
}

static int launch_keyswitch(eoc_engine *e, WS &W, const GateDesc *d_descs, uint32_t ngates, uint32_t S, hipStream_t st,
                            bool init_done = false, const GateDesc *inline_desc = nullptr)
{
    KSArgs a;
    a.inline_desc = inline_desc != nullptr;
    if (inline_desc) a.desc0 = *inline_desc;
    else a.desc0 = GateDesc{0, 0, nullptr, nullptr, nullptr, nullptr};
    a.ksk = e->ksk;
    a.u = W.d_u;
    a.ubar = W.d_ubar;
    a.n = e->p.n;
    a.n1p = (int)e->n1p;
    a.t = e->p.ks_t;
    a.basebit = e->p.ks_basebit;
    a.S = S;
    a.mu = (int32_t)(1u << 29);
    SpanGuard span(e, st, KIND_KEYSWITCH);
    if (!init_done) { // levels with MUX (two extracted samples are summed) and the stand-alone key switch
        hipLaunchKernelGGL(k_ks_init, dim3(S, ngates), dim3(256), 0, st, d_descs, a);
        HIP_TRY(hipGetLastError());
    }
    const uint32_t ntiles = (S + KS_GT - 1) / KS_GT;
    const int ncb = (int)(e->n1p / 64); // blocks of 64 key columns: 4 (n <= 255), 8 (Set A), 12 (Set B), 16
    const int bb = e->p.ks_basebit, t = e->p.ks_t;
#define EOC_KS_LAUNCH(TT, NWV, IWV)                                                                     \
    do {                                                                                                  \
        auto kfn = k_keyswitch_waves<TT, NWV, IWV>;                                                       \
        constexpr int lds = KS3Cfg<TT, NWV, IWV>::LDS_BYTES;                                              \
        constexpr int ns = KS3Cfg<TT, NWV, IWV>::NS;                                                      \
        hipLaunchKernelGGL(kfn, dim3(ntiles * (unsigned)ncb * ns, ngates), dim3(64 * NWV), lds, st, d_descs, a); \
    } while (0)
    // basebit 2, t 8 (both default sets): waves per workgroup x indices per wave chosen so that 1024 gates give every SIMD
    // its 3-4 waves: 4096 waves for n1p = 256 / 512 / 1024, 3072 (three 4-wave workgroups per CU) for n1p = 768
    const bool fast = bb == 2 && t == 8 && e->ks_waves_ok;
    if (fast && ncb == 4) EOC_KS_LAUNCH(8, 8, 16);
    else if (fast && ncb == 8) EOC_KS_LAUNCH(8, 8, 32);
    else if (fast && ncb == 12) EOC_KS_LAUNCH(8, 4, 64);
    else if (fast && ncb == 16) EOC_KS_LAUNCH(8, 8, 64);
    else // any other shape (no default set has one): the plain one-thread-per-word form
        hipLaunchKernelGGL(k_keyswitch_generic, dim3(S, ngates), dim3(256), 0, st, d_descs, a);
#undef EOC_KS_LAUNCH
    HIP_TRY(hipGetLastError());
    return EOC_OK;
}

static inline bool op_const(int op) { return op == OP_CONST0 || op == OP_CONST1; }
static inline bool op_free(int op) { return op == OP_NOT || op == OP_COPY || op_const(op); }
static inline bool op_valid(int op) { return (op >= 0 && op <= OP_XOR3); }
static inline bool op_lin3(int op) { return op == OP_MAJ || op == OP_XOR3; } // one bootstrap, three-operand linear stage
static inline int op_inputs(int op) { return op_const(op) ? 0 : (op_free(op) ? 1 : (op == OP_MUX || op_lin3(op) ? 3 : 2)); }

// kernels index gates with a grid dimension (y or z <= 65535): wider levels are cut into slices of this many gates
constexpr size_t kMaxGatesPerLaunch = 32768;

// Descriptors of one launch group: written into the pinned host ring, copied to the same position of the device ring
// ahead of the kernels.  The ring only wraps after a stream synchronisation, so a slot is never overwritten while a
// copy that reads it may still be in flight (sized by ensure_ws for everything a call sends, the wrap is rare).
static int push_descs(WS &W, const GateDesc *src, size_t count, hipStream_t st, GateDesc **d_out)
{
    if (stream_is_capturing(st)) {
        if (W.persist_pos + count > W.persist_cap) {
            eoc_set_error("graph capture: descriptor arena exhausted (%zu of %zu used, %zu wanted); "
                          "eoc_engine_reserve a larger max_descs before capturing", W.persist_pos, W.persist_cap, count);
            return EOC_ERR_STATE;
        }
        memcpy(W.h_persist + W.persist_pos, src, count * sizeof(GateDesc));
        HIP_TRY(hipMemcpyAsync(W.d_persist + W.persist_pos, W.h_persist + W.persist_pos, count * sizeof(GateDesc),
                               hipMemcpyHostToDevice, st));
        *d_out = W.d_persist + W.persist_pos;
        W.persist_pos += count;
        return EOC_OK;
    }
    if (count > W.ws_descs) {
        eoc_set_error("internal: descriptor ring too small (%zu > %zu)", count, W.ws_descs);
        return EOC_ERR_STATE;
    }
    if (W.desc_pos + count > W.ws_descs) {
        // wrap (rare: the ring is sized for everything a call sends).  The slots about to be rewritten were read by copies
        // and kernels of earlier calls: wait for the event ring_mark recorded behind the LAST consumers of the ring's final
        // quarter.  An engine's kernels run on one stream at a time (the workspace is shared: callers order their streams),
        // so that event is behind every earlier consumer too; only the engine's own work is waited for, not the device --
        // a neighbouring batch's copy streams and captures elsewhere in the process are left alone.
        if (W.ring_ev_set) HIP_TRY(hipEventSynchronize(W.ring_ev));
        else HIP_TRY(hipDeviceSynchronize()); // one push filled three quarters of the ring by itself: no mark yet
        // The mark covers what was queued BEFORE it.  Descriptor copies queued behind it -- earlier pushes of the SAME level
        // (a free-gate push followed by boot slices), or levels that ended before the ring was three quarters full -- may still
        // be reading their pinned slots: drain this stream too (only this engine's stream, still no device-wide
        // synchronise; in the common case -- the wrap is the first push after the mark -- nothing is queued and this is skipped)
        if (W.ring_ev_set && W.ring_pushed_since_mark) HIP_TRY(hipStreamSynchronize(st));
        W.ring_ev_set = false;
        W.desc_pos = 0;
    }
    memcpy(W.h_descs + W.desc_pos, src, count * sizeof(GateDesc));
    HIP_TRY(hipMemcpyAsync(W.d_descs + W.desc_pos, W.h_descs + W.desc_pos, count * sizeof(GateDesc),
                           hipMemcpyHostToDevice, st));
    *d_out = W.d_descs + W.desc_pos;
    W.desc_pos += count;
    W.ring_pushed_since_mark = true;
    return EOC_OK;
}

// called behind the kernels that consume pushed descriptors: marks the stream position the next wrap-around waits for
// (only once the ring is three quarters full: one event record per ~thousand launches; never under capture -- captured
// descriptors live in the arena)
static void ring_mark(WS &W, hipStream_t st)
{
    if (!W.ring_ev || W.desc_pos * 4 < W.ws_descs * 3 || stream_is_capturing(st)) return;
    if (hipEventRecord(W.ring_ev, st) == hipSuccess) {
        W.ring_ev_set = true;
        W.ring_pushed_since_mark = false;
    }
}

// One "level": a set of gates that all run over the same S instances.  descs are host-side and
// carry device pointers; free gates and bootstrapped gates are separated here.
static int run_level(eoc_engine *e, WS &W, std::vector<GateDesc> &boot, std::vector<GateDesc> &freeg, size_t S,
                     hipStream_t st)
{
    const int n = e->p.n;
    for (size_t g0 = 0; g0 < freeg.size(); g0 += kMaxGatesPerLaunch) {
        const size_t cnt = std::min(kMaxGatesPerLaunch, freeg.size() - g0);
        GateDesc *dd = nullptr;
        int rc = push_descs(W, freeg.data() + g0, cnt, st, &dd);
        if (rc) return rc;
        size_t total = S * (size_t)(n + 1);
        dim3 grid((unsigned)((total + 255) / 256), (unsigned)cnt);
        hipLaunchKernelGGL(k_free_gates, grid, dim3(256), 0, st, dd, total, n + 1, (int32_t)(1u << 29));
        HIP_TRY(hipGetLastError());
    }
    // bootstrapped gates, in slices whose job count fits the workspace and whose gate count fits a grid dimension
    size_t g0 = 0;
    while (g0 < boot.size()) {
        uint32_t jobs = 0;
        bool any_mux = false, any_lin3 = false;
        size_t g1 = g0;
        while (g1 < boot.size() && g1 - g0 < kMaxGatesPerLaunch) {
            const uint32_t w = (uint32_t)S * (boot[g1].op == OP_MUX ? 2u : 1u);
            if (g1 > g0 && (size_t)jobs + w > W.ws_jobs) break;
            boot[g1].job_base = jobs;
            jobs += w;
            any_mux |= boot[g1].op == OP_MUX;
            any_lin3 |= op_lin3(boot[g1].op);
            g1++;
        }
        if (jobs > W.ws_jobs) {
            eoc_set_error("internal: workspace too small for one gate (%u jobs > %zu)", jobs, W.ws_jobs);
            return EOC_ERR_STATE;
        }
        const size_t cnt = g1 - g0;
        // a level of ONE gate without MUX -- the plain eoc_gate_batch_device call -- sends its descriptor as a kernel
        // argument: no copy into the descriptor ring precedes the two launches (and nothing is consumed from the ring)
        const bool inline_one = cnt == 1 && !any_mux && !any_lin3 && !e->no_fold;
        GateDesc *dd = nullptr;
        int rc = inline_one ? EOC_OK : push_descs(W, boot.data() + g0, cnt, st, &dd);
        if (rc) return rc;
        // without MUX every gate has S jobs (job = gate * S + instance): the blind rotation's prologue derives its own
        // rotation amounts from the operand rows (k_prepare folded away) and its epilogue sets the key switch up
        // (k_ks_init folded away) -- one launch per level besides the key switch; EOC_TFHE_NO_FOLD=1 keeps the separate
        // launches (diagnostics)
        // The extension gates MAJ / XOR3 (three-operand linear stage) take their rotation amounts from k_prepare -- the
        // folded prologue stays the two-operand code it was -- but keep the folded key-switch set-up (every gate of the slice
        // has S jobs: job = gate * S + instance)
        const bool fold = !any_mux && !e->no_fold;
        const bool fold_prep = fold && !any_lin3;
        if (!fold_prep) {
            dim3 grid((unsigned)(S * (any_mux ? 2 : 1)), (unsigned)((n + 1 + 255) / 256), (unsigned)cnt);
            SpanGuard span(e, st, KIND_PREPARE);
            hipLaunchKernelGGL(k_prepare, grid, dim3(256), 0, st, dd, n, (uint32_t)S, W.d_bara, e->bara_stride);
            HIP_TRY(hipGetLastError());
        }
        const GateDesc *one = inline_one ? boot.data() + g0 : nullptr;
        rc = launch_blind_rotate(e, W, jobs, st, fold ? dd : nullptr, (uint32_t)S, one, fold_prep);
        if (rc) return rc;
        rc = launch_keyswitch(e, W, dd, (uint32_t)cnt, (uint32_t)S, st, fold, one);
        if (rc) return rc;
        e->stats[0] += 1;
        e->stats[1] += jobs;
        e->stats[2] += S * cnt;
        g0 = g1;
    }
    ring_mark(W, st);
    return EOC_OK;
}

// One POOL: bootstrapped gate groups that are independent of each other but run over DIFFERENT numbers of rows (the
// opcode runs of a mixed batch: the two-input block over j rows, the MUX run over m rows -- 2 m blind rotations).  Every
// job is "a row of rotation amounts", so all groups share ONE blind rotation over the concatenated jobs -- one partly
// filled last launch per call instead of one per group -- between a k_prepare and a key-switch set-up per group (which is
// where the groups differ: row count, linear stage, one extracted sample or the sum of two).  Bit-identical to running
// the groups as levels of their own: a job's result does not depend on its position in the launch.
struct PoolItem {
    GateDesc d;
    size_t S;
};
static int run_pool(eoc_engine *e, WS &W, std::vector<PoolItem> &pool, hipStream_t st)
{
    const int n = e->p.n;
    size_t jobs = 0;
    std::vector<GateDesc> descs;
    descs.reserve(pool.size());
    for (PoolItem &it : pool) {
        it.d.job_base = (uint32_t)jobs;
        jobs += it.S * (it.d.op == OP_MUX ? 2 : 1);
        descs.push_back(it.d);
    }
    if (jobs > W.ws_jobs || jobs > 0xFFFFFFFFull) {
        eoc_set_error("internal: workspace too small for a pooled level (%zu jobs > %zu)", jobs, W.ws_jobs);
        return EOC_ERR_STATE;
    }
    GateDesc *dd = nullptr;
    int rc = push_descs(W, descs.data(), descs.size(), st, &dd);
    if (rc) return rc;
    {
        SpanGuard span(e, st, KIND_PREPARE);
        for (size_t k = 0; k < pool.size(); k++) {
            const size_t S = pool[k].S;
            dim3 grid((unsigned)(S * (pool[k].d.op == OP_MUX ? 2 : 1)), (unsigned)((n + 1 + 255) / 256), 1);
            hipLaunchKernelGGL(k_prepare, grid, dim3(256), 0, st, dd + k, n, (uint32_t)S, W.d_bara, e->bara_stride);
        }
        HIP_TRY(hipGetLastError());
    }
    rc = launch_blind_rotate(e, W, (uint32_t)jobs, st);
    if (rc) return rc;
    for (size_t k = 0; k < pool.size(); k++) {
        rc = launch_keyswitch(e, W, dd + k, 1, (uint32_t)pool[k].S, st);
        if (rc) return rc;
        e->stats[2] += pool[k].S;
    }
    e->stats[0] += 1;
    e->stats[1] += jobs;
    ring_mark(W, st);
    return EOC_OK;
}

// ---- batch of independent gates --------------------------------------------------------------
static int gate_batch_ws(eoc_engine *e, WS &W, int op, const uint8_t *ops, const int32_t *d_in0, const int32_t *d_in1,
                         const int32_t *d_in2, int32_t *d_out, size_t count, hipStream_t st)
{
    const size_t stride = (size_t)e->p.n + 1;
    std::vector<GateDesc> boot, freeg;
    if (!ops) {
        if (!op_valid(op) || (!op_free(op) && !d_in1) || (op_inputs(op) == 3 && !d_in2)) {
            eoc_set_error("eoc_gate_batch_device: bad opcode %d or missing operand", op);
            return EOC_ERR_ARG;
        }
        GateDesc d{op, 0, d_in0, d_in1, d_in2, d_out};
        (op_free(op) ? freeg : boot).push_back(d);
        int rc = ensure_ws(e, W, count * (op == OP_MUX ? 2 : 1), 64, 0, st);
        if (rc) return rc;
        return run_level(e, W, boot, freeg, count, st);
    }
    // mixed batch.  Every maximal run of equal opcodes is one batch; when the caller's order has many
    // runs the rows are first gathered into opcode-sorted order on the device (stable, so equal opcodes keep
    // their relative order), evaluated run by run, and scattered back.
    for (size_t k = 0; k < count; k++) {
        if (!op_valid(ops[k])) {
            eoc_set_error("eoc_gate_batch_device: bad opcode %d at %zu", (int)ops[k], k);
            return EOC_ERR_ARG;
        }
        if ((!op_free(ops[k]) && !d_in1) || (op_inputs(ops[k]) == 3 && !d_in2) || (!op_const(ops[k]) && !d_in0)) {
            eoc_set_error("eoc_gate_batch_device: missing operand for opcode %d", (int)ops[k]);
            return EOC_ERR_ARG;
        }
    }
    size_t runs = 1;
    for (size_t k = 1; k < count; k++) runs += ops[k] != ops[k - 1];
    std::vector<uint8_t> sorted_ops;
    const uint8_t *run_ops = ops;
    const int32_t *in0 = d_in0, *in1 = d_in1, *in2 = d_in2;
    int32_t *out = d_out;
    const bool gather = runs > 15; // more runs than opcodes: sorting pays
    // every bootstrapped row of the call is one job of ONE blind rotation (run_pool), a MUX row two -- up to 2^20 jobs
    // (9.6 GB of extracted samples and rotation amounts: the bound eoc_circuit_run_device puts on a level); a wider call,
    // or EOC_TFHE_NO_POOL=1 (diagnostics), runs every opcode group as a level of its own, as before round 6
    size_t max_jobs = 0;
    for (size_t k = 0; k < count; k++) max_jobs += ops[k] == OP_MUX ? 2 : (op_free(ops[k]) ? 0 : 1);
    const bool use_pool = !e->no_pool && max_jobs <= ((size_t)1 << 20);
    if (!use_pool) {
        size_t cnt_op[OP_XOR3 + 1] = {0}, run = 0;
        max_jobs = 0;
        for (size_t k = 0; k < count; k++) {
            cnt_op[ops[k]]++;
            run = (k && ops[k] == ops[k - 1]) ? run + 1 : 1;
            if (!gather) max_jobs = std::max(max_jobs, run * (ops[k] == OP_MUX ? 2 : 1));
        }
        if (gather) {
            size_t two_input = 0;
            for (int o = 0; o < OP_MUX; o++) two_input += cnt_op[o];
            max_jobs = std::max({two_input, 2 * cnt_op[OP_MUX], cnt_op[OP_MAJ], cnt_op[OP_XOR3]});
        }
    }
    if (gather && count > (size_t)kPermIndexMask) {
        eoc_set_error("eoc_gate_batch_device: a mixed batch holds at most %u gates", kPermIndexMask);
        return EOC_ERR_ARG;
    }
    int rc = ensure_ws(e, W, max_jobs, 2 * (gather ? OP_XOR3 + 1 : runs) + 64, gather ? count : 0, st);
    if (rc) return rc;
    uint32_t *d_perm = nullptr;
    if (gather) {
        // stable counting sort by opcode on the host (the opcode array is a host array); the permutation goes through
        // the workspace's pinned buffer, so nothing here waits for the device
        uint32_t *perm = W.h_perm;
        d_perm = reinterpret_cast<uint32_t *>(W.d_mixed + 4 * W.ws_mixed * stride);
        const bool capturing = stream_is_capturing(st);
        if (capturing) {
            // A captured copy node reads its pinned source again at every replay, and h_perm is rewritten by the next
            // mixed batch: the permutation of a captured batch lives in the never-re-used capture arena instead (host and
            // device side), and perm_ev -- which would become a captured event -- is left alone.
            const size_t slots = (count * sizeof(uint32_t) + sizeof(GateDesc) - 1) / sizeof(GateDesc);
            if (W.persist_pos + slots > W.persist_cap) {
                eoc_set_error("graph capture: descriptor arena exhausted by a mixed batch's permutation (%zu of %zu slots "
                              "used, %zu wanted); eoc_engine_reserve a larger max_descs before capturing",
                              W.persist_pos, W.persist_cap, slots);
                return EOC_ERR_STATE;
            }
            perm = reinterpret_cast<uint32_t *>(W.h_persist + W.persist_pos);
            d_perm = reinterpret_cast<uint32_t *>(W.d_persist + W.persist_pos);
            W.persist_pos += slots;
        } else {
            HIP_TRY(hipEventSynchronize(W.perm_ev)); // the previous mixed batch's copy out of h_perm (long done, normally)
        }
        size_t bucket[OP_XOR3 + 2] = {0};
        for (size_t k = 0; k < count; k++) bucket[ops[k] + 1]++;
        for (int o = 1; o < OP_XOR3 + 2; o++) bucket[o] += bucket[o - 1];
        sorted_ops.resize(count);
        for (size_t k = 0; k < count; k++) {
            size_t pos = bucket[ops[k]]++;
            perm[pos] = (uint32_t)k | ((uint32_t)ops[k] << 28); // index | opcode: k_gather_rows masks, OP_MULTI reads the top
            sorted_ops[pos] = ops[k];
        }
        int32_t *g0 = W.d_mixed, *g1 = g0 + count * stride, *g2 = g1 + count * stride, *go = g2 + count * stride;
        HIP_TRY(hipMemcpyAsync(d_perm, perm, count * 4, hipMemcpyHostToDevice, st));
        if (!capturing) HIP_TRY(hipEventRecord(W.perm_ev, st));
        dim3 grid((unsigned)count, (unsigned)((stride + 255) / 256));
        if (d_in0) hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, st, d_in0, g0, d_perm, (int)stride, 0);
        if (d_in1) hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, st, d_in1, g1, d_perm, (int)stride, 0);
        if (d_in2) hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, st, d_in2, g2, d_perm, (int)stride, 0);
        HIP_TRY(hipGetLastError());
        in0 = d_in0 ? g0 : nullptr;
        in1 = d_in1 ? g1 : nullptr;
        in2 = d_in2 ? g2 : nullptr;
        out = go;
        run_ops = sorted_ops.data();
    }
    // The opcode runs are independent of each other.  Free runs (NOT / COPY / CONSTANT) go straight to k_free_gates; the
    // bootstrapped runs are collected and share ONE blind rotation (run_pool) -- unless there is only one, which keeps the
    // folded single-level path (its descriptor as a kernel argument, no k_prepare / k_ks_init launches).
    std::vector<PoolItem> pool;
    auto flush_one = [&](const GateDesc &d, size_t S) -> int {
        boot.clear();
        freeg.clear();
        (op_free(d.op) ? freeg : boot).push_back(d);
        return run_level(e, W, boot, freeg, S, st);
    };
    size_t i = 0;
    if (gather) {
        // opcode-sorted: the rows of the ten two-input opcodes come first and differ only in their linear stage -- one
        // group over all of them (a run of its own per opcode would end each in a partly filled launch)
        size_t j = 0;
        while (j < count && run_ops[j] < OP_MUX) j++;
        if (j > 0 && run_ops[0] != run_ops[j - 1]) {
            GateDesc d{OP_MULTI, 0, in0, in1, reinterpret_cast<const int32_t *>(d_perm), out};
            if (!use_pool) {
                rc = flush_one(d, j);
                if (rc) return rc;
            } else pool.push_back({d, j});
            i = j;
        }
    }
    while (i < count) {
        size_t j = i;
        while (j < count && run_ops[j] == run_ops[i]) j++;
        int o = run_ops[i];
        GateDesc d{o, 0, in0 ? in0 + i * stride : nullptr, in1 ? in1 + i * stride : nullptr,
                   in2 ? in2 + i * stride : nullptr, out + i * stride};
        if (op_free(o) || !use_pool) {
            rc = flush_one(d, j - i);
            if (rc) return rc;
        } else pool.push_back({d, j - i});
        i = j;
    }
    if (pool.size() == 1) rc = flush_one(pool[0].d, pool[0].S);
    else if (!pool.empty()) rc = run_pool(e, W, pool, st);
    if (rc) return rc;
    if (gather) {
        dim3 grid((unsigned)count, (unsigned)((stride + 255) / 256));
        hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, st, out, d_out, d_perm, (int)stride, 1);
        HIP_TRY(hipGetLastError());
    }
    return EOC_OK;
}

extern "C" int eoc_gate_batch_device(eoc_engine *e, int op, const uint8_t *ops, const int32_t *d_in0,
                                     const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, size_t count,
                                     void *hip_stream)
{
    if (!e || !d_out || (!d_in0 && !ops && !op_const(op))) {
        eoc_set_error("eoc_gate_batch_device: null argument");
        return EOC_ERR_ARG;
    }
    if (!count) return EOC_OK;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->bkfft || !e->ksk) {
        eoc_set_error("eoc_gate_batch_device: no cloud key loaded");
        return EOC_ERR_NO_KEY;
    }
    HIP_TRY(hipSetDevice(e->device));
    return gate_batch_ws(e, e->ws, op, ops, d_in0, d_in1, d_in2, d_out, count, (hipStream_t)hip_stream);
}

// ---- circuits -------------------------------------------------------------------------------
extern "C" size_t eoc_circuit_bootstraps(const eoc_gate *gates, size_t n_gates)
{
    size_t b = 0;
    for (size_t g = 0; g < n_gates; g++) {
        int op = gates[g].op;
        b += op == OP_MUX ? 2 : (op_free(op) ? 0 : 1);
    }
    return b;
}

extern "C" int eoc_circuit_run_device(eoc_engine *e, const eoc_gate *gates, size_t n_gates, int32_t *d_wires,
                                      size_t n_wires, size_t instances, void *hip_stream)
{
    if (!e || !gates || !d_wires) return EOC_ERR_ARG;
    if (!n_gates || !instances) return EOC_OK;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->bkfft || !e->ksk) {
        eoc_set_error("eoc_circuit_run_device: no cloud key loaded");
        return EOC_ERR_NO_KEY;
    }
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    WS &W = e->ws;
    // levelise: RAW, WAR and WAW hazards on wires (eoc_levelise, host.cpp -- the same levels eoc_netlist_cost prices)
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &q = gates[k];
        const int nin = op_valid(q.op) ? op_inputs(q.op) : 0;
        const int32_t ins[3] = {q.in0, q.in1, q.in2};
        if (!op_valid(q.op) || q.out < 0 || (size_t)q.out >= n_wires) {
            eoc_set_error("eoc_circuit_run_device: bad gate %zu", k);
            return EOC_ERR_ARG;
        }
        for (int a = 0; a < nin; a++)
            if (ins[a] < 0 || (size_t)ins[a] >= n_wires) {
                eoc_set_error("eoc_circuit_run_device: bad input wire in gate %zu", k);
                return EOC_ERR_ARG;
            }
    }
    std::vector<int> level(n_gates, 0);
    const int nlev = eoc_levelise(gates, n_gates, n_wires, level.data());
    std::vector<std::vector<GateDesc>> boot(nlev + 1), freeg(nlev + 1);
    const size_t wstride = instances * ((size_t)e->p.n + 1);
    size_t max_jobs = 0, max_gate_jobs = 0;
    for (size_t k = 0; k < n_gates; k++) {
        const eoc_gate &q = gates[k];
        GateDesc d{q.op, 0, q.in0 >= 0 ? d_wires + (size_t)q.in0 * wstride : nullptr,
                   q.in1 >= 0 ? d_wires + (size_t)q.in1 * wstride : nullptr,
                   q.in2 >= 0 ? d_wires + (size_t)q.in2 * wstride : nullptr, d_wires + (size_t)q.out * wstride};
        (op_free(q.op) ? freeg : boot)[level[k]].push_back(d);
    }
    for (int lv = 1; lv <= nlev; lv++) {
        size_t jobs = 0;
        for (auto &d : boot[lv]) {
            const size_t w = instances * (d.op == OP_MUX ? 2 : 1);
            jobs += w;
            max_gate_jobs = std::max(max_gate_jobs, w);
        }
        max_jobs = std::max(max_jobs, jobs);
    }
    // a level wider than the job cap is evaluated in slices (run_level), so the workspace is bounded: 2^20 blind
    // rotations in flight need 9.6 GB of extracted samples and rotation amounts; wider levels gain nothing
    const size_t job_cap = std::max<size_t>(max_gate_jobs, (size_t)1 << 20);
    int rc = ensure_ws(e, W, std::min(max_jobs, job_cap), n_gates + 64, 0, st);
    if (rc) return rc;
    for (int lv = 1; lv <= nlev; lv++) {
        rc = run_level(e, W, boot[lv], freeg[lv], instances, st);
        if (rc) return rc;
    }
    return EOC_OK;
}

// ---- building blocks ------------------------------------------------------------------------
extern "C" int eoc_blind_rotate_device(eoc_engine *e, const int32_t *d_t, int32_t *d_u, size_t count,
                                       void *hip_stream)
{
    if (!e || !d_t || !d_u) return EOC_ERR_ARG;
    if (!count) return EOC_OK;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->bkfft) return EOC_ERR_NO_KEY;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    WS &W = e->ws;
    int rc = ensure_ws(e, W, count, 64, 0, st);
    if (rc) return rc;
    GateDesc d{OP_RAW, 0, d_t, nullptr, nullptr, nullptr}, *dd = nullptr;
    rc = push_descs(W, &d, 1, st, &dd);
    if (rc) return rc;
    dim3 grid((unsigned)count, (unsigned)((e->p.n + 1 + 255) / 256), 1);
    hipLaunchKernelGGL(k_prepare, grid, dim3(256), 0, st, dd, e->p.n, (uint32_t)count, W.d_bara, e->bara_stride);
    HIP_TRY(hipGetLastError());
    rc = launch_blind_rotate(e, W, (uint32_t)count, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_u, W.d_u, count * (kN + 1) * 4, hipMemcpyDeviceToDevice, st));
    ring_mark(W, st);
    return EOC_OK;
}

extern "C" int eoc_keyswitch_device(eoc_engine *e, const int32_t *d_u, int32_t *d_out, size_t count,
                                    void *hip_stream)
{
    if (!e || !d_u || !d_out) return EOC_ERR_ARG;
    if (!count) return EOC_OK;
    std::lock_guard<std::mutex> g(e->mu);
    if (!e->ksk) return EOC_ERR_NO_KEY;
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;
    WS &W = e->ws;
    int rc = ensure_ws(e, W, count, 64, 0, st);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(W.d_u, d_u, count * (kN + 1) * 4, hipMemcpyDeviceToDevice, st));
    GateDesc d{OP_RAW, 0, nullptr, nullptr, nullptr, d_out}, *dd = nullptr;
    rc = push_descs(W, &d, 1, st, &dd);
    if (rc) return rc;
    rc = launch_keyswitch(e, W, dd, 1, (uint32_t)count, st);
    ring_mark(W, st);
    return rc;
}

extern "C" int eoc_engine_set_profiling(eoc_engine *e, int on)
{
    if (!e) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    e->profiling = on != 0;
    return EOC_OK;
}

// drains the recorded event pairs (synchronises the device) and returns accumulated times
extern "C" int eoc_engine_kernel_times(eoc_engine *e, double ms[3], uint64_t launches[3], int reset)
{
    if (!e || !ms || !launches) return EOC_ERR_ARG;
    std::lock_guard<std::mutex> g(e->mu);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    for (auto &sp : e->spans) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) {
            e->kernel_ms[sp.kind] += t;
            e->kernel_launches[sp.kind] += 1;
        }
        hipEventDestroy(sp.a);
        hipEventDestroy(sp.b);
    }
    e->spans.clear();
    for (int i = 0; i < 3; i++) {
        ms[i] = e->kernel_ms[i];
        launches[i] = e->kernel_launches[i];
        if (reset) {
            e->kernel_ms[i] = 0;
            e->kernel_launches[i] = 0;
        }
    }
    return EOC_OK;
}

#ifdef EOC_STAMPS
// diagnostic build only: allocate / read back the in-kernel stamp buffer ([waves][16] cycle sums)
extern "C" int eoc_dbg_stamps(eoc_engine *e, size_t waves, unsigned long long *host_out)
{
    if (!e) return EOC_ERR_ARG;
    HIP_TRY(hipSetDevice(e->device));
    if (!host_out) {
        hipFree(e->d_stamps);
        HIP_TRY(hipMalloc(&e->d_stamps, waves * 16 * 8));
        HIP_TRY(hipMemset(e->d_stamps, 0, waves * 16 * 8));
        return EOC_OK;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, e->d_stamps, waves * 16 * 8, hipMemcpyDeviceToHost));
    return EOC_OK;
}
#endif

extern "C" int eoc_engine_stats(eoc_engine *e, uint64_t out[3])
{
    if (!e || !out) return EOC_ERR_ARG;
    for (int i = 0; i < 3; i++) out[i] = e->stats[i];
    return EOC_OK;
}
extern "C" uint64_t eoc_engine_workspace_grows(eoc_engine *e) { return e ? e->ws_grows : 0; }
extern "C" uint64_t eoc_engine_blind_rotate_launches(eoc_engine *e) { return e ? e->br_launches : 0; }
extern "C" uint64_t eoc_engine_blind_rotate_wide_launches(eoc_engine *e) { return e ? e->br_wide_launches : 0; }
extern "C" size_t eoc_engine_resident_jobs(eoc_engine *e)
{
    if (!e) return 0;
    const bool wide = e->p.l == 2 && e->br_wide != 0;
    return (size_t)(wide ? 8 : 4) * (size_t)e->num_cus;
}
extern "C" int eoc_engine_device(eoc_engine *e) { return e ? e->device : -1; }
extern "C" const eoc_params *eoc_engine_params(eoc_engine *e) { return e ? &e->p : nullptr; }
