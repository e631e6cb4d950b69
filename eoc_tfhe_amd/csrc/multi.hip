// multi.hip -- layer 2 of include/eoc_tfhe_gpu.h: the process-global GPU context behind the host-buffer batch API
// (eoc_gpu_init[_multi], eoc_upload_cloud_key, eoc_gate_batch, eoc_circuit_run, eoc_stats).
//
// Mirrors the reference's single global key context (globalSecretKey / globalPublicKey, ao-tfhe/eoc-tfhe-run.cpp:38-40,
// reached through the luaopen_tfhe registry, ao-tfhe/eoc-tfhe-bindings.c:128-148): ONE host process, ONE key, any number
// of GPUs behind it.  SURVEY.md 8e: gates and circuit instances are independent given the cloud key, so every call cuts
// its instances into contiguous blocks (the same blocks as eoc_tfhe_amd.distributed.shard), one block per device, driven
// by one host thread per device; the only collective is the one-time replication of the two key images, an RCCL
// broadcast over xGMI when the devices are distinct (librccl is loaded on demand), device-to-device copies otherwise.
//
// Host-buffer traffic of one device: persistent device buffers owned by the slot (no hipMalloc / hipFree per call).
// Caller buffers from eoc_host_alloc (pinned, device-mapped) are read in place by the linear-stage kernel and receive
// the result by one asynchronous copy; pageable caller buffers go through the runtime's staged copies.
#include "common.h"
#include "../../include/eoc_tfhe_gpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the library is dlopen'ed when a broadcast is wanted

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            eoc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                        \
            return EOC_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

// result rows -> caller's pinned, device-mapped buffer (a kernel writes over PCIe at twice the rate of the copy
// engine's 2 MB transfer and needs no separate queue hand-off)
__global__ __launch_bounds__(256) void k_copy_words(const int32_t *__restrict__ src, int32_t *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

namespace {

// One persistent host thread per engine (slots 1..n-1; slot 0's block runs on the calling thread).  A call posts a job
// to the workers whose blocks are non-empty and waits for them; nobody is woken for an empty block, and no thread is
// created or joined on the call path (round 2 spawned one std::thread per engine per call, also for the string API's
// one-gate calls where every block but the first is empty).
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, stop = false;
    int rc = 0;
    std::string err; // the worker thread's error message of the last job (eoc_last_error is per thread)
    uint64_t wakeups = 0;
    void run(int device)
    {
        (void)hipSetDevice(device);
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (!has_job) return; // stop, and nothing posted: a posted job is finished first, so that a thread inside
                                  // wait() always sees has_job cleared (a slot destroyed under a live caller)
            std::function<int()> j = std::move(job);
            lk.unlock();
            const int r = j();
            std::string msg = r ? eoc_last_error() : "";
            lk.lock();
            rc = r;
            err.swap(msg);
            has_job = false;
            wakeups++;
            cv.notify_all();
        }
    }
    void post(std::function<int()> j)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            has_job = true;
        }
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !has_job; });
        if (rc) eoc_adopt_error(err.c_str()); // the caller's eoc_last_error() now explains the code it is about to get
        return rc;
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    // A process may exit (return from main, exit(), the interpreter's shutdown) without eoc_gpu_shutdown / resetGateKey:
    // the static context then destroys its slots, and a std::thread that is still joinable there calls std::terminate
    // (SIGABRT at exit).  Stop and join here; the thread only waits on `cv`, it makes no HIP call on the way out.
    ~Worker() { shutdown(); }
};

struct Slot {
    int device = 0;
    std::unique_ptr<Worker> worker; // null for slot 0 and in single-engine contexts
    eoc_engine *e = nullptr;
    hipStream_t st[3] = {nullptr, nullptr, nullptr}; // [0] kernels, [1] H2D copies, [2] D2H copies
    std::vector<hipEvent_t> ev;                       // chunk hand-offs between the three (grown on demand, re-used)
    // persistent buffers of the gate-batch path: 3 inputs + 1 output, `cap_rows` rows each
    int32_t *d_io[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap_rows = 0;
    // persistent wire buffer of the circuit path
    int32_t *d_wires = nullptr;
    size_t cap_wire_ints = 0;
    uint64_t grows = 0; // buffer growths (0 in steady state)
    // asynchronous batch path (eoc_gate_batch_submit / _wait): two buffer sets, so that batch k + 1's operands arrive
    // while batch k computes and batch k's results leave under batch k + 1's kernels
    int32_t *d_as[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
    size_t cap_async = 0;
    hipEvent_t ev_as[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; // operands in, kernels done, results out
};

struct Pending { // one submitted, not yet awaited batch
    uint64_t ticket = 0;
    bool active = false;
    std::vector<int> slots; // engines that received a non-empty block
};

struct Global {
    std::mutex mu;
    std::vector<Slot> slots;
    uint64_t next_ticket = 1;
    Pending ring[2];
    eoc_params p{};
    double bcast_s = 0.0;
    std::string bcast_method = "none";
    bool key_loaded = false;
};
Global G;

// contiguous block of `rank` among `world` (eoc_tfhe_amd.distributed.shard)
void shard_range(size_t total, int rank, int world, size_t *lo, size_t *hi)
{
    size_t base = total / (size_t)world, rem = total % (size_t)world;
    size_t l = (size_t)rank * base + std::min<size_t>((size_t)rank, rem);
    *lo = l;
    *hi = l + base + ((size_t)rank < rem ? 1 : 0);
}

// Registry of the pinned buffers handed out by eoc_host_alloc: base -> (size, device-side address).  The batch calls
// look caller pointers up here instead of asking the runtime (hipPointerGetAttributes costs tens of microseconds per
// call); anything not found is treated as pageable memory.
struct PinnedBlock { size_t size; char *dev; };
std::mutex g_pin_mu;
std::map<uintptr_t, PinnedBlock> g_pinned;

// device-side address of a caller pointer inside an eoc_host_alloc block, or nullptr
void *mapped_address(const void *p)
{
    if (!p) return nullptr;
    std::lock_guard<std::mutex> g(g_pin_mu);
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto it = g_pinned.upper_bound(a);
    if (it == g_pinned.begin()) return nullptr;
    --it;
    if (a >= it->first + it->second.size) return nullptr;
    return it->second.dev + (a - it->first);
}
bool is_pinned(const void *p) { return !p || mapped_address(p) != nullptr; }

int slot_reserve_rows(Slot &s, size_t rows, size_t stride_ints)
{
    if (rows <= s.cap_rows) return EOC_OK;
    HIP_TRY(hipSetDevice(s.device));
    HIP_TRY(hipDeviceSynchronize());
    size_t cap = std::max<size_t>(rows, 1024);
    for (int k = 0; k < 4; k++) {
        hipFree(s.d_io[k]);
        s.d_io[k] = nullptr;
    }
    s.cap_rows = 0;
    for (int k = 0; k < 4; k++) HIP_TRY(hipMalloc(&s.d_io[k], cap * stride_ints * 4));
    s.cap_rows = cap;
    s.grows++;
    return EOC_OK;
}

// one device's block of a gate batch.  Pinned caller buffers (eoc_host_alloc) are mapped into the device's address
// space: the linear stage (k_prepare / k_free_gates) reads the operands straight from host memory over PCIe, so no
// H2D copy and no staging exist on that path; the result is produced in a persistent device buffer (the key switch
// accumulates with atomics) and a small kernel writes it into the caller's mapped buffer.  Pageable buffers take the
// runtime's staged copies into the persistent device buffers.  (Cutting a 1024-gate batch into chunks on two streams
// was measured and dropped: kernels of two streams do not run concurrently here, 2 x 512 gates take 4.65 ms against
// 3.29 ms for 1 x 1024; chunking is used from 8192 gates on, where every chunk fills the device by itself.)
int slot_gate_block(Slot &s, int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                    int32_t *out, size_t count, size_t stride_ints)
{
    if (!count) return EOC_OK;
    int rc = slot_reserve_rows(s, count, stride_ints);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s.device));
    const int32_t *h[3] = {in0, in1, in2};
    const bool zero_copy = !getenv("EOC_TFHE_NO_ZERO_COPY");
    bool pin[3] = {false, false, false};
    const int32_t *dmap[3] = {nullptr, nullptr, nullptr};
    bool all_pinned = true;
    for (int k = 0; k < 3; k++) {
        if (!h[k]) continue;
        dmap[k] = static_cast<const int32_t *>(mapped_address(h[k]));
        pin[k] = dmap[k] != nullptr;
        all_pinned &= pin[k];
    }
    int32_t *out_map = static_cast<int32_t *>(mapped_address(out));
    all_pinned &= out_map != nullptr;
    // Batches wider than one resident set, from pinned buffers: chunks of one resident set (1024 gates on the pair kernel,
    // 2048 where the one-wave-per-ciphertext kernel applies: one single-round blind-rotate launch each).  ALL kernels stay on one stream (kernels of different streams do not overlap on this device and
    // would only interleave their launches); the copy engines run beside them on two copy streams: every chunk's
    // operands are DMA'd ahead, and a chunk's result leaves while the next chunk computes.  Exposed: the first chunk's
    // H2D and the last chunk's D2H, 6 MB in all, whatever the batch size.
    size_t nchunks = 1;
    const size_t chunk = std::max<size_t>(1024, eoc_engine_resident_jobs(s.e));
    if (all_pinned && count > chunk) nchunks = (count + chunk - 1) / chunk;
    if (const char *c = getenv("EOC_TFHE_HOST_CHUNKS")) nchunks = std::min<size_t>(std::max(1, atoi(c)), count);
    if (nchunks > 1) {
        const size_t row_bytes = stride_ints * 4;
        while (s.ev.size() < 2 * nchunks) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            s.ev.push_back(e);
        }
        // An error anywhere in the pipeline must not leave copies into (or out of) the caller's buffers in flight after
        // the call has returned, nor kernels that still use d_io: every exit drains the three streams first.
        auto drain = [&](int code) {
            for (int k = 0; k < 3; k++) (void)hipStreamSynchronize(s.st[k]);
            return code;
        };
#define PIPE_TRY(expr)                                                                                        \
        do {                                                                                                  \
            hipError_t _e = (expr);                                                                           \
            if (_e != hipSuccess) {                                                                           \
                eoc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);    \
                return drain(EOC_ERR_HIP);                                                                    \
            }                                                                                                 \
        } while (0)
        for (size_t c = 0; c < nchunks; c++) { // operands of every chunk, queued back to back on the H2D stream
            size_t lo, hi;
            shard_range(count, (int)c, (int)nchunks, &lo, &hi);
            for (int k = 0; k < 3; k++)
                if (h[k])
                    PIPE_TRY(hipMemcpyAsync(s.d_io[k] + lo * stride_ints, h[k] + lo * stride_ints, (hi - lo) * row_bytes,
                                            hipMemcpyHostToDevice, s.st[1]));
            PIPE_TRY(hipEventRecord(s.ev[2 * c], s.st[1]));
        }
        for (size_t c = 0; c < nchunks; c++) {
            size_t lo, hi;
            shard_range(count, (int)c, (int)nchunks, &lo, &hi);
            const size_t cnt = hi - lo;
            PIPE_TRY(hipStreamWaitEvent(s.st[0], s.ev[2 * c], 0));
            rc = eoc_gate_batch_device(s.e, op, ops ? ops + lo : nullptr, h[0] ? s.d_io[0] + lo * stride_ints : nullptr,
                                       h[1] ? s.d_io[1] + lo * stride_ints : nullptr,
                                       h[2] ? s.d_io[2] + lo * stride_ints : nullptr, s.d_io[3] + lo * stride_ints, cnt, s.st[0]);
            if (rc) return drain(rc);
            PIPE_TRY(hipEventRecord(s.ev[2 * c + 1], s.st[0]));
            PIPE_TRY(hipStreamWaitEvent(s.st[2], s.ev[2 * c + 1], 0));
            PIPE_TRY(hipMemcpyAsync(out + lo * stride_ints, s.d_io[3] + lo * stride_ints, cnt * row_bytes,
                                    hipMemcpyDeviceToHost, s.st[2]));
        }
#undef PIPE_TRY
        for (int k = 2; k >= 0; k--) HIP_TRY(hipStreamSynchronize(s.st[k]));
        return EOC_OK;
    }
    hipStream_t st = s.st[0];
    const size_t bytes = count * stride_ints * 4;
    const int32_t *d[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++) {
        if (!h[k]) continue;
        if (zero_copy && dmap[k]) {
            d[k] = dmap[k]; // read in place over PCIe by k_prepare / k_free_gates
            continue;
        }
        HIP_TRY(hipMemcpyAsync(s.d_io[k], h[k], bytes, hipMemcpyHostToDevice, st));
        d[k] = s.d_io[k];
    }
    rc = eoc_gate_batch_device(s.e, op, ops, d[0], d[1], d[2], s.d_io[3], count, st);
    if (rc) {
        (void)hipStreamSynchronize(st); // no operand copy may outlive the call
        return rc;
    }
    if (zero_copy && out_map) {
        const size_t n = count * stride_ints;
        hipLaunchKernelGGL(k_copy_words, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st,
                           s.d_io[3], out_map, n);
        HIP_TRY(hipGetLastError());
    } else {
        HIP_TRY(hipMemcpyAsync(out, s.d_io[3], bytes, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return EOC_OK;
}

// one device's block of circuit instances [lo, hi) of a host wire array [n_wires][instances][stride]
int slot_circuit_block(Slot &s, const eoc_gate *gates, size_t n_gates, int32_t *wires, size_t n_wires, size_t instances,
                       size_t lo, size_t hi, size_t stride_ints)
{
    const size_t blk = hi - lo;
    if (!blk) return EOC_OK;
    HIP_TRY(hipSetDevice(s.device));
    const size_t need = n_wires * blk * stride_ints;
    if (need > s.cap_wire_ints) {
        HIP_TRY(hipDeviceSynchronize());
        hipFree(s.d_wires);
        s.d_wires = nullptr;
        s.cap_wire_ints = 0;
        HIP_TRY(hipMalloc(&s.d_wires, need * 4));
        s.cap_wire_ints = need;
        s.grows++;
    }
    hipStream_t st = s.st[0];
    // Only the wires that matter cross PCIe: host -> device the wires some gate READS BEFORE any gate has written them
    // (the circuit's inputs), device -> host the wires some gate WRITES; a wire nobody touches keeps the caller's bytes
    // and a wire that is overwritten before it is read is never sent (an 8-bit adder moves 16 of its 57 wires in and
    // 41 out).  Gates are evaluated in netlist order semantics (eoc_circuit_run_device levelises on hazards), so the
    // scan below is in that order.  Consecutive wires travel as one strided copy: row w of the device array holds
    // instances [lo, hi) of wire w.
    std::vector<uint8_t> live_in(n_wires, 0), written(n_wires, 0);
    for (size_t k = 0; k < n_gates; k++) {
        const int32_t ins[3] = {gates[k].in0, gates[k].in1, gates[k].in2};
        for (int a = 0; a < 3; a++)
            if (ins[a] >= 0 && (size_t)ins[a] < n_wires && !written[ins[a]]) live_in[ins[a]] = 1;
        if (gates[k].out >= 0 && (size_t)gates[k].out < n_wires) written[gates[k].out] = 1;
    }
    auto copy_runs = [&](const std::vector<uint8_t> &sel, bool to_device) -> int {
        for (size_t w = 0; w < n_wires;) {
            if (!sel[w]) {
                w++;
                continue;
            }
            size_t e = w;
            while (e < n_wires && sel[e]) e++;
            int32_t *dev = s.d_wires + w * blk * stride_ints;
            int32_t *host = wires + (w * instances + lo) * stride_ints;
            if (to_device)
                HIP_TRY(hipMemcpy2DAsync(dev, blk * stride_ints * 4, host, instances * stride_ints * 4, blk * stride_ints * 4,
                                         e - w, hipMemcpyHostToDevice, st));
            else
                HIP_TRY(hipMemcpy2DAsync(host, instances * stride_ints * 4, dev, blk * stride_ints * 4, blk * stride_ints * 4,
                                         e - w, hipMemcpyDeviceToHost, st));
            w = e;
        }
        return EOC_OK;
    };
    int rc = copy_runs(live_in, true);
    if (rc == EOC_OK) rc = eoc_circuit_run_device(s.e, gates, n_gates, s.d_wires, n_wires, blk, st);
    if (rc == EOC_OK) rc = copy_runs(written, false);
    if (rc) {
        (void)hipStreamSynchronize(st); // no copy may outlive the call
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    return EOC_OK;
}

void destroy_slots_locked()
{
    for (auto &s : G.slots)
        if (s.worker) s.worker->shutdown();
    for (auto &pd : G.ring) pd = Pending();
    for (auto &s : G.slots) {
        hipSetDevice(s.device);
        hipDeviceSynchronize();
        for (int r = 0; r < 2; r++) {
            for (int k = 0; k < 4; k++) hipFree(s.d_as[r][k]);
            for (int k = 0; k < 3; k++)
                if (s.ev_as[r][k]) hipEventDestroy(s.ev_as[r][k]);
        }
        for (int k = 0; k < 4; k++) hipFree(s.d_io[k]);
        hipFree(s.d_wires);
        for (int k = 0; k < 3; k++)
            if (s.st[k]) hipStreamDestroy(s.st[k]);
        for (auto e : s.ev) hipEventDestroy(e);
        eoc_engine_destroy(s.e);
    }
    G.slots.clear();
    G.key_loaded = false;
    G.bcast_s = 0.0;
    G.bcast_method = "none";
}

// ---- RCCL, loaded on demand ---------------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    const char *origin = "not loaded";
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load()
    {
        if (lib) return true;
        static const char *const names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // a process that already maps an RCCL (a torch-hosting harness maps torch/lib/librccl.so) re-uses that copy:
        // RTLD_NOLOAD returns a handle only if the library is resident, so two RCCLs never share one process
        for (const char *name : names) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (lib) {
                origin = "already mapped";
                break;
            }
        }
        if (!lib && dlsym(RTLD_DEFAULT, "ncclCommInitAll")) { // mapped under another path, symbols globally visible
            lib = dlopen(nullptr, RTLD_NOW);
            origin = "process symbols";
        }
        for (size_t k = 0; !lib && k < sizeof names / sizeof names[0]; k++) {
            lib = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
            if (lib) origin = names[k];
        }
        if (!lib) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        return CommInitAll && Broadcast && GroupStart && GroupEnd && CommDestroy && GetErrorString;
    }
} R;

// replicate the two key images of slot 0 onto every other slot.  Returns the method used.
int replicate_key_locked(const void *bk0, const void *ksk0, std::vector<void *> &bk, std::vector<void *> &ksk)
{
    const size_t nb = eoc_bkfft_bytes(&G.p), nk = eoc_ksk_dev_bytes(&G.p);
    const int n = (int)G.slots.size();
    bool distinct = true;
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) distinct &= G.slots[i].device != G.slots[j].device;
    const char *force = getenv("EOC_TFHE_KEY_BCAST"); // "rccl" | "copy"
    bool use_rccl = distinct && n > 1 && !(force && !strcmp(force, "copy"));
    if (use_rccl && !R.load()) {
        if (force && !strcmp(force, "rccl")) {
            const char *why = dlerror(); // NULL when dlopen succeeded and a dlsym failed
            eoc_set_error("EOC_TFHE_KEY_BCAST=rccl but librccl could not be loaded: %s", why ? why : "missing symbol");
            return EOC_ERR_STATE;
        }
        use_rccl = false;
    }
    if (use_rccl) {
        std::vector<ncclComm_t> comms(n);
        std::vector<int> devs(n);
        for (int i = 0; i < n; i++) devs[i] = G.slots[i].device;
        ncclResult_t r = R.CommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            eoc_set_error("ncclCommInitAll failed: %s; falling back to peer copies", R.GetErrorString(r));
            use_rccl = false;
        } else {
            for (int img = 0; img < 2 && r == ncclSuccess; img++) {
                R.GroupStart();
                for (int i = 0; i < n; i++) {
                    void *buf = img == 0 ? bk[i] : ksk[i];
                    r = R.Broadcast(buf, buf, img == 0 ? nb : nk, ncclChar, 0, comms[i], G.slots[i].st[0]);
                    if (r != ncclSuccess) break;
                }
                ncclResult_t r2 = R.GroupEnd();
                if (r == ncclSuccess) r = r2;
            }
            for (int i = 0; i < n; i++) {
                hipSetDevice(G.slots[i].device);
                hipStreamSynchronize(G.slots[i].st[0]);
            }
            for (int i = 0; i < n; i++) R.CommDestroy(comms[i]);
            if (r != ncclSuccess) {
                eoc_set_error("ncclBroadcast failed: %s", R.GetErrorString(r));
                return EOC_ERR_HIP;
            }
            G.bcast_method = "rccl";
            return EOC_OK;
        }
    }
    // device-to-device copies from slot 0 (the only form available when several engines share one device)
    for (int i = 1; i < n; i++) {
        HIP_TRY(hipSetDevice(G.slots[i].device));
        if (G.slots[i].device == G.slots[0].device) {
            HIP_TRY(hipMemcpyAsync(bk[i], bk0, nb, hipMemcpyDeviceToDevice, G.slots[i].st[0]));
            HIP_TRY(hipMemcpyAsync(ksk[i], ksk0, nk, hipMemcpyDeviceToDevice, G.slots[i].st[0]));
        } else {
            HIP_TRY(hipMemcpyPeerAsync(bk[i], G.slots[i].device, bk0, G.slots[0].device, nb, G.slots[i].st[0]));
            HIP_TRY(hipMemcpyPeerAsync(ksk[i], G.slots[i].device, ksk0, G.slots[0].device, nk, G.slots[i].st[0]));
        }
    }
    for (int i = 1; i < n; i++) {
        HIP_TRY(hipSetDevice(G.slots[i].device));
        HIP_TRY(hipStreamSynchronize(G.slots[i].st[0]));
    }
    G.bcast_method = n > 1 ? "peer-copy" : "none";
    return EOC_OK;
}

// fn(i, lo, hi) evaluates block [lo, hi) of `total` instances on slot i.  Blocks are eoc_shard_range's; empty blocks wake
// nobody; the first non-empty block runs on the calling thread (so a call that fits one block -- every one-gate call of
// the string API -- costs what it costs on a single-engine context), the others on their slots' persistent workers.
template <class F> int for_each_block(size_t total, F fn)
{
    const int n = (int)G.slots.size();
    if (n == 1) return fn(0, (size_t)0, total);
    int first = -1;
    std::vector<int> posted;
    for (int i = 0; i < n; i++) {
        size_t lo, hi;
        shard_range(total, i, n, &lo, &hi);
        if (hi == lo) continue;
        if (first < 0 || !G.slots[i].worker) {
            if (first < 0) first = i;
            continue;
        }
        G.slots[i].worker->post([fn, i, lo, hi] { return fn(i, lo, hi); });
        posted.push_back(i);
    }
    int rc = EOC_OK;
    for (int i = 0; i < n; i++) { // inline blocks: the first non-empty one, and any slot without a worker
        size_t lo, hi;
        shard_range(total, i, n, &lo, &hi);
        if (hi == lo || (i != first && G.slots[i].worker)) continue;
        const int r = fn(i, lo, hi);
        if (r && !rc) rc = r;
    }
    for (int i : posted) {
        const int r = G.slots[i].worker->wait();
        if (r && !rc) rc = r;
    }
    return rc;
}

// ---- asynchronous batches ------------------------------------------------------------------------------------------
// wait for one pending submission: its results are in the caller's buffers afterwards
int finish_pending_locked(Pending &pd)
{
    int rc = EOC_OK;
    if (!pd.active) return rc;
    const int r = (int)(pd.ticket & 1);
    for (int i : pd.slots) {
        Slot &s = G.slots[i];
        if (hipSetDevice(s.device) != hipSuccess || hipEventSynchronize(s.ev_as[r][2]) != hipSuccess) {
            eoc_set_error("eoc_gate_batch_wait: waiting for ticket %llu failed on engine %d", (unsigned long long)pd.ticket, i);
            rc = EOC_ERR_HIP;
        }
    }
    pd.active = false;
    return rc;
}
// every synchronous entry point first drains the asynchronous ones (they share streams and the engines' workspaces)
int drain_async_locked()
{
    int rc = EOC_OK;
    const int first = G.ring[0].ticket < G.ring[1].ticket ? 0 : 1;
    for (int k = 0; k < 2; k++) {
        const int r = finish_pending_locked(G.ring[(first + k) & 1]);
        if (r && !rc) rc = r;
    }
    return rc;
}

int slot_async_reserve(Slot &s, size_t rows, size_t stride_ints)
{
    HIP_TRY(hipSetDevice(s.device));
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++)
            if (!s.ev_as[r][k]) HIP_TRY(hipEventCreateWithFlags(&s.ev_as[r][k], hipEventDisableTiming));
    if (rows <= s.cap_async) return EOC_OK;
    HIP_TRY(hipDeviceSynchronize());
    const size_t cap = std::max<size_t>(rows, 1024);
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 4; k++) {
            hipFree(s.d_as[r][k]);
            s.d_as[r][k] = nullptr;
        }
    s.cap_async = 0;
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 4; k++) HIP_TRY(hipMalloc(&s.d_as[r][k], cap * stride_ints * 4));
    s.cap_async = cap;
    s.grows++;
    return EOC_OK;
}

// queue one device's block of a submitted batch on buffer set r: operands by DMA on the H2D stream, kernels on the
// kernel stream behind them, results by DMA on the D2H stream behind the kernels.  Nothing here waits for the device.
int slot_async_block(Slot &s, int r, int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                     int32_t *out, size_t count, size_t stride_ints)
{
    int rc = slot_async_reserve(s, count, stride_ints);
    if (rc) return rc;
    const int32_t *h[3] = {in0, in1, in2};
    const size_t bytes = count * stride_ints * 4;
    auto fail = [&](int code) {
        for (int k = 0; k < 3; k++) (void)hipStreamSynchronize(s.st[k]);
        return code;
    };
#define AS_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            eoc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);       \
            return fail(EOC_ERR_HIP);                                                                        \
        }                                                                                                    \
    } while (0)
    for (int k = 0; k < 3; k++)
        if (h[k]) AS_TRY(hipMemcpyAsync(s.d_as[r][k], h[k], bytes, hipMemcpyHostToDevice, s.st[1]));
    AS_TRY(hipEventRecord(s.ev_as[r][0], s.st[1]));
    AS_TRY(hipStreamWaitEvent(s.st[0], s.ev_as[r][0], 0));
    rc = eoc_gate_batch_device(s.e, op, ops, h[0] ? s.d_as[r][0] : nullptr, h[1] ? s.d_as[r][1] : nullptr,
                               h[2] ? s.d_as[r][2] : nullptr, s.d_as[r][3], count, s.st[0]);
    if (rc) return fail(rc);
    AS_TRY(hipEventRecord(s.ev_as[r][1], s.st[0]));
    AS_TRY(hipStreamWaitEvent(s.st[2], s.ev_as[r][1], 0));
    AS_TRY(hipMemcpyAsync(out, s.d_as[r][3], bytes, hipMemcpyDeviceToHost, s.st[2]));
    AS_TRY(hipEventRecord(s.ev_as[r][2], s.st[2]));
#undef AS_TRY
    return EOC_OK;
}

} // namespace

// ---- public API -----------------------------------------------------------------------------------------------------
extern "C" void eoc_shard_range(size_t total, int rank, int world, size_t *lo, size_t *hi)
{
    if (world < 1 || rank < 0 || rank >= world) {
        *lo = *hi = 0;
        return;
    }
    shard_range(total, rank, world, lo, hi);
}

extern "C" int eoc_gpu_init_multi(const int *devices, int n_devices, const eoc_params *p)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (!G.slots.empty()) {
        eoc_set_error("eoc_gpu_init: engine already initialised");
        return EOC_ERR_STATE;
    }
    if (!devices || n_devices < 1 || n_devices > 64 || !p) {
        eoc_set_error("eoc_gpu_init_multi: bad arguments");
        return EOC_ERR_ARG;
    }
    G.p = *p;
    G.slots.resize(n_devices);
    for (int i = 0; i < n_devices; i++) {
        Slot &s = G.slots[i];
        s.device = devices[i];
        int rc = eoc_engine_create(devices[i], p, &s.e);
        if (rc == EOC_OK) {
            if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.st[0], hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&s.st[1], hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&s.st[2], hipStreamNonBlocking) != hipSuccess) {
                eoc_set_error("eoc_gpu_init_multi: stream creation failed on device %d", s.device);
                rc = EOC_ERR_HIP;
            }
        }
        if (rc) {
            destroy_slots_locked();
            return rc;
        }
    }
    for (int i = 1; i < n_devices; i++) { // slot 0's block always runs on the calling thread
        Slot &s = G.slots[i];
        s.worker.reset(new Worker());
        Worker *w = s.worker.get();
        const int dev = s.device;
        w->th = std::thread([w, dev] { w->run(dev); });
    }
    return EOC_OK;
}
extern "C" int eoc_gpu_init(int device, const eoc_params *p) { return eoc_gpu_init_multi(&device, 1, p); }

// The device list the string API brings up on its first key: eoc_gpu_set_devices (what a Lua / Node host calls) wins,
// then EOC_TFHE_DEVICES = "all" | "0,1,2,..." (no change to the host's calls at all), then device 0.
static std::mutex g_pref_mu;
static std::vector<int> g_pref_devices;
extern "C" int eoc_gpu_set_devices(const int *devices, int n_devices)
{
    if (n_devices < 0 || n_devices > 64 || (n_devices && !devices)) {
        eoc_set_error("eoc_gpu_set_devices: bad arguments");
        return EOC_ERR_ARG;
    }
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0) {
            eoc_set_error("eoc_gpu_set_devices: negative device index");
            return EOC_ERR_ARG;
        }
    std::lock_guard<std::mutex> g(g_pref_mu);
    g_pref_devices.assign(devices, devices + n_devices);
    return EOC_OK;
}
extern "C" int eoc_gpu_init_from_env(const eoc_params *p)
{
    const char *s = getenv("EOC_TFHE_DEVICES");
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> g(g_pref_mu);
        devs = g_pref_devices;
    }
    if (!devs.empty()) return eoc_gpu_init_multi(devs.data(), (int)devs.size(), p);
    if (!s || !*s) devs.push_back(0);
    else if (!strcmp(s, "all")) {
        int c = eoc_device_count();
        for (int i = 0; i < c; i++) devs.push_back(i);
        if (devs.empty()) devs.push_back(0);
    } else {
        const char *q = s;
        while (*q) {
            char *end = nullptr;
            long v = strtol(q, &end, 10);
            if (end == q) {
                eoc_set_error("EOC_TFHE_DEVICES: cannot parse '%s'", s);
                return EOC_ERR_ARG;
            }
            devs.push_back((int)v);
            q = *end == ',' ? end + 1 : end;
        }
    }
    return eoc_gpu_init_multi(devs.data(), (int)devs.size(), p);
}

extern "C" int eoc_gpu_engine_count(void)
{
    std::lock_guard<std::mutex> g(G.mu);
    return (int)G.slots.size();
}
extern "C" eoc_engine *eoc_global_engine_at(int i)
{
    std::lock_guard<std::mutex> g(G.mu);
    return (i >= 0 && (size_t)i < G.slots.size()) ? G.slots[i].e : nullptr;
}
extern "C" eoc_engine *eoc_global_engine(void) { return eoc_global_engine_at(0); }

extern "C" void eoc_gpu_shutdown(void)
{
    std::lock_guard<std::mutex> g(G.mu);
    destroy_slots_locked();
}

extern "C" int eoc_stats(uint64_t out[3])
{
    std::lock_guard<std::mutex> g(G.mu);
    if (G.slots.empty()) {
        eoc_set_error("eoc_stats: no global engine (eoc_gpu_init)");
        return EOC_ERR_STATE;
    }
    out[0] = out[1] = out[2] = 0;
    for (auto &s : G.slots) {
        uint64_t t[3];
        int rc = eoc_engine_stats(s.e, t);
        if (rc) return rc;
        for (int k = 0; k < 3; k++) out[k] += t[k];
    }
    return EOC_OK;
}
// per-device counters [engines][3], seconds the key replication took, and how it was done
extern "C" int eoc_stats_multi(uint64_t *per_device, int cap_devices, double *key_broadcast_seconds)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (G.slots.empty()) {
        eoc_set_error("eoc_stats_multi: no global engine (eoc_gpu_init)");
        return EOC_ERR_STATE;
    }
    const int n = (int)G.slots.size();
    for (int i = 0; i < n && i < cap_devices; i++) {
        int rc = eoc_engine_stats(G.slots[i].e, per_device + 3 * (size_t)i);
        if (rc) return rc;
    }
    if (key_broadcast_seconds) *key_broadcast_seconds = G.bcast_s;
    return n;
}
extern "C" const char *eoc_key_broadcast_method(void)
{
    std::lock_guard<std::mutex> g(G.mu);
    static thread_local std::string copy;
    copy = G.bcast_method;
    return copy.c_str();
}
// how often the persistent worker of engine i (i >= 1) was woken since eoc_gpu_init_multi; 0 for engine 0 (its
// blocks run on the calling thread).  Lets a test assert that calls whose other blocks are empty wake nobody.
extern "C" uint64_t eoc_worker_wakeups(int i)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (i < 0 || (size_t)i >= G.slots.size() || !G.slots[i].worker) return 0;
    std::lock_guard<std::mutex> lk(G.slots[i].worker->m);
    return G.slots[i].worker->wakeups;
}
extern "C" const char *eoc_rccl_origin(void) { return R.origin; }

// One-GPU rehearsal of the library's RCCL call path (every one-GPU test replicates keys by device-to-device copies, so
// the broadcast branch itself only runs on a multi-GPU node): loads librccl exactly as the key broadcast does, builds a
// ONE-rank communicator on `device` with ncclCommInitAll and runs a grouped out-of-place ncclBroadcast of `bytes` bytes
// through the dlopen'ed table, then compares source and destination.  Returns EOC_OK or an error with a message.
extern "C" int eoc_rccl_selftest(int device, size_t bytes)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (!bytes) bytes = 1 << 20;
    if (!R.load()) {
        const char *why = dlerror();
        eoc_set_error("eoc_rccl_selftest: librccl could not be loaded: %s", why ? why : "missing symbol");
        return EOC_ERR_STATE;
    }
    HIP_TRY(hipSetDevice(device));
    unsigned char *src = nullptr, *dst = nullptr;
    HIP_TRY(hipMalloc(&src, bytes));
    if (hipMalloc(&dst, bytes) != hipSuccess) {
        hipFree(src);
        eoc_set_error("eoc_rccl_selftest: hipMalloc failed");
        return EOC_ERR_ALLOC;
    }
    std::vector<unsigned char> h(bytes), back(bytes, 0);
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    int rc = EOC_OK;
    hipStream_t st = nullptr;
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclSuccess;
    // (the NULL-stream fill is drained before the non-blocking stream below may touch dst: such streams do not wait for it)
    if (hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dst, 0, bytes) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        eoc_set_error("eoc_rccl_selftest: HIP set-up failed");
        rc = EOC_ERR_HIP;
    }
    if (!rc && (r = R.CommInitAll(&comm, 1, &device)) != ncclSuccess) {
        eoc_set_error("eoc_rccl_selftest: ncclCommInitAll failed: %s", R.GetErrorString(r));
        rc = EOC_ERR_HIP;
    }
    if (!rc) {
        R.GroupStart();
        r = R.Broadcast(src, dst, bytes, ncclChar, 0, comm, st);
        ncclResult_t r2 = R.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess || hipStreamSynchronize(st) != hipSuccess) {
            eoc_set_error("eoc_rccl_selftest: ncclBroadcast failed: %s", R.GetErrorString(r));
            rc = EOC_ERR_HIP;
        }
    }
    if (!rc && (hipMemcpy(back.data(), dst, bytes, hipMemcpyDeviceToHost) != hipSuccess || back != h)) {
        eoc_set_error("eoc_rccl_selftest: broadcast result differs from its source");
        rc = EOC_ERR_STATE;
    }
    if (comm) R.CommDestroy(comm);
    if (st) hipStreamDestroy(st);
    hipFree(src);
    hipFree(dst);
    return rc;
}
extern "C" uint64_t eoc_host_path_buffer_grows(void)
{
    std::lock_guard<std::mutex> g(G.mu);
    uint64_t t = 0;
    for (auto &s : G.slots) t += s.grows;
    return t;
}

extern "C" int eoc_upload_cloud_key_arrays(const int32_t *bk, const int32_t *ksk)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (G.slots.empty()) {
        eoc_set_error("eoc_upload_cloud_key: call eoc_gpu_init first");
        return EOC_ERR_STATE;
    }
    if (!bk || !ksk) return EOC_ERR_NO_KEY;
    {   // pending submissions still read the key images that are about to be rebuilt
        int rc = drain_async_locked();
        if (rc) return rc;
    }
    // images are built once, on the first device (H2D of the torus form, forward transforms on the GPU, KSK padding)
    int rc = eoc_engine_load_cloud_key(G.slots[0].e, bk, ksk);
    if (rc) return rc;
    const int n = (int)G.slots.size();
    G.bcast_s = 0.0;
    G.bcast_method = "none";
    if (n > 1) {
        const void *bk0 = nullptr, *ksk0 = nullptr;
        rc = eoc_engine_cloud_key_device(G.slots[0].e, &bk0, &ksk0);
        if (rc) return rc;
        std::vector<void *> dbk(n, nullptr), dksk(n, nullptr);
        dbk[0] = const_cast<void *>(bk0);
        dksk[0] = const_cast<void *>(ksk0);
        // replicas not yet adopted by their engine are released on every error path
        auto release_from = [&](int first) {
            for (int i = std::max(first, 1); i < n; i++) {
                if (dbk[i]) eoc_device_free(G.slots[i].e, dbk[i]);
                if (dksk[i]) eoc_device_free(G.slots[i].e, dksk[i]);
                dbk[i] = dksk[i] = nullptr;
            }
        };
        for (int i = 1; i < n; i++) {
            // the engine owns replicas allocated through its own allocator entry points
            rc = eoc_device_alloc(G.slots[i].e, eoc_bkfft_bytes(&G.p), &dbk[i]);
            if (rc == EOC_OK) rc = eoc_device_alloc(G.slots[i].e, eoc_ksk_dev_bytes(&G.p), &dksk[i]);
            if (rc) {
                release_from(1);
                return rc;
            }
        }
        auto t0 = std::chrono::steady_clock::now();
        rc = replicate_key_locked(bk0, ksk0, dbk, dksk);
        G.bcast_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) {
            release_from(1);
            return rc;
        }
        for (int i = 1; i < n; i++) {
            rc = eoc_engine_adopt_cloud_key_device(G.slots[i].e, dbk[i], dksk[i]);
            if (rc) {
                release_from(i);
                return rc;
            }
        }
    }
    G.key_loaded = true;
    return EOC_OK;
}
extern "C" int eoc_upload_cloud_key(const eoc_secret_key *sk)
{
    if (!sk || !eoc_sk_bk(sk) || !eoc_sk_ksk(sk)) return EOC_ERR_NO_KEY;
    {   // the arrays are read with the ENGINE's shape: a key set of another shape is refused, not read out of bounds
        std::lock_guard<std::mutex> g(G.mu);
        const eoc_params *kp = eoc_sk_params(sk);
        if (!G.slots.empty() && kp &&
            (kp->n != G.p.n || kp->l != G.p.l || kp->Bgbit != G.p.Bgbit || kp->ks_t != G.p.ks_t ||
             kp->ks_basebit != G.p.ks_basebit)) {
            eoc_set_error("eoc_upload_cloud_key: key shape (n=%d l=%d Bgbit=%d ks %d x %d bits) differs from the engines' "
                          "(n=%d l=%d Bgbit=%d ks %d x %d bits)", kp->n, kp->l, kp->Bgbit, kp->ks_t, kp->ks_basebit,
                          G.p.n, G.p.l, G.p.Bgbit, G.p.ks_t, G.p.ks_basebit);
            return EOC_ERR_ARG;
        }
    }
    return eoc_upload_cloud_key_arrays(eoc_sk_bk(sk), eoc_sk_ksk(sk));
}

extern "C" void *eoc_host_alloc(size_t bytes)
{
    void *p = nullptr, *dp = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        eoc_set_error("eoc_host_alloc: hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        dp = p; // unified addressing: the host address is valid on the device
    }
    std::lock_guard<std::mutex> g(g_pin_mu);
    g_pinned[reinterpret_cast<uintptr_t>(p)] = PinnedBlock{bytes ? bytes : 1, static_cast<char *>(dp)};
    return p;
}
extern "C" void eoc_host_free(void *p)
{
    if (!p) return;
    {   // a buffer of a submission still in flight must not be released under its DMA (a Node Buffer finalizer, a Python
        // PinnedArray going out of scope): complete what is pending first
        std::lock_guard<std::mutex> g(G.mu);
        if (G.ring[0].active || G.ring[1].active) (void)drain_async_locked();
    }
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        g_pinned.erase(reinterpret_cast<uintptr_t>(p));
    }
    hipHostFree(p);
}

extern "C" int eoc_gate_batch(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                              int32_t *out, size_t count)
{
    std::lock_guard<std::mutex> g(G.mu); // calls on the global context are serialised, as in the reference (one global key)
    if (G.slots.empty()) {
        eoc_set_error("eoc_gate_batch: no GPU engine (eoc_gpu_init not called or failed); there is no CPU fallback");
        return EOC_ERR_NO_DEVICE;
    }
    const bool const_only = !ops && (op == EOC_CONST0 || op == EOC_CONST1);
    if (!out || (!in0 && !const_only)) {
        eoc_set_error("eoc_gate_batch: null argument");
        return EOC_ERR_ARG;
    }
    if (!count) return EOC_OK;
    {
        int rc = drain_async_locked();
        if (rc) return rc;
    }
    const size_t stride = (size_t)G.p.n + 1;
    return for_each_block(count, [=](int i, size_t lo, size_t hi) {
        return slot_gate_block(G.slots[i], op, ops ? ops + lo : nullptr, in0 ? in0 + lo * stride : nullptr,
                               in1 ? in1 + lo * stride : nullptr, in2 ? in2 + lo * stride : nullptr, out + lo * stride,
                               hi - lo, stride);
    });
}

extern "C" int eoc_circuit_run(const eoc_gate *gates, size_t n_gates, int32_t *wires, size_t n_wires, size_t instances)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (G.slots.empty()) {
        eoc_set_error("eoc_circuit_run: no GPU engine; there is no CPU fallback");
        return EOC_ERR_NO_DEVICE;
    }
    if (!gates || !wires) return EOC_ERR_ARG;
    if (!n_gates || !instances) return EOC_OK;
    {
        int rc = drain_async_locked();
        if (rc) return rc;
    }
    const size_t stride = (size_t)G.p.n + 1;
    // a whole circuit instance stays on one device: no wire ever crosses GPUs
    return for_each_block(instances, [=](int i, size_t lo, size_t hi) {
        return slot_circuit_block(G.slots[i], gates, n_gates, wires, n_wires, instances, lo, hi, stride);
    });
}

// Asynchronous form of eoc_gate_batch (the reference has no counterpart: it is one synchronous wasm call per operation,
// eoc-tfhe-bindings.c:12-24).  The call returns once the batch is QUEUED: operands travel on the H2D stream, kernels
// follow on the kernel stream, results leave on the D2H stream, so a host that keeps two batches in flight hides the PCIe
// time of one behind the kernels of the other.  All buffers must come from eoc_host_alloc and stay untouched until
// eoc_gate_batch_wait(ticket).  At most two submissions are in flight: a third first waits for the oldest.  Batches
// execute in submission order.  Every synchronous call on the global context drains the pending submissions first.
extern "C" int eoc_gate_batch_submit(int op, const uint8_t *ops, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                     int32_t *out, size_t count, uint64_t *ticket)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (G.slots.empty()) {
        eoc_set_error("eoc_gate_batch_submit: no GPU engine (eoc_gpu_init not called or failed); there is no CPU fallback");
        return EOC_ERR_NO_DEVICE;
    }
    const bool const_only = !ops && (op == EOC_CONST0 || op == EOC_CONST1);
    if (!ticket || !out || (!in0 && !const_only)) {
        eoc_set_error("eoc_gate_batch_submit: null argument");
        return EOC_ERR_ARG;
    }
    if (!is_pinned(in0) || !is_pinned(in1) || !is_pinned(in2) || !mapped_address(out)) {
        eoc_set_error("eoc_gate_batch_submit: every buffer must come from eoc_host_alloc (asynchronous copies from pageable "
                      "memory are staged synchronously by the runtime)");
        return EOC_ERR_ARG;
    }
    const uint64_t t = G.next_ticket;
    Pending &pd = G.ring[t & 1];
    int rc = finish_pending_locked(pd); // the submission that used this buffer set two tickets ago
    if (rc) return rc;
    pd.ticket = t;
    pd.slots.clear();
    const size_t stride = (size_t)G.p.n + 1;
    const int world = (int)G.slots.size();
    for (int i = 0; i < world && count; i++) { // queuing is asynchronous: done from the calling thread, device by device
        size_t lo, hi;
        shard_range(count, i, world, &lo, &hi);
        if (hi == lo) continue;
        rc = slot_async_block(G.slots[i], (int)(t & 1), op, ops ? ops + lo : nullptr, in0 ? in0 + lo * stride : nullptr,
                              in1 ? in1 + lo * stride : nullptr, in2 ? in2 + lo * stride : nullptr, out + lo * stride,
                              hi - lo, stride);
        if (rc) { // what was queued on earlier engines completes into the caller's buffers; the ticket is not issued
            pd.active = true;
            (void)finish_pending_locked(pd);
            return rc;
        }
        pd.slots.push_back(i);
    }
    pd.active = true;
    G.next_ticket = t + 1;
    *ticket = t;
    return EOC_OK;
}
// returns when the submission's results are in the caller's output buffer (immediately for a ticket already awaited)
extern "C" int eoc_gate_batch_wait(uint64_t ticket)
{
    std::lock_guard<std::mutex> g(G.mu);
    if (ticket == 0 || ticket >= G.next_ticket) {
        eoc_set_error("eoc_gate_batch_wait: unknown ticket %llu", (unsigned long long)ticket);
        return EOC_ERR_ARG;
    }
    Pending &pd = G.ring[ticket & 1];
    if (pd.ticket != ticket) return EOC_OK; // an older ticket: its buffer set was re-used, so it has completed
    return finish_pending_locked(pd);
}
