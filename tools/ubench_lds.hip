// Micro-benchmark (diagnostic): how LDS transposes and FP64 work overlap at 2 waves per SIMD (8 waves per CU).
// Each wave runs `iters` rounds of: 8 ds_write_b128 + 8 ds_read_b128 (a transpose of 8 complex doubles per lane through a
// private 9 KB scratch) and/or NF dependent-chain-free v_fma_f64.  Modes: 0 = LDS only, 1 = FMA only, 2 = both in sequence,
// 3 = both, compiler free to interleave.  Prints microseconds per round.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
#define NF 72
template <int MODE> __global__ __launch_bounds__(128, 4) void k(double *out, int iters)
{
    __shared__ d2 scr_all[2][568];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    d2 *scr = scr_all[w];
    d2 x[8], y[8];
    for (int r = 0; r < 8; r++) { x[r] = d2{lane * 0.5 + r, r * 0.25}; y[r] = d2{1.0 + r, 2.0 - lane}; }
    const double c = 1.0000001, d = 1e-9;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) scr[72 * r + lane] = x[r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = scr[72 * (lane >> 3) + 8 * r + (lane & 7)];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int k2 = 0; k2 < NF / 16; k2++)
#pragma unroll
                for (int r = 0; r < 8; r++) { y[r].x = __builtin_fma(y[r].x, c, d); y[r].y = __builtin_fma(y[r].y, c, d); }
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    }
    double s = 0;
    for (int r = 0; r < 8; r++) s += x[r].x + x[r].y + y[r].x + y[r].y;
    out[blockIdx.x * 128 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int blocks)
{
    double *out;
    hipMalloc(&out, (size_t)blocks * 128 * 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s blocks=%4d (%.1f waves/SIMD): %.4f us per round\n", name, blocks, blocks / 512.0, ms * 1e3 / iters);
    hipFree(out);
}
int main()
{
    for (int blocks : {1024, 1536, 2048}) {
        run<0>("transpose only (8 w128 + 8 r128)", blocks);
        run<1>("72 fma_f64 only", blocks);
        run<2>("transpose then fma (pinned)", blocks);
        run<3>("transpose + fma (compiler order)", blocks);
    }
    return 0;
}
