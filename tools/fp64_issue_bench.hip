// fp64_issue_bench.hip -- how many waves per SIMD does gfx950 need to issue v_fma_f64 at full rate?
// Each wave runs 16 independent FMA chains (no memory traffic); the grid puts W waves on every SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_issue_bench.hip -o gpurun_out/fp64_issue && gpurun_out/fp64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void k_fma(double *out, int iters, double a, double b)
{
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 1e-9 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = __builtin_fma(x[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    if (s == 12345.678) out[0] = s; // keep the chains alive
}

template <int CHAINS>
static void run(int waves_per_simd, int iters)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    double *d;
    hipMalloc(&d, 8);
    dim3 block(256 * 1), grid(cus * waves_per_simd); // 256 threads = 4 waves = 1 per SIMD; W blocks per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fma<CHAINS>, grid, block, 0, 0, d, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma<CHAINS>, grid, block, 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fmas = (double)grid.x * 256 * (double)iters * CHAINS;
    printf("chains=%2d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s (fma = 2 flop)\n", CHAINS, waves_per_simd, ms,
           2 * fmas / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main()
{
    for (int w = 1; w <= 4; w++) run<16>(w, 20000);
    for (int w = 1; w <= 4; w++) run<4>(w, 80000);
    for (int w = 1; w <= 2; w++) run<1>(w, 200000);
    // ILP sweep at the occupancy the blind-rotate kernel runs at (2 waves per SIMD)
    run<2>(2, 160000);
    run<3>(2, 100000);
    run<6>(2, 50000);
    run<8>(2, 40000);
    run<12>(2, 26000);
    run<24>(2, 13000);
    run<32>(2, 10000);
    return 0;
}
