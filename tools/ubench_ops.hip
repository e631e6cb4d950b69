// Micro-benchmark (diagnostic, not part of the library): issue cost in cycles of the candidate instructions for the
// blind-rotate kernel, at 1 and 2 waves per SIMD:  v_fma_f64, v_add_f64, v_trunc_f64, v_rndne_f64, v_cvt_f64_i32,
// v_permlane16_swap, v_permlane32_swap, v_mov_b32 with DPP, plus a semantic check of the two swaps.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_ops tools/ubench_ops.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITER 200

template <int OP> __global__ __launch_bounds__(256) void k(double *out, unsigned long long *cyc, int n)
{
    double a[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001 + i; u[i] = threadIdx.x * 7 + i; }
    const double c = 1.0000001, d = 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = __builtin_fma(a[i], c, d);
                if (OP == 1) a[i] = a[i] + c;
                if (OP == 2) asm volatile("v_trunc_f64 %0, %0" : "+v"(a[i]));
                if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
                if (OP == 4) { asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(u[i])); }
                if (OP == 5) { auto s = __builtin_amdgcn_permlane16_swap(u[i], u[(i + 1) & 7], false, false); u[i] = s[0]; u[(i + 1) & 7] = s[1]; }
                if (OP == 6) { auto s = __builtin_amdgcn_permlane32_swap(u[i], u[(i + 1) & 7], false, false); u[i] = s[0]; u[(i + 1) & 7] = s[1]; }
                if (OP == 7) u[i] = __builtin_amdgcn_update_dpp(u[i], u[(i + 3) & 7], 0x128 /*row_ror:8*/, 0xf, 0xc, false);
                if (OP == 8) u[i] = u[i] * 3u + 1u; // v_mad_u32_u24-ish / v_mul_lo: integer baseline
                if (OP == 9) u[i] = (u[i] + 64u) & 2047u;
                if (OP == 10) a[i] = a[i] * c;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    unsigned us = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; us += u[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = s + us;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void k_sem(unsigned *o)
{
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + threadIdx.x] = s[0];
    o[192 + threadIdx.x] = s[1];
}

template <int OP> void run(const char *name, int blocks_per_cu)
{
    int cus = 256;
    int blocks = cus * blocks_per_cu;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 8);
    hipMalloc(&cyc, blocks * 4 * 8);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, ITER);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= h.size();
    double per = m / ((double)ITER * REP); // s_memtime ticks (100 MHz? or shader clock) per instruction per wave
    printf("%-22s waves/SIMD=%d  ticks/instr/wave=%.3f  -> per SIMD issue=%.3f   wall %.3f ms\n", name, blocks_per_cu, per,
           per / blocks_per_cu, ms);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    unsigned *o;
    hipMalloc(&o, 256 * 4);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, o);
    unsigned h[256];
    hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    printf("permlane32_swap(a=lane, b=lane+100): r0 =");
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]);
    printf("\n                                     r1 =");
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]);
    printf("\npermlane16_swap: s0 =");
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[128 + i]);
    printf("\n                 s1 =");
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[192 + i]);
    printf("\n");
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f64", w);
        run<1>("v_add_f64", w);
        run<10>("v_mul_f64", w);
        run<2>("v_trunc_f64", w);
        run<3>("v_rndne_f64", w);
        run<4>("v_cvt_f64_i32", w);
        run<5>("v_permlane16_swap", w);
        run<6>("v_permlane32_swap", w);
        run<7>("v_mov_dpp row_ror:8", w);
        run<8>("int mul-add", w);
        run<9>("int add+and", w);
    }
    return 0;
}
