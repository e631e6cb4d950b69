// Micro-benchmark (diagnostic, round 4): would an 8 x 8 register/lane transpose of complex doubles be cheaper on the CU's LDS
// pipe if its stores were ds_write_addtid_b32 (2 cycles per 256-byte row, no address register) and its loads ds_read2_b32
// (one double per instruction out of two dword planes)?  8 waves per CU, as in k_blind_rotate.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds3.hip -o tools/_ubench_lds3
// MODE 0: 8 x ds_write_b128 + 8 x ds_read_b128 (what the kernel does)    1: 32 x ds_write_addtid_b32 only
//      2: 16 x ds_read2_b32 only   3: 32 x addtid + 16 x read2_b32 (the candidate transpose)   4: 8 x ds_write_b128 only
//      5: 8 x ds_read_b128 only
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int R = 66; // dword stride of a plane row: (4 hi + d) * 66 mod 64 = 8 hi + 2 d, the eight lane groups hit eight bank groups
template <int MODE> __global__ __launch_bounds__(128, 2) void k(double *out, int iters)
{
    __shared__ d2 scr_all[2][568];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    d2 *scr = scr_all[w];
    double *scd = reinterpret_cast<double *>(scr);
    d2 x[8];
    for (int r = 0; r < 8; r++) x[r] = d2{lane * 0.5 + r, r * 0.25};
    for (int i = lane; i < 568; i += 64) scr[i] = d2{1.0 * i, 2.0};
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)scr);
    const int rd = (int)base + (((lane >> 3) * 4) * R + (lane & 7)) * 4; // reader: planes of source register hi, column lo (+ 8 r)
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 8; r++) scr[72 * r + lane] = x[r];
        }
        if (MODE == 1 || MODE == 3) {
            unsigned m0s;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1" : "=&s"(m0s) : "s"(base) : "memory");
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(__double2loint(x[r].x)), "n"((4 * r + 0) * R * 4) : "memory");
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(__double2hiint(x[r].x)), "n"((4 * r + 1) * R * 4) : "memory");
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(__double2loint(x[r].y)), "n"((4 * r + 2) * R * 4) : "memory");
                asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(__double2hiint(x[r].y)), "n"((4 * r + 3) * R * 4) : "memory");
            }
            asm volatile("s_mov_b32 m0, %0" ::"s"(m0s) : "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (MODE == 0 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = scr[72 * (lane >> 3) + 8 * r + (lane & 7)];
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double re, im;
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(re) : "v"(rd), "n"(8 * r), "n"(8 * r + R) : "memory");
                asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(im) : "v"(rd), "n"(8 * r + 2 * R), "n"(8 * r + 3 * R) : "memory");
                x[r] = d2{re, im};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE == 1 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 8; r++) x[r].x += 1.0; // keep the stores live and distinct
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    double s = 0;
    for (int r = 0; r < 8; r++) s += x[r].x + x[r].y;
    out[blockIdx.x * 128 + threadIdx.x] = s + scd[lane];
}
template <int MODE> void run(const char *name)
{
    const int blocks = 1024; // four workgroups of two waves per CU: eight waves per CU
    double *out;
    (void)hipMalloc(&out, (size_t)blocks * 128 * 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-44s %.4f us per round (8 waves per CU)\n", name, ms * 1e3 / iters);
    (void)hipFree(out);
}
int main()
{
    run<0>("8 w128 + 8 r128 (kernel's transpose)");
    run<4>("8 w128 only");
    run<5>("8 r128 only");
    run<1>("32 ds_write_addtid_b32 only");
    run<2>("16 ds_read2_b32 only");
    run<3>("32 addtid + 16 read2_b32 (candidate)");
    return 0;
}
