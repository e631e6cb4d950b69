// Micro-benchmark (diagnostic): CU-wide cost of the LDS store/load forms a transpose could use, 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
// MODE 0: 8 x ds_write_b128 + 8 x ds_read_b128   1: 16 x ds_write_b64 + 8 x ds_read_b128   2: 8 x w128 + 16 x ds_read_b64
// 3: writes only (8 x b128)   4: reads only (8 x b128)   5: 16 x ds_write_b64 only   6: 8 x ds_write2_b64 only
// Planar forms (real parts in one plane, imaginary parts 4608 bytes above, lanes 8 bytes apart):
// 7: 16 x ds_write_b64   8: 8 x ds_write2st64_b64   9: 8 x ds_read2st64_b64   10: 16 x ds_read_b64
// 11: 32 x ds_write_b32 over four planes   12: 8 x write2st64_b64 + 8 x read2st64_b64 (a whole planar transpose)
// Accumulator image round trip against a crossbar rotation:
// 13: 16 x ds_write2st64_b32 + 16 x ds_read_b32 (rotated)   14: 16 x ds_bpermute_b32
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
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; r++) scr[72 * r + lane] = x[r];
        }
        if (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("ds_write_b64 %0, %1" ::"v"((72 * r + lane) * 16 + (int)(w * 568 * 16)), "v"(x[r].x) : "memory");
                asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"((72 * r + lane) * 16 + (int)(w * 568 * 16)), "v"(x[r].y) : "memory");
            }
        }
        if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" ::"v"((72 * r + lane) * 16 + (int)(w * 568 * 16)), "v"(x[r].x), "v"(x[r].y) : "memory");
        }
        if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("ds_write_b64 %0, %1" ::"v"((72 * r + lane) * 8 + (int)(w * 568 * 16)), "v"(x[r].x) : "memory");
                asm volatile("ds_write_b64 %0, %1 offset:4608" ::"v"((72 * r + lane) * 8 + (int)(w * 568 * 16)), "v"(x[r].y) : "memory");
            }
        }
        if (MODE == 8 || MODE == 12) {
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:9" ::"v"((72 * r + lane) * 8 + (int)(w * 568 * 16)), "v"(x[r].x), "v"(x[r].y) : "memory");
        }
        if (MODE == 11) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int a = (72 * r + lane) * 4 + (int)(w * 568 * 16);
                asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(__double2loint(x[r].x)) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:2304" ::"v"(a), "v"(__double2hiint(x[r].x)) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:4608" ::"v"(a), "v"(__double2loint(x[r].y)) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:6912" ::"v"(a), "v"(__double2hiint(x[r].y)) : "memory");
            }
        }
        if (MODE == 13 || MODE == 14) {
            int v[16], u[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = __double2loint(x[r & 7].x) + r;
            if (MODE == 13) {
                int *img = reinterpret_cast<int *>(scr);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    img[lane + 64 * r] = v[r];
                    img[lane + 64 * r + 1024] = -v[r];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int k = (lane - (it * 37 + 5)) & 2047;
#pragma unroll
                for (int r = 0; r < 16; r++) u[r] = img[(k + 64 * r) & 2047];
            } else {
                const int src = ((lane - (it * 37 + 5)) & 63) * 4;
#pragma unroll
                for (int r = 0; r < 16; r++) u[r] = __builtin_amdgcn_ds_bpermute(src, v[r]);
            }
            int acc = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) acc += u[r];
            x[0].x += (double)(acc & 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (MODE == 9 || MODE == 12) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                d2 v;
                asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:9" : "=v"(v) : "v"((72 * (lane >> 3) + 8 * r + (lane & 7)) * 8 + (int)(w * 568 * 16)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                x[r] = v;
            }
        }
        if (MODE == 10) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double a, b;
                asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"((72 * (lane >> 3) + 8 * r + (lane & 7)) * 8 + (int)(w * 568 * 16)) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:4608" : "=v"(b) : "v"((72 * (lane >> 3) + 8 * r + (lane & 7)) * 8 + (int)(w * 568 * 16)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                x[r] = d2{a, b};
            }
        }
        if (MODE == 0 || MODE == 1 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 8; r++) x[r] = scr[72 * (lane >> 3) + 8 * r + (lane & 7)];
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                double a, b;
                asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"((72 * (lane >> 3) + 8 * r + (lane & 7)) * 16 + (int)(w * 568 * 16)) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(b) : "v"((72 * (lane >> 3) + 8 * r + (lane & 7)) * 16 + (int)(w * 568 * 16)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                x[r] = d2{a, b};
            }
        }
        if (MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 11) {
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
    const int blocks = 1024;
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
    printf("%-40s %.4f us per round (8 waves per CU)\n", name, ms * 1e3 / iters);
    hipFree(out);
}
int main()
{
    run<0>("8 w128 + 8 r128");
    run<1>("16 w64 + 8 r128");
    run<2>("8 w128 + 16 r64");
    run<3>("8 w128 only");
    run<4>("8 r128 only");
    run<5>("16 w64 only");
    run<6>("8 write2_b64 only");
    run<7>("planar 16 w64 only");
    run<8>("planar 8 write2st64_b64 only");
    run<9>("planar 8 read2st64_b64 only");
    run<10>("planar 16 r64 only");
    run<11>("planar 32 w32 only");
    run<12>("planar 8 write2st64 + 8 read2st64");
    run<13>("image: 16 write2st64_b32 + 16 r32");
    run<14>("crossbar: 16 bpermute_b32");
    return 0;
}
