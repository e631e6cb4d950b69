// What address does ds_write_addtid_b32 use?  ISA text: LDS_ADDR = M0[15:0] + offset + TID * 4.  This probe answers the two
// questions the key-switch table builder depends on: is TID the lane within the wave or the thread within the workgroup,
// and does the sum wrap at 16 bits.   hipcc --offload-arch=gfx950 tools/ubench_addtid.hip -o tools/_ubench_addtid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, unsigned base, int probe_wrap)
{
    extern __shared__ int smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 20480; i += blockDim.x) smem[i] = -1;
    __syncthreads();
    unsigned m0_saved;
    int v = 1000 + tid;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1\n\tds_write_addtid_b32 %2 offset:64\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(__builtin_amdgcn_readfirstlane(base + (tid >> 6) * 1024)), "v"(v) : "memory");
    if (probe_wrap) { // base near 64 KiB: does M0[15:0] + offset + TID*4 carry into bit 16?
        int v2 = 5000 + tid;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 1\n\tds_write_addtid_b32 %2 offset:4096\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(__builtin_amdgcn_readfirstlane(65024u - (tid >> 6) * 1024)), "v"(v2) : "memory");
    }
    __syncthreads();
    for (int i = tid; i < 20480; i += blockDim.x) out[i] = smem[i];
}
int main()
{
    int *d;
    hipMalloc(&d, 20480 * 4);
    std::vector<int> h(20480);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 20480 * 4, 0, d, 1024u, 1);
    hipMemcpy(h.data(), d, 20480 * 4, hipMemcpyDeviceToHost);
    // thread t wrote 1000 + t at byte (1024 + 64 + X * 4): find X for t = 0, 1, 63, 64, 65, 255
    for (int t : {0, 1, 63, 64, 65, 128, 255}) {
        int where = -1;
        for (int i = 0; i < 20480; i++) if (h[i] == 1000 + t) where = i;
        const int b = (1024 + (t / 64) * 1024 + 64) / 4;   // every wave has its own base: M0 = 1024 + wave * 1024
        printf("thread %3d (wave %d lane %2d): dword %d = its wave's base dword %d + %d\n", t, t / 64, t % 64, where, b, where - b);
    }
    for (int t : {0, 64, 255}) {
        int where = -1;
        for (int i = 0; i < 20480; i++) if (h[i] == 5000 + t) where = i;
        const unsigned b = 65024u - (t / 64) * 1024;         // M0 + offset crosses 64 KiB for wave 0 only
        printf("wrap probe thread %3d: dword %d (no wrap would be %u, 16-bit wrap %u)\n", t, where, (b + 4096) / 4 + t % 64,
               ((b + 4096 + (t % 64) * 4) & 0xFFFF) / 4);
    }
    return 0;
}
