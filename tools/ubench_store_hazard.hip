// Does gfx950 need a wait state between `buffer_store_dwordx4 v[a:a+3], v, s[..], sN offen` and a VALU write of v[a..a+3]?
// It does.  hipcc 7.2 inserts none when the store's soffset is an SGPR (its hazard recogniser exempts that form; with an
// immediate soffset, and for global_store_dwordx4, it inserts wait states) -- seen in the ISA of the gadget-length-3
// prototype of the wide blind rotation, whose parked accumulators came back corrupted in 0.3 % of the steps
// (profiles/r05_wide_gadget3_attempt.txt section 2b).  This probe pins the HARDWARE half: the instruction
// sequence is fixed by one asm block -- fill v[40:43], store them, W wait states, overwrite v[40:43] -- and the row is read
// back and compared with what should have been stored.  The shipped library contains no buffer store.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_store_hazard.hip -o tools/_ubench_store_hazard && tools/_ubench_store_hazard
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef unsigned int u4 __attribute__((__vector_size__(16)));

#define STORE_THEN_CLOBBER(NOPS, SOFF)                                                                        \
    asm volatile("v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %3\n\t"       \
                 "s_nop 4\n\t"                                                                                 \
                 "buffer_store_dwordx4 v[40:43], %4, %5, " SOFF " offen\n\t" NOPS                              \
                 "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"            \
                 : : "v"(a), "v"(b), "v"(c), "v"(d), "v"(voff), "s"(rs), "s"(soff) : "v40", "v41", "v42", "v43", "memory")

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *buf, unsigned long long *bad, int iters, uint32_t bytes)
{
    const unsigned long long base = (unsigned long long)(uintptr_t)buf;
    const u4 rs = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, bytes, 0x00020000u};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t soff = __builtin_amdgcn_readfirstlane(wave * 1024u);          // wave-uniform: an SGPR
    uint32_t voff = lane * 16;
    if (MODE >= 4) voff += soff;                                                 // immediate-soffset forms: all in the VGPR
    uint32_t a = wave * 7919u + lane, b = a ^ 0x9e3779b9u, c = a + 0x7f4a7c15u, d = ~a;
    unsigned long long nbad = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) STORE_THEN_CLOBBER("", "%6");
        if (MODE == 1) STORE_THEN_CLOBBER("s_nop 0\n\t", "%6");
        if (MODE == 2) STORE_THEN_CLOBBER("s_nop 1\n\t", "%6");
        if (MODE == 3) STORE_THEN_CLOBBER("s_nop 2\n\t", "%6");
        if (MODE == 4) STORE_THEN_CLOBBER("", "0");
        if (MODE == 5) STORE_THEN_CLOBBER("s_nop 1\n\t", "0");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const u4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff, 16);
        nbad += (r[0] != a) + (r[1] != b) + (r[2] != c) + (r[3] != d);
        a += 0x01010101u; b += 0x00010001u; c += 3u; d += 0x10u;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int MODE>
static void run(const char *name, uint32_t *d_buf, unsigned long long *d_bad, int wgs, int iters)
{
    (void)hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(256), 0, 0, d_buf, d_bad, iters, (uint32_t)(wgs * 4 * 1024));
    unsigned long long bad = 0;
    (void)hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    printf("%-66s %12llu wrong dwords of %llu\n", name, bad, (unsigned long long)wgs * 256 * 4 * iters);
}
int main()
{
    const int wgs = 2048, iters = 5000;
    uint32_t *d_buf;
    unsigned long long *d_bad;
    (void)hipMalloc(&d_buf, (size_t)wgs * 4 * 1024);
    (void)hipMalloc(&d_bad, 8);
    run<0>("SGPR soffset, VALU overwrite directly behind the store:", d_buf, d_bad, wgs, iters);
    run<1>("SGPR soffset, s_nop 0 (1 wait state) in between:", d_buf, d_bad, wgs, iters);
    run<2>("SGPR soffset, s_nop 1 (2 wait states):", d_buf, d_bad, wgs, iters);
    run<3>("SGPR soffset, s_nop 2 (3 wait states):", d_buf, d_bad, wgs, iters);
    run<4>("soffset 0, VALU overwrite directly behind the store:", d_buf, d_bad, wgs, iters);
    run<5>("soffset 0, s_nop 1 (what the compiler inserts for this form):", d_buf, d_bad, wgs, iters);
    return 0;
}
