// One-shot hazard probe: the byte-merge sequence hipcc emitted for the 8-bit k_unpack of round 1
//   v_and_b32_sdwa t, a, m dst_sel:BYTE_1 dst_unused:UNUSED_PAD ; [N wait states] ; v_bitop3_b16 r, hi, t, 0xff bitop3:0xec
// with N = 0, 1 (what the compiler scheduled: one s_movk between the two), 2 and 4 wait states, checked against plain C on
// every lane of a full-chip grid.  Prints the number of wrong results per N.   hipcc --offload-arch=gfx950 -O2 -o sdwa_hazard sdwa_hazard.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define SEQ(NOPS)                                                                                                      \
    asm volatile("v_and_b32_sdwa %0, %2, %3 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"     \
                 NOPS "v_bitop3_b16 %1, %4, %0, %5 bitop3:0xec"                                                        \
                 : "=&v"(t), "=&v"(r) : "v"(a), "v"(m), "v"(hi), "s"(0xffu))
template <int N>
__global__ void probe(const uint32_t* __restrict__ in, uint32_t* __restrict__ bad, int iters) {
    uint32_t wrong = 0;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = in[tid & 4095] ^ tid;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = x, m = 0xffu >> (x >> 29), hi = (x >> 7) & 0xffffu;
        uint32_t t, r;
        if (N == 0) SEQ("");
        else if (N == 1) SEQ("s_movk_i32 s20, 0xff\n\t");
        else if (N == 2) SEQ("s_nop 1\n\t");
        else SEQ("s_nop 3\n\t");
        const uint32_t want = (((a & m) & 0xffu) << 8) | (hi & 0xffu);
        wrong += (r & 0xffffu) != want;
    }
    if (wrong) atomicAdd(&bad[N], wrong);
}
int main() {
    uint32_t *in, *bad;
    hipMalloc(&in, 4096 * 4); hipMalloc(&bad, 32);
    hipMemset(in, 0x5a, 4096 * 4); hipMemset(bad, 0, 32);
    const int grid = 256 * 32, iters = 4096;
    hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, 0, in, bad, iters);
    hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, 0, in, bad, iters);
    hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, in, bad, iters);
    hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, in, bad, iters);
    uint32_t h[8];
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost);
    printf("wrong results of %llu per variant: N=0: %u  N=1 (s_movk between, as compiled): %u  N=2: %u  N=4: %u\n",
           (unsigned long long)grid * 256 * iters, h[0], h[1], h[2], h[4]);
    return 0;
}
