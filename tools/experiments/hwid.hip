// Where do the four waves of a 256-thread workgroup land?  (HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 on gfx9)
// Build: hipcc -O3 --offload-arch=gfx950 tools/hwid.hip -o tools/hwid
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(uint32_t* out, int spin) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = id;
}
int main() {
    const int n = 2000;
    uint32_t* d; hipMalloc(&d, n * 16);
    k<<<n, 256>>>(d, 200);
    uint32_t* h = (uint32_t*)malloc(n * 16);
    hipMemcpy(h, d, n * 16, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < n; ++b) for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 6; ++b) printf("wg %d: %08x %08x %08x %08x\n", b, h[b*4], h[b*4+1], h[b*4+2], h[b*4+3]);
    return 0;
}
