// Cold instruction-cache cost: every workgroup runs N straight-line VALU instructions once (N*8 bytes of code).
// First-wave workgroups (cold I-cache) vs later ones (warm) -> microseconds per KB of cold code.
// build: hipcc -O3 --offload-arch=gfx950 tools/icachebench.hip -o tools/icachebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int N>
__global__ __launch_bounds__(256, 4) void k(uint64_t* stamps, uint32_t* sink) {
    __shared__ uint32_t lds[9800];
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    uint64_t t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x), "+v"(y));
#pragma unroll
    for (int j = 0; j < N; ++j) { x = x * (1664525u + 2 * j) + y; y = (y >> 3) ^ x ^ (uint32_t)(j * 977); }   // distinct literals: no loop re-rolling
    uint64_t t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x), "+v"(y));
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
    if (x == 12345u) sink[0] = y + lds[threadIdx.x];
}
template <int N> void run(uint64_t* d, uint32_t* s) {
    const int wgs = 4096;
    std::vector<uint64_t> h(wgs * 2);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<N>, dim3(wgs), dim3(256), 0, 0, d, s); hipDeviceSynchronize(); }
    hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
    printf("N=%d straight-line iterations (~%d KB of code):", N, N * 32 / 1024);
    for (int g = 0; g < wgs; g += 1024) {
        double du = 0;
        for (int i = g; i < g + 1024; ++i) du += (h[2 * i + 1] - h[2 * i]) / 100.0;
        printf("  wave%d %.2f us", g / 1024, du / 1024);
    }
    printf("\n");
}
int main() {
    uint64_t* d; uint32_t* s;
    hipMalloc(&d, 4096 * 16); hipMalloc(&s, 4);
    run<128>(d, s); run<256>(d, s); run<512>(d, s); run<1024>(d, s); run<1536>(d, s); run<2048>(d, s);
    return 0;
}
