// Does ALU throughput ramp up during the first tens of microseconds of a launch?  Every workgroup runs the same
// fixed VALU loop and stamps its start / end with s_memrealtime (100 MHz); prints the loop time per launch phase.
// build: hipcc -O3 --offload-arch=gfx950 tools/rampbench.hip -o tools/rampbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256, 4) void k(uint64_t* stamps, uint32_t* sink, int iters, int mode) {
    __shared__ uint32_t lds[9800];
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9e3779b9u;
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { x = x * 1664525u + y; y = (y >> 3) ^ x; }
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { atomicOr(&lds[(x >> 7) % 9800], y); x = x * 1664525u + y; y = (y >> 3) ^ x; }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
    if (x == 12345u) sink[0] = y + lds[threadIdx.x];
}
int main() {
    const int wgs = 8192;
    uint64_t* d; uint32_t* s;
    hipMalloc(&d, wgs * 16); hipMalloc(&s, 4);
    std::vector<uint64_t> h(wgs * 2);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, s, mode == 0 ? 200 : 100, mode);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, wgs * 16, hipMemcpyDeviceToHost);
        uint64_t t0 = h[0];
        for (int i = 0; i < wgs; ++i) t0 = std::min(t0, h[2 * i]);
        printf("mode %d (%s): per 1024 workgroups: mean start us / mean duration us\n", mode, mode ? "LDS atomics" : "VALU");
        for (int g = 0; g < wgs; g += 1024) {
            double st = 0, du = 0;
            for (int i = g; i < g + 1024; ++i) { st += (h[2 * i] - t0) / 100.0; du += (h[2 * i + 1] - h[2 * i]) / 100.0; }
            printf("  wg %5d..: start %8.2f  duration %8.2f\n", g, st / 1024, du / 1024);
        }
    }
    return 0;
}
