// VALU / SALU issue-rate microbenchmark: how many wave64 instructions per cycle does one CU retire at
// a given occupancy?  (Answers: is the encoder near the VALU issue limit?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = a * 3, c = a + 7, d = a ^ 0x55, e = a + 1, f = a + 2, g = a + 3, h = a + 4;
    uint32_t s = seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {          // 8 independent 32-bit VALU chains (v_xor / v_add / v_lshl_or mix)
                a = (a << 1) | b; b ^= c; c += d; d = (d << 3) | e; e ^= f; f += g; g = (g << 5) | h; h ^= a;
            } else if (MODE == 1) {   // same VALU + one SALU op per VALU pair
                a = (a << 1) | b; b ^= c; s = s * 5 + 1; c += d; d = (d << 3) | e; s ^= (s >> 3); e ^= f; f += g; s += 77; g = (g << 5) | h; h ^= a; s = s * 3 + 9;
            } else {                  // 64-bit shifts (the funnel used by the packer)
                uint64_t x = ((uint64_t)a << 32) | b; x >>= (c & 31); a = (uint32_t)x; b ^= (uint32_t)(x >> 32);
                uint64_t y = ((uint64_t)d << 32) | e; y >>= (f & 31); d = (uint32_t)y; e ^= (uint32_t)(y >> 32);
                c += g; f += h;
            }
        }
    }
    if ((a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ s) == 0x12345) out[0] = a;
}

int main() {
    uint32_t* out; CK(hipMalloc(&out, 4));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("clock %d kHz, CUs %d\n", p.clockRate, p.multiProcessorCount);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int wg_per_cu : {1, 2, 4, 8}) {
            const int grid = 256 * wg_per_cu;
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            auto run = [&] { if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1u);
                             else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1u);
                             else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1u); };
            run(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); run(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            const double valu_per_wave = (double)iters * 16 * (mode == 2 ? 10 : 8);   // rough count of VALU per wave
            const double waves_per_cu = 4.0 * wg_per_cu;
            const double per_simd_per_us = valu_per_wave * waves_per_cu / 4 / (ms * 1e3);
            printf("mode %d  %d WG/CU (%2.0f waves/CU): %.3f ms  -> %.0f VALU/us/SIMD = one per %.2f cycles @2.4GHz\n", mode, wg_per_cu,
                   waves_per_cu, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
        }
    return 0;
}
