// Write-bandwidth microbenchmark for the PROLIX decode floor (MI355X): what does a kernel that only writes
// 1 GB of pixels reach with (a) full-line 16-byte-per-lane stores, (b) the decoder's 24-byte-per-lane
// 16 + 8 byte stores, each plain and non-temporal, and (c) a decode-shaped mix (read 0.2 B per byte written)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/wrbench.hip -o tools/wrbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void fill16(u4* __restrict__ out, size_t n16) {
    const u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// lane owns 24 consecutive bytes: one 16-byte + one 8-byte store (the decoder's block store for 16-bit pixels)
template <bool NT>
__global__ __launch_bounds__(256) void fill24(uint8_t* __restrict__ out, size_t n24) {
    const u4 a = {threadIdx.x, blockIdx.x, 3u, 4u};
    const u2 b = {5u, 6u};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n24; i += (size_t)gridDim.x * 256) {
        uint8_t* p = out + i * 24;
        if (NT) { __builtin_nontemporal_store(a, (u4*)p); __builtin_nontemporal_store(b, (u2*)(p + 16)); }
        else { *(u4*)p = a; *(u2*)(p + 16) = b; }
    }
}
// the same 24 bytes per lane, but exchanged through LDS inside the wavefront so that the wave issues three
// 16-byte stores per two blocks, each covering 1 KB of consecutive bytes
template <bool NT>
__global__ __launch_bounds__(256) void fill24t(uint8_t* __restrict__ out, size_t n48) {
    __shared__ uint32_t s[4][64 * 12 + 8];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* w = s[wave];
    for (size_t g = (size_t)blockIdx.x * 4 + wave; g * 64 < n48; g += (size_t)gridDim.x * 4) {
        // lane's two blocks: 128 blocks per wave step = 3072 bytes
        for (int k = 0; k < 2; ++k) {
            u2* q = (u2*)(w + (k * 64 + lane) * 6);
            q[0] = u2{lane, (uint32_t)g}; q[1] = u2{3u, 4u}; q[2] = u2{5u, 6u};
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        uint8_t* base = out + g * 3072;
        for (int k = 0; k < 3; ++k) {
            const u4 v = *(const u4*)(w + (k * 64 + lane) * 4);
            if (NT) __builtin_nontemporal_store(v, (u4*)(base + (k * 64 + lane) * 16)); else *(u4*)(base + (k * 64 + lane) * 16) = v;
        }
    }
}
// decode-shaped: read 16 bytes for every 80 written (stream : pixels = 0.194)
template <bool NT>
__global__ __launch_bounds__(256) void mix(const u4* __restrict__ in, u4* __restrict__ out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        u4 v = in[i];
        for (int k = 0; k < 5; ++k) {
            v.x += k;
            if (NT) __builtin_nontemporal_store(v, out + i * 5 + k); else out[i * 5 + k] = v;
        }
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t *out, *in;
    CK(hipMalloc(&out, bytes + 4096));
    CK(hipMalloc(&in, bytes / 5 + 4096));
    CK(hipMemset(in, 1, bytes / 5));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    for (int grid : {2048, 4096, 16384}) {
        const float a = time([&] { hipLaunchKernelGGL(fill16<false>, dim3(grid), dim3(256), 0, 0, (u4*)out, bytes / 16); });
        const float b = time([&] { hipLaunchKernelGGL(fill16<true>, dim3(grid), dim3(256), 0, 0, (u4*)out, bytes / 16); });
        const float c = time([&] { hipLaunchKernelGGL(fill24<false>, dim3(grid), dim3(256), 0, 0, out, bytes / 24); });
        const float d = time([&] { hipLaunchKernelGGL(fill24<true>, dim3(grid), dim3(256), 0, 0, out, bytes / 24); });
        const float e = time([&] { hipLaunchKernelGGL(fill24t<false>, dim3(grid), dim3(256), 0, 0, out, bytes / 48); });
        const float f = time([&] { hipLaunchKernelGGL(fill24t<true>, dim3(grid), dim3(256), 0, 0, out, bytes / 48); });
        const float g = time([&] { hipLaunchKernelGGL(mix<false>, dim3(grid), dim3(256), 0, 0, (const u4*)in, (u4*)out, bytes / 80); });
        const float h = time([&] { hipLaunchKernelGGL(mix<true>, dim3(grid), dim3(256), 0, 0, (const u4*)in, (u4*)out, bytes / 80); });
        const double gb = bytes / 1e6;
        printf("grid %6d write GB/s: fill16 %.0f nt %.0f | fill24 %.0f nt %.0f | fill24 via LDS %.0f nt %.0f | mix(r+w bytes) %.0f nt %.0f\n", grid,
               gb / a, gb / b, gb / c, gb / d, gb / e, gb / f, gb * 1.2 / g, gb * 1.2 / h);
    }
    return 0;
}
