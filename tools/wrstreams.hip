// Write-pattern microbenchmark for k_decode_frames (MI355X): 2000 workgroups, each writing ITS OWN output stream
// sequentially (a frame's pixels), three waves per workgroup, 64 x 24 bytes (16 + 8, non-temporal) per wave step --
// the decoder's store pattern without any decoding.  Question: does the distance between the streams (512 KB =
// a 512x512 u16 frame, a power of two) or the order in which a workgroup's waves cover a frame matter?
// Build: hipcc -O3 --offload-arch=gfx950 tools/wrstreams.hip -o tools/wrstreams
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// mode 0: wave k of a step writes groups 4k..4k+3 (the decoder's order); mode 1: groups interleaved k, k+3, k+6, k+9;
// mode 2: full 16-byte-per-lane stores, wave k writes 6 KB contiguous as 6 x 1 KB
template <int MODE>
__global__ __launch_bounds__(256) void streams(uint8_t* __restrict__ out, size_t stride, size_t frame_bytes, int spin) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* base = out + (size_t)blockIdx.x * stride;
    const u4 a = {threadIdx.x, blockIdx.x, 3u, 4u};
    const u2 b = {5u, 6u};
    const size_t step_bytes = 12 * 1536;
    for (size_t off = 0; off + step_bytes <= frame_bytes; off += step_bytes) {
        if (wave) {
            for (int g = 0; g < 4; ++g) {
                if (MODE == 3) {        // quads of lanes: 96 bytes = one 64-byte run (4 lanes x 16) + one 32-byte run (2 lanes x 16)
                    uint8_t* p = base + off + (size_t)((wave - 1) * 4 + g) * 1536 + (lane >> 2) * 96;
                    __builtin_nontemporal_store(a, (u4*)(p + (lane & 3) * 16));
                    if ((lane & 3) < 2) __builtin_nontemporal_store(a, (u4*)(p + 64 + (lane & 3) * 16));
                } else if (MODE == 4) { // pairs of lanes: 48 bytes = one 32-byte run + one 16-byte piece
                    uint8_t* p = base + off + (size_t)((wave - 1) * 4 + g) * 1536 + (lane >> 1) * 48;
                    __builtin_nontemporal_store(a, (u4*)(p + (lane & 1) * 16));
                    if ((lane & 1) == 0) __builtin_nontemporal_store(a, (u4*)(p + 32));
                } else if (MODE == 5) { // 16 lanes: 384 bytes = 256-byte run (16 lanes) + 128-byte run (8 lanes)
                    uint8_t* p = base + off + (size_t)((wave - 1) * 4 + g) * 1536 + (lane >> 4) * 384;
                    __builtin_nontemporal_store(a, (u4*)(p + (lane & 15) * 16));
                    if ((lane & 15) < 8) __builtin_nontemporal_store(a, (u4*)(p + 256 + (lane & 15) * 16));
                } else if (MODE == 2) {
                    uint8_t* p = base + off + (size_t)(wave - 1) * 6144 + g * 1536;
                    __builtin_nontemporal_store(a, (u4*)(p + lane * 16));
                    if (lane < 32) __builtin_nontemporal_store(a, (u4*)(p + 1024 + lane * 16));
                } else {
                    const int gi = MODE == 0 ? (int)(wave - 1) * 4 + g : g * 3 + (int)(wave - 1);
                    uint8_t* p = base + off + (size_t)gi * 1536 + lane * 24;
                    __builtin_nontemporal_store(a, (u4*)p);
                    __builtin_nontemporal_store(b, (u2*)(p + 16));
                }
            }
        } else {
            for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);      // the walker's time per super-step
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int n = 2000;
    const size_t frame = 512 * 512 * 2;
    uint8_t* out;
    const size_t cap = (size_t)n * (frame + 65536) + (1 << 20);
    CK(hipMalloc(&out, cap));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        return best;
    };
    for (int spin : {0, 20, 40}) {
        for (size_t extra : {(size_t)0}) {
            const size_t stride = frame + extra;
            const float a = time([&] { hipLaunchKernelGGL(streams<0>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            const float b = time([&] { hipLaunchKernelGGL(streams<1>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            const float c = time([&] { hipLaunchKernelGGL(streams<2>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            const float d = time([&] { hipLaunchKernelGGL(streams<3>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            const float e = time([&] { hipLaunchKernelGGL(streams<4>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            const float f = time([&] { hipLaunchKernelGGL(streams<5>, dim3(n), dim3(256), 0, 0, out, stride, frame, spin); });
            printf("spin %2d stride 512K+%6zu: decoder order %.3f ms (%.0f GB/s) | interleaved %.3f ms | full-line %.3f ms | quads %.3f | pairs %.3f | rows16 %.3f\n", spin, extra, a,
                   n * (double)frame / a / 1e6, b, c, d, e, f);
        }
    }
    return 0;
}
