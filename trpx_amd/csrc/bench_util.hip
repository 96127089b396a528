// Measured memory ceilings for bench.py's roofline block (SURVEY.md section 8 row d: "also measure a plain device read /
// copy kernel on the box and report the fraction of both nominal and measured-achievable").  Not part of the codec: three
// grid-stride streaming kernels, 16 bytes per lane, non-temporal, the access shape the encoder's pixel loads and the
// decoder's pixel stores aim for -- and a fourth that writes the way the decoder's stores look when a frame starts inside a
// cache line (16 bytes per lane from a 2-byte aligned address: every wavefront's first and last line is shared with a neighbour's
// store): the ceiling for the kernels that keep such stores -- 3.5-5.0 TB/s over nine boxes of the pool against 4.8-6.7 aligned
// (tools/box_kind.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trpx_hip.h"

namespace {

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef u4 u4_a2 __attribute__((aligned(2)));                 // (dst is aligned to 16-bit pixels only: like unpack_common.hpp's u4_a1)
constexpr int kBenchThreads = 256;

__global__ __launch_bounds__(kBenchThreads) void k_bench_read(const u4* __restrict__ in, uint64_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBenchThreads + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kBenchThreads) {
        const u4 v = __builtin_nontemporal_load(in + i);
        acc |= v.x | v.y | v.z | v.w;
    }
    if (acc == 0x12345678u) *sink = acc;                     // (never true for the buffers bench.py passes; keeps the loads alive)
}
__global__ __launch_bounds__(kBenchThreads) void k_bench_write(u4* __restrict__ out, uint64_t n16) {
    const u4 v = {1u, 2u, 3u, 4u};
    for (uint64_t i = (uint64_t)blockIdx.x * kBenchThreads + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kBenchThreads)
        __builtin_nontemporal_store(v, out + i);
}
__global__ __launch_bounds__(kBenchThreads) void k_bench_write_misaligned(u4_a2* __restrict__ out, uint64_t n16) {
    const u4 v = {1u, 2u, 3u, 4u};
    for (uint64_t i = (uint64_t)blockIdx.x * kBenchThreads + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kBenchThreads)
        __builtin_nontemporal_store(v, out + i);
}
__global__ __launch_bounds__(kBenchThreads) void k_bench_copy(const u4* __restrict__ in, u4* __restrict__ out, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * kBenchThreads + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * kBenchThreads)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

}  // namespace

extern "C" int trpx_bench_stream(int mode, const void* src, void* dst, size_t bytes, void* stream) {
    if (mode == 3) {                                          // write from a 2-byte aligned address (any; bench.py passes base + 2)
        if (bytes < 16 || !dst || (uintptr_t)dst % 2) return TRPX_ERR_INVALID_ARG;
        const uint64_t n16m = bytes / 16, wantm = (n16m + kBenchThreads - 1) / kBenchThreads;
        hipLaunchKernelGGL(k_bench_write_misaligned, dim3((uint32_t)(wantm < 16384 ? wantm : 16384)), dim3(kBenchThreads), 0,
                           static_cast<hipStream_t>(stream), static_cast<u4_a2*>(dst), n16m);
        return hipGetLastError() == hipSuccess ? TRPX_OK : TRPX_ERR_HIP;
    }
    if (mode < 0 || mode > 2 || bytes < 16 || ((mode != 1) && (!src || (uintptr_t)src % 16)) || ((mode != 0) && (!dst || (uintptr_t)dst % 16)))
        return TRPX_ERR_INVALID_ARG;
    if (mode == 0 && (!dst || (uintptr_t)dst % 4)) return TRPX_ERR_INVALID_ARG;   // the read kernel's 4-byte sink
    const uint64_t n16 = bytes / 16;
    const uint64_t want = (n16 + kBenchThreads - 1) / kBenchThreads;
    const dim3 grid((uint32_t)(want < 16384 ? want : 16384));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 0) hipLaunchKernelGGL(k_bench_read, grid, dim3(kBenchThreads), 0, st, static_cast<const u4*>(src), n16, static_cast<uint32_t*>(dst));
    else if (mode == 1) hipLaunchKernelGGL(k_bench_write, grid, dim3(kBenchThreads), 0, st, static_cast<u4*>(dst), n16);
    else hipLaunchKernelGGL(k_bench_copy, grid, dim3(kBenchThreads), 0, st, static_cast<const u4*>(src), static_cast<u4*>(dst), n16);
    return hipGetLastError() == hipSuccess ? TRPX_OK : TRPX_ERR_HIP;
}
