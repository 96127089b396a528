// What is the shader clock RIGHT NOW?  One wavefront runs a fixed chain of dependent VALU operations and stamps it with both
// hardware counters: s_memrealtime (constant 100 MHz) and s_memtime (clock64(): shader cycles where the part counts them).  The
// time of the chain in 100 MHz ticks scales with 1 / shader clock whatever s_memtime turns out to count.  Launched in-stream right
// behind the work whose clock is in question (tools/experiments/idx_gap6.py).  Not part of the library.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/clockprobe.hip -o tools/libclockprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ __launch_bounds__(64) void k_clock_probe(uint64_t* out, int n_ops) {
    uint32_t x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9e3779b9u;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n_ops; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x = x * 1664525u + y; y = (y >> 3) ^ x; }          // every operation waits for the one before it
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; out[2] = (uint64_t)(x ^ y); }
}
extern "C" int clock_probe(uint64_t* out3, int n_ops, void* stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out3, n_ops);
    return (int)hipGetLastError();
}
