// Calibration microbenchmarks for the encode/decode kernels' memory patterns on MI355X:
//   read16   : coalesced 16 B/lane streaming read (the achievable-HBM reference, ~6.3 TB/s)
//   read24s  : each lane reads its own 24-byte block at 24-byte stride (dwordx4 + dwordx2): the
//              access shape of "one codec block of 12 u16 per lane"
//   read24l  : the same bytes fetched coalesced (16 B/lane) into LDS, then 3 x ds_read_b64 per lane
//   copy16   : 16 B/lane read + write
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read16(const uint4* __restrict__ in, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v = in[i];
        acc |= v.x | v.y | v.z | v.w;
    }
    if (acc == 0x12345678) *sink = acc;
}
__global__ __launch_bounds__(256) void read16nt(const uint4* __restrict__ in, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(in) + i);
        acc |= v.x | v.y | v.z | v.w;
    }
    if (acc == 0x12345678) *sink = acc;
}
struct alignas(8) B24 { uint2 a, b, c; };
__global__ __launch_bounds__(256) void read24s(const B24* __restrict__ in, size_t n24, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n24; i += (size_t)gridDim.x * 256) {
        B24 v = in[i];
        acc |= v.a.x | v.a.y | v.b.x | v.b.y | v.c.x | v.c.y;
    }
    if (acc == 0x12345678) *sink = acc;
}
// one workgroup tile = 256 blocks = 6144 B = 384 x 16 B
__global__ __launch_bounds__(256) void read24l(const uint4* __restrict__ in, size_t ntiles, uint32_t* sink) {
    __shared__ uint4 s[384];
    uint32_t acc = 0;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint4* p = in + t * 384;
        uint4 a = p[threadIdx.x];
        uint4 b = threadIdx.x < 128 ? p[256 + threadIdx.x] : uint4{0, 0, 0, 0};
        __syncthreads();
        s[threadIdx.x] = a;
        if (threadIdx.x < 128) s[256 + threadIdx.x] = b;
        __syncthreads();
        const uint2* q = reinterpret_cast<const uint2*>(s) + threadIdx.x * 3;
        uint2 x = q[0], y = q[1], z = q[2];
        acc |= x.x | x.y | y.x | y.y | z.x | z.y;
    }
    if (acc == 0x12345678) *sink = acc;
}
__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

template <typename F> float time_ms(F f, int reps = 10) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2000 * 512 * 512 * 2;     // the 2000-frame u16 stack: 1.05 GB
    void *in, *out; uint32_t* sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, bytes));
    const size_t n16 = bytes / 16, n24 = bytes / 24, ntiles = bytes / 6144;
    for (int grid : {2048, 4096, 8192, 16384, 65536, (int)ntiles}) {
        float a = time_ms([&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const uint4*)in, n16, sink); });
        float an = time_ms([&] { hipLaunchKernelGGL(read16nt, dim3(grid), dim3(256), 0, 0, (const uint4*)in, n16, sink); });
        float b = time_ms([&] { hipLaunchKernelGGL(read24s, dim3(grid), dim3(256), 0, 0, (const B24*)in, n24, sink); });
        float c = time_ms([&] { hipLaunchKernelGGL(read24l, dim3(grid), dim3(256), 0, 0, (const uint4*)in, ntiles, sink); });
        float d = time_ms([&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)in, (uint4*)out, n16); });
        printf("grid %7d: read16 %.3f ms %.0f GB/s | read16nt %.3f ms %.0f GB/s | read24s %.3f ms %.0f GB/s | read24l %.3f ms %.0f GB/s | copy16 %.3f ms %.0f GB/s (r+w)\n",
               grid, a, bytes / a / 1e6, an, bytes / an / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, 2.0 * bytes / d / 1e6);
    }
    return 0;
}
