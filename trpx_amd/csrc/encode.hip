// TERSE encode kernels for gfx950 (CDNA4).  Replaces jpa::Terse::f_compress
// (reference include/Terse.hpp:500-549) and the Bit_pointer.hpp pack primitives under it
// (Bit_range::append_range :700-730, operator|= :628-649, Bit::set :490).
//
// v1 pipeline (one launch each, all stream ordered, no host sync):
//   k_tile_bits   pixels -> bit length of every 256-block tile, prolix_bits (max width)
//   k_frame_scan  per frame: exclusive scan of tile bit lengths, S_f = 1 + bits/8 (Terse.hpp:547)
//   k_stack_scan  exclusive scan of S_f over the stack -> frame byte offsets (Terse.hpp:502)
//   k_zero_edges  zero the (few) output dwords that two tiles share
//   k_pack        pixels -> LDS bit staging at the scanned offsets -> coalesced dword stores
//
// Data layout in HBM: pixels [frame][value] contiguous; output = the compact reference stack;
// workspace = tile_bits u32[F*T], tile_off u64[F*T], frame_size u64[F].
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"

namespace trpx {

// ---------------------------------------------------------------------------------------------
// Loading one codec block (12 values) per lane.
// VEC path: n_values % 4 == 0 and 16-byte aligned base -> three (4*sizeof(T))-byte vector loads.
// ---------------------------------------------------------------------------------------------
template <typename T> struct alignas(4 * sizeof(T)) Quad { T x[4]; };

template <typename T, bool VEC>
__device__ __forceinline__ int load_block(const T* __restrict__ frame, uint64_t n_values, uint32_t b,
                                          T (&v)[kBlock]) {
    const uint64_t first = (uint64_t)b * kBlock;
    if (first + kBlock <= n_values) {
        if (VEC) {
            const Quad<T>* q = reinterpret_cast<const Quad<T>*>(frame + first);
            Quad<T> a = q[0], c = q[1], d = q[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = a.x[k]; v[4 + k] = c.x[k]; v[8 + k] = d.x[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < kBlock; ++k) v[k] = frame[first + k];
        }
        return kBlock;
    }
    int nb = (int)(n_values - first);                       // partial last block of the frame
#pragma unroll
    for (int k = 0; k < kBlock; ++k) v[k] = k < nb ? frame[first + k] : (T)0;
    return nb;
}

template <typename T>
__device__ __forceinline__ uint32_t block_width(const T (&v)[kBlock]) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < kBlock; ++k) m |= magnitude<T>(v[k]);   // OR-scan (Terse.hpp:508-514)
    return width_from_or<T>(m);
}

// Loads this thread's block, computes its width and -- through s_w -- the width of the block
// before it (0 at the start of a frame, Terse.hpp:505).  Contains one __syncthreads().
template <typename T, bool VEC>
__device__ __forceinline__ void load_and_widths(const T* __restrict__ frame, const FrameGeom& g, uint32_t t,
                                                uint32_t* s_w, T (&v)[kBlock], int& nb, uint32_t& w,
                                                uint32_t& w_prev, bool& valid) {
    const uint32_t tid = threadIdx.x;
    const uint32_t b = t * kTileBlocks + tid;
    valid = b < g.n_blocks;
    nb = 0;
    w = 0;
    if (valid) {
        nb = load_block<T, VEC>(frame, g.n_values, b, v);
        w = block_width<T>(v);
    }
    s_w[tid + 1] = w;
    if (tid == 0) {
        uint32_t hw = 0;
        if (t > 0) {                                        // halo: last block of the previous tile
            T h[kBlock];
            load_block<T, VEC>(frame, g.n_values, b - 1, h);
            hw = block_width<T>(h);
        }
        s_w[0] = hw;
    }
    __syncthreads();
    w_prev = s_w[tid];
}

// ---------------------------------------------------------------------------------------------
// K1: bit length of every tile + prolix_bits.
// ---------------------------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_tile_bits(const T* __restrict__ pixels, FrameGeom g,
                                                        uint32_t* __restrict__ tile_bits,
                                                        uint32_t* __restrict__ status) {
    __shared__ uint32_t s_w[kThreads + 1];
    __shared__ uint32_t s_tot[4];
    __shared__ uint32_t s_max[4];
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles);
    const uint32_t t = (uint32_t)(tile % g.n_tiles);
    const T* fp = pixels + (uint64_t)frame * g.n_values;

    T v[kBlock];
    int nb; uint32_t w, w_prev; bool valid;
    load_and_widths<T, VEC>(fp, g, t, s_w, v, nb, w, w_prev, valid);
    const uint32_t len = valid ? header_len(w, w_prev) + (uint32_t)nb * w : 0u;

    uint32_t total;
    block_exclusive_scan(len, s_tot, &total);
    uint32_t mx = wave_max(w);
    if (lane_id() == 0) s_max[wave_id()] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_bits[tile] = total;
        uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        // d_prolix_bits (Terse.hpp:516).  Read first: ~all tiles see a value that is already >= theirs,
        // and 172k same-address atomics would serialise at the memory side (~11 ns each).
        if (m > __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&status[1], m);
    }
}

// ---------------------------------------------------------------------------------------------
// K2a: per-frame exclusive scan of tile bits; frame size S_f = 1 + bits/8 (Terse.hpp:547).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_frame_scan(const uint32_t* __restrict__ tile_bits, FrameGeom g,
                                                         uint64_t* __restrict__ tile_off,
                                                         uint64_t* __restrict__ frame_size) {
    __shared__ uint32_t s_tot[4];
    const uint64_t frame = blockIdx.x;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < g.n_tiles; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t vlen = i < g.n_tiles ? tile_bits[frame * g.n_tiles + i] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(vlen, s_tot, &total);
        if (i < g.n_tiles) tile_off[frame * g.n_tiles + i] = carry + excl;
        carry += total;
        __syncthreads();                                    // s_tot is reused next iteration
    }
    if (threadIdx.x == 0) frame_size[frame] = 1 + carry / 8;
}

// ---------------------------------------------------------------------------------------------
// K2b: exclusive scan of frame sizes over the stack (single workgroup; F is small).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, off, 64);
        uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), off, 64);
        if (lane_id() >= off) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(kThreads) void k_stack_scan(const uint64_t* __restrict__ frame_size,
                                                         uint32_t n_frames, uint64_t out_capacity,
                                                         uint64_t* __restrict__ frame_offsets,
                                                         uint32_t* __restrict__ status) {
    __shared__ uint64_t s_tot[4];
    uint64_t carry = 0;
    for (uint32_t base = 0; base < n_frames; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < n_frames ? frame_size[i] : 0ull;
        const uint64_t inc = wave_inclusive_scan64(v);
        if (lane_id() == 63) s_tot[wave_id()] = inc;
        __syncthreads();
        uint64_t wbase = 0;
        for (int k = 0; k < wave_id(); ++k) wbase += s_tot[k];
        if (i < n_frames) frame_offsets[i] = carry + wbase + inc - v;
        carry += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        frame_offsets[n_frames] = carry;
        // dword-granular stores may touch up to 3 zero bytes past the stack's end
        if (align_up(carry, 4) > out_capacity) status[0] = 3u;   // TRPX_ERR_CAPACITY
    }
}

// ---------------------------------------------------------------------------------------------
// Output span of a tile in units of output dwords.
// ---------------------------------------------------------------------------------------------
struct Span {
    uint64_t d0;       // first output dword touched
    uint32_t s0;       // bit position of the tile's first bit inside dword d0
    uint32_t nbits;    // bits covered (the frame's last tile also covers the pad byte, Terse.hpp:547)
    uint32_t ndw;      // dwords touched
    bool first_partial, last_partial;
};

__device__ __forceinline__ Span tile_span(uint64_t frame_off, uint64_t frame_end, uint64_t t_off,
                                          uint32_t t_bits, bool last_tile) {
    Span s;
    const uint64_t p = 8 * frame_off + t_off;
    const uint64_t e = last_tile ? 8 * frame_end : p + t_bits;
    s.d0 = p >> 5;
    s.s0 = (uint32_t)(p & 31);
    s.nbits = (uint32_t)(e - p);
    const uint32_t span = s.s0 + s.nbits;
    s.ndw = (span + 31) >> 5;
    s.first_partial = s.s0 != 0 || span < 32;
    s.last_partial = (span & 31) != 0;
    return s;
}

// K2c: zero the dwords that k_pack will OR into (tile edges not aligned to a dword).
__global__ __launch_bounds__(kThreads) void k_zero_edges(FrameGeom g, uint32_t n_frames,
                                                         const uint64_t* __restrict__ tile_off,
                                                         const uint32_t* __restrict__ tile_bits,
                                                         const uint64_t* __restrict__ frame_offsets,
                                                         uint32_t* __restrict__ out32,
                                                         const uint32_t* __restrict__ status) {
    if (status[0] != 0) return;
    const uint64_t tile = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (tile >= (uint64_t)n_frames * g.n_tiles) return;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles);
    const uint32_t t = (uint32_t)(tile % g.n_tiles);
    const Span s = tile_span(frame_offsets[frame], frame_offsets[frame + 1], tile_off[tile], tile_bits[tile],
                             t + 1 == g.n_tiles);
    if (s.first_partial) out32[s.d0] = 0u;
    if (s.last_partial) out32[s.d0 + s.ndw - 1] = 0u;
}

// ---------------------------------------------------------------------------------------------
// K3: pack.  Each lane serialises its block (header + 12 fields of w bits, LSB first) into the
// workgroup's LDS staging image at its scanned bit offset; the image is then copied to the
// output with one coalesced dword store per lane (edge dwords shared with a neighbour tile are
// OR-ed into the pre-zeroed output instead).
// ---------------------------------------------------------------------------------------------
struct BitSink {
    uint32_t* stage;
    uint64_t acc;
    uint32_t fill;     // valid bits in acc, < 32 between puts
    uint32_t d;        // current staging dword
    bool first;        // the next flushed dword is this lane's first (may be shared)

    __device__ __forceinline__ void put(uint32_t val, uint32_t len) {   // len <= 32, val < 2^len
        acc |= (uint64_t)val << fill;
        fill += len;
        if (fill >= 32) {
            if (first) atomicOr(&stage[d], (uint32_t)acc);
            else stage[d] = (uint32_t)acc;                  // interior dword: owned by this lane alone
            first = false;
            ++d;
            acc >>= 32;
            fill -= 32;
        }
    }
    __device__ __forceinline__ void finish() {
        if (fill) atomicOr(&stage[d], (uint32_t)acc);
    }
};

template <typename T>
constexpr int stage_dwords() { return (31 + kTileBlocks * max_block_bits<T>() + 8 + 31) / 32 + 1; }

template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_pack(const T* __restrict__ pixels, FrameGeom g,
                                                   const uint64_t* __restrict__ tile_off,
                                                   const uint64_t* __restrict__ frame_offsets,
                                                   uint32_t* __restrict__ out32,
                                                   const uint32_t* __restrict__ status) {
    constexpr int kStage = stage_dwords<T>();
    __shared__ uint32_t s_stage[kStage];
    __shared__ uint32_t s_w[kThreads + 1];
    __shared__ uint32_t s_tot[4];
    if (status[0] != 0) return;                             // capacity error: write nothing
    const uint32_t tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles);
    const uint32_t t = (uint32_t)(tile % g.n_tiles);
    const T* fp = pixels + (uint64_t)frame * g.n_values;

    for (int i = tid; i < kStage; i += kThreads) s_stage[i] = 0u;

    T v[kBlock];
    int nb; uint32_t w, w_prev; bool valid;
    load_and_widths<T, VEC>(fp, g, t, s_w, v, nb, w, w_prev, valid);   // syncs (also covers the zeroing)
    const uint32_t hl = header_len(w, w_prev);
    const uint32_t len = valid ? hl + (uint32_t)nb * w : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(len, s_tot, &total);

    const Span s = tile_span(frame_offsets[frame], frame_offsets[frame + 1], tile_off[tile], total,
                             t + 1 == g.n_tiles);
    if (valid) {
        const uint32_t o = s.s0 + excl;
        BitSink sink{s_stage, 0ull, o & 31u, o >> 5, true};
        sink.put(header_val(w, w_prev), hl);
        if (w) {
            const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
            uint32_t u[kBlock];
#pragma unroll
            for (int k = 0; k < kBlock; ++k) u[k] = (uint32_t)v[k] & mask;   // Bit_pointer.hpp:707-710
            if (nb == kBlock) {
                if (w <= 8u) {                              // 4 values per <=32-bit field
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        sink.put(u[4 * q] | (u[4 * q + 1] << w) | (u[4 * q + 2] << (2 * w)) | (u[4 * q + 3] << (3 * w)), 4 * w);
                } else if (w <= 16u) {                      // 2 values per field
#pragma unroll
                    for (int p = 0; p < 6; ++p) sink.put(u[2 * p] | (u[2 * p + 1] << w), 2 * w);
                } else {
#pragma unroll
                    for (int k = 0; k < kBlock; ++k) sink.put(u[k], w);
                }
            } else {
#pragma unroll
                for (int k = 0; k < kBlock; ++k)
                    if (k < nb) sink.put(u[k], w);
            }
        }
        sink.finish();
    }
    __syncthreads();

    uint32_t* dst = out32 + s.d0;
    for (uint32_t j = tid; j < s.ndw; j += kThreads) {
        const uint32_t x = s_stage[j];
        const bool shared = (j == 0 && s.first_partial) || (j + 1 == s.ndw && s.last_partial);
        if (shared) { if (x) atomicOr(&dst[j], x); }
        else dst[j] = x;
    }
}

// ---------------------------------------------------------------------------------------------
// Host-side launcher (called by the C ABI in api.hip).
// ---------------------------------------------------------------------------------------------
template <typename T>
static hipError_t launch_encode_t(const EncodeArgs& a, hipStream_t st) {
    const FrameGeom g = a.geom;
    const uint64_t n_tiles_total = (uint64_t)a.n_frames * g.n_tiles;
    const bool vec = (g.n_values % 4 == 0) && ((uintptr_t)a.pixels % 16 == 0);
    const T* px = static_cast<const T*>(a.pixels);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(a.out);
    const dim3 grid((uint32_t)n_tiles_total), blk(kThreads);

    zero_status(a.status, st);
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    if (vec) hipLaunchKernelGGL((k_tile_bits<T, true>), grid, blk, 0, st, px, g, a.tile_bits, a.status);
    else     hipLaunchKernelGGL((k_tile_bits<T, false>), grid, blk, 0, st, px, g, a.tile_bits, a.status);
    prof.mark(st);
    hipLaunchKernelGGL(k_frame_scan, dim3(a.n_frames), blk, 0, st, a.tile_bits, g, a.tile_off, a.frame_size);
    prof.mark(st);
    hipLaunchKernelGGL(k_stack_scan, dim3(1), blk, 0, st, a.frame_size, a.n_frames, (uint64_t)a.out_capacity,
                       a.frame_offsets, a.status);
    prof.mark(st);
    hipLaunchKernelGGL(k_zero_edges, dim3((uint32_t)((n_tiles_total + kThreads - 1) / kThreads)), blk, 0, st, g,
                       a.n_frames, a.tile_off, a.tile_bits, a.frame_offsets, out32, a.status);
    prof.mark(st);
    if (vec) hipLaunchKernelGGL((k_pack<T, true>), grid, blk, 0, st, px, g, a.tile_off, a.frame_offsets, out32, a.status);
    else     hipLaunchKernelGGL((k_pack<T, false>), grid, blk, 0, st, px, g, a.tile_off, a.frame_offsets, out32, a.status);
    prof.mark(st);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Any block size (the `block` argument of Terse(Iterator, size, block), Terse.hpp:263-270, and the header's
// `block` attribute).  Correct-first kernels: one lane per block looping over its values, output pre-zeroed
// and written with 32-bit atomic ORs.  block == 12 never comes here.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ uint32_t block_width_g(const T* __restrict__ frame, uint64_t first, uint32_t nb) {
    if constexpr (sizeof(T) == 8) {
        uint64_t m = 0;
        for (uint32_t k = 0; k < nb; ++k) m |= magnitude64<T>(frame[first + k]);
        return width_from_or64<T>(m);
    } else {
        uint32_t m = 0;
        for (uint32_t k = 0; k < nb; ++k) m |= magnitude<T>(frame[first + k]);  // OR-scan (Terse.hpp:508-514)
        return width_from_or<T>(m);
    }
}

// width of block b and of the block before it (0 at the start of a frame); one __syncthreads()
template <typename T>
__device__ __forceinline__ void widths_g(const T* __restrict__ frame, const FrameGeom& g, uint32_t t, uint32_t* s_w,
                                         uint32_t& nb, uint32_t& w, uint32_t& w_prev, bool& valid) {
    const uint32_t tid = threadIdx.x;
    const uint32_t b = t * kTileBlocks + tid;
    valid = b < g.n_blocks;
    nb = 0;
    w = 0;
    if (valid) {
        const uint64_t first = (uint64_t)b * g.block;
        nb = (uint32_t)(first + g.block <= g.n_values ? g.block : g.n_values - first);
        w = block_width_g<T>(frame, first, nb);
    }
    s_w[tid + 1] = w;
    if (tid == 0) s_w[0] = t > 0 ? block_width_g<T>(frame, (uint64_t)(b - 1) * g.block, g.block) : 0u;
    __syncthreads();
    w_prev = s_w[tid];
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_tile_bits_g(const T* __restrict__ pixels, FrameGeom g,
                                                          uint32_t* __restrict__ tile_bits, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_w[kThreads + 1];
    __shared__ uint32_t s_tot[4];
    __shared__ uint32_t s_max[4];
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles), t = (uint32_t)(tile % g.n_tiles);
    uint32_t nb, w, w_prev;
    bool valid;
    widths_g<T>(pixels + (uint64_t)frame * g.n_values, g, t, s_w, nb, w, w_prev, valid);
    const uint32_t len = valid ? header_len(w, w_prev) + nb * w : 0u;
    uint32_t total;
    block_exclusive_scan(len, s_tot, &total);
    const uint32_t mx = wave_max(w);
    if (lane_id() == 0) s_max[wave_id()] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_bits[tile] = total;
        const uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (m > __hip_atomic_load(&status[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&status[1], m);
    }
}

__global__ __launch_bounds__(kThreads) void k_zero_out_g(const uint64_t* __restrict__ frame_offsets, uint32_t n_frames,
                                                         uint32_t* __restrict__ out32, const uint32_t* __restrict__ status) {
    if (status[0] != 0) return;
    const uint64_t n_dw = (frame_offsets[n_frames] + 3) / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n_dw; i += (uint64_t)gridDim.x * kThreads) out32[i] = 0u;
}

__device__ __forceinline__ void or_bits_g(uint32_t* __restrict__ out32, uint64_t abit, uint32_t val, uint32_t len) {
    if (!len) return;
    const uint64_t x = (uint64_t)val << (abit & 31);
    if ((uint32_t)x) atomicOr(&out32[abit >> 5], (uint32_t)x);
    if ((uint32_t)(x >> 32)) atomicOr(&out32[(abit >> 5) + 1], (uint32_t)(x >> 32));
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_pack_g(const T* __restrict__ pixels, FrameGeom g,
                                                     const uint64_t* __restrict__ tile_off,
                                                     const uint64_t* __restrict__ frame_offsets,
                                                     uint32_t* __restrict__ out32, const uint32_t* __restrict__ status) {
    __shared__ uint32_t s_w[kThreads + 1];
    __shared__ uint32_t s_tot[4];
    if (status[0] != 0) return;
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles), t = (uint32_t)(tile % g.n_tiles);
    const T* fp = pixels + (uint64_t)frame * g.n_values;
    uint32_t nb, w, w_prev;
    bool valid;
    widths_g<T>(fp, g, t, s_w, nb, w, w_prev, valid);
    const uint32_t hl = header_len(w, w_prev);
    const uint32_t len = valid ? hl + nb * w : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(len, s_tot, &total);
    if (!valid) return;
    uint64_t abit = 8 * frame_offsets[frame] + tile_off[tile] + excl;
    or_bits_g(out32, abit, header_val(w, w_prev), hl);
    abit += hl;
    if (w) {
        const uint64_t first = (uint64_t)(t * kTileBlocks + threadIdx.x) * g.block;
        if constexpr (sizeof(T) == 8) {                       // fields of up to 64 bits: low and high half
            const uint64_t mask = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
            for (uint32_t k = 0; k < nb; ++k, abit += w) {
                const uint64_t v = (uint64_t)fp[first + k] & mask;
                or_bits_g(out32, abit, (uint32_t)v, w < 32u ? w : 32u);
                if (w > 32u) or_bits_g(out32, abit + 32u, (uint32_t)(v >> 32), w - 32u);
            }
        } else {
            const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
            for (uint32_t k = 0; k < nb; ++k, abit += w) or_bits_g(out32, abit, (uint32_t)fp[first + k] & mask, w);   // Bit_pointer.hpp:707-711
        }
    }
}

template <typename T>
static hipError_t launch_encode_generic_t(const EncodeArgs& a, hipStream_t st) {
    const FrameGeom g = a.geom;
    const uint64_t n_tiles_total = (uint64_t)a.n_frames * g.n_tiles;
    const T* px = static_cast<const T*>(a.pixels);
    uint32_t* out32 = reinterpret_cast<uint32_t*>(a.out);
    const dim3 grid((uint32_t)n_tiles_total), blk(kThreads);
    zero_status(a.status, st);
    hipLaunchKernelGGL((k_tile_bits_g<T>), grid, blk, 0, st, px, g, a.tile_bits, a.status);
    hipLaunchKernelGGL(k_frame_scan, dim3(a.n_frames), blk, 0, st, a.tile_bits, g, a.tile_off, a.frame_size);
    hipLaunchKernelGGL(k_stack_scan, dim3(1), blk, 0, st, a.frame_size, a.n_frames, (uint64_t)a.out_capacity,
                       a.frame_offsets, a.status);
    hipLaunchKernelGGL(k_zero_out_g, dim3(1024), blk, 0, st, a.frame_offsets, a.n_frames, out32, a.status);
    hipLaunchKernelGGL((k_pack_g<T>), grid, blk, 0, st, px, g, a.tile_off, a.frame_offsets, out32, a.status);
    return hipGetLastError();
}

hipError_t launch_encode_generic(int dtype, const EncodeArgs& a, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_encode_generic_t<uint8_t>(a, st);
    case 1: return launch_encode_generic_t<int8_t>(a, st);
    case 2: return launch_encode_generic_t<uint16_t>(a, st);
    case 3: return launch_encode_generic_t<int16_t>(a, st);
    case 4: return launch_encode_generic_t<uint32_t>(a, st);
    case 5: return launch_encode_generic_t<int32_t>(a, st);
    case 8: return launch_encode_generic_t<uint64_t>(a, st);
    case 9: return launch_encode_generic_t<int64_t>(a, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_encode(int dtype, const EncodeArgs& a, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_encode_t<uint8_t>(a, st);
    case 1: return launch_encode_t<int8_t>(a, st);
    case 2: return launch_encode_t<uint16_t>(a, st);
    case 3: return launch_encode_t<int16_t>(a, st);
    case 4: return launch_encode_t<uint32_t>(a, st);
    case 5: return launch_encode_t<int32_t>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace trpx
