// Shared device-side definitions of the TERSE/PROLIX kernels (gfx950 / CDNA4 only).
//
// Geometry: the codec block is 12 values (Terse.hpp:264,:478).  One lane owns one block, one
// 64-lane wavefront owns 64 consecutive blocks, one 256-thread workgroup owns a TILE of 256
// consecutive blocks (3072 values) of ONE frame.  Frames never share a tile: the header state
// resets per frame (Terse.hpp:505,:359) and frames start byte aligned (Terse.hpp:502-504).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trpx {

constexpr int kBlock = 12;          // values per codec block on the tuned path
constexpr int kWave = 64;
constexpr int kThreads = 256;       // threads per workgroup = blocks per tile
constexpr int kTileBlocks = kThreads;
constexpr int kTileValues = kTileBlocks * kBlock;

struct FrameGeom {
    uint64_t n_values;   // per frame
    uint32_t n_blocks;   // ceil(n_values / 12)
    uint32_t n_tiles;    // ceil(n_blocks / 256)
    uint32_t block;      // values per codec block: 12 on every tuned path; the generic kernels take any value
};

template <typename T> struct PixelTraits;
template <> struct PixelTraits<uint8_t>  { using U = uint8_t;  static constexpr int bits = 8;  static constexpr bool is_signed = false; };
template <> struct PixelTraits<int8_t>   { using U = uint8_t;  static constexpr int bits = 8;  static constexpr bool is_signed = true;  };
template <> struct PixelTraits<uint16_t> { using U = uint16_t; static constexpr int bits = 16; static constexpr bool is_signed = false; };
template <> struct PixelTraits<int16_t>  { using U = uint16_t; static constexpr int bits = 16; static constexpr bool is_signed = true;  };
template <> struct PixelTraits<uint32_t> { using U = uint32_t; static constexpr int bits = 32; static constexpr bool is_signed = false; };
template <> struct PixelTraits<int32_t>  { using U = uint32_t; static constexpr int bits = 32; static constexpr bool is_signed = true;  };
// 64-bit containers (what src/terse.cpp:120-123 makes of float / double images): generic correct-first kernels only
template <> struct PixelTraits<uint64_t> { using U = uint64_t; static constexpr int bits = 64; static constexpr bool is_signed = false; };
template <> struct PixelTraits<int64_t>  { using U = uint64_t; static constexpr int bits = 64; static constexpr bool is_signed = true;  };

// Worst-case bits of one block: 12-bit header + 12 full-width values.
template <typename T> constexpr int max_block_bits() { return 12 + kBlock * PixelTraits<T>::bits; }

// ---- header code (Terse.hpp:517-535 encode, :361-372 decode) --------------------------------
// Returns the header length in bits for width w following a block of width w_prev.
__device__ __forceinline__ uint32_t header_len(uint32_t w, uint32_t w_prev) {
    return w == w_prev ? 1u : (w < 7u ? 4u : (w < 10u ? 6u : 12u));
}
// Header value (LSB first, bit 0 = the "same" flag).
__device__ __forceinline__ uint32_t header_val(uint32_t w, uint32_t w_prev) {
    if (w == w_prev) return 1u;                              // Terse.hpp:518
    if (w < 7u) return w << 1;                               // 0, then 3 bits     (:523)
    if (w < 10u) return (7u + ((w - 7u) << 3)) << 1;         // 0, 111, 2 bits     (:527)
    return (31u + ((w - 10u) << 5)) << 1;                    // 0, 111, 11, 6 bits (:531)
}

// Significant-bit width from the OR-reduction of a block (Terse.hpp:508-515, :551-560).
// `m` is OR(v) for unsigned T and OR(|v|) for signed T, as an unsigned 32-bit magnitude.
template <typename T>
__device__ __forceinline__ uint32_t width_from_or(uint32_t m) {
    uint32_t bl = 32u - (uint32_t)__builtin_clz(m | 0u) ;
    bl = m ? bl : 0u;
    if (PixelTraits<T>::is_signed) {
        uint32_t w = m ? bl + 1u : 0u;
        return w > (uint32_t)PixelTraits<T>::bits ? (uint32_t)PixelTraits<T>::bits : w;  // outside D3's domain
    }
    return bl;
}

// The same for 64-bit containers (generic kernels): OR-magnitude as 64 bits.
template <typename T>
__device__ __forceinline__ uint32_t width_from_or64(uint64_t m) {
    const uint32_t bl = m ? 64u - (uint32_t)__builtin_clzll(m) : 0u;
    if (PixelTraits<T>::is_signed) {
        const uint32_t w = m ? bl + 1u : 0u;
        return w > (uint32_t)PixelTraits<T>::bits ? (uint32_t)PixelTraits<T>::bits : w;  // outside D3's domain
    }
    return bl;
}
template <typename T>
__device__ __forceinline__ uint64_t magnitude64(T v) {
    if (PixelTraits<T>::is_signed) {
        const int64_t s = (int64_t)v;
        return s < 0 ? (uint64_t)0 - (uint64_t)s : (uint64_t)s;       // |v| (Terse.hpp:514)
    }
    return (uint64_t)(typename PixelTraits<T>::U)v;
}

template <typename T>
__device__ __forceinline__ uint32_t magnitude(T v) {
    if (PixelTraits<T>::is_signed) {
        int32_t s = (int32_t)v;
        return s < 0 ? 0u - (uint32_t)s : (uint32_t)s;       // |v| (Terse.hpp:514)
    }
    return (uint32_t)(typename PixelTraits<T>::U)v;
}

// ---- wavefront (64-lane) primitives ----------------------------------------------------------
// Inclusive +scan across the 64 lanes with DPP row shifts / broadcasts (no LDS traffic).
// Must be called with all 64 lanes active.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return v;
}

// Maximum over the 64 lanes (same DPP ladder as the scan; every lane gets the result).
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    auto step = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
// wave index as a SCALAR (readfirstlane): branches on it are uniform, so whatever a single wave computes from
// wave-uniform values stays in SGPRs / on the scalar unit instead of being treated as divergent per-lane data
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Workgroup-wide exclusive scan of one u32 per thread (256 threads = 4 waves).
// `wave_tot` is a 4-entry LDS array.  Returns the exclusive prefix; *total gets the tile sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t* total) {
    uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == 63) wave_tot[wave_id()] = inc;
    __syncthreads();
    uint32_t t0 = wave_tot[0], t1 = wave_tot[1], t2 = wave_tot[2], t3 = wave_tot[3];
    int w = wave_id();
    uint32_t base = (w > 0 ? t0 : 0u) + (w > 1 ? t1 : 0u) + (w > 2 ? t2 : 0u);
    *total = t0 + t1 + t2 + t3;
    return base + inc - v;
}

// ---- clearing device words on the launch stream ------------------------------------------------
// Every call clears its status block (and the encoder its descriptor words) with this kernel instead of
// hipMemsetAsync: kernel nodes replay reliably when the call is captured into a hipGraph, memset nodes did
// not (ROCm 7.2 / torch 2.10: from the second replay on the words stayed uncleared; tests/test_gpu_graph.py).
template <int kUnused = 0>
__global__ __launch_bounds__(kThreads) void k_zero_words(uint64_t* __restrict__ p, uint64_t n, uint64_t* __restrict__ q,
                                                         uint64_t m) {
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) p[i] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x < m) q[threadIdx.x] = 0ull;
}
inline void zero_status(uint32_t* status, hipStream_t st) {           // 8 x u32, 8-byte aligned
    hipLaunchKernelGGL(k_zero_words<0>, dim3(1), dim3(kThreads), 0, st, static_cast<uint64_t*>(nullptr), (uint64_t)0,
                       reinterpret_cast<uint64_t*>(status), (uint64_t)4);
}

// ---- workspace layouts (device memory, carved by the host API) -------------------------------
// encode: [status-shadow 64 B][frame_size u64 x F][tile_off u64 x F*T][tile_bits u32 x F*T]
// decode: [tile_off u64 x F*T][widths u8 x F*nblocks (padded to 16)]
__host__ __device__ inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }
// The list of the frames the per-frame decoder hands over (DecodeArgs::defer): word 0 = count, words 1 .. n_frames = entries.
// In FRONT of it (DecodeArgs::defer points at the count): kDeferSlots accumulators of the stack's statistics, one 128-byte line
// each -- width changes << 32 | blocks, summed over the frames' first super-steps -- by which frames near the hand-over line
// decide (decode_frame.hip); cleared with the count.  The accumulators are word 0 of every line; the last line's last four words
// are the large-frame routes': [-1] k_seg_fallback's barrier counter, [-2] the number of listed frames with bit 31 set
// (header-dense large frames: k_seg_wg's, decode_seg.hip), [-3] / [-4] the index route's vote and verdict (ChainVote, decode_part.hip).
constexpr uint32_t kDeferSlots = 64, kDeferSlotWords = 16;                 // (u64 words per slot: a cache line)
constexpr size_t kDeferFront = 8u * kDeferSlots * kDeferSlotWords;         // bytes in front of the list
inline size_t defer_bytes(size_t n_frames) { return kDeferFront + align_up(4 * (n_frames + 2), 256); }

}  // namespace trpx
