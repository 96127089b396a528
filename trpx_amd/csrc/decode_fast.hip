// PROLIX decode, tuned kernels for gfx950 (CDNA4).  Replaces jpa::Terse::prolix(Iterator, frame)
// (reference include/Terse.hpp:352-389) and Bit_range::get_range / operator T()
// (Bit_pointer.hpp:742-792, :597-617).  Same two stages as decode.hip, restructured:
//
//   k_walk_lds      one wavefront per frame walks the header chain (Terse.hpp:360-372) with the
//                   frame's stream staged through LDS in 16 KB chunks: a step costs an LDS round
//                   trip instead of an L2/HBM one.  64 candidate blocks are tested per step, so a run
//                   of equal-width blocks is one step.  Emits width[b] (u8) and the bit offset of
//                   every 256-block group.
//   k_unpack_tiles  tile = 1024 blocks (512 for 32-bit pixels), 4 (2) blocks per lane.  widths ->
//                   lengths -> wave/LDS scan -> the tile's stream bytes are fetched once, coalesced,
//                   into LDS -> every lane extracts its 12 fields with code specialised on the block
//                   width (static shifts, v_bfe) -> 24/48-byte streaming stores.
//
// HBM traffic per frame: S (stream, read twice: walk + unpack) + n_blocks (widths, written+read)
// + N*sizeof(T) (pixels, written once).  Algorithmic bytes: S + N*sizeof(T).
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include "unpack_common.hpp"
#include "unpack_tile.hpp"
#include "walk_lds.hpp"
#include <stdlib.h>
#include <string.h>

namespace trpx {

// ---------------------------------------------------------------------------------------------
// k_walk_lds (body: walk_lds.hpp)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void k_walk_lds(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                    const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                    uint32_t max_w, uint8_t* __restrict__ widths,
                                                    uint64_t* __restrict__ tile_off, const uint32_t* __restrict__ only,
                                                    const uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_chunk[kWalkChunkDw + 4];
    uint64_t frame = blockIdx.x;
    if (list) {                                       // (the frames k_decode_frames listed, see decode_seg.hip)
        if (blockIdx.x >= list[0]) return;
        frame = list[1u + blockIdx.x] & 0x7FFFFFFFu;
    }
    if (only && !only[frame]) return;                 // (frames the position-parallel walk has done)
    walk_lds_frame(terse, terse_bytes, frame_offsets, g, max_w, widths, tile_off, frame, s_chunk, status);
}

// ---------------------------------------------------------------------------------------------
// k_unpack_tiles
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 4 : 5) void k_unpack_tiles(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                              const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                              uint32_t tiles_per_frame,
                                                              const uint8_t* __restrict__ widths,
                                                              const uint64_t* __restrict__ tile_off,
                                                              T* __restrict__ pixels_out, uint32_t* __restrict__ status,
                                                              const uint32_t* __restrict__ frame_mode) {
    __shared__ uint32_t s_image[unpack_image_dwords<T>()];
    __shared__ uint32_t s_wtot[unpack_sub_tiles<T>() * 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[unpack_stage_dwords<T>()];
    if (status[0] != 0) return;                             // corrupt chain: produce nothing
    const uint64_t tile = blockIdx.x;
    if (frame_mode && frame_mode[tile / tiles_per_frame] == 0u) return;   // (large frames: this one is extracted part by part, decode_part.hip)
    unpack_tile<T>(terse, terse_bytes, frame_offsets, g, (uint32_t)(tile / tiles_per_frame), (uint32_t)(tile % tiles_per_frame),
                   widths, tile_off, pixels_out, status, s_image, s_wtot, s_stage);
}

// ---------------------------------------------------------------------------------------------
// Group states (row f1 for files): the chain state (bit offset, width of the block before) at every 256th block is
// all a decoder needs to walk a frame's groups independently.  k_index_group_states reads the states off a decode
// index; k_walk_groups rebuilds the index from them -- one lane per group, 256 dependent steps, every group checked
// against its successor's state (the last one against S_f = 1 + bits/8, Terse.hpp:547).
// ---------------------------------------------------------------------------------------------
constexpr uint64_t kStateOffMask = (1ull << 40) - 1;

__global__ __launch_bounds__(kThreads) void k_index_group_states(const uint8_t* __restrict__ widths, const uint64_t* __restrict__ tile_off,
                                                                 FrameGeom g, uint64_t n_groups_total, uint64_t* __restrict__ states) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_groups_total) return;
    const uint64_t frame = i / g.n_tiles, k = i % g.n_tiles;
    const uint32_t w_prev = k ? widths[frame * g.n_blocks + k * kTileBlocks - 1] : 0u;
    states[i] = (tile_off[i] & kStateOffMask) | ((uint64_t)w_prev << 40);
}

__global__ __launch_bounds__(kThreads) void k_walk_groups(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                          const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                          const uint64_t* __restrict__ states, uint64_t n_groups_total,
                                                          uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                          uint32_t* __restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_groups_total) return;
    const uint64_t frame = i / g.n_tiles;
    const uint32_t k = (uint32_t)(i % g.n_tiles);
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    bool bad = !(fe > fo && fe <= terse_bytes);
    const uint64_t limit = bad ? 0 : 8 * (fe - fo);
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const uint64_t st = states[i];
    uint64_t pos = st & kStateOffMask;
    uint32_t w = (uint32_t)(st >> 40);
    if (k == 0 && st != 0) bad = true;                       // a frame starts at bit 0 with width 0 (Terse.hpp:359, :505)
    if (pos > limit || w > max_w) bad = true;
    const uint32_t b0 = k * kTileBlocks;
    const uint32_t b1 = b0 + kTileBlocks < g.n_blocks ? b0 + kTileBlocks : g.n_blocks;
    const uint32_t nb_last = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock);
    if (!bad) tile_off[i] = pos;
    for (uint32_t b = b0; b < b1 && !bad; ++b) {
        const uint64_t abit = 8 * fo + pos, d = abit >> 5;
        const uint64_t two = (uint64_t)(d < n_dw ? s32[d] : 0u) | ((uint64_t)(d + 1 < n_dw ? s32[d + 1] : 0u) << 32);
        const uint32_t bits = (uint32_t)(two >> (abit & 31u));
        uint32_t hl = 1;
        if (!(bits & 1u)) {                                  // Terse.hpp:361-370
            w = (bits >> 1) & 7u; hl = 4;
            if (w == 7u) {
                w += (bits >> 4) & 3u; hl = 6;
                if (w == 10u) { w += (bits >> 6) & 63u; hl = 12; }
            }
        }
        if (w > max_w) { bad = true; break; }
        widths[frame * g.n_blocks + b] = (uint8_t)w;
        pos += hl + (uint64_t)(b + 1 == g.n_blocks ? nb_last : (uint32_t)kBlock) * w;
        if (pos > limit) bad = true;
    }
    if (!bad) {
        if (b1 < g.n_blocks) bad = ((pos & kStateOffMask) | ((uint64_t)w << 40)) != states[i + 1];   // lands in the next group's state
        else bad = 1 + pos / 8 != fe - fo;                                                           // S_f (Terse.hpp:547)
    }
    if (bad) atomicMax(&status[0], 5u);                      // TRPX_ERR_CORRUPT
}

hipError_t launch_index_group_states(const DecodeArgs& a, uint64_t* states, hipStream_t st) {
    const uint64_t n = (uint64_t)a.n_frames * a.geom.n_tiles;
    hipLaunchKernelGGL(k_index_group_states, dim3((uint32_t)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, a.widths, a.tile_off,
                       a.geom, n, states);
    return hipGetLastError();
}

hipError_t launch_walk_groups(const DecodeArgs& a, uint32_t max_w, const uint64_t* states, bool clear_status, hipStream_t st) {
    if (clear_status) zero_status(a.status, st);
    const uint64_t n = (uint64_t)a.n_frames * a.geom.n_tiles;
    // frames of less than 2^32 bits: the write pass of the position-parallel walk, one lane per group (k_seg_groups)
    const uint64_t worst_bits = (uint64_t)a.geom.n_blocks * (12u + (uint64_t)kBlock * max_w) + 8u;
    if (worst_bits < 0xF0000000ull) return launch_seg_groups(a, max_w, states, st);
    hipLaunchKernelGGL(k_walk_groups, dim3((uint32_t)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, a.terse,
                       (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w, states, n, a.widths, a.tile_off, a.status);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_decode_fast_t(const DecodeArgs& a, bool have_index, bool per_frame, hipStream_t st) {
    const FrameGeom g = a.geom;
    constexpr uint32_t tb = unpack_sub_tiles<T>() * kThreads;
    const uint32_t tpf = (g.n_blocks + tb - 1) / tb;
    const uint32_t max_w = PixelTraits<T>::bits;
    zero_status(a.status, st);
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    if (!have_index) {
        const hipError_t e = launch_walk_only(a, max_w, false, st);
        if (e != hipSuccess) return e;
    }
    prof.mark(st);
    if (have_index && per_frame) {                              // many small frames: the per-frame decoder with the widths given
        constexpr int dtype = PixelTraits<T>::bits == 8 ? (PixelTraits<T>::is_signed ? 1 : 0)
                              : PixelTraits<T>::bits == 16 ? (PixelTraits<T>::is_signed ? 3 : 2) : (PixelTraits<T>::is_signed ? 5 : 4);
        const hipError_t e = launch_decode_frames_indexed(dtype, a, nullptr, st);
        prof.mark(st);
        return e;
    }
#ifndef TRPX_INDEXED_LARGE_TILES
    if (have_index && sizeof(T) < 4 && g.n_blocks >= (1u << 18) &&
        8 * (uint64_t)g.n_blocks * (12u + 12u * max_w) < (1ull << 31)) {   // large frames of 8/16-bit pixels with their index: units of the per-frame decoder, as the index route extracts them (decode_frame.hip)
        constexpr int dtype = PixelTraits<T>::bits == 8 ? (PixelTraits<T>::is_signed ? 1 : 0) : (PixelTraits<T>::is_signed ? 3 : 2);
        const hipError_t e = launch_decode_units_indexed(dtype, a, st, nullptr);
        prof.mark(st);
        return e;
    }
#endif
    hipLaunchKernelGGL((k_unpack_tiles<T>), dim3((uint32_t)((uint64_t)a.n_frames * tpf)), dim3(kThreads), 0, st, a.terse,
                       (uint64_t)a.terse_bytes, a.frame_offsets, g, tpf, a.widths, a.tile_off,
                       static_cast<T*>(a.pixels_out), a.status, static_cast<const uint32_t*>(nullptr));
    prof.mark(st);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_unpack_tiles_t(const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode) {
    constexpr uint32_t tb = unpack_sub_tiles<T>() * kThreads;
    const uint32_t tpf = (a.geom.n_blocks + tb - 1) / tb;
    hipLaunchKernelGGL((k_unpack_tiles<T>), dim3((uint32_t)((uint64_t)a.n_frames * tpf)), dim3(kThreads), 0, st, a.terse,
                       (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, tpf, a.widths, a.tile_off, static_cast<T*>(a.pixels_out), a.status, frame_mode);
    return hipGetLastError();
}
hipError_t launch_unpack_tiles(int dtype, const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode) {
    switch (dtype) {
    case 0: return launch_unpack_tiles_t<uint8_t>(a, st, frame_mode);
    case 1: return launch_unpack_tiles_t<int8_t>(a, st, frame_mode);
    case 2: return launch_unpack_tiles_t<uint16_t>(a, st, frame_mode);
    case 3: return launch_unpack_tiles_t<int16_t>(a, st, frame_mode);
    case 4: return launch_unpack_tiles_t<uint32_t>(a, st, frame_mode);
    case 5: return launch_unpack_tiles_t<int32_t>(a, st, frame_mode);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_walk_lds_only(const DecodeArgs& a, uint32_t max_w, const uint32_t* only, hipStream_t st, const uint32_t* list = nullptr);

// Fast path preconditions (checked by the caller): block = 12, frame offsets known, frames of < 2^32 bits (T-aligned pointers, any pixel count).
hipError_t launch_walk_only(const DecodeArgs& a, uint32_t max_w, bool clear_status, hipStream_t st) {
    // the position-parallel walk; diagnostic builds: TRPX_WALK = lds keeps the one-wavefront-per-frame walk (A/B checks)
#ifdef TRPX_DIAGNOSTICS
    static const bool lds_walk = getenv("TRPX_WALK") && strcmp(getenv("TRPX_WALK"), "lds") == 0;
#else
    constexpr bool lds_walk = false;
#endif
    if (!lds_walk && a.seg_ws && a.chain && a.parts && a.defer) {
        // large frames: one walk of many short parts writes the index (every frame through k_chain_index: narrow = false); the
        // frames where that does not work out are listed and get theirs from the position-parallel walk
        hipError_t e = launch_chain_zero(a, max_w, clear_status, st);
        if (e != hipSuccess) return e;
        const uint32_t* mode = nullptr;
        e = launch_build_index_chain(a, max_w, false, &mode, st);
        if (e != hipSuccess) return e;
        return launch_seg_listed(a, max_w, st);
    }
    if (clear_status) zero_status(a.status, st);
    if (!lds_walk && a.seg_ws && a.index_per_frame && a.defer)
        return launch_index_frames(max_w, a, false, st);        // (status cleared above if asked)
    if (!lds_walk && a.seg_ws) return launch_seg_walk(a, max_w, st);
    return launch_walk_lds_only(a, max_w, nullptr, st);
}

hipError_t launch_walk_lds_only(const DecodeArgs& a, uint32_t max_w, const uint32_t* only, hipStream_t st, const uint32_t* list) {
    hipLaunchKernelGGL(k_walk_lds, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes,
                       a.frame_offsets, a.geom, max_w, a.widths, a.tile_off, only, list, a.status);
    return hipGetLastError();
}

hipError_t launch_decode_fast(int dtype, const DecodeArgs& a, bool have_index, hipStream_t st, bool per_frame) {
    switch (dtype) {
    case 0: return launch_decode_fast_t<uint8_t>(a, have_index, per_frame, st);
    case 1: return launch_decode_fast_t<int8_t>(a, have_index, per_frame, st);
    case 2: return launch_decode_fast_t<uint16_t>(a, have_index, per_frame, st);
    case 3: return launch_decode_fast_t<int16_t>(a, have_index, per_frame, st);
    case 4: return launch_decode_fast_t<uint32_t>(a, have_index, per_frame, st);
    case 5: return launch_decode_fast_t<int32_t>(a, have_index, per_frame, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace trpx
