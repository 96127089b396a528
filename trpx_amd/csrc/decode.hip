// PROLIX decode kernels for gfx950 (CDNA4).  Replaces jpa::Terse::prolix(Iterator, frame)
// (reference include/Terse.hpp:352-389), f_find_terse_frame (:562-585, intended semantics) and
// the Bit_pointer.hpp unpack primitives (Bit_range::get_range :742-792, operator T() :597-617).
//
// The .trpx stream stores no index: block b+1's bit position is only known after block b's
// header has been parsed (Terse.hpp:360-372).  v1 pipeline:
//   k_walk / k_walk_serial   one wavefront per frame walks the header chain.  All 64 lanes test
//                            the "same width" bit (Terse.hpp:361) of the next 64 candidate
//                            blocks at stride 1+12w in parallel, so a run of equal-width blocks
//                            costs one step; emits width[b] (u8) and the bit offset of every
//                            256-block tile.
//   k_unpack                 fully parallel: widths -> header/payload lengths -> workgroup scan
//                            -> every lane extracts its 12 fields and sign/zero-extends them.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include <type_traits>

namespace trpx {

// 64 bits of the stream starting at absolute bit `abit` of the (4-byte aligned) buffer.
// Dwords with no valid byte read as zero (an aligned dword holding >= 1 valid byte never
// crosses into an unmapped page).
__device__ __forceinline__ uint32_t ld_stream_dw(const uint32_t* __restrict__ s32, uint64_t idx, uint64_t n_dw) {
    return idx < n_dw ? s32[idx] : 0u;
}
__device__ __forceinline__ uint32_t peek32(const uint32_t* __restrict__ s32, uint64_t n_dw, uint64_t abit) {
    const uint64_t di = abit >> 5;
    const uint32_t sh = (uint32_t)(abit & 31);
    const uint64_t x = (uint64_t)ld_stream_dw(s32, di, n_dw) | ((uint64_t)ld_stream_dw(s32, di + 1, n_dw) << 32);
    return (uint32_t)(x >> sh);
}

// Walk one frame with one wavefront.  Returns the frame's total bit count, or ~0ull if the
// chain runs past `limit_bits` or a width exceeds `max_w` (corrupt stream).
__device__ uint64_t walk_frame(const uint32_t* __restrict__ s32, uint64_t n_dw, uint64_t frame_abit,
                               uint64_t limit_bits, const FrameGeom g, uint32_t max_w,
                               uint8_t* __restrict__ widths_f, uint64_t* __restrict__ tile_off_f) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t nb_last = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * g.block);
    uint32_t b = 0, w_prev = 0;
    uint64_t pos = 0;
    uint64_t final_pos = 0;
    bool bad = false;
    while (b < g.n_blocks) {
        const uint32_t stride = 1u + g.block * w_prev;
        const uint32_t cb = b + lane;
        const uint64_t cpos = pos + (uint64_t)lane * stride;
        const bool in_range = cb < g.n_blocks;
        const bool readable = in_range && cpos < limit_bits;
        const uint32_t bits = readable ? peek32(s32, n_dw, frame_abit + cpos) : 0u;
        const bool same = readable && (bits & 1u);                       // Terse.hpp:361
        const uint64_t not_same = __ballot(!same);
        const uint32_t first = not_same ? (uint32_t)__builtin_ctzll(not_same) : 64u;

        if (lane < first) widths_f[cb] = (uint8_t)w_prev;
        if (lane <= first && in_range && (cb & (kTileBlocks - 1)) == 0) tile_off_f[cb / kTileBlocks] = cpos;

        uint64_t npos = cpos;
        uint32_t nw = w_prev;
        bool lane_bad = false;
        if (lane == first && in_range) {                                 // explicit header
            if (!readable) lane_bad = true;
            uint32_t w = (bits >> 1) & 7u, hl = 4;                       // Terse.hpp:362
            if (w == 7u) {
                w += (bits >> 4) & 3u; hl = 6;                           // :365
                if (w == 10u) { w += (bits >> 6) & 63u; hl = 12; }       // :368
            }
            if (w > max_w) { lane_bad = true; w = 0; }
            const uint32_t nbv = cb + 1 == g.n_blocks ? nb_last : g.block;
            npos = cpos + hl + (uint64_t)nbv * w;
            nw = w;
            widths_f[cb] = (uint8_t)w;
        }
        // position after the frame's last block (a "same" last block may be partial)
        uint64_t fin = 0;
        if (in_range && cb + 1 == g.n_blocks) {
            if (lane < first) fin = cpos + 1 + (uint64_t)nb_last * w_prev;
            else if (lane == first) fin = npos;
        }
        const uint64_t fin_mask = __ballot(fin != 0 && lane <= first);
        if (fin_mask) {
            const int src = __builtin_ctzll(fin_mask);
            final_pos = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(fin >> 32), src, 64) << 32) |
                        (uint32_t)__shfl((int)(uint32_t)fin, src, 64);
        }
        if (__ballot(lane_bad)) { bad = true; break; }
        if (first < 64u) {
            const int src = (int)first;
            pos = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(npos >> 32), src, 64) << 32) |
                  (uint32_t)__shfl((int)(uint32_t)npos, src, 64);
            w_prev = (uint32_t)__shfl((int)nw, src, 64);
            b += first + 1;
        } else {
            pos += 64ull * stride;
            b += 64;
        }
    }
    if (bad || final_pos > limit_bits || final_pos / 8 + 1 > limit_bits / 8) return ~0ull;
    return final_pos;
}

// One wavefront per frame; frame byte ranges are known (frame_offsets given by the caller).
__global__ __launch_bounds__(kWave) void k_walk(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                uint32_t max_w, uint8_t* __restrict__ widths,
                                                uint64_t* __restrict__ tile_off, uint32_t* __restrict__ status) {
    const uint64_t frame = blockIdx.x;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    bool ok = fe > fo && fe <= terse_bytes;
    uint64_t bits = 0;
    if (ok) {
        bits = walk_frame(reinterpret_cast<const uint32_t*>(terse), (terse_bytes + 3) / 4, 8 * fo, 8 * (fe - fo), g,
                          max_w, widths + frame * g.n_blocks, tile_off + frame * g.n_tiles);
        ok = bits != ~0ull && 1 + bits / 8 == fe - fo;      // S_f = 1 + bits/8 (Terse.hpp:547)
    }
    if (!ok && lane_id() == 0) atomicMax(&status[0], 5u);   // TRPX_ERR_CORRUPT
}

// No frame index available: frames are located one after the other (Terse.hpp:562-585).
__global__ __launch_bounds__(kWave) void k_walk_serial(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       uint32_t n_frames, FrameGeom g, uint32_t max_w,
                                                       uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                       uint64_t* __restrict__ walk_offsets,
                                                       uint32_t* __restrict__ status) {
    uint64_t fo = 0;
    bool ok = true;
    for (uint32_t f = 0; f < n_frames; ++f) {
        if (lane_id() == 0) walk_offsets[f] = fo;
        uint64_t bits = ~0ull;
        if (ok && fo < terse_bytes)
            bits = walk_frame(reinterpret_cast<const uint32_t*>(terse), (terse_bytes + 3) / 4, 8 * fo,
                              8 * (terse_bytes - fo), g, max_w, widths + (uint64_t)f * g.n_blocks,
                              tile_off + (uint64_t)f * g.n_tiles);
        if (bits == ~0ull) { ok = false; bits = 0; }
        fo += ok ? 1 + bits / 8 : 0;
    }
    if (lane_id() == 0) {
        walk_offsets[n_frames] = fo;
        if (!ok) atomicMax(&status[0], 5u);
    }
}

template <typename T> struct alignas(4 * sizeof(T)) QuadOut { T x[4]; };

template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_unpack(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                     const uint8_t* __restrict__ widths,
                                                     const uint64_t* __restrict__ tile_off,
                                                     T* __restrict__ pixels_out, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_tot[4];
    if (status[0] != 0) return;                             // corrupt chain: produce nothing
    const uint32_t tid = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles);
    const uint32_t t = (uint32_t)(tile % g.n_tiles);
    const uint32_t b = t * kTileBlocks + tid;
    const bool valid = b < g.n_blocks;
    const uint8_t* wf = widths + (uint64_t)frame * g.n_blocks;

    uint32_t w = 0, w_prev = 0;
    int nb = 0;
    if (valid) {
        w = wf[b];
        w_prev = b ? wf[b - 1] : 0u;                        // significant_bits = 0 at frame start (:359)
        const uint64_t first = (uint64_t)b * kBlock;
        nb = first + kBlock <= g.n_values ? kBlock : (int)(g.n_values - first);
    }
    const uint32_t hl = header_len(w, w_prev);
    const uint32_t len = valid ? hl + (uint32_t)nb * w : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(len, s_tot, &total);
    if (!valid) return;

    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    const uint64_t pos = tile_off[tile] + excl + hl;        // first payload bit, relative to the frame
    // one 32-bit register per value: with T vals[] the compiler packs four 8-bit values per register and updates single
    // bytes under the refill / tail predicates -- that build produced run-to-run different pixels for 8-bit types
    // (tests/test_gpu_parity.py::test_encoder_chains_across_many_tiles_and_frames found it)
    uint32_t vals[kBlock];
#pragma unroll
    for (int k = 0; k < kBlock; ++k) vals[k] = 0u;           // w == 0 -> zeros (Terse.hpp:373-374)

    if (w) {
        if (pos + (uint64_t)nb * w > 8 * (fe - fo) || w > (uint32_t)PixelTraits<T>::bits) {
            atomicMax(&status[0], 5u);
        } else {
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(terse);
            const uint64_t n_dw = (terse_bytes + 3) / 4;
            const uint64_t abit = 8 * fo + pos;
            uint64_t di = abit >> 5;
            const uint32_t sh = (uint32_t)(abit & 31);
            uint64_t acc = ((uint64_t)ld_stream_dw(s32, di, n_dw) | ((uint64_t)ld_stream_dw(s32, di + 1, n_dw) << 32)) >> sh;
            uint32_t avail = 64u - sh;
            di += 2;
            const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
#pragma unroll
            for (int k = 0; k < kBlock; ++k) {
                if (k < nb) {
                    if (avail < w) {                        // refill (Bit_pointer.hpp:771-783)
                        acc |= (uint64_t)ld_stream_dw(s32, di++, n_dw) << avail;
                        avail += 32;
                    }
                    uint32_t u = (uint32_t)acc & mask;
                    acc >>= w;
                    avail -= w;
                    if (PixelTraits<T>::is_signed)          // sign-extend from bit w-1 (:784-789)
                        u = (uint32_t)((int32_t)(u << (32u - w)) >> (32u - w));
                    vals[k] = u;
                }
            }
        }
    }

    T* dst = pixels_out + (uint64_t)frame * g.n_values + (uint64_t)b * kBlock;
    if (nb == kBlock) {
        if (VEC) {
            QuadOut<T>* q = reinterpret_cast<QuadOut<T>*>(dst);
            QuadOut<T> a, c, d;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.x[k] = (T)vals[k]; c.x[k] = (T)vals[4 + k]; d.x[k] = (T)vals[8 + k]; }
            q[0] = a; q[1] = c; q[2] = d;
        } else {
#pragma unroll
            for (int k = 0; k < kBlock; ++k) dst[k] = (T)vals[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < kBlock; ++k)
            if (k < nb) dst[k] = (T)vals[k];
    }
}

// Any block size (see encode.hip's generic kernels): one lane per block, values read / written one by one.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_unpack_g(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                       const uint8_t* __restrict__ widths,
                                                       const uint64_t* __restrict__ tile_off,
                                                       T* __restrict__ pixels_out, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_tot[4];
    if (status[0] != 0) return;
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles), t = (uint32_t)(tile % g.n_tiles);
    const uint32_t b = t * kTileBlocks + threadIdx.x;
    const bool valid = b < g.n_blocks;
    const uint8_t* wf = widths + (uint64_t)frame * g.n_blocks;
    uint32_t w = 0, w_prev = 0, nb = 0;
    if (valid) {
        w = wf[b];
        w_prev = b ? wf[b - 1] : 0u;
        const uint64_t first = (uint64_t)b * g.block;
        nb = (uint32_t)(first + g.block <= g.n_values ? g.block : g.n_values - first);
    }
    const uint32_t hl = header_len(w, w_prev);
    const uint32_t len = valid ? hl + nb * w : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(len, s_tot, &total);
    if (!valid) return;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    const uint64_t pos = tile_off[tile] + excl + hl;
    T* dst = pixels_out + (uint64_t)frame * g.n_values + (uint64_t)b * g.block;
    if (w == 0) { for (uint32_t k = 0; k < nb; ++k) dst[k] = (T)0; return; }
    if (pos + (uint64_t)nb * w > 8 * (fe - fo) || w > (uint32_t)PixelTraits<T>::bits) { atomicMax(&status[0], 5u); return; }
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
    uint64_t abit = 8 * fo + pos;
    for (uint32_t k = 0; k < nb; ++k, abit += w) {
        const uint64_t two = (uint64_t)ld_stream_dw(s32, abit >> 5, n_dw) | ((uint64_t)ld_stream_dw(s32, (abit >> 5) + 1, n_dw) << 32);
        uint32_t u = (uint32_t)(two >> (abit & 31)) & mask;
        if (PixelTraits<T>::is_signed) u = (uint32_t)((int32_t)(u << (32u - w)) >> (32u - w));
        dst[k] = (T)u;
    }
}

template <typename T>
static hipError_t launch_decode_t(const DecodeArgs& a, bool have_offsets, hipStream_t st) {
    const FrameGeom g = a.geom;
    const uint64_t n_tiles_total = (uint64_t)a.n_frames * g.n_tiles;
    const bool vec = (g.n_values % 4 == 0) && ((uintptr_t)a.pixels_out % 16 == 0);
    const uint32_t max_w = PixelTraits<T>::bits;
    zero_status(a.status, st);
    const uint64_t* offs = a.frame_offsets;
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    if (have_offsets) {
        hipLaunchKernelGGL(k_walk, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, offs, g,
                           max_w, a.widths, a.tile_off, a.status);
    } else {
        hipLaunchKernelGGL(k_walk_serial, dim3(1), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.n_frames,
                           g, max_w, a.widths, a.tile_off, a.walk_offsets, a.status);
        offs = a.walk_offsets;
    }
    prof.mark(st);
    T* out = static_cast<T*>(a.pixels_out);
    if (g.block != (uint32_t)kBlock)
        hipLaunchKernelGGL((k_unpack_g<T>), dim3((uint32_t)n_tiles_total), dim3(kThreads), 0, st, a.terse,
                           (uint64_t)a.terse_bytes, offs, g, a.widths, a.tile_off, out, a.status);
    else if (vec) hipLaunchKernelGGL((k_unpack<T, true>), dim3((uint32_t)n_tiles_total), dim3(kThreads), 0, st, a.terse,
                                (uint64_t)a.terse_bytes, offs, g, a.widths, a.tile_off, out, a.status);
    else     hipLaunchKernelGGL((k_unpack<T, false>), dim3((uint32_t)n_tiles_total), dim3(kThreads), 0, st, a.terse,
                                (uint64_t)a.terse_bytes, offs, g, a.widths, a.tile_off, out, a.status);
    prof.mark(st);
    return hipGetLastError();
}

// Serial frame location only (no pixels): fills a.walk_offsets[0..n_frames].
hipError_t launch_walk_serial(const DecodeArgs& a, uint32_t max_w, hipStream_t st) {
    zero_status(a.status, st);
    hipLaunchKernelGGL(k_walk_serial, dim3(1), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.n_frames,
                       a.geom, max_w, a.widths, a.tile_off, a.walk_offsets, a.status);
    return hipGetLastError();
}

hipError_t launch_decode(int dtype, const DecodeArgs& a, bool have_offsets, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_decode_t<uint8_t>(a, have_offsets, st);
    case 1: return launch_decode_t<int8_t>(a, have_offsets, st);
    case 2: return launch_decode_t<uint16_t>(a, have_offsets, st);
    case 3: return launch_decode_t<int16_t>(a, have_offsets, st);
    case 4: return launch_decode_t<uint32_t>(a, have_offsets, st);
    case 5: return launch_decode_t<int32_t>(a, have_offsets, st);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// Converting decode: any output type, the stream's signedness given separately.  Values are zero- (unsigned
// stream) or sign-extended (signed stream) from the block width, then stored with clamping into a narrower
// integral type (Bit_pointer.hpp:747-763) or exactly into float / double (Terse.hpp:379-383, Bit_range::next
// :580-587).  Correct value semantics also for an unsigned stream into a signed type (reference defect D4).
// ---------------------------------------------------------------------------------------------
// `u`: the field zero-extended to 64 bits; `v`: the same sign-extended from its width (signed streams).
template <typename OutT>
__device__ __forceinline__ OutT convert_clamped(uint64_t u, int64_t v, bool stream_signed) {
    if constexpr (std::is_floating_point<OutT>::value) return stream_signed ? (OutT)v : (OutT)u;
    else if constexpr (sizeof(OutT) == 8) {
        if constexpr (std::is_signed<OutT>::value)                       // int64 out: an unsigned value >= 2^63 clamps
            return stream_signed ? (OutT)v : (u > (uint64_t)INT64_MAX ? (OutT)INT64_MAX : (OutT)u);
        else                                                             // uint64 out: negative values clamp to 0
            return stream_signed ? (v < 0 ? (OutT)0 : (OutT)v) : (OutT)u;
    } else {
        constexpr int64_t lo = std::is_signed<OutT>::value ? -(int64_t(1) << (8 * sizeof(OutT) - 1)) : 0;
        constexpr int64_t hi = std::is_signed<OutT>::value ? (int64_t(1) << (8 * sizeof(OutT) - 1)) - 1
                                                           : (int64_t(1) << (8 * sizeof(OutT))) - 1;
        if (!stream_signed) return (OutT)(u > (uint64_t)hi ? hi : (int64_t)u);
        return (OutT)(v < lo ? lo : (v > hi ? hi : v));
    }
}

template <typename OutT>
__global__ __launch_bounds__(kThreads) void k_unpack_conv(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                          const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                          int stream_signed, const uint8_t* __restrict__ widths,
                                                          const uint64_t* __restrict__ tile_off,
                                                          OutT* __restrict__ pixels_out, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_tot[4];
    if (status[0] != 0) return;
    const uint64_t tile = blockIdx.x;
    const uint32_t frame = (uint32_t)(tile / g.n_tiles), t = (uint32_t)(tile % g.n_tiles);
    const uint32_t b = t * kTileBlocks + threadIdx.x;
    const bool valid = b < g.n_blocks;
    const uint8_t* wf = widths + (uint64_t)frame * g.n_blocks;
    uint32_t w = 0, w_prev = 0, nb = 0;
    if (valid) {
        w = wf[b];
        w_prev = b ? wf[b - 1] : 0u;
        const uint64_t first = (uint64_t)b * g.block;
        nb = (uint32_t)(first + g.block <= g.n_values ? g.block : g.n_values - first);
    }
    const uint32_t hl = header_len(w, w_prev);
    const uint32_t len = valid ? hl + nb * w : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(len, s_tot, &total);
    if (!valid) return;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    const uint64_t pos = tile_off[tile] + excl + hl;
    OutT* dst = pixels_out + (uint64_t)frame * g.n_values + (uint64_t)b * g.block;
    if (w == 0) { for (uint32_t k = 0; k < nb; ++k) dst[k] = (OutT)0; return; }
    if (pos + (uint64_t)nb * w > 8 * (fe - fo) || w > 64u) { atomicMax(&status[0], 5u); return; }
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const uint64_t mask = w >= 64u ? ~0ull : ((1ull << w) - 1ull);
    uint64_t abit = 8 * fo + pos;
    for (uint32_t k = 0; k < nb; ++k, abit += w) {                       // fields of up to 64 bits: three dwords
        const uint64_t di = abit >> 5;
        const uint32_t sh = (uint32_t)(abit & 31);
        const uint64_t two = (uint64_t)ld_stream_dw(s32, di, n_dw) | ((uint64_t)ld_stream_dw(s32, di + 1, n_dw) << 32);
        uint64_t u = two >> sh;
        if (sh && w > 64u - sh) u |= (uint64_t)ld_stream_dw(s32, di + 2, n_dw) << (64u - sh);
        u &= mask;
        const int64_t v = w >= 64u ? (int64_t)u : (int64_t)(u << (64u - w)) >> (64u - w);   // sign extension (Bit_pointer.hpp:784-789)
        dst[k] = convert_clamped<OutT>(u, v, stream_signed != 0);
    }
}

template <typename OutT>
static hipError_t launch_decode_convert_t(const DecodeArgs& a, int stream_signed, bool have_offsets, hipStream_t st) {
    const FrameGeom g = a.geom;
    zero_status(a.status, st);
    const uint64_t* offs = a.frame_offsets;
    if (have_offsets) {
        hipLaunchKernelGGL(k_walk, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, offs, g, 64u,
                           a.widths, a.tile_off, a.status);
    } else {
        hipLaunchKernelGGL(k_walk_serial, dim3(1), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.n_frames, g, 64u,
                           a.widths, a.tile_off, a.walk_offsets, a.status);
        offs = a.walk_offsets;
    }
    hipLaunchKernelGGL((k_unpack_conv<OutT>), dim3((uint32_t)((uint64_t)a.n_frames * g.n_tiles)), dim3(kThreads), 0, st,
                       a.terse, (uint64_t)a.terse_bytes, offs, g, stream_signed, a.widths, a.tile_off,
                       static_cast<OutT*>(a.pixels_out), a.status);
    return hipGetLastError();
}

// dtype: 0..5 = the 8/16/32-bit integral pixel types, 6 = float, 7 = double, 8 / 9 = uint64 / int64
hipError_t launch_decode_convert(int dtype, const DecodeArgs& a, int stream_signed, bool have_offsets, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_decode_convert_t<uint8_t>(a, stream_signed, have_offsets, st);
    case 1: return launch_decode_convert_t<int8_t>(a, stream_signed, have_offsets, st);
    case 2: return launch_decode_convert_t<uint16_t>(a, stream_signed, have_offsets, st);
    case 3: return launch_decode_convert_t<int16_t>(a, stream_signed, have_offsets, st);
    case 4: return launch_decode_convert_t<uint32_t>(a, stream_signed, have_offsets, st);
    case 5: return launch_decode_convert_t<int32_t>(a, stream_signed, have_offsets, st);
    case 6: return launch_decode_convert_t<float>(a, stream_signed, have_offsets, st);
    case 7: return launch_decode_convert_t<double>(a, stream_signed, have_offsets, st);
    case 8: return launch_decode_convert_t<uint64_t>(a, stream_signed, have_offsets, st);
    case 9: return launch_decode_convert_t<int64_t>(a, stream_signed, have_offsets, st);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// synth-v1 generator (SURVEY.md section 8 row d) -- bench/test utility.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_synth(uint64_t seed, uint64_t frame0, uint64_t n_values,
                                                    uint64_t total, T* __restrict__ out) {
    for (uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * kThreads) {
        const uint64_t f = frame0 + idx / n_values, i = idx % n_values;
        const uint64_t r = synth_mix(seed + 0x9E3779B97F4A7C15ull * (f * n_values + i + 1));
        if (sizeof(T) == 2)
            out[idx] = ((r >> 40) & 0xFFF) == 0 ? (T)((r >> 24) & 0xFFF) : (T)__popcll(r & 0x3F);
        else
            out[idx] = ((r >> 40) & 0x3FF) == 0 ? (T)((r >> 8) & 0xFFFFFF) : (T)((int)__popcll(r & 0x3F) - 3);
    }
}

hipError_t launch_synth(int dtype, uint64_t seed, uint64_t frame0, size_t n_frames, size_t n_values, void* out,
                        hipStream_t st) {
    const uint64_t total = (uint64_t)n_frames * n_values;
    const uint32_t grid = (uint32_t)((total + kThreads - 1) / kThreads < 65536 ? (total + kThreads - 1) / kThreads : 65536);
    if (total == 0) return hipSuccess;
    if (dtype == 2) hipLaunchKernelGGL(k_synth<uint16_t>, dim3(grid), dim3(kThreads), 0, st, seed, frame0,
                                       (uint64_t)n_values, total, static_cast<uint16_t*>(out));
    else if (dtype == 5) hipLaunchKernelGGL(k_synth<int32_t>, dim3(grid), dim3(kThreads), 0, st, seed, frame0,
                                            (uint64_t)n_values, total, static_cast<int32_t*>(out));
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace trpx
