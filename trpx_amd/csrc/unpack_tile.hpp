// k_unpack_tiles' tile body as a device function (decode_fast.hip: one tile per workgroup; decode_seg.hip: the
// workgroup of a deferred frame loops over the frame's tiles).  Replaces Bit_range::get_range / operator T()
// (reference include/Bit_pointer.hpp:742-792, :597-617) for blocks whose widths are known.
#pragma once
#include "codec_common.hpp"
#include "unpack_common.hpp"

namespace trpx {

template <typename T> constexpr int unpack_sub_tiles() { return sizeof(T) <= 2 ? 4 : 2; }
// Pixels leave through a per-wavefront LDS row as whole 128-byte lines (store_group) for every pixel type: lane-owned 8..24-byte
// runs leave each line to be merged from several store instructions in L2, and under load the lines are written back half
// merged (decode_frame.hip).  2000 x 512 x 512 u16 with the decode index: 0.327 ms direct, 0.291 ms staged (five instead of six
// workgroups per CU for the rows); eight 4096 x 4096 int32 frames: 0.29 -> 0.12 ms.  TRPX_UNPACK_DIRECT: A/B build.
#ifdef TRPX_UNPACK_DIRECT
template <typename T> constexpr bool unpack_staged() { return sizeof(T) == 4; }
#else
template <typename T> constexpr bool unpack_staged() { return true; }
#endif
// (rows at a stride of one group behind one head room: store_group_lines reads up to 127 bytes in front of a row and uses only
// what lies inside the row when no group continues another, as here -- the bytes in front may be the neighbour's)
template <typename T> constexpr int unpack_stage_row_dwords() { return kWave * kBlock * (int)sizeof(T) / 4; }
template <typename T> constexpr int unpack_stage_dwords() { return unpack_staged<T>() ? 4 * unpack_stage_row_dwords<T>() + kStageCarryDw : 4; }
template <typename T>
constexpr int unpack_image_dwords() { return unpack_sub_tiles<T>() * ((kThreads * max_block_bits<T>() + 31) / 32) + 12; }

// One tile (kSub * 256 blocks) of one frame: widths -> lengths -> scan -> stream bytes to LDS -> extraction -> stores.
// Called by all 256 threads of a workgroup; s_image / s_wtot / s_stage are the workgroup's LDS (unpack_image_dwords<T>(),
// 4 * unpack_sub_tiles<T>() and unpack_stage_dwords<T>() dwords; s_stage: one row of 64 blocks per wavefront, through which
// a wavefront's pixels leave as whole 16-byte-per-lane stores).  Returns false when the index does not fit the frame (status set).
template <typename T>
__device__ __forceinline__ bool unpack_tile(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                            const uint64_t* __restrict__ frame_offsets, const FrameGeom& g, uint32_t frame,
                                            uint32_t t, const uint8_t* __restrict__ widths,
                                            const uint64_t* __restrict__ tile_off, T* __restrict__ pixels_out,
                                            uint32_t* __restrict__ status, uint32_t* __restrict__ s_image,
                                            uint32_t* __restrict__ s_wtot, uint32_t* __restrict__ s_stage) {
    constexpr int kSub = unpack_sub_tiles<T>();
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wave = wave_id();
    const uint32_t b0 = t * kSub * kThreads;
    const uint8_t* __restrict__ wf = widths + (uint64_t)frame * g.n_blocks;

    uint32_t w[kSub], hl[kSub], len[kSub], inc[kSub];
    int nb[kSub];
#pragma unroll
    for (int r = 0; r < kSub; ++r) {
        const uint32_t b = b0 + r * kThreads + tid;
        nb[r] = 0; w[r] = 0; hl[r] = 0;
        if (b < g.n_blocks) {
            w[r] = wf[b];
            const uint32_t w_prev = b ? wf[b - 1] : 0u;     // significant_bits = 0 at frame start (Terse.hpp:359)
            const uint64_t first = (uint64_t)b * kBlock;
            nb[r] = first + kBlock <= g.n_values ? kBlock : (int)(g.n_values - first);
            hl[r] = header_len(w[r], w_prev);
        }
        len[r] = nb[r] ? hl[r] + (uint32_t)nb[r] * w[r] : 0u;
        inc[r] = wave_inclusive_scan(len[r]);
        if (lane == 63) s_wtot[r * 4 + wave] = inc[r];
    }
    __syncthreads();
    // every wave: exclusive scan of the 16 (round, wave) piece sizes
    uint32_t off[kSub];
    uint32_t tile_bits;
    {
        const uint32_t tot = lane < kSub * 4 ? s_wtot[lane] : 0u;
        const uint32_t incl = wave_inclusive_scan(tot);
        const uint32_t excl = incl - tot;
#pragma unroll
        for (int r = 0; r < kSub; ++r)
            off[r] = (uint32_t)__shfl((int)excl, r * 4 + wave, 64) + inc[r] - len[r];   // tile-relative bit of my block
        tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, kSub * 4 - 1);
    }
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    const uint64_t t_off = tile_off[(uint64_t)frame * g.n_tiles + (uint64_t)t * kSub];   // walk records every 256 blocks
    if (fe > terse_bytes || fe <= fo || t_off > 8 * (fe - fo) || tile_bits > 8 * (fe - fo) - t_off) {   // chain / index inconsistent with the frame
        if (tid == 0) atomicMax(&status[0], 5u);
        return false;
    }
    // ---- stage the tile's stream bits [a0, a0 + tile_bits) in LDS, 16 bytes per lane, coalesced ----------
    const uint64_t a0 = 8 * fo + t_off;
    const uint64_t d_lo = (a0 >> 5) & ~3ull;                 // 16-byte aligned start (terse is 4-byte aligned: use dwords)
    const uint32_t n_dw = (uint32_t)(((a0 + tile_bits + 31) >> 5) - d_lo) + 1;   // + 1: alignbit peeks one dword further
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t total_dw = (terse_bytes + 3) / 4;
    const bool base16 = ((uintptr_t)terse & 15) == 0;
    for (uint32_t i = tid * 4; i < n_dw; i += kThreads * 4) {
        const uint64_t d = d_lo + i;
        uint4 x;
        if (base16 && d + 4 <= total_dw) x = *reinterpret_cast<const uint4*>(s32 + d);
        else {
            x.x = d < total_dw ? s32[d] : 0u; x.y = d + 1 < total_dw ? s32[d + 1] : 0u;
            x.z = d + 2 < total_dw ? s32[d + 2] : 0u; x.w = d + 3 < total_dw ? s32[d + 3] : 0u;
        }
        *reinterpret_cast<uint4*>(&s_image[i]) = x;
    }
    __syncthreads();
    const uint32_t img_bit0 = (uint32_t)(a0 - 32 * d_lo);   // image bit of the tile's first bit (< 128)

    // ---- extract + store ----------------------------------------------------------------------------------
    T* __restrict__ fout = pixels_out + (uint64_t)frame * g.n_values;
#pragma unroll
    for (int r = 0; r < kSub; ++r) {
        const uint32_t b = b0 + r * kThreads + tid;
        const uint32_t q = img_bit0 + off[r] + hl[r];       // first payload bit in the image
        uint32_t u[kBlock];
#pragma unroll
        for (int k = 0; k < kBlock; ++k) u[k] = 0u;         // w == 0 -> zeros (Terse.hpp:373-374)
        uint64_t todo = __ballot(nb[r] == kBlock && w[r] != 0u);
        while (todo) {
            const int l0 = __builtin_ctzll(todo);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w[r], l0);
            const bool mine = nb[r] == kBlock && w[r] == w0;
            uint32_t wd = w0 > (uint32_t)PixelTraits<T>::bits ? (uint32_t)PixelTraits<T>::bits : w0;
            asm volatile("" : "+s"(wd));                    // (a copy the compiler cannot equate with the lanes' own width: the dispatch stays scalar)
            uint32_t qq = q;
            asm volatile("" : "+v"(qq));                    // keep the specialised bodies out of LICM's reach
            if (mine) UnpackDispatch<T, 1, PixelTraits<T>::bits>::run(s_image, qq, wd, u);
            todo &= ~__ballot(mine);
        }
        if (unpack_staged<T>() && __ballot(nb[r] == kBlock) == ~0ull) {   // the wavefront's 64 blocks are all full: staged, coalesced stores
            if (w[r] > (uint32_t)PixelTraits<T>::bits) atomicMax(&status[0], 5u);
            uint32_t* const st = s_stage + kStageCarryDw + wave * unpack_stage_row_dwords<T>();
            T* const gdst = fout + (uint64_t)(b - (uint32_t)lane) * kBlock;
            stage_block<T>(st + lane * (kBlock * (int)sizeof(T) / 4), u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            store_group<T, false>(st, gdst);                                  // (from the group's first pixel, wherever in a line it lies: store_group_lines)
            __builtin_amdgcn_wave_barrier();                // (the row is rewritten in the next round)
        } else if (nb[r] == kBlock) {
            if (w[r] > (uint32_t)PixelTraits<T>::bits) atomicMax(&status[0], 5u);
            store_block<T>(fout + (uint64_t)b * kBlock, u);
        } else if (nb[r]) {                                 // the frame's last, partial block: generic
            const uint32_t ww = w[r] > (uint32_t)PixelTraits<T>::bits ? 0u : w[r];
            const uint32_t mask = ww >= 32u ? 0xFFFFFFFFu : ((1u << ww) - 1u);
            uint32_t p = q;
            for (int k = 0; k < nb[r]; ++k) {
                uint32_t f = 0;
                if (ww) {
                    const uint64_t two = (uint64_t)s_image[p >> 5] | ((uint64_t)s_image[(p >> 5) + 1] << 32);
                    f = (uint32_t)(two >> (p & 31u)) & mask;
                    if (PixelTraits<T>::is_signed) f = (uint32_t)((int32_t)(f << (32u - ww)) >> (32u - ww));
                }
                fout[(uint64_t)b * kBlock + k] = (T)f;
                p += ww;
            }
        }
    }
    return true;
}


}  // namespace trpx
