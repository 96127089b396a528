// The serial header walk with the stream staged through LDS (see decode_fast.hip: k_walk_lds), as a device function so that the
// position-parallel walk's one-launch fallback (decode_seg.hip: k_seg_fallback) can run it as its last phase.
#pragma once
#include "codec_common.hpp"

namespace trpx {

constexpr int kWalkChunkDw = 4096;                    // 16 KB of stream per LDS refill

// The serial walk of ONE frame by one wavefront; s_chunk: kWalkChunkDw + 4 dwords of LDS.  (k_walk_lds, decode_fast.hip, and the
// last phase of k_seg_fallback, decode_seg.hip.)
__device__ __forceinline__ void walk_lds_frame(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                               const uint64_t* __restrict__ frame_offsets, const FrameGeom& g,
                                               uint32_t max_w, uint8_t* __restrict__ widths,
                                               uint64_t* __restrict__ tile_off, uint64_t frame, uint32_t* __restrict__ s_chunk,
                                               uint32_t* __restrict__ status) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    if (!(fe > fo && fe <= terse_bytes)) {
        if (lane == 0) atomicMax(&status[0], 5u);
        return;
    }
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const bool base16 = ((uintptr_t)terse & 15) == 0;
    const uint64_t frame_abit = 8 * fo, limit_bits = 8 * (fe - fo);
    uint8_t* __restrict__ wf = widths + frame * g.n_blocks;
    uint64_t* __restrict__ tf = tile_off + frame * g.n_tiles;
    const uint32_t nb_last = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock);

    // All positions are frame-relative bit offsets in 32 bits (the launcher routes frames of >= 2^32 bits
    // to the basic path); everything that steers the loop is wave-uniform and lives in SGPRs: the serial
    // chain per step is one LDS read, one ballot and a few scalar instructions.
    const uint32_t limit = (uint32_t)limit_bits;
    const uint64_t frame_dw = frame_abit >> 5;        // absolute dword of the frame's first bit
    const uint32_t frame_sh = (uint32_t)(frame_abit & 31);
    const uint32_t n_blocks = g.n_blocks;
    int32_t c_lo = 0, c_hi = 0;                       // frame-relative dword range [c_lo, c_hi) held in s_chunk
                                                      // (c_lo may be -1..-3: chunks start on absolute 16-byte boundaries)
    uint32_t b = 0, w_prev = 0, pos = 0, final_pos = 0;
    bool bad = false;
#ifdef TRPX_WALK_STATS
    uint64_t st_t0 = __builtin_amdgcn_s_memtime(), st_refill = 0; uint32_t st_steps = 0, st_refills = 0;
#endif
    while (b < n_blocks) {
        // ---- fast steps (see k_decode_frames): 64 real candidates, none of them the frame's last block, all inside the
        // LDS window; every lane decodes "its" explicit header in parallel, the step itself is branch-free ----------------
        {
            uint32_t stride = 1u + kBlock * w_prev;
            int32_t pos_max = 32 * (c_hi - 1) - (int32_t)frame_sh - 63 * (int32_t)stride;
            uint32_t wide = 0;
            while (b + 64u < n_blocks && (int32_t)pos < pos_max) {
#ifdef TRPX_WALK_STATS
                ++st_steps;
#endif
                const uint32_t fbit = frame_sh + pos + __umul24(lane, stride) - 32u * (uint32_t)c_lo;
                const uint32_t bits = __builtin_amdgcn_alignbit(s_chunk[(fbit >> 5) + 1], s_chunk[fbit >> 5], fbit);
                const uint64_t stop = ~__ballot((bits & 1u) != 0u);
                const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                const bool run = first >= 64u;
                uint32_t e_w = w_prev, adv = 0;
                if (!run) {                                                     // (wave-uniform branch)
                    const uint32_t w3 = (bits >> 1) & 7u, wa = 7u + ((bits >> 4) & 3u), wb = 10u + ((bits >> 6) & 63u);
                    const uint32_t wk = w3 != 7u ? w3 : (wa != 10u ? wa : wb);
                    const uint32_t advk = (w3 != 7u ? 4u : (wa != 10u ? 6u : 12u)) + kBlock * wk;
                    e_w = (uint32_t)__builtin_amdgcn_readlane((int)wk, (int)first);
                    adv = (uint32_t)__builtin_amdgcn_readlane((int)advk, (int)first);
                }
                wide = e_w > wide ? e_w : wide;
                const uint32_t n_done = run ? 64u : first + 1u;
                if (lane < n_done) {
                    wf[b + lane] = (uint8_t)(lane < first ? w_prev : e_w);
                    if (((b + lane) & (kTileBlocks - 1)) == 0) tf[(b + lane) / kTileBlocks] = pos + __umul24(lane, stride);
                }
                pos += first * stride + adv;
                b += n_done;
                w_prev = e_w;
                stride = 1u + kBlock * e_w;
                pos_max = 32 * (c_hi - 1) - (int32_t)frame_sh - 63 * (int32_t)stride;
            }
            if (wide > max_w) { bad = true; break; }
        }
#ifdef TRPX_WALK_STATS
        ++st_steps;
#endif
        const uint32_t stride = 1u + kBlock * w_prev;
        // dwords needed this step: candidates pos .. pos + 63*stride, each peeking 12 bits (2 dwords)
        const uint32_t need_lo = (frame_sh + pos) >> 5;
        const uint32_t need_hi = ((frame_sh + pos + 63u * stride) >> 5) + 2;
        if ((int32_t)need_lo < c_lo || (int32_t)need_hi > c_hi) {   // refill (wave-uniform), 16-byte coalesced
#ifdef TRPX_WALK_STATS
            const uint64_t rt0 = __builtin_amdgcn_s_memtime(); ++st_refills;
#endif
            c_lo = (int32_t)(((frame_dw + need_lo) & ~3ull) - frame_dw);
            c_hi = c_lo + kWalkChunkDw;
            const uint64_t d0 = (uint64_t)((int64_t)frame_dw + c_lo);
            if (base16 && (d0 & 3) == 0 && d0 + kWalkChunkDw <= n_dw) {
                // whole chunk in bounds and 16-byte aligned: all 16 loads in flight before the first LDS write
                constexpr int kIt = kWalkChunkDw / (kWave * 4);
                uint4 x[kIt];
#pragma unroll
                for (int it = 0; it < kIt; ++it) x[it] = *reinterpret_cast<const uint4*>(s32 + d0 + it * kWave * 4 + lane * 4);
#pragma unroll
                for (int it = 0; it < kIt; ++it) *reinterpret_cast<uint4*>(&s_chunk[it * kWave * 4 + lane * 4]) = x[it];
            } else {
                for (uint32_t i = lane * 4; i < (uint32_t)kWalkChunkDw; i += kWave * 4) {
                    const uint64_t d = d0 + i;
                    uint4 x;
                    x.x = d < n_dw ? s32[d] : 0u; x.y = d + 1 < n_dw ? s32[d + 1] : 0u;
                    x.z = d + 2 < n_dw ? s32[d + 2] : 0u; x.w = d + 3 < n_dw ? s32[d + 3] : 0u;
                    *reinterpret_cast<uint4*>(&s_chunk[i]) = x;
                }
            }
#ifdef TRPX_WALK_STATS
            __builtin_amdgcn_s_waitcnt(0); st_refill += __builtin_amdgcn_s_memtime() - rt0;
#endif
            // (no explicit wait: one wave, LDS operations execute in order; an s_waitcnt here would also wait for the
            //  previous steps' width stores -- CDNA4 counts stores in vmcnt -- and serialise every step on HBM)
        }
        // every lane peeks at its candidate (bits past the frame's end read as whatever follows: harmless, the
        // chain is validated against S_f at the end and b never passes n_blocks)
        const uint32_t fbit = frame_sh + pos + lane * stride - 32u * (uint32_t)c_lo;   // bit index inside s_chunk
        const uint32_t bits = __builtin_amdgcn_alignbit(s_chunk[(fbit >> 5) + 1], s_chunk[fbit >> 5], fbit);
        const uint32_t left = n_blocks - b;                                // candidates that are real blocks
        const uint64_t valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
        const uint64_t same = __ballot((bits & 1u) != 0u) & valid;         // Terse.hpp:361
        const uint64_t stop = ~same;                                       // first explicit header or end of frame
        const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;

        uint32_t e_w = w_prev;
        const bool explicit_hdr = first < left && first < 64u;             // block b + first has an explicit header
        uint32_t new_pos, new_b;
        if (explicit_hdr) {
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)bits, first);   // scalar parse (Terse.hpp:362-370)
            uint32_t w = (eb >> 1) & 7u, hl = 4;
            if (w == 7u) {
                w += (eb >> 4) & 3u; hl = 6;
                if (w == 10u) { w += (eb >> 6) & 63u; hl = 12; }
            }
            if (w > max_w) { bad = true; break; }
            e_w = w;
            const uint32_t nbv = b + first + 1 == n_blocks ? nb_last : (uint32_t)kBlock;
            new_pos = pos + first * stride + hl + nbv * w;
            new_b = b + first + 1;
            if (new_b == n_blocks) final_pos = new_pos;
        } else {                                                           // every remaining candidate repeats w_prev
            const uint32_t cnt = left < 64u ? left : 64u;
            if (cnt == left) final_pos = pos + (cnt - 1) * stride + 1u + nb_last * w_prev;   // last block may be partial
            new_pos = pos + cnt * stride;
            new_b = b + cnt;
        }
        // widths of the blocks consumed by this step: w_prev for the run, e_w for the explicit block
        const uint32_t n_done = new_b - b;
        if (lane < n_done) wf[b + lane] = (uint8_t)(lane < first ? w_prev : e_w);
        // bit offset of every 256-block group that starts inside this step
        if (((b + n_done - 1) ^ (b - 1)) >= (uint32_t)kTileBlocks || b == 0) {
            const uint32_t cb = b + lane;
            if (lane < n_done && (cb & (kTileBlocks - 1)) == 0) tf[cb / kTileBlocks] = pos + lane * stride;
        }
        pos = new_pos;
        w_prev = e_w;
        b = new_b;
        if (pos > limit + 64u * 400u) { bad = true; break; }               // ran away (corrupt stream): stop before wrapping
    }
    const bool ok = !bad && final_pos <= limit && 1 + (uint64_t)final_pos / 8 == fe - fo;   // S_f (Terse.hpp:547)
    if (!ok && lane == 0) atomicMax(&status[0], 5u);                     // TRPX_ERR_CORRUPT
#ifdef TRPX_WALK_STATS
    if (lane == 0) { atomicAdd(&status[2], st_steps); atomicAdd(&status[3], st_refills); atomicAdd(&status[4], (uint32_t)(st_refill >> 4)); atomicAdd(&status[5], (uint32_t)((__builtin_amdgcn_s_memtime() - st_t0) >> 4)); }
#endif
}

}  // namespace trpx
