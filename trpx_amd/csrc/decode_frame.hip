// PROLIX decode, one workgroup per frame (gfx950 / CDNA4): the serial header walk and the parallel
// field extraction run side by side inside the workgroup, coupled through LDS only.
// Replaces jpa::Terse::prolix(Iterator, frame) (reference include/Terse.hpp:352-389) and
// Bit_range::get_range / operator T() (Bit_pointer.hpp:742-792, :597-617).
//
//   wave 0 ("walker")    walks the frame's header chain (Terse.hpp:360-372) exactly like k_walk_lds --
//                        64 candidate blocks per step, stream staged through a private LDS window --
//                        but deposits width[b] and the 64-block group offsets in LDS instead of HBM.
//   waves 1-3            one super-step (768 blocks = 12 groups of 64) behind the walker: each wave takes four
//                        groups; every lane loads the stream dwords of its own block straight from L2 (the
//                        walker has just pulled them through), extracts the 12 fields with width-specialised
//                        code on registers and stores the pixels (24/48 bytes per lane, non-temporal).
//   one barrier per super-step; width buffers are double buffered.
//
// The walk is the critical path (serial by construction of the format); the extraction hides under it.
// HBM traffic per frame: S read (once from HBM, once more from L2) + N*sizeof(T) written = algorithmic.
// Best for many small frames (one workgroup each); few huge frames are better served by the tiled
// kernels of decode_fast.hip with a decode index.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include "unpack_common.hpp"

namespace trpx {

#ifndef TRPX_FRAME_WAVES
#define TRPX_FRAME_WAVES 4
#endif
#ifndef TRPX_FRAME_GPW
#define TRPX_FRAME_GPW 4
#endif
constexpr int kFrameWaves = TRPX_FRAME_WAVES;           // waves per workgroup: 1 walker + (kFrameWaves - 1) unpackers
constexpr int kFrameThreads = kFrameWaves * kWave;
constexpr int kGroupsPerWave = TRPX_FRAME_GPW;          // 64-block groups per unpack wave and super-step
constexpr int kStepGroups = (kFrameWaves - 1) * kGroupsPerWave;   // 64-block groups per super-step
constexpr int kStepBlocks = kStepGroups * kWave;       // 768
#ifndef TRPX_FRAME_CHUNK_DW
#define TRPX_FRAME_CHUNK_DW 2048
#endif
constexpr int kFrameChunkDw = TRPX_FRAME_CHUNK_DW;     // walker's stream window: 8 KB.  Stream cache-resident (decode after decode): 2 / 4 / 8 /
                                                       // 16 KB -> 0.31 / 0.32 / 0.30 / 0.30 ms; cold (decode after an encode, the bench's round
                                                       // trip): 0.40 / 0.37 / 0.35 / 0.40 ms -- the unpack waves re-read the window's lines from
                                                       // L2, and 250 workgroups per XCD x 16 KB is all of its 4 MB

template <typename T>
__global__ __launch_bounds__(kFrameThreads, 2048 / kFrameThreads) void k_decode_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               T* __restrict__ pixels_out, uint32_t* __restrict__ defer,
                                                               uint32_t* __restrict__ status) {
    constexpr uint32_t kMaxW = PixelTraits<T>::bits;
    __shared__ uint32_t s_chunk[kFrameChunkDw + 4];    // walker's window of the stream
    __shared__ uint8_t s_w[2][kStepBlocks + 68];       // [0] = width of the block before the super-step, [1 + i] = widths of its blocks (double buffered); 64 spare bytes behind (fast steps)
    __shared__ uint32_t s_goff[3][kStepGroups < kWave ? kWave : kStepGroups];   // [0..1]: frame-relative bit offset of each group's first block; [2]: spare row (fast steps)
    uint8_t* const s_wb = &s_w[0][0];
    uint32_t* const s_goffx = &s_goff[0][0];
    constexpr uint32_t kGoffRow = kStepGroups < kWave ? kWave : kStepGroups;
    constexpr bool kStaged = sizeof(T) == 4;           // (8/16-bit pixels: direct stores are as fast -- int8 stacks 0.24 direct / 0.26 ms staged)
    __shared__ __attribute__((aligned(16))) uint32_t s_out[kFrameWaves - 1][kStaged ? kWave * kBlock * sizeof(T) / 4 : 4];   // a group's pixels, per extraction wave
    __shared__ uint32_t s_err;

    const uint32_t lane = (uint32_t)lane_id();
    const int wave = wave_id();
    const uint64_t frame = blockIdx.x;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    if (threadIdx.x == 0) s_err = (fe > fo && fe <= terse_bytes) ? 0u : 1u;
    __syncthreads();
    if (s_err) {
        if (threadIdx.x == 0) atomicMax(&status[0], 5u);
        return;
    }
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const bool base16 = ((uintptr_t)terse & 15) == 0;
    const uint64_t frame_abit = 8 * fo;
    const uint32_t limit = (uint32_t)(8 * (fe - fo));
    const uint64_t frame_dw = frame_abit >> 5;
    const uint32_t frame_sh = (uint32_t)(frame_abit & 31);
    const uint32_t n_blocks = g.n_blocks;
    const uint32_t nb_last = (uint32_t)(g.n_values - (uint64_t)(n_blocks - 1) * kBlock);
    const uint32_t n_steps = (n_blocks + kStepBlocks - 1) / kStepBlocks;
    T* __restrict__ fout = pixels_out + frame * g.n_values;

    // walker state (wave 0 only; wave-uniform)
    int32_t c_lo = 0, c_hi = 0;
    uint32_t b = 0, w_prev = 0, pos = 0, final_pos = 0;
    if (wave == 0) __builtin_amdgcn_s_setprio(3);      // the walk is the critical path

    for (uint32_t s = 0; s <= n_steps; ++s) {
        if (wave == 0) {
            if (s < n_steps) {
                const uint32_t buf = s & 1u;
                const uint32_t end_b = (s + 1) * kStepBlocks < n_blocks ? (s + 1) * kStepBlocks : n_blocks;
                if (lane == 0) s_w[buf][0] = (uint8_t)w_prev;
                bool bad = false;
                const uint32_t fast_end = end_b < n_blocks ? end_b : n_blocks - 1;   // the frame's last block: general step
                while (b < end_b) {
                    // ---- fast steps: 64 real candidates, all inside this super-step and inside the LDS window ------------
                    // The step's serial chain is short: address -> LDS read -> ballot -> s_ff1 -> two v_readlane -> a few
                    // scalar adds.  Every lane decodes "its" explicit header in parallel (Terse.hpp:362-370), so the
                    // block that ends the run only has to be picked, not parsed.  (Also consuming the block AFTER that one
                    // in the same step -- an isolated odd block is two explicit headers in a row -- cut the steps per
                    // synth-v1 frame from 602 to 436 but made each step 40 % longer: no gain, not kept.  For streams whose width
                    // changes every block or two, a scalar block-by-block walk over 64 dwords held in VGPRs -- two v_readlane,
                    // a 64-bit shift, the header decode, v_writelane of the width: ~25 instructions per block -- was measured
                    // too: 3.7 ms instead of 3.2 ms per noisy 2000-frame stack; dependent scalar chains run at ~15 clocks per
                    // instruction here.)
                    {
                        uint32_t stride = 1u + kBlock * w_prev;
                        int32_t pos_max = 32 * (c_hi - 1) - (int32_t)frame_sh - 63 * (int32_t)stride;   // window holds 64 candidates + peek
                        uint32_t wide = 0;                                // widest explicit block of these steps (checked once, after them)
                        const uint32_t wbase = (uint32_t)(buf * (kStepBlocks + 68)) + 1u - s * kStepBlocks;   // s_w index of block 0
                        while (b + 64u <= fast_end && (int32_t)pos < pos_max) {
                            // One step = one run of equal widths + the explicit header behind it.  Everything that does not
                            // need `first` is issued before the ballot (every lane decodes "its" explicit header,
                            // Terse.hpp:362-370), the step has one branch (the loop's), and the width / group-offset stores
                            // are unconditional: lanes behind the step's last block write values the next step overwrites,
                            // lanes that start no 64-block group write to a spare row.  (Walker alone: 0.23 -> 0.20 ms per
                            // 2000-frame stack against the branchy version.)
                            const uint32_t lpos = pos + __umul24(lane, stride);
                            const uint32_t fbit = frame_sh - 32u * (uint32_t)c_lo + lpos;
                            const uint32_t bits = __builtin_amdgcn_alignbit(s_chunk[(fbit >> 5) + 1], s_chunk[fbit >> 5], fbit);
                            const uint32_t w3 = (bits >> 1) & 7u, wa = 7u + ((bits >> 4) & 3u), wb = 10u + ((bits >> 6) & 63u);
                            const uint32_t wk = w3 != 7u ? w3 : (wa != 10u ? wa : wb);
                            const uint32_t advk = (w3 != 7u ? 4u : (wa != 10u ? 6u : 12u)) + kBlock * wk;   // header + payload bits
                            const uint64_t stop = ~__ballot((bits & 1u) != 0u);                           // Terse.hpp:361
                            const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                            const bool run = first >= 64u;                                              // all 64 repeat w_prev
                            const uint32_t src = run ? 63u : first;
                            const uint32_t x_w = (uint32_t)__builtin_amdgcn_readlane((int)wk, (int)src);
                            const uint32_t x_adv = (uint32_t)__builtin_amdgcn_readlane((int)advk, (int)src);
                            const uint32_t e_w = run ? w_prev : x_w, adv = run ? 0u : x_adv;
                            wide = e_w > wide ? e_w : wide;
                            const uint32_t n_done = run ? 64u : first + 1u;
                            s_wb[wbase + b + lane] = (uint8_t)(lane < first ? w_prev : e_w);
                            const uint32_t rl = b - s * kStepBlocks + lane;
                            s_goffx[(rl & (kWave - 1)) == 0 && lane < n_done ? buf * kGoffRow + (rl >> 6) : 2 * kGoffRow + lane] = lpos;
                            pos += first * stride + adv;                                                // (bounded by pos_max: inside the window)
                            b += n_done;
                            w_prev = e_w;
                            stride = 1u + kBlock * e_w;
                            pos_max = 32 * (c_hi - 1) - (int32_t)frame_sh - 63 * (int32_t)stride;
                        }
                        if (wide > kMaxW) bad = true;                     // a corrupt header: at worst the steps above wrote bogus widths to LDS
                        if (bad || b >= end_b) break;
                    }
                    const uint32_t stride = 1u + kBlock * w_prev;
                    const uint32_t need_lo = (frame_sh + pos) >> 5;
                    const uint32_t need_hi = ((frame_sh + pos + 63u * stride) >> 5) + 2;
                    if ((int32_t)need_lo < c_lo || (int32_t)need_hi > c_hi) {       // refill the window
                        c_lo = (int32_t)(((frame_dw + need_lo) & ~3ull) - frame_dw);
                        c_hi = c_lo + kFrameChunkDw;
                        const uint64_t d0 = (uint64_t)((int64_t)frame_dw + c_lo);
                        if (base16 && (d0 & 3) == 0 && d0 + kFrameChunkDw <= n_dw) {
                            constexpr int kIt = kFrameChunkDw / (kWave * 4);
                            uint4 x[kIt];
#pragma unroll
                            for (int it = 0; it < kIt; ++it) x[it] = *reinterpret_cast<const uint4*>(s32 + d0 + it * kWave * 4 + lane * 4);
#pragma unroll
                            for (int it = 0; it < kIt; ++it) *reinterpret_cast<uint4*>(&s_chunk[it * kWave * 4 + lane * 4]) = x[it];
                        } else {
                            for (uint32_t i = lane * 4; i < (uint32_t)kFrameChunkDw; i += kWave * 4) {
                                const uint64_t d = d0 + i;
                                uint4 x;
                                x.x = d < n_dw ? s32[d] : 0u; x.y = d + 1 < n_dw ? s32[d + 1] : 0u;
                                x.z = d + 2 < n_dw ? s32[d + 2] : 0u; x.w = d + 3 < n_dw ? s32[d + 3] : 0u;
                                *reinterpret_cast<uint4*>(&s_chunk[i]) = x;
                            }
                        }
                    }
                    const uint32_t fbit = frame_sh + pos + lane * stride - 32u * (uint32_t)c_lo;
                    const uint32_t bits = __builtin_amdgcn_alignbit(s_chunk[(fbit >> 5) + 1], s_chunk[fbit >> 5], fbit);
                    const uint32_t left = end_b - b;                                  // candidates inside this super-step
                    const uint64_t valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
                    const uint64_t same = __ballot((bits & 1u) != 0u) & valid;        // Terse.hpp:361
                    const uint64_t stop = ~same;
                    const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
                    uint32_t e_w = w_prev, new_pos, new_b;
                    if (first < left && first < 64u) {                                // explicit header at block b + first
                        const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)bits, first);
                        uint32_t w = (eb >> 1) & 7u, hl = 4;                          // Terse.hpp:362-370
                        if (w == 7u) {
                            w += (eb >> 4) & 3u; hl = 6;
                            if (w == 10u) { w += (eb >> 6) & 63u; hl = 12; }
                        }
                        if (w > kMaxW) { bad = true; break; }
                        e_w = w;
                        const uint32_t nbv = b + first + 1 == n_blocks ? nb_last : (uint32_t)kBlock;
                        new_pos = pos + first * stride + hl + nbv * w;
                        new_b = b + first + 1;
                        if (new_b == n_blocks) final_pos = new_pos;
                    } else {                                                          // the rest of the window repeats w_prev
                        const uint32_t cnt = left < 64u ? left : 64u;
                        if (b + cnt == n_blocks) final_pos = pos + (cnt - 1) * stride + 1u + nb_last * w_prev;
                        new_pos = pos + cnt * stride;
                        new_b = b + cnt;
                    }
                    const uint32_t n_done = new_b - b, rel = b - s * kStepBlocks;
                    if (lane < n_done) {
                        s_w[buf][1 + rel + lane] = (uint8_t)(lane < first ? w_prev : e_w);
                        if (((rel + lane) & (kWave - 1)) == 0) s_goff[buf][(rel + lane) >> 6] = pos + lane * stride;
                    }
                    pos = new_pos;
                    w_prev = e_w;
                    b = new_b;
                    if (pos > limit + 64u * 400u) { bad = true; break; }              // ran away: corrupt stream
                }
                if (!bad && b == n_blocks)                                            // S_f = 1 + bits/8 (Terse.hpp:547)
                    bad = !(final_pos <= limit && 1 + (uint64_t)final_pos / 8 == fe - fo);
                if (bad && lane == 0) s_err = 1u;
                // A stream with an explicit header every few blocks costs this walker a step per header (10 x the time of a
                // run-dominated frame); false chains merge quickly in such streams, so the frame goes to the
                // position-parallel walk instead (decode_seg.hip).  Decided after super-steps 0, 3 and 11 on the width
                // changes inside the super-step just walked (12 widths per lane, outside the step loop).  (Probing the first 256
                // blocks instead -- a second bound in the fast loop -- hands a header-dense frame over after 0.06 instead of
                // 0.13 ms but cost every other stack 6-60 %: the loop bound became loop-variant.)
                if (defer && !bad && (s == 0u || s == 3u || s == 11u) && end_b == (s + 1) * kStepBlocks && end_b < n_blocks) {
                    uint32_t changes = 0;
#pragma unroll
                    for (int i = 0; i < kStepBlocks / kWave; ++i) {
                        const uint32_t at = lane * (kStepBlocks / kWave) + i;
                        changes += s_w[buf][at] != s_w[buf][at + 1] ? 1u : 0u;
                    }
                    const uint32_t inc = wave_inclusive_scan(changes);
                    if ((uint32_t)__builtin_amdgcn_readlane((int)inc, 63) * 6u > (uint32_t)kStepBlocks && lane == 0) s_err = 2u;
                }
            }
#ifdef TRPX_DEC_WALK_ONLY
        } else if (false) {                            // diagnostic build (tools): time the walker alone
#else
        } else if (s >= 1) {
#endif
            // ---- unpack super-step s-1: this wave's groups, one after the other ------------------------------------
            // Every lane loads the stream dwords of its own block straight from L2 (the walker has just pulled them
            // through): one or more dwordx4 loads starting at the dword that holds the block's first payload bit, as many
            // as the widest block of the group needs; the width-specialised bodies then work on registers only.
            const uint32_t pbuf = (s - 1) & 1u;
            constexpr int NQ = RawQuads<T>::n;
            const uint32_t* __restrict__ fbase = s32 + frame_dw;          // wave-uniform base; per-lane 32-bit dword offsets
#pragma unroll 1
            for (int gq = 0; gq < kGroupsPerWave; ++gq) {
                const uint32_t gi = (uint32_t)(wave - 1) * (uint32_t)kGroupsPerWave + gq;
                const uint32_t rel = gi * kWave + lane;
                const uint32_t blk = (s - 1) * kStepBlocks + rel;
                if ((s - 1) * kStepBlocks + gi * kWave >= n_blocks) break;            // wave-uniform: group past the frame's end
                uint32_t w = 0, hl = 0;
                int nb = 0;
                if (blk < n_blocks) {
                    w = s_w[pbuf][1 + rel];
                    const uint32_t wp = s_w[pbuf][rel];
                    hl = header_len(w, wp);
                    nb = blk + 1 == n_blocks ? (int)nb_last : kBlock;
                }

                const uint32_t len = nb ? hl + __umul24((uint32_t)nb, w) : 0u;
                const uint32_t inc = wave_inclusive_scan(len);
#ifdef TRPX_DEC_NO_EXTRACT
                if (inc == 0xFFFFFFFFu) fout[0] = (T)w;                               // (diagnostic build: no loads, no extraction, no stores)
                continue;
#endif
                const uint32_t q = frame_sh + s_goff[pbuf][gi] + (inc - len) + hl;    // first payload bit, relative to dword frame_dw
                const uint32_t dq = q >> 5, sq = q & 31u;
                // (Issuing the NEXT group's loads before extracting this one was measured twice: a second set of raw registers
                // means 64 VGPRs with spills at 8 workgroups per CU, 0.39 ms instead of 0.32 ms; only the first 16 bytes
                // per lane in flight -- enough for widths <= 8 -- still 0.355 ms: the next group's width reads and scan
                // in front of the extraction cost more than the exposed L2 round trip.)
                // All NQ quads, whatever the widths: a 16-byte load more per lane is cheaper than a wavefront max of the
                // widths, and the extra bytes are the neighbours' (same cache lines).
                const uint32_t last_dw = (uint32_t)__builtin_amdgcn_readlane((int)dq, 63) + 4u * NQ;   // lanes ascend in position
                uint32_t raw[4 * NQ];
                if (frame_dw + last_dw <= n_dw) {                                     // wave-uniform: the loads stay inside the stream
#pragma unroll
                    for (int i = 0; i < NQ; ++i) {
                        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                        u4 x4;
                        __builtin_memcpy(&x4, fbase + dq + 4 * i, 16);                 // dword-aligned 16-byte load
                        raw[4 * i] = x4.x; raw[4 * i + 1] = x4.y; raw[4 * i + 2] = x4.z; raw[4 * i + 3] = x4.w;
                    }
                } else {                                                              // the stack's last bytes: guarded element loads
#pragma unroll
                    for (int i = 0; i < 4 * NQ; ++i) {
                        const uint64_t d = frame_dw + dq + i;
                        raw[i] = d < n_dw ? s32[d] : 0u;
                    }
                }
                T* __restrict__ dst = fout + (uint64_t)blk * kBlock;
                uint64_t todo = __ballot(nb == kBlock);
                if (kStaged && todo == ~0ull) {
                    // 64 full blocks: every lane leaves its 12 pixels in the wave's LDS row, then the wave stores the group
                    // 16 bytes per lane -- whole lines per store instruction instead of 24-byte runs
                    uint32_t* const stage = s_out[wave - 1];
                    uint32_t* const row = stage + lane * (kBlock * (uint32_t)sizeof(T) / 4u);
                    while (todo) {
                        const int l0 = __builtin_ctzll(todo);
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l0);
                        const bool mine = w == w0;
                        uint32_t ss = sq;
                        asm volatile("" : "+v"(ss));                                  // keep the specialised bodies out of LICM's reach
                        if (mine) UnpackStageDispatch<T, 0, PixelTraits<T>::bits>::run(raw, ss, w0, row);
                        todo &= ~__ballot(mine);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    store_group<T>(stage, fout + (uint64_t)((s - 1) * kStepBlocks + gi * kWave) * kBlock);
                    __builtin_amdgcn_wave_barrier();                                  // (the row is rewritten by the next group)
                    continue;
                }
                while (todo) {
                    const int l0 = __builtin_ctzll(todo);
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l0);
                    const bool mine = nb == kBlock && w == w0;
                    uint32_t ss = sq;
                    asm volatile("" : "+v"(ss));                                      // keep the specialised bodies out of LICM's reach
                    if (mine) UnpackStoreDispatch<T, 0, PixelTraits<T>::bits>::run(raw, ss, w0, dst);
                    todo &= ~__ballot(mine);
                }
                if (nb && nb != kBlock) {                                             // the frame's last, partial block
                    const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
                    uint32_t p = q;
                    for (int k = 0; k < nb; ++k) {
                        uint32_t f = 0;
                        if (w) {
                            const uint64_t d = frame_dw + (p >> 5);
                            const uint64_t two = (uint64_t)(d < n_dw ? s32[d] : 0u) | ((uint64_t)(d + 1 < n_dw ? s32[d + 1] : 0u) << 32);
                            f = (uint32_t)(two >> (p & 31u)) & mask;
                            if (PixelTraits<T>::is_signed) f = (uint32_t)((int32_t)(f << (32u - w)) >> (32u - w));
                        }
                        dst[k] = (T)f;
                        p += w;
                    }
                }
            }
        }
        __syncthreads();                               // super-step boundary: widths of step s published, step s-1 consumed
        if (s_err) break;
    }
    const bool deferred = s_err == 2u;
    if (s_err == 1u && threadIdx.x == 0) atomicMax(&status[0], 5u);        // TRPX_ERR_CORRUPT
    if (deferred && threadIdx.x == 0) defer[1u + atomicAdd(&defer[0], 1u)] = (uint32_t)frame;   // listed: k_seg_frames + k_unpack_listed do it
}

template <typename T>
static hipError_t launch_decode_frames_t(const DecodeArgs& a, hipStream_t st) {
    uint32_t* defer = a.defer && a.seg_ws && seg_single_wave(a.geom) ? a.defer : nullptr;
    hipLaunchKernelGGL(k_zero_words<0>, dim3(1), dim3(kThreads), 0, st, reinterpret_cast<uint64_t*>(defer), (uint64_t)(defer ? 1 : 0),
                       reinterpret_cast<uint64_t*>(a.status), (uint64_t)4);   // status block + the deferred-frame count
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    hipLaunchKernelGGL((k_decode_frames<T>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                       a.frame_offsets, a.geom, static_cast<T*>(a.pixels_out), defer, a.status);
    prof.mark(st);
    if (defer) {
        const hipError_t e = launch_decode_deferred(PixelTraits<T>::bits == 8 ? (PixelTraits<T>::is_signed ? 1 : 0)
                                                    : PixelTraits<T>::bits == 16 ? (PixelTraits<T>::is_signed ? 3 : 2)
                                                                                 : (PixelTraits<T>::is_signed ? 5 : 4), a, st);
        prof.mark(st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// Preconditions (checked by the caller): frame offsets known, n_values % 4 == 0, pixels_out 16-byte aligned,
// frames of < 2^32 bits.
hipError_t launch_decode_frames(int dtype, const DecodeArgs& a, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_decode_frames_t<uint8_t>(a, st);
    case 1: return launch_decode_frames_t<int8_t>(a, st);
    case 2: return launch_decode_frames_t<uint16_t>(a, st);
    case 3: return launch_decode_frames_t<int16_t>(a, st);
    case 4: return launch_decode_frames_t<uint32_t>(a, st);
    case 5: return launch_decode_frames_t<int32_t>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace trpx
