// PROLIX header walk of HEADER-DENSE frames (gfx950 / CDNA4): one speculative pass, link walks that stop where chains merge,
// a write pass.  Replaces the fix-point rounds of decode_seg.hip for the frames the per-frame decoder hands over (reference
// include/Terse.hpp:360-372: block b+1's position is only known after block b's header; detector data changes its block width
// every few blocks, so the serial walker's one step per explicit header is a step per block or two).
//
// decode_seg.hip walks such a frame with one wavefront, 64 segments, and REPEATS whole passes until every lane's IN state is
// its predecessor's OUT state: ~2.5 counting passes + the write pass = 3.5 x 341 dependent steps per 512 x 512 frame at two
// wavefronts per SIMD.  What the rounds recompute is almost all known after the first: a chain started in a wrong state merges
// with the frame's chain after a few dozen blocks (the state behind an explicit header does not depend on the width before
// it), and from there on its walk IS the frame's.  Here every pass is walked once:
//
//   spec    G = 64 W segments per frame (W wavefronts in one workgroup).  Lane j walks its region [B_j, B_j+1) from a guess --
//           inside a run of equal widths (seg_comb_guess) or the plain (X_j, 0) -- counts its blocks and leaves a CHECKPOINT
//           (state, blocks so far) at every window boundary (768 bits) it crosses.
//   link    lane j goes on from its OUT state into region j + 1, as a chain of its own: if lane j was on the frame's chain at its
//           end -- it is, unless its own guess chain never merged --, this is the frame's chain in region j + 1.  At every window
//           boundary it compares its state with lane j + 1's checkpoint; equal: merged, the rest of lane j + 1's walk is this
//           chain's, and the region's true block count follows from the two counts.  A chain that reaches the region's end
//           unmerged (lane j + 1's guess chain never met it) leaves its own count and end state and goes on into region j + 2,
//           depth 2, ... -- one record per (region, depth).
//   resolve S_0 = (0, 0) is true (Terse.hpp:359, :505).  Which record is region s's TRUE walk follows from region s - 1's: a
//           merged walk ends in lane s's OUT state, so region s + 1 is covered by lane s's own link (depth 1), or by lane s + 1's
//           own walk if its guess was that state; an unmerged walk covers the next region itself (depth + 1).  That is a state
//           machine over the segments with seven states -- a composition scan over 3-bit tables --, and prefix sums of the true
//           counts give every region's first block.
//   write   every lane walks its region once more from its TRUE state with its block number (seg_walk<true>: widths[], group
//           offsets) and checks that it ends in the next region's true state, the last one that S_f = 1 + bits/8 (Terse.hpp:547).
//
// Guesses, merges and records only steer the speed: a frame whose write pass does not close is walked again by one lane from
// (0, 0), block by block -- the serial walk, whose verdict on a corrupt stream is the final one.
//
// Dependent steps per 512 x 512 frame (W = 2, 128 segments of ~170 blocks): 170 (spec) + ~320 (the longest of 127 links: a guess
// chain merges at ~1.5 % per block) + 170 (write) against 3.5 x 341, at four wavefronts per SIMD instead of two.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include "seg_common.hpp"

namespace trpx {

constexpr uint32_t kDenseCk = 16;              // checkpoint entries per segment (LDS: 4 bytes each)
constexpr uint32_t kDenseSlots = 6;            // records per region: depth 1 .. 5, and one for every deeper walk
constexpr uint32_t kDenseMaxDepth = 24;        // regions a link walk crosses at most
constexpr uint32_t kStDeep = 6, kStFail = 7;   // states 0 (the lane's own walk) .. 5 = depth; 6 = deeper; 7 = no true walk known
enum : uint32_t { kRecMerged = 2, kRecThrough = 3, kRecFail = 4 };
struct DenseRec { uint64_t out; uint32_t cnt, flag; };
constexpr uint32_t kDenseIdent = 0u | 1u << 3 | 2u << 6 | 3u << 9 | 4u << 12 | 5u << 15 | 6u << 18;

#ifndef TRPX_DENSE_SEG_BLOCKS
#define TRPX_DENSE_SEG_BLOCKS 330
#endif
constexpr uint32_t kDenseSegBlocks = TRPX_DENSE_SEG_BLOCKS;   // blocks per segment, at least about
uint32_t dense_waves(const FrameGeom& g) {
    // segments of ~170 blocks (a guess chain has merged within its own segment 93 % of the time at 1.5 % per block)
    const uint32_t lanes = g.n_blocks / kDenseSegBlocks;
    return lanes <= 80u ? 1u : lanes <= 160u ? 2u : 4u;       // (W = 8 would not fit the LDS: 9 KB of windows + 4 KB of checkpoints per wavefront)
}
size_t dense_workspace_bytes(const FrameGeom& g, size_t n_frames) {
    const size_t segs = n_frames * 64u * dense_waves(g);
    return align_up(segs * (kDenseSlots * sizeof(DenseRec) + 8u), 256);
}

// first the map A, then B (3 bits per state 0 .. 6; 7 = fail stays fail)
__device__ __forceinline__ uint32_t dense_compose(uint32_t A, uint32_t B) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t e = 0; e < 7u; ++e) {
        const uint32_t a = (A >> (3u * e)) & 7u;
        const uint32_t b = a == 7u ? 7u : (B >> (3u * a)) & 7u;
        r |= b << (3u * e);
    }
    return r;
}
__device__ __forceinline__ uint32_t dense_scan_tables(uint32_t x) {             // inclusive, lane 0 first
    const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t prev = (uint32_t)__shfl_up((int)x, (int)d, 64);
        if (lane >= d) x = dense_compose(prev, x);
    }
    return x;
}
__device__ __forceinline__ uint32_t dense_wave_sum(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(v), 63); }

// A checkpoint: the state in which a lane's walk crosses a window boundary and the blocks it has counted up to there, 4 bytes in
// LDS (in HBM every 8-byte entry was a partial-line write and a sector fetch: the walk moved twice the bytes of the rounds it
// replaces) -- position relative to the boundary (a block is at most 396 bits), width, count.  0: none.
__device__ __forceinline__ uint32_t dense_ck_pack(uint32_t rel, uint32_t w, uint32_t cnt) {
    return rel < 512u && cnt < (1u << 16) ? 1u | rel << 1 | (w & 63u) << 10 | cnt << 16 : 0u;
}
struct DenseSpec {                             // seg_walk's hook of the speculative pass: leaves the checkpoints
    uint32_t* ck;                              // this lane's kDenseCk entries
    uint32_t every, X;
    __device__ __forceinline__ void open(uint32_t, bool) {}
    __device__ __forceinline__ bool close(uint32_t t, bool act0, uint32_t wend, uint32_t& pos, uint32_t& w, uint32_t& n, bool&, uint32_t&) {
        if (act0 && pos >= wend) {                                             // crossed the boundary behind window t
            const uint32_t i = (t + 1u) / every;
            if (i * every == t + 1u && i < kDenseCk) ck[i] = dense_ck_pack(pos - wend, w, n);
        }
        return false;
    }
};

// The link walk's bookkeeping (see the head of the file), called by seg_walk when a window closes.
struct DenseLink {
    const uint32_t* s_ck;      // the frame's checkpoints (LDS)
    DenseRec* rec;             // the frame's records
    const uint64_t* g_out;     // OUT states of the frame's lanes (global: written in front of the workgroup's barrier)
    const uint32_t *s_cnt, *s_bnd;
    uint32_t j, jl, n_w, every, limit;
    bool have;
    uint32_t s, d, depth, n_entry;
    uint64_t so;               // the OUT state of the region's own lane (requested when the region is entered)
    __device__ __forceinline__ uint32_t ck_index(uint32_t t) const {          // entry of region s for the boundary behind this lane's window t; ~0: none
        if (t + 1u < (s - j) * n_w) return ~0u;                                // (the region starts behind that boundary)
        const uint32_t b = t + 1u - (s - j) * n_w, i = b / every;             // boundary number inside region s
        return i * every == b && i < kDenseCk ? i : ~0u;
    }
    __device__ __forceinline__ void open(uint32_t, bool) {}
    // at or behind the region's end: the record, and on into the next region
    __device__ __forceinline__ void region_end(uint32_t& pos, uint32_t& w, uint32_t& n, bool& done, uint32_t& end) {
        while (have && (pos >= end || pos > limit)) {
            DenseRec r;
            const uint32_t at = s * kDenseSlots + d - 1u;
            const uint64_t here = seg_pack(pos, w);
            if (pos > limit) { r = DenseRec{0ull, 0u, kRecFail}; have = false; }                                   // (a chain that left the frame: not the frame's)
            else {
                if (here == so) { r = DenseRec{so, n - n_entry, kRecMerged}; have = false; }                       // merged by the region's end
                else {
                    r = DenseRec{here, n - n_entry, kRecThrough};
                    if (s + 1u >= jl) have = false;                                                                 // the last counted region: the tail starts here
                    else if (++depth > kDenseMaxDepth) { r.flag = kRecFail; have = false; }
                    else { ++s; d = d < kStDeep ? d + 1u : kStDeep; n_entry = n; end = s_bnd[s + 1u]; so = g_out[s]; }
                }
            }
            rec[at] = r;
        }
        done = !have;
    }
    __device__ __forceinline__ bool close(uint32_t t, bool act0, uint32_t wend, uint32_t& pos, uint32_t& w, uint32_t& n, bool& done, uint32_t& end) {
        if (!act0 || !have) return false;
        region_end(pos, w, n, done, end);
        if (have && pos >= wend) {                                             // a boundary inside the region: has the chain met the region's own walk?
            const uint32_t i = ck_index(t);
            if (i != ~0u) {
                const uint32_t ck = s_ck[s * kDenseCk + i];
                if (ck != 0u && (ck & 0xFFFFu) == (dense_ck_pack(pos - wend, w, 0u) & 0xFFFFu)) {
                    rec[s * kDenseSlots + d - 1u] = DenseRec{so, (n - n_entry) + (s_cnt[s] - (ck >> 16)), kRecMerged};
                    have = false; done = true;
                }
            }
        }
        return have && !done && pos < wend;
    }
};

#ifdef TRPX_DENSE_STAMPS
// diagnostic build (tools/r6_variant.sh ... -DTRPX_DENSE_STAMPS): per-frame phase times (10 ns ticks) folded into the status block --
// [3] max total, [4] max link, [5] sum of link / 16, [6] sum of total / 16, [7] max spec
#define TRPX_DENSE_STAMP_PRINT() do { if (threadIdx.x == 0) { const uint32_t tot = (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_t0), lnk = (uint32_t)(st_t2 - st_t1); \
    atomicMax(&status[3], tot); atomicMax(&status[4], lnk); atomicAdd(&status[5], lnk >> 4); atomicAdd(&status[6], tot >> 4); atomicMax(&status[7], (uint32_t)(st_t1 - st_t0)); (void)st_t3; (void)st_depths; } } while (0)
#endif
template <uint32_t W>
__global__ __launch_bounds__(64 * W, W >= 4 ? 1 : 2) void k_dense_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                            const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                            DenseRec* __restrict__ rec_all, uint64_t* __restrict__ out_all,
                                                            uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                            const uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    constexpr uint32_t G = 64u * W;
    __shared__ uint32_t s_win[W][kWave * kSegRow];
    __shared__ uint64_t s_exit[W];
    __shared__ uint32_t s_cnt[G], s_bnd[G + 1], s_ck[G * kDenseCk];
    __shared__ uint32_t s_agg[W], s_aggcnt[W][8];
    __shared__ uint32_t s_bad;
    const uint32_t slot = blockIdx.x;
    if (slot >= list[0]) return;
    const uint32_t entry = list[1u + slot];                    // bit 31: a width change every third block and more -- no run to look for
    const uint64_t frame = entry & 0x7FFFFFFFu;
    const bool run_guess = (entry >> 31) == 0u;
    const uint32_t lane = (uint32_t)lane_id(), k = (uint32_t)wave_id(), j = 64u * k + lane;
    uint32_t* const win = s_win[k];
    // regions of at least ~160 blocks: a link that meets a region's own walk only after the region's end has to cross it, and a
    // chain that has to cross regions all the time is a serial walk (small frames use fewer lanes)
    const uint32_t g_eff = g.n_blocks / kDenseSegBlocks < G ? (g.n_blocks / kDenseSegBlocks ? g.n_blocks / kDenseSegBlocks : 1u) : G;
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, g_eff, status)) {
        if (threadIdx.x == 0) atomicMax(&status[0], 5u);
        return;
    }
    c.L = (c.L + kSegAdv - 1u) / kSegAdv * kSegAdv;            // (a multiple of the window advance -- and of 128 bits, like seg_len_bits': every region's window grid is the frame's)
    uint8_t* const wf = widths + frame * g.n_blocks;
    uint64_t* const tf = tile_off + frame * g.n_tiles;
    seg_zero_widths(wf, g.n_blocks, k, W);
    const uint32_t jl = seg_last_live(c.limit, c.L, g_eff);
#ifdef TRPX_DENSE_FORCE_SERIAL                                  // test build (make denseserial): every listed frame takes the last resort
    if (threadIdx.x == 0) { s_bad = 1u; s_bnd[G] = G * c.L; }
#else
    if (threadIdx.x == 0) { s_bad = 0u; s_bnd[G] = G * c.L; }
#endif
    // checkpoints: every `every`-th window boundary, so that a segment's fit its kDenseCk entries
    const uint32_t n_win = (c.L + kSegSpan) / kSegAdv + 2u;
    const uint32_t every = (n_win + kDenseCk - 2u) / (kDenseCk - 1u);
    const uint64_t seg0 = frame * G;
    DenseRec* const rec = rec_all + seg0 * kDenseSlots;
    uint64_t* const g_out = out_all + seg0;

#ifdef TRPX_DENSE_STAMPS
    const uint64_t st_t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t st_t1 = 0, st_t2 = 0, st_t3 = 0;
    uint32_t st_depths = 0;
#endif
    // ---- spec: every lane walks its region from a guess ------------------------------------------------------------------------
    const bool walks = j < jl;
    uint64_t in = seg_pack(j * c.L, 0u);
    uint32_t B = j * c.L;
    if (run_guess && __ballot(walks)) {
        const uint32_t oct = lane & ~7u, piece = lane & 7u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t d0 = ((c.fa + (uint64_t)(64u * k + oct + q) * c.L) >> 5) & ~3ull;
            *reinterpret_cast<seg_u4*>(&win[(oct + q) * kSegRow + 4u * piece]) = seg_load16(c, d0 + 4u * piece);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef TRPX_SEG_STAMPS
        const uint64_t clkg = __builtin_amdgcn_s_memrealtime();
#endif
        const uint64_t gs = seg_comb_guess(c, win, j * c.L);
#ifdef TRPX_SEG_STAMPS
        c.clk_guess += __builtin_amdgcn_s_memrealtime() - clkg;
#endif
        if (gs != ~0ull && lane > 0u && walks) { in = gs; B = (uint32_t)gs; }   // (lane 0: the wave before ends at X)
        __builtin_amdgcn_wave_barrier();
    }
    if (j == 0u) { in = 0ull; B = 0u; }                        // the frame starts with width 0 at bit 0 (Terse.hpp:359, :505)
    uint32_t endB = (uint32_t)__shfl_down((int)B, 1, 64);
    if (lane == 63u) endB = (j + 1u) * c.L;
    uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = 0u;
    bool bad = false;
    {
#pragma unroll
        for (uint32_t i = 0; i < kDenseCk; ++i) s_ck[j * kDenseCk + i] = 0u;                                  // (no checkpoint)
        DenseSpec sp{s_ck + j * kDenseCk, every, j * c.L};
        seg_walk<false, DenseSpec>(c, win, 64u * k, walks, endB, false, pos, w, n, nullptr, nullptr, bad, nullptr, &sp);
    }
    const uint64_t out = seg_pack(pos, w);
    // the guess of the lane behind (a wavefront's lane 0 never takes a run guess)
    uint64_t in_next = (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)in, 1, 64) | ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(in >> 32), 1, 64) << 32);
    if (lane == 63u) in_next = seg_pack((j + 1u) * c.L, 0u);
    g_out[j] = out; s_cnt[j] = walks ? n : 0u; s_bnd[j] = B;
    __syncthreads();

#ifdef TRPX_DENSE_STAMPS
    st_t1 = __builtin_amdgcn_s_memrealtime();

#endif
    // ---- link: on from the OUT state into the regions behind, until the chain meets the walk of the lane that owns the region ----
    // (the lane's windows simply go on: with L a multiple of the window advance the grids of all regions coincide, window t of this
    // lane is window t - (s - j) L / 768 of region s)
    {
        DenseLink hk;
        hk.s_ck = s_ck; hk.rec = rec; hk.g_out = g_out; hk.s_cnt = s_cnt; hk.s_bnd = s_bnd;
        hk.j = j; hk.jl = jl; hk.n_w = c.L / kSegAdv; hk.every = every; hk.limit = c.limit;
        hk.have = j + 1u < jl && out != in_next;
        hk.s = j + 1u; hk.d = 1u; hk.depth = 1u; hk.n_entry = 0u;
        hk.so = hk.have ? g_out[j + 1u] : 0ull;
        n = 0u;
        uint32_t end2 = hk.have ? s_bnd[j + 2u] : 0u;
        bool done2 = !hk.have;
        if (hk.have) hk.region_end(pos, w, n, done2, end2);               // (a chain that is behind its first region already)
        const uint32_t t_mine = hk.have && !done2 ? (pos - j * c.L) / kSegAdv : 0xFFFFFFFFu;
        const uint32_t t_start = ~wave_max(~t_mine);
        if (t_start != 0xFFFFFFFFu)
            seg_walk<false, DenseLink>(c, win, 64u * k, hk.have && !done2, end2, false, pos, w, n, nullptr, nullptr, bad, nullptr, &hk, t_start);
#ifdef TRPX_DENSE_STAMPS
        st_depths = hk.depth;
#endif
    }
    __syncthreads();

#ifdef TRPX_DENSE_STAMPS
    st_t2 = __builtin_amdgcn_s_memrealtime();

#endif
    // ---- resolve: the true walk of every region, its first block, its start state ---------------------------------------------------
    uint32_t tbl = kDenseIdent, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
    DenseRec r1{}, r2{}, r3{}, r4{}, r5{}, r6{};
    if (j < jl) {
        const uint32_t mex = in_next == out ? 0u : 1u;      // behind a merged walk: lane j + 1's own walk if its guess was my OUT state, else my link
        tbl = mex;                                                              // state 0: the lane's own walk -- true in segment 0 and behind a closed link
        c0 = s_cnt[j];
        if (j > 0u) {
            r1 = rec[j * kDenseSlots + 0u]; r2 = rec[j * kDenseSlots + 1u]; r3 = rec[j * kDenseSlots + 2u];
            r4 = rec[j * kDenseSlots + 3u]; r5 = rec[j * kDenseSlots + 4u]; r6 = rec[j * kDenseSlots + 5u];
            auto nxt = [&](const DenseRec& r, uint32_t d) {
                return r.flag == kRecMerged ? mex : (r.flag == kRecThrough ? (d < kStDeep ? d + 1u : kStDeep) : kStFail);
            };
            tbl |= nxt(r1, 1u) << 3 | nxt(r2, 2u) << 6 | nxt(r3, 3u) << 9 | nxt(r4, 4u) << 12 | nxt(r5, 5u) << 15 | nxt(r6, 6u) << 18;
            c1 = r1.cnt; c2 = r2.cnt; c3 = r3.cnt; c4 = r4.cnt; c5 = r5.cnt; c6 = r6.cnt;
        } else tbl |= kStFail << 3 | kStFail << 6 | kStFail << 9 | kStFail << 12 | kStFail << 15 | kStFail << 18;
    }
    auto cnt_of = [&](uint32_t d) { return d == 0u ? c0 : d == 1u ? c1 : d == 2u ? c2 : d == 3u ? c3 : d == 4u ? c4 : d == 5u ? c5 : d == 6u ? c6 : 0u; };
    const uint32_t incl = dense_scan_tables(tbl);
    uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
    if (lane == 0u) excl = kDenseIdent;
    {
        const uint32_t agg = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#pragma unroll
        for (uint32_t e = 0; e < 7u; ++e) {                                     // the chunk's blocks for every state it may be entered in
            const uint32_t tot = dense_wave_sum(cnt_of((excl >> (3u * e)) & 7u));
            if (lane == 0u) s_aggcnt[k][e] = tot;
        }
        if (lane == 0u) s_agg[k] = agg;
    }
    __syncthreads();
    uint32_t st = 0u, base = 0u;                                                // the state in which this chunk is entered, the blocks in front of it
    for (uint32_t kk = 0; kk < k; ++kk) {
        if (st != kStFail) { base += s_aggcnt[kk][st]; st = (s_agg[kk] >> (3u * st)) & 7u; }
    }
    const uint32_t dl = st == kStFail ? kStFail : (excl >> (3u * st)) & 7u;    // the state in which this lane's region is entered
    const uint32_t my_cnt = j < jl ? cnt_of(dl) : 0u;
    const uint32_t my_base = base + wave_inclusive_scan(my_cnt) - my_cnt;
    uint64_t exit_state = out;                                                  // the state behind my region's true walk
    {
        const uint32_t fl = dl == 1u ? r1.flag : dl == 2u ? r2.flag : dl == 3u ? r3.flag : dl == 4u ? r4.flag : dl == 5u ? r5.flag : r6.flag;
        const uint64_t ro = dl == 1u ? r1.out : dl == 2u ? r2.out : dl == 3u ? r3.out : dl == 4u ? r4.out : dl == 5u ? r5.out : r6.out;
        if (dl >= 1u && dl <= 6u && fl == kRecThrough) exit_state = ro;
    }
    if (lane == 63u) s_exit[k] = exit_state;
    if (__ballot(dl == kStFail && j <= jl) && lane == 0u) s_bad = 1u;
    __syncthreads();
    uint64_t t_in = seg_shfl_up1(exit_state);
    if (lane == 0u) t_in = k > 0u ? s_exit[k - 1u] : 0ull;
    if (j == 0u) t_in = 0ull;

#ifdef TRPX_DENSE_STAMPS
    st_t3 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- write (and, for a frame that does not close, the serial walk) ----------------------------------------------------------------
    for (uint32_t attempt = 0; attempt < 2u; ++attempt) {
        const bool serial = attempt == 1u;
        if (serial) {
            if (!s_bad) break;
            __syncthreads();
            seg_zero_widths(wf, g.n_blocks, k, W);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(&status[2], 1u);                    // (frames that took the slow way)
            if (k != 0u) break;
        } else if (s_bad) continue;
        const bool part = serial ? lane == 0u : j <= jl;
        const bool last = serial ? true : j == jl;
        const SegOrigin org{serial ? 0u : j * c.L, g.n_blocks};
        pos = serial ? 0u : (uint32_t)t_in; w = serial ? 0u : (uint32_t)(t_in >> 32); n = serial ? 0u : my_base;
        bad = part && n > g.n_blocks;
        const uint32_t end_w = last ? 0xFFFFFFFFu : s_bnd[j + 1u < G ? j + 1u : G];
        seg_walk<true>(c, win, 0u, part && !bad, end_w, last, pos, w, n, wf, tf, bad, &org);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the walk's stores are inside asm statements: the compiler does not count them)
        // ends in the next region's true state, having counted the blocks the prefix sums gave it (a record that two deep link walks
        // shared may hold another chain's count with the right end state)
        if (part && !last && (seg_pack(pos, w) != exit_state || n != my_base + my_cnt)) bad = true;
        if (part && last && !(n == g.n_blocks && pos <= c.limit && 1u + pos / 8u == c.limit / 8u)) bad = true;   // S_f = 1 + bits/8 (Terse.hpp:547)
        if (serial) {
            if (__ballot(bad) && lane == 0u) atomicMax(&status[0], 5u);         // TRPX_ERR_CORRUPT: the serial walk's verdict
        } else {
            if (__ballot(bad) && lane == 0u) s_bad = 1u;
            __syncthreads();
        }
    }
#ifdef TRPX_DENSE_STAMPS
    TRPX_DENSE_STAMP_PRINT();
#endif
}

hipError_t launch_dense_listed(const DecodeArgs& a, uint32_t max_w, void* dense_ws, const uint32_t* list, hipStream_t st) {
    const uint32_t W = dense_waves(a.geom);
    const size_t segs = (size_t)a.n_frames * 64u * W;
    DenseRec* rec = static_cast<DenseRec*>(dense_ws);
    uint64_t* outs = reinterpret_cast<uint64_t*>(rec + segs * kDenseSlots);
#define TRPX_DENSE_LAUNCH(WW)                                                                                                             \
    hipLaunchKernelGGL((k_dense_frames<WW>), dim3(a.n_frames), dim3(64 * WW), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, \
                       a.geom, max_w, rec, outs, a.widths, a.tile_off, list, a.status)
    switch (W) {
    case 1: TRPX_DENSE_LAUNCH(1); break;
    case 2: TRPX_DENSE_LAUNCH(2); break;
    default: TRPX_DENSE_LAUNCH(4); break;
    }
#undef TRPX_DENSE_LAUNCH
    return hipGetLastError();
}

}  // namespace trpx
