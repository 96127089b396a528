// PROLIX header walk, position-parallel (gfx950 / CDNA4): the serial chain of jpa::Terse::prolix
// (reference include/Terse.hpp:360-372 -- block b+1's bit position is only known after block b's header) is
// broken into SEGMENTS of the frame's bit range that are walked speculatively, one lane per segment, and then
// verified.  Replaces the one-wavefront-per-frame walk (k_walk_lds) wherever the chain itself set the decode
// time: few large frames (1.4 M chained blocks in a 4096 x 4096 frame) and streams whose width changes every
// block or two (one walker step per explicit header).
//
//   chain state   (pos, w): bit position of the next block header and the width of the block before it.
//                 header bit 1 -> (pos + 1 + 12 w, w); header bit 0 -> w' from the 3/5/11 field bits,
//                 (pos + hl + 12 w', w') -- independent of w.
//   segments      the frame's bit range is cut at X_j = j * L (G segments, L a multiple of 128 bits).  Lane j
//                 owns the blocks whose header starts in [X_j, X_j+1).  Its IN state is the first block start
//                 at or after X_j and the width before it; its OUT state is the same thing at X_j+1.
//   speculation   lane 0 starts from the true state (0, 0) (Terse.hpp:359).  Every other lane starts from a guess:
//                 (X_j, 0) -- a wrong chain and the true chain merge as soon as they meet at an explicit header, or
//                 at any block start with equal widths, which on streams with frequent explicit headers happens
//                 within a few dozen blocks, so a guessed lane still ENDS in the true state -- or, where the stream
//                 is a run of equal-width blocks (nothing to merge at), a position inside the run found by its
//                 periodic header bits (seg_comb_guess; the lane's boundary B_j moves there).
//   fix point     in_j <- out_(j-1), re-walk the lanes whose IN state changed, repeat until nothing changes.
//                 in_0 is true and in_(j+1) = F_j(in_j) for every j, hence by induction every state is the true
//                 one; at most G rounds, typically 2-4.  Which of two conflicting states a lane believes (its own
//                 run guess or its predecessor's OUT state) only steers the speed, see seg_fixpoint.  Inside a
//                 wavefront the rounds are a loop; across wavefronts of one frame the last OUT state of wave k-1
//                 is read from memory by wave k on the next launch, and k_seg_resolve re-checks every such link
//                 (and re-runs a wave serially if one is still open) before anything is written.  A frame that
//                 does not converge (long runs of wide blocks: nothing to merge at, no guess) is handed to the
//                 serial walk (k_walk_lds), so the worst case is the old one.
//   write pass    block counts -> prefix sum -> every lane walks its segment once more from its verified IN
//                 state and stores width[b] (u8, array pre-zeroed: zero widths are not stored) and the bit
//                 offset of every 256-block group: the same decode index k_walk_lds emits, consumed by
//                 k_unpack_tiles.  The last lane runs to n_blocks and checks S_f = 1 + bits/8 (Terse.hpp:547).
//   kernels       k_seg_frames / k_seg_listed: a frame of up to 32 K blocks is ONE wavefront (rounds, prefix sum, write pass in one
//                 go; a lane that walks its segment again stops where it meets its walk before, SegMerge).  k_seg_wg: a
//                 header-dense large frame is one workgroup of 2 / 4 / 8 wavefronts, the links between them in LDS.  k_seg_round /
//                 k_seg_resolve / k_seg_write (launch_seg_multi) and the same as one persistent launch (k_seg_fallback): large
//                 frames with runs to start in, many wavefronts per frame, the links between them through memory.
//
// Stream access: each lane reads its own segment, so the bytes a wavefront needs at any moment are 64 separate
// 128-byte pieces.  They are fetched cooperatively -- eight lanes load one segment's piece as 8 x 16 bytes, so a
// load instruction covers eight full cache lines -- into a per-lane LDS window (position based: window t of lane j
// holds the bits [X_j + A t, X_j + A (t+1) + lookahead), A = kSegAdv), the next window is prefetched into registers while
// the current one is walked (windows of 256 bytes advancing by 1792 bits since round 6).  A run of zero-width blocks (header bits 1, 1 bit per block: empty detector
// regions) is consumed 32 blocks per step.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include "unpack_tile.hpp"
#include "walk_lds.hpp"
#include "seg_common.hpp"
#include <stdlib.h>

namespace trpx {

// (SegCtx, seg_walk, seg_comb_guess, seg_ctx, ...: seg_common.hpp)

// Segment states of one frame (device memory, seg_workspace_bytes()).
struct SegState {
    uint64_t* in;        // IN state of every segment
    uint64_t* out;       // OUT state
    uint32_t* cnt;       // blocks counted; bit 31: the IN state is still the lane's own run guess ("strong"); bit 30: IN changed, OUT / count are stale
    uint32_t* bnd;       // boundary B_j (X_j unless a run guess moved it)
    uint32_t* open;      // per wave: a link inside the wave is still open
    uint32_t* wtot;      // per wave: blocks counted by its 64 segments
    uint32_t* wbase;     // per wave: blocks in front of it (k_seg_resolve)
};

// Fix-point rounds of wave k of a frame (segments 64 k .. 64 k + 63).  `first`: no earlier launch has left states
// behind (start from the guesses).  `lane0_true`: the wave's first IN state is the true one (wave 0, or a re-run by
// k_seg_resolve) or at least the best there is (second launch: wave k-1's last OUT state).
//
// A lane whose IN state differs from its predecessor's OUT state (an open link) has to decide whom to believe.
// A lane without a run guess always takes the predecessor's state.  A lane WITH one ("strong") takes it when the
// predecessor can be trusted -- every link up to it is closed and lane 0 is true, or at least the predecessor's own
// IN state was hit by the chain before it -- and otherwise only tries it: it walks from it and keeps it if that chain
// ends where its own did (merged: nothing downstream changes), else it returns to its guess and waits.
// The first open link of a wave with a true lane 0 is always taken, so the loop ends with all links closed there.
__device__ __forceinline__ void seg_fixpoint(const SegCtx& c, uint32_t* __restrict__ win, uint32_t k, uint32_t jl, bool first,
                                             bool lane0_true, int max_rounds, const SegState& st, bool run_guess = true,
                                             uint32_t* __restrict__ ck = nullptr) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t j = 64u * k + lane;
    // merge stop (SegMerge, seg_common.hpp): `ck` = 64 x kSegCk words of LDS for this wavefront's checkpoints
    const uint32_t ck_win = (c.L + 1024u + kSegAdv - 1u) / kSegAdv + 1u;
#ifdef TRPX_SEG_NO_MERGE                                   // (A/B build: tools/r6_segvariant.sh nomerge -DTRPX_SEG_NO_MERGE)
    const uint32_t ck_every = 0u;
#else
    const uint32_t ck_every = ck != nullptr && c.L + 2048u < (1u << 15) ? (ck_win + kSegCk - 1u) / kSegCk : 0u;
#endif
    bool has_ck = false;
    const bool walks = j < jl;                                 // lane jl and the lanes behind it own no counted blocks
    uint64_t in = seg_pack(j * c.L, 0u), out = 0ull;
    uint32_t cnt = 0u, B = j * c.L;
    bool strong = false, dirty = walks;                        // dirty: (in, out) do not belong together yet -- the lane has to walk
    if (!first) {
        in = st.in[j]; out = st.out[j]; B = st.bnd[j];
        const uint32_t cs = st.cnt[j];
        cnt = cs & 0x3FFFFFFFu; strong = (cs >> 31) != 0u; dirty = ((cs >> 30) & 1u) != 0u && walks;   // (left dirty by a capped launch)
    } else if (run_guess && __ballot(walks)) {                 // run-dominated streams: start inside a run
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
        if (gs != ~0ull && lane > 0u && walks) { in = gs; B = (uint32_t)gs; strong = true; }   // (lane 0: the next wave's lane 63 ends at X)
        __builtin_amdgcn_wave_barrier();
    }
    if (j == 0u) { in = 0ull; B = 0u; strong = false; }       // the frame starts with width 0 at bit 0 (Terse.hpp:359, :505)
    if (lane == 0u && k > 0u && !first) {
        const uint64_t ni = __hip_atomic_load(&st.out[j - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ni != in) { in = ni; strong = false; dirty = walks; }
    }
    uint32_t endB = (uint32_t)__shfl_down((int)B, 1, 64);
    if (lane == 63u) endB = (j + 1u) * c.L;
    bool tent = false;
    uint64_t sav_in = 0ull, sav_out = 0ull, rej = ~0ull;
    uint32_t sav_cnt = 0u;
    for (int iter = 0;; ++iter) {
        if (__ballot(dirty)) {
            if (iter >= max_rounds) break;                     // capped: what is still dirty / open is left to the next launch or to k_seg_resolve
#ifdef TRPX_SEG_STATS
            if (lane == 0u) { atomicAdd(c.stat + 2, 1u); atomicAdd(c.stat + 4, (uint32_t)__builtin_popcountll(__ballot(dirty))); }
#endif
            uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = 0u;
            bool bad = false;
#ifdef TRPX_SEG_STAMPS
            ++c.clk_rounds;
#endif
            SegMerge mg{ck + lane, ck_every, has_ck, 0u, false, 0u, 0u};
            if (ck_every) seg_walk<false, SegMerge>(c, win, 64u * k, dirty, endB, false, pos, w, n, nullptr, nullptr, bad, nullptr, &mg);
            else seg_walk<false>(c, win, 64u * k, dirty, endB, false, pos, w, n, nullptr, nullptr, bad);
            if (dirty) {
                // a walk that met the lane's walk before ends where that one did, with its blocks from there on
                const uint64_t o = mg.merged ? out : seg_pack(pos, w);
                const uint32_t total = mg.merged ? n + (cnt - mg.n_old) : n;
                bool keep = true;
                if (tent) {
                    if (o == sav_out) { out = o; cnt = total; strong = false; }      // merged: the predecessor's state is as good as mine
                    else { rej = in; in = sav_in; out = sav_out; cnt = sav_cnt; keep = false; }   // back to the run guess
                } else { out = o; cnt = total; }
                if (ck_every) {
                    // the entries are to describe the chain the lane now holds: behind a merge the old ones, renumbered; entries the
                    // walk did not reach, or those of a chain that was turned down, go
                    const uint32_t shift = (n - mg.n_old) << 17;
                    for (uint32_t i = 0; i < kSegCk; ++i) {
                        if (!keep) ck[64u * i + lane] = 0u;
                        else if (mg.merged) { if (i >= mg.at) { const uint32_t v = ck[64u * i + lane]; if (v) ck[64u * i + lane] = v + shift; } }
                        else if (!((mg.wrote >> i) & 1u)) ck[64u * i + lane] = 0u;
                    }
                    has_ck = keep;
                }
            }
            dirty = false; tent = false;
        }
        // the links: who has to give up its IN state for its predecessor's OUT state?
        const uint64_t prev = seg_shfl_up1(out);
        const bool conflict = lane > 0u && j <= jl && prev != in;
        const uint64_t closed = __ballot(!conflict);
        const uint32_t first_open = ~closed ? (uint32_t)__builtin_ctzll(~closed) : 64u;
        const bool pred_ver = lane0_true && lane <= first_open;
        const bool pred_link = lane >= 2u ? ((closed >> (lane - 1u)) & 1ull) != 0ull : lane0_true;
        const bool trusted = !strong || !walks || pred_ver || pred_link;
        if (conflict && trusted) { in = prev; strong = false; dirty = walks; }
        else if (conflict && prev != rej) { sav_in = in; sav_out = out; sav_cnt = cnt; in = prev; tent = true; dirty = true; }
        if (!__ballot(dirty)) break;
    }
    if (tent) { in = sav_in; out = sav_out; cnt = sav_cnt; dirty = false; }             // (capped in the middle of a try: back to the guess)
    // anything still dirty or open is reported; a dirty lane is walked first thing by the next call
    const uint64_t prev = seg_shfl_up1(out);
    const bool still_open = (lane > 0u && j <= jl && prev != in) || dirty;
    const uint64_t any_open = __ballot(still_open);
    st.in[j] = in;
    st.cnt[j] = (walks ? cnt & 0x3FFFFFFFu : 0u) | (strong ? 0x80000000u : 0u) | (dirty ? 0x40000000u : 0u);
    st.bnd[j] = B;
    __hip_atomic_store(&st.out[j], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t total = wave_inclusive_scan(walks ? cnt & 0x3FFFFFFFu : 0u);
    if (lane == 63u) { st.open[k] = any_open ? 1u : 0u; st.wtot[k] = total; }
}

// Write pass of wave k: `base` = blocks in front of the lane's segment, `end` = the next lane's boundary.
__device__ __forceinline__ void seg_write(const SegCtx& c, uint32_t* __restrict__ win, uint32_t k, uint32_t jl, uint64_t in,
                                          uint64_t next_in, uint32_t end, uint32_t base, uint8_t* __restrict__ wf,
                                          uint64_t* __restrict__ tf, uint32_t S_bytes, uint32_t* __restrict__ status) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t j = 64u * k + lane;
    const bool part = j <= jl, last = j == jl;
    uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = base;
    bool bad = part && base > c.n_blocks;
    seg_walk<true>(c, win, 64u * k, part && !bad, last ? 0xFFFFFFFFu : end, last, pos, w, n, wf, tf, bad);
    if (part && !last && seg_pack(pos, w) != next_in) bad = true;                       // the chain the counts came from
    if (last && !(n == c.n_blocks && pos <= c.limit && 1u + pos / 8u == S_bytes)) bad = true;   // S_f = 1 + bits/8 (Terse.hpp:547)
    if (__ballot(bad) && lane == 0u) atomicMax(&status[0], 5u);                         // TRPX_ERR_CORRUPT
}

struct SegWs {            // carve of seg_workspace_bytes()
    uint64_t *in, *out;
    uint32_t *cnt, *bnd, *open, *wtot, *wbase, *fallback;
};
__host__ __device__ inline SegWs seg_carve(void* ws, size_t n_frames, uint32_t K) {
    const size_t segs = n_frames * (size_t)K * kWave;
    SegWs s;
    s.in = reinterpret_cast<uint64_t*>(ws);
    s.out = s.in + segs;
    s.cnt = reinterpret_cast<uint32_t*>(s.out + segs);
    s.bnd = s.cnt + segs;
    s.open = s.bnd + segs;
    s.wtot = s.open + n_frames * (size_t)K;
    s.wbase = s.wtot + n_frames * (size_t)K;
    s.fallback = s.wbase + n_frames * (size_t)K;
    return s;
}
__device__ __forceinline__ SegState seg_state(const SegWs& w, uint64_t frame, uint32_t K) {
    const uint64_t so = frame * K * kWave;
    return SegState{w.in + so, w.out + so, w.cnt + so, w.bnd + so, w.open + frame * K, w.wtot + frame * K, w.wbase + frame * K};
}

// ---- one wavefront per frame (G = 64): rounds, prefix sum and write pass in one go ---------------------------------------
__device__ __forceinline__ void seg_frame_walk(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                               const uint64_t* __restrict__ frame_offsets, const FrameGeom& g, uint32_t max_w,
                                               const SegWs& ws, uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                               uint64_t frame, uint32_t* __restrict__ win, uint32_t* __restrict__ status,
                                               bool run_guess = true, uint32_t* __restrict__ ck = nullptr) {
    const uint32_t lane = (uint32_t)lane_id();
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, kWave, status)) {
        if (lane == 0) atomicMax(&status[0], 5u);
        return;
    }
    uint8_t* wf = widths + frame * g.n_blocks;
    uint64_t* tf = tile_off + frame * g.n_tiles;
    seg_zero_widths(wf, g.n_blocks, 0u, 1u);
    const SegState st = seg_state(ws, frame, 1u);
    const uint32_t jl = seg_last_live(c.limit, c.L, kWave);
#ifdef TRPX_SEG_STAMPS
    const uint64_t t_a = __builtin_amdgcn_s_memrealtime();
#endif
    seg_fixpoint(c, win, 0u, jl, true, true, 70, st, run_guess, ck);
    __builtin_amdgcn_s_waitcnt(0);                             // the zeroes are in L2 before the write pass stores widths
#ifdef TRPX_SEG_STAMPS
    const uint64_t t_b = __builtin_amdgcn_s_memrealtime();
#endif
    const uint64_t in = st.in[lane];
    const uint32_t cnt = st.cnt[lane] & 0x3FFFFFFFu;
    const uint64_t next_in = (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)in, 1, 64) |
                             ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(in >> 32), 1, 64) << 32);
    const uint32_t end = (uint32_t)__shfl_down((int)st.bnd[lane], 1, 64);
    const uint32_t base = wave_inclusive_scan(cnt) - cnt;
    seg_write(c, win, 0u, jl, in, next_in, end, base, wf, tf, c.limit / 8u, status);
#ifdef TRPX_SEG_STAMPS
    if (lane == 0u) {      // tools/seg_time.py passes a status block of 16 + 8 * frames words: per-frame start / rounds / write ticks (100 MHz)
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t t_c = __builtin_amdgcn_s_memrealtime();
        status[16 + 8 * frame + 0] = (uint32_t)t_a;
        status[16 + 8 * frame + 1] = (uint32_t)(t_b - t_a);
        status[16 + 8 * frame + 2] = (uint32_t)(t_c - t_b);
        status[16 + 8 * frame + 3] = (uint32_t)c.clk_wait[0];
        status[16 + 8 * frame + 4] = (uint32_t)c.clk_step[0];
        status[16 + 8 * frame + 5] = (uint32_t)c.clk_guess;
        status[16 + 8 * frame + 6] = (uint32_t)c.clk_wait[1];
        status[16 + 8 * frame + 7] = (uint32_t)c.clk_step[1];
    }
#endif
}

__global__ __launch_bounds__(kWave) void k_seg_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                      const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                      SegWs ws, uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                      uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    seg_frame_walk(terse, terse_bytes, frame_offsets, g, max_w, ws, widths, tile_off, blockIdx.x, win, status);
}

// The frames k_decode_frames listed (list[0] = count, list[1 + i] = frame): one wavefront each, four to a workgroup (the
// grid is launched for every frame of the stack and is normally empty: fewer, larger workgroups exit faster).
__global__ __launch_bounds__(kWave * kSegWgWaves) void k_seg_listed(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                         const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                         SegWs ws, uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                         const uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kSegWgWaves][kWave * kSegRow];
    __shared__ uint32_t ck[kSegWgWaves][kWave * kSegCk];
    const uint32_t i = blockIdx.x * kSegWgWaves + (uint32_t)wave_id();
    if (i >= list[0]) return;
    const uint32_t entry = list[1 + i];                        // bit 31: a width change every third block and more -- no run to look for
    seg_frame_walk(terse, terse_bytes, frame_offsets, g, max_w, ws, widths, tile_off, entry & 0x7FFFFFFFu, win[wave_id()], status,
                   (entry >> 31) == 0u, ck[wave_id()]);
}

// ---- the index from recorded group states (row f1 for files): the write pass alone --------------------------------------------
// The chain state at every 256th block (trpx_index_group_states / the group_bit_offsets header attribute) makes every group a
// segment with a verified IN state: one lane per group walks it by count with the write pass's windows and steps, 64 consecutive
// groups of a frame to a wavefront.  Every group is checked against its successor's state, the last one against S_f = 1 + bits/8
// (Terse.hpp:547).  (k_walk_groups, decode_fast.hip, reads its headers straight from global memory: ~0.85 us per dependent step
// against ~0.17 here; it stays for frames of 2^32 bits and more.)
__global__ __launch_bounds__(kWave * kSegWgWaves) void k_seg_groups(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                         const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                         uint32_t n_frames, const uint64_t* __restrict__ states,
                                                         uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                         uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kSegWgWaves][kWave * kSegRow];
    constexpr uint64_t kOffMask = (1ull << 40) - 1;
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t wpf = (g.n_tiles + (uint32_t)kWave - 1u) / (uint32_t)kWave;       // wavefronts per frame
    const uint64_t wv = (uint64_t)blockIdx.x * kSegWgWaves + (uint32_t)wave_id();
    const uint64_t frame = wv / wpf;
    if (frame >= n_frames) return;
    const uint32_t k = (uint32_t)(wv % wpf) * (uint32_t)kWave + lane;                // this lane's group
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, kWave, status)) {
        if (lane == 0) atomicMax(&status[0], 5u);
        return;
    }
    uint8_t* wf = widths + frame * g.n_blocks;
    uint64_t* tf = tile_off + frame * g.n_tiles;
    const bool part = k < g.n_tiles;
    const uint64_t st = part ? states[frame * g.n_tiles + k] : 0ull;
    const uint32_t b0 = k * (uint32_t)kTileBlocks;
    const uint32_t b1 = part ? (b0 + (uint32_t)kTileBlocks < g.n_blocks ? b0 + (uint32_t)kTileBlocks : g.n_blocks) : b0;
    uint32_t pos = (uint32_t)(st & kOffMask), w = (uint32_t)(st >> 40), n = b0;
    bool bad = part && ((k == 0u && st != 0ull) || (st & kOffMask) > (uint64_t)c.limit || w > max_w);   // (a frame starts at bit 0 with width 0)
    // the general step leaves empty blocks to the zeroes: this wavefront's stretch of the width array first
    {
        const uint32_t z0 = (uint32_t)(wv % wpf) * (uint32_t)kWave * (uint32_t)kTileBlocks;
        const uint32_t z1 = z0 + (uint32_t)(kWave * kTileBlocks) < g.n_blocks ? z0 + (uint32_t)(kWave * kTileBlocks) : g.n_blocks;
        const uint64_t a0 = (uint64_t)(uintptr_t)(wf + z0);
        const uint32_t head = (uint32_t)((16u - (a0 & 15u)) & 15u) < z1 - z0 ? (uint32_t)((16u - (a0 & 15u)) & 15u) : z1 - z0;
        const uint32_t n16 = (z1 - z0 - head) / 16u;
        if (lane < head) wf[z0 + lane] = 0;
        if (z0 + head + 16u * n16 + lane < z1) wf[z0 + head + 16u * n16 + lane] = 0;                     // < 16 bytes
        seg_u4* q = reinterpret_cast<seg_u4*>(wf + z0 + head);
        const seg_u4 z = {0u, 0u, 0u, 0u};
        for (uint32_t i = lane; i < n16; i += kWave) q[i] = z;
        __builtin_amdgcn_s_waitcnt(0);
    }
    const SegOrigin org{pos, b1};
    seg_walk<true>(c, win[wave_id()], 0u, part && !bad && b1 > b0, 0xFFFFFFFFu, true, pos, w, n, wf, tf, bad, &org);
    if (part && !bad) {
        if (b1 < g.n_blocks) bad = ((uint64_t)pos | ((uint64_t)w << 40)) != states[frame * g.n_tiles + k + 1u];   // lands in the next group's state
        else bad = !(n == g.n_blocks && pos <= c.limit && 1u + pos / 8u == c.limit / 8u);                          // S_f
    }
    if (__ballot(bad) && lane == 0u) atomicMax(&status[0], 5u);                          // TRPX_ERR_CORRUPT
}

hipError_t launch_seg_groups(const DecodeArgs& a, uint32_t max_w, const uint64_t* states, hipStream_t st) {
    const uint64_t waves = (uint64_t)a.n_frames * ((a.geom.n_tiles + kWave - 1) / kWave);
    hipLaunchKernelGGL(k_seg_groups, dim3((uint32_t)((waves + kSegWgWaves - 1) / kSegWgWaves)), dim3(kWave * kSegWgWaves), 0, st, a.terse, (uint64_t)a.terse_bytes,
                       a.frame_offsets, a.geom, max_w, a.n_frames, states, a.widths, a.tile_off, a.status);
    return hipGetLastError();
}

// Tiles of the listed frames of more than 32 K blocks, after launch_seg_multi has written their index: a fixed grid strides over
// (listed frame, tile) pairs.  (Frames of one wavefront's worth go back to the per-frame decoder with the widths given -- 0.20
// against 0.31 ms for 2000 listed 512^2 frames --; a large frame is the other way round: its extraction by one workgroup's three
// waves is a 0.45 ms chain of 455 groups for 1024^2 pixels, its tiles spread over the GPU take 0.05 ms.)
template <typename T>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 4 : 5) void k_unpack_listed(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               const uint8_t* __restrict__ widths,
                                                               const uint64_t* __restrict__ tile_off,
                                                               const uint32_t* __restrict__ list, T* __restrict__ pixels_out,
                                                               uint32_t* __restrict__ status) {
    __shared__ uint32_t s_image[unpack_image_dwords<T>()];
    __shared__ uint32_t s_wtot[unpack_sub_tiles<T>() * 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[unpack_stage_dwords<T>()];
    const uint32_t count = list[0];
    if (count == 0u || status[0] != 0u) return;
    constexpr uint32_t tb = unpack_sub_tiles<T>() * kThreads;
    const uint32_t tpf = (g.n_blocks + tb - 1) / tb;
    const uint64_t total = (uint64_t)count * tpf;
    for (uint64_t i = blockIdx.x; i < total; i += gridDim.x) {
        const uint32_t frame = list[1 + (uint32_t)(i / tpf)] & 0x7FFFFFFFu;
        if (!unpack_tile<T>(terse, terse_bytes, frame_offsets, g, frame, (uint32_t)(i % tpf), widths, tile_off, pixels_out, status,
                            s_image, s_wtot, s_stage))
            return;
        __syncthreads();
    }
}

// A frame of the launch: slot itself, or -- list given (the frames k_decode_frames listed) -- list[1 + slot] (bit 31: header-dense,
// see k_seg_listed); false: the slot is behind the list's end.
__device__ __forceinline__ bool seg_listed_frame(const uint32_t* __restrict__ list, uint64_t slot, uint64_t& frame, bool& run_guess) {
    frame = slot; run_guess = true;
    if (!list) return true;
    if (slot >= list[0]) return false;
    const uint32_t e = list[1u + slot];
    frame = e & 0x7FFFFFFFu; run_guess = (e >> 31) == 0u;
    return true;
}

// ---- several wavefronts per frame (large frames): rounds / resolve / write are separate launches -------------------------
// (item = slot * K + k: wavefront k of the frame in slot `slot` of the launch -- see seg_listed_frame)
__device__ __forceinline__ void seg_round_item(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                               const uint64_t* __restrict__ frame_offsets, const FrameGeom& g, uint32_t max_w,
                                               uint32_t K, uint32_t first, const SegWs& ws, uint8_t* __restrict__ widths,
                                               const uint32_t* __restrict__ list, uint32_t* __restrict__ status, uint32_t item,
                                               uint32_t* __restrict__ win) {
    uint64_t frame;
    bool run_guess;
    if (!seg_listed_frame(list, item / K, frame, run_guess)) return;
    if (list && !run_guess) return;                            // (a listed header-dense frame: k_seg_wg's)
    const uint32_t k = item % K;
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave, status)) {
        if (threadIdx.x == 0 && k == 0) atomicMax(&status[0], 5u);
        return;
    }
    if (first) seg_zero_widths(widths + frame * g.n_blocks, g.n_blocks, k, K);
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    // (rounds per launch: 3 / 5 / 8 / 12 / 24 -> 37 (frames left to the serial walk) / 1.09 / 0.81 / 0.73 / 0.80 ms for eight 4096^2 frames)
#ifdef TRPX_SEG_STAMPS
    const uint64_t t_a = __builtin_amdgcn_s_memrealtime();
#endif
    // (the first launch, where no wavefront but the frame's first knows its lane 0 is right, stops after 4 rounds: what is open then
    // closes faster with the links of the second launch -- 2 / 3 / 4 / 6 / 8 / 12 rounds: 0.58 / 0.59 / 0.46 / 0.50 / 0.50 / 0.54 ms)
    seg_fixpoint(c, win, k, jl, first != 0u, k == 0u || first == 0u, first ? 4 : 12, seg_state(ws, frame, K), run_guess);
#ifdef TRPX_SEG_STAMPS
    if (first && threadIdx.x == 0) {                            // tools/c4_time.py (SEG_PER_WAVE=1): status block of 16 + 8 * waves words
        uint32_t* o = status + 16 + 8 * item;
        o[0] = (uint32_t)t_a; o[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_a);
        o[3] = (uint32_t)c.clk_wait[0]; o[4] = (uint32_t)c.clk_step[0]; o[5] = (uint32_t)c.clk_guess; o[6] = c.clk_rounds;
    }
#endif
}

__global__ __launch_bounds__(kWave) void k_seg_round(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                     uint32_t K, uint32_t first, SegWs ws, uint8_t* __restrict__ widths,
                                                     const uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    seg_round_item(terse, terse_bytes, frame_offsets, g, max_w, K, first, ws, widths, list, status, blockIdx.x, win);
}

// One wavefront per frame: closes the links the rounds left open -- between the frame's waves and inside them --
// serially, wave by wave (each re-run starts from a verified state), and turns the block counts into block bases.
// A frame with too many open waves is left to the serial walk (fallback flag).
__device__ __forceinline__ void seg_resolve_item(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                 const uint64_t* __restrict__ frame_offsets, const FrameGeom& g, uint32_t max_w,
                                                 uint32_t K, const SegWs& ws, const uint32_t* __restrict__ list,
                                                 uint32_t* __restrict__ status, uint32_t slot, uint32_t* __restrict__ win) {
    uint64_t frame;
    bool run_guess;
    if (!seg_listed_frame(list, slot, frame, run_guess)) return;
    if (list && !run_guess) return;                            // (k_seg_wg's)
    const uint32_t lane = (uint32_t)lane_id();
    if (lane == 0) ws.fallback[frame] = 0u;
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave, status)) return;    // (reported by k_seg_round)
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    const SegState st = seg_state(ws, frame, K);
    const uint32_t rerun_limit = 2u + K / 16u;
    uint32_t running = 0u, reruns = 0u;
    for (uint32_t k0 = 0; k0 < K && 64u * k0 <= jl; k0 += kWave) {            // 64 of the frame's waves per step, one per lane
        const uint32_t kk = k0 + lane;
        const bool valid = kk < K && 64u * kk <= jl;
        for (uint32_t from = k0;;) {                                           // links at or behind wave `from` may have changed
            bool need = false;
            if (valid && kk >= from) {
                need = __hip_atomic_load(&st.open[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (kk > 0u) {
                    const uint64_t a = __hip_atomic_load(&st.in[64u * kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t b = __hip_atomic_load(&st.out[64u * kk - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    need = need || a != b;
                }
            }
            const uint64_t mask = __ballot(need);
            if (!mask) break;
            if (++reruns > rerun_limit) {
                if (lane == 0) ws.fallback[frame] = 1u;
                return;
            }
            const uint32_t k = k0 + (uint32_t)__builtin_ctzll(mask);           // the first open one: its predecessors are final
            seg_fixpoint(c, win, k, jl, false, true, 70, st);
            __builtin_amdgcn_s_waitcnt(0);
            __threadfence();
            from = k + 1u;
        }
        const uint32_t tot = valid ? __hip_atomic_load(&st.wtot[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t inc = wave_inclusive_scan(tot);
        if (valid) st.wbase[kk] = running + inc - tot;
        running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
}

__global__ __launch_bounds__(kWave) void k_seg_resolve(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                       uint32_t K, SegWs ws, const uint32_t* __restrict__ list,
                                                       uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    seg_resolve_item(terse, terse_bytes, frame_offsets, g, max_w, K, ws, list, status, blockIdx.x, win);
}

__device__ __forceinline__ void seg_write_item(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                               const uint64_t* __restrict__ frame_offsets, const FrameGeom& g, uint32_t max_w,
                                               uint32_t K, const SegWs& ws, uint8_t* __restrict__ widths,
                                               uint64_t* __restrict__ tile_off, const uint32_t* __restrict__ list,
                                               uint32_t* __restrict__ status, uint32_t item, uint32_t* __restrict__ win) {
    uint64_t frame;
    bool run_guess;
    if (!seg_listed_frame(list, item / K, frame, run_guess)) return;
    if (list && !run_guess) return;                            // (k_seg_wg's)
    const uint32_t k = item % K;
    const uint32_t lane = (uint32_t)lane_id();
    if (ws.fallback[frame]) return;                            // the serial walk does this frame
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave, status)) return;
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    if (64u * k > jl) return;
    const SegState st = seg_state(ws, frame, K);
    const uint32_t j = 64u * k + lane;
    const bool has_next = j + 1u < K * kWave;
    const uint32_t cnt = st.cnt[j] & 0x3FFFFFFFu;
    const uint32_t base = st.wbase[k] + wave_inclusive_scan(cnt) - cnt;
    seg_write(c, win, k, jl, st.in[j], has_next ? st.in[j + 1u] : 0ull, has_next ? st.bnd[j + 1u] : 0xFFFFFFFFu, base,
              widths + frame * g.n_blocks, tile_off + frame * g.n_tiles, c.limit / 8u, status);
}

__global__ __launch_bounds__(kWave) void k_seg_write(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                     uint32_t K, SegWs ws, uint8_t* __restrict__ widths,
                                                     uint64_t* __restrict__ tile_off, const uint32_t* __restrict__ list,
                                                     uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    seg_write_item(terse, terse_bytes, frame_offsets, g, max_w, K, ws, widths, tile_off, list, status, blockIdx.x, win);
}

// ---- the same six launches as ONE, for a list that is normally empty -------------------------------------------------------------
// Behind the routes that handle large frames by themselves (decode_part.hip) the position-parallel walk is the fallback for the
// frames they list -- normally none, and then launch_seg_multi's six launches are six empty grids of a wavefront per (frame,
// segment group): 4.6 - 4.9 us each, 28 us = 7 - 14 % of such a decode.  k_seg_fallback is a small persistent grid instead:
// every workgroup reads the list's count and leaves if it is zero (one launch: 4 us); otherwise the grid strides over the items
// of each phase -- rounds x 3, resolve, write, the serial walk of frames that did not converge -- with a device-wide barrier
// between them (a counter every workgroup adds to and waits for, the waits bounded like the encoder's look-back; all kSegFbGrid
// workgroups are resident at once: 64 threads, 16.6 KB of LDS each).  Results are what the six launches produce.
constexpr uint32_t kSegFbGrid = 1024;            // at most; the launch asks the device how many of these workgroups it holds at once (seg_fallback_grid)
__device__ __forceinline__ bool seg_grid_barrier(uint64_t* __restrict__ ctr, uint32_t& epoch) {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    ++epoch;
    bool ok = true;
    if (lane_id() == 0) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t want = (uint64_t)epoch * gridDim.x;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(8);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { ok = false; break; }   // 1 s of the 100 MHz counter: a workgroup that never arrives
        }
    }
    ok = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");    // (uniform loads of what other workgroups wrote go through the scalar cache)
    return ok;
}
__global__ __launch_bounds__(kWave) void k_seg_fallback(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                        const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                        uint32_t K, SegWs ws, uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                        const uint32_t* __restrict__ list, uint64_t* __restrict__ barrier,
                                                        uint32_t* __restrict__ status) {
    __shared__ uint32_t s_lds[kWalkChunkDw + 4 > kWave * kSegRow ? kWalkChunkDw + 4 : kWave * kSegRow];
    const uint32_t count = __hip_atomic_load(&list[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count == 0u) return;                                                  // the normal case
    if ((uint64_t)count == __hip_atomic_load(barrier - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // every listed frame is header-dense: k_seg_wg's
    const uint32_t items = count * K;
    uint32_t epoch = 0;
    bool ok = true;
    for (uint32_t phase = 0; phase < 3u && ok; ++phase) {
        for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
            seg_round_item(terse, terse_bytes, frame_offsets, g, max_w, K, phase == 0u ? 1u : 0u, ws, widths, list, status, it, s_lds);
            __builtin_amdgcn_wave_barrier();
        }
        ok = seg_grid_barrier(barrier, epoch);
    }
    if (ok) {
        for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
            seg_resolve_item(terse, terse_bytes, frame_offsets, g, max_w, K, ws, list, status, it, s_lds);
            __builtin_amdgcn_wave_barrier();
        }
        ok = seg_grid_barrier(barrier, epoch);
    }
    if (ok) {
        for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
            seg_write_item(terse, terse_bytes, frame_offsets, g, max_w, K, ws, widths, tile_off, list, status, it, s_lds);
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {         // (the flags are the resolve phase's: no barrier needed in between)
            if ((list[1u + it] >> 31) != 0u) continue;                        // (k_seg_wg's)
            const uint64_t frame = list[1u + it] & 0x7FFFFFFFu;
            if (__hip_atomic_load(&ws.fallback[frame], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                walk_lds_frame(terse, terse_bytes, frame_offsets, g, max_w, widths, tile_off, frame, s_lds, status);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!ok && threadIdx.x == 0) atomicMax(&status[0], 7u);                   // TRPX_ERR_TIMEOUT: a device-wide barrier gave up
}

// ---- header-dense LARGE frames: one workgroup per frame ------------------------------------------------------------------------------
// The large-frame routes (decode_part.hip) walk a frame as a few thousand short parts, one serial walker each -- a step per run of
// equal widths, which on header-dense data (Poisson(3) counts: seven blocks of ten start with an explicit header) is a step per
// block: 0.12 us each, 370 us for 200 x (1030 x 1065) such frames.  The lane-per-segment walk takes 64 blocks per wavefront step of
// 0.1 us and pays for it with speculation (rounds: ~4.8 passes), as k_seg_listed shows on 512 x 512 frames.  Here a frame the
// routes' vote (ChainVote, decode_part.hip) calls header-dense gets ONE workgroup of W wavefronts: 64 W segments of some hundred blocks,
// counting rounds with the links between the wavefronts' edge lanes in LDS and a workgroup barrier per round, a prefix sum, the
// write pass -- no launches or device-wide barriers between the rounds, and a frame ends when ITS links are closed.  Plain guesses
// only (X_j, width 0): these frames have no runs to start in.  Every link is a lane's IN state against its predecessor's OUT state;
// lane 0 starts in the frame's true state, so closed links make every state the frame's chain's (and the write pass checks every
// segment's end against the next one's start, the last one against S_f, as everywhere).
// List entries without bit 31 are k_seg_fallback's.
#ifndef TRPX_SEG_WG_ROUNDS
#define TRPX_SEG_WG_ROUNDS 20
#endif
constexpr uint32_t kSegWgRounds = TRPX_SEG_WG_ROUNDS;         // (test build segwgrounds: 1 -- every frame handed back)
template <uint32_t W>
__global__ __launch_bounds__(kWave * W) void k_seg_wg(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                     uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                     uint32_t* __restrict__ list, uint64_t* __restrict__ n_dense,
                                                     uint32_t* __restrict__ status) {
    constexpr uint32_t G = kWave * W;
    constexpr bool kCk = W <= 4u;                              // (merge stop: 160 KB of LDS hold eight windows or the checkpoints)
    __shared__ uint32_t win[W][kWave * kSegRow];
    __shared__ uint32_t ckm[kCk ? W : 1u][kCk ? kWave * kSegCk : 1u];
    __shared__ uint64_t s_state[G];
    __shared__ uint32_t s_tot[W];
    if (__hip_atomic_load(n_dense, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) return;   // the normal case
    const uint32_t count = __hip_atomic_load(&list[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lane = (uint32_t)lane_id(), wv = (uint32_t)wave_id(), j = kWave * wv + lane;
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const uint32_t entry = list[1u + it];
        if ((entry >> 31) == 0u) continue;
        const uint64_t frame = entry & 0x7FFFFFFFu;
        SegCtx c;
        if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, G, status)) {
            if (threadIdx.x == 0) atomicMax(&status[0], 5u);
            continue;
        }
        uint8_t* wf = widths + frame * g.n_blocks;
        uint64_t* tf = tile_off + frame * g.n_tiles;
        seg_zero_widths(wf, g.n_blocks, wv, W);
        const uint32_t jl = seg_last_live(c.limit, c.L, G);
        const bool walks = j < jl;
        const uint32_t ck_win = (c.L + 1024u + kSegAdv - 1u) / kSegAdv + 1u;
        const uint32_t ck_every = kCk && c.L + 2048u < (1u << 15) ? (ck_win + kSegCk - 1u) / kSegCk : 0u;
        uint32_t* const ck = ckm[kCk ? wv : 0u];
        uint64_t in = j == 0u ? 0ull : seg_pack(j * c.L, 0u), out = 0ull;   // the frame starts with width 0 at bit 0 (Terse.hpp:359, :505)
        uint32_t cnt = 0u;
        bool dirty = walks, has_ck = false, gave_up = false;
        for (uint32_t round = 0;; ++round) {
            if (__ballot(dirty)) {
                uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = 0u;
                bool bad = false;
                SegMerge mg{ck + lane, ck_every, has_ck, 0u, false, 0u, 0u};
                if (ck_every) seg_walk<false, SegMerge>(c, win[wv], kWave * wv, dirty, (j + 1u) * c.L, false, pos, w, n, nullptr, nullptr, bad, nullptr, &mg);
                else seg_walk<false>(c, win[wv], kWave * wv, dirty, (j + 1u) * c.L, false, pos, w, n, nullptr, nullptr, bad);
                if (dirty) {
                    if (mg.merged) cnt = n + (cnt - mg.n_old);               // met the lane's walk before: its end, its blocks from there on
                    else { out = seg_pack(pos, w); cnt = n; }
                    if (ck_every) {                                          // (the entries describe the chain the lane now holds: see seg_fixpoint)
                        const uint32_t shift = (n - mg.n_old) << 17;
                        for (uint32_t i = 0; i < kSegCk; ++i) {
                            if (mg.merged) { if (i >= mg.at) { const uint32_t v = ck[64u * i + lane]; if (v) ck[64u * i + lane] = v + shift; } }
                            else if (!((mg.wrote >> i) & 1u)) ck[64u * i + lane] = 0u;
                        }
                        has_ck = true;
                    }
                }
                dirty = false;
            }
            s_state[j] = out;
            __syncthreads();
            const uint64_t prev = j > 0u ? s_state[j - 1u] : in;
            if (j > 0u && j <= jl && prev != in) { in = prev; dirty = walks; }   // an open link: the predecessor's state is the one to walk from
            if (!__syncthreads_or(dirty ? 1 : 0)) break;
            // Header-dense data closes its links in a handful of rounds (a false chain meets the frame's within a segment, seven
            // times in eight).  A frame that needs more was misjudged by its head -- run-dominated data, where a plain guess may not
            // merge for a whole segment and every round closes one link: 3.2 ms for 200 x (1030 x 1065) synth-v1 frames -- and goes
            // back to the list as a frame with runs to look for: k_seg_fallback, launched behind this kernel.
            if (round >= kSegWgRounds) { gave_up = true; break; }
        }
        if (gave_up) {
            if (threadIdx.x == 0) {
                list[1u + it] = entry & 0x7FFFFFFFu;
                __hip_atomic_fetch_add(n_dense, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            continue;
        }
        const uint32_t cw = walks ? cnt : 0u;
        const uint32_t inc = wave_inclusive_scan(cw);
        if (lane == 63u) s_tot[wv] = inc;
        s_state[j] = in;
        __builtin_amdgcn_s_waitcnt(0);                         // the zeroes are in L2 before the write pass stores widths
        __syncthreads();
        uint32_t base = inc - cw;
        for (uint32_t v = 0; v < wv; ++v) base += s_tot[v];
        const uint64_t next_in = j + 1u < G ? s_state[j + 1u] : 0ull;
        seg_write(c, win[wv], wv, jl, in, next_in, (j + 1u) * c.L, base, wf, tf, c.limit / 8u, status);
        __syncthreads();                                       // (LDS: the next frame's)
    }
}
#ifndef TRPX_SEG_WG_TARGET
#define TRPX_SEG_WG_TARGET 400
#endif
uint32_t seg_wg_waves(const FrameGeom& g) {                   // ~TRPX_SEG_WG_TARGET blocks per lane and less; eight wavefronts' windows are what the LDS holds
    return g.n_blocks <= 128u * TRPX_SEG_WG_TARGET ? 2u : (g.n_blocks <= 256u * TRPX_SEG_WG_TARGET ? 4u : 8u);
}

// Segments per frame: a multiple of 64.  Frames of up to 32 K blocks (512 x 512: 21 846) are one wavefront -- rounds, prefix
// sum and write pass in one launch, which is what the per-frame decoder's deferral needs; larger frames get segments of
// about kSegTargetBlocks blocks, i.e. enough wavefronts to fill the GPU even for a handful of frames (eight 4096 x 4096
// frames: 8 x 228 wavefronts; measured walk time for them with targets 320 / 160 / 96 / 64 blocks: 1.17 / 1.03 / 0.87 / 1.08 ms).
uint32_t seg_waves_per_frame(const FrameGeom& g, size_t n_frames) {
#ifdef TRPX_DIAGNOSTICS
    static const uint64_t kSegTargetBlocks = getenv("TRPX_SEG_TARGET") ? (uint64_t)atoi(getenv("TRPX_SEG_TARGET")) : 96;
#else
    constexpr uint64_t kSegTargetBlocks = 96;
#endif
    if (g.n_blocks <= single_part_blocks(n_frames)) return 1;
    const uint64_t k = ((uint64_t)g.n_blocks + 32 * kSegTargetBlocks) / (64 * kSegTargetBlocks);
    return (uint32_t)(k ? k : 1);
}
static size_t seg_state_bytes(const FrameGeom& g, size_t n_frames) {
    const size_t K = seg_waves_per_frame(g, n_frames), segs = n_frames * K * kWave;
    return align_up(segs * (8 + 8 + 4 + 4) + n_frames * K * 12 + n_frames * 4, 256);
}
size_t seg_workspace_bytes(const FrameGeom& g, size_t n_frames) {
    // behind the segment states: the scratch of the listed frames' dense walk (decode_dense.hip; frames of one wavefront's worth here)
    return seg_state_bytes(g, n_frames) + (seg_waves_per_frame(g, n_frames) == 1 ? dense_workspace_bytes(g, n_frames) : 0);
}
static bool g_dense_route = false;   // (off: measured slower than the rounds on Poisson(3) counts, see LABNOTES round 6; trpx_set_decode_path(6) / TRPX_DECODE_PATH=dense)
void set_dense_route(bool on) { g_dense_route = on; }

hipError_t launch_walk_lds_only(const DecodeArgs& a, uint32_t max_w, const uint32_t* only, hipStream_t st, const uint32_t* list = nullptr);   // decode_fast.hip

// k_seg_fallback's device-wide barriers need every workgroup of the grid resident at once: as many as the device holds (a
// partitioned or CU-masked device, another architecture: fewer than kSegFbGrid), at most kSegFbGrid; half of what fits, so that
// kernels of other streams -- the size gather of the sharded entry points -- leave room.  Cached per device.
static uint32_t seg_fallback_grid() {
    static thread_local int cached_dev = -1;
    static thread_local uint32_t cached = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 64u;
    if (dev == cached_dev && cached) return cached;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_seg_fallback, kWave, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 1;
    uint64_t fit = (uint64_t)per_cu * (uint64_t)cus / 2u;
#ifdef TRPX_SEG_FB_GRID
    fit = TRPX_SEG_FB_GRID;                                                    // (test build: a device that holds this many)
#endif
    cached = (uint32_t)(fit < 1u ? 1u : (fit > kSegFbGrid ? kSegFbGrid : fit));
    cached_dev = dev;
    return cached;
}

// Fills a.widths / a.tile_off (the decode index) from the stream.
// Several wavefronts per frame: every frame of the stack (list == nullptr) or the frames of a list.
static hipError_t launch_seg_multi(const DecodeArgs& a, uint32_t max_w, uint32_t K, const uint32_t* list, hipStream_t st) {
    const SegWs ws = seg_carve(a.seg_ws, a.n_frames, K);
    if (list && a.defer && list == a.defer) {          // a list that is normally empty: one launch (k_seg_fallback)
        // the barrier counter: the last word of the statistics slots in front of the list (codec_common.hpp), cleared with them by
        // the call's first launch and used by nothing else
        uint64_t* barrier = reinterpret_cast<uint64_t*>(a.defer) - 1;
        // first the listed frames of a stack the route's vote called header-dense (bit 31; their number: the word in front of the barrier's) --
        // k_seg_wg hands back what it cannot close --, then the others
        const uint32_t W = seg_wg_waves(a.geom);
        const uint32_t wg_grid = (uint32_t)(a.n_frames < (W == 8u ? 256u : 512u) ? a.n_frames : (W == 8u ? 256u : 512u));
        if (W == 2u)
            hipLaunchKernelGGL((k_seg_wg<2>), dim3(wg_grid), dim3(kWave * 2), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                               a.widths, a.tile_off, a.defer, barrier - 1, a.status);
        else if (W == 4u)
            hipLaunchKernelGGL((k_seg_wg<4>), dim3(wg_grid), dim3(kWave * 4), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                               a.widths, a.tile_off, a.defer, barrier - 1, a.status);
        else
            hipLaunchKernelGGL((k_seg_wg<8>), dim3(wg_grid), dim3(kWave * 8), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                               a.widths, a.tile_off, a.defer, barrier - 1, a.status);
        hipLaunchKernelGGL(k_seg_fallback, dim3(seg_fallback_grid()), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom,
                           max_w, K, ws, a.widths, a.tile_off, list, barrier, a.status);
        return hipGetLastError();
    }
    const dim3 grid((uint32_t)((size_t)a.n_frames * K));
    hipLaunchKernelGGL(k_seg_round, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, 1u, ws, a.widths, list, a.status);
    hipLaunchKernelGGL(k_seg_round, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, 0u, ws, a.widths, list, a.status);
    // (a third launch: what it leaves open, k_seg_resolve closes one wavefront at a time -- 0.22 ms for eight 4096^2 frames after
    // two launches, 0.02 ms after three, which cost 0.14 ms)
    hipLaunchKernelGGL(k_seg_round, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, 0u, ws, a.widths, list, a.status);
    hipLaunchKernelGGL(k_seg_resolve, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets,
                       a.geom, max_w, K, ws, list, a.status);
    hipLaunchKernelGGL(k_seg_write, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, ws, a.widths, a.tile_off, list, a.status);
    return launch_walk_lds_only(a, max_w, ws.fallback, st, list);    // frames that did not converge: the serial walk
}

hipError_t launch_seg_walk(const DecodeArgs& a, uint32_t max_w, hipStream_t st) {
    const uint32_t K = seg_waves_per_frame(a.geom, a.n_frames);
    if (K == 1) {
        const SegWs ws = seg_carve(a.seg_ws, a.n_frames, K);
        hipLaunchKernelGGL(k_seg_frames, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets,
                           a.geom, max_w, ws, a.widths, a.tile_off, a.status);
        return hipGetLastError();
    }
    return launch_seg_multi(a, max_w, K, nullptr, st);
}


// The decode index of the frames listed in a.defer: one wavefront per listed frame, or the launches of launch_seg_multi for
// frames of more than 32 K blocks.
hipError_t launch_seg_listed(const DecodeArgs& a, uint32_t max_w, hipStream_t st) {
    const uint32_t K = seg_waves_per_frame(a.geom, a.n_frames);
    if (K > 1) return launch_seg_multi(a, max_w, K, static_cast<const uint32_t*>(a.defer), st);
    if (g_dense_route)
        return launch_dense_listed(a, max_w, static_cast<char*>(a.seg_ws) + seg_state_bytes(a.geom, a.n_frames), static_cast<const uint32_t*>(a.defer), st);
    const SegWs ws = seg_carve(a.seg_ws, a.n_frames, 1u);
    hipLaunchKernelGGL(k_seg_listed, dim3((a.n_frames + kSegWgWaves - 1) / kSegWgWaves), dim3(kWave * kSegWgWaves), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets,
                       a.geom, max_w, ws, a.widths, a.tile_off, static_cast<const uint32_t*>(a.defer), a.status);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_decode_deferred_t(const DecodeArgs& a, hipStream_t st) {
    if (!a.index_given) {                                                   // (trpx_decode_indexed on frames that start inside a cache line: the index is the caller's)
        const hipError_t e0 = launch_seg_listed(a, (uint32_t)PixelTraits<T>::bits, st);
        if (e0 != hipSuccess) return e0;
    }
    if (seg_waves_per_frame(a.geom, a.n_frames) > 1u) {                     // large frames: their tiles, spread over the GPU
        constexpr uint32_t tb = unpack_sub_tiles<T>() * kThreads;
        const uint64_t tiles = (uint64_t)a.n_frames * ((a.geom.n_blocks + tb - 1) / tb);
        hipLaunchKernelGGL((k_unpack_listed<T>), dim3((uint32_t)(tiles < 1024 ? tiles : 1024)), dim3(kThreads), 0, st, a.terse,
                           (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, a.widths, a.tile_off, static_cast<const uint32_t*>(a.defer),
                           static_cast<T*>(a.pixels_out), a.status);
        return hipGetLastError();
    }
    // the listed frames' pixels: the per-frame decoder again, with the widths just written in place of its walker
    constexpr int dtype = PixelTraits<T>::bits == 8 ? (PixelTraits<T>::is_signed ? 1 : 0)
                          : PixelTraits<T>::bits == 16 ? (PixelTraits<T>::is_signed ? 3 : 2) : (PixelTraits<T>::is_signed ? 5 : 4);
    const hipError_t e = launch_decode_frames_indexed(dtype, a, static_cast<const uint32_t*>(a.defer), st);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

bool seg_single_wave(const FrameGeom& g, size_t n_frames) { return seg_waves_per_frame(g, n_frames) == 1; }

// Decodes the frames flagged in a.defer (see k_decode_frames); needs seg_single_wave(a.geom).
hipError_t launch_decode_deferred(int dtype, const DecodeArgs& a, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_decode_deferred_t<uint8_t>(a, st);
    case 1: return launch_decode_deferred_t<int8_t>(a, st);
    case 2: return launch_decode_deferred_t<uint16_t>(a, st);
    case 3: return launch_decode_deferred_t<int16_t>(a, st);
    case 4: return launch_decode_deferred_t<uint32_t>(a, st);
    case 5: return launch_decode_deferred_t<int32_t>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace trpx
