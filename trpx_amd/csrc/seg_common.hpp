// Shared device code of the position-parallel header walks (decode_seg.hip: rounds to a fix point inside a wavefront;
// decode_dense.hip: one speculative pass, pooled link walks, a write pass): segment context, the lane-per-segment window walk
// (counting / writing steps, hand-scheduled), the run guess.  See the head of decode_seg.hip for the scheme.
#pragma once
#include "codec_common.hpp"
#include "encode_kernels.hpp"

namespace trpx {

#ifndef TRPX_SEG_LOAD_DW
#define TRPX_SEG_LOAD_DW 64
#endif
constexpr uint32_t kSegLoadDw = TRPX_SEG_LOAD_DW;    // dwords loaded per lane and window: 32 (128 bytes, eight lanes x 16 bytes) or 64
constexpr uint32_t kSegPieces = kSegLoadDw / 32;     // 16-byte loads per lane and row
constexpr uint32_t kSegAdv = 32 * kSegLoadDw - 256;  // bits a window advances: 768 (1792)
//                                                   (127 (alignment) + advance + 44 (peek) bits <= the bits loaded, and a multiple of 128)
constexpr uint32_t kSegRow = kSegLoadDw + 4;      // LDS dwords per lane window (16-byte aligned rows)
constexpr uint32_t kSegWgWaves = kSegLoadDw > 32 ? 2 : 4;   // wavefronts per workgroup of the kernels that keep one window array per wavefront (64 KB of LDS per workgroup)
constexpr uint32_t kSegSpan = 864;                // window bits a first-round guess may use (boundaries move by < kSegSpan)
constexpr uint32_t kSegLiveMargin = 400 + kSegSpan;   // > longest block (12 + 12 * 32 bits) + boundary shift: see seg_last_live()

typedef uint32_t seg_u4 __attribute__((ext_vector_type(4)));

struct SegCtx {
    const uint32_t* s32;     // stream as dwords
    uint64_t n_dw;           // dwords that hold at least one stream byte
    uint64_t fa;             // absolute bit position of the frame's first bit
    uint32_t limit;          // 8 * S_f
    uint32_t L;              // segment length in bits (multiple of 128)
    uint32_t wsh;            // fa & 127: bit offset of a window's first wanted bit inside its 16-byte aligned load
    uint32_t n_blocks, nb_last, max_w;
    uint32_t* stat;          // status block (diagnostic build: [2] rounds, [3] wave steps, [4] lane walks)
#ifdef TRPX_SEG_STAMPS
    mutable uint64_t clk_wait[2] = {0, 0}, clk_step[2] = {0, 0}, clk_guess = 0;   // [WRITE]: window fill / stepping (100 MHz ticks)
    mutable uint32_t clk_rounds = 0;
#endif
};

__device__ __forceinline__ uint64_t seg_pack(uint32_t pos, uint32_t w) { return (uint64_t)pos | ((uint64_t)w << 32); }

// Last segment that owns blocks: the largest j with X_j + margin <= 8 (S_f - 1).  With the margin no block that
// starts before X_jl can be the frame's last (possibly partial) block, so the counting rounds may treat every
// block as 12 values; the blocks from X_jl on are walked by count in the write pass.
__device__ __forceinline__ uint32_t seg_last_live(uint32_t limit, uint32_t L, uint32_t G) {
    if (limit < 8u + kSegLiveMargin) return 0u;
    const uint32_t j = (limit - 8u - kSegLiveMargin) / L;
    return j < G - 1u ? j : G - 1u;
}

__device__ __forceinline__ uint32_t seg_len_bits(uint32_t limit, uint32_t G) {
    const uint32_t l = (limit + G - 1u) / G;
    const uint32_t r = (l + 127u) & ~127u;
    return r ? r : 128u;
}

// 16 stream bytes at dword index d (d % 4 == 0 when the buffer is 16-byte aligned; dword alignment is all the
// hardware needs); bytes past the end read as zero.
__device__ __forceinline__ seg_u4 seg_load16(const SegCtx& c, uint64_t d) {
    seg_u4 x;
    if (d + 4u <= c.n_dw) __builtin_memcpy(&x, c.s32 + d, 16);
    else {
        x.x = d < c.n_dw ? c.s32[d] : 0u; x.y = d + 1 < c.n_dw ? c.s32[d + 1] : 0u;
        x.z = d + 2 < c.n_dw ? c.s32[d + 2] : 0u; x.w = d + 3 < c.n_dw ? c.s32[d + 3] : 0u;
    }
    return x;
}

// One pass of a wavefront over its 64 segments seg0 .. seg0 + 63 (lanes with `part` walk, the others idle).
//   WRITE == false: count the block starts in [pos, end); leaves (pos, w) = OUT state, n = count.
//   WRITE == true : n is the global block index; stores widths / group offsets; a lane with by_count set stops
//                   at n == n_blocks (the frame's last blocks) instead of at `end`.
//   org (optional): where the lane's windows start and where a by_count lane stops, if not at the segment boundary (seg0 + lane) L
//                   and at n_blocks -- the walk of 256-block groups from their recorded start states (k_seg_groups).
struct SegOrigin {
    uint32_t X;          // bit position (inside the frame) of the lane's window 0
    uint32_t n_end;      // a by_count lane stops at n == n_end
};

// A caller's hook into a counting walk (decode_dense.hip: the link walks that go on across regions): open() when a window opens,
// close() when no lane is active in it any more -- it may change a lane's `end` / `done` and returns true for a lane that has
// more steps to take in this window.
struct SegNoHook {
    __device__ __forceinline__ void open(uint32_t, bool) {}
    __device__ __forceinline__ bool close(uint32_t, bool, uint32_t, uint32_t&, uint32_t&, uint32_t&, bool&, uint32_t&) { return false; }
};

// Merge stop of a repeated counting walk (seg_fixpoint, decode_seg.hip).  A lane that walks its segment a second time -- from its
// predecessor's OUT state instead of its guess -- is on a chain that merges with the one it walked before after a few dozen
// blocks (0.7-1.5 % per block on detector data) and IS that chain from there on: OUT state and the blocks behind the merge are
// known.  So every counting walk leaves the state in which it crosses a window boundary and its count up to there, 4 bytes in LDS
// per boundary (column `lane` of kSegCk rows of 64), and a later walk of the same lane that crosses a boundary in the state on
// record stops there.  Position relative to the boundary + 1 (0: no entry), width (counting walks hold widths up to 73),
// count: segments of fewer than 2^15 bits, so that no count overflows.
constexpr uint32_t kSegCk = 8;
__device__ __forceinline__ uint32_t seg_ck_pack(uint32_t rel, uint32_t w, uint32_t cnt) {
    return rel < 1023u && w < 128u && cnt < (1u << 15) ? (rel + 1u) | w << 10 | cnt << 17 : 0u;
}
struct SegMerge {
    uint32_t* ck;              // this lane's column (entry i at ck[64 i])
    uint32_t every;            // an entry per `every` window boundaries; 0: off
    bool cmp;                  // the entries are those of the lane's walk before
    uint32_t wrote;            // entries this walk has written
    bool merged;               // stopped at entry `at`, where the walk before had counted n_old blocks
    uint32_t at, n_old;
    __device__ __forceinline__ void open(uint32_t, bool) {}
    __device__ __forceinline__ bool close(uint32_t t, bool act0, uint32_t wend, uint32_t& pos, uint32_t& w, uint32_t& n, bool& done, uint32_t&) {
        if (every != 0u && act0 && !done && pos >= wend) {                     // crossed the boundary behind window t, not the segment's end
            const uint32_t b = t + 1u, q = b / every;
            if (q * every == b && q - 1u < kSegCk) {
                const uint32_t i = q - 1u, now = seg_ck_pack(pos - wend, w, n), old = ck[64u * i];
                if (cmp && now != 0u && ((old ^ now) & 0x1FFFFu) == 0u) { merged = true; at = i; n_old = old >> 17; done = true; }
                else { ck[64u * i] = now; wrote |= 1u << i; }
            }
        }
        return false;
    }
};

template <bool WRITE, class Hook = SegNoHook>
__device__ __forceinline__ void seg_walk(const SegCtx& c, uint32_t* __restrict__ win, uint32_t seg0, bool part, uint32_t end,
                                         bool by_count, uint32_t& pos, uint32_t& w, uint32_t& n,
                                         uint8_t* __restrict__ wf, uint64_t* __restrict__ tf, bool& bad, const SegOrigin* org = nullptr,
                                         Hook* hook = nullptr, uint32_t t_start = 0u) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t X = org ? org->X : (seg0 + lane) * c.L;
    const uint32_t wsh = org ? (uint32_t)((c.fa + X) & 127u) : c.wsh;      // bit offset of the window's first wanted bit in its 16-byte aligned load
    const uint32_t n_end = org ? org->n_end : c.n_blocks;
    const uint32_t oct = lane & ~7u, piece = lane & 7u;
    uint32_t Xo[8];                                            // window origins of the eight rows this lane helps to load
#pragma unroll
    for (int k = 0; k < 8; ++k) Xo[k] = org ? (uint32_t)__shfl((int)X, (int)(oct + k), 64) : (seg0 + oct + (uint32_t)k) * c.L;
    bool done = !part || (WRITE && by_count ? n >= n_end : pos >= end);
    // Counting passes do not check widths against the pixel type (the write pass does; a true chain never holds a wider one).
    // A false chain may hold anything, and one that jumps 12 x 73 bits at a time meets the true chain late: the 6-bit field is
    // cut to the bits a valid width needs, which bounds a false chain's blocks like the true one's (4096^2 int32 frames: 0.83 ms
    // for the walk against 1.68 with the whole field).
    const uint32_t wb_bits = c.max_w > 10u ? 32u - (uint32_t)__builtin_clz(c.max_w - 10u) : 0u;
    const uint32_t wb_mask = (1u << wb_bits) - 1u;
    seg_u4 pre[8 * kSegPieces];
    // window t of segment s: dwords [d0, d0 + 32) with d0 = ((fa + X_s + 768 t) >> 5) & ~3
    auto fetch = [&](uint32_t t, uint64_t live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t s = oct + k;
            if ((live >> s) & 1ull) {
                const uint64_t d0 = ((c.fa + (uint64_t)Xo[k] + (uint64_t)t * kSegAdv) >> 5) & ~3ull;
#pragma unroll
                for (uint32_t h = 0; h < kSegPieces; ++h) pre[k * kSegPieces + h] = seg_load16(c, d0 + 4u * (piece + 8u * h));
            }
        }
    };
    // While every row's window lies inside the stream (all but a frame's last windows at the stream's end) the loads need no bounds
    // tests and their addresses advance by a constant: window t < t_safe of row k is at rowp[k] + 56 t dwords (the 16 loads of a
    // window with their 64-bit index arithmetic and tests were a third of a window's overhead).
    uint32_t x_hi = X;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)x_hi, m, 64); x_hi = o > x_hi ? o : x_hi; }
    const int64_t room = ((int64_t)c.n_dw - (int64_t)kSegLoadDw - (int64_t)((c.fa + x_hi) >> 5)) / (int64_t)(kSegAdv / 32u);
#ifdef TRPX_SEG_NO_FAST_FETCH                                  // (A/B build)
    const uint32_t t_safe = 0u;
#else
    const uint32_t t_safe = room <= 0 ? 0u : (room > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)room);
#endif
    const uint32_t* rowp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) rowp[k] = c.s32 + (((c.fa + (uint64_t)Xo[k]) >> 5) & ~3ull) + 4u * piece;
    auto fetch_fast = [&](uint32_t t, uint64_t live) {
        const uint64_t adv = (uint64_t)t * (kSegAdv / 32u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((live >> (oct + k)) & 1ull) {
#pragma unroll
                for (uint32_t h = 0; h < kSegPieces; ++h) __builtin_memcpy(&pre[k * kSegPieces + h], rowp[k] + adv + 32u * h, 16);
            }
        }
    };
    uint64_t live = __ballot(!done);
    if (live) { if (t_start < t_safe) fetch_fast(t_start, live); else fetch(t_start, live); }
    for (uint32_t t = t_start; live; ++t) {
#ifdef TRPX_SEG_STAMPS
        const uint64_t clk0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t s = oct + k;
            if ((live >> s) & 1ull) {
#pragma unroll
                for (uint32_t h = 0; h < kSegPieces; ++h) *reinterpret_cast<seg_u4*>(&win[s * kSegRow + 4u * (piece + 8u * h)]) = pre[k * kSegPieces + h];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t w0 = X + t * kSegAdv, wend = w0 + kSegAdv;
        bool act = !done && pos < wend;
        [[maybe_unused]] const bool act0 = act;
        // (the hook's loads in FRONT of the next window's: loads return in order, and what is needed when this window closes must
        // not wait behind eight loads that are needed a window later)
        if constexpr (!WRITE) { if (hook) hook->open(t, act0); }
        if (t + 1u < t_safe) fetch_fast(t + 1u, live); else fetch(t + 1u, live);   // prefetch: consumed at the top of the next iteration
#ifdef TRPX_SEG_STAMPS
        const uint64_t clk1 = __builtin_amdgcn_s_memrealtime();
        c.clk_wait[WRITE] += clk1 - clk0;
#endif
        // Every lane executes every step (no exec-mask juggling, one branch per step); a lane that is not active
        // computes on a stale position and keeps its state.
        for (bool again = true; again;) {
        while (__ballot(act)) {
#ifdef TRPX_SEG_STATS
            if (lane == 0u) atomicAdd(c.stat + 3, 1u);
#endif
#if !defined(TRPX_SEG_STATS) && !defined(TRPX_SEG_PLAIN_STEP)
            if (!WRITE) {
                // Counting steps, hand-scheduled.  A walking wavefront is bound by the LATENCY of its step -- every instruction
                // hangs on the one before it, ~8 cycles each, plus the LDS read -- so the loop is software-pipelined around the
                // position: the lengths of both possible headers (same width: 1 + 12 w, known from the step before; explicit:
                // straight from the field bits) are ready when the header bit arrives, and the next step's read is issued as soon
                // as the position is known; widths, counters and the end tests follow in its shadow (tools/stepbench.hip: 143 ->
                // 95 ns per step at two wavefronts per SIMD; hipcc's version of the loop below: ~200 ns).
                // The position is kept as an LDS bit address (8 * row address + bit index inside the window; rows are 16-byte
                // aligned, so its low five bits are the funnel shift).  Lanes leave (exec) at the end of their window share or
                // segment; the loop ends with the last lane or when a lane reads 32 one bits, which may be a run of empty blocks:
                // the general step below takes those 32 at a time.
                // Round 6 (k_seg_listed on 2000 x 512^2 Poisson(3) frames 411 -> 334 us): a wavefront alone on its SIMD walks no faster
                // than two (1000 frames take what 2000 do) -- what counts is the wavefront's own instruction stream, and in it the
                // TAKEN branches: the headers of 6 and 12 bits (one block in three hundred) are parsed out of line, the loop is
                // unrolled four times (one taken branch per four steps), lanes leave through v_cmpx.
                const uint32_t endx = end < c.limit + 1u ? end : c.limit + 1u;                   // pos >= end or pos > limit: done
                const uint32_t k0 = 8u * (uint32_t)(uintptr_t)(win + lane * kSegRow) - (w0 - wsh);   // position -> LDS bit address
                uint32_t pw = pos + k0, ls = 1u + (uint32_t)kBlock * w;
                const uint32_t stop = act ? (endx < wend ? endx : wend) + k0 : 0u;
                const uint32_t c90 = 90u, c132 = 132u;
                uint64_t t_ex, t_sx;
                uint32_t t_a, t_bits, t_w3, t_wa, t_wb, t_lx, t_t, t_1, t_2;
// One counting step (copy K of the four the loop is unrolled to: a taken branch costs a walking wavefront about as much as
// eight instructions, so the steps fall through into each other and only the fourth branches back).
#define TRPX_SEG_COUNT_STEP(K)                                                                                                  \
                    "s_waitcnt lgkmcnt(0)\n\t"                                                                                  \
                    "v_alignbit_b32 %[bits], v63, v62, %[pw]\n\t"           /* 32 stream bits from the header on */             \
                    "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"                    /* Terse.hpp:362 */                                 \
                    "v_and_b32 %[t], 1, %[bits]\n\t"                                                                            \
                    "v_mad_u32_u24 %[lx], %[w3], 12, 4\n\t"                 /* block length behind a 4-bit header */            \
                    "v_cmp_eq_u32 %[sx], 7, %[w3]\n\t"                                                                          \
                    "v_cmp_eq_u32 vcc, 1, %[t]\n\t"                         /* header bit 1: same width (Terse.hpp:361) */      \
                    "s_andn2_b64 %[sx], %[sx], vcc\n\t"                     /* lanes with a 6- or 12-bit header (Terse.hpp:364-370): out of line -- */ \
                    "s_cbranch_scc1 5" #K "f\n"                             /* a block in three hundred of detector data, eleven instructions every step otherwise */ \
                    "2" #K ":\n\t"                                                                                              \
                    "v_cndmask_b32 %[lx], %[lx], %[ls], vcc\n\t"                                                                \
                    "v_add_u32 %[pw], %[pw], %[lx]\n\t"                                                                         \
                    "v_lshrrev_b32 %[a], 3, %[pw]\n\t"                                                                          \
                    "v_and_b32 %[a], -4, %[a]\n\t"                                                                              \
                    "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"             /* the next step's bits (lanes that leave: read and dropped) */ \
                    "v_cndmask_b32 %[w], %[w3], %[w], vcc\n\t"                                                                  \
                    "v_add_u32 %[n], 1, %[n]\n\t"                                                                               \
                    "v_mad_u32_u24 %[ls], %[w], 12, 1\n\t"                                                                      \
                    "v_cmpx_lt_u32 vcc, %[pw], %[stop]\n\t"                 /* lanes leave at the end of their window share or segment */ \
                    "s_cbranch_execz 9f\n\t"                                                                                    \
                    "v_cmp_eq_u32 vcc, -1, %[bits]\n\t"                     /* 32 one bits: maybe a run of empty blocks (the general step's) */ \
                    "s_cbranch_vccnz 9f\n\t"
#define TRPX_SEG_COUNT_WIDE(K, WB_BITS)                                                                                         \
                    "5" #K ":\n\t"                                          /* the wide lanes' length and width */              \
                    "v_bfe_u32 %[wa], %[bits], 4, 2\n\t"                                                                        \
                    "v_bfe_u32 %[wb], %[bits], 6, " WB_BITS "\n\t"                                                              \
                    "v_mad_u32_u24 %[t1], %[wa], 12, %[c90]\n\t"            /* a 6-bit header: 6 + 12 (7 + wa) */               \
                    "v_mad_u32_u24 %[t2], %[wb], 12, %[c132]\n\t"           /* a 12-bit header: 12 + 12 (10 + wb) */            \
                    "v_cmp_eq_u32 vcc, 3, %[wa]\n\t"                                                                            \
                    "v_cndmask_b32 %[t1], %[t1], %[t2], vcc\n\t"                                                                \
                    "v_add_u32 %[wa], 7, %[wa]\n\t"                                                                             \
                    "v_add_u32 %[wb], 10, %[wb]\n\t"                                                                            \
                    "v_cndmask_b32 %[wa], %[wa], %[wb], vcc\n\t"                                                                \
                    "v_cndmask_b32 %[lx], %[lx], %[t1], %[sx]\n\t"                                                              \
                    "v_cndmask_b32 %[w3], %[w3], %[wa], %[sx]\n\t"
                asm volatile(
                    "s_mov_b64 %[ex], exec\n\t"
                    "v_cmpx_lt_u32 vcc, %[pw], %[stop]\n\t"
                    "s_cbranch_execz 9f\n\t"
                    "v_lshrrev_b32 %[a], 3, %[pw]\n\t"
                    "v_and_b32 %[a], -4, %[a]\n\t"
                    "ds_read2_b32 v[62:63], %[a] offset1:1\n"
                    "1:\n\t"
                    TRPX_SEG_COUNT_STEP(0) TRPX_SEG_COUNT_STEP(1) TRPX_SEG_COUNT_STEP(2) TRPX_SEG_COUNT_STEP(3)
                    "s_branch 1b\n"
                    TRPX_SEG_COUNT_WIDE(0, "%[nb]") "v_cmp_eq_u32 vcc, 1, %[t]\n\t" "s_branch 20b\n"
                    TRPX_SEG_COUNT_WIDE(1, "%[nb]") "v_cmp_eq_u32 vcc, 1, %[t]\n\t" "s_branch 21b\n"
                    TRPX_SEG_COUNT_WIDE(2, "%[nb]") "v_cmp_eq_u32 vcc, 1, %[t]\n\t" "s_branch 22b\n"
                    TRPX_SEG_COUNT_WIDE(3, "%[nb]") "v_cmp_eq_u32 vcc, 1, %[t]\n\t" "s_branch 23b\n"
                    "9:\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, %[ex]\n"
                    : [pw] "+v"(pw), [w] "+v"(w), [n] "+v"(n), [ls] "+v"(ls), [ex] "=&s"(t_ex), [sx] "=&s"(t_sx), [a] "=&v"(t_a), [bits] "=&v"(t_bits),
                      [w3] "=&v"(t_w3), [wa] "=&v"(t_wa), [wb] "=&v"(t_wb), [lx] "=&v"(t_lx), [t] "=&v"(t_t), [t1] "=&v"(t_1), [t2] "=&v"(t_2)
                    : [stop] "v"(stop), [c90] "s"(c90), [c132] "s"(c132), [nb] "s"(wb_bits)
                    : "vcc", "scc", "memory", "v62", "v63");
                if (act) {
                    pos = pw - k0;
                    done = pos >= endx;
                }
                act = !done && pos < wend;
                if (!__ballot(act)) break;
            }
            if (WRITE) {
                // The same loop for the write pass: the width of every block goes to wf[n] (zero widths too: cheaper than a lane
                // mask), the bit position of every 256th block to tf; lanes also leave in front of the frame's last block, which
                // may be a partial one (the general step below knows how).  Every lane's widths land in a different cache line,
                // and a store instruction of 64 lines keeps the memory pipeline busy for longer than the rest of the step takes:
                // four widths are collected in a register and stored as one (unaligned) dword every fourth step; what a lane
                // holds when it leaves follows as bytes.
                const uint32_t endx = end < c.limit + 1u ? end : c.limit + 1u;
                const uint32_t k0 = 8u * (uint32_t)(uintptr_t)(win + lane * kSegRow) - (w0 - wsh);
                uint32_t pw = pos + k0, ls = 1u + (uint32_t)kBlock * w, wmax = w, acc = 0u;   // (wmax: the width the lane comes with, then every wide header's)
                const uint32_t n0 = n;
                const uint32_t stop = act ? (endx < wend ? endx : wend) + k0 : 0u;
                const uint32_t nstop = n_end - 1u;                                               // (n_end >= 1)
                const uint32_t c90 = 90u, c132 = 132u;
                // The position of a block that opens a 256-block group is latched without a branch and stored behind the loop, which takes
                // at most 252 steps at a time: one such block per lane.
                const uint32_t nt = (n + (uint32_t)kTileBlocks - 1u) & ~((uint32_t)kTileBlocks - 1u);
                uint32_t grp = 0xFFFFFFFFu, left = 62u;
                uint64_t t_ex, t_sx;
                uint32_t t_a, t_bits, t_w3, t_wa, t_wb, t_lx, t_t, t_1, t_2;
// One writing step (copy K of four, see TRPX_SEG_COUNT_STEP); STORE: the fourth copy's store of the last four widths.
#define TRPX_SEG_WRITE_STEP(K, STORE)                                                                                           \
                    "v_cmp_eq_u32 vcc, %[n], %[nt]\n\t"                     /* block n opens a 256-block group: its header position is kept */ \
                    "v_cndmask_b32 %[grp], %[grp], %[pw], vcc\n\t"                                                              \
                    "s_waitcnt lgkmcnt(0)\n\t"                                                                                  \
                    "v_alignbit_b32 %[bits], v63, v62, %[pw]\n\t"                                                               \
                    "v_bfe_u32 %[w3], %[bits], 1, 3\n\t"                                                                        \
                    "v_and_b32 %[t], 1, %[bits]\n\t"                                                                            \
                    "v_mad_u32_u24 %[lx], %[w3], 12, 4\n\t"                                                                     \
                    "v_cmp_eq_u32 %[sx], 7, %[w3]\n\t"                                                                          \
                    "v_cmp_eq_u32 vcc, 1, %[t]\n\t"                                                                             \
                    "s_andn2_b64 %[sx], %[sx], vcc\n\t"                     /* wide headers out of line, as in the counting loop */ \
                    "s_cbranch_scc1 5" #K "f\n"                                                                                 \
                    "4" #K ":\n\t"                                                                                              \
                    "v_cndmask_b32 %[lx], %[lx], %[ls], vcc\n\t"                                                                \
                    "v_add_u32 %[pw], %[pw], %[lx]\n\t"                                                                         \
                    "v_lshrrev_b32 %[a], 3, %[pw]\n\t"                                                                          \
                    "v_and_b32 %[a], -4, %[a]\n\t"                                                                              \
                    "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"                                                                 \
                    "v_cndmask_b32 %[w], %[w3], %[w], vcc\n\t"                                                                  \
                    "v_alignbyte_b32 %[acc], %[w], %[acc], 1\n\t"           /* the last four widths, oldest in the low byte */  \
                    "v_add_u32 %[n], 1, %[n]\n\t"                                                                               \
                    STORE                                                                                                       \
                    "v_mad_u32_u24 %[ls], %[w], 12, 1\n\t"                                                                      \
                    "v_cmpx_lt_u32 vcc, %[pw], %[stop]\n\t"                                                                     \
                    "v_cmpx_gt_u32 vcc, %[nstop], %[n]\n\t"                                                                     \
                    "s_cbranch_execz 9f\n\t"                                                                                    \
                    "v_cmp_eq_u32 vcc, -1, %[bits]\n\t"                                                                         \
                    "s_cbranch_vccnz 9f\n\t"
// (widths from 4-bit headers are at most 6: only the wide lanes' can be wider than the pixel type allows)
#define TRPX_SEG_WRITE_WIDE(K)                                                                                                  \
                    TRPX_SEG_COUNT_WIDE(K, "6")                                                                                 \
                    "v_cndmask_b32 %[t2], 0, %[wa], %[sx]\n\t"                                                                  \
                    "v_max_u32 %[wmax], %[wmax], %[t2]\n\t"                                                                     \
                    "v_cmp_eq_u32 vcc, 1, %[t]\n\t"                                                                             \
                    "s_branch 4" #K "b\n"
#ifndef TRPX_SEG_NO_STORE
#define TRPX_SEG_WIDTH_STORE "v_add_u32 %[t], -4, %[n]\n\t" "global_store_dword %[t], %[acc], %[wf]\n\t"   /* width[n - 4 .. n - 1] */
#else
#define TRPX_SEG_WIDTH_STORE ""                            /* (timing experiment: tools/r6_segvariant.sh nostore -DTRPX_SEG_NO_STORE) */
#endif
                asm volatile(
                    "s_mov_b64 %[ex], exec\n\t"
                    "v_cmpx_lt_u32 vcc, %[pw], %[stop]\n\t"
                    "v_cmpx_gt_u32 vcc, %[nstop], %[n]\n\t"
                    "s_cbranch_execz 9f\n\t"
                    "v_lshrrev_b32 %[a], 3, %[pw]\n\t"
                    "v_and_b32 %[a], -4, %[a]\n\t"
                    "ds_read2_b32 v[62:63], %[a] offset1:1\n"
                    "1:\n\t"
                    TRPX_SEG_WRITE_STEP(0, "") TRPX_SEG_WRITE_STEP(1, "") TRPX_SEG_WRITE_STEP(2, "") TRPX_SEG_WRITE_STEP(3, TRPX_SEG_WIDTH_STORE)
                    "s_sub_u32 %[left], %[left], 1\n\t"                     // (at most 252 steps at a time: one group opener per lane, see grp)
                    "s_cbranch_scc0 1b\n\t"
                    "s_branch 9f\n"
                    TRPX_SEG_WRITE_WIDE(0) TRPX_SEG_WRITE_WIDE(1) TRPX_SEG_WRITE_WIDE(2) TRPX_SEG_WRITE_WIDE(3)
                    "9:\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_mov_b64 exec, %[ex]\n"
                    : [pw] "+v"(pw), [w] "+v"(w), [n] "+v"(n), [ls] "+v"(ls), [wmax] "+v"(wmax), [acc] "+v"(acc), [grp] "+v"(grp), [left] "+s"(left), [ex] "=&s"(t_ex), [sx] "=&s"(t_sx),
                      [a] "=&v"(t_a), [bits] "=&v"(t_bits), [w3] "=&v"(t_w3), [wa] "=&v"(t_wa), [wb] "=&v"(t_wb), [lx] "=&v"(t_lx),
                      [t] "=&v"(t_t), [t1] "=&v"(t_1), [t2] "=&v"(t_2)
                    : [stop] "v"(stop), [nt] "v"(nt), [nstop] "v"(nstop), [c90] "s"(c90), [c132] "s"(c132), [wf] "s"(wf)
                    : "vcc", "scc", "memory", "v62", "v63");
                if (act) {
                    const uint32_t held = (n - n0) & 3u;                                        // widths of blocks n - held .. n - 1, in acc's top bytes
                    for (uint32_t q = 0; q < held; ++q) wf[n - held + q] = (uint8_t)(acc >> (8u * (4u - held + q)));
                    if (grp != 0xFFFFFFFFu) tf[nt / kTileBlocks] = grp - k0;                    // (nt < n: the lane has walked block nt)
                    pos = pw - k0;
                    bad = bad || wmax > c.max_w;
                    done = by_count ? n >= n_end : pos >= end;
                    if (pos > c.limit) { bad = bad || n < n_end || !by_count; done = true; }
                }
                act = !done && pos < wend;
                if (!__ballot(act)) break;
            }
#endif
            const uint32_t li = pos - w0 + wsh;                                           // bit index inside the lane's window
            const uint32_t dw = min(li >> 5, kSegRow - 2u);                               // (an inactive lane may be past its row)
            const uint32_t* row = win + lane * kSegRow + dw;
            const uint32_t bits = __builtin_amdgcn_alignbit(row[1], row[0], li);          // 32 stream bits from pos
            const bool same = (bits & 1u) != 0u;                                          // Terse.hpp:361
            const uint32_t w3 = (bits >> 1) & 7u, wa = 7u + ((bits >> 4) & 3u), wb = 10u + ((bits >> 6) & (WRITE ? 63u : wb_mask));
            const uint32_t wx = w3 != 7u ? w3 : (wa != 10u ? wa : wb);                    // Terse.hpp:362-370
            const uint32_t hx = w3 != 7u ? 4u : (wa != 10u ? 6u : 12u);
            uint32_t wn = same ? w : wx;
            const bool wide = WRITE && wn > c.max_w;                                       // (counting passes take any width: see above)
            wn = wide ? 0u : wn;
            // a run of empty blocks (header bits 1, no payload: 1 bit each) is taken up to 32 blocks at a time
            const bool zrun = same && wn == 0u;
            const uint32_t ones = min((uint32_t)(__ffs((int)~bits) - 1), 32u);
            const uint32_t room = WRITE && by_count ? n_end - n : end - pos;
            const uint32_t rep = zrun ? min(ones, room) : 1u;
            const uint32_t nv = WRITE && n + 1u == c.n_blocks ? c.nb_last : (uint32_t)kBlock;
            const uint32_t len = zrun ? rep : (same ? 1u : hx) + nv * wn;
            if (WRITE) {
                if (act) {
                    bad = bad || wide;
                    if (n + rep > c.n_blocks) { bad = true; done = true; }
                    else {
#ifndef TRPX_SEG_NO_STORE
                        if (wn) wf[n] = (uint8_t)wn;
#endif
                        const uint32_t m = (n + (uint32_t)kTileBlocks - 1u) & ~((uint32_t)kTileBlocks - 1u);
                        if (m < n + rep) tf[m / kTileBlocks] = pos + (m - n);               // rep > 1 only for 1-bit blocks
                    }
                }
            }
            pos = act ? pos + len : pos;
            n = act ? n + rep : n;
            w = act ? wn : w;
            done = done || (act && (WRITE && by_count ? n >= n_end : pos >= end));
            if (WRITE) { if (act && pos > c.limit) bad = bad || n < n_end || !by_count; }
            done = done || pos > c.limit;
            act = !done && pos < wend;
        }
        again = false;
        if constexpr (!WRITE) {
            if (hook) {
                const bool more = hook->close(t, act0, wend, pos, w, n, done, end);
                act = more && !done && pos < wend;
                again = __ballot(act) != 0ull;
            }
        }
        }
        __builtin_amdgcn_wave_barrier();                      // every lane is through with this window before it is overwritten
#ifdef TRPX_SEG_STAMPS
        c.clk_step[WRITE] += __builtin_amdgcn_s_memrealtime() - clk1;
#endif
        live = __ballot(!done);
        if ((uint64_t)t * kSegAdv > (uint64_t)c.limit + 2u * kSegAdv) break;   // (cannot happen: done is set past the limit)
    }
}

// First-round guess for run-dominated streams.  Inside a run of equal-width blocks the headers are single 1 bits
// at stride s = 1 + 12 w, and a false chain almost never merges into such a run (it would have to hit a block start
// with the right width by chance).  So every lane looks in the first window of its segment for the smallest
// w in 1..4 and then the first position q whose R header bits q, q + s, ..., q + (R-1) s are all 1 (bit-parallel: 32
// positions per AND chain; R = 12..32 so that the evidence spans ~560 bits), and, if it finds one, moves its
// boundary there and starts from (X + q, w).  Payload bits can pass the test too (a sign bit that is mostly 1):
// the guess is only a guess -- it is verified like any other state, see seg_fixpoint.  Inside a run of empty
// blocks (all bits 1) the plain start (X, 0) is itself a true state and is kept; runs wider than 4 bits do not fit
// the window often enough.
// Returns ~0 when there is no candidate.
__device__ __forceinline__ uint64_t seg_comb_guess(const SegCtx& c, const uint32_t* __restrict__ win, uint32_t X) {
    const uint32_t* row = win + (uint32_t)lane_id() * kSegRow;
    uint64_t best = ~0ull;
    {   // a run of empty blocks (every bit a header bit 1): (X, 0) is a true state, keep it
        uint32_t a = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < 3u; ++k) {
            const uint32_t q = c.wsh + 32u * k;
            a &= __builtin_amdgcn_alignbit(row[(q >> 5) + 1u], row[q >> 5], q);
        }
        if (a == 0xFFFFFFFFu) best = seg_pack(X, 0u);
    }
    const uint32_t w_hi = c.max_w < 4u ? c.max_w : 4u;
    // Second pass, for the lanes the first left without a guess in a stream that is run-dominated (others did find one): half
    // the evidence.  A peak inside the 560 bits -- two explicit headers -- leaves a lane on a false chain that does not merge
    // before the next explicit header, about a segment away, and every such lane in a row costs the wavefront a round.
    for (uint32_t pass = 0; pass < 2u; ++pass) {
        const uint32_t evid = pass ? 250u : 560u, rmin = pass ? 8u : 12u;
        for (uint32_t w = 1; w <= w_hi; ++w) {
            const uint32_t s = 1u + (uint32_t)kBlock * w;
            const uint32_t r0 = evid / s, R = r0 < rmin ? rmin : (r0 > 32u ? 32u : r0);
            const uint32_t fit = kSegSpan - (R - 1u) * s - 12u, range = w < 3u ? (4u * s < fit ? 4u * s : fit) : fit;
            const uint32_t words = (range + 31u) / 32u;
            for (uint32_t i = 0; i < words; ++i) {
                uint32_t a = i + 1u == words && (range & 31u) ? (1u << (range & 31u)) - 1u : 0xFFFFFFFFu;
                for (uint32_t k = 0; k < R; ++k) {
                    const uint32_t q = c.wsh + 32u * i + k * s;                               // (wave-uniform)
                    a &= __builtin_amdgcn_alignbit(row[(q >> 5) + 1u], row[q >> 5], q);
                }
                if (a && best == ~0ull) best = seg_pack(X + 32u * i + (uint32_t)__builtin_ctz(a), w);
            }
            if (!__ballot(best == ~0ull)) break;
        }
        const uint64_t none = __ballot(best == ~0ull);
        if (!none || __builtin_popcountll(~none) < 8) break;                  // all served, or not a run-dominated stream
    }
    return best;
}

// Frame-constant part of the context.
__device__ __forceinline__ bool seg_ctx(SegCtx& c, const uint8_t* terse, uint64_t terse_bytes, const uint64_t* frame_offsets,
                                        uint64_t frame, const FrameGeom& g, uint32_t max_w, uint32_t G, uint32_t* status) {
    c.stat = status;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    if (!(fe > fo && fe <= terse_bytes) || 8 * (fe - fo) >= 0xF0000000ull) return false;
    c.s32 = reinterpret_cast<const uint32_t*>(terse);
    c.n_dw = (terse_bytes + 3) / 4;
    c.fa = 8 * fo;
    c.limit = (uint32_t)(8 * (fe - fo));
    c.L = seg_len_bits(c.limit, G);
    c.wsh = (uint32_t)(c.fa & 127u);
    c.n_blocks = g.n_blocks;
    c.nb_last = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock);
    c.max_w = max_w;
    return true;
}

__device__ __forceinline__ uint64_t seg_shfl_up1(uint64_t v) {
    const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, 1, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), 1, 64);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// Zero this wave's share of the frame's width array (the write pass stores non-zero widths only).
__device__ __forceinline__ void seg_zero_widths(uint8_t* __restrict__ wf, uint32_t n_blocks, uint32_t k, uint32_t K) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t a = (uint64_t)(uintptr_t)wf;
    const uint32_t head = (uint32_t)((16u - (a & 15u)) & 15u) < n_blocks ? (uint32_t)((16u - (a & 15u)) & 15u) : n_blocks;
    const uint32_t n16 = (n_blocks - head) / 16u;
    if (k == 0u) {
        if (lane < head) wf[lane] = 0;
        const uint32_t tail0 = head + 16u * n16;
        if (tail0 + lane < n_blocks) wf[tail0 + lane] = 0;                               // < 16 bytes
    }
    seg_u4* q = reinterpret_cast<seg_u4*>(wf + head);
    const uint32_t per = (n16 + K - 1u) / K, lo = k * per, hi = lo + per < n16 ? lo + per : n16;
    const seg_u4 z = {0u, 0u, 0u, 0u};
    for (uint32_t i = lo + lane; i < hi; i += kWave) q[i] = z;
}

}  // namespace trpx
