// PROLIX decode of LARGE frames on the per-frame route (gfx950 / CDNA4): cutting a frame into PARTS.
//
// The per-frame decoder (decode_frame.hip) puts one serial header walker (reference include/Terse.hpp:360-372: block b+1's
// position is only known after block b's header) and three extraction waves on a frame.  For the 22 K blocks of a 512 x 512
// frame that is the right grain; a 1030 x 1065 frame (91 K blocks) is a 4 x longer chain on 4 x fewer workgroups, a 2048 x
// 2048 frame (350 K blocks) 16 x -- 200 frames of 1030 x 1065 decoded at 0.13 of the HBM peak, 128 frames of 2048 x 2048 at
// 0.07 (on the tiled route).  A frame of more than kPartMaxBlocks blocks is therefore cut into P parts, every part a unit of
// work of its own for the per-frame decoder (k_decode_parts), which needs the chain state in front of every part: the
// bit position of its first block's header, the width of the block before it, and -- for the place of its pixels -- the
// number of its first block.  This file finds them, with the serial walker's own cheap steps (one per RUN of equal widths):
//
//   k_part_guess    a start state S_p for every part p >= 1.  The parts are cut at bit positions X_p = p * L.  Behind X_p the
//                   wavefront looks for a run of blocks that repeat their width: kPartEvid header bits 1 at stride 1 + 12 w
//                   (bit-parallel: 2048 candidate positions per width and pass); a position kPartSkip blocks inside such a
//                   run is a block start with all but certainty.  No run within 16 K bits: the plain (X_p, 0), which is
//                   almost surely not a state of the chain -- but a chain started there merges with the true one at the first
//                   explicit header it meets on a block start (its state after an explicit header does not depend on the
//                   width before, Terse.hpp:362-370).
//   k_part_walk     one wavefront per part p < P - 1 walks from S_p to the position of S_(p+1), counts the blocks and leaves
//                   the state it arrives in (OUT_p) plus a checkpoint (state, blocks so far) every few thousand bits.
//                   A part whose explicit headers are too dense for the serial walker (more than 1 in 6 blocks) stops:
//                   its frame is the position-parallel walk's case.
//   k_part_repair   one wavefront per link p -> p + 1 that is OPEN (OUT_p != S_(p+1)): the blocks of part p + 1 counted again
//                   from OUT_p -- only until this chain meets the one k_part_walk walked there (a checkpoint), which in
//                   run-dominated data is the next explicit header; the count follows from the checkpoint's.
//   k_part_resolve  one wavefront per frame.  S_0 = (0, 0) is true (Terse.hpp:359, :505); a chain that starts in a true state
//                   is true to its end; a link is closed, or repaired from the true OUT_p -- by induction every part's
//                   start state, block count and end state are the true chain's, the prefix sums of the counts are the
//                   parts' first blocks, and the part table is written.  A frame where this does not work out (a dense
//                   part, a repair that ran into another repair, a part of more than kPartMaxBlocks blocks, a corrupt
//                   stream) is listed as a whole for the route large frames took before (position-parallel walk + tiled
//                   extraction, decode_seg.hip): guesses and repairs only ever steer the speed, never the result -- and
//                   every part's decoder checks again that it ends in the next part's state, the last one that
//                   S_f = 1 + bits/8 (Terse.hpp:547).
//
// HBM traffic: the stream once more (walk only: no pixels), 0.2 of the algorithmic bytes of a u16 stack.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include <stdlib.h>

namespace trpx {

#ifndef TRPX_PART_CHUNK_DW
#define TRPX_PART_CHUNK_DW 2048
#endif
constexpr int kPartChunkDw = TRPX_PART_CHUNK_DW;   // the walker's stream window: 8 KB
constexpr uint32_t kPartEvid = 32;             // header bits of evidence for a start inside a run: a stack has thousands of cuts x 10^5 candidates each, and with 24 bits one guess per stack WAS wrong -- on a chain that, in run-dominated data, does not merge before its part ends (the frame then takes the other route: +4 ms)
constexpr uint32_t kPartSkip = 12;             // the start lies this many blocks inside the evidence (the bits in front of a run are 1 half the time)
#ifndef TRPX_PART_SEARCH
#define TRPX_PART_SEARCH 8
#endif
constexpr uint32_t kPartSearch = TRPX_PART_SEARCH;   // passes of 2048 candidate positions behind X_p
constexpr uint32_t kPartCk = 256;              // checkpoints per part
constexpr uint32_t kPartWeak = 0x80000000u;    // PartState::w: a plain guess (not expected to be a state of the chain)
constexpr uint32_t kPartAmbig = 0x40000000u;   //   ... because runs were found, but not which of their passing positions is the header's
constexpr uint32_t kPartRuns = 0x20000000u;    //   ... (index route) although runs of equal widths lie nearby: run-dominated data, where a false chain is slow to merge
constexpr uint32_t kPartReady = 0x10000000u;   //   (index route) the state has been published: k_chain_walk's wavefronts wait for their neighbour's
constexpr uint32_t kPartFlags = kPartWeak | kPartAmbig | kPartRuns | kPartReady;

struct PartState { uint32_t pos, w; };
struct PartCk { uint32_t pos, w, cnt; };
struct PartWalk {                              // what k_part_walk leaves per (frame, part p < P - 1)
    uint32_t o_pos, o_w;                       // OUT_p: the state at the chain's first block start >= T_p
    uint32_t cnt;                              // blocks started in [S_p, T_p)
    uint32_t flags;                            // 1: the chain left the frame / held an illegal width; 2: too dense for the serial walker
    uint32_t n_ck;                             // checkpoints left behind
    uint32_t s_pos, s_w;                       // (index route) the state the counting walk started in: the guess, or behind the last illegal width
    uint32_t pad;                              // (index route) explicit headers among the blocks counted
};
struct PartFix {                               // what k_part_repair leaves per (frame, part 1 <= p < P - 1)
    uint32_t state;                            // 0: link closed, nothing done; 1: merged into the part's walk; 2: walked to T_p on its own; 3: failed
    uint32_t cnt, o_pos, o_w;                  // the part's block count (and, state 2, its end state) from the true start
};

// Checkpoints and walk records are read by OTHER wavefronts of the launch that writes them (index route: k_chain_walk's links), on
// other XCDs: agent-scope atomic words -- write-through stores, loads that do not trust a cached line -- ordered by the record's
// `published` bit alone.  (Fences instead write back or invalidate the XCD's whole L2, dirty with the walk's entries: measured,
// 45 - 100 us per launch of 4000 wavefronts.)
constexpr uint32_t kWalkPublished = 0x80000000u;   // PartWalk::flags
#if defined(TRPX_CHAIN_FORCE_TIMEOUT)
constexpr uint64_t kChainWaitTicks = 200000ull;        // (test build, see k_chain_walk: 2 ms)
#elif defined(TRPX_PART_STATS)
constexpr uint64_t kChainWaitTicks = 3000000000ull;    // (diagnostic build: its printfs hold wavefronts up for milliseconds)
#else
constexpr uint64_t kChainWaitTicks = 25000000ull;      // how long a wavefront waits for another's word: 0.25 s of the 100 MHz counter
#endif
__device__ __forceinline__ void part_ck_store(PartCk* dst, const PartCk& c) {
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    __hip_atomic_store(d, c.pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 1, c.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(d + 2, c.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ PartCk part_ck_load(const PartCk* src) {
    const uint32_t* d = reinterpret_cast<const uint32_t*>(src);
    PartCk c;
    c.pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));   // (wave-uniform address)
    c.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    c.cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return c;
}

// Which frames stay whole on the per-frame route (one workgroup per frame; header-dense ones handed to the one-wavefront
// position-parallel walk): up to kPartMaxBlocks blocks always; in a stack of kManyFrames frames and more up to kManyBlocks -- the
// workgroups of such a stack fill the GPU by themselves, and cutting its frames costs more than it levels (1280 x 640^2 Poisson(3)
// frames through the index route: 1.42 ms; 2000 x 512^2, the same bytes: 0.74).
static uint32_t g_many_frames = kManyFrames, g_many_blocks = kManyBlocks;
void set_single_part_rule(uint32_t many_frames, uint32_t many_blocks) { g_many_frames = many_frames; g_many_blocks = many_blocks; }
uint32_t single_part_blocks(size_t n_frames) {
    return n_frames >= g_many_frames && g_many_blocks > kPartMaxBlocks ? g_many_blocks : kPartMaxBlocks;
}

uint32_t parts_per_frame(const FrameGeom& g, size_t n_frames) {
    if (g.n_blocks <= single_part_blocks(n_frames) || n_frames == 0) return 1u;
    // enough parts to fill the GPU one and a half times over (8 workgroups per CU), of 4 K .. 16 K blocks
#ifdef TRPX_DIAGNOSTICS
    static const uint64_t want_total = getenv("TRPX_PART_TOTAL") ? (uint64_t)atoi(getenv("TRPX_PART_TOTAL")) : 3072u;
    static const uint64_t min_blocks = getenv("TRPX_PART_MIN") ? (uint64_t)atoi(getenv("TRPX_PART_MIN")) : 4096u;
#else
    constexpr uint64_t want_total = 3072u, min_blocks = 4096u;
#endif
    const uint64_t p_min = (g.n_blocks + kPartBlocks - 1u) / kPartBlocks, p_max = g.n_blocks / min_blocks;
    const uint64_t p_want = (want_total + n_frames - 1u) / n_frames;
    const uint64_t hi = p_max > p_min ? p_max : p_min;
    const uint64_t P = p_want < p_min ? p_min : (p_want > hi ? hi : p_want);
    return (uint32_t)P;
}
struct PartWs { size_t states, walks, fixes, cks, total; };
static PartWs part_ws_layout(size_t n_frames, size_t P) {
    PartWs w;
    w.states = 0;
    w.walks = align_up(w.states + n_frames * P * sizeof(PartState), 256);
    w.fixes = align_up(w.walks + n_frames * (P - 1) * sizeof(PartWalk), 256);
    w.cks = align_up(w.fixes + n_frames * (P - 1) * sizeof(PartFix), 256);
    w.total = align_up(w.cks + n_frames * (P - 1) * kPartCk * sizeof(PartCk), 256);
    return w;
}
size_t part_workspace_bytes(const FrameGeom& g, size_t n_frames) {
    const size_t P = parts_per_frame(g, n_frames);
    return P > 1 ? part_ws_layout(n_frames, P).total : 0;
}

// One LDS-DMA piece (see decode_frame.hip): lane l's 16 bytes at `src` land at LDS byte address lds_base + 16 * l.
__device__ __forceinline__ void part_lds_dma16(const uint32_t* src_uniform, uint32_t lane_byte_offset, uint32_t lds_base) {
    uint32_t keep;
    // (wave-uniform by construction; said again for the builds in which the compiler loses sight of it: "invalid operand")
    const uint64_t a = (uint64_t)(uintptr_t)src_uniform;
    const uint32_t* const src = reinterpret_cast<const uint32_t*>((uintptr_t)((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                                                                              ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32)));
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_offset), "s"(src), "s"(base) : "memory");
}

struct PartWin {                               // the frame's stream behind an LDS window
    const uint32_t* s32;
    uint64_t n_dw, frame_dw;                   // dwords of the stream; dword of the frame's first bit
    uint32_t frame_sh;                         // bit of the frame's first bit inside that dword
    bool base16;
    int32_t c_lo, c_hi;                        // the window holds dwords [c_lo, c_hi) of the frame
#ifdef TRPX_CHAIN_STAMPS
    uint32_t n_fill = 0, t_fill = 0;           // diagnostic build: window fills and the 10 ns ticks they took
#endif
};

// Window := the kPartChunkDw dwords from the (16-byte aligned) dword that holds frame bit `pos` on.
__device__ __forceinline__ void part_fill(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t pos) {
    const uint32_t lane = (uint32_t)lane_id();
#ifdef TRPX_CHAIN_STAMPS
    const uint64_t st_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t need_lo = (W.frame_sh + pos) >> 5;
    W.c_lo = (int32_t)(((W.frame_dw + need_lo) & ~3ull) - W.frame_dw);
    W.c_hi = W.c_lo + kPartChunkDw;
    const uint64_t d0 = (uint64_t)((int64_t)W.frame_dw + W.c_lo);
    __builtin_amdgcn_wave_barrier();           // (every lane is through with the old window)
    if (W.base16 && (d0 & 3) == 0 && d0 + kPartChunkDw <= W.n_dw) {
#pragma unroll
        for (int it = 0; it < kPartChunkDw / (kWave * 4); ++it)
            part_lds_dma16(W.s32 + d0 + it * kWave * 4, lane * 16u, (uint32_t)(uintptr_t)&s_chunk[it * kWave * 4]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (uint32_t i = lane * 4; i < (uint32_t)kPartChunkDw; i += kWave * 4) {
            const uint64_t d = d0 + i;
            uint4 x;
            x.x = d < W.n_dw ? W.s32[d] : 0u; x.y = d + 1 < W.n_dw ? W.s32[d + 1] : 0u;
            x.z = d + 2 < W.n_dw ? W.s32[d + 2] : 0u; x.w = d + 3 < W.n_dw ? W.s32[d + 3] : 0u;
            *reinterpret_cast<uint4*>(&s_chunk[i]) = x;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef TRPX_CHAIN_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    ++W.n_fill; W.t_fill += (uint32_t)(__builtin_amdgcn_s_memrealtime() - st_t0);
#endif
}

// 32 stream bits from frame bit q on (inside the window)
__device__ __forceinline__ uint32_t part_bits(const PartWin& W, const uint32_t* __restrict__ s_chunk, uint32_t q) {
    const uint32_t fbit = W.frame_sh + q - 32u * (uint32_t)W.c_lo;
    const uint32_t i = fbit >> 5 < (uint32_t)kPartChunkDw ? fbit >> 5 : (uint32_t)kPartChunkDw - 1u;   // (s_chunk has 4 dwords of slack)
    return __builtin_amdgcn_alignbit(s_chunk[i + 1u], s_chunk[i], fbit);
}

// A block start inside a run of equal widths at or behind frame bit X, or the plain guess.  kPartEvid header bits 1 at stride
// s = 1 + 12 w from some q in [X, X + 2048 kPartSearch) on -- 32 candidate positions per lane and AND chain, the widths in
// ascending order, for each of them the range in passes of 2048 positions -- make (q + kPartSkip * s, w) the state; inside a run of EMPTY blocks (every bit a
// header bit 1) X itself is one.  (Payload bits pass the test with probability 2^-32 per candidate, and the first kPartSkip
// of the evidence may be a neighbour's bits: a guess is only a guess, k_part_resolve verifies.)
//
// Pixels with a pedestal (raw detector counts: every value in 100 .. 107, say) have CONSTANT payload bits, and those pass the
// test at the run's own stride just like its header bits: the first passing position is then a payload bit's more often than
// not, a chain started there stays beside the frame's for as long as the run lasts, and every cut of every frame is wrong the
// same way.  But the passing positions of one stride are a header's plus, for every constant bit b of the pixels, the twelve
// positions header + 1 + b + j w of that bit in the block's twelve fields: the header is the one passing position relative to
// which all the others fall into complete classes of twelve (part_header_phase; one passing position per stride, the case
// without a pedestal, needs no such test).
__device__ __forceinline__ bool part_pm_bit(const uint32_t* __restrict__ pm, uint32_t i) { return ((pm[i >> 5] >> (i & 31u)) & 1u) != 0u; }

// pm: pass bits of 2048 consecutive positions; r0: the first passing one; s, w: the stride and its width.  Returns the index of
// the header among the passing positions in [r0, r0 + s), or ~0 if that cannot be told (fewer than s positions left, no or
// several consistent ones).
__device__ __forceinline__ uint32_t part_header_phase(const uint32_t* __restrict__ pm, uint32_t r0, uint32_t s, uint32_t w) {
    const uint32_t lane = (uint32_t)lane_id();
    if (r0 + 2u * s > 2048u) return ~0u;
    uint32_t n = 0;
    for (uint32_t i = lane; i < s; i += kWave) n += part_pm_bit(pm, r0 + i) ? 1u : 0u;
    n = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n), 63);
    if (n == 1u) return r0;
    if (n % 12u != 1u) return ~0u;
    uint32_t found = ~0u, n_found = 0, found_any = ~0u, n_any = 0;
    for (uint32_t base = 0; base < s; base += kWave) {
        const uint32_t c = base + lane;                                        // candidate header: position r0 + c
        bool ok = c < s && part_pm_bit(pm, r0 + c);
        // (the grid of twelve fields may be shifted along a run of constant bits and still fit -- constant bits 5 and 6 of 7 read
        // as {0, 6} one position to the left, as {0, 1} two --: a pedestal's constant bits are the TOP ones, w - k .. w - 1, and
        // only the candidate whose complete classes are exactly such a run counts.  `top` = length of the run of complete classes
        // that ends at bit w - 1.)
        uint32_t classes = 0, top = 0;
        for (uint32_t b = 0; b < w && __ballot(ok); ++b) {
            uint32_t cnt = 0;
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)kBlock; ++j) {
                uint32_t o = c + 1u + b + j * w;                               // < 2 s
                o = o >= s ? o - s : o;
                cnt += ok && part_pm_bit(pm, r0 + o) ? 1u : 0u;
            }
            ok = ok && (cnt == 0u || cnt == (uint32_t)kBlock);
            classes += cnt == (uint32_t)kBlock ? 1u : 0u;
            top = cnt == (uint32_t)kBlock ? top + 1u : 0u;
        }
        ok = ok && 1u + (uint32_t)kBlock * classes == n;
        // (The grid shifted by one bit ALWAYS fits too: a pedestal's top bit w - 1 is constant, and seen from the position in
        // front of the header -- the last field's top bit -- the header bits are class 0 and every class b is class b + 1.  What
        // tells them apart: the block's width says bit w - 1 is set somewhere, in a pedestal everywhere -- the true grid HAS class
        // w - 1, the shifted one only if bit w - 2 is constant too, and then the top-run rule decides.  Pixels 40 or 41: bits 5
        // and 3 -- no top run, but only one grid with class 5.)
        const uint64_t m_any = __ballot(ok && top >= 1u);
        if (m_any) { n_any += (uint32_t)__builtin_popcountll(m_any); found_any = r0 + base + (uint32_t)__builtin_ctzll(m_any); }
        const uint64_t m = __ballot(ok && top == classes);
        if (m) { n_found += (uint32_t)__builtin_popcountll(m); found = r0 + base + (uint32_t)__builtin_ctzll(m); }
    }
    return n_any == 1u ? found_any : (n_found == 1u ? found : ~0u);
}

// reach: the search (and the state it returns) stays inside [X, X + reach).
__device__ __forceinline__ PartState part_guess(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t* __restrict__ s_pm, uint32_t X, uint32_t limit,
                                                uint32_t max_w, uint32_t reach = 0xFFFFFFFFu, uint32_t max_passes = kPartSearch) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t s_max = 1u + (uint32_t)kBlock * max_w;
    const PartState plain{X, kPartWeak};
#ifdef TRPX_PART_FORCE_WEAK
    return plain;                              // test build (make weakparts): every link is open and goes through k_part_repair
#endif
    // the search range and its evidence lie inside one window: [X, X + 2048 kPartSearch + kPartEvid s_max) < 64 K bits
    static_assert(2048u * kPartSearch + kPartEvid * (1u + 12u * 32u) + 256u < 32u * (uint32_t)kPartChunkDw - 128u, "one window per guess");
    uint32_t passes = max_passes < kPartSearch ? max_passes : kPartSearch;
    while (passes && (uint64_t)X + 2048ull * passes + (uint64_t)kPartEvid * s_max + 128u > (uint64_t)limit) --passes;   // too close to the frame's end
    while (passes && 2048ull * passes + (uint64_t)kPartSkip * s_max > (uint64_t)reach) --passes;
    if (passes == 0u) return plain;
    part_fill(W, s_chunk, X);
#ifdef TRPX_PART_NOGUESS
    return plain;
#endif
#ifdef TRPX_PART_MAXW
    max_w = max_w < TRPX_PART_MAXW ? max_w : TRPX_PART_MAXW;
#endif
    {
        const uint32_t a = lane < 4u ? part_bits(W, s_chunk, X + 32u * lane) : 0xFFFFFFFFu;
        if (!__ballot(a != 0xFFFFFFFFu)) return PartState{X, 0u};
    }
    // widths outside, passes inside: a stream whose runs are short (a change every ten blocks) has no run of kPartEvid blocks in
    // most passes, and all max_w widths of a pass without a hit cost 30 x what the first pass with a hit does
    // (widths above 8 get two passes: a cut without a run of a small width in reach is rare, and sweeping every width of a
    // 32-bit type over the whole range for it cost eight 4096 x 4096 int32 frames 70 of 100 us here)
    bool ambiguous = false;
    for (uint32_t w = 1; w <= max_w; ++w) {
        const uint32_t s = 1u + (uint32_t)kBlock * w;
        const uint32_t passes_w = w <= 8u || passes < 2u ? passes : (max_passes < kPartSearch ? 1u : 2u);
        for (uint32_t pass = 0; pass < passes_w; ++pass) {
            const uint32_t X0 = X + 2048u * pass;
            uint32_t a = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < kPartEvid; k += 4u) {                   // (a wrong width's candidates are gone after a dozen bits)
#pragma unroll
                for (uint32_t i = 0; i < 4u; ++i) a &= part_bits(W, s_chunk, X0 + 32u * lane + (k + i) * s);
                if (!__ballot(a != 0u)) break;
            }
            const uint64_t hits = __ballot(a != 0u);
            if (hits) {
                const int l0 = __builtin_ctzll(hits);
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)a, l0);
                const uint32_t r0 = 32u * (uint32_t)l0 + (uint32_t)__builtin_ctz(a0);
                __builtin_amdgcn_wave_barrier();
                s_pm[lane] = a;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t h = part_header_phase(s_pm, r0, s, w);
                if (h != ~0u) return PartState{X0 + h + kPartSkip * s, w};
                ambiguous = true;               // (cannot tell the header here: the range's later passes, then other widths)
            }
        }
    }
    return ambiguous ? PartState{X, kPartWeak | kPartAmbig} : plain;
}

// One (width, pass) of that search, for the index route's staged variant: kPartEvid header bits 1 at stride 1 + 12 w from the 2048
// positions behind X0.  Returns the state's position or ~0; depth = evidence bits the last candidates survived.
__device__ __forceinline__ uint32_t part_try(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t* __restrict__ s_pm, uint32_t X0, uint32_t w,
                                             uint32_t& depth, uint32_t& surv12) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t s = 1u + (uint32_t)kBlock * w;
    uint32_t a = 0xFFFFFFFFu, k = 0;
    surv12 = 0;
    for (; k < kPartEvid; k += 4u) {
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) a &= part_bits(W, s_chunk, X0 + 32u * lane + (k + i) * s);
        if (!__ballot(a != 0u)) break;
        if (k == 8u) surv12 = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan((uint32_t)__builtin_popcount(a)), 63);   // positions with 12 header bits 1 at this stride
    }
    depth = k;
    const uint64_t hits = __ballot(a != 0u);
    if (!hits) return ~0u;
    const int l0 = __builtin_ctzll(hits);
    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)a, l0);
    const uint32_t r0 = 32u * (uint32_t)l0 + (uint32_t)__builtin_ctz(a0);
    __builtin_amdgcn_wave_barrier();
    s_pm[lane] = a;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t h = part_header_phase(s_pm, r0, s, w);
    return h != ~0u ? X0 + h + kPartSkip * s : ~0u;
}

// The index route's search (k_chain_guess), staged by what the data show: (1) widths up to 8, the first 4096 positions -- two
// thirds of the cuts of a stack with a width change every nine blocks end here, and header-dense data, where no run exists, pays
// for no more than this; (2) the width whose candidates survived longest, if they survived 20 header bits (chance: 2^-20 per
// candidate, 32 K candidates), over the rest of the range: runs of that width exist nearby; (3) wider widths, one pass each,
// only if no small width has runs here (fewer than 20 positions of a pass with 12 header bits 1: pedestals, wide data --
// header-dense narrow data shows 20 and more and mostly skips this stage).  A cut that ends without a run starts from a plain guess.
struct ChainVote;
__device__ __forceinline__ bool chain_vote_look(ChainVote& v);
__device__ __forceinline__ PartState chain_guess(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t* __restrict__ s_pm, uint32_t X, uint32_t limit,
                                                 uint32_t max_w, uint32_t reach, ChainVote* vote = nullptr) {   // (vote: a header-dense stack needs no guesses)
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t s_max = 1u + (uint32_t)kBlock * max_w;
    const PartState plain{X, kPartWeak};
#ifdef TRPX_PART_FORCE_WEAK
    return plain;
#endif
    uint32_t passes = kPartSearch;
    while (passes && (uint64_t)X + 2048ull * passes + (uint64_t)kPartEvid * s_max + 128u > (uint64_t)limit) --passes;   // too close to the frame's end
    while (passes && 2048ull * passes + (uint64_t)kPartSkip * s_max > (uint64_t)reach) --passes;
    if (X + 256u > limit) return plain;
    part_fill(W, s_chunk, X);
    {
        const uint32_t a = lane < 4u ? part_bits(W, s_chunk, X + 32u * lane) : 0xFFFFFFFFu;
        if (!__ballot(a != 0xFFFFFFFFu)) return PartState{X, 0u};            // inside a run of empty blocks (parts of any size: an all-zero frame's are short)
    }
    if (passes == 0u) return plain;
    const uint32_t w1 = max_w < 8u ? max_w : 8u, p1 = passes < 2u ? passes : 2u;
    uint32_t best_w = 0, best_d = 0, most12 = 0;
    for (uint32_t w = 1; w <= w1; ++w)
        for (uint32_t pass = 0; pass < p1; ++pass) {
            uint32_t d, n12;
            if (vote && chain_vote_look(*vote)) return plain;
            const uint32_t q = part_try(W, s_chunk, s_pm, X + 2048u * pass, w, d, n12);
            if (q != ~0u) return PartState{q, w};
            if (d > best_d) { best_d = d; best_w = w; }
            most12 = n12 > most12 ? n12 : most12;
        }
    if (best_d >= 20u)
        for (uint32_t pass = p1; pass < passes; ++pass) {
            uint32_t d, n12;
            const uint32_t q = part_try(W, s_chunk, s_pm, X + 2048u * pass, best_w, d, n12);
            if (q != ~0u) return PartState{q, best_w};
        }
    if (most12 < 20u)                                                         // (no small width has runs here: wide data, a pedestal -- whose payload shows up to 16 by itself)
        for (uint32_t w = w1 + 1u; w <= max_w; ++w) {
            uint32_t d, n12;
            const uint32_t q = part_try(W, s_chunk, s_pm, X, w, d, n12);
            if (q != ~0u) return PartState{q, w};
            most12 = n12 > most12 ? n12 : most12;
        }
    // How run-dominated?  What tells is how MANY of a pass's ~55 block starts carry 12 header bits 1 at one stride: ~45 where one
    // block in sixty changes its width (synth-v1), ~28 at one in nine (eight 4096^2 int32 frames), ~22 in Poisson(3) counts, a
    // handful in noise -- the deepest single candidate does not tell: runs of 28 blocks occur a few times per thousand cuts in
    // Poisson(3) counts too.  Only the first kind gets the flag (a walk that stops instead of starting again, k_chain_walk):
    // there a false chain may not merge for a whole part; in the others it merges within a few K bits, and a stopped part
    // costs its repair a whole part's walk (71 of 4200 parts stopped: the repair launch 236 instead of ~20 us).
    // (32-bit pixels: from 24 on -- their parts hold half the blocks per bit, a stopped part's repair is a short walk (65 us for
    // eight 4096^2 frames' parts, where two walks that kept starting again for 100 Kbits made k_chain_walk 187 instead of ~100 us))
#ifdef TRPX_GUESS_STATS    // (not part of TRPX_PART_STATS: a printf in front of the start state's publication holds up every wavefront that waits for it -- hundreds ran into the diagnostic build's 30 s bound)
    if (lane == 0 && (X & 0xFFu) < 24u) printf("guess: no run behind %u: best width %u depth %u, most12 %u, passes %u, max_w %u, limit %u reach %u\n", X, best_w, best_d, most12, passes, max_w, limit, reach);
#endif
    return most12 >= (max_w > 16u ? 24u : 40u) ? PartState{X, kPartWeak | kPartRuns} : plain;
}

// Walks the chain from (pos, w) and counts the blocks that start in front of frame bit T; leaves the state at the first block
// start >= T, and in ck[] the state and count at the first step end behind every `ck_every` bits (a state of the chain at a
// block boundary: another chain that has merged with this one passes through it).  dense: more than one explicit header in 6
// blocks over 2048 blocks and more -- the walk stops (two serial walks of such a part, this one and the decoder's,
// cost more than the position-parallel walk of the frame).  tolerant: the walk started from a plain guess, i.e. on a chain
// that is not the frame's until it has merged with it -- such a chain may hold anything, also widths the pixel type does not
// have: it goes on from there with width 0 (what it counts in front of the merge is never used).  The steps are decode_frame.hip's (one per run of equal widths + the explicit header that ends it, the header
// parsed on the scalar unit) without the position entries; the fast loop only takes steps whose 64 candidates all lie in
// front of T and inside the window, the general step does the rest.
//
// STORE (the index route, see k_chain_walk): every step also leaves ent[j] = width of the block BEFORE the part's block j, for the 64
// candidates of the step (one byte store per lane, no mask: what lies behind the step's last block is overwritten by the next
// step, a wavefront's stores to one address arrive in program order), indices clamped to ent_cap - 1 -- a part that counts more
// blocks than its entry array holds has no entries, which the caller sees from the count.  ent[count] = the width in front of
// the block the walk stopped at.  ck == nullptr: no checkpoints and no density stop (ck_cap: entries ck[] has room for;
// stop_dense: the serial walker's hand-over test, the old parts route only).
// ---- is the stack header-dense?  (the index route's vote, k_chain_walk) ----------------------------------------------------------------
// The serial walkers of the index route take a step per explicit header: 370 us for 200 x (1030 x 1065) Poisson(3) frames (one block
// in four starts with one), which the lane-per-segment walk of decode_seg.hip (k_seg_wg: one workgroup per frame) does in 290.  On
// run-dominated frames it is the other way round (plain guesses inside runs do not merge: 3.2 ms for 200 synth-v1 frames).  What
// tells the two apart is the share of explicit headers on a TRUE chain (Poisson(3): one block in four; the int32 test frames: one in
// nine; synth-v1: one in sixty) -- the position-parallel walk's run search does not (random bits pass its eight-header test
// somewhere in most windows) --, and part 0 of every frame starts ON the true chain.  So the wavefronts of part 0 of up to sixteen
// frames spread over the stack first walk their share of ~1000 blocks and VOTE (blocks and explicit headers into one atomic
// word; ~10 us), the last voter writes the verdict (more than one block in six: header-dense), and every part looks at it between
// the passes of its guess and at its checkpoints until it reads "run-dominated": a header-dense stack's walkers stop within
// ~20 us of the launch, k_chain_resolve lists its frames (bit 31) and k_seg_wg walks them.  The verdict is the STACK's: a frame judged alone by a hundred blocks is misjudged once in a few
// hundred, and ONE frame on the other route costs the call that route's whole latency.  Only the speed hangs on it -- either walk is
// checked the same way --; where the frames' heads mislead, blank heads keep the stack on this route (round 5's time) and k_seg_wg
// hands a frame back whose links do not close.  The two words live in front of the hand-over list (codec_common.hpp: [-3] votes,
// [-4] verdict), cleared by the call's first launch.
constexpr uint32_t kChainVoters = 16;
__host__ __device__ inline uint32_t chain_voters(uint32_t n_frames) { return n_frames < kChainVoters ? n_frames : kChainVoters; }
struct ChainVote {
    const uint64_t* words;     // words[0] = verdict (0: none yet, 1: run-dominated, 2: header-dense), words[1] = voters << 56 | blocks << 28 | explicit headers
    bool settled;              // the verdict was "run-dominated" when last looked at: it is final, no more looks
};
__device__ __forceinline__ uint32_t chain_verdict(const uint64_t* words) {
#ifdef TRPX_CHAIN_NO_CLASSIFY                                  // (test build, make noclassify: every stack through the serial walkers, as in round 5)
    return 1u;
#elif defined(TRPX_CHAIN_ALL_DENSE)                            // (test build, make alldense: every stack to k_seg_wg)
    return 2u;
#else
    return (uint32_t)__hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// A look at the verdict (at a checkpoint of a walk, between the passes of a guess); true: the stack is header-dense, stop.
__device__ __forceinline__ bool chain_vote_look(ChainVote& v) {
    if (v.settled) return false;
    const uint32_t verdict = chain_verdict(v.words);
    v.settled = verdict == 1u;
    return verdict == 2u;
}
// A voter's vote: b blocks with n_exp explicit headers; the last of the stack's voters writes the verdict.
__device__ __forceinline__ void chain_vote_cast(uint64_t* words, uint32_t voters, uint32_t b, uint32_t n_exp) {
    if (lane_id() != 0) return;
    const uint64_t mine = (1ull << 56) | ((uint64_t)(b < 0xFFFFFu ? b : 0xFFFFFu) << 28) | (uint64_t)(n_exp < 0xFFFFFu ? n_exp : 0xFFFFFu);
    const uint64_t all = __hip_atomic_fetch_add(words + 1, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
    if ((uint32_t)(all >> 56) == voters) {
        const uint64_t blocks = (all >> 28) & 0xFFFFFFFull, expl = all & 0xFFFFFFFull;
        __hip_atomic_store(words, blocks >= 512u && 6u * expl > blocks ? 2ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool STORE = false>
__device__ __forceinline__ void part_walk(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t& pos, uint32_t& w_prev, uint32_t T,
                                          uint32_t limit, uint32_t max_w, uint32_t& count, bool& bad, bool& dense,
                                          PartCk* __restrict__ ck, uint32_t ck_every, uint32_t& n_ck, bool tolerant,
                                          uint32_t ck_cap = kPartCk, bool stop_dense = true, uint8_t* __restrict__ ent = nullptr,
                                          uint32_t ent_cap = 1u, bool abort_illegal = false, uint32_t prio_span = 0u,
                                          uint32_t* __restrict__ exp_out = nullptr, bool ratio_stop = false, ChainVote* vote = nullptr) {
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t b = 0, n_exp = 0, b_ref = 0, exp_ref = 0;
    uint32_t ck_next = ck ? pos + ck_every : 0xFFFFFFFFu;
    [[maybe_unused]] const uint32_t capm1 = ent_cap - 1u;
    n_ck = 0;
    while (pos < T && !bad) {
        if (pos >= ck_next) {
            if (n_ck < ck_cap && lane == 0) part_ck_store(ck + n_ck, PartCk{pos, w_prev, b});
            n_ck = n_ck < ck_cap ? n_ck + 1u : n_ck;
            ck_next = pos + ck_every;
            if (vote && chain_vote_look(*vote)) { dense = true; break; }      // (k_chain_walk: a header-dense stack is k_seg_wg's)
            if (ratio_stop) {
                // (run-dominated locales, k_chain_walk: more than 35 % explicit headers over a checkpoint interval is a chain that is
                // not the frame's -- a false chain reads one at every other block -- even if it has not read an illegal width yet)
                if (b - b_ref >= 32u && (n_exp - exp_ref) * 20u > (b - b_ref) * 7u) { dense = true; break; }
                b_ref = b; exp_ref = n_exp;
            }
            if (prio_span) {
                // The SIMD's arbiter serves equal-priority waves oldest first: of the ~17 walkers of a CU the last to arrive finished
                // at 115 - 137 us, the median at 65 (eight 4096 x 4096 frames, tools/chain_stamps.py) -- and a walker alone on its SIMD
                // cannot use the issue slots the others left.  Priority by what is LEFT of the part keeps them level.
                const uint32_t left = T - pos;
                if (4u * left > 3u * prio_span) __builtin_amdgcn_s_setprio(3);
                else if (2u * left > prio_span) __builtin_amdgcn_s_setprio(2);
                else if (4u * left > prio_span) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            // (a chain that is not the frame's yet may look like anything until it has merged: a tolerant walk's count starts
            // at its sixteenth checkpoint)
            // (the weakparts test build never stops one: every part there has to get through k_part_repair)
            if (tolerant && n_ck == 16u) { b_ref = b; exp_ref = n_exp; }
#ifdef TRPX_PART_FORCE_WEAK
            constexpr bool stoppable = false;
#else
            const bool stoppable = stop_dense && (!tolerant || n_ck >= 16u);
#endif
            if (stoppable && b - b_ref >= 2048u && (n_exp - exp_ref) * 6u > b - b_ref) { dense = true; break; }
        }
        uint32_t stride = 1u + (uint32_t)kBlock * w_prev;
        {   // the window holds the 64 candidates of a step from here
            const uint32_t need_lo = (W.frame_sh + pos) >> 5;
            const uint32_t need_hi = ((W.frame_sh + pos + 63u * stride) >> 5) + 2u;
            if ((int32_t)need_lo < W.c_lo || (int32_t)need_hi > W.c_hi) part_fill(W, s_chunk, pos);
        }
        {
            const uint32_t base8 = 8u * (uint32_t)(uintptr_t)&s_chunk[0];               // the window's LDS address, in bits
            uint32_t pw = W.frame_sh + pos - 32u * (uint32_t)W.c_lo + base8;             // the header's bit address in LDS
            uint32_t pw_end = 32u * (uint32_t)(W.c_hi - W.c_lo - 1) + base8;
            const uint64_t t_rel = (uint64_t)W.frame_sh + T - 32ull * (uint64_t)(int64_t)W.c_lo;   // (c_lo >= -3; T >= pos: inside or behind the window)
            if (t_rel + base8 < (uint64_t)pw_end) pw_end = (uint32_t)t_rel + base8;       // candidates stay in front of T
            {   // ... and the fast steps end where the next checkpoint is due
                const uint32_t c_pw = W.frame_sh + ck_next - 32u * (uint32_t)W.c_lo + base8 + 64u * stride;
                if (ck_next - pos < 0x4000000u && c_pw < pw_end) pw_end = c_pw;
            }
            uint32_t pw_max = pw_end - 63u * stride;
            uint32_t v_ls = __umul24(lane, stride);
            uint32_t s_bad = 0, t_first, t_h, t_t, t_p, t_a, t_bits;
            // (the steps' state lives in scalar registers -- wave-uniform by construction, which the compiler cannot always see through
            // the callers' loops: "illegal VGPR to SGPR copy" in some builds, not in others)
            auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
            pw = uni(pw); pw_end = uni(pw_end); pw_max = uni(pw_max); stride = uni(stride); b = uni(b); n_exp = uni(n_exp); w_prev = uni(w_prev);
#define TRPX_PART_STEP_ASM(ST_STEP, ST_WIDTH)                                                                                  \
                "s_cmp_lt_i32 %[pw], %[pwmax]\n\t"                                                                             \
                "s_cbranch_scc0 9f\n"                                                                                          \
                "1:\n\t"                                                                                                       \
                "v_add_u32 %[p], %[pw], %[ls]\n\t"                 /* this lane's candidate header */                          \
                "v_lshrrev_b32 %[a], 3, %[p]\n\t"                                                                              \
                "v_and_b32 %[a], 0x1ffffffc, %[a]\n\t"                                                                         \
                "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"                                                                    \
                ST_STEP                                              /* (in the LDS read's shadow) */                          \
                "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
                "v_alignbit_b32 %[bits], v63, v62, %[p]\n\t"                                                                   \
                "v_and_b32 %[a], 1, %[bits]\n\t"                                                                               \
                "v_cmp_eq_u32 vcc, 0, %[a]\n\t"                   /* lanes whose block has an explicit header (Terse.hpp:361) */ \
                "s_cbranch_vccz 5f\n\t"                                                                                        \
                "s_ff1_i32_b64 %[first], vcc\n\t"                                                                              \
                "v_readlane_b32 %[h], %[bits], %[first]\n\t"                                                                   \
                "s_bfe_u32 %[w], %[h], 0x30001\n\t"               /* Terse.hpp:362 */                                          \
                "s_cmp_lg_u32 %[w], 7\n\t"                                                                                     \
                "s_cbranch_scc0 6f\n\t"                                                                                        \
                "s_addc_u32 %[b], %[b], %[first]\n\t"             /* b += first + 1 (SCC = 1) */                               \
                "s_add_i32 %[nexp], %[nexp], 1\n\t"                                                                            \
                "s_mul_i32 %[t], %[first], %[stride]\n\t"                                                                      \
                "s_mul_i32 %[stride], %[w], 12\n\t"                                                                            \
                "s_add_i32 %[pw], %[pw], %[t]\n\t"                                                                             \
                "s_add_i32 %[stride], %[stride], 1\n\t"                                                                        \
                "s_add_i32 %[pw], %[pw], %[stride]\n\t"                                                                        \
                "s_add_i32 %[pw], %[pw], 3\n"                      /* pos += first * stride + 4 + 12 w */                       \
                "3:\n\t"                                                                                                       \
                ST_WIDTH                                                                                                       \
                "v_mul_u32_u24 %[ls], %[stride], %[lane]\n\t"                                                                  \
                "s_mul_i32 %[t], %[stride], 63\n\t"                                                                            \
                "s_sub_i32 %[pwmax], %[pwend], %[t]\n"                                                                         \
                "4:\n\t"                                                                                                       \
                "s_cmp_lt_i32 %[pw], %[pwmax]\n\t"                                                                             \
                "s_cbranch_scc1 1b\n\t"                                                                                        \
                "s_branch 9f\n"                                                                                                \
                "5:\n\t"                                           /* 64 blocks repeat the width */                            \
                "s_lshl_b32 %[t], %[stride], 6\n\t"                                                                            \
                "s_add_i32 %[pw], %[pw], %[t]\n\t"                                                                             \
                "s_add_i32 %[b], %[b], 64\n\t"                                                                                 \
                "s_branch 4b\n"                                                                                                \
                "6:\n\t"                                           /* widths >= 7: Terse.hpp:364-370 */                        \
                "s_bfe_u32 %[t], %[h], 0x20004\n\t"                                                                            \
                "s_add_i32 %[w], %[t], 7\n\t"                                                                                  \
                "s_mul_i32 %[t], %[first], %[stride]\n\t"                                                                      \
                "s_add_i32 %[pw], %[pw], %[t]\n\t"                                                                             \
                "s_add_i32 %[pw], %[pw], 6\n\t"                                                                                \
                "s_cmp_lg_u32 %[w], 10\n\t"                                                                                    \
                "s_cbranch_scc1 7f\n\t"                                                                                        \
                "s_bfe_u32 %[t], %[h], 0x60006\n\t"                                                                            \
                "s_add_i32 %[w], %[t], 10\n\t"                                                                                 \
                "s_add_i32 %[pw], %[pw], 6\n"                                                                                  \
                "7:\n\t"                                                                                                       \
                "s_cmp_gt_u32 %[w], %[maxw]\n\t"                                                                               \
                "s_cbranch_scc1 8f\n\t"                                                                                        \
                "s_add_i32 %[b], %[b], %[first]\n\t"                                                                           \
                "s_add_i32 %[b], %[b], 1\n\t"                                                                                  \
                "s_add_i32 %[nexp], %[nexp], 1\n\t"                                                                            \
                "s_mul_i32 %[stride], %[w], 12\n\t"                                                                            \
                "s_add_i32 %[pw], %[pw], %[stride]\n\t"                                                                        \
                "s_add_i32 %[stride], %[stride], 1\n\t"                                                                        \
                "s_branch 3b\n"                                                                                                \
                "8:\n\t"                                                                                                       \
                "s_mov_b32 %[bad], 1\n"                                                                                        \
                "9:\n"
            if constexpr (STORE) {
                uint32_t t_eo, v_wv = w_prev;
                const uint32_t capm1_u = uni(capm1), max_w_u = uni(max_w);
                const uint64_t ent_a = (uint64_t)(uintptr_t)ent;
                uint8_t* const ent_u = reinterpret_cast<uint8_t*>((uintptr_t)((uint64_t)uni((uint32_t)ent_a) | ((uint64_t)uni((uint32_t)(ent_a >> 32)) << 32)));
                asm volatile(TRPX_PART_STEP_ASM("v_add_u32 %[eo], %[b], %[lane]\n\t"
                                                "v_min_u32 %[eo], %[capm1], %[eo]\n\t"
                                                "global_store_byte %[eo], %[wv], %[ebase]\n\t",
                                                "v_mov_b32 %[wv], %[w]\n\t")
                    : [pw] "+s"(pw), [b] "+s"(b), [nexp] "+s"(n_exp), [w] "+s"(w_prev), [stride] "+s"(stride), [pwmax] "+s"(pw_max),
                      [ls] "+v"(v_ls), [bad] "+s"(s_bad), [first] "=&s"(t_first), [h] "=&s"(t_h), [t] "=&s"(t_t), [p] "=&v"(t_p),
                      [a] "=&v"(t_a), [bits] "=&v"(t_bits), [eo] "=&v"(t_eo), [wv] "+v"(v_wv)
                    : [lane] "v"(lane), [pwend] "s"(pw_end), [maxw] "s"(max_w_u), [capm1] "s"(capm1_u), [ebase] "s"(ent_u)
                    : "vcc", "scc", "memory", "v62", "v63");
            } else {
                const uint32_t max_w_u = uni(max_w);
                asm volatile(TRPX_PART_STEP_ASM("", "")
                    : [pw] "+s"(pw), [b] "+s"(b), [nexp] "+s"(n_exp), [w] "+s"(w_prev), [stride] "+s"(stride), [pwmax] "+s"(pw_max),
                      [ls] "+v"(v_ls), [bad] "+s"(s_bad), [first] "=&s"(t_first), [h] "=&s"(t_h), [t] "=&s"(t_t), [p] "=&v"(t_p),
                      [a] "=&v"(t_a), [bits] "=&v"(t_bits)
                    : [lane] "v"(lane), [pwend] "s"(pw_end), [maxw] "s"(max_w_u)
                    : "vcc", "scc", "memory", "v62", "v63");
            }
#undef TRPX_PART_STEP_ASM
            pos = pw - base8 + 32u * (uint32_t)W.c_lo - W.frame_sh;
            if (s_bad) {
                if (!tolerant) { bad = true; break; }
                if (abort_illegal) { dense = true; break; }                               // (a chain that is not the frame's: see k_chain_walk)
                w_prev = 0u;                                                              // (pos: behind the illegal header)
            }
        }
        if (pos >= T) break;
        if (pos > limit) { bad = true; break; }
        if (pos >= ck_next) continue;                                             // (the fast steps stopped for a checkpoint: no general step)
        stride = 1u + (uint32_t)kBlock * w_prev;
        {
            const uint32_t need_lo = (W.frame_sh + pos) >> 5;
            const uint32_t need_hi = ((W.frame_sh + pos + 63u * stride) >> 5) + 2u;
            if ((int32_t)need_lo < W.c_lo || (int32_t)need_hi > W.c_hi) { part_fill(W, s_chunk, pos); continue; }   // the window's end stopped the fast steps
        }
        // ---- general step: the candidates in front of T (Terse.hpp:360-372) ----
        const uint32_t lpos = pos + lane * stride;
        const uint32_t bits = part_bits(W, s_chunk, lpos);
        const uint32_t k_raw = (T - pos + stride - 1u) / stride;              // candidates with lpos < T (>= 1)
        const uint32_t kT = k_raw < 64u ? k_raw : 64u;
        const uint64_t valid = kT >= 64u ? ~0ull : ((1ull << kT) - 1ull);
        const uint64_t stop = ~(__ballot((bits & 1u) != 0u) & valid);
        const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        if constexpr (STORE) {                                                // the width in front of this step's blocks
            const uint32_t n_done = first < kT ? first + 1u : kT;
            if (lane < n_done) ent[b + lane < capm1 ? b + lane : capm1] = (uint8_t)w_prev;
        }
        if (first < kT) {                                                     // explicit header at candidate `first`
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)bits, (int)first);
            uint32_t w = (eb >> 1) & 7u, hl = 4;                              // Terse.hpp:362-370
            if (w == 7u) {
                w += (eb >> 4) & 3u; hl = 6;
                if (w == 10u) { w += (eb >> 6) & 63u; hl = 12; }
            }
            if (w > max_w) {
                if (!tolerant) { bad = true; break; }
                if (abort_illegal) {                                          // (stops behind the illegal header, as the fast steps do)
                    pos += first * stride + hl; w_prev = 0u; b += first + 1u;
                    dense = true;
                    break;
                }
                w = 0u;
            }
            pos += first * stride + hl + (uint32_t)kBlock * w;
            w_prev = w;
            b += first + 1u;
            ++n_exp;
        } else {                                                              // they all repeat the width
            pos += kT * stride;
            b += kT;
        }
        if (pos > limit) { bad = true; break; }
    }
    count = b;
    if (exp_out) *exp_out = n_exp;                                            // explicit headers among the blocks counted
    if constexpr (STORE) {
        if (lane == 0) ent[b < capm1 ? b : capm1] = (uint8_t)w_prev;
    }
}

// Counts the blocks of a part again from the true state (pos, w) -- general steps only -- until the chain passes through a
// checkpoint of the part's first walk (the chains have merged: the rest of that walk is this chain's) or reaches T.
// Returns 1: merged, `count` = the part's blocks; 2: at T, `count` / (pos, w) are the part's count and end state; 3: failed.
__device__ __forceinline__ uint32_t part_rewalk(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t& pos, uint32_t& w_prev, uint32_t T,
                                                uint32_t limit, uint32_t max_w, const PartCk* __restrict__ ck, uint32_t n_ck,
                                                uint32_t walk_cnt, uint32_t& count) {
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t b = 0, ci = 0;
    PartCk c{0xFFFFFFFFu, 0u, 0u};
    if (n_ck) c = ck[0];
    while (pos < T) {
        while (ci < n_ck && c.pos < pos) { ++ci; c = ci < n_ck ? ck[ci] : PartCk{0xFFFFFFFFu, 0u, 0u}; }
        const uint32_t stride = 1u + (uint32_t)kBlock * w_prev;
        {
            const uint32_t need_lo = (W.frame_sh + pos) >> 5;
            const uint32_t need_hi = ((W.frame_sh + pos + 63u * stride) >> 5) + 2u;
            if ((int32_t)need_lo < W.c_lo || (int32_t)need_hi > W.c_hi) part_fill(W, s_chunk, pos);
        }
        const uint32_t lpos = pos + lane * stride;
        const uint32_t bits = part_bits(W, s_chunk, lpos);
        const uint32_t k_raw = (T - pos + stride - 1u) / stride;              // candidates with lpos < T (>= 1)
        const uint32_t kT = k_raw < 64u ? k_raw : 64u;
        const uint64_t valid = kT >= 64u ? ~0ull : ((1ull << kT) - 1ull);
        const uint64_t stop = ~(__ballot((bits & 1u) != 0u) & valid);
        const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        const uint32_t n_run = first < kT ? first : kT - 1u;                  // candidates 0 .. n_run are block starts in front of T, width before them: w_prev
        // a checkpoint among them?
        while (ci < n_ck && c.pos <= pos + n_run * stride) {
            if ((c.pos - pos) % stride == 0u && c.w == w_prev) { count = b + (c.pos - pos) / stride + (walk_cnt - c.cnt); return 1u; }
            ++ci; c = ci < n_ck ? ck[ci] : PartCk{0xFFFFFFFFu, 0u, 0u};
        }
        if (first < kT) {                                                     // explicit header at candidate `first` (Terse.hpp:362-370)
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)bits, (int)first);
            uint32_t w = (eb >> 1) & 7u, hl = 4;
            if (w == 7u) {
                w += (eb >> 4) & 3u; hl = 6;
                if (w == 10u) { w += (eb >> 6) & 63u; hl = 12; }
            }
            if (w > max_w) return 3u;
            pos += first * stride + hl + (uint32_t)kBlock * w;
            w_prev = w;
            b += first + 1u;
        } else {
            pos += kT * stride;
            b += kT;
        }
        if (pos > limit) return 3u;
    }
    count = b;
    return 2u;
}

// A frame many of whose cuts found no run to start in (run-dominated stacks: one cut in a thousand) is header-dense: 1; one
// many of whose cuts found runs but not the headers in them is run-dominated with a problem (part_header_phase): 2; else 0.
__device__ __forceinline__ uint32_t part_frame_kind(const PartState* __restrict__ sf, uint32_t P) {
#ifdef TRPX_PART_FORCE_WEAK
    return 0u;
#endif
    uint32_t n_weak = 0, n_amb = 0;
    for (uint32_t q = (uint32_t)lane_id(); q < P; q += kWave) {
        const uint32_t w = sf[q].w;
        n_weak += (w & kPartWeak) != 0u && (w & kPartAmbig) == 0u ? 1u : 0u;
        n_amb += (w & kPartAmbig) != 0u ? 1u : 0u;
    }
    n_weak = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n_weak), 63);
    n_amb = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n_amb), 63);
    return 4u * n_weak > P ? 1u : (4u * n_amb > P ? 2u : 0u);
}

// The frame's cut positions: X_p = p * L.
__device__ __forceinline__ uint32_t part_len_bits(uint32_t limit, uint32_t P) {
    return (uint32_t)((((uint64_t)limit + P - 1u) / P + 127u) & ~127ull);
}

struct PartFrame { PartWin W; uint32_t limit, L; bool ok; };
__device__ __forceinline__ PartFrame part_frame(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                const uint64_t* __restrict__ frame_offsets, uint32_t frame, uint32_t P) {
    PartFrame f{};
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    f.ok = fe > fo && fe <= terse_bytes && 8 * (fe - fo) < 0xF0000000ull;
    if (!f.ok) return f;
    f.W.s32 = reinterpret_cast<const uint32_t*>(terse);
    f.W.n_dw = (terse_bytes + 3) / 4;
    f.W.frame_dw = (8 * fo) >> 5;
    f.W.frame_sh = (uint32_t)((8 * fo) & 31);
    f.W.base16 = ((uintptr_t)terse & 15) == 0;
    f.W.c_lo = f.W.c_hi = 0;
    f.limit = (uint32_t)(8 * (fe - fo));
    f.L = part_len_bits(f.limit, P);
    return f;
}

__global__ __launch_bounds__(kWave) void k_part_guess(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                      const uint64_t* __restrict__ frame_offsets, uint32_t max_w, uint32_t P,
                                                      PartState* __restrict__ states) {
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kPartChunkDw + 4];
    __shared__ uint32_t s_pm[kWave + 2];
    const uint32_t frame = blockIdx.x / P, p = blockIdx.x % P;
    const uint32_t lane = (uint32_t)lane_id();
    if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
    if (lane < 2u) s_pm[kWave + lane] = 0u;
    PartState s{0u, 0u};                                                      // a frame starts at bit 0 with width 0 (Terse.hpp:359, :505)
    if (p != 0u) {
        PartFrame f = part_frame(terse, terse_bytes, frame_offsets, frame, P);
        s = f.ok ? part_guess(f.W, s_chunk, s_pm, p * f.L, f.limit, max_w) : PartState{0u, kPartWeak};
    }
    if (lane == 0) states[blockIdx.x] = s;
}

__global__ __launch_bounds__(kWave) void k_part_walk(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, uint32_t max_w, uint32_t P,
                                                     const PartState* __restrict__ states, PartWalk* __restrict__ walks,
                                                     PartCk* __restrict__ cks) {
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kPartChunkDw + 4];
    const uint32_t frame = blockIdx.x / (P - 1u), p = blockIdx.x % (P - 1u);
    const uint32_t lane = (uint32_t)lane_id();
    if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
    PartWalk r{};
    r.flags = 1u;
    PartFrame f = part_frame(terse, terse_bytes, frame_offsets, frame, P);
    const PartState s = states[(uint64_t)frame * P + p], t = states[(uint64_t)frame * P + p + 1u];
    // a frame a quarter of whose cuts found no run to start in is header-dense: not worth a walk (it takes the other route)
    const uint32_t kind = part_frame_kind(states + (uint64_t)frame * P, P);
    if (kind) r.flags = kind == 1u ? 2u : 4u;                               // (4: not dense, but not worth a walk either)
    else if (f.ok && t.pos > s.pos && t.pos < f.limit) {
        uint32_t pos = s.pos, w = s.w & ~(kPartWeak | kPartAmbig), cnt = 0, n_ck = 0;
        bool bad = false, dense = false;
        const uint32_t span = t.pos - s.pos;
        const uint32_t every = span / (kPartCk - 8u) > 4096u ? span / (kPartCk - 8u) : 4096u;
        // (only the frame's first part is known to start on the frame's chain: a run guess can be wrong too -- one in a few
        // thousand was, with 24 bits of evidence -- and its walk then has to get to the merge like a plain guess's)
        part_walk(f.W, s_chunk, pos, w, t.pos, f.limit, max_w, cnt, bad, dense, cks + (uint64_t)blockIdx.x * kPartCk, every, n_ck,
                  p != 0u);
        r.o_pos = pos; r.o_w = w; r.cnt = cnt; r.n_ck = n_ck; r.flags = (bad ? 1u : 0u) | (dense ? 2u : 0u);
    }
    if (lane == 0) walks[blockIdx.x] = r;
}

// One wavefront per part 1 <= p <= P - 2 (the frame's last part is not walked here: its start is simply OUT_(P-2)).
__global__ __launch_bounds__(kWave) void k_part_repair(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       const uint64_t* __restrict__ frame_offsets, uint32_t max_w, uint32_t P,
                                                       const PartState* __restrict__ states, const PartWalk* __restrict__ walks,
                                                       const PartCk* __restrict__ cks, PartFix* __restrict__ fixes) {
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kPartChunkDw + 4];
    const uint32_t frame = blockIdx.x / (P - 1u), p = blockIdx.x % (P - 1u);
    const uint32_t lane = (uint32_t)lane_id();
    if (p == 0u) return;                                                      // (part 0 starts in the true state)
    const uint64_t wi = (uint64_t)frame * (P - 1u) + p;
    const PartWalk prev = walks[wi - 1u], mine = walks[wi];
    const PartState s = states[(uint64_t)frame * P + p], t = states[(uint64_t)frame * P + p + 1u];
    PartFix x{0u, 0u, 0u, 0u};
    // a frame a quarter of whose links are open has a systematic problem with its cuts (see k_part_resolve: pedestals): counting
    // every part again from its neighbour's end state would be a second walk of the frame for nothing
    uint32_t n_open = 0;
    for (uint32_t q = 1u + lane; q + 1u < P; q += kWave) {
        const PartWalk wq = walks[(uint64_t)frame * (P - 1u) + q - 1u];
        const PartState sq = states[(uint64_t)frame * P + q];
        n_open += wq.flags == 0u && !(wq.o_pos == sq.pos && wq.o_w == sq.w) ? 1u : 0u;
    }
    n_open = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n_open), 63);
#ifdef TRPX_PART_FORCE_WEAK
    n_open = 0;
#endif
    if (part_frame_kind(states + (uint64_t)frame * P, P) != 0u || 4u * n_open > P) x.state = 3u;   // (the frame takes another route)
    else if (prev.flags == 0u && !(prev.o_pos == s.pos && prev.o_w == s.w)) { // an open link behind a walk that arrived somewhere
        x.state = 3u;
        if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
        PartFrame f = part_frame(terse, terse_bytes, frame_offsets, frame, P);
        if (f.ok && mine.flags == 0u && prev.o_pos < t.pos && t.pos < f.limit) {
            uint32_t pos = prev.o_pos, w = prev.o_w, cnt = 0;
            x.state = part_rewalk(f.W, s_chunk, pos, w, t.pos, f.limit, max_w, cks + wi * kPartCk, mine.n_ck < kPartCk ? mine.n_ck : kPartCk,
                                  mine.cnt, cnt);
            x.cnt = cnt; x.o_pos = pos; x.o_w = w;
        }
    }
    if (lane == 0) fixes[wi] = x;
}

// See the head of the file.  list[0] = count, list[1 + i] = frame | (dense: bit 31).
__global__ __launch_bounds__(kWave) void k_part_resolve(const PartState* __restrict__ states, const PartWalk* __restrict__ walks,
                                                        const PartFix* __restrict__ fixes, FrameGeom g, uint32_t P,
                                                        const uint64_t* __restrict__ frame_offsets, uint64_t terse_bytes,
                                                        PartDesc* __restrict__ parts, uint32_t* __restrict__ list,
                                                        uint32_t* __restrict__ status) {
    const uint32_t frame = blockIdx.x, lane = (uint32_t)lane_id();
    const PartState* __restrict__ sf = states + (uint64_t)frame * P;
    const PartWalk* __restrict__ wf = walks + (uint64_t)frame * (P - 1u);
    const PartFix* __restrict__ xf = fixes + (uint64_t)frame * (P - 1u);
    PartDesc* __restrict__ pf = parts + (uint64_t)frame * P;
    bool ok = true, dense = false;
    uint32_t running = 0;
    for (uint32_t base = 0; base < P - 1u; base += kWave) {
        const uint32_t p = base + lane;
        const bool valid = p < P - 1u;
        PartWalk r{};
        PartFix x{};
        PartState s{}, t{};
        bool good = true;
        uint32_t cnt = 0;
        PartState st{0u, 0u}, en{0u, 0u};                                     // the part's true start and end states
        if (valid) {
            r = wf[p]; s = sf[p]; t = sf[p + 1u];
            if (p > 0u) x = xf[p];
            good = r.flags == 0u;
            if (p == 0u || x.state == 0u) {                                   // starts in its guess: S_0, or the link in front of it is closed
                st = PartState{s.pos, s.w & ~(kPartWeak | kPartAmbig)}; en = PartState{r.o_pos, r.o_w}; cnt = r.cnt;
            } else {
                const PartWalk q = wf[p - 1u];
                st = PartState{q.o_pos, q.o_w};
                cnt = x.cnt;
                if (x.state == 1u) en = PartState{r.o_pos, r.o_w};
                else if (x.state == 2u) { en = PartState{x.o_pos, x.o_w}; good = good && x.o_pos == r.o_pos && x.o_w == r.o_w; }   // (the next link was judged by r's end state)
                else good = false;
            }
            good = good && cnt >= 1u && cnt <= kPartMaxBlocks;
        }
        ok = ok && !__ballot(valid && !good);
        dense = dense || __ballot(valid && (r.flags & 2u) != 0u) != 0ull;
#ifdef TRPX_PART_STATS
        {   // diagnostic build: status[2..7] = fallback frames, plain guesses, bad / dense walks, repaired links, failed repairs, oversized parts
            const uint32_t n3 = (uint32_t)__builtin_popcountll(__ballot(valid && p > 0u && (s.w & kPartWeak) != 0u));
            const uint32_t n4 = (uint32_t)__builtin_popcountll(__ballot(valid && r.flags != 0u));
            const uint32_t n5 = (uint32_t)__builtin_popcountll(__ballot(valid && p > 0u && (x.state == 1u || x.state == 2u)));
            const uint32_t n6 = (uint32_t)__builtin_popcountll(__ballot(valid && p > 0u && x.state == 3u));
            const uint32_t n7 = (uint32_t)__builtin_popcountll(__ballot(valid && r.flags == 0u && (cnt < 1u || cnt > kPartMaxBlocks)));
            if (lane == 0) { atomicAdd(status + 3, n3); atomicAdd(status + 4, n4); atomicAdd(status + 5, n5); atomicAdd(status + 6, n6); atomicAdd(status + 7, n7); }
        }
#endif
        const uint32_t c = valid && good ? cnt : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (valid && good) {
            PartDesc d;
            d.frame = frame; d.b0 = running + inc - c; d.b1 = d.b0 + c;
            d.pos0 = st.pos; d.w0 = st.w; d.pos1 = en.pos; d.w1 = en.w; d.pad = 0u;
            pf[p] = d;
            if (p == P - 2u) {                                                // the frame's last part starts where this one ends
                PartDesc e;
                e.frame = frame; e.b0 = d.b1; e.b1 = g.n_blocks; e.pos0 = en.pos; e.w0 = en.w; e.pos1 = 0u; e.w1 = 0u; e.pad = 0u;
                pf[P - 1u] = e;
            }
        }
        running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (running >= g.n_blocks) ok = false;                                // (the last part holds at least the frame's last block)
    }
    if (ok && g.n_blocks - running > kPartMaxBlocks) ok = false;
    if (!ok) {
        // No part table for this frame.  Header-dense: the position-parallel walk + tiled extraction (the list).  Otherwise --
        // run-dominated, but the cuts did not work out: e.g. pixels with a pedestal, whose constant payload bits look like
        // header bits at the same stride, so that every cut starts beside the chain and stays there -- the frame is ONE unit for
        // its one workgroup's serial walker, as before there were parts (0.4 ms for 200 frames of 1030 x 1065 against 2.9 ms on
        // the list's route), provided its bits fit the decoder's 26-bit positions.
        const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
        const bool one_unit = !dense && fe > fo && fe <= terse_bytes && 8 * (fe - fo) + (1u << 17) < (1ull << 26);
        __builtin_amdgcn_s_waitcnt(0);
        for (uint32_t p = lane; p < P; p += kWave) {
            PartDesc d{};
            d.frame = frame;
            if (one_unit && p == 0u) d.b1 = g.n_blocks;
            pf[p] = d;
        }
        if (!one_unit && lane == 0) {
            list[1u + atomicAdd(&list[0], 1u)] = frame | (dense ? 0x80000000u : 0u);
            if (dense) atomicAdd(reinterpret_cast<unsigned long long*>(list) - 2, 1ull);   // (k_seg_wg's: their number, codec_common.hpp)
        }
#ifdef TRPX_PART_STATS
        if (lane == 0) atomicAdd(status + 2, 1u);
#endif
    }
}

hipError_t launch_build_parts(const DecodeArgs& a, uint32_t max_w, hipStream_t st) {
    const uint32_t P = a.parts_per_frame;
    if (P < 2u || !a.parts || !a.part_ws || !a.defer) return hipErrorInvalidValue;
    const PartWs l = part_ws_layout(a.n_frames, P);
    char* ws = static_cast<char*>(a.part_ws);
    PartState* states = reinterpret_cast<PartState*>(ws + l.states);
    PartWalk* walks = reinterpret_cast<PartWalk*>(ws + l.walks);
    PartFix* fixes = reinterpret_cast<PartFix*>(ws + l.fixes);
    PartCk* cks = reinterpret_cast<PartCk*>(ws + l.cks);
    const dim3 links(a.n_frames * (P - 1u));
    hipLaunchKernelGGL(k_part_guess, dim3(a.n_frames * P), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, max_w, P,
                       states);
    hipLaunchKernelGGL(k_part_walk, links, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, max_w, P,
                       static_cast<const PartState*>(states), walks, cks);
    hipLaunchKernelGGL(k_part_repair, links, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, max_w, P,
                       static_cast<const PartState*>(states), static_cast<const PartWalk*>(walks), static_cast<const PartCk*>(cks), fixes);
    hipLaunchKernelGGL(k_part_resolve, dim3(a.n_frames), dim3(kWave), 0, st, static_cast<const PartState*>(states),
                       static_cast<const PartWalk*>(walks), static_cast<const PartFix*>(fixes), a.geom, P, a.frame_offsets, (uint64_t)a.terse_bytes,
                       a.parts, a.defer, a.status);
    return hipGetLastError();
}

// =================================================================================================================================
// The INDEX route for large frames (round 5): one walk, then the extraction with the widths given.
//
// The parts route above walks every block twice -- k_part_walk counts, k_decode_parts' walker walks again to extract -- because a
// part's pixels cannot be placed before the blocks in front of it are counted.  Here the counting walk leaves what it saw: per
// block one byte, the width in front of it, in a per-part entry array (relative block numbers: the part's first block is not
// known yet).  Once k_chain_resolve has the block numbers, k_chain_index turns the entries into the decode index the extraction
// kernels consume (widths[] by absolute block, bit offset of every 256th block: encode_kernels.hpp) -- 35 vector instructions per
// 64 BLOCKS, no dependent chain -- and the frames are extracted as if the caller had brought the index (k_unpack_tiles /
// k_decode_frames_indexed).  With the decoder's grain out of the way the walk's parts are SMALL (about kChainBlocks blocks): a
// serial walker is a latency chain, ~130 ns per run of equal widths whatever else runs on the SIMD, so the walk's time is the
// longest part's, and ten times more, ten times shorter chains cost nothing but their start states:
//
//   k_chain_guess   a start state inside a run of equal widths behind every cut X_p = p * L (part_guess, as above), searched over at
//                   most half a part; the frame's last cut lies kChainTail bits in front of its end.
//   k_chain_walk    one wavefront per part but the last.  A part whose cut found no run -- header-dense data -- WARMS UP: it starts
//                   kChainWarm bits in front of its cut from a plain guess, tolerant, and takes the state in which that chain
//                   crosses the cut (false chains merge with the frame's at the first explicit header they meet on a block start:
//                   a median of 94 blocks in Poisson(3) counts, tools/merge_stats.py); the part in front of it arrives at the same
//                   block start if it is on the frame's chain.  Then the counting walk with the entry stores (part_walk<true>).
//   k_chain_repair  open links (OUT_(p-1) != the state part p started in): part p counted again from OUT_(p-1) until that chain
//                   passes through a checkpoint of the part's own walk, as above.
//   k_chain_resolve one wavefront per frame: S_0 true, links closed or repaired => by induction every part's start, count and end
//                   are the frame's chain's; prefix sums; the part table.  pad bit 0: the part's entries are the true chain's (it
//                   started in the state it was walked from, and they fit).
//   k_chain_index   one wavefront per part: entries -> widths[b0 ..), group offsets by a prefix sum over header + payload lengths,
//                   the end checked against the next part's start.  Parts without entries -- repaired, overflowed (runs of empty
//                   blocks: 1 bit each, 64 per step), the frame's last part (whose last block may be partial: only the block
//                   count tells where the frame ends) -- are walked from their true start, by count, writing the index directly.
//
// A frame where this does not work out is listed for the position-parallel walk (decode_seg.hip), which writes the same index;
// the extraction follows for all frames at once.  Guesses, warm-ups and repairs only ever steer the speed: every part's end is
// checked against its successor's start, the last one against S_f = 1 + bits/8 (Terse.hpp:547).
// =================================================================================================================================
#ifndef TRPX_CHAIN_WAVES
#define TRPX_CHAIN_WAVES 4352
#endif
#ifndef TRPX_CHAIN_WAVES_NARROW
#define TRPX_CHAIN_WAVES_NARROW 4096
#endif
constexpr uint32_t kChainCk = 64;              // checkpoints per part
constexpr uint32_t kChainTail = 2048;          // bits of the frame's last part (walked by count in k_chain_index)
constexpr uint32_t kChainPatience = 8192;      // bits into a part after which a walk that reads an illegal width stops instead of starting again
constexpr uint32_t kChainEntSlack = 80;

// Parts per frame: as many walkers as the GPU holds at once (8 KB of LDS each: 19 per CU) -- a second round of a few hundred
// stragglers doubled k_chain_walk's time (eight 4096 x 4096 frames, 5464 parts: 146 us; tools/chain_stamps.py) --, of 1 K ..
// 16 K blocks each; + the tail part.
uint32_t chain_parts_per_frame(const FrameGeom& g, size_t n_frames, size_t pixel_bytes) {
    if (g.n_blocks <= single_part_blocks(n_frames) || n_frames == 0) return 1u;
    // (8 / 16-bit pixels: 4096 -- the parts are also the units k_decode_parts extracts, eight workgroups per CU: two whole rounds
    // of them instead of two and a tenth; 200 x (1030 x 1065) 0.202 -> 0.193 ms, 128 x 2048^2 0.43 -> 0.405, Poisson(3) mid-size
    // 0.559 -> 0.537; the int32 frames, extracted by tiles, lose 6 % to the longer parts: 0.3105 -> 0.331)
#ifdef TRPX_DIAGNOSTICS
    static const uint64_t waves_env = getenv("TRPX_CHAIN_WAVES") ? (uint64_t)atoi(getenv("TRPX_CHAIN_WAVES")) : 0u;
    const uint64_t waves = waves_env ? waves_env : (pixel_bytes < 4 ? (uint64_t)TRPX_CHAIN_WAVES_NARROW : (uint64_t)TRPX_CHAIN_WAVES);
#else
    const uint64_t waves = pixel_bytes < 4 ? (uint64_t)TRPX_CHAIN_WAVES_NARROW : (uint64_t)TRPX_CHAIN_WAVES;
#endif
    const uint64_t lo = ((uint64_t)g.n_blocks + kPartBlocks - 1u) / kPartBlocks, hi = (uint64_t)g.n_blocks / 1024u;
    uint64_t n = waves / n_frames;
    n = n < lo ? lo : (n > hi ? hi : n);
    return (uint32_t)(n < 3u ? 4u : n + 1u);
}
__host__ __device__ inline uint32_t chain_ent_cap(uint32_t n_blocks, uint32_t P) { return 2u * ((n_blocks + P - 1u) / P) + kChainEntSlack; }
struct ChainFix {                              // what k_chain_repair leaves per (frame, part 1 <= p < P - 1)
    uint32_t state;                            // 0: link closed, nothing done; 1: merged into the part's walk; 2: walked to T_p on its own; 3: failed
    uint32_t cnt, o_pos, o_w;                  // the part's block count (and, state 2, its end state) from the true start
    uint32_t b_merge, ck_cnt;                  // state 1: the repair's block b_merge is the part's own walk's block ck_cnt (a checkpoint)
    uint32_t n_fix;                            // entries the repair left for its blocks [0, b_merge] (0: they did not fit)
    uint32_t pad;
};
struct ChainWs { size_t states, walks, fixes, cks, ents, fixents, modes, total; };
static ChainWs chain_ws_layout(const FrameGeom& g, size_t n_frames, size_t P) {
    ChainWs w;
    w.states = 0;                                                             // [states, cks): start states, walk records (with their ready / published bits) and link records, cleared in front of every call (launch_chain_zero)
    w.walks = align_up(w.states + n_frames * P * sizeof(PartState), 256);
    w.fixes = align_up(w.walks + n_frames * P * sizeof(PartWalk), 256);
    w.cks = align_up(w.fixes + n_frames * P * sizeof(ChainFix), 256);
    w.ents = align_up(w.cks + n_frames * P * kChainCk * sizeof(PartCk), 256);
    w.fixents = align_up(w.ents + n_frames * P * (size_t)chain_ent_cap(g.n_blocks, (uint32_t)P) + 16, 256);
    w.modes = align_up(w.fixents + n_frames * P * (size_t)chain_ent_cap(g.n_blocks, (uint32_t)P) + 16, 256);   // (a repair's entries: as many as a walk's)
    w.total = align_up(w.modes + 4 * n_frames, 256);
    return w;
}
size_t chain_workspace_bytes(const FrameGeom& g, size_t n_frames, size_t pixel_bytes) {
    const size_t P = chain_parts_per_frame(g, n_frames, pixel_bytes);
    return P > 1 ? chain_ws_layout(g, n_frames, P).total : 0;
}

// The frame's cuts: X_p = p * L for p < P, the last one kChainTail bits (or a sixteenth of a small frame) in front of the end.
__device__ __forceinline__ PartFrame chain_frame(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                 const uint64_t* __restrict__ frame_offsets, uint32_t frame, uint32_t P) {
    PartFrame f = part_frame(terse, terse_bytes, frame_offsets, frame, P);
    if (!f.ok) return f;
    const uint32_t tail = f.limit > 16u * kChainTail ? kChainTail : f.limit / 16u;
    f.L = (f.limit - tail + (P - 2u)) / (P - 1u);
    f.ok = f.L >= 128u && tail > P;                                         // (a frame of a few bits per part: another route; an all-zero 4096 x 4096 frame has 1 Kbit parts)
    return f;
}

// The blocks of a part counted again from the true state (pos, w), checkpoint by checkpoint of the part's own walk: fast steps to
// the next checkpoint's position (part_walk: it stops at the first block start at or behind it), and if the chain arrives IN the
// checkpoint's state the two have merged -- the rest of that walk is this chain's.  Leaves the entries of the blocks it counted in
// fix[] (part_walk<true>) as long as they fit.  Returns the ChainFix state.
__device__ __forceinline__ uint32_t chain_rewalk(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t& pos, uint32_t& w, uint32_t T,
                                                 uint32_t limit, uint32_t max_w, const PartCk* __restrict__ ck, uint32_t n_ck,
                                                 uint32_t walk_cnt, uint8_t* __restrict__ fix, uint32_t fix_cap, ChainFix& x,
                                                 uint32_t b = 0, bool fits = true) {   // (b, fits: blocks already counted by chain_walk_ahead)
    uint32_t k = 0;
    bool bad = false, dense = false;
    for (;;) {
        PartCk c{};
        while (k < n_ck && (c = part_ck_load(ck + k)).pos <= pos) {
            if (c.pos == pos && c.w == w) {                                   // (starts in a checkpoint: nothing to count)
                x.b_merge = b; x.ck_cnt = c.cnt; x.cnt = b + (walk_cnt - c.cnt); x.n_fix = fits ? b + 1u : 0u;
                return 1u;
            }
            ++k;
        }
        const uint32_t target = k < n_ck ? c.pos : T;                         // (c: checkpoint k, the first one behind pos)
        uint32_t cnt = 0, n0 = 0;
        if (fits && b + 160u < fix_cap)
            part_walk<true>(W, s_chunk, pos, w, target, limit, max_w, cnt, bad, dense, nullptr, 0u, n0, false, 0u, false, fix + b, fix_cap - b);
        else
            part_walk<false>(W, s_chunk, pos, w, target, limit, max_w, cnt, bad, dense, nullptr, 0u, n0, false, 0u, false);
        if (bad) return 3u;
        b += cnt;
        fits = fits && b + 1u < fix_cap;
        if (k >= n_ck) break;                                                 // at T on its own
    }
    x.cnt = b; x.o_pos = pos; x.o_w = w; x.b_merge = b; x.ck_cnt = 0u; x.n_fix = fits ? b + 1u : 0u;
    return 2u;
}

// A walk record / a start state of ANOTHER wavefront of this launch, once it has been published (bounded wait: a record that does
// not come reads as a bad walk, a state as position 0 -- the link fails and the frame takes the other route).
__device__ __forceinline__ PartWalk chain_walk_wait(const PartWalk* src) {
    const uint64_t* d = reinterpret_cast<const uint64_t*>(src);
    uint64_t v[4] = {0, 0, 0, 0};
    if (lane_id() == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            v[1] = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // cnt | flags << 32
            if (((v[1] >> 32) & kWalkPublished) != 0u) break;
            __builtin_amdgcn_s_sleep(64);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kChainWaitTicks) { v[1] = 1ull << 32; break; }   // 0.25 s: a bad walk
        }
        // (the record's other words -- and whatever the caller reads of the part's checkpoints -- only after the published bit has
        // been SEEN: the flag's value has arrived here, the statement keeps the compiler from moving the relaxed loads below
        // in front of it; the hardware returns loads in order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        v[0] = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[2] = __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[3] = __hip_atomic_load(d + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[2 * i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v[i]);
        u[2 * i + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v[i] >> 32));
    }
    PartWalk r;
    __builtin_memcpy(&r, u, 32);
    return r;
}
__device__ __forceinline__ void chain_walk_publish(PartWalk* dst, PartWalk r) {
    static_assert(sizeof(PartWalk) == 32, "four 64-bit words");
    r.flags |= kWalkPublished;
    uint64_t v[4];
    __builtin_memcpy(v, &r, 32);
    uint64_t* d = reinterpret_cast<uint64_t*>(dst);
    if (lane_id() == 0) {
        __hip_atomic_store(d, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(d + 3, v[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the record's other words and the checkpoints (write-through) have arrived
    if (lane_id() == 0) __hip_atomic_store(d + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... without waiting: false if the record is not there yet
__device__ __forceinline__ bool chain_walk_poll(const PartWalk* src, PartWalk& r) {
    const uint64_t* d = reinterpret_cast<const uint64_t*>(src);
    uint64_t v[4] = {0, 0, 0, 0};
    if (lane_id() == 0) {
        v[1] = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((v[1] >> 32) & kWalkPublished) != 0u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (as in chain_walk_wait)
            v[0] = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[2] = __hip_atomic_load(d + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[3] = __hip_atomic_load(d + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u[2 * i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v[i]);
        u[2 * i + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v[i] >> 32));
    }
    __builtin_memcpy(&r, u, 32);
    return (r.flags & kWalkPublished) != 0u;
}
// The part behind, counted from the true state while its own wavefront is still walking it: kChainAhead bits at a time, its record
// polled in between.  True: the record has come (`mine`); the chain stands at (pos, w), b blocks in, and chain_rewalk goes on from
// there with the record's checkpoints.  False: the chain is at T (or `bad`) and the record still is not there.
constexpr uint32_t kChainAhead = 8192;
__device__ __forceinline__ bool chain_walk_ahead(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t& pos, uint32_t& w, uint32_t T, uint32_t limit,
                                                 uint32_t max_w, uint8_t* __restrict__ fix, uint32_t fix_cap, uint32_t& b, bool& fits, bool& bad,
                                                 const PartWalk* rec, PartWalk& mine) {
    bool dense = false;
    for (;;) {
        if (chain_walk_poll(rec, mine)) return true;
        if (pos >= T || bad) return false;
        const uint32_t target = T - pos > kChainAhead ? pos + kChainAhead : T;
        uint32_t cnt = 0, n0 = 0;
        if (fits && b + 160u < fix_cap)
            part_walk<true>(W, s_chunk, pos, w, target, limit, max_w, cnt, bad, dense, nullptr, 0u, n0, false, 0u, false, fix + b, fix_cap - b);
        else
            part_walk<false>(W, s_chunk, pos, w, target, limit, max_w, cnt, bad, dense, nullptr, 0u, n0, false, 0u, false);
        b += cnt;
        fits = fits && b + 1u < fix_cap;
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    }
}
__device__ __forceinline__ PartState chain_state_wait(const PartState* src) {
    const uint64_t* d = reinterpret_cast<const uint64_t*>(src);
    uint64_t v = 0;
    if (lane_id() == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            v = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((v >> 32) & kPartReady) != 0u) break;
            __builtin_amdgcn_s_sleep(32);
            if (__builtin_amdgcn_s_memrealtime() - t0 > kChainWaitTicks) { v = 0; break; }           // 0.25 s of the 100 MHz counter
        }
    }
    return PartState{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32))};
}

// The link INTO part p (1 <= p <= P - 2), by the wavefront that has just walked part p - 1 (`prev`: its record) and arrived in front
// of it.  Closed -- the part's walk started where this chain arrives -- nothing is counted.  Open: the part's blocks are counted again
// from here, checkpoint by checkpoint of the part's own walk, until the chains have merged (chain_rewalk).  A part whose walk stopped
// on a false chain (flag 16) is counted as a whole (state 2: nothing to merge into), and the same wavefront goes on into the parts
// behind it for as long as they are stopped parts: their own wavefronts, which cannot know the true end of a stopped part, leave the
// link behind them alone.  Every part's ChainFix is written by exactly one wavefront.
// (Until round 5 a launch of its own behind the walk: its critical path -- the longest-lived false chain of 4000, a part's restart
// far into it, a stopped part -- was 47 us for eight 4096^2 frames and 130 us for 200 x (1030 x 1065) Poisson(3) frames, with the
// GPU all but idle; here it falls into the slack every wavefront but the launch's slowest has.)
__device__ __forceinline__ void chain_link_into(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                const uint64_t* __restrict__ frame_offsets, uint32_t max_w, uint32_t P,
                                                const PartState* states, const PartWalk* walks, const PartCk* cks, ChainFix* __restrict__ fixes,
                                                uint8_t* __restrict__ fixents, uint32_t fix_cap, uint32_t frame, uint32_t p, const PartWalk& prev,
                                                PartFrame& f, uint32_t* __restrict__ s_chunk) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint64_t slot0 = (uint64_t)frame * P;
    uint32_t pos = prev.o_pos, w = prev.o_w;                                  // the true end of the part in front (if ITS start was true: k_chain_resolve)
    // arrives in the state the part's walk was started in: closed (a walk that began again behind an illegal width, or stopped, did
    // so on a chain this one would then be too)
    const PartState g = chain_state_wait(states + slot0 + p);
    if ((prev.flags & 1u) != 0u || (g.pos == pos && (g.w & ~kPartFlags) == w)) {
        if (lane == 0) fixes[slot0 + p] = ChainFix{};
        return;
    }
    // A part that started from a plain guess (no run behind its cut: header-dense data, wide data) is not on the frame's chain until
    // it has merged with it, so this link is open: the count from here starts at once, ahead of the part's record (chain_walk_ahead)
    // -- waiting for it first left the launch's last wavefronts idle for 60 us and more, behind the slowest walks of all.
    const bool ahead = (g.w & kPartWeak) != 0u;
#ifdef TRPX_PART_STATS
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t cur = p;;) {
        ChainFix x{};
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);             // (wave-uniform by construction)
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
        cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
        const PartState t = chain_state_wait(states + slot0 + cur + 1u);
        const bool can_walk = f.ok && pos < t.pos && t.pos < f.limit;
        const uint32_t pos_in = pos, w_in = w;
        uint8_t* const fix = fixents + (slot0 + cur) * (uint64_t)fix_cap;
        uint32_t b = 0;
        bool fits = true, bad = false, have = false;
        PartWalk mine{};
        if (ahead && cur == p && can_walk) have = chain_walk_ahead(f.W, s_chunk, pos, w, t.pos, f.limit, max_w, fix, fix_cap, b, fits, bad, walks + slot0 + cur, mine);
        if (!have) mine = chain_walk_wait(walks + slot0 + cur);
        const bool open = !(pos_in == mine.s_pos && w_in == mine.s_w) || (mine.flags & 16u) != 0u;
        if (open) {
            x.state = 3u;
            if (can_walk && !bad && (mine.flags & 1u) == 0u) {
                const uint32_t n_ck = (mine.flags & 16u) != 0u ? 0u : (mine.n_ck < kChainCk ? mine.n_ck : kChainCk);   // (a stopped walk: nothing to merge into)
                x.state = chain_rewalk(f.W, s_chunk, pos, w, t.pos, f.limit, max_w, cks + (slot0 + cur) * kChainCk, n_ck, mine.cnt, fix, fix_cap, x, b, fits);
                if (x.state == 1u && (mine.flags & 8u) != 0u) x.n_fix = 0u;  // (the part's own entries did not fit: nothing to splice onto)
#ifdef TRPX_PART_STATS
                if (x.state == 3u && lane == 0)
                    printf("repair failed in its walk: frame %u part %u of %u at %u/%u target %u limit %u, own walk %u/%u -> %u/%u cnt %u n_ck %u flags %u\n",
                           frame, cur, P, pos, w, t.pos, f.limit, mine.s_pos, mine.s_w, mine.o_pos, mine.o_w, mine.cnt, mine.n_ck, mine.flags);
#endif
            }
        } else { pos = pos_in; w = w_in; }                                    // (closed after all: what was counted ahead is dropped)
        if (lane == 0) fixes[slot0 + cur] = x;
#ifdef TRPX_PART_STATS
        {
            const uint64_t t_now = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && t_now - t_start > 30000u)
                printf("slow repair: frame %u part %u (from %u) %u us: state %u cnt %u merge at %u (own block %u), own walk cnt %u n_ck %u flags %u start %u/%u, true start %u\n", frame, cur, p,
                       (uint32_t)(t_now - t_start) / 100u, x.state, x.cnt, x.b_merge, x.ck_cnt, mine.cnt, mine.n_ck, mine.flags, mine.s_pos, mine.s_w, prev.o_pos);
        }
#endif
        // on into the next part?  only behind a stopped part that this wavefront has counted to its end
        if (x.state != 2u || (mine.flags & 16u) == 0u || cur + 1u > P - 2u) break;
        ++cur;                                                                // (pos, w: the true end of the part just counted)
    }
}

// One wavefront per part: its start state first -- a position inside a run of equal widths behind the cut, or a plain guess
// (chain_guess) --, published for the wavefront of the part in front, whose walk ends there; then the walk to the start state of the
// part behind, which that part's wavefront publishes.  (One launch instead of a guessing and a walking one: the waits are for a
// wavefront that needs nothing from anybody -- a neighbour's guess is 15 .. 30 us of work from its dispatch --, bounded, and a
// wavefront that gives up reports a bad walk: the frame takes the other route.)
__global__ __launch_bounds__(kWave) void k_chain_walk(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                      const uint64_t* __restrict__ frame_offsets, uint32_t max_w, uint32_t P,
                                                      uint32_t ent_cap, PartState* states, PartWalk* walks, PartCk* cks,
                                                      uint8_t* __restrict__ ents, ChainFix* __restrict__ fixes, uint8_t* __restrict__ fixents,
                                                      uint64_t* __restrict__ vote_words, [[maybe_unused]] uint32_t* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kPartChunkDw + 4];
    __shared__ uint32_t s_pm[kWave + 2];
    const uint32_t item = blockIdx.x;
    const uint32_t frame = item / P, p = item % P;
    const uint32_t lane = (uint32_t)lane_id();
    // (the stack's vote, see ChainVote: part 0 of up to sixteen frames votes, every part looks at the verdict at its checkpoints)
    const uint32_t n_frames = gridDim.x / P, n_voters = chain_voters(n_frames);
    const uint32_t my_voter = (uint32_t)(((uint64_t)frame * n_voters + n_frames - 1u) / n_frames);   // the voter whose frame this would be
    const bool voter = p == 0u && my_voter < n_voters && (uint32_t)((uint64_t)my_voter * n_frames / n_voters) == frame;
    ChainVote vote{vote_words, false};
    if (chain_verdict(vote_words) == 2u) {                                    // (the test build alldense; a wavefront that starts late)
        if (p + 1u < P) { PartWalk r0{}; r0.flags = 1u; chain_walk_publish(walks + (uint64_t)frame * P + p, r0); }
        if (lane == 0) __hip_atomic_store(reinterpret_cast<uint64_t*>(states) + (uint64_t)frame * P + p, (uint64_t)(p * 128u) | ((uint64_t)(kPartWeak | kPartReady) << 32),
                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (somebody may wait for this part's start state)
        return;
    }
#ifdef TRPX_CHAIN_STAMPS
    const uint64_t st_a = __builtin_amdgcn_s_memrealtime();
    uint64_t st_b = st_a, st_c = st_a;
#endif
    if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
    if (lane < 2u) s_pm[kWave + lane] = 0u;
    const uint64_t slot = (uint64_t)frame * P + p;
    PartWalk r{};
    r.flags = 1u;
    PartFrame f = chain_frame(terse, terse_bytes, frame_offsets, frame, P);
    if (voter && chain_verdict(vote_words) == 0u) {                           // this wavefront's vote first: its share of ~1000 blocks from the frame's true start, 4 Kbit at a time
        const uint32_t want = 1024u / n_voters < 64u ? 64u : 1024u / n_voters;
        uint32_t vp = 0u, vw = 0u, vb = 0u, ve = 0u;
        bool vbad = !f.ok;                                                    // (every voter votes, if with nothing: the verdict waits for the last one)
        __builtin_amdgcn_s_setprio(3);                                        // (everybody else is looking for a start state that a header-dense stack will not need)
        for (uint32_t T = 4096u; !vbad && vb < want && T <= 32768u && T < f.limit / 2u; T += 4096u) {   // (the walker's fast steps need their 64 candidates in front of T)
            uint32_t c1 = 0u, e1 = 0u, n1 = 0u;
            bool dn = false;
            part_walk<false>(f.W, s_chunk, vp, vw, T, f.limit, max_w, c1, vbad, dn, nullptr, 0u, n1, false, kPartCk, false, nullptr, 1u, false, 0u, &e1);
            vb += c1; ve += e1;
        }
        chain_vote_cast(vote_words, n_voters, vbad ? 0u : vb, vbad ? 0u : ve);
        __builtin_amdgcn_s_setprio(0);
    }
    PartState s{0u, 0u};                                                      // a frame starts at bit 0 with width 0 (Terse.hpp:359, :505)
    if (p != 0u) {
        if (!f.ok) s = PartState{0u, kPartWeak};
        else {
            const uint32_t X = p * f.L;
            // the search stays inside the part's first half: a start state lies in front of the next cut
            const uint32_t reach = p + 1u < P ? f.L / 2u : (f.limit - X) / 2u;
            s = chain_guess(f.W, s_chunk, s_pm, X, f.limit, max_w, reach, &vote);
        }
    }
    s.pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.pos);
    s.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.w);
    uint64_t* const sw = reinterpret_cast<uint64_t*>(states);
#ifdef TRPX_CHAIN_FORCE_TIMEOUT
    // test build (make chain_timeout, tests/test_gpu_parity.py): frame 0's part 3 never publishes its start state, frame 1's part 5
    // never its record -- the wavefronts that wait for them give up after kChainWaitTicks, the frames take the other route
    const bool mute_start = frame == 0u && p == 3u, mute_record = frame == 1u && p == 5u;
#else
    constexpr bool mute_start = false, mute_record = false;
#endif
    if (lane == 0 && !mute_start)
        __hip_atomic_store(sw + slot, (uint64_t)s.pos | ((uint64_t)(s.w | kPartReady) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the word is all there is to publish: no fence)
    if (p + 1u == P) return;                                                  // (the frame's last part is walked by count: k_chain_index)
#ifdef TRPX_CHAIN_STAMPS
    st_b = __builtin_amdgcn_s_memrealtime();
#endif
    const PartState t = chain_state_wait(states + slot + 1u);                 // (never came: t.pos = 0 -> a bad walk)
    if (chain_verdict(vote_words) == 2u) {                                    // (the verdict came while this part looked for its start)
        PartWalk r0{}; r0.flags = 1u;
        if (!mute_record) chain_walk_publish(walks + slot, r0);
        return;
    }
    if (f.ok && t.pos > s.pos && t.pos < f.limit) {
        uint32_t pos = s.pos, w = s.w & ~kPartFlags, cnt = 0, n_ck = 0;
        bool bad = false, dense = false, stopped = false;
        uint32_t n_exp = 0;
        const uint32_t X = pos;
        // (how long a walk that reads illegal widths keeps starting again: for ever where the cut found no runs at all -- header-dense
        // data: false chains merge within a few K bits, and a stopped part costs its repair a whole part's walk, the repair launch's
        // critical path: 233 against 15 us for 200 x (1030 x 1065) Poisson(3) counts --, kChainPatience bits where runs lie nearby or
        // the start was a run guess, which is then a wrong one: run-dominated data)
        // (... and even there not for ever: a chain that still reads illegal widths a quarter of the part in is one of the few --
        // two of eight 4096^2 frames' 4352 -- that will not merge in time: 184 against 70 us)
        const uint32_t patience = (s.w & kPartWeak) != 0u && (s.w & kPartRuns) == 0u ? (t.pos - X) / 4u : kChainPatience;
#ifdef TRPX_CHAIN_STAMPS
        st_c = __builtin_amdgcn_s_memrealtime();
#endif
        // A cut without a run to start in (header-dense data; the weakparts test build: every cut) starts ON the cut from a plain
        // guess: the chain is not the frame's until it has merged with it -- at the first explicit header it meets on a block start,
        // a median of 94 blocks in Poisson(3) counts (tools/merge_stats.py) -- and the link in front of the part is open: the
        // repair counts from the true state to the first checkpoint behind the merge, a few hundred blocks, in parallel for all
        // parts.  (A warm-up stretch in FRONT of the cut, so that the chain arrives merged -- round 5's first version, 16 K bits --
        // closed 95 % of such links and cost every wavefront 28 .. 52 us of walking on a false chain, which takes a step every
        // other block: more than the repairs of all links together.)
        // A width the pixel type does not have says the chain is not the frame's yet (2 - 6 % of a false chain's explicit headers
        // read one; a true chain's none).  Early in the part the walk RESTARTS behind it -- count, checkpoints and entries begin
        // again: what they held was not the frame's --; kChainPatience bits into the part it STOPS: in run-dominated data a
        // false chain may not merge for a whole part, at five times a true chain's time per bit, and the launch ends with its
        // slowest wavefront (one such part of 4200: 227 against 50 us); the repair then counts the part from its true start.
        for (;;) {
            pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);         // (wave-uniform by construction: the steps' state lives in scalar registers)
            w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
            r.s_pos = pos; r.s_w = w;
            const uint32_t span = t.pos - pos;
            const uint32_t every = span / (kChainCk - 8u) > 4096u ? span / (kChainCk - 8u) : 4096u;
            dense = false;
            part_walk<true>(f.W, s_chunk, pos, w, t.pos, f.limit, max_w, cnt, bad, dense, cks + slot * kChainCk, every, n_ck, p != 0u,
                            kChainCk, false, ents + slot * ent_cap, ent_cap, p != 0u, t.pos - X, &n_exp, false, &vote);
            if (dense && chain_verdict(vote_words) == 2u) {                   // a header-dense stack: k_seg_wg's (the record says: no walk)
                PartWalk r0{}; r0.flags = 1u;
                if (!mute_record) chain_walk_publish(walks + slot, r0);
                return;
            }
            if (!dense || bad) break;
            if (pos >= t.pos || pos - X >= patience) {
                stopped = true;
#ifdef TRPX_PART_STATS
                if (lane == 0) printf("walk stopped: frame %u part %u guess flags %x start %u at %u (target %u) patience %u cnt %u\n", frame, p, s.w >> 28, X, pos, t.pos, patience, cnt);
#endif
                break;
            }
            w = 0u;                                                           // (pos: behind the illegal header)
        }
        r.o_pos = pos; r.o_w = w; r.cnt = cnt; r.n_ck = n_ck; r.pad = n_exp;
        r.flags = (bad ? 1u : 0u) | (cnt + 1u > ent_cap - 1u ? 8u : 0u) | (stopped ? 16u : 0u);   // 8: more blocks than entries; 16: stopped on a false chain
    }
    if (!mute_record) chain_walk_publish(walks + slot, r);
#ifdef TRPX_CHAIN_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t st_d = __builtin_amdgcn_s_memrealtime();
#endif
    // the link into the part behind (parts 1 .. P - 2; the last part's is checked by k_chain_index, which walks it by count) -- unless
    // this walk stopped on a false chain: the wavefront that counts this part from its true start goes on through that link
    if (p + 2u < P && (r.flags & 16u) == 0u)
        chain_link_into(terse, terse_bytes, frame_offsets, max_w, P, states, walks, cks, fixes, fixents, ent_cap, frame, p + 1u, r, f, s_chunk);
#ifdef TRPX_CHAIN_STAMPS
    if (lane == 0) {       // diagnostic build (tools/chain_stamps.py): a status block of 16 + 8 * parts words
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t st_e = __builtin_amdgcn_s_memrealtime();
        uint32_t* o = stamps + 16 + 8 * slot;
        o[0] = (uint32_t)st_a; o[1] = (uint32_t)(st_b - st_a); o[2] = (uint32_t)(st_c - st_b); o[3] = (uint32_t)(st_d - st_c);
        o[4] = f.W.n_fill; o[5] = (uint32_t)(st_e - st_d); o[6] = r.cnt; o[7] = r.n_ck;
    }
#endif
}

// list[0] = count, list[1 + i] = frame (bit 31 clear: the position-parallel walk may look for runs).
// mode[frame]: how the frame's pixels are extracted -- 1: through the decode index (k_chain_index, then the kernels that take the
// widths as given); 0: part by part by the per-frame decoder's walker + extraction waves (k_decode_parts on this table: a second
// walk, but one that hides under the extraction where explicit headers are rare -- 200 x (1030 x 1065) u16 synth-v1: 106 us
// against k_chain_index 24 + k_unpack_tiles 122; where they are frequent the walker is that kernel's bound: Poisson(3) counts
// 200 against 25 + 131).  `narrow` (8 / 16-bit pixels): by the frame's explicit headers per block; 32-bit pixels: always 1.
// One wavefront per frame.
__device__ __forceinline__ void chain_resolve_frame(const PartWalk* walks, const ChainFix* fixes, const FrameGeom& g, uint32_t frame,
                                                    uint32_t P, uint32_t narrow, PartDesc* __restrict__ parts, uint32_t* __restrict__ mode,
                                                    uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    const uint32_t lane = (uint32_t)lane_id();
    const bool hd = chain_verdict(reinterpret_cast<const uint64_t*>(list) - 4) == 2u;   // (the stack's verdict, ChainVote)
    const PartWalk* wf = walks + (uint64_t)frame * P;
    const ChainFix* xf = fixes + (uint64_t)frame * P;
    PartDesc* __restrict__ pf = parts + (uint64_t)frame * P;
    bool ok = true;
    uint32_t running = 0, n_explicit = 0;
    for (uint32_t base = 0; base < P - 1u; base += kWave) {
        const uint32_t p = base + lane;
        const bool valid = p < P - 1u;
        PartWalk r{};
        ChainFix x{};
        bool good = true, own = true, spliced = false;
        uint32_t cnt = 0;
        PartState st{0u, 0u}, en{0u, 0u};                                     // the part's true start and end states
        if (valid) {
            // (Everything is loaded up front, with clamped indices, and combined without branches: fixes[] of a frame's part 0 is
            // never written and never used -- every use is masked by p.  A first version with the loads inside nested divergent
            // ifs marked EVERY frame for the other route in the optimised build and none with a printf in the loop.)
            const uint32_t p1 = p > 0u ? p - 1u : 0u;
            r = wf[p];
            x = xf[p];
            const PartWalk q = wf[p1];
            const ChainFix xq = xf[p1];
            if (p == 0u) x = ChainFix{};
            // The true end of the part in front: its walk's; of a part whose walk STOPPED on a false chain (flag 16), its repair's
            // (state 2) -- and this part's ChainFix was then left by that same wavefront, from that true end (k_chain_repair).
            // A part in front whose walk ran to the cut on a false chain without reading an illegal width (its repair met nothing:
            // state 2 with another end than the walk's) leaves this part's link judged from a false state: the other route.
            const bool q_stopped = p > 0u && (q.flags & 16u) != 0u;
            const bool q_lost = p > 1u && !q_stopped && xq.state == 2u && !(xq.o_pos == q.o_pos && xq.o_w == q.o_w);
            const PartState te = q_stopped ? PartState{xq.o_pos, xq.o_w} : PartState{q.o_pos, q.o_w};
            const bool closed = p == 0u || (te.pos == r.s_pos && te.w == r.s_w);
            const bool use_own = closed && (r.flags & 16u) == 0u;             // starts in the state it was walked from, on the frame's chain
            const bool use_fix = !use_own && (x.state == 1u || x.state == 2u);
            good = (r.flags & 1u) == 0u && (use_own || use_fix) && !q_lost && !(q_stopped && xq.state != 2u);
            st = use_own ? PartState{r.s_pos, r.s_w} : te;
            cnt = use_own ? r.cnt : x.cnt;
            en = use_own || x.state == 1u ? PartState{r.o_pos, r.o_w} : PartState{x.o_pos, x.o_w};
            own = use_own && (r.flags & 8u) == 0u;
            spliced = use_fix && x.n_fix != 0u;                               // (entries: the repair's up to the merge, the walk's own behind it)
            good = good && cnt >= 1u;
        }
        ok = ok && !__ballot(valid && !good);
#ifdef TRPX_PART_STATS
        {   // diagnostic build: status[4..7] = bad walks, repaired links, failed repairs, parts without entries
            const uint32_t n4 = (uint32_t)__builtin_popcountll(__ballot(valid && (r.flags & 1u) != 0u));
            const uint32_t n5 = (uint32_t)__builtin_popcountll(__ballot(valid && p > 0u && (x.state == 1u || x.state == 2u)));
            const uint32_t n6 = (uint32_t)__builtin_popcountll(__ballot(valid && p > 0u && x.state == 3u));
            const uint32_t n7 = (uint32_t)__builtin_popcountll(__ballot(valid && good && !own && !spliced));
            if (lane == 0) { atomicAdd(status + 4, n4); atomicAdd(status + 5, n5); atomicAdd(status + 6, n6); atomicAdd(status + 7, n7); }
        }
#endif
        n_explicit += (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(valid ? r.pad : 0u), 63);
        const uint32_t c = valid && good ? cnt : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (valid && good) {
            PartDesc d;
            d.frame = frame; d.b0 = running + inc - c; d.b1 = d.b0 + c;
            d.pos0 = st.pos; d.w0 = st.w; d.pos1 = en.pos; d.w1 = en.w; d.pad = own ? 1u : (spliced ? 2u : 0u);
            pf[p] = d;
            if (p == P - 2u) {                                                // the frame's last part starts where this one ends
                PartDesc e;
                e.frame = frame; e.b0 = d.b1; e.b1 = g.n_blocks; e.pos0 = en.pos; e.w0 = en.w; e.pos1 = 0u; e.w1 = 0u; e.pad = 0u;
                pf[P - 1u] = e;
            }
        }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (tot >= g.n_blocks - running) ok = false;                          // (the last part holds at least the frame's last block)
        running += ok ? tot : 0u;
    }
    if (lane == 0) mode[frame] = !narrow || !ok || 12u * n_explicit > running ? 1u : 0u;   // (a listed frame's index comes from the other route)
    if (!ok) {
#ifdef TRPX_PART_STATS
        if (frame < 3u && lane == 0) printf("resolve: frame %u not ok, running %u of %u blocks, P %u\n", frame, running, g.n_blocks, P);
#endif
        __builtin_amdgcn_s_waitcnt(0);
        for (uint32_t p = lane; p < P; p += kWave) {
            PartDesc d{};
            d.frame = frame;
            pf[p] = d;                                                        // b1 <= b0: nothing to do for k_chain_index
        }
        if (lane == 0) {
            // bit 31: k_seg_wg's (their number: the word in front of the barrier counter, codec_common.hpp)
            list[1u + atomicAdd(&list[0], 1u)] = frame | (hd ? 0x80000000u : 0u);
            if (hd) atomicAdd(reinterpret_cast<unsigned long long*>(list) - 2, 1ull);
            atomicAdd(status + 2, 1u);                                        // status[2]: frames the index route handed to the position-parallel walk
        }
    }
}

__global__ __launch_bounds__(kWave) void k_chain_resolve(const PartWalk* __restrict__ walks, const ChainFix* __restrict__ fixes, FrameGeom g,
                                                         uint32_t P, uint32_t narrow, PartDesc* __restrict__ parts, uint32_t* __restrict__ mode,
                                                         uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    chain_resolve_frame(walks, fixes, g, blockIdx.x, P, narrow, parts, mode, list, status);
}

// Walks `nb` blocks from (pos, w_prev) -- general steps, Terse.hpp:360-372 -- and writes their widths to wf[0 .. nb) and the bit
// offset of every block whose frame number b0 + i is a multiple of 256 to tf[(b0 + i) / 256].  at_end: the last of them is the
// frame's last block (nb_last values).  Returns false on a width the pixel type does not have or a chain that leaves the frame.
__device__ __forceinline__ bool chain_walk_abs(PartWin& W, uint32_t* __restrict__ s_chunk, uint32_t& pos, uint32_t& w_prev, uint32_t nb,
                                               bool at_end, uint32_t nb_last, uint32_t limit, uint32_t max_w, uint8_t* __restrict__ wf,
                                               uint64_t* __restrict__ tf, uint32_t b0) {
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t b = 0;
    while (b < nb) {
        const uint32_t stride = 1u + (uint32_t)kBlock * w_prev;
        {
            const uint32_t need_lo = (W.frame_sh + pos) >> 5;
            const uint32_t need_hi = ((W.frame_sh + pos + 63u * stride) >> 5) + 2u;
            if ((int32_t)need_lo < W.c_lo || (int32_t)need_hi > W.c_hi) part_fill(W, s_chunk, pos);
        }
        const uint32_t lpos = pos + lane * stride;
        const uint32_t bits = part_bits(W, s_chunk, lpos);
        const uint32_t left = nb - b;
        const uint64_t valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
        const uint64_t stop = ~(__ballot((bits & 1u) != 0u) & valid);
        const uint32_t first = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
        uint32_t e_w = w_prev, new_pos, new_b;
        if (first < left && first < 64u) {                                    // explicit header at block b + first
            const uint32_t eb = (uint32_t)__builtin_amdgcn_readlane((int)bits, (int)first);
            uint32_t w = (eb >> 1) & 7u, hl = 4;                              // Terse.hpp:362-370
            if (w == 7u) {
                w += (eb >> 4) & 3u; hl = 6;
                if (w == 10u) { w += (eb >> 6) & 63u; hl = 12; }
            }
            if (w > max_w) return false;
            e_w = w;
            const uint32_t nbv = at_end && b + first + 1u == nb ? nb_last : (uint32_t)kBlock;
            new_pos = pos + first * stride + hl + nbv * w;
            new_b = b + first + 1u;
        } else {                                                              // they all repeat the width
            const uint32_t cnt = left < 64u ? left : 64u;
            new_pos = pos + cnt * stride;
            if (at_end && b + cnt == nb) new_pos = pos + (cnt - 1u) * stride + 1u + nb_last * w_prev;
            new_b = b + cnt;
        }
        const uint32_t n_done = new_b - b;
        if (lane < n_done) {
            wf[b + lane] = (uint8_t)(lane < first ? w_prev : e_w);
            if (((b0 + b + lane) & (uint32_t)(kTileBlocks - 1)) == 0u) tf[(b0 + b + lane) / (uint32_t)kTileBlocks] = lpos;
        }
        pos = new_pos; w_prev = e_w; b = new_b;
        if (pos > limit) return false;
    }
    return true;
}

// Entries j .. j + 3 of a part as one dword: its own walk's (E), or -- a repaired part -- the repair's up to the block where the
// chains merged and the walk's own, shifted, behind it.  (Unaligned dword loads; bytes behind the part's last entry are the
// neighbour's or slack and are not used.)
__device__ __forceinline__ uint32_t chain_ent4(const uint8_t* __restrict__ E, const uint8_t* __restrict__ F, uint32_t j, uint32_t b_merge,
                                               int32_t shift) {
    uint32_t q;
    if (!F || j > b_merge) { __builtin_memcpy(&q, E + (int64_t)j + (F ? shift : 0), 4); return q; }
    __builtin_memcpy(&q, F + j, 4);
    if (j + 3u > b_merge) {
        uint32_t e;
        __builtin_memcpy(&e, E + (int64_t)j + shift, 4);
        const uint32_t keep = 0xFFFFFFFFu >> (8u * (3u - (b_merge - j)));   // bytes j .. b_merge from the repair
        q = (q & keep) | (e & ~keep);
    }
    return q;
}

__global__ __launch_bounds__(kWave) void k_chain_index(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w, uint32_t P,
                                                       uint32_t ent_cap, PartDesc* __restrict__ parts, uint8_t* __restrict__ ents,
                                                       const ChainFix* __restrict__ fixes, const uint8_t* __restrict__ fixents,
                                                       uint8_t* __restrict__ widths, uint64_t* __restrict__ tile_off,
                                                       uint32_t* __restrict__ mode, uint32_t* __restrict__ list, uint32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kPartChunkDw + 4];
    const uint32_t lane = (uint32_t)lane_id();
    const PartDesc d = parts[blockIdx.x];
    if (d.b1 <= d.b0 || mode[d.frame] == 0u) return;                          // the frame took another route / is extracted part by part
    const uint32_t frame = d.frame, cnt = d.b1 - d.b0;
    const bool at_end = d.b1 == g.n_blocks;
    const uint32_t nb_last = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock);
    uint8_t* __restrict__ wf = widths + (uint64_t)frame * g.n_blocks + d.b0;
    uint64_t* __restrict__ tf = tile_off + (uint64_t)frame * g.n_tiles;
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    bool done = false, have = (d.pad & 3u) != 0u;
    uint8_t* __restrict__ E = ents + (uint64_t)blockIdx.x * ent_cap;
    if (!have && !at_end && cnt + 2u < ent_cap) {
        // No entries of the true chain (a repair's did not fit, the walk's own overflowed into ...): the part's counting walk once
        // more, fast steps, from its true start to its true end -- into the part's own entry array.
        if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
        PartFrame f = chain_frame(terse, terse_bytes, frame_offsets, frame, P);
        if (f.ok && d.pos0 < d.pos1 && d.pos1 < f.limit) {
            uint32_t pos = d.pos0, w = d.w0, c2 = 0, n0 = 0;
            bool bad = false, dense = false;
            part_walk<true>(f.W, s_chunk, pos, w, d.pos1, f.limit, max_w, c2, bad, dense, nullptr, 0u, n0, false, 0u, false, E, ent_cap);
            __builtin_amdgcn_s_waitcnt(0);                                    // (the entries are read back below)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            have = !bad && c2 == cnt && pos == d.pos1 && w == d.w1;
        }
    }
    if (have && !at_end) {
        // ---- entries -> index: lengths from the widths (Terse.hpp:517-535), positions by a prefix sum; 256 blocks per step, four
        // consecutive ones per lane (one dword of entries in, one dword of widths out), the next step's entries requested first ----
        const uint8_t* __restrict__ F = nullptr;
        uint32_t b_merge = 0;
        int32_t shift = 0;
        if ((d.pad & 2u) != 0u) {
            const ChainFix x = fixes[blockIdx.x];
            F = fixents + (uint64_t)blockIdx.x * ent_cap;
            b_merge = x.b_merge; shift = (int32_t)x.ck_cnt - (int32_t)x.b_merge;
        }
        uint32_t pos = d.pos0;
        bool bad = false;
        uint32_t q = chain_ent4(E, F, 4u * lane, b_merge, shift), w_last = 0;
        const uint32_t w_first = (uint32_t)__builtin_amdgcn_readlane((int)q, 0) & 0xFFu;
        for (uint32_t c = 0; c < cnt; c += 4u * kWave) {
            const uint32_t j0 = c + 4u * lane;
            const uint32_t qn = c + 4u * kWave <= cnt ? chain_ent4(E, F, j0 + 4u * kWave, b_merge, shift) : 0u;   // (entry cnt included)
            uint32_t e4 = (uint32_t)__shfl_down((int)(q & 0xFFu), 1, 64);
            if (lane == 63u) e4 = (uint32_t)__builtin_amdgcn_readlane((int)qn, 0) & 0xFFu;
            const uint32_t e[5] = {q & 0xFFu, (q >> 8) & 0xFFu, (q >> 16) & 0xFFu, q >> 24, e4};
            uint32_t len[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = j0 + (uint32_t)k < cnt;
                len[k] = in ? header_len(e[k + 1], e[k]) + (uint32_t)kBlock * e[k + 1] : 0u;
                bad = bad || (in && e[k + 1] > max_w);
                sum += len[k];
            }
            const uint32_t incl = wave_inclusive_scan(sum);
            uint32_t at = pos + incl - sum;
            const uint32_t packed = e[1] | (e[2] << 8) | (e[3] << 16) | (e[4] << 24);
            if (j0 + 4u <= cnt) __builtin_memcpy(wf + j0, &packed, 4);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (j0 + (uint32_t)k < cnt) wf[j0 + (uint32_t)k] = (uint8_t)e[k + 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (j0 + (uint32_t)k < cnt && ((d.b0 + j0 + (uint32_t)k) & (uint32_t)(kTileBlocks - 1)) == 0u)
                    tf[(d.b0 + j0 + (uint32_t)k) / (uint32_t)kTileBlocks] = at;
                at += len[k];
            }
            pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            // the width of the part's last block (entry cnt): whichever lane holds it in this step
            const uint32_t rel = cnt - c;                                    // entry cnt = e[rel - 4 * lane] of lane (rel - 1) / 4 ... or lane rel / 4's e[0]
            if (rel <= 4u * kWave) {
                const uint32_t ln = (rel - 1u) / 4u, kk = rel - 4u * ln;     // 1 .. 4
                const uint32_t v = kk == 1u ? e[1] : kk == 2u ? e[2] : kk == 3u ? e[3] : e[4];
                w_last = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)ln);
            }
            q = qn;
        }
        done = !__ballot(bad) && pos == d.pos1 && w_last == d.w1 && w_first == d.w0;
        // (not what the walk arrived at: e.g. a stream whose writer spelled out a width that repeats -- the walk below reads
        // the headers themselves)
    }
    if (!done) {
        if (lane < 4u) s_chunk[kPartChunkDw + lane] = 0u;
        PartFrame f = chain_frame(terse, terse_bytes, frame_offsets, frame, P);
        uint32_t pos = d.pos0, w = d.w0;
        bool ok = f.ok && d.pos0 < f.limit && chain_walk_abs(f.W, s_chunk, pos, w, cnt, at_end, nb_last, f.limit, max_w, wf, tf, d.b0);
        if (ok) ok = at_end ? (pos <= f.limit && 1u + (uint64_t)pos / 8u == fe - fo) : (pos == d.pos1 && w == d.w1);   // S_f (Terse.hpp:547) / the next part's start
        // Not the frame's chain after all (a table k_chain_resolve put together from walks that agreed by chance, or a damaged
        // stream): the frame is listed -- once -- for the position-parallel walk, which writes its whole index again and reports
        // what is corrupt.
        if (!ok && lane == 0 && (atomicOr(&parts[(uint64_t)frame * P].pad, 0x80000000u) & 0x80000000u) == 0u) {
            list[1u + atomicAdd(&list[0], 1u)] = frame;
            atomicAdd(status + 2, 1u);
        }
    }
}

// The words the index route needs cleared in front of every call: the status block (if asked), the deferred-frame count and the
// stack statistics in front of it, and the route's start states, walk records and link records (ChainWs: [states, cks) -- the
// link records too: a wavefront that skips a link writes none, and a record left by an earlier call must not be read as this
// call's).  One launch.
__global__ __launch_bounds__(kThreads) void k_chain_zero(uint64_t* __restrict__ p, uint64_t n, uint64_t* __restrict__ q, uint64_t m,
                                                         uint64_t* __restrict__ r, uint64_t k) {
    const uint64_t i0 = (uint64_t)blockIdx.x * kThreads + threadIdx.x, stride = (uint64_t)gridDim.x * kThreads;
    for (uint64_t i = i0; i < k; i += stride) r[i] = 0ull;
    for (uint64_t i = i0; i < n; i += stride) p[i] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x < m) q[threadIdx.x] = 0ull;
}
hipError_t launch_chain_zero(const DecodeArgs& a, uint32_t max_w, bool clear_status, hipStream_t st) {
    (void)max_w;
    const uint32_t P = a.parts_per_frame;
    if (P < 4u || !a.part_ws || !a.defer) return hipErrorInvalidValue;
    const ChainWs l = chain_ws_layout(a.geom, a.n_frames, P);
    const uint64_t k = l.cks / 8;
    hipLaunchKernelGGL(k_chain_zero, dim3((uint32_t)((k + 4 * kThreads - 1) / (4 * kThreads))), dim3(kThreads), 0, st,
                       reinterpret_cast<uint64_t*>(a.defer) - kDeferSlots * kDeferSlotWords, (uint64_t)(kDeferSlots * kDeferSlotWords + 1),
                       reinterpret_cast<uint64_t*>(a.status), (uint64_t)(clear_status ? 4 : 0), reinterpret_cast<uint64_t*>(a.part_ws), k);
    return hipGetLastError();
}

// Fills a.widths / a.tile_off for every frame that works out and lists the others in a.defer.  The caller has cleared the route's
// words (launch_chain_zero) earlier on the stream.
hipError_t launch_build_index_chain(const DecodeArgs& a, uint32_t max_w, bool narrow, const uint32_t** frame_mode, hipStream_t st) {
    const uint32_t P = a.parts_per_frame;
    if (P < 4u || !a.parts || !a.part_ws || !a.defer) return hipErrorInvalidValue;
    const ChainWs l = chain_ws_layout(a.geom, a.n_frames, P);
    char* ws = static_cast<char*>(a.part_ws);
    PartState* states = reinterpret_cast<PartState*>(ws + l.states);
    PartWalk* walks = reinterpret_cast<PartWalk*>(ws + l.walks);
    ChainFix* fixes = reinterpret_cast<ChainFix*>(ws + l.fixes);
    PartCk* cks = reinterpret_cast<PartCk*>(ws + l.cks);
    uint8_t* ents = reinterpret_cast<uint8_t*>(ws + l.ents);
    uint8_t* fixents = reinterpret_cast<uint8_t*>(ws + l.fixents);
    uint32_t* modes = reinterpret_cast<uint32_t*>(ws + l.modes);
    uint64_t* vote_words = reinterpret_cast<uint64_t*>(a.defer) - 4;             // (ChainVote: [0] verdict, [1] votes; cleared by launch_chain_zero)
    const uint32_t cap = chain_ent_cap(a.geom.n_blocks, P);
    hipLaunchKernelGGL(k_chain_walk, dim3(a.n_frames * P), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, max_w, P, cap,
                       states, walks, cks, ents, fixes, fixents, vote_words, a.status);
    hipLaunchKernelGGL(k_chain_resolve, dim3(a.n_frames), dim3(kWave), 0, st, static_cast<const PartWalk*>(walks),
                       static_cast<const ChainFix*>(fixes), a.geom, P, narrow ? 1u : 0u, a.parts, modes, a.defer, a.status);
    hipLaunchKernelGGL(k_chain_index, dim3(a.n_frames * P), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom,
                       max_w, P, cap, a.parts, ents, static_cast<const ChainFix*>(fixes), static_cast<const uint8_t*>(fixents), a.widths,
                       a.tile_off, modes, a.defer, a.status);
    *frame_mode = modes;
    return hipGetLastError();
}

}  // namespace trpx
