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
//   speculation   lane 0 starts from the true state (0, 0) (Terse.hpp:359).  Every other lane first starts from
//                 the guess (X_j, 0).  A wrong chain and the true chain merge as soon as they meet at an explicit
//                 header, or at any block start with equal widths -- which they do within a few hundred blocks on
//                 real data -- so most guessed lanes still END in the true state.
//   fix point     in_j <- out_(j-1), re-walk the lanes whose IN state changed, repeat until nothing changes.
//                 in_0 is true and in_(j+1) = F_j(in_j) for every j, hence by induction every state is the true
//                 one; at most G rounds, typically 2-4.  Inside a wavefront the rounds are a loop (states move one
//                 lane up with a DPP/LDS shuffle); across wavefronts of one frame the last OUT state of wave k-1
//                 is read from memory by wave k on the next launch, and k_seg_resolve re-checks every such link
//                 (and re-runs a wave serially if one is still open) before anything is written.
//   write pass    block counts -> prefix sum -> every lane walks its segment once more from its verified IN
//                 state and stores width[b] (u8, array pre-zeroed: zero widths are not stored) and the bit
//                 offset of every 256-block group: the same decode index k_walk_lds emits, consumed by
//                 k_unpack_tiles.  The last lane runs to n_blocks and checks S_f = 1 + bits/8 (Terse.hpp:547).
//
// Stream access: each lane reads its own segment, so the bytes a wavefront needs at any moment are 64 separate
// 128-byte pieces.  They are fetched cooperatively -- eight lanes load one segment's piece as 8 x 16 bytes, so a
// load instruction covers eight full cache lines -- into a per-lane LDS window (position based: window t of lane j
// holds the bits [X_j + 768 t, X_j + 768 (t+1) + lookahead)), the next window is prefetched into registers while
// the current one is walked.  A run of zero-width blocks (header bits 1, 1 bit per block: empty detector
// regions) is consumed 32 blocks per step.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"

namespace trpx {

constexpr uint32_t kSegAdv = 768;                 // bits a window advances
constexpr uint32_t kSegWinDw = 32;                // dwords loaded per window: 127 (alignment) + 768 + 44 (peek) bits <= 1024
constexpr uint32_t kSegRow = 36;                  // LDS dwords per lane window (16-byte aligned rows)
constexpr uint32_t kSegLiveMargin = 400;          // > longest block (12 + 12 * 32 bits): see seg_last_live()

typedef uint32_t seg_u4 __attribute__((ext_vector_type(4)));

struct SegCtx {
    const uint32_t* s32;     // stream as dwords
    uint64_t n_dw;           // dwords that hold at least one stream byte
    uint64_t fa;             // absolute bit position of the frame's first bit
    uint32_t limit;          // 8 * S_f
    uint32_t L;              // segment length in bits (multiple of 128)
    uint32_t wsh;            // fa & 127: bit offset of a window's first wanted bit inside its 16-byte aligned load
    uint32_t n_blocks, nb_last, max_w;
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
template <bool WRITE>
__device__ __forceinline__ void seg_walk(const SegCtx& c, uint32_t* __restrict__ win, uint32_t seg0, bool part, uint32_t end,
                                         bool by_count, uint32_t& pos, uint32_t& w, uint32_t& n,
                                         uint8_t* __restrict__ wf, uint64_t* __restrict__ tf, bool& bad) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t X = (seg0 + lane) * c.L;
    const uint32_t oct = lane & ~7u, piece = lane & 7u;
    bool done = !part || (WRITE && by_count ? n >= c.n_blocks : pos >= end);
    seg_u4 pre[8];
    // window t of segment s: dwords [d0, d0 + 32) with d0 = ((fa + X_s + 768 t) >> 5) & ~3
    auto fetch = [&](uint32_t t, uint64_t live) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t s = oct + k;
            if ((live >> s) & 1ull) {
                const uint64_t d0 = ((c.fa + (uint64_t)(seg0 + s) * c.L + (uint64_t)t * kSegAdv) >> 5) & ~3ull;
                pre[k] = seg_load16(c, d0 + 4u * piece);
            }
        }
    };
    uint64_t live = __ballot(!done);
    if (live) fetch(0, live);
    for (uint32_t t = 0; live; ++t) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t s = oct + k;
            if ((live >> s) & 1ull) *reinterpret_cast<seg_u4*>(&win[s * kSegRow + 4u * piece]) = pre[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fetch(t + 1, live);                                   // prefetch: consumed at the top of the next iteration
        const uint32_t w0 = X + t * kSegAdv, wend = w0 + kSegAdv;
        bool act = !done && pos < wend;
        while (__ballot(act)) {
            if (act) {
                const uint32_t li = pos - w0 + c.wsh;                                     // bit index inside the lane's window
                const uint32_t* row = win + lane * kSegRow + (li >> 5);
                const uint32_t bits = __builtin_amdgcn_alignbit(row[1], row[0], li);      // 32 stream bits from pos
                const bool same = (bits & 1u) != 0u;                                      // Terse.hpp:361
                const uint32_t w3 = (bits >> 1) & 7u, wa = 7u + ((bits >> 4) & 3u), wb = 10u + ((bits >> 6) & 63u);
                const uint32_t wx = w3 != 7u ? w3 : (wa != 10u ? wa : wb);                // Terse.hpp:362-370
                const uint32_t hx = w3 != 7u ? 4u : (wa != 10u ? 6u : 12u);
                uint32_t wn = same ? w : wx;
                if (wn > c.max_w) { wn = 0u; if (WRITE) bad = true; }
                uint32_t rep = 1u, len;
                if (same && wn == 0u) {                                                   // run of empty blocks: 1 bit each
                    rep = (uint32_t)__builtin_ctz(~bits | 0x80000000u) + ((bits == 0xFFFFFFFFu) ? 1u : 0u);
                    const uint32_t room = WRITE && by_count ? c.n_blocks - n : end - pos;
                    rep = rep < room ? rep : room;
                    len = rep;
                } else {
                    const uint32_t nv = WRITE && n + 1u == c.n_blocks ? c.nb_last : (uint32_t)kBlock;
                    len = (same ? 1u : hx) + nv * wn;
                }
                if (WRITE) {
                    if (n + rep > c.n_blocks) { bad = true; done = true; }
                    else {
                        if (wn) wf[n] = (uint8_t)wn;
                        const uint32_t m = (n + (uint32_t)kTileBlocks - 1u) & ~((uint32_t)kTileBlocks - 1u);
                        if (m < n + rep) tf[m / kTileBlocks] = pos + (m - n);               // rep > 1 only for 1-bit blocks
                    }
                }
                pos += len; n += rep; w = wn;
                if (WRITE && by_count) done = done || n >= c.n_blocks;
                else done = done || pos >= end;
                if (pos > c.limit) { done = true; if (WRITE) bad = bad || n < c.n_blocks || !by_count; }
                act = !done && pos < wend;
            }
        }
        __builtin_amdgcn_wave_barrier();                      // every lane is through with this window before it is overwritten
        live = __ballot(!done);
        if ((uint64_t)t * kSegAdv > (uint64_t)c.limit + 2u * kSegAdv) break;   // (cannot happen: done is set past the limit)
    }
}

// Frame-constant part of the context.
__device__ __forceinline__ bool seg_ctx(SegCtx& c, const uint8_t* terse, uint64_t terse_bytes, const uint64_t* frame_offsets,
                                        uint64_t frame, const FrameGeom& g, uint32_t max_w, uint32_t G) {
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

// Fix-point rounds of wave k of a frame (segments 64 k .. 64 k + 63).  `first`: no earlier launch has left
// states behind (start from the guesses).  Lane 0 of a wave k > 0 takes wave k-1's last OUT state as its IN state
// when `link` is set.  Leaves in / out / cnt of its 64 segments in memory.
__device__ __forceinline__ void seg_fixpoint(const SegCtx& c, uint32_t* __restrict__ win, uint32_t k, uint32_t jl, bool first,
                                             bool link, uint64_t* __restrict__ s_in, uint64_t* __restrict__ s_out,
                                             uint32_t* __restrict__ s_cnt) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t j = 64u * k + lane;
    const bool walks = j < jl;                                 // lane jl and the lanes behind it own no counted blocks
    uint64_t in = first ? seg_pack(j * c.L, 0u) : s_in[j];
    uint64_t out = first ? 0ull : s_out[j];
    uint32_t cnt = first ? 0u : s_cnt[j];
    bool dirty = first && walks;
    if (j == 0u) in = 0ull;                                    // the frame starts with width 0 at bit 0 (Terse.hpp:359, :505)
    if (lane == 0u && k > 0u && link) {
        const uint64_t ni = __hip_atomic_load(&s_out[j - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ni != in) { in = ni; dirty = walks; }
    }
    for (int iter = 0; iter < 66; ++iter) {
        if (!__ballot(dirty)) break;
        uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = 0u;
        bool bad = false;
        seg_walk<false>(c, win, 64u * k, dirty, (j + 1u) * c.L, false, pos, w, n, nullptr, nullptr, bad);
        if (dirty) { out = seg_pack(pos, w); cnt = n; }
        const uint64_t prev = seg_shfl_up1(out);
        bool nd = false;
        if (lane > 0u && j <= jl && prev != in) { in = prev; nd = walks; }
        dirty = nd;
    }
    s_in[j] = in;
    s_cnt[j] = walks ? cnt : 0u;
    __hip_atomic_store(&s_out[j], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Write pass of wave k: `base` = blocks in front of the lane's segment.
__device__ __forceinline__ void seg_write(const SegCtx& c, uint32_t* __restrict__ win, uint32_t k, uint32_t jl, uint64_t in,
                                          uint64_t next_in, uint32_t base, uint8_t* __restrict__ wf, uint64_t* __restrict__ tf,
                                          uint32_t S_bytes, uint32_t* __restrict__ status) {
    const uint32_t lane = (uint32_t)lane_id();
    const uint32_t j = 64u * k + lane;
    const bool part = j <= jl, last = j == jl;
    uint32_t pos = (uint32_t)in, w = (uint32_t)(in >> 32), n = base;
    bool bad = part && base > c.n_blocks;
    seg_walk<true>(c, win, 64u * k, part && !bad, last ? 0xFFFFFFFFu : (j + 1u) * c.L, last, pos, w, n, wf, tf, bad);
    if (part && !last && seg_pack(pos, w) != next_in) bad = true;                       // the chain the counts came from
    if (last && !(n == c.n_blocks && pos <= c.limit && 1u + pos / 8u == S_bytes)) bad = true;   // S_f = 1 + bits/8 (Terse.hpp:547)
    if (__ballot(bad) && lane == 0u) atomicMax(&status[0], 5u);                         // TRPX_ERR_CORRUPT
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

// ---- one wavefront per frame (G = 64): rounds, prefix sum and write pass in one launch ---------------------------------
__global__ __launch_bounds__(kWave) void k_seg_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                      const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                      uint64_t* __restrict__ seg_in, uint64_t* __restrict__ seg_out,
                                                      uint32_t* __restrict__ seg_cnt, uint8_t* __restrict__ widths,
                                                      uint64_t* __restrict__ tile_off, const uint32_t* __restrict__ only,
                                                      uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    const uint64_t frame = blockIdx.x;
    if (only && !only[frame]) return;                          // (frames the per-frame decoder kept for itself)
    const uint32_t lane = (uint32_t)lane_id();
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, kWave)) {
        if (lane == 0) atomicMax(&status[0], 5u);
        return;
    }
    uint8_t* wf = widths + frame * g.n_blocks;
    uint64_t* tf = tile_off + frame * g.n_tiles;
    seg_zero_widths(wf, g.n_blocks, 0u, 1u);
    uint64_t* s_in = seg_in + frame * kWave;
    uint64_t* s_out = seg_out + frame * kWave;
    uint32_t* s_cnt = seg_cnt + frame * kWave;
    const uint32_t jl = seg_last_live(c.limit, c.L, kWave);
    seg_fixpoint(c, win, 0u, jl, true, false, s_in, s_out, s_cnt);
    __builtin_amdgcn_s_waitcnt(0);                             // the zeroes are in L2 before the write pass stores widths
    const uint64_t in = s_in[lane];
    const uint32_t cnt = s_cnt[lane];
    const uint64_t next_in = (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)in, 1, 64) |
                             ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(in >> 32), 1, 64) << 32);
    const uint32_t base = wave_inclusive_scan(cnt) - cnt;
    seg_write(c, win, 0u, jl, in, next_in, base, wf, tf, c.limit / 8u, status);
}

// ---- several wavefronts per frame (large frames): rounds / resolve / write are separate launches -------------------------
__global__ __launch_bounds__(kWave) void k_seg_round(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                     uint32_t K, uint32_t first, uint64_t* __restrict__ seg_in,
                                                     uint64_t* __restrict__ seg_out, uint32_t* __restrict__ seg_cnt,
                                                     uint8_t* __restrict__ widths, uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    const uint64_t frame = blockIdx.x / K;
    const uint32_t k = blockIdx.x % K;
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave)) {
        if (threadIdx.x == 0 && k == 0) atomicMax(&status[0], 5u);
        return;
    }
    if (first) seg_zero_widths(widths + frame * g.n_blocks, g.n_blocks, k, K);
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    const uint64_t so = frame * K * kWave;
    seg_fixpoint(c, win, k, jl, first != 0u, first == 0u, seg_in + so, seg_out + so, seg_cnt + so);
}

// One wavefront per frame: closes the links between the frame's waves that the rounds left open (serially, wave by
// wave: each re-run starts from a verified state) and turns the block counts into block bases.
__global__ __launch_bounds__(kWave) void k_seg_resolve(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                       const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                       uint32_t K, uint64_t* __restrict__ seg_in, uint64_t* __restrict__ seg_out,
                                                       uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ seg_base,
                                                       uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    const uint64_t frame = blockIdx.x;
    const uint32_t lane = (uint32_t)lane_id();
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave)) return;    // (reported by k_seg_round)
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    const uint64_t so = frame * K * kWave;
    uint32_t running = 0u;
    for (uint32_t k = 0; k < K; ++k) {
        if (k > 0u && 64u * k <= jl) {
            const uint64_t a = __hip_atomic_load(&seg_in[so + 64u * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t b = __hip_atomic_load(&seg_out[so + 64u * k - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a != b) {
                seg_fixpoint(c, win, k, jl, false, true, seg_in + so, seg_out + so, seg_cnt + so);
                __builtin_amdgcn_s_waitcnt(0);
                __threadfence();
            }
        }
        const uint32_t cnt = __hip_atomic_load(&seg_cnt[so + 64u * k + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t inc = wave_inclusive_scan(cnt);
        seg_base[so + 64u * k + lane] = running + inc - cnt;
        running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
}

__global__ __launch_bounds__(kWave) void k_seg_write(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                     const uint64_t* __restrict__ frame_offsets, FrameGeom g, uint32_t max_w,
                                                     uint32_t K, const uint64_t* __restrict__ seg_in,
                                                     const uint32_t* __restrict__ seg_base, uint8_t* __restrict__ widths,
                                                     uint64_t* __restrict__ tile_off, uint32_t* __restrict__ status) {
    __shared__ uint32_t win[kWave * kSegRow];
    const uint64_t frame = blockIdx.x / K;
    const uint32_t k = blockIdx.x % K;
    const uint32_t lane = (uint32_t)lane_id();
    SegCtx c;
    if (!seg_ctx(c, terse, terse_bytes, frame_offsets, frame, g, max_w, K * kWave)) return;
    const uint32_t jl = seg_last_live(c.limit, c.L, K * kWave);
    if (64u * k > jl) return;
    const uint64_t so = frame * K * kWave;
    const uint32_t j = 64u * k + lane;
    const uint64_t in = seg_in[so + j];
    const uint64_t next_in = j + 1u < K * kWave ? seg_in[so + j + 1u] : 0ull;
    seg_write(c, win, k, jl, in, next_in, seg_base[so + j], widths + frame * g.n_blocks, tile_off + frame * g.n_tiles,
              c.limit / 8u, status);
}

// Segments per frame: a multiple of 64, about kSegTargetBlocks blocks each.
uint32_t seg_waves_per_frame(const FrameGeom& g) {
    constexpr uint64_t kSegTargetBlocks = 320;
    const uint64_t k = ((uint64_t)g.n_blocks + 32 * kSegTargetBlocks) / (64 * kSegTargetBlocks);
    return (uint32_t)(k ? k : 1);
}
size_t seg_workspace_bytes(const FrameGeom& g, size_t n_frames) {
    const size_t segs = n_frames * (size_t)seg_waves_per_frame(g) * kWave;
    return align_up(segs * (8 + 8 + 4 + 4), 256);
}

// Fills a.widths / a.tile_off (the decode index) from the stream; `only` (device, u32 per frame, may be null) limits
// the single-wave variant to the flagged frames.
hipError_t launch_seg_walk(const DecodeArgs& a, uint32_t max_w, const uint32_t* only, hipStream_t st) {
    const uint32_t K = seg_waves_per_frame(a.geom);
    const size_t segs = (size_t)a.n_frames * K * kWave;
    uint64_t* s_in = reinterpret_cast<uint64_t*>(a.seg_ws);
    uint64_t* s_out = s_in + segs;
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_out + segs);
    uint32_t* s_base = s_cnt + segs;
    if (K == 1) {
        hipLaunchKernelGGL(k_seg_frames, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets,
                           a.geom, max_w, s_in, s_out, s_cnt, a.widths, a.tile_off, only, a.status);
        return hipGetLastError();
    }
    const dim3 grid((uint32_t)((size_t)a.n_frames * K));
    hipLaunchKernelGGL(k_seg_round, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, 1u, s_in, s_out, s_cnt, a.widths, a.status);
    hipLaunchKernelGGL(k_seg_round, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, 0u, s_in, s_out, s_cnt, a.widths, a.status);
    hipLaunchKernelGGL(k_seg_resolve, dim3(a.n_frames), dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets,
                       a.geom, max_w, K, s_in, s_out, s_cnt, s_base, a.status);
    hipLaunchKernelGGL(k_seg_write, grid, dim3(kWave), 0, st, a.terse, (uint64_t)a.terse_bytes, a.frame_offsets, a.geom, max_w,
                       K, s_in, s_base, a.widths, a.tile_off, a.status);
    return hipGetLastError();
}

}  // namespace trpx
