// PROLIX decode, one workgroup per frame (gfx950 / CDNA4): the serial header walk and the parallel
// field extraction run side by side inside the workgroup, coupled through LDS only.
// Replaces jpa::Terse::prolix(Iterator, frame) (reference include/Terse.hpp:352-389) and
// Bit_range::get_range / operator T() (Bit_pointer.hpp:742-792, :597-617).
//
//   one wave ("walker")  walks the frame's header chain (Terse.hpp:360-372): 64 candidate blocks per step, one step per
//                        run of equal widths, the stream staged through a private 8 KB LDS window (refilled by LDS-DMA,
//                        no staging registers).  For every block it leaves "bit position of the first payload bit |
//                        width << 26" in LDS.
//   three waves          one super-step (768 blocks = 12 groups of 64) behind the walker: each wave takes four groups;
//                        every lane loads the stream dwords of its own block straight from L2 (the walker has just
//                        pulled them through) at the position the walker left, extracts the 12 fields with
//                        width-specialised code on registers and stores the pixels (12 / 24 bytes per lane,
//                        non-temporal; 32-bit pixels go through a per-wave LDS row and leave as whole lines).
//   one barrier per super-step; the position buffers are double buffered.
//
// The walk is the critical path (serial by construction of the format); the extraction hides under it as far as the
// CU's shared issue resources allow.
// HBM traffic per frame: S read (once from HBM, once more from L2) + N*sizeof(T) written = algorithmic.
// Best for many small frames (one workgroup each); few huge frames are better served by the tiled
// kernels of decode_fast.hip.
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include "unpack_common.hpp"

namespace trpx {

#ifndef TRPX_FRAME_WAVES
#define TRPX_FRAME_WAVES 4
#endif
constexpr int kFrameWaves = TRPX_FRAME_WAVES;           // waves per workgroup: 1 walker + the extraction waves
constexpr int kFrameThreads = kFrameWaves * kWave;

#ifndef TRPX_FRAME_CHUNK_DW
#define TRPX_FRAME_CHUNK_DW 2048
#endif
#ifndef TRPX_IDX_TOUCH
#define TRPX_IDX_TOUCH 1        // MODE 1: the filler touches the stream lines of the super-step it has just laid out (A/B: make idxtouch)
#endif
template <typename T>
struct FrameCfg {
#ifdef TRPX_FRAME_GPW
    static constexpr int kGpw = TRPX_FRAME_GPW;
#else
    static constexpr int kGpw = 4;                                       // 64-block groups per extraction wave and super-step (LDS: 20 KB per workgroup, 24.6 KB for 32-bit pixels)
#endif
    static constexpr int kStepGroups = (kFrameWaves - 1) * kGpw;
    static constexpr int kStepBlocks = kStepGroups * kWave;              // 768 / 1152
    // walker's stream window: 8 KB.  Stream cache-resident (decode after decode): 2 / 4 / 8 / 16 KB -> 0.31 / 0.32 / 0.30 /
    // 0.30 ms; cold (decode after an encode, the bench's round trip): 0.40 / 0.37 / 0.35 / 0.40 ms -- the extraction waves
    // re-read the window's lines from L2, and 250 workgroups per XCD x 16 KB is all of its 4 MB
    static constexpr int kChunkDw = TRPX_FRAME_CHUNK_DW;
    // Pixels leave through the extraction wave's LDS row, 16 bytes per lane and store: every store instruction writes whole
    // 128-byte lines.  (Lane-owned 24-byte runs as 16 + 8 byte stores leave every line to be merged from two instructions in
    // L2; with 2048 frames in flight lines were written back half merged -- WRITE_SIZE 1.17 x the pixels -- and the kernel fell
    // from 0.26 to 0.38 ms.  TRPX_DEC_DIRECT_STORES: the round-2 stores for 8/16-bit pixels, A/B only.)
#ifdef TRPX_DEC_DIRECT_STORES
    static constexpr bool kOutStaged = sizeof(T) == 4;
#else
    static constexpr bool kOutStaged = true;
#endif
    static constexpr int kRawDw = 4 * RawQuads<T>::n;                    // stream dwords a lane loads for its block
    static constexpr int kOutDw = kWave * kBlock * (int)sizeof(T) / 4;   // an extraction wave's output row: 64 blocks of pixels
    static_assert(kChunkDw % (kWave * 4) == 0, "window = whole 1 KB pieces");
};

// One LDS-DMA piece: lane l's 16 bytes at `src` land at LDS byte address lds_base + 16 * l; no staging registers.  The
// caller waits with s_waitcnt vmcnt(0) before it reads the bytes.  (An asm statement: with the builtin in the kernel's body
// the host pass of hipcc 7.2 silently dropped the kernel's launch stubs.)
__device__ __forceinline__ void lds_dma16(const uint32_t* src_uniform, uint32_t lane_byte_offset, uint32_t lds_base) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_offset), "s"(src_uniform), "s"(lds_base) : "memory");
}

// MODE 0: decode.  MODE 1: the block widths are known (a decode index: widths[] and the bit offset of every 256th block,
// encode_kernels.hpp) and wave 0 turns them into the position entries with a prefix sum instead of walking the header chain --
// everything else is the same kernel; used for trpx_decode_indexed on stacks of small frames and for the frames k_decode_frames
// hands to the position-parallel walk, once that has written their index.  MODE 2: the walker walks, and the other waves write
// the index (widths, group offsets) instead of pixels: trpx_build_index on stacks of small frames (header-dense frames are
// handed to the position-parallel walk like in MODE 0).
// LINES: the frames / parts may start anywhere inside a 128-byte line of the output and the extraction waves write line images
// (store_group_lines); without it the groups are stored from their first pixel (line-aligned frames -- the host checks --, and
// the kernels without a walker, which are faster that way even when the frames are not aligned).
template <typename T, int MODE, bool PARTS = false, bool LINES = false>
__device__ __forceinline__ void decode_frame_body(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                  const uint64_t* __restrict__ frame_offsets, const FrameGeom& g,
                                                  T* __restrict__ pixels_out, uint32_t* __restrict__ defer,
                                                  uint32_t* __restrict__ status, const uint64_t item,
                                                  const uint8_t* __restrict__ idx_widths, const uint64_t* __restrict__ idx_group_off,
                                                  const PartDesc* __restrict__ parts = nullptr) {
    using Cfg = FrameCfg<T>;
    constexpr bool IDX = MODE == 1;
    constexpr int kStepBlocks = Cfg::kStepBlocks, kChunkDw = Cfg::kChunkDw;
    constexpr uint32_t kMaxW = PixelTraits<T>::bits;
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[kChunkDw + 4];  // walker's window of the stream
    // Per block b of a super-step, entry [b - first block of the super-step] = H(b): bit position of the block's HEADER
    // (relative to dword frame_dw, 26 bits) | width of the block BEFORE it << 26; one entry more for the block behind the
    // super-step.  H(b + 1) is all the extraction needs of block b: the block's width and the end of its payload.
    // (Double buffered.  The walker's last step of a super-step may run up to 63 blocks over its end, and a fast step writes
    // 64 entries: the entries behind the super-step's end are copied to the front of the other buffer when the next one starts.)
    constexpr int kPosEntries = kStepBlocks + 64 + 68;
    __shared__ uint32_t s_pos[2][kPosEntries];
    uint32_t* const s_posx = &s_pos[0][0];
    constexpr uint32_t kPosBits = 26, kPosMask = (1u << kPosBits) - 1u;
    __shared__ __attribute__((aligned(16))) uint32_t s_out[kFrameWaves - 1][Cfg::kOutStaged ? Cfg::kOutDw + kStageCarryDw : 4];  // a group's pixels as an image of its cache lines (store_group_lines), per extraction wave
    __shared__ uint32_t s_err;
#ifdef TRPX_DEC_LDS_PAD
    __shared__ uint32_t s_pad[TRPX_DEC_LDS_PAD / 4];   // diagnostic build: fewer workgroups per CU
    if (terse_bytes == 1) s_pad[threadIdx.x] = 0;
#endif
    __shared__ uint32_t s_role[kFrameWaves];
    // What only the epilogue needs (status / list pointers, the frame number) waits in LDS: as live scalar registers across the
    // super-step loop they were spilled to a VGPR's lanes (the kernel sits at the 80-SGPR limit of eight workgroups per CU).
    __shared__ uint64_t s_keep[3];

    const uint32_t lane = (uint32_t)lane_id();
    const int hw_wave = wave_id();
    // The unit of work: a whole frame (item = frame) or, with a part table, the blocks [b0, b1) of a frame with the chain state
    // in front of them (encode_kernels.hpp: PartDesc).  Below, everything is relative to the unit: block numbers, bit positions
    // (`limit` = the frame's bits behind the unit's first), the output pointer.
    uint64_t frame = item;
    uint32_t pb0 = 0, pb1 = g.n_blocks, ppos0 = 0, pw0 = 0, ppos1 = 0, pw1 = 0;
    if constexpr (PARTS && MODE == 1) {
        // UNITS of a frame whose decode index is known: blocks [u * unit_blocks, (u + 1) * unit_blocks) -- a multiple of 256, so the
        // index holds the chain state in front of every unit (the group's bit offset, the width of the block before it)
        const uint32_t unit_blocks = (uint32_t)(uintptr_t)parts;          // (the part-table argument carries the unit size here)
        const uint32_t upf = (g.n_blocks + unit_blocks - 1u) / unit_blocks;
        frame = item / upf;
        pb0 = (uint32_t)(item % upf) * unit_blocks;
        pb1 = pb0 + unit_blocks < g.n_blocks ? pb0 + unit_blocks : g.n_blocks;
        ppos0 = (uint32_t)idx_group_off[frame * g.n_tiles + pb0 / (uint32_t)kTileBlocks];
        pw0 = pb0 ? (uint32_t)idx_widths[frame * g.n_blocks + pb0 - 1u] : 0u;
        if (pb1 < g.n_blocks) {
            ppos1 = (uint32_t)idx_group_off[frame * g.n_tiles + pb1 / (uint32_t)kTileBlocks];
            pw1 = (uint32_t)idx_widths[frame * g.n_blocks + pb1 - 1u];
        }
    } else if constexpr (PARTS) {                                         // (without a part table all of this folds to constants)
        const PartDesc d = parts[item];                                   // (uniform address: scalar loads)
        frame = d.frame; pb0 = d.b0; pb1 = d.b1; ppos0 = d.pos0; pw0 = d.w0; ppos1 = d.pos1; pw1 = d.w1;
        if (pb1 <= pb0) return;                                           // the frame took another route
    }
    const uint64_t fo = frame_offsets[frame], fe = frame_offsets[frame + 1];
    if (threadIdx.x == 0) {
        s_err = (fe > fo && fe <= terse_bytes && (uint64_t)ppos0 < 8 * (fe - fo) && pb1 <= g.n_blocks) ? 0u : 1u;
        s_keep[0] = (uint64_t)(uintptr_t)status; s_keep[1] = (uint64_t)(uintptr_t)defer; s_keep[2] = item;
    }
    // Which wave walks.  A workgroup's four waves land on the CU's four SIMDs, the first one on a SIMD that rotates from
    // workgroup to workgroup, and the k-th workgroup to arrive on a CU gets wave slot k on every SIMD (measured,
    // tools/hwid.hip).  "The wave whose SIMD number equals its slot number mod 4 walks" puts exactly two of a CU's eight
    // walkers on every SIMD; with "wave 0 walks" they are spread at random, up to five on one SIMD (3 % slower).  (HW_ID: slot =
    // bits 3:0, SIMD = bits 5:4.)  If no wave or more than one matches -- another placement -- the lowest match, else wave 0.
    uint32_t hw_slot = 0;
    {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (lane == 0) s_role[hw_wave] = ((hwid >> 4) & 3u) == (hwid & 3u) ? 1u : 0u;
        hw_slot = hwid & 7u;
    }
    __syncthreads();
    int walker_wave = 0;
#pragma unroll
    for (int i = kFrameWaves - 1; i >= 0; --i) walker_wave = s_role[i] ? i : walker_wave;
    walker_wave = __builtin_amdgcn_readfirstlane(walker_wave);
    const int wave = (hw_wave - walker_wave + kFrameWaves) % kFrameWaves;     // role: 0 walks, 1.. extract (wave-uniform scalar)
    if (s_err) {
        if (threadIdx.x == 0) atomicMax(&status[0], 5u);
        return;
    }
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(terse);
    const uint64_t n_dw = (terse_bytes + 3) / 4;
    const bool base16 = ((uintptr_t)terse & 15) == 0;
    const uint64_t frame_abit = 8 * fo + ppos0;
    const uint32_t limit = (uint32_t)(8 * (fe - fo) - ppos0);
    const uint64_t frame_dw = frame_abit >> 5;
    const uint32_t frame_sh = (uint32_t)(frame_abit & 31);
    const uint32_t n_blocks = pb1 - pb0;
    const bool at_end = pb1 == g.n_blocks;                                // the unit ends with the frame's last block
    const uint32_t nb_last = at_end ? (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock) : (uint32_t)kBlock;
    // Super-steps: step 0 is short when frames can be handed over (one group per extraction wave: the decision "this
    // frame's headers are too dense for the serial walk" falls after 192 blocks instead of 768 -- a header-dense stack spends
    // 0.04 instead of 0.11 ms here before its frames go to the position-parallel walk); every later step is kStepBlocks.
    // The boundaries are multiples of 64 blocks, so an extraction group's pixels are whole 128-byte lines written by one
    // wave (groups that cut lines -- super-steps ending where the walker's last step happened to end -- cost the stores
    // 0.25 ms per stack: partial-line writes).
    const uint32_t sb0 = defer ? (uint32_t)((kFrameWaves - 1) * kWave) : (uint32_t)kStepBlocks;   // (a multiple of 64 either way)
    const uint32_t n_steps = n_blocks <= sb0 ? 1u : 1u + (n_blocks - sb0 + kStepBlocks - 1) / kStepBlocks;
    auto step_begin = [&](uint32_t t) -> uint32_t { return t == 0u ? 0u : sb0 + (t - 1u) * kStepBlocks; };
    T* __restrict__ fout = pixels_out + frame * g.n_values + (uint64_t)pb0 * kBlock;
    // offset of the frame's first pixel inside its 128-byte line = that of every 64-block group (768 pixels: whole lines)
    const uint32_t out_c = LINES ? (uint32_t)((uintptr_t)fout & 127u) : 0u;

#ifdef TRPX_DEC_NO_STORE
    uint32_t diag_acc = 0;
#endif
    // walker state (wave-uniform)
    int32_t c_lo = 0, c_hi = 0;                        // the window holds dwords [c_lo, c_hi) of the frame
    uint32_t b = 0, w_prev = pw0, pos = 0;             // next block, width of the block before it, bit position of its header (all relative to the unit)
#ifndef TRPX_DEC_WALK_PRIO
#define TRPX_DEC_WALK_PRIO 3
#endif
#ifndef TRPX_DEFER_NUM                               // hand-over line: more than NUM width changes in DEN blocks
#define TRPX_DEFER_NUM 2
#define TRPX_DEFER_DEN 9
#endif
    if (wave == 0) __builtin_amdgcn_s_setprio(TRPX_DEC_WALK_PRIO);      // the walk is the critical path
    // MODE 1: the widths of the super-step after the one being filled wait in LDS (s_wnext[chunk * 64 + lane]); the filler
    // requests them before it computes and parks them behind -- in registers across the loop they would be live in the
    // extraction waves' code too (spills at 64 VGPRs)
    // (requested four widths per lane and load -- one unaligned dword: three registers in flight per 768 blocks, not twelve)
    constexpr int kIdxChunks = IDX ? kStepBlocks / kWave : 1;
    constexpr int kIdxQuads = IDX ? kStepBlocks / (4 * kWave) : 1;
    static_assert(!IDX || kStepBlocks % (4 * kWave) == 0, "the width prefetch covers a super-step in whole quads of 64 lanes (TRPX_FRAME_GPW: a multiple of 4 per three extraction waves)");
    __shared__ __attribute__((aligned(4))) uint8_t s_wnext[IDX ? kStepBlocks : 4];
    [[maybe_unused]] auto idx_load = [&](uint32_t ss, uint32_t (&dst)[kIdxQuads]) {
        const uint8_t* __restrict__ wfl = idx_widths + frame * g.n_blocks + pb0;
        const uint32_t bb = step_begin(ss), be = step_begin(ss + 1u) < n_blocks ? step_begin(ss + 1u) : n_blocks;
#pragma unroll
        for (int j = 0; j < kIdxQuads; ++j) {                                         // (all loads in flight before the first is used)
            const uint32_t bi = bb + 4u * ((uint32_t)j * kWave + lane);
            uint32_t v = 0u;
            if (bi + 4u <= be) __builtin_memcpy(&v, wfl + bi, 4);
            else {
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) v |= bi + k < be ? (uint32_t)wfl[bi + k] << (8u * k) : 0u;
            }
            dst[j] = v;
        }
    };
    [[maybe_unused]] auto idx_park = [&](const uint32_t (&src)[kIdxQuads]) {
#pragma unroll
        for (int j = 0; j < kIdxQuads; ++j) reinterpret_cast<uint32_t*>(s_wnext)[j * kWave + lane] = src[j];
    };
    if constexpr (IDX) {
        if (wave == 0) {
            uint32_t w0[kIdxQuads];
            idx_load(0u, w0);
            idx_park(w0);
        }
    }

#ifdef TRPX_DEC_STAMPS
    uint64_t st_work = 0, st_wait = 0, st_t0 = __builtin_readcyclecounter(), st_start = st_t0;
#endif
    [[maybe_unused]] uint32_t idx_touch = 0;
    for (uint32_t s = 0; s <= n_steps; ++s) {
        if (wave == 0) {
            if (IDX && s < n_steps) {
#if TRPX_IDX_TOUCH
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(idx_touch) : : "memory");     // (the touch loads of the super-step before: long back)
#endif
                // ---- positions from the index: H(b) for the super-step's blocks by a prefix sum over header + payload lengths ----
                const uint32_t beg_b = step_begin(s);
                const uint32_t end_nom = step_begin(s + 1u) < n_blocks ? step_begin(s + 1u) : n_blocks;
                uint32_t* const ent = s_posx + (s & 1u) * kPosEntries;
                // this super-step's widths were requested a super-step ago (idx_next: the filler's loop would otherwise wait for a
                // global load's round trip per super-step -- 29 per 512 x 512 frame, where the walker refills its window 13 times)
                uint32_t wq[kIdxChunks], wn[kIdxQuads];
#pragma unroll
                for (int i = 0; i < kIdxChunks; ++i) wq[i] = s_wnext[i * kWave + lane];
                if (s + 1u < n_steps) idx_load(s + 1u, wn);
                bool bad = false;
                [[maybe_unused]] const uint32_t pos_begin = pos;
#pragma unroll
                for (int i = 0; i < kIdxChunks; ++i) {
                    const uint32_t b0 = beg_b + (uint32_t)i * kWave;
                    if (b0 >= end_nom) continue;                                      // (wave-uniform; no break: the loop has to unroll, wq[] lives in registers)
                    const uint32_t bi = b0 + lane, wi = wq[i];
                    uint32_t wp = (uint32_t)__shfl_up((int)wi, 1, 64);
                    if (lane == 0) wp = w_prev;
                    const uint32_t hl = wi == wp ? 1u : (wi < 7u ? 4u : (wi < 10u ? 6u : 12u));   // Terse.hpp:517-535
                    const uint32_t nv = bi + 1u == n_blocks ? nb_last : (uint32_t)kBlock;
                    const uint32_t len = bi < end_nom ? hl + nv * wi : 0u;
                    const uint32_t incl = wave_inclusive_scan(len);
                    const uint32_t p = pos + incl - len;
                    bad = bad || wi > kMaxW;
                    if (bi < end_nom) {
                        ent[bi - beg_b] = (frame_sh + p) | (wp << kPosBits);
                        if (((pb0 + bi) & (uint32_t)(kTileBlocks - 1)) == 0u && idx_group_off &&
                            idx_group_off[frame * g.n_tiles + (pb0 + bi) / (uint32_t)kTileBlocks] != (uint64_t)ppos0 + p) bad = true;
                    }
                    pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    const uint32_t lastl = end_nom - b0 < (uint32_t)kWave ? end_nom - b0 - 1u : (uint32_t)kWave - 1u;
                    w_prev = (uint32_t)__builtin_amdgcn_readlane((int)wi, (int)lastl);
                    if (pos > limit) bad = true;                                      // (positions stay below 2^26: the index is not trusted)
                }
                b = end_nom;
                if (!__ballot(bad) && b == n_blocks)                                  // S_f = 1 + bits/8 (Terse.hpp:547) / the next part's state
                    bad = at_end ? !(pos <= limit && 1 + ((uint64_t)ppos0 + pos) / 8 == fe - fo) : !(ppos0 + pos == ppos1 && w_prev == pw1);
                if (__ballot(bad) && lane == 0) s_err = 1u;
                if (lane == 0) ent[b - beg_b] = (frame_sh + (pos <= limit ? pos : limit)) | (w_prev << kPosBits);
#if TRPX_IDX_TOUCH
                {
                    // The stream lines of this super-step's blocks, one dword per 128-byte line and lane (64 lines = 8 KB per
                    // instruction): the extraction waves load per lane and block, two overlapping 16-byte loads each, an iteration
                    // from now -- with the lines in L2 by then they miss the vector cache only.  (The walker of MODE 0 pulls the
                    // stream through its LDS window for the same effect; without it the extraction's requests queue up between
                    // the texture-address unit and the vector cache whenever HBM answers late: the kernel's slow state,
                    // profiles/r04_idx_gap.txt.)  The value is never used; the register is given up at the next super-step.
                    // (Touching the next super-step's expected span as well, a super-step further ahead: no different.  Line images
                    // for frames that start inside a cache line, as MODE 0 writes them: 0.351 against 0.33 ms, again.)
                    const uint64_t a0 = (4u * frame_dw + ((frame_sh + pos_begin) >> 3)) & ~127ull;
                    const uint64_t a1 = 4u * frame_dw + ((frame_sh + (pos <= limit ? pos : limit)) >> 3);
                    for (uint64_t my = a0 + 128u * lane; my <= a1 && my + 4u <= terse_bytes; my += 128u * kWave)
                        asm volatile("global_load_dword %0, %1, off" : "=v"(idx_touch) : "v"(terse + my) : "memory");
                }
#endif
                if (s + 1u < n_steps) idx_park(wn);
            } else if (s < n_steps) {
                const uint32_t buf = s & 1u;
                const uint32_t beg_b = step_begin(s);
                const uint32_t end_nom = step_begin(s + 1u) < n_blocks ? step_begin(s + 1u) : n_blocks;
                // fast steps need 64 candidate blocks none of which is the frame's last (its value count differs)
                const int32_t fast_lim = (int32_t)end_nom < (int32_t)n_blocks - 64 ? (int32_t)end_nom : (int32_t)n_blocks - 64;
                uint32_t* const ent = s_posx + buf * kPosEntries;                  // this super-step's entries
                bool bad = false;
                if (s > 0u && lane <= b - beg_b)                                   // what the step before walked behind its end, H(b) included
                    ent[lane] = s_posx[(buf ^ 1u) * kPosEntries + (beg_b - step_begin(s - 1u)) + lane];
                while (true) {
                    // ---- fast steps: 64 candidate headers inside the LDS window ------------------------------------------------
                    // One step = one run of blocks that repeat the width before them (header bit 1, Terse.hpp:361) + the
                    // explicit header that ends it.  A single wave issues an instruction every ~4-5 cycles whatever the
                    // dependencies, and eight walkers share a CU's scalar unit, so the step is built for the smallest
                    // instruction COUNT: every lane tests one candidate at stride 1 + 12 w and stores H (header position |
                    // previous width) for it unconditionally -- entries behind the step's last block are overwritten by the
                    // next step --; the header that ends the run is parsed on the scalar unit after the pick, widths < 7 (one
                    // 3-bit field) on the straight path (Terse.hpp:362-370); runs of 64 take a second, shorter path.  The
                    // state lives in SGPRs.  (Round 2's step parsed every lane's header before the ballot and wrote payload
                    // positions: 27 vector + 43 scalar instructions against 12 + 21 here.)
                    {
                        // (hand-scheduled: hipcc's structurizer spends ~25 copies and flag materialisations per step on this loop)
                        const uint32_t base8 = 8u * (uint32_t)(uintptr_t)&s_chunk[0];         // the window's LDS address, in bits (a multiple of 128)
                        uint32_t pw = frame_sh + pos - 32u * (uint32_t)c_lo + base8;          // the header's bit address in LDS
                        uint32_t stride = 1u + kBlock * w_prev;
                        const uint32_t pw_end = 32u * (uint32_t)(c_hi - c_lo - 1) + base8;    // (signed compares below: c_hi == c_lo before the first refill)
                        uint32_t pw_max = pw_end - 63u * stride;                              // 64 candidates + one dword inside the window
                        uint32_t v_ls = __umul24(lane, stride);
                        const uint32_t e_base = 32u * (uint32_t)c_lo - base8;                 // LDS bit address -> position relative to dword frame_dw
                        uint32_t s_e = e_base + (w_prev << kPosBits);
                        const uint32_t v_entl = (uint32_t)(uintptr_t)ent - 4u * beg_b + 4u * lane;   // LDS address of this lane's entry for b = 0
                        uint32_t s_bad = 0, t_first, t_h, t_t, t_p, t_a, t_bits;
                        asm volatile(
                            "s_cmp_lt_i32 %[b], %[lim]\n\t"
                            "s_cbranch_scc0 9f\n\t"
                            "s_cmp_lt_i32 %[pw], %[pwmax]\n\t"
                            "s_cbranch_scc0 9f\n"
                            "1:\n\t"
                            "v_add_u32 %[p], %[pw], %[ls]\n\t"                 // this lane's candidate header
                            "v_lshrrev_b32 %[a], 3, %[p]\n\t"
                            "v_and_b32 %[a], 0x1ffffffc, %[a]\n\t"
                            "ds_read2_b32 v[62:63], %[a] offset1:1\n\t"
                            "v_add_u32 %[bits], %[se], %[p]\n\t"              // H = header position | previous width
                            "v_lshl_add_u32 %[a], %[b], 2, %[entl]\n\t"
                            "ds_write_b32 %[a], %[bits]\n\t"
                            "s_waitcnt lgkmcnt(1)\n\t"
                            "v_alignbit_b32 %[bits], v63, v62, %[p]\n\t"
                            "v_and_b32 %[a], 1, %[bits]\n\t"
                            "v_cmp_eq_u32 vcc, 0, %[a]\n\t"                   // lanes whose block has an explicit header (Terse.hpp:361)
                            "s_cbranch_vccz 5f\n\t"
                            "s_ff1_i32_b64 %[first], vcc\n\t"
                            "v_readlane_b32 %[h], %[bits], %[first]\n\t"
                            "s_bfe_u32 %[w], %[h], 0x30001\n\t"               // Terse.hpp:362
                            "s_cmp_lg_u32 %[w], 7\n\t"
                            "s_cbranch_scc0 6f\n\t"
                            "s_addc_u32 %[b], %[b], %[first]\n\t"             // b += first + 1 (SCC = 1)
                            "s_mul_i32 %[t], %[first], %[stride]\n\t"
                            "s_mul_i32 %[stride], %[w], 12\n\t"
                            "s_add_i32 %[pw], %[pw], %[t]\n\t"
                            "s_add_i32 %[stride], %[stride], 1\n\t"
                            "s_add_i32 %[pw], %[pw], %[stride]\n\t"
                            "s_add_i32 %[pw], %[pw], 3\n"                      // pos += first * stride + 4 + 12 w
                            "3:\n\t"
                            "v_mul_u32_u24 %[ls], %[stride], %[lane]\n\t"
                            "s_lshl_b32 %[t], %[w], 26\n\t"
                            "s_add_i32 %[se], %[ebase], %[t]\n\t"
                            "s_mul_i32 %[t], %[stride], 63\n\t"
                            "s_sub_i32 %[pwmax], %[pwend], %[t]\n"
                            "4:\n\t"
                            "s_cmp_lt_i32 %[b], %[lim]\n\t"
                            "s_cbranch_scc0 9f\n\t"
                            "s_cmp_lt_i32 %[pw], %[pwmax]\n\t"
                            "s_cbranch_scc1 1b\n\t"
                            "s_branch 9f\n"
                            "5:\n\t"                                           // 64 blocks repeat the width
                            "s_lshl_b32 %[t], %[stride], 6\n\t"
                            "s_add_i32 %[pw], %[pw], %[t]\n\t"
                            "s_add_i32 %[b], %[b], 64\n\t"
                            "s_branch 4b\n"
                            "6:\n\t"                                           // widths >= 7: Terse.hpp:364-370
                            "s_bfe_u32 %[t], %[h], 0x20004\n\t"
                            "s_add_i32 %[w], %[t], 7\n\t"
                            "s_mul_i32 %[t], %[first], %[stride]\n\t"
                            "s_add_i32 %[pw], %[pw], %[t]\n\t"
                            "s_add_i32 %[pw], %[pw], 6\n\t"
                            "s_cmp_lg_u32 %[w], 10\n\t"
                            "s_cbranch_scc1 7f\n\t"
                            "s_bfe_u32 %[t], %[h], 0x60006\n\t"
                            "s_add_i32 %[w], %[t], 10\n\t"
                            "s_add_i32 %[pw], %[pw], 6\n"
                            "7:\n\t"
                            "s_cmp_gt_u32 %[w], %[maxw]\n\t"
                            "s_cbranch_scc1 8f\n\t"
                            "s_add_i32 %[b], %[b], %[first]\n\t"
                            "s_add_i32 %[b], %[b], 1\n\t"
                            "s_mul_i32 %[stride], %[w], 12\n\t"
                            "s_add_i32 %[pw], %[pw], %[stride]\n\t"
                            "s_add_i32 %[stride], %[stride], 1\n\t"
                            "s_branch 3b\n"
                            "8:\n\t"
                            "s_mov_b32 %[bad], 1\n"
                            "9:\n"
                            : [pw] "+s"(pw), [b] "+s"(b), [w] "+s"(w_prev), [stride] "+s"(stride), [se] "+s"(s_e), [pwmax] "+s"(pw_max),
                              [ls] "+v"(v_ls), [bad] "+s"(s_bad), [first] "=&s"(t_first), [h] "=&s"(t_h), [t] "=&s"(t_t), [p] "=&v"(t_p),
                              [a] "=&v"(t_a), [bits] "=&v"(t_bits)
                            : [lim] "s"(fast_lim), [lane] "v"(lane), [entl] "v"(v_entl), [ebase] "s"(e_base), [pwend] "s"(pw_end),
                              [maxw] "n"(kMaxW)
                            : "vcc", "scc", "memory", "v62", "v63");
                        pos = pw - base8 + 32u * (uint32_t)c_lo - frame_sh;
                        bad = s_bad != 0u;
                    }
                    if (bad || b >= end_nom || b == n_blocks) break;                             // the super-step is complete
                    // ---- refill the window ------------------------------------------------------------------------------------
                    const uint32_t stride = 1u + kBlock * w_prev;
                    const uint32_t need_lo = (frame_sh + pos) >> 5;
                    const uint32_t need_hi = ((frame_sh + pos + 63u * stride) >> 5) + 2;
                    if ((int32_t)need_lo < c_lo || (int32_t)need_hi > c_hi) {
                        c_lo = (int32_t)(((frame_dw + need_lo) & ~3ull) - frame_dw);
                        c_hi = c_lo + kChunkDw;
                        const uint64_t d0 = (uint64_t)((int64_t)frame_dw + c_lo);
                        if (base16 && (d0 & 3) == 0 && d0 + kChunkDw <= n_dw) {
                            // straight into LDS (LDS-DMA: 1 KB per instruction, destination = wave-uniform base + lane * 16):
                            // no staging registers
#pragma unroll
                            for (int it = 0; it < kChunkDw / (kWave * 4); ++it)
                                lds_dma16(s32 + d0 + it * kWave * 4, lane * 16u, (uint32_t)(uintptr_t)&s_chunk[it * kWave * 4]);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            // (Touching the window behind this one -- one dword of each of its lines by LDS-DMA into a scratch
                            // row, so that the next refill finds them in L2: 1.5 % faster right behind an encode, 16 % slower
                            // with the stream cache-resident: dropped.)
                        } else {
                            for (uint32_t i = lane * 4; i < (uint32_t)kChunkDw; i += kWave * 4) {
                                const uint64_t d = d0 + i;
                                uint4 x;
                                x.x = d < n_dw ? s32[d] : 0u; x.y = d + 1 < n_dw ? s32[d + 1] : 0u;
                                x.z = d + 2 < n_dw ? s32[d + 2] : 0u; x.w = d + 3 < n_dw ? s32[d + 3] : 0u;
                                *reinterpret_cast<uint4*>(&s_chunk[i]) = x;
                            }
                        }
                        if ((int32_t)b < fast_lim) continue;                          // go on with fast steps
                    }
                    // ---- general step: the frame's last (up to 64) blocks -------------------------------------------------------
                    const uint32_t lpos = pos + lane * stride;
                    const uint32_t fbit = frame_sh + lpos - 32u * (uint32_t)c_lo;
                    const uint32_t bits = __builtin_amdgcn_alignbit(s_chunk[(fbit >> 5) + 1], s_chunk[fbit >> 5], fbit);
                    const uint32_t left = n_blocks - b;                               // candidates left in the frame
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
                    } else {                                                          // the rest of the window repeats w_prev
                        const uint32_t cnt = left < 64u ? left : 64u;
                        new_pos = pos + cnt * stride;
                        if (b + cnt == n_blocks) new_pos = pos + (cnt - 1) * stride + 1u + nb_last * w_prev;
                        new_b = b + cnt;
                    }
                    if (lane < new_b - b) ent[b - beg_b + lane] = frame_sh + lpos + (w_prev << kPosBits);
                    pos = new_pos;
                    w_prev = e_w;
                    b = new_b;
                    if (pos > limit + 64u * 400u) { bad = true; break; }              // ran away: corrupt stream
                    if (b == n_blocks) break;
                }
                if (!bad && b == n_blocks)                                            // S_f = 1 + bits/8 (Terse.hpp:547) / the next part's state
                    bad = at_end ? !(pos <= limit && 1 + ((uint64_t)ppos0 + pos) / 8 == fe - fo) : !(ppos0 + pos == ppos1 && w_prev == pw1);
                if (bad && lane == 0) s_err = 1u;
                if (lane == 0) ent[b - beg_b] = (frame_sh + pos) | (w_prev << kPosBits);   // H(first block the walker has not seen)
                // A stream with an explicit header every few blocks costs this walker a step per header (10 x the time of a
                // run-dominated frame); false chains merge quickly in such streams, so the frame goes to the
                // position-parallel walk instead (decode_seg.hip).  Decided after super-steps 0, 3 and 11 on the width
                // changes inside the super-step just walked (9 or 12 widths per lane, outside the step loop).  (Probing the
                // first 256 blocks instead -- a second bound in the fast loop -- hands a header-dense frame over after 0.06
                // instead of 0.13 ms but cost every other stack 6-60 %: the loop bound became loop-variant.)
                if (defer && !bad && (s <= 1u || s == 3u || s == 11u) && end_nom == step_begin(s + 1u) && end_nom < n_blocks) {   // (1: a frame near the line that was early at 0 -- 37 % of a Poisson(3) stack -- sees the whole stack's count now, not three super-steps later)
                    const uint32_t per = (end_nom - beg_b) / kWave;                   // widths per lane (whole groups)
                    uint32_t changes = 0;
#pragma unroll
                    for (int i = 0; i < kStepBlocks / kWave; ++i) {
                        const uint32_t at = lane * per + i;
                        if ((uint32_t)i < per) changes += (ent[at] >> kPosBits) != (ent[at + 1] >> kPosBits) ? 1u : 0u;
                    }
                    const uint32_t inc = wave_inclusive_scan(changes);
                    const uint32_t chg = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    // (2000 x 512^2 u16 frames whose blocks change their width with probability d, runs geometric like detector noise,
                    // tools/dens_time.py: kept 0.36 / 0.57 / 1.19 / 1.33 ms at d = 0.10 / 0.18 / 0.26 / 0.30 -- steeper than the number
                    // of steps: the extraction waves share the walker's CU --, handed over 0.72 .. 0.95 ms whatever d: the line is at
                    // 2 changes in 9 blocks.  Round 3's line, 2 in 7, came from REGULAR patterns, where the walker does better.)
                    // What a stack must not do is SPLIT: the frames handed over start when the last kept one is through, and their
                    // walk and extraction take 0.5 ms however few they are -- Poisson(3) counts (d = 0.25) half kept, half not:
                    // 1.26 ms; d = 0.18 with the line at 2 in 9: a few frames' first 192 blocks read 23 %, 1.11 instead of 0.57 ms.
                    // So a frame whose own count is within 8 % of the line decides by the STACK's: every frame adds its first
                    // super-step's counts to one of 64 accumulators (a fire-and-forget atomic each, 64 lines: 4000 same-line atomics
                    // with their results awaited cost 0.6 ms) and reads their sum.
                    const uint32_t blk = per * kWave;
                    uint64_t* const slots = reinterpret_cast<uint64_t*>(defer) - kDeferSlots * kDeferSlotWords;
                    if (s == 0u && lane == 0)
                        __hip_atomic_fetch_add(slots + (blockIdx.x % kDeferSlots) * kDeferSlotWords, ((uint64_t)chg << 32) | blk, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                    uint64_t d_chg = chg, d_blk = blk;
                    if (chg * 100u > 14u * blk && chg * 100u < 30u * blk) {
                        // (the stack's frames reach this point within microseconds of each other: a frame that is early waits -- a
                        // few polls, bounded -- until 64 of them have reported, or it would decide on its own count after all and
                        // be handed over three super-steps later)
                        const uint32_t enough = (gridDim.x < 64u ? gridDim.x : 64u) * blk;
                        uint32_t c = 0, n = 0;
                        for (int poll = 0; poll < 24; ++poll) {
                            const uint64_t v = __hip_atomic_load(slots + lane * kDeferSlotWords, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            c = (uint32_t)(v >> 32); n = (uint32_t)v;            // (2000 frames x 768 blocks: the sums fit 32 bits per slot and in all)
                            c = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(c), 63);
                            n = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n), 63);
                            if (s != 0u || n >= enough) break;
                            __builtin_amdgcn_s_sleep(16);
                        }
                        if (s == 0u) { d_chg += c; d_blk += n; }                 // (own counts may or may not have arrived: once more or less)
                        else if (n) { d_chg = c; d_blk = n; }
                    }
                    // (3: listed WITHOUT a search for runs -- a frame past this line has runs of four or five blocks, nothing the run
                    // guess of decode_seg.hip could start in: the search cost every Poisson(3) frame 44 of 350 us)
                    if (d_chg * (uint64_t)TRPX_DEFER_DEN > (uint64_t)TRPX_DEFER_NUM * d_blk && lane == 0) s_err = 3u;
#ifdef TRPX_DEFER_STATS
                    if (lane == 0) {       // diagnostic build: status[2..4] = frames handed over after super-step 0 / 3 / 11, [5] = decisions by the stack's count, [6] = sum of own densities (per mille) at step 0
                        if (d_chg * (uint64_t)TRPX_DEFER_DEN > (uint64_t)TRPX_DEFER_NUM * d_blk) atomicAdd(status + (s == 0u ? 2 : s == 3u ? 3 : 4), 1u);
                        if (d_blk != blk) atomicAdd(status + 5, 1u);
                        if (s == 0u) atomicAdd(status + 6, chg * 1000u / blk);
                    }
#endif
                }
            }
        } else if (s >= 1) {
            // The SIMD's arbiter serves equal-priority waves oldest first: with all extraction waves at priority 0 the
            // workgroups that arrived first on the CU finished after 0.17 ms, the last ones after 0.30 ms (tools/dec_stamps.py),
            // a long tail with few waves left to hide latency.  Alternating the extraction waves' priority between 0 and 1
            // from super-step to super-step, in opposite phase for odd and even wave slots, gives every workgroup the same
            // share over time: all finish within 20 % of each other, the kernel 6 % sooner.  (Three levels, rotating the
            // walkers' priority as well, or a rotation every second super-step: no better.)
#ifndef TRPX_DEC_PRIO_FLAT
            if (((hw_slot + s) & 1u) != 0u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
            // ---- extraction of super-step s-1: this wave's groups, one after the other ----------------------------------
            // One LDS read gives a lane the end of its block's payload and the block's width (H of the NEXT block).  Every
            // lane loads the stream dwords of its own block straight from L2 (the walker has just pulled them through):
            // dwordx4 loads starting at the dword that holds the block's first payload bit, all kRawDw dwords whatever the
            // widths (a 16-byte load more per lane is cheaper than a wavefront max of the widths, and the extra bytes are
            // the neighbours').  The width-specialised bodies work on registers only and leave the packed pixels in
            // registers; the stores follow once per group, behind all passes.  (Measured and dropped, DESIGN.md 4.3: the
            // next group's loads issued before this group's stores -- a second register set, 64 VGPRs with spills; a
            // group's bytes fetched into LDS by LDS-DMA and its pixels leaving through LDS rows as whole lines -- fewer,
            // cheaper vector-memory instructions, but 60 % more scalar and LDS instructions: 0.39 instead of 0.29 ms.)
            const uint32_t pbuf = (s - 1) & 1u;
            constexpr int kRawDw = Cfg::kRawDw;
            const uint32_t* __restrict__ fbase = s32 + frame_dw;          // wave-uniform base; per-lane 32-bit dword offsets
            const uint32_t step0 = step_begin(s - 1u);
            const uint32_t step1 = step_begin(s) < n_blocks ? step_begin(s) : n_blocks;
            const uint32_t* const ent = s_posx + pbuf * kPosEntries;
#ifdef TRPX_DEC_WALK_ONLY
            const uint32_t gpw = 0;                                                   // diagnostic build (tools): time the walker alone
#else
            const uint32_t gpw = (step_begin(s) - step0) / (uint32_t)((kFrameWaves - 1) * kWave);   // groups per wave in this super-step: 1 or kGpw
#endif
            // (blocks in front of n_whole are full blocks of this super-step: a group of 64 of them leaves as whole lines)
            [[maybe_unused]] const uint32_t n_whole = nb_last == (uint32_t)kBlock || step1 < n_blocks ? step1 : step1 - 1u;
            [[maybe_unused]] bool carry_live = false;                                 // the row's front holds the group's first out_c bytes
#pragma unroll 1
            for (uint32_t gq = 0; gq < gpw; ++gq) {
                const uint32_t g0 = step0 + ((uint32_t)(wave - 1) * gpw + gq) * kWave;
                if (g0 >= step1) break;                                               // wave-uniform: group past the frame's end
                const uint32_t blk = g0 + lane;
                uint32_t w = 0, q = frame_sh + limit;                                 // (lanes behind the super-step's end: a position behind every block's)
                if (blk < step1) {
                    const uint32_t pw = ent[blk - step0 + 1u];                        // H(blk + 1)
                    w = pw >> kPosBits;
                    q = (pw & kPosMask) - (blk + 1u == n_blocks ? nb_last : (uint32_t)kBlock) * w;   // first payload bit, relative to dword frame_dw
                }
                if constexpr (MODE == 2) {                                            // the index instead of the pixels
                    if (blk < step1) {
                        const_cast<uint8_t*>(idx_widths)[frame * g.n_blocks + pb0 + blk] = (uint8_t)w;
                        if (((pb0 + blk) & (uint32_t)(kTileBlocks - 1)) == 0u)
                            const_cast<uint64_t*>(idx_group_off)[frame * g.n_tiles + (pb0 + blk) / (uint32_t)kTileBlocks] =
                                (uint64_t)ppos0 + ((ent[blk - step0] & kPosMask) - frame_sh);
                    }
                    continue;
                }
                const uint32_t dq = q >> 5, sq = q & 31u;
                const uint32_t last_dw = (uint32_t)__builtin_amdgcn_readlane((int)dq, 63) + (uint32_t)kRawDw;   // lanes ascend in position
                uint32_t raw[kRawDw];
                if (frame_dw + last_dw <= n_dw) {                                     // wave-uniform: the loads stay inside the stream
#pragma unroll
                    for (int i = 0; i < kRawDw / 4; ++i) {
                        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                        u4 x4;
                        __builtin_memcpy(&x4, fbase + dq + 4 * i, 16);                 // dword-aligned 16-byte load
                        raw[4 * i] = x4.x; raw[4 * i + 1] = x4.y; raw[4 * i + 2] = x4.z; raw[4 * i + 3] = x4.w;
                    }
                } else {                                                              // the stack's last bytes: guarded element loads
#pragma unroll
                    for (int i = 0; i < kRawDw; ++i) {
                        const uint64_t d = frame_dw + dq + i;
                        raw[i] = d < n_dw ? s32[d] : 0u;
                    }
                }
#ifdef TRPX_DEC_NO_EXTRACT
                if (q == 0xFFFFFFFFu) fout[0] = (T)(w + raw[0]);                      // (diagnostic build: loads only)
                continue;
#endif
                const bool full = blk < step1 && (blk + 1 < n_blocks || nb_last == (uint32_t)kBlock);
                T* __restrict__ dst = fout + (uint64_t)blk * kBlock;
                uint64_t todo = __ballot(full);
                if (Cfg::kOutStaged && todo == ~0ull) {
                    // 64 full blocks of 32-bit pixels: every lane leaves its 12 pixels in the wave's LDS row, then the wave
                    // stores the group 16 bytes per lane -- whole lines per store instruction instead of 48-byte runs
                    uint32_t* const stage = s_out[wave - 1] + kStageCarryDw;            // (head room in front: store_group_lines)
                    uint32_t* const row = stage + lane * (kBlock * (uint32_t)sizeof(T) / 4u);
                    [[maybe_unused]] const bool cont = LINES && gq + 1u < gpw && g0 + 2u * (uint32_t)kWave <= n_whole;   // this wave's next group is staged too
                    while (todo) {
                        const int l0 = __builtin_ctzll(todo);
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l0);
                        uint32_t wd = w0;
                        asm volatile("" : "+s"(wd));                                  // (a copy the compiler cannot equate with the lanes' own w: the dispatch on it stays scalar)
                        const uint64_t mine = __ballot(w == w0);
                        uint32_t ss = sq;
                        asm volatile("" : "+v"(ss));                                  // keep the specialised bodies out of LICM's reach
                        // Every lane runs the picked width's body and only the lanes of that width keep the result; the
                        // loop has no divergent branch (the lane mask is applied inside stage_packed_masked), so its
                        // wave-uniform branches stay plain s_cmp / s_cbranch pairs.  With `if (mine) body` the compiler
                        // structurises the dispatch tree: a flag register and three scalar instructions more per level,
                        // 30 per pass -- and the CU's scalar unit is this kernel's busiest.
                        uint32_t o[PackedDwords<T>::n];
                        UnpackRegsDispatch<T, 0, PixelTraits<T>::bits>::run(raw, ss, wd, o);
                        stage_packed_masked<T>(row, o, mine);
                        todo &= ~mine;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if constexpr (!LINES) store_group<T, false>(stage, fout + (uint64_t)g0 * kBlock);
                    else {
                        store_group_lines<T>(stage, fout + (uint64_t)g0 * kBlock, out_c, carry_live, cont);
                        carry_live = cont;
                    }
                    __builtin_amdgcn_wave_barrier();                                  // (the row is rewritten by the next group)
                } else if constexpr (sizeof(T) == 4) {                                // 32-bit pixels, a group with fewer than 64 full blocks: stores inside the bodies
                    while (todo) {
                        const int l0 = __builtin_ctzll(todo);
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l0);
                        uint32_t wd = w0;
                        asm volatile("" : "+s"(wd));                                  // (a copy the compiler cannot equate with the lanes' own w: the dispatch on it stays scalar)
                        const bool mine = full && w == w0;
                        uint32_t ss = sq;
                        asm volatile("" : "+v"(ss));
                        if (mine) UnpackStoreDispatch<T, 0, PixelTraits<T>::bits>::run(raw, ss, wd, dst);
                        todo &= ~__ballot(mine);
                    }
                } else {
                    uint32_t o[PackedDwords<T>::n];
                    while (todo) {
                        const int l0 = __builtin_ctzll(todo);
                        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)w, l0);
                        uint32_t wd = w0;
                        asm volatile("" : "+s"(wd));                                  // (a copy the compiler cannot equate with the lanes' own w: the dispatch on it stays scalar)
                        const bool mine = full && w == w0;
                        uint32_t ss = sq;
                        asm volatile("" : "+v"(ss));                                  // keep the specialised bodies out of LICM's reach
                        if (mine) UnpackRegsDispatch<T, 0, PixelTraits<T>::bits>::run(raw, ss, wd, o);
                        todo &= ~__ballot(mine);
                    }
#ifdef TRPX_DEC_NO_STORE
                    if (full) diag_acc ^= o[0];
#else
                    if (full) store_packed<T>(dst, o);
#endif
                }
                if (blk + 1 == n_blocks && blk < step1 && !full) {                    // the frame's last, partial block
                    const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
                    uint32_t p = q;
                    for (uint32_t k = 0; k < nb_last; ++k) {
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
#ifdef TRPX_DEC_STAMPS
        { const uint64_t t1 = __builtin_readcyclecounter(); st_work += t1 - st_t0; st_t0 = t1; }
#endif
        __syncthreads();                               // super-step boundary: entries of step s published, step s-1 consumed
#ifdef TRPX_DEC_STAMPS
        { const uint64_t t1 = __builtin_readcyclecounter(); st_wait += t1 - st_t0; st_t0 = t1; }
#endif
        if (s_err) break;
    }
#ifdef TRPX_DEC_STAMPS
    // diagnostic build (tools/dec_stamps.py): per wave role, cycles spent working / waiting at the super-step barrier, and
    // the wave's SIMD + slot; overwrites the first pixels of the frame
    if (lane == 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t* dbg = reinterpret_cast<uint32_t*>(fout) + 8 * wave;
        dbg[0] = (uint32_t)st_work; dbg[1] = (uint32_t)st_wait; dbg[2] = hwid; dbg[3] = (uint32_t)(st_t0 - st_start);
        dbg[4] = (uint32_t)(st_start & 0xFFFFFFFFu);
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[5] = xcc;
        dbg[6] = dbg[7] = 0;
    }
#endif
#ifdef TRPX_DEC_NO_STORE
    if (diag_acc == 0x12345678u) fout[threadIdx.x] = (T)diag_acc;
#endif
    if (threadIdx.x == 0 && s_err != 0u) {
        asm volatile("" ::: "memory");                                     // (the LDS copies, not the registers they came from)
        uint32_t* const status_l = reinterpret_cast<uint32_t*>((uintptr_t)s_keep[0]);
        uint32_t* const defer_l = reinterpret_cast<uint32_t*>((uintptr_t)s_keep[1]);
        if (s_err == 1u) atomicMax(&status_l[0], 5u);                      // TRPX_ERR_CORRUPT
        // listed: k_seg_listed + k_decode_frames_indexed do it (bit 31: so dense that a search for runs is a waste of time)
        if (s_err >= 2u) {
            defer_l[1u + atomicAdd(&defer_l[0], 1u)] = (uint32_t)s_keep[2] | (s_err == 3u ? 0x80000000u : 0u);
            // (their number, for frames of several wavefronts' worth -- k_seg_wg, decode_seg.hip; codec_common.hpp: the word at [-2])
            if (s_err == 3u) atomicAdd(reinterpret_cast<unsigned long long*>(defer_l) - 2, 1ull);
        }
    }
}

template <typename T, bool LINES>
__global__ __launch_bounds__(kFrameThreads, sizeof(T) == 4 ? 6 : 8) void k_decode_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               T* __restrict__ pixels_out, uint32_t* __restrict__ defer,
                                                               uint32_t* __restrict__ status) {
    decode_frame_body<T, 0, false, LINES>(terse, terse_bytes, frame_offsets, g, pixels_out, defer, status, blockIdx.x, nullptr, nullptr);
}

// The same over the PARTS of large frames (decode_part.hip has built the table): one workgroup per part.
template <typename T>
__global__ __launch_bounds__(kFrameThreads, sizeof(T) == 4 ? 6 : 8) void k_decode_parts(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               T* __restrict__ pixels_out, uint32_t* __restrict__ status,
                                                               const PartDesc* __restrict__ parts, const uint32_t* __restrict__ frame_mode) {
    if (status[0] != 0u) return;
    if (frame_mode && frame_mode[parts[blockIdx.x].frame] != 0u) return;   // (index route: this frame is extracted through its decode index)
    decode_frame_body<T, 0, true, true>(terse, terse_bytes, frame_offsets, g, pixels_out, nullptr, status, blockIdx.x, nullptr, nullptr, parts);
}

// The same with the widths given (IDX above): every frame of the stack (list == nullptr) or the frames list[1 .. list[0]].
template <typename T>
__global__ __launch_bounds__(kFrameThreads, sizeof(T) == 4 ? 6 : 8) void k_decode_frames_indexed(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               const uint8_t* __restrict__ widths, const uint64_t* __restrict__ group_off,
                                                               const uint32_t* __restrict__ list, T* __restrict__ pixels_out,
                                                               uint32_t* __restrict__ status) {
    uint64_t frame = blockIdx.x;
    if (list) {
        if (blockIdx.x >= list[0] || status[0] != 0u) return;
        frame = list[1u + blockIdx.x] & 0x7FFFFFFFu;
    }
#ifdef TRPX_IDX_LINES                                     // (experiment: line images in the indexed kernel, as k_decode_frames<T, true> writes them)
    decode_frame_body<T, 1, false, true>(terse, terse_bytes, frame_offsets, g, pixels_out, nullptr, status, frame, widths, group_off);
#else
    decode_frame_body<T, 1>(terse, terse_bytes, frame_offsets, g, pixels_out, nullptr, status, frame, widths, group_off);
#endif
}

// Large frames whose index is known, in units of `unit_blocks` blocks (a multiple of 256 and of a super-step): the same body, one
// workgroup per unit -- for 8- and 16-bit pixels faster than the tiled kernel (128 x 2048^2 u16: k_unpack_tiles 370 us).
template <typename T>
__global__ __launch_bounds__(kFrameThreads, sizeof(T) == 4 ? 6 : 8) void k_decode_units_indexed(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               const uint8_t* __restrict__ widths, const uint64_t* __restrict__ group_off,
                                                               uint32_t unit_blocks, T* __restrict__ pixels_out, uint32_t* __restrict__ status,
                                                               const uint32_t* __restrict__ frame_mode) {
    if (status[0] != 0u) return;
    if (frame_mode && frame_mode[blockIdx.x / ((g.n_blocks + unit_blocks - 1u) / unit_blocks)] == 0u) return;   // (extracted part by part)
    decode_frame_body<T, 1, true>(terse, terse_bytes, frame_offsets, g, pixels_out, nullptr, status, blockIdx.x, widths, group_off,
                                  reinterpret_cast<const PartDesc*>((uintptr_t)unit_blocks));
}
template <typename T>
static uint32_t units_choose(const DecodeArgs& a, uint64_t* cost_out);
template <typename T>
static hipError_t launch_decode_units_indexed_t(const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode) {
    // Unit size: whole super-steps (a multiple of 256 blocks: units start on index groups), 8 .. 32 of them, such that the stack's
    // units come out as whole rounds of the workgroups the GPU holds at once -- 128 x 2048^2 u16 frames: 57 units of 6144 blocks per
    // frame are 3.6 rounds, the last one 0.55 full, and every unit pays a super-step of pipeline fill; 16 units of 22 272 blocks are
    // one round (the 512 x 512 stack's shape: one workgroup per 21 846-block frame).
    const uint32_t unit_blocks = units_choose<T>(a, nullptr);
    const uint32_t upf = (a.geom.n_blocks + unit_blocks - 1u) / unit_blocks;
    hipLaunchKernelGGL((k_decode_units_indexed<T>), dim3(a.n_frames * upf), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                       a.frame_offsets, a.geom, static_cast<const uint8_t*>(a.widths), static_cast<const uint64_t*>(a.tile_off), unit_blocks,
                       static_cast<T*>(a.pixels_out), a.status, frame_mode);
    return hipGetLastError();
}
// (cost: rounds of resident workgroups x (blocks per unit + a super-step of pipeline fill); a round = what is resident at once --
// a stack of 2128 frames of 480 x 512 pixels decodes whole in two rounds, the second one 4 % full: 0.267 ms against 0.230 for 2000
// frames of 512 x 512, the same bytes)
constexpr uint64_t units_resident(size_t pixel_bytes) { return (uint64_t)(pixel_bytes == 4 ? 6 : 8) * 256u; }
template <typename T>
static uint32_t units_choose(const DecodeArgs& a, uint64_t* cost_out) {
    constexpr uint32_t step = FrameCfg<T>::kStepBlocks;
    static_assert(step % kTileBlocks == 0, "units start on index groups");
    // (less a margin: 2048 units on 2048 slots ran as two rounds)
    const uint64_t resident = units_resident(sizeof(T)) * 15u / 16u;
    uint32_t unit_blocks = 8u * step;
    uint64_t best = ~0ull;
    for (uint32_t k = 8; k <= 32; ++k) {
        const uint64_t u = (uint64_t)k * step, units = (uint64_t)a.n_frames * ((a.geom.n_blocks + u - 1) / u);
        const uint64_t cost = ((units + resident - 1) / resident) * (u + step);
        if (cost < best) { best = cost; unit_blocks = (uint32_t)u; }
    }
    if (cost_out) *cost_out = best;
    return unit_blocks;
}
hipError_t launch_decode_units_indexed(int dtype, const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode) {
    switch (dtype) {
    case 0: return launch_decode_units_indexed_t<uint8_t>(a, st, frame_mode);
    case 1: return launch_decode_units_indexed_t<int8_t>(a, st, frame_mode);
    case 2: return launch_decode_units_indexed_t<uint16_t>(a, st, frame_mode);
    case 3: return launch_decode_units_indexed_t<int16_t>(a, st, frame_mode);
    case 4: return launch_decode_units_indexed_t<uint32_t>(a, st, frame_mode);
    case 5: return launch_decode_units_indexed_t<int32_t>(a, st, frame_mode);
    }
    return hipErrorInvalidValue;
}

// The index of every frame of a stack of small frames (MODE 2 above); T stands for the width limit only.
template <typename T>
__global__ __launch_bounds__(kFrameThreads, sizeof(T) == 4 ? 6 : 8) void k_index_frames(const uint8_t* __restrict__ terse, uint64_t terse_bytes,
                                                               const uint64_t* __restrict__ frame_offsets, FrameGeom g,
                                                               uint8_t* __restrict__ widths, uint64_t* __restrict__ group_off,
                                                               uint32_t* __restrict__ defer, uint32_t* __restrict__ status) {
    decode_frame_body<T, 2>(terse, terse_bytes, frame_offsets, g, static_cast<T*>(nullptr), defer, status, blockIdx.x, widths, group_off);
}

// Writes a.widths / a.tile_off with the per-frame walker; the frames it lists in a.defer are left to launch_seg_listed.
hipError_t launch_index_frames(uint32_t max_w, const DecodeArgs& a, bool clear_status, hipStream_t st) {
    hipLaunchKernelGGL(k_zero_words<0>, dim3(1), dim3(kThreads), 0, st, reinterpret_cast<uint64_t*>(a.defer) - kDeferSlots * kDeferSlotWords,
                       (uint64_t)(kDeferSlots * kDeferSlotWords + 1), reinterpret_cast<uint64_t*>(a.status), (uint64_t)(clear_status ? 4 : 0));
    if (max_w <= 8u)
        hipLaunchKernelGGL((k_index_frames<uint8_t>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                           a.frame_offsets, a.geom, a.widths, a.tile_off, a.defer, a.status);
    else if (max_w <= 16u)
        hipLaunchKernelGGL((k_index_frames<uint16_t>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                           a.frame_offsets, a.geom, a.widths, a.tile_off, a.defer, a.status);
    else
        hipLaunchKernelGGL((k_index_frames<uint32_t>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                           a.frame_offsets, a.geom, a.widths, a.tile_off, a.defer, a.status);
    return launch_seg_listed(a, max_w, st);
}

template <typename T>
static hipError_t launch_decode_frames_indexed_t(const DecodeArgs& a, const uint32_t* list, hipStream_t st) {
#ifdef TRPX_IDX_AS_UNITS                                  // (experiment: the units instantiation on whole small frames)
    if (!list) return launch_decode_units_indexed_t<T>(a, st, nullptr);
#endif
    if (!list && a.geom.n_blocks > 8u * FrameCfg<T>::kStepBlocks) {
        // A frame count a little over a multiple of what the GPU holds leaves the last round of whole frames nearly empty: units of
        // the frames -- the same body -- where their rounds come out at least a tenth cheaper.
        uint64_t cost_units = 0;
        units_choose<T>(a, &cost_units);
        const uint64_t resident = units_resident(sizeof(T));
        const uint64_t cost_whole = ((a.n_frames + resident - 1) / resident) * ((uint64_t)a.geom.n_blocks + FrameCfg<T>::kStepBlocks);
        if (10u * cost_units < 9u * cost_whole) return launch_decode_units_indexed_t<T>(a, st, nullptr);
    }
    hipLaunchKernelGGL((k_decode_frames_indexed<T>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                       a.frame_offsets, a.geom, static_cast<const uint8_t*>(a.widths), static_cast<const uint64_t*>(a.tile_off), list,
                       static_cast<T*>(a.pixels_out), a.status);
    return hipGetLastError();
}

// Per-frame decode of frames whose index (a.widths, a.tile_off) is known; same preconditions as launch_decode_frames.
hipError_t launch_decode_frames_indexed(int dtype, const DecodeArgs& a, const uint32_t* list, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_decode_frames_indexed_t<uint8_t>(a, list, st);
    case 1: return launch_decode_frames_indexed_t<int8_t>(a, list, st);
    case 2: return launch_decode_frames_indexed_t<uint16_t>(a, list, st);
    case 3: return launch_decode_frames_indexed_t<int16_t>(a, list, st);
    case 4: return launch_decode_frames_indexed_t<uint32_t>(a, list, st);
    case 5: return launch_decode_frames_indexed_t<int32_t>(a, list, st);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_decode_frames_t(const DecodeArgs& a, hipStream_t st) {
    uint32_t* defer = a.defer && a.seg_ws ? a.defer : nullptr;
    const bool chain = a.chain && a.parts && a.parts_per_frame > 1u;
    if (chain) {                                                              // (the same words and the index route's own, in one launch)
        if (!defer) return hipErrorInvalidValue;
        const hipError_t e = launch_chain_zero(a, (uint32_t)PixelTraits<T>::bits, true, st);
        if (e != hipSuccess) return e;
    } else
        hipLaunchKernelGGL(k_zero_words<0>, dim3(1), dim3(kThreads), 0, st,
                           defer ? reinterpret_cast<uint64_t*>(defer) - kDeferSlots * kDeferSlotWords : static_cast<uint64_t*>(nullptr),
                           (uint64_t)(defer ? kDeferSlots * kDeferSlotWords + 1 : 0),
                           reinterpret_cast<uint64_t*>(a.status), (uint64_t)4);   // status block + the deferred-frame count and the stack statistics in front of it
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    if (chain) {
        // Large frames by the index route (decode_part.hip): one walk of many short parts writes the decode index, the frames
        // where that does not work out are listed and get theirs from the position-parallel walk, then every frame's tiles are
        // extracted with the widths given.
        if (!defer) return hipErrorInvalidValue;
        constexpr int dt = PixelTraits<T>::bits == 8 ? (PixelTraits<T>::is_signed ? 1 : 0)
                           : PixelTraits<T>::bits == 16 ? (PixelTraits<T>::is_signed ? 3 : 2) : (PixelTraits<T>::is_signed ? 5 : 4);
        constexpr bool narrow = sizeof(T) < 4;
        const uint32_t* frame_mode = nullptr;
        hipError_t e = launch_build_index_chain(a, (uint32_t)PixelTraits<T>::bits, narrow, &frame_mode, st);
        if (e != hipSuccess) return e;
        prof.mark(st);
        e = launch_seg_listed(a, (uint32_t)PixelTraits<T>::bits, st);
        if (e != hipSuccess) return e;
        // (the tiled kernel, or units of the per-frame decoder with the widths given: eight 4096^2 int32 frames 0.406 / 0.444 ms,
        // 200 x (1030 x 1065) u16 0.246 / 0.268, 128 x 2048^2 u16 0.594 / 0.554 -- tools/r5_ab.sh xtiles / xunits)
#ifdef TRPX_CHAIN_EXTRACT_TILES
        const bool tiles = TRPX_CHAIN_EXTRACT_TILES != 0;
#else
        const bool tiles = sizeof(T) == 4 || a.geom.n_blocks < (1u << 18);
#endif
        e = tiles ? launch_unpack_tiles(dt, a, st, narrow ? frame_mode : nullptr) : launch_decode_units_indexed(dt, a, st, narrow ? frame_mode : nullptr);
        if (e != hipSuccess) return e;
        if constexpr (narrow)                                                 // frames with few explicit headers: part by part, walker + extraction fused
            hipLaunchKernelGGL((k_decode_parts<T>), dim3(a.n_frames * a.parts_per_frame), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                               a.frame_offsets, a.geom, static_cast<T*>(a.pixels_out), a.status, static_cast<const PartDesc*>(a.parts), frame_mode);
        prof.mark(st);
        return hipGetLastError();
    } else if (a.parts && a.parts_per_frame > 1u) {
        // Large frames: cut into parts first (decode_part.hip: a walk-only pass from guessed states inside runs of equal widths,
        // verified link by link); frames whose parts cannot be established -- no runs to start from: header-dense data -- are
        // listed in a.defer as whole frames and take the position-parallel walk + tiled extraction below.
        if (!defer) return hipErrorInvalidValue;
        const hipError_t e = launch_build_parts(a, (uint32_t)PixelTraits<T>::bits, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_decode_parts<T>), dim3(a.n_frames * a.parts_per_frame), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                           a.frame_offsets, a.geom, static_cast<T*>(a.pixels_out), a.status, static_cast<const PartDesc*>(a.parts),
                           static_cast<const uint32_t*>(nullptr));
    } else if ((a.geom.n_values * sizeof(T)) % 128u == 0u && (uintptr_t)a.pixels_out % 128u == 0u)   // every frame starts a cache line
        hipLaunchKernelGGL((k_decode_frames<T, false>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
                           a.frame_offsets, a.geom, static_cast<T*>(a.pixels_out), defer, a.status);
    else
        hipLaunchKernelGGL((k_decode_frames<T, true>), dim3(a.n_frames), dim3(kFrameThreads), 0, st, a.terse, (uint64_t)a.terse_bytes,
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

// Preconditions (checked by the caller): block = 12, frame offsets known, frames of < 2^26 bits (less one step's overshoot);
// pixels_out aligned to the pixel type, any pixel count per frame.
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
