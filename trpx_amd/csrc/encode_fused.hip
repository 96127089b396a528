#include <stddef.h>
#include <utility>
// Single-pass TERSE encoder for gfx950 (CDNA4): pixels are read from HBM exactly once and every
// stream byte is written exactly once (algorithmic traffic N*sizeof(T) + S per frame).
// Replaces jpa::Terse::f_compress (reference include/Terse.hpp:500-549) and the Bit_pointer.hpp
// pack primitives (Bit_range::append_range :700-730, operator|= :628-649, Bit::set :490).
//
// Work decomposition
//   tile      = 1024 consecutive codec blocks (12 288 values; 768 blocks for 32-bit pixels) of ONE frame, one
//               256-thread workgroup; grid = (tiles per frame, frames), x fastest = dispatch in tile order.
//   sub-tile  = 256 blocks; lane `tid` owns block (r*256 + tid) of sub-tile r = 0..3, so every
//               wave-level load covers 1536 contiguous bytes (u16).
//   The serial bit cursor of the reference (Terse.hpp:504) becomes three prefix sums:
//     lanes -> wavefront DPP scan, wavefronts/sub-tiles -> LDS, tiles and frames -> 8-byte words in HBM
//     (agent-scope relaxed atomics, the data is the flag):
//       tile chain  (inside a frame)   : decoupled look-back over the frame's tile descriptors (bits)
//       frame chain (across the stack) : every tile ADDS its bits to its frame's accumulator; only the frame's first
//                                        tile walks the chain of complete frames (S_f = 1 + bits/8, Terse.hpp:547)
//                                        and hands the frame's base to its siblings through the frame's own line
//   Packing: every lane serialises its block -- header and payload as one bit string -- with code specialised
//   on the block's width W (all shifts static), ORs it into the workgroup's LDS staging image at its scanned
//   bit offset; the tile-relative image is flushed to HBM in 16-byte stores through one funnel shift.  A dword
//   shared by two tiles is stored by k_stitch from the bits both sides deposit in an exchange word, so
//   the output needs no pre-zeroing and nothing at a tile's end waits for another tile.
//
// Forward progress: tile i only ever waits for tiles < i.  Tiles are taken in blockIdx order,
// which the hardware dispatches in order; every wait is bounded and raises status[0] =
// TRPX_ERR_HIP-class timeout (7) instead of hanging, in which case the host API re-runs the
// stack through the two-pass pipeline (encode.hip).
#include "codec_common.hpp"
#include "encode_kernels.hpp"
#include "profile.hpp"
#include <stdlib.h>
#include <mutex>
#include <vector>

namespace trpx {

// sub-tiles per tile: 4 (1024 blocks) for 8/16-bit pixels, 5 (1280 blocks, 60 KB of pixels) for 32-bit pixels.  Measured with the
// round-1 chain design, 2000 x 512^2 u16: 3 @ 8 / 4 @ 6 / 5 @ 4 / 6 @ 4 sub-tiles @ workgroups per CU -> 0.40 / 0.325 / 0.38 / 0.354 ms;
// round 3, u16: 4 @ 8 / 5 @ 6 / 6 @ 5 / 5 @ 7 (6 VGPRs spilled) / 6 @ 6 (17 spilled) -> 0.258 / 0.253 / 0.256 / 0.270 / 0.277 ms (kept: 4 @ 8);
// round 3, eight 4096^2 int32 frames: 2 @ 6..8 / 3 @ 5 / 4 @ 4 / 5 @ 4 / 6 @ 3 -> 0.29 / 0.208 / 0.173 / 0.164 / 0.166 ms, 1000 noisy
// 512^2 int32 frames 0.39-0.44 / 0.333 / 0.272 / 0.249 / 0.260 ms: a 32-bit tile's fixed costs (look-back, flush, two barriers) weigh
// more than the workgroups its registers cost.
#ifndef TRPX_FUSED_SUB32
#define TRPX_FUSED_SUB32 5
#endif
#ifndef TRPX_FUSED_SUB16
#define TRPX_FUSED_SUB16 4
#endif
template <typename T> constexpr int sub_tiles() { return sizeof(T) <= 2 ? TRPX_FUSED_SUB16 : TRPX_FUSED_SUB32; }
// Workgroups per CU (what the VGPR budget and the half-size LDS image of fused_phase_rounds() allow).  8/16-bit pixels: eight
// (13.4 KB image; 64 VGPRs once a round's block metadata share one register: 0.268 -> 0.258 ms per 2000-frame u16 stack against
// seven).  32-bit pixels: four (five sub-tiles: 38.4 KB image, 117 VGPRs; with three sub-tiles: five workgroups, 83 VGPRs).
#ifndef TRPX_FUSED_OCC16
#define TRPX_FUSED_OCC16 8
#endif
#ifndef TRPX_FUSED_OCC32
#define TRPX_FUSED_OCC32 4
#endif
template <typename T> constexpr int fused_occupancy() { return sizeof(T) == 1 ? 8 : (sizeof(T) == 2 ? TRPX_FUSED_OCC16 : TRPX_FUSED_OCC32); }   // workgroups per CU (LDS image + VGPR budget)
// Every wait on another tile is bounded in WALL time: a poll loop gives up kWaitTicks of the 100 MHz realtime counter
// after it started (0.25 s; the whole 2000-frame launch takes 0.3 ms, so this only ever triggers when tiles do not
// make progress at all), checked every 64 polls.  The caller then sees TRPX_ERR_TIMEOUT in status[0]
// (trpx_encode_checked / the host wrappers re-run through the two-pass pipeline).
[[maybe_unused]] constexpr uint64_t kWaitTicks = 25u * 1000u * 1000u;
struct SpinGuard {
    uint64_t t0 = 0;
    uint32_t spins = 0;
    __device__ __forceinline__ bool expired() {
#ifdef TRPX_FORCE_TIMEOUT
        return true;                                     // test build: every wait that does not succeed at once gives up
#else
        if ((++spins & 63u) != 0u) return false;
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        if (t0 == 0) { t0 = now; return false; }
        return now - t0 > kWaitTicks;
#endif
    }
};

// Diagnostics (tools/stamps.py, tools/enc_time.py): only in builds with -DTRPX_DIAGNOSTICS; the product build folds
// every `diag(a) & x` test to false.  Bits: 1 = skip the look-back waits, 2 = skip the tail wait (both give WRONG
// output, timing experiments only), 4 = write s_memrealtime stamps per tile.
#ifdef TRPX_DIAGNOSTICS
#define TRPX_DIAG(a) ((a).debug)
#else
#define TRPX_DIAG(a) 0u
#endif

constexpr uint64_t kStInvalid = 0, kStAgg = 1, kStPrefix = 2;
constexpr uint32_t kGridY = 32768;                                      // frames per grid.z slice
constexpr int kWmaxShift = 56;                                           // bnd_pos[tile]: dword index | head side pending << 55 | widest block << 56
constexpr uint64_t kBndHead = 1ull << 55;                                // (= the head flag of the boundary's exchange word: k_stitch judges its neighbours by it)
constexpr uint64_t kHeadFlag = 1ull << 63, kTailFlag = 1ull << 62;   // boundary exchange words (k_encode_fused's end, k_stitch)
__device__ __forceinline__ uint64_t make_desc(uint64_t st, uint64_t v) { return (st << 62) | (v & ((1ull << 62) - 1)); }
__device__ __forceinline__ uint64_t desc_status(uint64_t d) { return d >> 62; }
__device__ __forceinline__ uint64_t desc_value(uint64_t d) { return d & ((1ull << 62) - 1); }

__device__ __forceinline__ uint64_t ld_desc(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of one value per lane over the wavefront (values < 2^62): three 21-bit limbs, each summed with the DPP scan
// ladder (64 x 2^21 < 2^32) -- the look-backs sit on every tile's critical path between packing and flush, and the
// ds_bpermute butterfly this replaces (12 LDS-crossbar round trips) cost ~0.4 us there.
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
    const uint32_t l0 = (uint32_t)v & 0x1FFFFFu, l1 = (uint32_t)(v >> 21) & 0x1FFFFFu, l2 = (uint32_t)(v >> 42);
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(l0), 63);
    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(l1), 63);
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(l2), 63);
    return (uint64_t)s0 + ((uint64_t)s1 << 21) + ((uint64_t)s2 << 42);
}

// Decoupled look-back by one full wavefront: sum of the values of desc[lo .. idx-1], using the
// nearest PREFIX descriptor as a shortcut.  Returns false on timeout.
__device__ bool lookback(const uint64_t* __restrict__ desc, int64_t idx, int64_t lo, uint64_t* result) {
    const int lane = lane_id();
    uint64_t acc = 0;
    int64_t pos = idx - 1;
    while (pos >= lo) {
        const int64_t my = pos - lane;
        const bool in = my >= lo;
        uint64_t g = in ? 0ull : make_desc(kStPrefix, 0);                // below `lo`: prefix 0 ends the chain
        uint32_t first_p = 64;
        SpinGuard guard;
        uint64_t need = ~0ull;
        for (;;) {
            // re-read only what is still missing AND still matters (lanes behind the nearest prefix never do): the polls
            // of ~1000 waiting tiles all hit the same few cache lines, i.e. one memory channel
            if (desc_status(g) == kStInvalid && ((need >> lane) & 1ull)) g = ld_desc(desc + my);
            const uint64_t pmask = __ballot(desc_status(g) == kStPrefix);
            const uint64_t imask = __ballot(desc_status(g) == kStInvalid);
            first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
            need = first_p >= 63u ? ~0ull : ((2ull << first_p) - 1ull);
            if ((imask & need) == 0) break;
            if (guard.expired()) return false;
            __builtin_amdgcn_s_sleep(2);
        }
        acc += wave_sum64((uint32_t)lane <= first_p ? desc_value(g) : 0ull);
        if (first_p < 64u) break;
        pos -= 64;
    }
    *result = acc;
    return true;
}

// Frame chain: byte offset of frame `frame` in the stack = sum of S_f' = 1 + bits(f')/8 (Terse.hpp:547) over the
// frames before it.  bits(f') is accumulated by the frame's tiles themselves -- each adds {1 << 40 | its bits} to
// acc[f'] with a fire-and-forget atomic as soon as it knows its size -- so a reader is ONE memory hop behind the
// slowest contributing tile (a chain of published per-frame sizes would be two).  pref[f'] = {1 << 63 | inclusive
// bytes} is the shortcut the frame's last tile leaves behind once it knows its own base.
// The accumulators live on cache lines of their own -- one per frame, up to kAccLines per frame of many tiles (tile t adds to line
// t % n_acc; the reader sums them: count and bits are both sums).  Same-line atomics are served one at a time (~10 ns): with the
// accumulators of sixteen frames in one line, as they were until round 6, the 2048 tiles in flight of a stack of 2048 x 2048
// frames (six frames: ONE line) queued for 20 us, every frame's base waited for the last of them, and the encoder ran at 0.29 of
// the HBM peak on such a stack against 0.60 on 512 x 512 frames (200 x (1030 x 1065): 0.187 -> 0.125 ms, 128 x 2048^2: 0.557 -> 0.343).
constexpr int kAccShift = 40;                                            // bits < 2^40, tiles per frame < 2^24
#ifndef TRPX_FUSED_ACC_LINES
#define TRPX_FUSED_ACC_LINES 2
#endif
constexpr uint32_t kAccLines = TRPX_FUSED_ACC_LINES;
__host__ __device__ inline uint32_t fused_acc_lines(uint32_t tiles_per_frame) {                    // ~32 tiles and more per line
    const uint32_t k = (tiles_per_frame + 31u) / 32u;
    return k < 1u ? 1u : (k > kAccLines ? kAccLines : k);
}
constexpr uint64_t kPrefFlag = 1ull << 63;
__device__ bool lookback_frames(const uint64_t* __restrict__ acc_w, const uint64_t* __restrict__ pref_w, int64_t frame,
                                uint64_t tiles_per_frame, uint64_t* result, uint64_t stride, uint32_t n_acc) {
    const int lane = lane_id();
    uint64_t acc = 0;
    int64_t pos = frame - 1;
    while (pos >= 0) {
        const int64_t my = pos - lane;
        const bool in = my >= 0;
        uint64_t p = in ? 0ull : kPrefFlag, c = 0;                       // below frame 0: prefix 0 ends the chain
        uint32_t first_p = 64;
        SpinGuard guard;
        uint64_t need = ~0ull;
        bool complete = false;
        for (;;) {
            // re-read only what is still missing and still matters (see lookback)
            if (in && !(p & kPrefFlag) && !complete && ((need >> lane) & 1ull)) {
                p = ld_desc(pref_w + my * stride);
                c = ld_desc(acc_w + my * stride);
                for (uint32_t k = 1; k < n_acc; ++k) c += ld_desc(acc_w + my * stride + 16u * k);   // (count and bits are both sums)
                complete = (c >> kAccShift) == tiles_per_frame;
            }
            const uint64_t pmask = __ballot((p & kPrefFlag) != 0);
            const uint64_t cmask = __ballot(complete);
            first_p = pmask ? (uint32_t)__builtin_ctzll(pmask) : 64u;
            need = first_p >= 64u ? ~0ull : ((1ull << first_p) - 1ull);  // the lanes in front of the prefix
            if ((~cmask & need) == 0) break;
            if (guard.expired()) return false;
            __builtin_amdgcn_s_sleep(2);
        }
        const uint64_t mine = (uint32_t)lane < first_p ? 1 + (c & ((1ull << kAccShift) - 1)) / 8
                                                       : ((uint32_t)lane == first_p ? p & ~kPrefFlag : 0ull);
        acc += wave_sum64(mine);
        if (first_p < 64u) break;
        pos -= 64;
    }
    *result = acc;
    return true;
}

// ---------------------------------------------------------------------------------------------
// A codec block (12 values) lives in registers as the raw little-endian dwords it was loaded as:
// 3 dwords for 8-bit, 6 for 16-bit, 12 for 32-bit pixels.  Field k is a static bit-field extract.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Raw {
    static constexpr int bits = PixelTraits<T>::bits;
    static constexpr int per = 32 / bits;                 // values per dword
    static constexpr int dw = kBlock / per;               // dwords per block
};

template <typename T>
__device__ __forceinline__ uint32_t raw_field(const uint32_t (&raw)[Raw<T>::dw], int k) {   // zero-extended value k
    constexpr int per = Raw<T>::per, bits = Raw<T>::bits;
    const uint32_t x = raw[k / per] >> ((k % per) * bits);
    return bits == 32 ? x : x & ((1u << (bits & 31)) - 1u);
}

template <typename T> struct QuadVec;
template <> struct QuadVec<uint8_t>  { typedef uint32_t type; };
template <> struct QuadVec<int8_t>   { typedef uint32_t type; };
template <> struct QuadVec<uint16_t> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct QuadVec<int16_t>  { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct QuadVec<uint32_t> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct QuadVec<int32_t>  { typedef uint32_t type __attribute__((ext_vector_type(4))); };

// A full, vector-aligned block (3 x 4 values).  (A software prefetch of a future tile -- one dword per 128-byte line of the
// tile 768..6144 positions further on, issued after barrier #1 -- was measured: 0.32-0.39 ms instead of 0.31.)
template <typename T>
__device__ __forceinline__ void load_raw_nt(const T* __restrict__ p, uint32_t (&raw)[Raw<T>::dw]) {
    // 16-bit pixels: 24 bytes as 16 + 8, 8-bit pixels: 12 bytes in one load (dword alignment is all a global load needs; three
    // 8-byte / 4-byte loads: 0.268 instead of 0.266 ms per 2000-frame stack).  Plain loads: non-temporal ones cost this kernel 12 %
    // (0.30 ms) -- they do leave the freshly written stream in the caches for a decode that follows (0.23-0.24 instead of 0.25-0.27
    // ms), which does not pay the difference back.
    if constexpr (sizeof(T) == 2) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u4 lo; u2 hi;
        __builtin_memcpy(&lo, p, 16);
        __builtin_memcpy(&hi, reinterpret_cast<const char*>(p) + 16, 8);
        raw[0] = lo.x; raw[1] = lo.y; raw[2] = lo.z; raw[3] = lo.w; raw[4] = hi.x; raw[5] = hi.y;
        return;
    } else if constexpr (sizeof(T) == 1) {
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
        u3 x;
        __builtin_memcpy(&x, p, 12);
        raw[0] = x.x; raw[1] = x.y; raw[2] = x.z;
        return;
    }
    using V = typename QuadVec<T>::type;
    constexpr int q = Raw<T>::dw / 3;
    const V* src = reinterpret_cast<const V*>(p);
    union U { V vec; uint32_t x[q]; };
    U a, b, c;
    a.vec = src[0];
    b.vec = src[1];
    c.vec = src[2];
#pragma unroll
    for (int i = 0; i < q; ++i) { raw[i] = a.x[i]; raw[q + i] = b.x[i]; raw[2 * q + i] = c.x[i]; }
}
// A full block of 16-bit pixels at an address that is 2 mod 4: a 2-byte misaligned 16-byte load costs the encoder a third of its
// time on such frames (2000 x (513 x 511): 0.313 against 0.268 ms, every second frame).  The same 24 bytes as dword-aligned loads
// from the dword in front -- 16 + 8 bytes, and the block's last pixel on its own, so that nothing behind the block is read -- and
// six funnel shifts.
template <typename T>
__device__ __forceinline__ void load_raw_mis2(const T* __restrict__ p, uint32_t (&raw)[Raw<T>::dw]) {
    static_assert(sizeof(T) == 2, "16-bit pixels");
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const char* q = reinterpret_cast<const char*>(p) - 2;
    u4 lo; u2 hi; uint16_t last;
    __builtin_memcpy(&lo, q, 16);
    __builtin_memcpy(&hi, q + 16, 8);
    __builtin_memcpy(&last, q + 24, 2);
    raw[0] = __builtin_amdgcn_alignbit(lo.y, lo.x, 16); raw[1] = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
    raw[2] = __builtin_amdgcn_alignbit(lo.w, lo.z, 16); raw[3] = __builtin_amdgcn_alignbit(hi.x, lo.w, 16);
    raw[4] = __builtin_amdgcn_alignbit(hi.y, hi.x, 16); raw[5] = __builtin_amdgcn_alignbit((uint32_t)last, hi.y, 16);
}
// The frame's last (partial) block: element loads, missing values read as zero.
template <typename T>
__device__ __forceinline__ void load_raw_partial(const T* __restrict__ p, int nb, uint32_t (&raw)[Raw<T>::dw]) {
    using UT = typename PixelTraits<T>::U;
#pragma unroll
    for (int i = 0; i < Raw<T>::dw; ++i) raw[i] = 0;
#pragma unroll
    for (int k = 0; k < kBlock; ++k)
        if (k < nb) raw[k / Raw<T>::per] |= (uint32_t)(UT)p[k] << ((k % Raw<T>::per) * Raw<T>::bits);
}

// OR-scan + bit length (Terse.hpp:508-515, :551-560) on the raw dwords.
template <typename T>
__device__ __forceinline__ uint32_t raw_width(const uint32_t (&raw)[Raw<T>::dw]) {
    constexpr int bits = Raw<T>::bits;
    uint32_t m = 0;
    if (!PixelTraits<T>::is_signed) {
#pragma unroll
        for (int i = 0; i < Raw<T>::dw; ++i) m |= raw[i];
        if (bits == 16) m = (m | (m >> 16)) & 0xFFFFu;
        if (bits == 8) { m |= m >> 16; m = (m | (m >> 8)) & 0xFFu; }
    } else if (bits == 32) {
#pragma unroll
        for (int i = 0; i < Raw<T>::dw; ++i) {                           // |x| in unsigned arithmetic: -INT32_MIN overflows as an int,
            const uint32_t u = raw[i];                                  // and a compiler that may assume it does not drops the clamp
            m |= (int32_t)u < 0 ? 0u - u : u;                           // of width_from_or (33-bit widths for blocks that hold INT32_MIN)
        }
    } else if (bits == 16) {
        typedef short s2 __attribute__((ext_vector_type(2)));
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < Raw<T>::dw; ++i) {                          // packed |x| = max(x, -x); the negation in unsigned arithmetic
            union { uint32_t u; s2 v; us2 w; } x, n, y;                 // (-(-32768) wraps to -32768 by definition, not by luck)
            x.u = raw[i];
            n.w = (us2)(0) - x.w;
            y.v = __builtin_elementwise_max(x.v, n.v);
            acc |= y.u;
        }
        m = (acc | (acc >> 16)) & 0xFFFFu;                              // |−32768| reads 0x8000: bitlen 16 -> clamped below
    } else {
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const int32_t x = (int32_t)(int8_t)(raw[k / 4] >> ((k % 4) * 8));
            m |= (uint32_t)(x < 0 ? -x : x);
        }
    }
    return width_from_or<T>(m);
}

// ---------------------------------------------------------------------------------------------
// Width-specialised block serialiser.
// ---------------------------------------------------------------------------------------------
// OR an ND-dword little-endian bit string (exactly 12*W valid bits, rest zero) into the LDS image
// at bit position `pos`.  First / last touched dwords may be shared with other lanes.
// `stage` is the image shifted by one pad dword (image dword i = stage[i + 1]): with rs = -pos mod 32 the string
// lands in dwords ceil(pos/32)-1 .. +ND, each one funnel shift (v_alignbit) of two neighbouring pieces -- no 64-bit
// shifts, hence no register-pair constraints on the block's dwords.
// The block's header rides along: `hv_top` holds the header code in its TOP bits (the code's last bit in bit 31), i.e.
// it is the 32-bit piece in front of the payload string, so the header costs one more funnel shift instead of a
// 64-bit shift, an address and two atomics of its own.
template <int ND>
__device__ __forceinline__ void lds_or_string(uint32_t* __restrict__ stage, uint32_t pos, uint32_t hv_top,
                                              const uint32_t* __restrict__ p) {
    const uint32_t d = (pos + 31u) >> 5, rs = (0u - pos) & 31u;
    const uint32_t xm = __builtin_amdgcn_alignbit(hv_top, 0u, rs);              // header bits below the payload's first dword
    if (xm) atomicOr(&stage[d - 1], xm);
    if constexpr (ND == 0) {
        const uint32_t x0 = __builtin_amdgcn_alignbit(0u, hv_top, rs);
        if (x0) atomicOr(&stage[d], x0);
    } else {
        atomicOr(&stage[d], __builtin_amdgcn_alignbit(p[0], hv_top, rs));
#pragma unroll
        for (int j = 1; j < ND; ++j) {
            const uint32_t x = __builtin_amdgcn_alignbit(p[j], p[j - 1], rs);
            if (j >= ND - 1) atomicOr(&stage[d + j], x);
            else stage[d + j] = x;                          // interior dword: owned by this block alone
        }
        const uint32_t x = __builtin_amdgcn_alignbit(0u, p[ND - 1], rs);
        if (x) atomicOr(&stage[d + ND], x);
    }
}

template <typename T, int W>
__device__ __forceinline__ void pack_payload_w(uint32_t* __restrict__ stage, uint32_t pos, uint32_t hv_top,
                                               const uint32_t (&raw)[Raw<T>::dw]) {
    constexpr int NB = kBlock * W;
    constexpr int ND = (NB + 31) / 32;
    constexpr int per = Raw<T>::per, bits = Raw<T>::bits;
    constexpr uint32_t MASK = W >= 32 ? 0xFFFFFFFFu : ((1u << (W & 31)) - 1u);
    uint32_t p[ND ? ND : 1];
#pragma unroll
    for (int j = 0; j < ND; ++j) p[j] = 0;
    if constexpr (W == 0) {
    } else if constexpr (bits == 16 && W < 16) {
        // Two values per dword.  v_dot2_u32_u16 computes lo * c.lo + hi * c.hi + acc in one instruction: with the constants
        // {2^s, 2^(s+W)} it places BOTH values of a dword at bit s of a running piece, and chained through the accumulator it
        // strings L dwords together for as long as the constants fit 16 bits (s + W <= 15): W = 3 -> two chains of three
        // dwords, 6 instructions for the 12 values (shift + v_bfi + shift + or per pair before: 18).  Sums of disjoint bit
        // fields are ORs: every value of a block of width W is < 2^W (signed pixels are truncated to W bits first,
        // Bit_pointer.hpp:707-710).  The kernel is bound by vector instruction issue.
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        constexpr int L = (15 - W) / (2 * W) + 1;                               // dwords per chain
        constexpr uint32_t kPairMask = MASK | (MASK << 16);
#pragma unroll
        for (int c = 0; c * L < kBlock / 2; ++c) {
            uint32_t acc = 0;
            int n_pairs = 0;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const int j = c * L + i;
                if (j < kBlock / 2) {
                    union { uint32_t u; us2 v; } x, k;
                    x.u = PixelTraits<T>::is_signed ? raw[j] & kPairMask : raw[j];
                    k.u = (1u << (2 * W * i)) | (1u << (2 * W * i + W + 16));
                    acc = __builtin_amdgcn_udot2(x.v, k.v, acc, false);
                    ++n_pairs;
                }
            }
            const int bit = 2 * W * L * c, nbits = 2 * W * n_pairs;
            p[bit >> 5] |= acc << (bit & 31);
            if ((bit & 31) + nbits > 32) p[(bit >> 5) + 1] |= acc >> (32 - (bit & 31));
        }
    } else if constexpr (bits == 16) {                                          // W = 16: the dwords as they are
#pragma unroll
        for (int j = 0; j < kBlock / 2; ++j) p[j] = raw[j];
    } else {
#pragma unroll
        for (int k = 0; k < kBlock; ++k) {
            const uint32_t u = (raw[k / per] >> ((k % per) * bits)) & MASK;   // value mod 2^W (Bit_pointer.hpp:707-710)
            const int bit = k * W;
            p[bit >> 5] |= u << (bit & 31);
            if ((bit & 31) + W > 32) p[(bit >> 5) + 1] |= u >> (32 - (bit & 31));
        }
    }
    lds_or_string<ND>(stage, pos, hv_top, p);
}

// Binary dispatch on a wave-uniform width (scalar compares only, one specialised body executes).
template <typename T, int LO, int HI>
struct PackDispatch {
    static __device__ __forceinline__ void run(uint32_t* stage, uint32_t pos, uint32_t hv_top, uint32_t w0,
                                               const uint32_t (&raw)[Raw<T>::dw]) {
        if constexpr (LO == HI) pack_payload_w<T, LO>(stage, pos, hv_top, raw);
        else {
            constexpr int MID = (LO + HI) / 2;
            if (w0 <= (uint32_t)MID) PackDispatch<T, LO, MID>::run(stage, pos, hv_top, w0, raw);
            else PackDispatch<T, MID + 1, HI>::run(stage, pos, hv_top, w0, raw);
        }
    }
};

// Generic (runtime width, partial block) serialiser: only the last block of a frame uses it.
template <typename T>
__device__ __forceinline__ void pack_payload_generic(uint32_t* __restrict__ stage, uint32_t pos, uint32_t w, int nb,
                                                     const uint32_t (&raw)[Raw<T>::dw]) {
    const uint32_t mask = w >= 32u ? 0xFFFFFFFFu : ((1u << w) - 1u);
#pragma unroll
    for (int k = 0; k < kBlock; ++k) {
        if (k < nb) {
            const uint64_t x = (uint64_t)(raw_field<T>(raw, k) & mask) << (pos & 31u);
            if ((uint32_t)x) atomicOr(&stage[pos >> 5], (uint32_t)x);
            if ((uint32_t)(x >> 32)) atomicOr(&stage[(pos >> 5) + 1], (uint32_t)(x >> 32));
            pos += w;
        }
    }
}

// The LDS image holds fused_phase_rounds() rounds of worst-case blocks (+ a carried dword); larger tiles take two phases.
template <typename T> constexpr int fused_phase_rounds() { return (sub_tiles<T>() + 1) / 2; }
template <typename T>
constexpr int fused_stage_dwords() { return fused_phase_rounds<T>() * ((kThreads * max_block_bits<T>() + 31) / 32) + 12; }

struct FusedArgs {
    FrameGeom g;
    uint32_t n_frames;
    uint32_t tiles_per_frame;      // ceil(n_blocks / (sub_tiles * 256))
    uint64_t out_capacity;
    uint64_t* tile_desc;           // [F * tpf]  AGG: tile bits, PREFIX: inclusive bits inside the frame
    uint64_t* tail_desc;           // [F * tpf]  exchange word of the boundary between tile-1 and tile (see the kernel's end)
    uint64_t* bnd_pos;             // [F * tpf]  output dword index of the boundary's shared dword (head side writes it)
    uint64_t* frame_acc;           // [F * acc_stride]  word 0 of each of the frame's n_acc lines: tiles contributed << 40 | bits so far (atomic adds)
    uint64_t* frame_pref;          // = frame_acc + 1: word 1 of the frame's first line: 1 << 63 | inclusive bytes through this frame, once known
    uint64_t* frame_base;          // [F * 16]   1 << 63 | first byte of the frame (one 128-byte line per frame): tile 0 -> the frame's other tiles
    uint32_t acc_stride;           // words between two frames' accumulator lines
    uint32_t n_acc;                // accumulator lines per frame: tile t adds to line t % n_acc (word 0; line 0's word 1 is the frame's prefix)
    uint64_t* ws_tag;              // [1]        what the call before this one left behind (see launch_fused_t); checked against expect_tag if that is not 0
    uint64_t expect_tag;
    uint64_t* frame_offsets;       // [F + 1]    output
    uint32_t* out32;
    uint32_t* status;
    uint8_t* idx_widths;           // optional decode index (see include/trpx_hip.h): width of every block
    uint64_t* idx_group_off;       //   and frame-relative bit offset of every 256-block group
    uint32_t debug;                // TRPX_DIAGNOSTICS builds only (see TRPX_DIAG)
    uint64_t* stamps;              // [tiles][8]
};

// Arguments that are only needed behind the packing (the tile's place in the stack, the flush) are fetched from the
// kernel-argument segment where they are used instead of living in SGPRs from the kernel's first instruction on: `kb` is
// the segment's base made opaque at that point (a scalar load of a few dwords per tile against ~20 SGPRs less to spill
// around the specialised packing bodies).  Layout: `pixels` at byte 0, the FusedArgs struct at byte 8.
typedef const char __attribute__((address_space(4)))* karg_ptr;
#define TRPX_KARG(kb, field) (*reinterpret_cast<const decltype(FusedArgs::field) __attribute__((address_space(4)))*>((kb) + 8 + offsetof(FusedArgs, field)))

// MIS2 (16-bit pixels): frames may start at an address that is 2 mod 4 -- every second frame of a stack whose frames hold an odd
// number of pixels (513 x 511, 1030 x 1065 ...) -- and such a frame's blocks are loaded from the dword in front (load_raw_mis2).
template <typename T, bool MIS2>
__device__ __forceinline__ void encode_fused_body(const T* __restrict__ pixels, const FusedArgs& a) {
    constexpr int kSub = sub_tiles<T>();
    constexpr int kFusedTileBlocks = kSub * kThreads;
    constexpr int kStage = fused_stage_dwords<T>();
    constexpr int kStage4 = (kStage + 1 + 3) / 4;  // in 16-byte units
    __shared__ __attribute__((aligned(16))) uint32_t s_stage_pad[4 * kStage4];   // [0] = pad dword, always zero (lds_or_string, flush)
    uint32_t* const s_stage = s_stage_pad + 1;    // the tile-relative bit image
    __shared__ uint32_t s_hdr[40];          // explicit header code of width w, top aligned (see lds_or_string)
    __shared__ uint32_t s_wtot[kSub * 4];   // bits of each (round, wave) piece without its lane-0 header
    __shared__ uint32_t s_wfl[kSub * 4];    // first width | last width << 8 | lane 0 valid << 16
    __shared__ uint64_t s_excl_bits;       // bits of this frame before the tile
    __shared__ uint64_t s_base_bytes;      // first byte of this frame in the stack
    __shared__ uint32_t s_abort;
    __shared__ uint32_t s_carry;           // two-phase tiles: phase 0's bits of the output dword the phases share
    __shared__ uint32_t s_halo;            // width of the block in front of the tile (wave 0 computes it, every wave needs it)

    const uint32_t tid = threadIdx.x;
#define TRPX_STAMP(slot) do { if ((TRPX_DIAG(a) & 4u) && threadIdx.x == 0) a.stamps[((uint64_t)(blockIdx.z * kGridY + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    TRPX_STAMP(0);
    const int lane = lane_id(), wave = wave_id();
    const FrameGeom g = a.g;
    // grid = (tiles per frame, frames in chunks of kGridY): x runs fastest, so the linear dispatch order is tile order
    const uint32_t t = blockIdx.x;
    const uint32_t frame = blockIdx.z * kGridY + blockIdx.y;
    const uint32_t n_frames = a.n_frames, tpf = a.tiles_per_frame;
    asm volatile("" :: "s"(g.n_values), "s"(g.n_blocks), "s"(n_frames), "s"(tpf), "s"(pixels));   // one kernarg fetch, one wait
    if (frame >= n_frames) return;
    const uint64_t tile = (uint64_t)frame * tpf + t;
    const bool last_tile_of_frame = t + 1 == tpf;
    if (tile == 0 && tid == 0) {
        // The stack's first tile clears the status block -- it is dispatched first, and nothing is reported there before a tile
        // has looked back over the chain that starts with THIS tile's descriptor, which is published behind these stores'
        // acknowledgement (errors: CAPACITY by the last tile, TIMEOUT after 0.25 s, the tag below by this thread).  The
        // descriptor words themselves were left cleared by the k_stitch of the call before (launch_fused_t); the tag tells
        // whether that is still what the workspace holds.
        karg_ptr kb0 = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        uint64_t* const st64 = reinterpret_cast<uint64_t*>(TRPX_KARG(kb0, status));
#pragma unroll
        for (int i = 0; i < 4; ++i) st_desc(st64 + i, 0ull);
        const uint64_t want = TRPX_KARG(kb0, expect_tag);
        if (want != 0ull && ld_desc(TRPX_KARG(kb0, ws_tag)) != want) atomicMax(&TRPX_KARG(kb0, status)[0], 7u);   // somebody else wrote into the workspace
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const T* fp = pixels + (uint64_t)frame * g.n_values;
    const uint32_t b0 = t * kFusedTileBlocks;

    // ---- load, widths, header / payload lengths ------------------------------------------------
    // Every round's loads are issued back to back and unconditionally (no branch, no register shuffle between
    // them, so the compiler has no reason to wait for round r before it issues round r+1 -- the per-round
    // `s_waitcnt vmcnt` of the branchy version made this phase latency bound).  A lane whose block does not exist or
    // is the frame's partial last block reads the frame's last full block instead; its data are never used.
    uint32_t v[kSub][Raw<T>::dw];
    uint32_t w[kSub], up[kSub], len[kSub], inc[kSub];
    int nb[kSub];
    const bool has_full = g.n_values >= (uint64_t)kBlock;
    const uint64_t max_first = has_full ? g.n_values - kBlock : 0;      // first value of the frame's last full block (any pixel count: loads are T-aligned)
    uint32_t h[Raw<T>::dw];              // block b0-1 (never the frame's last: full), for the tile's first header: wave 0 only
    // (MIS2: the frame's first pixel lies 2 bytes behind a dword boundary, and not at the stack's first byte -- the two bytes in
    // front of it are the frame's before it)
    [[maybe_unused]] const bool mis2 = MIS2 && ((uintptr_t)fp & 2u) != 0u && frame != 0u;
    if (has_full) {
        bool loaded = false;
        if constexpr (MIS2 && sizeof(T) == 2) {
            if (mis2) {
#pragma unroll
                for (int r = 0; r < kSub; ++r) {
                    const uint64_t first = (uint64_t)(b0 + r * kThreads + tid) * kBlock;
                    load_raw_mis2<T>(fp + (first < max_first ? first : max_first), v[r]);
                }
                loaded = true;
            }
        }
        if (!loaded) {
#pragma unroll
            for (int r = 0; r < kSub; ++r) {
                const uint64_t first = (uint64_t)(b0 + r * kThreads + tid) * kBlock;
                load_raw_nt<T>(fp + (first < max_first ? first : max_first), v[r]);
            }
        }
        if (wave == 0) load_raw_nt<T>(fp + (uint64_t)(b0 ? b0 - 1 : 0) * kBlock, h);
        else {
#pragma unroll
            for (int i = 0; i < Raw<T>::dw; ++i) h[i] = 0u;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kSub; ++r)
#pragma unroll
            for (int i = 0; i < Raw<T>::dw; ++i) v[r][i] = 0u;
#pragma unroll
        for (int i = 0; i < Raw<T>::dw; ++i) h[i] = 0u;
    }
    const uint32_t nb_tail = (uint32_t)(g.n_values - (uint64_t)(g.n_blocks - 1) * kBlock);   // values of the frame's last block (scalar)
#pragma unroll
    for (int r = 0; r < kSub; ++r) {
        const uint32_t b1 = b0 + r * kThreads + tid + 1u;
        nb[r] = b1 < g.n_blocks ? kBlock : (b1 == g.n_blocks ? (int)nb_tail : 0);
    }
    if (tid == 0) s_abort = 0;
    if (tid <= 32u) s_hdr[tid] = header_val(tid, tid + 1u) << (32u - header_len(tid, tid + 1u));
    {                                                                    // zero the image: unrolled ds_write_b128
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4* z = reinterpret_cast<u4*>(s_stage_pad);
#pragma unroll
        for (int i = 0; i < (kStage4 + kThreads - 1) / kThreads; ++i)
            if (i * kThreads + (int)tid < kStage4) z[i * kThreads + tid] = (u4)(0u);
    }
    // The frame's partial last block (at most one lane of the frame's last tile) lives in its own registers, so that
    // v[][] stay exactly the registers the vector loads wrote (a merge would put a copy -- and a wait -- behind each load).
    uint32_t pv[Raw<T>::dw];
#pragma unroll
    for (int i = 0; i < Raw<T>::dw; ++i) pv[i] = 0u;
    if (last_tile_of_frame && g.n_values % kBlock) {
        const uint32_t bp = g.n_blocks - 1;                             // element loads, missing values read as zero
        if (bp >= b0 && (bp - b0) % kThreads == tid) load_raw_partial<T>(fp + (uint64_t)bp * kBlock, (int)(g.n_values % kBlock), pv);
    }
    TRPX_STAMP(7);                                                       // loads issued
    uint32_t w_part = 0u;
    if (last_tile_of_frame && g.n_values % kBlock) w_part = raw_width<T>(pv);             // (workgroup-uniform branch)
    // width of the block before the tile's first block (w_{-1} = 0 at frame start, Terse.hpp:505): wave 0 computes it,
    // every wave reads it behind barrier #1 (each redoes the piece scan itself)
    if (wave == 0) {
        const uint32_t hw0 = b0 > 0 ? raw_width<T>(h) : 0u;
        if (lane == 0) s_halo = hw0;
    }
    // Lanes 1..63 get the previous block's width from their neighbour; lane 0's header depends on the
    // previous wavefront's last width, so it is left out of the scan here and added after barrier #1.
    uint32_t wmax = 0;
    uint32_t hlr[kSub];                  // header length of lanes 1..63 (lane 0: fixed up after barrier #1)
#pragma unroll
    for (int r = 0; r < kSub; ++r) {
        w[r] = nb[r] == kBlock ? raw_width<T>(v[r]) : (nb[r] ? w_part : 0u);
        wmax = w[r] > wmax ? w[r] : wmax;
        if (a.idx_widths && nb[r]) a.idx_widths[(uint64_t)frame * g.n_blocks + b0 + r * kThreads + tid] = (uint8_t)w[r];
        up[r] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[r], 0x138, 0xf, 0xf, false);   // wave_shr:1
        hlr[r] = header_len(w[r], up[r]);
        len[r] = nb[r] ? (lane ? hlr[r] : 0u) + (uint32_t)nb[r] * w[r] : 0u;
    }
    // wave scans, two rounds per 32-bit DPP scan: a wave's 64 blocks are < 2^16 bits (64 * 396 = 25 344)
    static_assert(kWave * max_block_bits<T>() < 65536, "packed scan needs 16-bit wave totals");
#pragma unroll
    for (int r = 0; r + 1 < kSub; r += 2) {
        const uint32_t two = wave_inclusive_scan(len[r] | (len[r + 1] << 16));
        inc[r] = two & 0xFFFFu;
        inc[r + 1] = two >> 16;
    }
    if (kSub & 1) inc[kSub - 1] = wave_inclusive_scan(len[kSub - 1]);
    const uint32_t wmax_wave = wave_max(wmax);                           // widest block of this wave's 6 x 64
#pragma unroll
    for (int r = 0; r < kSub; ++r) {
        const uint32_t w_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[r]);
        const uint32_t v_first = (uint32_t)__builtin_amdgcn_readfirstlane(nb[r]);
        if (lane == 63) {
            s_wtot[r * 4 + wave] = inc[r];
            s_wfl[r * 4 + wave] = w_first | (w[r] << 8) | ((v_first ? 1u : 0u) << 16) | (r == 0 ? wmax_wave << 24 : 0u);
        }
    }
    TRPX_STAMP(5);                                                       // wave 0: data arrived, widths and scans done
    __syncthreads();                                                     // #1: wave totals + staging zeroed
    TRPX_STAMP(1);

    // Every wave, lanes 0..15: lane-0 header of each of the 16 (round, wave) pieces -- it depends on the
    // previous piece's last width -- then the exclusive scan of the piece sizes (tile-relative bit offsets).
    const uint32_t tile_halo = s_halo;
    uint32_t rb[kSub + 1];               // tile-relative bit where round r starts; rb[kSub] = tile bits
    uint32_t off[kSub];                  // tile-relative bit position of this lane's block
    // what the packing needs of a block besides its pixels, in ONE register per round (an eighth workgroup per CU is a matter
    // of ~10 VGPRs): width | width of the block before << 8 | header length << 16 | full block << 24 | block exists << 25
    uint32_t meta[kSub];
    {
        uint32_t tot = 0, h0 = 0, wprev = 0;
        if (lane < kSub * 4) {
            const uint32_t wfl = s_wfl[lane];
            wprev = lane == 0 ? tile_halo : (s_wfl[lane - 1] >> 8) & 0xFFu;
            h0 = (wfl >> 16) & 1u ? header_len(wfl & 0xFFu, wprev) : 0u;
            tot = s_wtot[lane] + h0;
        }
        const uint32_t incl = wave_inclusive_scan(tot);
        const uint32_t excl = incl - tot;
        const uint32_t hw = h0 | (wprev << 8);
        const int swave = __builtin_amdgcn_readfirstlane(wave);          // wave index as a scalar: v_readlane, no LDS shuffle
#pragma unroll
        for (int r = 0; r < kSub; ++r) {
            rb[r] = (uint32_t)__builtin_amdgcn_readlane((int)excl, r * 4);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane((int)excl, r * 4 + swave);
            const uint32_t ph = (uint32_t)__builtin_amdgcn_readlane((int)hw, r * 4 + swave);
            off[r] = pb + (lane ? (ph & 0xFFu) : 0u) + inc[r] - len[r];
            const uint32_t wp_r = lane ? up[r] : ph >> 8;
            const uint32_t hl_r = lane ? hlr[r] : (ph & 0xFFu);          // lane 0's header length from the fix-up
            meta[r] = w[r] | (wp_r << 8) | (hl_r << 16) | (nb[r] == kBlock ? 1u << 24 : 0u) | (nb[r] ? 1u << 25 : 0u);
        }
        rb[kSub] = (uint32_t)__builtin_amdgcn_readlane((int)incl, kSub * 4 - 1);
    }
    const uint32_t tile_total = rb[kSub];

    // publish this tile's bit count at once (decoupled look-back: nobody waits for our look-back)
    if (tid == 0 && !(TRPX_DIAG(a) & 1u)) {
        st_desc(a.tile_desc + tile, make_desc(t == 0 ? kStPrefix : kStAgg, tile_total));
        __hip_atomic_fetch_add(a.frame_acc + (uint64_t)frame * a.acc_stride + 16u * (t % a.n_acc), (1ull << kAccShift) | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (Reading the chains' first windows before the packing, to hide their round trip, was measured: 0.326 -> 0.396 ms --
    // early reads mostly miss and every extra request on the chains' hot lines delays the stores that publish them.)
    // d_prolix_bits (Terse.hpp:516): the tile's widest block travels in bnd_pos[tile]'s top bits and k_stitch reduces
    // them -- an atomicMax per wave on one status word meant ~4000 same-address atomics (~11 ns each, serialised) from
    // the first wave of tiles, and every wave's next s_waitcnt vmcnt sat behind its own: a ~25 us stall per launch.
    // ---- pack into the tile-relative LDS image, look back meanwhile, flush with a funnel shift ---------
    // The image holds kPhaseRounds rounds of worst-case blocks (half a worst-case tile: 13 KB for 16-bit pixels, which is
    // what lets a seventh workgroup onto the CU).  A tile that needs more -- more than ~100 bits per block on average -- is
    // packed and flushed in TWO phases (rounds [0, kPhaseRounds), then the rest): phase 1's image starts at the output
    // dword in which phase 0 ended, and the bits phase 0 left in that dword are carried into it.
    constexpr int kPhaseRounds = fused_phase_rounds<T>();
    constexpr uint32_t kCapBits = (uint32_t)kPhaseRounds * kThreads * max_block_bits<T>();
    const bool two_phase = tile_total > kCapBits;                         // (workgroup-uniform)
    uint64_t p0 = 0, p_end = 0;                                          // absolute bit range of the tile (known behind barrier #2)
    uint64_t img_p0 = 0;                                                 // absolute bit of image bit 0 in the current phase
    uint32_t bias = 0;                                                   // tile-relative bit of image bit 0
    bool writable = false;
#ifndef TRPX_ABLATE
#define TRPX_ABLATE 0
#endif
#define TRPX_WSTAMP(slot) do { if ((TRPX_DIAG(a) & 8u) && lane == 0 && tile < 4096) a.stamps[tile * 32 + wave * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // Full blocks: header + payload as ONE bit string, one pass per distinct width present in the wavefront (w = 0:
    // the header alone), each with static shifts.  The frame's partial last block goes the generic way.
    auto pack_round = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t pos = off[r] - bias;
        const uint32_t mr = meta[r];
        const uint32_t hl = (mr >> 16) & 0xFFu;                          // (lane 0: from the piece fix-up)
        const uint32_t wr = mr & 0xFFu;
        const uint32_t hv_top = wr == ((mr >> 8) & 0xFFu) ? 0x80000000u : s_hdr[wr];   // header code, top aligned (Terse.hpp:517-535)
        const bool full = (mr >> 24) & 1u;
        if (((mr >> 25) & 1u) && !full && !(TRPX_ABLATE & 2)) {         // the frame's partial last block
            const uint64_t hx = (uint64_t)(hv_top >> (32u - hl)) << (pos & 31u);
            atomicOr(&s_stage[pos >> 5], (uint32_t)hx);
            if ((uint32_t)(hx >> 32)) atomicOr(&s_stage[(pos >> 5) + 1], (uint32_t)(hx >> 32));
            if (wr) pack_payload_generic<T>(s_stage, pos + hl, wr, (int)nb_tail, pv);
        }
        uint64_t todo = (TRPX_ABLATE & 1) ? 0ull : __ballot(full);
        while (todo) {
            const int l0 = __builtin_ctzll(todo);
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)wr, l0);
            const bool mine = full && wr == w0;
            uint32_t wd = w0;
            asm volatile("" : "+s"(wd));          // a copy the compiler cannot equate with the lanes' own width: with w0 itself it dispatched on
                                                  // the VECTOR (14 v_cmp per round and exec-mask branches per tree level instead of scalar compares)
#pragma unroll
            for (int i = 0; i < Raw<T>::dw; ++i) asm volatile("" : "+v"(v[r][i]));       // keep the bodies out of LICM's reach
            if (mine) PackDispatch<T, 0, PixelTraits<T>::bits>::run(s_stage_pad, pos + hl, hv_top, wd, v[r]);
            todo &= ~__ballot(mine);
        }
        TRPX_WSTAMP(1 + r);
    };
    // cross-tile prefixes (decoupled look-back), after this wave's share of the packing
    auto look_back = [&]() {
        if (wave == 0) {
            uint64_t excl = 0;
            bool ok = true;
            if (TRPX_DIAG(a) & 1u) excl = (uint64_t)t * 40000u;
            else if (t != 0) {
                ok = lookback(a.tile_desc, (int64_t)tile, (int64_t)(tile - t), &excl);
                if (ok && lane == 0) st_desc(a.tile_desc + tile, make_desc(kStPrefix, excl + tile_total));
            }
            if (lane == 0) { s_excl_bits = excl; if (!ok) s_abort = 1; }
            TRPX_STAMP(2);
        } else if (wave == 1) {
            uint64_t base = 0;
            bool ok = true;
            if (TRPX_DIAG(a) & 1u) base = (uint64_t)frame * 120000u;
            else if (t == 0) {
                // Only the frame's FIRST tile walks the stack-wide frame chain and hands the result to its siblings
                // through the frame's own cache line: ~70 pollers on the chain's hot lines instead of ~1000 (requests
                // to one line are served one by one, ~10 ns each, and the publishing stores queue behind them).
                ok = lookback_frames(a.frame_acc, a.frame_pref, (int64_t)frame, a.tiles_per_frame, &base, a.acc_stride, a.n_acc);
                if (ok && lane == 0) st_desc(a.frame_base + (uint64_t)frame * 16, kPrefFlag | base);
            } else {
                // (a second level -- the tiles of a frame of hundreds polling the line of their group of 8 / 32 / 128, filled by the
                // group's first tile from the frame's -- was measured in round 6: no change at 90, 342 or 1093 tiles per frame)
                uint64_t vb = 0;
                SpinGuard guard;
                for (;;) {
                    vb = ld_desc(a.frame_base + (uint64_t)frame * 16);   // same address in every lane: one request
                    if (vb & kPrefFlag) break;
                    if (guard.expired()) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                base = vb & ~kPrefFlag;
            }
            if (lane == 0) { s_base_bytes = base; if (!ok) s_abort = 1; }
        }
    };
    // Behind the barrier that follows the look-back: the tile's place in the stack.  Returns false if a wait gave up.
    auto place_tile = [&]() -> bool {
        karg_ptr kb = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        const bool aborted = s_abort != 0;
        const uint64_t excl_bits = s_excl_bits, base_bytes = s_base_bytes;
        const uint64_t frame_size = 1 + (excl_bits + tile_total) / 8;        // valid for the frame's last tile
        if (last_tile_of_frame && tid == 0 && !aborted) {
            st_desc(TRPX_KARG(kb, frame_pref) + (uint64_t)frame * TRPX_KARG(kb, acc_stride), kPrefFlag | (base_bytes + frame_size));
            TRPX_KARG(kb, frame_offsets)[frame + 1] = base_bytes + frame_size;
            if (frame == 0) TRPX_KARG(kb, frame_offsets)[0] = 0;
            if (frame + 1 == TRPX_KARG(kb, n_frames) && align_up(base_bytes + frame_size, 4) > TRPX_KARG(kb, out_capacity))
                atomicMax(&TRPX_KARG(kb, status)[0], 3u);                                 // TRPX_ERR_CAPACITY
        }
        if (aborted) {
            if (tid == 0) atomicMax(&TRPX_KARG(kb, status)[0], 7u);                       // look-back timeout
            return false;
        }
        if (TRPX_KARG(kb, idx_group_off) && tid < (uint32_t)kSub && b0 + tid * kThreads < g.n_blocks) {
            uint32_t rbv = 0;
#pragma unroll
            for (int r = 0; r < kSub; ++r) rbv = tid == (uint32_t)r ? rb[r] : rbv;
            TRPX_KARG(kb, idx_group_off)[(uint64_t)frame * g.n_tiles + (uint64_t)t * kSub + tid] = excl_bits + rbv;
        }
        p0 = 8 * base_bytes + excl_bits;                                     // absolute bit of the tile's first bit
        // bits this tile must materialise: its blocks, plus the frame's pad up to the byte S_f
        p_end = last_tile_of_frame ? 8 * (base_bytes + frame_size) : p0 + tile_total;
        writable = align_up((p_end + 7) / 8, 4) <= TRPX_KARG(kb, out_capacity);           // sizes-only query / too small: no stores
        img_p0 = p0;
        return true;
    };
    // The image's bits [img_p0, q_end) -> memory.  first: the tile's first phase (records the boundary bookkeeping);
    // carry_out: not the tile's last phase (the bits of the last, shared output dword stay in s_carry).
    auto flush = [&](bool first, bool carry_out, uint64_t q_end) {
        karg_ptr kb = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        uint32_t* const out32 = TRPX_KARG(kb, out32);
        uint64_t* const tail_desc = TRPX_KARG(kb, tail_desc);
        const uint32_t s0 = (uint32_t)(img_p0 & 31), sh = (32u - s0) & 31u;
        const bool head_pending = s0 != 0;                                   // first dword also holds the previous tile's bits
        const uint64_t d_first = img_p0 >> 5, d_last = q_end >> 5;
        // ---- tile boundary dwords ---------------------------------------------------------------------------
        // A dword that straddles two tiles (or frames) is written by nobody here: both sides OR their bits into the
        // boundary's exchange word {bit 63: head side there, bit 62: tail side there, low 32: the bits} with a
        // fire-and-forget atomic and the head side records the dword's index; k_stitch (a ~44 000-thread kernel
        // right behind this one) stores the merged dwords.  Nothing in this kernel waits for another tile's data.
        // (A dword that straddles the two PHASES of one tile is carried into the second phase's image instead.)
        const int32_t k_first = head_pending ? -1 : 0;
        const uint32_t n_out = (uint32_t)(d_last - d_first);
        const bool completed_first = d_last > d_first;                       // finished at least its first dword
        const bool is_last_tile = tile + 1 == (uint64_t)TRPX_KARG(kb, n_frames) * TRPX_KARG(kb, tiles_per_frame);
        if (tid == 0) {
            const int32_t kt = k_first + (int32_t)n_out;
            const uint32_t tail_bits = (q_end & 31) ? __builtin_amdgcn_alignbit(s_stage[kt + 1], s_stage[kt], sh) : 0u;
            if (head_pending) {                                              // our share of dword d_first (a tile that does not
                const uint32_t head_bits = __builtin_amdgcn_alignbit(s_stage[0], 0u, sh);   // complete it: all of its bits)
                __hip_atomic_fetch_or(tail_desc + tile, kHeadFlag | head_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (first) {
                uint32_t wm = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) wm = (s_wfl[i] >> 24) > wm ? (s_wfl[i] >> 24) : wm;
                TRPX_KARG(kb, bnd_pos)[tile] = (head_pending ? d_first | kBndHead : 0ull) | ((uint64_t)wm << kWmaxShift);
            }
            if (carry_out) s_carry = tail_bits;                              // this phase's share of dword d_last -> next image
            else if ((q_end & 31) != 0 && (completed_first || !head_pending)) {   // our share of dword d_last
                if (is_last_tile) { if (writable) __builtin_nontemporal_store(tail_bits, out32 + d_last); }
                else __hip_atomic_fetch_or(tail_desc + tile + 1, kTailFlag | tail_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- global dword d_first + j = {image[k+1], image[k]} >> sh, k = j - (s0 != 0) --------------------------
        // 16-byte groups: one thread turns two aligned ds_read_b128 into one 16-byte aligned non-temporal store (the
        // image-side misalignment m is tile-uniform: four specialised bodies behind scalar branches).  The <= 7 dwords in
        // front of / behind the aligned groups go one per thread.
        if (!(TRPX_ABLATE & 4) && writable) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            uint32_t al = (0u - ((uint32_t)((uintptr_t)out32 >> 2) + (uint32_t)d_first)) & 3u;
            if (head_pending && al == 0) al = 4;                              // dword 0 is k_stitch's: keep it out of the groups
            al = al < n_out ? al : n_out;
            const uint32_t n4 = (n_out - al) >> 2, rest0 = al + 4 * n4;
            if (tid < al + (n_out - rest0)) {
                const uint32_t j = tid < al ? tid : rest0 + (tid - al);
                if (!(j == 0 && head_pending)) {
                    const int32_t k = k_first + (int32_t)j;                   // k = -1: the pad dword
                    __builtin_nontemporal_store(__builtin_amdgcn_alignbit(s_stage[k + 1], s_stage[k], sh), out32 + d_first + j);
                }
            }
            const uint32_t kk0 = (uint32_t)(k_first + 1) + al;               // s_stage_pad index of the first group's low dword
            const uint32_t m = kk0 & 3u;
            const u4* pad4 = reinterpret_cast<const u4*>(s_stage_pad) + (kk0 >> 2);
            u4* dst = reinterpret_cast<u4*>(out32 + d_first + al);
#ifdef TRPX_ENC_PLAIN_STORES
#define TRPX_STREAM_STORE(v, p) (*(p) = (v))
#else
#define TRPX_STREAM_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#define TRPX_FLUSH_GROUPS(M)                                                                                   \
            for (uint32_t gq = tid; gq < n4; gq += kThreads) {                                                 \
                const u4 A = pad4[gq], B = pad4[gq + 1];                                                       \
                const uint32_t e[8] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w};                                \
                u4 x;                                                                                          \
                x.x = __builtin_amdgcn_alignbit(e[M + 1], e[M + 0], sh);                                       \
                x.y = __builtin_amdgcn_alignbit(e[M + 2], e[M + 1], sh);                                       \
                x.z = __builtin_amdgcn_alignbit(e[M + 3], e[M + 2], sh);                                       \
                x.w = __builtin_amdgcn_alignbit(e[M + 4], e[M + 3], sh);                                       \
                TRPX_STREAM_STORE(x, dst + gq);                                                                \
            }
            if (m == 0) { TRPX_FLUSH_GROUPS(0) } else if (m == 1) { TRPX_FLUSH_GROUPS(1) }
            else if (m == 2) { TRPX_FLUSH_GROUPS(2) } else { TRPX_FLUSH_GROUPS(3) }
#undef TRPX_FLUSH_GROUPS
        }
    };

    TRPX_WSTAMP(0);
    // rounds [0, kPhaseRounds)
    [&]<int... R>(std::integer_sequence<int, R...>) { (pack_round(std::integral_constant<int, R>{}), ...); }(std::make_integer_sequence<int, kPhaseRounds>{});
    if (two_phase) {
        // the image is full: the tile's place has to be known now, its first half leaves, the second half gets a fresh
        // image whose bit 0 is bit 0 of the output dword the halves share
        look_back();
        __syncthreads();
        if (!place_tile()) return;
        flush(true, true, p0 + rb[kPhaseRounds]);
        const uint64_t d_last = (p0 + rb[kPhaseRounds]) >> 5;
        __syncthreads();                                                     // every reader of the image is through
        {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4* z = reinterpret_cast<u4*>(s_stage_pad);
            for (int i = (int)tid; i < kStage4; i += kThreads) z[i] = (u4)(0u);
        }
        __syncthreads();
        if (tid == 0 && s_carry) atomicOr(&s_stage[0], s_carry);
        img_p0 = 32 * d_last;
        bias = (uint32_t)(img_p0 - p0);
    }
    // rounds [kPhaseRounds, kSub)
    [&]<int... R>(std::integer_sequence<int, R...>) { (pack_round(std::integral_constant<int, kPhaseRounds + R>{}), ...); }(std::make_integer_sequence<int, kSub - kPhaseRounds>{});
    if (!two_phase) look_back();
    __syncthreads();                                                         // #2: tile packed, prefixes known
    TRPX_STAMP(3);
    if (!two_phase && !place_tile()) return;
    flush(!two_phase, false, p_end);
    TRPX_STAMP(4);
    if ((TRPX_DIAG(a) & 4u) && tid == 0) { uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); a.stamps[tile * 8 + 6] = xcc; }
}

template <typename T>
__global__ __launch_bounds__(kThreads, fused_occupancy<T>()) void k_encode_fused(const T* __restrict__ pixels, FusedArgs a) {
    encode_fused_body<T, false>(pixels, a);
}
// The same for stacks of 16-bit pixels some of whose frames start 2 bytes behind a dword boundary (one workgroup less per CU: the
// realigned loads keep a dword more per block in flight).
template <typename T>
__global__ __launch_bounds__(kThreads, fused_occupancy<T>() - 1) void k_encode_fused_mis2(const T* __restrict__ pixels, FusedArgs a) {
    encode_fused_body<T, true>(pixels, a);
}

// Stores every dword that two (or, with tiny tiles, more) tiles share: the OR of what they deposited.  Also reduces the
// tiles' widest-block values into status[1] (d_prolix_bits, Terse.hpp:516): one atomic per workgroup.
// And it leaves the workspace as the NEXT call needs it: every descriptor word this call polled or OR-ed into is cleared
// again -- by the thread that read it last: a run of boundaries inside one dword is read and cleared by its first boundary's
// thread alone, which the others tell from bnd_pos (never from a neighbour's exchange word) -- and the tag is set.  A call
// that reported an error (a tile that gave up leaves no bnd_pos) clears everything and stitches nothing.
__global__ __launch_bounds__(kThreads) void k_stitch(uint64_t* __restrict__ tile_desc, uint64_t* __restrict__ xw, const uint64_t* __restrict__ pos,
                                                     uint64_t n_tiles, uint64_t* __restrict__ frame_words, uint64_t n_frame_words,
                                                     uint64_t* __restrict__ ws_tag, uint64_t tag, uint32_t* __restrict__ out32,
                                                     uint64_t out_capacity, uint32_t* __restrict__ status) {
    __shared__ uint32_t s_max[4];
    const uint64_t b = (uint64_t)blockIdx.x * kThreads + threadIdx.x;     // tile b; boundary b = between tile b-1 and tile b
    constexpr uint64_t kPosMask = kBndHead - 1;
    // (everything this thread may need is requested at once: the kernel is a handful of round trips, not work)
    const bool in = b < n_tiles;
    const uint32_t st0 = __hip_atomic_load(&status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t pw = in ? pos[b] : 0ull;
    const uint64_t pb = in && b > 0 ? pos[b - 1] : 0ull;
    const uint64_t xb = in ? xw[b] : 0ull;
    const bool failed = st0 != 0u;
    const uint32_t wm = wave_max(failed ? 0u : (uint32_t)(pw >> kWmaxShift));
    if (lane_id() == 0) s_max[wave_id()] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = s_max[0];
        for (int i = 1; i < 4; ++i) m = s_max[i] > m ? s_max[i] : m;
        if (m) atomicMax(&status[1], m);
    }
    for (uint64_t i = b; i < n_frame_words; i += (uint64_t)gridDim.x * kThreads) frame_words[i] = 0ull;
    if (b == 0 && ws_tag) ws_tag[0] = tag;                                // (captured calls leave the tag alone: see launch_fused_t)
    if (b >= n_tiles) return;
    tile_desc[b] = 0ull;
    if (failed || b == 0) { xw[b] = 0ull; return; }
    if (!(pw & kBndHead)) { xw[b] = 0ull; return; }                      // dword-aligned boundary: nothing shared (and nothing deposited)
    const uint64_t d = pw & kPosMask;
    if (b > 1 && (pb & kBndHead) && (pb & kPosMask) == d) return;         // middle of a run (tiny tiles): its first boundary stores and clears
    uint32_t acc = (uint32_t)xb;
    xw[b] = 0ull;
    for (uint64_t k = b + 1; k < n_tiles; ++k) {
        const uint64_t pk = pos[k];
        if (!((pk & kBndHead) && (pk & kPosMask) == d)) break;
        acc |= (uint32_t)xw[k];
        xw[k] = 0ull;
    }
    if (4 * d + 4 <= out_capacity) out32[d] = acc;
}

template <typename T>
static uint32_t fused_tiles_per_frame(const FrameGeom& g) {
    constexpr uint32_t tb = sub_tiles<T>() * kThreads;
    return (g.n_blocks + tb - 1) / tb;
}

size_t fused_workspace_bytes(const FrameGeom& g, size_t n_frames) {
    const size_t tpf = ((size_t)g.n_blocks + 2 * kThreads - 1) / (2 * kThreads);   // finest tiling (32-bit pixels)
    // descriptor words per tile; per frame the accumulator lines and the base line; the tag; the diagnostic stamps
    return align_up(8 * (3 * n_frames * tpf + 16 * ((size_t)kAccLines + 1) * n_frames + 1), 256) + 64 * n_frames * tpf;
}

// ---- workspaces the library knows to be clean ---------------------------------------------------------------------------------
// k_encode_fused polls and ORs into ~1 MB of descriptor words that have to be zero when it starts.  Clearing them took a launch
// of its own in front of every call (k_zero_words: 7 us of a 270 us call, launch latency, not work).  Now the call's LAST kernel,
// k_stitch, leaves every word it or the encoder used cleared again and writes a tag; the library remembers (device, address,
// geometry, tag) of the workspaces whose last user, as far as it knows, was such a call, and skips the clearing launch for
// them.  What it cannot know is what others did to the memory in between, hence the rules (include/trpx_hip.h): a workspace
// is the library's between calls -- whoever writes into it, frees it or hands the address to something else calls
// trpx_workspace_invalidate -- every entry point of this library that is given a workspace for another purpose forgets it by
// itself, a call that is being captured into a graph always clears (a replay does not pass through here), and the encoder's
// first tile compares the tag: a workspace that is not what the library remembers makes the call report TRPX_ERR_TIMEOUT -- the
// status the checked entry points answer with the two-pass pipeline, which uses none of these words.
namespace {
struct CleanSig {                                                         // the geometry a clean workspace was laid out for, field by field
    uint32_t elem, tiles_per_frame, n_frames, n_blocks;
    bool operator==(const CleanSig& o) const { return elem == o.elem && tiles_per_frame == o.tiles_per_frame && n_frames == o.n_frames && n_blocks == o.n_blocks; }
};
struct CleanWs { int device; const void* ptr; CleanSig sig; uint64_t tag; };
std::mutex g_clean_mu;
std::vector<CleanWs> g_clean;                                             // a handful of entries: linear search
uint64_t g_next_tag = 0x5452505800000001ull;
}  // namespace
void fused_ws_forget(const void* lo, size_t bytes, const void* keep) {
    std::lock_guard<std::mutex> lk(g_clean_mu);
    const char* a = static_cast<const char*>(lo);
    for (size_t i = 0; i < g_clean.size();) {
        const char* p = static_cast<const char*>(g_clean[i].ptr);
        if (p != keep && (!lo || (p >= a && p < a + bytes))) { g_clean[i] = g_clean.back(); g_clean.pop_back(); } else ++i;
    }
}

template <typename T>
static hipError_t launch_fused_t(const EncodeArgs& e, void* ws, hipStream_t st) {
    FusedArgs a;
    a.g = e.geom;
    a.n_frames = e.n_frames;
    a.tiles_per_frame = fused_tiles_per_frame<T>(e.geom);
    a.out_capacity = e.out_capacity;
    const size_t tiles = (size_t)e.n_frames * a.tiles_per_frame;
    a.tile_desc = static_cast<uint64_t*>(ws);
    a.tail_desc = a.tile_desc + tiles;
    a.bnd_pos = a.tail_desc + tiles;
    // per frame: n_acc accumulator lines (line 0's word 1: the prefix) and the base line -- 128 bytes each
    a.n_acc = fused_acc_lines(a.tiles_per_frame);
    a.acc_stride = 16u * a.n_acc;
    a.frame_acc = a.bnd_pos + tiles;
    a.frame_pref = a.frame_acc + 1;
    a.frame_base = a.frame_acc + (size_t)a.acc_stride * e.n_frames;
    a.frame_offsets = e.frame_offsets;
    a.out32 = reinterpret_cast<uint32_t*>(e.out);
    a.status = e.status;
    a.idx_widths = e.idx_widths;
    a.idx_group_off = e.idx_group_off;
    const size_t frame_words = 16 * ((size_t)a.n_acc + 1) * e.n_frames;
    uint64_t* const frame_words_base = a.frame_acc;
    a.ws_tag = frame_words_base + frame_words;
    a.stamps = reinterpret_cast<uint64_t*>(static_cast<char*>(ws) + align_up(8 * (3 * tiles + frame_words + 1), 256));
#ifdef TRPX_DIAGNOSTICS
    a.debug = getenv("TRPX_FUSED_DEBUG") ? (uint32_t)atoi(getenv("TRPX_FUSED_DEBUG")) : 0u;
#else
    a.debug = 0u;
#endif
    // Is the workspace known to be clean (see above)?  A call that is being CAPTURED never relies on it (a replay does not pass
    // through here): it always clears, and it leaves the registry and the TAG WORD alone -- its clearing launch stops in front
    // of the tag and its k_stitch does not write one -- so that replays and eager calls can alternate on one workspace: every
    // call, replayed or not, leaves the descriptor words clean, and the tag stays the last eager call's.  (A replay that wrote its
    // baked-in tag made the next eager call find "not the workspace I remember": a spurious TRPX_ERR_TIMEOUT.)
    // The current device is the one the kernels below run on, i.e. the one that has to own the workspace.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusActive; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    int device = -1;
    (void)hipGetDevice(&device);
    const CleanSig sig{(uint32_t)sizeof(T), a.tiles_per_frame, e.n_frames, e.geom.n_blocks};
    uint64_t expect = 0, tag = 0;
    if (!capturing) {
        std::lock_guard<std::mutex> lk(g_clean_mu);
        tag = g_next_tag++;
        for (size_t i = 0; i < g_clean.size(); ++i)
            if (g_clean[i].device == device && g_clean[i].ptr == ws) {
                if (g_clean[i].sig == sig) expect = g_clean[i].tag;
                g_clean[i] = g_clean.back(); g_clean.pop_back();            // (re-entered below if this call leaves it clean)
                break;
            }
    }
    a.expect_tag = expect;
    Profiler& prof = profiler();
    prof.begin();
    prof.mark(st);
    if (expect == 0)   // unknown workspace: every polled / OR-ed word, the tag and the status block (a kernel, not a memset node: see k_zero_words)
        hipLaunchKernelGGL(k_zero_words<0>, dim3(256), dim3(kThreads), 0, st, static_cast<uint64_t*>(ws),
                           (uint64_t)(3 * tiles + frame_words + (capturing ? 0 : 1)), reinterpret_cast<uint64_t*>(e.status), (uint64_t)4);
    prof.mark(st);
    const dim3 grid(a.tiles_per_frame, e.n_frames < kGridY ? e.n_frames : kGridY, (e.n_frames + kGridY - 1) / kGridY);
    bool mis2 = false;
    if constexpr (sizeof(T) == 2) mis2 = (((uintptr_t)e.pixels & 2u) != 0u || (e.geom.n_values & 1u) != 0u) && e.n_frames > 1u;
    if constexpr (sizeof(T) == 2) {
        if (mis2) hipLaunchKernelGGL((k_encode_fused_mis2<T>), grid, dim3(kThreads), 0, st, static_cast<const T*>(e.pixels), a);
    }
    if (!mis2) hipLaunchKernelGGL((k_encode_fused<T>), grid, dim3(kThreads), 0, st, static_cast<const T*>(e.pixels), a);
    prof.mark(st);
    hipLaunchKernelGGL(k_stitch, dim3((uint32_t)((tiles + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       a.tile_desc, a.tail_desc, static_cast<const uint64_t*>(a.bnd_pos), (uint64_t)tiles, frame_words_base, (uint64_t)frame_words,
                       capturing ? static_cast<uint64_t*>(nullptr) : a.ws_tag, tag, a.out32, (uint64_t)e.out_capacity, a.status);
    prof.mark(st);
    const hipError_t err = hipGetLastError();
    if (err == hipSuccess && !capturing) {
        std::lock_guard<std::mutex> lk(g_clean_mu);
        if (g_clean.size() >= 64) g_clean.erase(g_clean.begin());
        g_clean.push_back(CleanWs{device, ws, sig, tag});
    }
    return err;
}

hipError_t launch_encode_fused(int dtype, const EncodeArgs& e, void* ws, hipStream_t st) {
    switch (dtype) {
    case 0: return launch_fused_t<uint8_t>(e, ws, st);
    case 1: return launch_fused_t<int8_t>(e, ws, st);
    case 2: return launch_fused_t<uint16_t>(e, ws, st);
    case 3: return launch_fused_t<int16_t>(e, ws, st);
    case 4: return launch_fused_t<uint32_t>(e, ws, st);
    case 5: return launch_fused_t<int32_t>(e, ws, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace trpx
