// Width-specialised field extraction and block stores shared by the PROLIX decode kernels
// (decode_fast.hip: k_unpack_tiles, decode_frame.hip: k_decode_frames).  Replaces
// Bit_range::get_range / operator T() (reference include/Bit_pointer.hpp:742-792, :597-617).
#pragma once
#include "codec_common.hpp"
#include <type_traits>

namespace trpx {

// Vector types for accesses that are only aligned to the pixel type (frames of any pixel count start anywhere): the
// compiler is told the real alignment and still emits one dwordx2 / dwordx4 (gfx950: unaligned-access-mode,
// unaligned-ds-access).
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
typedef u4_t u4_a1 __attribute__((aligned(1)));
typedef u2_t u2_a1 __attribute__((aligned(1)));
typedef uint32_t u1_a1 __attribute__((aligned(1)));

template <typename T> struct OutVec;
template <> struct OutVec<uint8_t>  { typedef uint32_t type; };
template <> struct OutVec<int8_t>   { typedef uint32_t type; };
template <> struct OutVec<uint16_t> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct OutVec<int16_t>  { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct OutVec<uint32_t> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct OutVec<int32_t>  { typedef uint32_t type __attribute__((ext_vector_type(4))); };

// 12 decoded values (already sign/zero-extended to 32 bits) -> three vector stores of 4 values.
template <typename T>
__device__ __forceinline__ void store_block(T* __restrict__ dst, const uint32_t (&u)[kBlock]) {
    using V = typename OutVec<T>::type;
    constexpr int bits = PixelTraits<T>::bits;
    typedef V Vu __attribute__((aligned(1)));               // dst is aligned to T only
    Vu* q = reinterpret_cast<Vu*>(dst);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        union { V vec; uint32_t x[sizeof(V) / 4]; } o;
        if constexpr (bits == 32) { o.x[0] = u[4 * i]; o.x[1] = u[4 * i + 1]; o.x[2] = u[4 * i + 2]; o.x[3] = u[4 * i + 3]; }
        else if constexpr (bits == 16) {
            o.x[0] = (u[4 * i] & 0xFFFFu) | (u[4 * i + 1] << 16);
            o.x[1] = (u[4 * i + 2] & 0xFFFFu) | (u[4 * i + 3] << 16);
        } else {
            o.x[0] = (u[4 * i] & 0xFFu) | ((u[4 * i + 1] & 0xFFu) << 8) | ((u[4 * i + 2] & 0xFFu) << 16) | (u[4 * i + 3] << 24);
        }
        __builtin_nontemporal_store(o.vec, q + i);   // (plain stores remove the 1.23x partial-write traffic but run 17 % slower)
    }
}

// 12 decoded values -> the lane's LDS staging row (see unpack_stage_w / store_group).
template <typename T>
__device__ __forceinline__ void stage_block(uint32_t* __restrict__ row, const uint32_t (&u)[kBlock]) {
    constexpr int bits = PixelTraits<T>::bits;
    typedef u2_t u2;
    if constexpr (bits == 32) {
#pragma unroll
        for (int i = 0; i < 6; ++i) reinterpret_cast<u2*>(row)[i] = u2_t{u[2 * i], u[2 * i + 1]};
    } else if constexpr (bits == 16) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            reinterpret_cast<u2*>(row)[i] = u2_t{__builtin_amdgcn_perm(u[4 * i + 1], u[4 * i], 0x05040100u),
                                                 __builtin_amdgcn_perm(u[4 * i + 3], u[4 * i + 2], 0x05040100u)};
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(u[4 * i + 1], u[4 * i], 0x0c0c0400u);
            const uint32_t hi = __builtin_amdgcn_perm(u[4 * i + 3], u[4 * i + 2], 0x0c0c0400u);
            row[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
    }
}

// Extract the 12 W-bit fields that start at bit `q` of the LDS image; static shifts.
template <typename T, int W>
__device__ __forceinline__ void unpack_payload_w(const uint32_t* __restrict__ image, uint32_t q, uint32_t (&u)[kBlock]) {
    constexpr int NBITS = kBlock * W;
    constexpr int ND = (NBITS + 31) / 32;
    const uint32_t d = q >> 5, s = q & 31u;
    uint32_t raw[ND + 1];
#pragma unroll
    for (int j = 0; j <= ND; ++j) raw[j] = image[d + j];
    uint32_t x[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) x[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], s);   // string aligned to bit 0
#pragma unroll
    for (int k = 0; k < kBlock; ++k) {
        const int bit = k * W;
        uint32_t f = x[bit >> 5] >> (bit & 31);
        if ((bit & 31) + W > 32) f |= x[(bit >> 5) + 1] << (32 - (bit & 31));
        if (PixelTraits<T>::is_signed) u[k] = (uint32_t)((int32_t)(f << (32 - W)) >> (32 - W));   // sign-extend (:784-789)
        else u[k] = W >= 32 ? f : f & ((1u << (W & 31)) - 1u);
    }
}

// ---- register-resident variant (decode_frame.hip) ------------------------------------------------------------------
// The lane's stream dwords are already in registers (`raw`, loaded straight from L2 with dwordx4 loads starting at the
// dword that holds the block's first payload bit; s = that bit's position in raw[0]).  Extracts the 12 W-bit fields
// with static shifts (v_bfe_u32 / v_bfe_i32), packs them to the pixel type and stores the block: no LDS, no merge of
// the specialised bodies' results.
template <typename T> struct RawQuads { static constexpr int n = PixelTraits<T>::bits == 32 ? 4 : (PixelTraits<T>::bits == 16 ? 2 : 1); };

template <typename T, int W>
__device__ __forceinline__ void unpack_store_w(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, T* __restrict__ dst) {
    constexpr int bits = PixelTraits<T>::bits;
    constexpr int NBITS = kBlock * W;
    constexpr int ND = (NBITS + 31) / 32;
    static_assert(ND + 1 <= 4 * RawQuads<T>::n, "field string must fit the loaded quads");
    uint32_t x[ND ? ND : 1];
#pragma unroll
    for (int j = 0; j < ND; ++j) x[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], s);   // string aligned to bit 0
    uint32_t f[kBlock];
#pragma unroll
    for (int k = 0; k < kBlock; ++k) {
        if constexpr (W == 0) f[k] = 0u;
        else {
            const int bit = k * W;
            uint32_t y = x[bit >> 5] >> (bit & 31);
            if ((bit & 31) + W > 32) y |= x[(bit >> 5) + 1] << (32 - (bit & 31));
            if (PixelTraits<T>::is_signed) f[k] = (uint32_t)((int32_t)(y << (32 - W)) >> (32 - W));   // sign-extend (:784-789)
            else f[k] = W >= 32 ? y : y & ((1u << (W & 31)) - 1u);
        }
    }
    typedef u4_a1 u4;                                                                // (dst is aligned to T only)
    typedef u2_a1 u2;
    if constexpr (bits == 32) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            u4_t o = {f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]};
            __builtin_nontemporal_store(o, reinterpret_cast<u4*>(dst) + i);
        }
    } else if constexpr (bits == 16) {                                               // v_perm_b32: {hi.lo16, lo.lo16}
        uint32_t o[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = __builtin_amdgcn_perm(f[2 * i + 1], f[2 * i], 0x05040100u);
        u4_t a = {o[0], o[1], o[2], o[3]};
        u2_t c = {o[4], o[5]};
        __builtin_nontemporal_store(a, reinterpret_cast<u4*>(dst));                  // 24 bytes: one 16-byte + one 8-byte store
        __builtin_nontemporal_store(c, reinterpret_cast<u2*>(dst) + 2);
    } else {
        uint32_t o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(f[4 * i + 1], f[4 * i], 0x0c0c0400u);       // {0, 0, b.byte0, a.byte0}
            const uint32_t hi = __builtin_amdgcn_perm(f[4 * i + 3], f[4 * i + 2], 0x0c0c0400u);
            o[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
        u2_t a = {o[0], o[1]};
        __builtin_nontemporal_store(a, reinterpret_cast<u2*>(dst));   // 12 bytes
        __builtin_nontemporal_store(o[2], reinterpret_cast<u1_a1*>(dst) + 2);
    }
}

// The same extraction into registers: the block's 12 pixels packed to the pixel type (3 / 6 / 12 dwords).  The caller
// stores them after ALL specialised passes of a group: the number of store instructions per group is then static, which
// lets a load issued before them (the next group's stream dwords) be awaited with s_waitcnt vmcnt(#stores) -- on gfx950
// loads and stores retire through one in-order counter, so a store issued inside a data-dependent number of passes
// would force vmcnt(0), i.e. a wait for the stores' round trip, in front of every group.
template <typename T> struct PackedDwords { static constexpr int n = kBlock * (int)sizeof(T) / 4; };

template <typename T, int W>
__device__ __forceinline__ void unpack_regs_w(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, uint32_t (&o)[PackedDwords<T>::n]) {
    constexpr int bits = PixelTraits<T>::bits;
    constexpr int NBITS = kBlock * W;
    constexpr int ND = (NBITS + 31) / 32;
    static_assert(ND + 1 <= 4 * RawQuads<T>::n, "field string must fit the loaded quads");
    uint32_t x[ND ? ND : 1];
#pragma unroll
    for (int j = 0; j < ND; ++j) x[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], s);   // string aligned to bit 0
    uint32_t f[kBlock];
#pragma unroll
    for (int k = 0; k < kBlock; ++k) {
        if constexpr (W == 0) f[k] = 0u;
        else {
            const int bit = k * W;
            uint32_t y = x[bit >> 5] >> (bit & 31);
            if ((bit & 31) + W > 32) y |= x[(bit >> 5) + 1] << (32 - (bit & 31));
            if (PixelTraits<T>::is_signed) f[k] = (uint32_t)((int32_t)(y << (32 - W)) >> (32 - W));   // sign-extend (:784-789)
            else f[k] = W >= 32 ? y : y & ((1u << (W & 31)) - 1u);
        }
    }
    if constexpr (bits == 32) {
#pragma unroll
        for (int i = 0; i < 12; ++i) o[i] = f[i];
    } else if constexpr (bits == 16) {                                               // v_perm_b32: {hi.lo16, lo.lo16}
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = __builtin_amdgcn_perm(f[2 * i + 1], f[2 * i], 0x05040100u);
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(f[4 * i + 1], f[4 * i], 0x0c0c0400u);       // {0, 0, b.byte0, a.byte0}
            const uint32_t hi = __builtin_amdgcn_perm(f[4 * i + 3], f[4 * i + 2], 0x0c0c0400u);
            o[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
    }
}

// A block's packed pixels -> memory: 12 bytes as 8 + 4, 24 bytes as 16 + 8, 48 bytes as 3 x 16 (non-temporal).
template <typename T>
__device__ __forceinline__ void store_packed(T* __restrict__ dst, const uint32_t (&o)[PackedDwords<T>::n]) {
    typedef u4_a1 u4;                                                                // (dst is aligned to T only)
    typedef u2_a1 u2;
    constexpr int bits = PixelTraits<T>::bits;
    if constexpr (bits == 32) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            u4_t v = {o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
            __builtin_nontemporal_store(v, reinterpret_cast<u4*>(dst) + i);
        }
    } else if constexpr (bits == 16) {
        u4_t a = {o[0], o[1], o[2], o[3]};
        u2_t c = {o[4], o[5]};
        __builtin_nontemporal_store(a, reinterpret_cast<u4*>(dst));
        __builtin_nontemporal_store(c, reinterpret_cast<u2*>(dst) + 2);
    } else {
        u2_t a = {o[0], o[1]};
        __builtin_nontemporal_store(a, reinterpret_cast<u2*>(dst));
        __builtin_nontemporal_store(o[2], reinterpret_cast<u1_a1*>(dst) + 2);
    }
}

// A block's packed pixels -> the lane's LDS staging row (row = staging + lane * 3 * sizeof(T) dwords): see store_group.
template <typename T>
__device__ __forceinline__ void stage_packed(uint32_t* __restrict__ row, const uint32_t (&o)[PackedDwords<T>::n]) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<u4*>(row)[i] = u4{o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) reinterpret_cast<u2*>(row)[i] = u2{o[2 * i], o[2 * i + 1]};
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) row[i] = o[i];
    }
}

// The same under a lane mask, without a divergent branch in the caller's control flow (one asm statement that narrows
// EXEC around the LDS stores): with `if (mine) stage_packed(...)` in a loop the compiler structurises every wave-uniform
// branch of that loop as well -- a flag register and three more scalar instructions per level of the width dispatch.
// (LDS executes a wave's instructions in order: the wave's later reads of the row need no wait on these stores.)
template <typename T>
__device__ __forceinline__ void stage_packed_masked(uint32_t* __restrict__ row, const uint32_t (&o)[PackedDwords<T>::n], uint64_t lanes) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    const uint32_t addr = (uint32_t)(uintptr_t)row;
    uint64_t keep;
    if constexpr (sizeof(T) == 4) {
        const u4 a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]}, c = {o[8], o[9], o[10], o[11]};
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %5\n\tds_write_b128 %1, %2\n\tds_write_b128 %1, %3 offset:16\n\t"
                     "ds_write_b128 %1, %4 offset:32\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(addr), "v"(a), "v"(b), "v"(c), "s"(lanes) : "memory", "scc");
    } else if constexpr (sizeof(T) == 2) {
        const u2 a = {o[0], o[1]}, b = {o[2], o[3]}, c = {o[4], o[5]};
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %5\n\tds_write_b64 %1, %2\n\tds_write_b64 %1, %3 offset:8\n\t"
                     "ds_write_b64 %1, %4 offset:16\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(addr), "v"(a), "v"(b), "v"(c), "s"(lanes) : "memory", "scc");
    } else {
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %5\n\tds_write2_b32 %1, %2, %3 offset1:1\n\t"
                     "ds_write_b32 %1, %4 offset:8\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(addr), "v"(o[0]), "v"(o[1]), "v"(o[2]), "s"(lanes) : "memory", "scc");
    }
}

template <typename T, int LO, int HI>
struct UnpackRegsDispatch {
    static __device__ __forceinline__ void run(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, uint32_t w0, uint32_t (&o)[PackedDwords<T>::n]) {
        if constexpr (LO == HI) unpack_regs_w<T, LO>(raw, s, o);
        else {
            constexpr int MID = (LO + HI) / 2;
            if (w0 <= (uint32_t)MID) UnpackRegsDispatch<T, LO, MID>::run(raw, s, w0, o);
            else UnpackRegsDispatch<T, MID + 1, HI>::run(raw, s, w0, o);
        }
    }
};

// The same extraction, but the block's 12 pixels go to the wavefront's LDS staging row instead of straight to memory
// (`row` = staging + lane * 3 * sizeof(T) dwords, 8-byte aligned for 16/32-bit pixels): the wavefront then writes its
// 64 blocks as whole 16-byte-per-lane stores (store_group), every store instruction covering consecutive bytes.
// A lane-owned 24-byte run written as 16 + 8 byte non-temporal stores reaches 2.4 TB/s on MI355X, the same bytes
// exchanged through LDS and stored 16 bytes per lane 5.4 TB/s (tools/wrbench.hip).
template <typename T, int W>
__device__ __forceinline__ void unpack_stage_w(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, uint32_t* __restrict__ row) {
    constexpr int bits = PixelTraits<T>::bits;
    constexpr int NBITS = kBlock * W;
    constexpr int ND = (NBITS + 31) / 32;
    static_assert(ND + 1 <= 4 * RawQuads<T>::n, "field string must fit the loaded quads");
    uint32_t x[ND ? ND : 1];
#pragma unroll
    for (int j = 0; j < ND; ++j) x[j] = __builtin_amdgcn_alignbit(raw[j + 1], raw[j], s);   // string aligned to bit 0
    uint32_t f[kBlock];
#pragma unroll
    for (int k = 0; k < kBlock; ++k) {
        if constexpr (W == 0) f[k] = 0u;
        else {
            const int bit = k * W;
            uint32_t y = x[bit >> 5] >> (bit & 31);
            if ((bit & 31) + W > 32) y |= x[(bit >> 5) + 1] << (32 - (bit & 31));
            if (PixelTraits<T>::is_signed) f[k] = (uint32_t)((int32_t)(y << (32 - W)) >> (32 - W));   // sign-extend (:784-789)
            else f[k] = W >= 32 ? y : y & ((1u << (W & 31)) - 1u);
        }
    }
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    if constexpr (bits == 32) {
#pragma unroll
        for (int i = 0; i < 6; ++i) reinterpret_cast<u2*>(row)[i] = u2{f[2 * i], f[2 * i + 1]};
    } else if constexpr (bits == 16) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            reinterpret_cast<u2*>(row)[i] = u2{__builtin_amdgcn_perm(f[4 * i + 1], f[4 * i], 0x05040100u),
                                               __builtin_amdgcn_perm(f[4 * i + 3], f[4 * i + 2], 0x05040100u)};
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(f[4 * i + 1], f[4 * i], 0x0c0c0400u);       // {0, 0, b.byte0, a.byte0}
            const uint32_t hi = __builtin_amdgcn_perm(f[4 * i + 3], f[4 * i + 2], 0x0c0c0400u);
            row[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        }
    }
}

// The wavefront's staged group (64 blocks = 768 pixels, consecutive in `staging`) -> memory, `group_dst` 128-byte aligned.
// Every store instruction is executed by all 64 lanes and writes whole 128-byte lines: 3072 bytes as three 16-byte-per-lane
// stores, 1536 bytes as one 16-byte and one 8-byte-per-lane store, 768 bytes as one 8-byte and one 4-byte-per-lane store (no
// lane masks, no branches).  skip_line0: the lanes of the first instruction that cover the first line do not store.
template <typename T, bool ALIGNED = true>
__device__ __forceinline__ void store_group(const uint32_t* __restrict__ staging, T* __restrict__ group_dst, bool skip_line0 = false) {
    typedef uint32_t u4l __attribute__((ext_vector_type(4)));
    typedef uint32_t u2l __attribute__((ext_vector_type(2)));
    // ALIGNED = false: group_dst is aligned to T only (a frame that starts inside a cache line, stored from its first pixel: the
    // hardware's unaligned access mode splits the accesses -- what the kernels without a walker beside them do, see store_group_lines)
    typedef std::conditional_t<ALIGNED, u4l, u4_a1> u4;
    typedef std::conditional_t<ALIGNED, u2l, u2_a1> u2;
    typedef std::conditional_t<ALIGNED, uint32_t, u1_a1> u1;
    const int lane = lane_id();
    uint32_t* dst = reinterpret_cast<uint32_t*>(group_dst);
    if constexpr (sizeof(T) == 4) {
        if (!(skip_line0 && lane < 8))
            __builtin_nontemporal_store(reinterpret_cast<const u4l*>(staging)[lane], reinterpret_cast<u4*>(dst) + lane);
#pragma unroll
        for (int i = 1; i < 3; ++i)
            __builtin_nontemporal_store(reinterpret_cast<const u4l*>(staging)[i * kWave + lane], reinterpret_cast<u4*>(dst) + i * kWave + lane);
    } else if constexpr (sizeof(T) == 2) {
        if (!(skip_line0 && lane < 8))
            __builtin_nontemporal_store(reinterpret_cast<const u4l*>(staging)[lane], reinterpret_cast<u4*>(dst) + lane);
        __builtin_nontemporal_store(reinterpret_cast<const u2l*>(staging + 256)[lane], reinterpret_cast<u2*>(dst + 256) + lane);
    } else {
        if (!(skip_line0 && lane < 16))
            __builtin_nontemporal_store(reinterpret_cast<const u2l*>(staging)[lane], reinterpret_cast<u2*>(dst) + lane);
        __builtin_nontemporal_store((staging + 128)[lane], reinterpret_cast<u1*>(dst + 128) + lane);
    }
}

// The same for a frame that starts anywhere inside a cache line (most detectors: 1030 x 1065, 513 x 511 pixels ...).  A group
// is 6 / 12 / 24 whole lines long, so every group of a frame starts at the same offset c = (address of the frame's first
// pixel) & 127 inside its line.  The lanes stage their blocks exactly as for aligned frames (row byte = lane * block bytes:
// aligned LDS writes); the row has kStageCarryDw dwords of head room IN FRONT of it, and the wave stores the 128-byte LINES
// [group_dst - c, group_dst - c + G): each lane reads its 16 (8, 4) bytes c bytes in front of where an aligned frame would
// read them -- dword-aligned LDS reads + one funnel shift per dword when c is no multiple of 4 -- so that every global store
// instruction writes whole lines at aligned addresses.  The group's last c bytes are left over: a wave's next group, if it
// follows at once (`cont`), finds them in the head room (`carry_live`); only the two ends of such a run are partial lines
// (2 bytes per lane).  (Measured, 2000 x (513 x 511) u16, k_decode_frames: lane-owned rows stored from the frame's first pixel --
// every 16-byte store misaligned -- 0.32 ms; line images with the blocks staged at row byte c + ..., i.e. misaligned 8-byte LDS
// WRITES, 0.36 ms; this version 0.26 ms; 512 x 512: 0.22 ms.  The kernels WITHOUT a walker beside the extraction waves --
// k_decode_frames_indexed 0.27 ms, k_unpack_tiles -- are faster with the plain misaligned stores (0.27 against 0.36 ms, 0.126
// against 0.136 ms for 200 x (1030 x 1065)): there the misaligned stores do not hold up a walker's window loads in the CU's
// one vector-memory pipeline, and the shifted reads are all cost.)
constexpr int kStageCarryDw = 32;
template <int N>                                         // N dwords from LDS byte address `addr` (any 2-byte / 1-byte offset)
__device__ __forceinline__ void lds_read_shifted(const uint32_t* __restrict__ lds0, uint32_t addr, uint32_t (&out)[N]) {
    const uint32_t* p = lds0 + (addr >> 2);
    const uint32_t sh = 8u * (addr & 3u);
    uint32_t d[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) d[i] = p[i];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
}
// `row` = the group's row (the head room lies in front of it); all offsets below are row-relative bytes, biased by the head room
template <typename T>
__device__ __forceinline__ void store_group_lines(uint32_t* __restrict__ row, T* __restrict__ group_dst, uint32_t c,
                                                  bool carry_live, bool cont) {
    if (c == 0u) { store_group<T>(row, group_dst); return; }                  // (wave-uniform: line-aligned frames)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    constexpr uint32_t G = kWave * kBlock * (uint32_t)sizeof(T);               // group bytes: a multiple of 128
    constexpr uint32_t H = 4u * kStageCarryDw;                                 // head room, bytes
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t* const buf = row - kStageCarryDw;                                 // buffer byte H + i = row byte i = address group_dst + i
    char* const bufb = reinterpret_cast<char*>(buf);
    char* const base = reinterpret_cast<char*>(group_dst) - c;                 // 128-byte aligned
    const uint32_t o0 = H - c;                                                 // buffer byte of the first line's first byte
    const bool skip0 = !carry_live;                                            // the run's first line: only its bytes [c, 128) are ours
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            uint32_t x[4];
            lds_read_shifted<4>(buf, o0 + 16u * (i * kWave + lane), x);
            if (!(i == 0 && skip0 && lane < 8u))
                __builtin_nontemporal_store(u4{x[0], x[1], x[2], x[3]}, reinterpret_cast<u4*>(base) + i * kWave + lane);
        }
    } else if constexpr (sizeof(T) == 2) {
        uint32_t x[4], y[2];
        lds_read_shifted<4>(buf, o0 + 16u * lane, x);
        lds_read_shifted<2>(buf, o0 + 1024u + 8u * lane, y);
        if (!(skip0 && lane < 8u)) __builtin_nontemporal_store(u4{x[0], x[1], x[2], x[3]}, reinterpret_cast<u4*>(base) + lane);
        __builtin_nontemporal_store(u2{y[0], y[1]}, reinterpret_cast<u2*>(base + 1024) + lane);
    } else {
        uint32_t x[2], y[1];
        lds_read_shifted<2>(buf, o0 + 8u * lane, x);
        lds_read_shifted<1>(buf, o0 + 512u + 4u * lane, y);
        if (!(skip0 && lane < 16u)) __builtin_nontemporal_store(u2{x[0], x[1]}, reinterpret_cast<u2*>(base) + lane);
        __builtin_nontemporal_store(y[0], reinterpret_cast<uint32_t*>(base + 512) + lane);
    }
    if (skip0) {                                                             // the run's first line: its bytes [c, 128) = row bytes [0, 128 - c)
        if constexpr (sizeof(T) == 1) {
            if (2u * lane >= c) base[2u * lane] = bufb[o0 + 2u * lane];
            if (2u * lane + 1u >= c) base[2u * lane + 1u] = bufb[o0 + 2u * lane + 1u];
        } else {
            if (2u * lane >= c) *reinterpret_cast<uint16_t*>(base + 2u * lane) = *reinterpret_cast<const uint16_t*>(bufb + o0 + 2u * lane);
        }
    }
    // the group's last c bytes (row bytes [G - c, G)): into the head room for the next group, or out as the run's last line
    if constexpr (sizeof(T) == 1) {
#pragma unroll
        for (uint32_t k = 0; k < 2u; ++k) {
            const uint32_t i = lane + 64u * k;
            if (i < c) {
                const char v = bufb[H + G - c + i];
                if (cont) bufb[o0 + i] = v; else (base + G)[i] = v;
            }
        }
    } else {
        if (2u * lane < c) {
            const uint16_t v = *reinterpret_cast<const uint16_t*>(bufb + H + G - c + 2u * lane);
            if (cont) *reinterpret_cast<uint16_t*>(bufb + o0 + 2u * lane) = v;
            else *reinterpret_cast<uint16_t*>(base + G + 2u * lane) = v;
        }
    }
}

template <typename T, int LO, int HI>
struct UnpackStageDispatch {
    static __device__ __forceinline__ void run(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, uint32_t w0, uint32_t* row) {
        if constexpr (LO == HI) unpack_stage_w<T, LO>(raw, s, row);
        else {
            constexpr int MID = (LO + HI) / 2;
            if (w0 <= (uint32_t)MID) UnpackStageDispatch<T, LO, MID>::run(raw, s, w0, row);
            else UnpackStageDispatch<T, MID + 1, HI>::run(raw, s, w0, row);
        }
    }
};

template <typename T, int LO, int HI>
struct UnpackStoreDispatch {
    static __device__ __forceinline__ void run(const uint32_t (&raw)[4 * RawQuads<T>::n], uint32_t s, uint32_t w0, T* dst) {
        if constexpr (LO == HI) unpack_store_w<T, LO>(raw, s, dst);
        else {
            constexpr int MID = (LO + HI) / 2;
            if (w0 <= (uint32_t)MID) UnpackStoreDispatch<T, LO, MID>::run(raw, s, w0, dst);
            else UnpackStoreDispatch<T, MID + 1, HI>::run(raw, s, w0, dst);
        }
    }
};

template <typename T, int LO, int HI>
struct UnpackDispatch {
    static __device__ __forceinline__ void run(const uint32_t* image, uint32_t q, uint32_t w0, uint32_t (&u)[kBlock]) {
        if constexpr (LO == HI) unpack_payload_w<T, LO>(image, q, u);
        else {
            constexpr int MID = (LO + HI) / 2;
            if (w0 <= (uint32_t)MID) UnpackDispatch<T, LO, MID>::run(image, q, w0, u);
            else UnpackDispatch<T, MID + 1, HI>::run(image, q, w0, u);
        }
    }
};

}  // namespace trpx
