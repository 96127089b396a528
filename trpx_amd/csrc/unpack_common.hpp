// Width-specialised field extraction and block stores shared by the PROLIX decode kernels
// (decode_fast.hip: k_unpack_tiles, decode_frame.hip: k_decode_frames).  Replaces
// Bit_range::get_range / operator T() (reference include/Bit_pointer.hpp:742-792, :597-617).
#pragma once
#include "codec_common.hpp"

namespace trpx {

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
    V* q = reinterpret_cast<V*>(dst);
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
