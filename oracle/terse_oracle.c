/*
 * oracle/terse_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement (plain C) of the reference TERSE encode / PROLIX decode loop of
 * senikm/trpx @ 2024_08_07.  It is the parity CHECKER for the HIP path and the fallback
 * "port" CPU baseline of bench.py.  Nothing in the product path (trpx_amd/, include/) may
 * import, link, call or execute anything in this directory; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this file against
 *   (a) the known-answer vectors of the reference's doc comments (Terse.hpp:53-57, :127-154),
 *   (b) committed golden fixtures under tests/golden/ that were produced by the REAL reference
 *       (oracle/_ref/libtrpx_ref.so = oracle/ref_shim.cpp compiled against
 *       /root/reference/include in place; generator: tests/golden/make_golden.py),
 *   (c) live differential runs against oracle/_ref whenever that library is present.
 *
 * Written from the normative bitstream description, not from the
 * reference's class structure.  Reference lines each function follows are cited inline
 * (paths relative to /root/reference/).
 *
 * Behaviour outside the reference's validity domain (signed values whose block
 * width would reach the bit size of T, the type minimum, u32 >= 2^31 on decode) is DEFINED here
 * as the mathematically consistent extension (width clamped to the bit size of T, proper
 * masking); the reference is broken there, so no parity is claimed outside the domain.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <time.h>

enum { TRPX_U8 = 0, TRPX_I8 = 1, TRPX_U16 = 2, TRPX_I16 = 3, TRPX_U32 = 4, TRPX_I32 = 5,
       TRPX_U64 = 6, TRPX_I64 = 7,
       ORACLE_F32 = 8, ORACLE_F64 = 9 };   /* decode output only: Terse.hpp:379-383 (non-integral iterators) */

static unsigned dtype_bytes(int dt) {
    switch (dt) {
    case TRPX_U8: case TRPX_I8: return 1;
    case TRPX_U16: case TRPX_I16: return 2;
    case TRPX_U32: case TRPX_I32: return 4;
    case TRPX_U64: case TRPX_I64: case ORACLE_F64: return 8;
    case ORACLE_F32: return 4;
    }
    return 0;
}
static int dtype_signed(int dt) { return dt & 1; }

/* Load element i of a typed array as (sign- or zero-extended) 64-bit integer. */
static inline int64_t load_elem(int dt, const void* p, size_t i) {
    switch (dt) {
    case TRPX_U8:  return ((const uint8_t*)p)[i];
    case TRPX_I8:  return ((const int8_t*)p)[i];
    case TRPX_U16: return ((const uint16_t*)p)[i];
    case TRPX_I16: return ((const int16_t*)p)[i];
    case TRPX_U32: return ((const uint32_t*)p)[i];
    case TRPX_I32: return ((const int32_t*)p)[i];
    case TRPX_U64: return (int64_t)((const uint64_t*)p)[i];
    default:       return ((const int64_t*)p)[i];
    }
}

/* Bit length of an unsigned magnitude (Terse.hpp:555-558: shift loop until zero). */
static inline unsigned bitlen64(uint64_t v) { return v ? 64u - (unsigned)__builtin_clzll(v) : 0u; }

/*
 * Slot size that can hold any encoding of one frame.  The reference allocates
 * ceil(N*(sizeof(T) + 12/(block*8))) (Terse.hpp:503), which is up to 2 bytes short when every
 * block carries a 12-bit header and N % block != 0; this bound is exact:
 * N*sizeof(T) payload + 12 header bits per block + the trailing byte of Terse.hpp:547.
 */
size_t trpx_oracle_worst_case_bytes(int dtype, size_t n, unsigned block) {
    size_t nblocks = (n + block - 1) / block;
    return n * dtype_bytes(dtype) + (12 * nblocks + 7) / 8 + 1;
}

/* Significant-bit width of one block (Terse.hpp:508-515 OR-scan, :551-560 bit length). */
static inline unsigned block_width(int dt, const void* px, size_t from, size_t to) {
    const unsigned tbits = 8 * dtype_bytes(dt);
    uint64_t m = 0;
    if (dtype_signed(dt)) {
        for (size_t i = from; i < to; ++i) {                /* setbits |= abs(v)  (Terse.hpp:514) */
            int64_t v = load_elem(dt, px, i);
            m |= v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        }
        unsigned w = m ? bitlen64(m) + 1 : 0;               /* 1 + bitlen(|v|)   (Terse.hpp:554) */
        return w > tbits ? tbits : w;                       /* extension outside D3's domain     */
    }
    for (size_t i = from; i < to; ++i) m |= (uint64_t)load_elem(dt, px, i);   /* Terse.hpp:512 */
    if (tbits < 64) m &= (((uint64_t)1 << tbits) - 1);
    return bitlen64(m);
}

/* Per-block widths only (used to check the GPU width scan on its own). Returns max width. */
unsigned trpx_oracle_widths(int dtype, const void* px, size_t n, unsigned block, uint8_t* w_out) {
    unsigned mx = 0;
    size_t b = 0;
    for (size_t from = 0; from < n; from += block, ++b) {
        size_t to = from + block < n ? from + block : n;
        unsigned w = block_width(dtype, px, from, to);
        if (w_out) w_out[b] = (uint8_t)w;
        if (w > mx) mx = w;
    }
    return mx;
}

/* LSB-first bit writer over a zero-initialised byte buffer (Bit_pointer.hpp:438,490,634,711). */
typedef struct { uint8_t* base; size_t bit; } bitw_t;

static inline void put_bits(bitw_t* b, uint64_t v, unsigned w) {   /* v < 2^w, w <= 64 */
    while (w) {
        unsigned sh = (unsigned)(b->bit & 7);
        unsigned take = 8 - sh < w ? 8 - sh : w;
        b->base[b->bit >> 3] |= (uint8_t)((v & (((uint64_t)1 << take) - 1)) << sh);
        v = take < 64 ? v >> take : 0;
        w -= take;
        b->bit += take;
    }
}

/*
 * Encode one frame (Terse.hpp:500-549).  `out` must hold trpx_oracle_worst_case_bytes().
 * Returns the frame's stream length S = 1 + total_bits/8 (Terse.hpp:547) or -1 on overflow.
 * *prolix_bits is max-updated (Terse.hpp:516) -- pass a zero-initialised accumulator.
 */
long trpx_oracle_encode(int dtype, const void* px, size_t n, unsigned block, uint8_t* out,
                        size_t cap, unsigned* prolix_bits) {
    size_t need = trpx_oracle_worst_case_bytes(dtype, n, block);
    if (cap < need || block == 0) return -1;
    memset(out, 0, need);                                   /* zero-initialised (Terse.hpp:503) */
    bitw_t bw = { out, 0 };
    unsigned prev = 0;                                      /* prevbits = 0 per frame (:505)    */
    for (size_t from = 0; from < n; from += block) {
        size_t to = from + block < n ? from + block : n;
        unsigned w = block_width(dtype, px, from, to);
        if (prolix_bits && w > *prolix_bits) *prolix_bits = w;
        if (w == prev) {
            put_bits(&bw, 1, 1);                            /* "same" bit (Terse.hpp:517-520)   */
        } else {
            put_bits(&bw, 0, 1);
            if (w < 7) put_bits(&bw, w, 3);                                   /* :522-525 */
            else if (w < 10) put_bits(&bw, 7u + ((w - 7) << 3), 5);           /* :526-529 */
            else put_bits(&bw, 31u + ((uint64_t)(w - 10) << 5), 11);          /* :530-533 */
            prev = w;
        }
        if (w) {                                            /* payload (Bit_pointer.hpp:700-730) */
            uint64_t mask = w < 64 ? (((uint64_t)1 << w) - 1) : ~(uint64_t)0;
            for (size_t i = from; i < to; ++i)
                put_bits(&bw, (uint64_t)load_elem(dtype, px, i) & mask, w);
        }
    }
    return (long)(1 + bw.bit / 8);                          /* Terse.hpp:547 */
}

/* LSB-first bit reader with bounds check (the reference has none: SURVEY.md section 5). */
typedef struct { const uint8_t* base; size_t nbits; size_t bit; int err; } bitr_t;

static inline uint64_t get_bits(bitr_t* b, unsigned w) {
    uint64_t v = 0;
    unsigned got = 0;
    if (b->bit + w > b->nbits) { b->err = 1; return 0; }
    while (got < w) {
        unsigned sh = (unsigned)(b->bit & 7);
        unsigned take = 8 - sh < w - got ? 8 - sh : w - got;
        uint64_t piece = (b->base[b->bit >> 3] >> sh) & ((1u << take) - 1);
        v |= piece << got;
        got += take;
        b->bit += take;
    }
    return v;
}

static inline void store_elem_clamped(int dt, void* p, size_t i, int64_t v, int v_is_unsigned64) {
    /* Narrowing conversions clamp (Bit_pointer.hpp:747-763); widening keeps the value. */
    switch (dt) {
    case TRPX_U8:  ((uint8_t*)p)[i]  = (uint8_t)(v_is_unsigned64 || v > 0xFF ? 0xFF : v < 0 ? 0 : v); break;
    case TRPX_I8:  ((int8_t*)p)[i]   = (int8_t)(v_is_unsigned64 || v > 127 ? 127 : v < -128 ? -128 : v); break;
    case TRPX_U16: ((uint16_t*)p)[i] = (uint16_t)(v_is_unsigned64 || v > 0xFFFF ? 0xFFFF : v < 0 ? 0 : v); break;
    case TRPX_I16: ((int16_t*)p)[i]  = (int16_t)(v_is_unsigned64 || v > 32767 ? 32767 : v < -32768 ? -32768 : v); break;
    case TRPX_U32: ((uint32_t*)p)[i] = (uint32_t)(v_is_unsigned64 || v > 0xFFFFFFFFLL ? 0xFFFFFFFFu : v < 0 ? 0 : v); break;
    case TRPX_I32: ((int32_t*)p)[i]  = (int32_t)(v_is_unsigned64 || v > 2147483647LL ? 2147483647 : v < -2147483648LL ? -2147483648LL : v); break;
    case TRPX_U64: ((uint64_t*)p)[i] = v_is_unsigned64 ? (uint64_t)v : (v < 0 ? 0 : (uint64_t)v); break;
    /* non-integral output: begin[i] = double(std::uint64_t(bitr)) / double(std::int64_t(bitr)) (Terse.hpp:379-383) */
    case ORACLE_F32: ((float*)p)[i]  = (float)(v_is_unsigned64 ? (double)(uint64_t)v : (double)v); break;
    case ORACLE_F64: ((double*)p)[i] = v_is_unsigned64 ? (double)(uint64_t)v : (double)v; break;
    default:       ((int64_t*)p)[i]  = v_is_unsigned64 ? INT64_MAX : v; break;
    }
}

/*
 * Decode one frame (Terse.hpp:352-389 header state machine, Bit_pointer.hpp:742-792 unpack).
 * stream_signed = the header's `signed` attribute.  Values are zero-extended (unsigned stream)
 * or sign-extended from bit w-1 (signed stream, Bit_pointer.hpp:784-789), then stored into
 * dtype_out with clamping when narrower.  Same-type decode is the reference's contract;
 * cross-type decode implements the *intended* value semantics.
 * Returns the number of stream bytes consumed S (= 1 + bits/8) or -1 on a truncated stream.
 */
long trpx_oracle_decode(int dtype_out, int stream_signed, const uint8_t* in, size_t nbytes,
                        size_t n, unsigned block, void* out) {
    bitr_t br = { in, nbytes * 8, 0, 0 };
    unsigned w = 0;                                         /* significant_bits = 0 (:359) */
    if (block == 0) return -1;
    for (size_t from = 0; from < n; from += block) {
        size_t to = from + block < n ? from + block : n;
        if (get_bits(&br, 1) == 0) {                        /* Terse.hpp:361 */
            w = (unsigned)get_bits(&br, 3);                 /* :362 */
            if (w == 7) {
                w += (unsigned)get_bits(&br, 2);            /* :365 */
                if (w == 10) w += (unsigned)get_bits(&br, 6);   /* :368 */
            }
        }
        if (br.err || w > 64) return -1;
        if (w == 0) {
            for (size_t i = from; i < to; ++i) store_elem_clamped(dtype_out, out, i, 0, 0);   /* :373 */
            continue;
        }
        for (size_t i = from; i < to; ++i) {
            uint64_t u = get_bits(&br, w);
            if (br.err) return -1;
            if (stream_signed) {
                int64_t s = (int64_t)u;
                if (w < 64 && (u >> (w - 1)) & 1) s = (int64_t)(u | ~(((uint64_t)1 << w) - 1));
                store_elem_clamped(dtype_out, out, i, s, 0);
            } else {
                store_elem_clamped(dtype_out, out, i, (int64_t)u, (u >> 63) != 0);
            }
        }
    }
    if (br.bit / 8 + 1 > nbytes) return -1;                 /* the +1 byte of Terse.hpp:547 */
    return (long)(1 + br.bit / 8);
}

/* Length in bytes of the frame starting at `in` without producing pixels (intended semantics
 * of Terse.hpp:562-585 with defects D1/D2 fixed: partial last block skips n_b*w bits). */
long trpx_oracle_frame_bytes(const uint8_t* in, size_t nbytes, size_t n, unsigned block) {
    bitr_t br = { in, nbytes * 8, 0, 0 };
    unsigned w = 0;
    for (size_t from = 0; from < n; from += block) {
        size_t to = from + block < n ? from + block : n;
        if (get_bits(&br, 1) == 0) {
            w = (unsigned)get_bits(&br, 3);
            if (w == 7) { w += (unsigned)get_bits(&br, 2); if (w == 10) w += (unsigned)get_bits(&br, 6); }
        }
        if (br.err) return -1;
        br.bit += (size_t)w * (to - from);
        if (br.bit > br.nbits) return -1;
    }
    if (br.bit / 8 + 1 > nbytes) return -1;
    return (long)(1 + br.bit / 8);
}

/* ------------------------------------------------------------------------------------------
 * synth-v1 generator (SURVEY.md section 8 row d): counter based, identical on CPU and GPU.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t synth_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t synth_r(uint64_t seed, uint64_t f, uint64_t n, uint64_t i) {
    return synth_mix(seed + 0x9E3779B97F4A7C15ULL * (f * n + i + 1));
}
void trpx_oracle_synth_u16(uint64_t seed, uint64_t frame0, size_t frames, size_t n, uint16_t* out) {
    for (size_t f = 0; f < frames; ++f)
        for (size_t i = 0; i < n; ++i) {
            uint64_t r = synth_r(seed, frame0 + f, n, i);
            out[f * n + i] = ((r >> 40) & 0xFFF) == 0 ? (uint16_t)((r >> 24) & 0xFFF)
                                                      : (uint16_t)__builtin_popcountll(r & 0x3F);
        }
}
void trpx_oracle_synth_i32(uint64_t seed, uint64_t frame0, size_t frames, size_t n, int32_t* out) {
    for (size_t f = 0; f < frames; ++f)
        for (size_t i = 0; i < n; ++i) {
            uint64_t r = synth_r(seed, frame0 + f, n, i);
            out[f * n + i] = ((r >> 40) & 0x3FF) == 0 ? (int32_t)((r >> 8) & 0xFFFFFF)
                                                      : (int32_t)__builtin_popcountll(r & 0x3F) - 3;
        }
}

uint64_t trpx_oracle_fnv1a64(const void* p, size_t nbytes) {
    const uint8_t* b = (const uint8_t*)p;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < nbytes; ++i) h = (h ^ b[i]) * 0x100000001b3ULL;
    return h;
}

/* ------------------------------------------------------------------------------------------
 * CPU-baseline timing ("port" kind): frames strided over `threads` pthreads, one independent
 * encode per frame, then decode + verify.  Seconds are wall-clock for the whole
 * batch (max over threads).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dtype; const uint8_t* px; size_t n, frames; unsigned block; int tid, nthreads;
    uint8_t* slots; size_t slot; long* sizes; void* back; int phase; int ok;
} job_t;

static void* job_main(void* arg) {
    job_t* j = (job_t*)arg;
    size_t esz = dtype_bytes(j->dtype);
    uint8_t* back = (uint8_t*)j->back + (size_t)j->tid * j->n * esz;
    for (size_t f = (size_t)j->tid; f < j->frames; f += (size_t)j->nthreads) {
        const uint8_t* src = j->px + f * j->n * esz;
        if (j->phase == 0) {
            unsigned pb = 0;
            j->sizes[f] = trpx_oracle_encode(j->dtype, src, j->n, j->block, j->slots + f * j->slot, j->slot, &pb);
        } else {
            long s = trpx_oracle_decode(j->dtype, dtype_signed(j->dtype), j->slots + f * j->slot,
                                        (size_t)j->sizes[f], j->n, j->block, back);
            if (s != j->sizes[f] || memcmp(back, src, j->n * esz) != 0) j->ok = 0;
        }
    }
    return NULL;
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int trpx_oracle_time(int dtype, const void* px, size_t n, size_t frames, unsigned block, int threads,
                     double* enc_s, double* dec_s, size_t* total_bytes, int* roundtrip_ok) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    size_t slot = trpx_oracle_worst_case_bytes(dtype, n, block);
    uint8_t* slots = (uint8_t*)malloc(slot * frames);
    long* sizes = (long*)calloc(frames, sizeof(long));
    void* back = malloc((size_t)threads * n * dtype_bytes(dtype));
    job_t jobs[256]; pthread_t th[256];
    if (!slots || !sizes || !back) { free(slots); free(sizes); free(back); return -1; }
    *roundtrip_ok = 1;
    for (int phase = 0; phase < 2; ++phase) {
        double t0 = now_s();
        for (int t = 0; t < threads; ++t) {
            job_t j = { dtype, (const uint8_t*)px, n, frames, block, t, threads, slots, slot, sizes, back, phase, 1 };
            jobs[t] = j;
            pthread_create(&th[t], NULL, job_main, &jobs[t]);
        }
        for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); if (!jobs[t].ok) *roundtrip_ok = 0; }
        double dt = now_s() - t0;
        if (phase == 0) *enc_s = dt; else *dec_s = dt;
    }
    *total_bytes = 0;
    for (size_t f = 0; f < frames; ++f) { if (sizes[f] < 0) *roundtrip_ok = 0; else *total_bytes += (size_t)sizes[f]; }
    free(slots); free(sizes); free(back);
    return 0;
}
