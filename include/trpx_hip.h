/*
 * trpx_hip.h -- C ABI of libtrpx_hip.so: the MI355X (gfx950) implementation of the
 * TERSE encode / PROLIX decode hot path of senikm/trpx.
 *
 * The reference has no FFI/plugin boundary: the hot path is the two inline bodies
 * jpa::Terse::f_compress (include/Terse.hpp:500-549) and jpa::Terse::prolix(Iterator, frame)
 * (include/Terse.hpp:352-389) plus the Bit_pointer.hpp primitives they use.  This header is
 * the boundary a maintainer would bind instead (INTEGRATION.md shows the binding).  Plain
 * pointers and sizes only, no C++/torch types, no exceptions across the boundary.
 *
 * Conventions
 *   - every function returning int returns TRPX_OK (0) or a trpx_status error code;
 *     trpx_last_error_string() gives the thread's last error text.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - functions taking DEVICE pointers only enqueue work on `stream`: no allocation, no
 *     host synchronisation (they can be captured into a hipGraph); the caller supplies the
 *     workspace.  The *_host convenience entry points own their staging and synchronise.
 *   - the bitstream is bit-identical to the reference's: a stack is the
 *     plain concatenation of its frames, frame k starting at byte frame_offsets[k]
 *     (Terse.hpp:502-504; the intended semantics of Terse.hpp:562-585).
 */
#ifndef TRPX_HIP_H
#define TRPX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the opaque decode index / workspace layouts grew (hand-over list with its statistics, part table); trpx_bench_stream.
 * A caller compiled against one version must not run against a library of another: size its buffers with THIS library's
 * trpx_index_bytes / trpx_*_workspace_bytes and compare trpx_abi_version() with TRPX_ABI_VERSION first (the Terse classes do). */
#define TRPX_ABI_VERSION 3

typedef enum trpx_status {
    TRPX_OK = 0,
    TRPX_ERR_INVALID_ARG = 1,   /* null/misaligned pointer, zero sizes, unknown dtype        */
    TRPX_ERR_UNSUPPORTED = 2,   /* block outside 1..4096, decode index with block != 12, ...   */
    TRPX_ERR_CAPACITY = 3,      /* output or workspace too small                              */
    TRPX_ERR_HIP = 4,           /* a HIP runtime call failed (text has the HIP error)         */
    TRPX_ERR_CORRUPT = 5,       /* bitstream runs past its frame / buffer                     */
    TRPX_ERR_NO_DEVICE = 6,     /* no gfx950 device visible                                   */
    TRPX_ERR_TIMEOUT = 7        /* single-pass encoder: a bounded inter-workgroup wait expired;
                                   re-issue through the two-pass pipeline (trpx_set_encode_path) */
} trpx_status;

/* Pixel types = the reference CLI's dispatch set (src/terse.cpp:113-118). Odd = signed. */
typedef enum trpx_dtype {
    TRPX_U8 = 0, TRPX_I8 = 1, TRPX_U16 = 2, TRPX_I16 = 3, TRPX_U32 = 4, TRPX_I32 = 5,
    TRPX_F32 = 6, TRPX_F64 = 7,     /* output types of trpx_decode_convert / trpx_decode_host only */
    TRPX_U64 = 8, TRPX_I64 = 9      /* 64-bit containers (what src/terse.cpp:120-123 makes of float / double images; Terse.hpp:249
                                     * takes any integral type): correct-first generic kernels, fields of up to 64 bits, no decode
                                     * index; as an output type: values of narrower streams widened, wider ones clamped */
} trpx_dtype;

/* Device status block written by the kernels (u32 words); zeroed by each call's first node. */
enum { TRPX_STATUS_WORDS = 8 };
/*   word 0: 0 = ok, else a trpx_status (CAPACITY / CORRUPT) detected on the device
 *   word 1: prolix_bits of this call (max significant bits over all blocks, Terse.hpp:516)   */

int         trpx_abi_version(void);
const char* trpx_last_error_string(void);
size_t      trpx_dtype_size(int dtype);          /* 0 for an unknown dtype */
int         trpx_dtype_is_signed(int dtype);
int         trpx_device_count(void);             /* gfx950 devices visible to HIP; 0 = none   */

/*
 * Bytes that can hold ANY encoding of one frame: N*sizeof(T) + ceil(12*nblocks/8) + 1.
 * Replaces the (slightly short) bound of Terse.hpp:503.  Pure arithmetic.
 */
size_t trpx_worst_case_bytes(int dtype, size_t n_values, unsigned block);

/* Workspace sizes (bytes) for the device entry points below.  Pure arithmetic. */
size_t trpx_encode_workspace_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block);
size_t trpx_decode_workspace_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block);
/* How many parts trpx_decode cuts a frame of this geometry into on the current route (1: frames are not cut; large frames --
 * more than 32 K blocks -- are: Terse.hpp:360-372's serial chain is walked as many short chains, decode_part.hip).  For tests
 * and diagnostics; pure arithmetic. */
unsigned trpx_decode_parts_per_frame(int dtype, size_t n_values, size_t n_frames, unsigned block);

/*
 * Encode n_frames frames of n_values pixels each (contiguous, frame-major, DEVICE memory) into
 * one compact stack.  Replaces jpa::Terse::f_compress called once per frame from the ctor /
 * push_back (Terse.hpp:263-270, :290-302, :500-549) and the Bit_range::append_range /
 * operator|= / Bit::set primitives under it (Bit_pointer.hpp:700-730, :628-649, :490).
 *
 *   pixels        DEVICE  const T[n_frames * n_values], aligned to T (16-byte aligned with n_values % 4 == 0 is the fastest case:
 *                           every vector access then stays inside aligned lines; any n_values runs on the same kernels)
 *   out           DEVICE  uint8_t[out_capacity], 16-byte aligned; receives sum(S_f) bytes
 *   frame_offsets DEVICE  uint64_t[n_frames + 1]; [k] = first byte of frame k, [n_frames] = total
 *   status        DEVICE  uint32_t[TRPX_STATUS_WORDS], 8-byte aligned (see above); word 1 = prolix_bits
 *   workspace     DEVICE  >= trpx_encode_workspace_bytes(...), 16-byte aligned
 *
 * If the stack does not fit out_capacity, frame_offsets is still valid, status[0] =
 * TRPX_ERR_CAPACITY and the contents of out[0 .. out_capacity) are unspecified (nothing beyond
 * out_capacity is ever touched; sizes-only query: pass out = NULL, out_capacity = 0).
 * Bytes [total, align_up(total, 4)) of `out` are zeroed (stores are dword granular).
 */
int trpx_encode(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block,
                uint8_t* out, size_t out_capacity, uint64_t* frame_offsets, uint32_t* status,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Decode n_frames frames.  Replaces jpa::Terse::prolix(Iterator, frame) (Terse.hpp:352-389),
 * f_find_terse_frame (Terse.hpp:562-585, intended semantics: offset_k = sum of S_j, j < k) and
 * Bit_range::get_range / operator T() (Bit_pointer.hpp:742-792, :597-617).
 *
 *   stream_signed  the header's `signed` attribute; must equal trpx_dtype_is_signed(out_dtype)
 *                  (same-type decode is the reference's contract)
 *   terse          DEVICE const uint8_t[terse_bytes], 4-byte aligned
 *   frame_offsets  DEVICE const uint64_t[n_frames + 1], or NULL: the frames are then located by
 *                  a serial header walk on the device (the .trpx format stores no index)
 *   pixels_out     DEVICE T[n_frames * n_values], aligned to T (16-byte aligned with n_values % 4 == 0: the fastest case)
 *   status         DEVICE uint32_t[TRPX_STATUS_WORDS]; word 0 = TRPX_ERR_CORRUPT if a frame's
 *                  bits run past its end (the reference does not check; we do)
 */
int trpx_decode(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                const uint64_t* frame_offsets, size_t n_values, size_t n_frames, unsigned block,
                void* pixels_out, uint32_t* status, void* workspace, size_t workspace_bytes,
                void* stream);

/*
 * Converting decode: the output type is free and the stream's signedness is given separately.  Replaces the
 * cross-type branches of jpa::Terse::prolix / Bit_range::get_range: narrower integral output clamps to
 * numeric_limits (Bit_pointer.hpp:747-763), float / double output is exact (Terse.hpp:379-383,
 * Bit_range::next :580-587), wider integral output keeps the value -- also for an unsigned stream into a
 * signed type, where the reference sign-extends wrongly.  out_dtype: TRPX_U8 .. TRPX_F64;
 * workspace: trpx_decode_workspace_bytes().  Correct-first kernels (one lane per block).
 * trpx_decode_host picks this path by itself when out_dtype does not match the stream's signedness or is a
 * floating-point type.
 */
int trpx_decode_convert(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                        const uint64_t* frame_offsets, size_t n_values, size_t n_frames, unsigned block,
                        void* pixels_out, uint32_t* status, void* workspace, size_t workspace_bytes,
                        void* stream);

/*
 * Decode index.  The .trpx stream stores no index, so a plain trpx_decode first walks
 * every frame's header chain (serial per frame, Terse.hpp:360-372).  An encoder that is about to decode its
 * own stack again -- or a reader that decodes a stack more than once -- can keep what that walk produces:
 * width[b] (1 byte per block) and the bit offset of every 256-block group.  The index is an opaque DEVICE
 * buffer of trpx_index_bytes(); it is NOT part of the bitstream (files stay byte-identical).
 *   trpx_encode_indexed   = trpx_encode that also fills `index` (index == NULL: plain trpx_encode)
 *   trpx_build_index      fills `index` from an existing stack (the walk alone)
 *   trpx_decode_indexed   = trpx_decode without the walk; the index is validated against the frame sizes
 *                           (inconsistent -> status[0] = TRPX_ERR_CORRUPT), frame_offsets is mandatory.
 *                           Stacks of >= 1024 small frames that start inside a 128-byte line (pixel bytes per frame no
 *                           multiple of 128) take the walking decoder, whose stores are line images, and only the
 *                           header-dense frames it hands over are extracted through the index -- never slower than
 *                           trpx_decode.  That needs a few KB for the hand-over list: one grow-only device buffer per
 *                           calling thread, device and stream, allocated at the first such call (a call that is being
 *                           captured into a graph takes the plain indexed route and allocates nothing).
 */
size_t trpx_index_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block);
int trpx_encode_indexed(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block,
                        uint8_t* out, size_t out_capacity, uint64_t* frame_offsets, uint32_t* status,
                        void* index, void* workspace, size_t workspace_bytes, void* stream);
int trpx_build_index(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                     size_t n_values, size_t n_frames, unsigned block, void* index, uint32_t* status,
                     void* stream);
int trpx_decode_indexed(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                        const uint64_t* frame_offsets, const void* index, size_t n_values, size_t n_frames,
                        unsigned block, void* pixels_out, uint32_t* status, void* stream);

/*
 * Host-pointer convenience wrappers (what the C++ trpx::Terse class calls): allocate device
 * staging, copy in, run the device entry point, copy out, synchronise.  `out` must hold
 * n_frames * trpx_worst_case_bytes(); *total_bytes receives sum(S_f); frame_offsets (host,
 * n_frames + 1) and prolix_bits may be NULL.  device < 0 keeps the current HIP device.
 */
int trpx_encode_host(int dtype, const void* pixels, size_t n_values, size_t n_frames,
                     unsigned block, uint8_t* out, size_t out_capacity, size_t* total_bytes,
                     uint64_t* frame_offsets, uint32_t* prolix_bits, int device);
int trpx_decode_host(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                     const uint64_t* frame_offsets, size_t n_values, size_t n_frames,
                     unsigned block, void* pixels_out, int device);

/*
 * Locate the frames of a stack that comes without an index (a .trpx file): serial header walk on
 * the device, the intended semantics of jpa::Terse::f_find_terse_frame (Terse.hpp:562-585).
 * frame_offsets: HOST uint64_t[n_frames + 1].  max_bits = widest legal block (8, 16 or 32).
 */
int trpx_frame_offsets_host(const uint8_t* terse, size_t terse_bytes, size_t n_values, size_t n_frames,
                            unsigned block, unsigned max_bits, uint64_t* frame_offsets, int device);

/*
 * synth-v1 frame generator (SURVEY.md section 8 row d) -- bench/test utility so that the GPU
 * box regenerates exactly the pixels the oracle anchors were computed on.  dtype U16 or I32.
 */
int trpx_synth_fill(int dtype, uint64_t seed, uint64_t frame0, size_t n_frames, size_t n_values,
                    void* pixels_dev, void* stream);

/*
 * Workspaces between calls.  The single-pass encoder polls and ORs into ~1 MB of descriptor words inside the caller's
 * workspace that have to be zero when it starts; its last kernel leaves them zero again, and the library REMEMBERS
 * (device, address, geometry) of the workspaces whose last user, as far as it can know, was such a call, and then skips the
 * clearing launch in front of the next one (7 us of a 270 us call).  What it cannot see is what others do to that memory,
 * hence the rule: between two trpx_encode* calls a workspace belongs to the library.  Whoever writes into it, frees it or
 * hands its address to something else calls trpx_workspace_invalidate(workspace, workspace_bytes) first (NULL, 0: every
 * workspace the library remembers).  Entry points of this library that are given the memory for another purpose (trpx_decode
 * with the same workspace, another geometry, the two-pass pipeline, the *_host entry points' own buffers) do so by themselves;
 * a call that is captured into a HIP graph always clears and neither reads nor writes what the library remembers (replays and
 * eager calls may alternate on one workspace); the device that is current at the call is the one the kernels run on and has
 * to own the workspace; and the encoder's first tile checks a tag it left in the workspace: a workspace that is not what the
 * library remembers makes the call report TRPX_ERR_TIMEOUT in the status block, which trpx_encode_checked and the host
 * wrappers answer with the two-pass pipeline (identical stream, none of these words).  Host-side, thread safe, no device call.
 */
int trpx_workspace_invalidate(const void* workspace, size_t workspace_bytes);

/*
 * Measured memory ceilings for the benchmark's roofline block (SURVEY.md section 8 row d): one grid-stride streaming
 * kernel over `bytes` bytes, 16 bytes per lane, non-temporal.  mode 0 = read `src` (dst: a 4-byte device sink),
 * 1 = write `dst`, 2 = copy src -> dst.  Device pointers, 16-byte aligned; stream-ordered.  Bench utility, not codec.
 * 3 = write `dst` from a 2-byte aligned address (the shape of the decoder's stores when a frame starts inside a cache line;
 * the benchmark reports it next to mode 1 as the ceiling of the kernels that keep such stores).
 */
int trpx_bench_stream(int mode, const void* src, void* dst, size_t bytes, void* stream);

/*
 * Encoder selection: 0 = auto (default: the single-pass look-back encoder), 1 = always the two-pass pipeline.
 * Process-wide; also settable with the environment variable TRPX_ENCODE_PATH=twopass.
 */
int trpx_set_encode_path(int path);

/*
 * Decoder selection (trpx_decode and the entry points built on it): 0 = auto (default: the per-frame decoder for
 * frames of < 2^26 bits -- block = 12, frame offsets given, any pixel count per frame, pointers aligned to the pixel
 * type only --, the position-parallel walk + tiled extraction for larger frames, the basic kernels for other block
 * sizes and missing frame offsets), 1 = always the basic kernels, 2 = the tiled route whenever its preconditions hold,
 * 3 = the per-frame decoder whenever its preconditions hold (= auto for trpx_decode; trpx_decode_indexed, which takes
 * it from 1024 frames on, then uses it for any number), 4 = large frames by round 4's parts route, 5 = auto with the
 * header-dense frames the per-frame decoder hands over walked by the dense walk (decode_dense.hip: one speculative pass,
 * link walks, a verified write pass) instead of the fix-point rounds of decode_seg.hip.  Every route yields the same
 * pixels (Terse.hpp:352-389); the setter exists for tests and A/B measurements.
 * Process-wide; also settable with the environment variable TRPX_DECODE_PATH=basic|tiles|frames|parts|dense.
 * Frames of more than 32 K blocks are cut into parts (one walk of many short parts, then extraction through the decode
 * index it yields; a stack the walk's own vote finds header-dense -- more than one block in six with an explicit header --
 * is walked one workgroup per frame, lane per segment, instead: decode_seg.hip, k_seg_wg) unless the stack holds 768 frames
 * or more, which keep them whole on the per-frame decoder -- such a
 * stack fills the GPU by itself; TRPX_SINGLE_PART=<frames>,<blocks> moves that line for tuning runs (stacks of <frames>
 * frames and more keep frames of up to <blocks> blocks whole).
 * These three variables are the only ones the library reads; further switches exist in -DTRPX_DIAGNOSTICS builds only.
 * The tuned kernels issue 8- and 16-byte accesses at addresses that are only aligned to the pixel type (frames of any
 * pixel count): they rely on the HSA unaligned-access mode, which ROCm enables on gfx950.
 */
int trpx_set_decode_path(int path);

/*
 * trpx_encode_indexed + the one thing a stream-ordered call cannot do for itself: it waits for `stream`, reads the
 * status block and, if a look-back wait of the single-pass encoder gave up (TRPX_ERR_TIMEOUT: tiles wait for earlier
 * tiles of the same launch -- bounded to 0.25 s of wall time, never seen with in-order workgroup dispatch, but the
 * hardware does not promise it), runs the call again through the two-pass pipeline, which has no inter-tile waits and
 * writes the identical stream (f_compress is deterministic, Terse.hpp:500-549).  Returns the final status as its
 * return value (TRPX_OK = stack complete) and, if host_status is not NULL, the 8 status words.  `index` may be NULL.
 * Callers that stay asynchronous use trpx_encode and apply the same rule when they read the status block.
 */
int trpx_encode_checked(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block, uint8_t* out,
                        size_t out_capacity, uint64_t* frame_offsets, uint32_t* status, void* index, void* workspace,
                        size_t workspace_bytes, void* stream, uint32_t* host_status);

/*
 * Per-kernel timing for bench.py's roofline leg.  While enabled (per calling thread), trpx_encode /
 * trpx_decode record a hipEvent on `stream` before their first and after each of their kernels;
 * trpx_profile_read waits for the last launch and returns the elapsed ms of each stage
 * (single-pass encode: memset (k_zero_words), encode_fused, stitch; two-pass encode: tile_bits, frame_scan,
 * stack_scan, zero_edges, pack; per-frame decode: decode_frames, deferred_frames; tiled decode: walk, unpack).
 * Returns the number of stages written.  Not graph-capturable while enabled.
 */
int trpx_profile_enable(int on);
int trpx_profile_read(float* stage_ms, int capacity);

/* ---- stream-serialise surface: the ASCII header of Terse::write (Terse.hpp:454-474) and its
 * reader (Terse.hpp:485-498 over XML_element.hpp:216-224, :296-307).  Host only. ------------ */
typedef struct trpx_header {
    unsigned prolix_bits;
    int      is_signed;
    unsigned block;
    uint64_t memory_size;       /* payload bytes following the header            */
    uint64_t number_of_values;  /* per frame                                     */
    uint64_t number_of_frames;
    unsigned n_dims;            /* 0 = no `dimensions` attribute                 */
    uint64_t dims[8];
} trpx_header;

/* Writes the exact header text of Terse::write into buf (NUL-terminated); returns its length
 * (excluding the NUL) or 0 if buf_cap is too small. */
size_t trpx_header_format(const trpx_header* h, char* buf, size_t buf_cap);
/* Finds "<Terse" in data[0..len), parses the attributes (unknown ones are ignored, as the
 * reference does) and sets *payload_offset to the byte after "/>".  number_of_frames is
 * mandatory. */
int trpx_header_parse(const char* data, size_t len, trpx_header* h, size_t* payload_offset);

/* Frame index side-channel in the FILE.  The reference's file stores no frame sizes
 * (Terse.hpp:454-474), so a reader has to walk every frame's header chain just to find the next frame
 * (Terse.hpp:562-585).  trpx_header_format_indexed writes the same header with one more attribute,
 *   frame_sizes="S_0 S_1 ... S_{F-1}"      (bytes per frame, their sum = memory_size),
 * in front of number_of_frames.  The reference reader looks attributes up by name and ignores the others
 * (XML_element.hpp:296-307), so such a file stays readable by the reference tools; trpx_header_frame_sizes returns
 * the number of sizes found (0: no such attribute) and fills at most `capacity` of them.  Returns 0 / leaves buf
 * untouched if buf_cap is too small (2000 frames need ~14 KB). */
size_t trpx_header_format_indexed(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes, char* buf, size_t buf_cap);
size_t trpx_header_frame_sizes(const char* data, size_t len, uint64_t* frame_sizes, size_t capacity);

/* Group states (row f1, second half): even with the frame sizes a reader has to walk each frame's header chain before it can
 * expand anything (Terse.hpp:360-372: a block's position is only known after the header of the block before it).  The chain
 * state at every 256th block -- bit offset inside the frame and width of the block before it, packed as
 * offset | width << 40 -- lets every 256-block group be walked on its own: trpx_group_count(n_values, 12) states per frame.
 *   trpx_index_group_states          reads them off a decode index (device);
 *   trpx_index_from_group_states     rebuilds the decode index from them (device, one lane per group) and checks every
 *                                    group against its successor's state and the frame's size (TRPX_ERR_CORRUPT);
 *   trpx_group_states_host           the states of a stack in host memory (what `terse -index` writes);
 *   trpx_decode_host_grouped         trpx_decode_host with the states: no walk; falls back to trpx_decode_host when the
 *                                    output type needs conversion or the states do not fit the stream;
 *   trpx_header_format_grouped       the header text with frame_sizes="..." and group_bit_offsets="o:w o:w ..." (~0.8 % of a
 *                                    synth-v1 stack's size; ignored by the reference reader like any unknown attribute,
 *                                    XML_element.hpp:296-307), trpx_header_group_states reads the attribute back. */
size_t trpx_group_count(size_t n_values, unsigned block);
int trpx_index_group_states(const void* index, size_t n_values, size_t n_frames, unsigned block, uint64_t* group_states, void* stream);
int trpx_index_from_group_states(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                                 const uint64_t* group_states, size_t n_values, size_t n_frames, unsigned block, void* index,
                                 uint32_t* status, void* stream);
int trpx_group_states_host(const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets, size_t n_values,
                           size_t n_frames, unsigned block, unsigned max_bits, uint64_t* group_states, int device);
int trpx_decode_host_grouped(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                             const uint64_t* group_states, size_t n_values, size_t n_frames, unsigned block, void* pixels_out,
                             int device);
size_t trpx_header_format_grouped(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes, const uint64_t* group_states,
                                  size_t n_states, char* buf, size_t buf_cap);
size_t trpx_header_group_states(const char* data, size_t len, uint64_t* group_states, size_t capacity);

/* ---- host callers that work frame by frame ---------------------------------------------
 * src/prolix.cpp:69-92 expands a .trpx file with one `trpx_data.prolix(image, i)` per frame, src/terse.cpp:63-69 pushes
 * one image at a time.  The *_host entry points keep their device buffers per calling thread between calls (no
 * allocation per frame; trpx_host_release frees them).  For the expanding loop a stack object keeps the compressed stack
 * on the device: trpx_stack_open uploads it once (frame_offsets may be NULL: the frames are then located by the serial
 * walk, Terse.hpp:562-585, max_bits as in trpx_frame_offsets_host; group_states, if the file carried them, make the
 * expansion walk-free), trpx_stack_read(frame) expands a window of frames
 * starting at `frame` in ONE device call on a miss (<= 64 MB of pixels) and afterwards only copies the frame asked for
 * -- any output type of trpx_decode_host.  Not thread safe per object (the reference's prolix is not const either,
 * Terse.hpp:387-388). */
typedef struct trpx_stack trpx_stack;
int trpx_stack_open(trpx_stack** handle, int stream_signed, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                    const uint64_t* group_states /* may be NULL */, size_t n_values, size_t n_frames, unsigned block,
                    unsigned max_bits, int device);
int trpx_stack_read(trpx_stack* stack, size_t frame, int out_dtype, void* pixels_out);
void trpx_stack_close(trpx_stack* stack);
void trpx_host_release(void);

/* ---- multi-GPU (SURVEY.md section 8 row e): frames shard across GPUs, one process per GPU ----------------------
 * A frame's stream does not depend on its neighbours -- the header state resets per frame and every frame starts byte
 * aligned (Terse.hpp:502-505, :359) -- so rank r encodes its contiguous frame range with trpx_encode and keeps its
 * bytes.  What the serial encoder's cursor (Terse.hpp:502-504) would have been for the WHOLE stack is recovered by
 * the one collective of the path: trpx_gather_frame_offsets all-gathers the per-frame sizes (S_f, Terse.hpp:547) of
 * every rank over RCCL / xGMI (ncclAllGather, 8 bytes per frame) on `stream` and writes
 *   global_offsets[0 .. F_total]  byte offset of every frame of the global stack (ranks in order) + its total size,
 *   *prolix_bits                  maximum over the ranks' d_prolix_bits (Terse.hpp:516; encode_status = the status
 *                                 block trpx_encode filled, may be NULL), may be NULL,
 *   rank_base[0 .. world)         first byte of every rank's local stack inside the global one, may be NULL.
 * comm is an ncclComm_t of the caller (or one made by trpx_comm_init); n_slot >= every rank's n_local is the common
 * message size (shards may be ragged); all pointers are device pointers; stream-ordered, no host sync, no allocation.
 * The payload never crosses xGMI.  RCCL is resolved at run time (TRPX_ERR_UNSUPPORTED without it). */
size_t trpx_gather_workspace_bytes(size_t n_slot, int world);
int trpx_gather_frame_offsets(void* comm, const uint64_t* local_offsets, size_t n_local, size_t n_slot,
                              const uint32_t* encode_status, uint64_t* global_offsets, uint32_t* prolix_bits,
                              uint64_t* rank_base, void* workspace, size_t workspace_bytes, void* stream);
/* The two kernels of that call for callers that move the messages themselves (MPI, a host gather): trpx_gather_pack
 * writes this rank's message, u64[n_slot + 2] = { S_0 .. S_{n_local-1}, 0 .., n_local, prolix_bits }; trpx_gather_scan
 * turns the `world` messages, concatenated in rank order, into global_offsets / prolix_bits / rank_base as above. */
int trpx_gather_pack(const uint64_t* local_offsets, size_t n_local, size_t n_slot, const uint32_t* encode_status, uint64_t* message,
                     void* stream);
int trpx_gather_scan(const uint64_t* all_messages, int world, size_t n_slot, uint64_t* global_offsets, uint32_t* prolix_bits,
                     uint64_t* rank_base, void* stream);
/* One rank's share of a sharded stack in ONE stream-ordered call (SURVEY.md section 8 row b: trpx_encode_sharded /
 * trpx_decode_sharded).  trpx_encode_sharded = trpx_encode of this rank's n_local frames (f_compress on its share of the stack,
 * Terse.hpp:500-549; `out`, `local_offsets`, `status` exactly as trpx_encode leaves them) + trpx_gather_frame_offsets on the
 * caller's communicator (`global_offsets`, `prolix_bits`, `rank_base` as above).  gather_stream: NULL -- the gather follows the
 * encode on `stream`; another stream of the same device -- the gather runs there, ordered behind the encode by an event, so
 * that what the caller enqueues on `stream` next (the decode of its own frames needs no global offset) overlaps the collective;
 * the caller joins the two streams where it reads the table.  workspace: trpx_encode_sharded_workspace_bytes().
 * trpx_decode_sharded expands this rank's frames -- frames [first_frame, first_frame + n_local) of the global stack, whose
 * bytes it holds as `local_terse` -- from the GLOBAL offset table (prolix(it, frame) for its share, Terse.hpp:352-389; frames
 * are independent, Terse.hpp:502-505): no collective, one small kernel in front of trpx_decode.  Errors of both:
 * trpx_shard_last_error(). */
size_t trpx_encode_sharded_workspace_bytes(int dtype, size_t n_values, size_t n_local, size_t n_slot, unsigned block, int world);
int trpx_encode_sharded(void* comm, int dtype, const void* pixels, size_t n_values, size_t n_local, size_t n_slot, unsigned block,
                        uint8_t* out, size_t out_capacity, uint64_t* local_offsets, uint32_t* status, uint64_t* global_offsets,
                        uint32_t* prolix_bits, uint64_t* rank_base, void* workspace, size_t workspace_bytes, void* stream,
                        void* gather_stream);
size_t trpx_decode_sharded_workspace_bytes(int dtype, size_t n_values, size_t n_local, unsigned block);
int trpx_decode_sharded(int stream_signed, int out_dtype, const uint8_t* local_terse, size_t local_bytes, const uint64_t* global_offsets,
                        size_t first_frame, size_t n_values, size_t n_local, unsigned block, void* pixels_out, uint32_t* status,
                        void* workspace, size_t workspace_bytes, void* stream);
/* A communicator for callers that have none: rank 0 makes the 128-byte id, every rank gets it by its own means
 * (MPI, torch.distributed, a file) and calls trpx_comm_init with the GPU it will use already selected. */
int trpx_comm_unique_id(void* id128);
int trpx_comm_init(void** comm, int world, int rank, const void* id128);
int trpx_comm_destroy(void* comm);
/* What RCCL itself says about a communicator (ncclCommCount / ncclCommUserRank): a caller that reports a sharded run can state
 * how many ranks the size gather really spanned.  Either pointer may be NULL. */
int trpx_comm_info(void* comm, int* world, int* rank);
const char* trpx_shard_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TRPX_HIP_H */
