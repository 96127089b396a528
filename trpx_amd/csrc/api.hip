// C ABI of libtrpx_hip.so (declared in include/trpx_hip.h).  Thin: argument checks, workspace
// carving, kernel launches on the caller's stream.  No CPU codec lives here: without a gfx950
// device every compute entry point fails with TRPX_ERR_NO_DEVICE / TRPX_ERR_HIP.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/trpx_hip.h"
#include "encode_kernels.hpp"
#include "profile.hpp"

namespace trpx {
Profiler& profiler() {
    static thread_local Profiler p;
    return p;
}
}  // namespace trpx

namespace {

thread_local char g_err[512] = "";
thread_local bool t_force_two_pass = false;   // trpx_encode_host's retry after a look-back timeout

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(TRPX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr unsigned kMaxBlock = 4096;   // generic kernels: tile bit counts stay well inside 32 bits

bool geom_of(size_t n_values, unsigned block, trpx::FrameGeom* g) {
    if (n_values == 0 || block == 0 || block > kMaxBlock) return false;
    const uint64_t nb = (n_values + block - 1) / block;
    if (nb > 0xFFFFFFFFull) return false;
    g->n_values = n_values;
    g->n_blocks = (uint32_t)nb;
    g->n_tiles = (uint32_t)((nb + trpx::kTileBlocks - 1) / trpx::kTileBlocks);
    g->block = block;
    return true;
}

// One size check for every entry point (sizes may come straight from an untrusted .trpx header): frame count and
// tiles must fit the 31-bit grids, and n_values * n_frames * 8 (the widest element) must not wrap.
bool sizes_ok(const trpx::FrameGeom& g, size_t n_frames) {
    if (n_frames == 0 || n_frames > 0x7FFFFFFFull / g.n_tiles) return false;
    const unsigned __int128 bytes = (unsigned __int128)g.n_values * n_frames * 8;
    return bytes < ((unsigned __int128)1 << 62);
}

// workspace layouts ------------------------------------------------------------------------
struct EncWs { size_t frame_size, tile_off, tile_bits, fused, total; };
EncWs enc_ws(const trpx::FrameGeom& g, size_t n_frames) {
    EncWs w;
    const size_t tiles = n_frames * (size_t)g.n_tiles;
    w.frame_size = 0;
    w.tile_off = trpx::align_up(w.frame_size + 8 * n_frames, 16);
    w.tile_bits = trpx::align_up(w.tile_off + 8 * tiles, 16);
    w.fused = trpx::align_up(w.tile_bits + 4 * tiles, 256);          // descriptors of the single-pass encoder
    w.total = w.fused + trpx::fused_workspace_bytes(g, n_frames);
    return w;
}

// 0 = auto (single-pass encoder), 1 = force the two-pass pipeline.
// Initialised from $TRPX_ENCODE_PATH ("twopass" / "fused"), changed by trpx_set_encode_path().
int g_encode_path = [] {
    const char* e = getenv("TRPX_ENCODE_PATH");
    return e && strcmp(e, "twopass") == 0 ? 1 : 0;
}();
// Decode route: 0 = auto, 1 = basic (decode.hip), 2 = tiled (position-parallel walk + k_unpack_tiles), 3 = per-frame
// decoder for any number of frames, 4 = large frames by round 4's parts route (two walks) instead of the index route.
// Initialised from $TRPX_DECODE_PATH ("basic" / "tiles" / "frames" / "parts"), changed by trpx_set_decode_path().  A forced
// route is still subject to its preconditions (alignment, block = 12, frame size).
int g_decode_path = [] {
    const char* e = getenv("TRPX_DECODE_PATH");
    if (!e) return 0;
    return strcmp(e, "basic") == 0 ? 1 : (strcmp(e, "tiles") == 0 || strcmp(e, "seg") == 0) ? 2 : strcmp(e, "frames") == 0 ? 3
           : strcmp(e, "parts") == 0 ? 4 : strcmp(e, "dense") == 0 ? (trpx::set_dense_route(true), 0) : 0;
}();
// $TRPX_SINGLE_PART = "frames,blocks": stacks of that many frames and more keep frames of up to that many blocks on the per-frame
// route (encode_kernels.hpp: single_part_blocks; tuning runs -- the built-in rule otherwise)
[[maybe_unused]] static const int g_single_part_env = [] {
    const char* e = getenv("TRPX_SINGLE_PART");
    unsigned f = 0, b = 0;
    if (e && sscanf(e, "%u,%u", &f, &b) == 2) trpx::set_single_part_rule(f, b);
    return 0;
}();
struct IdxLayout { size_t group_off, widths, seg, defer, parts, part_ws, total; };
IdxLayout idx_layout(const trpx::FrameGeom& g, size_t n_frames, size_t pixel_bytes = 4) {   // (pixel_bytes: only the scratch behind the index proper depends on it)
    IdxLayout l;
    l.group_off = 0;
    l.widths = trpx::align_up(8 * n_frames * (size_t)g.n_tiles, 16);
    l.seg = trpx::align_up(l.widths + n_frames * (size_t)g.n_blocks, 256);   // scratch of trpx_build_index's walk
    l.defer = l.seg + trpx::seg_workspace_bytes(g, n_frames);                 // list of the frames the per-frame walker hands over
    l.parts = l.defer + trpx::defer_bytes(n_frames);                          // large frames: the index route's part table and scratch (decode_part.hip)
    const size_t P = trpx::chain_parts_per_frame(g, n_frames, pixel_bytes);
    l.part_ws = l.parts + (P > 1 ? trpx::align_up(sizeof(trpx::PartDesc) * n_frames * P, 256) : 0);
    l.total = l.part_ws + trpx::chain_workspace_bytes(g, n_frames, pixel_bytes);
    return l;
}
struct DecWs { size_t walk_offsets, tile_off, widths, seg, defer, parts, part_ws, total; };
DecWs dec_ws(const trpx::FrameGeom& g, size_t n_frames, size_t pixel_bytes) {
    DecWs w;
    const size_t tiles = n_frames * (size_t)g.n_tiles;
    w.walk_offsets = 0;
    w.tile_off = trpx::align_up(w.walk_offsets + 8 * (n_frames + 1), 16);
    w.widths = trpx::align_up(w.tile_off + 8 * tiles, 16);
    w.seg = trpx::align_up(w.widths + n_frames * (size_t)g.n_blocks, 256);
    w.defer = w.seg + trpx::seg_workspace_bytes(g, n_frames);
    w.parts = w.defer + trpx::defer_bytes(n_frames);                            // large frames on the per-frame route: the part table
    // (two routes for large frames share these two areas: the index route -- many short parts -- and round 4's parts route)
    const size_t P = std::max<size_t>(trpx::parts_per_frame(g, n_frames), trpx::chain_parts_per_frame(g, n_frames, pixel_bytes));
    w.part_ws = w.parts + (P > 1 ? trpx::align_up(sizeof(trpx::PartDesc) * n_frames * P, 256) : 0);
    w.total = w.part_ws + std::max(trpx::part_workspace_bytes(g, n_frames), trpx::chain_workspace_bytes(g, n_frames, pixel_bytes));
    return w;
}

// trpx_decode_indexed's hand-over list (see there): per calling thread, one buffer per (device, stream), grow-only, freed with the thread.
struct IdxScratch {
    struct Slot { int dev; hipStream_t st; void* p; size_t bytes; };
    std::vector<Slot> slots;
    ~IdxScratch() { for (auto& s : slots) if (s.p) (void)hipFree(s.p); }
};
void* indexed_scratch(size_t bytes, hipStream_t st) {
    static thread_local IdxScratch t;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    // never under stream capture: a graph would keep the buffer's address, and a later, larger call frees it
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    for (auto& s : t.slots)
        if (s.dev == dev && s.st == st) {
            if (s.bytes >= bytes) return s.p;
            if (hipStreamSynchronize(st) != hipSuccess) return nullptr;        // (calls that still use the smaller buffer)
            (void)hipFree(s.p);
            s.p = nullptr; s.bytes = 0;
            if (hipMalloc(&s.p, bytes) != hipSuccess) { s.p = nullptr; (void)hipGetLastError(); return nullptr; }
            s.bytes = bytes;
            return s.p;
        }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    t.slots.push_back({dev, st, p, bytes});
    return p;
}

}  // namespace

extern "C" {

int trpx_abi_version(void) { return TRPX_ABI_VERSION; }
const char* trpx_last_error_string(void) { return g_err; }

size_t trpx_dtype_size(int dtype) {
    switch (dtype) {
    case TRPX_U8: case TRPX_I8: return 1;
    case TRPX_U16: case TRPX_I16: return 2;
    case TRPX_U32: case TRPX_I32: return 4;
    case TRPX_U64: case TRPX_I64: return 8;
    }
    return 0;
}
int trpx_dtype_is_signed(int dtype) { return dtype >= 0 && dtype <= TRPX_I32 ? (dtype & 1) : (dtype == TRPX_I64 ? 1 : 0); }
static bool is64(int dtype) { return dtype == TRPX_U64 || dtype == TRPX_I64; }

int trpx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t trpx_worst_case_bytes(int dtype, size_t n_values, unsigned block) {
    const size_t es = trpx_dtype_size(dtype);
    if (!es || !block) return 0;
    const size_t nblocks = (n_values + block - 1) / block;
    return n_values * es + (12 * nblocks + 7) / 8 + 1;
}

size_t trpx_encode_workspace_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(dtype) || !geom_of(n_values, block, &g)) return 0;
    return enc_ws(g, n_frames).total;
}
size_t trpx_decode_workspace_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(dtype) || !geom_of(n_values, block, &g)) return 0;
    return dec_ws(g, n_frames, trpx_dtype_size(dtype)).total;
}

unsigned trpx_decode_parts_per_frame(int dtype, size_t n_values, size_t n_frames, unsigned block) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(dtype) || is64(dtype) || block != (unsigned)trpx::kBlock || !geom_of(n_values, block, &g)) return 1;
    const bool bits32 = 8 * (uint64_t)trpx_worst_case_bytes(dtype, n_values, block) < 0xF0000000ull;
    return g_decode_path != 4 && bits32 ? trpx::chain_parts_per_frame(g, n_frames, trpx_dtype_size(dtype)) : trpx::parts_per_frame(g, n_frames);
}

static int build_index_impl(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                            size_t n_values, size_t n_frames, unsigned block, void* index, uint32_t* status,
                            bool clear_status, void* stream);

size_t trpx_index_bytes(int dtype, size_t n_values, size_t n_frames, unsigned block) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(dtype) || !geom_of(n_values, block, &g)) return 0;
    return idx_layout(g, n_frames, trpx_dtype_size(dtype)).total;
}

int trpx_encode(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block, uint8_t* out,
                size_t out_capacity, uint64_t* frame_offsets, uint32_t* status, void* workspace,
                size_t workspace_bytes, void* stream) {
    return trpx_encode_indexed(dtype, pixels, n_values, n_frames, block, out, out_capacity, frame_offsets, status,
                               nullptr, workspace, workspace_bytes, stream);
}

int trpx_encode_indexed(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block, uint8_t* out,
                        size_t out_capacity, uint64_t* frame_offsets, uint32_t* status, void* index,
                        void* workspace, size_t workspace_bytes, void* stream) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(dtype)) return fail(TRPX_ERR_INVALID_ARG, "trpx_encode: unknown dtype %d", dtype);
    if (block == 0 || block > kMaxBlock)
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_encode: block=%u (supported: 1..%u; 12 is the tuned default, Terse.hpp:264)", block, kMaxBlock);
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames))
        return fail(TRPX_ERR_INVALID_ARG, "trpx_encode: bad sizes n_values=%zu n_frames=%zu", n_values, n_frames);
    if (!pixels || !frame_offsets || !status || !workspace || (!out && out_capacity))
        return fail(TRPX_ERR_INVALID_ARG, "trpx_encode: null pointer");
    if (((uintptr_t)out | (uintptr_t)workspace | (uintptr_t)frame_offsets) % 8 || (uintptr_t)out % 16 ||
        (uintptr_t)status % 8 || (uintptr_t)pixels % trpx_dtype_size(dtype))
        return fail(TRPX_ERR_INVALID_ARG, "trpx_encode: misaligned pointer (out needs 16 B, workspace/offsets 8 B)");
    const EncWs w = enc_ws(g, n_frames);
    if (workspace_bytes < w.total)
        return fail(TRPX_ERR_CAPACITY, "trpx_encode: workspace %zu < %zu", workspace_bytes, w.total);

    trpx::EncodeArgs a;
    a.pixels = pixels;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.out = out;
    a.out_capacity = out_capacity;
    a.frame_offsets = frame_offsets;
    a.status = status;
    char* ws = static_cast<char*>(workspace);
    a.frame_size = reinterpret_cast<uint64_t*>(ws + w.frame_size);
    a.tile_off = reinterpret_cast<uint64_t*>(ws + w.tile_off);
    a.tile_bits = reinterpret_cast<uint32_t*>(ws + w.tile_bits);
    if ((uintptr_t)index % 16) return fail(TRPX_ERR_INVALID_ARG, "trpx_encode_indexed: index must be 16-byte aligned");
    const IdxLayout il = idx_layout(g, n_frames, trpx_dtype_size(dtype));
    a.idx_group_off = index ? reinterpret_cast<uint64_t*>(static_cast<char*>(index) + il.group_off) : nullptr;
    a.idx_widths = index ? reinterpret_cast<uint8_t*>(static_cast<char*>(index) + il.widths) : nullptr;
    if (block != (unsigned)trpx::kBlock || is64(dtype)) {  // any other block size, 64-bit containers: generic (correct-first) kernels
        if (index) return fail(TRPX_ERR_UNSUPPORTED, "trpx_encode_indexed: the decode index needs block=12 and pixels of <= 32 bits");
        trpx::fused_ws_forget(workspace, workspace_bytes);
        HIP_TRY(trpx::launch_encode_generic(dtype, a, static_cast<hipStream_t>(stream)));
        return TRPX_OK;
    }
    // (frames need not be vector aligned -- most detectors' pixel counts are not multiples of 4: 1030 x 1065, 2463 x 2527 --: the
    // kernels' 16-byte accesses only need what the hardware needs, which in HSA's unaligned access mode is nothing)
    const bool vec_ok = (uint64_t)g.n_blocks * 396 < (1ull << 40);       // frame bits fit the fused encoder's 40-bit accumulator
    if (g_encode_path == 0 && !t_force_two_pass && vec_ok) {
        trpx::fused_ws_forget(workspace, workspace_bytes, ws + w.fused);   // (a clean descriptor block of another geometry in this memory is no longer)
        HIP_TRY(trpx::launch_encode_fused(dtype, a, ws + w.fused, static_cast<hipStream_t>(stream)));
    } else {
        trpx::fused_ws_forget(workspace, workspace_bytes);
        HIP_TRY(trpx::launch_encode(dtype, a, static_cast<hipStream_t>(stream)));
        if (index && out)   // the two-pass pipeline does not emit the index: build it from the stream it just wrote
            return build_index_impl(dtype, out, out_capacity, frame_offsets, n_values, n_frames, block, index, status, false, stream);
    }
    return TRPX_OK;
}

int trpx_encode_checked(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block, uint8_t* out,
                        size_t out_capacity, uint64_t* frame_offsets, uint32_t* status, void* index, void* workspace,
                        size_t workspace_bytes, void* stream, uint32_t* host_status) {
    uint32_t st[TRPX_STATUS_WORDS] = {0};
    for (int attempt = 0; attempt < 2; ++attempt) {
        t_force_two_pass = attempt == 1;                     // (this thread's next call only)
        const int rc = trpx_encode_indexed(dtype, pixels, n_values, n_frames, block, out, out_capacity, frame_offsets, status,
                                           index, workspace, workspace_bytes, stream);
        t_force_two_pass = false;
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(st, status, sizeof st, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
        HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));   // the caller's stream alone, not the device
        if (st[0] != TRPX_ERR_TIMEOUT) break;                // a look-back wait gave up: the two-pass pipeline has no waits
    }
    if (host_status) memcpy(host_status, st, sizeof st);
    if (st[0]) return fail((int)st[0], "trpx_encode_checked: device status %u", st[0]);
    return TRPX_OK;
}

int trpx_decode(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                const uint64_t* frame_offsets, size_t n_values, size_t n_frames, unsigned block, void* pixels_out,
                uint32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(out_dtype)) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode: unknown dtype %d", out_dtype);
    if (is64(out_dtype))                                   // 64-bit containers: the converting decoder (fields of up to 64 bits)
        return trpx_decode_convert(stream_signed, out_dtype, terse, terse_bytes, frame_offsets, n_values, n_frames, block, pixels_out,
                                   status, workspace, workspace_bytes, stream);
    if (block == 0 || block > kMaxBlock)
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode: block=%u (supported: 1..%u)", block, kMaxBlock);
    if ((stream_signed != 0) != (trpx_dtype_is_signed(out_dtype) != 0))
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode: stream signed=%d into dtype %d: only same-signedness decode "
                    "is defined by the reference (Terse.hpp:356-357)", stream_signed, out_dtype);
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || terse_bytes == 0)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode: bad sizes");
    if (!terse || !pixels_out || !status || !workspace) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode: null pointer");
    if ((uintptr_t)terse % 4 || (uintptr_t)workspace % 8 || (uintptr_t)frame_offsets % 8 || (uintptr_t)status % 8 ||
        (uintptr_t)pixels_out % trpx_dtype_size(out_dtype))
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode: misaligned pointer (terse needs 4 B, workspace 8 B)");
    const DecWs w = dec_ws(g, n_frames, trpx_dtype_size(out_dtype));
    if (workspace_bytes < w.total)
        return fail(TRPX_ERR_CAPACITY, "trpx_decode: workspace %zu < %zu", workspace_bytes, w.total);

    trpx::fused_ws_forget(workspace, workspace_bytes);      // (an encoder's clean descriptor words in this memory are about to be overwritten)
    trpx::DecodeArgs a{};
    a.terse = terse;
    a.terse_bytes = terse_bytes;
    a.frame_offsets = frame_offsets;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.pixels_out = pixels_out;
    a.status = status;
    char* ws = static_cast<char*>(workspace);
    a.walk_offsets = reinterpret_cast<uint64_t*>(ws + w.walk_offsets);
    a.tile_off = reinterpret_cast<uint64_t*>(ws + w.tile_off);
    a.widths = reinterpret_cast<uint8_t*>(ws + w.widths);
    a.seg_ws = ws + w.seg;
#ifdef TRPX_DIAGNOSTICS
    static const bool no_defer = getenv("TRPX_NO_DEFER") != nullptr;          // (diagnostic build: the per-frame decoder keeps every frame)
#else
    constexpr bool no_defer = false;
#endif
    a.defer = no_defer ? nullptr : reinterpret_cast<uint32_t*>(ws + w.defer + trpx::kDeferFront);
    const int route = g_decode_path;                                          // trpx_set_decode_path / $TRPX_DECODE_PATH
    const bool basic = route == 1, force_tiles = route == 2;
    const bool bits32 = 8 * (uint64_t)trpx_worst_case_bytes(out_dtype, n_values, block) < 0xF0000000ull;   // 32-bit frame-relative bit offsets
    const bool fast_ok = frame_offsets && !basic && bits32 && block == (unsigned)trpx::kBlock;
    // frames whose worst case fits 2^26 bits: one workgroup per frame, the walk and the extraction overlap inside it -- whatever
    // the number of frames (since the round-3 walker a single 512^2 frame takes 0.10 ms this way against 0.24 ms through the
    // position-parallel walk + tiled extraction, eight 1024^2 frames 0.37 against 0.73 ms); larger frames: the tiled kernels
    // (the per-frame decoder packs a block's bit position with its width into 32 bits: frames of < 2^26 bits less the
    // walker's ring offset and one step's overshoot)
    const bool frame26 = 8 * (uint64_t)trpx_worst_case_bytes(out_dtype, n_values, block) + (1u << 17) < (1ull << 26);
    // larger frames, or frames of more than 32 K blocks: cut into parts of the size of a 512 x 512 frame first (decode_part.hip);
    // a part's positions are relative to its own first bit, so the 2^26 limit applies to the part
    a.chain = route != 4 && route != 2 && bits32;                              // (frame-relative 32-bit positions; a forced tiled route walks large frames position-parallel, as build_index_impl does)
    a.parts_per_frame = a.chain ? trpx::chain_parts_per_frame(g, n_frames, trpx_dtype_size(out_dtype)) : trpx::parts_per_frame(g, n_frames);
    const bool parts_ok = a.parts_per_frame > 1u && n_frames * (uint64_t)a.parts_per_frame < 0x7FFFFFFFull && a.defer;
    if (parts_ok) {
        a.parts = reinterpret_cast<trpx::PartDesc*>(ws + w.parts);
        a.part_ws = ws + w.part_ws;
    } else a.parts_per_frame = 1;
    if (fast_ok && (parts_ok || (frame26 && trpx::parts_per_frame(g, n_frames) == 1u)) && !force_tiles)
        HIP_TRY(trpx::launch_decode_frames(out_dtype, a, static_cast<hipStream_t>(stream)));
    else if (fast_ok)
        HIP_TRY(trpx::launch_decode_fast(out_dtype, a, false, static_cast<hipStream_t>(stream)));
    else
        HIP_TRY(trpx::launch_decode(out_dtype, a, frame_offsets != nullptr, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

static int build_index_impl(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                            size_t n_values, size_t n_frames, unsigned block, void* index, uint32_t* status,
                            bool clear_status, void* stream) {
    trpx::FrameGeom g;
    if (block != (unsigned)trpx::kBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_build_index: the decode index needs block=12");
    if (is64(dtype)) return fail(TRPX_ERR_UNSUPPORTED, "trpx_build_index: no decode index for 64-bit containers (generic kernels)");
    if (!trpx_dtype_size(dtype) || !geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || !terse_bytes)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_build_index: bad dtype/sizes");
    if (!terse || !frame_offsets || !index || !status) return fail(TRPX_ERR_INVALID_ARG, "trpx_build_index: null pointer");
    if ((uintptr_t)terse % 4 || (uintptr_t)index % 16 || (uintptr_t)frame_offsets % 8 || (uintptr_t)status % 8)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_build_index: misaligned pointer");
    if (8 * (uint64_t)trpx_worst_case_bytes(dtype, n_values, block) >= 0xF0000000ull)
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_build_index: frames of >= 2^32 bits");
    const IdxLayout il = idx_layout(g, n_frames, trpx_dtype_size(dtype));
    trpx::DecodeArgs a{};
    a.terse = terse;
    a.terse_bytes = terse_bytes;
    a.frame_offsets = frame_offsets;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.status = status;
    a.tile_off = reinterpret_cast<uint64_t*>(static_cast<char*>(index) + il.group_off);
    a.widths = reinterpret_cast<uint8_t*>(static_cast<char*>(index) + il.widths);
    a.seg_ws = static_cast<char*>(index) + il.seg;
    a.defer = reinterpret_cast<uint32_t*>(static_cast<char*>(index) + il.defer + trpx::kDeferFront);
    // frames of < 2^26 bits: the per-frame decoder's walker writes the index (the conditions of trpx_decode's per-frame route)
    a.index_per_frame = 8 * (uint64_t)trpx_worst_case_bytes(dtype, n_values, block) + (1u << 17) < (1ull << 26) && g_decode_path != 2;
    // frames of more than 32 K blocks: the index route's one walk of many short parts (decode_part.hip), unless the tiled route is forced
    a.parts_per_frame = trpx::chain_parts_per_frame(g, n_frames, trpx_dtype_size(dtype));
    a.chain = a.parts_per_frame > 1u && n_frames * (uint64_t)a.parts_per_frame < 0x7FFFFFFFull && g_decode_path != 2 && g_decode_path != 4;
    if (a.chain) {
        a.parts = reinterpret_cast<trpx::PartDesc*>(static_cast<char*>(index) + il.parts);
        a.part_ws = static_cast<char*>(index) + il.part_ws;
    } else a.parts_per_frame = 1;
    HIP_TRY(trpx::launch_walk_only(a, (uint32_t)(8 * trpx_dtype_size(dtype)), clear_status, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

int trpx_build_index(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                     size_t n_values, size_t n_frames, unsigned block, void* index, uint32_t* status, void* stream) {
    return build_index_impl(dtype, terse, terse_bytes, frame_offsets, n_values, n_frames, block, index, status, true, stream);
}

int trpx_decode_indexed(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                        const uint64_t* frame_offsets, const void* index, size_t n_values, size_t n_frames,
                        unsigned block, void* pixels_out, uint32_t* status, void* stream) {
    trpx::FrameGeom g;
    if (!trpx_dtype_size(out_dtype)) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_indexed: unknown dtype %d", out_dtype);
    if (block != (unsigned)trpx::kBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode_indexed: block=%u", block);
    if (is64(out_dtype)) return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode_indexed: no decode index for 64-bit containers (generic kernels)");
    if ((stream_signed != 0) != (trpx_dtype_is_signed(out_dtype) != 0))
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode_indexed: only same-signedness decode (Terse.hpp:356-357)");
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || terse_bytes == 0)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_indexed: bad sizes");
    if (!terse || !pixels_out || !status || !index || !frame_offsets)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_indexed: null pointer");
    if ((uintptr_t)terse % 4 || (uintptr_t)index % 16 || (uintptr_t)frame_offsets % 8 ||
        (uintptr_t)pixels_out % trpx_dtype_size(out_dtype) || (uintptr_t)status % 8)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_indexed: misaligned pointer (terse needs 4 B, index 16 B)");
    const IdxLayout il = idx_layout(g, n_frames, trpx_dtype_size(out_dtype));
    trpx::DecodeArgs a{};
    a.terse = terse;
    a.terse_bytes = terse_bytes;
    a.frame_offsets = frame_offsets;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.pixels_out = pixels_out;
    a.status = status;
    a.tile_off = reinterpret_cast<uint64_t*>(const_cast<char*>(static_cast<const char*>(index)) + il.group_off);
    a.widths = reinterpret_cast<uint8_t*>(const_cast<char*>(static_cast<const char*>(index)) + il.widths);
    // a thousand frames and more: one workgroup per frame; fewer: the tiled kernel spreads a frame's tiles over the whole GPU
    // (512^2 u16 frames, tiled / per-frame: 128 frames 0.023 / 0.121 ms, 600 frames 0.092 / 0.137, 1000 frames 0.148 / 0.142, 2000
    // frames 0.30 / 0.21 -- both scale with the blocks per frame, so the frame count alone decides)
    const bool frame26 = 8 * (uint64_t)trpx_worst_case_bytes(out_dtype, n_values, block) + (1u << 17) < (1ull << 26);
    const bool per_frame = frame26 && g_decode_path != 2 && (g_decode_path == 3 || n_frames >= 1024);
    // Frames that start inside a cache line (513 x 511 u16: most detectors): the indexed extraction's 16-byte stores are then
    // misaligned and it LOSES to the walking decoder, whose extraction waves write line images (2000 frames: 0.33 against 0.27 ms).
    // Such stacks take the walking decoder, and the frames it hands over -- header-dense ones, where the walk is what costs --
    // are extracted with the caller's index instead of being walked position-parallel: never slower than trpx_decode.  The
    // hand-over list needs a few KB of scratch, which this entry point has no argument for: one grow-only buffer per calling
    // thread, device and stream (calls on one stream are ordered; a call that is being captured into a graph takes the plain indexed
    // route: a graph would keep the buffer's address, and a later, larger call frees it).
    const bool misaligned = (g.n_values * trpx_dtype_size(out_dtype)) % 128u != 0u || (uintptr_t)pixels_out % 128u != 0u;
    if (per_frame && misaligned && g_decode_path == 0 && trpx::parts_per_frame(g, n_frames) == 1u) {
        void* scratch = indexed_scratch(trpx::defer_bytes(n_frames), static_cast<hipStream_t>(stream));
        if (scratch) {
            a.defer = reinterpret_cast<uint32_t*>(static_cast<char*>(scratch) + trpx::kDeferFront);
            a.seg_ws = scratch;                                               // (not used: no walk of the listed frames)
            a.index_given = true;
            HIP_TRY(trpx::launch_decode_frames(out_dtype, a, static_cast<hipStream_t>(stream)));
            return TRPX_OK;
        }
    }
    HIP_TRY(trpx::launch_decode_fast(out_dtype, a, true, static_cast<hipStream_t>(stream), per_frame));
    return TRPX_OK;
}

size_t trpx_group_count(size_t n_values, unsigned block) {
    trpx::FrameGeom g;
    if (block != (unsigned)trpx::kBlock || !geom_of(n_values, block, &g)) return 0;
    return g.n_tiles;
}

int trpx_index_group_states(const void* index, size_t n_values, size_t n_frames, unsigned block, uint64_t* states, void* stream) {
    trpx::FrameGeom g;
    if (block != (unsigned)trpx::kBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_index_group_states: the decode index needs block=12");
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames)) return fail(TRPX_ERR_INVALID_ARG, "trpx_index_group_states: bad sizes");
    if (!index || !states || (uintptr_t)index % 16 || (uintptr_t)states % 8) return fail(TRPX_ERR_INVALID_ARG, "trpx_index_group_states: null / misaligned pointer");
    const IdxLayout il = idx_layout(g, n_frames);
    trpx::DecodeArgs a{};
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.tile_off = reinterpret_cast<uint64_t*>(const_cast<char*>(static_cast<const char*>(index)) + il.group_off);
    a.widths = reinterpret_cast<uint8_t*>(const_cast<char*>(static_cast<const char*>(index)) + il.widths);
    HIP_TRY(trpx::launch_index_group_states(a, states, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

int trpx_index_from_group_states(int dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                                 const uint64_t* states, size_t n_values, size_t n_frames, unsigned block, void* index,
                                 uint32_t* status, void* stream) {
    trpx::FrameGeom g;
    if (block != (unsigned)trpx::kBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_index_from_group_states: the decode index needs block=12");
    if (is64(dtype)) return fail(TRPX_ERR_UNSUPPORTED, "trpx_index_from_group_states: no decode index for 64-bit containers (generic kernels)");
    if (!trpx_dtype_size(dtype) || !geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || !terse_bytes)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_index_from_group_states: bad dtype/sizes");
    if (!terse || !frame_offsets || !states || !index || !status) return fail(TRPX_ERR_INVALID_ARG, "trpx_index_from_group_states: null pointer");
    if ((uintptr_t)terse % 4 || (uintptr_t)index % 16 || ((uintptr_t)frame_offsets | (uintptr_t)states | (uintptr_t)status) % 8)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_index_from_group_states: misaligned pointer");
    if (8 * (uint64_t)trpx_worst_case_bytes(dtype, n_values, block) >= (1ull << 40))
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_index_from_group_states: frames of >= 2^40 bits");
    const IdxLayout il = idx_layout(g, n_frames);
    trpx::DecodeArgs a{};
    a.terse = terse;
    a.terse_bytes = terse_bytes;
    a.frame_offsets = frame_offsets;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.status = status;
    a.tile_off = reinterpret_cast<uint64_t*>(static_cast<char*>(index) + il.group_off);
    a.widths = reinterpret_cast<uint8_t*>(static_cast<char*>(index) + il.widths);
    HIP_TRY(trpx::launch_walk_groups(a, (uint32_t)(8 * trpx_dtype_size(dtype)), states, true, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

int trpx_decode_convert(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                        const uint64_t* frame_offsets, size_t n_values, size_t n_frames, unsigned block, void* pixels_out,
                        uint32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    trpx::FrameGeom g;
    const size_t es = out_dtype == TRPX_F32 ? 4 : out_dtype == TRPX_F64 ? 8 : trpx_dtype_size(out_dtype);
    if (!es) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_convert: unknown dtype %d", out_dtype);
    if (block == 0 || block > kMaxBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_decode_convert: block=%u", block);
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || terse_bytes == 0)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_convert: bad sizes");
    if (!terse || !pixels_out || !status || !workspace) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_convert: null pointer");
    if ((uintptr_t)terse % 4 || (uintptr_t)workspace % 8 || (uintptr_t)frame_offsets % 8 || (uintptr_t)status % 8 ||
        (uintptr_t)pixels_out % es)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_convert: misaligned pointer");
    const DecWs w = dec_ws(g, n_frames, es);
    if (workspace_bytes < w.total) return fail(TRPX_ERR_CAPACITY, "trpx_decode_convert: workspace %zu < %zu", workspace_bytes, w.total);
    trpx::fused_ws_forget(workspace, workspace_bytes);
    trpx::DecodeArgs a{};
    a.terse = terse;
    a.terse_bytes = terse_bytes;
    a.frame_offsets = frame_offsets;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.pixels_out = pixels_out;
    a.status = status;
    char* ws = static_cast<char*>(workspace);
    a.walk_offsets = reinterpret_cast<uint64_t*>(ws + w.walk_offsets);
    a.tile_off = reinterpret_cast<uint64_t*>(ws + w.tile_off);
    a.widths = reinterpret_cast<uint8_t*>(ws + w.widths);
    a.seg_ws = ws + w.seg;
    HIP_TRY(trpx::launch_decode_convert(out_dtype, a, stream_signed != 0, frame_offsets != nullptr, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

int trpx_workspace_invalidate(const void* workspace, size_t workspace_bytes) {
    trpx::fused_ws_forget(workspace, workspace_bytes);
    return TRPX_OK;
}

int trpx_set_encode_path(int path) {
    if (path != 0 && path != 1) return fail(TRPX_ERR_INVALID_ARG, "trpx_set_encode_path: 0 = auto, 1 = two-pass");
    g_encode_path = path;
    return TRPX_OK;
}

int trpx_set_decode_path(int path) {
    if (path < 0 || path > 5) return fail(TRPX_ERR_INVALID_ARG, "trpx_set_decode_path: 0 = auto, 1 = basic, 2 = tiled, 3 = per-frame, 4 = parts route for large frames, 5 = auto with the dense walk for listed frames");
    trpx::set_dense_route(path == 5);
    g_decode_path = path == 5 ? 0 : path;
    return TRPX_OK;
}

int trpx_profile_enable(int on) {
    trpx::profiler().enabled = on != 0;
    return TRPX_OK;
}
int trpx_profile_read(float* stage_ms, int capacity) {
    if (!stage_ms || capacity <= 0) return 0;
    return trpx::profiler().read(stage_ms, capacity);
}

int trpx_synth_fill(int dtype, uint64_t seed, uint64_t frame0, size_t n_frames, size_t n_values, void* pixels_dev,
                    void* stream) {
    if (dtype != TRPX_U16 && dtype != TRPX_I32)
        return fail(TRPX_ERR_UNSUPPORTED, "trpx_synth_fill: synth-v1 is defined for U16 and I32");
    if (!pixels_dev) return fail(TRPX_ERR_INVALID_ARG, "trpx_synth_fill: null pointer");
    HIP_TRY(trpx::launch_synth(dtype, seed, frame0, n_frames, n_values, pixels_dev, static_cast<hipStream_t>(stream)));
    return TRPX_OK;
}

// ---- host-pointer convenience wrappers ---------------------------------------------------
// The callers of the reference's API work frame by frame from host memory (src/terse.cpp:63-69 pushes one image at a
// time, src/prolix.cpp:69-92 expands one frame at a time): a device allocation per call would cost more than the
// codec.  Every thread keeps ONE grow-only set of device buffers per role (freed by trpx_host_release or at exit of the
// process); the host wrappers carve their pixels / stream / offsets / status / workspace from it.
}  // extern "C"
namespace {
struct Arena {
    enum { kPixels, kStream, kOffsets, kStatus, kWorkspace, kSlots };
    void* p[kSlots] = {};
    size_t cap[kSlots] = {};
    int device = -1;
    hipStream_t stream = nullptr;                            // this thread's private, non-blocking stream: the host wrappers
                                                             // order their copies and kernels on it and wait for IT alone
    hipError_t on_device() {
        int dev = 0;
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev != device) { release(); device = dev; }
        return hipSuccess;
    }
    // (the workspace slot is the one piece of memory trpx_encode registers as a clean descriptor block: every OTHER user of
    // it -- an index, a walk's scratch -- makes the library forget that first; `encoder` = trpx_encode_host itself)
    hipError_t get(int slot, size_t n, void** out, bool encoder = false) {
        hipError_t e = on_device();
        if (e != hipSuccess) return e;
        if (slot == kWorkspace && !encoder && p[slot]) trpx::fused_ws_forget(p[slot], cap[slot]);
        if (cap[slot] < n) {
            if (p[slot] && slot == kWorkspace) trpx::fused_ws_forget(p[slot], cap[slot]);
            if (p[slot]) (void)hipFree(p[slot]);
            p[slot] = nullptr; cap[slot] = 0;
            const size_t want = n + n / 8 + 256;             // a little head room: stacks of slightly different sizes reuse it
            e = hipMalloc(&p[slot], want);
            if (e != hipSuccess) return e;
            cap[slot] = want;
        }
        *out = p[slot];
        return hipSuccess;
    }
    hipError_t get_stream(hipStream_t* out) {
        hipError_t e = on_device();
        if (e != hipSuccess) return e;
        if (!stream && (e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) return e;
        *out = stream;
        return hipSuccess;
    }
    void release() {
        if (p[kWorkspace]) trpx::fused_ws_forget(p[kWorkspace], cap[kWorkspace]);
        for (int i = 0; i < kSlots; ++i) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; }
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
    ~Arena() { release(); }                                  // thread exit: the buffers go back (errors of a runtime that is already down are ignored)
};
Arena& arena() {
    static thread_local Arena a;
    return a;
}
// copy on the caller's private stream and wait for that stream (never for the device)
hipError_t copy_sync(hipStream_t hs, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, hs);
    return e != hipSuccess ? e : hipStreamSynchronize(hs);
}
}  // namespace
extern "C" {

void trpx_host_release(void) { arena().release(); }

int trpx_encode_host(int dtype, const void* pixels, size_t n_values, size_t n_frames, unsigned block, uint8_t* out,
                     size_t out_capacity, size_t* total_bytes, uint64_t* frame_offsets, uint32_t* prolix_bits,
                     int device) {
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_encode_host: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    const size_t es = trpx_dtype_size(dtype);
    if (!es || !pixels || !out || !total_bytes) return fail(TRPX_ERR_INVALID_ARG, "trpx_encode_host: bad argument");
    {
        trpx::FrameGeom g0;
        if (!geom_of(n_values, block, &g0)) return fail(block == 0 || block > kMaxBlock ? TRPX_ERR_UNSUPPORTED : TRPX_ERR_INVALID_ARG, "trpx_encode_host: unsupported sizes/block (block=%u)", block);
        if (!sizes_ok(g0, n_frames)) return fail(TRPX_ERR_INVALID_ARG, "trpx_encode_host: bad sizes n_values=%zu n_frames=%zu", n_values, n_frames);
    }
    const size_t in_bytes = n_values * n_frames * es;
    const size_t cap = trpx::align_up(n_frames * trpx_worst_case_bytes(dtype, n_values, block), 16);
    const size_t ws_bytes = trpx_encode_workspace_bytes(dtype, n_values, n_frames, block);
    if (!ws_bytes) return fail(block != 12 ? TRPX_ERR_UNSUPPORTED : TRPX_ERR_INVALID_ARG,
                               "trpx_encode_host: unsupported sizes/block (block=%u)", block);
    struct { void* p; } d_px, d_out, d_off, d_st, d_ws;
    Arena& A = arena();
    hipStream_t hs = nullptr;
    HIP_TRY(A.get_stream(&hs));
    HIP_TRY(A.get(Arena::kPixels, in_bytes, &d_px.p));
    HIP_TRY(A.get(Arena::kStream, cap, &d_out.p));
    HIP_TRY(A.get(Arena::kOffsets, 8 * (n_frames + 1), &d_off.p));
    HIP_TRY(A.get(Arena::kStatus, 4 * TRPX_STATUS_WORDS, &d_st.p));
    HIP_TRY(A.get(Arena::kWorkspace, ws_bytes, &d_ws.p, true));
    HIP_TRY(copy_sync(hs, d_px.p, pixels, in_bytes, hipMemcpyHostToDevice));
    int rc = trpx_encode(dtype, d_px.p, n_values, n_frames, block, static_cast<uint8_t*>(d_out.p), cap,
                         static_cast<uint64_t*>(d_off.p), static_cast<uint32_t*>(d_st.p), d_ws.p, ws_bytes, hs);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(hs));
    uint32_t st[TRPX_STATUS_WORDS];
    HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    if (st[0] == TRPX_ERR_TIMEOUT && g_encode_path == 0) {   // never seen in practice; keeps the API total
        t_force_two_pass = true;                             // (this thread's next call only: other threads are not affected)
        rc = trpx_encode(dtype, d_px.p, n_values, n_frames, block, static_cast<uint8_t*>(d_out.p), cap,
                         static_cast<uint64_t*>(d_off.p), static_cast<uint32_t*>(d_st.p), d_ws.p, ws_bytes, hs);
        t_force_two_pass = false;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(hs));
        HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    }
    if (st[0]) return fail((int)st[0], "trpx_encode_host: device status %u", st[0]);
    std::vector<uint64_t> offs(n_frames + 1);
    HIP_TRY(copy_sync(hs, offs.data(), d_off.p, 8 * (n_frames + 1), hipMemcpyDeviceToHost));
    const size_t total = (size_t)offs[n_frames];
    if (total > out_capacity) return fail(TRPX_ERR_CAPACITY, "trpx_encode_host: need %zu bytes, have %zu", total, out_capacity);
    HIP_TRY(copy_sync(hs, out, d_out.p, total, hipMemcpyDeviceToHost));
    *total_bytes = total;
    if (frame_offsets) memcpy(frame_offsets, offs.data(), 8 * (n_frames + 1));
    if (prolix_bits) *prolix_bits = st[1];
    return TRPX_OK;
}

int trpx_decode_host(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes,
                     const uint64_t* frame_offsets, size_t n_values, size_t n_frames, unsigned block,
                     void* pixels_out, int device) {
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_decode_host: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    const size_t es = out_dtype == TRPX_F32 ? 4 : out_dtype == TRPX_F64 ? 8 : trpx_dtype_size(out_dtype);
    if (!es || !terse || !pixels_out || !terse_bytes) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_host: bad argument");
    {
        trpx::FrameGeom g0;
        if (geom_of(n_values, block, &g0) && !sizes_ok(g0, n_frames)) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_host: bad sizes");
    }
    const size_t out_bytes = n_values * n_frames * es;
    const size_t ws_bytes = trpx_decode_workspace_bytes(TRPX_U8, n_values, n_frames, block);
    if (!ws_bytes) return fail(block != 12 ? TRPX_ERR_UNSUPPORTED : TRPX_ERR_INVALID_ARG,
                               "trpx_decode_host: unsupported sizes/block (block=%u)", block);
    struct { void* p = nullptr; } d_in, d_out, d_off, d_st, d_ws;
    Arena& A = arena();
    hipStream_t hs = nullptr;
    HIP_TRY(A.get_stream(&hs));
    HIP_TRY(A.get(Arena::kStream, trpx::align_up(terse_bytes, 4) + 8, &d_in.p));
    HIP_TRY(A.get(Arena::kPixels, out_bytes, &d_out.p));
    HIP_TRY(A.get(Arena::kStatus, 4 * TRPX_STATUS_WORDS, &d_st.p));
    HIP_TRY(A.get(Arena::kWorkspace, ws_bytes, &d_ws.p));
    HIP_TRY(hipMemsetAsync(static_cast<char*>(d_in.p) + (terse_bytes & ~size_t(3)), 0, trpx::align_up(terse_bytes, 4) + 8 - (terse_bytes & ~size_t(3)), hs));   // the bytes behind the stream read as zero
    HIP_TRY(copy_sync(hs, d_in.p, terse, terse_bytes, hipMemcpyHostToDevice));
    if (frame_offsets) {
        HIP_TRY(A.get(Arena::kOffsets, 8 * (n_frames + 1), &d_off.p));
        HIP_TRY(copy_sync(hs, d_off.p, frame_offsets, 8 * (n_frames + 1), hipMemcpyHostToDevice));
    }
    // Same signedness: the tuned decoders.  They report CORRUPT for a block wider than the output type, which is also
    // what a legitimately wider stream looks like (e.g. u16 data into a u8 container, Bit_pointer.hpp:747-763): those
    // -- and every cross-kind / float / double request -- take the converting decoder, whose verdict is final.
    const uint64_t* offs_dev = frame_offsets ? static_cast<const uint64_t*>(d_off.p) : nullptr;
    bool convert = !(out_dtype <= TRPX_I32 && (stream_signed != 0) == (trpx_dtype_is_signed(out_dtype) != 0));
    uint32_t st[TRPX_STATUS_WORDS];
    for (;;) {
        const int rc = convert ? trpx_decode_convert(stream_signed, out_dtype, static_cast<const uint8_t*>(d_in.p), terse_bytes,
                                                     offs_dev, n_values, n_frames, block, d_out.p,
                                                     static_cast<uint32_t*>(d_st.p), d_ws.p, ws_bytes, hs)
                               : trpx_decode(stream_signed, out_dtype, static_cast<const uint8_t*>(d_in.p), terse_bytes, offs_dev,
                                             n_values, n_frames, block, d_out.p, static_cast<uint32_t*>(d_st.p), d_ws.p,
                                             ws_bytes, hs);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(hs));
        HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
        if (st[0] == TRPX_ERR_CORRUPT && !convert && es <= 4) { convert = true; continue; }   // (32-bit containers: a stream of 64-bit pixels)
        break;
    }
    if (st[0]) return fail((int)st[0], "trpx_decode_host: corrupt or truncated stream (device status %u)", st[0]);
    HIP_TRY(copy_sync(hs, pixels_out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return TRPX_OK;
}

int trpx_frame_offsets_host(const uint8_t* terse, size_t terse_bytes, size_t n_values, size_t n_frames,
                            unsigned block, unsigned max_bits, uint64_t* frame_offsets, int device) {
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_frame_offsets_host: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    trpx::FrameGeom g;
    if (!terse || !terse_bytes || !frame_offsets || !n_frames || max_bits == 0 || max_bits > 64)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_frame_offsets_host: bad argument");
    if (!geom_of(n_values, block, &g))
        return fail(block != 12 ? TRPX_ERR_UNSUPPORTED : TRPX_ERR_INVALID_ARG,
                    "trpx_frame_offsets_host: unsupported sizes/block (block=%u)", block);
    if (!sizes_ok(g, n_frames) || n_frames > terse_bytes)                     // every frame is at least one byte (Terse.hpp:547)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_frame_offsets_host: bad sizes n_values=%zu n_frames=%zu", n_values, n_frames);
    const DecWs w = dec_ws(g, n_frames, 4);
    struct { void* p = nullptr; } d_in, d_st, d_ws;
    Arena& A = arena();
    hipStream_t hs = nullptr;
    HIP_TRY(A.get_stream(&hs));
    HIP_TRY(A.get(Arena::kStream, trpx::align_up(terse_bytes, 4) + 8, &d_in.p));
    HIP_TRY(A.get(Arena::kStatus, 4 * TRPX_STATUS_WORDS, &d_st.p));
    HIP_TRY(A.get(Arena::kWorkspace, w.total, &d_ws.p));
    HIP_TRY(hipMemsetAsync(static_cast<char*>(d_in.p) + (terse_bytes & ~size_t(3)), 0, trpx::align_up(terse_bytes, 4) + 8 - (terse_bytes & ~size_t(3)), hs));
    HIP_TRY(copy_sync(hs, d_in.p, terse, terse_bytes, hipMemcpyHostToDevice));
    trpx::DecodeArgs a{};
    a.terse = static_cast<const uint8_t*>(d_in.p);
    a.terse_bytes = terse_bytes;
    a.frame_offsets = nullptr;
    a.geom = g;
    a.n_frames = (uint32_t)n_frames;
    a.pixels_out = nullptr;
    a.status = static_cast<uint32_t*>(d_st.p);
    char* ws = static_cast<char*>(d_ws.p);
    a.walk_offsets = reinterpret_cast<uint64_t*>(ws + w.walk_offsets);
    a.tile_off = reinterpret_cast<uint64_t*>(ws + w.tile_off);
    a.widths = reinterpret_cast<uint8_t*>(ws + w.widths);
    a.seg_ws = ws + w.seg;
    HIP_TRY(trpx::launch_walk_serial(a, max_bits, hs));
    HIP_TRY(hipStreamSynchronize(hs));
    uint32_t st[TRPX_STATUS_WORDS];
    HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    if (st[0]) return fail((int)st[0], "trpx_frame_offsets_host: corrupt or truncated stack");
    HIP_TRY(copy_sync(hs, frame_offsets, a.walk_offsets, 8 * (n_frames + 1), hipMemcpyDeviceToHost));
    return TRPX_OK;
}


int trpx_group_states_host(const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets, size_t n_values,
                           size_t n_frames, unsigned block, unsigned max_bits, uint64_t* group_states, int device) {
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_group_states_host: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    trpx::FrameGeom g;
    if (!terse || !terse_bytes || !frame_offsets || !group_states || max_bits == 0 || max_bits > 32)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_group_states_host: bad argument");
    if (block != (unsigned)trpx::kBlock) return fail(TRPX_ERR_UNSUPPORTED, "trpx_group_states_host: block=%u", block);
    if (!geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || n_frames > terse_bytes)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_group_states_host: bad sizes");
    const int dtype = max_bits <= 8 ? TRPX_U8 : max_bits <= 16 ? TRPX_U16 : TRPX_U32;
    const size_t ib = trpx_index_bytes(dtype, n_values, n_frames, block), ng = n_frames * (size_t)g.n_tiles;
    struct { void* p = nullptr; } d_in, d_off, d_st, d_idx;
    Arena& A = arena();
    hipStream_t hs = nullptr;
    HIP_TRY(A.get_stream(&hs));
    HIP_TRY(A.get(Arena::kStream, trpx::align_up(terse_bytes, 4) + 8, &d_in.p));
    HIP_TRY(A.get(Arena::kOffsets, 8 * (n_frames + 1) + 8 * ng, &d_off.p));
    HIP_TRY(A.get(Arena::kStatus, 4 * TRPX_STATUS_WORDS, &d_st.p));
    HIP_TRY(A.get(Arena::kWorkspace, ib, &d_idx.p));
    HIP_TRY(hipMemsetAsync(static_cast<char*>(d_in.p) + (terse_bytes & ~size_t(3)), 0, trpx::align_up(terse_bytes, 4) + 8 - (terse_bytes & ~size_t(3)), hs));
    HIP_TRY(copy_sync(hs, d_in.p, terse, terse_bytes, hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(hs, d_off.p, frame_offsets, 8 * (n_frames + 1), hipMemcpyHostToDevice));
    uint64_t* d_states = static_cast<uint64_t*>(d_off.p) + (n_frames + 1);
    int rc = trpx_build_index(dtype, static_cast<const uint8_t*>(d_in.p), terse_bytes, static_cast<const uint64_t*>(d_off.p), n_values,
                              n_frames, block, d_idx.p, static_cast<uint32_t*>(d_st.p), hs);
    if (rc) return rc;
    rc = trpx_index_group_states(d_idx.p, n_values, n_frames, block, d_states, hs);
    if (rc) return rc;
    uint32_t st[TRPX_STATUS_WORDS];
    HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    if (st[0]) return fail((int)st[0], "trpx_group_states_host: corrupt or truncated stack (device status %u)", st[0]);
    HIP_TRY(copy_sync(hs, group_states, d_states, 8 * ng, hipMemcpyDeviceToHost));
    return TRPX_OK;
}

int trpx_decode_host_grouped(int stream_signed, int out_dtype, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                             const uint64_t* group_states, size_t n_values, size_t n_frames, unsigned block, void* pixels_out,
                             int device) {
    // the tuned, walk-free route needs a same-signedness integer type; everything else (and a
    // state table that does not fit the stream) goes the general way
    trpx::FrameGeom g;
    const bool tuned = group_states && frame_offsets && out_dtype <= TRPX_I32 && trpx_dtype_size(out_dtype) &&
                       (stream_signed != 0) == (trpx_dtype_is_signed(out_dtype) != 0) && block == (unsigned)trpx::kBlock &&
                       geom_of(n_values, block, &g) && sizes_ok(g, n_frames);
    if (!tuned) return trpx_decode_host(stream_signed, out_dtype, terse, terse_bytes, frame_offsets, n_values, n_frames, block, pixels_out, device);
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_decode_host_grouped: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    if (!terse || !terse_bytes || !pixels_out) return fail(TRPX_ERR_INVALID_ARG, "trpx_decode_host_grouped: bad argument");
    const size_t es = trpx_dtype_size(out_dtype), out_bytes = n_values * n_frames * es, ng = n_frames * (size_t)g.n_tiles;
    const size_t ib = trpx_index_bytes(out_dtype, n_values, n_frames, block);
    struct { void* p = nullptr; } d_in, d_out, d_off, d_st, d_idx;
    Arena& A = arena();
    hipStream_t hs = nullptr;
    HIP_TRY(A.get_stream(&hs));
    HIP_TRY(A.get(Arena::kStream, trpx::align_up(terse_bytes, 4) + 8, &d_in.p));
    HIP_TRY(A.get(Arena::kPixels, out_bytes, &d_out.p));
    HIP_TRY(A.get(Arena::kOffsets, 8 * (n_frames + 1) + 8 * ng, &d_off.p));
    HIP_TRY(A.get(Arena::kStatus, 4 * TRPX_STATUS_WORDS, &d_st.p));
    HIP_TRY(A.get(Arena::kWorkspace, ib, &d_idx.p));
    HIP_TRY(hipMemsetAsync(static_cast<char*>(d_in.p) + (terse_bytes & ~size_t(3)), 0, trpx::align_up(terse_bytes, 4) + 8 - (terse_bytes & ~size_t(3)), hs));
    HIP_TRY(copy_sync(hs, d_in.p, terse, terse_bytes, hipMemcpyHostToDevice));
    HIP_TRY(copy_sync(hs, d_off.p, frame_offsets, 8 * (n_frames + 1), hipMemcpyHostToDevice));
    uint64_t* d_states = static_cast<uint64_t*>(d_off.p) + (n_frames + 1);
    HIP_TRY(copy_sync(hs, d_states, group_states, 8 * ng, hipMemcpyHostToDevice));
    int rc = trpx_index_from_group_states(out_dtype, static_cast<const uint8_t*>(d_in.p), terse_bytes, static_cast<const uint64_t*>(d_off.p),
                                          d_states, n_values, n_frames, block, d_idx.p, static_cast<uint32_t*>(d_st.p), hs);
    if (rc) return rc;
    uint32_t st[TRPX_STATUS_WORDS];
    HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    if (st[0] == TRPX_ERR_CORRUPT)     // the states do not describe this stream (or the data are wider than the output type): general route
        return trpx_decode_host(stream_signed, out_dtype, terse, terse_bytes, frame_offsets, n_values, n_frames, block, pixels_out, device);
    rc = trpx_decode_indexed(stream_signed, out_dtype, static_cast<const uint8_t*>(d_in.p), terse_bytes, static_cast<const uint64_t*>(d_off.p),
                             d_idx.p, n_values, n_frames, block, d_out.p, static_cast<uint32_t*>(d_st.p), hs);
    if (rc) return rc;
    HIP_TRY(copy_sync(hs, st, d_st.p, sizeof st, hipMemcpyDeviceToHost));
    if (st[0]) return fail((int)st[0], "trpx_decode_host_grouped: corrupt or truncated stream (device status %u)", st[0]);
    HIP_TRY(copy_sync(hs, pixels_out, d_out.p, out_bytes, hipMemcpyDeviceToHost));
    return TRPX_OK;
}

// ---- a compressed stack kept on the device, read frame by frame (src/prolix.cpp:69-92's loop shape) --------------------
struct trpx_stack {
    int device = 0;
    int stream_signed = 0;
    size_t n_values = 0, n_frames = 0, terse_bytes = 0;
    unsigned block = 12;
    void* d_terse = nullptr;       // the stack (+ zero tail)
    void* d_offs = nullptr;        // u64[n_frames + 1]
    void* d_status = nullptr;
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    void* d_states = nullptr;      // u64[n_frames * groups]: chain state at every 256th block (may be absent)
    void* d_index = nullptr;       // decode index of the window's frames, rebuilt from the states (walk-free expansion)
    size_t groups = 0;
    void* d_window = nullptr;      // decoded frames [win_first, win_first + win_count) as win_dtype
    size_t window_cap = 0, win_first = 0, win_count = 0, window_frames = 0;
    int win_dtype = -1;
    std::vector<uint64_t> offs;    // host copy of the frame offsets
};

static void stack_free(trpx_stack* s) {
    if (!s) return;
    for (void* q : {s->d_terse, s->d_offs, s->d_status, s->d_ws, s->d_window, s->d_states, s->d_index}) if (q) (void)hipFree(q);
    delete s;
}

int trpx_stack_open(trpx_stack** handle, int stream_signed, const uint8_t* terse, size_t terse_bytes, const uint64_t* frame_offsets,
                    const uint64_t* group_states, size_t n_values, size_t n_frames, unsigned block, unsigned max_bits, int device) {
    if (!handle) return fail(TRPX_ERR_INVALID_ARG, "trpx_stack_open: null handle");
    *handle = nullptr;
    if (trpx_device_count() == 0) return fail(TRPX_ERR_NO_DEVICE, "trpx_stack_open: no HIP device");
    if (device >= 0) HIP_TRY(hipSetDevice(device));
    trpx::FrameGeom g;
    if (!terse || !terse_bytes || !geom_of(n_values, block, &g) || !sizes_ok(g, n_frames) || n_frames > terse_bytes)
        return fail(TRPX_ERR_INVALID_ARG, "trpx_stack_open: bad argument / sizes");
    std::vector<uint64_t> offs(n_frames + 1);
    if (frame_offsets) memcpy(offs.data(), frame_offsets, 8 * (n_frames + 1));
    else {
        const int rc = trpx_frame_offsets_host(terse, terse_bytes, n_values, n_frames, block, max_bits ? max_bits : 32, offs.data(), -1);
        if (rc) return rc;
    }
    if (offs[0] != 0 || offs[n_frames] > terse_bytes) return fail(TRPX_ERR_CORRUPT, "trpx_stack_open: frame offsets do not fit the stack");
    for (size_t f = 0; f < n_frames; ++f)
        if (offs[f + 1] <= offs[f]) return fail(TRPX_ERR_CORRUPT, "trpx_stack_open: frame offsets are not increasing");
    hipStream_t hs = nullptr;                                                  // the calling thread's private stream (see Arena)
    HIP_TRY(arena().get_stream(&hs));
    trpx_stack* s = new trpx_stack;
    auto bail = [&](hipError_t e, const char* what) { stack_free(s); return fail(TRPX_ERR_HIP, "trpx_stack_open: %s: %s", what, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipGetDevice(&s->device)) != hipSuccess) return bail(e, "hipGetDevice");
    s->stream_signed = stream_signed != 0;
    s->n_values = n_values; s->n_frames = n_frames; s->terse_bytes = terse_bytes; s->block = block;
    s->offs.swap(offs);
    const size_t frame_bytes = n_values * 8;                                   // widest output (double)
    s->window_frames = std::max<size_t>(1, std::min<size_t>(n_frames, (size_t(64) << 20) / frame_bytes));   // <= 64 MB of decoded frames
    s->ws_bytes = trpx_decode_workspace_bytes(TRPX_U8, n_values, s->window_frames, block);
    if ((e = hipMalloc(&s->d_terse, trpx::align_up(terse_bytes, 4) + 8)) != hipSuccess) return bail(e, "hipMalloc(stack)");
    if ((e = hipMalloc(&s->d_offs, 8 * (n_frames + 1))) != hipSuccess) return bail(e, "hipMalloc(offsets)");
    if ((e = hipMalloc(&s->d_status, 4 * TRPX_STATUS_WORDS)) != hipSuccess) return bail(e, "hipMalloc(status)");
    if ((e = hipMalloc(&s->d_ws, s->ws_bytes ? s->ws_bytes : 256)) != hipSuccess) return bail(e, "hipMalloc(workspace)");
    if ((e = hipMemsetAsync(s->d_terse, 0, trpx::align_up(terse_bytes, 4) + 8, hs)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = copy_sync(hs, s->d_terse, terse, terse_bytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(stack)");
    if ((e = copy_sync(hs, s->d_offs, s->offs.data(), 8 * (n_frames + 1), hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(offsets)");
    if (group_states && block == (unsigned)trpx::kBlock) {   // row f1: the file carried its group states
        s->groups = g.n_tiles;
        if ((e = hipMalloc(&s->d_states, 8 * n_frames * s->groups)) != hipSuccess) return bail(e, "hipMalloc(states)");
        if ((e = copy_sync(hs, s->d_states, group_states, 8 * n_frames * s->groups, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy(states)");
        if ((e = hipMalloc(&s->d_index, trpx_index_bytes(TRPX_U8, n_values, s->window_frames, block))) != hipSuccess) return bail(e, "hipMalloc(index)");
    }
    *handle = s;
    return TRPX_OK;
}

int trpx_stack_read(trpx_stack* s, size_t frame, int out_dtype, void* pixels_out) {
    if (!s || !pixels_out || frame >= s->n_frames) return fail(TRPX_ERR_INVALID_ARG, "trpx_stack_read: bad argument");
    const size_t es = out_dtype == TRPX_F32 ? 4 : out_dtype == TRPX_F64 ? 8 : trpx_dtype_size(out_dtype);
    if (!es) return fail(TRPX_ERR_INVALID_ARG, "trpx_stack_read: unknown dtype %d", out_dtype);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t hs = nullptr;                                                  // the calling thread's private stream
    HIP_TRY(arena().get_stream(&hs));
    if (out_dtype != s->win_dtype || frame < s->win_first || frame >= s->win_first + s->win_count) {
        // a miss: expand the window of frames that starts here (the callers of the reference walk the stack in order)
        const size_t count = std::min(s->window_frames, s->n_frames - frame);
        const size_t need = count * s->n_values * es;
        if (s->window_cap < need) {
            if (s->d_window) (void)hipFree(s->d_window);
            s->d_window = nullptr; s->window_cap = 0;
            HIP_TRY(hipMalloc(&s->d_window, need));
            s->window_cap = need;
        }
        s->win_count = 0;
        const uint64_t first_byte = s->offs[frame] & ~uint64_t(3);            // trpx_decode wants a 4-byte aligned stream
        const uint8_t* base = static_cast<const uint8_t*>(s->d_terse) + first_byte;
        std::vector<uint64_t> rel(count + 1);
        for (size_t i = 0; i <= count; ++i) rel[i] = s->offs[frame + i] - first_byte;
        HIP_TRY(copy_sync(hs, static_cast<uint64_t*>(s->d_offs), rel.data(), 8 * (count + 1), hipMemcpyHostToDevice));
        const size_t bytes = (size_t)rel[count];
        bool convert = !(out_dtype <= TRPX_I32 && (s->stream_signed != 0) == (trpx_dtype_is_signed(out_dtype) != 0));
        uint32_t st[TRPX_STATUS_WORDS];
        bool done = false;
        if (s->d_states && !convert) {                                         // walk-free: index from the file's group states
            int rc = trpx_index_from_group_states(out_dtype, base, bytes, static_cast<const uint64_t*>(s->d_offs),
                                                  static_cast<const uint64_t*>(s->d_states) + frame * s->groups, s->n_values, count,
                                                  s->block, s->d_index, static_cast<uint32_t*>(s->d_status), hs);
            if (rc) return rc;
            HIP_TRY(copy_sync(hs, st, s->d_status, sizeof st, hipMemcpyDeviceToHost));
            if (st[0] == 0) {
                rc = trpx_decode_indexed(s->stream_signed, out_dtype, base, bytes, static_cast<const uint64_t*>(s->d_offs), s->d_index,
                                         s->n_values, count, s->block, s->d_window, static_cast<uint32_t*>(s->d_status), hs);
                if (rc) return rc;
                HIP_TRY(copy_sync(hs, st, s->d_status, sizeof st, hipMemcpyDeviceToHost));
                done = st[0] == 0;
            }                                                                  // (states that do not fit the stream: the general route decides)
        }
        for (; !done;) {                                                       // (same routing as trpx_decode_host)
            const int rc = convert ? trpx_decode_convert(s->stream_signed, out_dtype, base, bytes, static_cast<const uint64_t*>(s->d_offs),
                                                         s->n_values, count, s->block, s->d_window, static_cast<uint32_t*>(s->d_status),
                                                         s->d_ws, s->ws_bytes, hs)
                                   : trpx_decode(s->stream_signed, out_dtype, base, bytes, static_cast<const uint64_t*>(s->d_offs), s->n_values,
                                                 count, s->block, s->d_window, static_cast<uint32_t*>(s->d_status), s->d_ws, s->ws_bytes, hs);
            if (rc) return rc;
            HIP_TRY(copy_sync(hs, st, s->d_status, sizeof st, hipMemcpyDeviceToHost));
            if (st[0] == TRPX_ERR_CORRUPT && !convert && es <= 4) { convert = true; continue; }   // (32-bit containers: a stream of 64-bit pixels)
            break;
        }
        if (st[0]) return fail((int)st[0], "trpx_stack_read: corrupt or truncated stream (device status %u)", st[0]);
        s->win_first = frame; s->win_count = count; s->win_dtype = out_dtype;
    }
    const size_t fb = s->n_values * es;
    HIP_TRY(copy_sync(hs, pixels_out, static_cast<const char*>(s->d_window) + (frame - s->win_first) * fb, fb, hipMemcpyDeviceToHost));
    return TRPX_OK;
}

void trpx_stack_close(trpx_stack* s) {
    if (s) { (void)hipSetDevice(s->device); stack_free(s); }
}

}  // extern "C"
