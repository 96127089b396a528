// Host-visible launcher interface between api.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "codec_common.hpp"

namespace trpx {

struct EncodeArgs {
    const void* pixels;        // device, [n_frames][n_values]
    FrameGeom   geom;
    uint32_t    n_frames;
    uint8_t*    out;           // device, compact stack
    size_t      out_capacity;
    uint64_t*   frame_offsets; // device, n_frames + 1
    uint32_t*   status;        // device, 8 words
    // workspace carve
    uint64_t*   frame_size;    // n_frames
    uint64_t*   tile_off;      // n_frames * n_tiles
    uint32_t*   tile_bits;     // n_frames * n_tiles
    // optional decode index (NULL = not wanted): what the header walk would produce
    uint8_t*    idx_widths;    // n_frames * n_blocks
    uint64_t*   idx_group_off; // n_frames * n_tiles (frame-relative bit offset of every 256-block group)
};

// A PART of a frame: the blocks [b0, b1) with the chain state in front of b0 (decode_part.hip).  Frames of more than
// kPartMaxBlocks blocks are cut into parts of about kPartBlocks blocks so that the per-frame decoder -- one serial walker per
// workgroup -- works on pieces of the size of a 512 x 512 frame whatever the frame size: `pos0` / `w0` = bit position of block
// b0's header inside the frame and width of the block before it, `pos1` / `w1` = the same for block b1 (checked by the part's
// walker when it gets there; the frame's last part checks S_f = 1 + bits/8 instead, Terse.hpp:547).  b1 <= b0: nothing to do
// (the frame took another route).
struct PartDesc {
    uint32_t frame, b0, b1, pos0, w0, pos1, w1, pad;
};
constexpr uint32_t kPartBlocks = 16384;            // blocks per part, at most about (parts are cut at bit positions; stacks of few frames get smaller parts)
constexpr uint32_t kPartMaxBlocks = 32768;         // frames of up to this many blocks are one part; no part may hold more
// Stacks of kManyFrames frames and more keep frames of up to kManyBlocks blocks whole (single_part_blocks; the per-frame kernels'
// own limit, 2^26 bits per frame, is checked where the route is chosen).  tools/experiments/route_sweep.sh, ~1 GB of u16 pixels,
// index route / whole frames: Poisson(3) 1280 x 640^2 1.42 / 0.86 ms, 888 x 768^2 1.13 / 0.96, 960 x (1030 x 1065) 2.20 / 1.68, 500 x
// 1024^2 1.08 / 1.36, 250 x 1448^2 1.07 / 2.28; synth-v1 0.52 / 0.23, 0.41 / 0.28, 0.87 / 0.73, 0.41 / 0.43, 0.43 / 0.81.
constexpr uint32_t kManyFrames = 768;
constexpr uint32_t kManyBlocks = 1u << 20;
uint32_t single_part_blocks(size_t n_frames);
void set_single_part_rule(uint32_t many_frames, uint32_t many_blocks);   // ($TRPX_SINGLE_PART = "frames,blocks": tuning runs)

struct DecodeArgs {
    const uint8_t*  terse;         // device
    size_t          terse_bytes;
    const uint64_t* frame_offsets; // device, n_frames + 1 (never null here: api fills it by a walk)
    FrameGeom       geom;
    uint32_t        n_frames;
    void*           pixels_out;    // device
    uint32_t*       status;        // device
    // workspace carve
    uint64_t*       tile_off;      // n_frames * n_tiles : bit offset of each tile inside its frame
    uint8_t*        widths;        // n_frames * n_blocks : significant bits of every block
    uint64_t*       walk_offsets;  // n_frames + 1 : frame offsets produced by the serial walk
    uint32_t*       defer;         // 2 + n_frames words, 8-byte aligned (may be null): [0] = count, [1 + i] = frames the per-frame decoder hands to the position-parallel path
    void*           seg_ws;        // seg_workspace_bytes(): segment states of the position-parallel walk (decode_seg.hip)
    bool            index_per_frame = false;   // launch_walk_only: many small frames, use the per-frame walker (launch_index_frames)
    // frames of more than kPartMaxBlocks blocks on the per-frame route: the part table and the scratch of its construction
    PartDesc*       parts = nullptr;           // n_frames * parts_per_frame entries
    uint32_t        parts_per_frame = 1;
    void*           part_ws = nullptr;         // part_workspace_bytes() / chain_workspace_bytes()
    bool            index_given = false;       // a.widths / a.tile_off are the CALLER's decode index (read only): the frames the per-frame decoder lists are extracted with it, no walk
    bool            chain = false;             // large frames by the index route (decode_part.hip: one walk -> the decode index -> extraction with the widths given); parts / parts_per_frame / part_ws are that route's
};
uint32_t parts_per_frame(const FrameGeom& g, size_t n_frames);
size_t part_workspace_bytes(const FrameGeom& g, size_t n_frames);
// decode_part.hip: fills a.parts for every frame; frames whose parts cannot be established are listed in a.defer (whole frames)
hipError_t launch_build_parts(const DecodeArgs& a, uint32_t max_w, hipStream_t st);
// decode_part.hip, the index route: a.widths / a.tile_off of every large frame from ONE walk of many short parts; frames where
// that does not work out are listed in a.defer (the position-parallel walk writes their index: launch_seg_listed)
uint32_t chain_parts_per_frame(const FrameGeom& g, size_t n_frames, size_t pixel_bytes);   // (pixel_bytes: of the type decoded into)
size_t chain_workspace_bytes(const FrameGeom& g, size_t n_frames, size_t pixel_bytes);
// narrow: 8 / 16-bit pixels -- frames with few explicit headers are left to k_decode_parts on a.parts (*frame_mode: per frame, 1 = the
// index was written, 0 = extract part by part)
hipError_t launch_build_index_chain(const DecodeArgs& a, uint32_t max_w, bool narrow, const uint32_t** frame_mode, hipStream_t st);
// ... and what has to be cleared in front of it, in one launch: the deferred-frame list's count and statistics, the route's ready
// flags, and (clear_status) the status block
hipError_t launch_chain_zero(const DecodeArgs& a, uint32_t max_w, bool clear_status, hipStream_t st);
// decode_fast.hip: the tiled extraction of every frame with a.widths / a.tile_off given (no status clear, no profiler marks)
hipError_t launch_unpack_tiles(int dtype, const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode = nullptr);   // frame_mode: only frames with mode 1

hipError_t launch_encode(int dtype, const EncodeArgs& a, hipStream_t st);
// any block size (encode.hip, correct-first kernels): geom.block != 12
hipError_t launch_encode_generic(int dtype, const EncodeArgs& a, hipStream_t st);
// single-pass encoder (encode_fused.hip); `ws` = fused_workspace_bytes() of descriptor words
size_t fused_workspace_bytes(const FrameGeom& g, size_t n_frames);
hipError_t launch_encode_fused(int dtype, const EncodeArgs& a, void* ws, hipStream_t st);
// the library's memory of workspaces its single-pass encoder left clean: forget those inside [lo, lo + bytes) (lo == nullptr: all)
void fused_ws_forget(const void* lo, size_t bytes, const void* keep = nullptr);
hipError_t launch_decode(int dtype, const DecodeArgs& a, bool have_offsets, hipStream_t st);
// converting decode (decode.hip): any integral output type with clamping, float, double; stream signedness given
hipError_t launch_decode_convert(int dtype, const DecodeArgs& a, int stream_signed, bool have_offsets, hipStream_t st);
// tuned decode (decode_fast.hip): needs block = 12, frame offsets, frames of < 2^32 bits; pixels_out aligned to the pixel type (any pixel count)
hipError_t launch_decode_fast(int dtype, const DecodeArgs& a, bool have_index, hipStream_t st, bool per_frame = false);   // per_frame (with an index): k_decode_frames_indexed
// one workgroup per frame, walk and extraction fused through LDS (decode_frame.hip): many small frames
hipError_t launch_decode_frames(int dtype, const DecodeArgs& a, hipStream_t st);
hipError_t launch_decode_frames_indexed(int dtype, const DecodeArgs& a, const uint32_t* list, hipStream_t st);   // widths / group offsets given
hipError_t launch_decode_units_indexed(int dtype, const DecodeArgs& a, hipStream_t st, const uint32_t* frame_mode = nullptr);                         // the same over units of 6144 blocks of large frames
hipError_t launch_index_frames(uint32_t max_w, const DecodeArgs& a, bool clear_status, hipStream_t st);   // the index by the per-frame walker (needs a.defer, a.seg_ws)
hipError_t launch_seg_listed(const DecodeArgs& a, uint32_t max_w, hipStream_t st);                       // decode_seg.hip: index of the frames listed in a.defer
hipError_t launch_seg_groups(const DecodeArgs& a, uint32_t max_w, const uint64_t* states, hipStream_t st);   // decode_seg.hip: index from group states (frames < 2^32 bits)
// header walk only (fills a.widths / a.tile_off from the stream): builds the decode index of an existing stack
hipError_t launch_walk_only(const DecodeArgs& a, uint32_t max_w, bool clear_status, hipStream_t st);
hipError_t launch_walk_serial(const DecodeArgs& a, uint32_t max_w, hipStream_t st);
// group states (chain state at every 256th block): read them off an index (a.widths / a.tile_off) / rebuild the index from them
hipError_t launch_index_group_states(const DecodeArgs& a, uint64_t* states, hipStream_t st);
hipError_t launch_walk_groups(const DecodeArgs& a, uint32_t max_w, const uint64_t* states, bool clear_status, hipStream_t st);
// position-parallel header walk (decode_seg.hip): fills a.widths / a.tile_off like launch_walk_only; needs a.seg_ws
size_t seg_workspace_bytes(const FrameGeom& g, size_t n_frames);   // (includes dense_workspace_bytes(): the listed frames' walk, decode_dense.hip)
// decode_dense.hip: the decode index of the frames in `list` (frames of up to single_part_blocks() blocks) -- one speculative pass,
// link walks, a write pass per frame; dense_ws: dense_workspace_bytes()
size_t dense_workspace_bytes(const FrameGeom& g, size_t n_frames);
hipError_t launch_dense_listed(const DecodeArgs& a, uint32_t max_w, void* dense_ws, const uint32_t* list, hipStream_t st);
void set_dense_route(bool on);                                       // (off: the fix-point rounds of decode_seg.hip, A/B and tests)
hipError_t launch_seg_walk(const DecodeArgs& a, uint32_t max_w, hipStream_t st);
// frames k_decode_frames flagged in a.defer (explicit headers every few blocks): position-parallel walk + tiled extraction
bool seg_single_wave(const FrameGeom& g, size_t n_frames);
hipError_t launch_decode_deferred(int dtype, const DecodeArgs& a, hipStream_t st);
hipError_t launch_synth(int dtype, uint64_t seed, uint64_t frame0, size_t n_frames, size_t n_values,
                        void* out, hipStream_t st);

}  // namespace trpx
