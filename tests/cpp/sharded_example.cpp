// A C++ caller of the single-call sharded entry points (include/trpx_hip.h: trpx_encode_sharded / trpx_decode_sharded, SURVEY.md
// section 8 rows b, e) on a ONE-rank RCCL communicator: what each process of an N-GPU job does with its share of the stack.
// Frames are independent and byte aligned (reference include/Terse.hpp:502-505), so the global stack is the ranks' stacks in rank
// order; nothing but the per-frame sizes is exchanged.  The encode of the same frames through trpx::Terse (one device call,
// Terse.hpp:249-322's surface) is the check.  Needs a GPU and librccl.
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "trpx/Terse.hpp"
#include "trpx_hip.h"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s (%s | %s)\n", __FILE__, __LINE__, #c, trpx_last_error_string(), trpx_shard_last_error()); return 1; } } while (0)

int main() {
    const std::size_t n = 300 * 211, frames = 19;
    std::vector<std::uint16_t> px(frames * n);
    for (std::size_t i = 0; i < px.size(); ++i) px[i] = (std::uint16_t)(((i * 2654435761u) >> 28) + ((i % 4099) == 0 ? 900 : 0));
    trpx::Terse ref;
    ref.push_back(px.data(), n, frames);                                     // the single-process stack of the same frames

    char id[128];
    void* comm = nullptr;
    REQUIRE(trpx_comm_unique_id(id) == TRPX_OK);
    REQUIRE(trpx_comm_init(&comm, 1, 0, id) == TRPX_OK);
    const int world = 1, rank = 0;

    const std::size_t cap = (frames * trpx_worst_case_bytes(TRPX_U16, n, 12) + 15) / 16 * 16;
    const std::size_t ws_e = trpx_encode_sharded_workspace_bytes(TRPX_U16, n, frames, frames, 12, world);
    const std::size_t ws_d = trpx_decode_sharded_workspace_bytes(TRPX_U16, n, frames, 12);
    void *d_px, *d_out, *d_loc, *d_glob, *d_st, *d_pb, *d_base, *d_wse, *d_wsd, *d_back;
    REQUIRE(hipMalloc(&d_px, px.size() * 2) == hipSuccess && hipMalloc(&d_out, cap) == hipSuccess && hipMalloc(&d_loc, 8 * (frames + 1)) == hipSuccess);
    REQUIRE(hipMalloc(&d_glob, 8 * (world * frames + 1)) == hipSuccess && hipMalloc(&d_st, 64) == hipSuccess && hipMalloc(&d_pb, 8) == hipSuccess);
    REQUIRE(hipMalloc(&d_base, 8 * world) == hipSuccess && hipMalloc(&d_wse, ws_e) == hipSuccess && hipMalloc(&d_wsd, ws_d) == hipSuccess);
    REQUIRE(hipMalloc(&d_back, px.size() * 2) == hipSuccess);
    REQUIRE(hipMemcpy(d_px, px.data(), px.size() * 2, hipMemcpyHostToDevice) == hipSuccess);
    hipStream_t st, side;
    REQUIRE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&side, hipStreamNonBlocking) == hipSuccess);

    for (int pass = 0; pass < 2; ++pass) {                                   // the gather on the call's stream, then on a second stream
        REQUIRE(trpx_encode_sharded(comm, TRPX_U16, d_px, n, frames, frames, 12, static_cast<std::uint8_t*>(d_out), cap,
                                    static_cast<std::uint64_t*>(d_loc), static_cast<std::uint32_t*>(d_st), static_cast<std::uint64_t*>(d_glob),
                                    static_cast<std::uint32_t*>(d_pb), static_cast<std::uint64_t*>(d_base), d_wse, ws_e, st, pass ? side : nullptr) == TRPX_OK);
        REQUIRE(hipStreamSynchronize(st) == hipSuccess && hipStreamSynchronize(side) == hipSuccess);
        std::vector<std::uint64_t> glob(frames + 1);
        std::uint32_t status[TRPX_STATUS_WORDS], pb = 0;
        std::uint64_t base = 1;
        REQUIRE(hipMemcpy(glob.data(), d_glob, 8 * (frames + 1), hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(hipMemcpy(status, d_st, sizeof status, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(&pb, d_pb, 4, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(hipMemcpy(&base, d_base, 8, hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(status[0] == 0 && pb == ref.bits_per_val() && base == 0 && glob[frames] == ref.terse_size());
        std::vector<std::uint8_t> stack(glob[frames]);
        REQUIRE(hipMemcpy(stack.data(), d_out, stack.size(), hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(stack == ref.data());                                        // byte-identical to the single-process encode (Terse.hpp:500-549)
    }
    REQUIRE(hipMemset(d_back, 0xEE, px.size() * 2) == hipSuccess);
    REQUIRE(trpx_decode_sharded(0, TRPX_U16, static_cast<const std::uint8_t*>(d_out), cap, static_cast<const std::uint64_t*>(d_glob), rank * frames, n,
                                frames, 12, d_back, static_cast<std::uint32_t*>(d_st), d_wsd, ws_d, st) == TRPX_OK);
    REQUIRE(hipStreamSynchronize(st) == hipSuccess);
    std::vector<std::uint16_t> back(px.size());
    std::uint32_t status[TRPX_STATUS_WORDS];
    REQUIRE(hipMemcpy(back.data(), d_back, back.size() * 2, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(status, d_st, sizeof status, hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(status[0] == 0 && back == px);
    REQUIRE(trpx_comm_destroy(comm) == TRPX_OK);
    for (void* q : {d_px, d_out, d_loc, d_glob, d_st, d_pb, d_base, d_wse, d_wsd, d_back}) (void)hipFree(q);
    std::printf("OK sharded example: %zu frames, %zu bytes, one rank\n", frames, (std::size_t)ref.terse_size());
    return 0;
}
