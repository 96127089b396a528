// Multi-GPU entry points of the C ABI (SURVEY.md section 8 row e).  Frames are independent units -- the header state
// resets per frame and every frame starts byte aligned (reference include/Terse.hpp:502-505) -- so a stack shards by
// contiguous frame ranges, one process per GPU, with NO payload exchange.  The single collective is the per-frame
// size gather: every rank contributes the sizes of its frames (+ its frame count and prolix_bits), one
// ncclAllGather over RCCL / xGMI moves them (16 KB per rank at 2000 frames: latency bound), and one small kernel
// turns them into the global byte offset of every frame (what the serial encoder's cursor would have been,
// Terse.hpp:502-504) and the stack-wide prolix_bits (Terse.hpp:516).
//
// RCCL is bound at run time (dlopen of the process's librccl: the caller's communicator and these calls then share
// one RCCL instance); libtrpx_hip.so itself does not link against it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include "../../include/trpx_hip.h"
#include "codec_common.hpp"

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
    char why[256] = "";
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {          // an instance the process already has, if any
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (h) break;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (h) break;
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!h) {
            snprintf(x.why, sizeof x.why, "librccl not found: %s", dlerror());
            return x;
        }
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(h, "ncclAllGather"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.CommCount = reinterpret_cast<decltype(x.CommCount)>(dlsym(h, "ncclCommCount"));
        x.CommUserRank = reinterpret_cast<decltype(x.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.GetErrorString && x.CommCount && x.CommUserRank;
        if (!x.ok) snprintf(x.why, sizeof x.why, "librccl lacks a needed symbol");
        return x;
    }();
    return r;
}

thread_local char g_shard_err[256] = "";
int shard_fail(int code, const char* what, const char* detail) {
    snprintf(g_shard_err, sizeof g_shard_err, "%s: %s", what, detail ? detail : "");
    return code;
}

// message of one rank: u64[n_slot + 2] = { S_0 .. S_{n_local-1}, 0 .., n_local, prolix_bits }
__global__ __launch_bounds__(trpx::kThreads) void k_gather_pack(const uint64_t* __restrict__ local_offsets, uint32_t n_local,
                                                               uint32_t n_slot, const uint32_t* __restrict__ enc_status,
                                                               uint64_t* __restrict__ msg) {
    for (uint32_t i = blockIdx.x * trpx::kThreads + threadIdx.x; i < n_slot + 2u; i += gridDim.x * trpx::kThreads) {
        uint64_t v = 0;
        if (i < n_local) v = local_offsets[i + 1] - local_offsets[i];          // S_f = 1 + bits/8 (Terse.hpp:547)
        else if (i == n_slot) v = n_local;
        else if (i == n_slot + 1u) v = enc_status ? enc_status[1] : 0u;        // d_prolix_bits (Terse.hpp:516)
        msg[i] = v;
    }
}

// One workgroup: exclusive prefix sum over the ranks' sizes in rank order = the byte offset of every frame in the
// global stack (Terse.hpp:502-504), its total behind the last frame, and the maximum of the ranks' prolix_bits.
__global__ __launch_bounds__(trpx::kThreads) void k_gather_scan(const uint64_t* __restrict__ all, uint32_t world, uint32_t n_slot,
                                                               uint64_t* __restrict__ global_offsets, uint32_t* __restrict__ prolix_bits,
                                                               uint64_t* __restrict__ rank_base) {
    // (rank_base may be null)
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    uint64_t out_idx = 0;                                                    // frames in front of the current rank
    uint32_t pb = 0;
    for (uint32_t r = 0; r < world; ++r) {
        const uint64_t* m = all + (uint64_t)r * (n_slot + 2u);
        const uint32_t cnt = (uint32_t)m[n_slot];
        pb = max(pb, (uint32_t)m[n_slot + 1u]);
        if (tid == 0 && rank_base) rank_base[r] = s_carry;                   // first byte of rank r's stack in the global one
        for (uint32_t i0 = 0; i0 < cnt; i0 += trpx::kThreads) {
            const uint32_t i = i0 + tid;
            const uint64_t v = i < cnt ? m[i] : 0ull;
            uint64_t inc = v;                                                // wavefront inclusive scan (64-bit: shuffles)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d, 64);
                if ((int)lane >= d) inc += (uint64_t)lo | ((uint64_t)hi << 32);
            }
            if (lane == 63u) s_wave[wave] = inc;
            __syncthreads();
            uint64_t base = s_carry;
            for (uint32_t k = 0; k < wave; ++k) base += s_wave[k];
            if (i < cnt) global_offsets[out_idx + i] = base + inc - v;
            __syncthreads();
            if (tid == trpx::kThreads - 1) s_carry = base + inc;
            __syncthreads();
        }
        out_idx += cnt;
    }
    if (tid == 0) {
        global_offsets[out_idx] = s_carry;
        if (prolix_bits) *prolix_bits = pb;
    }
}

}  // namespace

extern "C" {

const char* trpx_shard_last_error(void) { return g_shard_err; }

size_t trpx_gather_workspace_bytes(size_t n_slot, int world) {
    if (world <= 0) return 0;
    return trpx::align_up((n_slot + 2) * 8 * ((size_t)world + 1) + 8 * (size_t)world, 256);
}

int trpx_gather_frame_offsets(void* comm, const uint64_t* local_offsets, size_t n_local, size_t n_slot,
                              const uint32_t* encode_status, uint64_t* global_offsets, uint32_t* prolix_bits,
                              uint64_t* rank_base, void* workspace, size_t workspace_bytes, void* stream) {
    const Rccl& R = rccl();
    if (!R.ok) return shard_fail(TRPX_ERR_UNSUPPORTED, "trpx_gather_frame_offsets", R.why);
    if (!comm || !local_offsets || !global_offsets || !workspace || n_local > n_slot || n_slot == 0 || n_slot > 0x7FFFFFF0ull)
        return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_gather_frame_offsets", "bad argument");
    int world = 0;
    ncclResult_t rc = R.CommCount(static_cast<ncclComm_t>(comm), &world);
    if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclCommCount", R.GetErrorString(rc));
    if (workspace_bytes < trpx_gather_workspace_bytes(n_slot, world) || ((uintptr_t)workspace | (uintptr_t)global_offsets | (uintptr_t)local_offsets) % 8)
        return shard_fail(TRPX_ERR_CAPACITY, "trpx_gather_frame_offsets", "workspace too small or misaligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint64_t* msg = static_cast<uint64_t*>(workspace);
    uint64_t* all = msg + (n_slot + 2);
    uint64_t* bases = rank_base ? rank_base : all + (size_t)world * (n_slot + 2);
    hipLaunchKernelGGL(k_gather_pack, dim3((unsigned)((n_slot + 2 + trpx::kThreads - 1) / trpx::kThreads)), dim3(trpx::kThreads), 0, st,
                       local_offsets, (uint32_t)n_local, (uint32_t)n_slot, encode_status, msg);
    rc = R.AllGather(msg, all, n_slot + 2, ncclUint64, static_cast<ncclComm_t>(comm), st);
    if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclAllGather", R.GetErrorString(rc));
    hipLaunchKernelGGL(k_gather_scan, dim3(1), dim3(trpx::kThreads), 0, st, all, (uint32_t)world, (uint32_t)n_slot, global_offsets,
                       prolix_bits, bases);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return shard_fail(TRPX_ERR_HIP, "trpx_gather_frame_offsets", hipGetErrorString(e));
    return TRPX_OK;
}

int trpx_gather_pack(const uint64_t* local_offsets, size_t n_local, size_t n_slot, const uint32_t* encode_status, uint64_t* message,
                     void* stream) {
    if (!local_offsets || !message || n_local > n_slot || n_slot == 0 || n_slot > 0x7FFFFFF0ull)
        return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_gather_pack", "bad argument");
    hipLaunchKernelGGL(k_gather_pack, dim3((unsigned)((n_slot + 2 + trpx::kThreads - 1) / trpx::kThreads)), dim3(trpx::kThreads), 0,
                       static_cast<hipStream_t>(stream), local_offsets, (uint32_t)n_local, (uint32_t)n_slot, encode_status, message);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRPX_OK : shard_fail(TRPX_ERR_HIP, "trpx_gather_pack", hipGetErrorString(e));
}

int trpx_gather_scan(const uint64_t* all_messages, int world, size_t n_slot, uint64_t* global_offsets, uint32_t* prolix_bits,
                     uint64_t* rank_base, void* stream) {
    if (!all_messages || !global_offsets || world <= 0 || n_slot == 0 || n_slot > 0x7FFFFFF0ull)
        return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_gather_scan", "bad argument");
    hipLaunchKernelGGL(k_gather_scan, dim3(1), dim3(trpx::kThreads), 0, static_cast<hipStream_t>(stream), all_messages, (uint32_t)world,
                       (uint32_t)n_slot, global_offsets, prolix_bits, rank_base);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? TRPX_OK : shard_fail(TRPX_ERR_HIP, "trpx_gather_scan", hipGetErrorString(e));
}

// ---- one call per rank: encode + size gather / decode from the global offsets ---------------------------------------------------
namespace {
// local_offsets[i] = global_offsets[first + i] - global_offsets[first]: this rank's frames inside its own stack
__global__ __launch_bounds__(trpx::kThreads) void k_rebase_offsets(const uint64_t* __restrict__ global_offsets, uint64_t first, uint32_t n,
                                                                  uint64_t* __restrict__ local_offsets) {
    const uint64_t base = global_offsets[first];
    for (uint32_t i = blockIdx.x * trpx::kThreads + threadIdx.x; i <= n; i += gridDim.x * trpx::kThreads)
        local_offsets[i] = global_offsets[first + i] - base;
}
struct SideEvent {                                            // one cached event per calling thread (no allocation per call)
    hipEvent_t ev = nullptr;
    int device = -1;
    ~SideEvent() { if (ev) (void)hipEventDestroy(ev); }
};
thread_local SideEvent t_side;
}  // namespace

size_t trpx_encode_sharded_workspace_bytes(int dtype, size_t n_values, size_t n_local, size_t n_slot, unsigned block, int world) {
    const size_t e = trpx_encode_workspace_bytes(dtype, n_values, n_local, block), g = trpx_gather_workspace_bytes(n_slot, world);
    return e && g ? trpx::align_up(e, 256) + g : 0;
}

int trpx_encode_sharded(void* comm, int dtype, const void* pixels, size_t n_values, size_t n_local, size_t n_slot, unsigned block,
                        uint8_t* out, size_t out_capacity, uint64_t* local_offsets, uint32_t* status, uint64_t* global_offsets,
                        uint32_t* prolix_bits, uint64_t* rank_base, void* workspace, size_t workspace_bytes, void* stream,
                        void* gather_stream) {
    const Rccl& R = rccl();
    if (!R.ok) return shard_fail(TRPX_ERR_UNSUPPORTED, "trpx_encode_sharded", R.why);
    if (!comm || !workspace || n_local == 0 || n_local > n_slot) return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_encode_sharded", "bad argument");
    int world = 0;
    const ncclResult_t rc = R.CommCount(static_cast<ncclComm_t>(comm), &world);
    if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclCommCount", R.GetErrorString(rc));
    const size_t e_bytes = trpx::align_up(trpx_encode_workspace_bytes(dtype, n_values, n_local, block), 256);
    const size_t g_bytes = trpx_gather_workspace_bytes(n_slot, world);
    if (e_bytes == 0 || workspace_bytes < e_bytes + g_bytes) return shard_fail(TRPX_ERR_CAPACITY, "trpx_encode_sharded", "workspace too small");
    // this rank's frames: the reference's f_compress on its share of the stack (Terse.hpp:500-549), bytes kept locally
    int r = trpx_encode(dtype, pixels, n_values, n_local, block, out, out_capacity, local_offsets, status, workspace, e_bytes, stream);
    if (r != TRPX_OK) return shard_fail(r, "trpx_encode_sharded", trpx_last_error_string());
    hipStream_t gs = static_cast<hipStream_t>(stream);
    if (gather_stream && gather_stream != stream) {
        // the gather on the caller's second stream, behind the encode: whatever the caller enqueues on `stream` next (the decode of
        // its own frames needs no global offset) runs beside the collective; the caller joins the two streams where it needs the table
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return shard_fail(TRPX_ERR_HIP, "trpx_encode_sharded", "hipGetDevice");
        if (!t_side.ev || t_side.device != dev) {
            if (t_side.ev) (void)hipEventDestroy(t_side.ev);
            t_side.ev = nullptr;
            if (hipEventCreateWithFlags(&t_side.ev, hipEventDisableTiming) != hipSuccess) return shard_fail(TRPX_ERR_HIP, "trpx_encode_sharded", "hipEventCreate");
            t_side.device = dev;
        }
        gs = static_cast<hipStream_t>(gather_stream);
        if (hipEventRecord(t_side.ev, static_cast<hipStream_t>(stream)) != hipSuccess || hipStreamWaitEvent(gs, t_side.ev, 0) != hipSuccess)
            return shard_fail(TRPX_ERR_HIP, "trpx_encode_sharded", "event fork");
    }
    return trpx_gather_frame_offsets(comm, local_offsets, n_local, n_slot, status, global_offsets, prolix_bits, rank_base,
                                     static_cast<char*>(workspace) + e_bytes, g_bytes, gs);
}

size_t trpx_decode_sharded_workspace_bytes(int dtype, size_t n_values, size_t n_local, unsigned block) {
    const size_t d = trpx_decode_workspace_bytes(dtype, n_values, n_local, block);
    return d ? trpx::align_up(d, 256) + trpx::align_up(8 * (n_local + 1), 256) : 0;
}

int trpx_decode_sharded(int stream_signed, int out_dtype, const uint8_t* local_terse, size_t local_bytes, const uint64_t* global_offsets,
                        size_t first_frame, size_t n_values, size_t n_local, unsigned block, void* pixels_out, uint32_t* status,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!global_offsets || !workspace || n_local == 0 || n_local > 0x7FFFFFF0ull || (uintptr_t)global_offsets % 8 || (uintptr_t)workspace % 8)
        return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_decode_sharded", "bad argument");
    const size_t d_bytes = trpx::align_up(trpx_decode_workspace_bytes(out_dtype, n_values, n_local, block), 256);
    if (d_bytes == 0 || workspace_bytes < d_bytes + trpx::align_up(8 * (n_local + 1), 256))
        return shard_fail(TRPX_ERR_CAPACITY, "trpx_decode_sharded", "workspace too small");
    uint64_t* local_offsets = reinterpret_cast<uint64_t*>(static_cast<char*>(workspace) + d_bytes);
    hipLaunchKernelGGL(k_rebase_offsets, dim3((unsigned)((n_local + 1 + trpx::kThreads - 1) / trpx::kThreads)), dim3(trpx::kThreads), 0,
                       static_cast<hipStream_t>(stream), global_offsets, (uint64_t)first_frame, (uint32_t)n_local, local_offsets);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return shard_fail(TRPX_ERR_HIP, "trpx_decode_sharded", hipGetErrorString(e));
    const int r = trpx_decode(stream_signed, out_dtype, local_terse, local_bytes, local_offsets, n_values, n_local, block, pixels_out, status,
                              workspace, d_bytes, stream);
    return r == TRPX_OK ? TRPX_OK : shard_fail(r, "trpx_decode_sharded", trpx_last_error_string());
}

int trpx_comm_unique_id(void* id128) {
    const Rccl& R = rccl();
    if (!R.ok) return shard_fail(TRPX_ERR_UNSUPPORTED, "trpx_comm_unique_id", R.why);
    if (!id128) return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_comm_unique_id", "null pointer");
    ncclUniqueId id;
    const ncclResult_t rc = R.GetUniqueId(&id);
    if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclGetUniqueId", R.GetErrorString(rc));
    memcpy(id128, &id, sizeof id);
    return TRPX_OK;
}

int trpx_comm_init(void** comm, int world, int rank, const void* id128) {
    const Rccl& R = rccl();
    if (!R.ok) return shard_fail(TRPX_ERR_UNSUPPORTED, "trpx_comm_init", R.why);
    if (!comm || !id128 || world <= 0 || rank < 0 || rank >= world) return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_comm_init", "bad argument");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = R.CommInitRank(&c, world, id, rank);
    if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclCommInitRank", R.GetErrorString(rc));
    *comm = c;
    return TRPX_OK;
}

int trpx_comm_info(void* comm, int* world, int* rank) {
    const Rccl& R = rccl();
    if (!R.ok) return shard_fail(TRPX_ERR_UNSUPPORTED, "trpx_comm_info", R.why);
    if (!comm) return shard_fail(TRPX_ERR_INVALID_ARG, "trpx_comm_info", "null communicator");
    int v = 0;
    if (world) {
        const ncclResult_t rc = R.CommCount(static_cast<ncclComm_t>(comm), &v);
        if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclCommCount", R.GetErrorString(rc));
        *world = v;
    }
    if (rank) {
        const ncclResult_t rc = R.CommUserRank(static_cast<ncclComm_t>(comm), &v);
        if (rc != ncclSuccess) return shard_fail(TRPX_ERR_HIP, "ncclCommUserRank", R.GetErrorString(rc));
        *rank = v;
    }
    return TRPX_OK;
}

int trpx_comm_destroy(void* comm) {
    const Rccl& R = rccl();
    if (!R.ok || !comm) return TRPX_OK;
    const ncclResult_t rc = R.CommDestroy(static_cast<ncclComm_t>(comm));
    return rc == ncclSuccess ? TRPX_OK : shard_fail(TRPX_ERR_HIP, "ncclCommDestroy", R.GetErrorString(rc));
}

}  // extern "C"
