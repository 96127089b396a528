// The code blocks of INTEGRATION.md sections 2 and 4, compiled (and, on the GPU box, run): a stand-in class with the
// reference class's data members (include/Terse.hpp:476-483) carries the two patched bodies exactly as the document shows
// them; tests/test_abi.py checks that the document's blocks are these lines.  Needs a GPU to RUN (it calls the *_host
// entry points); `make -C tests/cpp` only has to compile and link it.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <iterator>
#include <numeric>
#include <stdexcept>
#include <type_traits>
#include <vector>
#include <rccl/rccl.h>
// [snippet:include]
#include "trpx_hip.h"
// [/snippet]

namespace jpa_patched {

template <typename V> constexpr int dtype_of =
    std::is_same_v<V, float> ? TRPX_F32 : std::is_same_v<V, double> ? TRPX_F64
    : (sizeof(V) == 1 ? TRPX_U8 : sizeof(V) == 2 ? TRPX_U16 : sizeof(V) == 4 ? TRPX_U32 : TRPX_U64) + std::is_signed_v<V>;

class Terse {                                   // the reference's members (Terse.hpp:476-483) + the device-side stack
public:
    template <typename Iterator>
    Terse(Iterator data, std::size_t size, unsigned block = 12)
        : d_signed(std::is_signed_v<typename std::iterator_traits<Iterator>::value_type>), d_block(block), d_size(size) {
        d_terse_frames.push_back(0);
        f_compress(data);
    }
    ~Terse() { if (d_stack) trpx_stack_close(d_stack); }
    Terse(Terse const&) = delete;
    template <typename Iterator>
    void push_back(Iterator data, std::size_t) {
        if (d_stack) { trpx_stack_close(d_stack); d_stack = nullptr; }
        d_terse_frames.push_back(0);
        f_compress(data);
    }
    std::size_t number_of_frames() const { return d_terse_frames.size(); }

    template <typename Iterator>
    void prolix(Iterator begin, std::size_t frame = 0) {
        using value_type = typename std::iterator_traits<Iterator>::value_type;
        const std::uint64_t* frame_offsets = d_frame_offsets.data();
        const std::uint64_t* group_states = nullptr;            // or: the file's group_bit_offsets attribute (section 5)
// [snippet:prolix]
        // include/Terse.hpp:352-389  -- body of prolix(Iterator begin, frame), any output type
        if (!d_stack && trpx_stack_open(&d_stack, d_signed, d_terse_data.data(), d_terse_data.size(), frame_offsets /* or nullptr */,
                                        group_states /* or nullptr */, d_size, number_of_frames(), d_block, /*max_bits*/ 0, -1) != TRPX_OK)
            throw std::runtime_error(trpx_last_error_string());
        if (trpx_stack_read(d_stack, frame, dtype_of<value_type>, &*begin) != TRPX_OK)
            throw std::runtime_error(trpx_last_error_string());
// [/snippet]
    }

private:
// [snippet:f_compress]
    // include/Terse.hpp:500-549  -- body of f_compress(Iterator data), contiguous integral data
    template <typename Iterator>
    void f_compress(Iterator data) {
        using T = typename std::iterator_traits<Iterator>::value_type;
        constexpr int dtype = (sizeof(T) == 1 ? TRPX_U8 : sizeof(T) == 2 ? TRPX_U16 : sizeof(T) == 4 ? TRPX_U32 : TRPX_U64) + std::is_signed_v<T>;
        std::size_t const prev = d_terse_data.size();
        std::size_t const cap  = trpx_worst_case_bytes(dtype, d_size, d_block);      // replaces the bound of :503
        d_terse_data.resize(prev + cap);
        std::size_t total = 0; unsigned pb = 0;
        if (trpx_encode_host(dtype, &*data, d_size, 1, d_block, d_terse_data.data() + prev, cap,
                             &total, nullptr, &pb, /*device*/ -1) != TRPX_OK)
            throw std::runtime_error(trpx_last_error_string());
        d_terse_data.resize(prev + total);                                            // :547
        d_prolix_bits = std::max(d_prolix_bits, pb);                                  // :516
        d_frame_offsets.push_back(prev + total);                                      // (the sizes f_find_terse_frame :562-585 recomputes)
    }
// [/snippet]
    bool d_signed;
    unsigned const d_block;
    std::size_t d_size;
    unsigned d_prolix_bits = 0;
    std::vector<std::uint8_t> d_terse_data;
    std::vector<std::size_t> d_terse_frames;
    std::vector<std::uint64_t> d_frame_offsets{0};
    trpx_stack* d_stack = nullptr;
};

}  // namespace jpa_patched

// one rank's step of a sharded encode (compiled; run by tests with a one-rank communicator)
int sharded_step(ncclComm_t comm, int world, const void* d_pixels, size_t n_values, size_t frames_per_rank, uint8_t* d_out, size_t cap,
                 uint64_t* d_local_offsets, uint32_t* d_status, void* d_ws, size_t ws, uint64_t* d_global_offsets,
                 uint32_t* d_prolix_bits, uint64_t* d_rank_base, void* d_gather_ws, hipStream_t stream) {
// [snippet:sharded]
    size_t gws = trpx_gather_workspace_bytes(frames_per_rank, world);
    trpx_encode(TRPX_U16, d_pixels, n_values, frames_per_rank, 12, d_out, cap, d_local_offsets, d_status, d_ws, ws, stream);
    trpx_gather_frame_offsets(comm, d_local_offsets, frames_per_rank, /*n_slot*/ frames_per_rank, d_status,
                              d_global_offsets /* world * frames_per_rank + 1 */, d_prolix_bits, d_rank_base /* world */,
                              d_gather_ws, gws, stream);          // pack kernel + ncclAllGather (8 B per frame) + scan kernel
// [/snippet]
    return 0;
}

// INTEGRATION.md section 4, the single-call form (ABI version 3)
int sharded_single_call(ncclComm_t comm, int world, int rank, const void* d_pixels, size_t n_values, size_t frames_per_rank, uint8_t* d_out,
                        size_t cap, uint64_t* d_local_offsets, uint32_t* d_status, uint64_t* d_global_offsets, uint32_t* d_prolix_bits,
                        uint64_t* d_rank_base, void* d_ws_e, void* d_ws_d, void* d_pixels_back, hipStream_t stream, hipStream_t side_stream) {
// [snippet:sharded1]
    size_t ws_e = trpx_encode_sharded_workspace_bytes(TRPX_U16, n_values, frames_per_rank, /*n_slot*/ frames_per_rank, 12, world);
    trpx_encode_sharded(comm, TRPX_U16, d_pixels, n_values, frames_per_rank, frames_per_rank, 12, d_out, cap, d_local_offsets, d_status,
                        d_global_offsets, d_prolix_bits, d_rank_base, d_ws_e, ws_e, stream, /*gather_stream*/ side_stream /* or NULL */);
    size_t ws_d = trpx_decode_sharded_workspace_bytes(TRPX_U16, n_values, frames_per_rank, 12);
    trpx_decode_sharded(/*stream_signed*/ 0, TRPX_U16, d_out, cap, d_global_offsets, /*first_frame*/ rank * frames_per_rank, n_values,
                        frames_per_rank, 12, d_pixels_back, d_status, d_ws_d, ws_d, stream);
// [/snippet]
    return 0;
}

int main() {
    std::vector<std::uint16_t> a(5000), b(5000), out(5000);
    std::iota(a.begin(), a.end(), 0);
    for (std::size_t i = 0; i < b.size(); ++i) b[i] = (std::uint16_t)((i * 7919u) & 0x3ff);
    jpa_patched::Terse t(a.begin(), a.size());
    t.push_back(b.begin(), b.size());
    t.prolix(out.begin(), 1);
    if (out != b) { std::printf("FAIL frame 1\n"); return 1; }
    std::vector<double> outd(5000);
    t.prolix(outd.begin(), 0);
    for (std::size_t i = 0; i < a.size(); ++i) if (outd[i] != (double)a[i]) { std::printf("FAIL frame 0 as double\n"); return 1; }
    std::vector<std::int64_t> wide(2400);
    for (std::size_t i = 0; i < wide.size(); ++i) wide[i] = ((std::int64_t)i << 33) - 77;
    jpa_patched::Terse w(wide.begin(), wide.size());               // src/terse.cpp:120-123: 64-bit containers
    std::vector<std::int64_t> wout(2400);
    w.prolix(wout.begin());
    if (wout != wide) { std::printf("FAIL 64-bit\n"); return 1; }
    std::printf("OK integration snippets\n");
    return 0;
}
