// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// Thin extern "C" shim that compiles the REAL reference codec (jpa::Terse, senikm/trpx
// @ 2024_08_07) from the headers where they lie under /root/reference/include and exposes
// it to the parity tests / golden-vector generator / bench.py's cpu_baseline leg.
// No reference source is copied into this repository: this file only #includes the headers
// in place and is built by oracle/Makefile into oracle/_ref/libtrpx_ref.so (git-ignored).
//
// Caveats handled here (SURVEY.md section 4 "reference defects"):
//   * <cmath> must be included before Terse.hpp (Terse.hpp:503 uses std::ceil).
//   * D1/D6: every frame is encoded / decoded as its OWN single-frame jpa::Terse object.
//   * The stream constructor only takes std::ifstream& (Terse.hpp:279) -> decode goes through
//     a real temporary file under /dev/shm (or $TMPDIR).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>
#include "Terse.hpp"

namespace {

// Encode one frame with the reference; returns the payload (header stripped) and the header text.
template <typename T>
long ref_encode_impl(const T* px, size_t n, unsigned block, uint8_t* out, size_t cap,
                     unsigned* prolix_bits, char* header, size_t header_cap) {
    jpa::Terse t(px, n, block);
    std::ostringstream os;
    t.write(os);
    std::string s = os.str();
    size_t hdr_end = s.find("/>");
    if (hdr_end == std::string::npos) return -1;
    hdr_end += 2;
    size_t payload = s.size() - hdr_end;
    if (payload != t.terse_size()) return -2;
    if (prolix_bits) *prolix_bits = t.bits_per_val();
    if (header && header_cap) {
        size_t k = hdr_end < header_cap - 1 ? hdr_end : header_cap - 1;
        memcpy(header, s.data(), k);
        header[k] = 0;
    }
    if (out) {
        if (payload > cap) return -3;
        memcpy(out, s.data() + hdr_end, payload);
    }
    return (long)payload;
}

std::string tmp_path() {
    const char* d = access("/dev/shm", W_OK) == 0 ? "/dev/shm" : (getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp");
    char buf[512];
    static thread_local unsigned long ctr = 0;
    snprintf(buf, sizeof buf, "%s/trpx_ref_%d_%lx_%lu.trpx", d, (int)getpid(),
             (unsigned long)(uintptr_t)&ctr, ctr++);
    return buf;
}

// Decode one single-frame stream with the reference (through a temp file, Terse.hpp:279).
template <typename T>
int ref_decode_impl(const uint8_t* payload, size_t nbytes, size_t n, unsigned block, int is_signed,
                    unsigned prolix_bits, T* out) {
    std::string path = tmp_path();
    {
        std::ofstream f(path, std::ios::binary);
        if (!f) return -1;
        f << "<Terse prolix_bits=\"" << prolix_bits << "\" signed=\"" << (is_signed ? 1 : 0)
          << "\" block=\"" << block << "\" memory_size=\"" << nbytes << "\" number_of_values=\"" << n
          << "\" number_of_frames=\"1\"/>";
        f.write((const char*)payload, (std::streamsize)nbytes);
    }
    int rc = 0;
    try {
        std::ifstream f(path, std::ios::binary);
        jpa::Terse t(f);
        t.prolix(out, 0);
    } catch (...) { rc = -2; }
    unlink(path.c_str());
    return rc;
}

}  // namespace

#define REF_API(T, SFX)                                                                              \
    extern "C" long trpx_ref_encode_##SFX(const T* px, size_t n, unsigned block, uint8_t* out,       \
                                          size_t cap, unsigned* prolix_bits, char* header,           \
                                          size_t header_cap) {                                       \
        return ref_encode_impl<T>(px, n, block, out, cap, prolix_bits, header, header_cap);          \
    }                                                                                                \
    extern "C" int trpx_ref_decode_##SFX(const uint8_t* payload, size_t nbytes, size_t n,            \
                                         unsigned block, int is_signed, unsigned prolix_bits,        \
                                         T* out) {                                                   \
        return ref_decode_impl<T>(payload, nbytes, n, block, is_signed, prolix_bits, out);           \
    }

REF_API(uint8_t, u8)
REF_API(int8_t, i8)
REF_API(uint16_t, u16)
REF_API(int16_t, i16)
REF_API(uint32_t, u32)
REF_API(int32_t, i32)
REF_API(uint64_t, u64)
REF_API(int64_t, i64)

// Multi-frame stack through the reference's own push_back path (O(F^2), D6: small stacks only).
// Returns total payload bytes; header text (with number_of_frames=F) optionally returned.
extern "C" long trpx_ref_encode_stack_u16(const uint16_t* px, size_t n, size_t frames, uint8_t* out,
                                          size_t cap, char* header, size_t header_cap,
                                          const size_t* dims, size_t ndims) {
    jpa::Terse t;
    for (size_t f = 0; f < frames; ++f) t.push_back(px + f * n, n);
    if (ndims) t.dim(std::vector<size_t>(dims, dims + ndims));
    std::ostringstream os;
    t.write(os);
    std::string s = os.str();
    size_t hdr_end = s.find("/>") + 2;
    size_t payload = s.size() - hdr_end;
    if (header && header_cap) {
        size_t k = hdr_end < header_cap - 1 ? hdr_end : header_cap - 1;
        memcpy(header, s.data(), k);
        header[k] = 0;
    }
    if (out) {
        if (payload > cap) return -3;
        memcpy(out, s.data() + hdr_end, payload);
    }
    return (long)payload;
}

// CPU-baseline timing helper: encode `frames` frames (one jpa::Terse object per frame, D6) and
// decode them again; returns seconds spent in encode / decode (wall clock, this thread only).
#include <chrono>
extern "C" int trpx_ref_time_u16(const uint16_t* px, size_t n, size_t frames, double* enc_s,
                                 double* dec_s, size_t* total_bytes, int* roundtrip_ok) {
    using clk = std::chrono::steady_clock;
    std::vector<jpa::Terse> objs;
    objs.reserve(frames);
    auto t0 = clk::now();
    for (size_t f = 0; f < frames; ++f) objs.emplace_back(px + f * n, n);
    auto t1 = clk::now();
    std::vector<uint16_t> back(n);
    size_t tot = 0;
    int ok = 1;
    double dec = 0;
    for (size_t f = 0; f < frames; ++f) {
        auto a = clk::now();
        objs[f].prolix(back.data(), 0);
        auto b = clk::now();
        dec += std::chrono::duration<double>(b - a).count();
        tot += objs[f].terse_size();
        if (memcmp(back.data(), px + f * n, n * sizeof(uint16_t)) != 0) ok = 0;
    }
    *enc_s = std::chrono::duration<double>(t1 - t0).count();
    *dec_s = dec;
    *total_bytes = tot;
    *roundtrip_ok = ok;
    return 0;
}
extern "C" int trpx_ref_time_i32(const int32_t* px, size_t n, size_t frames, double* enc_s,
                                 double* dec_s, size_t* total_bytes, int* roundtrip_ok) {
    using clk = std::chrono::steady_clock;
    std::vector<jpa::Terse> objs;
    objs.reserve(frames);
    auto t0 = clk::now();
    for (size_t f = 0; f < frames; ++f) objs.emplace_back(px + f * n, n);
    auto t1 = clk::now();
    std::vector<int32_t> back(n);
    size_t tot = 0;
    int ok = 1;
    double dec = 0;
    for (size_t f = 0; f < frames; ++f) {
        auto a = clk::now();
        objs[f].prolix(back.data(), 0);
        auto b = clk::now();
        dec += std::chrono::duration<double>(b - a).count();
        tot += objs[f].terse_size();
        if (memcmp(back.data(), px + f * n, n * sizeof(int32_t)) != 0) ok = 0;
    }
    *enc_s = std::chrono::duration<double>(t1 - t0).count();
    *dec_s = dec;
    *total_bytes = tot;
    *roundtrip_ok = ok;
    return 0;
}
