// Host-code sanitizer tier (CPU build only: g++ -fsanitize=address,undefined; no HIP, no GPU).
// Everything in the product that parses untrusted bytes on the host is driven with valid, truncated and corrupted
// inputs: the .trpx header text (trpx_amd/csrc/header_text.cpp: trpx_header_parse / _format / _frame_sizes), the TIFF
// reader (include/trpx/Grey_tif.hpp: scan) and the stream constructor of the Terse class (include/trpx/Terse.hpp:
// f_read).  A failure is an ASan / UBSan report (the process aborts) or a wrong answer; rejected inputs must be
// rejected with an error code / exception, never by luck.
//
// The device entry points Terse.hpp references are stubbed below (this binary never reaches a GPU): f_read only calls
// trpx_frame_offsets_host, for multi-frame files without a frame_sizes attribute, and must pass the error on.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "trpx/Grey_tif.hpp"
#include "trpx/Terse.hpp"

extern "C" {
static int no_device() { return TRPX_ERR_NO_DEVICE; }
const char* trpx_last_error_string(void) { return "host sanitizer build: no device"; }
int trpx_encode_host(int, const void*, size_t, size_t, unsigned, uint8_t*, size_t, size_t*, uint64_t*, uint32_t*, int) { return no_device(); }
int trpx_decode_host(int, int, const uint8_t*, size_t, const uint64_t*, size_t, size_t, unsigned, void*, int) { return no_device(); }
int trpx_frame_offsets_host(const uint8_t*, size_t, size_t, size_t, unsigned, unsigned, uint64_t*, int) { return no_device(); }
int trpx_stack_open(trpx_stack**, int, const uint8_t*, size_t, const uint64_t*, const uint64_t*, size_t, size_t, unsigned, unsigned, int) { return no_device(); }
int trpx_group_states_host(const uint8_t*, size_t, const uint64_t*, size_t, size_t, unsigned, unsigned, uint64_t*, int) { return no_device(); }
int trpx_decode_host_grouped(int, int, const uint8_t*, size_t, const uint64_t*, const uint64_t*, size_t, size_t, unsigned, void*, int) { return no_device(); }
size_t trpx_group_count(size_t n, unsigned block) { return block == 12 ? ((n + 11) / 12 + 255) / 256 : 0; }
int trpx_stack_read(trpx_stack*, size_t, int, void*) { return no_device(); }
void trpx_stack_close(trpx_stack*) {}
size_t trpx_worst_case_bytes(int, size_t n, unsigned) { return 8 * n + 64; }
size_t trpx_dtype_size(int d) { return d < 2 ? 1 : d < 4 ? 2 : 4; }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

// ---- header text -------------------------------------------------------------------------------------------------
static void headers() {
    trpx_header h{};
    h.prolix_bits = 12; h.is_signed = 0; h.block = 12; h.memory_size = 305036; h.number_of_values = 262144; h.number_of_frames = 3;
    h.n_dims = 2; h.dims[0] = 512; h.dims[1] = 512;
    char buf[512];
    const size_t len = trpx_header_format(&h, buf, sizeof buf);
    EXPECT(len > 0 && len < sizeof buf);
    trpx_header g{};
    size_t off = 0;
    EXPECT(trpx_header_parse(buf, len, &g, &off) == TRPX_OK && off == len && g.memory_size == h.memory_size && g.n_dims == 2);
    char tiny[8];
    EXPECT(trpx_header_format(&h, tiny, sizeof tiny) == 0);
    const uint64_t sizes[3] = {101683, 101603, 101750};
    char ibuf[1024];
    const size_t ilen = trpx_header_format_indexed(&h, sizes, 3, ibuf, sizeof ibuf);
    EXPECT(ilen > len);
    uint64_t back[3] = {0, 0, 0};
    EXPECT(trpx_header_frame_sizes(ibuf, ilen, back, 3) == 3 && back[2] == 101750);
    uint64_t one[1];
    (void)trpx_header_frame_sizes(ibuf, ilen, one, 1);                      // capacity smaller than the attribute
    {   // group states: "offset:width" tokens, round trip and hostile values
        const uint64_t gs[4] = {0, 12345ull | (3ull << 40), (1ull << 40) - 1, 77ull | (64ull << 40)};
        char gbuf[1024];
        const size_t glen = trpx_header_format_grouped(&h, sizes, 3, gs, 4, gbuf, sizeof gbuf);
        EXPECT(glen > ilen);
        uint64_t gb[4] = {1, 1, 1, 1};
        EXPECT(trpx_header_group_states(gbuf, glen, gb, 4) == 4 && gb[1] == gs[1] && gb[2] == gs[2] && gb[3] == gs[3]);
        EXPECT(trpx_header_frame_sizes(gbuf, glen, back, 3) == 3);
        trpx_header t{};
        size_t o = 0;
        EXPECT(trpx_header_parse(gbuf, glen, &t, &o) == TRPX_OK && o == glen);
        for (size_t n = 0; n < glen; ++n) { std::vector<char> cut(gbuf, gbuf + n); uint64_t x[2]; (void)trpx_header_group_states(cut.data(), cut.size(), x, 2); }
        const char* evil[] = {"<Terse group_bit_offsets=\"1:2 3: 4:5\"/>", "<Terse group_bit_offsets=\"99999999999999999999:1\"/>",
                              "<Terse group_bit_offsets=\"1:999 2:3\"/>", "<Terse group_bit_offsets=\":::: 1:1\"/>", "<Terse group_bit_offsets=\"5:6"};
        for (const char* e : evil) { std::vector<char> m(e, e + std::strlen(e)); uint64_t x[3]; EXPECT(trpx_header_group_states(m.data(), m.size(), x, 3) <= 3); }
    }
    // every prefix: parse must fail cleanly (never read past `len`; the buffers are exact-size heap blocks for ASan)
    for (size_t n = 0; n < ilen; ++n) {
        std::vector<char> cut(ibuf, ibuf + n);
        trpx_header t{};
        size_t o = 0;
        const int rc = trpx_header_parse(cut.data(), cut.size(), &t, &o);
        EXPECT(rc != TRPX_OK || o <= n);
        uint64_t fs[4];
        (void)trpx_header_frame_sizes(cut.data(), cut.size(), fs, 4);
    }
    // random corruption of a valid header + absurd numbers
    for (int it = 0; it < 10000; ++it) {
        std::vector<char> m(ibuf, ibuf + ilen);
        const int flips = 1 + (int)(rnd() % 4);
        for (int k = 0; k < flips; ++k) m[rnd() % m.size()] = (char)rnd();
        trpx_header t{};
        size_t o = 0;
        if (trpx_header_parse(m.data(), m.size(), &t, &o) == TRPX_OK) EXPECT(o <= m.size() && t.n_dims <= 8);
        uint64_t fs[8];
        EXPECT(trpx_header_frame_sizes(m.data(), m.size(), fs, 8) <= 8 || true);
    }
    const char* absurd[] = {
        "<Terse prolix_bits=\"99999999999999999999999\" signed=\"1\" block=\"12\" memory_size=\"18446744073709551616\" number_of_values=\"1\" number_of_frames=\"1\"/>",
        "<Terse prolix_bits=\"-1\" signed=\"x\" block=\"\" memory_size=\"\" number_of_values=\"\" number_of_frames=\"\"/>",
        "<Terse dimensions=\"1 2 3 4 5 6 7 8 9 10 11 12\" prolix_bits=\"1\" signed=\"0\" block=\"12\" memory_size=\"1\" number_of_values=\"1\" number_of_frames=\"1\"/>",
        "<Terse prolix_bits=\"1\" signed=\"0\" block=\"12\" memory_size=\"1\" number_of_values=\"1\" number_of_frames=\"1\" frame_sizes=\"1 2 3",
        "<Terse", "<Terse/>", "", "<<<<Terse Terse <Terse a=\"", "<Terse a=b c=\"d\" e='f'/>"};
    for (const char* a : absurd) {
        std::vector<char> m(a, a + std::strlen(a));
        trpx_header t{};
        size_t o = 0;
        (void)trpx_header_parse(m.data(), m.size(), &t, &o);
        uint64_t fs[2];
        (void)trpx_header_frame_sizes(m.data(), m.size(), fs, 2);
    }
}

// ---- TIFF reader -------------------------------------------------------------------------------------------------
static std::string valid_tiff(bool big_endian) {
    // two 7 x 5 16-bit images written by the product's own writer; the big-endian twin is made by hand below
    trpx::Grey_tif tif;
    std::vector<std::uint16_t> a(35), b(35);
    for (int i = 0; i < 35; ++i) { a[i] = (std::uint16_t)(i * 7); b[i] = (std::uint16_t)(60000 - i); }
    std::memcpy(tif.push_back<std::uint16_t>(7, 5), a.data(), 70);
    std::memcpy(tif.push_back<std::uint16_t>(7, 5), b.data(), 70);
    std::ostringstream os;
    tif.write(os);
    std::string s = os.str();
    if (!big_endian) return s;
    // byte-swap header, IFDs and pixels into an MM file
    auto r16 = [&](size_t at) { return (uint16_t)((uint8_t)s[at] | ((uint8_t)s[at + 1] << 8)); };
    auto r32 = [&](size_t at) { return (uint32_t)r16(at) | ((uint32_t)r16(at + 2) << 16); };
    auto w16 = [&](size_t at, uint16_t v) { s[at] = (char)(v >> 8); s[at + 1] = (char)v; };
    auto w32 = [&](size_t at, uint32_t v) { w16(at, (uint16_t)(v >> 16)); w16(at + 2, (uint16_t)v); };
    size_t ifd = r32(4);
    s[0] = 'M'; s[1] = 'M'; w16(2, 42); w32(4, (uint32_t)ifd);
    while (ifd) {
        const unsigned n = r16(ifd);
        size_t pix = 0;
        for (unsigned i = 0; i < n; ++i) {
            const size_t e = ifd + 2 + 12 * i;
            const unsigned tag = r16(e), type = r16(e + 2);
            const uint32_t count = r32(e + 4);
            if (tag == 0x111) pix = r32(e + 8);
            w16(e, (uint16_t)tag); w16(e + 2, (uint16_t)type); w32(e + 4, count);
            if (type == 3 && count == 1) { const uint16_t v = r16(e + 8); w16(e + 8, v); s[e + 10] = s[e + 11] = 0; }
            else w32(e + 8, r32(e + 8));
        }
        for (size_t k = 0; k < 35; ++k) w16(pix + 2 * k, r16(pix + 2 * k));
        const size_t link = ifd + 2 + 12 * n;
        const uint32_t next = r32(link);
        w16(ifd, (uint16_t)n);
        w32(link, next);
        ifd = next;
    }
    return s;
}

static bool try_tiff(const std::string& bytes, std::size_t* images) {
    std::istringstream is(bytes);
    try {
        trpx::Grey_tif t(is);
        *images = t.image_stack_size();
        for (std::size_t i = 0; i < t.image_stack_size(); ++i) {           // touch every pixel the reader vouches for
            const trpx::Tif_image& im = t.image(i);
            const std::uint8_t* p = t.pixels(i);
            unsigned acc = 0;
            for (std::size_t k = 0; k < im.pixels() * im.bytes_per_pixel; ++k) acc += p[k];
            if (acc == 0xFFFFFFFFu) std::printf(" ");
        }
        return true;
    } catch (const std::exception&) {
        return false;
    }
}

static void tiffs() {
    for (int be = 0; be < 2; ++be) {
        const std::string good = valid_tiff(be != 0);
        std::size_t n = 0;
        EXPECT(try_tiff(good, &n) && n == 2);
        for (size_t cut = 0; cut < good.size(); ++cut) { std::size_t k; (void)try_tiff(good.substr(0, cut), &k); }
        for (int it = 0; it < 8000; ++it) {
            std::string m = good;
            const int flips = 1 + (int)(rnd() % 3);
            for (int k = 0; k < flips; ++k) m[rnd() % m.size()] = (char)rnd();
            std::size_t k;
            (void)try_tiff(m, &k);
        }
        // the advisor's case: width 2^31, height 2^30, 64-bit samples -> pixels * 8 wraps to 0; and a 2^32 - 1 entry array
        auto patch = [&](std::string m, unsigned tag, uint32_t value, uint32_t count, bool set_count) {
            const bool mm = m[0] == 'M';
            auto r16 = [&](size_t at) { return mm ? (uint16_t)(((uint8_t)m[at] << 8) | (uint8_t)m[at + 1]) : (uint16_t)((uint8_t)m[at] | ((uint8_t)m[at + 1] << 8)); };
            auto r32 = [&](size_t at) { return mm ? ((uint32_t)r16(at) << 16) | r16(at + 2) : (uint32_t)r16(at) | ((uint32_t)r16(at + 2) << 16); };
            auto w32 = [&](size_t at, uint32_t v) { for (int b = 0; b < 4; ++b) m[at + (mm ? 3 - b : b)] = (char)(v >> (8 * b)); };
            auto w16 = [&](size_t at, uint16_t v) { for (int b = 0; b < 2; ++b) m[at + (mm ? 1 - b : b)] = (char)(v >> (8 * b)); };
            const size_t ifd = r32(4);
            const unsigned cnt = r16(ifd);
            for (unsigned i = 0; i < cnt; ++i) {
                const size_t e = ifd + 2 + 12 * i;
                if (r16(e) == tag) { w16(e + 2, 4); w32(e + 8, value); if (set_count) w32(e + 4, count); }
            }
            return m;
        };
        std::string evil = patch(patch(patch(good, 0x100, 1u << 31, 0, false), 0x101, 1u << 30, 0, false), 0x102, 64, 0, false);
        std::size_t k;
        EXPECT(!try_tiff(evil, &k));
        EXPECT(!try_tiff(patch(good, 0x111, 8, 0xFFFFFFFFu, true), &k));
    }
}

// ---- Terse(std::ifstream&) -----------------------------------------------------------------------------------------
static bool try_trpx(const std::string& bytes, const char* path) {
    { std::ofstream f(path, std::ios::binary); f.write(bytes.data(), (std::streamsize)bytes.size()); }
    std::ifstream f(path, std::ios::binary);
    try {
        trpx::Terse t(f);
        return t.terse_size() <= bytes.size();
    } catch (const std::exception&) {
        return false;
    }
}

static void trpx_files(const char* path) {
    const std::string payload(1152, '\x5a');
    const std::string good = "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"1\"/>" + payload;
    EXPECT(try_trpx(good, path));
    EXPECT(try_trpx("junk in front " + good, path));
    for (size_t cut = 0; cut < good.size(); cut += 7) (void)try_trpx(good.substr(0, cut), path);
    const char* bad[] = {
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"99999999999999\" number_of_values=\"1000\" number_of_frames=\"1\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"18446744073709551615\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"5000\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"0\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"1\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"0\" number_of_frames=\"1\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"3\" frame_sizes=\"1 2 99999999999999999999\"/>",
        "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" number_of_frames=\"3\"/>"};   // needs the device walk: stubbed -> error
    for (const char* b : bad) EXPECT(!try_trpx(std::string(b) + payload, path));
    const std::string three = "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" number_of_values=\"1000\" frame_sizes=\"400 400 352\" number_of_frames=\"3\"/>" + payload;
    EXPECT(try_trpx(three, path));                                             // the frame index in the file: no device walk needed
    for (int it = 0; it < 1500; ++it) {
        std::string m = three.substr(0, 200 + rnd() % 64);
        const int flips = 1 + (int)(rnd() % 4);
        for (int k = 0; k < flips; ++k) m[rnd() % 170] = (char)rnd();
        (void)try_trpx(m, path);
    }
}

int main(int argc, char** argv) {
    headers();
    tiffs();
    trpx_files(argc > 1 ? argv[1] : "/tmp/trpx_host_sanitize.trpx");
    std::printf(failures ? "FAILED (%d)\n" : "OK host_sanitize: header text, TIFF reader and stream constructor survive truncated / corrupt input\n", failures);
    return failures ? 1 : 0;
}
