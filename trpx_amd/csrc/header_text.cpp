// Stream-serialise surface, host only: the ASCII header that precedes the raw stack in a .trpx
// file.  trpx_header_format reproduces jpa::Terse::write byte for byte (reference
// include/Terse.hpp:454-470: attribute order prolix_bits, signed, block, memory_size,
// number_of_values, [dimensions], number_of_frames; `dimensions` only when set).
// trpx_header_parse accepts what the reference reader accepts (Terse.hpp:485-498 over
// XML_element.hpp:216-224, :296-307, :428-452): arbitrary bytes before "<Terse", attributes in
// any order, either quote character, unknown attributes ignored.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include "../../include/trpx_hip.h"

static std::string header_string(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes, const uint64_t* group_states = nullptr,
                                 size_t n_states = 0);

extern "C" size_t trpx_header_format(const trpx_header* h, char* buf, size_t buf_cap) {
    if (!h || !buf) return 0;
    const std::string s = header_string(h, nullptr, 0);
    if (s.size() + 1 > buf_cap) return 0;
    memcpy(buf, s.c_str(), s.size() + 1);
    return s.size();
}

extern "C" size_t trpx_header_format_indexed(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes, char* buf,
                                             size_t buf_cap) {
    if (!h || !buf || (!frame_sizes && n_sizes)) return 0;
    const std::string s = header_string(h, frame_sizes, n_sizes);
    if (s.size() + 1 > buf_cap) return 0;
    memcpy(buf, s.c_str(), s.size() + 1);
    return s.size();
}

extern "C" size_t trpx_header_format_grouped(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes,
                                             const uint64_t* group_states, size_t n_states, char* buf, size_t buf_cap) {
    if (!h || !buf || (!frame_sizes && n_sizes) || (!group_states && n_states)) return 0;
    const std::string s = header_string(h, frame_sizes, n_sizes, group_states, n_states);
    if (s.size() + 1 > buf_cap) return 0;
    memcpy(buf, s.c_str(), s.size() + 1);
    return s.size();
}

static std::string header_string(const trpx_header* h, const uint64_t* frame_sizes, size_t n_sizes, const uint64_t* group_states,
                                 size_t n_states) {
    std::string s = "<Terse prolix_bits=\"" + std::to_string(h->prolix_bits) + "\"";
    s += " signed=\"" + std::to_string(h->is_signed ? 1 : 0) + "\"";
    s += " block=\"" + std::to_string(h->block) + "\"";
    s += " memory_size=\"" + std::to_string((unsigned long long)h->memory_size) + "\"";
    s += " number_of_values=\"" + std::to_string((unsigned long long)h->number_of_values) + "\"";
    if (h->n_dims) {
        s += " dimensions=\"";
        for (unsigned i = 0; i < h->n_dims && i < 8; ++i) {
            if (i) s += " ";
            s += std::to_string((unsigned long long)h->dims[i]);
        }
        s += "\"";
    }
    if (n_sizes) {                                            // row f1: optional, ignored by the reference reader
        s += " frame_sizes=\"";
        for (size_t i = 0; i < n_sizes; ++i) {
            if (i) s += " ";
            s += std::to_string((unsigned long long)frame_sizes[i]);
        }
        s += "\"";
    }
    if (n_states) {                                           // row f1: chain state at every 256th block, "bit_offset:width_before"
        s += " group_bit_offsets=\"";
        for (size_t i = 0; i < n_states; ++i) {
            if (i) s += " ";
            s += std::to_string((unsigned long long)(group_states[i] & ((1ull << 40) - 1))) + ":" + std::to_string((unsigned)(group_states[i] >> 40));
        }
        s += "\"";
    }
    s += " number_of_frames=\"" + std::to_string((unsigned long long)h->number_of_frames) + "\"/>";
    return s;
}

static bool is_white(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }

extern "C" int trpx_header_parse(const char* data, size_t len, trpx_header* h, size_t* payload_offset) {
    if (!data || !h) return TRPX_ERR_INVALID_ARG;
    static const char tag[] = "<Terse";
    const size_t tl = sizeof(tag) - 1;
    size_t p = 0;
    bool found = false;
    for (; p + tl < len; ++p)
        if (data[p] == '<' && memcmp(data + p, tag, tl) == 0 && (is_white(data[p + tl]) || data[p + tl] == '/' || data[p + tl] == '>')) {
            found = true;
            break;
        }
    if (!found) return TRPX_ERR_CORRUPT;
    size_t q = p + tl;
    memset(h, 0, sizeof *h);
    bool got_pb = false, got_sg = false, got_bl = false, got_ms = false, got_nv = false, got_nf = false;
    while (q < len && data[q] != '>') {
        if (is_white(data[q]) || data[q] == '/') { ++q; continue; }
        size_t ns = q;
        while (q < len && data[q] != '=' && data[q] != '>' && !is_white(data[q])) ++q;
        std::string name(data + ns, q - ns);
        while (q < len && is_white(data[q])) ++q;
        if (q >= len || data[q] != '=') return TRPX_ERR_CORRUPT;
        ++q;
        while (q < len && is_white(data[q])) ++q;
        if (q >= len) return TRPX_ERR_CORRUPT;
        const char quote = data[q++];
        size_t vs = q;
        while (q < len && data[q] != quote) ++q;
        if (q >= len) return TRPX_ERR_CORRUPT;
        std::string val(data + vs, q - vs);
        ++q;
        char* end = nullptr;
        if (name == "prolix_bits") { h->prolix_bits = (unsigned)strtoul(val.c_str(), &end, 10); got_pb = end != val.c_str(); }
        else if (name == "signed") { h->is_signed = strtoul(val.c_str(), &end, 10) != 0; got_sg = end != val.c_str(); }
        else if (name == "block") { h->block = (unsigned)strtoul(val.c_str(), &end, 10); got_bl = end != val.c_str(); }
        else if (name == "memory_size") { h->memory_size = (uint64_t)strtold(val.c_str(), &end); got_ms = end != val.c_str(); }
        else if (name == "number_of_values") { h->number_of_values = strtoull(val.c_str(), &end, 10); got_nv = end != val.c_str(); }
        else if (name == "number_of_frames") { h->number_of_frames = strtoull(val.c_str(), &end, 10); got_nf = end != val.c_str(); }
        else if (name == "dimensions") {
            const char* s = val.c_str();
            while (*s && h->n_dims < 8) {
                unsigned long long d = strtoull(s, &end, 10);
                if (end == s) break;
                h->dims[h->n_dims++] = d;
                s = end;
            }
        }
    }
    if (q >= len) return TRPX_ERR_CORRUPT;
    // the reference's stoul/stoull/stold throw on a missing attribute
    if (!(got_pb && got_sg && got_bl && got_ms && got_nv && got_nf)) return TRPX_ERR_CORRUPT;
    if (payload_offset) *payload_offset = q + 1;
    return TRPX_OK;
}

// Value of one attribute of the <Terse .../> element: same scan as trpx_header_parse.  Returns false if it is not there.
static bool find_attribute(const char* data, size_t len, const char* wanted, std::string* value) {
    if (!data) return false;
    static const char tag[] = "<Terse";
    const size_t tl = sizeof(tag) - 1;
    size_t p = 0;
    for (;; ++p) {
        if (p + tl >= len) return false;
        if (data[p] == '<' && memcmp(data + p, tag, tl) == 0 && (is_white(data[p + tl]) || data[p + tl] == '/' || data[p + tl] == '>')) break;
    }
    size_t q = p + tl;
    while (q < len && data[q] != '>') {
        if (is_white(data[q]) || data[q] == '/') { ++q; continue; }
        const size_t ns = q;
        while (q < len && data[q] != '=' && data[q] != '>' && !is_white(data[q])) ++q;
        const std::string name(data + ns, q - ns);
        while (q < len && is_white(data[q])) ++q;
        if (q >= len || data[q] != '=') return false;
        ++q;
        while (q < len && is_white(data[q])) ++q;
        if (q >= len) return false;
        const char quote = data[q++];
        const size_t vs = q;
        while (q < len && data[q] != quote) ++q;
        if (q >= len) return false;
        if (name == wanted) {
            value->assign(data + vs, q - vs);
            return true;
        }
        ++q;
    }
    return false;
}

// The optional frame_sizes attribute (trpx_header_format_indexed).
extern "C" size_t trpx_header_frame_sizes(const char* data, size_t len, uint64_t* frame_sizes, size_t capacity) {
    std::string val;
    if (!find_attribute(data, len, "frame_sizes", &val)) return 0;
    const char* t = val.c_str();
    size_t n = 0;
    for (;;) {
        char* end = nullptr;
        const unsigned long long v = strtoull(t, &end, 10);
        if (end == t) break;
        if (frame_sizes && n < capacity) frame_sizes[n] = v;
        ++n;
        t = end;
    }
    return n;
}

// The optional group_bit_offsets attribute (trpx_header_format_grouped): tokens "offset:width"; a malformed token ends the list.
extern "C" size_t trpx_header_group_states(const char* data, size_t len, uint64_t* group_states, size_t capacity) {
    std::string val;
    if (!find_attribute(data, len, "group_bit_offsets", &val)) return 0;
    const char* t = val.c_str();
    size_t n = 0;
    for (;;) {
        char* end = nullptr;
        const unsigned long long off = strtoull(t, &end, 10);
        if (end == t || *end != ':') break;
        t = end + 1;
        const unsigned long long w = strtoull(t, &end, 10);
        if (end == t) break;
        if (off >= (1ull << 40) || w > 64) break;
        if (group_states && n < capacity) group_states[n] = off | ((uint64_t)w << 40);
        ++n;
        t = end;
    }
    return n;
}
