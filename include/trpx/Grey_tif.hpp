// trpx::Grey_tif -- minimal uncompressed greyscale TIFF stack reader / writer for the `terse` and `prolix`
// command line tools.  It covers what the reference's jpa::Grey_tif<std::byte>
// (senikm/trpx include/Grey_tif.hpp:364-375 reader, :675-827 IFD parser, :477-557 / :602-625 writer) is used for by
// src/terse.cpp and src/prolix.cpp:
//   * reading: "II" or "MM" files, a chain of IFDs, one image per IFD, 8/16/32/64-bit samples (tag 0x102),
//     uncompressed (0x103 == 1), greyscale (0x106 <= 1, 0x115 == 1), one strip or consecutive strips (0x111 / 0x117),
//     sample format from tag 0x153 (1 unsigned, 2 signed, 3 float); pixels are converted to host byte order in place;
//   * writing: byte-for-byte the layout the reference writes -- 8-byte header, then per image: pixel data, one pad
//     byte if the file length is odd, a 7-entry IFD (0x100 width, 0x101 height, 0x102 bits, 0x103 = 1, 0x106 = 1,
//     0x111 data offset, 0x153 sample format) and the 4-byte offset of the next IFD.
// Unlike the reference it checks every offset against the file size and throws std::runtime_error.
#ifndef TRPX_GREY_TIF_HPP
#define TRPX_GREY_TIF_HPP

#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace trpx {

struct Tif_image {
    std::uint32_t width = 0, height = 0;      // dim() of the reference = {width, height}
    unsigned bytes_per_pixel = 0;             // 1, 2, 4, 8
    bool is_signed = false, is_integral = true;
    std::size_t offset = 0;                   // of the first pixel in the file image
    std::size_t pixels() const { return std::size_t(width) * height; }
};

class Grey_tif {
public:
    /// An empty little-endian stack (Grey_tif.hpp:336-343).
    Grey_tif() : d_tif(8, 0) {
        d_tif[0] = d_tif[1] = 'I';
        put16(2, 42);
    }

    /// Reads a whole TIFF stream (Grey_tif.hpp:364-375).
    explicit Grey_tif(std::istream& is) {
        is.seekg(0, std::ios::end);
        const std::streamoff size = is.tellg();
        is.seekg(0, std::ios::beg);
        if (size < 8) throw std::runtime_error("not a TIFF file");
        d_tif.resize((std::size_t)size);
        if (!is.read(reinterpret_cast<char*>(d_tif.data()), size)) throw std::runtime_error("cannot read the TIFF file");
        const bool little = d_tif[0] == 'I' && d_tif[1] == 'I', big = d_tif[0] == 'M' && d_tif[1] == 'M';
        if (!little && !big) throw std::runtime_error("not a TIFF file");
        d_swap = big;                                           // this code runs on little-endian hosts only (x86-64 + gfx950)
        if (get16(2) != 42) throw std::runtime_error("not a TIFF file");
        scan();
    }

    std::size_t image_stack_size() const { return d_img.size(); }
    Tif_image const& image(std::size_t i) const { return d_img.at(i); }
    std::uint8_t const* pixels(std::size_t i) const { return d_tif.data() + d_img.at(i).offset; }
    std::uint8_t* pixels(std::size_t i) { return d_tif.data() + d_img.at(i).offset; }
    std::size_t raw_data_size() const { return d_tif.size(); }  // Grey_tif.hpp:458

    /// Appends an all-zero image of pixel type T and returns its pixels (Grey_tif.hpp:602-625).
    template <typename T>
    T* push_back(std::uint32_t width, std::uint32_t height) {
        static_assert(std::is_arithmetic_v<T> && !std::is_same_v<T, bool>);
        const std::size_t bytes = std::size_t(width) * height * sizeof(T);
        if (d_tif.size() + bytes + 91 > 0xFFFFFFFFull) throw std::runtime_error("TIFF files are limited to 4 GB");
        const std::size_t data_start = d_tif.size();
        d_tif.resize(data_start + bytes);
        if (d_tif.size() & 1) d_tif.push_back(0);
        std::size_t at = d_tif.size();
        d_tif.resize(at + 2 + 7 * 12 + 4);
        put32(d_last_ifd_link, (std::uint32_t)at);
        put16(at, 7);
        at += 2;
        const std::uint32_t fmt = std::is_integral_v<T> ? (std::is_signed_v<T> ? 2u : 1u) : 3u;
        const std::uint32_t entries[7][3] = {{0x100, 3, width}, {0x101, 3, height}, {0x102, 3, 8 * (std::uint32_t)sizeof(T)}, {0x103, 3, 1},
                                             {0x106, 3, 1},     {0x111, 4, (std::uint32_t)data_start}, {0x153, 3, fmt}};
        for (auto const& e : entries) {
            put16(at, (std::uint16_t)e[0]);
            put16(at + 2, (std::uint16_t)e[1]);
            put32(at + 4, 1);
            if (e[1] == 3) put16(at + 8, (std::uint16_t)e[2]);
            else put32(at + 8, e[2]);
            at += 12;
        }
        d_last_ifd_link = at;                                    // the next-IFD offset, zero for now
        Tif_image img;
        img.width = width;
        img.height = height;
        img.bytes_per_pixel = sizeof(T);
        img.is_signed = std::is_signed_v<T>;
        img.is_integral = std::is_integral_v<T>;
        img.offset = data_start;
        d_img.push_back(img);
        return reinterpret_cast<T*>(d_tif.data() + data_start);  // (valid until the next push_back)
    }

    void write(std::ostream& os) const { os.write(reinterpret_cast<char const*>(d_tif.data()), (std::streamsize)d_tif.size()); }

private:
    std::vector<std::uint8_t> d_tif;
    std::vector<Tif_image> d_img;
    std::size_t d_last_ifd_link = 4;
    bool d_swap = false;

    void need(std::size_t at, std::size_t n) const {
        if (at > d_tif.size() || n > d_tif.size() - at) throw std::runtime_error("corrupt TIFF file: offset beyond the end");
    }
    std::uint16_t get16(std::size_t at) const {
        need(at, 2);
        std::uint16_t v;
        std::memcpy(&v, &d_tif[at], 2);
        return d_swap ? (std::uint16_t)((v << 8) | (v >> 8)) : v;
    }
    std::uint32_t get32(std::size_t at) const {
        need(at, 4);
        std::uint32_t v;
        std::memcpy(&v, &d_tif[at], 4);
        return d_swap ? __builtin_bswap32(v) : v;
    }
    void put16(std::size_t at, std::uint16_t v) { std::memcpy(&d_tif[at], &v, 2); }
    void put32(std::size_t at, std::uint32_t v) { std::memcpy(&d_tif[at], &v, 4); }

    // value of an IFD entry of type BYTE / SHORT / LONG (the only ones the tags below use)
    std::uint32_t entry_value(std::size_t at, unsigned type) const {
        if (type == 1 || type == 2 || type == 6 || type == 7) { need(at, 1); return d_tif[at]; }
        if (type == 3 || type == 8) return get16(at);
        return get32(at);
    }
    std::vector<std::uint32_t> entry_array(std::size_t at, unsigned type, std::uint32_t count) const {
        const unsigned esz = (type == 3 || type == 8) ? 2 : 4;
        if (count > d_tif.size() / esz) throw std::runtime_error("corrupt TIFF file: array longer than the file");   // before allocating
        std::vector<std::uint32_t> v(count);
        const std::size_t src = (std::size_t)count * esz <= 4 ? at : get32(at);
        for (std::uint32_t i = 0; i < count; ++i) v[i] = esz == 2 ? get16(src + 2 * (std::size_t)i) : get32(src + 4 * (std::size_t)i);
        return v;
    }

    void scan() {                                                // Grey_tif.hpp:675-708 / :710-827
        std::size_t ifd = get32(4);
        std::size_t guard = 0;
        while (ifd != 0) {
            if (++guard > (1u << 24)) throw std::runtime_error("corrupt TIFF file: IFD loop");
            const unsigned n = get16(ifd);
            Tif_image img;
            unsigned bits = 0;
            std::vector<std::uint32_t> strip_off(1, 0), strip_len;
            for (unsigned i = 0; i < n; ++i) {
                const std::size_t e = ifd + 2 + 12 * (std::size_t)i;
                const unsigned tag = get16(e), type = get16(e + 2);
                const std::uint32_t count = get32(e + 4);
                const std::uint32_t val = entry_value(e + 8, type);
                switch (tag) {
                case 0x100: img.width = val; break;
                case 0x101: img.height = val; break;
                case 0x102:
                    if (val != 8 && val != 16 && val != 32 && val != 64) throw std::runtime_error("Incompatible TIFF file: only 8-, 16-, 32- or 64-bit grey pixels");
                    bits = val;
                    break;
                case 0x103: if (val != 1) throw std::runtime_error("Incompatible TIFF file: compressed"); break;
                case 0x106: if (val > 1) throw std::runtime_error("Incompatible TIFF file: colour"); break;
                case 0x111: strip_off = count == 1 ? std::vector<std::uint32_t>(1, val) : entry_array(e + 8, type, count); break;
                case 0x115: if (val != 1) throw std::runtime_error("Incompatible TIFF file: more than one sample per pixel"); break;
                case 0x117: strip_len = count == 1 ? std::vector<std::uint32_t>(1, val) : entry_array(e + 8, type, count); break;
                case 0x153: img.is_signed = val != 1; img.is_integral = val != 3; break;
                default: break;
                }
            }
            if (!bits || !img.width || !img.height || strip_off.empty()) throw std::runtime_error("Incompatible TIFF file: missing tags");
            for (std::size_t i = 0; i + 1 < strip_off.size(); ++i)
                if (i >= strip_len.size() || strip_len[i] != strip_off[i + 1] - strip_off[i])
                    throw std::runtime_error("Incompatible TIFF file: non-consecutive strips");
            img.bytes_per_pixel = bits / 8;
            img.offset = strip_off[0];
            if (img.pixels() > d_tif.size() / img.bytes_per_pixel)                  // (a product could wrap: 2^31 x 2^30 x 8 bytes = 2^64)
                throw std::runtime_error("corrupt TIFF file: image larger than the file");
            need(img.offset, img.pixels() * img.bytes_per_pixel);
            if (d_swap && img.bytes_per_pixel > 1) {             // pixels to host byte order, in place
                std::uint8_t* p = d_tif.data() + img.offset;
                for (std::size_t k = 0; k < img.pixels(); ++k, p += img.bytes_per_pixel)
                    for (unsigned a = 0, b = img.bytes_per_pixel - 1; a < b; ++a, --b) std::swap(p[a], p[b]);
            }
            d_img.push_back(img);
            d_last_ifd_link = ifd + 2 + 12 * (std::size_t)n;
            ifd = get32(d_last_ifd_link);
        }
    }
};

}  // namespace trpx
#endif
