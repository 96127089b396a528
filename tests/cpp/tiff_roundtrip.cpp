// CPU-only check of include/trpx/Grey_tif.hpp: reads a TIFF stack and writes it again, every image converted to
// 16- or 32-bit pixels of the same signedness (what `prolix` writes) or left as it is.  tests/test_cli_tiff.py compares
// the output byte for byte with files in the layout of the reference's writer.
//   tiff_roundtrip in.tif out.tif [16|32|same]      prints: <images> <width> <height> <bytes per pixel> <signed> <integral>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "trpx/Grey_tif.hpp"

template <typename S, typename D>
static void copy_image(trpx::Grey_tif const& in, std::size_t i, trpx::Grey_tif& out) {
    trpx::Tif_image const& im = in.image(i);
    D* dst = out.push_back<D>(im.width, im.height);
    S const* src = reinterpret_cast<S const*>(in.pixels(i));
    for (std::size_t k = 0; k < im.pixels(); ++k) dst[k] = (D)src[k];
}

template <typename D>
static void convert(trpx::Grey_tif const& in, std::size_t i, trpx::Grey_tif& out) {
    trpx::Tif_image const& im = in.image(i);
    switch (im.bytes_per_pixel * 2 + (im.is_signed ? 1 : 0)) {
    case 2: copy_image<std::uint8_t, D>(in, i, out); break;
    case 3: copy_image<std::int8_t, D>(in, i, out); break;
    case 4: copy_image<std::uint16_t, D>(in, i, out); break;
    case 5: copy_image<std::int16_t, D>(in, i, out); break;
    case 8: copy_image<std::uint32_t, D>(in, i, out); break;
    case 9: copy_image<std::int32_t, D>(in, i, out); break;
    default: throw std::runtime_error("unsupported pixel type");
    }
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        std::ifstream f(argv[1], std::ios::binary);
        trpx::Grey_tif in(f);
        const std::string mode = argc > 3 ? argv[3] : "same";
        trpx::Grey_tif out;
        for (std::size_t i = 0; i < in.image_stack_size(); ++i) {
            trpx::Tif_image const& im = in.image(i);
            if (!im.is_integral) throw std::runtime_error("float pixels");
            const unsigned bytes = mode == "16" ? 2 : mode == "32" ? 4 : im.bytes_per_pixel;
            if (bytes == 1) { if (im.is_signed) convert<std::int8_t>(in, i, out); else convert<std::uint8_t>(in, i, out); }
            else if (bytes == 2) { if (im.is_signed) convert<std::int16_t>(in, i, out); else convert<std::uint16_t>(in, i, out); }
            else { if (im.is_signed) convert<std::int32_t>(in, i, out); else convert<std::uint32_t>(in, i, out); }
        }
        std::ofstream o(argv[2], std::ios::binary);
        out.write(o);
        trpx::Tif_image const& a = in.image(0);
        std::cout << in.image_stack_size() << " " << a.width << " " << a.height << " " << a.bytes_per_pixel << " " << a.is_signed << " "
                  << a.is_integral << "\n";
    } catch (std::exception const& e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
