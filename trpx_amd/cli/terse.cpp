// terse -- compresses greyscale TIFF stacks to .trpx files on the MI355X.
// Command line and behaviour of the reference tool (senikm/trpx src/terse.cpp:20-104, type dispatch :107-125):
//   terse [-help] [-verbose] [-delete] [-index] [file ...]
// every argument with a .tif / .tiff / .TIF / .TIFF extension is read, all images of its stack are pushed into ONE
// Terse object (one device call for the whole stack) and written next to it as <name>.trpx; -verbose prints the
// reference's report.  Differences: the input is kept unless -delete is given (the reference always deletes it,
// terse.cpp:82); float / double TIFFs are converted to 64-bit integers like the reference does (:120-123): values that all
// fit 32 bits take the tuned kernels (the stream is the same), wider ones the generic 64-bit kernels.
#include <chrono>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "trpx/Grey_tif.hpp"
#include "trpx/Terse.hpp"

namespace fs = std::filesystem;

template <typename T>
static void compress_stack(trpx::Terse& out, trpx::Grey_tif const& tif) {
    const std::size_t n = tif.image(0).pixels(), frames = tif.image_stack_size();
    std::vector<T> stack(n * frames);                       // the images of a TIFF file are separated by their IFDs
    for (std::size_t i = 0; i < frames; ++i) std::memcpy(stack.data() + i * n, tif.pixels(i), n * sizeof(T));
    out.dim({tif.image(0).width, tif.image(0).height});     // what push_back(image) captures from image.dim() (Terse.hpp:314-317)
    out.push_back(stack.data(), n, frames);
}

// float / double images: converted to 64-bit integers like the reference does (static_cast, i.e. truncation towards
// zero) and pushed as a plain vector -- which is why such files carry no `dimensions` attribute (terse.cpp:120-123).
static void compress_float_stack(trpx::Terse& out, trpx::Grey_tif const& tif) {
    const std::size_t n = tif.image(0).pixels(), frames = tif.image_stack_size();
    std::vector<std::int64_t> stack(n * frames);
    for (std::size_t i = 0; i < frames; ++i) {
        if (tif.image(i).bytes_per_pixel == 4) {
            float const* p = reinterpret_cast<float const*>(tif.pixels(i));
            for (std::size_t k = 0; k < n; ++k) stack[i * n + k] = static_cast<std::int64_t>(p[k]);
        } else {
            double const* p = reinterpret_cast<double const*>(tif.pixels(i));
            for (std::size_t k = 0; k < n; ++k) stack[i * n + k] = static_cast<std::int64_t>(p[k]);
        }
    }
    out.push_back(stack.data(), n, frames);
}

int main(int argc, char const* argv[]) {
    bool help = false, verbose = false, del = false, index = false;
    std::vector<fs::path> params;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-help") help = true;
        else if (a == "-verbose") verbose = true;
        else if (a == "-delete") del = true;
        else if (a == "-index") index = true;
        else params.emplace_back(a);
    }
    if (help) {
        std::cout << "terse [-help] [-verbose] [-delete] [-index] [file ...]\n"
                     "  compresses all files with .tiff or .tif extensions to terse files with .trpx extensions (on the GPU).\n"
                     "Examples:\n"
                     "   terse *                   // all tiff files in this directory are compressed to trpx files.\n"
                     "   terse ~/dir/my_img*       // compresses all tiff files in the directory ~/dir that start with my_img\n"
                     "\nkeywords:\n"
                     "  -help      print help\n"
                     "  -verbose   print compressed filenames, compute times and compression rate\n"
                     "  -delete    delete each TIFF file after it has been compressed (the reference tool always does)\n"
                     "  -index     add the frame_sizes and group_bit_offsets attributes to the header (ignored by the reference reader;\n"
                     "             they let prolix locate the frames of a stack and expand them without walking their headers)\n";
        return 0;
    }
    std::chrono::duration<double> user_time(0), io_time(0);
    double total_trpx = 0, total_tiff = 0;
    std::size_t compressed_files = 0;
    int rc = 0;
    for (fs::path const& tif_name : params) {
        const std::string ext = tif_name.extension().string();
        if (!fs::is_regular_file(tif_name) || !(ext == ".tiff" || ext == ".tif" || ext == ".TIFF" || ext == ".TIF")) continue;
        try {
            const auto t0 = std::chrono::high_resolution_clock::now();
            std::ifstream in(tif_name, std::ios::binary);
            if (!in.is_open()) {
                std::cerr << "Failed to open input file " << tif_name << std::endl;
                rc = 1;
                continue;
            }
            trpx::Grey_tif tif(in);
            in.close();
            if (tif.image_stack_size() == 0) throw std::runtime_error("TIFF file contains no images.");
            total_tiff += (double)tif.raw_data_size();
            const auto t1 = std::chrono::high_resolution_clock::now();
            trpx::Tif_image const& first = tif.image(0);
            for (std::size_t i = 1; i < tif.image_stack_size(); ++i) {
                trpx::Tif_image const& im = tif.image(i);
                if (im.width != first.width || im.height != first.height)
                    throw std::runtime_error("TIFF file contains a stack of images with varying sizes.");
                if (im.bytes_per_pixel != first.bytes_per_pixel || im.is_signed != first.is_signed || im.is_integral != first.is_integral)
                    throw std::runtime_error("TIFF file contains a stack of images with varying pixel types.");
            }
            trpx::Terse compressed;
            if (!first.is_integral) {                                            // terse.cpp:119-124: float / double -> int64_t
                compress_float_stack(compressed, tif);
            } else if (first.bytes_per_pixel > 4)
                throw std::runtime_error("64-bit integer pixels are not supported.");
            else
            switch (first.bytes_per_pixel * 2 + (first.is_signed ? 1 : 0)) {      // terse.cpp:113-118
            case 2: compress_stack<std::uint8_t>(compressed, tif); break;
            case 3: compress_stack<std::int8_t>(compressed, tif); break;
            case 4: compress_stack<std::uint16_t>(compressed, tif); break;
            case 5: compress_stack<std::int16_t>(compressed, tif); break;
            case 8: compress_stack<std::uint32_t>(compressed, tif); break;
            case 9: compress_stack<std::int32_t>(compressed, tif); break;
            default: throw std::runtime_error("unsupported pixel type.");
            }
            total_trpx += (double)compressed.terse_size();
            if (verbose && !compressed.imagej_readable())
                std::cout << "Note: " << tif_name << " is " << (compressed.is_signed() ? "signed, " : "unsigned, ") << compressed.bits_per_val()
                          << " bits per value: the ImageJ TRPX reader only opens unsigned data of at most 16 bits." << std::endl;
            fs::path trpx_name = tif_name;
            trpx_name.replace_extension(".trpx");
            std::ofstream out(trpx_name, std::ios::binary);
            if (!out.is_open()) throw std::runtime_error("Failed to open trpx file for output.");
            compressed.write(out, index);
            out.close();
            if (del) {
                std::cout << "Deleting original TIFF file: " << tif_name << std::endl;
                fs::remove(tif_name);
            }
            ++compressed_files;
            const auto t2 = std::chrono::high_resolution_clock::now();
            user_time += t2 - t1;
            io_time += t1 - t0;
        } catch (std::exception const& e) {
            std::cerr << "Error processing " << tif_name << ": " << e.what() << std::endl;
            rc = 1;
        }
    }
    if (verbose) {
        for (fs::path const& p : params) std::cout << "Compressed: " << p << std::endl;
        std::cout << "Terse compressed: " << compressed_files << " files\n";
        std::cout << "User time       : " << user_time.count() << " seconds\n";
        std::cout << "IO time         : " << io_time.count() << " seconds\n";
        if (total_tiff > 0) std::cout << "Compression rate: " << std::round(1000 * (1 - total_trpx / total_tiff)) / 10 << "%\n";
    }
    return rc;
}
