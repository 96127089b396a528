// prolix -- expands .trpx files to greyscale TIFF stacks on the MI355X.
// Command line and behaviour of the reference tool (senikm/trpx src/prolix.cpp:18-128):
//   prolix [-help] [-verbose] [-delete] [file ...]
// every argument with a .trpx extension is read (Terse(std::ifstream&)), the pixel type is chosen from bits_per_val() and
// is_signed() (16-bit up to 16 bits, else 32-bit; :69-92), all frames are decoded in one device call into a TIFF stack
// written next to it as <name>.tif.  Differences: 32-bit stacks are decoded into 32-bit images (the reference passes
// image<int16_t> there, defect D5), frames >= 2 of a stack are located correctly (D1/D2), and the input is kept unless
// -delete is given (the reference always deletes it, :111).
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
static void expand_stack(trpx::Terse& data, std::uint32_t w, std::uint32_t h, trpx::Grey_tif& tif) {
    const std::size_t n = data.size(), frames = data.number_of_frames();
    std::vector<T> stack(n * frames);
    data.prolix_all(stack.data());
    for (std::size_t i = 0; i < frames; ++i) {
        T* dst = tif.push_back<T>(w, h);
        std::memcpy(dst, stack.data() + i * n, n * sizeof(T));
    }
}

int main(int argc, char const* argv[]) {
    bool help = false, verbose = false, del = false;
    std::vector<fs::path> params;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-help") help = true;
        else if (a == "-verbose") verbose = true;
        else if (a == "-delete") del = true;
        else params.emplace_back(a);
    }
    if (help) {
        std::cout << "prolix [-help] [-verbose] [-delete] [file ...]\n"
                     "  expands trpx files to tiff files (on the GPU).\n"
                     "Examples:\n"
                     "   prolix *              // all TRPX files with .trpx extensions are expanded to tiff files with .tif extensions.\n"
                     "   prolix ~/dir/my_img*  // decompresses all trpx files in the directory ~/dir that start with my_img\n"
                     "\nkeywords:\n"
                     "  -help      print help\n"
                     "  -verbose   print expanded file names and compute times\n"
                     "  -delete    delete each trpx file after it has been expanded (the reference tool always does)\n";
        return 0;
    }
    std::chrono::duration<double> user_time(0), io_time(0);
    std::size_t expanded_files = 0;
    int rc = 0;
    for (fs::path const& name : params) {
        if (!fs::is_regular_file(name) || name.extension() != ".trpx") continue;
        try {
            auto t0 = std::chrono::high_resolution_clock::now();
            std::ifstream in(name, std::ios::binary);
            if (!in.is_open()) {
                std::cerr << "Failed to open input file " << name << std::endl;
                rc = 1;
                continue;
            }
            trpx::Terse data(in);
            in.close();
            io_time += std::chrono::high_resolution_clock::now() - t0;
            const auto t1 = std::chrono::high_resolution_clock::now();
            std::uint32_t w, h;
            if (data.dim().empty()) w = h = (std::uint32_t)std::sqrt((double)data.size());   // no dimensions: assume a square image (:64-65)
            else if (data.dim().size() >= 2) { w = (std::uint32_t)data.dim()[0]; h = (std::uint32_t)data.dim()[1]; }
            else { w = (std::uint32_t)data.dim()[0]; h = 1; }
            if ((std::size_t)w * h != data.size()) throw std::runtime_error("frame dimensions do not match the number of values.");
            trpx::Grey_tif tif;
            if (data.bits_per_val() <= 16) {
                if (data.is_signed()) expand_stack<std::int16_t>(data, w, h, tif);
                else expand_stack<std::uint16_t>(data, w, h, tif);
            } else if (data.bits_per_val() <= 32) {
                if (data.is_signed()) expand_stack<std::int32_t>(data, w, h, tif);
                else expand_stack<std::uint32_t>(data, w, h, tif);
            } else {
                std::cerr << "Terse file " << name << " encodes data that requires 64 bits per pixel." << std::endl;
                std::cerr << "Prolix cannot process such trpx-stacks." << std::endl;
                rc = 1;
                continue;
            }
            user_time += std::chrono::high_resolution_clock::now() - t1;
            t0 = std::chrono::high_resolution_clock::now();
            fs::path tif_name = name;
            tif_name.replace_extension(".tif");
            std::ofstream out(tif_name, std::ios::binary);
            if (!out.is_open()) throw std::runtime_error("Failed to open tiff file for output.");
            tif.write(out);
            out.close();
            if (del) fs::remove(name);
            ++expanded_files;
            io_time += std::chrono::high_resolution_clock::now() - t0;
        } catch (std::exception const& e) {
            std::cerr << "Error processing " << name << ": " << e.what() << std::endl;
            rc = 1;
        }
    }
    if (verbose) {
        for (fs::path const& p : params) std::cout << "Expanded: " << p << std::endl;
        std::cout << "Prolix expanded : " << expanded_files << " files\n";
        std::cout << "User time       : " << user_time.count() << " seconds\n";
        std::cout << "IO time         : " << io_time.count() << " seconds\n";
    }
    return rc;
}
