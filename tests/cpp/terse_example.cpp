// C++ drop-in check of trpx::Terse (include/trpx/Terse.hpp) -- reads like the reference's own
// (disabled) test, test/terse_tests.cpp:15-33: compress iota(-500..499) as int, write to a file,
// read it back, prolix, compare -- plus a multi-frame u16 stack and the error conventions.
// Needs a GPU: every encode / decode goes through libtrpx_hip.so.
#include <cstdio>
#include <fstream>
#include <iostream>
#include <numeric>
#include <sstream>
#include <vector>
#include "trpx/Terse.hpp"

namespace jpa = trpx;   // a caller written against jpa::Terse switches with this one line

#define REQUIRE(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/junk.terse";
    {   // README / Terse.hpp:127-154 example
        std::vector<int> numbers(1000);
        std::iota(numbers.begin(), numbers.end(), -500);
        jpa::Terse compressed(numbers);
        REQUIRE(compressed.terse_size() == 1152);         // "compression rate 0.29" (Terse.hpp:129)
        REQUIRE(compressed.bits_per_val() == 10);
        REQUIRE(compressed.is_signed());
        std::ostringstream hdr;
        const jpa::Terse& as_const = compressed;               // write() is const like the reference's (Terse.hpp:454)
        as_const.write(hdr);
        jpa::Terse assigned;                                   // assignable and movable like the reference class
        assigned = compressed;
        jpa::Terse moved(std::move(assigned));
        REQUIRE(moved.terse_size() == compressed.terse_size() && moved.bits_per_val() == 10);
        const std::string want = "<Terse prolix_bits=\"10\" signed=\"1\" block=\"12\" memory_size=\"1152\" "
                                 "number_of_values=\"1000\" number_of_frames=\"1\"/>";
        REQUIRE(hdr.str().substr(0, want.size()) == want);
        std::ofstream outfile(path, std::ios::binary);
        compressed.write(outfile);
        outfile.close();
        std::ifstream infile(path, std::ios::binary);
        jpa::Terse from_file(infile);
        std::vector<int> uncompressed(1000);
        from_file.prolix(uncompressed.begin());
        REQUIRE(uncompressed == numbers);
    }
    {   // 3-frame u16 stack with dimensions, frame >= 2 decodes correctly (reference defect D1)
        const std::size_t n = 35 * 20;
        std::vector<std::uint16_t> stack(3 * n);
        for (std::size_t i = 0; i < stack.size(); ++i) stack[i] = (std::uint16_t)((i * 2654435761u >> 27) & (i < n ? 7 : i < 2 * n ? 255 : 0));
        jpa::Terse t;
        for (int f = 0; f < 3; ++f) t.push_back(stack.data() + f * n, n);
        t.dim({35, 20});
        REQUIRE(t.number_of_frames() == 3 && t.size() == n && !t.is_signed());
        jpa::Terse batched;
        batched.push_back(stack.data(), n, 3);            // one device call
        REQUIRE(batched.data() == t.data() && batched.frame_sizes() == t.frame_sizes());
        std::ofstream o(path, std::ios::binary);
        t.write(o);
        o.close();
        std::ifstream in(path, std::ios::binary);
        jpa::Terse r(in);
        REQUIRE(r.number_of_frames() == 3 && r.dim() == std::vector<std::size_t>({35, 20}));
        REQUIRE(r.frame_sizes() == t.frame_sizes());
        for (int f = 2; f >= 0; --f) {
            std::vector<std::uint16_t> back(n);
            r.prolix(back, f);
            REQUIRE(std::equal(back.begin(), back.end(), stack.begin() + f * n));
        }
        bool threw = false;
        try { std::vector<std::uint16_t> small(n - 1); r.prolix(small, 0); } catch (std::invalid_argument const&) { threw = true; }
        REQUIRE(threw);
    }
    std::remove(path);
    std::printf("OK\n");
    return 0;
}
