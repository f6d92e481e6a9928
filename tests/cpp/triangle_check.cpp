// Dumps the triangle-filter gather lists the product builds on the host (directxtex_amd/csrc/triangle_filter.h) for
// tests/test_bounds_cpu.py, which compares them bit for bit with the reference's CreateTriangleFilter (filters.h:249-419, through
// oracle/_ref). usage: triangle_check <source> <dest> <wrap> ...   -> one line per triple: "n  dst:src:weightbits ..." in gather order
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <vector>
#include "../../directxtex_amd/csrc/triangle_filter.h"

int main(int argc, char** argv)
{
    for (int i = 1; i + 2 < argc; i += 3)
    {
        const size_t source = std::strtoull(argv[i], nullptr, 10), dest = std::strtoull(argv[i + 1], nullptr, 10);
        const bool wrap = std::atoi(argv[i + 2]) != 0;
        std::vector<uint32_t> ofs; std::vector<dxtex::TriEntry> ent;
        dxtex::build_triangle_axis(source, dest, wrap, ofs, ent);
        std::printf("%zu", ent.size());
        for (size_t d = 0; d < dest; ++d)
            for (uint32_t k = ofs[d]; k < ofs[d + 1]; ++k) std::printf(" %zu:%u:%u", d, ent[k].src, ent[k].weightBits);
        std::puts("");
    }
    return 0;
}
