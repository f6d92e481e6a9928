// Host check of csrc/cubic_filter.h: cubic_half1 (the folded form the 2:1 RGBA8 mip kernels use) must give the bits of cubic1(0.5f, ...)
// (the reference's CUBIC_INTERPOLATE, operation for operation) on every input: 8-bit codes / 255, x-filtered values, tiny and large
// magnitudes. Built with the flags of the reference-faithful kernels (-ffp-contract=off, no fast-math); prints the mismatch count.
#include "../../directxtex_amd/csrc/cubic_filter.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

int main(int argc, char** argv)
{
    const long n = argc > 1 ? std::atol(argv[1]) : 20000000;
    uint32_t s = 12345;
    long bad = 0;
    for (long i = 0; i < n; ++i)
    {
        float p[4];
        const int kind = int(i & 3);
        for (int k = 0; k < 4; ++k)
        {
            const uint32_t r = lcg(s);
            const float u = float(r >> 8) * (1.6f / 16777216.0f) - 0.3f;           // [-0.3, 1.3)
            p[k] = kind == 0 ? float(r >> 24) * (1.0f / 255.0f) : kind == 1 ? u : kind == 2 ? u * 1e-20f : u * 3e4f;
        }
        if (kind == 1 && (i & 4)) { p[1] = p[0]; p[2] = p[0]; }                   // flat neighbourhoods: exact zeros in the differences
        const float x = dxtex::cubic1(0.5f, p[0], p[1], p[2], p[3]), y = dxtex::cubic_half1(p[0], p[1], p[2], p[3]);
        uint32_t a, b; std::memcpy(&a, &x, 4); std::memcpy(&b, &y, 4);
        if (a != b) { if (bad < 5) std::printf("differ at %g %g %g %g: %a vs %a\n", p[0], p[1], p[2], p[3], x, y); ++bad; }
    }
    std::printf("%ld of %ld differ\n", bad, n);
    return bad ? 1 : 0;
}
