// development statistic: Newton trips of bc7_rough_kernel's fits per wavefront (max over 64 lanes) under different dealings
#define DXTEX_HOST_DEBUG 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../directxtex_amd/csrc/bc67_tables.h"
#include "../../directxtex_amd/csrc/bc7_core.h"
using namespace dxtex; using namespace dxtex::bc7;
int main(int argc, char** argv)
{
    const int n = atoi(argv[2]);
    std::vector<uint8_t> tiles(size_t(n) * 64);
    FILE* f = fopen(argv[1], "rb"); fread(tiles.data(), 1, tiles.size(), f); fclose(f);
    std::vector<int> trips(size_t(n) * 128), nps(size_t(n) * 128);
    long hist[10] = {};
    for (int t = 0; t < n; ++t)
    {
        float fp[64]; bool opaque = true;
        for (int i = 0; i < 64; ++i) fp[i] = float(tiles[size_t(t) * 64 + i]) * (1.0f / 255.0f);
        for (int i = 0; i < 16; ++i) if (fp[i * 4 + 3] != 1.0f) opaque = false;
        for (int code = 0; code < 128; ++code)
        {
            const uint32_t m1 = kPart2Mask[code >> 1];
            const uint32_t m = (code & 1u) ? m1 : ((~m1) & 0xFFFFu);
            const int np = __builtin_popcount(m);
            int it = 0;
            if (np > 2)
            {
                float X[4], Y[4];
                bool run = opaque ? fit_setup<true, false, true>(fp, m, X, Y) : fit_setup<true>(fp, m, X, Y);
                if (run) for (int k = 0; k < 8; ++k) { ++it; if (opaque ? fit_iterate<true, false, true>(fp, m, X, Y) : fit_iterate<true>(fp, m, X, Y)) break; }
            }
            trips[size_t(t) * 128 + code] = it; nps[size_t(t) * 128 + code] = np; ++hist[it];
        }
    }
    printf("trips histogram:"); for (int i = 0; i <= 8; ++i) printf(" %d:%.1f%%", i, 100.0 * hist[i] / (n * 128.0)); printf("\n");
    // current dealing: wave w of a workgroup of 4 tiles takes group w then group 7 - w (16 entries of kFit2Order each) of all 4 tiles
    double cur = 0, ideal = 0, inter = 0, curT = 0, idealT = 0, interT = 0;      // trips and texel-trips
    for (int t0 = 0; t0 + 4 <= n; t0 += 4)
        for (int w = 0; w < 4; ++w)
        {
            int mx[2] = { 0, 0 }, npm[2] = { 0, 0 }; double sum = 0, sumT = 0; int mxSum = 0;
            for (int b = 0; b < 4; ++b) for (int e = 0; e < 16; ++e)
            {
                int s = 0;
                for (int k = 0; k < 2; ++k)
                {
                    const int g = k ? 7 - w : w; const int code = kFit2Order[g * 16 + e];
                    const int it = trips[size_t(t0 + b) * 128 + code], np = nps[size_t(t0 + b) * 128 + code];
                    mx[k] = std::max(mx[k], it); npm[k] = std::max(npm[k], np); sum += it; sumT += it * np; s += it;
                }
                mxSum = std::max(mxSum, s);
            }
            cur += mx[0] + mx[1]; curT += mx[0] * npm[0] + mx[1] * npm[1];
            ideal += sum / 64.0; idealT += sumT / 64.0;
            inter += mxSum; interT += mxSum * 0.5 * (npm[0] + npm[1]);
        }
    printf("wave-level Newton trips per wavefront: now %.2f, two fits interleaved per lane %.2f, perfectly packed %.2f; texel-trips: now %.1f, interleaved ~%.1f, packed %.1f\n",
           cur / (n), inter / n, ideal / n, curT / n, interT / n, idealT / n);
    return 0;
}
