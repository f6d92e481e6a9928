// CPU property test of the exact-pruning bounds of directxtex_amd/csrc/bc7_core.h (compiled for the host through
// DXTEX_HOST_DEBUG): for random texels, subsets, rotations and RANDOM endpoints, the error of the palette those endpoints
// generate - with the best possible index per texel, which no index rule can beat - must never be below
// subset_lower_bound / scalar_kmeans_lower_bound. Also checks the bound against a crude search for good endpoints.
// usage: bc7_bound_check [trials]
#define DXTEX_HOST_DEBUG 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <algorithm>
#include <cstring>
#include "../../directxtex_amd/csrc/bc67_tables.h"
#include "../../directxtex_amd/csrc/bc7_core.h"
#include "../../tools/bc7_bound2.h"

using namespace dxtex;
using namespace dxtex::bc7;

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { g_s ^= g_s << 13; g_s ^= g_s >> 7; g_s ^= g_s << 17; return uint32_t(g_s >> 32); }

static const int kW2[4] = { 0, 21, 43, 64 }, kW3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 }, kW4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

// error of a subset against the palette of (ea, eb) with `bits`-bit indices over the channels [0, C), best index per texel
static long palette_error(const uint32_t* px, uint32_t mask, int C, int bits, const int* ea, const int* eb)
{
    const int n = 1 << bits; const int* w = bits == 2 ? kW2 : bits == 3 ? kW3 : kW4;
    long tot = 0;
    for (int i = 0; i < 16; ++i)
    {
        if (!((mask >> i) & 1u)) continue;
        long best = -1;
        for (int j = 0; j < n; ++j)
        {
            long e = 0;
            for (int c = 0; c < C; ++c)
            {
                const int q = (ea[c] * (64 - w[j]) + eb[c] * w[j] + 32) >> 6;
                const int d = int((px[i] >> (8 * c)) & 0xFF) - q;
                e += long(d) * d;
            }
            if (best < 0 || e < best) best = e;
        }
        tot += best;
    }
    return tot;
}

int main(int argc, char** argv)
{
    const int trials = argc > 1 ? std::atoi(argv[1]) : 20000;
    long violations = 0, checked = 0, lineChecked = 0, lineOrder = 0, lineGain = 0; double tightest = 1e30;
    for (int t = 0; t < trials; ++t)
    {
        // texels: a mix of noisy, near-linear and few-colour blocks
        uint32_t px[16];
        const int kind = rnd() % 4;
        int a[4], b[4];
        for (int c = 0; c < 4; ++c) { a[c] = rnd() % 256; b[c] = rnd() % 256; }
        for (int i = 0; i < 16; ++i)
        {
            uint32_t p = 0;
            for (int c = 0; c < 4; ++c)
            {
                int v;
                if (kind == 0) v = rnd() % 256;
                else if (kind == 1) { const int s = rnd() % 65; v = (a[c] * (64 - s) + b[c] * s) / 64 + int(rnd() % 9) - 4; }
                else if (kind == 2) v = (rnd() & 1) ? a[c] : b[c];
                else v = a[c] + int(rnd() % 40) - 20;
                v = std::min(255, std::max(0, v));
                p |= uint32_t(v) << (8 * c);
            }
            px[i] = p;
        }
        // a third of the blocks get one or two CONSTANT channels (alpha of an opaque block, grey ramps): the bound charges rounding slack
        // only to the channels that vary, and palettes that move off the constant must not get below it
        if (rnd() % 3 == 0)
        {
            const int c0 = rnd() % 4, c1 = (rnd() & 1) ? int(rnd() % 4) : c0;
            const uint32_t v0 = (rnd() & 1) ? 255u : rnd() % 256, v1 = rnd() % 256;
            for (int i = 0; i < 16; ++i)
            {
                px[i] = (px[i] & ~(0xFFu << (8 * c0))) | (v0 << (8 * c0));
                if (c1 != c0) px[i] = (px[i] & ~(0xFFu << (8 * c1))) | (v1 << (8 * c1));
            }
        }
        const uint32_t mask = (rnd() % 3 == 0) ? 0xFFFFu : (kPart2Mask[rnd() % 64] ^ ((rnd() & 1) ? 0xFFFFu : 0u)) & 0xFFFFu;
        if (!mask) continue;
        const int C = (rnd() & 1) ? 3 : 4;
        const int bits = 2 + rnd() % 3;
        const int lbPlain = subset_lower_bound(px, mask, 0, C);
        // the along-the-line bound of bc7_bound2.h (2-bit indices: free and fixed-weight term; 3-bit: free term) must hold for every palette too,
        // and is never below the plain one
        int lb = lbPlain;
        if (bits == 2) { const int f = subset_lower_bound_line<4, false>(px, mask, 0, C), x = subset_lower_bound_line<4, true>(px, mask, 0, C); lineChecked += 2; if (f < lbPlain - 1 || x < f - 1) ++lineOrder; lineGain += (x > lbPlain); lb = std::max(lb, std::max(f, x)); }
        if (bits == 3) { const int f = subset_lower_bound_line<8, false>(px, mask, 0, C); ++lineChecked; if (f < lbPlain - 1) ++lineOrder; lineGain += (f > lbPlain); lb = std::max(lb, f); }
        // random endpoints, endpoints drawn from the texels, and a little hill climbing from the best of those
        long bestErr = -1; int be[4] = { 0, 0, 0, 0 }, bf[4] = { 0, 0, 0, 0 };
        for (int k = 0; k < 60; ++k)
        {
            int ea[4], eb[4];
            if (k & 1) { for (int c = 0; c < 4; ++c) { ea[c] = rnd() % 256; eb[c] = rnd() % 256; } }
            else
            {
                int i0, i1;
                do { i0 = rnd() % 16; } while (!((mask >> i0) & 1u));
                do { i1 = rnd() % 16; } while (!((mask >> i1) & 1u));
                for (int c = 0; c < 4; ++c) { ea[c] = (px[i0] >> (8 * c)) & 0xFF; eb[c] = (px[i1] >> (8 * c)) & 0xFF; }
            }
            const long e = palette_error(px, mask, C, bits, ea, eb);
            ++checked;
            if (e < lb) { ++violations; std::printf("VIOLATION: error %ld < bound %d (trial %d)\n", e, lb, t); }
            if (bestErr < 0 || e < bestErr) { bestErr = e; for (int c = 0; c < 4; ++c) { be[c] = ea[c]; bf[c] = eb[c]; } }
        }
        for (int it = 0; it < 200; ++it)
        {
            int ea[4], eb[4];
            for (int c = 0; c < 4; ++c) { ea[c] = std::min(255, std::max(0, be[c] + int(rnd() % 7) - 3)); eb[c] = std::min(255, std::max(0, bf[c] + int(rnd() % 7) - 3)); }
            const long e = palette_error(px, mask, C, bits, ea, eb);
            ++checked;
            if (e < lb) { ++violations; std::printf("VIOLATION: error %ld < bound %d (trial %d, climbing)\n", e, lb, t); }
            if (e < bestErr) { bestErr = e; for (int c = 0; c < 4; ++c) { be[c] = ea[c]; bf[c] = eb[c]; } }
        }
        if (lb > 0) tightest = std::min(tightest, double(bestErr) / double(lb));
        // the scalar slot of modes 4 / 5: any K values against the optimum of 1-D K-means
        const uint32_t rot = rnd() % 4;
        const int lb4 = scalar_kmeans_lower_bound<4>(px, rot), lb8 = scalar_kmeans_lower_bound<8>(px, rot);
        for (int k = 0; k < 40; ++k)
        {
            const int K = (k & 1) ? 4 : 8;
            int lv[8], e0 = rnd() % 256, e1 = rnd() % 256;
            for (int j = 0; j < K; ++j) lv[j] = (K == 4) ? (e0 * (64 - kW2[j]) + e1 * kW2[j] + 32) >> 6 : (e0 * (64 - kW3[j]) + e1 * kW3[j] + 32) >> 6;
            long e = 0;
            for (int i = 0; i < 16; ++i)
            {
                const int v = int(rotate_pixel(px[i], rot) >> 24);
                long best = 1 << 30;
                for (int j = 0; j < K; ++j) best = std::min<long>(best, long(v - lv[j]) * (v - lv[j]));
                e += best;
            }
            ++checked;
            if (e < ((K == 4) ? lb4 : lb8)) { ++violations; std::printf("VIOLATION: scalar error %ld < k-means bound %d (K=%d)\n", e, (K == 4) ? lb4 : lb8, K); }
        }
    }
    // The opaque-block variant of the RGBA fit (fit_setup / fit_iterate with A1): bit-identical end points to the plain RGBA fit on
    // blocks whose alpha is exactly 1.0f, for every subset mask - flat, two-colour, gradient and noisy blocks, texels as LoadScanline
    // makes them (byte * float(1 / 255)).
    long fits = 0, fitDiff = 0;
    for (int t = 0; t < trials; ++t)
    {
        float f[64];
        const int kind = rnd() % 5;
        int a[3], b[3];
        for (int c = 0; c < 3; ++c) { a[c] = rnd() % 256; b[c] = rnd() % 256; }
        for (int i = 0; i < 16; ++i)
        {
            for (int c = 0; c < 3; ++c)
            {
                int v;
                if (kind == 0) v = rnd() % 256;
                else if (kind == 1) { const int sgrad = rnd() % 65; v = (a[c] * (64 - sgrad) + b[c] * sgrad) / 64 + int(rnd() % 9) - 4; }
                else if (kind == 2) v = (rnd() & 1) ? a[c] : b[c];
                else if (kind == 3) v = a[c] + int(rnd() % 6) - 3;
                else v = a[c];
                v = std::min(255, std::max(0, v));
                f[i * 4 + c] = float(v) * (1.0f / 255.0f);
            }
            f[i * 4 + 3] = 255.0f * (1.0f / 255.0f);
            if (f[i * 4 + 3] != 1.0f) { std::printf("alpha 255 does not load as 1.0f\n"); return 1; }
        }
        const uint32_t masks[3] = { 0xFFFFu, uint32_t(kPart2Mask[rnd() % 64]), uint32_t(~kPart2Mask[rnd() % 64]) & 0xFFFFu };
        for (uint32_t mask : masks)
        {
            if (__builtin_popcount(mask) < 3) continue;
            float X0[4], Y0[4], X1[4], Y1[4];
            seed_fit<true, false, false>(f, mask, X0, Y0);
            seed_fit<true, false, true>(f, mask, X1, Y1);
            ++fits;
            if (std::memcmp(X0, X1, sizeof(X0)) != 0 || std::memcmp(Y0, Y1, sizeof(Y0)) != 0)
            {
                ++fitDiff;
                if (fitDiff < 4) std::printf("A1 FIT DIFFERS (trial %d, mask %04x): X %g %g %g %g | %g %g %g %g\n", t, mask, X0[0], X0[1], X0[2], X0[3], X1[0], X1[1], X1[2], X1[3]);
            }
        }
    }
    std::printf("%ld palettes checked, %ld violations, best found error / bound >= %.3f\n", checked, violations, tightest);
    std::printf("along-the-line bound (bc7_bound2.h): %ld evaluated, above the plain bound in %ld, below the plain bound or out of order in %ld\n", lineChecked, lineGain, lineOrder);
    if (lineOrder) violations += lineOrder;
    std::printf("%ld opaque fits, %ld differ between the RGBA fit and its opaque-block variant\n", fits, fitDiff);
    return (violations || fitDiff) ? 1 : 0;
}
