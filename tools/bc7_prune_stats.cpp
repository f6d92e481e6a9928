// DEVELOPMENT TOOL (host only): how much of the BC7 search of modes 1 / 3 could pruning remove at best? For tiles of the benchmark
// image, every ranked candidate's lower bound (subset_lower_bound), unoptimised error and final error, against three tables:
//   now      - what bc7_pre_kernel sees: the smallest unoptimised error of the mode's candidates (and the results of finished modes)
//   in-mode  - the smallest FINAL error of the mode's candidates (a second pruning pass after searching the best-ranked few)
//   oracle   - a perfect bound: a candidate is searched only if its final error is the block's minimum
// build + run: tools/run_bc7_prune_stats.sh <tiles.bin> <ntiles>
#define DXTEX_HOST_DEBUG 1
#define DXTEX_COUNT_EVALS 1
#define DXTEX_BC7_USE_LOCKSTEP 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../directxtex_amd/csrc/bc67_tables.h"
#include "../directxtex_amd/csrc/bc7_core.h"
using namespace dxtex; using namespace dxtex::bc7;

struct HB { float f[64]; uint32_t ldr[16]; };
static void make_block(HB& b, const uint8_t* px)
{
    for (int i = 0; i < 16; ++i)
    {
        uint32_t l = 0;
        for (int c = 0; c < 4; ++c)
        {
            const float v = float(px[i * 4 + c]) * (1.0f / 255.0f);
            b.f[i * 4 + c] = v;
            float q = v * 255.0f + 0.01f; q = (q < 255.0f) ? q : 255.0f; q = (0.0f < q) ? q : 0.0f;
            l |= (uint32_t(q) & 0xFF) << (8 * c);
        }
        b.ldr[i] = l;
    }
}
static void seeds2(const HB& b, uint32_t shape, int region, Region& rg, uint32_t& A, uint32_t& B, uint32_t& anchor, uint32_t& mask)
{
    const uint32_t m1 = kPart2Mask[shape];
    mask = region ? m1 : ((~m1) & 0xFFFF);
    region_init(rg, b.ldr, mask);
    if (rg.np == 1) { A = b.ldr[rg.pos(0)]; B = A; }
    else if (rg.np == 2) { A = b.ldr[rg.pos(0)]; B = b.ldr[rg.pos(1)]; }
    else seed_endpoints<true>(b.f, mask, A, B);
    anchor = region ? kAnchor2[shape] : 0;
}
struct CandStat { int lb, org, fin, np[2], orgS[2], finS[2]; double cost, resid; };
static double line_resid(const uint32_t* pix, uint32_t mask);
template<int MODE> static CandStat cand(const HB& b, uint32_t shape)
{
    CandStat c; c.lb = 0; c.org = 0; c.cost = 0; c.resid = 0; int opt = 0;
    for (int r = 0; r < 2; ++r)
    {
        Region rg; uint32_t A, B, anchor, mask; seeds2(b, shape, r, rg, A, B, anchor, mask);
        const long e0 = g_evalCount[MODE], b0 = g_boundCount[MODE];
        SubsetResult res; refine_subset<MODE, 0>(rg, A, B, anchor, res);
        c.cost += (double(g_evalCount[MODE] - e0) + 0.45 * double(g_boundCount[MODE] - b0)) * rg.np;      // texel-evaluations, a bound ~0.45 of an exact one
        c.org += res.orgErr; opt += res.optErr; c.np[r] = rg.np; c.orgS[r] = res.orgErr; c.finS[r] = res.optErr;
        c.lb += subset_lower_bound(b.ldr, mask, 0u, 3); c.resid += line_resid(b.ldr, mask);
    }
    c.fin = std::min(c.org, opt);
    return c;
}
// residual of the best-fit line through the subset's texels (3 channels), exact eigen solve in double (power iteration)
static double line_resid(const uint32_t* pix, uint32_t mask)
{
    double n = 0, s[3] = {0,0,0}, q[3][3] = {};
    for (int i = 0; i < 16; ++i) if ((mask >> i) & 1u) { double v[3] = { double(pix[i] & 0xFF), double((pix[i] >> 8) & 0xFF), double((pix[i] >> 16) & 0xFF) }; n += 1; for (int a = 0; a < 3; ++a) { s[a] += v[a]; for (int b = 0; b < 3; ++b) q[a][b] += v[a] * v[b]; } }
    if (n < 2) return 0;
    double S[3][3]; double tr = 0; for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] = q[a][b] - s[a] * s[b] / n; for (int a = 0; a < 3; ++a) tr += S[a][a];
    double v[3] = { 1, 0.7, 0.3 }, lam = 0;
    for (int it = 0; it < 200; ++it) { double w[3]; for (int a = 0; a < 3; ++a) w[a] = S[a][0] * v[0] + S[a][1] * v[1] + S[a][2] * v[2]; const double nn = sqrt(w[0]*w[0] + w[1]*w[1] + w[2]*w[2]); if (nn < 1e-12) { lam = 0; break; } lam = nn; for (int a = 0; a < 3; ++a) v[a] = w[a] / nn; }
    return std::max(0.0, tr - lam);
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); const int n = atoi(argv[2]);
    std::vector<uint8_t> tiles(size_t(n) * 64); if (fread(tiles.data(), 1, tiles.size(), f) != tiles.size()) return 2; fclose(f);
    double noslack[2] = {0, 0}, fr[2] = {0, 0}, frN[2] = {0, 0}, tot[2] = {0, 0}, now[2] = {0, 0}, inmode[2] = {0, 0}, oracle[2] = {0, 0}, top4[2] = {0, 0}, flat[2] = {0, 0}; long winRankHist[2][16] = {};
    double ratioSum[2] = {0, 0}; long ratioN[2] = {0, 0};
    double lbHist[2][12] = {}; double costByBest[2][8] = {};
    for (int t = 0; t < n; ++t)
    {
        HB b; make_block(b, &tiles[size_t(t) * 64]);
        int e3[64], e2[64]; uint32_t l3[16], l2[16];
        for (uint32_t s = 0; s < 64; ++s)
        {
            e3[s] = e2[s] = 0;
            for (int r = 0; r < 2; ++r) { Region rg; uint32_t A, B, an, m; seeds2(b, s, r, rg, A, B, an, m); e3[s] += rough_error<3, 0>(rg, A, B); e2[s] += rough_error<2, 0>(rg, A, B); }
        }
        for (int pass = 0; pass < 2; ++pass)
        {
            int e[64]; uint32_t sh[64];
            for (int s = 0; s < 64; ++s) { e[s] = pass ? e2[s] : e3[s]; sh[s] = s; }
            for (int i = 0; i < 16; ++i) for (int j = i + 1; j < 64; ++j) if (e[i] > e[j]) { std::swap(e[i], e[j]); std::swap(sh[i], sh[j]); }
            for (int i = 0; i < 16; ++i) (pass ? l2 : l3)[i] = sh[i];
        }
        int prevBest = 0x7FFFFFFF;
        static const bool order31 = getenv("DXTEX_STATS_ORDER31") != nullptr;      // what if mode 3 ran before mode 1?
        for (int mm = 0; mm < 2; ++mm)
        {
            const int m = order31 ? 1 - mm : mm;
            CandStat c[16];
            for (int i = 0; i < 16; ++i) c[i] = m ? cand<3>(b, l2[i]) : cand<1>(b, l3[i]);
            int tabNow = prevBest, best = 0x7FFFFFFF, bi = 0;
            for (int i = 0; i < 16; ++i) { tabNow = std::min(tabNow, c[i].org); if (c[i].fin < best) { best = c[i].fin; bi = i; } }
            const int tabIn = std::min(prevBest, best);
            // the best four by unoptimised error (phase A of a two-phase mode), the rest pruned against their results
            int idx[16]; for (int i = 0; i < 16; ++i) idx[i] = i;
            std::sort(idx, idx + 16, [&](int x, int y) { return c[x].org < c[y].org; });
            int tabA = prevBest; for (int k = 0; k < 4; ++k) tabA = std::min(tabA, c[idx[k]].fin);
            for (int i = 0; i < 16; ++i)
            {
                if (c[i].org == 0) continue;
                const double w = c[i].cost;
                tot[m] += w;
                const bool sNow = !(c[i].lb > tabNow);
                if (sNow) now[m] += w;
                if (!(c[i].resid > tabNow)) noslack[m] += w;
                if (sNow && c[i].resid > 1) { fr[m] += w * double(c[i].fin) / c[i].resid; frN[m] += w; }
                if (!(c[i].lb > tabIn)) inmode[m] += w;
                if (c[i].fin <= std::min(prevBest, best)) oracle[m] += w;
                bool inTop = false; for (int k = 0; k < 4; ++k) inTop |= idx[k] == i;
                if (sNow && (inTop || !(c[i].lb > tabA))) top4[m] += w;
                if (sNow && c[i].lb > 0) { ratioSum[m] += double(c[i].fin) / c[i].lb; ++ratioN[m]; }
                if (sNow) { const double r = c[i].lb > 0 ? double(c[i].lb) / std::max(1, tabNow) : 0.0; lbHist[m][std::min(11, int(r * 10))] += w; }
                if (sNow) { int k = 0; for (int v = best; v > 0 && k < 7; v >>= 2) ++k; costByBest[m][k] += w; }
            }
            winRankHist[m][bi]++;
            prevBest = std::min(prevBest, best);
        }
    }
    for (int m = 0; m < 2; ++m)
    {
        printf("mode %d: %.0f cost units (texel-evaluations) unpruned; searched now %.1f %%; with the in-mode final table %.1f %%; two-phase (best 4 by org first) %.1f %%; oracle %.1f %%; mean final/LB of the searched %.2f\n",
               m ? 3 : 1, tot[m], 100.0 * now[m] / tot[m], 100.0 * inmode[m] / tot[m], 100.0 * top4[m] / tot[m], 100.0 * oracle[m] / tot[m], ratioSum[m] / std::max(1L, ratioN[m]));
        printf("  searched if the bound were the plain line residual (no rounding slack; NOT valid, potential only): %.1f %%; cost-weighted final / residual of the searched: %.2f\n", 100.0 * noslack[m] / tot[m], fr[m] / std::max(1.0, frN[m]));
        printf("  cost share by LB / table of the searched (0.0-0.1 ... 1.0+):"); for (int k = 0; k < 12; ++k) printf(" %.1f%%", 100.0 * lbHist[m][k] / now[m]); printf("\n");
        printf("  cost share by the mode's best final error (0, 1-3, 4-15, 16-63, 64-255, 256-1023, 1024-4095, 4096+):"); for (int k = 0; k < 8; ++k) printf(" %.1f%%", 100.0 * costByBest[m][k] / now[m]); printf("\n");
        printf("  rank of the winning candidate:"); for (int k = 0; k < 16; ++k) printf(" %ld", winRankHist[m][k]); printf("\n");
    }
    return 0;
}
