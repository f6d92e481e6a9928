// TEST INFRASTRUCTURE / DEVELOPMENT TOOL (not part of the product; built by oracle/Makefile into oracle/_ref/bc6h_core_check and
// run by tests/test_bc7_core_cpu.py): runs the BC6H core of
// directxtex_amd/csrc/bc6h_core.h on the HOST, in the order the kernels of bc6h_encode.hip use it, next to the
// reference's D3DX_BC6H::Encode compiled in place, and compares the emitted blocks.
// Build + run:  tools/run_bc6h_debug.sh [ntiles] [seed]
#define DXTEX_HOST_DEBUG 1
#define private public
#define protected public
#include "BC6HBC7.cpp"     // the reference, in place (-I/root/reference/DirectXTex), against oracle/shim
#undef private
#undef protected

#include <cstdio>
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <vector>

static inline uint32_t dxtex_host_float_to_half(float v) { return DirectX::PackedVector::XMConvertFloatToHalf(v); }
#include "../directxtex_amd/csrc/bc7_core.h"
#include "../directxtex_amd/csrc/bc6h_core.h"

using namespace dxtex;
using namespace dxtex::bc6h;

#if defined(DXTEX_COUNT_EVALS6)
// statistics of the PerturbOne bound (bc6h_core.h: perturb6_bound): [entries 8 / 16][step class: first, large, 4, 2, 1]
static unsigned long long g_cand[2][5], g_pass[2][5], g_texels[2][5], g_passTexels[2][5], g_unsound;
static double g_worst = 0.0;
namespace dxtex { namespace bc6h {
void count_bound6(int n, int np, int step, float bound, float exact, float best)
{
    const int a = n == 16, c = step == 0 ? 0 : step > 4 ? 1 : step == 4 ? 2 : step == 2 ? 3 : 4;
    ++g_cand[a][c]; g_texels[a][c] += np;
    if (bound < best) { ++g_pass[a][c]; g_passTexels[a][c] += np; }
    if (bound > exact) { ++g_unsound; if (g_unsound <= 5) printf("UNSOUND bound %.9g > exact %.9g (np %d step %d)\n", bound, exact, np, step); }
    if (exact > 0 && bound / exact > g_worst && bound <= exact) g_worst = bound / exact;
}
} }
// experiment: after how many texels does the prefix of the bound exclude a candidate - in the region's order and "outside in" along the
// channel of largest range. [order][step class][checkpoint 4 / 8 / 12 / never before the end]
static unsigned long long g_pref[2][5][4];
namespace dxtex { namespace bc6h {
void count_prefix6(const float* r, const float* g, const float* b, int stride, int np, const float* pr, const float* pg, const float* pb, float best, int step)
{
    const int c = step == 0 ? 0 : step > 4 ? 1 : step == 4 ? 2 : step == 2 ? 3 : 4;
    double D[16];
    for (int k = 0; k < np; ++k)
    {
        double m = 1e300;
        for (int i = 0; i < 8; ++i) { const double dr = r[k * stride] - pr[i], dg = g[k * stride] - pg[i], db = b[k * stride] - pb[i]; m = std::min(m, dr * dr + dg * dg + db * db); }
        D[k] = m;
    }
    int order[2][16];
    for (int k = 0; k < np; ++k) order[0][k] = k;
    // outside in along the channel of largest range
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    for (int k = 0; k < np; ++k) { const float v[3] = { r[k * stride], g[k * stride], b[k * stride] }; for (int ch = 0; ch < 3; ++ch) { lo[ch] = std::min(lo[ch], v[ch]); hi[ch] = std::max(hi[ch], v[ch]); } }
    int cm = 0; for (int ch = 1; ch < 3; ++ch) if (hi[ch] - lo[ch] > hi[cm] - lo[cm]) cm = ch;
    const float* pl = cm == 0 ? r : cm == 1 ? g : b;
    for (int k = 0; k < np; ++k)
    {
        int rank = 0;
        for (int j = 0; j < np; ++j) rank += (pl[j * stride] < pl[k * stride]) || (pl[j * stride] == pl[k * stride] && j < k);
        const int pos = (rank < (np + 1) / 2) ? 2 * rank : 2 * (np - 1 - rank) + 1;
        order[1][pos] = k;
    }
    for (int o = 0; o < 2; ++o)
    {
        double sum = 0; int at = 3;
        for (int k = 0; k < np; ++k)
        {
            sum += D[order[o][k]];
            if ((k & 3) == 3 && k < 12 && sum * 0.9999 >= best) { at = k >> 2; break; }
        }
        ++g_pref[o][c][at];
    }
}
} }
static void print_bound_stats()
{
    const char* ord[2] = { "region order", "outside in  " }; const char* cl[5] = { "first", ">4", "4", "2", "1" };
    for (int o = 0; o < 2; ++o)
        for (int c = 0; c < 5; ++c)
        {
            const double t = double(g_pref[o][c][0] + g_pref[o][c][1] + g_pref[o][c][2] + g_pref[o][c][3]);
            if (t > 0) printf("%s step %-5s: excluded after 4 texels %6.2f %%, 8: %6.2f %%, 12: %6.2f %%, later or never %6.2f %%\n", ord[o], cl[c],
                              100 * g_pref[o][c][0] / t, 100 * g_pref[o][c][1] / t, 100 * g_pref[o][c][2] / t, 100 * g_pref[o][c][3] / t);
        }
    const char* cls[5] = { "first", ">4", "4", "2", "1" };
    for (int a = 0; a < 2; ++a)
    {
        unsigned long long tc = 0, tp = 0, tt = 0, tpt = 0;
        for (int c = 0; c < 5; ++c)
        {
            if (!g_cand[a][c]) continue;
            printf("entries %2d step %-5s: %10llu candidates, %5.1f %% pass the bound (texel-weighted %5.1f %%)\n", a ? 16 : 8, cls[c], g_cand[a][c],
                   100.0 * g_pass[a][c] / g_cand[a][c], 100.0 * g_passTexels[a][c] / g_texels[a][c]);
            tc += g_cand[a][c]; tp += g_pass[a][c]; tt += g_texels[a][c]; tpt += g_passTexels[a][c];
        }
        if (tc) printf("entries %2d all steps : %10llu candidates, %5.1f %% pass (texel-weighted %5.1f %%), mean texels %.1f\n", a ? 16 : 8, tc, 100.0 * tp / tc, 100.0 * tpt / tt, double(tt) / tc);
    }
    printf("bound above the exact error: %llu candidates; tightest bound / exact = %.6f\n", g_unsound, g_worst);
}
#endif

static uint32_t g_rng = 1;
static uint32_t rnd() { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }
static float frand() { return float(rnd() & 0xFFFF) / 65535.0f; }

static ModeRt mode_rt(int i)
{
    ModeRt m; const Bc6hMode& k = kBc6hModes[i];
    m.index = i; m.code = k.code; m.regions2 = k.regions2; m.transformed = k.transformed; m.prec = k.prec[0];
    for (int c = 0; c < 3; ++c) m.delta[c] = k.delta[c];
    return m;
}

struct Region6 { float r[16], g[16], b[16]; uint64_t pos; int np; };
static void gather(const int ipx[16][3], uint32_t mask, Region6& rg)
{
    rg.np = 0; rg.pos = 0;
    for (int i = 0; i < 16; ++i)
        if ((mask >> i) & 1) { rg.r[rg.np] = float(ipx[i][0]); rg.g[rg.np] = float(ipx[i][1]); rg.b[rg.np] = float(ipx[i][2]); rg.pos |= uint64_t(i) << (4 * rg.np); ++rg.np; }
}
static Texels tex(const Region6& rg) { Texels t; t.r = rg.r; t.g = rg.g; t.b = rg.b; t.stride = 1; t.np = rg.np; return t; }

static EndPts seed_region(const float* fpx, const int ipx[16][3], uint32_t mask, bool isSigned, int& np)
{
    EndPts s; np = __builtin_popcount(mask);
    int first = -1, second = -1;
    for (int i = 0; i < 16; ++i) if ((mask >> i) & 1) { if (first < 0) first = i; else if (second < 0) second = i; }
    if (np == 1) { for (int c = 0; c < 3; ++c) { s.A[c] = ipx[first][c]; s.B[c] = ipx[first][c]; } return s; }
    if (np == 2) { for (int c = 0; c < 3; ++c) { s.A[c] = ipx[first][c]; s.B[c] = ipx[second][c]; } return s; }
    float X[4], Y[4];
    bc7::seed_fit<false>(fpx, mask, X, Y);
    for (int c = 0; c < 3; ++c) { s.A[c] = clamp_seed(float_to_int16f(X[c], isSigned), isSigned); s.B[c] = clamp_seed(float_to_int16f(Y[c], isSigned), isSigned); }
    return s;
}

static int g_onlyMode = -1; static bool g_noSearch = false;
template<int N>
static bool refine_host(const ModeRt& m, bool isSigned, uint32_t shape, const EndPts seeds[2], const int ipx[16][3], float& bestErr, uint64_t& lo, uint64_t& hi)
{
    const int nreg = m.regions2 ? 2 : 1;
    const uint32_t m1 = m.regions2 ? kPart2Mask[shape] : 0u;
    const uint32_t masks[2] = { m.regions2 ? ((~m1) & 0xFFFFu) : 0xFFFFu, m1 };
    const uint32_t anchors[2] = { 0u, m.regions2 ? uint32_t(kAnchor2[shape]) : 0u };
    EndPts org[2], opt[2]; float orgErr[2] = { 0, 0 }, optErr[2] = { 0, 0 }; uint64_t orgIdx[2] = { 0, 0 }, optIdx[2] = { 0, 0 };
    Region6 rg[2], all; gather(ipx, 0xFFFF, all);
    for (int r = 0; r < nreg; ++r)
    {
        gather(ipx, masks[r], rg[r]);
        for (int c = 0; c < 3; ++c) { org[r].A[c] = quantize(seeds[r].A[c], m.prec, isSigned); org[r].B[c] = quantize(seeds[r].B[c], m.prec, isSigned); }
        orgErr[r] = assign_indices6<N>(tex(rg[r]), rg[r].pos, org[r], m.prec, isSigned, anchors[r], orgIdx[r]);
    }
    int a0[3] = { org[0].A[0], org[0].A[1], org[0].A[2] };
    bool fit = true;
    EndPts orgT[2];
    for (int r = 0; r < nreg; ++r) { orgT[r] = m.transformed ? transform_forward(org[r], r, a0) : org[r]; fit = fit && endpoints_fit(orgT[r], r, m, isSigned); }
    if (getenv("DXTEX_BC6H_DUMP")) { printf("  mode %d shape %u fit %d:", m.index, shape, int(fit)); for (int r = 0; r < nreg; ++r) { printf(" | A"); for (int c = 0; c < 3; ++c) printf(" %d", orgT[r].A[c]); printf(" B"); for (int c = 0; c < 3; ++c) printf(" %d", orgT[r].B[c]); } printf(" err %.9g %.9g\n", orgErr[0], orgErr[1]); }
    if (!fit) return false;
    for (int r = 0; r < nreg; ++r)
    {
        const Region6& sr = (r == 0) ? all : rg[r];          // the reference's region-0 quirk
        if (g_noSearch) opt[r] = org[r]; else optimize_one6<N>(tex(sr), org[r], orgErr[r], m.prec, isSigned, opt[r]);
        optErr[r] = assign_indices6<N>(tex(rg[r]), rg[r].pos, opt[r], m.prec, isSigned, anchors[r], optIdx[r]);
    }
    float orgTot = 0.0f, optTot = 0.0f;
    for (int r = 0; r < nreg; ++r) { orgTot += orgErr[r]; optTot += optErr[r]; }
    int b0[3] = { opt[0].A[0], opt[0].A[1], opt[0].A[2] };
    bool fitOpt = true; EndPts optT[2];
    for (int r = 0; r < nreg; ++r) { optT[r] = m.transformed ? transform_forward(opt[r], r, b0) : opt[r]; fitOpt = fitOpt && endpoints_fit(optT[r], r, m, isSigned); }
    const bool useOpt = fitOpt && optTot < orgTot;
    const float err = useOpt ? optTot : orgTot;
    if (!(err < bestErr)) return true;
    bestErr = err;
    int ep[4][3] = {};
    for (int r = 0; r < nreg; ++r)
        for (int c = 0; c < 3; ++c) { ep[2 * r][c] = (useOpt ? optT[r] : orgT[r]).A[c]; ep[2 * r + 1][c] = (useOpt ? optT[r] : orgT[r]).B[c]; }
    const uint64_t idx = useOpt ? (optIdx[0] | optIdx[1]) : (orgIdx[0] | orgIdx[1]);
    emit_block6(m, shape, ep, idx, anchors[1], lo, hi);
    return true;
}

static void encode_host(const float* fpx, bool isSigned, uint64_t& lo, uint64_t& hi)
{
    int ipx[16][3];
    for (int i = 0; i < 16; ++i) for (int c = 0; c < 3; ++c) ipx[i][c] = float_to_int16f(fpx[i * 4 + c], isSigned);
    EndPts seeds2[32][2], seed1[2];
    float rough[32]; uint32_t shp[32];
    Region6 rg;
    for (uint32_t s = 0; s < 32; ++s)
    {
        const uint32_t m1 = kPart2Mask[s];
        const uint32_t masks[2] = { (~m1) & 0xFFFFu, m1 };
        rough[s] = 0.0f; shp[s] = s;
        for (int r = 0; r < 2; ++r)
        {
            int np; seeds2[s][r] = seed_region(fpx, ipx, masks[r], isSigned, np);
            if (np > 2) { gather(ipx, masks[r], rg); rough[s] += rough_error6<8>(tex(rg), seeds2[s][r]); }
        }
    }
    for (int i = 0; i < 8; ++i) for (int j = i + 1; j < 32; ++j) if (rough[i] > rough[j]) { std::swap(rough[i], rough[j]); std::swap(shp[i], shp[j]); }
    if (getenv("DXTEX_BC6H_DUMP")) { printf("shapes:"); for (int i = 0; i < 8; ++i) printf(" %u", shp[i]); printf("\n  seeds rank0:"); for (int r = 0; r < 2; ++r) { for (int c = 0; c < 3; ++c) printf(" %d", seeds2[shp[0]][r].A[c]); for (int c = 0; c < 3; ++c) printf(" %d", seeds2[shp[0]][r].B[c]); } printf("\n  rough:"); for (int i = 0; i < 10; ++i) printf(" %.9g", rough[i]); printf("\n"); }
    { int np; seed1[0] = seed_region(fpx, ipx, 0xFFFF, isSigned, np); }
    float best = FLT_MAX; lo = hi = 0;
    for (int mi = 0; mi < 14 && best > 0; ++mi)
    {
        if (g_onlyMode >= 0 && mi != g_onlyMode) continue;
        const ModeRt m = mode_rt(mi);
        if (m.regions2) { for (int i = 0; i < 8 && best > 0; ++i) refine_host<8>(m, isSigned, shp[i], seeds2[shp[i]], ipx, best, lo, hi); }
        else refine_host<16>(m, isSigned, 0, seed1, ipx, best, lo, hi);
    }
}

#if defined(DXTEX_PRUNE_STATS6)
// DEVELOPMENT STATISTICS (round 5): what would a tighter pruning bound be worth for BC6H? The kernels prune a candidate (mode, shape) when the sum of its
// regions' lower bounds exceeds an error already on the table (bc6h_pre_kernel). The bound in use is the residual of the best-fit line minus the rounding
// slack (region_lower_bound6); `line` adds the error ALONG the line - eight palette points against the texels' projections (free 8-means, exact dynamic
// programme), minimised over the line's direction as tools/bc7_bound2.h derives it. Search cost = texels scored x (2 prec - 1), region 0 against all
// sixteen texels (the reference's quirk); the second and third mode of a trio reuse the first one's search.
static double g_costAll, g_costCur, g_costNew, g_lbCur, g_lbNew, g_errSum; static unsigned long long g_nCand, g_newBelow, g_violCur, g_violNew;
static void scatter3(const Region6& rg, double& n, double (&mean)[3], double (&S)[3][3])
{
    n = rg.np; mean[0] = mean[1] = mean[2] = 0;
    for (int k = 0; k < rg.np; ++k) { mean[0] += rg.r[k]; mean[1] += rg.g[k]; mean[2] += rg.b[k]; }
    for (int c = 0; c < 3; ++c) mean[c] /= n;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] = 0;
    for (int k = 0; k < rg.np; ++k)
    {
        const double d[3] = { rg.r[k] - mean[0], rg.g[k] - mean[1], rg.b[k] - mean[2] };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] += d[i] * d[j];
    }
}
static double kmeans8(double* t, int n)
{
    if (n <= 8) return 0.0;
    std::sort(t, t + n);
    double s1[17] = { 0 }, s2[17] = { 0 };
    for (int i = 0; i < n; ++i) { s1[i + 1] = s1[i] + t[i]; s2[i + 1] = s2[i] + t[i] * t[i]; }
    auto sse = [&](int i, int j) { const double m = s1[j + 1] - s1[i]; const double c = (s2[j + 1] - s2[i]) - m * m / double(j - i + 1); return c > 0 ? c : 0.0; };
    double d[16], e[16];
    for (int j = 0; j < n; ++j) d[j] = sse(0, j);
    for (int k = 2; k <= 8; ++k)
    {
        for (int j = 0; j < n; ++j) { double best = (j < k) ? 0.0 : 1e300; for (int i = k - 1; i <= j; ++i) best = std::min(best, d[i - 1] + sse(i, j)); e[j] = best; }
        for (int j = 0; j < n; ++j) d[j] = e[j];
    }
    return d[n - 1];
}
// returns (cur, line) lower bounds of one region
static void region_bounds(const Region6& rg, double& cur, double& line)
{
    cur = line = 0.0;
    if (rg.np < 2) return;
    double n, mean[3], S[3][3]; scatter3(rg, n, mean, S);
    const double trS = S[0][0] + S[1][1] + S[2][2];
    if (!(trS > 0)) return;
    double A[3][3], B[3][3], Q[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = S[i][j] / trS; B[i][j] = A[i][j]; }
    for (int k = 0; k < 4; ++k)
    {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int l = 0; l < 3; ++l) v += B[i][l] * B[l][j]; Q[i][j] = v; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) B[i][j] = Q[i][j];
    }
    double f2 = 0; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) f2 += B[i][j] * B[i][j];
    const double lam = std::sqrt(std::sqrt(std::sqrt(std::sqrt(std::sqrt(f2))))) * (1.0 + 1e-9);
    const double rPlain = lam < 1.0 ? trS * (1.0 - lam) : 0.0;
    auto finish = [&](double G) { const double d = std::sqrt(G > 0 ? G : 0.0) - 3.0 * std::sqrt(n) - 1e-3; return d > 0 ? d * d * 0.9999 : 0.0; };
    cur = finish(rPlain);
    int jc = 0; for (int j = 1; j < 3; ++j) if (B[j][j] > B[jc][jc]) jc = j;
    double u0[3] = { B[0][jc], B[1][jc], B[2][jc] };
    const double nu = std::sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
    double G = rPlain;
    if (nu > 1e-150)
    {
        for (int i = 0; i < 3; ++i) u0[i] /= nu;
        double Su[3]; for (int i = 0; i < 3; ++i) Su[i] = S[i][0] * u0[0] + S[i][1] * u0[1] + S[i][2] * u0[2];
        const double a = u0[0] * Su[0] + u0[1] * Su[1] + u0[2] * Su[2];
        double g2 = 0; for (int i = 0; i < 3; ++i) { const double r = Su[i] - a * u0[i]; g2 += r * r; }
        const double g = std::sqrt(g2) * (1.0 + 1e-9) + 1e-9 * trS, mu = (trS - a) * (1.0 + 1e-9) + 1e-9 * trS;
        double t[16];
        for (int k = 0; k < rg.np; ++k) t[k] = (rg.r[k] - mean[0]) * u0[0] + (rg.g[k] - mean[1]) * u0[1] + (rg.b[k] - mean[2]) * u0[2];
        const double K0 = kmeans8(t, rg.np) * (1.0 - 1e-9);
        if (K0 > 0)
        {
            const double bq = a - K0, e = (-bq + std::sqrt(bq * bq + 4.0 * K0 * mu)) / (2.0 * K0);
            if (e > 0 && e < 1) { const double x1 = (trS - a) + (1.0 - e) * K0, x2 = trS - mu / e; const double Gn = std::min(x1, x2) * (1.0 - 1e-9) - g; G = std::max(G, Gn); }
        }
    }
    line = finish(G);
}
static void prune_stats_host(const float* fpx, bool isSigned)
{
    int ipx[16][3];
    for (int i = 0; i < 16; ++i) for (int c = 0; c < 3; ++c) ipx[i][c] = float_to_int16f(fpx[i * 4 + c], isSigned);
    EndPts seeds2[32][2]; float rough[32]; uint32_t shp[32]; Region6 rg;
    for (uint32_t s = 0; s < 32; ++s)
    {
        const uint32_t m1 = kPart2Mask[s]; const uint32_t masks[2] = { (~m1) & 0xFFFFu, m1 };
        rough[s] = 0.0f; shp[s] = s;
        for (int r = 0; r < 2; ++r) { int np; seeds2[s][r] = seed_region(fpx, ipx, masks[r], isSigned, np); if (np > 2) { gather(ipx, masks[r], rg); rough[s] += rough_error6<8>(tex(rg), seeds2[s][r]); } }
    }
    for (int i = 0; i < 8; ++i) for (int j = i + 1; j < 32; ++j) if (rough[i] > rough[j]) { std::swap(rough[i], rough[j]); std::swap(shp[i], shp[j]); }
    double lbCur[8], lbNew[8]; int np1[8];
    for (int i = 0; i < 8; ++i)
    {
        const uint32_t m1 = kPart2Mask[shp[i]]; const uint32_t masks[2] = { (~m1) & 0xFFFFu, m1 };
        lbCur[i] = lbNew[i] = 0;
        for (int r = 0; r < 2; ++r) { gather(ipx, masks[r], rg); double c, l; region_bounds(rg, c, l); lbCur[i] += c; lbNew[i] += l; if (r == 1) np1[i] = rg.np; if (l < c - 1e-6) ++g_newBelow; }
    }
    static const int order[10] = { 5, 6, 7, 8, 0, 2, 3, 4, 1, 9 };
    float best = FLT_MAX; int prevPrec = -1;
    bool searchedCur[8] = {}, searchedNew[8] = {};
    for (int oi = 0; oi < 10; ++oi)
    {
        const ModeRt m = mode_rt(order[oi]);
        const bool same = m.prec == prevPrec; prevPrec = m.prec;
        if (!same) for (int i = 0; i < 8; ++i) searchedCur[i] = searchedNew[i] = false;
        // pre: unoptimised errors of the candidates that fit -> the table
        float orgTot[8]; bool fit[8]; float table = best;
        for (int i = 0; i < 8; ++i)
        {
            const uint32_t shape = shp[i]; const uint32_t m1 = kPart2Mask[shape]; const uint32_t masks[2] = { (~m1) & 0xFFFFu, m1 };
            const uint32_t anchors[2] = { 0u, uint32_t(kAnchor2[shape]) };
            EndPts org[2]; float e[2]; uint64_t idx[2]; Region6 r2[2];
            for (int r = 0; r < 2; ++r)
            {
                gather(ipx, masks[r], r2[r]);
                for (int c = 0; c < 3; ++c) { org[r].A[c] = quantize(seeds2[shape][r].A[c], m.prec, isSigned); org[r].B[c] = quantize(seeds2[shape][r].B[c], m.prec, isSigned); }
                e[r] = assign_indices6<8>(tex(r2[r]), r2[r].pos, org[r], m.prec, isSigned, anchors[r], idx[r]);
            }
            int a0[3] = { org[0].A[0], org[0].A[1], org[0].A[2] };
            fit[i] = true;
            for (int r = 0; r < 2; ++r) { const EndPts t = m.transformed ? transform_forward(org[r], r, a0) : org[r]; fit[i] = fit[i] && endpoints_fit(t, r, m, isSigned); }
            orgTot[i] = e[0] + e[1];
            if (fit[i]) table = std::min(table, orgTot[i] * 1.00001f);
        }
        const double w = double(2 * m.prec - 1);
        for (int i = 0; i < 8; ++i)
        {
            if (!fit[i] || !(orgTot[i] > 0.0f)) continue;
            const double cost = w * double(16 + np1[i]);
            const bool liveCur = !(lbCur[i] > table), liveNew = !(lbNew[i] > table);
            g_costAll += same ? 0.0 : cost;
            if (liveCur && !searchedCur[i]) { g_costCur += cost; searchedCur[i] = true; }
            if (liveNew && !searchedNew[i]) { g_costNew += cost; searchedNew[i] = true; }
        }
        // the mode's results fold into the running best (whatever was pruned could not have lowered it)
        uint64_t lo, hi;
        for (int i = 0; i < 8 && best > 0; ++i)
        {
            float b2 = FLT_MAX; uint64_t l2, h2;
            if (refine_host<8>(m, isSigned, shp[i], seeds2[shp[i]], ipx, b2, l2, h2) && b2 < FLT_MAX)
            {
                ++g_nCand; g_lbCur += lbCur[i]; g_lbNew += lbNew[i]; g_errSum += b2;
                if (lbCur[i] > b2 * 1.0001) ++g_violCur;
                if (lbNew[i] > b2 * 1.0001) ++g_violNew;
                if (b2 < best) best = b2;
            }
        }
        (void)lo; (void)hi;
    }
}
#endif

int main(int argc, char** argv)
{
#if defined(DXTEX_PRUNE_STATS6)
    if (argc >= 4 && !strcmp(argv[1], "prune"))
    {
        FILE* f = fopen(argv[2], "rb"); const bool sg = atoi(argv[3]) != 0;
        alignas(16) float px[64]; int t = 0;
        while (fread(px, sizeof(px), 1, f) == 1) { prune_stats_host(px, sg); ++t; }
        printf("%d blocks: search cost of the two-region modes, all candidates that fit = 100 %%: bound in use %.1f %%, with the along-the-line term %.1f %%\n", t,
               100.0 * g_costCur / g_costAll, 100.0 * g_costNew / g_costAll);
        printf("  mean bound / final error over %llu refined candidates: %.3f in use, %.3f with the term; term below the plain bound in %llu regions; bounds above a final error: %llu / %llu\n",
               g_nCand, g_lbCur / g_errSum, g_lbNew / g_errSum, g_newBelow, g_violCur, g_violNew);
        return 0;
    }
#endif
    if (argc >= 4 && !strcmp(argv[1], "file"))
    {
        FILE* f = fopen(argv[2], "rb"); const bool sg = atoi(argv[3]) != 0;
        if (argc > 4) g_onlyMode = atoi(argv[4]);
        if (argc > 5) g_noSearch = atoi(argv[5]) != 0;
        alignas(16) float px[64]; int t = 0;
        while (fread(px, sizeof(px), 1, f) == 1) { uint64_t lo, hi; encode_host(px, sg, lo, hi); if (!getenv("DXTEX_BC6H_QUIET")) printf("%d %016llx %016llx\n", t, (unsigned long long)lo, (unsigned long long)hi); ++t; }
#if defined(DXTEX_COUNT_EVALS6)
        print_bound_stats();
#endif
        return 0;
    }
    // nbits (bc6h_core.h, count-leading-zeros form) against the reference's NBits loops, every value EndPointsFit can see
    for (int n = -140000; n <= 140000; ++n)
        for (int sg = 0; sg < 2; ++sg)
            if (n >= 0 || sg)
                if (nbits(n, sg != 0) != NBits(n, sg != 0)) { printf("nbits(%d, %d) = %d, the reference's NBits gives %d\n", n, sg, nbits(n, sg != 0), NBits(n, sg != 0)); return 3; }
    const int ntiles = argc > 1 ? atoi(argv[1]) : 100;
    g_rng = argc > 2 ? uint32_t(atoi(argv[2])) : 1u;
    int nbad = 0;
    for (int t = 0; t < ntiles; ++t)
    {
        alignas(16) float px[64];
        const int kind = rnd() % 6;
        const float scale = (float[]){ 0.001f, 0.1f, 1.0f, 8.0f, 200.0f, 30000.0f }[rnd() % 6];
        const float base[3] = { frand(), frand(), frand() };
        for (int i = 0; i < 16; ++i)
        {
            for (int c = 0; c < 3; ++c)
            {
                float v;
                if (kind == 0) v = base[c];
                else if (kind == 1) v = base[c] + 0.02f * frand();
                else if (kind == 2) v = base[c] * (0.5f + frand());
                else if (kind == 3) v = frand();
                else if (kind == 4) v = (i & 3) * 0.25f * base[c] + 0.1f * frand();
                else v = frand() - 0.3f;            // some negatives
                px[i * 4 + c] = v * scale;
            }
            px[i * 4 + 3] = 1.0f;
        }
        for (int sg = 0; sg < 2; ++sg)
        {
            alignas(16) uint8_t ref[16];
            reinterpret_cast<D3DX_BC6H*>(ref)->Encode(sg != 0, reinterpret_cast<const HDRColorA*>(px));
            uint64_t lo, hi; encode_host(px, sg != 0, lo, hi);
            uint64_t rlo, rhi; memcpy(&rlo, ref, 8); memcpy(&rhi, ref + 8, 8);
            if (rlo == lo && rhi == hi) continue;
            ++nbad;
            if (nbad <= 5) printf("tile %d signed %d kind %d scale %g MISMATCH ref %016llx%016llx ours %016llx%016llx (mode bits ref %x ours %x)\n", t, sg, kind, scale,
                                  (unsigned long long)rhi, (unsigned long long)rlo, (unsigned long long)hi, (unsigned long long)lo, unsigned(rlo & 31), unsigned(lo & 31));
        }
    }
#if defined(DXTEX_COUNT_EVALS6)
    print_bound_stats();
#endif
    printf("%d of %d encodes differ\n", nbad, ntiles * 2);
#if defined(DXTEX_COUNT_EVALS6)
    if (g_unsound) return 2;
#endif
    return nbad ? 1 : 0;
}
