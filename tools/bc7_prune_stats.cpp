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
#include <array>
#include <cmath>
#include "../directxtex_amd/csrc/bc67_tables.h"
#include "../directxtex_amd/csrc/bc7_core.h"
#include "bc7_bound2.h"
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
// EXPERIMENT (round 4, host only): a lower bound that also counts the error ALONG the palette's line. For a palette on a line with direction u
// the texels' squared distances to the real line points split into the distance to the line and the distance, along the line, to the nearest
// of the N points (Pythagoras): sum >= R(u) + K_N(u), R(u) = tr S - u'Su, K_N(u) = optimum of 1-D N-means of the projections. With u = c u1 +
// s v (u1 the principal axis): R(u) >= tr S - c^2 l1 - s^2 l2, sqrt K_N(u) >= c sqrt K_N(u1) - s sqrt l2 (the root of a k-means cost is
// 1-Lipschitz and homogeneous), and (x - y)+^2 >= (1 - e) x^2 - (1/e - 1) y^2 makes the sum linear in c^2: >= min(R1 + (1 - e) K0, tr S - l2 / e)
// for every e in (0, 1); the best e solves K0 e^2 + (l1 - K0) e - l2 = 0. The rounding of the entries costs 1/2 sqrt(C n) under the root as before.
static void jacobi3(double A[3][3], double w[3], double V[3][3])
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 60; ++sweep)
    {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q)
        {
            if (fabs(A[p][q]) < 1e-300) continue;
            const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    for (int i = 0; i < 3; ++i) w[i] = A[i][i];
}
static double kmeans1d(std::vector<double> t, int N)
{
    std::sort(t.begin(), t.end()); const int n = int(t.size());
    if (n <= N) return 0.0;
    std::vector<double> s1(n + 1, 0.0), s2(n + 1, 0.0);
    for (int i = 0; i < n; ++i) { s1[i + 1] = s1[i] + t[i]; s2[i + 1] = s2[i] + t[i] * t[i]; }
    auto sse = [&](int i, int j) { const double m = s1[j + 1] - s1[i]; const double c = (s2[j + 1] - s2[i]) - m * m / double(j - i + 1); return c > 0 ? c : 0.0; };
    std::vector<double> d(n), e(n);
    for (int j = 0; j < n; ++j) d[j] = sse(0, j);
    for (int k = 2; k <= N; ++k)
    {
        for (int j = 0; j < n; ++j)
        {
            double best = (j < k) ? 0.0 : 1e300;
            if (j >= k) for (int i = k - 1; i <= j; ++i) best = std::min(best, d[i - 1] + sse(i, j));
            e[j] = best;
        }
        d = e;
    }
    return d[n - 1];
}
// exact optimum of "N equally weighted points alpha + beta w_i / 64 against the sorted values t": every monotone assignment, least squares for (alpha, beta)
static double kconstr1d(std::vector<double> t, int N)
{
    static const double w3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 }, w2[4] = { 0, 21, 43, 64 };
    const double* w = (N == 8) ? w3 : w2;
    std::sort(t.begin(), t.end()); const int n = int(t.size());
    double st = 0, stt = 0; for (double v : t) { st += v; stt += v * v; }
    double best = 1e300;
    // depth-first over non-decreasing index sequences
    struct F { int k, i; double sx, sxx, sxt; };
    std::vector<F> stack; stack.push_back({ 0, 0, 0, 0, 0 });
    while (!stack.empty())
    {
        F f = stack.back(); stack.pop_back();
        if (f.k == n)
        {
            const double vx = f.sxx - f.sx * f.sx / n, cxt = f.sxt - f.sx * st / n, vt = stt - st * st / n;
            const double c = (vx > 1e-12) ? vt - cxt * cxt / vx : vt;
            best = std::min(best, c > 0 ? c : 0.0);
            continue;
        }
        for (int i = f.i; i < N; ++i) { const double x = w[i] / 64.0; stack.push_back({ f.k + 1, i, f.sx + x, f.sxx + x * x, f.sxt + x * t[f.k] }); }
    }
    return best;
}
static double g_lb2Diag[4];
static double g_diagSum[2][6];
static int subset_lower_bound2(const uint32_t* pix, uint32_t mask, int N)
{
    double n = 0, m[3] = {0,0,0}; std::vector<std::array<double,3>> P;
    for (int i = 0; i < 16; ++i) if ((mask >> i) & 1u) { std::array<double,3> v = { double(pix[i] & 0xFF), double((pix[i] >> 8) & 0xFF), double((pix[i] >> 16) & 0xFF) }; P.push_back(v); n += 1; for (int a = 0; a < 3; ++a) m[a] += v[a]; }
    if (n < 2) return 0;
    for (int a = 0; a < 3; ++a) m[a] /= n;
    double S[3][3] = {};
    for (auto& v : P) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] += (v[a] - m[a]) * (v[b] - m[b]);
    const double tr = S[0][0] + S[1][1] + S[2][2];
    if (!(tr > 0)) return 0;
    double A[3][3]; memcpy(A, S, sizeof(A)); double w[3], V[3][3]; jacobi3(A, w, V);
    int i1 = 0; for (int i = 1; i < 3; ++i) if (w[i] > w[i1]) i1 = i;
    double l1 = w[i1], l2 = -1e300; for (int i = 0; i < 3; ++i) if (i != i1) l2 = std::max(l2, w[i]);
    l1 *= (1.0 + 1e-9); l2 = std::max(0.0, l2) * (1.0 + 1e-9) + 1e-9;
    std::vector<double> t; for (auto& v : P) t.push_back((v[0] - m[0]) * V[0][i1] + (v[1] - m[1]) * V[1][i1] + (v[2] - m[2]) * V[2][i1]);
    static const bool constrained = getenv("DXTEX_STATS_CONSTRAINED_K") != nullptr;
    const double K0 = (constrained ? kconstr1d(t, N) : kmeans1d(t, N)) * (1.0 - 1e-9);
    const double R1 = std::max(0.0, tr - l1);
    double G = R1;
    if (K0 > 0)
    {
        const double bq = l1 - K0, e = (-bq + sqrt(bq * bq + 4.0 * K0 * l2)) / (2.0 * K0);
        if (e > 0 && e < 1) { const double ee = std::min(1.0, e * (1.0 + 1e-9) + 1e-12); G = std::min(R1 + (1.0 - ee) * K0, tr - l2 / ee); G = std::max(G, R1); }
    }
    g_lb2Diag[0] = R1; g_lb2Diag[1] = K0; g_lb2Diag[2] = G;
    // slack-free: the texels of one entry share ONE rounding vector, so their error is at least their scatter about their own centroid, whatever
    // the entry is: error >= within-cluster scatter of SOME partition into N clusters >= sum over three orthogonal axes of the 1-D N-means optimum
    double W3 = 0;
    for (int ax = 0; ax < 3; ++ax) { std::vector<double> q; for (auto& v : P) q.push_back((v[0] - m[0]) * V[0][ax] + (v[1] - m[1]) * V[1][ax] + (v[2] - m[2]) * V[2][ax]); W3 += kmeans1d(q, N); }
    g_lb2Diag[3] = W3;
    // and without any rounding slack: whatever the N palette entries are, the error is at least the optimum of N-means in space, which is at
    // least the optimum of N-means of the projections on any one direction (K0)
    const double d = sqrt(G) - 0.5 * sqrt(3.0 * n) - 1e-3;
    double lb = (d > 0) ? d * d * 0.99999 - 1.0 : 0.0;
    static const bool noFree = getenv("DXTEX_STATS_NO_FREE_K") != nullptr;
    if (!noFree) lb = std::max(lb, K0 * 0.99999 - 1.0);
    static const bool useW3 = getenv("DXTEX_STATS_W3") != nullptr;
    if (useW3) lb = std::max(lb, W3 * 0.99999 - 1.0);
    return lb > 0 ? int(lb) : 0;
}
struct CandStat { int lb, lb2, org, fin, np[2], orgS[2], finS[2], lb2S[2]; double cost, resid; };
static double line_resid(const uint32_t* pix, uint32_t mask);
template<int MODE> static CandStat cand(const HB& b, uint32_t shape)
{
    CandStat c; c.lb = 0; c.lb2 = 0; c.org = 0; c.cost = 0; c.resid = 0; int opt = 0;
    for (int r = 0; r < 2; ++r)
    {
        Region rg; uint32_t A, B, anchor, mask; seeds2(b, shape, r, rg, A, B, anchor, mask);
        const long e0 = g_evalCount[MODE], b0 = g_boundCount[MODE];
        SubsetResult res; refine_subset<MODE, 0>(rg, A, B, anchor, res);
        c.cost += (double(g_evalCount[MODE] - e0) + 0.45 * double(g_boundCount[MODE] - b0)) * rg.np;      // texel-evaluations, a bound ~0.45 of an exact one
        c.org += res.orgErr; opt += res.optErr; c.np[r] = rg.np; c.orgS[r] = res.orgErr; c.finS[r] = res.optErr;
        c.lb += subset_lower_bound(b.ldr, mask, 0u, 3); c.resid += line_resid(b.ldr, mask);
        // DXTEX_STATS_DEVICE_FORM: the device-ready function of bc7_bound2.h (approximate axis from the squared matrix) instead of this file's
        // Jacobi version; DXTEX_STATS_CONSTRAINED_K selects the fixed-weight term in both (2-bit indices only in the device form)
        static const bool devForm = getenv("DXTEX_STATS_DEVICE_FORM") != nullptr, fixedK = getenv("DXTEX_STATS_CONSTRAINED_K") != nullptr;
        if (devForm) c.lb2S[r] = (MODE == 1) ? subset_lower_bound_line<8, false>(b.ldr, mask, 0u, 3) : (fixedK ? subset_lower_bound_line<4, true>(b.ldr, mask, 0u, 3) : subset_lower_bound_line<4, false>(b.ldr, mask, 0u, 3));
        else c.lb2S[r] = subset_lower_bound2(b.ldr, mask, (MODE == 1) ? 8 : 4);
        c.lb2 += c.lb2S[r];
        if (!devForm) { g_diagSum[MODE == 1 ? 0 : 1][0] += g_lb2Diag[0]; g_diagSum[MODE == 1 ? 0 : 1][1] += g_lb2Diag[1]; g_diagSum[MODE == 1 ? 0 : 1][2] += g_lb2Diag[2]; g_diagSum[MODE == 1 ? 0 : 1][3] += res.optErr < res.orgErr ? res.optErr : res.orgErr; g_diagSum[MODE == 1 ? 0 : 1][4] += c.lb2S[r]; g_diagSum[MODE == 1 ? 0 : 1][5] += g_lb2Diag[3]; }
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
    long lateAll[3] = {0, 0, 0}, latePruned[3][2] = {};
    double lbHist[2][12] = {}; double costByBest[2][8] = {}; double now2[2] = {0, 0}, lbSum[2] = {0,0}, lb2Sum[2] = {0,0}, finSum[2] = {0,0}; long viol[2] = {0, 0}, violC[2] = {0, 0};
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
                if (!(std::max(c[i].lb, c[i].lb2) > tabNow)) now2[m] += w;
                for (int r = 0; r < 2; ++r) if (c[i].lb2S[r] > c[i].finS[r]) { ++viol[m]; if (viol[m] <= 3) printf("VIOLATION mode %d: subset bound %d > final %d (np %d)\n", m ? 3 : 1, c[i].lb2S[r], c[i].finS[r], c[i].np[r]); }
                if (c[i].lb2 > c[i].fin) ++violC[m];
                lbSum[m] += c[i].lb; lb2Sum[m] += std::max(c[i].lb, c[i].lb2); finSum[m] += c[i].fin;
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
            if (getenv("DXTEX_STATS_DUMP") && t < atoi(getenv("DXTEX_STATS_DUMP")))
            {
                printf("block %d mode %d:", t, m ? 3 : 1);
                for (int i = 0; i < 16; ++i) printf(" [%d lb %d lb2 %d org %d fin %d]", i, c[i].lb, c[i].lb2, c[i].org, c[i].fin);
                printf("\n");
            }
            winRankHist[m][bi]++;
            prevBest = std::min(prevBest, best);
        }
        // the late modes' tasks (modes 4 / 5: the whole block per rotation): today's universal bound and the along-the-line one against what modes 1 / 3 left
        for (uint32_t rot = 0; rot < 4; ++rot)
        {
            const int colour = subset_lower_bound(b.ldr, 0xFFFFu, rot, 3);
            const int line4 = std::max(colour, subset_lower_bound_line<4, true>(b.ldr, 0xFFFFu, rot, 3)), line8 = std::max(colour, subset_lower_bound_line<8, false>(b.ldr, 0xFFFFu, rot, 3));
            const int sc4 = rot ? scalar_kmeans_lower_bound<4>(b.ldr, rot) : 0, sc8 = rot ? scalar_kmeans_lower_bound<8>(b.ldr, rot) : 0;
            const int oldB[3] = { colour + sc4, colour + sc8, colour + sc4 }, newB[3] = { line4 + sc4, line4 + sc8, line8 + sc4 };
            for (int k = 0; k < 3; ++k) { ++lateAll[k]; latePruned[k][0] += oldB[k] > prevBest; latePruned[k][1] += newB[k] > prevBest; }
        }
    }
    printf("late modes (whole-block tasks against the best of modes 1 / 3): pruned now / with the along-the-line term -");
    { const char* nm[3] = { "mode 5", "mode 4 im0", "mode 4 im1" }; for (int k = 0; k < 3; ++k) printf(" %s %.1f %% / %.1f %% of %ld;", nm[k], 100.0 * latePruned[k][0] / lateAll[k], 100.0 * latePruned[k][1] / lateAll[k], lateAll[k]); }
    printf("\n");
    for (int m = 0; m < 2; ++m)
    {
        printf("mode %d: %.0f cost units (texel-evaluations) unpruned; searched now %.1f %%; with the in-mode final table %.1f %%; two-phase (best 4 by org first) %.1f %%; oracle %.1f %%; mean final/LB of the searched %.2f\n",
               m ? 3 : 1, tot[m], 100.0 * now[m] / tot[m], 100.0 * inmode[m] / tot[m], 100.0 * top4[m] / tot[m], 100.0 * oracle[m] / tot[m], ratioSum[m] / std::max(1L, ratioN[m]));
        printf("  EXPERIMENT along-the-line bound: searched %.1f %% (bound above a subset's final error: %ld subsets, %ld candidates); mean bound / final: now %.3f, new %.3f\n", 100.0 * now2[m] / tot[m], viol[m], violC[m], lbSum[m] / finSum[m], lb2Sum[m] / finSum[m]);
        printf("  sums over all subsets: line residual R1 %.0f, along-the-line term K0 %.0f, G = min over directions %.0f, bound after the rounding slack %.0f, final errors %.0f; slack-free sum of per-axis N-means %.0f\n", g_diagSum[m][0], g_diagSum[m][1], g_diagSum[m][2], g_diagSum[m][4], g_diagSum[m][3], g_diagSum[m][5]);
        printf("  searched if the bound were the plain line residual (no rounding slack; NOT valid, potential only): %.1f %%; cost-weighted final / residual of the searched: %.2f\n", 100.0 * noslack[m] / tot[m], fr[m] / std::max(1.0, frN[m]));
        printf("  cost share by LB / table of the searched (0.0-0.1 ... 1.0+):"); for (int k = 0; k < 12; ++k) printf(" %.1f%%", 100.0 * lbHist[m][k] / now[m]); printf("\n");
        printf("  cost share by the mode's best final error (0, 1-3, 4-15, 16-63, 64-255, 256-1023, 1024-4095, 4096+):"); for (int k = 0; k < 8; ++k) printf(" %.1f%%", 100.0 * costByBest[m][k] / now[m]); printf("\n");
        printf("  rank of the winning candidate:"); for (int k = 0; k < 16; ++k) printf(" %ld", winRankHist[m][k]); printf("\n");
    }
    return 0;
}
