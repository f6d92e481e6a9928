// A tighter exact pruning bound for BC7 - DEVELOPMENT MATERIAL, not part of the product (round 4: derived and checked on the host; round 5:
// moved out of directxtex_amd/csrc because no kernel uses it - its free 4-means term would remove 3 - 4 ms of mode 3's search for about 2 ms
// of fp64 sorting and dynamic programming per image in `pre`). tests/cpp/bc7_bound_check.cpp keeps it under the
// property test (never above the error of any palette), tools/bc7_prune_stats.cpp measures what it would prune.
//
// subset_lower_bound (bc7_core.h) is the residual of the best-fit line minus the rounding slack: it ignores that a palette has only N
// points on its line. For a palette on a line of direction u the texels' squared distances to the REAL line points split exactly into
// the distance to the line and the distance, along the line, to the nearest of the N points (Pythagoras):
//      sum_k min_i |p_k - L_i|^2  >=  R(u) + K_N(u),   R(u) = tr S - u'Su,   K_N(u) = optimum of "N points against the projections on u"
// (free N-means: an exact dynamic programme over the sorted projections; or, tighter, N points with BC7's fixed weights: every monotone
// assignment, least squares for offset and scale). The minimum over all directions: write u = c u0 + s v with v orthogonal to a fixed unit
// vector u0 (the principal axis, approximately), a = u0'Su0, g = |S u0 - a u0|, mu = tr S - a (>= v'Sv for every such v). Then
//      u'Su <= c^2 a + 2 c s g + s^2 mu <= c^2 a + s^2 mu + g,
//      sqrt K_N(u) >= c sqrt K_N(u0) - s sqrt mu        (the root of such a cost is the distance to a union of subspaces: 1-Lipschitz and
//                                                        homogeneous; the projections on u are c t(u0) + s t(v), |t(v)|^2 = v'Sv <= mu),
//      (x - y)+^2 >= (1 - e) x^2 - (1/e - 1) y^2        for every e in (0, 1),
// so R(u) + K_N(u) >= tr S - g - s^2 mu / e + c^2 ((1 - e) K0 - a), linear in c^2: >= min(tr S - a + (1 - e) K0, tr S - mu / e) - g. Any e
// is valid; the best one solves K0 e^2 + (a - K0) e - mu = 0. The rounding of the palette entries to integers costs 1/2 sqrt(C n) under the
// root exactly as in subset_lower_bound (Minkowski), and the plain residual bound is kept as a floor.
// Host statistics (tools/bc7_prune_stats.cpp, 1 500 blocks of the benchmark image, against the final error of every candidate of the lockstep
// search): 0 bounds above a final error; mode 3's searched share of its unpruned cost 77.7 % -> 69.1 % (free term), 63.1 % (fixed weights).
#pragma once
#include "../directxtex_amd/csrc/bc7_core.h"

namespace dxtex
{
namespace bc7
{
// optimum of N-means of n <= 16 sorted values (clusters are runs): D_k(j) = min_i D_{k-1}(i - 1) + SSE(i .. j)
template<int N>
DXTEX_HD double line_kmeans_free(const double (&t)[16], int n)
{
    if (n <= N) return 0.0;
    double s1[17], s2[17];
    s1[0] = 0.0; s2[0] = 0.0;
    for (int i = 0; i < n; ++i) { s1[i + 1] = s1[i] + t[i]; s2[i + 1] = s2[i] + t[i] * t[i]; }
    double d[16], e[16];
    for (int j = 0; j < n; ++j) { const double m = s1[j + 1]; const double c = s2[j + 1] - m * m / double(j + 1); d[j] = c > 0.0 ? c : 0.0; }
    for (int k = 2; k <= N; ++k)
    {
        for (int j = 0; j < n; ++j)
        {
            double best = (j < k) ? 0.0 : 1.0e300;
            for (int i = k - 1; i <= j; ++i)
            {
                const double m = s1[j + 1] - s1[i];
                const double c = (s2[j + 1] - s2[i]) - m * m / double(j - i + 1);
                const double v = d[i - 1] + (c > 0.0 ? c : 0.0);
                best = v < best ? v : best;
            }
            e[j] = best;
        }
        for (int j = 0; j < n; ++j) d[j] = e[j];
    }
    return d[n - 1];
}

// optimum of "the four points alpha + beta w_i / 64, w = 0, 21, 43, 64 (2-bit indices), against n sorted values": the nearest-point assignment
// of sorted values is monotone, so every choice of three cut positions, with least squares for (alpha, beta)
DXTEX_HD double line_fixed4(const double (&t)[16], int n)
{
    double p1[17];
    p1[0] = 0.0;
    double stt = 0.0;
    for (int i = 0; i < n; ++i) { p1[i + 1] = p1[i] + t[i]; stt += t[i] * t[i]; }
    const double st = p1[n], rn = 1.0 / double(n);
    const double vt = stt - st * st * rn;
    const double w1 = 21.0 / 64.0, w2 = 43.0 / 64.0;
    double best = vt > 0.0 ? vt : 0.0;                 // all texels on one point
    for (int b1 = 0; b1 <= n; ++b1)
        for (int b2 = b1; b2 <= n; ++b2)
            for (int b3 = b2; b3 <= n; ++b3)
            {
                const double n1 = double(b2 - b1), n2 = double(b3 - b2), n3 = double(n - b3);
                const double sx = n1 * w1 + n2 * w2 + n3, sxx = n1 * w1 * w1 + n2 * w2 * w2 + n3;
                const double vx = sxx - sx * sx * rn;
                if (!(vx > 1.0e-9)) continue;
                const double sxt = w1 * (p1[b2] - p1[b1]) + w2 * (p1[b3] - p1[b2]) + (p1[n] - p1[b3]);
                const double cxt = sxt - sx * st * rn;
                const double c = vt - cxt * cxt / vx;
                const double cc = c > 0.0 ? c : 0.0;
                best = cc < best ? cc : best;
            }
    return best;
}

// Lower bound of the error of a subset against ANY palette of N interpolated entries over C channels. N = 4 or 8 (free term), FIXED: N == 4
// with the fixed weights. Never below subset_lower_bound's value (the same residual bound is its floor).
template<int N, bool FIXED>
DXTEX_HD int subset_lower_bound_line(const uint32_t* pix, uint32_t mask16, uint32_t rot, int C)
{
    static_assert(!FIXED || N == 4, "fixed weights: 2-bit indices only");
    uint32_t n = 0, s[4] = { 0, 0, 0, 0 }, ss[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    double P[16][4];
    for (uint32_t i = 0; i < 16; ++i)
        if ((mask16 >> i) & 1u)
        {
            const uint32_t p = rotate_pixel(pix[i], rot);
            const uint32_t c0 = p & 0xFFu, c1 = (p >> 8) & 0xFFu, c2 = (p >> 16) & 0xFFu, c3 = (C == 4) ? (p >> 24) : 0u;
            P[n][0] = double(c0); P[n][1] = double(c1); P[n][2] = double(c2); P[n][3] = double(c3);
            ++n; s[0] += c0; s[1] += c1; s[2] += c2; s[3] += c3;
            ss[0] += c0 * c0; ss[1] += c0 * c1; ss[2] += c0 * c2; ss[3] += c0 * c3;
            ss[4] += c1 * c1; ss[5] += c1 * c2; ss[6] += c1 * c3;
            ss[7] += c2 * c2; ss[8] += c2 * c3; ss[9] += c3 * c3;
        }
    if (n < 2) return 0;
    // M = n * S, exact integers
    const int M00 = int(n * ss[0]) - int(s[0] * s[0]), M01 = int(n * ss[1]) - int(s[0] * s[1]), M02 = int(n * ss[2]) - int(s[0] * s[2]), M03 = int(n * ss[3]) - int(s[0] * s[3]);
    const int M11 = int(n * ss[4]) - int(s[1] * s[1]), M12 = int(n * ss[5]) - int(s[1] * s[2]), M13 = int(n * ss[6]) - int(s[1] * s[3]);
    const int M22 = int(n * ss[7]) - int(s[2] * s[2]), M23 = int(n * ss[8]) - int(s[2] * s[3]), M33 = int(n * ss[9]) - int(s[3] * s[3]);
    const int T = M00 + M11 + M22 + M33;
    if (T <= 0) return 0;
    const double inv = 1.0 / double(T);
    const double A[4][4] = { { M00 * inv, M01 * inv, M02 * inv, M03 * inv }, { M01 * inv, M11 * inv, M12 * inv, M13 * inv },
                             { M02 * inv, M12 * inv, M22 * inv, M23 * inv }, { M03 * inv, M13 * inv, M23 * inv, M33 * inv } };     // S / tr S
    // N^16 by four squarings: its Frobenius norm bounds lambda_max from above, its columns point along the principal axis
    double B[4][4], Q[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] = A[i][j];
    for (int k = 0; k < 4; ++k)
    {
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { double v = 0.0; for (int l = 0; l < 4; ++l) v += B[i][l] * B[l][j]; Q[i][j] = v; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) B[i][j] = Q[i][j];
    }
    double f2 = 0.0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) f2 += B[i][j] * B[i][j];
    double lam = sqrt(sqrt(sqrt(sqrt(sqrt(f2))))) * (1.0 + 1e-9);            // lambda_max(S) / tr S from above
    const double trS = double(T) / double(n);
    const double rPlain = (lam < 1.0) ? trS * (1.0 - lam) : 0.0;             // tr S - lambda_max: subset_lower_bound's residual
    // u0: the column of N^16 with the largest diagonal entry, normalised
    int jc = 0;
    for (int j = 1; j < 4; ++j) if (B[j][j] > B[jc][jc]) jc = j;
    double u0[4] = { B[0][jc], B[1][jc], B[2][jc], B[3][jc] };
    double nu = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2] + u0[3] * u0[3]);
    double G = rPlain;
    if (nu > 1e-150)
    {
        for (int i = 0; i < 4; ++i) u0[i] /= nu;
        // a = u0'Su0, g = |S u0 - a u0|, mu = tr S - a  (S = A * tr S)
        double Su[4];
        for (int i = 0; i < 4; ++i) Su[i] = (A[i][0] * u0[0] + A[i][1] * u0[1] + A[i][2] * u0[2] + A[i][3] * u0[3]) * trS;
        const double uu = u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2] + u0[3] * u0[3];          // 1 up to rounding
        const double a = (u0[0] * Su[0] + u0[1] * Su[1] + u0[2] * Su[2] + u0[3] * Su[3]) / uu;
        double g2 = 0.0;
        for (int i = 0; i < 4; ++i) { const double r = Su[i] - a * u0[i]; g2 += r * r; }
        const double g = sqrt(g2) * (1.0 + 1e-9) + 1e-9 * trS;
        const double mu = (trS - a) * (1.0 + 1e-9) + 1e-9 * trS;
        // the projections, sorted
        const double m0 = double(s[0]) / double(n), m1 = double(s[1]) / double(n), m2 = double(s[2]) / double(n), m3 = double(s[3]) / double(n);
        double t[16];
        for (uint32_t k = 0; k < n; ++k) t[k] = (P[k][0] - m0) * u0[0] + (P[k][1] - m1) * u0[1] + (P[k][2] - m2) * u0[2] + (P[k][3] - m3) * u0[3];
        for (uint32_t k = n; k < 16; ++k) t[k] = 0.0;
        for (uint32_t i = 1; i < n; ++i) { const double v = t[i]; int j = int(i) - 1; while (j >= 0 && t[j] > v) { t[j + 1] = t[j]; --j; } t[j + 1] = v; }
        const double K0 = (FIXED ? line_fixed4(t, int(n)) : line_kmeans_free<N>(t, int(n))) * (1.0 - 1e-9);
        if (K0 > 0.0)          // mu > 0 by construction
        {
            const double bq = a - K0;
            const double e = (-bq + sqrt(bq * bq + 4.0 * K0 * mu)) / (2.0 * K0);
            if (e > 0.0 && e < 1.0)
            {
                const double x1 = (trS - a) + (1.0 - e) * K0, x2 = trS - mu / e;
                const double Gn = (x1 < x2 ? x1 : x2) * (1.0 - 1e-9) - g;
                G = Gn > G ? Gn : G;
            }
        }
    }
    const int varying = (M00 > 0 ? 1 : 0) + (M11 > 0 ? 1 : 0) + (M22 > 0 ? 1 : 0) + (M33 > 0 ? 1 : 0);      // as subset_lower_bound: only channels that vary pay rounding slack
    const double d = sqrt(G > 0.0 ? G : 0.0) - 0.5 * sqrt(double(varying) * double(n)) - 1e-3;
    if (d <= 0.0) return 0;
    const double lb = d * d * 0.99999 - 1.0;
    return (lb > 0.0) ? int(lb) : 0;
}
} // namespace bc7
} // namespace dxtex
