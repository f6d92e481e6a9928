// BC6H integer pipeline shared by the decoder and the encoder (BC6HBC7.cpp: INTColor :452-575, Quantize :1864-1889,
// Unquantize :1893-1926, FinishUnquantize :1930-1940, TransformInverse :1153-1165).
// Texel components are half-float bit patterns carried as ints; unsigned formats drop negatives to 0.
#pragma once
#include <stdint.h>

#if defined(DXTEX_HOST_DEBUG)
#define DXTEX_HD6 __host__ __device__ inline
#else
#define DXTEX_HD6 __device__ __forceinline__
#endif

namespace dxtex
{
namespace bc6h
{
enum : int { F16S_MASK = 0x8000, F16EM_MASK = 0x7FFF, F16MAX = 0x7BFF };

// SIGN_EXTEND(x, nb) (BC6HBC7.cpp:47)
DXTEX_HD6 int sign_extend(int x, int nb) { return ((x & (1 << (nb - 1))) ? ((~0) ^ ((1 << nb) - 1)) : 0) | x; }

// INTColor::F16ToINT (:533-551): half bits -> int
DXTEX_HD6 int f16_to_int(uint32_t h, bool isSigned)
{
    if (isSigned)
    {
        const int s = int(h & F16S_MASK);
        int m = int(h & F16EM_MASK);
        if (m > F16MAX) m = F16MAX;
        return s ? -m : m;
    }
    return (h & F16S_MASK) ? 0 : int(h);
}

// INTColor::INT2F16 (:553-575): int -> half bits
DXTEX_HD6 uint32_t int_to_f16(int v, bool isSigned)
{
    if (isSigned)
    {
        int s = 0;
        if (v < 0) { s = F16S_MASK; v = -v; }
        return uint32_t(s | v) & 0xFFFFu;
    }
    return uint32_t(v) & 0xFFFFu;
}

DXTEX_HD6 int quantize(int v, int prec, bool isSigned)
{
    int q;
    if (isSigned)
    {
        int s = 0;
        if (v < 0) { s = 1; v = -v; }
        q = (prec >= 16) ? v : (v << (prec - 1)) / (F16MAX + 1);
        if (s) q = -q;
    }
    else
        q = (prec >= 15) ? v : (v << prec) / (F16MAX + 1);
    return q;
}

DXTEX_HD6 int unquantize(int comp, int bits, bool isSigned)
{
    int unq;
    if (isSigned)
    {
        if (bits >= 16) unq = comp;
        else
        {
            int s = 0;
            if (comp < 0) { s = 1; comp = -comp; }
            if (comp == 0) unq = 0;
            else if (comp >= ((1 << (bits - 1)) - 1)) unq = 0x7FFF;
            else unq = ((comp << 15) + 0x4000) >> (bits - 1);
            if (s) unq = -unq;
        }
    }
    else
    {
        if (bits >= 15) unq = comp;
        else if (comp == 0) unq = 0;
        else if (comp == ((1 << bits) - 1)) unq = 0xFFFF;
        else unq = ((comp << 16) + 0x8000) >> bits;
    }
    return unq;
}

// 24-bit multiply (operands here are at most 18 bits): v_mul_i32_i24 instead of the quarter-rate v_mul_lo_u32 the compiler emits when
// it cannot see the operand range.
DXTEX_HD6 int mul24i(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}

DXTEX_HD6 int finish_unquantize(int comp, bool isSigned)
{
    // (:1930-1940) the signed form works on the magnitude; written with selects - a per-lane branch here sat in the innermost
    // palette loop of the search kernels
    if (isSigned)
    {
        const int a = (comp < 0) ? -comp : comp;
        const int r = mul24i(a, 31) >> 5;
        return (comp < 0) ? -r : r;
    }
    return mul24i(comp, 31) >> 6;
}

} // namespace bc6h
} // namespace dxtex

// ======================================================================================================================
// BC6H encoder core (D3DX_BC6H::Encode, BC6HBC7.cpp:1817-1859): everything one lane computes for one region of one
// candidate (mode, shape). The kernels in bc6h_encode.hip pair the lanes of a candidate's two regions up and reduce
// over candidates. Unlike BC7 the errors are NOT exact integers here: they are fp32 sums of fp32 squares of values up
// to 2^17, accumulated texel by texel (Norm :1167-1173, MapColorsQuantized :2044-2077), and every comparison the
// search makes depends on those roundings. They are reproduced operation for operation (-ffp-contract=off), which is
// why the texels and palettes are kept as floats holding exact integers: float(a) - float(b) == float(a - b) here.
// Quirks kept: OptimizeEndPoints hands region 0 ALL sixteen texels (it indexes g_aPartitionTable with the region
// number instead of the region count, :2215); PerturbOne never tries a negative endpoint (:2112, :2118);
// RoughMSE skips the error of 1- and 2-texel regions (:2523-2534).
// ======================================================================================================================
#include "bc67_tables.h"

namespace dxtex
{
namespace bc6h
{
struct ModeRt
{
    int index;          // 0..13, position in the encoder's mode order
    int code;           // mode bits
    int regions2;       // 1 = two regions (32 shapes), 0 = one region
    int transformed;
    int prec;           // base endpoint precision (the same for r, g, b in every mode)
    int delta[3];       // precision of the other endpoints
};

DXTEX_HD6 int weight3(int i) { return i == 0 ? 0 : i == 1 ? 9 : i == 2 ? 18 : i == 3 ? 27 : i == 4 ? 37 : i == 5 ? 46 : i == 6 ? 55 : 64; }
DXTEX_HD6 int weight4(int i)
{
    return i == 0 ? 0 : i == 1 ? 4 : i == 2 ? 9 : i == 3 ? 13 : i == 4 ? 17 : i == 5 ? 21 : i == 6 ? 26 : i == 7 ? 30
         : i == 8 ? 34 : i == 9 ? 38 : i == 10 ? 43 : i == 11 ? 47 : i == 12 ? 51 : i == 13 ? 55 : i == 14 ? 60 : 64;
}
template<int N> DXTEX_HD6 int weight_of(int i) { return N == 8 ? weight3(i) : weight4(i); }

// NBits (:1176-1194). The reference counts with shift loops: for n > 0 the position of the highest set bit (+ 1 for a sign bit), for n < 0
// the shifts until -1 is left = the position of the highest CLEAR bit, + 1; both are 32 - clz (of n resp. ~n; clz(0) = 32 gives 0 for n = -1).
// As loops they were a third of bc6h_pre_kernel's and half of bc6h_post_kernel's instructions (six per-lane loops of up to 17 trips per
// EndPointsFit). The host search (tools/bc6h_debug.cpp) runs this function against D3DX_BC6H::Encode; main() there also compares it with the loops.
DXTEX_HD6 int nbits(int n, bool isSigned)
{
    const uint32_t m = uint32_t(n < 0 ? ~n : n);
#if defined(__HIP_DEVICE_COMPILE__)
    const int bits = 32 - __clz(int(m));
#else
    const int bits = m ? 32 - __builtin_clz(m) : 0;
#endif
    return n == 0 ? 0 : bits + ((n < 0 || isSigned) ? 1 : 0);
}

// Norm (:1167-1173) on floats that hold exact integers
DXTEX_HD6 float norm3(float pr, float pg, float pb, float qr, float qg, float qb)
{
    const float dr = pr - qr, dg = pg - qg, db = pb - qb;
    return dr * dr + dg * dg + db * db;
}

// The early-breaking palette scan shared by MapColors, MapColorsQuantized and AssignIndices: first local minimum.
template<int N>
DXTEX_HD6 float scan_min(const float (&e)[N])
{
    float res = e[N - 1];
#pragma unroll
    for (int i = N - 1; i >= 1; --i) res = (e[i] > e[i - 1]) ? e[i - 1] : res;
    return res;
}
template<int N>
DXTEX_HD6 float scan_min_idx(const float (&e)[N], uint32_t& idx)
{
    float best = e[0];
    bool done = false;
    idx = 0;
#pragma unroll
    for (int i = 1; i < N; ++i)
    {
        done = done || (e[i] > best) || !(best > 0.0f);
        if (!done && e[i] < best) { best = e[i]; idx = uint32_t(i); }
    }
    return best;
}

// A region's texels: texel k of the region is at block position pos(k); values as floats holding exact ints.
struct Texels
{
    typedef float T;
    const float* r; const float* g; const float* b;     // r[k * stride] etc.
    int stride;
    int np;
    DXTEX_HD6 void fetch(int k, uint32_t /*blockPos*/, float& pr, float& pg, float& pb) const { pr = r[k * stride]; pg = g[k * stride]; pb = b[k * stride]; }
};

// The same columns as 16-bit integers (texel components are half-float bit patterns, |v| <= 32767): half the LDS of a search kernel's
// columns, one conversion per fetch. What bc6h_perturb_filter_kernel keeps.
struct Texels16
{
    typedef int16_t T;
    const int16_t* r; const int16_t* g; const int16_t* b;
    int stride;
    int np;
    DXTEX_HD6 void fetch(int k, uint32_t /*blockPos*/, float& pr, float& pg, float& pb) const { pr = float(r[k * stride]); pg = float(g[k * stride]); pb = float(b[k * stride]); }
};

// The same texels read in place from the block's planes (r[16], g[16], b[16], one copy per block shared by the lanes that work on
// it): texel k of the region is at its block position. No per-lane copy - what pre / post of the two-region modes use.
struct TileTexels
{
    const float* planes;
    int np;
    DXTEX_HD6 void fetch(int /*k*/, uint32_t blockPos, float& pr, float& pg, float& pb) const { pr = planes[blockPos]; pg = planes[16 + blockPos]; pb = planes[32 + blockPos]; }
};

// Palette of quantised endpoints (GeneratePaletteQuantized, :1990-2040) for one channel
template<int N>
DXTEX_HD6 void palette_channel_unq(int ua, int ub, bool isSigned, float (&out)[N])      // from unquantised endpoints
{
    // (ua (64 - w) + ub w + 32) >> 6 as (64 ua + 32 + w (ub - ua)) >> 6: one 24-bit multiply-add per entry (|ub - ua| < 2^17)
    const int b64 = ua * 64 + 32, d = ub - ua;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        const int w = weight_of<N>(i);
        out[i] = float(finish_unquantize((mul24i(d, w) + b64) >> 6, isSigned));
    }
}
template<int N>
DXTEX_HD6 void palette_channel(int qa, int qb, int prec, bool isSigned, float (&out)[N])
{
    palette_channel_unq<N>(unquantize(qa, prec, isSigned), unquantize(qb, prec, isSigned), isSigned, out);
}

struct EndPts { int A[3], B[3]; };

// MapColorsQuantized (:2044-2077): total fp32 error of the texels against the palette of `ep`
template<int N, class TX>
DXTEX_HD6 float map_colors_q(const TX& tx, const float (&pr)[N], const float (&pg)[N], const float (&pb)[N], uint64_t pos = 0)
{
    float tot = 0.0f;
    uint64_t rest = pos;            // block positions (TileTexels only)
    for (int k = 0; k < tx.np; ++k)
    {
        float r, g, b;
        tx.fetch(k, uint32_t(rest) & 15u, r, g, b);
        rest >>= 4;
        float e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = norm3(r, g, b, pr[i], pg[i], pb[i]);
        tot += scan_min(e);
    }
    return tot;
}

// AssignIndices for one region (:2260-2301) + SwapIndices (:2228-2255). `pos` = 4-bit block positions of the region's
// texels; indices come back as 4 bits per block position.
template<int N, class TX>
DXTEX_HD6 float assign_indices6(const TX& tx, uint64_t pos, EndPts& ep, int prec, bool isSigned, uint32_t anchorPos, uint64_t& idxOut)
{
    float pr[N], pg[N], pb[N];
    palette_channel<N>(ep.A[0], ep.B[0], prec, isSigned, pr);
    palette_channel<N>(ep.A[1], ep.B[1], prec, isSigned, pg);
    palette_channel<N>(ep.A[2], ep.B[2], prec, isSigned, pb);
    float tot = 0.0f;
    uint64_t idx = 0, member = 0;
    uint64_t rest = pos;            // block positions of the texels still to come, 4 bits each
    for (int k = 0; k < tx.np; ++k)
    {
        const uint32_t p = uint32_t(rest) & 15u;
        rest >>= 4;
        float r, g, b;
        tx.fetch(k, p, r, g, b);
        float e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = norm3(r, g, b, pr[i], pg[i], pb[i]);
        uint32_t ix;
        tot += scan_min_idx(e, ix);
        idx |= uint64_t(ix) << (4 * p);
        member |= uint64_t(0xF) << (4 * p);
    }
    if ((idx >> (4 * anchorPos)) & uint64_t(N >> 1))
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int t = ep.A[c]; ep.A[c] = ep.B[c]; ep.B[c] = t; }
        idx ^= member & (uint64_t(N - 1) * 0x1111111111111111ull);
    }
    idxOut = idx;
    return tot;
}

// EndPointsFit for the endpoints ONE lane holds (:1945-1986): region 0 -> (A absolute, B delta); region 1 -> both delta
DXTEX_HD6 bool endpoints_fit(const EndPts& t, int region, const ModeRt& m, bool isSigned)
{
    const bool ds = m.transformed || isSigned;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        if (region == 0) ok = ok && (nbits(t.A[c], isSigned) <= m.prec);
        else ok = ok && (nbits(t.A[c], ds) <= m.delta[c]);
        ok = ok && (nbits(t.B[c], ds) <= m.delta[c]);
    }
    return ok;
}

// TransformForward for one lane's endpoints (:1146-1151); a0 = region 0's A
DXTEX_HD6 EndPts transform_forward(const EndPts& e, int region, const int (&a0)[3])
{
    EndPts t = e;
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        if (region != 0) t.A[c] = e.A[c] - a0[c];
        t.B[c] = e.B[c] - a0[c];
    }
    return t;
}

// ---- OptimizeOne (:2145-2194) as lockstep pieces ---------------------------------------------------------------------------
struct Perturb6
{
    EndPts ep;
    float err;
    int ch;         // 0..2, 3 = finished
    int sub;        // 0 = first pass on A, 1 = first pass on B, 2 = alternating loop
    int do_b;
    float err0;     // result of the first pass on A
    int new0;       // value found by the first pass on A
};

DXTEX_HD6 Perturb6 perturb6_begin(const EndPts& ep, float err)
{
    Perturb6 s; s.ep = ep; s.err = err; s.ch = 0; s.sub = 0; s.do_b = 0; s.err0 = 0.0f; s.new0 = 0;
    return s;
}

// Error of the region when channel s.ch of the endpoint being perturbed is `tmp` (the other endpoint's channel is fixedQ).
template<int N>
DXTEX_HD6 float perturb6_candidate(const Texels& tx, const Perturb6& s, const float (&base)[3][N], int fixedQ, int tmp, int prec, bool isSigned)
{
    float var[N];
    palette_channel<N>(s.do_b ? fixedQ : tmp, s.do_b ? tmp : fixedQ, prec, isSigned, var);
    float pr[N], pg[N], pb[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        pr[i] = (s.ch == 0) ? var[i] : base[0][i];
        pg[i] = (s.ch == 1) ? var[i] : base[1][i];
        pb[i] = (s.ch == 2) ? var[i] : base[2][i];
    }
    return map_colors_q<N>(tx, pr, pg, pb);
}

// ---- a LOWER BOUND on perturb6_candidate's result (bc6h_perturb_filter_kernel) ------------------------------------------------
// PerturbOne accepts a candidate only if its error is BELOW the best so far (:2124), so a candidate whose error provably is not
// needs no exact evaluation. The reference's error is Σ_k fl(first local minimum over the entries of fl(|p_k - q_i|^2)), every
// operation rounded to fp32: with u = 2^-24 it is at least (1 - 18 u) E*, E* = Σ_k min_i |p_k - q_i|^2 in real arithmetic (all
// terms are non-negative, a region has at most 16 texels; the first local minimum is never below the minimum). E* is evaluated
// through scores, |p - q|^2 = |p|^2 - (2 p.q - |q|^2): three FMAs per (texel, entry) and a running maximum instead of eight ordered
// operations and a compare / select pair, with the coordinates moved to the region's first texel o (an exact integer translation)
// so that the cancellation in |p|^2 - score is of the order of the region's spread, not of its brightness. Rounding: every
// intermediate of a score is at most G = |p'|^2 + 2 |q'|^2 in magnitude and passes six roundings (|q'|^2: three, the FMAs: three);
// the sum of the maxima adds np - 1 roundings of partial sums below H = Σ|p'_k|^2 + 2 np max_i |q'_i|^2, the sum of |p'_k|^2 np + 2,
// the subtraction one: |computed - E*| <= 42 u H, and 18 u E* <= 36 u H (E* <= 2 H). The bound subtracts 2^-17 H = 128 u H. The same
// holds for a PREFIX of the texels (the other texels' errors are >= 0; Bound6::pre holds the prefixes of Σ|p'_k|^2 over 4 / 8 / 12
// texels), which is what lets the loop stop as soon as a prefix already excludes the candidate(s).
// tools/bc6h_debug.cpp (-DDXTEX_COUNT_EVALS6) checks bound <= exact on every candidate of the host search and counts what passes.
#if defined(DXTEX_COUNT_EVALS6)
void count_bound6(int n, int np, int step, float bound, float exact, float best);     // tools/bc6h_debug.cpp
void count_prefix6(const float* r, const float* g, const float* b, int stride, int np, const float* pr, const float* pg, const float* pb, float best, int step);
#endif
struct Bound6 { float o[3]; float pp; float pre[3]; };   // per task: the centre, fl(Σ |p_k - o|^2), and the sum's prefixes over 4 / 8 / 12 texels

template<class TX>
DXTEX_HD6 Bound6 bound6_begin(const TX& tx)
{
    Bound6 b; b.o[0] = float(tx.r[0]); b.o[1] = float(tx.g[0]); b.o[2] = float(tx.b[0]); b.pre[0] = b.pre[1] = b.pre[2] = 0.0f;
    float pp = 0.0f;
    for (int k = 0; k < tx.np; ++k)
    {
        const float x = float(tx.r[k * tx.stride]) - b.o[0], y = float(tx.g[k * tx.stride]) - b.o[1], z = float(tx.b[k * tx.stride]) - b.o[2];
        pp += __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x));
        if (k == 3) b.pre[0] = pp;
        if (k == 7) b.pre[1] = pp;
        if (k == 11) b.pre[2] = pp;
    }
    b.pp = pp;
    return b;
}

// per PerturbOne call: the walked channel first (the order of a dot product's terms is free in a bound), the other two channels'
// centred palettes and minus their squares
template<int N, class T = float>
struct MacroBound6
{
    const T* pv; const T* p1; const T* p2;                  // texel planes: walked channel, the two fixed ones
    float ov, n2ov, n2o1, n2o2;
    float f1[N], f2[N], baseN[N];
};

template<int N, class TX>
DXTEX_HD6 MacroBound6<N, typename TX::T> bound6_macro(const TX& tx, const Bound6& bd, int ch, const float (&fix1)[N], const float (&fix2)[N])      // fix1 / fix2: the palettes of channels (ch + 1) % 3 and (ch + 2) % 3
{
    MacroBound6<N, typename TX::T> m;
    const int og = int(tx.g - tx.r), ob2 = int(tx.b - tx.r);              // plane offsets (one array on the device: the pointers stay LDS pointers)
    m.pv = tx.r + ((ch == 0) ? 0 : (ch == 1) ? og : ob2);
    m.p1 = tx.r + ((ch == 0) ? og : (ch == 1) ? ob2 : 0);
    m.p2 = tx.r + ((ch == 0) ? ob2 : (ch == 1) ? 0 : og);
    const float o0 = bd.o[0], o1 = bd.o[1], o2 = bd.o[2];
    m.ov = (ch == 0) ? o0 : (ch == 1) ? o1 : o2;
    const float oa = (ch == 0) ? o1 : (ch == 1) ? o2 : o0, ob = (ch == 0) ? o2 : (ch == 1) ? o0 : o1;
    m.n2ov = -2.0f * m.ov; m.n2o1 = -2.0f * oa; m.n2o2 = -2.0f * ob;
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        const float a = fix1[i] - oa;
        const float b = fix2[i] - ob;
        m.f1[i] = a; m.f2[i] = b;
        m.baseN[i] = -__builtin_fmaf(b, b, a * a);
    }
    return m;
}

// The two candidates of a PerturbOne step (cur - step, cur + step) in one pass over the texels; the texel fetches and the doubled centred
// coordinates are shared. `best` is the error a candidate has to beat: every fourth texel a lane whose prefix bounds already exclude both
// candidates leaves the loop (the large steps of the logarithmic search end after four texels: their candidates are hopeless).
// Returns lower bounds of the two candidates' errors (of a prefix of the texels when the loop was left early).
template<int N, class TX>
DXTEX_HD6 void perturb6_bound_pair(const TX& tx, const Bound6& bd, const MacroBound6<N, typename TX::T>& m, const float (&varM)[N], const float (&varP)[N],
                                   float best, float& lbM, float& lbP)
{
    float vqM[N], qnM[N], vqP[N], qnP[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        vqM[i] = varM[i] - m.ov; qnM[i] = __builtin_fmaf(-vqM[i], vqM[i], m.baseN[i]);
        vqP[i] = varP[i] - m.ov; qnP[i] = __builtin_fmaf(-vqP[i], vqP[i], m.baseN[i]);
    }
    float qminM = qnM[0], qminP = qnP[0];
#pragma unroll
    for (int i = 1; i < N; ++i) { qminM = __builtin_fminf(qminM, qnM[i]); qminP = __builtin_fminf(qminP, qnP[i]); }
    const float n2 = float(2 * tx.np);
    const float mgM = 0x1p-17f * __builtin_fmaf(n2, -qminM, bd.pp), mgP = 0x1p-17f * __builtin_fmaf(n2, -qminP, bd.pp);      // 2^-17 H, exact scaling
    float SM = 0.0f, SP = 0.0f, ppk = bd.pp;
    for (int k = 0; k < tx.np; ++k)
    {
        const float a = __builtin_fmaf(float(m.pv[k * tx.stride]), 2.0f, m.n2ov), b = __builtin_fmaf(float(m.p1[k * tx.stride]), 2.0f, m.n2o1), c = __builtin_fmaf(float(m.p2[k * tx.stride]), 2.0f, m.n2o2);
        float mM = __builtin_fmaf(a, vqM[0], __builtin_fmaf(b, m.f1[0], __builtin_fmaf(c, m.f2[0], qnM[0])));
        float mP = __builtin_fmaf(a, vqP[0], __builtin_fmaf(b, m.f1[0], __builtin_fmaf(c, m.f2[0], qnP[0])));
#pragma unroll
        for (int i = 1; i < N; ++i)
        {
            mM = __builtin_fmaxf(mM, __builtin_fmaf(a, vqM[i], __builtin_fmaf(b, m.f1[i], __builtin_fmaf(c, m.f2[i], qnM[i]))));
            mP = __builtin_fmaxf(mP, __builtin_fmaf(a, vqP[i], __builtin_fmaf(b, m.f1[i], __builtin_fmaf(c, m.f2[i], qnP[i]))));
        }
        SM = (k == 0) ? mM : SM + mM;
        SP = (k == 0) ? mP : SP + mP;
        if ((k & 3) == 3 && k < 12)
        {
            const float pk = (k == 3) ? bd.pre[0] : (k == 7) ? bd.pre[1] : bd.pre[2];
            if (!((pk - SM) - mgM < best) && !((pk - SP) - mgP < best)) { ppk = pk; break; }
        }
    }
    lbM = (ppk - SM) - mgM;
    lbP = (ppk - SP) - mgP;
}

template<int N, class TX>
DXTEX_HD6 float perturb6_bound(const TX& tx, const Bound6& bd, const MacroBound6<N, typename TX::T>& m, const float (&var)[N], float best)
{
    float lbM, lbP;
    perturb6_bound_pair<N>(tx, bd, m, var, var, best, lbM, lbP);
    return lbM;
}

// One PerturbOne call (:2081-2141): 2 * prec - 1 candidate evaluations, straight-line.
template<int N>
DXTEX_HD6 void perturb6_macro(const Texels& tx, const Perturb6& s, int prec, bool isSigned, float& outErr, int& outVal)
{
    // palettes of the three channels for the current endpoints; channel s.ch is rebuilt per candidate
    float base[3][N];
#pragma unroll
    for (int c = 0; c < 3; ++c) palette_channel<N>(s.ep.A[c], s.ep.B[c], prec, isSigned, base[c]);
    const int fixedQ = (s.ch == 0) ? (s.do_b ? s.ep.A[0] : s.ep.B[0]) : (s.ch == 1) ? (s.do_b ? s.ep.A[1] : s.ep.B[1]) : (s.do_b ? s.ep.A[2] : s.ep.B[2]);
    int cur = (s.ch == 0) ? (s.do_b ? s.ep.B[0] : s.ep.A[0]) : (s.ch == 1) ? (s.do_b ? s.ep.B[1] : s.ep.A[1]) : (s.do_b ? s.ep.B[2] : s.ep.A[2]);
    float minErr = s.err;
#if defined(DXTEX_COUNT_EVALS6)
    const Bound6 cbd = bound6_begin(tx);
    const MacroBound6<N, float> cmb = bound6_macro<N>(tx, cbd, s.ch, base[(s.ch + 1) % 3], base[(s.ch + 2) % 3]);
#define DXTEX_COUNT6(tmp_, e_, step_) do { if (valid) { float var_[N]; palette_channel<N>(s.do_b ? fixedQ : (tmp_), s.do_b ? (tmp_) : fixedQ, prec, isSigned, var_); \
        count_bound6(N, tx.np, step_, perturb6_bound<N>(tx, cbd, cmb, var_, minErr), e_, minErr); \
        if (N == 8) { float pr_[N], pg_[N], pb_[N]; for (int i_ = 0; i_ < N; ++i_) { pr_[i_] = (s.ch == 0) ? var_[i_] : base[0][i_]; pg_[i_] = (s.ch == 1) ? var_[i_] : base[1][i_]; pb_[i_] = (s.ch == 2) ? var_[i_] : base[2][i_]; } \
                      count_prefix6(tx.r, tx.g, tx.b, tx.stride, tx.np, pr_, pg_, pb_, minErr, step_); } } } while (0)
#else
#define DXTEX_COUNT6(tmp_, e_, step_) do { } while (0)
#endif
    // The first step is half the range: cur - step is legal only for cur >= step, cur + step only for cur < step - never both, and the
    // reference skips the other one (:2112). One evaluation instead of two.
    {
        const int half = 1 << (prec - 1);
        const int tmp = (cur >= half) ? cur - half : cur + half;
        const bool valid = (tmp >= 0) && (tmp < (1 << prec));
        const float e = perturb6_candidate<N>(tx, s, base, fixedQ, tmp, prec, isSigned);
        DXTEX_COUNT6(tmp, e, 0);
        if (valid && e < minErr) { minErr = e; cur = tmp; }
    }
#pragma unroll 1
    for (int step = 1 << (prec - 1) >> 1; step; step >>= 1)
    {
        int beststep = 0;
#pragma unroll 1
        for (int sign = -1; sign <= 1; sign += 2)
        {
            const int tmp = cur + sign * step;
            const bool valid = (tmp >= 0) && (tmp < (1 << prec));
            const float e = perturb6_candidate<N>(tx, s, base, fixedQ, tmp, prec, isSigned);
            DXTEX_COUNT6(tmp, e, step);
            if (valid && e < minErr) { minErr = e; beststep = sign * step; }
        }
        cur += beststep;
    }
    outErr = minErr; outVal = cur;
}

DXTEX_HD6 void set_channel(EndPts& ep, int ch, int do_b, int v)
{
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c == ch) { if (do_b) ep.B[c] = v; else ep.A[c] = v; }
}

DXTEX_HD6 Perturb6 perturb6_transition(const Perturb6& in, float e, int val)
{
    Perturb6 s = in;
    if (in.sub == 0) { s.err0 = e; s.new0 = val; s.sub = 1; s.do_b = 1; return s; }
    if (in.sub == 1)
    {
        // e = fErr1 (B side, value `val`), err0 / new0 = A side
        if (in.err0 < e)
        {
            if (in.err0 >= in.err) { s.ch = in.ch + 1; s.sub = 0; s.do_b = 0; return s; }
            set_channel(s.ep, in.ch, 0, in.new0); s.err = in.err0; s.do_b = 1;
        }
        else
        {
            if (e >= in.err) { s.ch = in.ch + 1; s.sub = 0; s.do_b = 0; return s; }
            set_channel(s.ep, in.ch, 1, val); s.err = e; s.do_b = 0;
        }
        s.sub = 2;
        return s;
    }
    if (e >= in.err) { s.ch = in.ch + 1; s.sub = 0; s.do_b = 0; return s; }
    set_channel(s.ep, in.ch, in.do_b, val); s.err = e; s.do_b = 1 - in.do_b;
    return s;
}

template<int N>
DXTEX_HD6 void optimize_one6(const Texels& tx, const EndPts& org, float orgErr, int prec, bool isSigned, EndPts& opt)
{
    Perturb6 s = perturb6_begin(org, orgErr);
    while (s.ch < 3)
    {
        float e; int v;
        perturb6_macro<N>(tx, s, prec, isSigned, e, v);
        s = perturb6_transition(s, e, v);
    }
    opt = s.ep;
}

// ---- rough pass pieces -----------------------------------------------------------------------------------------------------
// INTColor::Set (:498-508) for one float: XMStoreHalf4 (round to nearest even) then F16ToINT
DXTEX_HD6 int float_to_int16f(float v, bool isSigned)
{
#if defined(DXTEX_HOST_DEBUG)
    const uint32_t h = dxtex_host_float_to_half(v);
#else
    const uint32_t h = __half_as_ushort(__float2half_rn(v));
#endif
    return f16_to_int(h, isSigned);
}

DXTEX_HD6 int clamp_seed(int v, bool isSigned)
{
    const int lo = isSigned ? -int(F16MAX) : 0, hi = int(F16MAX);
    v = (v > lo) ? v : lo;              // std::min(iMax, std::max(iMin, v))
    return (v < hi) ? v : hi;
}

// MapColors (:2467-2494): rough error of a region against the UNQUANTISED palette of the seed (:2430-2464)
template<int N, class TX>
DXTEX_HD6 float rough_error6(const TX& tx, const EndPts& seed, uint64_t pos = 0)
{
    float pr[N], pg[N], pb[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
    {
        const int w = weight_of<N>(i);
        pr[i] = float((seed.A[0] * (64 - w) + seed.B[0] * w + 32) >> 6);
        pg[i] = float((seed.A[1] * (64 - w) + seed.B[1] * w + 32) >> 6);
        pb[i] = float((seed.A[2] * (64 - w) + seed.B[2] * w + 32) >> 6);
    }
    return map_colors_q<N>(tx, pr, pg, pb, pos);
}

// ---- EmitBlock (:2330-2373) --------------------------------------------------------------------------------------------------
struct Bits128w
{
    uint64_t lo, hi; uint32_t pos;
    DXTEX_HD6 void put(uint32_t nbits, uint32_t value)
    {
        if (!nbits) return;
        const uint64_t v = uint64_t(value) & ((uint64_t(1) << nbits) - 1);
        if (pos < 64) { lo |= v << pos; if (pos + nbits > 64) hi |= v >> (64 - pos); }
        else hi |= v << (pos - 64);
        pos += nbits;
    }
};

// ep[0..3] = A0, B0, A1, B1 as they go into the block (delta-transformed when the mode is)
DXTEX_HD6 void emit_block6(const ModeRt& m, uint32_t shape, const int (&ep)[4][3], uint64_t idx, uint32_t anchor1, uint64_t& lo, uint64_t& hi)
{
    Bits128w w; w.lo = 0; w.hi = 0; w.pos = 0;
    const uint32_t headerBits = m.regions2 ? 82u : 65u;
    const uint8_t* desc = kBc6hHeader[m.index];
    for (uint32_t bit = 0; bit < headerBits; ++bit)
    {
        const uint32_t f = desc[bit] >> 4, k = desc[bit] & 15u;
        uint32_t v;
        if (f == 1) v = uint32_t(m.code) >> k;
        else if (f == 2) v = shape >> k;
        else { const uint32_t q = f - 3; v = uint32_t(ep[q & 3][q >> 2] >> k); }
        w.put(1, v & 1u);
    }
    const uint32_t ib = m.regions2 ? 3u : 4u;
    for (uint32_t i = 0; i < 16; ++i)
    {
        const bool isAnchor = (i == 0) || (m.regions2 && i == anchor1);
        w.put(isAnchor ? ib - 1 : ib, uint32_t(idx >> (4 * i)) & 15u);
    }
    lo = w.lo; hi = w.hi;
}
} // namespace bc6h
} // namespace dxtex
