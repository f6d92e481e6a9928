// BC1 / BC2 / BC3 / BC4 / BC5 encoders for gfx950: one 4x4 block per lane.
//
// Every lane gathers its own tile with four coalesced 16-byte row loads (lane = block, so a wave reads
// 1 KiB contiguous per tile row), runs the whole endpoint fit in registers and stores one 8- or 16-byte
// block; consecutive lanes store consecutive blocks. No LDS, no cross-lane traffic.
//
// The arithmetic restates, operation for operation in IEEE fp32 without FMA contraction, what the
// reference CPU encoders compute, so the output is bit-identical to them:
//   BC1 colour block  : BC.cpp:370-685 (EncodeBC1) + BC.cpp:65-314 (OptimizeRGB) + :44-61 (Encode565)
//   BC1 entry / alpha : BC.cpp:738-795   BC2: BC.cpp:828-895   BC3: BC.cpp:944-1141
//   alpha endpoint fit: BC.h:187-311 (OptimizeAlpha<bRange>)
//   BC4/BC5           : BC4BC5.cpp:183-293 (FindEndPoints*), :325-377 (FindClosest*), :419-562 (entry points)
// This translation unit must be compiled with -ffp-contract=off (see csrc/Makefile).
#include "dxtex_device.h"
#include "dxtex_dev.h"
#include "dxtex_kernels.h"

#if !defined(DXTEX_BC15_CODES)
#define DXTEX_BC15_CODES 1             // 0: the colour fit on 48 floats per lane as in rounds 2 - 3 (A/B builds)
#endif

namespace dxtex
{
namespace
{
// Floyd-Steinberg taps inside the 4x4 tile: right 7/16, down-left 3/16, down 5/16, down-right 1/16
// (BC.cpp:451-481). `i` is a compile-time constant after unrolling.
template<int N>
__device__ __forceinline__ void diffuse(float (&err)[N], int i, float d)
{
    if (3 != (i & 3)) err[i + 1] += d * (7.0f / 16.0f);
    if (i < 12)
    {
        if (i & 3) err[i + 3] += d * (3.0f / 16.0f);
        err[i + 4] += d * (5.0f / 16.0f);
        if (3 != (i & 3)) err[i + 5] += d * (1.0f / 16.0f);
    }
}

__device__ __forceinline__ uint32_t encode565(float r, float g, float b)
{
    r = (r < 0.0f) ? 0.0f : (r > 1.0f) ? 1.0f : r;
    g = (g < 0.0f) ? 0.0f : (g > 1.0f) ? 1.0f : g;
    b = (b < 0.0f) ? 0.0f : (b > 1.0f) ? 1.0f : b;
    return (uint32_t(int32_t(r * 31.0f + 0.5f)) << 11) | (uint32_t(int32_t(g * 63.0f + 0.5f)) << 5) | uint32_t(int32_t(b * 31.0f + 0.5f));
}

__device__ __forceinline__ void decode565(uint32_t w, float& r, float& g, float& b)
{
    r = float((w >> 11) & 31) * (1.0f / 31.0f);
    g = float((w >> 5) & 63) * (1.0f / 63.0f);
    b = float(w & 31) * (1.0f / 31.0f);
}

// Interpolation coefficient tables of the Newton fits (pC3 / pD3, pC4 / pD4, BC.cpp:26-31), selected without indexing memory and
// without control flow: the middle entries are picked once per block from the step count, an entry is three selects.
struct StepCoef
{
    // pC3 = {1, 1/2, 0}, pD3 = {0, 1/2, 1}; pC4 = {1, 2/3, 1/3, 0}, pD4 = {0, 1/3, 2/3, 1} (BC.cpp:26-31) as ONE multiplication each:
    // pD[k] = k * r and pC[k] = (steps - 1 - k) * r with r = 1/2 or float(1/3). Exact for every entry: k * 0.5f is exact, and with
    // r = 0x3EAAAAAB (1/3 rounded up) 2 r is the table's 2.0f/3.0f (a power-of-two scaling) and 3 r = 1.00000003 rounds to 1.0f.
    // (Round 2 picked the entries with three float compare-selects per coefficient: 12 of the ~45 operations per texel and trip.)
    float r, last;
    __device__ __forceinline__ explicit StepCoef(uint32_t cSteps)
    {
        const bool three = (cSteps == 3);
        r = three ? 0.5f : (1.0f / 3.0f);
        last = three ? 2.0f : 3.0f;
    }
    // k = the step index as a float (0, 1, 2 or 3)
    __device__ __forceinline__ float c(float k) const { return (last - k) * r; }
    __device__ __forceinline__ float d(float k) const { return k * r; }
};

// What the colour fit reads: the sixteen quantised (5:6:5 grid), luminance-weighted colours of the block (BC.cpp:418-490).
// ArrayColors: as 48 floats (any tile, the dithered path). CodeColors: as one dword of 5:6:5 codes per texel, decoded where a component is
// read - float(code) * (1/31 | 1/63), then the luminance weight: the operations that produced the array's floats, so the same floats - which
// keeps 16 registers live across the Newton loop instead of 48. The fit needs ~200 registers with the array (two waves per SIMD, where a
// SIMD that loses one wave to a stall issues at a single wave's rate - 4 cycles per instruction even for the 2-cycle opcodes,
// profiles/r02_valu_rates.md, W = 1 column); with the codes it fits four.
struct ArrayColors
{
    const float (&pr)[16]; const float (&pg)[16]; const float (&pb)[16];
    __device__ __forceinline__ float r(int i) const { return pr[i]; }
    __device__ __forceinline__ float g(int i) const { return pg[i]; }
    __device__ __forceinline__ float b(int i) const { return pb[i]; }
    __device__ __forceinline__ void launder() const {}
    template<class F>
    __device__ __forceinline__ void for_texels(F&& f)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) f(i);
    }
};
struct CodeColors
{
    uint32_t q[16];          // r5 | g6 << 8 | b5 << 16
    float wr, wb;            // luminance weights of red and blue (1.0f with BC_FLAGS_UNIFORM: x * 1.0f is x; green's weight is 1)
    __device__ __forceinline__ float r(int i) const { return float(q[i] & 0xFFu) * (1.0f / 31.0f) * wr; }
    __device__ __forceinline__ float g(int i) const { return float((q[i] >> 8) & 0xFFu) * (1.0f / 63.0f); }
    __device__ __forceinline__ float b(int i) const { return float((q[i] >> 16) & 0xFFu) * (1.0f / 31.0f) * wb; }
    // keeps the optimiser from hoisting the 48 decoded floats out of a loop (they are loop invariants): no instruction is emitted
    __device__ __forceinline__ void launder()
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(q[i]));
    }
    // f(k) for the texels in order, four per trip of a ROLLED loop: the trip works on q[0..3] and then rotates the sixteen registers by
    // four (sixteen moves), so the body exists once - four texels of temporaries - instead of sixteen times interleaved by the scheduler,
    // which is what held the fit at ~200 registers whatever the texels were kept as. After four trips the order is the original one.
    template<class F>
    __device__ __forceinline__ void for_texels(F&& f)
    {
#pragma unroll 1
        for (int grp = 0; grp < 4; ++grp)
        {
            f(0); f(1); f(2); f(3);
            const uint32_t t0 = q[0], t1 = q[1], t2 = q[2], t3 = q[3];
#pragma unroll
            for (int k = 0; k < 12; ++k) q[k] = q[k + 4];
            q[12] = t0; q[13] = t1; q[14] = t2; q[15] = t3;
        }
    }
};

// 6-D Newton endpoint fit over 16 points (BC.cpp:65-314).
template<class CS>
__device__ __forceinline__ void optimize_rgb16(CS& cs, uint32_t cSteps, bool uniform,
                               float& oXr, float& oXg, float& oXb, float& oYr, float& oYg, float& oYb)
{
    constexpr float fEpsilon = (0.25f / 64.0f) * (0.25f / 64.0f);

    float Xr = uniform ? 1.0f : (0.2125f / 0.7154f);
    float Xg = 1.0f;
    float Xb = uniform ? 1.0f : (0.0721f / 0.7154f);
    float Yr = 0.0f, Yg = 0.0f, Yb = 0.0f;

    cs.for_texels([&](int i)
    {
        const float vr = cs.r(i), vg = cs.g(i), vb = cs.b(i);
        if (vr < Xr) Xr = vr;
        if (vg < Xg) Xg = vg;
        if (vb < Xb) Xb = vb;
        if (vr > Yr) Yr = vr;
        if (vg > Yg) Yg = vg;
        if (vb > Yb) Yb = vb;
    });

    const float ABr = Yr - Xr, ABg = Yg - Xg, ABb = Yb - Xb;
    const float fAB = ABr * ABr + ABg * ABg + ABb * ABb;

    if (fAB < 1.175494351e-38f)   // FLT_MIN: single colour block
    {
        oXr = Xr; oXg = Xg; oXb = Xb; oYr = Yr; oYg = Yg; oYb = Yb;
        return;
    }

    const float fABInv = 1.0f / fAB;
    float Dr = ABr * fABInv, Dg = ABg * fABInv, Db = ABb * fABInv;
    const float Mr = (Xr + Yr) * 0.5f, Mg = (Xg + Yg) * 0.5f, Mb = (Xb + Yb) * 0.5f;

    float fDir0 = 0.0f, fDir1 = 0.0f, fDir2 = 0.0f, fDir3 = 0.0f;
    cs.for_texels([&](int i)
    {
        const float Pr = (cs.r(i) - Mr) * Dr;
        const float Pg = (cs.g(i) - Mg) * Dg;
        const float Pb = (cs.b(i) - Mb) * Db;
        float f;
        f = Pr + Pg + Pb; fDir0 += f * f;
        f = Pr + Pg - Pb; fDir1 += f * f;
        f = Pr - Pg + Pb; fDir2 += f * f;
        f = Pr - Pg - Pb; fDir3 += f * f;
    });

    float fDirMax = fDir0; uint32_t iDirMax = 0;
    if (fDir1 > fDirMax) { fDirMax = fDir1; iDirMax = 1; }
    if (fDir2 > fDirMax) { fDirMax = fDir2; iDirMax = 2; }
    if (fDir3 > fDirMax) { fDirMax = fDir3; iDirMax = 3; }

    if (iDirMax & 2) { const float f = Xg; Xg = Yg; Yg = f; }
    if (iDirMax & 1) { const float f = Xb; Xb = Yb; Yb = f; }

    if (fAB < 1.0f / 4096.0f)   // two colour block
    {
        oXr = Xr; oXg = Xg; oXb = Xb; oYr = Yr; oYg = Yg; oYb = Yb;
        return;
    }

    const float fSteps = float(cSteps - 1);
    const StepCoef coef(cSteps);

#pragma unroll 1
    for (int iter = 0; iter < 8; ++iter)
    {
        Dr = Yr - Xr; Dg = Yg - Xg; Db = Yb - Xb;
        const float fLen = Dr * Dr + Dg * Dg + Db * Db;
        if (fLen < (1.0f / 4096.0f))
            break;

        const float fScale = fSteps / fLen;
        Dr *= fScale; Dg *= fScale; Db *= fScale;

        float d2X = 0.0f, d2Y = 0.0f;
        float dXr = 0.0f, dXg = 0.0f, dXb = 0.0f, dYr = 0.0f, dYg = 0.0f, dYb = 0.0f;

        cs.for_texels([&](int i)
        {
            const float vr = cs.r(i), vg = cs.g(i), vb = cs.b(i);
            const float fDot = (vr - Xr) * Dr + (vg - Xg) * Dg + (vb - Xb) * Db;

            // fDot <= 0 -> 0, fDot >= fSteps -> cSteps - 1, else uint32(fDot + 0.5f) (BC.cpp:225-231): the clamp maps the two outer
            // cases onto the same conversion, so the three-way branch becomes straight-line code (inputs are finite)
            const float kStep = float(uint32_t(fminf(fmaxf(fDot, 0.0f), fSteps) + 0.5f));

            const float c = coef.c(kStep), d = coef.d(kStep);
            // pSteps[iStep] = X * pC[iStep] + Y * pD[iStep], evaluated where it is used
            const float diffR = (Xr * c + Yr * d) - vr;
            const float diffG = (Xg * c + Yg * d) - vg;
            const float diffB = (Xb * c + Yb * d) - vb;

            const float fC = c * (1.0f / 8.0f);
            const float fD = d * (1.0f / 8.0f);

            d2X += fC * c;
            dXr += fC * diffR; dXg += fC * diffG; dXb += fC * diffB;
            d2Y += fD * d;
            dYr += fD * diffR; dYg += fD * diffG; dYb += fD * diffB;
            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four texels at a time: keeps the temporaries of an iteration within the register budget
        });

        if (d2X > 0.0f)
        {
            const float f = -1.0f / d2X;
            Xr += dXr * f; Xg += dXg * f; Xb += dXb * f;
        }
        if (d2Y > 0.0f)
        {
            const float f = -1.0f / d2Y;
            Yr += dYr * f; Yg += dYg * f; Yb += dYb * f;
        }

        if ((dXr * dXr < fEpsilon) && (dXg * dXg < fEpsilon) && (dXb * dXb < fEpsilon) &&
            (dYr * dYr < fEpsilon) && (dYg * dYg < fEpsilon) && (dYb * dYb < fEpsilon))
            break;
    }

    oXr = Xr; oXg = Xg; oXb = Xb; oYr = Yr; oYg = Yg; oYb = Yb;
}

// Two views of a lane's 4x4 tile. TileView: the generic float tile (any source format, partial blocks, conversions). PackedTile: the
// sixteen RGBA8 texels as loaded - 16 registers instead of 64 - with XMLoadUByteN4's conversion (b * (1 / 255.f), the same expression
// load_tile uses) applied wherever a component is read; with it the BC1-BC5 kernels fit twice as many waves per SIMD.
struct TileView
{
    static constexpr bool kPacked = false;
    const Tile& t; const float (&al)[16];
    __device__ __forceinline__ void launder() const {}
    __device__ __forceinline__ float r(int i) const { return t.r[i]; }
    __device__ __forceinline__ float g(int i) const { return t.g[i]; }
    __device__ __forceinline__ float b(int i) const { return t.b[i]; }
    __device__ __forceinline__ float a(int i) const { return al[i]; }
};
struct PackedTile
{
    static constexpr bool kPacked = true;
    uint32_t px[16];
    __device__ __forceinline__ float r(int i) const { return float(px[i] & 0xFFu) * (1.0f / 255.0f); }
    __device__ __forceinline__ float g(int i) const { return float((px[i] >> 8) & 0xFFu) * (1.0f / 255.0f); }
    __device__ __forceinline__ float b(int i) const { return float((px[i] >> 16) & 0xFFu) * (1.0f / 255.0f); }
    __device__ __forceinline__ float a(int i) const { return float(px[i] >> 24) * (1.0f / 255.0f); }
    // Makes the texels opaque to the optimiser: conversions after this point are recomputed from the packed words instead of being
    // kept alive (as 48 floats) from their first use across the Newton fit.
    __device__ __forceinline__ void launder()
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(px[i]));
    }
};

// BC1 colour block (BC.cpp:370-685). Alpha is only read when bColorKey is set.
template<bool DITHER, class TS, bool CODES = false>
__device__ __forceinline__ uint2 encode_bc1_color(TS& s, bool bColorKey, float threshold, uint32_t flags)
{
    const bool uniform = (flags & BCF_UNIFORM) != 0;
    constexpr float LumR = 0.2125f / 0.7154f, LumB = 0.0721f / 0.7154f;
    constexpr float LumInvR = 0.7154f / 0.2125f, LumInvB = 0.7154f / 0.0721f;

    uint32_t uSteps = 4;
    if (bColorKey)
    {
        uint32_t uColorKey = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (s.a(i) < threshold) uColorKey++;

        if (uColorKey == 16)
            return make_uint2(0xffff0000u, 0xffffffffu);   // rgb[0] = 0x0000, rgb[1] = 0xffff

        uSteps = (uColorKey > 0) ? 3u : 4u;
    }

    float er[DITHER ? 16 : 1], eg[DITHER ? 16 : 1], eb[DITHER ? 16 : 1];
    float Ar, Ag, Ab, Br, Bg, Bb;
    if constexpr (!DITHER && TS::kPacked && CODES && DXTEX_BC15_CODES)
    {
        // Quantise to the 5:6:5 grid; the codes stay packed and the fit decodes them where it reads them (CodeColors)
        CodeColors cc;
        cc.wr = uniform ? 1.0f : LumR; cc.wb = uniform ? 1.0f : LumB;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            cc.q[i] = uint32_t(int32_t(s.r(i) * 31.0f + 0.5f)) | (uint32_t(int32_t(s.g(i) * 63.0f + 0.5f)) << 8) | (uint32_t(int32_t(s.b(i) * 31.0f + 0.5f)) << 16);
        optimize_rgb16(cc, uSteps, uniform, Ar, Ag, Ab, Br, Bg, Bb);
    }
    else
    {
    // Quantise to the 565 grid (optionally error-diffused), then weight by luminance.
    float cr[16], cg[16], cb[16];
    if constexpr (DITHER)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { er[i] = 0.0f; eg[i] = 0.0f; eb[i] = 0.0f; }
    }

#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        float r = s.r(i), g = s.g(i), b = s.b(i);
        if constexpr (DITHER) { r += er[i]; g += eg[i]; b += eb[i]; }

        cr[i] = float(int32_t(r * 31.0f + 0.5f)) * (1.0f / 31.0f);
        cg[i] = float(int32_t(g * 63.0f + 0.5f)) * (1.0f / 63.0f);
        cb[i] = float(int32_t(b * 31.0f + 0.5f)) * (1.0f / 31.0f);

        if constexpr (DITHER)
        {
            // Color[i].a == 1.0f in the default build, so Diff = 1.0f * (Clr - Color)
            diffuse(er, i, 1.0f * (r - cr[i]));
            diffuse(eg, i, 1.0f * (g - cg[i]));
            diffuse(eb, i, 1.0f * (b - cb[i]));
        }

        if (!uniform) { cr[i] *= LumR; cg[i] *= 1.0f; cb[i] *= LumB; }
    }

    ArrayColors ac{ cr, cg, cb };
    optimize_rgb16(ac, uSteps, uniform, Ar, Ag, Ab, Br, Bg, Bb);
    }
    s.launder();

    float Cr, Cg, Cb, Dr, Dg, Db;
    if (uniform) { Cr = Ar; Cg = Ag; Cb = Ab; Dr = Br; Dg = Bg; Db = Bb; }
    else
    {
        Cr = Ar * LumInvR; Cg = Ag * 1.0f; Cb = Ab * LumInvB;
        Dr = Br * LumInvR; Dg = Bg * 1.0f; Db = Bb * LumInvB;
    }

    const uint32_t wA = encode565(Cr, Cg, Cb);
    const uint32_t wB = encode565(Dr, Dg, Db);

    if ((uSteps == 4) && (wA == wB))
        return make_uint2(wA | (wB << 16), 0u);

    decode565(wA, Cr, Cg, Cb);
    decode565(wB, Dr, Dg, Db);

    if (uniform) { Ar = Cr; Ag = Cg; Ab = Cb; Br = Dr; Bg = Dg; Bb = Db; }
    else
    {
        Ar = Cr * LumR; Ag = Cg * 1.0f; Ab = Cb * LumB;
        Br = Dr * LumR; Bg = Dg * 1.0f; Bb = Db * LumB;
    }

    // Endpoint order decides 3- vs 4-colour decoding.
    float S0r, S0g, S0b, S1r, S1g, S1b;
    uint32_t rgb0, rgb1;
    if ((3 == uSteps) == (wA <= wB))
    {
        rgb0 = wA; rgb1 = wB;
        S0r = Ar; S0g = Ag; S0b = Ab; S1r = Br; S1g = Bg; S1b = Bb;
    }
    else
    {
        rgb0 = wB; rgb1 = wA;
        S0r = Br; S0g = Bg; S0b = Bb; S1r = Ar; S1g = Ag; S1b = Ab;
    }

    // Step[2], Step[3] (only needed to feed the dither error)
    float S2r, S2g, S2b, S3r = 0.0f, S3g = 0.0f, S3b = 0.0f;
    if (3 == uSteps)
    {
        S2r = S0r + 0.5f * (S1r - S0r); S2g = S0g + 0.5f * (S1g - S0g); S2b = S0b + 0.5f * (S1b - S0b);
    }
    else
    {
        S2r = S0r + (1.0f / 3.0f) * (S1r - S0r); S2g = S0g + (1.0f / 3.0f) * (S1g - S0g); S2b = S0b + (1.0f / 3.0f) * (S1b - S0b);
        S3r = S0r + (2.0f / 3.0f) * (S1r - S0r); S3g = S0g + (2.0f / 3.0f) * (S1g - S0g); S3b = S0b + (2.0f / 3.0f) * (S1b - S0b);
    }

    float dirR = S1r - S0r, dirG = S1g - S0g, dirB = S1b - S0b;
    const float fSteps = float(uSteps - 1);
    const float fScale = (wA != wB) ? (fSteps / (dirR * dirR + dirG * dirG + dirB * dirB)) : 0.0f;
    dirR *= fScale; dirG *= fScale; dirB *= fScale;

    uint32_t dw = 0;
    if constexpr (DITHER)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) { er[i] = 0.0f; eg[i] = 0.0f; eb[i] = 0.0f; }
    }

#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        if ((3 == uSteps) && (s.a(i) < threshold))
        {
            dw = (3u << 30) | (dw >> 2);
        }
        else
        {
            float r, g, b;
            if (uniform) { r = s.r(i); g = s.g(i); b = s.b(i); }
            else { r = s.r(i) * LumR; g = s.g(i) * 1.0f; b = s.b(i) * LumB; }

            if constexpr (DITHER) { r += er[i]; g += eg[i]; b += eb[i]; }

            const float fDot = (r - S0r) * dirR + (g - S0g) * dirG + (b - S0b) * dirB;

            // fDot <= 0 -> 0, fDot >= fSteps -> 1, else pSteps[uint32(fDot + 0.5f)] with pSteps3 = {0,2,1}, pSteps4 = {0,2,3,1}
            // (BC.cpp:640-647): with the clamp, k = 0 and k = uSteps - 1 are exactly the two outer cases
            const uint32_t k = uint32_t(fminf(fmaxf(fDot, 0.0f), fSteps) + 0.5f);
            const uint32_t iStep = (k == 0) ? 0u : (k == uSteps - 1) ? 1u : (k + 1);

            dw = (iStep << 30) | (dw >> 2);

            if constexpr (DITHER)
            {
                const float qr = (iStep == 0) ? S0r : (iStep == 1) ? S1r : (iStep == 2) ? S2r : S3r;
                const float qg = (iStep == 0) ? S0g : (iStep == 1) ? S1g : (iStep == 2) ? S2g : S3g;
                const float qb = (iStep == 0) ? S0b : (iStep == 1) ? S1b : (iStep == 2) ? S2b : S3b;
                diffuse(er, i, 1.0f * (r - qr));
                diffuse(eg, i, 1.0f * (g - qg));
                diffuse(eb, i, 1.0f * (b - qb));
            }
        }
    }

    return make_uint2(rgb0 | (rgb1 << 16), dw);
}

// x / n for small non-negative integers x <= n, n = 5 or 7, rn = float(1 / n): equal to the IEEE quotient (checked for every operand
// the fits use), at three operations instead of the dozen of a full division.
__device__ __forceinline__ float small_quotient(float x, float n, float rn)
{
    const float q = x * rn;
    return fmaf(fmaf(-q, n, x), rn, q);
}

// 1-D Newton endpoint fit for BC3 alpha / BC4 / BC5 (BC.h:187-311).
template<bool bRange>
__device__ __forceinline__ void optimize_alpha(float& outX, float& outY, const float (&p)[16], uint32_t cSteps)
{
    constexpr float MAX_VALUE = 1.0f;
    constexpr float MIN_VALUE = bRange ? -1.0f : 0.0f;

    float fX = MAX_VALUE, fY = MIN_VALUE;

    if (8 == cSteps)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            if (p[i] < fX) fX = p[i];
            if (p[i] > fY) fY = p[i];
        }
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            if (p[i] < fX && p[i] > MIN_VALUE) fX = p[i];
            if (p[i] > fY && p[i] < MAX_VALUE) fY = p[i];
        }
        if (fX == fY) fY = MAX_VALUE;
    }

    const float fSteps = float(cSteps - 1);
    const bool six = (6 == cSteps);
    const float rSteps = six ? (1.0f / 5.0f) : (1.0f / 7.0f);

    for (int iter = 0; iter < 8; ++iter)
    {
        if ((fY - fX) < (1.0f / 256.0f))
            break;

        const float fScale = fSteps / (fY - fX);

        float dX = 0.0f, dY = 0.0f, d2X = 0.0f, d2Y = 0.0f;

#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            const float fDot = (p[i] - fX) * fScale;

            // BC.h:255-283, without control flow: below / above the end points the texel takes step 0 / cSteps - 1 - or, in the 6-step
            // codec, one of the two fixed values (steps 6 / 7), which take no part in the fit; in between uint32(fDot + 0.5f). The clamp
            // maps the outer cases onto the same conversion.
            const bool low = fDot <= 0.0f, high = fDot >= fSteps;
            const float k = float(uint32_t(fminf(fmaxf(fDot, 0.0f), fSteps) + 0.5f));
            const bool fixedValue = six && ((low && (p[i] <= (fX + MIN_VALUE) * 0.5f)) || (high && (p[i] >= (fY + MAX_VALUE) * 0.5f)));

            // pC6[i] = (5-i)/5, pD6[i] = i/5; pC8[i] = (7-i)/7, pD8[i] = i/7: the correctly rounded quotients, by one Newton step on the
            // product with the rounded reciprocal (exact for these operands; a plain multiplication is not: 3/7, 6/7)
            const float c = small_quotient(fSteps - k, fSteps, rSteps), d = small_quotient(k, fSteps, rSteps);
            const float fDiff = (c * fX + d * fY) - p[i];

            dX = fixedValue ? dX : dX + c * fDiff;
            d2X = fixedValue ? d2X : d2X + c * c;
            dY = fixedValue ? dY : dY + d * fDiff;
            d2Y = fixedValue ? d2Y : d2Y + d * d;
        }

        if (d2X > 0.0f) fX -= dX / d2X;
        if (d2Y > 0.0f) fY -= dY / d2Y;

        if (fX > fY) { const float f = fX; fX = fY; fY = f; }

        if ((dX * dX < (1.0f / 64.0f)) && (dY * dY < (1.0f / 64.0f)))
            break;
    }

    outX = (fX < MIN_VALUE) ? MIN_VALUE : (fX > MAX_VALUE) ? MAX_VALUE : fX;
    outY = (fY < MIN_VALUE) ? MIN_VALUE : (fY > MAX_VALUE) ? MAX_VALUE : fY;
}

// BC3 alpha block (BC.cpp:957-1140). Returns the 8 bytes {alpha0, alpha1, 48 index bits}.
__device__ __forceinline__ uint2 encode_bc3_alpha(const float (&pa)[16], uint32_t flags)
{
    const bool dither = (flags & BCF_DITHER_A) != 0;

    float fAlpha[16], fError[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fError[i] = 0.0f;

    float fMinAlpha = pa[0], fMaxAlpha = pa[0];
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        float fAlph = pa[i];
        if (dither) fAlph += fError[i];

        fAlpha[i] = float(int32_t(fAlph * 255.0f + 0.5f)) * (1.0f / 255.0f);

        if (fAlpha[i] < fMinAlpha) fMinAlpha = fAlpha[i];
        else if (fAlpha[i] > fMaxAlpha) fMaxAlpha = fAlpha[i];

        if (dither) diffuse(fError, i, fAlph - fAlpha[i]);
    }

    if (1.0f == fMinAlpha)
        return make_uint2(0x0000ffffu, 0u);

    const uint32_t uSteps = ((0.0f == fMinAlpha) || (1.0f == fMaxAlpha)) ? 6u : 8u;

    float fAlphaA, fAlphaB;
    optimize_alpha<false>(fAlphaA, fAlphaB, fAlpha, uSteps);

    const uint32_t bAlphaA = uint32_t(int32_t(fAlphaA * 255.0f + 0.5f)) & 0xFF;
    const uint32_t bAlphaB = uint32_t(int32_t(fAlphaB * 255.0f + 0.5f)) & 0xFF;

    fAlphaA = float(bAlphaA) * (1.0f / 255.0f);
    fAlphaB = float(bAlphaB) * (1.0f / 255.0f);

    if ((8 == uSteps) && (bAlphaA == bAlphaB))
        return make_uint2(bAlphaA | (bAlphaB << 8), 0u);

    float fStep[8];
    uint32_t a0, a1;
    if (6 == uSteps)
    {
        a0 = bAlphaA; a1 = bAlphaB;
        fStep[0] = fAlphaA; fStep[1] = fAlphaB;
#pragma unroll
        for (int i = 1; i < 5; ++i)
            fStep[i + 1] = (fStep[0] * float(5 - i) + fStep[1] * float(i)) * (1.0f / 5.0f);
        fStep[6] = 0.0f; fStep[7] = 1.0f;
    }
    else
    {
        a0 = bAlphaB; a1 = bAlphaA;
        fStep[0] = fAlphaB; fStep[1] = fAlphaA;
#pragma unroll
        for (int i = 1; i < 7; ++i)
            fStep[i + 1] = (fStep[0] * float(7 - i) + fStep[1] * float(i)) * (1.0f / 7.0f);
    }

    const float fSteps = float(uSteps - 1);
    const float fScale = (fStep[0] != fStep[1]) ? (fSteps / (fStep[1] - fStep[0])) : 0.0f;

    if (dither)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) fError[i] = 0.0f;
    }

    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        float fAlph = pa[i];
        if (dither) fAlph += fError[i];
        const float fDot = (fAlph - fStep[0]) * fScale;

        uint32_t iStep;
        if (fDot <= 0.0f)
            iStep = ((6 == uSteps) && (fAlph <= fStep[0] * 0.5f)) ? 6u : 0u;
        else if (fDot >= fSteps)
            iStep = ((6 == uSteps) && (fAlph >= (fStep[1] + 1.0f) * 0.5f)) ? 7u : 1u;
        else
        {
            // pSteps6 = {0,2,3,4,5,1}; pSteps8 = {0,2,3,4,5,6,7,1}
            const uint32_t k = uint32_t(fDot + 0.5f);
            iStep = (k == 0) ? 0u : (k == uSteps - 1) ? 1u : (k + 1);
        }

        bits |= uint64_t(iStep) << (3 * i);

        if (dither)
        {
            float q = fStep[0];
#pragma unroll
            for (int s = 1; s < 8; ++s) q = (iStep == uint32_t(s)) ? fStep[s] : q;
            diffuse(fError, i, fAlph - q);
        }
    }

    const uint64_t blk = uint64_t(a0) | (uint64_t(a1) << 8) | (bits << 16);
    return make_uint2(uint32_t(blk), uint32_t(blk >> 32));
}

// BC2 explicit 4-bit alpha (BC.cpp:841-883).
__device__ __forceinline__ uint2 encode_bc2_alpha(const float (&pa)[16], uint32_t flags)
{
    const bool dither = (flags & BCF_DITHER_A) != 0;
    float fError[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fError[i] = 0.0f;

    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        float fAlph = pa[i];
        if (dither) fAlph += fError[i];
        const uint32_t u = uint32_t(fAlph * 15.0f + 0.5f);
        bits |= uint64_t(u & 0xF) << (4 * i);
        if (dither) diffuse(fError, i, fAlph - float(u) * (1.0f / 15.0f));
    }
    return make_uint2(uint32_t(bits), uint32_t(bits >> 32));
}

// BC4 channel block. SNORM selects BC4S/BC5S (BC4BC5.cpp:183-293, :325-377).
template<bool SNORM>
__device__ __forceinline__ uint2 encode_bc4_channel(const float (&t)[16])
{
    constexpr float MIN_NORM = SNORM ? -1.0f : 0.0f;
    constexpr float MAX_NORM = 1.0f;

    float fBlockMax = t[0], fBlockMin = t[0];
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        if (t[i] < fBlockMin) fBlockMin = t[i];
        else if (t[i] > fBlockMax) fBlockMax = t[i];
    }

    const bool bUsing4BlockCodec = (MIN_NORM == fBlockMin || MAX_NORM == fBlockMax);

    float fStart, fEnd;
    optimize_alpha<SNORM>(fStart, fEnd, t, bUsing4BlockCodec ? 6u : 8u);

    int32_t iStart, iEnd;
    if (SNORM)
    {
        // FloatToSNorm (BC4BC5.cpp:158-179); inputs are already clamped to [-1, 1] by optimize_alpha
        float a = fStart, b = fEnd;
        if (a != a) a = 0.0f; else if (a > 1.0f) a = 1.0f; else if (a < -1.0f) a = -1.0f;
        if (b != b) b = 0.0f; else if (b > 1.0f) b = 1.0f; else if (b < -1.0f) b = -1.0f;
        a = a * 127.0f; b = b * 127.0f;
        a = (a >= 0.0f) ? (a + 0.5f) : (a - 0.5f);
        b = (b >= 0.0f) ? (b + 0.5f) : (b - 0.5f);
        iStart = int32_t(a); iEnd = int32_t(b);
    }
    else
    {
        iStart = int32_t(fStart * 255.0f) & 0xFF;   // truncating, BC4BC5.cpp:219-220
        iEnd = int32_t(fEnd * 255.0f) & 0xFF;
    }

    int32_t e0, e1;
    if (!bUsing4BlockCodec) { e0 = iEnd; e1 = iStart; }
    else { e0 = iStart; e1 = iEnd; }

    // Decode the 8 representable values exactly as DecodeFromIndex does (BC4BC5.cpp:47-69 / :103-128).
    float grad[8];
    float f0, f1;
    if (SNORM)
    {
        const int32_t s0 = (e0 == -128) ? -127 : e0;
        const int32_t s1 = (e1 == -128) ? -127 : e1;
        f0 = float(s0) / 127.0f; f1 = float(s1) / 127.0f;
    }
    else
    {
        f0 = float(e0) / 255.0f; f1 = float(e1) / 255.0f;
    }
    grad[0] = f0; grad[1] = f1;
    if (e0 > e1)
    {
#pragma unroll
        for (int i = 1; i < 7; ++i)
            grad[i + 1] = (f0 * float(7 - i) + f1 * float(i)) / 7.0f;
    }
    else
    {
#pragma unroll
        for (int i = 1; i < 5; ++i)
            grad[i + 1] = (f0 * float(5 - i) + f1 * float(i)) / 5.0f;
        grad[6] = MIN_NORM;
        grad[7] = 1.0f;
    }

    uint64_t bits = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        uint32_t uBest = 0;
        float fBestDelta = 100000.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const float fCur = fabsf(grad[k] - t[i]);
            if (fCur < fBestDelta) { uBest = uint32_t(k); fBestDelta = fCur; }
        }
        bits |= uint64_t(uBest) << (3 * i);
    }

    const uint64_t blk = uint64_t(uint32_t(e0) & 0xFF) | (uint64_t(uint32_t(e1) & 0xFF) << 8) | (bits << 16);
    return make_uint2(uint32_t(blk), uint32_t(blk >> 32));
}

// BC1 entry: optional alpha error diffusion to {0,1} before colour keying (BC.cpp:744-784).
__device__ __forceinline__ void bc1_dither_alpha(float (&pa)[16])
{
    float fError[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fError[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        const float fAlph = pa[i] + fError[i];
        const float q = float(int32_t(pa[i] + fError[i] + 0.5f));
        pa[i] = q;
        diffuse(fError, i, fAlph - q);
    }
}

struct EncodeArgs
{
    SrcView src;
    uint8_t* dst;
    uint64_t dstRowPitch;   // bytes per row of blocks
    uint32_t nbw, nbh;      // blocks per row / column
    int dstFormat;
    uint32_t flags;
    float threshold;
};



// KIND: 1..3 = BC1..BC3, 4/5 = BC4/BC5 unsigned, 6/7 = BC4/BC5 signed. One instantiation per format keeps
// each kernel's register footprint to what that codec needs.
// One block from a tile of floats (any source format, partial blocks, conversions): block `nb` of the image behind `a`
template<int KIND, bool DITHER>
__device__ __forceinline__ void encode_block_generic(const EncodeArgs& a, uint32_t bx, uint32_t by)
{
    uint8_t* out = a.dst + uint64_t(by) * a.dstRowPitch;
    Tile t;
    load_tile(a.src, bx, by, t);

    if constexpr (KIND == 1)
    {
        if (a.flags & BCF_DITHER_A) bc1_dither_alpha(t.a);
        reinterpret_cast<uint2*>(out)[bx] = [&] { TileView tv{ t, t.a }; return encode_bc1_color<DITHER>(tv, true, a.threshold, a.flags); }();
    }
    else if constexpr (KIND == 2)
    {
        const uint2 al = encode_bc2_alpha(t.a, a.flags);
        const uint2 c = [&] { TileView tv{ t, t.a }; return encode_bc1_color<DITHER>(tv, false, 0.0f, a.flags); }();
        reinterpret_cast<uint4*>(out)[bx] = make_uint4(al.x, al.y, c.x, c.y);
    }
    else if constexpr (KIND == 3)
    {
        const uint2 al = encode_bc3_alpha(t.a, a.flags);
        const uint2 c = [&] { TileView tv{ t, t.a }; return encode_bc1_color<DITHER>(tv, false, 0.0f, a.flags); }();
        reinterpret_cast<uint4*>(out)[bx] = make_uint4(al.x, al.y, c.x, c.y);
    }
    else if constexpr (KIND == 4)
        reinterpret_cast<uint2*>(out)[bx] = encode_bc4_channel<false>(t.r);
    else if constexpr (KIND == 6)
        reinterpret_cast<uint2*>(out)[bx] = encode_bc4_channel<true>(t.r);
    else if constexpr (KIND == 5)
    {
        const uint2 u = encode_bc4_channel<false>(t.r), v = encode_bc4_channel<false>(t.g);
        reinterpret_cast<uint4*>(out)[bx] = make_uint4(u.x, u.y, v.x, v.y);
    }
    else
    {
        const uint2 u = encode_bc4_channel<true>(t.r), v = encode_bc4_channel<true>(t.g);
        reinterpret_cast<uint4*>(out)[bx] = make_uint4(u.x, u.y, v.x, v.y);
    }
}

// The packed-tile path (RGBA8 source, whole blocks): load and encode of one block, shared by the one-tile-per-wavefront kernel and the
// streaming kernel below.
__device__ __forceinline__ void load_packed_tile(const EncodeArgs& a, uint32_t bx, uint32_t by, PackedTile& t)
{
#pragma unroll
    for (uint32_t y = 0; y < 4; ++y)
    {
        const uint4 v = *reinterpret_cast<const uint4*>(a.src.pixels + uint64_t(by * 4 + y) * a.src.rowPitch + uint64_t(bx) * 16);
        t.px[y * 4 + 0] = v.x; t.px[y * 4 + 1] = v.y; t.px[y * 4 + 2] = v.z; t.px[y * 4 + 3] = v.w;
    }
}

template<int KIND, bool DITHER>
__device__ __forceinline__ void encode_packed(PackedTile& t, uint8_t* out, uint32_t bx, const EncodeArgs& a)
{
    if constexpr (KIND == 1)
        reinterpret_cast<uint2*>(out)[bx] = encode_bc1_color<DITHER>(t, true, a.threshold, a.flags);
    else if constexpr (KIND == 2 || KIND == 3)
    {
        uint2 al;
        {
            float pa[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pa[i] = t.a(i);
            al = (KIND == 2) ? encode_bc2_alpha(pa, a.flags) : encode_bc3_alpha(pa, a.flags);
        }
        // BC3: the fit on packed codes (four waves per SIMD, no spill); BC2 keeps the float array at two (measured, see bc15_encode_kernel)
        const uint2 c = encode_bc1_color<DITHER, PackedTile, KIND == 3>(t, false, 0.0f, a.flags);
        reinterpret_cast<uint4*>(out)[bx] = make_uint4(al.x, al.y, c.x, c.y);
    }
    else
    {
        float ch[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ch[i] = t.r(i);
        const uint2 u = encode_bc4_channel<false>(ch);
        if constexpr (KIND == 4) reinterpret_cast<uint2*>(out)[bx] = u;
        else
        {
#pragma unroll
            for (int i = 0; i < 16; ++i) ch[i] = t.g(i);
            const uint2 v = encode_bc4_channel<false>(ch);
            reinterpret_cast<uint4*>(out)[bx] = make_uint4(u.x, u.y, v.x, v.y);
        }
    }
}

#if !defined(DXTEX_BC15_COLOR_WGS)
#define DXTEX_BC15_COLOR_WGS 4         // workgroups per CU of the BC1 - BC3 instantiations whose fit reads packed 5:6:5 codes (CodeColors)
#endif
#if !defined(DXTEX_BC15_PACKED_WGS)
#define DXTEX_BC15_PACKED_WGS 3        // workgroups per CU the other packed-tile instantiations of BC3 (dithered / fit on floats) are compiled for
#endif
#if !defined(DXTEX_BC15_BC45_WGS)
#define DXTEX_BC15_BC45_WGS 4          // BC4 / BC5: no colour fit, 120 registers are enough (at 3 the allocator took 168 and BC5 of a 4096^2 image went 0.077 -> 0.092 ms)
#endif
// Occupancy per codec, measured (round 4, 4096^2 cfg2 image / BC3 of the 8192^2 cfg4 chain with random alpha, same box): BC1 and BC2 with the
// fit on 48 floats at 2 workgroups per CU (no spill; BC2 0.128 ms against 0.144 at 3) - on packed codes at 3 - 4 workgroups they are slower
// (0.146 against 0.135 / 0.128: decoding costs 8 operations per texel and trip, a quarter more VALU work, and these kernels are VALU-bound at
// any occupancy). BC3, whose alpha fit shares the registers, spilled 128 bytes per lane at 3 workgroups and was slower still at 2; on codes
// at 4 workgroups it has no scratch: 0.163 -> 0.147 ms per 4096^2 image, the cfg4 chain 1.16 -> 0.93 ms.
template<int KIND, bool DITHER, bool PACKED8>
__global__ void __launch_bounds__(256, PACKED8 ? ((KIND == 3 && !DITHER && DXTEX_BC15_CODES) ? DXTEX_BC15_COLOR_WGS : (KIND <= 2 ? 2 : (KIND >= 4 ? DXTEX_BC15_BC45_WGS : DXTEX_BC15_PACKED_WGS))) : 1) bc15_encode_kernel(EncodeArgs a)
{
    // A wavefront takes an 8 x 8 tile of blocks (32 x 32 texels), not 64 blocks of one block row: what the lanes of a wavefront do
    // differs by content - flat blocks leave the fit at once, noisy ones run its eight Newton trips - and content is coherent in two
    // dimensions, so a square tile keeps more lanes in step than a 256-texel strip (36 -> of 64 lanes active on the benchmark image
    // with strips). Loads stay 128-byte runs (8 blocks x 16 bytes per row), stores 64 / 128-byte runs. Images narrower or lower than
    // 8 blocks keep the linear order.
    uint32_t bx, by;
    if (KIND <= 3 && a.nbw >= 8 && a.nbh >= 8)          // BC4 / BC5 have no content-dependent paths: linear order, 1 KiB runs
    {
        const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
        const uint32_t tilesX = (a.nbw + 7u) >> 3;
        const uint32_t ty = wave / tilesX, tx = wave - ty * tilesX;
        bx = tx * 8u + (lane & 7u); by = ty * 8u + (lane >> 3);
        if (bx >= a.nbw || by >= a.nbh) return;
    }
    else
    {
        const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
        if (nb >= a.nbw * a.nbh) return;
        by = nb / a.nbw; bx = nb - by * a.nbw;
    }
    uint8_t* out = a.dst + uint64_t(by) * a.dstRowPitch;

    if constexpr (PACKED8)
    {
        // RGBA8 source, whole blocks, no tile conversion (the launcher checks): four coalesced 16-byte row loads, texels stay packed
        PackedTile t;
        load_packed_tile(a, bx, by, t);
        encode_packed<KIND, DITHER>(t, out, bx, a);
        return;
    }
    else
        encode_block_generic<KIND, DITHER>(a, bx, by);
}

// Several SMALL images in one launch (the tail of a mip chain, a set of icons): a kernel over one of them lasts as long as its slowest
// block - about 15 us, whatever the size below 256^2 - so a launch each is a serial chain of latencies; together they are one.
constexpr int kMultiMax = 12;
constexpr uint32_t kMultiMaxBlocks = 4096;           // images of at most 256 x 256 texels
struct MultiImage { SrcView src; uint8_t* dst; uint64_t dstRowPitch; uint32_t nbw, nbh, first, pad; };
struct MultiArgs { MultiImage img[kMultiMax]; uint32_t n, total; int dstFormat; uint32_t flags; float threshold; };
template<int KIND, bool DITHER>
__global__ void __launch_bounds__(256) bc15_encode_multi_kernel(MultiArgs m)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= m.total) return;
    uint32_t i = 0;
    for (uint32_t k = 1; k < m.n; ++k) if (m.img[k].first <= g) i = k;
    EncodeArgs a;
    a.src = m.img[i].src; a.dst = m.img[i].dst; a.dstRowPitch = m.img[i].dstRowPitch; a.nbw = m.img[i].nbw; a.nbh = m.img[i].nbh;
    a.dstFormat = m.dstFormat; a.flags = m.flags; a.threshold = m.threshold;
    const uint32_t nb = g - m.img[i].first, by = nb / a.nbw;
    encode_block_generic<KIND, DITHER>(a, nb - by * a.nbw, by);
}
} // namespace

hipError_t launch_bc15_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                              uint32_t flags, float threshold, hipStream_t stream)
{
    EncodeArgs a;
    a.src = src; a.dst = dst; a.dstRowPitch = dstRowPitch;
    a.nbw = (src.width + 3) / 4; a.nbh = (src.height + 3) / 4;
    a.dstFormat = dstFormat; a.flags = flags; a.threshold = threshold;
    const uint64_t nblocks = uint64_t(a.nbw) * a.nbh;
    if (!nblocks) return hipSuccess;
    // bc15_encode_kernel: a wavefront per 8 x 8 tile of blocks when the image has at least 8 x 8 of them, else blocks in linear order
    const uint64_t tiles = uint64_t((a.nbw + 7) / 8) * ((a.nbh + 7) / 8);
    const bool colourFit = dstFormat == FMT_BC1_UNORM || dstFormat == FMT_BC1_UNORM_SRGB || dstFormat == FMT_BC2_UNORM || dstFormat == FMT_BC2_UNORM_SRGB ||
                           dstFormat == FMT_BC3_UNORM || dstFormat == FMT_BC3_UNORM_SRGB;         // kernel KIND <= 3
    const dim3 grid((colourFit && a.nbw >= 8 && a.nbh >= 8) ? uint32_t((tiles + 3) / 4) : uint32_t((nblocks + 255) / 256)), block(256);
    const bool dither = (flags & BCF_DITHER_RGB) != 0;
    // packed-tile kernels: RGBA8 texels need no conversion on their way into the encoder, every block is whole, rows are 16-byte aligned;
    // BC1's alpha dithering rewrites the tile's alpha and keeps the float tile
    const bool packed = src.format == FMT_R8G8B8A8_UNORM && src.tcv == TCV_NONE && src.tsw == TSW_NONE && (src.width % 4) == 0 && (src.height % 4) == 0 &&
                        (src.rowPitch % 16) == 0 && (reinterpret_cast<uintptr_t>(src.pixels) % 16) == 0;
#define DXTEX_LAUNCH2(KIND, DITHER, PACKED) hipLaunchKernelGGL((bc15_encode_kernel<KIND, DITHER, PACKED>), grid, block, 0, stream, a)
#define DXTEX_LAUNCH(KIND, PACKOK) do { const bool pk_ = packed && (PACKOK); \
                                         if (dither) { if (pk_) DXTEX_LAUNCH2(KIND, true, true); else DXTEX_LAUNCH2(KIND, true, false); } \
                                         else { if (pk_) DXTEX_LAUNCH2(KIND, false, true); else DXTEX_LAUNCH2(KIND, false, false); } } while (0)
    switch (dstFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: DXTEX_LAUNCH(1, (flags & BCF_DITHER_A) == 0); break;
    case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB: DXTEX_LAUNCH(2, true); break;
    case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB: DXTEX_LAUNCH(3, true); break;
    case FMT_BC4_UNORM: if (packed) DXTEX_LAUNCH2(4, false, true); else DXTEX_LAUNCH2(4, false, false); break;
    case FMT_BC5_UNORM: if (packed) DXTEX_LAUNCH2(5, false, true); else DXTEX_LAUNCH2(5, false, false); break;
    case FMT_BC4_SNORM: DXTEX_LAUNCH2(6, false, false); break;
    case FMT_BC5_SNORM: DXTEX_LAUNCH2(7, false, false); break;
    default: return hipErrorInvalidValue;
    }
#undef DXTEX_LAUNCH2
#undef DXTEX_LAUNCH
    return hipGetLastError();
}
bool bc15_small_image(uint32_t width, uint32_t height) { return uint64_t((width + 3) / 4) * ((height + 3) / 4) <= kMultiMaxBlocks; }
int bc15_small_batch_max() { return kMultiMax; }

hipError_t launch_bc15_encode_small(const BcImage* images, int count, int dstFormat, uint32_t flags, float threshold, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    if (count > kMultiMax) return hipErrorInvalidValue;
    MultiArgs m;
    uint32_t total = 0;
    for (int i = 0; i < count; ++i)
    {
        MultiImage& g = m.img[i];
        g.src = images[i].src; g.dst = images[i].dst; g.dstRowPitch = images[i].dstRowPitch;
        g.nbw = (images[i].src.width + 3) / 4; g.nbh = (images[i].src.height + 3) / 4; g.first = total; g.pad = 0;
        total += g.nbw * g.nbh;
    }
    for (int i = count; i < kMultiMax; ++i) m.img[i] = m.img[0];
    m.n = uint32_t(count); m.total = total; m.dstFormat = dstFormat; m.flags = flags; m.threshold = threshold;
    if (!total) return hipSuccess;
    const dim3 grid((total + 255) / 256), block(256);
    const bool dither = (flags & BCF_DITHER_RGB) != 0;
#define DXTEX_MULTI(KIND) do { if (dither) hipLaunchKernelGGL((bc15_encode_multi_kernel<KIND, true>), grid, block, 0, stream, m); \
                               else hipLaunchKernelGGL((bc15_encode_multi_kernel<KIND, false>), grid, block, 0, stream, m); } while (0)
    switch (dstFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: DXTEX_MULTI(1); break;
    case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB: DXTEX_MULTI(2); break;
    case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB: DXTEX_MULTI(3); break;
    case FMT_BC4_UNORM: hipLaunchKernelGGL((bc15_encode_multi_kernel<4, false>), grid, block, 0, stream, m); break;
    case FMT_BC5_UNORM: hipLaunchKernelGGL((bc15_encode_multi_kernel<5, false>), grid, block, 0, stream, m); break;
    case FMT_BC4_SNORM: hipLaunchKernelGGL((bc15_encode_multi_kernel<6, false>), grid, block, 0, stream, m); break;
    case FMT_BC5_SNORM: hipLaunchKernelGGL((bc15_encode_multi_kernel<7, false>), grid, block, 0, stream, m); break;
    default: return hipErrorInvalidValue;
    }
#undef DXTEX_MULTI
    return hipGetLastError();
}
} // namespace dxtex
