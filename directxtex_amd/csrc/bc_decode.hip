// BC1-BC7 decoders for gfx950: DecompressBC (DirectXTexCompress.cpp:425-535) with the block decoders
// D3DXDecodeBC1/2/3 (BC.cpp:318-364, :802-826, :902-941), D3DXDecodeBC4U/4S/5U/5S (BC4BC5.cpp:36-151, :389-494),
// D3DX_BC6H::Decode (BC6HBC7.cpp:1658-1813) and D3DX_BC7::Decode (:2566-2780).
//
// One lane decodes one 4x4 block to 16 fp32 texels (what the reference's BC_DECODE hooks produce), runs the
// ConvertScanline plan and stores the texels that fall inside the image with StoreScanline semantics. Lanes of a
// wavefront own consecutive blocks of a block row, so every one of the four row stores of a wave is contiguous.
// HBM-bound: 0.5 or 1 byte read + bytes-per-texel of the target written per texel.
#include "dxtex_kernels.h"
#include "dxtex_store.h"
#include "bc67_tables.h"
#include "bc6h_core.h"

namespace dxtex
{
namespace
{
struct DecodeArgs
{
    const uint8_t* src; uint64_t srcRowPitch; int srcFormat;
    uint8_t* dst; uint64_t dstRowPitch; int dstFormat;
    uint32_t width, height, nbw, nbh;
    ConvertPlan plan;
};

// XMVectorLerp(a, b, t) = a + (b - a) * t, unfused (DirectXMath, SSE2 shape)
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + (b - a) * t; }

__device__ __forceinline__ void decode_bc1(const uint8_t* p, bool isbc1, Texel (&out)[16])
{
    const uint32_t c01 = *reinterpret_cast<const uint32_t*>(p);
    const uint32_t bitmap = *reinterpret_cast<const uint32_t*>(p + 4);
    const uint32_t w0 = c01 & 0xFFFF, w1 = c01 >> 16;
    // XMLoadU565 -> (x = bits 0-4, y = bits 5-10, z = bits 11-15) * (1/31, 1/63, 1/31), swizzled to (z, y, x)
    Texel clr[4];
    const uint32_t ws[2] = { w0, w1 };
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        clr[i].r = float((ws[i] >> 11) & 31) * (1.0f / 31.0f);
        clr[i].g = float((ws[i] >> 5) & 63) * (1.0f / 63.0f);
        clr[i].b = float(ws[i] & 31) * (1.0f / 31.0f);
        clr[i].a = 1.0f;
    }
    if (isbc1 && w0 <= w1)
    {
        clr[2].r = lerp1(clr[0].r, clr[1].r, 0.5f); clr[2].g = lerp1(clr[0].g, clr[1].g, 0.5f);
        clr[2].b = lerp1(clr[0].b, clr[1].b, 0.5f); clr[2].a = lerp1(clr[0].a, clr[1].a, 0.5f);
        clr[3].r = clr[3].g = clr[3].b = clr[3].a = 0.0f;
    }
    else
    {
        clr[2].r = lerp1(clr[0].r, clr[1].r, 1.0f / 3.0f); clr[2].g = lerp1(clr[0].g, clr[1].g, 1.0f / 3.0f);
        clr[2].b = lerp1(clr[0].b, clr[1].b, 1.0f / 3.0f); clr[2].a = lerp1(clr[0].a, clr[1].a, 1.0f / 3.0f);
        clr[3].r = lerp1(clr[0].r, clr[1].r, 2.0f / 3.0f); clr[3].g = lerp1(clr[0].g, clr[1].g, 2.0f / 3.0f);
        clr[3].b = lerp1(clr[0].b, clr[1].b, 2.0f / 3.0f); clr[3].a = lerp1(clr[0].a, clr[1].a, 2.0f / 3.0f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        const uint32_t s = (bitmap >> (2 * i)) & 3u;
        out[i].r = (s == 0) ? clr[0].r : (s == 1) ? clr[1].r : (s == 2) ? clr[2].r : clr[3].r;
        out[i].g = (s == 0) ? clr[0].g : (s == 1) ? clr[1].g : (s == 2) ? clr[2].g : clr[3].g;
        out[i].b = (s == 0) ? clr[0].b : (s == 1) ? clr[1].b : (s == 2) ? clr[2].b : clr[3].b;
        out[i].a = (s == 0) ? clr[0].a : (s == 1) ? clr[1].a : (s == 2) ? clr[2].a : clr[3].a;
    }
}

__device__ __forceinline__ void decode_bc3_alpha(const uint8_t* p, Texel (&out)[16])
{
    const uint64_t d = *reinterpret_cast<const uint64_t*>(p);
    const uint32_t a0 = uint32_t(d & 0xFF), a1 = uint32_t((d >> 8) & 0xFF);
    float f[8];
    f[0] = float(a0) * (1.0f / 255.0f);
    f[1] = float(a1) * (1.0f / 255.0f);
    if (a0 > a1)
    {
#pragma unroll
        for (int i = 1; i < 7; ++i) f[i + 1] = (f[0] * float(7 - i) + f[1] * float(i)) * (1.0f / 7.0f);
    }
    else
    {
#pragma unroll
        for (int i = 1; i < 5; ++i) f[i + 1] = (f[0] * float(5 - i) + f[1] * float(i)) * (1.0f / 5.0f);
        f[6] = 0.0f; f[7] = 1.0f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        const uint32_t s = uint32_t(d >> (16 + 3 * i)) & 7u;
        float v = f[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) v = (s == uint32_t(k)) ? f[k] : v;
        out[i].a = v;
    }
}

// BC4_UNORM::R / BC4_SNORM::R (BC4BC5.cpp:36-151): true divisions, unlike BC3's alpha
__device__ __forceinline__ float bc4_value(uint64_t d, int i, bool isSigned)
{
    const uint32_t idx = uint32_t(d >> (16 + 3 * i)) & 7u;
    float f0, f1; bool gt;
    if (isSigned)
    {
        int r0 = int(int8_t(d & 0xFF)), r1 = int(int8_t((d >> 8) & 0xFF));
        gt = r0 > r1;
        if (r0 == -128) r0 = -127;
        if (r1 == -128) r1 = -127;
        f0 = float(r0) / 127.0f; f1 = float(r1) / 127.0f;
    }
    else
    {
        const uint32_t r0 = uint32_t(d & 0xFF), r1 = uint32_t((d >> 8) & 0xFF);
        gt = r0 > r1;
        f0 = float(r0) / 255.0f; f1 = float(r1) / 255.0f;
    }
    if (idx == 0) return f0;
    if (idx == 1) return f1;
    if (gt) { const uint32_t k = idx - 1; return (f0 * float(7u - k) + f1 * float(k)) / 7.0f; }
    if (idx == 6) return isSigned ? -1.0f : 0.0f;
    if (idx == 7) return 1.0f;
    const uint32_t k = idx - 1;
    return (f0 * float(5u - k) + f1 * float(k)) / 5.0f;
}

// ---- BC6H / BC7 bit reader -------------------------------------------------------------------------------------------
struct Bits { uint64_t lo, hi; uint32_t pos; };
__device__ __forceinline__ uint32_t get_bits(Bits& b, uint32_t n)
{
    if (n == 0) return 0;
    uint64_t v;
    if (b.pos >= 64) v = b.hi >> (b.pos - 64);
    else { v = b.lo >> b.pos; if (b.pos + n > 64) v |= b.hi << (64 - b.pos); }
    b.pos += n;
    return uint32_t(v & ((uint64_t(1) << n) - 1));
}

__device__ __forceinline__ void fill_error(Texel (&out)[16])
{
    // FillWithErrorColors, release flavour: opaque black (:1638-1650)
#pragma unroll
    for (int i = 0; i < 16; ++i) { out[i].r = out[i].g = out[i].b = 0.0f; out[i].a = 1.0f; }
}

// ms_aInfo (BC6HBC7.cpp:1106-1124): subsets-1, partition bits, p-bits, rotation bits, index-mode bits, index bits,
// second index bits, colour / alpha endpoint bits without and with the p-bit.
struct Bc7ModeRt { uint8_t parts, partBits, pBits, rotBits, imBits, ib, ib2, cp, ap, cpp, app; };
__device__ static const Bc7ModeRt kBc7Modes[8] = {
    { 2, 4, 6, 0, 0, 3, 0, 4, 0, 5, 0 },
    { 1, 6, 2, 0, 0, 3, 0, 6, 0, 7, 0 },
    { 2, 6, 0, 0, 0, 2, 0, 5, 0, 5, 0 },
    { 1, 6, 4, 0, 0, 2, 0, 7, 0, 8, 0 },
    { 0, 0, 0, 2, 1, 2, 3, 5, 6, 5, 6 },
    { 0, 0, 0, 2, 0, 2, 2, 7, 8, 7, 8 },
    { 0, 0, 2, 0, 0, 4, 0, 7, 7, 8, 8 },
    { 1, 6, 4, 0, 0, 2, 0, 5, 5, 6, 6 },
};

__device__ __forceinline__ uint32_t bc67_weight(uint32_t bits, uint32_t i)
{
    // g_aWeights2/3/4 (:327-329)
    const uint32_t w2 = 0x40u << 24 | 43u << 16 | 21u << 8;                 // {0, 21, 43, 64}
    if (bits == 2) return (w2 >> (8 * i)) & 0xFF;
    if (bits == 3) { const uint8_t w3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 }; return w3[i & 7]; }
    const uint8_t w4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
    return w4[i & 15];
}

__device__ __forceinline__ uint32_t region_of(uint32_t parts, uint32_t shape, uint32_t i)
{
    if (parts == 0) return 0;
    if (parts == 1) return (uint32_t(kPart2Mask[shape]) >> i) & 1u;
    return (kPart3Bits[shape] >> (2 * i)) & 3u;
}

__device__ __forceinline__ bool is_anchor(uint32_t parts, uint32_t shape, uint32_t i)
{
    // IsFixUpOffset (:1132-1144)
    if (i == 0) return true;
    if (parts == 1) return i == kAnchor2[shape];
    if (parts == 2) return i == uint32_t(kAnchor3[shape] & 15) || i == uint32_t(kAnchor3[shape] >> 4);
    return false;
}

__device__ __forceinline__ uint32_t bc7_unq(uint32_t c, uint32_t prec)
{
    // D3DX_BC7::Unquantize (:826-831)
    if (prec == 0) return 255u;      // alpha of the colour-only modes is forced to 255 before unquantising
    const uint32_t s = (c << (8 - prec)) & 0xFFu;
    return s | (s >> prec);
}

__device__ __forceinline__ void decode_bc7(const uint8_t* p, Texel (&out)[16])
{
    Bits b; b.lo = reinterpret_cast<const uint64_t*>(p)[0]; b.hi = reinterpret_cast<const uint64_t*>(p)[1]; b.pos = 0;
    const uint32_t low8 = uint32_t(b.lo & 0xFF);
    if (low8 == 0)
    {
        // reserved mode 8 (or no mode bit in the first byte): transparent black (:2771-2778)
#pragma unroll
        for (int i = 0; i < 16; ++i) { out[i].r = out[i].g = out[i].b = out[i].a = 0.0f; }
        return;
    }
    const uint32_t mode = uint32_t(__ffs(int(low8))) - 1u;
    b.pos = mode + 1;
    const Bc7ModeRt mi = kBc7Modes[mode];
    const uint32_t nEnd = (uint32_t(mi.parts) + 1u) << 1;
    const uint32_t shape = get_bits(b, mi.partBits);
    const uint32_t rot = get_bits(b, mi.rotBits);
    const uint32_t im = get_bits(b, mi.imBits);

    uint32_t c[6][4];
    for (uint32_t ch = 0; ch < 4; ++ch)
    {
        const uint32_t prec = (ch == 3) ? mi.ap : mi.cp;
        for (uint32_t i = 0; i < nEnd; ++i) c[i][ch] = (ch == 3 && prec == 0) ? 255u : get_bits(b, prec);
    }
    uint32_t P[6] = { 0, 0, 0, 0, 0, 0 };
    for (uint32_t i = 0; i < mi.pBits; ++i) P[i] = get_bits(b, 1);
    if (mi.pBits)
    {
        for (uint32_t i = 0; i < nEnd; ++i)
        {
            const uint32_t pi = i * mi.pBits / nEnd;
            for (uint32_t ch = 0; ch < 4; ++ch)
            {
                const uint32_t pr = (ch == 3) ? mi.ap : mi.cp, prp = (ch == 3) ? mi.app : mi.cpp;
                if (pr != prp) c[i][ch] = ((c[i][ch] << 1) | P[pi]) & 0xFFu;
            }
        }
    }
    for (uint32_t i = 0; i < nEnd; ++i)
        for (uint32_t ch = 0; ch < 4; ++ch)
            c[i][ch] = bc7_unq(c[i][ch], (ch == 3) ? mi.app : mi.cpp);

    uint32_t w1[16], w2[16];
    for (uint32_t i = 0; i < 16; ++i)
        w1[i] = get_bits(b, is_anchor(mi.parts, shape, i) ? mi.ib - 1u : mi.ib);
    if (mi.ib2)
        for (uint32_t i = 0; i < 16; ++i)
            w2[i] = get_bits(b, i ? mi.ib2 : mi.ib2 - 1u);
    if (b.pos > 128) { fill_error(out); return; }

    for (uint32_t i = 0; i < 16; ++i)
    {
        const uint32_t rg = region_of(mi.parts, shape, i);
        const uint32_t* e0 = c[rg << 1];
        const uint32_t* e1 = c[(rg << 1) + 1];
        uint32_t wc, wa, wcp, wap;
        if (mi.ib2 == 0) { wc = wa = w1[i]; wcp = wap = mi.ib; }
        else if (im == 0) { wc = w1[i]; wa = w2[i]; wcp = mi.ib; wap = mi.ib2; }
        else { wc = w2[i]; wa = w1[i]; wcp = mi.ib2; wap = mi.ib; }
        const uint32_t kc = bc67_weight(wcp, wc), ka = bc67_weight(wap, wa);
        uint32_t px[4];
        for (uint32_t ch = 0; ch < 3; ++ch) px[ch] = (e0[ch] * (64u - kc) + e1[ch] * kc + 32u) >> 6;
        px[3] = (e0[3] * (64u - ka) + e1[3] * ka + 32u) >> 6;
        if (rot == 1) { const uint32_t t = px[0]; px[0] = px[3]; px[3] = t; }
        else if (rot == 2) { const uint32_t t = px[1]; px[1] = px[3]; px[3] = t; }
        else if (rot == 3) { const uint32_t t = px[2]; px[2] = px[3]; px[3] = t; }
        out[i].r = float(px[0]) * (1.0f / 255.0f); out[i].g = float(px[1]) * (1.0f / 255.0f);
        out[i].b = float(px[2]) * (1.0f / 255.0f); out[i].a = float(px[3]) * (1.0f / 255.0f);
    }
}

__device__ __forceinline__ void decode_bc6h(const uint8_t* p, bool isSigned, Texel (&out)[16])
{
    Bits b; b.lo = reinterpret_cast<const uint64_t*>(p)[0]; b.hi = reinterpret_cast<const uint64_t*>(p)[1]; b.pos = 0;
    uint32_t mode = get_bits(b, 2);
    if (mode != 0 && mode != 1) mode = (get_bits(b, 3) << 2) | mode;
    const int mi = kBc6hModeIndex[mode];
    if (mi < 0) { fill_error(out); return; }      // reserved modes decode to opaque black (:1805-1811)
    const Bc6hMode info = kBc6hModes[mi];
    const uint8_t* desc = kBc6hHeader[mi];

    int ep[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };   // A0, B0, A1, B1
    uint32_t shape = 0;
    const uint32_t headerBits = info.regions2 ? 82u : 65u;
    while (b.pos < headerBits)
    {
        const uint32_t cur = b.pos;
        if (get_bits(b, 1))
        {
            const uint32_t f = desc[cur] >> 4, bit = desc[cur] & 15u;
            if (f == 2) shape |= 1u << bit;
            else if (f >= 3) ep[(f - 3) & 3][(f - 3) >> 2] |= 1 << bit;
            else if (f == 0) { fill_error(out); return; }           // a set bit in an unused header position (:1703-1711)
        }
    }
    // sign extension and inverse delta transform
    if (isSigned)
        for (int ch = 0; ch < 3; ++ch) ep[0][ch] = bc6h::sign_extend(ep[0][ch], info.prec[ch]);
    if (isSigned || info.transformed)
    {
        const int nreg = info.regions2 ? 2 : 1;
        for (int r = 0; r < nreg; ++r)
            for (int ch = 0; ch < 3; ++ch)
            {
                if (r != 0) ep[2][ch] = bc6h::sign_extend(ep[2][ch], info.delta[ch]);
                ep[2 * r + 1][ch] = bc6h::sign_extend(ep[2 * r + 1][ch], info.delta[ch]);
            }
    }
    if (info.transformed)
    {
        // TransformInverse (:1153-1165): applied to both regions' slots regardless of the region count
        for (int ch = 0; ch < 3; ++ch)
        {
            const int wrap = (1 << info.prec[ch]) - 1;
            for (int e = 1; e < 4; ++e)
            {
                ep[e][ch] = (ep[e][ch] + ep[0][ch]) & wrap;
                if (isSigned) ep[e][ch] = bc6h::sign_extend(ep[e][ch], info.prec[ch]);
            }
        }
    }

    for (uint32_t i = 0; i < 16; ++i)
    {
        const uint32_t nbits = is_anchor(info.regions2, shape & 31u, i) ? info.indexBits - 1u : info.indexBits;
        if (b.pos + nbits > 128) { fill_error(out); return; }
        const uint32_t idx = get_bits(b, nbits);
        const uint32_t rg = info.regions2 ? region_of(1, shape & 31u, i) : 0u;
        const int w = int(bc67_weight(info.regions2 ? 3 : 4, idx));
        uint32_t h[3];
        for (int ch = 0; ch < 3; ++ch)
        {
            const int a = bc6h::unquantize(ep[2 * rg][ch], info.prec[ch], isSigned);
            const int bq = bc6h::unquantize(ep[2 * rg + 1][ch], info.prec[ch], isSigned);
            const int v = bc6h::finish_unquantize((a * (64 - w) + bq * w + 32) >> 6, isSigned);
            h[ch] = bc6h::int_to_f16(v, isSigned);
        }
        out[i].r = __half2float(__ushort_as_half(uint16_t(h[0])));
        out[i].g = __half2float(__ushort_as_half(uint16_t(h[1])));
        out[i].b = __half2float(__ushort_as_half(uint16_t(h[2])));
        out[i].a = 1.0f;
    }
}

__global__ void __launch_bounds__(256) bc_decode_kernel(DecodeArgs a)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nbw * a.nbh) return;
    const uint32_t by = nb / a.nbw, bx = nb - by * a.nbw;
    const uint32_t bb = (a.srcFormat == FMT_BC1_UNORM || a.srcFormat == FMT_BC1_UNORM_SRGB || a.srcFormat == FMT_BC4_UNORM || a.srcFormat == FMT_BC4_SNORM) ? 8u : 16u;
    const uint8_t* p = a.src + uint64_t(by) * a.srcRowPitch + uint64_t(bx) * bb;

    Texel t[16];
    switch (a.srcFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB:
        decode_bc1(p, true, t);
        break;
    case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB:
    {
        decode_bc1(p + 8, false, t);
        const uint64_t al = *reinterpret_cast<const uint64_t*>(p);
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i].a = float(uint32_t(al >> (4 * i)) & 15u) * (1.0f / 15.0f);
        break;
    }
    case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB:
        decode_bc1(p + 8, false, t);
        decode_bc3_alpha(p, t);
        break;
    case FMT_BC4_UNORM: case FMT_BC4_SNORM:
    {
        const uint64_t d = *reinterpret_cast<const uint64_t*>(p);
        const bool sg = a.srcFormat == FMT_BC4_SNORM;
#pragma unroll
        for (int i = 0; i < 16; ++i) { t[i].r = bc4_value(d, i, sg); t[i].g = 0.0f; t[i].b = 0.0f; t[i].a = 1.0f; }
        break;
    }
    case FMT_BC5_UNORM: case FMT_BC5_SNORM:
    {
        const uint64_t d0 = reinterpret_cast<const uint64_t*>(p)[0], d1 = reinterpret_cast<const uint64_t*>(p)[1];
        const bool sg = a.srcFormat == FMT_BC5_SNORM;
#pragma unroll
        for (int i = 0; i < 16; ++i) { t[i].r = bc4_value(d0, i, sg); t[i].g = bc4_value(d1, i, sg); t[i].b = 0.0f; t[i].a = 1.0f; }
        break;
    }
    case FMT_BC6H_UF16: decode_bc6h(p, false, t); break;
    case FMT_BC6H_SF16: decode_bc6h(p, true, t); break;
    default: decode_bc7(p, t); break;
    }

    const uint32_t x0 = bx * 4, y0 = by * 4;
    const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
#pragma unroll
    for (uint32_t y = 0; y < 4; ++y)
    {
        if (y >= ph) break;
        uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x)
            if (x < pw) store_texel(row, x0 + x, a.dstFormat, apply_plan(t[y * 4 + x], a.plan));
    }
}
} // namespace

hipError_t launch_bc_decode(const uint8_t* src, uint64_t srcRowPitch, int srcFormat, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                            uint32_t width, uint32_t height, const ConvertPlan& plan, hipStream_t stream)
{
    DecodeArgs a;
    a.src = src; a.srcRowPitch = srcRowPitch; a.srcFormat = srcFormat;
    a.dst = dst; a.dstRowPitch = dstRowPitch; a.dstFormat = dstFormat;
    a.width = width; a.height = height; a.nbw = (width + 3) / 4; a.nbh = (height + 3) / 4;
    a.plan = plan;
    const uint64_t n = uint64_t(a.nbw) * a.nbh;
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(bc_decode_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}
} // namespace dxtex
