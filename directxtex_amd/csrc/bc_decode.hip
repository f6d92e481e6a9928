// BC1-BC7 decoders for gfx950: DecompressBC (DirectXTexCompress.cpp:425-535) with the block decoders
// D3DXDecodeBC1/2/3 (BC.cpp:318-364, :802-826, :902-941), D3DXDecodeBC4U/4S/5U/5S (BC4BC5.cpp:36-151, :389-494),
// D3DX_BC6H::Decode (BC6HBC7.cpp:1658-1813) and D3DX_BC7::Decode (:2566-2780).
//
// One lane decodes one 4x4 block to 16 fp32 texels (what the reference's BC_DECODE hooks produce), runs the
// ConvertScanline plan and stores the texels that fall inside the image with StoreScanline semantics. Lanes of a
// wavefront own consecutive blocks of a block row, so every one of the four row stores of a wave is contiguous.
// The decoders keep everything in registers (BC6H: its scattered header goes through LDS); no scratch.
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
    int vec16;       // dst and dstRowPitch are 16-byte aligned
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

// ---- BC6H / BC7 -------------------------------------------------------------------------------------------------------
// Everything below is written so that no array is ever indexed by a per-lane value: endpoints live in constant-indexed
// registers (packed RGBA8 for BC7), region / anchor lookups become shifts of two table words read once, and the index
// fields are read in place with running bit positions. (A runtime-indexed local array would be spilled to scratch,
// i.e. to HBM-backed memory - which is what made the first version of this kernel 10x slower than its traffic.)
struct Bits128 { uint32_t w0, w1, w2, w3; };

// n <= 16 bits starting at bit `pos` (pos + n <= 128)
__device__ __forceinline__ uint32_t peek_bits(const Bits128& b, uint32_t pos, uint32_t n)
{
    const uint32_t k = pos >> 5;
    const uint32_t lo = (k == 0) ? b.w0 : (k == 1) ? b.w1 : (k == 2) ? b.w2 : b.w3;
    const uint32_t hi = (k == 0) ? b.w1 : (k == 1) ? b.w2 : (k == 2) ? b.w3 : 0u;
    return __builtin_amdgcn_alignbit(hi, lo, pos & 31u) & ((1u << n) - 1u);
}

__device__ __forceinline__ void fill_error(Texel (&out)[16])
{
    // FillWithErrorColors, release flavour: opaque black (:1638-1650)
#pragma unroll
    for (int i = 0; i < 16; ++i) { out[i].r = out[i].g = out[i].b = 0.0f; out[i].a = 1.0f; }
}

// ms_aInfo (BC6HBC7.cpp:1106-1124), one 32-bit word per mode:
// parts(2) | partBits(3)<<2 | pBits(3)<<5 | rotBits(2)<<8 | imBits(1)<<10 | ib(3)<<11 | ib2(2)<<14 | cp(4)<<16 | ap(4)<<20 | pbit-per-endpoint(1)<<24
#define BC7_MODE_WORD(parts, partBits, pBits, rotBits, imBits, ib, ib2, cp, ap) \
    (uint32_t(parts) | uint32_t(partBits) << 2 | uint32_t(pBits) << 5 | uint32_t(rotBits) << 8 | uint32_t(imBits) << 10 | uint32_t(ib) << 11 | \
     uint32_t(ib2) << 14 | uint32_t(cp) << 16 | uint32_t(ap) << 20)
__device__ __forceinline__ uint32_t bc7_mode_word(uint32_t mode)
{
    switch (mode)
    {
    case 0: return BC7_MODE_WORD(2, 4, 6, 0, 0, 3, 0, 4, 0);
    case 1: return BC7_MODE_WORD(1, 6, 2, 0, 0, 3, 0, 6, 0);
    case 2: return BC7_MODE_WORD(2, 6, 0, 0, 0, 2, 0, 5, 0);
    case 3: return BC7_MODE_WORD(1, 6, 4, 0, 0, 2, 0, 7, 0);
    case 4: return BC7_MODE_WORD(0, 0, 0, 2, 1, 2, 3, 5, 6);
    case 5: return BC7_MODE_WORD(0, 0, 0, 2, 0, 2, 2, 7, 8);
    case 6: return BC7_MODE_WORD(0, 0, 2, 0, 0, 4, 0, 7, 7);
    default: return BC7_MODE_WORD(1, 6, 4, 0, 0, 2, 0, 5, 5);
    }
}

// g_aWeights2/3/4 (:327-329) in closed form: w(i) = ((64 i + (n-1)/2) * ceil(65536 / (n-1))) >> 16 with n = 2^bits
__device__ __forceinline__ uint32_t weight_magic(uint32_t bits) { return (bits == 2) ? 21846u : (bits == 3) ? 9363u : 4370u; }
__device__ __forceinline__ uint32_t bc67_weight(uint32_t bits, uint32_t magic, uint32_t i)
{
    return ((i * 64u + (((1u << bits) - 1u) >> 1)) * magic) >> 16;
}

// D3DX_BC7::Unquantize (:826-831) of one component that has `prec` significant bits
__device__ __forceinline__ uint32_t bc7_unq(uint32_t c, uint32_t prec)
{
    const uint32_t s = (c << (8u - prec)) & 0xFFu;
    return s | (s >> prec);
}

__device__ __forceinline__ void decode_bc7(const uint8_t* p, Texel (&out)[16])
{
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    Bits128 b; b.w0 = raw.x; b.w1 = raw.y; b.w2 = raw.z; b.w3 = raw.w;
    const uint32_t low8 = b.w0 & 0xFFu;
    if (low8 == 0)
    {
        // reserved mode 8 (or no mode bit in the first byte): transparent black (:2771-2778)
#pragma unroll
        for (int i = 0; i < 16; ++i) { out[i].r = out[i].g = out[i].b = out[i].a = 0.0f; }
        return;
    }
    const uint32_t mode = uint32_t(__ffs(int(low8))) - 1u;
    const uint32_t mw = bc7_mode_word(mode);
    const uint32_t parts = mw & 3u, partBits = (mw >> 2) & 7u, pBits = (mw >> 5) & 7u, rotBits = (mw >> 8) & 3u, imBits = (mw >> 10) & 1u;
    const uint32_t ib = (mw >> 11) & 7u, ib2 = (mw >> 14) & 3u, cp = (mw >> 16) & 15u, ap = (mw >> 20) & 15u;
    const uint32_t nEnd = (parts + 1u) << 1;
    uint32_t pos = mode + 1u;
    const uint32_t shape = peek_bits(b, pos, partBits); pos += partBits;
    const uint32_t rot = peek_bits(b, pos, rotBits); pos += rotBits;
    const uint32_t im = peek_bits(b, pos, imBits); pos += imBits;

    // endpoints, channel-major in the stream (:2618-2640); e[i] = packed RGBA of endpoint i
    uint32_t e[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (uint32_t ch = 0; ch < 4; ++ch)
    {
        const uint32_t prec = (ch == 3) ? ap : cp;
#pragma unroll
        for (uint32_t i = 0; i < 6; ++i)
            if (i < nEnd) { e[i] |= peek_bits(b, pos, prec) << (8u * ch); pos += prec; }
    }
    // p-bits (:2643-2660): one per endpoint, or one per endpoint pair (mode 1)
    const uint32_t pb = peek_bits(b, pos, pBits); pos += pBits;
    const bool perPair = pBits != 0 && pBits != nEnd;
    const uint32_t cpp = cp + (pBits ? 1u : 0u), app = (ap && pBits) ? ap + 1u : ap;
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i)
    {
        const uint32_t pbit = pBits ? (pb >> (perPair ? (i >> 1) : i)) & 1u : 0u;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t ch = 0; ch < 4; ++ch)
        {
            uint32_t c = (e[i] >> (8u * ch)) & 0xFFu;
            const uint32_t pr = (ch == 3) ? ap : cp, prp = (ch == 3) ? app : cpp;
            if (pr != prp) c = ((c << 1) | pbit) & 0xFFu;
            c = (ch == 3 && ap == 0) ? 255u : bc7_unq(c, prp);            // colour-only modes: alpha = 255
            v |= c << (8u * ch);
        }
        e[i] = v;
    }

    // partition row and anchors, read once
    const uint32_t reg2 = (parts == 1) ? uint32_t(kPart2Mask[shape]) : 0u;          // 1 bit per texel
    const uint32_t reg3 = (parts == 2) ? kPart3Bits[shape] : 0u;                    // 2 bits per texel
    const uint32_t an3 = (parts == 2) ? uint32_t(kAnchor3[shape]) : 0u;
    const uint32_t anchorA = (parts == 1) ? uint32_t(kAnchor2[shape]) : (parts == 2) ? (an3 & 15u) : 0u;   // 0 == "texel 0", always an anchor
    const uint32_t anchorB = (parts == 2) ? (an3 >> 4) : 0u;

    uint32_t pos1 = pos;                                         // first index set: 16 ib-bit fields minus one bit per anchor
    uint32_t pos2 = pos + 16u * ib - (parts + 1u);               // second index set (modes 4 and 5)
    const uint32_t wcBits = (ib2 && im) ? ib2 : ib, waBits = ib2 ? (im ? ib : ib2) : ib;
    const uint32_t mc = weight_magic(wcBits), ma = weight_magic(waBits);
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i)
    {
        const bool anchor = (i == 0) || (i == anchorA) || (i == anchorB);
        const uint32_t n1 = ib - (anchor ? 1u : 0u);
        const uint32_t i1 = peek_bits(b, pos1, n1); pos1 += n1;
        uint32_t i2 = 0;
        if (ib2) { const uint32_t n2 = ib2 - (i == 0 ? 1u : 0u); i2 = peek_bits(b, pos2, n2); pos2 += n2; }
        const uint32_t rg = ((reg2 >> i) & 1u) | ((reg3 >> (2u * i)) & 3u);
        const uint32_t e0 = (rg == 0) ? e[0] : (rg == 1) ? e[2] : e[4];
        const uint32_t e1 = (rg == 0) ? e[1] : (rg == 1) ? e[3] : e[5];
        const uint32_t wc = ib2 ? (im ? i2 : i1) : i1, wa = ib2 ? (im ? i1 : i2) : i1;
        const uint32_t kc = bc67_weight(wcBits, mc, wc), ka = bc67_weight(waBits, ma, wa);
        uint32_t px[4];
#pragma unroll
        for (uint32_t ch = 0; ch < 4; ++ch)
        {
            const uint32_t k = (ch == 3) ? ka : kc;
            px[ch] = (((e0 >> (8u * ch)) & 0xFFu) * (64u - k) + ((e1 >> (8u * ch)) & 0xFFu) * k + 32u) >> 6;
        }
        // rotation (:2745-2753): swap alpha with channel rot-1
        const uint32_t al = px[3];
        if (rot == 1) { px[3] = px[0]; px[0] = al; }
        else if (rot == 2) { px[3] = px[1]; px[1] = al; }
        else if (rot == 3) { px[3] = px[2]; px[2] = al; }
        out[i].r = float(px[0]) * (1.0f / 255.0f); out[i].g = float(px[1]) * (1.0f / 255.0f);
        out[i].b = float(px[2]) * (1.0f / 255.0f); out[i].a = float(px[3]) * (1.0f / 255.0f);
    }
}

// ep: [field][lane] in LDS, field = endpoint * 3 + channel (A0, B0, A1, B1). The header is scattered bit by bit
// (ms_aDesc, :879-1048), which needs a store whose target depends on the lane's mode: LDS takes that in one ds_or.
__device__ __forceinline__ void decode_bc6h(const uint8_t* p, bool isSigned, Texel (&out)[16], int (*epLds)[256])
{
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    Bits128 b; b.w0 = raw.x; b.w1 = raw.y; b.w2 = raw.z; b.w3 = raw.w;
    const uint32_t lane = threadIdx.x;
    uint32_t mode = b.w0 & 3u, pos = 2;
    if (mode != 0 && mode != 1) { mode = (b.w0 & 31u); pos = 5; }
    const int mi = kBc6hModeIndex[mode];
    if (mi < 0) { fill_error(out); return; }      // reserved modes decode to opaque black (:1805-1811)
    const Bc6hMode info = kBc6hModes[mi];
    const uint8_t* desc = kBc6hHeader[mi];

#pragma unroll
    for (int f = 0; f < 12; ++f) epLds[f][lane] = 0;
    uint32_t shape = 0;
    const uint32_t headerBits = info.regions2 ? 82u : 65u;
    bool bad = false;
    for (; pos < headerBits; ++pos)
    {
        if (!peek_bits(b, pos, 1)) continue;
        const uint32_t f = desc[pos] >> 4, bit = desc[pos] & 15u;
        if (f == 2) shape |= 1u << bit;
        else if (f >= 3) { const uint32_t k = f - 3u; atomicOr(&epLds[(k & 3u) * 3u + (k >> 2)][lane], 1 << bit); }
        else if (f == 0) bad = true;               // a set bit in an unused header position (:1703-1711)
    }
    if (bad) { fill_error(out); return; }
    int ep[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) ep[e][ch] = epLds[e * 3 + ch][lane];

    // sign extension and inverse delta transform (:1735-1767)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
    {
        const int prec = info.prec[ch], delta = info.delta[ch];
        if (isSigned) ep[0][ch] = bc6h::sign_extend(ep[0][ch], prec);
        if (isSigned || info.transformed)
        {
            ep[1][ch] = bc6h::sign_extend(ep[1][ch], delta);
            if (info.regions2) { ep[2][ch] = bc6h::sign_extend(ep[2][ch], delta); ep[3][ch] = bc6h::sign_extend(ep[3][ch], delta); }
        }
        if (info.transformed)
        {
            // TransformInverse (:1153-1165): applied to both regions' slots regardless of the region count
            const int wrap = (1 << prec) - 1;
#pragma unroll
            for (int e = 1; e < 4; ++e)
            {
                ep[e][ch] = (ep[e][ch] + ep[0][ch]) & wrap;
                if (isSigned) ep[e][ch] = bc6h::sign_extend(ep[e][ch], prec);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ep[e][ch] = bc6h::unquantize(ep[e][ch], prec, isSigned);
    }

    const uint32_t sh = shape & 31u;
    const uint32_t reg2 = info.regions2 ? uint32_t(kPart2Mask[sh]) : 0u;
    const uint32_t anchorA = info.regions2 ? uint32_t(kAnchor2[sh]) : 0u;
    const uint32_t ibits = info.indexBits, magic = weight_magic(ibits);
    pos = headerBits;
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i)
    {
        const uint32_t nb = ibits - ((i == 0 || i == anchorA) ? 1u : 0u);
        const uint32_t idx = peek_bits(b, pos, nb); pos += nb;
        const bool r1 = ((reg2 >> i) & 1u) != 0;
        const int w = int(bc67_weight(ibits, magic, idx));
        uint32_t h[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
        {
            const int a = r1 ? ep[2][ch] : ep[0][ch], bq = r1 ? ep[3][ch] : ep[1][ch];
            const int v = bc6h::finish_unquantize((a * (64 - w) + bq * w + 32) >> 6, isSigned);
            h[ch] = bc6h::int_to_f16(v, isSigned);
        }
        out[i].r = __half2float(__ushort_as_half(uint16_t(h[0])));
        out[i].g = __half2float(__ushort_as_half(uint16_t(h[1])));
        out[i].b = __half2float(__ushort_as_half(uint16_t(h[2])));
        out[i].a = 1.0f;
    }
}

enum : int { FAM_BC123 = 0, FAM_BC45 = 1, FAM_BC6H = 2, FAM_BC7 = 3 };

// One instantiation per codec family, so the register budget (and the LDS of BC6H) is that of the family in use.
template <int FAM>
__global__ void __launch_bounds__(256) bc_decode_kernel(DecodeArgs a)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nbw * a.nbh) return;
    const uint32_t by = nb / a.nbw, bx = nb - by * a.nbw;
    const uint32_t bb = (a.srcFormat == FMT_BC1_UNORM || a.srcFormat == FMT_BC1_UNORM_SRGB || a.srcFormat == FMT_BC4_UNORM || a.srcFormat == FMT_BC4_SNORM) ? 8u : 16u;
    const uint8_t* p = a.src + uint64_t(by) * a.srcRowPitch + uint64_t(bx) * bb;

    Texel t[16];
    if constexpr (FAM == FAM_BC123)
    {
        if (a.srcFormat == FMT_BC1_UNORM || a.srcFormat == FMT_BC1_UNORM_SRGB)
            decode_bc1(p, true, t);
        else
        {
            decode_bc1(p + 8, false, t);
            if (a.srcFormat == FMT_BC2_UNORM || a.srcFormat == FMT_BC2_UNORM_SRGB)
            {
                const uint64_t al = *reinterpret_cast<const uint64_t*>(p);
#pragma unroll
                for (int i = 0; i < 16; ++i) t[i].a = float(uint32_t(al >> (4 * i)) & 15u) * (1.0f / 15.0f);
            }
            else
                decode_bc3_alpha(p, t);
        }
    }
    else if constexpr (FAM == FAM_BC45)
    {
        const bool sg = a.srcFormat == FMT_BC4_SNORM || a.srcFormat == FMT_BC5_SNORM;
        if (a.srcFormat == FMT_BC4_UNORM || a.srcFormat == FMT_BC4_SNORM)
        {
            const uint64_t d = *reinterpret_cast<const uint64_t*>(p);
#pragma unroll
            for (int i = 0; i < 16; ++i) { t[i].r = bc4_value(d, i, sg); t[i].g = 0.0f; t[i].b = 0.0f; t[i].a = 1.0f; }
        }
        else
        {
            const uint64_t d0 = reinterpret_cast<const uint64_t*>(p)[0], d1 = reinterpret_cast<const uint64_t*>(p)[1];
#pragma unroll
            for (int i = 0; i < 16; ++i) { t[i].r = bc4_value(d0, i, sg); t[i].g = bc4_value(d1, i, sg); t[i].b = 0.0f; t[i].a = 1.0f; }
        }
    }
    else if constexpr (FAM == FAM_BC6H)
    {
        __shared__ int epLds[12][256];
        decode_bc6h(p, a.srcFormat == FMT_BC6H_SF16, t, epLds);
    }
    else
        decode_bc7(p, t);

    const uint32_t x0 = bx * 4, y0 = by * 4;
    const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
    // full-width rows of 4-, 8- and 16-byte texels leave as 16-byte stores (lanes of a wave own consecutive blocks, so a
    // wave writes 1, 2 or 4 KiB contiguous per row); anything else goes texel by texel
    const bool vec = a.vec16 && pw == 4;
    const bool packed32 = is_packed32(a.dstFormat);
#pragma unroll
    for (uint32_t y = 0; y < 4; ++y)
    {
        if (y >= ph) break;
        uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
        Texel q[4];
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) q[x] = apply_plan(t[y * 4 + x], a.plan);
        if (vec && packed32)
            reinterpret_cast<uint4*>(row)[bx] = make_uint4(pack_texel32(a.dstFormat, q[0]), pack_texel32(a.dstFormat, q[1]),
                                                           pack_texel32(a.dstFormat, q[2]), pack_texel32(a.dstFormat, q[3]));
        else if (vec && a.dstFormat == FMT_R16G16B16A16_FLOAT)
        {
            const uint2 h0 = pack_texel_half4(q[0]), h1 = pack_texel_half4(q[1]), h2 = pack_texel_half4(q[2]), h3 = pack_texel_half4(q[3]);
            reinterpret_cast<uint4*>(row)[bx * 2] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            reinterpret_cast<uint4*>(row)[bx * 2 + 1] = make_uint4(h2.x, h2.y, h3.x, h3.y);
        }
        else
        {
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x)
                if (x < pw) store_texel(row, x0 + x, a.dstFormat, q[x]);
        }
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
    a.vec16 = ((reinterpret_cast<uintptr_t>(dst) | dstRowPitch) & 15u) == 0;
    const uint64_t n = uint64_t(a.nbw) * a.nbh;
    if (!n) return hipSuccess;
    const dim3 grid(uint32_t((n + 255) / 256)), wg(256);
    switch (srcFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB: case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB:
        hipLaunchKernelGGL(bc_decode_kernel<FAM_BC123>, grid, wg, 0, stream, a); break;
    case FMT_BC4_UNORM: case FMT_BC4_SNORM: case FMT_BC5_UNORM: case FMT_BC5_SNORM:
        hipLaunchKernelGGL(bc_decode_kernel<FAM_BC45>, grid, wg, 0, stream, a); break;
    case FMT_BC6H_UF16: case FMT_BC6H_SF16:
        hipLaunchKernelGGL(bc_decode_kernel<FAM_BC6H>, grid, wg, 0, stream, a); break;
    case FMT_BC7_UNORM: case FMT_BC7_UNORM_SRGB:
        hipLaunchKernelGGL(bc_decode_kernel<FAM_BC7>, grid, wg, 0, stream, a); break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
} // namespace dxtex
