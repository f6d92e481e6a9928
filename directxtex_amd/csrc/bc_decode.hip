// BC1-BC7 decoders for gfx950: DecompressBC (DirectXTexCompress.cpp:425-535) with the block decoders
// D3DXDecodeBC1/2/3 (BC.cpp:318-364, :802-826, :902-941), D3DXDecodeBC4U/4S/5U/5S (BC4BC5.cpp:36-151, :389-494),
// D3DX_BC6H::Decode (BC6HBC7.cpp:1658-1813) and D3DX_BC7::Decode (:2566-2780).
//
// One lane decodes one 4x4 block to 16 fp32 texels (what the reference's BC_DECODE hooks produce), runs the
// ConvertScanline plan and stores the texels that fall inside the image with StoreScanline semantics. Lanes of a
// wavefront own consecutive blocks of a block row, so every one of the four row stores of a wave is contiguous.
// The decoders keep everything in registers; no scratch, no LDS.
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
    int direct8;     // BC1-3 / BC7 -> R8G8B8A8_UNORM(_SRGB), BC4 -> R8, BC5 -> R8G8 with an empty plan: texels are picked from stored palette bytes
    int direct16;    // BC6H -> R16G16B16A16_FLOAT with an empty plan: the decoder's halves are the texels
    ConvertPlan plan;
};

// XMVectorLerp(a, b, t) = a + (b - a) * t, unfused (DirectXMath, SSE2 shape)
__device__ __forceinline__ float lerp1(float a, float b, float t) { return a + (b - a) * t; }

// The four colours of a BC1-style colour block (D3DXDecodeBC1's clr[], BC.cpp:318-364) and its 2-bit index word
__device__ __forceinline__ void bc1_palette(const uint8_t* p, bool isbc1, Texel (&clr)[4], uint32_t& bitmap)
{
    const uint32_t c01 = *reinterpret_cast<const uint32_t*>(p);
    bitmap = *reinterpret_cast<const uint32_t*>(p + 4);
    const uint32_t w0 = c01 & 0xFFFF, w1 = c01 >> 16;
    // XMLoadU565 -> (x = bits 0-4, y = bits 5-10, z = bits 11-15) * (1/31, 1/63, 1/31), swizzled to (z, y, x)
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
}

__device__ __forceinline__ void decode_bc1(const uint8_t* p, bool isbc1, Texel (&out)[16])
{
    Texel clr[4]; uint32_t bitmap;
    bc1_palette(p, isbc1, clr, bitmap);
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

// Eight-entry palettes are kept in eight separate scalars, not an array: the optimiser rewrites a select chain over the elements of
// one array into a load with a per-lane index, which pins the array to scratch or LDS.
#define DXTEX_PAL8(n) float n##0, n##1, n##2, n##3, n##4, n##5, n##6, n##7
#define DXTEX_PAL8_ARGS(n) n##0, n##1, n##2, n##3, n##4, n##5, n##6, n##7
#define DXTEX_PAL8_REFS(n) float& n##0, float& n##1, float& n##2, float& n##3, float& n##4, float& n##5, float& n##6, float& n##7

// The eight alpha values of a BC3 alpha block (D3DXDecodeBC3, BC.cpp:902-941: multiplications by 1/7 and 1/5)
__device__ __forceinline__ void bc3_alpha_palette(uint64_t d, DXTEX_PAL8_REFS(f))
{
    const uint32_t a0 = uint32_t(d & 0xFF), a1 = uint32_t((d >> 8) & 0xFF);
    f0 = float(a0) * (1.0f / 255.0f);
    f1 = float(a1) * (1.0f / 255.0f);
    if (a0 > a1)
    {
        f2 = (f0 * 6.0f + f1 * 1.0f) * (1.0f / 7.0f); f3 = (f0 * 5.0f + f1 * 2.0f) * (1.0f / 7.0f); f4 = (f0 * 4.0f + f1 * 3.0f) * (1.0f / 7.0f);
        f5 = (f0 * 3.0f + f1 * 4.0f) * (1.0f / 7.0f); f6 = (f0 * 2.0f + f1 * 5.0f) * (1.0f / 7.0f); f7 = (f0 * 1.0f + f1 * 6.0f) * (1.0f / 7.0f);
    }
    else
    {
        f2 = (f0 * 4.0f + f1 * 1.0f) * (1.0f / 5.0f); f3 = (f0 * 3.0f + f1 * 2.0f) * (1.0f / 5.0f);
        f4 = (f0 * 2.0f + f1 * 3.0f) * (1.0f / 5.0f); f5 = (f0 * 1.0f + f1 * 4.0f) * (1.0f / 5.0f);
        f6 = 0.0f; f7 = 1.0f;
    }
}

#define DXTEX_PAL8_VALS(n) float n##0, float n##1, float n##2, float n##3, float n##4, float n##5, float n##6, float n##7
__device__ __forceinline__ float select8(uint32_t s, DXTEX_PAL8_VALS(f))
{
    float v = f0;
    v = (s == 1u) ? f1 : v; v = (s == 2u) ? f2 : v; v = (s == 3u) ? f3 : v; v = (s == 4u) ? f4 : v;
    v = (s == 5u) ? f5 : v; v = (s == 6u) ? f6 : v; v = (s == 7u) ? f7 : v;
    return v;
}

__device__ __forceinline__ void decode_bc3_alpha(const uint8_t* p, Texel (&out)[16])
{
    const uint64_t d = *reinterpret_cast<const uint64_t*>(p);
    DXTEX_PAL8(f);
    bc3_alpha_palette(d, DXTEX_PAL8_ARGS(f));
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i].a = select8(uint32_t(d >> (16 + 3 * i)) & 7u, DXTEX_PAL8_ARGS(f));
}

// The eight values of a BC4 / BC5 channel block: BC4_UNORM::R / BC4_SNORM::R (BC4BC5.cpp:36-151) for index 0..7 - true divisions,
// unlike BC3's alpha
__device__ __forceinline__ void bc4_palette(uint64_t d, bool isSigned, DXTEX_PAL8_REFS(f))
{
    bool gt;
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
    if (gt)
    {
        f2 = (f0 * 6.0f + f1 * 1.0f) / 7.0f; f3 = (f0 * 5.0f + f1 * 2.0f) / 7.0f; f4 = (f0 * 4.0f + f1 * 3.0f) / 7.0f;
        f5 = (f0 * 3.0f + f1 * 4.0f) / 7.0f; f6 = (f0 * 2.0f + f1 * 5.0f) / 7.0f; f7 = (f0 * 1.0f + f1 * 6.0f) / 7.0f;
    }
    else
    {
        f2 = (f0 * 4.0f + f1 * 1.0f) / 5.0f; f3 = (f0 * 3.0f + f1 * 2.0f) / 5.0f;
        f4 = (f0 * 2.0f + f1 * 3.0f) / 5.0f; f5 = (f0 * 1.0f + f1 * 4.0f) / 5.0f;
        f6 = isSigned ? -1.0f : 0.0f; f7 = 1.0f;
    }
}

// Byte selectors (v_perm_b32) for the four texels of row y of a 3-bit index block: byte k of the result = index of texel 4y + k
__device__ __forceinline__ uint32_t row_selector3(uint64_t d, uint32_t y)
{
    const uint32_t x = uint32_t(d >> (16u + 12u * y)) & 0xFFFu;
    return (x & 7u) | ((x & 0x38u) << 5) | ((x & 0x1C0u) << 10) | ((x & 0xE00u) << 15);
}
// eight palette bytes (lo = entries 0-3, hi = 4-7) -> the four bytes of a row
__device__ __forceinline__ uint32_t row_bytes3(uint32_t lo, uint32_t hi, uint64_t d, uint32_t y) { return __builtin_amdgcn_perm(hi, lo, row_selector3(d, y)); }

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

// b >>= s, s < 32
__device__ __forceinline__ void shr128(Bits128& b, uint32_t s)
{
    b.w0 = __builtin_amdgcn_alignbit(b.w1, b.w0, s);
    b.w1 = __builtin_amdgcn_alignbit(b.w2, b.w1, s);
    b.w2 = __builtin_amdgcn_alignbit(b.w3, b.w2, s);
    b.w3 = __builtin_amdgcn_alignbit(0u, b.w3, s);
}

// v with a zero bit inserted at position p (the implicit top bit of an anchor index, :2663-2700); `on` false: v unchanged
__device__ __forceinline__ uint64_t insert_zero(uint64_t v, uint32_t p, bool on)
{
    const uint64_t keep = on ? ((1ull << p) - 1ull) : ~0ull;
    return v + (v & ~keep);
}

// 16 mask bits -> 16 two-bit fields
__device__ __forceinline__ uint32_t spread16(uint32_t x)
{
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// D3DX_BC7::Decode (:2566-2780) in integers: out[i] = texel i as packed R8G8B8A8 (the LDRColorA the reference converts to floats with
// * 1/255). The block is consumed as a 128-bit shift register: the header fields, then one shift per endpoint channel (the nEnd fields
// of a channel are at most 30 bits), then the p-bits; what remains is the first index stream. Index streams get their anchors' implicit
// zero bits inserted once, so the 16 texels read fixed-width fields; endpoints are unquantized three colour channels at a time.
__device__ __forceinline__ void decode_bc7_packed(const uint8_t* p, uint32_t (&out)[16])
{
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    Bits128 b; b.w0 = raw.x; b.w1 = raw.y; b.w2 = raw.z; b.w3 = raw.w;
    const uint32_t low8 = b.w0 & 0xFFu;
    if (low8 == 0)
    {
        // reserved mode 8 (or no mode bit in the first byte): transparent black (:2771-2778)
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = 0u;
        return;
    }
    const uint32_t mode = uint32_t(__ffs(int(low8))) - 1u;
    const uint32_t mw = bc7_mode_word(mode);
    const uint32_t parts = mw & 3u, partBits = (mw >> 2) & 7u, pBits = (mw >> 5) & 7u, rotBits = (mw >> 8) & 3u, imBits = (mw >> 10) & 1u;
    const uint32_t ib = (mw >> 11) & 7u, ib2 = (mw >> 14) & 3u, cp = (mw >> 16) & 15u, ap = (mw >> 20) & 15u;
    const uint32_t nEnd = (parts + 1u) << 1;
    uint32_t pos = mode + 1u;                                    // at most 14 header bits, all in the first word
    const uint32_t shape = __builtin_amdgcn_ubfe(b.w0, pos, partBits); pos += partBits;
    const uint32_t rot = __builtin_amdgcn_ubfe(b.w0, pos, rotBits); pos += rotBits;
    const uint32_t im = __builtin_amdgcn_ubfe(b.w0, pos, imBits); pos += imBits;
    shr128(b, pos);

    // endpoints, channel-major in the stream (:2618-2640); e[i] = packed RGBA of endpoint i. Fields of endpoints the mode does not have
    // (i >= nEnd) read as garbage below 2^prec; no texel ever selects them.
    uint32_t e[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (uint32_t ch = 0; ch < 3; ++ch)
    {
#pragma unroll
        for (uint32_t i = 0; i < 6; ++i) e[i] |= __builtin_amdgcn_ubfe(b.w0, i * cp, cp) << (8u * ch);
        shr128(b, nEnd * cp);
    }
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i) e[i] |= __builtin_amdgcn_ubfe(b.w0, i * ap, ap) << 24;
    shr128(b, nEnd * ap);
    // p-bits (:2643-2660): one per endpoint, or one per endpoint pair (mode 1)
    const uint32_t pb = __builtin_amdgcn_ubfe(b.w0, 0u, pBits);
    shr128(b, pBits);
    const uint32_t hasP = pBits ? 1u : 0u;
    const bool perPair = pBits != 0 && pBits != nEnd;
    const uint32_t cpp = cp + hasP, app = ap ? ap + hasP : 0u;
    const uint32_t cmask = (0xFFu >> cpp) * 0x010101u;           // the bits of (s >> cpp) that stay inside their own byte
#pragma unroll
    for (uint32_t i = 0; i < 6; ++i)
    {
        const uint32_t pbit = (pb >> (perPair ? (i >> 1) : i)) & 1u;                   // pb == 0 without p-bits
        // D3DX_BC7::Unquantize (:826-831) of the three colour channels at once: every channel is below 2^cpp, so the left shift
        // cannot cross a byte and the right shift's spill into the byte below is masked
        uint32_t c3 = ((e[i] & 0xFFFFFFu) << hasP) | (pbit ? 0x010101u : 0u);
        c3 <<= 8u - cpp;
        c3 |= (c3 >> cpp) & cmask;
        uint32_t al = ((e[i] >> 24) << (ap ? hasP : 0u)) | (ap ? pbit : 0u);
        al <<= 8u - app;
        al = ap ? (al | (al >> app)) : 255u;                                               // colour-only modes: alpha = 255
        e[i] = c3 | (al << 24);
    }

    // partition (2 bits per texel) and anchors, read once
    const uint32_t reg = (parts == 2) ? kPart3Bits[shape] : (parts == 1) ? spread16(uint32_t(kPart2Mask[shape])) : 0u;
    const uint32_t an3 = uint32_t(kAnchor3[shape]);
    const uint32_t anchorA = (parts == 2) ? (an3 & 15u) : uint32_t(kAnchor2[shape]), anchorB = an3 >> 4;
    const uint32_t pA = (anchorA + 1u) * ib - 1u, pB = (anchorB + 1u) * ib - 1u;
    const uint32_t pLo = (parts == 2) ? min(pA, pB) : pA, pHi = max(pA, pB);

    // index streams: the first is what is left in the register; the second (modes 4 and 5, ib == 2) starts 16 * ib - 1 = 31 bits later
    uint64_t s1 = uint64_t(b.w0) | (uint64_t(b.w1) << 32);
    uint64_t s2 = uint64_t(__builtin_amdgcn_alignbit(b.w1, b.w0, 31u)) | (uint64_t(__builtin_amdgcn_alignbit(b.w2, b.w1, 31u)) << 32);
    s1 = insert_zero(s1, ib - 1u, true);
    s1 = insert_zero(s1, pLo, parts >= 1);
    s1 = insert_zero(s1, pHi, parts == 2);
    s2 = ib2 ? insert_zero(s2, ib2 - 1u, true) : s1;
    const uint32_t nb2 = ib2 ? ib2 : ib;
    uint64_t sc = im ? s2 : s1, sa = im ? s1 : s2;               // colour and alpha streams (:2703-2720)
    const uint32_t bC = im ? nb2 : ib, bA = im ? ib : nb2;
    const uint32_t mC = (1u << bC) - 1u, mA = (1u << bA) - 1u, hC = mC >> 1, hA = mA >> 1;
    const uint32_t magC = weight_magic(bC), magA = weight_magic(bA);
    // rotation (:2745-2753) swaps alpha with channel rot - 1: a byte permute of the packed texel
    const uint32_t rsel = (rot == 1) ? 0x00020103u : (rot == 2) ? 0x01020300u : (rot == 3) ? 0x02030100u : 0x03020100u;
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i)
    {
        const uint32_t wc = uint32_t(sc) & mC, wa = uint32_t(sa) & mA;
        sc >>= bC; sa >>= bA;
        const uint32_t kc = __umul24(wc * 64u + hC, magC) >> 16, ka = __umul24(wa * 64u + hA, magA) >> 16;
        const uint32_t rg = (reg >> (2u * i)) & 3u;
        const uint32_t e0 = (rg == 0) ? e[0] : (rg == 1) ? e[2] : e[4];
        const uint32_t e1 = (rg == 0) ? e[1] : (rg == 1) ? e[3] : e[5];
        const uint32_t rb = ((__umul24(e0 & 0x00FF00FFu, 64u - kc) + __umul24(e1 & 0x00FF00FFu, kc) + 0x00200020u) >> 6) & 0x00FF00FFu;
        const uint32_t g = (__umul24((e0 >> 8) & 0xFFu, 64u - kc) + __umul24((e1 >> 8) & 0xFFu, kc) + 32u) >> 6;
        const uint32_t al = (__umul24(e0 >> 24, 64u - ka) + __umul24(e1 >> 24, ka) + 32u) >> 6;
        out[i] = __builtin_amdgcn_perm(0u, rb | (g << 8) | (al << 24), rsel);
    }
}

__device__ __forceinline__ void decode_bc7(const uint8_t* p, Texel (&out)[16])
{
    uint32_t pk[16];
    decode_bc7_packed(p, pk);
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        out[i].r = float(pk[i] & 0xFFu) * (1.0f / 255.0f); out[i].g = float((pk[i] >> 8) & 0xFFu) * (1.0f / 255.0f);
        out[i].b = float((pk[i] >> 16) & 0xFFu) * (1.0f / 255.0f); out[i].a = float(pk[i] >> 24) * (1.0f / 255.0f);
    }
}

// n <= 16 bits at a position the caller knows at compile time (the word selects fold away)
__device__ __forceinline__ uint32_t take_bits(const Bits128& b, uint32_t pos, uint32_t n)
{
    const uint32_t k = pos >> 5;
    const uint32_t lo = (k == 0) ? b.w0 : (k == 1) ? b.w1 : (k == 2) ? b.w2 : b.w3;
    const uint32_t hi = (k == 0) ? b.w1 : (k == 1) ? b.w2 : (k == 2) ? b.w3 : 0u;
    return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(hi, lo, pos & 31u), 0u, n);
}

// One run of the gather list (bc67_tables.h): the header bits [pos, pos + len) of a field, shifted to their place in the field
// (the words by value: a select between loads through a reference becomes a load through a selected pointer, which pins the block
// to scratch memory)
__device__ __forceinline__ uint32_t bc6h_extra_run(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t d)
{
    const uint32_t pos = d & 127u, len = (d >> 7) & 7u, lo = (d >> 10) & 15u;
    const uint32_t k = pos >> 5;                                 // header bits end at 82: words 0..2
    const uint32_t wl = (k == 0) ? w0 : (k == 1) ? w1 : w2;
    const uint32_t wh = (k == 0) ? w1 : (k == 1) ? w2 : w3;
    uint32_t v = __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(wh, wl, pos & 31u), 0u, len);      // len == 0 (no run): 0
    const uint32_t r = __brev(v) >> ((32u - len) & 31u);                                            // stored high bit first
    v = (d & 0x4000u) ? r : v;
    return v << lo;
}

// D3DX_BC6H::Decode (:1658-1813): out[i] = texel i as halves, (r | g << 16, b | 1.0h << 16). The header (ms_aDesc, :879-1048) is read as
// a gather: twelve fields at fixed positions whose lengths depend on the mode, plus up to fourteen short runs listed per mode, each with
// a compile-time destination field - so endpoints stay in registers. `false` = reserved mode: FillWithErrorColors (:1805-1811).
__device__ __forceinline__ bool decode_bc6h_half(const uint8_t* p, bool isSigned, uint2 (&out)[16])
{
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    Bits128 b; b.w0 = raw.x; b.w1 = raw.y; b.w2 = raw.z; b.w3 = raw.w;
    uint32_t mode = b.w0 & 3u;
    if (mode != 0 && mode != 1) mode = b.w0 & 31u;
    const int mi = kBc6hModeIndex[mode];
    if (mi < 0) return false;
    const Bc6hMode info = kBc6hModes[mi];
    const uint64_t lens = kBc6hMainLen[mi];
    const uint4 x0 = reinterpret_cast<const uint4*>(kBc6hExtra[mi])[0], x1 = reinterpret_cast<const uint4*>(kBc6hExtra[mi])[1];
    const uint32_t xw[8] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w };

    // (a set bit in an unused header position, :1703-1711, cannot occur: no mode has one below its header length - checked by the
    // table generator)
    const uint32_t shape = take_bits(b, kBc6hMainPos[0], uint32_t(lens) & 15u);
    int ep[4][3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            const int q = 1 + 4 * ch + e;                        // field 3 + 4 * channel + endpoint, minus 2
            if (q < 12) ep[e][ch] = int(take_bits(b, kBc6hMainPos[q], uint32_t(lens >> (4 * q)) & 15u));
            else ep[e][ch] = 0;                                  // B1.b has no fixed position
        }
#pragma unroll
    for (int sl = 0; sl < 14; ++sl)
    {
        const int f = kBc6hExtraField[sl] - 3;
        const uint32_t d = (sl & 1) ? (xw[sl >> 1] >> 16) : (xw[sl >> 1] & 0xFFFFu);
        ep[f & 3][f >> 2] |= int(bc6h_extra_run(raw.x, raw.y, raw.z, raw.w, d));
    }

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
    const uint32_t ibits = info.indexBits, magic = weight_magic(ibits), imask = (1u << ibits) - 1u, half = imask >> 1;
    // indices: 46 bits from header bit 82 (two regions) or 63 bits from bit 65, the anchors' implicit zero bits inserted once
    uint64_t st = info.regions2 ? (uint64_t(__builtin_amdgcn_alignbit(b.w3, b.w2, 18u)) | (uint64_t(b.w3 >> 18) << 32))
                                : (uint64_t(__builtin_amdgcn_alignbit(b.w3, b.w2, 1u)) | (uint64_t(b.w3 >> 1) << 32));
    st = insert_zero(st, ibits - 1u, true);
    st = insert_zero(st, (uint32_t(kAnchor2[sh]) + 1u) * ibits - 1u, info.regions2 != 0);
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i)
    {
        const uint32_t idx = uint32_t(st) & imask;
        st >>= ibits;
        const bool r1 = ((reg2 >> i) & 1u) != 0;
        const int w = int(__umul24(idx * 64u + half, magic) >> 16);
        uint32_t h[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
        {
            const int a = r1 ? ep[2][ch] : ep[0][ch], bq = r1 ? ep[3][ch] : ep[1][ch];
            const int v = bc6h::finish_unquantize((a * (64 - w) + bq * w + 32) >> 6, isSigned);
            h[ch] = bc6h::int_to_f16(v, isSigned) & 0xFFFFu;
        }
        out[i] = make_uint2(h[0] | (h[1] << 16), h[2] | 0x3C000000u);
    }
    return true;
}

__device__ __forceinline__ void decode_bc6h(const uint8_t* p, bool isSigned, Texel (&out)[16])
{
    uint2 hv[16];
    if (!decode_bc6h_half(p, isSigned, hv)) { fill_error(out); return; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        out[i].r = __half2float(__ushort_as_half(uint16_t(hv[i].x & 0xFFFFu)));
        out[i].g = __half2float(__ushort_as_half(uint16_t(hv[i].x >> 16)));
        out[i].b = __half2float(__ushort_as_half(uint16_t(hv[i].y & 0xFFFFu)));
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
        const bool isbc1 = a.srcFormat == FMT_BC1_UNORM || a.srcFormat == FMT_BC1_UNORM_SRGB;
        const bool isbc2 = a.srcFormat == FMT_BC2_UNORM || a.srcFormat == FMT_BC2_UNORM_SRGB;
        if (a.direct8)
        {
            // RGBA8 target, empty plan: the four colours (and BC3's eight alphas) are stored once per block with StoreScanline's
            // arithmetic, the texels pick bytes
            Texel clr[4]; uint32_t bitmap;
            bc1_palette(isbc1 ? p : p + 8, isbc1, clr, bitmap);
            uint32_t c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = pack_texel32(FMT_R8G8B8A8_UNORM, clr[k]);
            const uint64_t ad = isbc1 ? 0ull : *reinterpret_cast<const uint64_t*>(p);
            uint32_t alo = 0, ahi = 0;
            if (!isbc1 && !isbc2)
            {
                DXTEX_PAL8(f);
                bc3_alpha_palette(ad, DXTEX_PAL8_ARGS(f));
                alo = store_ubn_biased(f0) | (store_ubn_biased(f1) << 8) | (store_ubn_biased(f2) << 16) | (store_ubn_biased(f3) << 24);
                ahi = store_ubn_biased(f4) | (store_ubn_biased(f5) << 8) | (store_ubn_biased(f6) << 16) | (store_ubn_biased(f7) << 24);
            }
            const uint32_t x0 = bx * 4, y0 = by * 4;
            const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
#pragma unroll
            for (uint32_t y = 0; y < 4; ++y)
            {
                if (y < ph)
                {
                    uint32_t a4 = 0;
                    if (isbc2)
                    {
                        // float(n) * (1 / 15.f) stores as 17 n for every nibble n
                        uint32_t n = uint32_t(ad >> (16u * y)) & 0xFFFFu;
                        n = (n | (n << 8)) & 0x00FF00FFu;
                        n = (n | (n << 4)) & 0x0F0F0F0Fu;
                        a4 = n | (n << 4);
                    }
                    else if (!isbc1)
                        a4 = row_bytes3(alo, ahi, ad, y);
                    uint32_t px[4];
#pragma unroll
                    for (uint32_t x = 0; x < 4; ++x)
                    {
                        const uint32_t sidx = (bitmap >> (2u * (4u * y + x))) & 3u;
                        const uint32_t v = (sidx == 0) ? c[0] : (sidx == 1) ? c[1] : (sidx == 2) ? c[2] : c[3];
                        px[x] = isbc1 ? v : __builtin_amdgcn_perm(a4, v, 0x04020100u + (x << 24));
                    }
                    uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
                    if (a.vec16 && pw == 4)
                        reinterpret_cast<uint4*>(row)[bx] = make_uint4(px[0], px[1], px[2], px[3]);
                    else
                    {
#pragma unroll
                        for (uint32_t x = 0; x < 4; ++x)
                            if (x < pw) reinterpret_cast<uint32_t*>(row)[x0 + x] = px[x];
                    }
                }
            }
            return;
        }
        if (isbc1)
            decode_bc1(p, true, t);
        else
        {
            decode_bc1(p + 8, false, t);
            if (isbc2)
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
        const bool one = a.srcFormat == FMT_BC4_UNORM || a.srcFormat == FMT_BC4_SNORM;
        const uint64_t d0 = reinterpret_cast<const uint64_t*>(p)[0], d1 = one ? 0ull : reinterpret_cast<const uint64_t*>(p)[1];
        DXTEX_PAL8(f); DXTEX_PAL8(g);
        bc4_palette(d0, sg, DXTEX_PAL8_ARGS(f));
        g0 = g1 = g2 = g3 = g4 = g5 = g6 = g7 = 0.0f;
        if (!one) bc4_palette(d1, sg, DXTEX_PAL8_ARGS(g));
        if (a.direct8)
        {
            // R8 / R8G8 target of the block's own kind (UNORM or SNORM), empty plan: eight stored bytes per channel, texels pick.
            // StoreScanline R8_UNORM: biased truncation; R8_SNORM: round half away from zero; R8G8: XMStoreUByteN2 / XMStoreByteN2
            // (dxtex_store.h)
            auto byte_of = [&](float v) -> uint32_t
            {
                if (one)
                {
                    if (!sg) return store_ubn_biased(v);
                    float c = (v < 1.0f) ? v : 1.0f; c = (c > -1.0f) ? c : -1.0f;
                    return uint32_t(int32_t(roundf(c * 127.0f))) & 0xFFu;
                }
                return sg ? (uint32_t(store_bn(v)) & 0xFFu) : store_ubn2(v);
            };
            const uint32_t lo0 = byte_of(f0) | (byte_of(f1) << 8) | (byte_of(f2) << 16) | (byte_of(f3) << 24);
            const uint32_t hi0 = byte_of(f4) | (byte_of(f5) << 8) | (byte_of(f6) << 16) | (byte_of(f7) << 24);
            uint32_t lo1 = 0, hi1 = 0;
            if (!one)
            {
                lo1 = byte_of(g0) | (byte_of(g1) << 8) | (byte_of(g2) << 16) | (byte_of(g3) << 24);
                hi1 = byte_of(g4) | (byte_of(g5) << 8) | (byte_of(g6) << 16) | (byte_of(g7) << 24);
            }
            const uint32_t x0 = bx * 4, y0 = by * 4;
            const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
#pragma unroll
            for (uint32_t y = 0; y < 4; ++y)
            {
                if (y < ph)
                {
                    uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
                    const uint32_t r4 = row_bytes3(lo0, hi0, d0, y);
                    if (one)
                    {
                        if (a.vec16 && pw == 4) reinterpret_cast<uint32_t*>(row)[bx] = r4;
                        else
                        {
#pragma unroll
                            for (uint32_t x = 0; x < 4; ++x)
                                if (x < pw) row[x0 + x] = uint8_t(r4 >> (8 * x));
                        }
                    }
                    else
                    {
                        const uint32_t g4 = row_bytes3(lo1, hi1, d1, y);
                        const uint32_t w0 = __builtin_amdgcn_perm(g4, r4, 0x05010400u), w1 = __builtin_amdgcn_perm(g4, r4, 0x07030602u);
                        if (a.vec16 && pw == 4) reinterpret_cast<uint2*>(row)[bx] = make_uint2(w0, w1);
                        else
                        {
#pragma unroll
                            for (uint32_t x = 0; x < 4; ++x)
                                if (x < pw) reinterpret_cast<uint16_t*>(row)[x0 + x] = uint16_t(((x < 2) ? w0 : w1) >> (16 * (x & 1)));
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            t[i].r = select8(uint32_t(d0 >> (16 + 3 * i)) & 7u, DXTEX_PAL8_ARGS(f));
            t[i].g = one ? 0.0f : select8(uint32_t(d1 >> (16 + 3 * i)) & 7u, DXTEX_PAL8_ARGS(g));
            t[i].b = 0.0f; t[i].a = 1.0f;
        }
    }
    else if constexpr (FAM == FAM_BC6H)
    {
        if (a.direct16)
        {
            // half -> float -> StoreScanline's half (clamped to +-65504) gives the decoder's bits back, except for the one infinity the
            // signed format can produce (-32768 * 31 >> 5 = 0x7C00, the 16-bit mode), which the clamp turns into -65504
            uint2 hv[16];
            const bool sf = a.srcFormat == FMT_BC6H_SF16;
            if (!decode_bc6h_half(p, sf, hv))
            {
#pragma unroll
                for (int i = 0; i < 16; ++i) hv[i] = make_uint2(0u, 0x3C000000u);                  // opaque black
            }
            else if (sf)
            {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                {
                    // magnitudes never exceed 0x7C00, so "bit 10..14 all set" is exactly the infinity
                    const uint32_t ix = hv[i].x & 0x7C007C00u, iy = hv[i].y & 0x00007C00u;
                    uint32_t fx = 0u;
                    if ((ix & 0xFFFFu) == 0x7C00u) fx |= 0x00000001u;
                    if ((ix >> 16) == 0x7C00u) fx |= 0x00010000u;
                    hv[i].x -= fx;                                                                  // 0x7C00 -> 0x7BFF, sign kept
                    hv[i].y -= (iy == 0x7C00u) ? 1u : 0u;
                }
            }
            const uint32_t x0 = bx * 4, y0 = by * 4;
            const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
#pragma unroll
            for (uint32_t y = 0; y < 4; ++y)
            {
                if (y < ph)
                {
                    uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
                    if (a.vec16 && pw == 4)
                    {
                        reinterpret_cast<uint4*>(row)[bx * 2] = make_uint4(hv[y * 4].x, hv[y * 4].y, hv[y * 4 + 1].x, hv[y * 4 + 1].y);
                        reinterpret_cast<uint4*>(row)[bx * 2 + 1] = make_uint4(hv[y * 4 + 2].x, hv[y * 4 + 2].y, hv[y * 4 + 3].x, hv[y * 4 + 3].y);
                    }
                    else
                    {
#pragma unroll
                        for (uint32_t x = 0; x < 4; ++x)
                            if (x < pw) reinterpret_cast<uint2*>(row)[x0 + x] = hv[y * 4 + x];
                    }
                }
            }
            return;
        }
        decode_bc6h(p, a.srcFormat == FMT_BC6H_SF16, t);
    }
    else
    {
        if (a.direct8)
        {
            // float(c) * (1/255) stored with XMStoreUByteN4's bias / scale / truncate gives c back for every byte c, so the
            // round trip through fp32 texels is skipped
            uint32_t pk[16];
            decode_bc7_packed(p, pk);
            const uint32_t x0 = bx * 4, y0 = by * 4;
            const uint32_t pw = min(4u, a.width - x0), ph = min(4u, a.height - y0);
#pragma unroll
            for (uint32_t y = 0; y < 4; ++y)
            {
                if (y < ph)
                {
                    uint8_t* row = a.dst + uint64_t(y0 + y) * a.dstRowPitch;
                    if (a.vec16 && pw == 4)
                        reinterpret_cast<uint4*>(row)[bx] = make_uint4(pk[y * 4], pk[y * 4 + 1], pk[y * 4 + 2], pk[y * 4 + 3]);
                    else
                    {
#pragma unroll
                        for (uint32_t x = 0; x < 4; ++x)
                            if (x < pw) reinterpret_cast<uint32_t*>(row)[x0 + x] = pk[y * 4 + x];
                    }
                }
            }
            return;
        }
        decode_bc7(p, t);
    }

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
    const bool emptyPlan = !plan.srgbIn && !plan.srgbOut && plan.tcv == TCV_NONE && plan.tsw == TSW_NONE && !plan.depth;
    const bool rgba8 = dstFormat == FMT_R8G8B8A8_UNORM || dstFormat == FMT_R8G8B8A8_UNORM_SRGB;
    switch (srcFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB: case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB:
    case FMT_BC7_UNORM: case FMT_BC7_UNORM_SRGB:
        a.direct8 = rgba8 && emptyPlan; break;
    case FMT_BC4_UNORM: a.direct8 = dstFormat == FMT_R8_UNORM && emptyPlan; break;
    case FMT_BC4_SNORM: a.direct8 = dstFormat == FMT_R8_SNORM && emptyPlan; break;
    case FMT_BC5_UNORM: a.direct8 = dstFormat == FMT_R8G8_UNORM && emptyPlan; break;
    case FMT_BC5_SNORM: a.direct8 = dstFormat == FMT_R8G8_SNORM && emptyPlan; break;
    default: a.direct8 = 0; break;
    }
    a.direct16 = (srcFormat == FMT_BC6H_UF16 || srcFormat == FMT_BC6H_SF16) && dstFormat == FMT_R16G16B16A16_FLOAT &&
                 emptyPlan;
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
