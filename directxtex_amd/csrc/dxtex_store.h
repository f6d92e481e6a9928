// Device-side StoreScanline (DirectXTexConvert.cpp:1629-2533) for the formats this library writes, and the general
// ConvertScanline plan (DirectXTexConvert.cpp:3080-3854) resolved once per call on the host.
//
// The packed stores live in DirectXMath (not vendored by the reference). What is restated
// here is its SSE2 behaviour, the one every x64 build of the reference gets (since round 6 the checker is the reference's own
// StoreScanline compiled over oracle/shim's leaf stand-ins of these same stores):
//   XMStoreUByteN4  : clamp to [0,1], multiply by 255 (lanes pre-shifted by powers of two, which does not change the
//                     rounding), TRUNCATE (_mm_cvttps_epi32). The reference adds g_8BitBias = 0.5/255 beforehand
//                     (:1767) precisely because of that truncation.
//   XMStoreUByteN2  : saturate, v * 255 + 0.5 (two roundings), truncate.
//   XMStoreByteN2/N4: clamp to [-1,1], v * 127, round to nearest even.
//   XMStoreUShortN4 : saturate, v * 65535, round to nearest even (_mm_cvtps_epi32).
//   XMStoreHalf4 / XMConvertFloatToHalf: IEEE round to nearest even (== v_cvt_f16_f32), after the reference's clamp
//                     to +-65504 (:1695-1699).
#pragma once
#include "dxtex_device.h"

namespace dxtex
{
// ---- ConvertScanline plan ---------------------------------------------------------------------------------------
enum : int
{
    TSW_RGB_TO_R_GRAY = 4,   // RGB(A) UNORM -> R format: x = dot3(v, g_Grayscale) (:3758-3768)
    TSW_SPLAT_X = 5,         // -> A format / COPY_RED: (x,x,x,x) (:3631-3641)
    TSW_SPLAT_Y = 6,         // COPY_GREEN
    TSW_SPLAT_Z = 7,         // COPY_BLUE
    TSW_GRAY_SPLAT = 8,      // !A UNORM RGB -> A format: splat(dot3(v, g_Grayscale)) (:3612-3622)
    TSW_G_TO_R = 9,          // RGB -> R with COPY_GREEN: x = y (:3712-3722)
    TSW_B_TO_R = 10,         // x = z
    TSW_A_TO_R = 11,         // x = w
    TSW_RA_TO_RG = 12, TSW_GA_TO_RG = 13, TSW_BA_TO_RG = 14,   // RGBA -> RG with COPY_ALPHA (:3777-3812)
    TSW_RB_TO_RG = 15, TSW_GB_TO_RG = 16,                      // RGB -> RG with COPY flags (:3817-3838)
};

struct ConvertPlan
{
    int srgbIn;      // XMColorSRGBToRGB first
    int tcv;         // TCV_*
    int tsw;         // TSW_*
    int srgbOut;     // XMColorRGBToSRGB last
    int depth;       // TDP_* word (dxtex_device.h): the depth conversions, in place of the range conversion
};

__device__ __forceinline__ Texel apply_plan(Texel t, const ConvertPlan& p)
{
    if (p.srgbIn) { t.r = srgb_to_linear1(t.r); t.g = srgb_to_linear1(t.g); t.b = srgb_to_linear1(t.b); }
    if (p.depth) t = apply_depth(t, p.depth);
    if (p.tcv != TCV_NONE)
    {
        t.r = tcv1(t.r, p.tcv); t.g = tcv1(t.g, p.tcv); t.b = tcv1(t.b, p.tcv); t.a = tcv1(t.a, p.tcv);
    }
    switch (p.tsw)
    {
    case TSW_R_TO_RGB: t.g = t.r; t.b = t.r; break;
    case TSW_R_TO_RG: t.g = t.r; break;
    case TSW_A_TO_RGB: t.r = t.a; t.g = t.a; t.b = t.a; break;
    case TSW_RGB_TO_R_GRAY:
        // XMVector3Dot, SSE2 shape: (x*gx + z*gz) ... the shim and this agree on ((x*gx + y*gy) + z*gz)
        t.r = (t.r * 0.2125f + t.g * 0.7154f) + t.b * 0.0721f; break;
    case TSW_GRAY_SPLAT: { const float d = (t.r * 0.2125f + t.g * 0.7154f) + t.b * 0.0721f; t.r = t.g = t.b = t.a = d; break; }
    case TSW_SPLAT_X: t.g = t.b = t.a = t.r; break;
    case TSW_SPLAT_Y: t.r = t.b = t.a = t.g; break;
    case TSW_SPLAT_Z: t.r = t.g = t.a = t.b; break;
    case TSW_G_TO_R: t.r = t.g; break;
    case TSW_B_TO_R: t.r = t.b; break;
    case TSW_A_TO_R: t.r = t.a; break;
    case TSW_RA_TO_RG: t.g = t.a; break;
    case TSW_GA_TO_RG: t.r = t.g; t.g = t.a; break;
    case TSW_BA_TO_RG: t.r = t.b; t.g = t.a; break;
    case TSW_RB_TO_RG: t.g = t.b; break;
    case TSW_GB_TO_RG: t.r = t.g; t.g = t.b; break;
    default: break;
    }
    if (p.srgbOut) { t.r = linear_to_srgb1(t.r); t.g = linear_to_srgb1(t.g); t.b = linear_to_srgb1(t.b); }
    return t;
}

// ---- packed stores ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t store_ubn_biased(float v)
{
    // v + g_8BitBias, then XMStoreUByteN4's clamp / scale / truncate
    float s = v + (0.5f / 255.0f);
    s = (s > 0.0f) ? s : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    return uint32_t(s * 255.0f);
}

__device__ __forceinline__ uint32_t store_ubn2(float v)
{
    float s = (v > 0.0f) ? v : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    return uint32_t(s * 255.0f + 0.5f);
}

__device__ __forceinline__ int32_t store_bn(float v)
{
    float s = (v > -1.0f) ? v : -1.0f; s = (s < 1.0f) ? s : 1.0f;
    return int32_t(rintf(s * 127.0f));
}

__device__ __forceinline__ uint16_t store_half(float v)
{
    float s = (v > -65504.0f) ? v : -65504.0f; s = (s < 65504.0f) ? s : 65504.0f;      // XMVectorClamp(v, g_HalfMin, g_HalfMax)
    return __half_as_ushort(__float2half_rn(s));
}

__device__ __forceinline__ uint32_t store_usn(float v)
{
    float s = (v > 0.0f) ? v : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    return uint32_t(rintf(s * 65535.0f));
}

// XMStoreShortN4 / XMStoreShortN2: clamp to [-1,1], * 32767, round to nearest even (_mm_cvtps_epi32), saturating pack.
__device__ __forceinline__ uint32_t store_sn16(float v)
{
    float s = (v > -1.0f) ? v : -1.0f; s = (s < 1.0f) ? s : 1.0f;
    return uint32_t(int32_t(rintf(s * 32767.0f))) & 0xFFFFu;
}

// XMStoreUDecN4, SSE2 path: saturate, scale, TRUNCATE (_mm_cvttps_epi32), mask. The lanes are pre-scaled by powers of two and masked
// back, which leaves floor(v * 1023) per colour channel and floor(v * 3) for alpha.
__device__ __forceinline__ uint32_t store_udecn4(const Texel& t)
{
    auto sat = [](float v) { float s = (v > 0.0f) ? v : 0.0f; return (s < 1.0f) ? s : 1.0f; };
    return (uint32_t(sat(t.r) * 1023.0f) & 0x3FFu) | ((uint32_t(sat(t.g) * 1023.0f) & 0x3FFu) << 10) |
           ((uint32_t(sat(t.b) * 1023.0f) & 0x3FFu) << 20) | ((uint32_t(sat(t.a) * 3.0f) & 0x3u) << 30);
}

// One component of XMStoreFloat3PK: fp32 -> unsigned small float with 5 exponent bits and MB mantissa bits, round to nearest even;
// negative values and values below half the smallest denormal go to 0, values above the largest finite go to it, NaN / +INF keep
// their class.
__device__ __forceinline__ uint32_t store_float11(float v, int MB)
{
    const uint32_t bits = __float_as_uint(v);
    const bool sign = (bits & 0x80000000u) != 0;
    uint32_t I = bits & 0x7FFFFFFFu;
    const uint32_t expAll = 0x1Fu << MB, all = expAll | ((1u << MB) - 1u);
    const int shift = 23 - MB;                           // 17 for x / y, 18 for z
    if ((I & 0x7F800000u) == 0x7F800000u)
    {
        if (I & 0x7FFFFFu) return all;                   // NaN
        return sign ? 0u : expAll;                       // -INF is clamped to 0
    }
    if (sign || I < ((MB == 6) ? 0x35800000u : 0x36000000u)) return 0u;      // positive only; below the smallest denormal (2^-20 / 2^-19)
    if (I > ((MB == 6) ? 0x477E0000u : 0x477C0000u)) return expAll - 1u;     // larger than the largest finite value: clamp to it
    if (I < 0x38800000u)
    {
        const uint32_t Shift = 113u - (I >> 23);         // denormal in the small format
        I = (0x800000u | (I & 0x7FFFFFu)) >> Shift;
    }
    else I += 0xC8000000u;                               // re-bias the exponent
    return ((I + ((1u << (shift - 1)) - 1u) + ((I >> shift) & 1u)) >> shift) & all;
}

// XMStoreFloat3SE (DirectXMath >= 3.10, which is what the reference calls: DirectXTexConvert.cpp:155-156). The mantissas are rounded
// to nearest EVEN (DirectXMath's Internal::round_to_nearest); the reference's private copy for DirectXMath < 3.10 (:158-191) uses lroundf
// (half away from zero) and is not compiled against a current DirectXMath.
__device__ __forceinline__ uint32_t store_float3se(const Texel& t)
{
    const float maxf9 = float(0x1FF << 7), minf9 = 1.0f / float(1 << 16);
    const float x = (t.r >= 0.0f) ? ((t.r > maxf9) ? maxf9 : t.r) : 0.0f;
    const float y = (t.g >= 0.0f) ? ((t.g > maxf9) ? maxf9 : t.g) : 0.0f;
    const float z = (t.b >= 0.0f) ? ((t.b > maxf9) ? maxf9 : t.b) : 0.0f;
    const float max_xy = (x > y) ? x : y;
    const float max_xyz = (max_xy > z) ? max_xy : z;
    const float maxColor = (max_xyz > minf9) ? max_xyz : minf9;
    const uint32_t fi = __float_as_uint(maxColor) + 0x00004000u;      // round up leaving 9 bits in the fraction (including the assumed 1)
    const uint32_t e = fi >> 23;
    const float scaleR = __uint_as_float(0x83000000u - (e << 23));
    return (uint32_t(int32_t(rintf(x * scaleR))) & 0x1FFu) | ((uint32_t(int32_t(rintf(y * scaleR))) & 0x1FFu) << 9) |
           ((uint32_t(int32_t(rintf(z * scaleR))) & 0x1FFu) << 18) | (((e - 0x6Fu) & 0x1Fu) << 27);
}

// XMStoreU565 / XMStoreU555 / XMStoreUNibble4 after the reference's scaling (v * 31 | 63 | 15, no bias on x64): clamp to [0, max], round
// to nearest even (_mm_cvtps_epi32).
__device__ __forceinline__ uint32_t store_scaled_rne(float v, float scale)
{
    float s = v * scale;
    s = (s > 0.0f) ? s : 0.0f; s = (s < scale) ? s : scale;
    return uint32_t(int32_t(rintf(s)));
}


// ---- integer stores --------------------------------------------------------------------------------------------------------
// XMStoreUInt4/3/2 and XMConvertVectorFloatToUInt(v, 0), SSE2 path: max(v, 0) (a NaN becomes 0: maxps returns its second operand),
// values above g_XMMaxUInt = 65536 * 65536 - 256 = 4294967040.0f (the largest fp32 below 2^32) give 0xFFFFFFFF, everything else is
// TRUNCATED (cvttps2dq, with the usual subtract-2^31 detour above 2^31, which is exact).
__device__ __forceinline__ uint32_t store_u32(float v)
{
    const float s = (v > 0.0f) ? v : 0.0f;
    if (s > 4294967040.0f) return 0xFFFFFFFFu;
    if (s < 2147483648.0f) return uint32_t(int32_t(s));
    return uint32_t(int32_t(s - 2147483648.0f)) ^ 0x80000000u;
}
// XMStoreSInt4/3/2 and XMConvertVectorFloatToInt(v, 0): truncation; above 2147483520.0f (65536 * 32768 - 128) the result is
// 0x7FFFFFFF; below -2^31 and for NaN cvttps2dq's "integer indefinite" 0x80000000.
__device__ __forceinline__ uint32_t store_s32(float v)
{
    if (v > 2147483520.0f) return 0x7FFFFFFFu;
    if (!(v >= -2147483648.0f)) return 0x80000000u;
    return uint32_t(int32_t(v));
}
// XMStoreUShort4/2, XMStoreShort4/2, XMStoreUByte4/2, XMStoreByte4/2: clamp to [lo, hi] (maxps then minps: a NaN becomes lo), round to
// nearest even (cvtps2dq). lo is -32767 / -127 for the signed ones (g_ShortMin / g_ByteMin).
__device__ __forceinline__ int32_t store_clamp_rne(float v, float lo, float hi)
{
    float s = (v > lo) ? v : lo; s = (s < hi) ? s : hi;
    return int32_t(rintf(s));
}
// The single-channel integer formats are the reference's own scalar code (:1913-2016): std::min / std::max, then a C++ cast (truncation)
__device__ __forceinline__ int32_t store_clamp_trunc(float v, float lo, float hi)
{
    float s = (hi < v) ? hi : v;        // std::min<float>(v, hi)
    s = (s < lo) ? lo : s;              // std::max<float>(., lo); a NaN passes through both and converts to 0 here (the reference's cast
    return int32_t(s);                  // of a NaN is undefined; x86-64 stores 0 in the 8- and 16-bit fields)
}
// XMStoreUDec4: maxps / minps (a NaN becomes lo), then cvttps2dq
__device__ __forceinline__ int32_t store_clamp_trunc_sse(float v, float lo, float hi)
{
    float s = (v > lo) ? v : lo; s = (s < hi) ? s : hi;
    return int32_t(s);
}
// XMStoreUDecN4_XR: v * (510, 510, 510, 3) + (384, 384, 384, 0), clamp to [0, (1023, 1023, 1023, 3)], truncate
__device__ __forceinline__ uint32_t store_xr(const Texel& t)
{
    auto one = [](float v, float scale, float bias, float hi) { float s = v * scale + bias; s = (s > 0.0f) ? s : 0.0f; s = (s < hi) ? s : hi; return uint32_t(s); };
    return (one(t.r, 510.0f, 384.0f, 1023.0f) & 0x3FFu) | ((one(t.g, 510.0f, 384.0f, 1023.0f) & 0x3FFu) << 10) |
           ((one(t.b, 510.0f, 384.0f, 1023.0f) & 0x3FFu) << 20) | (one(t.a, 3.0f, 0.0f, 3.0f) << 30);
}
// XMStoreUByteN4 WITHOUT the reference's bias (the AYUV store calls it on the raw vector, :2184): saturate, * 255, truncate
__device__ __forceinline__ uint32_t store_ubn_plain(float v)
{
    float s = (v > 0.0f) ? v : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    return uint32_t(s * 255.0f);
}

// The 4-byte formats as one packed word (same expressions as the cases of store_texel below).
__device__ __forceinline__ bool is_packed32(int format)
{
    switch (format)
    {
    case FMT_R8G8B8A8_UNORM: case FMT_R8G8B8A8_UNORM_SRGB: case FMT_B8G8R8A8_UNORM: case FMT_B8G8R8A8_UNORM_SRGB:
    case FMT_B8G8R8X8_UNORM: case FMT_B8G8R8X8_UNORM_SRGB: case FMT_R8G8B8A8_SNORM:
        return true;
    default:
        return false;
    }
}

__device__ __forceinline__ uint32_t pack_texel32(int format, const Texel& t)
{
    switch (format)
    {
    case FMT_R8G8B8A8_UNORM:
    case FMT_R8G8B8A8_UNORM_SRGB:
        return store_ubn_biased(t.r) | (store_ubn_biased(t.g) << 8) | (store_ubn_biased(t.b) << 16) | (store_ubn_biased(t.a) << 24);
    case FMT_B8G8R8A8_UNORM:
    case FMT_B8G8R8A8_UNORM_SRGB:
        return store_ubn_biased(t.b) | (store_ubn_biased(t.g) << 8) | (store_ubn_biased(t.r) << 16) | (store_ubn_biased(t.a) << 24);
    case FMT_B8G8R8X8_UNORM:
    case FMT_B8G8R8X8_UNORM_SRGB:
        // XMVectorPermute<2,1,0,7>(v, g_XMIdentityR3): w = 1 (:2157-2171)
        return store_ubn_biased(t.b) | (store_ubn_biased(t.g) << 8) | (store_ubn_biased(t.r) << 16) | (store_ubn_biased(1.0f) << 24);
    default:   // FMT_R8G8B8A8_SNORM
        return (uint32_t(store_bn(t.r)) & 0xFF) | ((uint32_t(store_bn(t.g)) & 0xFF) << 8) |
               ((uint32_t(store_bn(t.b)) & 0xFF) << 16) | ((uint32_t(store_bn(t.a)) & 0xFF) << 24);
    }
}

__device__ __forceinline__ uint2 pack_texel_half4(const Texel& t)
{
    return make_uint2(uint32_t(store_half(t.r)) | (uint32_t(store_half(t.g)) << 16), uint32_t(store_half(t.b)) | (uint32_t(store_half(t.a)) << 16));
}

// One texel, StoreScanline semantics. Returns false for a format this library cannot write.
__device__ __forceinline__ void store_texel(uint8_t* row, uint32_t x, int format, const Texel& t, float threshold = 0.0f)   // the default of StoreScanline[Linear] (DirectXTexP.h): only Convert passes one
{
    switch (format)
    {
    case FMT_R32G32B32A32_FLOAT:
        reinterpret_cast<float4*>(row)[x] = make_float4(t.r, t.g, t.b, t.a);
        break;
    case FMT_R16G16B16A16_FLOAT:
        reinterpret_cast<uint2*>(row)[x] = pack_texel_half4(t);
        break;
    case FMT_R16G16B16A16_UNORM:
        reinterpret_cast<uint2*>(row)[x] = make_uint2(store_usn(t.r) | (store_usn(t.g) << 16), store_usn(t.b) | (store_usn(t.a) << 16));
        break;
    case FMT_R8G8B8A8_UNORM: case FMT_R8G8B8A8_UNORM_SRGB: case FMT_B8G8R8A8_UNORM: case FMT_B8G8R8A8_UNORM_SRGB:
    case FMT_B8G8R8X8_UNORM: case FMT_B8G8R8X8_UNORM_SRGB: case FMT_R8G8B8A8_SNORM:
        reinterpret_cast<uint32_t*>(row)[x] = pack_texel32(format, t);
        break;
    case FMT_R32G32_FLOAT:
        reinterpret_cast<float2*>(row)[x] = make_float2(t.r, t.g);
        break;
    case FMT_R16G16_FLOAT:
        reinterpret_cast<uint32_t*>(row)[x] = uint32_t(store_half(t.r)) | (uint32_t(store_half(t.g)) << 16);
        break;
    case FMT_R16G16_UNORM:
        reinterpret_cast<uint32_t*>(row)[x] = store_usn(t.r) | (store_usn(t.g) << 16);
        break;
    case FMT_D32_FLOAT:              // :1810-1822, shares R32_FLOAT's case
    case FMT_R32_FLOAT:
        reinterpret_cast<float*>(row)[x] = t.r;
        break;
    case FMT_D32_FLOAT_S8X24_UINT:   // depth as it is, stencil = uint8(min(255, max(0, y))), the other three bytes zero (:1725-1744)
    {
        float sv = (0.0f < t.g) ? t.g : 0.0f;            // std::max<float>(0.f, y): NaN -> 0
        sv = (sv < 255.0f) ? sv : 255.0f;
        reinterpret_cast<uint2*>(row)[x] = make_uint2(__float_as_uint(t.r), uint32_t(sv) & 0xFFu);
        break;
    }
    case FMT_D24_UNORM_S8_UINT:      // XMVectorClamp(v, 0, (1, 255)); uint32(x * 16777215.f) & 0xFFFFFF | (uint32(y) & 0xFF) << 24 (:1852-1869)
    {
        float d = (t.r > 0.0f) ? t.r : 0.0f; d = (d < 1.0f) ? d : 1.0f;
        float sv = (t.g > 0.0f) ? t.g : 0.0f; sv = (sv < 255.0f) ? sv : 255.0f;
        reinterpret_cast<uint32_t*>(row)[x] = (uint32_t(d * 16777215.0f) & 0xFFFFFFu) | ((uint32_t(sv) & 0xFFu) << 24);
        break;
    }
    case FMT_R8G8_UNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t(store_ubn2(t.r) | (store_ubn2(t.g) << 8));
        break;
    case FMT_R8G8_SNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((uint32_t(store_bn(t.r)) & 0xFF) | ((uint32_t(store_bn(t.g)) & 0xFF) << 8));
        break;
    case FMT_R16_FLOAT:
        reinterpret_cast<uint16_t*>(row)[x] = store_half(t.r);       // std::max(std::min(v, 65504), -65504) (:1891)
        break;
    case FMT_D16_UNORM:              // :1897, shares R16_UNORM's case
    case FMT_R16_UNORM:
    {
        // v = clamp(x, 0, 1); uint16(v * 65535 + 0.5) (:1898-1912)
        float s = (t.r < 1.0f) ? t.r : 1.0f; s = (s > 0.0f) ? s : 0.0f;
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t(s * 65535.0f + 0.5f);
        break;
    }
    case FMT_R8_UNORM:
    {
        // v = x + g_8BitBias; clamp; uint8(v * 255) (:1958-1971)
        float s = t.r + (0.5f / 255.0f);
        s = (s < 1.0f) ? s : 1.0f; s = (s > 0.0f) ? s : 0.0f;
        row[x] = uint8_t(s * 255.0f);
        break;
    }
    case FMT_A8_UNORM:
    {
        float s = t.a + (0.5f / 255.0f);
        s = (s < 1.0f) ? s : 1.0f; s = (s > 0.0f) ? s : 0.0f;
        row[x] = uint8_t(s * 255.0f);
        break;
    }
    case FMT_R8_SNORM:
    {
        // clamp to [-1,1]; int8(lroundf(v * 127)) - round half away from zero (:1988-2001)
        float s = (t.r < 1.0f) ? t.r : 1.0f; s = (s > -1.0f) ? s : -1.0f;
        row[x] = uint8_t(int8_t(int32_t(roundf(s * 127.0f))));
        break;
    }
    case FMT_R32G32B32_FLOAT:
    {
        float* d = reinterpret_cast<float*>(row) + 3 * size_t(x);             // XMStoreFloat3
        d[0] = t.r; d[1] = t.g; d[2] = t.b;
        break;
    }
    case FMT_R16G16B16A16_SNORM:
        reinterpret_cast<uint2*>(row)[x] = make_uint2(store_sn16(t.r) | (store_sn16(t.g) << 16), store_sn16(t.b) | (store_sn16(t.a) << 16));
        break;
    case FMT_R16G16_SNORM:
        reinterpret_cast<uint32_t*>(row)[x] = store_sn16(t.r) | (store_sn16(t.g) << 16);
        break;
    case FMT_R16_SNORM:
    {
        // clamp to [-1,1]; int16(lroundf(v * 32767)) (:1928-1941)
        float s = (t.r < 1.0f) ? t.r : 1.0f; s = (s > -1.0f) ? s : -1.0f;
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t(int16_t(int32_t(roundf(s * 32767.0f))));
        break;
    }
    case FMT_R10G10B10A2_UNORM:
        reinterpret_cast<uint32_t*>(row)[x] = store_udecn4(t);
        break;
    case FMT_R11G11B10_FLOAT:
        reinterpret_cast<uint32_t*>(row)[x] = store_float11(t.r, 6) | (store_float11(t.g, 6) << 11) | (store_float11(t.b, 5) << 22);
        break;
    case FMT_R9G9B9E5_SHAREDEXP:
        reinterpret_cast<uint32_t*>(row)[x] = store_float3se(t);
        break;
    case FMT_B5G6R5_UNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((store_scaled_rne(t.b, 31.0f) & 0x1Fu) | ((store_scaled_rne(t.g, 63.0f) & 0x3Fu) << 5) | ((store_scaled_rne(t.r, 31.0f) & 0x1Fu) << 11));
        break;
    case FMT_B5G5R5A1_UNORM:
        // XMStoreU555, then the alpha bit from the caller's threshold (:2116-2139)
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((store_scaled_rne(t.b, 31.0f) & 0x1Fu) | ((store_scaled_rne(t.g, 31.0f) & 0x1Fu) << 5) |
                                                       ((store_scaled_rne(t.r, 31.0f) & 0x1Fu) << 10) | ((t.a > threshold) ? 0x8000u : 0u));
        break;
    case FMT_B4G4R4A4_UNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((store_scaled_rne(t.b, 15.0f) & 0xFu) | ((store_scaled_rne(t.g, 15.0f) & 0xFu) << 4) |
                                                       ((store_scaled_rne(t.r, 15.0f) & 0xFu) << 8) | ((store_scaled_rne(t.a, 15.0f) & 0xFu) << 12));
        break;
    case FMT_A4B4G4R4_UNORM:         // (a, b, g, r) * 15 through XMStoreUNibble4 (:2419-2437)
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((store_scaled_rne(t.a, 15.0f) & 0xFu) | ((store_scaled_rne(t.b, 15.0f) & 0xFu) << 4) |
                                                       ((store_scaled_rne(t.g, 15.0f) & 0xFu) << 8) | ((store_scaled_rne(t.r, 15.0f) & 0xFu) << 12));
        break;
    // ---- integer formats (see store_u32 / store_s32 / store_clamp_rne / store_clamp_trunc above)
    case FMT_R32G32B32A32_UINT: reinterpret_cast<uint4*>(row)[x] = make_uint4(store_u32(t.r), store_u32(t.g), store_u32(t.b), store_u32(t.a)); break;     // :1674-1675
    case FMT_R32G32B32A32_SINT: reinterpret_cast<uint4*>(row)[x] = make_uint4(store_s32(t.r), store_s32(t.g), store_s32(t.b), store_s32(t.a)); break;     // :1677-1678
    case FMT_R32G32B32_UINT: { uint32_t* d = reinterpret_cast<uint32_t*>(row) + 3 * size_t(x); d[0] = store_u32(t.r); d[1] = store_u32(t.g); d[2] = store_u32(t.b); break; }      // :1683-1684
    case FMT_R32G32B32_SINT: { uint32_t* d = reinterpret_cast<uint32_t*>(row) + 3 * size_t(x); d[0] = store_s32(t.r); d[1] = store_s32(t.g); d[2] = store_s32(t.b); break; }      // :1686-1687
    case FMT_R16G16B16A16_UINT:     // XMStoreUShort4, :1707-1708
        reinterpret_cast<uint2*>(row)[x] = make_uint2(uint32_t(store_clamp_rne(t.r, 0.0f, 65535.0f)) | (uint32_t(store_clamp_rne(t.g, 0.0f, 65535.0f)) << 16),
                                                      uint32_t(store_clamp_rne(t.b, 0.0f, 65535.0f)) | (uint32_t(store_clamp_rne(t.a, 0.0f, 65535.0f)) << 16));
        break;
    case FMT_R16G16B16A16_SINT:     // XMStoreShort4, :1713-1714
        reinterpret_cast<uint2*>(row)[x] = make_uint2((uint32_t(store_clamp_rne(t.r, -32767.0f, 32767.0f)) & 0xFFFFu) | (uint32_t(store_clamp_rne(t.g, -32767.0f, 32767.0f)) << 16),
                                                      (uint32_t(store_clamp_rne(t.b, -32767.0f, 32767.0f)) & 0xFFFFu) | (uint32_t(store_clamp_rne(t.a, -32767.0f, 32767.0f)) << 16));
        break;
    case FMT_R32G32_UINT: reinterpret_cast<uint2*>(row)[x] = make_uint2(store_u32(t.r), store_u32(t.g)); break;      // :1719-1720
    case FMT_R32G32_SINT: reinterpret_cast<uint2*>(row)[x] = make_uint2(store_s32(t.r), store_s32(t.g)); break;      // :1722-1723
    case FMT_R10G10B10A2_UINT:      // XMStoreUDec4 (:1753-1754): clamp to [0, (1023, 1023, 1023, 3)], truncate
        reinterpret_cast<uint32_t*>(row)[x] = uint32_t(store_clamp_trunc_sse(t.r, 0.0f, 1023.0f)) | (uint32_t(store_clamp_trunc_sse(t.g, 0.0f, 1023.0f)) << 10) |
                                              (uint32_t(store_clamp_trunc_sse(t.b, 0.0f, 1023.0f)) << 20) | (uint32_t(store_clamp_trunc_sse(t.a, 0.0f, 3.0f)) << 30);
        break;
    case FMT_R10G10B10_XR_BIAS_A2_UNORM: reinterpret_cast<uint32_t*>(row)[x] = store_xr(t); break;      // :1750-1751
    case FMT_R8G8B8A8_UINT:         // XMStoreUByte4, :1774-1775
        reinterpret_cast<uint32_t*>(row)[x] = uint32_t(store_clamp_rne(t.r, 0.0f, 255.0f)) | (uint32_t(store_clamp_rne(t.g, 0.0f, 255.0f)) << 8) |
                                              (uint32_t(store_clamp_rne(t.b, 0.0f, 255.0f)) << 16) | (uint32_t(store_clamp_rne(t.a, 0.0f, 255.0f)) << 24);
        break;
    case FMT_R8G8B8A8_SINT:         // XMStoreByte4, :1780-1781
        reinterpret_cast<uint32_t*>(row)[x] = (uint32_t(store_clamp_rne(t.r, -127.0f, 127.0f)) & 0xFFu) | ((uint32_t(store_clamp_rne(t.g, -127.0f, 127.0f)) & 0xFFu) << 8) |
                                              ((uint32_t(store_clamp_rne(t.b, -127.0f, 127.0f)) & 0xFFu) << 16) | (uint32_t(store_clamp_rne(t.a, -127.0f, 127.0f)) << 24);
        break;
    case FMT_R16G16_UINT: reinterpret_cast<uint32_t*>(row)[x] = uint32_t(store_clamp_rne(t.r, 0.0f, 65535.0f)) | (uint32_t(store_clamp_rne(t.g, 0.0f, 65535.0f)) << 16); break;      // :1801-1802
    case FMT_R16G16_SINT: reinterpret_cast<uint32_t*>(row)[x] = (uint32_t(store_clamp_rne(t.r, -32767.0f, 32767.0f)) & 0xFFFFu) | (uint32_t(store_clamp_rne(t.g, -32767.0f, 32767.0f)) << 16); break;   // :1807-1808
    case FMT_R32_UINT: reinterpret_cast<uint32_t*>(row)[x] = store_u32(t.r); break;      // XMConvertVectorFloatToUInt(v, 0), :1824-1836
    case FMT_R32_SINT: reinterpret_cast<uint32_t*>(row)[x] = store_s32(t.r); break;      // :1838-1850
    case FMT_R8G8_UINT: reinterpret_cast<uint16_t*>(row)[x] = uint16_t(uint32_t(store_clamp_rne(t.r, 0.0f, 255.0f)) | (uint32_t(store_clamp_rne(t.g, 0.0f, 255.0f)) << 8)); break;     // :1873-1874
    case FMT_R8G8_SINT: reinterpret_cast<uint16_t*>(row)[x] = uint16_t((uint32_t(store_clamp_rne(t.r, -127.0f, 127.0f)) & 0xFFu) | ((uint32_t(store_clamp_rne(t.g, -127.0f, 127.0f)) & 0xFFu) << 8)); break;   // :1879-1880
    case FMT_R16_UINT: reinterpret_cast<uint16_t*>(row)[x] = uint16_t(store_clamp_trunc(t.r, 0.0f, 65535.0f)); break;                  // :1913-1926
    case FMT_R16_SINT: reinterpret_cast<uint16_t*>(row)[x] = uint16_t(int16_t(store_clamp_trunc(t.r, -32767.0f, 32767.0f))); break;    // :1943-1956
    case FMT_R8_UINT: row[x] = uint8_t(store_clamp_trunc(t.r, 0.0f, 255.0f)); break;                                                   // :1973-1986
    case FMT_R8_SINT: row[x] = uint8_t(int8_t(store_clamp_trunc(t.r, -127.0f, 127.0f))); break;                                        // :2003-2016
    // ---- 4:4:4 video formats: quantise with the format's RGB store, then the reference's fixed-point matrices (:2173-2268)
    case FMT_AYUV:
    {
        const int r = int(store_ubn_plain(t.r)), g = int(store_ubn_plain(t.g)), b = int(store_ubn_plain(t.b));
        const int y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16, u = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128, v = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128;
        reinterpret_cast<uint32_t*>(row)[x] = uint32_t(min(max(v, 0), 255)) | (uint32_t(min(max(u, 0), 255)) << 8) | (uint32_t(min(max(y, 0), 255)) << 16) | (store_ubn_plain(t.a) << 24);
        break;
    }
    case FMT_Y410:
    {
        const uint32_t q = store_udecn4(t);
        const long long r = q & 0x3FFu, g = (q >> 10) & 0x3FFu, b = (q >> 20) & 0x3FFu;
        const int y = int((16780 * r + 32942 * g + 6544 * b + 32768) >> 16) + 64, u = int((-9683 * r - 19017 * g + 28700 * b + 32768) >> 16) + 512,
                  v = int((28700 * r - 24033 * g - 4667 * b + 32768) >> 16) + 512;
        reinterpret_cast<uint32_t*>(row)[x] = uint32_t(min(max(u, 0), 1023)) | (uint32_t(min(max(y, 0), 1023)) << 10) | (uint32_t(min(max(v, 0), 1023)) << 20) | (q & 0xC0000000u);
        break;
    }
    case FMT_Y416:
    {
        const long long r = store_usn(t.r), g = store_usn(t.g), b = store_usn(t.b);
        const int y = int((16763 * r + 32910 * g + 6537 * b + 32768) >> 16) + 4096, u = int((-9674 * r - 18998 * g + 28672 * b + 32768) >> 16) + 32768,
                  v = int((28672 * r - 24010 * g - 4662 * b + 32768) >> 16) + 32768;
        reinterpret_cast<uint2*>(row)[x] = make_uint2(uint32_t(min(max(u, 0), 65535)) | (uint32_t(min(max(y, 0), 65535)) << 16),
                                                      uint32_t(min(max(v, 0), 65535)) | (store_usn(t.a) << 16));
        break;
    }
    default:
        break;
    }
}

// ---- formats whose memory element holds more than one texel (FC_GROUP): one element from its texels ------------------------------
// texels per element
__host__ __device__ inline uint32_t group_texels(int format) { return format == FMT_R1_UNORM ? 8u : 2u; }
__host__ __device__ inline uint32_t group_bytes(int format) { return format == FMT_R1_UNORM ? 1u : (format == FMT_Y210 || format == FMT_Y216) ? 8u : 4u; }

// Element `g` of a row from texels t[0 .. n) (n = the texels of the element that exist: the last element of an odd-width row is
// short; StoreScanline's missing second texel is a zero vector - which still goes through the colour matrix of the video formats).
__device__ __forceinline__ void store_group(uint8_t* row, uint32_t g, int format, const Texel* t, uint32_t n)
{
    const Texel zero = { 0.0f, 0.0f, 0.0f, 0.0f };
    const Texel t0 = t[0], t1 = (n > 1) ? t[1] : zero;
    switch (format)
    {
    case FMT_R1_UNORM:               // bit set where x > 0.25, first texel in the most significant bit (:2033-2055)
    {
        uint32_t bits = 0;
        for (uint32_t k = 0; k < n; ++k) if (t[k].r > 0.25f) bits |= 0x80u >> k;
        row[g] = uint8_t(bits);
        break;
    }
    case FMT_R8G8_B8G8_UNORM:        // (R0, G0, B0, G1) + g_8BitBias through XMStoreUByteN4 (:2060-2075)
        reinterpret_cast<uint32_t*>(row)[g] = store_ubn_biased(t0.r) | (store_ubn_biased(t0.g) << 8) | (store_ubn_biased(t0.b) << 16) | (store_ubn_biased(t1.g) << 24);
        break;
    case FMT_G8R8_G8B8_UNORM:        // (G0, R0, G1, B0) (:2077-2094)
        reinterpret_cast<uint32_t*>(row)[g] = store_ubn_biased(t0.g) | (store_ubn_biased(t0.r) << 8) | (store_ubn_biased(t1.g) << 16) | (store_ubn_biased(t0.b) << 24);
        break;
    case FMT_YUY2:                   // XMStoreUByteN4 (no bias) of both texels, BT.601 matrix each, chroma averaged (:2274-2308)
    {
        int yy[2], uu[2], vv[2];
        const Texel* src[2] = { &t0, &t1 };
        for (int k = 0; k < 2; ++k)
        {
            const int r = int(store_ubn_plain(src[k]->r)), gr = int(store_ubn_plain(src[k]->g)), b = int(store_ubn_plain(src[k]->b));
            yy[k] = ((66 * r + 129 * gr + 25 * b + 128) >> 8) + 16; uu[k] = ((-38 * r - 74 * gr + 112 * b + 128) >> 8) + 128; vv[k] = ((112 * r - 94 * gr - 18 * b + 128) >> 8) + 128;
        }
        reinterpret_cast<uint32_t*>(row)[g] = uint32_t(min(max(yy[0], 0), 255)) | (uint32_t(min(max((uu[0] + uu[1]) >> 1, 0), 255)) << 8) |
                                              (uint32_t(min(max(yy[1], 0), 255)) << 16) | (uint32_t(min(max((vv[0] + vv[1]) >> 1, 0), 255)) << 24);
        break;
    }
    case FMT_Y210:                   // XMStoreUDecN4 of both texels, the 10-bit matrix, results shifted up six bits (:2310-2353)
    {
        int yy[2], uu[2], vv[2];
        const Texel* src[2] = { &t0, &t1 };
        for (int k = 0; k < 2; ++k)
        {
            const uint32_t q = store_udecn4(*src[k]);
            const long long r = q & 0x3FFu, gr = (q >> 10) & 0x3FFu, b = (q >> 20) & 0x3FFu;
            yy[k] = int((16780 * r + 32942 * gr + 6544 * b + 32768) >> 16) + 64; uu[k] = int((-9683 * r - 19017 * gr + 28700 * b + 32768) >> 16) + 512;
            vv[k] = int((28700 * r - 24033 * gr - 4667 * b + 32768) >> 16) + 512;
        }
        reinterpret_cast<uint2*>(row)[g] = make_uint2((uint32_t(min(max(yy[0], 0), 1023)) << 6) | (uint32_t(min(max((uu[0] + uu[1]) >> 1, 0), 1023)) << 22),
                                                      (uint32_t(min(max(yy[1], 0), 1023)) << 6) | (uint32_t(min(max((vv[0] + vv[1]) >> 1, 0), 1023)) << 22));
        break;
    }
    case FMT_Y216:                   // XMStoreUShortN4 of both texels, the 16-bit matrix (:2355-2397)
    {
        int yy[2], uu[2], vv[2];
        const Texel* src[2] = { &t0, &t1 };
        for (int k = 0; k < 2; ++k)
        {
            const long long r = store_usn(src[k]->r), gr = store_usn(src[k]->g), b = store_usn(src[k]->b);
            yy[k] = int((16763 * r + 32910 * gr + 6537 * b + 32768) >> 16) + 4096; uu[k] = int((-9674 * r - 18998 * gr + 28672 * b + 32768) >> 16) + 32768;
            vv[k] = int((28672 * r - 24010 * gr - 4662 * b + 32768) >> 16) + 32768;
        }
        reinterpret_cast<uint2*>(row)[g] = make_uint2(uint32_t(min(max(yy[0], 0), 65535)) | (uint32_t(min(max((uu[0] + uu[1]) >> 1, 0), 65535)) << 16),
                                                      uint32_t(min(max(yy[1], 0), 65535)) | (uint32_t(min(max((vv[0] + vv[1]) >> 1, 0), 65535)) << 16));
        break;
    }
    default:
        break;
    }
}

} // namespace dxtex
