// Device-side StoreScanline (DirectXTexConvert.cpp:1629-2533) for the formats this library writes, and the general
// ConvertScanline plan (DirectXTexConvert.cpp:3080-3854) resolved once per call on the host.
//
// The packed stores live in DirectXMath (not vendored by the reference). What is restated
// here is its SSE2 behaviour, the one every x64 build of the reference gets:
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
};

__device__ __forceinline__ Texel apply_plan(Texel t, const ConvertPlan& p)
{
    if (p.srgbIn) { t.r = srgb_to_linear1(t.r); t.g = srgb_to_linear1(t.g); t.b = srgb_to_linear1(t.b); }
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

// XMStoreFloat3SE (DirectXMath >= 3.10; the reference carries the same code for older versions, DirectXTexConvert.cpp:158-191)
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
    // lroundf: round half away from zero
    return (uint32_t(int32_t(roundf(x * scaleR))) & 0x1FFu) | ((uint32_t(int32_t(roundf(y * scaleR))) & 0x1FFu) << 9) |
           ((uint32_t(int32_t(roundf(z * scaleR))) & 0x1FFu) << 18) | (((e - 0x6Fu) & 0x1Fu) << 27);
}

// XMStoreU565 / XMStoreU555 / XMStoreUNibble4 after the reference's scaling (v * 31 | 63 | 15, no bias on x64): clamp to [0, max], round
// to nearest even (_mm_cvtps_epi32).
__device__ __forceinline__ uint32_t store_scaled_rne(float v, float scale)
{
    float s = v * scale;
    s = (s > 0.0f) ? s : 0.0f; s = (s < scale) ? s : scale;
    return uint32_t(int32_t(rintf(s)));
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
    case FMT_R32_FLOAT:
        reinterpret_cast<float*>(row)[x] = t.r;
        break;
    case FMT_R8G8_UNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t(store_ubn2(t.r) | (store_ubn2(t.g) << 8));
        break;
    case FMT_R8G8_SNORM:
        reinterpret_cast<uint16_t*>(row)[x] = uint16_t((uint32_t(store_bn(t.r)) & 0xFF) | ((uint32_t(store_bn(t.g)) & 0xFF) << 8));
        break;
    case FMT_R16_FLOAT:
        reinterpret_cast<uint16_t*>(row)[x] = store_half(t.r);       // std::max(std::min(v, 65504), -65504) (:1891)
        break;
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
    default:
        break;
    }
}

} // namespace dxtex
