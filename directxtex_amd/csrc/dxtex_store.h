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
__device__ __forceinline__ void store_texel(uint8_t* row, uint32_t x, int format, const Texel& t)
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
    default:
        break;
    }
}

} // namespace dxtex
