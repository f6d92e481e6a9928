// Device-side helpers shared by the HIP kernels: DXGI format ids, HRESULT values, and the 4x4 tile
// loader that plays the role of the reference's LoadScanline + partial-block replication +
// ConvertScanline (DirectXTexCompress.cpp:291-343, DirectXTexConvert.cpp:779-1619, :3080-3854).
// gfx950 only. Compile with -ffp-contract=off: BC1-BC5 parity is bit-exact fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace dxtex
{
// Public DXGI numbering (only the formats this library understands).
enum : int
{
    FMT_UNKNOWN = 0,
    FMT_R32G32B32A32_FLOAT = 2, FMT_R32G32B32A32_UINT = 3, FMT_R32G32B32A32_SINT = 4,
    FMT_R32G32B32_FLOAT = 6, FMT_R32G32B32_UINT = 7, FMT_R32G32B32_SINT = 8,
    FMT_R16G16B16A16_FLOAT = 10,
    FMT_R16G16B16A16_UNORM = 11, FMT_R16G16B16A16_UINT = 12,
    FMT_R16G16B16A16_SNORM = 13, FMT_R16G16B16A16_SINT = 14,
    FMT_R32G32_FLOAT = 16, FMT_R32G32_UINT = 17, FMT_R32G32_SINT = 18,
    FMT_D32_FLOAT_S8X24_UINT = 20,
    FMT_R10G10B10A2_UNORM = 24, FMT_R10G10B10A2_UINT = 25,
    FMT_R11G11B10_FLOAT = 26,
    FMT_R8G8B8A8_UNORM = 28,
    FMT_R8G8B8A8_UNORM_SRGB = 29, FMT_R8G8B8A8_UINT = 30,
    FMT_R8G8B8A8_SNORM = 31, FMT_R8G8B8A8_SINT = 32,
    FMT_R16G16_FLOAT = 34,
    FMT_R16G16_UNORM = 35, FMT_R16G16_UINT = 36,
    FMT_R16G16_SNORM = 37, FMT_R16G16_SINT = 38,
    FMT_D32_FLOAT = 40, FMT_R32_FLOAT = 41, FMT_R32_UINT = 42, FMT_R32_SINT = 43,
    FMT_D24_UNORM_S8_UINT = 45,
    FMT_R8G8_UNORM = 49, FMT_R8G8_UINT = 50,
    FMT_R8G8_SNORM = 51, FMT_R8G8_SINT = 52,
    FMT_R16_FLOAT = 54, FMT_D16_UNORM = 55,
    FMT_R16_UNORM = 56, FMT_R16_UINT = 57,
    FMT_R16_SNORM = 58, FMT_R16_SINT = 59,
    FMT_R8_UNORM = 61, FMT_R8_UINT = 62,
    FMT_R8_SNORM = 63, FMT_R8_SINT = 64,
    FMT_A8_UNORM = 65, FMT_R1_UNORM = 66,
    FMT_R9G9B9E5_SHAREDEXP = 67, FMT_R8G8_B8G8_UNORM = 68, FMT_G8R8_G8B8_UNORM = 69,
    FMT_BC1_UNORM = 71, FMT_BC1_UNORM_SRGB = 72,
    FMT_BC2_UNORM = 74, FMT_BC2_UNORM_SRGB = 75,
    FMT_BC3_UNORM = 77, FMT_BC3_UNORM_SRGB = 78,
    FMT_BC4_UNORM = 80, FMT_BC4_SNORM = 81,
    FMT_BC5_UNORM = 83, FMT_BC5_SNORM = 84,
    FMT_B5G6R5_UNORM = 85, FMT_B5G5R5A1_UNORM = 86,
    FMT_B8G8R8A8_UNORM = 87, FMT_B8G8R8X8_UNORM = 88, FMT_R10G10B10_XR_BIAS_A2_UNORM = 89,
    FMT_B8G8R8A8_UNORM_SRGB = 91, FMT_B8G8R8X8_UNORM_SRGB = 93,
    FMT_BC6H_UF16 = 95, FMT_BC6H_SF16 = 96,
    FMT_BC7_UNORM = 98, FMT_BC7_UNORM_SRGB = 99,
    FMT_AYUV = 100, FMT_Y410 = 101, FMT_Y416 = 102, FMT_YUY2 = 107, FMT_Y210 = 108, FMT_Y216 = 109,
    FMT_B4G4R4A4_UNORM = 115,
    FMT_A4B4G4R4_UNORM = 191,
};

// BC_FLAGS == TEX_COMPRESS_FLAGS bit-for-bit (BC.h:30-48, DirectXTex.h:887-917).
enum : uint32_t
{
    BCF_DITHER_RGB = 0x10000,
    BCF_DITHER_A = 0x20000,
    BCF_UNIFORM = 0x40000,
    BCF_USE_3SUBSETS = 0x80000,
    BCF_BC7_QUICK = 0x100000,
};

// What ConvertScanline does to a loaded tile on the compress path (DirectXTexConvert.cpp:3080-3854
// restricted to the in/out pairs Compress can produce; sRGB one-sided conversions are rejected on
// the host). Chosen on the host from (srcFormat, dstFormat).
enum : int
{
    TCV_NONE = 0,
    TCV_SATURATE = 1,        // FLOAT -> UNORM  (:3481-3486)
    TCV_CLAMP_SNORM = 2,     // FLOAT -> SNORM  (:3521-3526)
    TCV_UNORM_TO_SNORM = 3,  // UNORM -> SNORM  v*2 + -1, unfused (:3495-3501)
    TCV_SNORM_TO_UNORM = 4,  // SNORM -> UNORM  v*0.5 + 0.5 (:3457-3463)
    TCV_X2BIAS_TO_UNORM = 5, // FLOAT -> UNORM with TEX_FILTER_FLOAT_X2BIAS: clamp(v,-1,1)*0.5 + 0.5 (:3469-3477)
    TCV_SAT_TO_SNORM = 6,    // positive-only FLOAT (x2 bias) -> SNORM / FLOAT: saturate(v)*2 + -1 (:3506-3515, :3549-3561)
};
enum : int
{
    TSW_NONE = 0,
    TSW_R_TO_RGB = 1,        // R format -> RGB format: (x,x,x,w) (:3670-3680)
    TSW_R_TO_RG = 2,         // R format -> RG format:  (x,x,z,w) (:3683-3693)
    TSW_A_TO_RGB = 3,        // A format -> !A: splat w (:3657-3666)
};

struct SrcView
{
    const uint8_t* pixels;
    uint32_t width, height;
    uint64_t rowPitch;
    int format;
    int tcv;      // TCV_*
    int tsw;      // TSW_*
};

struct Texel { float r, g, b, a; };

// One component of XMLoadFloat3PK: an unsigned small float (5-bit exponent, MB-bit mantissa) widened to fp32. Infinity and NaN keep
// their class (exponent 31), denormals are normalised, everything else is re-biased by 127 - 15.
__device__ __forceinline__ float load_float11(uint32_t bits, int MB)
{
    uint32_t mant = bits & ((1u << MB) - 1u);
    uint32_t expo = bits >> MB;
    if (expo == 0x1Fu) return __uint_as_float(0x7F800000u | (mant << (23 - MB)));
    if (expo == 0)
    {
        if (mant == 0) return 0.0f;
        expo = 1;
        do { --expo; mant <<= 1; } while ((mant & (1u << MB)) == 0);       // normalise
        mant &= (1u << MB) - 1u;
    }
    return __uint_as_float(((expo + 112u) << 23) | (mant << (23 - MB)));
}

// XMLoadUInt4 / XMConvertVectorUIntToFloat, SSE2 path: cvtdq2ps of the low 31 bits, + 2^31 if the top bit was set
__device__ __forceinline__ float load_u32f(uint32_t v)
{
    const float lo = float(int32_t(v & 0x7FFFFFFFu));
    return (v & 0x80000000u) ? lo + 2147483648.0f : lo;
}

// One texel, LoadScanline semantics for the supported formats.
__device__ __forceinline__ Texel load_texel(const uint8_t* row, uint32_t x, int format)
{
    Texel t;
    switch (format)
    {
    case FMT_R8G8B8A8_UNORM:
    case FMT_R8G8B8A8_UNORM_SRGB:
    {
        // XMLoadUByteN4 (DirectXTexConvert.cpp:909-911): float(byte) * (1/255)
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0xFF) * (1.0f / 255.0f);
        t.g = float((v >> 8) & 0xFF) * (1.0f / 255.0f);
        t.b = float((v >> 16) & 0xFF) * (1.0f / 255.0f);
        t.a = float(v >> 24) * (1.0f / 255.0f);
        break;
    }
    case FMT_B8G8R8A8_UNORM:
    case FMT_B8G8R8A8_UNORM_SRGB:
    case FMT_B8G8R8X8_UNORM:
    case FMT_B8G8R8X8_UNORM_SRGB:
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.b = float(v & 0xFF) * (1.0f / 255.0f);
        t.g = float((v >> 8) & 0xFF) * (1.0f / 255.0f);
        t.r = float((v >> 16) & 0xFF) * (1.0f / 255.0f);
        t.a = (format == FMT_B8G8R8X8_UNORM || format == FMT_B8G8R8X8_UNORM_SRGB) ? 1.0f : float(v >> 24) * (1.0f / 255.0f);
        break;
    }
    case FMT_R16G16B16A16_FLOAT:
    {
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = __half2float(__ushort_as_half(uint16_t(v.x & 0xFFFF)));
        t.g = __half2float(__ushort_as_half(uint16_t(v.x >> 16)));
        t.b = __half2float(__ushort_as_half(uint16_t(v.y & 0xFFFF)));
        t.a = __half2float(__ushort_as_half(uint16_t(v.y >> 16)));
        break;
    }
    case FMT_R32G32B32A32_FLOAT:
    {
        const float4 v = reinterpret_cast<const float4*>(row)[x];
        t.r = v.x; t.g = v.y; t.b = v.z; t.a = v.w;
        break;
    }
    case FMT_R16G16B16A16_UNORM:
    {
        // XMLoadUShortN4: float(v) * (1/65535)
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = float(v.x & 0xFFFF) * (1.0f / 65535.0f); t.g = float(v.x >> 16) * (1.0f / 65535.0f);
        t.b = float(v.y & 0xFFFF) * (1.0f / 65535.0f); t.a = float(v.y >> 16) * (1.0f / 65535.0f);
        break;
    }
    case FMT_R8G8B8A8_SNORM:
    {
        // XMLoadByteN4: float(int8) * (1/127), then max with -1
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        const float c[4] = { float(int8_t(v & 0xFF)), float(int8_t((v >> 8) & 0xFF)), float(int8_t((v >> 16) & 0xFF)), float(int8_t(v >> 24)) };
        t.r = fmaxf(c[0] * (1.0f / 127.0f), -1.0f); t.g = fmaxf(c[1] * (1.0f / 127.0f), -1.0f);
        t.b = fmaxf(c[2] * (1.0f / 127.0f), -1.0f); t.a = fmaxf(c[3] * (1.0f / 127.0f), -1.0f);
        break;
    }
    case FMT_R32G32_FLOAT:
    {
        const float2 v = reinterpret_cast<const float2*>(row)[x];
        t.r = v.x; t.g = v.y; t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R16G16_FLOAT:
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = __half2float(__ushort_as_half(uint16_t(v & 0xFFFF))); t.g = __half2float(__ushort_as_half(uint16_t(v >> 16)));
        t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R16G16_UNORM:
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0xFFFF) * (1.0f / 65535.0f); t.g = float(v >> 16) * (1.0f / 65535.0f); t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R8G8_SNORM:
    {
        const uint16_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.r = fmaxf(float(int8_t(v & 0xFF)) * (1.0f / 127.0f), -1.0f); t.g = fmaxf(float(int8_t(v >> 8)) * (1.0f / 127.0f), -1.0f);
        t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_D16_UNORM:              // :1053, shares R16_UNORM's case
    case FMT_R16_UNORM:
        t.r = float(reinterpret_cast<const uint16_t*>(row)[x]) / 65535.0f; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;   // :1054-1065
        break;
    case FMT_R8_UNORM:
        // true division here, unlike the packed loads (DirectXTexConvert.cpp:1113)
        t.r = float(row[x]) / 255.0f; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;
        break;
    case FMT_R8G8_UNORM:
    {
        const uint16_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.r = float(v & 0xFF) * (1.0f / 255.0f); t.g = float(v >> 8) * (1.0f / 255.0f); t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R8_SNORM:
        t.r = float(int8_t(row[x])) / 127.0f; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;   // :1139, no clamp
        break;
    case FMT_D32_FLOAT:              // :937-950, shares R32_FLOAT's case
    case FMT_R32_FLOAT:
        t.r = reinterpret_cast<const float*>(row)[x]; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;
        break;
    case FMT_D32_FLOAT_S8X24_UINT:   // (depth, stencil byte, 0, 1), :844-860
    {
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = __uint_as_float(v.x); t.g = float(v.y & 0xFFu); t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_D24_UNORM_S8_UINT:      // (d / 16777215.f, s, 0, 1): a true division, :982-997
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0xFFFFFFu) / 16777215.0f; t.g = float(v >> 24); t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R16_FLOAT:
        t.r = __half2float(__ushort_as_half(reinterpret_cast<const uint16_t*>(row)[x])); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;
        break;
    case FMT_A8_UNORM:
        t.r = 0.0f; t.g = 0.0f; t.b = 0.0f; t.a = float(row[x]) / 255.0f;   // :1165
        break;
    case FMT_R32G32B32_FLOAT:
    {
        // XMLoadFloat3, w from g_XMIdentityR3 (LOAD_SCANLINE3, :811-812)
        const float* v = reinterpret_cast<const float*>(row) + 3 * size_t(x);
        t.r = v[0]; t.g = v[1]; t.b = v[2]; t.a = 1.0f;
        break;
    }
    case FMT_R16G16B16A16_SNORM:
    {
        // XMLoadShortN4 (:829-830): float(int16) * (1/32767), then max with -1 (the -32768 code)
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = fmaxf(float(int16_t(v.x & 0xFFFF)) * (1.0f / 32767.0f), -1.0f); t.g = fmaxf(float(int16_t(v.x >> 16)) * (1.0f / 32767.0f), -1.0f);
        t.b = fmaxf(float(int16_t(v.y & 0xFFFF)) * (1.0f / 32767.0f), -1.0f); t.a = fmaxf(float(int16_t(v.y >> 16)) * (1.0f / 32767.0f), -1.0f);
        break;
    }
    case FMT_R16G16_SNORM:
    {
        // XMLoadShortN2 (:931-932), z, w from g_XMIdentityR3
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = fmaxf(float(int16_t(v & 0xFFFF)) * (1.0f / 32767.0f), -1.0f); t.g = fmaxf(float(int16_t(v >> 16)) * (1.0f / 32767.0f), -1.0f);
        t.b = 0.0f; t.a = 1.0f;
        break;
    }
    case FMT_R16_SNORM:
        t.r = float(int16_t(reinterpret_cast<const uint16_t*>(row)[x])) / 32767.0f; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;   // true division, no clamp (:1080-1091)
        break;
    case FMT_R10G10B10A2_UNORM:
    {
        // XMLoadUDecN4 (:897-898): the SSE2 path scales the fields in place by 1/1023, 1/(1023*2^10), 1/(1023*2^20), 1/(3*2^30); the
        // powers of two are exact, so every channel is float(field) * float(1/1023) (alpha: * float(1/3))
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0x3FFu) * (1.0f / 1023.0f); t.g = float((v >> 10) & 0x3FFu) * (1.0f / 1023.0f);
        t.b = float((v >> 20) & 0x3FFu) * (1.0f / 1023.0f); t.a = float(v >> 30) * (1.0f / 3.0f);
        break;
    }
    case FMT_R11G11B10_FLOAT:
    {
        // XMLoadFloat3PK (:906-907): 6-bit mantissa / 5-bit exponent (x, y), 5-bit mantissa (z), no sign; w from g_XMIdentityR3
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = load_float11(v & 0x7FFu, 6); t.g = load_float11((v >> 11) & 0x7FFu, 6); t.b = load_float11((v >> 22) & 0x3FFu, 5); t.a = 1.0f;
        break;
    }
    case FMT_R9G9B9E5_SHAREDEXP:
    {
        // XMLoadFloat3SE (:1189-1190): mantissa * 2^(e - 24)
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        const float scale = __uint_as_float(0x33800000u + ((v >> 27) << 23));
        t.r = scale * float(v & 0x1FFu); t.g = scale * float((v >> 9) & 0x1FFu); t.b = scale * float((v >> 18) & 0x1FFu); t.a = 1.0f;
        break;
    }
    case FMT_B5G6R5_UNORM:
    {
        // XMLoadU565 -> (b5, g6, r5) as floats, * (1/31, 1/63, 1/31), swizzled to RGB, w = 1 (:1227-1242)
        const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.b = float(v & 0x1Fu) * (1.0f / 31.0f); t.g = float((v >> 5) & 0x3Fu) * (1.0f / 63.0f); t.r = float((v >> 11) & 0x1Fu) * (1.0f / 31.0f); t.a = 1.0f;
        break;
    }
    case FMT_B5G5R5A1_UNORM:
    {
        // XMLoadU555 (:1244-1258)
        const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.b = float(v & 0x1Fu) * (1.0f / 31.0f); t.g = float((v >> 5) & 0x1Fu) * (1.0f / 31.0f); t.r = float((v >> 10) & 0x1Fu) * (1.0f / 31.0f); t.a = float(v >> 15);
        break;
    }
    case FMT_B4G4R4A4_UNORM:
    {
        // XMLoadUNibble4 * 1/15 (:1511-1525)
        const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.b = float(v & 0xFu) * (1.0f / 15.0f); t.g = float((v >> 4) & 0xFu) * (1.0f / 15.0f); t.r = float((v >> 8) & 0xFu) * (1.0f / 15.0f); t.a = float(v >> 12) * (1.0f / 15.0f);
        break;
    }
    case FMT_A4B4G4R4_UNORM:
    {
        // XMLoadUNibble4 * 1/15, swizzled <3, 2, 1, 0> (:1527-1541)
        const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x];
        t.a = float(v & 0xFu) * (1.0f / 15.0f); t.b = float((v >> 4) & 0xFu) * (1.0f / 15.0f); t.g = float((v >> 8) & 0xFu) * (1.0f / 15.0f); t.r = float(v >> 12) * (1.0f / 15.0f);
        break;
    }
    // ---- integer formats: the VALUE as a float, not normalised. XMLoadUInt* (SSE2): the low 31 bits through cvtdq2ps (round to
    // nearest even), plus 2^31 when the top bit is set - two roundings above 2^31, as there. XMLoadSInt*: cvtdq2ps. The 8- / 16-bit
    // loads are exact. Missing channels come from g_XMIdentityR3 = (0, 0, 0, 1) (LOAD_SCANLINE2 / 3, :750-776).
    case FMT_R32G32B32A32_UINT: { const uint4 v = reinterpret_cast<const uint4*>(row)[x]; t.r = load_u32f(v.x); t.g = load_u32f(v.y); t.b = load_u32f(v.z); t.a = load_u32f(v.w); break; }   // :805-806
    case FMT_R32G32B32A32_SINT: { const int4 v = reinterpret_cast<const int4*>(row)[x]; t.r = float(v.x); t.g = float(v.y); t.b = float(v.z); t.a = float(v.w); break; }                      // :808-809
    case FMT_R32G32B32_UINT: { const uint32_t* v = reinterpret_cast<const uint32_t*>(row) + 3 * size_t(x); t.r = load_u32f(v[0]); t.g = load_u32f(v[1]); t.b = load_u32f(v[2]); t.a = 1.0f; break; }   // :814-815
    case FMT_R32G32B32_SINT: { const int32_t* v = reinterpret_cast<const int32_t*>(row) + 3 * size_t(x); t.r = float(v[0]); t.g = float(v[1]); t.b = float(v[2]); t.a = 1.0f; break; }              // :817-818
    case FMT_R16G16B16A16_UINT:     // XMLoadUShort4, :826-827
    {
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = float(v.x & 0xFFFFu); t.g = float(v.x >> 16); t.b = float(v.y & 0xFFFFu); t.a = float(v.y >> 16);
        break;
    }
    case FMT_R16G16B16A16_SINT:     // XMLoadShort4, :832-833
    {
        const uint2 v = reinterpret_cast<const uint2*>(row)[x];
        t.r = float(int16_t(v.x & 0xFFFFu)); t.g = float(int16_t(v.x >> 16)); t.b = float(int16_t(v.y & 0xFFFFu)); t.a = float(int16_t(v.y >> 16));
        break;
    }
    case FMT_R32G32_UINT: { const uint2 v = reinterpret_cast<const uint2*>(row)[x]; t.r = load_u32f(v.x); t.g = load_u32f(v.y); t.b = 0.0f; t.a = 1.0f; break; }       // :838-839
    case FMT_R32G32_SINT: { const int2 v = reinterpret_cast<const int2*>(row)[x]; t.r = float(v.x); t.g = float(v.y); t.b = 0.0f; t.a = 1.0f; break; }                   // :841-842
    case FMT_R10G10B10A2_UINT:      // XMLoadUDec4, :903-904
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0x3FFu); t.g = float((v >> 10) & 0x3FFu); t.b = float((v >> 20) & 0x3FFu); t.a = float(v >> 30);
        break;
    }
    case FMT_R10G10B10_XR_BIAS_A2_UNORM:    // XMLoadUDecN4_XR, :900-901
    {
        // the SSE2 path (the one an x64 build runs, like XMLoadUDecN4 above) subtracts the bias in place and multiplies by 1/510,
        // 1/(510*2^10), 1/(510*2^20), 1/(3*2^30): float(field - 0x180) * float(1/510) per channel - NOT the scalar path's division,
        // which differs in the last place for some fields (alpha * float(1/3) equals alpha / 3 for all four values)
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(int32_t(v & 0x3FFu) - 0x180) * (1.0f / 510.0f); t.g = float(int32_t((v >> 10) & 0x3FFu) - 0x180) * (1.0f / 510.0f);
        t.b = float(int32_t((v >> 20) & 0x3FFu) - 0x180) * (1.0f / 510.0f); t.a = float(v >> 30) * (1.0f / 3.0f);
        break;
    }
    case FMT_R8G8B8A8_UINT:         // XMLoadUByte4, :913-914
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(v & 0xFFu); t.g = float((v >> 8) & 0xFFu); t.b = float((v >> 16) & 0xFFu); t.a = float(v >> 24);
        break;
    }
    case FMT_R8G8B8A8_SINT:         // XMLoadByte4, :919-920
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x];
        t.r = float(int8_t(v & 0xFFu)); t.g = float(int8_t((v >> 8) & 0xFFu)); t.b = float(int8_t((v >> 16) & 0xFFu)); t.a = float(int8_t(v >> 24));
        break;
    }
    case FMT_R16G16_UINT: { const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x]; t.r = float(v & 0xFFFFu); t.g = float(v >> 16); t.b = 0.0f; t.a = 1.0f; break; }                        // :928-929
    case FMT_R16G16_SINT: { const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x]; t.r = float(int16_t(v & 0xFFFFu)); t.g = float(int16_t(v >> 16)); t.b = 0.0f; t.a = 1.0f; break; }      // :934-935
    case FMT_R32_UINT: t.r = load_u32f(reinterpret_cast<const uint32_t*>(row)[x]); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;      // XMConvertVectorUIntToFloat(v, 0), :952-966
    case FMT_R32_SINT: t.r = float(reinterpret_cast<const int32_t*>(row)[x]); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;           // :968-981
    case FMT_R8G8_UINT: { const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x]; t.r = float(v & 0xFFu); t.g = float(v >> 8); t.b = 0.0f; t.a = 1.0f; break; }                            // :1031-1032
    case FMT_R8G8_SINT: { const uint32_t v = reinterpret_cast<const uint16_t*>(row)[x]; t.r = float(int8_t(v & 0xFFu)); t.g = float(int8_t(v >> 8)); t.b = 0.0f; t.a = 1.0f; break; }            // :1037-1038
    case FMT_R16_UINT: t.r = float(reinterpret_cast<const uint16_t*>(row)[x]); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;          // :1067-1078
    case FMT_R16_SINT: t.r = float(reinterpret_cast<const int16_t*>(row)[x]); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;           // :1093-1104
    case FMT_R8_UINT: t.r = float(row[x]); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;                                              // :1119-1130
    case FMT_R8_SINT: t.r = float(int8_t(row[x])); t.g = 0.0f; t.b = 0.0f; t.a = 1.0f; break;                                      // :1145-1156
    // ---- 4:4:4 video formats: the reference's own fixed-point BT.601 matrices (:1291-1392), clamp, true division by the channel maximum
    // ---- formats whose memory element holds more than one texel: texel x of its element ----
    case FMT_R1_UNORM:               // eight texels a byte, most significant bit first (:1171-1188)
        t.r = ((row[x >> 3] >> (7u - (x & 7u))) & 1u) ? 1.0f : 0.0f; t.g = 0.0f; t.b = 0.0f; t.a = 1.0f;
        break;
    case FMT_R8G8_B8G8_UNORM:        // XMLoadUByteN4 (x * (1/255.f)) of (R, G0, B, G1): texel 0 = (R, G0, B), texel 1 = (R, G1, B) (:1192-1207)
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x >> 1];
        t.r = float(v & 0xFFu) * (1.0f / 255.0f); t.g = float((v >> ((x & 1u) ? 24 : 8)) & 0xFFu) * (1.0f / 255.0f);
        t.b = float((v >> 16) & 0xFFu) * (1.0f / 255.0f); t.a = 1.0f;
        break;
    }
    case FMT_G8R8_G8B8_UNORM:        // (G0, R, G1, B): texel 0 = (R, G0, B), texel 1 = (R, G1, B) (:1209-1225)
    {
        const uint32_t v = reinterpret_cast<const uint32_t*>(row)[x >> 1];
        t.r = float((v >> 8) & 0xFFu) * (1.0f / 255.0f); t.g = float((v >> ((x & 1u) ? 16 : 0)) & 0xFFu) * (1.0f / 255.0f);
        t.b = float(v >> 24) * (1.0f / 255.0f); t.a = 1.0f;
        break;
    }
    case FMT_YUY2:                   // (Y0, U, Y1, V) bytes, BT.601 integer matrix, true divisions by 255 (:1399-1434)
    {
        const uint32_t w = reinterpret_cast<const uint32_t*>(row)[x >> 1];
        const int y = int((w >> ((x & 1u) ? 16 : 0)) & 0xFFu) - 16, u = int((w >> 8) & 0xFFu) - 128, v = int(w >> 24) - 128;
        const int r = (298 * y + 409 * v + 128) >> 8, g = (298 * y - 100 * u - 208 * v + 128) >> 8, b = (298 * y + 516 * u + 128) >> 8;
        t.r = float(min(max(r, 0), 255)) / 255.0f; t.g = float(min(max(g, 0), 255)) / 255.0f; t.b = float(min(max(b, 0), 255)) / 255.0f; t.a = 1.0f;
        break;
    }
    case FMT_Y210:                   // Y216's layout with the low six bits of every word unused (:1436-1472)
    case FMT_Y216:                   // (Y0, U, Y1, V) words (:1474-1510)
    {
        const uint2 w = reinterpret_cast<const uint2*>(row)[x >> 1];
        const uint32_t wy = (x & 1u) ? (w.y & 0xFFFFu) : (w.x & 0xFFFFu), wu = w.x >> 16, wv = w.y >> 16;
        if (format == FMT_Y210)
        {
            const long long y = (long long)(wy >> 6) - 64, u = (long long)(wu >> 6) - 512, v = (long long)(wv >> 6) - 512;
            const int r = int((76533 * y + 104905 * v + 32768) >> 16), g = int((76533 * y - 25747 * u - 53425 * v + 32768) >> 16), b = int((76533 * y + 132590 * u + 32768) >> 16);
            t.r = float(min(max(r, 0), 1023)) / 1023.0f; t.g = float(min(max(g, 0), 1023)) / 1023.0f; t.b = float(min(max(b, 0), 1023)) / 1023.0f;
        }
        else
        {
            const long long y = (long long)wy - 4096, u = (long long)wu - 32768, v = (long long)wv - 32768;
            const int r = int((76607 * y + 105006 * v + 32768) >> 16), g = int((76607 * y - 25772 * u - 53477 * v + 32768) >> 16), b = int((76607 * y + 132718 * u + 32768) >> 16);
            t.r = float(min(max(r, 0), 65535)) / 65535.0f; t.g = float(min(max(g, 0), 65535)) / 65535.0f; t.b = float(min(max(b, 0), 65535)) / 65535.0f;
        }
        t.a = 1.0f;
        break;
    }
    case FMT_AYUV:
    {
        const uint32_t w = reinterpret_cast<const uint32_t*>(row)[x];
        const int v = int(w & 0xFFu) - 128, u = int((w >> 8) & 0xFFu) - 128, y = int((w >> 16) & 0xFFu) - 16;
        const int r = (298 * y + 409 * v + 128) >> 8, g = (298 * y - 100 * u - 208 * v + 128) >> 8, b = (298 * y + 516 * u + 128) >> 8;
        t.r = float(min(max(r, 0), 255)) / 255.0f; t.g = float(min(max(g, 0), 255)) / 255.0f; t.b = float(min(max(b, 0), 255)) / 255.0f;
        t.a = float(w >> 24) / 255.0f;
        break;
    }
    case FMT_Y410:
    {
        const uint32_t w = reinterpret_cast<const uint32_t*>(row)[x];
        const long long u = int(w & 0x3FFu) - 512, y = int((w >> 10) & 0x3FFu) - 64, v = int((w >> 20) & 0x3FFu) - 512;
        const int r = int((76533 * y + 104905 * v + 32768) >> 16), g = int((76533 * y - 25747 * u - 53425 * v + 32768) >> 16), b = int((76533 * y + 132590 * u + 32768) >> 16);
        t.r = float(min(max(r, 0), 1023)) / 1023.0f; t.g = float(min(max(g, 0), 1023)) / 1023.0f; t.b = float(min(max(b, 0), 1023)) / 1023.0f;
        t.a = float(w >> 30) / 3.0f;
        break;
    }
    case FMT_Y416:
    {
        const uint2 w = reinterpret_cast<const uint2*>(row)[x];
        const long long u = (long long)(w.x & 0xFFFFu) - 32768, y = (long long)(w.x >> 16) - 4096, v = (long long)(w.y & 0xFFFFu) - 32768;
        const int a = int(w.y >> 16);
        const int r = int((76607 * y + 105006 * v + 32768) >> 16), g = int((76607 * y - 25772 * u - 53477 * v + 32768) >> 16), b = int((76607 * y + 132718 * u + 32768) >> 16);
        t.r = float(min(max(r, 0), 65535)) / 65535.0f; t.g = float(min(max(g, 0), 65535)) / 65535.0f; t.b = float(min(max(b, 0), 65535)) / 65535.0f;
        t.a = float(min(max(a, 0), 65535)) / 65535.0f;
        break;
    }
    default:
        t.r = t.g = t.b = 0.0f; t.a = 1.0f;
        break;
    }
    return t;
}

__device__ __forceinline__ float tcv1(float v, int tcv)
{
    switch (tcv)
    {
    case TCV_SATURATE: { float m = (v > 0.0f) ? v : 0.0f; return (m < 1.0f) ? m : 1.0f; }   // maxps then minps
    case TCV_CLAMP_SNORM: { float m = (v > -1.0f) ? v : -1.0f; return (m < 1.0f) ? m : 1.0f; }
    case TCV_UNORM_TO_SNORM: return v * 2.0f + -1.0f;
    case TCV_SNORM_TO_UNORM: return v * 0.5f + 0.5f;
    case TCV_X2BIAS_TO_UNORM: { float m = (v > -1.0f) ? v : -1.0f; m = (m < 1.0f) ? m : 1.0f; return m * 0.5f + 0.5f; }
    case TCV_SAT_TO_SNORM: { float m = (v > 0.0f) ? v : 0.0f; m = (m < 1.0f) ? m : 1.0f; return m * 2.0f + -1.0f; }
    default: return v;
    }
}

// XMColorSRGBToRGB / XMColorRGBToSRGB per component. DirectXMath's SSE2 build evaluates XMVectorPow with scalar powf(); pow() in
// double precision rounded once to fp32 is the correctly rounded powf, which is what the host libm returns for every value an
// 8-bit channel can take and for > 99.9 % of arbitrary floats (tests/test_scanline_parity.py pins both).
__device__ __forceinline__ float pow_rn(float x, float y) { return float(pow(double(x), double(y))); }

__device__ __forceinline__ float srgb_to_linear1(float v)
{
    // V = saturate(srgb); V > 0.04045 ? pow((V + 0.055) * (1 / 1.055), 2.4) : V * (1 / 12.92) - DirectXMath multiplies by the reciprocal
    // CONSTANTS (ILinear, Scale), it does not divide (oracle/shim/DirectXMath.h, XMColorSRGBToRGB)
    float s = (v > 0.0f) ? v : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    const float lo = s * (1.0f / 12.92f);
    const float hi = pow_rn((s + 0.055f) * (1.0f / 1.055f), 2.4f);
    return (s > 0.04045f) ? hi : lo;
}

__device__ __forceinline__ float linear_to_srgb1(float v)
{
    // V = saturate(rgb); V < 0.0031308 ? V * 12.92 : 1.055 * pow(V, 1/2.4) - 0.055
    float s = (v > 0.0f) ? v : 0.0f; s = (s < 1.0f) ? s : 1.0f;
    const float lo = s * 12.92f;
    const float hi = 1.055f * pow_rn(s, 1.0f / 2.4f) - 0.055f;
    return (s < 0.0031308f) ? lo : hi;
}

enum : int { TCV_SRGB_TO_LINEAR = 0x100, TCV_LINEAR_TO_SRGB = 0x200 };   // or-ed into SrcView::tcv by the compress path

// ConvertScanline's depth conversions (:3186-3451), one code per step, packed TDP_A | TDP_B << 4 | TDP_C << 8 (0 = none). Depth
// travels in x and stencil in y of the float row (LoadScanline); the steps run in the reference's order: stencil -> alpha, depth ->
// RGB (depth source), or channel -> depth, its range conversion, alpha -> stencil (depth target).
enum : int
{
    TDP_S2A_UNORM = 1,   // w = clamp(y, 0, 255) / 255 (:3196-3209)
    TDP_S2A_SNORM = 2,   // w = clamp(y, 0, 255) / 255 * 2 - 1 (:3210-3224)
    TDP_S2A_RAW = 3,     // w = y (:3225-3235)
    TDP_A2S_UNORM = 4,   // y = w * 255 (:3396-3408)
    TDP_A2S_SNORM = 5,   // y = (w * 0.5 + 0.5) * 255 (:3409-3422)
    TDP_A2S_RAW = 6,     // y = w (:3423-3433)
    TDP_D2RGB_SAT = 1,   // xyz = saturate(x): float depth -> UNORM (:3239-3250)
    TDP_D2RGB_U2S = 2,   // xyz = x * 2 - 1: UNORM depth -> SNORM (:3253-3265)
    TDP_D2RGB_CLAMPS = 3,// xyz = clamp(x, -1, 1): float depth -> SNORM (:3266-3278)
    TDP_D2RGB_RAW = 4,   // xyz = x (:3280-3290)
    TDP_X_FROM_Y = 5, TDP_X_FROM_Z = 6, TDP_X_FROM_W = 7,    // TEX_FILTER_RGB_COPY_GREEN / BLUE / ALPHA -> depth (:3295-3328)
    TDP_X_GRAY = 8,      // x = dot3(v, g_Grayscale): UNORM RGB source without a copy flag (:3330-3343)
    TDP_X_S2U = 1,       // x = x * 0.5 + 0.5 (:3368-3379)
    TDP_X_SAT = 2,       // x = saturate(x) (:3380-3391, :3440-3449)
};

__device__ __forceinline__ Texel apply_depth(Texel t, int op)
{
    const int a = op & 15, b = (op >> 4) & 15, c = (op >> 8) & 15;
    if (a == TDP_S2A_UNORM || a == TDP_S2A_SNORM)
    {
        float s = (t.g > 0.0f) ? t.g : 0.0f; s = (s < 255.0f) ? s : 255.0f;      // XMVectorClamp: max with the lower bound, then min
        s = s / 255.0f;
        t.a = (a == TDP_S2A_SNORM) ? (s * 2.0f + -1.0f) : s;
    }
    else if (a == TDP_S2A_RAW) t.a = t.g;
    switch (b)
    {
    case TDP_D2RGB_SAT: { float v = (t.r > 0.0f) ? t.r : 0.0f; v = (v < 1.0f) ? v : 1.0f; t.r = t.g = t.b = v; break; }
    case TDP_D2RGB_U2S: { const float v = t.r * 2.0f + -1.0f; t.r = t.g = t.b = v; break; }
    case TDP_D2RGB_CLAMPS: { float v = (t.r > -1.0f) ? t.r : -1.0f; v = (v < 1.0f) ? v : 1.0f; t.r = t.g = t.b = v; break; }
    case TDP_D2RGB_RAW: t.g = t.b = t.r; break;
    case TDP_X_FROM_Y: t.r = t.g; break;
    case TDP_X_FROM_Z: t.r = t.b; break;
    case TDP_X_FROM_W: t.r = t.a; break;
    case TDP_X_GRAY: t.r = (t.r * 0.2125f + t.g * 0.7154f) + t.b * 0.0721f; break;
    default: break;
    }
    if (c == TDP_X_S2U) t.r = t.r * 0.5f + 0.5f;
    else if (c == TDP_X_SAT) { float v = (t.r > 0.0f) ? t.r : 0.0f; t.r = (v < 1.0f) ? v : 1.0f; }
    if (a == TDP_A2S_UNORM) t.g = t.a * 255.0f;
    else if (a == TDP_A2S_SNORM) t.g = (t.a * 0.5f + 0.5f) * 255.0f;
    else if (a == TDP_A2S_RAW) t.g = t.a;
    return t;
}

// tsw carries the depth steps of a depth source in its upper bits (tsw = TSW_* | TDP word << 8): Compress from a depth format
__device__ __forceinline__ Texel convert_texel(Texel t, int tcv, int tsw)
{
    if (tcv & TCV_SRGB_TO_LINEAR) { t.r = srgb_to_linear1(t.r); t.g = srgb_to_linear1(t.g); t.b = srgb_to_linear1(t.b); }   // first, :3170-3180
    const int srgbOut = tcv & TCV_LINEAR_TO_SRGB;
    tcv &= 0xFF;
    if (tsw >> 8) t = apply_depth(t, tsw >> 8);
    tsw &= 0xFF;
    if (tcv != TCV_NONE)
    {
        t.r = tcv1(t.r, tcv); t.g = tcv1(t.g, tcv); t.b = tcv1(t.b, tcv); t.a = tcv1(t.a, tcv);
    }
    switch (tsw)
    {
    case TSW_R_TO_RGB: t.g = t.r; t.b = t.r; break;
    case TSW_R_TO_RG: t.g = t.r; break;
    case TSW_A_TO_RGB: t.r = t.a; t.g = t.a; t.b = t.a; break;
    default: break;
    }
    if (srgbOut) { t.r = linear_to_srgb1(t.r); t.g = linear_to_srgb1(t.g); t.b = linear_to_srgb1(t.b); }                    // last, :3843-3853
    return t;
}

// Gathers block (bx, by) as 16 texels, row-major temp[y*4+x], replicating partial blocks exactly like
// the reference (uSrc = {0,0,0,1}: column/row s >= extent copies column/row uSrc[s], applied in
// increasing s so a 1-wide block replicates column 0 everywhere; DirectXTexCompress.cpp:315-341).
struct Tile { float r[16], g[16], b[16], a[16]; };

__device__ __forceinline__ uint32_t replicate_src(uint32_t s, uint32_t extent)
{
    if (s < extent) return s;
    return (s == 3 && extent > 1) ? 1u : 0u;
}

__device__ __forceinline__ void load_tile(const SrcView& src, uint32_t bx, uint32_t by, Tile& t)
{
    const uint32_t x0 = bx * 4, y0 = by * 4;
    const uint32_t pw = min(4u, src.width - x0);
    const uint32_t ph = min(4u, src.height - y0);
    const int format = src.format;
#pragma unroll
    for (uint32_t y = 0; y < 4; ++y)
    {
        const uint32_t sy = y0 + replicate_src(y, ph);
        const uint8_t* row = src.pixels + uint64_t(sy) * src.rowPitch;
        if (pw == 4 && (format == FMT_R8G8B8A8_UNORM || format == FMT_R8G8B8A8_UNORM_SRGB))
        {
            // fast path: one 16-byte load per tile row, coalesced across the wave (lane = block)
            const uint4 v = *reinterpret_cast<const uint4*>(row + uint64_t(x0) * 4);
            const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x)
            {
                Texel px;
                px.r = float(w[x] & 0xFF) * (1.0f / 255.0f);
                px.g = float((w[x] >> 8) & 0xFF) * (1.0f / 255.0f);
                px.b = float((w[x] >> 16) & 0xFF) * (1.0f / 255.0f);
                px.a = float(w[x] >> 24) * (1.0f / 255.0f);
                px = convert_texel(px, src.tcv, src.tsw);
                t.r[y * 4 + x] = px.r; t.g[y * 4 + x] = px.g; t.b[y * 4 + x] = px.b; t.a[y * 4 + x] = px.a;
            }
        }
        else
        {
#pragma unroll
            for (uint32_t x = 0; x < 4; ++x)
            {
                const uint32_t sx = x0 + replicate_src(x, pw);
                Texel px = convert_texel(load_texel(row, sx, format), src.tcv, src.tsw);
                t.r[y * 4 + x] = px.r; t.g[y * 4 + x] = px.g; t.b[y * 4 + x] = px.b; t.a[y * 4 + x] = px.a;
            }
        }
    }
}

} // namespace dxtex
