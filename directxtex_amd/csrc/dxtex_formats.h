// Host-side format tables for the formats this library handles (DirectXTexUtil.cpp:340-1186 semantics).
#pragma once
#include "dxtex_device.h"
#include <stddef.h>

namespace dxtex
{
enum FmtClass : uint32_t
{
    FC_UNORM = 1, FC_SNORM = 2, FC_FLOAT = 4, FC_BC = 8,
    FC_R = 0x10, FC_G = 0x20, FC_B = 0x40, FC_A = 0x80, FC_SRGB = 0x100,
    FC_POS_ONLY = 0x200,     // CONVF_POS_ONLY: unsigned float formats (R11G11B10_FLOAT, R9G9B9E5_SHAREDEXP)
    FC_UINT = 0x400, FC_SINT = 0x800,      // CONVF_UINT / CONVF_SINT: the value itself travels through the float row
    FC_XR = 0x1000,          // CONVF_XR (R10G10B10_XR_BIAS_A2_UNORM)
    FC_YUV = 0x2000,         // CONVF_YUV: converted to / from RGB inside LoadScanline / StoreScanline
    FC_PACKED = 0x10000,     // CONVF_PACKED: two texels share an element (LoadScanline / StoreScanline unpack / pack)
    FC_GROUP = 0x20000,      // ours: an element holds several texels (the packed formats and R1_UNORM) - stores go through store_group()
    FC_DEPTH = 0x4000, FC_STENCIL = 0x8000,    // CONVF_DEPTH / CONVF_STENCIL: depth in x, stencil in y of the float row; no R / G / B / A bits
};

struct FmtInfo { int format; uint32_t bpp; uint32_t cls; };

inline const FmtInfo* format_info(int format)
{
    static const FmtInfo table[] = {
        { FMT_R32G32B32A32_FLOAT, 128, FC_FLOAT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R16G16B16A16_FLOAT, 64, FC_FLOAT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R8G8B8A8_UNORM, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_R8G8B8A8_UNORM_SRGB, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
        { FMT_B8G8R8A8_UNORM, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_B8G8R8A8_UNORM_SRGB, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
        { FMT_B8G8R8X8_UNORM, 32, FC_UNORM | FC_R | FC_G | FC_B },
        { FMT_B8G8R8X8_UNORM_SRGB, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_SRGB },
        { FMT_R16G16B16A16_UNORM, 64, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_R8G8B8A8_SNORM, 32, FC_SNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_R32G32_FLOAT, 64, FC_FLOAT | FC_R | FC_G },
        { FMT_R16G16_FLOAT, 32, FC_FLOAT | FC_R | FC_G },
        { FMT_R16G16_UNORM, 32, FC_UNORM | FC_R | FC_G },
        { FMT_R8G8_SNORM, 16, FC_SNORM | FC_R | FC_G },
        { FMT_R16_UNORM, 16, FC_UNORM | FC_R },
        { FMT_R8G8_UNORM, 16, FC_UNORM | FC_R | FC_G },
        { FMT_R8_UNORM, 8, FC_UNORM | FC_R },
        { FMT_R8_SNORM, 8, FC_SNORM | FC_R },
        { FMT_A8_UNORM, 8, FC_UNORM | FC_A },
        { FMT_R32_FLOAT, 32, FC_FLOAT | FC_R },
        { FMT_R16_FLOAT, 16, FC_FLOAT | FC_R },
        { FMT_R32G32B32_FLOAT, 96, FC_FLOAT | FC_R | FC_G | FC_B },
        { FMT_R16G16B16A16_SNORM, 64, FC_SNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_R16G16_SNORM, 32, FC_SNORM | FC_R | FC_G },
        { FMT_R16_SNORM, 16, FC_SNORM | FC_R },
        { FMT_R10G10B10A2_UNORM, 32, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_R11G11B10_FLOAT, 32, FC_FLOAT | FC_POS_ONLY | FC_R | FC_G | FC_B },
        { FMT_R9G9B9E5_SHAREDEXP, 32, FC_FLOAT | FC_POS_ONLY | FC_R | FC_G | FC_B },
        { FMT_B5G6R5_UNORM, 16, FC_UNORM | FC_R | FC_G | FC_B },
        { FMT_B5G5R5A1_UNORM, 16, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_B4G4R4A4_UNORM, 16, FC_UNORM | FC_R | FC_G | FC_B | FC_A },
        { FMT_A4B4G4R4_UNORM, 16, FC_UNORM | FC_R | FC_G | FC_B | FC_A },        // WIN11_DXGI_FORMAT_A4B4G4R4_UNORM = 191 (:3046)
        // g_ConvertTable (DirectXTexConvert.cpp:2960-3047): integer, extended-range and 4:4:4 video formats
        { FMT_R32G32B32A32_UINT, 128, FC_UINT | FC_R | FC_G | FC_B | FC_A }, { FMT_R32G32B32A32_SINT, 128, FC_SINT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R32G32B32_UINT, 96, FC_UINT | FC_R | FC_G | FC_B }, { FMT_R32G32B32_SINT, 96, FC_SINT | FC_R | FC_G | FC_B },
        { FMT_R16G16B16A16_UINT, 64, FC_UINT | FC_R | FC_G | FC_B | FC_A }, { FMT_R16G16B16A16_SINT, 64, FC_SINT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R32G32_UINT, 64, FC_UINT | FC_R | FC_G }, { FMT_R32G32_SINT, 64, FC_SINT | FC_R | FC_G },
        { FMT_R10G10B10A2_UINT, 32, FC_UINT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R8G8B8A8_UINT, 32, FC_UINT | FC_R | FC_G | FC_B | FC_A }, { FMT_R8G8B8A8_SINT, 32, FC_SINT | FC_R | FC_G | FC_B | FC_A },
        { FMT_R16G16_UINT, 32, FC_UINT | FC_R | FC_G }, { FMT_R16G16_SINT, 32, FC_SINT | FC_R | FC_G },
        { FMT_R32_UINT, 32, FC_UINT | FC_R }, { FMT_R32_SINT, 32, FC_SINT | FC_R },
        { FMT_R8G8_UINT, 16, FC_UINT | FC_R | FC_G }, { FMT_R8G8_SINT, 16, FC_SINT | FC_R | FC_G },
        { FMT_R16_UINT, 16, FC_UINT | FC_R }, { FMT_R16_SINT, 16, FC_SINT | FC_R },
        { FMT_R8_UINT, 8, FC_UINT | FC_R }, { FMT_R8_SINT, 8, FC_SINT | FC_R },
        { FMT_R10G10B10_XR_BIAS_A2_UNORM, 32, FC_UNORM | FC_XR | FC_R | FC_G | FC_B | FC_A },
        { FMT_AYUV, 32, FC_UNORM | FC_YUV | FC_R | FC_G | FC_B | FC_A }, { FMT_Y410, 32, FC_UNORM | FC_YUV | FC_R | FC_G | FC_B | FC_A },
        { FMT_Y416, 64, FC_UNORM | FC_YUV | FC_R | FC_G | FC_B | FC_A },
        // several texels per element (:3010, :3012-3013, :3038-3040); bpp = bits of an element / texels in it
        { FMT_R1_UNORM, 1, FC_UNORM | FC_R | FC_GROUP },
        { FMT_R8G8_B8G8_UNORM, 16, FC_UNORM | FC_PACKED | FC_GROUP | FC_R | FC_G | FC_B }, { FMT_G8R8_G8B8_UNORM, 16, FC_UNORM | FC_PACKED | FC_GROUP | FC_R | FC_G | FC_B },
        { FMT_YUY2, 16, FC_UNORM | FC_YUV | FC_PACKED | FC_GROUP | FC_R | FC_G | FC_B },
        { FMT_Y210, 32, FC_UNORM | FC_YUV | FC_PACKED | FC_GROUP | FC_R | FC_G | FC_B }, { FMT_Y216, 32, FC_UNORM | FC_YUV | FC_PACKED | FC_GROUP | FC_R | FC_G | FC_B },
        // depth / stencil (:2976, :2990, :2994, :3000)
        { FMT_D32_FLOAT_S8X24_UINT, 64, FC_FLOAT | FC_DEPTH | FC_STENCIL }, { FMT_D32_FLOAT, 32, FC_FLOAT | FC_DEPTH },
        { FMT_D24_UNORM_S8_UINT, 32, FC_UNORM | FC_DEPTH | FC_STENCIL }, { FMT_D16_UNORM, 16, FC_UNORM | FC_DEPTH },
        { FMT_BC1_UNORM, 4, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC1_UNORM_SRGB, 4, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
        { FMT_BC2_UNORM, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC2_UNORM_SRGB, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
        { FMT_BC3_UNORM, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC3_UNORM_SRGB, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
        { FMT_BC4_UNORM, 4, FC_UNORM | FC_BC | FC_R },
        { FMT_BC4_SNORM, 4, FC_SNORM | FC_BC | FC_R },
        { FMT_BC5_UNORM, 8, FC_UNORM | FC_BC | FC_R | FC_G },
        { FMT_BC5_SNORM, 8, FC_SNORM | FC_BC | FC_R | FC_G },
        { FMT_BC6H_UF16, 8, FC_FLOAT | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC6H_SF16, 8, FC_FLOAT | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC7_UNORM, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A },
        { FMT_BC7_UNORM_SRGB, 8, FC_UNORM | FC_BC | FC_R | FC_G | FC_B | FC_A | FC_SRGB },
    };
    for (const FmtInfo& f : table)
        if (f.format == format) return &f;
    return nullptr;
}

inline bool is_bc(int format) { const FmtInfo* f = format_info(format); return f && (f->cls & FC_BC); }
inline size_t bc_block_bytes(int format)
{
    switch (format)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: case FMT_BC4_UNORM: case FMT_BC4_SNORM: return 8;
    default: return is_bc(format) ? 16 : 0;
    }
}
} // namespace dxtex
