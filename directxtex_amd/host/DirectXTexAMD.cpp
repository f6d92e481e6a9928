// DirectXTexAMD.cpp - see DirectXTexAMD.h. Validation order and error codes follow the reference functions cited at
// each entry point; the work itself is done by libdxtex_amd.so (HIP kernels) through include/dxtex_amd.h.
#include "DirectXTexAMD.h"
#include "../../include/dxtex_amd.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

namespace DirectXTexAMD
{
namespace
{
    inline dxtex_image View(const Image& i) noexcept
    {
        dxtex_image v;
        v.width = i.width; v.height = i.height; v.format = int32_t(i.format);
        v.rowPitch = i.rowPitch; v.slicePitch = i.slicePitch; v.pixels = i.pixels;
        return v;
    }

    inline bool IsKnown(DXGI_FORMAT fmt) noexcept { return dxtex_bits_per_pixel(int32_t(fmt)) != 0; }

    DXGI_FORMAT DefaultDecompress(DXGI_FORMAT format) noexcept
    {
        switch (format)
        {
        case DXGI_FORMAT_BC1_UNORM: case DXGI_FORMAT_BC2_UNORM: case DXGI_FORMAT_BC3_UNORM: case DXGI_FORMAT_BC7_UNORM: return DXGI_FORMAT_R8G8B8A8_UNORM;
        case DXGI_FORMAT_BC1_UNORM_SRGB: case DXGI_FORMAT_BC2_UNORM_SRGB: case DXGI_FORMAT_BC3_UNORM_SRGB: case DXGI_FORMAT_BC7_UNORM_SRGB: return DXGI_FORMAT_R8G8B8A8_UNORM_SRGB;
        case DXGI_FORMAT_BC4_UNORM: return DXGI_FORMAT_R8_UNORM;
        case DXGI_FORMAT_BC4_SNORM: return DXGI_FORMAT_R8_SNORM;
        case DXGI_FORMAT_BC5_UNORM: return DXGI_FORMAT_R8G8_UNORM;
        case DXGI_FORMAT_BC5_SNORM: return DXGI_FORMAT_R8G8_SNORM;
        case DXGI_FORMAT_BC6H_UF16: case DXGI_FORMAT_BC6H_SF16: return DXGI_FORMAT_R32G32B32A32_FLOAT;
        default: return DXGI_FORMAT_UNKNOWN;
        }
    }

    inline bool ispow2(size_t x) noexcept { return x != 0 && (x & (x - 1)) == 0; }
}

// ---- format facts (DirectXTexUtil.cpp:340-1247), for every DXGI format ------------------------------------------------------
bool IsSupportedOnDevice(DXGI_FORMAT fmt) noexcept { return IsKnown(fmt); }

bool IsCompressed(DXGI_FORMAT fmt) noexcept
{
    const uint32_t f = uint32_t(fmt);
    return (f >= DXGI_FORMAT_BC1_TYPELESS && f <= DXGI_FORMAT_BC5_SNORM) || (f >= DXGI_FORMAT_BC6H_TYPELESS && f <= DXGI_FORMAT_BC7_UNORM_SRGB);
}

bool IsPacked(DXGI_FORMAT fmt) noexcept
{
    return fmt == DXGI_FORMAT_R8G8_B8G8_UNORM || fmt == DXGI_FORMAT_G8R8_G8B8_UNORM || fmt == DXGI_FORMAT_YUY2 || fmt == DXGI_FORMAT_Y210 || fmt == DXGI_FORMAT_Y216;
}

bool IsPlanar(DXGI_FORMAT fmt) noexcept
{
    switch (uint32_t(fmt))
    {
    case DXGI_FORMAT_NV12: case DXGI_FORMAT_P010: case DXGI_FORMAT_P016: case DXGI_FORMAT_420_OPAQUE: case DXGI_FORMAT_NV11:
    case DXGI_FORMAT_P208: case DXGI_FORMAT_V208: case DXGI_FORMAT_V408:
    case 118: case 119: case 120:            // the Xbox depth-stencil planes (D16_UNORM_S8_UINT and its views)
        return true;
    default:
        return false;
    }
}

bool IsPalettized(DXGI_FORMAT fmt) noexcept { return uint32_t(fmt) >= DXGI_FORMAT_AI44 && uint32_t(fmt) <= DXGI_FORMAT_A8P8; }

bool IsSRGB(DXGI_FORMAT fmt) noexcept
{
    switch (fmt)
    {
    case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: case DXGI_FORMAT_BC1_UNORM_SRGB: case DXGI_FORMAT_BC2_UNORM_SRGB: case DXGI_FORMAT_BC3_UNORM_SRGB:
    case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: case DXGI_FORMAT_BC7_UNORM_SRGB:
        return true;
    default:
        return false;
    }
}

bool HasAlpha(DXGI_FORMAT fmt) noexcept
{
    // DirectXTexUtil.cpp:215-276, as runs of the DXGI numbering
    static const struct { uint16_t first, last; } runs[] = {
        { 1, 4 }, { 9, 14 }, { 23, 25 }, { 27, 32 }, { 65, 65 }, { 70, 78 }, { 86, 87 }, { 89, 91 }, { 97, 102 }, { 111, 112 }, { 114, 117 }, { 189, 189 }, { 191, 191 },
    };
    const uint32_t f = uint32_t(fmt);
    for (const auto& r : runs)
        if (f >= r.first && f <= r.last) return true;
    return false;
}

namespace
{
    inline bool InRuns(uint32_t f, const uint16_t (*runs)[2], size_t n) noexcept
    {
        for (size_t i = 0; i < n; ++i)
            if (f >= runs[i][0] && f <= runs[i][1]) return true;
        return false;
    }

    // The typeless families of the DXGI numbering: the typeless id, its typed members, and which member "UNORM" / "FLOAT"
    // means for it (DirectXTexUtil.cpp:1481-1693). 0 = none.
    struct Family { uint16_t typeless, first, last, extra[3], unorm, flt; };
    const Family kFamilies[] = {
        { 1, 2, 4, { 0, 0, 0 }, 0, 2 }, { 5, 6, 8, { 0, 0, 0 }, 0, 6 }, { 9, 10, 14, { 0, 0, 0 }, 11, 10 }, { 15, 16, 18, { 0, 0, 0 }, 0, 16 },
        { 23, 24, 25, { 116, 117, 189 }, 24, 0 },           // + the Xbox 10:10:10 float / snorm variants
        { 27, 28, 32, { 0, 0, 0 }, 28, 0 }, { 33, 34, 38, { 0, 0, 0 }, 35, 34 }, { 39, 40, 43, { 0, 0, 0 }, 0, 41 }, { 48, 49, 52, { 0, 0, 0 }, 49, 0 },
        { 53, 54, 59, { 0, 0, 0 }, 56, 54 }, { 60, 61, 64, { 190, 0, 0 }, 61, 0 },
        { 70, 71, 72, { 0, 0, 0 }, 71, 0 }, { 73, 74, 75, { 0, 0, 0 }, 74, 0 }, { 76, 77, 78, { 0, 0, 0 }, 77, 0 }, { 79, 80, 81, { 0, 0, 0 }, 80, 0 },
        { 82, 83, 84, { 0, 0, 0 }, 83, 0 },
        { 90, 87, 87, { 91, 0, 0 }, 87, 0 }, { 92, 88, 88, { 93, 0, 0 }, 88, 0 },
        { 94, 95, 96, { 0, 0, 0 }, 0, 0 }, { 97, 98, 99, { 0, 0, 0 }, 98, 0 },
    };
}

bool IsVideo(DXGI_FORMAT fmt) noexcept
{
    static const uint16_t runs[][2] = { { 100, 114 }, { 130, 132 } };
    return InRuns(uint32_t(fmt), runs, 2);
}

bool IsDepthStencil(DXGI_FORMAT fmt) noexcept
{
    static const uint16_t runs[][2] = { { 19, 22 }, { 40, 40 }, { 44, 47 }, { 55, 55 }, { 118, 120 } };
    return InRuns(uint32_t(fmt), runs, 5);
}

bool IsBGR(DXGI_FORMAT fmt) noexcept
{
    static const uint16_t runs[][2] = { { 85, 88 }, { 90, 93 }, { 115, 115 }, { 191, 191 } };
    return InRuns(uint32_t(fmt), runs, 4);
}

bool IsTypeless(DXGI_FORMAT fmt, bool partialTypeless) noexcept
{
    const uint32_t f = uint32_t(fmt);
    for (const Family& fam : kFamilies)
        if (f == fam.typeless) return true;
    if (f == DXGI_FORMAT_R32G8X24_TYPELESS || f == DXGI_FORMAT_R24G8_TYPELESS) return true;
    // typed in one plane, typeless in the other
    if (f == DXGI_FORMAT_R32_FLOAT_X8X24_TYPELESS || f == DXGI_FORMAT_X32_TYPELESS_G8X24_UINT || f == DXGI_FORMAT_R24_UNORM_X8_TYPELESS
        || f == DXGI_FORMAT_X24_TYPELESS_G8_UINT || f == 119 || f == 120) return partialTypeless;
    return false;
}

size_t BitsPerColor(DXGI_FORMAT fmt) noexcept
{
    static const struct { uint16_t first, last, bits; } runs[] = {
        { 1, 8, 32 }, { 9, 14, 16 }, { 15, 22, 32 }, { 23, 25, 10 }, { 26, 26, 11 }, { 27, 32, 8 }, { 33, 38, 16 }, { 39, 43, 32 }, { 44, 47, 24 },
        { 48, 52, 8 }, { 53, 59, 16 }, { 60, 65, 8 }, { 66, 66, 1 }, { 67, 67, 14 }, { 68, 69, 8 }, { 70, 78, 6 }, { 79, 84, 8 }, { 85, 85, 6 },
        { 86, 86, 5 }, { 87, 88, 8 }, { 89, 89, 10 }, { 90, 93, 8 }, { 94, 96, 16 }, { 97, 99, 7 }, { 100, 100, 8 }, { 101, 101, 10 }, { 102, 102, 16 },
        { 103, 103, 8 }, { 104, 104, 10 }, { 105, 105, 16 }, { 106, 107, 8 }, { 108, 108, 10 }, { 109, 109, 16 }, { 110, 110, 8 }, { 115, 115, 4 },
        { 116, 117, 10 }, { 118, 120, 16 }, { 130, 132, 8 }, { 189, 189, 10 }, { 190, 191, 4 },
    };
    const uint32_t f = uint32_t(fmt);
    for (const auto& r : runs)
        if (f >= r.first && f <= r.last) return r.bits;
    return 0;
}

size_t BytesPerBlock(DXGI_FORMAT fmt) noexcept { return IsCompressed(fmt) ? (BitsPerPixel(fmt) == 4 ? 8 : 16) : 0; }

DXGI_FORMAT MakeLinear(DXGI_FORMAT fmt) noexcept { return IsSRGB(fmt) ? DXGI_FORMAT(uint32_t(fmt) - 1 - (fmt == DXGI_FORMAT_B8G8R8A8_UNORM_SRGB ? 3 : fmt == DXGI_FORMAT_B8G8R8X8_UNORM_SRGB ? 4 : 0)) : fmt; }

DXGI_FORMAT MakeTypeless(DXGI_FORMAT fmt) noexcept
{
    const uint32_t f = uint32_t(fmt);
    for (const Family& fam : kFamilies)
        if ((f >= fam.first && f <= fam.last) || (f && (f == fam.extra[0] || f == fam.extra[1] || f == fam.extra[2]))) return DXGI_FORMAT(fam.typeless);
    return fmt;
}

DXGI_FORMAT MakeTypelessUNORM(DXGI_FORMAT fmt) noexcept
{
    for (const Family& fam : kFamilies)
        if (uint32_t(fmt) == fam.typeless && fam.unorm) return DXGI_FORMAT(fam.unorm);
    return fmt;
}

DXGI_FORMAT MakeTypelessFLOAT(DXGI_FORMAT fmt) noexcept
{
    for (const Family& fam : kFamilies)
        if (uint32_t(fmt) == fam.typeless && fam.flt) return DXGI_FORMAT(fam.flt);
    return fmt;
}

DXGI_FORMAT MakeSRGB(DXGI_FORMAT fmt) noexcept
{
    switch (fmt)
    {
    case DXGI_FORMAT_R8G8B8A8_UNORM: return DXGI_FORMAT_R8G8B8A8_UNORM_SRGB;
    case DXGI_FORMAT_BC1_UNORM: return DXGI_FORMAT_BC1_UNORM_SRGB;
    case DXGI_FORMAT_BC2_UNORM: return DXGI_FORMAT_BC2_UNORM_SRGB;
    case DXGI_FORMAT_BC3_UNORM: return DXGI_FORMAT_BC3_UNORM_SRGB;
    case DXGI_FORMAT_B8G8R8A8_UNORM: return DXGI_FORMAT_B8G8R8A8_UNORM_SRGB;
    case DXGI_FORMAT_B8G8R8X8_UNORM: return DXGI_FORMAT_B8G8R8X8_UNORM_SRGB;
    case DXGI_FORMAT_BC7_UNORM: return DXGI_FORMAT_BC7_UNORM_SRGB;
    default: return fmt;
    }
}

size_t BitsPerPixel(DXGI_FORMAT fmt) noexcept
{
    // runs of the DXGI numbering that share a size
    static const struct { uint16_t first, last, bits; } runs[] = {
        { 1, 4, 128 }, { 5, 8, 96 }, { 9, 22, 64 }, { 23, 47, 32 }, { 48, 59, 16 }, { 60, 65, 8 }, { 66, 66, 1 }, { 67, 69, 32 },
        { 70, 72, 4 }, { 73, 78, 8 }, { 79, 81, 4 }, { 82, 84, 8 }, { 85, 86, 16 }, { 87, 93, 32 }, { 94, 99, 8 },
        { 100, 101, 32 }, { 102, 102, 64 }, { 103, 103, 12 }, { 104, 105, 24 }, { 106, 106, 12 }, { 107, 107, 32 }, { 108, 109, 64 },
        { 110, 110, 12 }, { 111, 113, 8 }, { 114, 115, 16 },
        { 116, 117, 32 }, { 118, 120, 24 },          // Xbox: 10:10:10 float + A2, D16 + S8 planes
        { 130, 131, 16 }, { 132, 132, 24 },          // P208, V208, V408
        { 189, 189, 32 }, { 190, 190, 8 }, { 191, 191, 16 },          // (Xbox) R10G10B10_SNORM_A2_UNORM, R4G4_UNORM; A4B4G4R4_UNORM
    };
    const uint32_t f = uint32_t(fmt);
    for (const auto& r : runs)
        if (f >= r.first && f <= r.last) return r.bits;
    return 0;
}

HRESULT ComputePitch(DXGI_FORMAT fmt, size_t width, size_t height, size_t& rowPitch, size_t& slicePitch, CP_FLAGS flags) noexcept
{
    // 128-bit arithmetic: a hostile header (32-bit width and height, 128 bits per texel) must not wrap the slice size
    using u128 = unsigned __int128;
    const u128 w = width, h = height;
    u128 pitch = 0, slice = 0;
    if (fmt == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    if (IsCompressed(fmt))
    {
        const u128 bytes = (BitsPerPixel(fmt) == 4) ? 8 : 16;
        if (flags & CP_FLAGS_BAD_DXTN_TAILS)
        {
            // files whose writer rounded the block counts down (DirectXTexUtil.cpp:980-986)
            pitch = std::max<u128>(1, (w >> 2) * bytes);
            slice = std::max<u128>(1, pitch * (h >> 2));
        }
        else
        {
            pitch = std::max<u128>(1, (w + 3) / 4) * bytes;
            slice = pitch * std::max<u128>(1, (h + 3) / 4);
        }
    }
    else if (IsPacked(fmt))
    {
        pitch = ((w + 1) >> 1) * ((fmt == DXGI_FORMAT_Y210 || fmt == DXGI_FORMAT_Y216) ? 8 : 4);
        slice = pitch * h;
    }
    else if (IsPlanar(fmt))
    {
        switch (uint32_t(fmt))
        {
        case DXGI_FORMAT_NV12: case DXGI_FORMAT_420_OPAQUE:
            if (h % 2) return E_INVALIDARG;
            pitch = ((w + 1) >> 1) * 2; slice = pitch * (h + ((h + 1) >> 1));
            break;
        case DXGI_FORMAT_P010: case DXGI_FORMAT_P016:
            if (h % 2) return E_INVALIDARG;
            pitch = ((w + 1) >> 1) * 4; slice = pitch * (h + ((h + 1) >> 1));
            break;
        case 118: case 119: case 120:
            pitch = ((w + 1) >> 1) * 4; slice = pitch * (h + ((h + 1) >> 1));
            break;
        case DXGI_FORMAT_NV11:
            pitch = ((w + 3) >> 2) * 4; slice = pitch * h * 2;
            break;
        case DXGI_FORMAT_P208:
            pitch = ((w + 1) >> 1) * 2; slice = pitch * h * 2;
            break;
        case DXGI_FORMAT_V208:
            if (h % 2) return E_INVALIDARG;
            pitch = w; slice = pitch * (h + (((h + 1) >> 1) * 2));
            break;
        default:        // V408
            pitch = w; slice = pitch * (h + ((h >> 1) * 4));
            break;
        }
    }
    else
    {
        const u128 bpp = (flags & CP_FLAGS_24BPP) ? 24 : (flags & CP_FLAGS_16BPP) ? 16 : (flags & CP_FLAGS_8BPP) ? 8 : BitsPerPixel(fmt);
        if (!bpp) return E_INVALIDARG;
        // row alignment in bits: 4 KiB page, zmm, ymm, paragraph, DWORD, else bytes
        const u128 align = (flags & CP_FLAGS_PAGE4K) ? 32768 : (flags & CP_FLAGS_ZMM) ? 512 : (flags & CP_FLAGS_YMM) ? 256
                             : (flags & CP_FLAGS_PARAGRAPH) ? 128 : (flags & CP_FLAGS_LEGACY_DWORD) ? 32 : 8;
        pitch = ((w * bpp + align - 1) / align) * (align / 8);
        slice = pitch * h;
    }
    if (pitch > UINT64_MAX || slice > UINT64_MAX) { rowPitch = slicePitch = 0; return HRESULT_E_ARITHMETIC_OVERFLOW; }      // the reference checks this on 32-bit builds only (:1172-1178)
    rowPitch = size_t(pitch); slicePitch = size_t(slice);
    return S_OK;
}

size_t ComputeScanlines(DXGI_FORMAT fmt, size_t height) noexcept
{
    if (fmt == DXGI_FORMAT_UNKNOWN) return 0;
    if (IsCompressed(fmt)) return std::max<size_t>(1, (height + 3) / 4);
    switch (uint32_t(fmt))
    {
    case DXGI_FORMAT_NV11: case DXGI_FORMAT_P208: return height * 2;
    case DXGI_FORMAT_V208: return height + (((height + 1) >> 1) * 2);
    case DXGI_FORMAT_V408: return height + ((height >> 1) * 4);
    case DXGI_FORMAT_NV12: case DXGI_FORMAT_P010: case DXGI_FORMAT_P016: case DXGI_FORMAT_420_OPAQUE: case 118: case 119: case 120:
        return height + ((height + 1) >> 1);
    default: return height;
    }
}

bool CalculateMipLevels(size_t width, size_t height, size_t& mipLevels) noexcept
{
    size_t full = 1;
    for (size_t w = width, h = height; w > 1 || h > 1; ++full) { if (w > 1) w >>= 1; if (h > 1) h >>= 1; }
    if (mipLevels > 1) { if (mipLevels > full) return false; }
    else if (mipLevels == 0) mipLevels = full;
    else mipLevels = 1;
    return true;
}

// A 64 KiB tile holds 2^k texels (or 4x4 blocks); the k doublings go round the dimensions, width first. That reproduces the
// table of standard tile shapes (DirectXTexUtil.cpp:1259-1405).
HRESULT ComputeTileShape(DXGI_FORMAT fmt, TEX_DIMENSION dimension, TileShape& tiling) noexcept
{
    tiling = TileShape{ 0, 0, 0 };
    if (IsVideo(fmt) || IsPacked(fmt)) return E_INVALIDARG;
    const size_t bpp = BitsPerPixel(fmt);
    if (!bpp || bpp == 1 || bpp == 24 || bpp == 96) return E_INVALIDARG;
    const bool compressed = IsCompressed(fmt);
    if (dimension == TEX_DIMENSION_TEXTURE1D)
    {
        if (compressed) return E_INVALIDARG;
        tiling = TileShape{ 65536 * 8 / bpp, 1, 1 };
        return S_OK;
    }
    if (dimension != TEX_DIMENSION_TEXTURE2D && dimension != TEX_DIMENSION_TEXTURE3D) return E_INVALIDARG;
    size_t unitBytes = compressed ? BytesPerBlock(fmt) : 1;          // a block, or a texel rounded up to a power of two of bytes
    if (!compressed) { while (unitBytes * 8 < bpp) unitBytes <<= 1; if (unitBytes > 16) return E_INVALIDARG; }
    unsigned k = 0;
    while ((size_t(65536) >> (k + 1)) >= unitBytes) ++k;            // 65536 / unitBytes = 2^k
    const unsigned dims = (dimension == TEX_DIMENSION_TEXTURE3D) ? 3 : 2;
    size_t e[3] = { 1, 1, 1 };
    for (unsigned i = 0; i < k; ++i) e[(i % dims)] <<= 1;             // round-robin: width gets the extra doubling first
    // the doublings were dealt lowest-first; what matters is how many each dimension received
    const size_t unit = compressed ? 4 : 1;
    tiling = TileShape{ e[0] * unit, e[1] * unit, e[2] };
    return S_OK;
}

uint32_t TexMetadata::CalculateSubresource(size_t mip, size_t item) const noexcept { return CalculateSubresource(mip, item, 0); }

uint32_t TexMetadata::CalculateSubresource(size_t mip, size_t item, size_t plane) const noexcept
{
    if (mip >= mipLevels) return uint32_t(-1);
    if (dimension == TEX_DIMENSION_TEXTURE1D || dimension == TEX_DIMENSION_TEXTURE2D)
        return (item < arraySize) ? uint32_t(mip + item * mipLevels + plane * mipLevels * arraySize) : uint32_t(-1);
    if (dimension == TEX_DIMENSION_TEXTURE3D) return (item == 0) ? uint32_t(mip + plane * mipLevels) : uint32_t(-1);       // no arrays of volumes
    return uint32_t(-1);
}

size_t TexMetadata::ComputeIndex(size_t mip, size_t item, size_t slice) const noexcept
{
    if (mip >= mipLevels) return size_t(-1);
    if (dimension == TEX_DIMENSION_TEXTURE3D)
    {
        // a volume: level by level, the level's slices consecutive; no arrays of volumes (DirectXTexUtil.cpp:1714-1738)
        if (item > 0) return size_t(-1);
        size_t index = 0, d = depth;
        for (size_t level = 0; level < mip; ++level) { index += d; if (d > 1) d >>= 1; }
        return (slice < d) ? index + slice : size_t(-1);
    }
    if (slice > 0 || item >= arraySize) return size_t(-1);
    return item * mipLevels + mip;
}

bool CalculateMipLevels3D(size_t width, size_t height, size_t depth, size_t& mipLevels) noexcept
{
    // CountMips3D (DirectXTexMipmaps.cpp:45-66): until all three dimensions reach 1
    size_t maxMips = 1;
    for (size_t w = width, h = height, d = depth; w > 1 || h > 1 || d > 1; ++maxMips) { if (w > 1) w >>= 1; if (h > 1) h >>= 1; if (d > 1) d >>= 1; }
    if (mipLevels > 1) return mipLevels <= maxMips;
    mipLevels = (mipLevels == 0) ? maxMips : 1;
    return true;
}

// ---- ScratchImage --------------------------------------------------------------------------------------------------------
ScratchImage& ScratchImage::operator=(ScratchImage&& o) noexcept
{
    if (this != &o)
    {
        Release();
        m_nimages = o.m_nimages; m_size = o.m_size; m_metadata = o.m_metadata;
        m_images = std::move(o.m_images); m_memory = o.m_memory;
        o.m_nimages = 0; o.m_size = 0; o.m_memory = nullptr;
    }
    return *this;
}

void ScratchImage::Release() noexcept
{
    m_nimages = 0; m_size = 0;
    m_images.reset();
    if (m_memory) { std::free(m_memory); m_memory = nullptr; }
    m_metadata = TexMetadata();
}

// DetermineImageArray / SetupImageArray (DirectXTexImage.cpp:34-268, argument checks :310-340): item-major then mip; a volume goes level by
// level with the level's slices consecutive. `pixels` of every entry is its byte OFFSET in the blob; the caller adds its base pointer.
namespace
{
HRESULT LayoutImages(const TexMetadata& mdata, CP_FLAGS flags, std::unique_ptr<Image[]>& images, size_t& nimagesOut, uint64_t& totalOut, size_t& mipLevelsOut,
                     bool& argumentsAccepted) noexcept
{
    argumentsAccepted = false;
    if (!IsValid(mdata.format)) return E_INVALIDARG;
    if (IsPalettized(mdata.format)) return HRESULT_E_NOT_SUPPORTED;
    size_t mipLevels = mdata.mipLevels;
    switch (mdata.dimension)              // DirectXTexImage.cpp:310-340
    {
    case TEX_DIMENSION_TEXTURE1D:
        if (!mdata.width || mdata.height != 1 || mdata.depth != 1 || !mdata.arraySize) return E_INVALIDARG;
        if (!CalculateMipLevels(mdata.width, 1, mipLevels)) return E_INVALIDARG;
        break;
    case TEX_DIMENSION_TEXTURE2D:
        if (!mdata.width || !mdata.height || mdata.depth != 1 || !mdata.arraySize) return E_INVALIDARG;
        if (mdata.IsCubemap() && (mdata.arraySize % 6) != 0) return E_INVALIDARG;
        if (!CalculateMipLevels(mdata.width, mdata.height, mipLevels)) return E_INVALIDARG;
        break;
    case TEX_DIMENSION_TEXTURE3D:
        if (!mdata.width || !mdata.height || !mdata.depth || mdata.arraySize != 1) return E_INVALIDARG;
        if (!CalculateMipLevels3D(mdata.width, mdata.height, mdata.depth, mipLevels)) return E_INVALIDARG;
        break;
    default:
        return HRESULT_E_NOT_SUPPORTED;
    }
    argumentsAccepted = true;
    const bool volume = mdata.dimension == TEX_DIMENSION_TEXTURE3D;
    size_t nimages = 0;
    uint64_t total = 0;
    const size_t items = volume ? 1 : mdata.arraySize;
    for (size_t item = 0; item < items; ++item)
    {
        size_t w = mdata.width, h = mdata.height, d = volume ? mdata.depth : 1;
        for (size_t level = 0; level < mipLevels; ++level)
        {
            size_t rp, sp;
            const HRESULT hr = ComputePitch(mdata.format, w, h, rp, sp, flags);
            if (FAILED(hr)) return hr;
            if (d && sp > (UINT64_MAX - total) / d) return E_OUTOFMEMORY;          // more than an address space: no wrap-around
            total += uint64_t(sp) * d;
            nimages += d;
            if (h > 1) h >>= 1;
            if (w > 1) w >>= 1;
            if (d > 1) d >>= 1;
        }
    }
    images.reset(new (std::nothrow) Image[nimages]);
    if (!images) return E_OUTOFMEMORY;
    size_t at = 0, index = 0;
    for (size_t item = 0; item < items; ++item)
    {
        size_t w = mdata.width, h = mdata.height, d = volume ? mdata.depth : 1;
        for (size_t level = 0; level < mipLevels; ++level)
        {
            for (size_t slice = 0; slice < d; ++slice, ++index)
            {
                Image& im = images[index];
                im.width = w; im.height = h; im.format = mdata.format;
                ComputePitch(mdata.format, w, h, im.rowPitch, im.slicePitch, flags);
                im.pixels = reinterpret_cast<uint8_t*>(at);
                at += im.slicePitch;
            }
            if (h > 1) h >>= 1;
            if (w > 1) w >>= 1;
            if (d > 1) d >>= 1;
        }
    }
    nimagesOut = nimages; totalOut = total; mipLevelsOut = mipLevels;
    return S_OK;
}
}

HRESULT ScratchImage::Initialize(const TexMetadata& mdata, CP_FLAGS flags) noexcept
{
    std::unique_ptr<Image[]> images;
    size_t nimages = 0, mipLevels = 0;
    uint64_t total = 0;
    bool accepted = false;
    const HRESULT hr = LayoutImages(mdata, flags, images, nimages, total, mipLevels, accepted);
    if (FAILED(hr)) { if (accepted) Release(); return hr; }      // the argument checks come before the Release (:310-342)
    Release();
    m_metadata = mdata;
    m_metadata.mipLevels = mipLevels;
    const size_t bytes = (size_t(total) + 15) & ~size_t(15);
    m_memory = static_cast<uint8_t*>(std::aligned_alloc(16, std::max<size_t>(bytes, 16)));
    if (!m_memory) { Release(); return E_OUTOFMEMORY; }
    std::memset(m_memory, 0, std::max<size_t>(bytes, 16));        // zero-filled like the reference (DirectXTexImage.cpp:376)
    m_size = size_t(total);
    m_nimages = nimages;
    m_images = std::move(images);
    for (size_t i = 0; i < nimages; ++i) m_images[i].pixels = m_memory + reinterpret_cast<size_t>(m_images[i].pixels);
    return S_OK;
}

HRESULT ScratchImage::Initialize3D(DXGI_FORMAT fmt, size_t width, size_t height, size_t depth, size_t mipLevels, CP_FLAGS flags) noexcept
{
    if (depth > INT16_MAX) return E_INVALIDARG;                  // DirectXTexImage.cpp:464-465
    TexMetadata m;
    m.width = width; m.height = height; m.depth = depth; m.arraySize = 1; m.mipLevels = mipLevels;
    m.format = fmt; m.dimension = TEX_DIMENSION_TEXTURE3D;
    return Initialize(m, flags);
}

HRESULT ScratchImage::Initialize2D(DXGI_FORMAT fmt, size_t width, size_t height, size_t arraySize, size_t mipLevels, CP_FLAGS flags) noexcept
{
    TexMetadata m;
    m.width = width; m.height = height; m.depth = 1; m.arraySize = arraySize; m.mipLevels = mipLevels;
    m.format = fmt; m.dimension = TEX_DIMENSION_TEXTURE2D;
    return Initialize(m, flags);
}

HRESULT ScratchImage::Initialize1D(DXGI_FORMAT fmt, size_t length, size_t arraySize, size_t mipLevels, CP_FLAGS flags) noexcept
{
    TexMetadata m;
    m.width = length; m.height = 1; m.depth = 1; m.arraySize = arraySize; m.mipLevels = mipLevels;
    m.format = fmt; m.dimension = TEX_DIMENSION_TEXTURE1D;
    return Initialize(m, flags);
}

HRESULT ScratchImage::InitializeCube(DXGI_FORMAT fmt, size_t width, size_t height, size_t nCubes, size_t mipLevels, CP_FLAGS flags) noexcept
{
    if (!nCubes) return E_INVALIDARG;
    TexMetadata m;
    m.width = width; m.height = height; m.depth = 1; m.arraySize = nCubes * 6; m.mipLevels = mipLevels;
    m.format = fmt; m.dimension = TEX_DIMENSION_TEXTURE2D; m.miscFlags = TEX_MISC_TEXTURECUBE;
    return Initialize(m, flags);
}

// the pixels of caller images copied in, row by row over min(pitches) (DirectXTexImage.cpp:534-723)
namespace
{
    HRESULT CopyIn(const Image* src, const Image* dst, size_t count) noexcept
    {
        const size_t rows = ComputeScanlines(src[0].format, src[0].height);
        if (!rows) return HRESULT(0x8000FFFF);           // E_UNEXPECTED
        for (size_t i = 0; i < count; ++i)
        {
            if (!src[i].pixels || !dst[i].pixels) return E_POINTER;
            const size_t n = std::min(src[i].rowPitch, dst[i].rowPitch);
            for (size_t y = 0; y < rows; ++y) std::memcpy(dst[i].pixels + y * dst[i].rowPitch, src[i].pixels + y * src[i].rowPitch, n);
        }
        return S_OK;
    }
    HRESULT SameShape(const Image* images, size_t n) noexcept
    {
        for (size_t i = 0; i < n; ++i)
        {
            if (!images[i].pixels) return E_POINTER;
            if (images[i].format != images[0].format || images[i].width != images[0].width || images[i].height != images[0].height) return E_FAIL;
        }
        return S_OK;
    }
}

HRESULT ScratchImage::InitializeFromImage(const Image& src, bool allow1D, CP_FLAGS flags) noexcept
{
    const HRESULT hr = (src.height > 1 || !allow1D) ? Initialize2D(src.format, src.width, src.height, 1, 1, flags) : Initialize1D(src.format, src.width, 1, 1, flags);
    if (FAILED(hr)) return hr;
    return CopyIn(&src, m_images.get(), 1);
}

HRESULT ScratchImage::InitializeArrayFromImages(const Image* images, size_t nImages, bool allow1D, CP_FLAGS flags) noexcept
{
    if (!images || !nImages) return E_INVALIDARG;
    HRESULT hr = SameShape(images, nImages);
    if (FAILED(hr)) return hr;
    hr = (images[0].height > 1 || !allow1D) ? Initialize2D(images[0].format, images[0].width, images[0].height, nImages, 1, flags)
                                            : Initialize1D(images[0].format, images[0].width, nImages, 1, flags);
    if (FAILED(hr)) return hr;
    return CopyIn(images, m_images.get(), nImages);
}

HRESULT ScratchImage::InitializeCubeFromImages(const Image* images, size_t nImages, CP_FLAGS flags) noexcept
{
    if (!images || !nImages || (nImages % 6) != 0) return E_INVALIDARG;
    const HRESULT hr = InitializeArrayFromImages(images, nImages, false, flags);
    if (FAILED(hr)) return hr;
    m_metadata.miscFlags |= TEX_MISC_TEXTURECUBE;
    return S_OK;
}

HRESULT ScratchImage::Initialize3DFromImages(const Image* images, size_t depth, CP_FLAGS flags) noexcept
{
    if (!images || !depth || depth > INT16_MAX) return E_INVALIDARG;
    HRESULT hr = SameShape(images, depth);
    if (FAILED(hr)) return hr;
    hr = Initialize3D(images[0].format, images[0].width, images[0].height, depth, 1, flags);
    if (FAILED(hr)) return hr;
    return CopyIn(images, m_images.get(), depth);
}

bool ScratchImage::OverrideFormat(DXGI_FORMAT f) noexcept
{
    if (!m_images || !IsValid(f) || IsPlanar(f) || IsPalettized(f)) return false;
    for (size_t i = 0; i < m_nimages; ++i) m_images[i].format = f;
    m_metadata.format = f;
    return true;
}

const Image* ScratchImage::GetImage(size_t mip, size_t item, size_t slice) const noexcept
{
    const size_t i = m_metadata.ComputeIndex(mip, item, slice);
    return (i < m_nimages) ? &m_images[i] : nullptr;
}

// ---- Device ----------------------------------------------------------------------------------------------------------------
Device::~Device() { if (m_ctx) dxtex_ctx_destroy(m_ctx); }
HRESULT Device::Create(int hipDevice) noexcept
{
    if (m_ctx) { dxtex_ctx_destroy(m_ctx); m_ctx = nullptr; }
    return dxtex_ctx_create(hipDevice, &m_ctx);
}
HRESULT Device::Prepare(size_t width, size_t height, DXGI_FORMAT srcFormat, DXGI_FORMAT bcFormat, TEX_COMPRESS_FLAGS flags, size_t count) noexcept
{
    if (!m_ctx) return E_POINTER;
    return dxtex_ctx_prepare(m_ctx, width, height, int32_t(srcFormat), int32_t(bcFormat), uint32_t(flags), count, nullptr);
}
const char* Device::LastError() const noexcept { return m_ctx ? dxtex_ctx_last_error(m_ctx) : "no device"; }

// ---- the two memory spaces of the array entry points ---------------------------------------------------------------------------
// Host images (ScratchImage out; the C ABI's host-pointer functions stage through the device) and device-resident images
// (DeviceScratchImage out; the *_device functions, stream-ordered). Each array entry point below is ONE template over the space, so the
// validation, its order and the error codes are the same code in both.
namespace
{
struct HostSpace
{
    using Out = ScratchImage;
    static HRESULT Init(Device&, Out& out, const TexMetadata& m) noexcept { return out.Initialize(m); }
    static HRESULT CompressMany(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, size_t n, uint32_t f, float t) noexcept { return dxtex_compress_many(c, s, d, n, f, t); }
    static HRESULT Decompress(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d) noexcept { return dxtex_decompress(c, s, d); }
    static HRESULT Mips(dxtex_ctx* c, const dxtex_image* l, size_t n, uint32_t f) noexcept { return dxtex_generate_mips(c, l, n, f); }
    static HRESULT Mips3D(dxtex_ctx* c, const dxtex_volume* l, size_t n, uint32_t f) noexcept { return dxtex_generate_mips3d(c, l, n, f); }
    static HRESULT Resize(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f) noexcept { return dxtex_resize(c, s, d, f); }
    static HRESULT Convert(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f, float t) noexcept { return dxtex_convert(c, s, d, f, t); }
    static HRESULT PMAlpha(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f) noexcept { return dxtex_premultiply_alpha(c, s, d, f); }
    static HRESULT Coverage(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, size_t n, float r) noexcept { return dxtex_scale_mips_alpha_for_coverage(c, s, d, n, r); }
    static HRESULT CopyRows(dxtex_ctx*, uint8_t* dst, size_t dstPitch, const uint8_t* src, size_t srcPitch, size_t rowBytes, size_t rows) noexcept
    {
        for (size_t y = 0; y < rows; ++y) std::memcpy(dst + y * dstPitch, src + y * srcPitch, rowBytes);
        return S_OK;
    }
};
struct DeviceSpace
{
    using Out = DeviceScratchImage;
    static HRESULT Init(Device& dev, Out& out, const TexMetadata& m) noexcept { return out.Initialize(dev, m); }
    static HRESULT CompressMany(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, size_t n, uint32_t f, float t) noexcept { return dxtex_compress_many_device(c, s, d, n, f, t); }
    static HRESULT Decompress(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d) noexcept { return dxtex_decompress_device(c, s, d); }
    static HRESULT Mips(dxtex_ctx* c, const dxtex_image* l, size_t n, uint32_t f) noexcept { return dxtex_generate_mips_device(c, l, n, f); }
    static HRESULT Mips3D(dxtex_ctx* c, const dxtex_volume* l, size_t n, uint32_t f) noexcept { return dxtex_generate_mips3d_device(c, l, n, f); }
    static HRESULT Resize(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f) noexcept { return dxtex_resize_device(c, s, d, f); }
    static HRESULT Convert(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f, float t) noexcept { return dxtex_convert_device(c, s, d, f, t); }
    static HRESULT PMAlpha(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, uint32_t f) noexcept { return dxtex_premultiply_alpha_device(c, s, d, f); }
    static HRESULT Coverage(dxtex_ctx* c, const dxtex_image* s, const dxtex_image* d, size_t n, float r) noexcept { return dxtex_scale_mips_alpha_for_coverage_device(c, s, d, n, r); }
    static HRESULT CopyRows(dxtex_ctx* c, uint8_t* dst, size_t dstPitch, const uint8_t* src, size_t srcPitch, size_t rowBytes, size_t rows) noexcept
    {
        return dxtex_copy_rows_device(c, dst, dstPitch, src, srcPitch, rowBytes, rows);
    }
};
}

// ---- Compress (CompressEx, DirectXTexCompress.cpp:664-850) --------------------------------------------------------------------
namespace
{
// Progress and cancellation for one image (the role of the per-row callbacks of CompressBC, DirectXTexCompress.cpp:113-121):
// blocks are independent, so the image is submitted as bands of whole block rows - large enough to fill the GPU - and the
// callback is asked between bands. The bytes written are those of the one-submission path.
// Device::SetProgressBands overrides the band sizes (the tests use it to get several bands out of a small image).
size_t BandSize(size_t chosen, size_t def) noexcept { return chosen ? chosen : def; }
constexpr size_t kBandBlocks = 262144;

HRESULT CompressBands(Device& device, const Image& src, const Image& dst, const CompressOptions& options, const StatusCallback& statusCallback)
{
    const size_t nbW = std::max<size_t>(1, (src.width + 3) / 4), nbH = std::max<size_t>(1, (src.height + 3) / 4);
    const size_t bandRows = std::max<size_t>(1, BandSize(device.ProgressBandBlocks(), kBandBlocks) / nbW);          // block rows per band
    for (size_t by = 0; by < nbH; by += bandRows)
    {
        const size_t y = by * 4;
        if (by && !statusCallback(y, src.height)) return E_ABORT;            // (0, height) was reported by the caller
        const size_t rows = std::min(src.height - y, bandRows * 4);
        dxtex_image s = View(src), d = View(dst);
        s.pixels = src.pixels + y * src.rowPitch; s.height = rows; s.slicePitch = src.rowPitch * rows;
        d.pixels = dst.pixels + by * dst.rowPitch; d.height = rows; d.slicePitch = dst.rowPitch * ((rows + 3) / 4);
        const HRESULT hr = dxtex_compress(device.Get(), &s, &d, uint32_t(options.flags), options.threshold);
        if (FAILED(hr)) return hr;
    }
    return S_OK;
}
}

HRESULT CompressEx(Device& device, const Image& srcImage, DXGI_FORMAT format, const CompressOptions& options, ScratchImage& image,
                   StatusCallback statusCallback)
{
    if (!device) return E_POINTER;
    if (IsCompressed(srcImage.format) || !IsCompressed(format) || srcImage.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    if (!IsKnown(srcImage.format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;             // typeless, planar, palettised, ... (:674-676)
    HRESULT hr = image.Initialize2D(format, srcImage.width, srcImage.height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* img = image.GetImage(0, 0, 0);
    if (!img) { image.Release(); return E_POINTER; }
    if (statusCallback)
    {
        if (!srcImage.pixels) { image.Release(); return E_POINTER; }
        if (!statusCallback(0, img->height)) { image.Release(); return E_ABORT; }                    // :690-697
        hr = CompressBands(device, srcImage, *img, options, statusCallback);
        if (FAILED(hr)) { image.Release(); return hr; }
        if (!statusCallback(img->height, img->height)) { image.Release(); return E_ABORT; }          // :719-726
        return S_OK;
    }
    const dxtex_image s = View(srcImage), d = View(*img);
    hr = dxtex_compress(device.Get(), &s, &d, uint32_t(options.flags), options.threshold);
    if (FAILED(hr)) image.Release();
    return hr;
}

namespace
{
template <class Space>
HRESULT CompressArrayT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                       const CompressOptions& options, typename Space::Out& cImages, const StatusCallback& statusCallback)
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages) return E_INVALIDARG;
    if (IsCompressed(metadata.format) || !IsCompressed(format)) return E_INVALIDARG;
    if (!IsKnown(metadata.format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    cImages.Release();
    if constexpr (std::is_same_v<Space, HostSpace>)
    {
        if (statusCallback && nimages == 1 && !metadata.IsVolumemap() && metadata.mipLevels == 1 && metadata.arraySize == 1)
            return CompressEx(device, srcImages[0], format, options, cImages, statusCallback);         // progress inside the image, :753-764
    }
    TexMetadata m2 = metadata;
    m2.format = format;
    HRESULT hr = Space::Init(device, cImages, m2);
    if (FAILED(hr)) return hr;
    if (nimages != cImages.GetImageCount()) { cImages.Release(); return E_FAIL; }
    const Image* dest = cImages.GetImages();
    std::vector<dxtex_image> s(nimages), d(nimages);
    for (size_t i = 0; i < nimages; ++i)
    {
        if (srcImages[i].format != metadata.format) { cImages.Release(); return E_FAIL; }
        if (srcImages[i].width != dest[i].width || srcImages[i].height != dest[i].height) { cImages.Release(); return E_FAIL; }     // :800-804
        s[i] = View(srcImages[i]); d[i] = View(dest[i]);
    }
    if (!statusCallback)
    {
        hr = Space::CompressMany(device.Get(), s.data(), d.data(), nimages, uint32_t(options.flags), options.threshold);   // the whole array in one submission
        if (FAILED(hr)) cImages.Release();
        return hr;
    }
    // with a callback the images go one at a time so that a cancel stops the work where the reference's would (:785-843)
    if (!statusCallback(0, nimages)) { cImages.Release(); return E_ABORT; }
    for (size_t i = 0; i < nimages; ++i)
    {
        hr = dxtex_compress(device.Get(), &s[i], &d[i], uint32_t(options.flags), options.threshold);
        if (FAILED(hr)) { cImages.Release(); return hr; }
        if (!statusCallback(i, nimages)) { cImages.Release(); return E_ABORT; }
    }
    if (!statusCallback(nimages, nimages)) { cImages.Release(); return E_ABORT; }
    return S_OK;
}
}

HRESULT CompressEx(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                   const CompressOptions& options, ScratchImage& cImages, StatusCallback statusCallback)
{
    return CompressArrayT<HostSpace>(device, srcImages, nimages, metadata, format, options, cImages, statusCallback);
}

// Compress = CompressEx without a callback (DirectXTexCompress.cpp:632-661). The exception barrier is for the noexcept
// contract only: nothing on the callback-free path throws short of bad_alloc.
HRESULT Compress(Device& device, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& image) noexcept
{
    CompressOptions options = {};
    options.flags = compress; options.threshold = threshold;
    try { return CompressEx(device, srcImage, format, options, image, nullptr); }
    catch (...) { image.Release(); return E_OUTOFMEMORY; }
}

HRESULT Compress(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                 TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImages) noexcept
{
    CompressOptions options = {};
    options.flags = compress; options.threshold = threshold;
    try { return CompressEx(device, srcImages, nimages, metadata, format, options, cImages, nullptr); }
    catch (...) { cImages.Release(); return E_OUTOFMEMORY; }
}

// ---- Decompress (DirectXTexCompress.cpp:852-979) -------------------------------------------------------------------------------
HRESULT Decompress(Device& device, const Image& cImage, DXGI_FORMAT format, ScratchImage& image) noexcept
{
    if (!device) return E_POINTER;
    if (!IsCompressed(cImage.format) || IsCompressed(format)) return E_INVALIDARG;
    if (!IsKnown(cImage.format)) return HRESULT_E_NOT_SUPPORTED;
    if (format == DXGI_FORMAT_UNKNOWN)
    {
        format = DefaultDecompress(cImage.format);
        if (format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    }
    else if (!IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    HRESULT hr = image.Initialize2D(format, cImage.width, cImage.height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* img = image.GetImage(0, 0, 0);
    if (!img) { image.Release(); return E_POINTER; }
    const dxtex_image s = View(cImage), d = View(*img);
    hr = dxtex_decompress(device.Get(), &s, &d);
    if (FAILED(hr)) image.Release();
    return hr;
}

namespace
{
template <class Space>
HRESULT DecompressArrayT(Device& device, const Image* cImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format, typename Space::Out& images) noexcept
{
    if (!device) return E_POINTER;
    if (!cImages || !nimages) return E_INVALIDARG;
    if (!IsCompressed(metadata.format) || IsCompressed(format)) return E_INVALIDARG;
    if (!IsKnown(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    if (format == DXGI_FORMAT_UNKNOWN)
    {
        format = DefaultDecompress(cImages[0].format);
        if (format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    }
    else if (!IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    images.Release();
    TexMetadata m2 = metadata;
    m2.format = format;
    HRESULT hr = Space::Init(device, images, m2);
    if (FAILED(hr)) return hr;
    if (nimages != images.GetImageCount()) { images.Release(); return E_FAIL; }
    const Image* dest = images.GetImages();
    for (size_t i = 0; i < nimages; ++i)
    {
        if (cImages[i].format != metadata.format) { images.Release(); return E_FAIL; }
        const dxtex_image s = View(cImages[i]), d = View(dest[i]);
        hr = Space::Decompress(device.Get(), &s, &d);
        if (FAILED(hr)) { images.Release(); return hr; }
    }
    return S_OK;
}
}

HRESULT Decompress(Device& device, const Image* cImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format, ScratchImage& images) noexcept
{
    return DecompressArrayT<HostSpace>(device, cImages, nimages, metadata, format, images);
}

// ---- GenerateMipMaps (DirectXTexMipmaps.cpp:2828-3247) ---------------------------------------------------------------------------
HRESULT GenerateMipMaps(Device& device, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept
{
    if (!device) return E_POINTER;
    if (baseImage.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    if (!baseImage.pixels) return E_POINTER;
    if (!CalculateMipLevels(baseImage.width, baseImage.height, levels)) return E_INVALIDARG;
    if (levels <= 1) return E_INVALIDARG;
    if (IsCompressed(baseImage.format) || !IsKnown(baseImage.format)) return HRESULT_E_NOT_SUPPORTED;

    HRESULT hr = mipChain.Initialize2D(baseImage.format, baseImage.width, baseImage.height, 1, levels);
    if (FAILED(hr)) return hr;
    // Setup2DMips (:851-904): the base image goes to the top of the chain
    const Image* top = mipChain.GetImage(0, 0, 0);
    for (size_t y = 0; y < baseImage.height; ++y)
        std::memcpy(top->pixels + y * top->rowPitch, baseImage.pixels + y * baseImage.rowPitch, std::min(top->rowPitch, baseImage.rowPitch));
    std::vector<dxtex_image> views(levels);
    for (size_t l = 0; l < levels; ++l) views[l] = View(*mipChain.GetImage(l, 0, 0));
    hr = dxtex_generate_mips(device.Get(), views.data(), levels, uint32_t(filter));
    if (FAILED(hr)) mipChain.Release();
    return hr;
}

// ---- one image over several devices (in-process strong scaling; dxtex_compress_multi / dxtex_generate_mips_multi) -------------------------
namespace
{
bool Contexts(Device* const* devices, size_t ndevices, std::vector<dxtex_ctx*>& out)
{
    if (!devices || !ndevices) return false;
    for (size_t i = 0; i < ndevices; ++i)
    {
        if (!devices[i] || !*devices[i]) return false;
        out.push_back(devices[i]->Get());
    }
    return true;
}
}

HRESULT Compress(Device* const* devices, size_t ndevices, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& image) noexcept
{
    try
    {
        std::vector<dxtex_ctx*> ctxs;
        if (!Contexts(devices, ndevices, ctxs)) return E_POINTER;
        if (IsCompressed(srcImage.format) || !IsCompressed(format) || srcImage.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
        if (!srcImage.pixels) return E_POINTER;                      // as the single-device overload (DirectXTexCompress.cpp:613-614)
        if (!IsKnown(srcImage.format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
        HRESULT hr = image.Initialize2D(format, srcImage.width, srcImage.height, 1, 1);
        if (FAILED(hr)) return hr;
        const Image* img = image.GetImage(0, 0, 0);
        if (!img) { image.Release(); return E_POINTER; }
        const dxtex_image s = View(srcImage), d = View(*img);
        hr = dxtex_compress_multi(ctxs.data(), ctxs.size(), &s, &d, uint32_t(compress), threshold);
        if (FAILED(hr)) image.Release();
        return hr;
    }
    catch (...) { image.Release(); return E_OUTOFMEMORY; }
}

HRESULT GenerateMipMaps(Device* const* devices, size_t ndevices, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept
{
    try
    {
        std::vector<dxtex_ctx*> ctxs;
        if (!Contexts(devices, ndevices, ctxs)) return E_POINTER;
        if (baseImage.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
        if (!baseImage.pixels) return E_POINTER;
        if (!CalculateMipLevels(baseImage.width, baseImage.height, levels)) return E_INVALIDARG;
        if (levels <= 1) return E_INVALIDARG;
        if (IsCompressed(baseImage.format) || !IsKnown(baseImage.format)) return HRESULT_E_NOT_SUPPORTED;
        HRESULT hr = mipChain.Initialize2D(baseImage.format, baseImage.width, baseImage.height, 1, levels);
        if (FAILED(hr)) return hr;
        const Image* top = mipChain.GetImage(0, 0, 0);
        for (size_t y = 0; y < baseImage.height; ++y)
            std::memcpy(top->pixels + y * top->rowPitch, baseImage.pixels + y * baseImage.rowPitch, std::min(top->rowPitch, baseImage.rowPitch));
        std::vector<dxtex_image> views(levels);
        for (size_t l = 0; l < levels; ++l) views[l] = View(*mipChain.GetImage(l, 0, 0));
        hr = dxtex_generate_mips_multi(ctxs.data(), ctxs.size(), views.data(), levels, uint32_t(filter));
        if (FAILED(hr)) mipChain.Release();
        return hr;
    }
    catch (...) { mipChain.Release(); return E_OUTOFMEMORY; }
}

namespace
{
template <class Space>
HRESULT GenerateMipMapsArrayT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_FILTER_FLAGS filter,
                              size_t levels, typename Space::Out& mipChain) noexcept
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages || metadata.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    if (metadata.dimension == TEX_DIMENSION_TEXTURE3D || IsCompressed(metadata.format) || !IsKnown(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    if (!CalculateMipLevels(metadata.width, metadata.height, levels)) return E_INVALIDARG;
    if (levels <= 1) return E_INVALIDARG;
    // base images = mip 0 of every array item (:3160-3188)
    std::vector<const Image*> base;
    for (size_t item = 0; item < metadata.arraySize; ++item)
    {
        const size_t index = metadata.ComputeIndex(0, item, 0);
        if (index >= nimages) return E_FAIL;
        const Image& src = srcImages[index];
        if (!src.pixels) return E_POINTER;
        if (src.format != metadata.format || src.width != metadata.width || src.height != metadata.height) return E_FAIL;
        base.push_back(&src);
    }
    TexMetadata m2 = metadata;
    m2.mipLevels = levels;
    HRESULT hr = Space::Init(device, mipChain, m2);
    if (FAILED(hr)) return hr;
    for (size_t item = 0; item < metadata.arraySize; ++item)
    {
        const Image* top = mipChain.GetImage(0, item, 0);
        hr = Space::CopyRows(device.Get(), top->pixels, top->rowPitch, base[item]->pixels, base[item]->rowPitch, std::min(top->rowPitch, base[item]->rowPitch), top->height);
        if (FAILED(hr)) { mipChain.Release(); return hr; }
        std::vector<dxtex_image> views(levels);
        for (size_t l = 0; l < levels; ++l) views[l] = View(*mipChain.GetImage(l, item, 0));
        hr = Space::Mips(device.Get(), views.data(), levels, uint32_t(filter));
        if (FAILED(hr)) { mipChain.Release(); return hr; }
    }
    return S_OK;
}
}

HRESULT GenerateMipMaps(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_FILTER_FLAGS filter,
                        size_t levels, ScratchImage& mipChain) noexcept
{
    return GenerateMipMapsArrayT<HostSpace>(device, srcImages, nimages, metadata, filter, levels, mipChain);
}

// ---- GenerateMipMaps3D (DirectXTexMipmaps.cpp:3254-3361) -------------------------------------------------------------------------------
namespace
{
template <class Space>
HRESULT GenerateMipMaps3DT(Device& device, const Image* baseImages, size_t depth, TEX_FILTER_FLAGS filter, size_t levels, typename Space::Out& mipChain) noexcept
{
    if (!device) return E_POINTER;
    if (!baseImages || !depth || depth > INT16_MAX) return E_INVALIDARG;
    const DXGI_FORMAT format = baseImages[0].format;
    const size_t width = baseImages[0].width, height = baseImages[0].height;
    if (!CalculateMipLevels3D(width, height, depth, levels)) return E_INVALIDARG;
    if (levels <= 1) return E_INVALIDARG;
    for (size_t slice = 0; slice < depth; ++slice)
    {
        if (!baseImages[slice].pixels) return E_POINTER;
        if (baseImages[slice].format != format || baseImages[slice].width != width || baseImages[slice].height != height) return E_FAIL;
    }
    if (IsCompressed(format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    // Setup3DMips (:1608-1664): the base slices go to the top level
    TexMetadata m3;
    m3.width = width; m3.height = height; m3.depth = depth; m3.arraySize = 1; m3.mipLevels = levels; m3.format = format; m3.dimension = TEX_DIMENSION_TEXTURE3D;
    HRESULT hr = Space::Init(device, mipChain, m3);               // ScratchImage::Initialize3D (depth <= INT16_MAX was checked above)
    if (FAILED(hr)) return hr;
    for (size_t slice = 0; slice < depth; ++slice)
    {
        const Image* dest = mipChain.GetImage(0, 0, slice);
        if (!dest) { mipChain.Release(); return E_POINTER; }
        hr = Space::CopyRows(device.Get(), dest->pixels, dest->rowPitch, baseImages[slice].pixels, baseImages[slice].rowPitch,
                             std::min(dest->rowPitch, baseImages[slice].rowPitch), height);
        if (FAILED(hr)) { mipChain.Release(); return hr; }
    }
    std::vector<dxtex_volume> lv(levels);
    size_t d = depth;
    for (size_t l = 0; l < levels; ++l)
    {
        const Image* first = mipChain.GetImage(l, 0, 0);
        lv[l] = dxtex_volume{ first->width, first->height, d, int32_t(first->format), first->rowPitch, first->slicePitch, first->pixels };
        if (d > 1) d >>= 1;
    }
    hr = Space::Mips3D(device.Get(), lv.data(), lv.size(), uint32_t(filter));
    if (FAILED(hr)) mipChain.Release();
    return hr;
}
}

HRESULT GenerateMipMaps3D(Device& device, const Image* baseImages, size_t depth, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept
{
    return GenerateMipMaps3DT<HostSpace>(device, baseImages, depth, filter, levels, mipChain);
}

HRESULT GenerateMipMaps3D(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_FILTER_FLAGS filter, size_t levels,
                          ScratchImage& mipChain) noexcept
{
    // the complex overload (:3364-3480): the metadata must describe a volume, the base slices are its first `depth` images
    if (!srcImages || !nimages || metadata.dimension != TEX_DIMENSION_TEXTURE3D) return E_INVALIDARG;
    if (metadata.depth > nimages) return E_FAIL;
    for (size_t slice = 0; slice < metadata.depth; ++slice)
        if (srcImages[slice].format != metadata.format || srcImages[slice].width != metadata.width || srcImages[slice].height != metadata.height) return E_FAIL;
    return GenerateMipMaps3D(device, srcImages, metadata.depth, filter, levels, mipChain);
}

// ---- Resize (DirectXTexResize.cpp:854-930) ----------------------------------------------------------------------------------------
HRESULT Resize(Device& device, const Image& srcImage, size_t width, size_t height, TEX_FILTER_FLAGS filter, ScratchImage& image) noexcept
{
    if (!device) return E_POINTER;
    if (width == 0 || height == 0) return E_INVALIDARG;
    if (srcImage.width > UINT32_MAX || srcImage.height > UINT32_MAX || width > UINT32_MAX || height > UINT32_MAX) return E_INVALIDARG;
    if (!srcImage.pixels) return E_POINTER;
    if (IsCompressed(srcImage.format) || !IsKnown(srcImage.format)) return HRESULT_E_NOT_SUPPORTED;
    HRESULT hr = image.Initialize2D(srcImage.format, width, height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* rimage = image.GetImage(0, 0, 0);
    if (!rimage) return E_POINTER;
    const dxtex_image s = View(srcImage), d = View(*rimage);
    hr = dxtex_resize(device.Get(), &s, &d, uint32_t(filter));
    if (FAILED(hr)) image.Release();
    return hr;
}

// Resize (complex), DirectXTexResize.cpp:942-1103: only mip 0 of every array item / depth slice is resized, the result has
// one mip level.
namespace
{
template <class Space>
HRESULT ResizeArrayT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t width, size_t height,
                     TEX_FILTER_FLAGS filter, typename Space::Out& result) noexcept
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages || width == 0 || height == 0) return E_INVALIDARG;
    if (width > UINT32_MAX || height > UINT32_MAX || metadata.width > UINT32_MAX || metadata.height > UINT32_MAX) return E_INVALIDARG;
    if (IsCompressed(metadata.format) || !IsKnown(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    TexMetadata mdata2 = metadata;
    mdata2.width = width;
    mdata2.height = height;
    mdata2.mipLevels = 1;
    HRESULT hr = Space::Init(device, result, mdata2);
    if (FAILED(hr)) return hr;
    const bool volume = metadata.dimension == TEX_DIMENSION_TEXTURE3D;
    const size_t count = volume ? metadata.depth : metadata.arraySize;
    for (size_t i = 0; i < count; ++i)
    {
        const size_t srcIndex = volume ? metadata.ComputeIndex(0, 0, i) : metadata.ComputeIndex(0, i, 0);
        if (srcIndex >= nimages) { result.Release(); return E_FAIL; }
        const Image& srcimg = srcImages[srcIndex];
        const Image* destimg = volume ? result.GetImage(0, 0, i) : result.GetImage(0, i, 0);
        if (!destimg || !srcimg.pixels) { result.Release(); return E_POINTER; }
        if (srcimg.format != metadata.format) { result.Release(); return E_FAIL; }
        if (srcimg.width > UINT32_MAX || srcimg.height > UINT32_MAX) { result.Release(); return E_FAIL; }
        const dxtex_image s = View(srcimg), d = View(*destimg);
        hr = Space::Resize(device.Get(), &s, &d, uint32_t(filter));
        if (FAILED(hr)) { result.Release(); return hr; }
    }
    return S_OK;
}
}

HRESULT Resize(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t width, size_t height,
               TEX_FILTER_FLAGS filter, ScratchImage& result) noexcept
{
    return ResizeArrayT<HostSpace>(device, srcImages, nimages, metadata, width, height, filter, result);
}

// ---- Convert (ConvertEx, DirectXTexConvert.cpp:5107-5176) ---------------------------------------------------------------------------
namespace
{
// rows are independent in Convert (dither is out of scope), so progress / cancel works on bands of rows (the per-row
// callbacks of ConvertCustom, DirectXTexConvert.cpp:4834-4896)
constexpr size_t kBandTexels = size_t(1) << 24;

HRESULT ConvertBands(Device& device, const Image& src, const Image& dst, const ConvertOptions& options, const StatusCallback& statusCallback)
{
    const size_t bandRows = std::max<size_t>(1, BandSize(device.ProgressBandTexels(), kBandTexels) / std::max<size_t>(1, src.width));
    for (size_t y = 0; y < src.height; y += bandRows)
    {
        if (y && !statusCallback(y, src.height)) return E_ABORT;
        const size_t rows = std::min(src.height - y, bandRows);
        dxtex_image s = View(src), d = View(dst);
        s.pixels = src.pixels + y * src.rowPitch; s.height = rows; s.slicePitch = src.rowPitch * rows;
        d.pixels = dst.pixels + y * dst.rowPitch; d.height = rows; d.slicePitch = dst.rowPitch * rows;
        const HRESULT hr = dxtex_convert(device.Get(), &s, &d, uint32_t(options.filter), options.threshold);
        if (FAILED(hr)) return hr;
    }
    return S_OK;
}
}

HRESULT ConvertEx(Device& device, const Image& srcImage, DXGI_FORMAT format, const ConvertOptions& options, ScratchImage& image,
                  StatusCallback statusCallback)
{
    if (!device) return E_POINTER;
    if (srcImage.format == format || format == DXGI_FORMAT_UNKNOWN || srcImage.format == DXGI_FORMAT_UNKNOWN) return E_INVALIDARG;
    if (!srcImage.pixels) return E_POINTER;
    if (IsCompressed(srcImage.format) || IsCompressed(format) || !IsKnown(srcImage.format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    if (srcImage.width > UINT32_MAX || srcImage.height > UINT32_MAX) return E_INVALIDARG;
    HRESULT hr = image.Initialize2D(format, srcImage.width, srcImage.height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* rimage = image.GetImage(0, 0, 0);
    if (!rimage) { image.Release(); return E_POINTER; }
    if (statusCallback)
    {
        if (!statusCallback(0, rimage->height)) { image.Release(); return E_ABORT; }                  // :5141-5148
        hr = ConvertBands(device, srcImage, *rimage, options, statusCallback);
        if (FAILED(hr)) { image.Release(); return hr; }
        if (!statusCallback(rimage->height, rimage->height)) { image.Release(); return E_ABORT; }     // :5166-5173
        return S_OK;
    }
    const dxtex_image s = View(srcImage), d = View(*rimage);
    hr = dxtex_convert(device.Get(), &s, &d, uint32_t(options.filter), options.threshold);
    if (FAILED(hr)) image.Release();
    return hr;
}

// ConvertEx (complex), DirectXTexConvert.cpp:5198-5405: every image of the set (all items, mips, slices) is converted.
namespace
{
template <class Space>
HRESULT ConvertArrayT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                      const ConvertOptions& options, typename Space::Out& result, const StatusCallback& statusCallback)
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages || metadata.format == format || format == DXGI_FORMAT_UNKNOWN || metadata.format == DXGI_FORMAT_UNKNOWN)
        return E_INVALIDARG;
    if (IsCompressed(metadata.format) || IsCompressed(format) || !IsKnown(metadata.format) || !IsKnown(format)) return HRESULT_E_NOT_SUPPORTED;
    if (metadata.width > UINT32_MAX || metadata.height > UINT32_MAX) return E_INVALIDARG;
    if constexpr (std::is_same_v<Space, HostSpace>)
    {
        if (statusCallback && nimages == 1 && !metadata.IsVolumemap() && metadata.mipLevels == 1 && metadata.arraySize == 1)
            return ConvertEx(device, srcImages[0], format, options, result, statusCallback);              // :5224-5235
    }
    TexMetadata mdata2 = metadata;
    mdata2.format = format;
    HRESULT hr = Space::Init(device, result, mdata2);
    if (FAILED(hr)) return hr;
    if (nimages != result.GetImageCount()) { result.Release(); return E_FAIL; }
    const Image* dest = result.GetImages();
    if (!dest) { result.Release(); return E_POINTER; }
    if (statusCallback && !statusCallback(0, nimages)) { result.Release(); return E_ABORT; }
    for (size_t i = 0; i < nimages; ++i)
    {
        const Image& src = srcImages[i];
        if (src.format != metadata.format) { result.Release(); return E_FAIL; }
        if (src.width > UINT32_MAX || src.height > UINT32_MAX) { result.Release(); return E_FAIL; }
        if (src.width != dest[i].width || src.height != dest[i].height) { result.Release(); return E_FAIL; }
        if (!src.pixels) { result.Release(); return E_POINTER; }
        const dxtex_image s = View(src), d = View(dest[i]);
        hr = Space::Convert(device.Get(), &s, &d, uint32_t(options.filter), options.threshold);
        if (FAILED(hr)) { result.Release(); return hr; }
        if (statusCallback && !statusCallback(i, nimages)) { result.Release(); return E_ABORT; }
    }
    if (statusCallback && !statusCallback(nimages, nimages)) { result.Release(); return E_ABORT; }
    return S_OK;
}
}

HRESULT ConvertEx(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                  const ConvertOptions& options, ScratchImage& result, StatusCallback statusCallback)
{
    return ConvertArrayT<HostSpace>(device, srcImages, nimages, metadata, format, options, result, statusCallback);
}

HRESULT Convert(Device& device, const Image& srcImage, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, ScratchImage& image) noexcept
{
    ConvertOptions options = {};
    options.filter = filter; options.threshold = threshold;
    try { return ConvertEx(device, srcImage, format, options, image, nullptr); }
    catch (...) { image.Release(); return E_OUTOFMEMORY; }
}

HRESULT Convert(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                TEX_FILTER_FLAGS filter, float threshold, ScratchImage& result) noexcept
{
    ConvertOptions options = {};
    options.filter = filter; options.threshold = threshold;
    try { return ConvertEx(device, srcImages, nimages, metadata, format, options, result, nullptr); }
    catch (...) { result.Release(); return E_OUTOFMEMORY; }
}

// ---- PremultiplyAlpha (DirectXTexPMAlpha.cpp:214-341) ----------------------------------------------------------------------------------
HRESULT PremultiplyAlpha(Device& device, const Image& srcImage, TEX_PMALPHA_FLAGS flags, ScratchImage& image) noexcept
{
    if (!device) return E_POINTER;
    if (!srcImage.pixels) return E_POINTER;
    if (IsCompressed(srcImage.format) || !IsKnown(srcImage.format) || !HasAlpha(srcImage.format)) return HRESULT_E_NOT_SUPPORTED;
    if (srcImage.width > UINT32_MAX || srcImage.height > UINT32_MAX) return E_INVALIDARG;
    HRESULT hr = image.Initialize2D(srcImage.format, srcImage.width, srcImage.height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* rimage = image.GetImage(0, 0, 0);
    if (!rimage) { image.Release(); return E_POINTER; }
    const dxtex_image s = View(srcImage), d = View(*rimage);
    hr = dxtex_premultiply_alpha(device.Get(), &s, &d, uint32_t(flags));
    if (FAILED(hr)) image.Release();
    return hr;
}

namespace
{
template <class Space>
HRESULT PremultiplyAlphaArrayT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_PMALPHA_FLAGS flags, typename Space::Out& result) noexcept
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages) return E_INVALIDARG;
    if (IsCompressed(metadata.format) || !IsKnown(metadata.format) || !HasAlpha(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    if (metadata.width > UINT32_MAX || metadata.height > UINT32_MAX) return E_INVALIDARG;
    const bool isPM = metadata.IsPMAlpha();
    if (isPM != ((flags & TEX_PMALPHA_REVERSE) != 0)) return E_FAIL;                      // :283-284
    TexMetadata mdata2 = metadata;
    mdata2.SetAlphaMode((flags & TEX_PMALPHA_REVERSE) ? TEX_ALPHA_MODE_STRAIGHT : TEX_ALPHA_MODE_PREMULTIPLIED);
    HRESULT hr = Space::Init(device, result, mdata2);
    if (FAILED(hr)) return hr;
    if (nimages != result.GetImageCount()) { result.Release(); return E_FAIL; }
    const Image* dest = result.GetImages();
    if (!dest) { result.Release(); return E_POINTER; }
    for (size_t i = 0; i < nimages; ++i)
    {
        const Image& src = srcImages[i];
        if (src.format != metadata.format) { result.Release(); return E_FAIL; }
        if (src.width > UINT32_MAX || src.height > UINT32_MAX) { result.Release(); return E_FAIL; }
        if (src.width != dest[i].width || src.height != dest[i].height) { result.Release(); return E_FAIL; }
        const dxtex_image s = View(src), d = View(dest[i]);
        hr = Space::PMAlpha(device.Get(), &s, &d, uint32_t(flags));
        if (FAILED(hr)) { result.Release(); return hr; }
    }
    return S_OK;
}
}

HRESULT PremultiplyAlpha(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_PMALPHA_FLAGS flags, ScratchImage& result) noexcept
{
    return PremultiplyAlphaArrayT<HostSpace>(device, srcImages, nimages, metadata, flags, result);
}

// ---- ScaleMipMapsAlphaForCoverage (DirectXTexMipmaps.cpp:3483-3556) ------------------------------------------------------------------
namespace
{
template <class Space>
HRESULT ScaleMipMapsAlphaForCoverageT(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t item,
                                      float alphaReference, typename Space::Out& mipChain) noexcept
{
    if (!device) return E_POINTER;
    if (!srcImages || !nimages || metadata.format == DXGI_FORMAT_UNKNOWN || nimages > metadata.mipLevels || !mipChain.GetImages()) return E_INVALIDARG;
    if (metadata.dimension == TEX_DIMENSION_TEXTURE3D || IsCompressed(metadata.format) || !IsKnown(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    if (srcImages[0].format != metadata.format || srcImages[0].width != metadata.width || srcImages[0].height != metadata.height) return E_FAIL;
    if (nimages < metadata.mipLevels) return E_FAIL;                                       // :3532-3533, reached at the first missing level
    std::vector<dxtex_image> s(metadata.mipLevels), d(metadata.mipLevels);
    for (size_t level = 0; level < metadata.mipLevels; ++level)
    {
        const Image* dst = mipChain.GetImage(level, item, 0);
        if (!dst || !dst->pixels || !srcImages[level].pixels) return E_POINTER;
        s[level] = View(srcImages[level]); d[level] = View(*dst);
    }
    return Space::Coverage(device.Get(), s.data(), d.data(), s.size(), alphaReference);
}
}

HRESULT ScaleMipMapsAlphaForCoverage(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t item,
                                     float alphaReference, ScratchImage& mipChain) noexcept
{
    return ScaleMipMapsAlphaForCoverageT<HostSpace>(device, srcImages, nimages, metadata, item, alphaReference, mipChain);
}

// ---- ComputeMSE (DirectXTexMisc.cpp:181-260): compressed inputs are decompressed first -----------------------------------------------
HRESULT ComputeMSE(Device& device, const Image& image1, const Image& image2, float& mse, float* mseV) noexcept
{
    if (!device) return E_POINTER;
    if (!image1.pixels || !image2.pixels) return E_POINTER;
    if (image1.width != image2.width || image1.height != image2.height) return E_INVALIDARG;
    ScratchImage t1, t2;
    const Image* a = &image1;
    const Image* b = &image2;
    if (IsCompressed(image1.format))
    {
        const HRESULT hr = Decompress(device, image1, DXGI_FORMAT_UNKNOWN, t1);
        if (FAILED(hr)) return hr;
        a = t1.GetImage(0, 0, 0);
    }
    if (IsCompressed(image2.format))
    {
        const HRESULT hr = Decompress(device, image2, DXGI_FORMAT_UNKNOWN, t2);
        if (FAILED(hr)) return hr;
        b = t2.GetImage(0, 0, 0);
    }
    // stage both on the device, reduce there
    void* da = nullptr; void* db = nullptr;
    const size_t na = a->rowPitch * a->height, nb = b->rowPitch * b->height;
    HRESULT hr = dxtex_device_alloc(device.Get(), na, &da);
    if (FAILED(hr)) return hr;
    hr = dxtex_device_alloc(device.Get(), nb, &db);
    if (FAILED(hr)) { dxtex_device_free(device.Get(), da); return hr; }
    double v[4] = { 0, 0, 0, 0 };
    hr = dxtex_memcpy_h2d(device.Get(), da, a->pixels, na);
    if (SUCCEEDED(hr)) hr = dxtex_memcpy_h2d(device.Get(), db, b->pixels, nb);
    if (SUCCEEDED(hr))
    {
        dxtex_image va = View(*a), vb = View(*b);
        va.pixels = static_cast<uint8_t*>(da); vb.pixels = static_cast<uint8_t*>(db);
        hr = dxtex_compute_mse_device(device.Get(), &va, &vb, v);
    }
    dxtex_device_free(device.Get(), da);
    dxtex_device_free(device.Get(), db);
    if (FAILED(hr)) return hr;
    if (mseV) for (int c = 0; c < 4; ++c) mseV[c] = float(v[c]);
    mse = float(v[0]) + float(v[1]) + float(v[2]) + float(v[3]);
    return S_OK;
}

// ---- DeviceScratchImage: a ScratchImage whose blob lives in HBM ---------------------------------------------------------------------
DeviceScratchImage& DeviceScratchImage::operator=(DeviceScratchImage&& o) noexcept
{
    if (this != &o)
    {
        Release();
        m_device = o.m_device; m_nimages = o.m_nimages; m_size = o.m_size; m_metadata = o.m_metadata;
        m_images = std::move(o.m_images); m_memory = o.m_memory;
        o.m_device = nullptr; o.m_nimages = 0; o.m_size = 0; o.m_memory = nullptr; o.m_metadata = TexMetadata();
    }
    return *this;
}

void DeviceScratchImage::Release() noexcept
{
    m_nimages = 0; m_size = 0;
    m_images.reset();
    if (m_memory && m_device && *m_device) dxtex_device_free(m_device->Get(), m_memory);      // waits for the work that still uses it
    m_memory = nullptr; m_device = nullptr;
    m_metadata = TexMetadata();
}

HRESULT DeviceScratchImage::Initialize(Device& device, const TexMetadata& mdata, CP_FLAGS flags) noexcept
{
    if (!device) return E_POINTER;
    std::unique_ptr<Image[]> images;
    size_t nimages = 0, mipLevels = 0;
    uint64_t total = 0;
    bool accepted = false;
    HRESULT hr = LayoutImages(mdata, flags, images, nimages, total, mipLevels, accepted);
    if (FAILED(hr)) { if (accepted) Release(); return hr; }
    Release();
    const size_t bytes = std::max<size_t>((size_t(total) + 15) & ~size_t(15), 16);
    void* mem = nullptr;
    hr = dxtex_device_alloc(device.Get(), bytes, &mem);
    if (FAILED(hr)) return hr;
    hr = dxtex_device_memset(device.Get(), mem, 0, bytes);                  // zero-filled like ScratchImage (DirectXTexImage.cpp:376)
    if (FAILED(hr)) { dxtex_device_free(device.Get(), mem); return hr; }
    m_device = &device;
    m_memory = static_cast<uint8_t*>(mem);
    m_metadata = mdata;
    m_metadata.mipLevels = mipLevels;
    m_size = size_t(total);
    m_nimages = nimages;
    m_images = std::move(images);
    for (size_t i = 0; i < nimages; ++i) m_images[i].pixels = m_memory + reinterpret_cast<size_t>(m_images[i].pixels);
    return S_OK;
}

HRESULT DeviceScratchImage::Upload(Device& device, const ScratchImage& src) noexcept
{
    if (!src.GetPixels() || !src.GetImageCount()) return E_INVALIDARG;
    HRESULT hr = Initialize(device, src.GetMetadata());
    if (FAILED(hr)) return hr;
    // A ScratchImage built with CP_FLAGS other than NONE has other pitches: go image by image then.
    bool sameLayout = src.GetImageCount() == m_nimages && src.GetPixelsSize() == m_size;
    for (size_t i = 0; sameLayout && i < m_nimages; ++i)
        sameLayout = src.GetImages()[i].rowPitch == m_images[i].rowPitch && src.GetImages()[i].slicePitch == m_images[i].slicePitch &&
                     size_t(src.GetImages()[i].pixels - src.GetPixels()) == size_t(m_images[i].pixels - m_memory);
    if (!sameLayout) return Upload(device, src.GetImages(), src.GetImageCount(), src.GetMetadata());
    hr = dxtex_memcpy_h2d(device.Get(), m_memory, src.GetPixels(), m_size);
    if (FAILED(hr)) Release();
    return hr;
}

HRESULT DeviceScratchImage::Upload(Device& device, const Image* images, size_t nimages, const TexMetadata& metadata) noexcept
{
    if (!images || !nimages) return E_INVALIDARG;
    HRESULT hr = Initialize(device, metadata);
    if (FAILED(hr)) return hr;
    if (nimages != m_nimages) { Release(); return E_FAIL; }
    for (size_t i = 0; i < nimages; ++i)
    {
        const Image& s = images[i];
        const Image& d = m_images[i];
        if (!s.pixels) { Release(); return E_POINTER; }
        if (s.format != d.format || s.width != d.width || s.height != d.height) { Release(); return E_FAIL; }
        const size_t rows = ComputeScanlines(d.format, d.height);
        if (s.rowPitch == d.rowPitch) hr = dxtex_memcpy_h2d_async(device.Get(), d.pixels, s.pixels, d.rowPitch * rows);
        else
            for (size_t y = 0; y < rows && SUCCEEDED(hr); ++y)
                hr = dxtex_memcpy_h2d_async(device.Get(), d.pixels + y * d.rowPitch, s.pixels + y * s.rowPitch, std::min(s.rowPitch, d.rowPitch));
        if (FAILED(hr)) { Release(); return hr; }
    }
    hr = dxtex_ctx_synchronize(device.Get());          // the caller's images may go away once this returns
    if (FAILED(hr)) Release();
    return hr;
}

HRESULT DeviceScratchImage::Download(ScratchImage& dst) const noexcept
{
    if (!m_memory || !m_device || !*m_device) return E_POINTER;
    HRESULT hr = dst.Initialize(m_metadata);
    if (FAILED(hr)) return hr;
    if (dst.GetPixelsSize() != m_size) { dst.Release(); return E_FAIL; }
    hr = dxtex_memcpy_d2h(m_device->Get(), dst.GetPixels(), m_memory, m_size);
    if (FAILED(hr)) dst.Release();
    return hr;
}

bool DeviceScratchImage::OverrideFormat(DXGI_FORMAT f) noexcept
{
    if (!m_images || !IsValid(f) || IsPlanar(f) || IsPalettized(f)) return false;
    for (size_t i = 0; i < m_nimages; ++i) m_images[i].format = f;
    m_metadata.format = f;
    return true;
}

const Image* DeviceScratchImage::GetImage(size_t mip, size_t item, size_t slice) const noexcept
{
    const size_t i = m_metadata.ComputeIndex(mip, item, slice);
    return (i < m_nimages) ? &m_images[i] : nullptr;
}

// ---- the resident steps: the array templates above in DeviceSpace ---------------------------------------------------------------------
namespace
{
inline bool Resident(const Device& device, const DeviceScratchImage& src) noexcept { return src.GetImages() && src.GetDevice() == &device; }
}

HRESULT Compress(Device& device, const DeviceScratchImage& src, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, DeviceScratchImage& cImages) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    CompressOptions options = {};
    options.flags = compress; options.threshold = threshold;
    try { return CompressArrayT<DeviceSpace>(device, src.GetImages(), src.GetImageCount(), src.GetMetadata(), format, options, cImages, nullptr); }
    catch (...) { cImages.Release(); return E_OUTOFMEMORY; }
}

HRESULT Decompress(Device& device, const DeviceScratchImage& cImages, DXGI_FORMAT format, DeviceScratchImage& images) noexcept
{
    if (!Resident(device, cImages)) return E_INVALIDARG;
    return DecompressArrayT<DeviceSpace>(device, cImages.GetImages(), cImages.GetImageCount(), cImages.GetMetadata(), format, images);
}

HRESULT GenerateMipMaps(Device& device, const DeviceScratchImage& src, TEX_FILTER_FLAGS filter, size_t levels, DeviceScratchImage& mipChain) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    return GenerateMipMapsArrayT<DeviceSpace>(device, src.GetImages(), src.GetImageCount(), src.GetMetadata(), filter, levels, mipChain);
}

HRESULT GenerateMipMaps3D(Device& device, const DeviceScratchImage& src, TEX_FILTER_FLAGS filter, size_t levels, DeviceScratchImage& mipChain) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    const TexMetadata& metadata = src.GetMetadata();
    // the complex overload's checks (DirectXTexMipmaps.cpp:3364-3480)
    if (metadata.dimension != TEX_DIMENSION_TEXTURE3D) return E_INVALIDARG;
    if (metadata.depth > src.GetImageCount()) return E_FAIL;
    return GenerateMipMaps3DT<DeviceSpace>(device, src.GetImages(), metadata.depth, filter, levels, mipChain);
}

HRESULT Resize(Device& device, const DeviceScratchImage& src, size_t width, size_t height, TEX_FILTER_FLAGS filter, DeviceScratchImage& result) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    return ResizeArrayT<DeviceSpace>(device, src.GetImages(), src.GetImageCount(), src.GetMetadata(), width, height, filter, result);
}

HRESULT Convert(Device& device, const DeviceScratchImage& src, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, DeviceScratchImage& result) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    ConvertOptions options = {};
    options.filter = filter; options.threshold = threshold;
    try { return ConvertArrayT<DeviceSpace>(device, src.GetImages(), src.GetImageCount(), src.GetMetadata(), format, options, result, nullptr); }
    catch (...) { result.Release(); return E_OUTOFMEMORY; }
}

HRESULT PremultiplyAlpha(Device& device, const DeviceScratchImage& src, TEX_PMALPHA_FLAGS flags, DeviceScratchImage& result) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    return PremultiplyAlphaArrayT<DeviceSpace>(device, src.GetImages(), src.GetImageCount(), src.GetMetadata(), flags, result);
}

HRESULT ScaleMipMapsAlphaForCoverage(Device& device, const DeviceScratchImage& src, float alphaReference, DeviceScratchImage& mipChain) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    const TexMetadata& info = src.GetMetadata();
    HRESULT hr = mipChain.Initialize(device, info);
    if (FAILED(hr)) return hr;
    for (size_t item = 0; item < info.arraySize; ++item)
    {
        const Image* first = src.GetImage(0, item, 0);
        if (!first) { mipChain.Release(); return E_FAIL; }
        hr = ScaleMipMapsAlphaForCoverageT<DeviceSpace>(device, first, info.mipLevels, info, item, alphaReference, mipChain);
        if (FAILED(hr)) { mipChain.Release(); return hr; }
    }
    return S_OK;
}

HRESULT CopyTopLevels(Device& device, const DeviceScratchImage& src, DeviceScratchImage& result) noexcept
{
    if (!Resident(device, src)) return E_INVALIDARG;
    TexMetadata m = src.GetMetadata();
    m.mipLevels = 1;
    HRESULT hr = result.Initialize(device, m);
    if (FAILED(hr)) return hr;
    const bool volume = m.dimension == TEX_DIMENSION_TEXTURE3D;
    for (size_t i = 0; i < (volume ? m.depth : m.arraySize); ++i)
    {
        const Image* s = volume ? src.GetImage(0, 0, i) : src.GetImage(0, i, 0);
        const Image* d = volume ? result.GetImage(0, 0, i) : result.GetImage(0, i, 0);
        if (!s || !d) { result.Release(); return E_FAIL; }
        const size_t rows = ComputeScanlines(d->format, d->height);
        hr = dxtex_copy_rows_device(device.Get(), d->pixels, d->rowPitch, s->pixels, s->rowPitch, std::min(s->rowPitch, d->rowPitch), rows);
        if (FAILED(hr)) { result.Release(); return hr; }
    }
    return S_OK;
}

bool IsAlphaAllOpaque(Device& device, const Image* deviceImages, size_t nimages) noexcept
{
    if (!device || !deviceImages || !nimages) return false;
    if (!HasAlpha(deviceImages[0].format)) return true;                  // DirectXTexImage.cpp:805-806
    try
    {
        std::vector<dxtex_image> v(nimages);
        for (size_t i = 0; i < nimages; ++i) v[i] = View(deviceImages[i]);
        int opaque = 0;
        return SUCCEEDED(dxtex_alpha_all_opaque_device(device.Get(), v.data(), nimages, &opaque)) && opaque != 0;
    }
    catch (...) { return false; }
}

bool IsAlphaAllOpaque(Device& device, const DeviceScratchImage& image) noexcept
{
    if (!Resident(device, image)) return false;
    return IsAlphaAllOpaque(device, image.GetImages(), image.GetImageCount());
}

void GetTransferBytes(Device& device, uint64_t& hostToDevice, uint64_t& deviceToHost, bool reset) noexcept
{
    hostToDevice = deviceToHost = 0;
    if (device) dxtex_ctx_transfer_bytes(device.Get(), &hostToDevice, &deviceToHost, reset ? 1 : 0);
}
} // namespace DirectXTexAMD
