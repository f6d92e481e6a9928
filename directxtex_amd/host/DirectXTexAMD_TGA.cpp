// Truevision TGA reader / writer of the host layer. Behaviour follows DirectXTexTGA.cpp: which headers are accepted and the error
// code for each that is not (DecodeTGAHeader, :110-258); 8-bit grey, 16-bit 5:5:5:1, 24- and 32-bit true colour, raw or run-length
// encoded, and 8-bit colour-mapped with a 24-bit palette (:381-1180), either row order, either column order; the all-zero /
// all-opaque alpha rules; the TGA 2.0 extension area for the alpha mode and gamma -> sRGB (:1381-1440); on output the header choice
// per format, 24-bit rows for B8G8R8X8, the extension area and footer (:1183-1380, :2249-2330). One routine handles every texel
// size here where the reference has a copy per format; tests/test_hdr_tga_cpu.py checks files, pixels, metadata and HRESULTs
// against the reference's own codec (oracle/_ref), including truncated and mutated files. Host code.
#include "DirectXTexAMD.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <vector>

namespace DirectXTexAMD
{
namespace
{
    const char kSignature[] = "TRUEVISION-XFILE.";            // 18 bytes with its NUL, as the footer stores it
    enum : uint8_t { IMG_NONE = 0, IMG_MAPPED = 1, IMG_TRUECOLOR = 2, IMG_GREY = 3, IMG_MAPPED_RLE = 9, IMG_TRUECOLOR_RLE = 10, IMG_GREY_RLE = 11 };
    enum : uint8_t { DESC_INVERTX = 0x10, DESC_INVERTY = 0x20, DESC_INTERLEAVED = 0xC0 };
    enum : uint8_t { ATTR_NONE = 0, ATTR_IGNORED = 1, ATTR_UNDEFINED = 2, ATTR_ALPHA = 3, ATTR_PREMULTIPLIED = 4 };
    constexpr size_t kHeaderLen = 18, kFooterLen = 26, kExtensionLen = 495;
    // offsets inside the extension area (TGA 2.0): size, time stamp, software id + version, gamma, attributes type
    constexpr size_t EXT_STAMP = 367, EXT_SOFTWARE = 426, EXT_VERSION = 467, EXT_VERSION_LETTER = 469, EXT_GAMMA = 478, EXT_ATTRIBUTES = 494;

    struct Header { uint8_t idLength, colorMapType, imageType; uint16_t colorMapFirst, colorMapLength; uint8_t colorMapSize; uint16_t width, height; uint8_t bitsPerPixel, descriptor; };
    inline uint16_t rd16(const uint8_t* p) noexcept { return uint16_t(p[0] | (p[1] << 8)); }
    inline uint32_t rd32(const uint8_t* p) noexcept { uint32_t v; std::memcpy(&v, p, 4); return v; }
    inline void wr16(uint8_t* p, uint32_t v) noexcept { p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); }
    Header ParseHeader(const uint8_t* p) noexcept
    {
        Header h;
        h.idLength = p[0]; h.colorMapType = p[1]; h.imageType = p[2]; h.colorMapFirst = rd16(p + 3); h.colorMapLength = rd16(p + 5); h.colorMapSize = p[7];
        h.width = rd16(p + 12); h.height = rd16(p + 14); h.bitsPerPixel = p[16]; h.descriptor = p[17];
        return h;
    }

    enum : uint32_t { RD_EXPAND = 0x1, RD_INVERTX = 0x2, RD_INVERTY = 0x4, RD_RLE = 0x8, RD_PALETTED = 0x10 };

    HRESULT DecodeHeader(const uint8_t* pSource, size_t size, uint32_t flags, TexMetadata& metadata, size_t& offset, uint32_t* how) noexcept
    {
        if (!pSource) return E_POINTER;
        metadata = TexMetadata();
        metadata.dimension = TEX_DIMENSION(0);
        if (size < kHeaderLen) return HRESULT_E_INVALID_DATA;
        const Header h = ParseHeader(pSource);
        if (h.descriptor & DESC_INTERLEAVED) return HRESULT_E_NOT_SUPPORTED;
        if (!h.width || !h.height) return HRESULT_E_INVALID_DATA;
        const bool bgr = (flags & TGA_FLAGS_BGR) != 0;
        auto rgb24 = [&]()        // a 24-bit colour lands in RGBA8 (opaque) or, on request, stays BGR in B8G8R8X8
        {
            if (bgr) metadata.format = DXGI_FORMAT_B8G8R8X8_UNORM;
            else { metadata.format = DXGI_FORMAT_R8G8B8A8_UNORM; metadata.SetAlphaMode(TEX_ALPHA_MODE_OPAQUE); }
        };
        switch (h.imageType)
        {
        case IMG_NONE: case IMG_MAPPED_RLE:
            return HRESULT_E_NOT_SUPPORTED;
        case IMG_MAPPED:
            if (h.colorMapType != 1 || h.colorMapLength == 0 || h.bitsPerPixel != 8 || h.colorMapSize != 24) return HRESULT_E_NOT_SUPPORTED;
            rgb24();
            if (how) *how |= RD_PALETTED;
            break;
        case IMG_TRUECOLOR: case IMG_TRUECOLOR_RLE:
            if (h.colorMapType != 0 || h.colorMapLength != 0) return HRESULT_E_NOT_SUPPORTED;
            switch (h.bitsPerPixel)
            {
            case 16: metadata.format = DXGI_FORMAT_B5G5R5A1_UNORM; break;
            case 24: rgb24(); if (how) *how |= RD_EXPAND; break;
            case 32: metadata.format = bgr ? DXGI_FORMAT_B8G8R8A8_UNORM : DXGI_FORMAT_R8G8B8A8_UNORM; break;
            default: return HRESULT_E_NOT_SUPPORTED;
            }
            if (how && h.imageType == IMG_TRUECOLOR_RLE) *how |= RD_RLE;
            break;
        case IMG_GREY: case IMG_GREY_RLE:
            if (h.colorMapType != 0 || h.colorMapLength != 0 || h.bitsPerPixel != 8) return HRESULT_E_NOT_SUPPORTED;
            metadata.format = DXGI_FORMAT_R8_UNORM;
            if (how && h.imageType == IMG_GREY_RLE) *how |= RD_RLE;
            break;
        default:
            return HRESULT_E_INVALID_DATA;
        }
        if (uint64_t(h.width) * uint64_t(h.height) * uint64_t(h.bitsPerPixel) / 8 > UINT32_MAX) return HRESULT_E_ARITHMETIC_OVERFLOW;
        metadata.width = h.width; metadata.height = h.height;
        metadata.depth = metadata.arraySize = metadata.mipLevels = 1;
        metadata.dimension = TEX_DIMENSION_TEXTURE2D;
        if (how)
        {
            if (h.descriptor & DESC_INVERTX) *how |= RD_INVERTX;
            if (h.descriptor & DESC_INVERTY) *how |= RD_INVERTY;
        }
        offset = kHeaderLen + h.idLength;
        return S_OK;
    }

    // the 24-bit colour map as RGBA / BGRA words; entries outside [first, first + length) stay zero (ReadPalette, :260-310)
    HRESULT ReadPalette(const Header& h, const uint8_t* bytes, size_t size, uint32_t flags, uint32_t palette[256], size_t& mapBytes) noexcept
    {
        if (h.colorMapType != 1 || h.colorMapLength == 0 || h.colorMapLength > 256 || h.colorMapSize != 24) return HRESULT_E_NOT_SUPPORTED;
        const size_t last = size_t(h.colorMapFirst) + h.colorMapLength;
        if (last > 256) return HRESULT_E_NOT_SUPPORTED;
        mapBytes = size_t(h.colorMapLength) * 3;
        if (mapBytes > size) return HRESULT_E_INVALID_DATA;
        for (size_t i = h.colorMapFirst; i < last; ++i, bytes += 3)
            palette[i] = (flags & TGA_FLAGS_BGR) ? (uint32_t(bytes[0]) | (uint32_t(bytes[1]) << 8) | (uint32_t(bytes[2]) << 16) | 0xFF000000u)
                                                 : (uint32_t(bytes[2]) | (uint32_t(bytes[1]) << 8) | (uint32_t(bytes[0]) << 16) | 0xFF000000u);
        return S_OK;
    }

    struct AlphaRange { uint32_t lo = 255, hi = 0; void see(uint32_t a) noexcept { lo = std::min(lo, a); hi = std::max(hi, a); } };

    // One texel of the file (B bytes at s) as the image stores it, noting its alpha.
    inline uint32_t Texel(const uint8_t* s, DXGI_FORMAT format, bool expand, const uint32_t* palette, AlphaRange& alpha) noexcept
    {
        if (palette) return palette[s[0]];
        switch (format)
        {
        case DXGI_FORMAT_R8_UNORM: return s[0];
        case DXGI_FORMAT_B5G5R5A1_UNORM: { const uint32_t t = rd16(s); alpha.see((t & 0x8000) ? 255 : 0); return t; }
        case DXGI_FORMAT_R8G8B8A8_UNORM:          // the file is B, G, R (, A)
            if (expand) { alpha.lo = alpha.hi = 255; return (uint32_t(s[0]) << 16) | (uint32_t(s[1]) << 8) | s[2] | 0xFF000000u; }
            alpha.see(s[3]);
            return (uint32_t(s[0]) << 16) | (uint32_t(s[1]) << 8) | s[2] | (uint32_t(s[3]) << 24);
        case DXGI_FORMAT_B8G8R8A8_UNORM: alpha.see(s[3]); return rd32(s);
        default: return uint32_t(s[0]) | (uint32_t(s[1]) << 8) | (uint32_t(s[2]) << 16);          // B8G8R8X8 from 24 bits: X stays 0
        }
    }

    // UncompressPixels / CopyPixels (:381-1180). S_FALSE: every alpha turned out opaque (or was made so).
    HRESULT ReadPixels(const uint8_t* s, size_t size, uint32_t flags, const Image& image, uint32_t how, const uint32_t* palette) noexcept
    {
        if (!s || !image.pixels) return E_POINTER;
        const uint8_t* end = s + size;
        const bool expand = (how & RD_EXPAND) != 0, flipX = (how & RD_INVERTX) != 0;
        const DXGI_FORMAT format = image.format;
        size_t B, D;              // bytes per texel in the file and in the image
        switch (format)
        {
        case DXGI_FORMAT_R8_UNORM: B = D = 1; break;
        case DXGI_FORMAT_B5G5R5A1_UNORM: B = D = 2; break;
        case DXGI_FORMAT_R8G8B8A8_UNORM: B = expand ? 3 : 4; D = 4; break;
        case DXGI_FORMAT_B8G8R8A8_UNORM: B = D = 4; break;
        case DXGI_FORMAT_B8G8R8X8_UNORM: B = 3; D = 4; break;
        default: return E_FAIL;
        }
        if (palette) B = 1;
        AlphaRange alpha;
        auto put = [&](uint8_t*& d, uint32_t t)
        {
            if (D == 1) *d = uint8_t(t); else if (D == 2) wr16(d, t); else std::memcpy(d, &t, 4);
            if (flipX) d -= D; else d += D;
        };
        for (size_t y = 0; y < image.height; ++y)
        {
            // the file's first row is the image's bottom row unless the descriptor says top-down; right-to-left likewise
            uint8_t* d = image.pixels + image.rowPitch * ((how & RD_INVERTY) ? y : (image.height - y - 1)) + (flipX ? (image.width - 1) * D : 0);
            if (!(how & RD_RLE))
            {
                for (size_t x = 0; x < image.width; ++x, s += B)
                {
                    if (s + (B - 1) >= end) return E_FAIL;
                    put(d, Texel(s, format, expand, palette, alpha));
                }
                continue;
            }
            for (size_t x = 0; x < image.width;)
            {
                if (s >= end) return E_FAIL;
                size_t n = size_t(*s & 0x7F) + 1;
                const bool run = (*s & 0x80) != 0;
                ++s;
                if (run)
                {
                    if (s + (B - 1) >= end) return E_FAIL;
                    const uint32_t t = Texel(s, format, expand, nullptr, alpha);
                    s += B;
                    for (; n > 0; --n, ++x)
                    {
                        if (x >= image.width) return E_FAIL;           // packets do not cross rows
                        put(d, t);
                    }
                }
                else
                {
                    if (s + n * B > end) return E_FAIL;
                    for (; n > 0; --n, ++x, s += B)
                    {
                        if (x >= image.width) return E_FAIL;
                        put(d, Texel(s, format, expand, nullptr, alpha));
                    }
                }
            }
        }
        if (palette || format == DXGI_FORMAT_R8_UNORM || format == DXGI_FORMAT_B8G8R8X8_UNORM) return S_OK;
        if (alpha.hi == 0 && !(flags & TGA_FLAGS_ALLOW_ALL_ZERO_ALPHA))
        {
            // an alpha channel that is zero everywhere was not meant: make it opaque
            for (size_t y = 0; y < image.height; ++y)
            {
                uint8_t* row = image.pixels + y * image.rowPitch;
                if (D == 2) for (size_t x = 0; x < image.width; ++x) row[x * 2 + 1] |= 0x80;
                else for (size_t x = 0; x < image.width; ++x) row[x * 4 + 3] = 0xFF;
            }
            return 1;         // S_FALSE
        }
        return (alpha.lo == 255) ? 1 : S_OK;
    }

    // the extension area the footer points at, if the file has a TGA 2.0 footer and the area lies inside the file
    const uint8_t* FindExtension(const uint8_t* p, size_t size) noexcept
    {
        if (size < kFooterLen) return nullptr;
        const uint8_t* footer = p + size - kFooterLen;
        if (std::memcmp(footer + 8, kSignature, sizeof(kSignature)) != 0) return nullptr;
        const uint32_t at = rd32(footer);
        if (at == 0 || size_t(at) + kExtensionLen > size) return nullptr;
        return p + at;
    }

    TEX_ALPHA_MODE AlphaModeOf(const uint8_t* ext) noexcept
    {
        if (!ext || rd16(ext) != kExtensionLen) return TEX_ALPHA_MODE_UNKNOWN;
        switch (ext[EXT_ATTRIBUTES])
        {
        case ATTR_IGNORED: return TEX_ALPHA_MODE_OPAQUE;
        case ATTR_UNDEFINED: return TEX_ALPHA_MODE_CUSTOM;
        case ATTR_ALPHA: return TEX_ALPHA_MODE_STRAIGHT;
        case ATTR_PREMULTIPLIED: return TEX_ALPHA_MODE_PREMULTIPLIED;
        default: return TEX_ALPHA_MODE_UNKNOWN;
        }
    }

    // a gamma of 2.2 or 2.4 in the extension area means sRGB; with no usable gamma the caller's default decides (:1409-1440)
    DXGI_FORMAT ApplyGamma(const uint8_t* ext, DXGI_FORMAT format, uint32_t flags, ScratchImage* image) noexcept
    {
        bool srgb;
        if (ext && rd16(ext) == kExtensionLen && rd16(ext + EXT_GAMMA + 2) != 0)
        {
            const float gamma = float(rd16(ext + EXT_GAMMA)) / float(rd16(ext + EXT_GAMMA + 2));
            srgb = std::fabs(gamma - 2.2f) < 0.01f || std::fabs(gamma - 2.4f) < 0.01f;
        }
        else srgb = (flags & TGA_FLAGS_DEFAULT_SRGB) != 0;
        if (srgb)
        {
            format = MakeSRGB(format);
            if (image) image->OverrideFormat(format);
        }
        return format;
    }

    HRESULT ReadAll(const char* szFile, std::vector<uint8_t>& buf, size_t minimum) noexcept
    {
        FILE* f = std::fopen(szFile, "rb");
        if (!f) return E_FAIL;
        std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        if (n < 0) { std::fclose(f); return E_FAIL; }
        if (uint64_t(n) > UINT32_MAX) { std::fclose(f); return HRESULT_E_FILE_TOO_LARGE; }
        if (size_t(n) < minimum) { std::fclose(f); return E_FAIL; }
        try { buf.resize(size_t(n)); } catch (...) { std::fclose(f); return E_OUTOFMEMORY; }
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        return got == buf.size() ? S_OK : E_FAIL;
    }
}

HRESULT GetMetadataFromTGAMemory(const void* pSource, size_t size, TGA_FLAGS flags, TexMetadata& metadata) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    const uint8_t* p = static_cast<const uint8_t*>(pSource);
    size_t offset;
    const HRESULT hr = DecodeHeader(p, size, flags, metadata, offset, nullptr);
    if (FAILED(hr)) return hr;
    const uint8_t* ext = FindExtension(p, size);
    if (ext) metadata.SetAlphaMode(AlphaModeOf(ext));
    if (!(flags & TGA_FLAGS_IGNORE_SRGB)) metadata.format = ApplyGamma(ext, metadata.format, flags, nullptr);
    return S_OK;
}

HRESULT GetMetadataFromTGAFile(const char* szFile, TGA_FLAGS flags, TexMetadata& metadata) noexcept
{
    if (!szFile) return E_INVALIDARG;
    std::vector<uint8_t> buf;
    const HRESULT hr = ReadAll(szFile, buf, kHeaderLen);
    if (FAILED(hr)) return hr;
    return GetMetadataFromTGAMemory(buf.data(), buf.size(), flags, metadata);
}

// LoadFromTGAMemory (DirectXTexTGA.cpp:1640-1745)
HRESULT LoadFromTGAMemory(const void* pSource, size_t size, TGA_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    image.Release();
    const uint8_t* p = static_cast<const uint8_t*>(pSource);
    size_t offset; uint32_t how = 0;
    TexMetadata mdata;
    HRESULT hr = DecodeHeader(p, size, flags, mdata, offset, &how);
    if (FAILED(hr)) return hr;
    if (offset > size) return HRESULT_E_INVALID_DATA;
    size_t mapBytes = 0;
    uint32_t palette[256] = {};
    if (how & RD_PALETTED)
    {
        if (size - offset == 0) return E_FAIL;
        hr = ReadPalette(ParseHeader(p), p + offset, size - offset, flags, palette, mapBytes);
        if (FAILED(hr)) return hr;
    }
    const size_t remaining = size - offset - mapBytes;
    if (remaining == 0) return HRESULT_E_HANDLE_EOF;
    hr = image.Initialize2D(mdata.format, mdata.width, mdata.height, 1, 1);
    if (FAILED(hr)) return hr;
    hr = ReadPixels(p + offset + mapBytes, remaining, flags, *image.GetImage(0, 0, 0), how, (how & RD_PALETTED) ? palette : nullptr);
    if (FAILED(hr)) { image.Release(); return hr; }
    const uint8_t* ext = FindExtension(p, size);
    if (!(flags & TGA_FLAGS_IGNORE_SRGB)) mdata.format = ApplyGamma(ext, mdata.format, flags, &image);
    if (metadata)
    {
        *metadata = mdata;
        if (hr == 1) metadata->SetAlphaMode(TEX_ALPHA_MODE_OPAQUE);
        else if (ext) metadata->SetAlphaMode(AlphaModeOf(ext));
    }
    return S_OK;
}

HRESULT LoadFromTGAFile(const char* szFile, TGA_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept
{
    if (!szFile) return E_INVALIDARG;
    image.Release();
    std::vector<uint8_t> buf;
    const HRESULT hr = ReadAll(szFile, buf, kHeaderLen);
    if (FAILED(hr)) return hr;
    return LoadFromTGAMemory(buf.data(), buf.size(), flags, metadata, image);
}

// SaveToTGAMemory (DirectXTexTGA.cpp:2249-2330): uncompressed, top-down; with metadata a TGA 2.0 extension area is added
HRESULT SaveToTGAMemory(const Image& image, TGA_FLAGS flags, Blob& blob, const TexMetadata* metadata) noexcept
{
    if ((flags & (TGA_FLAGS_FORCE_LINEAR | TGA_FLAGS_FORCE_SRGB)) != 0 && !metadata) return E_INVALIDARG;
    if (!image.pixels) return E_POINTER;
    if (image.width > UINT16_MAX || image.height > UINT16_MAX) return HRESULT_E_NOT_SUPPORTED;
    uint8_t type, bits, descriptor;
    enum { ROW_COPY, ROW_SWAP_RB, ROW_DROP_X } row = ROW_COPY;
    switch (image.format)
    {
    case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: type = IMG_TRUECOLOR; bits = 32; descriptor = DESC_INVERTY | 8; row = ROW_SWAP_RB; break;
    case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: type = IMG_TRUECOLOR; bits = 32; descriptor = DESC_INVERTY | 8; break;
    case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: type = IMG_TRUECOLOR; bits = 24; descriptor = DESC_INVERTY; row = ROW_DROP_X; break;
    case DXGI_FORMAT_R8_UNORM: case DXGI_FORMAT_A8_UNORM: type = IMG_GREY; bits = 8; descriptor = DESC_INVERTY; break;
    case DXGI_FORMAT_B5G5R5A1_UNORM: type = IMG_TRUECOLOR; bits = 16; descriptor = DESC_INVERTY | 1; break;
    default: return HRESULT_E_NOT_SUPPORTED;
    }
    blob.Release();
    const size_t rowPitch = image.width * (bits / 8), slicePitch = rowPitch * image.height;
    HRESULT hr = blob.Initialize(kHeaderLen + slicePitch + (metadata ? kExtensionLen : 0) + kFooterLen);
    if (FAILED(hr)) return hr;
    uint8_t* base = blob.GetBufferPointer();
    std::memset(base, 0, kHeaderLen);
    base[2] = type; wr16(base + 12, uint32_t(image.width)); wr16(base + 14, uint32_t(image.height)); base[16] = bits; base[17] = descriptor;
    uint8_t* d = base + kHeaderLen;
    for (size_t y = 0; y < image.height; ++y, d += rowPitch)
    {
        const uint8_t* s = image.pixels + y * image.rowPitch;
        if (row == ROW_DROP_X) { for (size_t x = 0; x < image.width && x * 4 + 3 < image.rowPitch; ++x) { d[x * 3] = s[x * 4]; d[x * 3 + 1] = s[x * 4 + 1]; d[x * 3 + 2] = s[x * 4 + 2]; } }
        else if (row == ROW_SWAP_RB) { for (size_t x = 0; x < image.width && x * 4 + 3 < image.rowPitch; ++x) { d[x * 4] = s[x * 4 + 2]; d[x * 4 + 1] = s[x * 4 + 1]; d[x * 4 + 2] = s[x * 4]; d[x * 4 + 3] = s[x * 4 + 3]; } }
        else std::memcpy(d, s, std::min(rowPitch, image.rowPitch));
    }
    uint32_t extOffset = 0;
    if (metadata)
    {
        // SetExtension (:1322-1380): who wrote it, gamma, what the alpha channel means, when
        std::memset(d, 0, kExtensionLen);
        wr16(d, kExtensionLen);
        std::memcpy(d + EXT_SOFTWARE, "DirectXTex", sizeof("DirectXTex"));
        wr16(d + EXT_VERSION, 211); d[EXT_VERSION_LETTER] = ' ';          // DIRECTX_TEX_VERSION of the reference this mirrors (DirectXTex.h:50)
        const bool srgb = !(flags & TGA_FLAGS_FORCE_LINEAR) && ((flags & TGA_FLAGS_FORCE_SRGB) || IsSRGB(metadata->format));
        if (srgb) { wr16(d + EXT_GAMMA, 22); wr16(d + EXT_GAMMA + 2, 10); }
        else if (flags & TGA_FLAGS_FORCE_LINEAR) { wr16(d + EXT_GAMMA, 1); wr16(d + EXT_GAMMA + 2, 1); }
        switch (metadata->GetAlphaMode())
        {
        case TEX_ALPHA_MODE_STRAIGHT: d[EXT_ATTRIBUTES] = ATTR_ALPHA; break;
        case TEX_ALPHA_MODE_PREMULTIPLIED: d[EXT_ATTRIBUTES] = ATTR_PREMULTIPLIED; break;
        case TEX_ALPHA_MODE_OPAQUE: d[EXT_ATTRIBUTES] = ATTR_IGNORED; break;
        case TEX_ALPHA_MODE_CUSTOM: d[EXT_ATTRIBUTES] = ATTR_UNDEFINED; break;
        default: d[EXT_ATTRIBUTES] = HasAlpha(metadata->format) ? ATTR_UNDEFINED : ATTR_NONE; break;
        }
        std::time_t now = {};
        std::time(&now);
        if (const std::tm* t = std::gmtime(&now))
        {
            wr16(d + EXT_STAMP, uint32_t(t->tm_mon + 1)); wr16(d + EXT_STAMP + 2, uint32_t(t->tm_mday)); wr16(d + EXT_STAMP + 4, uint32_t(t->tm_year + 1900));
            wr16(d + EXT_STAMP + 6, uint32_t(t->tm_hour)); wr16(d + EXT_STAMP + 8, uint32_t(t->tm_min)); wr16(d + EXT_STAMP + 10, uint32_t(t->tm_sec));
        }
        extOffset = uint32_t(d - base);
        d += kExtensionLen;
    }
    std::memcpy(d, &extOffset, 4);
    std::memset(d + 4, 0, 4);
    std::memcpy(d + 8, kSignature, sizeof(kSignature));
    return S_OK;
}

HRESULT SaveToTGAFile(const Image& image, TGA_FLAGS flags, const char* szFile, const TexMetadata* metadata) noexcept
{
    if (!szFile) return E_INVALIDARG;
    Blob blob;
    const HRESULT hr = SaveToTGAMemory(image, flags, blob, metadata);
    if (FAILED(hr)) return hr;
    FILE* f = std::fopen(szFile, "wb");
    if (!f) return E_FAIL;
    const size_t n = std::fwrite(blob.GetBufferPointer(), 1, blob.GetBufferSize(), f);
    const bool closed = std::fclose(f) == 0;
    if (n != blob.GetBufferSize() || !closed) { std::remove(szFile); return E_FAIL; }
    return S_OK;
}
} // namespace DirectXTexAMD
