// DDS reader / writer of the host layer (see DirectXTexAMD.h). Format facts follow the public DDS layout as the
// reference implements it: magic 0x20534444 + 124-byte DDS_HEADER (DDS.h:262-278) + optional 20-byte DDS_HEADER_DXT10
// (:280-287); which formats get a legacy pixel format and which the 'DX10' extension is EncodeDDSHeader's choice
// (DirectXTexDDS.cpp:711-1033); the payload is the ScratchImage order (item-major, mips inside) with default pitches.
#include "DirectXTexAMD.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace DirectXTexAMD
{
namespace
{
    constexpr uint32_t kMagic = 0x20534444u;       // "DDS "
    constexpr uint32_t FOURCC = 0x4, RGB = 0x40, RGBA = 0x41, LUM = 0x20000, LUMA = 0x20001, ALPHA = 0x2, BUMPDUDV = 0x80000;
    constexpr uint32_t HF_TEXTURE = 0x1007, HF_MIPMAP = 0x20000, HF_PITCH = 0x8, HF_LINEARSIZE = 0x80000, HF_VOLUME = 0x800000;
    constexpr uint32_t CAPS_TEXTURE = 0x1000, CAPS_MIPMAP = 0x400008, CAPS_CUBEMAP = 0x8, CAPS2_CUBEMAP_ALL = 0xFE00, CAPS2_CUBEMAP = 0x200, CAPS2_VOLUME = 0x200000;

#pragma pack(push, 1)
    struct PixelFormat { uint32_t size, flags, fourCC, bitCount, rMask, gMask, bMask, aMask; };
    struct Header { uint32_t size, flags, height, width, pitchOrLinearSize, depth, mipMapCount, reserved1[11]; PixelFormat pf; uint32_t caps, caps2, caps3, caps4, reserved2; };
    struct HeaderDX10 { uint32_t dxgiFormat, resourceDimension, miscFlag, arraySize, miscFlags2; };
#pragma pack(pop)
    static_assert(sizeof(Header) == 124 && sizeof(HeaderDX10) == 20, "DDS header layout");

    constexpr uint32_t cc(char a, char b, char c, char d) { return uint32_t(uint8_t(a)) | (uint32_t(uint8_t(b)) << 8) | (uint32_t(uint8_t(c)) << 16) | (uint32_t(uint8_t(d)) << 24); }

    struct Legacy { DXGI_FORMAT format; PixelFormat pf; bool write; };      // write: EncodeDDSHeader uses it for this format
    const Legacy kLegacy[] = {
        { DXGI_FORMAT_R8G8B8A8_UNORM, { 32, RGBA, 0, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0xff000000 }, true },
        { DXGI_FORMAT_R16G16_UNORM,   { 32, RGB, 0, 32, 0x0000ffff, 0xffff0000, 0, 0 }, true },
        { DXGI_FORMAT_R8G8_UNORM,     { 32, LUMA, 0, 16, 0x00ff, 0, 0, 0xff00 }, true },
        { DXGI_FORMAT_R16_UNORM,      { 32, LUM, 0, 16, 0xffff, 0, 0, 0 }, true },
        { DXGI_FORMAT_R8_UNORM,       { 32, LUM, 0, 8, 0xff, 0, 0, 0 }, true },
        { DXGI_FORMAT_A8_UNORM,       { 32, ALPHA, 0, 8, 0, 0, 0, 0xff }, true },
        { DXGI_FORMAT_R8G8_SNORM,     { 32, BUMPDUDV, 0, 16, 0x00ff, 0xff00, 0, 0 }, true },
        { DXGI_FORMAT_R8G8B8A8_SNORM, { 32, BUMPDUDV, 0, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0xff000000 }, true },
        { DXGI_FORMAT_B8G8R8A8_UNORM, { 32, RGBA, 0, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000 }, true },
        { DXGI_FORMAT_B8G8R8X8_UNORM, { 32, RGB, 0, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0 }, true },
        { DXGI_FORMAT_BC1_UNORM, { 32, FOURCC, cc('D', 'X', 'T', '1'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC2_UNORM, { 32, FOURCC, cc('D', 'X', 'T', '3'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC3_UNORM, { 32, FOURCC, cc('D', 'X', 'T', '5'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC4_UNORM, { 32, FOURCC, cc('B', 'C', '4', 'U'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC4_SNORM, { 32, FOURCC, cc('B', 'C', '4', 'S'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC5_UNORM, { 32, FOURCC, cc('B', 'C', '5', 'U'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC5_SNORM, { 32, FOURCC, cc('B', 'C', '5', 'S'), 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_BC4_UNORM, { 32, FOURCC, cc('A', 'T', 'I', '1'), 0, 0, 0, 0, 0 }, false },
        { DXGI_FORMAT_BC5_UNORM, { 32, FOURCC, cc('A', 'T', 'I', '2'), 0, 0, 0, 0, 0 }, false },
        { DXGI_FORMAT_BC2_UNORM, { 32, FOURCC, cc('D', 'X', 'T', '2'), 0, 0, 0, 0, 0 }, false },
        { DXGI_FORMAT_BC3_UNORM, { 32, FOURCC, cc('D', 'X', 'T', '4'), 0, 0, 0, 0, 0 }, false },
        // legacy D3DX files carry the D3DFMT enum value as FourCC
        { DXGI_FORMAT_R32G32B32A32_FLOAT, { 32, FOURCC, 116, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R16G16B16A16_FLOAT, { 32, FOURCC, 113, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R16G16B16A16_UNORM, { 32, FOURCC, 36, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R32G32_FLOAT, { 32, FOURCC, 115, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R16G16_FLOAT, { 32, FOURCC, 112, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R32_FLOAT, { 32, FOURCC, 114, 0, 0, 0, 0, 0 }, true },
        { DXGI_FORMAT_R16_FLOAT, { 32, FOURCC, 111, 0, 0, 0, 0, 0 }, true },
    };

    bool SamePF(const PixelFormat& a, const PixelFormat& b) noexcept
    {
        if (a.flags & FOURCC) return (b.flags & FOURCC) && a.fourCC == b.fourCC;
        return a.flags == b.flags && a.bitCount == b.bitCount && a.rMask == b.rMask && a.gMask == b.gMask && a.bMask == b.bMask && a.aMask == b.aMask;
    }

    HRESULT DecodeHeader(const void* pSource, size_t size, TexMetadata& m, size_t& offset) noexcept
    {
        if (!pSource) return E_INVALIDARG;
        if (size < 4 + sizeof(Header)) return HRESULT(0x80070026);        // HRESULT_E_HANDLE_EOF
        const uint8_t* p = static_cast<const uint8_t*>(pSource);
        uint32_t magic; std::memcpy(&magic, p, 4);
        if (magic != kMagic) return E_FAIL;
        Header h; std::memcpy(&h, p + 4, sizeof(h));
        if (h.size != sizeof(Header) || h.pf.size != sizeof(PixelFormat)) return E_FAIL;
        m = TexMetadata();
        m.mipLevels = h.mipMapCount ? h.mipMapCount : 1;
        offset = 4 + sizeof(Header);
        if ((h.pf.flags & FOURCC) && h.pf.fourCC == cc('D', 'X', '1', '0'))
        {
            if (size < offset + sizeof(HeaderDX10)) return E_FAIL;
            HeaderDX10 x; std::memcpy(&x, p + offset, sizeof(x));
            offset += sizeof(HeaderDX10);
            m.arraySize = x.arraySize;
            if (m.arraySize == 0) return HRESULT(0x8007000D);                 // HRESULT_E_INVALID_DATA
            m.format = DXGI_FORMAT(x.dxgiFormat);
            if (BitsPerPixel(m.format) == 0) return HRESULT_E_NOT_SUPPORTED;
            m.miscFlags = x.miscFlag & ~uint32_t(TEX_MISC_TEXTURECUBE);
            m.miscFlags2 = x.miscFlags2;
            switch (x.resourceDimension)
            {
            case TEX_DIMENSION_TEXTURE1D:
                if ((h.flags & 0x2) && h.height != 1) return HRESULT(0x8007000D);
                m.width = h.width; m.height = 1; m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE1D;
                break;
            case TEX_DIMENSION_TEXTURE2D:
                if (x.miscFlag & TEX_MISC_TEXTURECUBE) { m.miscFlags |= TEX_MISC_TEXTURECUBE; m.arraySize *= 6; }
                m.width = h.width; m.height = h.height; m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE2D;
                break;
            case TEX_DIMENSION_TEXTURE3D:
                if (!(h.flags & HF_VOLUME)) return HRESULT(0x8007000D);                      // DirectXTexDDS.cpp:465-478
                if (m.arraySize > 1) return HRESULT_E_NOT_SUPPORTED;
                m.width = h.width; m.height = h.height; m.depth = h.depth; m.dimension = TEX_DIMENSION_TEXTURE3D;
                break;
            default:
                return HRESULT(0x8007000D);
            }
        }
        else
        {
            m.arraySize = 1;
            m.width = h.width; m.height = h.height; m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE2D;
            if (h.flags & HF_VOLUME) { m.depth = h.depth; m.dimension = TEX_DIMENSION_TEXTURE3D; }        // :497-504
            else if (h.caps2 & CAPS2_CUBEMAP)
            {
                if ((h.caps2 & CAPS2_CUBEMAP_ALL) != CAPS2_CUBEMAP_ALL) return HRESULT_E_NOT_SUPPORTED;      // all six faces required
                m.arraySize = 6; m.miscFlags |= TEX_MISC_TEXTURECUBE;
            }
            m.format = DXGI_FORMAT_UNKNOWN;
            for (const Legacy& l : kLegacy)
                if (SamePF(l.pf, h.pf)) { m.format = l.format; break; }
            if (m.format == DXGI_FORMAT_UNKNOWN) return HRESULT_E_NOT_SUPPORTED;
            if ((h.pf.flags & FOURCC) && (h.pf.fourCC == cc('D', 'X', 'T', '2') || h.pf.fourCC == cc('D', 'X', 'T', '4')))
                m.miscFlags2 = 2;         // TEX_ALPHA_MODE_PREMULTIPLIED
        }
        if (!m.width || !m.height || !m.depth) return HRESULT(0x8007000D);
        size_t full = 0;
        const bool okMips = (m.dimension == TEX_DIMENSION_TEXTURE3D) ? CalculateMipLevels3D(m.width, m.height, m.depth, full) : CalculateMipLevels(m.width, m.height, full);
        if (!okMips || m.mipLevels > full) return HRESULT(0x8007000D);
        return S_OK;
    }
}

HRESULT Blob::Initialize(size_t size) noexcept
{
    if (!size) return E_INVALIDARG;
    Release();
    m_buffer = static_cast<uint8_t*>(std::aligned_alloc(16, (size + 15) & ~size_t(15)));
    if (!m_buffer) return E_OUTOFMEMORY;
    m_size = size;
    return S_OK;
}
void Blob::Release() noexcept { if (m_buffer) { std::free(m_buffer); m_buffer = nullptr; } m_size = 0; }

HRESULT GetMetadataFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS, TexMetadata& metadata) noexcept
{
    size_t offset = 0;
    return DecodeHeader(pSource, size, metadata, offset);
}

HRESULT LoadFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS, TexMetadata* metadata, ScratchImage& image) noexcept
{
    image.Release();
    TexMetadata m; size_t offset = 0;
    HRESULT hr = DecodeHeader(pSource, size, m, offset);
    if (FAILED(hr)) return hr;
    hr = image.Initialize(m);
    if (FAILED(hr)) return hr;
    if (size - offset < image.GetPixelsSize()) { image.Release(); return HRESULT(0x80070026); }
    // the payload is laid out exactly like the ScratchImage (DirectXTexDDS.cpp:1706-1780 with matching pitches)
    std::memcpy(image.GetPixels(), static_cast<const uint8_t*>(pSource) + offset, image.GetPixelsSize());
    if (metadata) *metadata = image.GetMetadata();
    return S_OK;
}

HRESULT LoadFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept
{
    if (!szFile) return E_INVALIDARG;
    FILE* f = std::fopen(szFile, "rb");
    if (!f) return HRESULT(0x80070002);           // ERROR_FILE_NOT_FOUND
    std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(n > 0 ? size_t(n) : 0);
    const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) return E_FAIL;
    return LoadFromDDSMemory(buf.data(), buf.size(), flags, metadata, image);
}

HRESULT SaveToDDSMemory(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, Blob& blob) noexcept
{
    if (!images || !nimages) return E_INVALIDARG;
    if (BitsPerPixel(metadata.format) == 0) return E_INVALIDARG;
    const bool volume = metadata.dimension == TEX_DIMENSION_TEXTURE3D;
    if (volume && (metadata.depth > 0xFFFF || metadata.arraySize != 1)) return E_INVALIDARG;
    uint32_t fl = uint32_t(flags);
    const bool cube = (metadata.miscFlags & TEX_MISC_TEXTURECUBE) != 0;
    if (metadata.arraySize > 1 && !(metadata.arraySize == 6 && metadata.dimension == TEX_DIMENSION_TEXTURE2D && cube)) fl |= DDS_FLAGS_FORCE_DX10_EXT;
    if (fl & DDS_FLAGS_FORCE_DX10_EXT_MISC2) fl |= DDS_FLAGS_FORCE_DX10_EXT;
    const Legacy* legacy = nullptr;
    if (!(fl & DDS_FLAGS_FORCE_DX10_EXT))
        for (const Legacy& l : kLegacy)
            if (l.write && l.format == metadata.format) { legacy = &l; break; }
    if (metadata.mipLevels > 0xFFFF || metadata.arraySize > 0xFFFF) return E_INVALIDARG;
    if (metadata.width > 0xFFFFFFFFull || metadata.height > 0xFFFFFFFFull) return E_INVALIDARG;

    // every image with its default pitch, in ScratchImage order
    size_t payload = 0;
    std::vector<size_t> rp(nimages), sp(nimages);
    size_t expected = metadata.arraySize * (metadata.mipLevels ? metadata.mipLevels : 1);
    if (volume)
    {
        expected = 0;
        for (size_t l = 0, d = metadata.depth; l < (metadata.mipLevels ? metadata.mipLevels : 1); ++l) { expected += d; if (d > 1) d >>= 1; }
    }
    if (nimages != expected) return E_FAIL;
    for (size_t i = 0; i < nimages; ++i)
    {
        if (!images[i].pixels) return E_POINTER;
        if (images[i].format != metadata.format) return E_FAIL;
        const HRESULT hr = ComputePitch(metadata.format, images[i].width, images[i].height, rp[i], sp[i]);
        if (FAILED(hr)) return hr;
        payload += sp[i];
    }
    const size_t headerBytes = 4 + sizeof(Header) + (legacy ? 0 : sizeof(HeaderDX10));
    HRESULT hr = blob.Initialize(headerBytes + payload);
    if (FAILED(hr)) return hr;
    uint8_t* p = blob.GetBufferPointer();
    std::memcpy(p, &kMagic, 4);
    Header h; std::memset(&h, 0, sizeof(h));
    h.size = sizeof(Header); h.flags = HF_TEXTURE; h.caps = CAPS_TEXTURE;
    if (metadata.mipLevels > 0)
    {
        h.flags |= HF_MIPMAP; h.mipMapCount = uint32_t(metadata.mipLevels);
        if (h.mipMapCount > 1) h.caps |= CAPS_MIPMAP;
    }
    h.width = uint32_t(metadata.width);
    h.height = (metadata.dimension == TEX_DIMENSION_TEXTURE1D) ? 1u : uint32_t(metadata.height);
    h.depth = 1;
    if (volume) { h.flags |= HF_VOLUME; h.caps2 |= CAPS2_VOLUME; h.depth = uint32_t(metadata.depth); }      // :951-962
    if (metadata.dimension == TEX_DIMENSION_TEXTURE2D && cube) { h.caps |= CAPS_CUBEMAP; h.caps2 |= CAPS2_CUBEMAP_ALL; }
    size_t rp0, sp0;
    ComputePitch(metadata.format, metadata.width, metadata.height, rp0, sp0);
    if (rp0 > 0xFFFFFFFFull || sp0 > 0xFFFFFFFFull) { blob.Release(); return E_FAIL; }
    if (IsCompressed(metadata.format)) { h.flags |= HF_LINEARSIZE; h.pitchOrLinearSize = uint32_t(sp0); }
    else { h.flags |= HF_PITCH; h.pitchOrLinearSize = uint32_t(rp0); }
    if (legacy) h.pf = legacy->pf;
    else
    {
        const PixelFormat dx10 = { 32, FOURCC, cc('D', 'X', '1', '0'), 0, 0, 0, 0, 0 };
        h.pf = dx10;
        HeaderDX10 x; std::memset(&x, 0, sizeof(x));
        x.dxgiFormat = uint32_t(metadata.format); x.resourceDimension = uint32_t(metadata.dimension);
        x.miscFlag = metadata.miscFlags & ~uint32_t(TEX_MISC_TEXTURECUBE);
        if (cube)
        {
            if (metadata.arraySize % 6) { blob.Release(); return E_INVALIDARG; }
            x.miscFlag |= TEX_MISC_TEXTURECUBE; x.arraySize = uint32_t(metadata.arraySize / 6);
        }
        else x.arraySize = uint32_t(metadata.arraySize);
        if (fl & DDS_FLAGS_FORCE_DX10_EXT_MISC2) x.miscFlags2 = metadata.miscFlags2;
        std::memcpy(p + 4 + sizeof(Header), &x, sizeof(x));
    }
    std::memcpy(p + 4, &h, sizeof(h));
    uint8_t* d = p + headerBytes;
    for (size_t i = 0; i < nimages; ++i)
    {
        const size_t rows = sp[i] / rp[i];
        for (size_t y = 0; y < rows; ++y) std::memcpy(d + y * rp[i], images[i].pixels + y * images[i].rowPitch, rp[i] < images[i].rowPitch ? rp[i] : images[i].rowPitch);
        d += sp[i];
    }
    return S_OK;
}

HRESULT SaveToDDSFile(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, const char* szFile) noexcept
{
    if (!szFile) return E_INVALIDARG;
    Blob blob;
    const HRESULT hr = SaveToDDSMemory(images, nimages, metadata, flags, blob);
    if (FAILED(hr)) return hr;
    FILE* f = std::fopen(szFile, "wb");
    if (!f) return E_FAIL;
    const size_t n = std::fwrite(blob.GetBufferPointer(), 1, blob.GetBufferSize(), f);
    std::fclose(f);
    return n == blob.GetBufferSize() ? S_OK : E_FAIL;
}
} // namespace DirectXTexAMD
