// DDS reader / writer of the host layer (see DirectXTexAMD.h). Behaviour follows DirectXTexDDS.cpp: magic 0x20534444 + 124-byte
// DDS_HEADER (DDS.h:262-278) + optional 20-byte DDS_HEADER_DXT10 (:280-287); legacy (Direct3D 9) pixel formats are recognised by
// the rules of GetDXGIFormat (DirectXTexDDS.cpp:62-318) and brought to a DXGI format by the row conversions of CopyImage
// (:1040-1870); which formats are written with a legacy pixel format and which with the 'DX10' extension is EncodeDDSHeader's
// choice (:711-1033). The payload is the ScratchImage order (item-major, mips inside; volumes level by level).
// All of this is host code: it is the container either side of the GPU path, and tests/test_dds_cpu.py compares it with the
// reference's own reader and writer (oracle/_ref) file by file.
#include "DirectXTexAMD.h"

#include <algorithm>
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
    // DDS_PIXELFORMAT.flags
    constexpr uint32_t PF_ALPHAPIXELS = 0x1, PF_ALPHA = 0x2, PF_FOURCC = 0x4, PF_PAL8 = 0x20, PF_RGB = 0x40, PF_RGBA = 0x41, PF_LUM = 0x20000, PF_LUMA = 0x20001,
                       PF_BUMPLUM = 0x40000, PF_BUMPDUDV = 0x80000, PF_BUMPDUDVA = 0x80001, PF_NVTT_SRGB = 0x40000000, PF_NVTT_BITS = 0xC0000000;
    // DDS_HEADER.flags / caps / caps2
    constexpr uint32_t HF_TEXTURE = 0x1007, HF_HEIGHT = 0x2, HF_MIPMAP = 0x20000, HF_PITCH = 0x8, HF_LINEARSIZE = 0x80000, HF_VOLUME = 0x800000;
    constexpr uint32_t CAPS_TEXTURE = 0x1000, CAPS_MIPMAP = 0x400008, CAPS_CUBEMAP = 0x8, CAPS2_CUBEMAP_ALL = 0xFE00, CAPS2_CUBEMAP = 0x200, CAPS2_VOLUME = 0x200000;

#pragma pack(push, 1)
    struct PixelFormat { uint32_t size, flags, fourCC, bitCount, rMask, gMask, bMask, aMask; };
    struct Header { uint32_t size, flags, height, width, pitchOrLinearSize, depth, mipMapCount, reserved1[11]; PixelFormat pf; uint32_t caps, caps2, caps3, caps4, reserved2; };
    struct HeaderDX10 { uint32_t dxgiFormat, resourceDimension, miscFlag, arraySize, miscFlags2; };
#pragma pack(pop)
    static_assert(sizeof(Header) == 124 && sizeof(HeaderDX10) == 20, "DDS header layout");
    constexpr size_t kMinHeader = 4 + sizeof(Header), kDX10Header = kMinHeader + sizeof(HeaderDX10);

    constexpr uint32_t cc(char a, char b, char c, char d) { return uint32_t(uint8_t(a)) | (uint32_t(uint8_t(b)) << 8) | (uint32_t(uint8_t(c)) << 16) | (uint32_t(uint8_t(d)) << 24); }
    constexpr uint32_t kNVTT = cc('N', 'V', 'T', 'T');

    // What has to happen to the rows of a file on the way into the ScratchImage.
    enum : uint32_t
    {
        CV_EXPAND = 0x1,          // the texel grows (source rows are narrower than the result's)
        CV_NOALPHA = 0x2,         // alpha is forced to opaque
        CV_SWIZZLE = 0x4,         // red and blue trade places (or UYVY -> YUY2)
        CV_PAL8 = 0x8, CV_A8P8 = 0x800,          // 256-entry RGBA palette after the header; with an alpha byte per texel
        CV_888 = 0x10, CV_565 = 0x20, CV_5551 = 0x40, CV_4444 = 0x80, CV_44 = 0x100, CV_332 = 0x200, CV_8332 = 0x400,
        CV_ABGR4 = 0x1000,        // the 4:4:4:4 source is A4B4G4R4 (DXGI 191), not B4G4R4A4
        CV_DX10 = 0x10000, CV_PMALPHA = 0x20000,
        CV_L8 = 0x40000, CV_L16 = 0x80000, CV_A8L8 = 0x100000, CV_L6V5U5 = 0x200000, CV_L8U8V8 = 0x400000, CV_WUV10 = 0x800000,
    };

    // The legacy table, in the order GetDXGIFormat searches it (first hit wins; DirectXTexDDS.cpp:62-199).
    struct Legacy { DXGI_FORMAT format; uint32_t conv; PixelFormat pf; };
    constexpr PixelFormat four(uint32_t code) { return PixelFormat{ 32, PF_FOURCC, code, 0, 0, 0, 0, 0 }; }
    constexpr PixelFormat masks(uint32_t flags, uint32_t bits, uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return PixelFormat{ 32, flags, 0, bits, r, g, b, a }; }
    constexpr PixelFormat kA8R8G8B8 = masks(PF_RGBA, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0xff000000), kX8R8G8B8 = masks(PF_RGB, 32, 0x00ff0000, 0x0000ff00, 0x000000ff, 0),
                          kA8B8G8R8 = masks(PF_RGBA, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0xff000000), kR8G8B8 = masks(PF_RGB, 24, 0xff0000, 0x00ff00, 0x0000ff, 0),
                          kA2B10G10R10 = masks(PF_RGBA, 32, 0x3ff00000, 0x000ffc00, 0x000003ff, 0xc0000000);
    const Legacy kLegacy[] = {
        { DXGI_FORMAT_BC1_UNORM, 0, four(cc('D', 'X', 'T', '1')) },
        { DXGI_FORMAT_BC2_UNORM, 0, four(cc('D', 'X', 'T', '3')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('D', 'X', 'T', '5')) },
        { DXGI_FORMAT_BC2_UNORM, CV_PMALPHA, four(cc('D', 'X', 'T', '2')) },
        { DXGI_FORMAT_BC3_UNORM, CV_PMALPHA, four(cc('D', 'X', 'T', '4')) },
        // DXT5 with swizzled channels (normal-map tricks): handed over unchanged as BC3
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('A', '2', 'D', '5')) }, { DXGI_FORMAT_BC3_UNORM, 0, four(cc('x', 'G', 'B', 'R')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('R', 'x', 'B', 'G')) }, { DXGI_FORMAT_BC3_UNORM, 0, four(cc('R', 'B', 'x', 'G')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('x', 'R', 'B', 'G')) }, { DXGI_FORMAT_BC3_UNORM, 0, four(cc('R', 'G', 'x', 'B')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('x', 'G', 'x', 'R')) }, { DXGI_FORMAT_BC3_UNORM, 0, four(cc('G', 'X', 'R', 'B')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('G', 'R', 'X', 'B')) }, { DXGI_FORMAT_BC3_UNORM, 0, four(cc('R', 'X', 'G', 'B')) },
        { DXGI_FORMAT_BC3_UNORM, 0, four(cc('B', 'R', 'G', 'X')) },
        { DXGI_FORMAT_BC4_UNORM, 0, four(cc('B', 'C', '4', 'U')) }, { DXGI_FORMAT_BC4_SNORM, 0, four(cc('B', 'C', '4', 'S')) },
        { DXGI_FORMAT_BC5_UNORM, 0, four(cc('B', 'C', '5', 'U')) }, { DXGI_FORMAT_BC5_SNORM, 0, four(cc('B', 'C', '5', 'S')) },
        { DXGI_FORMAT_BC4_UNORM, 0, four(cc('A', 'T', 'I', '1')) }, { DXGI_FORMAT_BC5_UNORM, 0, four(cc('A', 'T', 'I', '2')) },
        { DXGI_FORMAT_BC5_UNORM, 0, four(cc('A', '2', 'X', 'Y')) },
        { DXGI_FORMAT_BC6H_UF16, 0, four(cc('B', 'C', '6', 'H')) }, { DXGI_FORMAT_BC7_UNORM, 0, four(cc('B', 'C', '7', 'L')) },
        { DXGI_FORMAT_BC7_UNORM, 0, four(cc('B', 'C', '7', '\0')) },
        { DXGI_FORMAT_R8G8_B8G8_UNORM, 0, four(cc('R', 'G', 'B', 'G')) }, { DXGI_FORMAT_G8R8_G8B8_UNORM, 0, four(cc('G', 'R', 'G', 'B')) },
        { DXGI_FORMAT_B8G8R8A8_UNORM, 0, kA8R8G8B8 },
        { DXGI_FORMAT_B8G8R8X8_UNORM, 0, kX8R8G8B8 },
        { DXGI_FORMAT_R8G8B8A8_UNORM, 0, kA8B8G8R8 },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_NOALPHA, masks(PF_RGB, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0) },
        { DXGI_FORMAT_R16G16_UNORM, 0, masks(PF_RGB, 32, 0x0000ffff, 0xffff0000, 0, 0) },
        // D3DX wrote the 10:10:10:2 masks the wrong way round, so the file that CLAIMS A2R10G10B10 is the one to leave alone
        { DXGI_FORMAT_R10G10B10A2_UNORM, CV_SWIZZLE, masks(PF_RGBA, 32, 0x000003ff, 0x000ffc00, 0x3ff00000, 0xc0000000) },
        { DXGI_FORMAT_R10G10B10A2_UNORM, 0, kA2B10G10R10 },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_EXPAND | CV_NOALPHA | CV_888, kR8G8B8 },
        { DXGI_FORMAT_B5G6R5_UNORM, CV_565, masks(PF_RGB, 16, 0xf800, 0x07e0, 0x001f, 0) },
        { DXGI_FORMAT_B5G5R5A1_UNORM, CV_5551, masks(PF_RGBA, 16, 0x7c00, 0x03e0, 0x001f, 0x8000) },
        { DXGI_FORMAT_B5G5R5A1_UNORM, CV_5551 | CV_NOALPHA, masks(PF_RGB, 16, 0x7c00, 0x03e0, 0x001f, 0) },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_EXPAND | CV_8332, masks(PF_RGBA, 16, 0x00e0, 0x001c, 0x0003, 0xff00) },
        { DXGI_FORMAT_B5G6R5_UNORM, CV_EXPAND | CV_332, masks(PF_RGB, 8, 0xe0, 0x1c, 0x03, 0) },
        { DXGI_FORMAT_R8_UNORM, 0, masks(PF_LUM, 8, 0xff, 0, 0, 0) },
        { DXGI_FORMAT_R16_UNORM, 0, masks(PF_LUM, 16, 0xffff, 0, 0, 0) },
        { DXGI_FORMAT_R8G8_UNORM, 0, masks(PF_LUMA, 16, 0x00ff, 0, 0, 0xff00) },
        { DXGI_FORMAT_R8G8_UNORM, 0, masks(PF_LUMA, 8, 0x00ff, 0, 0, 0xff00) },            // alternative bit count in the wild
        // NVTT 1 wrote luminance as RGB
        { DXGI_FORMAT_R8_UNORM, 0, masks(PF_RGB, 8, 0xff, 0, 0, 0) },
        { DXGI_FORMAT_R16_UNORM, 0, masks(PF_RGB, 16, 0xffff, 0, 0, 0) },
        { DXGI_FORMAT_R8G8_UNORM, 0, masks(PF_RGBA, 16, 0x00ff, 0, 0, 0xff00) },
        { DXGI_FORMAT_A8_UNORM, 0, masks(PF_ALPHA, 8, 0, 0, 0, 0xff) },
        // D3DX files carry the D3DFMT enum value as FourCC
        { DXGI_FORMAT_R16G16B16A16_UNORM, 0, four(36) }, { DXGI_FORMAT_R16G16B16A16_SNORM, 0, four(110) },
        { DXGI_FORMAT_R16_FLOAT, 0, four(111) }, { DXGI_FORMAT_R16G16_FLOAT, 0, four(112) }, { DXGI_FORMAT_R16G16B16A16_FLOAT, 0, four(113) },
        { DXGI_FORMAT_R32_FLOAT, 0, four(114) }, { DXGI_FORMAT_R32G32_FLOAT, 0, four(115) }, { DXGI_FORMAT_R32G32B32A32_FLOAT, 0, four(116) },
        { DXGI_FORMAT_R32_FLOAT, 0, masks(PF_RGB, 32, 0xffffffff, 0, 0, 0) },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_EXPAND | CV_PAL8 | CV_A8P8, masks(PF_PAL8 | PF_ALPHAPIXELS, 16, 0, 0, 0, 0xff00) },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_EXPAND | CV_PAL8, masks(PF_PAL8, 8, 0, 0, 0, 0) },
        { DXGI_FORMAT_B4G4R4A4_UNORM, CV_4444, masks(PF_RGBA, 16, 0x0f00, 0x00f0, 0x000f, 0xf000) },
        { DXGI_FORMAT_B4G4R4A4_UNORM, CV_NOALPHA | CV_4444, masks(PF_RGB, 16, 0x0f00, 0x00f0, 0x000f, 0) },
        { DXGI_FORMAT_B4G4R4A4_UNORM, CV_EXPAND | CV_44, masks(PF_LUMA, 8, 0x0f, 0, 0, 0xf0) },
        { DXGI_FORMAT_YUY2, 0, four(cc('Y', 'U', 'Y', '2')) }, { DXGI_FORMAT_YUY2, CV_SWIZZLE, four(cc('U', 'Y', 'V', 'Y')) },
        { DXGI_FORMAT_R8G8_SNORM, 0, masks(PF_BUMPDUDV, 16, 0x00ff, 0xff00, 0, 0) },
        { DXGI_FORMAT_R8G8B8A8_SNORM, 0, masks(PF_BUMPDUDV, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0xff000000) },
        { DXGI_FORMAT_R16G16_SNORM, 0, masks(PF_BUMPDUDV, 32, 0x0000ffff, 0xffff0000, 0, 0) },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_L6V5U5 | CV_EXPAND, masks(PF_BUMPLUM, 16, 0x001f, 0x03e0, 0xfc00, 0) },
        { DXGI_FORMAT_R8G8B8A8_UNORM, CV_L8U8V8, masks(PF_BUMPLUM, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0) },
        { DXGI_FORMAT_R10G10B10A2_UNORM, CV_WUV10, masks(PF_BUMPDUDVA, 32, 0x3ff00000, 0x000ffc00, 0x000003ff, 0xc0000000) },
    };

    // Does the file's pixel format name this table entry? Only the masks a kind of format defines are compared.
    bool Names(const Legacy& e, const PixelFormat& pf, uint32_t pfFlags, uint32_t& flags) noexcept
    {
        if ((pfFlags & PF_FOURCC) && (e.pf.flags & PF_FOURCC)) return pf.fourCC == e.pf.fourCC;       // other flag bits are ignored with a FourCC
        if (pfFlags != e.pf.flags || pf.bitCount != e.pf.bitCount) return false;
        if (e.pf.flags & PF_PAL8) return true;
        if (e.pf.flags & PF_ALPHA) return pf.aMask == e.pf.aMask;
        if (e.pf.flags & PF_LUM) return pf.rMask == e.pf.rMask && (!(e.pf.flags & PF_ALPHAPIXELS) || pf.aMask == e.pf.aMask);
        if (e.pf.flags & PF_BUMPDUDV)
        {
            if (!(e.pf.flags & PF_ALPHAPIXELS)) return pf.rMask == e.pf.rMask;
            if (pf.rMask != e.pf.rMask || pf.aMask != e.pf.aMask) return false;
            flags &= ~uint32_t(DDS_FLAGS_NO_R10B10G10A2_FIXUP);          // A2W10V10U10 is never "fixed up"
            return true;
        }
        if (pf.rMask != e.pf.rMask || pf.gMask != e.pf.gMask || pf.bMask != e.pf.bMask) return false;
        return !(e.pf.flags & PF_ALPHAPIXELS) || pf.aMask == e.pf.aMask;
    }

    DXGI_FORMAT LegacyFormat(const Header& h, uint32_t flags, uint32_t& conv) noexcept
    {
        const PixelFormat& pf = h.pf;
        const bool nvtt = h.reserved1[9] == kNVTT;
        const uint32_t pfFlags = nvtt ? (pf.flags & ~PF_NVTT_BITS) : pf.flags;            // NVTT's non-standard sRGB / normal-map bits
        const Legacy* hit = nullptr;
        if (pf.size == 0 && pf.flags == 0 && pf.fourCC != 0)
        {
            // files whose pixel format is blank except for the FourCC
            for (const Legacy& e : kLegacy)
                if ((e.pf.flags & PF_FOURCC) && e.pf.fourCC == pf.fourCC) { hit = &e; break; }
        }
        else
        {
            for (const Legacy& e : kLegacy)
                if (Names(e, pf, pfFlags, flags)) { hit = &e; break; }
        }
        if (!hit) return DXGI_FORMAT_UNKNOWN;
        uint32_t c = hit->conv;
        DXGI_FORMAT format = hit->format;
        if ((c & CV_EXPAND) && (flags & DDS_FLAGS_NO_LEGACY_EXPANSION)) return DXGI_FORMAT_UNKNOWN;
        if (format == DXGI_FORMAT_R10G10B10A2_UNORM && (flags & DDS_FLAGS_NO_R10B10G10A2_FIXUP)) c ^= CV_SWIZZLE;
        if (nvtt && (pf.flags & PF_NVTT_SRGB)) format = MakeSRGB(format);
        conv = c;
        return format;
    }

    // DecodeDDSHeader (DirectXTexDDS.cpp:324-694)
    HRESULT DecodeHeader(const void* pSource, size_t size, uint32_t flags, TexMetadata& m, DDSMetaData* ddpf, uint32_t& conv) noexcept
    {
        if (!pSource) return E_POINTER;
        m = TexMetadata();
        m.dimension = TEX_DIMENSION(0);
        if (ddpf) *ddpf = DDSMetaData{};
        if (size < kMinHeader) return HRESULT_E_INVALID_DATA;
        const uint8_t* p = static_cast<const uint8_t*>(pSource);
        uint32_t magic; std::memcpy(&magic, p, 4);
        if (magic != kMagic) return E_FAIL;
        Header h; std::memcpy(&h, p + 4, sizeof(h));
        const bool permissive = (flags & DDS_FLAGS_PERMISSIVE) != 0;
        // known variants: a header size of 24, a pixel format size of 0 or 24
        if (h.size != sizeof(Header) && !(permissive && h.size == 24)) return HRESULT_E_NOT_SUPPORTED;
        if (h.pf.size != sizeof(PixelFormat) && !(permissive && (h.pf.size == 0 || h.pf.size == 24))) return HRESULT_E_NOT_SUPPORTED;
        m.mipLevels = h.mipMapCount ? h.mipMapCount : 1;

        if ((h.pf.flags & PF_FOURCC) && h.pf.fourCC == cc('D', 'X', '1', '0'))
        {
            if (h.size != sizeof(Header) || h.pf.size != sizeof(PixelFormat)) return E_FAIL;          // no variants with the extension header
            if (size < kDX10Header) return E_FAIL;
            HeaderDX10 x; std::memcpy(&x, p + kMinHeader, sizeof(x));
            conv |= CV_DX10;
            m.arraySize = x.arraySize ? x.arraySize : 1;
            m.format = DXGI_FORMAT(x.dxgiFormat);
            if (!IsValid(m.format) || IsPalettized(m.format)) return HRESULT_E_NOT_SUPPORTED;
            m.miscFlags = x.miscFlag & ~uint32_t(TEX_MISC_TEXTURECUBE);
            switch (x.resourceDimension)
            {
            case TEX_DIMENSION_TEXTURE1D:
                if ((h.flags & HF_HEIGHT) && h.height != 1) return HRESULT_E_INVALID_DATA;          // D3DX writes 1D textures with a height of 1
                m.width = h.width; m.height = 1; m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE1D;
                break;
            case 0:                                     // dimension unknown: a known variant that means 2D
                if (!permissive) return HRESULT_E_INVALID_DATA;
                [[fallthrough]];
            case TEX_DIMENSION_TEXTURE2D:
                if (x.miscFlag & TEX_MISC_TEXTURECUBE) { m.miscFlags |= TEX_MISC_TEXTURECUBE; m.arraySize *= 6; }
                m.width = h.width; m.height = h.height; m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE2D;
                break;
            case TEX_DIMENSION_TEXTURE3D:
                if (!(h.flags & HF_VOLUME)) return HRESULT_E_INVALID_DATA;
                if (m.arraySize > 1) return HRESULT_E_NOT_SUPPORTED;
                m.width = h.width; m.height = h.height; m.depth = h.depth; m.dimension = TEX_DIMENSION_TEXTURE3D;
                break;
            default:
                return HRESULT_E_INVALID_DATA;
            }
            m.miscFlags2 = x.miscFlags2;
        }
        else
        {
            m.arraySize = 1;
            m.width = h.width; m.height = h.height;
            if (h.flags & HF_VOLUME) { m.depth = h.depth; m.dimension = TEX_DIMENSION_TEXTURE3D; }
            else
            {
                if (h.caps2 & CAPS2_CUBEMAP)
                {
                    if ((h.caps2 & CAPS2_CUBEMAP_ALL) != CAPS2_CUBEMAP_ALL) return HRESULT_E_NOT_SUPPORTED;      // all six faces required
                    m.arraySize = 6; m.miscFlags |= TEX_MISC_TEXTURECUBE;
                }
                m.depth = 1; m.dimension = TEX_DIMENSION_TEXTURE2D;           // a legacy file cannot say "1D"
            }
            if (permissive)
            {
                // tolerate a mip count that was computed wrongly
                size_t maxMips = 0;
                if (m.dimension == TEX_DIMENSION_TEXTURE3D) CalculateMipLevels3D(m.width, m.height, m.depth, maxMips);
                else CalculateMipLevels(m.width, m.height, maxMips);
                m.mipLevels = std::min(m.mipLevels, maxMips);
            }
            m.format = LegacyFormat(h, flags, conv);
            if (m.format == DXGI_FORMAT_UNKNOWN) return HRESULT_E_NOT_SUPPORTED;
            if (flags & DDS_FLAGS_EXPAND_LUMINANCE)
            {
                if (m.format == DXGI_FORMAT_R8_UNORM) { m.format = DXGI_FORMAT_R8G8B8A8_UNORM; conv |= CV_L8 | CV_EXPAND; }
                else if (m.format == DXGI_FORMAT_R8G8_UNORM) { m.format = DXGI_FORMAT_R8G8B8A8_UNORM; conv |= CV_A8L8 | CV_EXPAND; }
                else if (m.format == DXGI_FORMAT_R16_UNORM) { m.format = DXGI_FORMAT_R16G16B16A16_UNORM; conv |= CV_L16 | CV_EXPAND; }
            }
        }

        if (flags & DDS_FLAGS_FORCE_RGB)
        {
            switch (m.format)
            {
            case DXGI_FORMAT_B8G8R8A8_UNORM: m.format = DXGI_FORMAT_R8G8B8A8_UNORM; conv |= CV_SWIZZLE; break;
            case DXGI_FORMAT_B8G8R8X8_UNORM: m.format = DXGI_FORMAT_R8G8B8A8_UNORM; conv |= CV_SWIZZLE | CV_NOALPHA; break;
            case DXGI_FORMAT_B8G8R8A8_TYPELESS: m.format = DXGI_FORMAT_R8G8B8A8_TYPELESS; conv |= CV_SWIZZLE; break;
            case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: m.format = DXGI_FORMAT_R8G8B8A8_UNORM_SRGB; conv |= CV_SWIZZLE; break;
            case DXGI_FORMAT_B8G8R8X8_TYPELESS: m.format = DXGI_FORMAT_R8G8B8A8_TYPELESS; conv |= CV_SWIZZLE | CV_NOALPHA; break;
            case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: m.format = DXGI_FORMAT_R8G8B8A8_UNORM_SRGB; conv |= CV_SWIZZLE | CV_NOALPHA; break;
            default: break;
            }
        }
        if (flags & DDS_FLAGS_NO_16BPP)
        {
            switch (m.format)
            {
            case DXGI_FORMAT_B5G6R5_UNORM: case DXGI_FORMAT_B5G5R5A1_UNORM: case DXGI_FORMAT_B4G4R4A4_UNORM: case DXGI_FORMAT_A4B4G4R4_UNORM:
                if (m.format == DXGI_FORMAT_B5G6R5_UNORM) conv |= CV_NOALPHA;
                if (m.format == DXGI_FORMAT_A4B4G4R4_UNORM) conv |= CV_4444 | CV_ABGR4;
                m.format = DXGI_FORMAT_R8G8B8A8_UNORM;
                conv |= CV_EXPAND;
                break;
            default: break;
            }
        }
        if (conv & CV_NOALPHA) m.SetAlphaMode(TEX_ALPHA_MODE_OPAQUE);
        else if (conv & CV_PMALPHA) m.SetAlphaMode(TEX_ALPHA_MODE_PREMULTIPLIED);

        // beyond what Direct3D hardware has to support (16k textures, 15 mips, 2048 array items / depth slices)
        if (!(flags & DDS_FLAGS_ALLOW_LARGE_FILES))
            if (m.width > 16384u || m.height > 16384u || m.mipLevels > 15u || m.arraySize > 2048u || m.depth > 2048u) return HRESULT_E_NOT_SUPPORTED;
        if ((flags & DDS_FLAGS_IGNORE_MIPS) && m.arraySize == 1) m.mipLevels = 1;
        if (ddpf)
        {
            ddpf->size = h.pf.size; ddpf->flags = h.pf.flags; ddpf->fourCC = h.pf.fourCC; ddpf->RGBBitCount = h.pf.bitCount;
            ddpf->RBitMask = h.pf.rMask; ddpf->GBitMask = h.pf.gMask; ddpf->BBitMask = h.pf.bMask; ddpf->ABitMask = h.pf.aMask;
        }
        return S_OK;
    }

    // ---- row conversions ------------------------------------------------------------------------------------------------
    inline uint32_t rd16(const uint8_t* p) noexcept { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
    inline uint32_t rd32(const uint8_t* p) noexcept { uint32_t v; std::memcpy(&v, p, 4); return v; }
    inline void wr16(uint8_t* p, uint32_t v) noexcept { p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); }
    inline void wr32(uint8_t* p, uint32_t v) noexcept { std::memcpy(p, &v, 4); }
    // an n-bit channel widened to `to` bits by repeating its bit pattern (what all the legacy expansions do)
    inline uint32_t widen(uint32_t v, unsigned n, unsigned to) noexcept
    {
        uint32_t out = 0; unsigned have = 0;
        while (have < to) { out = (out << n) | v; have += n; }
        return out >> (have - to);
    }
    inline uint32_t rgba8(uint32_t r, uint32_t g, uint32_t b, uint32_t a) noexcept { return r | (g << 8) | (b << 16) | (a << 24); }
    inline uint32_t flip(uint32_t v, unsigned bits) noexcept { return v ^ (1u << (bits - 1)); }        // two's complement -> offset binary

    enum RowOp { ROW_COPY, ROW_SWIZZLE, ROW_565, ROW_5551, ROW_4444, ROW_ABGR4, ROW_888, ROW_332, ROW_8332, ROW_P8, ROW_A8P8, ROW_44, ROW_L8, ROW_L16,
                 ROW_A8L8, ROW_L6V5U5, ROW_L8U8V8, ROW_WUV10, ROW_FAIL };

    // which conversion a row of this file gets (the dispatch of CopyImage, DirectXTexDDS.cpp:1644-1691)
    RowOp PickRowOp(uint32_t conv) noexcept
    {
        if (conv & CV_EXPAND)
        {
            if (conv & CV_4444) return (conv & CV_ABGR4) ? ROW_ABGR4 : ROW_4444;
            if (conv & CV_565) return ROW_565;
            if (conv & CV_5551) return ROW_5551;
            if (conv & CV_PAL8) return (conv & CV_A8P8) ? ROW_A8P8 : ROW_P8;
            if (conv & CV_888) return ROW_888;
            if (conv & CV_332) return ROW_332;
            if (conv & CV_8332) return ROW_8332;
            if (conv & CV_44) return ROW_44;
            if (conv & CV_L8) return ROW_L8;
            if (conv & CV_L16) return ROW_L16;
            if (conv & CV_A8L8) return ROW_A8L8;
            if (conv & CV_L6V5U5) return ROW_L6V5U5;
            return ROW_FAIL;               // an expansion nobody implements (e.g. a 'DX10' B5G6R5 file read with NO_16BPP)
        }
        if (conv & CV_SWIZZLE) return ROW_SWIZZLE;
        if (conv & CV_L8U8V8) return ROW_L8U8V8;
        if (conv & CV_WUV10) return ROW_WUV10;
        return ROW_COPY;
    }

    inline bool Is8888(DXGI_FORMAT f) noexcept
    {
        switch (f)
        {
        case DXGI_FORMAT_R8G8B8A8_TYPELESS: case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB:
        case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8A8_TYPELESS: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB:
        case DXGI_FORMAT_B8G8R8X8_TYPELESS: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB:
            return true;
        default: return false;
        }
    }
    inline bool Is1010102(DXGI_FORMAT f) noexcept
    {
        return f == DXGI_FORMAT_R10G10B10A2_TYPELESS || f == DXGI_FORMAT_R10G10B10A2_UNORM || f == DXGI_FORMAT_R10G10B10A2_UINT
            || f == DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM || uint32_t(f) == 189;
    }

    // CopyScanline with TEXP_SCANLINE_SETALPHA (DirectXTexConvert.cpp:207-430): the formats whose alpha can be forced opaque
    void CopyOpaque(uint8_t* d, size_t dn, const uint8_t* s, size_t sn, DXGI_FORMAT f) noexcept
    {
        const size_t n = std::min(dn, sn);
        switch (uint32_t(f))
        {
        case DXGI_FORMAT_R32G32B32A32_TYPELESS: case DXGI_FORMAT_R32G32B32A32_FLOAT: case DXGI_FORMAT_R32G32B32A32_UINT: case DXGI_FORMAT_R32G32B32A32_SINT:
        {
            const uint32_t a = (f == DXGI_FORMAT_R32G32B32A32_FLOAT) ? 0x3f800000u : (f == DXGI_FORMAT_R32G32B32A32_SINT) ? 0x7fffffffu : 0xffffffffu;
            for (size_t i = 0; i + 16 <= n; i += 16) { std::memcpy(d + i, s + i, 12); wr32(d + i + 12, a); }
            return;
        }
        case DXGI_FORMAT_R16G16B16A16_TYPELESS: case DXGI_FORMAT_R16G16B16A16_FLOAT: case DXGI_FORMAT_R16G16B16A16_UNORM: case DXGI_FORMAT_R16G16B16A16_UINT:
        case DXGI_FORMAT_R16G16B16A16_SNORM: case DXGI_FORMAT_R16G16B16A16_SINT: case DXGI_FORMAT_Y416:
        {
            const uint32_t a = (f == DXGI_FORMAT_R16G16B16A16_FLOAT) ? 0x3c00u : (f == DXGI_FORMAT_R16G16B16A16_SNORM || f == DXGI_FORMAT_R16G16B16A16_SINT) ? 0x7fffu : 0xffffu;
            for (size_t i = 0; i + 8 <= n; i += 8) { std::memcpy(d + i, s + i, 6); wr16(d + i + 6, a); }
            return;
        }
        case DXGI_FORMAT_R10G10B10A2_TYPELESS: case DXGI_FORMAT_R10G10B10A2_UNORM: case DXGI_FORMAT_R10G10B10A2_UINT: case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM:
        case DXGI_FORMAT_Y410: case 116: case 117: case 189:
            for (size_t i = 0; i + 4 <= n; i += 4) wr32(d + i, rd32(s + i) | 0xC0000000u);
            return;
        case DXGI_FORMAT_R8G8B8A8_TYPELESS: case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: case DXGI_FORMAT_R8G8B8A8_UINT:
        case DXGI_FORMAT_R8G8B8A8_SNORM: case DXGI_FORMAT_R8G8B8A8_SINT: case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_TYPELESS:
        case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: case DXGI_FORMAT_AYUV:
        {
            const uint32_t a = (f == DXGI_FORMAT_R8G8B8A8_SNORM || f == DXGI_FORMAT_R8G8B8A8_SINT) ? 0x7f000000u : 0xff000000u;
            for (size_t i = 0; i + 4 <= n; i += 4) wr32(d + i, (rd32(s + i) & 0xFFFFFFu) | a);
            return;
        }
        case DXGI_FORMAT_B5G5R5A1_UNORM: case DXGI_FORMAT_B4G4R4A4_UNORM: case DXGI_FORMAT_A4B4G4R4_UNORM:
        {
            const uint32_t a = (f == DXGI_FORMAT_B4G4R4A4_UNORM) ? 0xF000u : (f == DXGI_FORMAT_A4B4G4R4_UNORM) ? 0x000Fu : 0x8000u;
            for (size_t i = 0; i + 2 <= n; i += 2) wr16(d + i, rd16(s + i) | a);
            return;
        }
        case DXGI_FORMAT_A8_UNORM:
            std::memset(d, 0xff, dn);
            return;
        default:
            std::memcpy(d, s, n);
            return;
        }
    }

    // One row. Returns false where the reference's expansion refuses the (source, result format) pair.
    bool ConvertRow(RowOp op, uint8_t* d, size_t dn, const uint8_t* s, size_t sn, DXGI_FORMAT outFormat, bool opaque, const uint32_t* pal8) noexcept
    {
        const bool to8888 = outFormat == DXGI_FORMAT_R8G8B8A8_UNORM;
        switch (op)
        {
        case ROW_COPY:
            if (opaque) CopyOpaque(d, dn, s, sn, outFormat);
            else std::memcpy(d, s, std::min(dn, sn));
            return true;
        case ROW_SWIZZLE:           // SwizzleScanline with TEXP_SCANLINE_LEGACY (DirectXTexConvert.cpp:440-605)
        {
            const size_t n = std::min(dn, sn);
            if (Is1010102(outFormat))
                for (size_t i = 0; i + 4 <= n; i += 4)
                {
                    const uint32_t t = rd32(s + i);
                    wr32(d + i, ((t >> 20) & 0x3ffu) | ((t & 0x3ffu) << 20) | (t & 0x000ffc00u) | (opaque ? 0xC0000000u : (t & 0xC0000000u)));
                }
            else if (Is8888(outFormat))
                for (size_t i = 0; i + 4 <= n; i += 4)
                {
                    const uint32_t t = rd32(s + i);
                    wr32(d + i, ((t >> 16) & 0xffu) | ((t & 0xffu) << 16) | (t & 0x0000ff00u) | (opaque ? 0xff000000u : (t & 0xff000000u)));
                }
            else if (outFormat == DXGI_FORMAT_YUY2)         // UYVY -> YUY2
                for (size_t i = 0; i + 4 <= n; i += 4)
                {
                    const uint32_t t = rd32(s + i);
                    wr32(d + i, ((t & 0x00ff00ffu) << 8) | ((t & 0xff00ff00u) >> 8));
                }
            else std::memcpy(d, s, n);
            return true;
        }
        case ROW_565:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4)
            {
                const uint32_t t = rd16(s + i);
                // sic (ExpandScanline, DirectXTexConvert.cpp:643): the two bits that should fill green's low end are shifted into
                // bits 4-5 of the word - the red byte - so green keeps zeros there and red picks them up
                const uint32_t g6 = (t >> 5) & 0x3f;
                wr32(d + o, rgba8(widen(t >> 11, 5, 8) | ((g6 >> 4) << 4), g6 << 2, widen(t & 0x1f, 5, 8), 0xff));
            }
            return true;
        case ROW_5551:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4)
            {
                const uint32_t t = rd16(s + i);
                wr32(d + o, rgba8(widen((t >> 10) & 0x1f, 5, 8), widen((t >> 5) & 0x1f, 5, 8), widen(t & 0x1f, 5, 8), (opaque || (t & 0x8000)) ? 0xff : 0));
            }
            return true;
        case ROW_4444: case ROW_ABGR4:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4)
            {
                const uint32_t t = rd16(s + i);
                // nibbles from the top: B4G4R4A4 = A R G B, A4B4G4R4 = R G B A
                const uint32_t n3 = t >> 12, n2 = (t >> 8) & 0xf, n1 = (t >> 4) & 0xf, n0 = t & 0xf;
                if (op == ROW_4444) wr32(d + o, rgba8(n2 * 17, n1 * 17, n0 * 17, opaque ? 0xff : n3 * 17));
                else wr32(d + o, rgba8(n3 * 17, n2 * 17, n1 * 17, opaque ? 0xff : n0 * 17));
            }
            return true;
        case ROW_888:                // 24 bpp files are B, G, R in memory
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 3 <= sn && o + 4 <= dn; i += 3, o += 4) wr32(d + o, rgba8(s[i + 2], s[i + 1], s[i], 0xff));
            return true;
        case ROW_332:
            if (to8888)
                for (size_t i = 0, o = 0; i < sn && o + 4 <= dn; ++i, o += 4)
                    wr32(d + o, rgba8(widen(s[i] >> 5, 3, 8), widen((s[i] >> 2) & 7, 3, 8), widen(s[i] & 3, 2, 8), 0xff));
            else if (outFormat == DXGI_FORMAT_B5G6R5_UNORM)
                for (size_t i = 0, o = 0; i < sn && o + 2 <= dn; ++i, o += 2)
                    wr16(d + o, (widen(s[i] >> 5, 3, 5) << 11) | (widen((s[i] >> 2) & 7, 3, 6) << 5) | widen(s[i] & 3, 2, 5));
            else return false;
            return true;
        case ROW_8332:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4)
            {
                const uint32_t t = rd16(s + i);
                wr32(d + o, rgba8(widen((t >> 5) & 7, 3, 8), widen((t >> 2) & 7, 3, 8), widen(t & 3, 2, 8), opaque ? 0xff : (t >> 8)));
            }
            return true;
        case ROW_P8:
            if (!to8888 || !pal8) return false;
            for (size_t i = 0, o = 0; i < sn && o + 4 <= dn; ++i, o += 4) wr32(d + o, pal8[s[i]]);
            return true;
        case ROW_A8P8:               // the texel's alpha is OR-ed onto the palette entry's
            if (!to8888 || !pal8) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4) wr32(d + o, pal8[s[i]] | (opaque ? 0xff000000u : (uint32_t(s[i + 1]) << 24)));
            return true;
        case ROW_44:
            if (to8888)
                for (size_t i = 0, o = 0; i < sn && o + 4 <= dn; ++i, o += 4)
                {
                    const uint32_t l = (s[i] & 0xfu) * 17;
                    wr32(d + o, rgba8(l, l, l, opaque ? 0xff : (s[i] >> 4) * 17u));
                }
            else if (outFormat == DXGI_FORMAT_B4G4R4A4_UNORM)
                for (size_t i = 0, o = 0; i < sn && o + 2 <= dn; ++i, o += 2)
                {
                    const uint32_t l = s[i] & 0xfu;
                    wr16(d + o, l | (l << 4) | (l << 8) | (opaque ? 0xf000u : (uint32_t(s[i] >> 4) << 12)));
                }
            else return false;
            return true;
        case ROW_L8:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i < sn && o + 4 <= dn; ++i, o += 4) wr32(d + o, rgba8(s[i], s[i], s[i], 0xff));
            return true;
        case ROW_L16:
            if (outFormat != DXGI_FORMAT_R16G16B16A16_UNORM) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 8 <= dn; i += 2, o += 8)
            {
                const uint32_t l = rd16(s + i);
                wr16(d + o, l); wr16(d + o + 2, l); wr16(d + o + 4, l); wr16(d + o + 6, 0xffff);
            }
            return true;
        case ROW_A8L8:
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4) wr32(d + o, rgba8(s[i], s[i], s[i], opaque ? 0xff : s[i + 1]));
            return true;
        case ROW_L6V5U5:             // unsigned 6-bit luminance, signed 5-bit v and u -> (L, U, V, 1) as unsigned bytes
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 2 <= sn && o + 4 <= dn; i += 2, o += 4)
            {
                const uint32_t t = rd16(s + i);
                wr32(d + o, rgba8(widen(t >> 10, 6, 8), widen(flip(t & 0x1f, 5), 5, 8), widen(flip((t >> 5) & 0x1f, 5), 5, 8), 0xff));
            }
            return true;
        case ROW_L8U8V8:             // X8L8V8U8: (L, U, V, 1) with the signed bytes rebased to unsigned
            if (!to8888) return false;
            for (size_t i = 0, o = 0; i + 4 <= sn && o + 4 <= dn; i += 4, o += 4)
            {
                const uint32_t t = rd32(s + i);
                wr32(d + o, rgba8((t >> 16) & 0xff, flip(t & 0xff, 8), flip((t >> 8) & 0xff, 8), 0xff));
            }
            return true;
        case ROW_WUV10:              // A2W10V10U10: three signed 10-bit fields rebased to unsigned, alpha kept
            if (outFormat != DXGI_FORMAT_R10G10B10A2_UNORM) return false;
            for (size_t i = 0, o = 0; i + 4 <= sn && o + 4 <= dn; i += 4, o += 4)
            {
                const uint32_t t = rd32(s + i);
                wr32(d + o, flip(t & 0x3ff, 10) | (flip((t >> 10) & 0x3ff, 10) << 10) | (flip((t >> 20) & 0x3ff, 10) << 20) | (opaque ? 0xC0000000u : (t & 0xC0000000u)));
            }
            return true;
        default:
            return false;
        }
    }

    // CopyImage (DirectXTexDDS.cpp:1518-1808): the payload, laid out by the FILE's pitch rule, into the ScratchImage
    HRESULT CopyPayload(const uint8_t* pixels, size_t size, const TexMetadata& m, uint32_t cp, uint32_t conv, const uint32_t* pal8, const ScratchImage& image) noexcept
    {
        if (!size) return E_FAIL;
        if (conv & CV_EXPAND)
        {
            if (conv & CV_888) cp |= CP_FLAGS_24BPP;
            else if (conv & (CV_565 | CV_5551 | CV_4444 | CV_8332 | CV_A8P8 | CV_L16 | CV_A8L8 | CV_L6V5U5)) cp |= CP_FLAGS_16BPP;
            else if (conv & (CV_44 | CV_332 | CV_PAL8 | CV_L8)) cp |= CP_FLAGS_8BPP;
        }
        // where each image of the file starts, and its pitches
        struct Src { size_t offset, rowPitch, slicePitch; };
        std::vector<Src> src;
        const bool volume = m.dimension == TEX_DIMENSION_TEXTURE3D;
        uint64_t total = 0;
        for (size_t item = 0; item < (volume ? 1 : m.arraySize); ++item)
        {
            size_t w = m.width, h = m.height, d = volume ? m.depth : 1;
            for (size_t level = 0; level < m.mipLevels; ++level)
            {
                size_t rp, sp;
                const HRESULT hr = ComputePitch(m.format, w, h, rp, sp, CP_FLAGS(cp));
                if (FAILED(hr)) return hr;
                for (size_t slice = 0; slice < d; ++slice)
                {
                    if (sp > UINT64_MAX - total) return HRESULT_E_HANDLE_EOF;          // no file holds that much
                    src.push_back({ size_t(total), rp, sp }); total += sp;
                }
                if (h > 1) h >>= 1;
                if (w > 1) w >>= 1;
                if (d > 1) d >>= 1;
            }
        }
        if (src.empty() || src.size() != image.GetImageCount()) return E_FAIL;
        if (total > size) return HRESULT_E_HANDLE_EOF;
        const Image* images = image.GetImages();
        if (!images) return E_FAIL;
        if (m.dimension != TEX_DIMENSION_TEXTURE1D && m.dimension != TEX_DIMENSION_TEXTURE2D && !volume) return E_FAIL;

        const bool opaque = (conv & CV_NOALPHA) != 0;
        const RowOp op = PickRowOp(conv);
        const bool compressed = IsCompressed(m.format), planar = IsPlanar(m.format);
        if (planar && volume) return HRESULT_E_NOT_SUPPORTED;
        size_t index = 0;
        const size_t chains = volume ? 1 : m.arraySize;
        for (size_t item = 0; item < chains; ++item)
        {
            size_t lastgood = 0, d = volume ? m.depth : 1;          // sic: the reference restarts at image 0 for every array item
            for (size_t level = 0; level < m.mipLevels; ++level)
            {
                for (size_t slice = 0; slice < d; ++slice, ++index)
                {
                    const Image& dst = images[index];
                    const Src& from = src[index];
                    const uint8_t* sp = pixels + from.offset;
                    if (compressed)
                    {
                        std::memcpy(dst.pixels, sp, std::min(dst.slicePitch, from.slicePitch));
                        if (cp & CP_FLAGS_BAD_DXTN_TAILS)
                        {
                            // levels smaller than a block were not stored: take the bytes of the last level that was
                            if (dst.width < 4 || dst.height < 4)
                            {
                                const Src& good = src[lastgood + (volume ? slice : 0)];
                                std::memcpy(dst.pixels, pixels + good.offset, std::min(dst.slicePitch, good.slicePitch));
                            }
                            else if (!volume || slice == 0) lastgood = index;
                        }
                        continue;
                    }
                    const size_t rows = planar ? ComputeScanlines(m.format, dst.height) : dst.height;
                    if (planar && !rows) return E_FAIL;
                    uint8_t* dp = dst.pixels;
                    for (size_t y = 0; y < rows; ++y, sp += from.rowPitch, dp += dst.rowPitch)
                    {
                        if (planar) std::memcpy(dp, sp, std::min(dst.rowPitch, from.rowPitch));
                        else if (!ConvertRow(op, dp, dst.rowPitch, sp, from.rowPitch, m.format, opaque, pal8)) return E_FAIL;
                    }
                }
                if (d > 1) d >>= 1;
            }
        }
        return S_OK;
    }

    HRESULT ReadWholeFile(const char* szFile, std::vector<uint8_t>& buf) noexcept
    {
        FILE* f = std::fopen(szFile, "rb");
        if (!f) return E_FAIL;
        std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        if (n < 0) { std::fclose(f); return E_FAIL; }
        if (uint64_t(n) > UINT32_MAX) { std::fclose(f); return HRESULT_E_FILE_TOO_LARGE; }
        try { buf.resize(size_t(n)); } catch (...) { std::fclose(f); return E_OUTOFMEMORY; }
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        if (got != buf.size()) return E_FAIL;
        if (buf.size() < kMinHeader) return E_FAIL;               // the file readers say E_FAIL here (DirectXTexDDS.cpp:1975-1978, :2178-2181)
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

// ---- reading --------------------------------------------------------------------------------------------------------------
HRESULT GetMetadataFromDDSMemoryEx(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata& metadata, DDSMetaData* ddPixelFormat) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    uint32_t conv = 0;
    return DecodeHeader(pSource, size, uint32_t(flags), metadata, ddPixelFormat, conv);
}

HRESULT GetMetadataFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata& metadata) noexcept
{
    return GetMetadataFromDDSMemoryEx(pSource, size, flags, metadata, nullptr);
}

HRESULT GetMetadataFromDDSFileEx(const char* szFile, DDS_FLAGS flags, TexMetadata& metadata, DDSMetaData* ddPixelFormat) noexcept
{
    if (!szFile) return E_INVALIDARG;
    FILE* f = std::fopen(szFile, "rb");
    if (!f) return E_FAIL;
    uint8_t header[kDX10Header] = {};
    const size_t got = std::fread(header, 1, sizeof(header), f);
    std::fclose(f);
    if (got < kMinHeader) return E_FAIL;
    uint32_t conv = 0;
    return DecodeHeader(header, got, uint32_t(flags), metadata, ddPixelFormat, conv);
}

HRESULT GetMetadataFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata& metadata) noexcept
{
    return GetMetadataFromDDSFileEx(szFile, flags, metadata, nullptr);
}

// LoadFromDDSMemoryEx (DirectXTexDDS.cpp:2019-2107)
HRESULT LoadFromDDSMemoryEx(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata* metadata, DDSMetaData* ddPixelFormat, ScratchImage& image) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    image.Release();
    uint32_t conv = 0;
    TexMetadata m;
    HRESULT hr = DecodeHeader(pSource, size, uint32_t(flags), m, ddPixelFormat, conv);
    if (FAILED(hr)) return hr;
    size_t offset = (conv & CV_DX10) ? kDX10Header : kMinHeader;
    const uint8_t* bytes = static_cast<const uint8_t*>(pSource);
    std::vector<uint32_t> pal8;
    if (conv & CV_PAL8)
    {
        if (size < offset + 256 * sizeof(uint32_t)) return E_FAIL;
        try { pal8.resize(256); } catch (...) { return E_OUTOFMEMORY; }
        std::memcpy(pal8.data(), bytes + offset, 256 * sizeof(uint32_t));
        offset += 256 * sizeof(uint32_t);
    }
    const size_t remaining = size - offset;
    if (remaining == 0) return E_FAIL;
    hr = image.Initialize(m);
    if (FAILED(hr)) return hr;
    if ((flags & DDS_FLAGS_PERMISSIVE) && (m.miscFlags & TEX_MISC_TEXTURECUBE) && (conv & CV_DX10) && image.GetPixelsSize() > remaining && (m.arraySize % 6) == 0)
    {
        // a writer that stored 6 * cubes where the number of cubes belongs
        m.arraySize /= 6;
        hr = image.Initialize(m);
        if (FAILED(hr)) return hr;
        if (image.GetPixelsSize() > remaining) { image.Release(); return HRESULT_E_HANDLE_EOF; }
    }
    uint32_t cp = 0;
    if (flags & DDS_FLAGS_LEGACY_DWORD) cp |= CP_FLAGS_LEGACY_DWORD;
    if (flags & DDS_FLAGS_BAD_DXTN_TAILS) cp |= CP_FLAGS_BAD_DXTN_TAILS;
    hr = CopyPayload(bytes + offset, remaining, m, cp, conv, pal8.empty() ? nullptr : pal8.data(), image);
    if (FAILED(hr)) { image.Release(); return hr; }
    if (metadata) *metadata = m;
    return S_OK;
}

HRESULT LoadFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept
{
    return LoadFromDDSMemoryEx(pSource, size, flags, metadata, nullptr, image);
}

HRESULT LoadFromDDSFileEx(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, DDSMetaData* ddPixelFormat, ScratchImage& image) noexcept
{
    if (!szFile) return E_INVALIDARG;
    image.Release();
    std::vector<uint8_t> buf;
    const HRESULT hr = ReadWholeFile(szFile, buf);
    if (FAILED(hr)) return hr;
    return LoadFromDDSMemoryEx(buf.data(), buf.size(), flags, metadata, ddPixelFormat, image);
}

HRESULT LoadFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept
{
    return LoadFromDDSFileEx(szFile, flags, metadata, nullptr, image);
}

// ---- writing --------------------------------------------------------------------------------------------------------------
// EncodeDDSHeader (DirectXTexDDS.cpp:711-1033)
HRESULT EncodeDDSHeader(const TexMetadata& metadata, DDS_FLAGS ddsFlags, uint8_t* pDestination, size_t maxsize, size_t& required) noexcept
{
    if (!IsValid(metadata.format)) return E_INVALIDARG;
    if (IsPalettized(metadata.format)) return HRESULT_E_NOT_SUPPORTED;
    uint32_t flags = uint32_t(ddsFlags);
    const bool dx9 = (flags & DDS_FLAGS_FORCE_DX9_LEGACY) != 0;
    if (metadata.arraySize > 1 && (metadata.arraySize != 6 || metadata.dimension != TEX_DIMENSION_TEXTURE2D || !metadata.IsCubemap()))
    {
        // 1D / 2D arrays and cubemap arrays need the 'DX10' header
        if (dx9) return HRESULT_E_CANNOT_MAKE;
        flags |= DDS_FLAGS_FORCE_DX10_EXT;
    }
    if (flags & DDS_FLAGS_FORCE_DX10_EXT_MISC2) flags |= DDS_FLAGS_FORCE_DX10_EXT;

    // the legacy pixel format, where one is written
    PixelFormat pf = {};
    uint32_t pitchFlags = CP_FLAGS_NONE;
    if (!(flags & DDS_FLAGS_FORCE_DX10_EXT))
    {
        const bool pm = metadata.IsPMAlpha();
        switch (metadata.format)
        {
        case DXGI_FORMAT_R8G8B8A8_UNORM: pf = kA8B8G8R8; break;
        case DXGI_FORMAT_R16G16_UNORM: pf = masks(PF_RGB, 32, 0x0000ffff, 0xffff0000, 0, 0); break;
        case DXGI_FORMAT_R8G8_UNORM: pf = masks(PF_LUMA, 16, 0x00ff, 0, 0, 0xff00); break;
        case DXGI_FORMAT_R16_UNORM: pf = masks(PF_LUM, 16, 0xffff, 0, 0, 0); break;
        case DXGI_FORMAT_R8_UNORM: pf = masks(PF_LUM, 8, 0xff, 0, 0, 0); break;
        case DXGI_FORMAT_A8_UNORM: pf = masks(PF_ALPHA, 8, 0, 0, 0, 0xff); break;
        case DXGI_FORMAT_R8G8_B8G8_UNORM: pf = four(cc('R', 'G', 'B', 'G')); break;
        case DXGI_FORMAT_G8R8_G8B8_UNORM: pf = four(cc('G', 'R', 'G', 'B')); break;
        case DXGI_FORMAT_BC1_UNORM: pf = four(cc('D', 'X', 'T', '1')); break;
        case DXGI_FORMAT_BC2_UNORM: pf = four(pm ? cc('D', 'X', 'T', '2') : cc('D', 'X', 'T', '3')); break;
        case DXGI_FORMAT_BC3_UNORM:
            pf = four(pm ? cc('D', 'X', 'T', '4') : cc('D', 'X', 'T', '5'));
            if (flags & DDS_FLAGS_FORCE_DXT5_RXGB) pf.fourCC = cc('R', 'X', 'G', 'B');
            break;
        case DXGI_FORMAT_BC4_UNORM: pf = four(dx9 ? cc('A', 'T', 'I', '1') : cc('B', 'C', '4', 'U')); break;
        case DXGI_FORMAT_BC4_SNORM: pf = four(cc('B', 'C', '4', 'S')); break;
        case DXGI_FORMAT_BC5_UNORM: pf = four(dx9 ? cc('A', 'T', 'I', '2') : cc('B', 'C', '5', 'U')); break;
        case DXGI_FORMAT_BC5_SNORM: pf = four(cc('B', 'C', '5', 'S')); break;
        case DXGI_FORMAT_B5G6R5_UNORM: pf = masks(PF_RGB, 16, 0xf800, 0x07e0, 0x001f, 0); break;
        case DXGI_FORMAT_B5G5R5A1_UNORM: pf = masks(PF_RGBA, 16, 0x7c00, 0x03e0, 0x001f, 0x8000); break;
        case DXGI_FORMAT_R8G8_SNORM: pf = masks(PF_BUMPDUDV, 16, 0x00ff, 0xff00, 0, 0); break;
        case DXGI_FORMAT_R8G8B8A8_SNORM: pf = masks(PF_BUMPDUDV, 32, 0x000000ff, 0x0000ff00, 0x00ff0000, 0xff000000); break;
        case DXGI_FORMAT_R16G16_SNORM: pf = masks(PF_BUMPDUDV, 32, 0x0000ffff, 0xffff0000, 0, 0); break;
        case DXGI_FORMAT_B8G8R8A8_UNORM: pf = kA8R8G8B8; break;
        case DXGI_FORMAT_B8G8R8X8_UNORM:
            if (flags & DDS_FLAGS_FORCE_24BPP_RGB) { pf = kR8G8B8; pitchFlags |= CP_FLAGS_24BPP; }       // no DXGI equivalent
            else pf = kX8R8G8B8;
            break;
        case DXGI_FORMAT_B4G4R4A4_UNORM: pf = masks(PF_RGBA, 16, 0x0f00, 0x00f0, 0x000f, 0xf000); break;
        case DXGI_FORMAT_YUY2: pf = four(cc('Y', 'U', 'Y', '2')); break;
        // D3DX's convention: the D3DFMT enum value as FourCC
        case DXGI_FORMAT_R32G32B32A32_FLOAT: pf = four(116); break;
        case DXGI_FORMAT_R16G16B16A16_FLOAT: pf = four(113); break;
        case DXGI_FORMAT_R16G16B16A16_UNORM: pf = four(36); break;
        case DXGI_FORMAT_R16G16B16A16_SNORM: pf = four(110); break;
        case DXGI_FORMAT_R32G32_FLOAT: pf = four(115); break;
        case DXGI_FORMAT_R16G16_FLOAT: pf = four(112); break;
        case DXGI_FORMAT_R32_FLOAT: pf = four(114); break;
        case DXGI_FORMAT_R16_FLOAT: pf = four(111); break;
        default:
            // only when a Direct3D 9 file is insisted on: sRGB loses its tag, 10:10:10:2 gets D3DX's reversed masks
            if (dx9)
                switch (metadata.format)
                {
                case DXGI_FORMAT_R10G10B10A2_UNORM: pf = kA2B10G10R10; break;
                case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: pf = kA8B8G8R8; break;
                case DXGI_FORMAT_BC1_UNORM_SRGB: pf = four(cc('D', 'X', 'T', '1')); break;
                case DXGI_FORMAT_BC2_UNORM_SRGB: pf = four(pm ? cc('D', 'X', 'T', '2') : cc('D', 'X', 'T', '3')); break;
                case DXGI_FORMAT_BC3_UNORM_SRGB: pf = four(pm ? cc('D', 'X', 'T', '4') : cc('D', 'X', 'T', '5')); break;
                case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: pf = kA8R8G8B8; break;
                case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: pf = kX8R8G8B8; break;
                default: break;
                }
            break;
        }
    }
    required = kMinHeader;
    if (pf.size == 0)
    {
        if (dx9) return HRESULT_E_CANNOT_MAKE;
        required += sizeof(HeaderDX10);
    }
    if (!pDestination) return S_OK;
    if (maxsize < required) return HRESULT(0x8007007A);           // E_NOT_SUFFICIENT_BUFFER

    std::memcpy(pDestination, &kMagic, 4);
    Header h; std::memset(&h, 0, sizeof(h));
    h.size = sizeof(Header); h.flags = HF_TEXTURE; h.caps = CAPS_TEXTURE;
    if (metadata.mipLevels > 0)
    {
        h.flags |= HF_MIPMAP;
        if (metadata.mipLevels > UINT16_MAX) return E_INVALIDARG;
        h.mipMapCount = uint32_t(metadata.mipLevels);
        if (h.mipMapCount > 1) h.caps |= CAPS_MIPMAP;
    }
    switch (metadata.dimension)
    {
    case TEX_DIMENSION_TEXTURE1D:
        if (metadata.width > UINT32_MAX) return E_INVALIDARG;
        h.width = uint32_t(metadata.width); h.height = h.depth = 1;
        break;
    case TEX_DIMENSION_TEXTURE2D:
        if (metadata.height > UINT32_MAX || metadata.width > UINT32_MAX) return E_INVALIDARG;
        h.height = uint32_t(metadata.height); h.width = uint32_t(metadata.width); h.depth = 1;
        if (metadata.IsCubemap()) { h.caps |= CAPS_CUBEMAP; h.caps2 |= CAPS2_CUBEMAP_ALL; }
        break;
    case TEX_DIMENSION_TEXTURE3D:
        if (metadata.height > UINT32_MAX || metadata.width > UINT32_MAX || metadata.depth > UINT16_MAX) return E_INVALIDARG;
        h.flags |= HF_VOLUME; h.caps2 |= CAPS2_VOLUME;
        h.height = uint32_t(metadata.height); h.width = uint32_t(metadata.width); h.depth = uint32_t(metadata.depth);
        break;
    default:
        return E_FAIL;
    }
    size_t rowPitch, slicePitch;
    const HRESULT hr = ComputePitch(metadata.format, metadata.width, metadata.height, rowPitch, slicePitch, CP_FLAGS(pitchFlags));
    if (FAILED(hr)) return hr;
    if (slicePitch > UINT32_MAX || rowPitch > UINT32_MAX) return E_FAIL;
    if (IsCompressed(metadata.format)) { h.flags |= HF_LINEARSIZE; h.pitchOrLinearSize = uint32_t(slicePitch); }
    else { h.flags |= HF_PITCH; h.pitchOrLinearSize = uint32_t(rowPitch); }
    if (pf.size == 0)
    {
        h.pf = four(cc('D', 'X', '1', '0'));
        HeaderDX10 x; std::memset(&x, 0, sizeof(x));
        x.dxgiFormat = uint32_t(metadata.format); x.resourceDimension = uint32_t(metadata.dimension);
        if (metadata.arraySize > UINT16_MAX) return E_INVALIDARG;
        x.miscFlag = metadata.miscFlags & ~uint32_t(TEX_MISC_TEXTURECUBE);
        if (metadata.miscFlags & TEX_MISC_TEXTURECUBE)
        {
            x.miscFlag |= TEX_MISC_TEXTURECUBE;
            if (metadata.arraySize % 6) return E_INVALIDARG;
            x.arraySize = uint32_t(metadata.arraySize / 6);           // the number of cubes
        }
        else x.arraySize = uint32_t(metadata.arraySize);
        if (flags & DDS_FLAGS_FORCE_DX10_EXT_MISC2) x.miscFlags2 = metadata.miscFlags2;       // D3DX10 / 11 reject anything but 0 here
        std::memcpy(pDestination + kMinHeader, &x, sizeof(x));
    }
    else h.pf = pf;
    std::memcpy(pDestination + 4, &h, sizeof(h));
    return S_OK;
}

// SaveToDDSMemory (DirectXTexDDS.cpp:2403-2698)
HRESULT SaveToDDSMemory(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, Blob& blob) noexcept
{
    if (!images || !nimages) return E_INVALIDARG;
    size_t required = 0;
    HRESULT hr = EncodeDDSHeader(metadata, flags, nullptr, 0, required);
    if (FAILED(hr)) return hr;
    const bool use24bpp = metadata.format == DXGI_FORMAT_B8G8R8X8_UNORM && (flags & DDS_FLAGS_FORCE_24BPP_RGB) && !(flags & (DDS_FLAGS_FORCE_DX10_EXT | DDS_FLAGS_FORCE_DX10_EXT_MISC2));
    const CP_FLAGS cp = use24bpp ? CP_FLAGS_24BPP : CP_FLAGS_NONE;
    std::vector<size_t> rp, sp;
    try { rp.resize(nimages); sp.resize(nimages); } catch (...) { return E_OUTOFMEMORY; }
    for (size_t i = 0; i < nimages; ++i)
    {
        if (!images[i].pixels) return E_POINTER;
        if (images[i].format != metadata.format) return E_FAIL;
        hr = ComputePitch(metadata.format, images[i].width, images[i].height, rp[i], sp[i], cp);
        if (FAILED(hr)) return hr;
        required += sp[i];
    }
    blob.Release();
    hr = blob.Initialize(required);
    if (FAILED(hr)) return hr;
    uint8_t* d = blob.GetBufferPointer();
    hr = EncodeDDSHeader(metadata, flags, d, blob.GetBufferSize(), required);
    if (FAILED(hr)) { blob.Release(); return hr; }
    size_t remaining = blob.GetBufferSize() - required;
    d += required;
    if (!remaining) { blob.Release(); return E_FAIL; }

    // how many images the metadata describes, in file order (arrays: item-major; volumes: level by level)
    size_t expected = 0;
    switch (metadata.dimension)
    {
    case TEX_DIMENSION_TEXTURE1D: case TEX_DIMENSION_TEXTURE2D:
        expected = metadata.arraySize * metadata.mipLevels;
        break;
    case TEX_DIMENSION_TEXTURE3D:
        if (metadata.arraySize != 1) { blob.Release(); return E_FAIL; }
        for (size_t level = 0, depth = metadata.depth; level < metadata.mipLevels; ++level) { expected += depth; if (depth > 1) depth >>= 1; }
        break;
    default:
        blob.Release();
        return E_FAIL;
    }
    if (expected > nimages) { blob.Release(); return E_FAIL; }
    for (size_t i = 0; i < expected; ++i)
    {
        const Image& im = images[i];
        if (sp[i] > remaining) { blob.Release(); return E_FAIL; }
        if (use24bpp)
        {
            // B8G8R8X8 rows lose their fourth byte
            for (size_t y = 0; y < im.height; ++y)
            {
                const uint8_t* s = im.pixels + y * im.rowPitch;
                uint8_t* o = d + y * rp[i];
                for (size_t x = 0; x < im.width; ++x, s += 4, o += 3) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
            }
        }
        else if (im.rowPitch == rp[i] && im.slicePitch == sp[i]) std::memcpy(d, im.pixels, sp[i]);
        else
        {
            const size_t lines = ComputeScanlines(metadata.format, im.height), n = std::min(im.rowPitch, rp[i]);
            for (size_t y = 0; y < lines; ++y) std::memcpy(d + y * rp[i], im.pixels + y * im.rowPitch, n);
        }
        d += sp[i];
        remaining -= sp[i];
    }
    return S_OK;
}

HRESULT SaveToDDSMemory(const Image& image, DDS_FLAGS flags, Blob& blob) noexcept
{
    // DirectXTex.inl:151-164
    TexMetadata m;
    m.width = image.width; m.height = image.height; m.depth = 1; m.arraySize = 1; m.mipLevels = 1;
    m.format = image.format; m.dimension = TEX_DIMENSION_TEXTURE2D;
    return SaveToDDSMemory(&image, 1, m, flags, blob);
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
    const bool closed = std::fclose(f) == 0;
    if (n != blob.GetBufferSize() || !closed) { std::remove(szFile); return E_FAIL; }         // no partial files left behind
    return S_OK;
}

HRESULT SaveToDDSFile(const Image& image, DDS_FLAGS flags, const char* szFile) noexcept
{
    TexMetadata m;
    m.width = image.width; m.height = image.height; m.depth = 1; m.arraySize = 1; m.mipLevels = 1;
    m.format = image.format; m.dimension = TEX_DIMENSION_TEXTURE2D;
    return SaveToDDSFile(&image, 1, m, flags, szFile);
}
} // namespace DirectXTexAMD
