// dxtexconv - a small texconv-style batch converter on top of the MI355X host layer (DirectXTexAMD.h).
// Same pipeline order as the reference tool (Texconv/texconv.cpp:2609 resize -> :3109 convert -> :3434 mipmaps ->
// :3711 compress), DDS in, DDS out; every image-processing step runs on the GPU.
//
//   dxtexconv [-w <width>] [-h <height>] [-m <miplevels, 0 = full chain>] [-f <DXGI format name or number>]
//             [-if <POINT|LINEAR|CUBIC|BOX|TRIANGLE>[_WRAP|_MIRROR]] [-bc <q|x|d|u>...] [-srgb] [-gpu <n>] [-dx10]
//             -o <out.dds> <in.dds>
#include "../host/DirectXTexAMD.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace DirectXTexAMD;

namespace
{
struct Name { const char* name; uint32_t value; };
const Name kFormats[] = {
    { "R32G32B32A32_FLOAT", 2 }, { "R16G16B16A16_FLOAT", 10 }, { "R16G16B16A16_UNORM", 11 }, { "R32G32_FLOAT", 16 }, { "R8G8B8A8_UNORM", 28 },
    { "R8G8B8A8_UNORM_SRGB", 29 }, { "R8G8B8A8_SNORM", 31 }, { "R16G16_FLOAT", 34 }, { "R16G16_UNORM", 35 }, { "R32_FLOAT", 41 }, { "R8G8_UNORM", 49 },
    { "R8G8_SNORM", 51 }, { "R16_FLOAT", 54 }, { "R16_UNORM", 56 }, { "R8_UNORM", 61 }, { "R8_SNORM", 63 }, { "A8_UNORM", 65 },
    { "BC1_UNORM", 71 }, { "BC1_UNORM_SRGB", 72 }, { "BC2_UNORM", 74 }, { "BC2_UNORM_SRGB", 75 }, { "BC3_UNORM", 77 }, { "BC3_UNORM_SRGB", 78 },
    { "BC4_UNORM", 80 }, { "BC4_SNORM", 81 }, { "BC5_UNORM", 83 }, { "BC5_SNORM", 84 }, { "B8G8R8A8_UNORM", 87 }, { "B8G8R8X8_UNORM", 88 },
    { "B8G8R8A8_UNORM_SRGB", 91 }, { "B8G8R8X8_UNORM_SRGB", 93 }, { "BC6H_UF16", 95 }, { "BC6H_SF16", 96 }, { "BC7_UNORM", 98 }, { "BC7_UNORM_SRGB", 99 },
};
const Name kFilters[] = {
    { "POINT", TEX_FILTER_POINT }, { "LINEAR", TEX_FILTER_LINEAR }, { "CUBIC", TEX_FILTER_CUBIC }, { "BOX", TEX_FILTER_BOX }, { "FANT", TEX_FILTER_BOX },
    { "TRIANGLE", TEX_FILTER_TRIANGLE },
    { "POINT_WRAP", TEX_FILTER_POINT | TEX_FILTER_WRAP }, { "LINEAR_WRAP", TEX_FILTER_LINEAR | TEX_FILTER_WRAP }, { "CUBIC_WRAP", TEX_FILTER_CUBIC | TEX_FILTER_WRAP },
    { "TRIANGLE_WRAP", TEX_FILTER_TRIANGLE | TEX_FILTER_WRAP }, { "LINEAR_MIRROR", TEX_FILTER_LINEAR | TEX_FILTER_MIRROR }, { "CUBIC_MIRROR", TEX_FILTER_CUBIC | TEX_FILTER_MIRROR },
};

bool lookup(const Name* t, size_t n, const char* s, uint32_t& out)
{
    for (size_t i = 0; i < n; ++i) if (!strcasecmp(t[i].name, s)) { out = t[i].value; return true; }
    char* end = nullptr; const unsigned long v = std::strtoul(s, &end, 0);
    if (end && *end == 0 && end != s) { out = uint32_t(v); return true; }
    return false;
}

int fail(const char* what, HRESULT hr, Device& dev)
{
    std::fprintf(stderr, "FAILED [%s] (%08X) %s\n", what, unsigned(hr), dev ? dev.LastError() : "");
    return 1;
}

// the level-0 image of every array item as a stand-alone single-mip texture
HRESULT TopLevels(const ScratchImage& in, ScratchImage& out)
{
    TexMetadata m = in.GetMetadata();
    m.mipLevels = 1;
    HRESULT hr = out.Initialize(m);
    if (FAILED(hr)) return hr;
    for (size_t item = 0; item < m.arraySize; ++item)
    {
        const Image* s = in.GetImage(0, item, 0); const Image* d = out.GetImage(0, item, 0);
        std::memcpy(d->pixels, s->pixels, d->slicePitch);
    }
    return S_OK;
}
}

int main(int argc, char** argv)
{
    size_t width = 0, height = 0, mipLevels = 1;
    uint32_t format = 0, filter = 0, compress = 0;
    int gpu = 0; bool dx10 = false, haveMips = false;
    const char* outFile = nullptr; const char* inFile = nullptr;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "-w") width = std::strtoull(next(), nullptr, 10);
        else if (a == "-h") height = std::strtoull(next(), nullptr, 10);
        else if (a == "-m") { mipLevels = std::strtoull(next(), nullptr, 10); haveMips = true; }
        else if (a == "-f") { if (!lookup(kFormats, sizeof(kFormats) / sizeof(Name), next(), format)) { std::fprintf(stderr, "unknown format\n"); return 1; } }
        else if (a == "-if") { if (!lookup(kFilters, sizeof(kFilters) / sizeof(Name), next(), filter)) { std::fprintf(stderr, "unknown filter\n"); return 1; } }
        else if (a == "-bc")
        {
            for (const char* p = next(); *p; ++p)
                switch (*p)
                {
                case 'q': compress |= TEX_COMPRESS_BC7_QUICK; break;
                case 'x': compress |= TEX_COMPRESS_BC7_USE_3SUBSETS; break;
                case 'd': compress |= TEX_COMPRESS_DITHER; break;
                case 'u': compress |= TEX_COMPRESS_UNIFORM; break;
                default: std::fprintf(stderr, "unknown -bc flag %c\n", *p); return 1;
                }
        }
        else if (a == "-gpu") gpu = std::atoi(next());
        else if (a == "-dx10") dx10 = true;
        else if (a == "-o") outFile = next();
        else if (a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 1; }
        else inFile = argv[i];
    }
    if (!inFile || !outFile) { std::fprintf(stderr, "usage: dxtexconv [-w W] [-h H] [-m N] [-f FORMAT] [-if FILTER] [-bc qxdu] [-gpu N] [-dx10] -o out.dds in.dds\n"); return 1; }

    Device dev;
    HRESULT hr = dev.Create(gpu);
    if (FAILED(hr)) { std::fprintf(stderr, "no gfx950 device %d (this tool has no CPU path)\n", gpu); return 1; }

    ScratchImage image; TexMetadata info;
    hr = LoadFromDDSFile(inFile, DDS_FLAGS_NONE, &info, image);
    if (FAILED(hr)) return fail("load", hr, dev);
    std::printf("reading %s (%zux%zu, %zu mips, %zu items, format %u)\n", inFile, info.width, info.height, info.mipLevels, info.arraySize, unsigned(info.format));
    const DXGI_FORMAT tformat = format ? DXGI_FORMAT(format) : info.format;

    // --- decompress a block-compressed source when anything has to be done to its texels (texconv.cpp:2325-2372)
    if (IsCompressed(info.format) && (width || height || haveMips || tformat != info.format))
    {
        ScratchImage t;
        hr = Decompress(dev, image.GetImages(), image.GetImageCount(), info, DXGI_FORMAT_UNKNOWN, t);
        if (FAILED(hr)) return fail("decompress", hr, dev);
        image = std::move(t); info = image.GetMetadata();
    }

    // --- resize (level 0 of every item; the mip chain is regenerated or dropped like texconv does)
    const size_t tw = width ? width : info.width, th = height ? height : info.height;
    if (tw != info.width || th != info.height)
    {
        TexMetadata m = info; m.width = tw; m.height = th; m.mipLevels = 1;
        ScratchImage t;
        hr = t.Initialize(m);
        if (FAILED(hr)) return fail("resize", hr, dev);
        for (size_t item = 0; item < info.arraySize; ++item)
        {
            ScratchImage one;
            hr = Resize(dev, *image.GetImage(0, item, 0), tw, th, TEX_FILTER_FLAGS(filter), one);
            if (FAILED(hr)) return fail("resize", hr, dev);
            std::memcpy(t.GetImage(0, item, 0)->pixels, one.GetPixels(), one.GetPixelsSize());
        }
        image = std::move(t); info = image.GetMetadata();
    }

    // --- convert to the uncompressed target format
    if (!IsCompressed(tformat) && tformat != info.format)
    {
        TexMetadata m = info; m.format = tformat;
        ScratchImage t;
        hr = t.Initialize(m);
        if (FAILED(hr)) return fail("convert", hr, dev);
        for (size_t i = 0; i < image.GetImageCount(); ++i)
        {
            ScratchImage one;
            hr = Convert(dev, image.GetImages()[i], tformat, TEX_FILTER_FLAGS(filter), TEX_THRESHOLD_DEFAULT, one);
            if (FAILED(hr)) return fail("convert", hr, dev);
            std::memcpy(t.GetImages()[i].pixels, one.GetPixels(), one.GetPixelsSize());
        }
        image = std::move(t); info = image.GetMetadata();
    }

    // --- mipmaps
    if (haveMips && mipLevels != 1)
    {
        ScratchImage tops;
        const ScratchImage* base = &image;
        if (info.mipLevels != 1) { hr = TopLevels(image, tops); if (FAILED(hr)) return fail("mipmaps", hr, dev); base = &tops; }
        ScratchImage t;
        hr = GenerateMipMaps(dev, base->GetImages(), base->GetImageCount(), base->GetMetadata(), TEX_FILTER_FLAGS(filter), mipLevels, t);
        if (FAILED(hr)) return fail("mipmaps", hr, dev);
        image = std::move(t); info = image.GetMetadata();
    }
    else if (haveMips && mipLevels == 1 && info.mipLevels != 1)
    {
        ScratchImage t;
        hr = TopLevels(image, t);
        if (FAILED(hr)) return fail("mipmaps", hr, dev);
        image = std::move(t); info = image.GetMetadata();
    }

    // --- compress
    if (IsCompressed(tformat) && tformat != info.format)
    {
        ScratchImage t;
        hr = Compress(dev, image.GetImages(), image.GetImageCount(), info, tformat, TEX_COMPRESS_FLAGS(compress), TEX_THRESHOLD_DEFAULT, t);
        if (FAILED(hr)) return fail("compress", hr, dev);
        image = std::move(t); info = image.GetMetadata();
    }

    hr = SaveToDDSFile(image.GetImages(), image.GetImageCount(), info, dx10 ? DDS_FLAGS_FORCE_DX10_EXT : DDS_FLAGS_NONE, outFile);
    if (FAILED(hr)) return fail("save", hr, dev);
    std::printf("writing %s (%zux%zu, %zu mips, %zu items, format %u)\n", outFile, info.width, info.height, info.mipLevels, info.arraySize, unsigned(info.format));
    return 0;
}
