// dxtexconv - a texconv-style batch converter on top of the MI355X host layer (DirectXTexAMD.h): DDS in, DDS out, every
// image-processing step on the GPU. The pipeline and its order are the reference tool's (Texconv/texconv.cpp): load (:2077-2090)
// -> decompress (:2325-2480) -> undo premultiplied alpha (:2482-2530) -> resize (:2577-2640) -> convert (:3100-3140) ->
// mipmaps (:3302-3460) -> alpha-coverage preservation (:3462-3500) -> premultiply alpha (:3502-3545) -> compress (:3547-3735) ->
// alpha mode (:3738-3766) -> save (:3858-3878). Option names are texconv's; what has no GPU implementation here (flips,
// swizzles, normal maps, tone mapping, WIC codecs, dithered conversion) is refused, not approximated.
//
//   dxtexconv [options] -o <out.dds | output directory> <in.dds | in.hdr | in.tga>...        (-ft hdr | tga: Radiance / TGA output of level 0)
//     -w <n> -h <n>        target size                         -pow2             fit to a power of two (keeps the aspect ratio)
//     -m <n>               mip levels, 0 = full chain           -fl <9.1 .. 12.2> feature level: largest texture side allowed
//     -f <format>          DXGI format name or number           -if <filter>      POINT LINEAR CUBIC FANT BOX TRIANGLE
//     -wrap -mirror        addressing of the filters            -srgb -srgbi -srgbo   sRGB on both sides / input / output
//     -pmalpha -alpha      to / from premultiplied alpha        -keepcoverage <ref>   keep alpha-test coverage in the mips
//     -at <threshold>      alpha threshold (BC1, 1-bit alpha)   -bc <q|x|d|u>...  BC7 quick / 3 subsets, dither, uniform weights
//     -x2bias              *2 - 1 on conversions to / from SNORM -sepalpha        resize / mip alpha separately (alpha mode custom)
//     -dword -badtails -permissive -ignoremips -xlum            DDS reader tolerances (DDS_FLAGS)
//     -dx10 -dx9           force the 'DX10' header (+ alpha mode) / a Direct3D 9 file        -tga20   TGA output with the 2.0 extension area
//     -px <s> -sx <s> -l   output name prefix / suffix, lower case    -y   overwrite    -info (print what the files hold, no GPU)    -timing -nologo -gpu <n> | -gpus <a,b,...> (files dealt out over the GPUs)
//     -overlap <n>         workers (contexts) per GPU, default 2: the transfers and file codecs of one file overlap the kernels of another
#include "../host/DirectXTexAMD.h"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/stat.h>
#include <atomic>
#include <thread>
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
    { "R32G32B32_FLOAT", 6 }, { "R16G16B16A16_SNORM", 13 }, { "R10G10B10A2_UNORM", 24 }, { "R11G11B10_FLOAT", 26 }, { "R16G16_SNORM", 37 }, { "R16_SNORM", 58 },
    { "R9G9B9E5_SHAREDEXP", 67 }, { "B5G6R5_UNORM", 85 }, { "B5G5R5A1_UNORM", 86 }, { "B4G4R4A4_UNORM", 115 },
    { "R32G32B32A32_UINT", 3 }, { "R32G32B32A32_SINT", 4 }, { "R32G32B32_UINT", 7 }, { "R32G32B32_SINT", 8 }, { "R16G16B16A16_UINT", 12 }, { "R16G16B16A16_SINT", 14 },
    { "R32G32_UINT", 17 }, { "R32G32_SINT", 18 }, { "R10G10B10A2_UINT", 25 }, { "R8G8B8A8_UINT", 30 }, { "R8G8B8A8_SINT", 32 }, { "R16G16_UINT", 36 }, { "R16G16_SINT", 38 },
    { "R32_UINT", 42 }, { "R32_SINT", 43 }, { "R8G8_UINT", 50 }, { "R8G8_SINT", 52 }, { "R16_UINT", 57 }, { "R16_SINT", 59 }, { "R8_UINT", 62 }, { "R8_SINT", 64 },
    { "R10G10B10_XR_BIAS_A2_UNORM", 89 }, { "AYUV", 100 }, { "Y410", 101 }, { "Y416", 102 },
    { "D32_FLOAT_S8X24_UINT", 20 }, { "D32_FLOAT", 40 }, { "D24_UNORM_S8_UINT", 45 }, { "D16_UNORM", 55 },
    { "R1_UNORM", 66 }, { "R8G8_B8G8_UNORM", 68 }, { "G8R8_G8B8_UNORM", 69 }, { "YUY2", 107 }, { "Y210", 108 }, { "Y216", 109 }, { "A4B4G4R4_UNORM", 191 },
    // texconv's aliases (texconv.cpp:420-440)
    { "DXT1", 71 }, { "DXT2", 74 }, { "DXT3", 74 }, { "DXT4", 77 }, { "DXT5", 77 }, { "RGBA", 28 }, { "BGRA", 87 }, { "BGR", 88 }, { "FP16", 10 }, { "FP32", 2 },
    { "BC4", 80 }, { "BC5", 83 }, { "BC6H", 95 }, { "BC7", 98 },
};
const Name kFilters[] = {
    { "POINT", TEX_FILTER_POINT }, { "LINEAR", TEX_FILTER_LINEAR }, { "CUBIC", TEX_FILTER_CUBIC }, { "BOX", TEX_FILTER_BOX }, { "FANT", TEX_FILTER_BOX },
    { "TRIANGLE", TEX_FILTER_TRIANGLE },
    { "POINT_WRAP", TEX_FILTER_POINT | TEX_FILTER_WRAP }, { "LINEAR_WRAP", TEX_FILTER_LINEAR | TEX_FILTER_WRAP }, { "CUBIC_WRAP", TEX_FILTER_CUBIC | TEX_FILTER_WRAP },
    { "TRIANGLE_WRAP", TEX_FILTER_TRIANGLE | TEX_FILTER_WRAP }, { "LINEAR_MIRROR", TEX_FILTER_LINEAR | TEX_FILTER_MIRROR }, { "CUBIC_MIRROR", TEX_FILTER_CUBIC | TEX_FILTER_MIRROR },
};
const Name kFeatureLevels[] = {         // largest 2D texture side (texconv.cpp:880-905)
    { "9.1", 2048 }, { "9.2", 2048 }, { "9.3", 4096 }, { "10.0", 8192 }, { "10.1", 8192 }, { "11.0", 16384 }, { "11.1", 16384 }, { "12.0", 16384 }, { "12.1", 16384 }, { "12.2", 16384 },
};
constexpr uint32_t TEX_FILTER_SEPARATE_ALPHA = 0x100;

bool lookup(const Name* t, size_t n, const char* s, uint32_t& out, bool numbers = true)
{
    for (size_t i = 0; i < n; ++i) if (!strcasecmp(t[i].name, s)) { out = t[i].value; return true; }
    if (!numbers) return false;
    char* end = nullptr; const unsigned long v = std::strtoul(s, &end, 0);
    if (end && *end == 0 && end != s) { out = uint32_t(v); return true; }
    return false;
}

struct Options
{
    size_t width = 0, height = 0, mipLevels = 0, maxSize = 16384;          // mipLevels 0: keep a chain the input has, else build the full one
    bool pow2 = false, pmalpha = false, demul = false, dx10 = false, dx9 = false, sepalpha = false, lower = false, overwrite = false,
         timing = false, nologo = false, hdrOut = false, tgaOut = false, tga20 = false, info = false;
    uint32_t format = 0, filter = 0, filterOpts = 0, srgb = 0, convert = 0, compress = 0, ddsRead = DDS_FLAGS_ALLOW_LARGE_FILES;
    float alphaThreshold = TEX_THRESHOLD_DEFAULT, keepCoverage = 0.f;
    std::vector<int> gpus;              // one worker (own Device, own host thread) per entry; input i goes to worker i mod n
    size_t overlap = 2;                 // workers per listed GPU: file k + 1 is read, decoded and uploaded while file k's kernels run
    std::string prefix, suffix, out;
    std::vector<std::string> inputs;
};

bool ispow2(size_t x) { return x && !(x & (x - 1)); }

// FitPowerOf2 (texconv.cpp:1019-1057): the larger side snaps down to a power of two, the other takes the power of two that keeps
// the aspect ratio best
void FitPowerOf2(size_t origx, size_t origy, size_t& targetx, size_t& targety, size_t maxsize)
{
    const float aspect = float(origx) / float(origy);
    size_t& major = (origx > origy) ? targetx : targety;
    size_t& minor = (origx > origy) ? targety : targetx;
    size_t m;
    for (m = maxsize; m > 1; m >>= 1) if (m <= major) break;
    major = m;
    float best = 3.402823466e+38f;
    for (size_t o = maxsize; o > 0; o >>= 1)
    {
        const float score = std::fabs(((origx > origy) ? float(m) / float(o) : float(o) / float(m)) - aspect);
        if (score < best) { best = score; minor = o; }
    }
}

// mip level 0 of every array item / depth slice as a texture with one level (texconv.cpp:3324-3380), host images
HRESULT TopLevels(const ScratchImage& in, ScratchImage& out)
{
    TexMetadata m = in.GetMetadata();
    m.mipLevels = 1;
    HRESULT hr = out.Initialize(m);
    if (FAILED(hr)) return hr;
    const bool volume = m.dimension == TEX_DIMENSION_TEXTURE3D;
    for (size_t i = 0; i < (volume ? m.depth : m.arraySize); ++i)
    {
        const Image* s = volume ? in.GetImage(0, 0, i) : in.GetImage(0, i, 0);
        const Image* d = volume ? out.GetImage(0, 0, i) : out.GetImage(0, i, 0);
        if (!s || !d) return E_FAIL;
        std::memcpy(d->pixels, s->pixels, d->slicePitch);
    }
    return S_OK;
}

int usage()
{
    std::fprintf(stderr, "usage: dxtexconv [-w W] [-h H] [-pow2] [-fl LEVEL] [-m N] [-f FORMAT] [-if FILTER] [-wrap] [-mirror] [-srgb|-srgbi|-srgbo]\n"
                         "                 [-pmalpha|-alpha] [-keepcoverage REF] [-at T] [-bc qxdu] [-x2bias] [-sepalpha] [-dword] [-badtails] [-permissive]\n"
                         "                 [-ignoremips] [-xlum] [-dx10|-dx9] [-px S] [-sx S] [-l] [-y] [-timing] [-nologo] [-gpu N | -gpus A,B,...] [-overlap N] -o <out.dds | dir> in.dds...\n");
    return 1;
}

bool Parse(int argc, char** argv, Options& o)
{
    for (int i = 1; i < argc; ++i)
    {
        std::string a = argv[i];
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') a.erase(0, 1);          // --long-form spelled like the short one
        bool missing = false;
        auto next = [&]() -> const char* { if (i + 1 < argc) return argv[++i]; missing = true; return ""; };
        if (a == "-w") o.width = std::strtoull(next(), nullptr, 10);
        else if (a == "-h") o.height = std::strtoull(next(), nullptr, 10);
        else if (a == "-m") o.mipLevels = std::strtoull(next(), nullptr, 10);
        else if (a == "-f") { if (!lookup(kFormats, sizeof(kFormats) / sizeof(Name), next(), o.format)) { std::fprintf(stderr, "unknown format\n"); return false; } }
        else if (a == "-if") { if (!lookup(kFilters, sizeof(kFilters) / sizeof(Name), next(), o.filter)) { std::fprintf(stderr, "unknown filter (dithered filters have no GPU path)\n"); return false; } }
        else if (a == "-fl") { uint32_t v; if (!lookup(kFeatureLevels, sizeof(kFeatureLevels) / sizeof(Name), next(), v, false)) { std::fprintf(stderr, "unknown feature level\n"); return false; } o.maxSize = v; }
        else if (a == "-bc")
        {
            for (const char* p = next(); *p; ++p)
                switch (*p)
                {
                case 'q': o.compress |= TEX_COMPRESS_BC7_QUICK; break;
                case 'x': o.compress |= TEX_COMPRESS_BC7_USE_3SUBSETS; break;
                case 'd': o.compress |= TEX_COMPRESS_DITHER; break;
                case 'u': o.compress |= TEX_COMPRESS_UNIFORM; break;
                default: std::fprintf(stderr, "unknown -bc flag %c\n", *p); return false;
                }
        }
        else if (a == "-srgb") o.srgb |= TEX_FILTER_SRGB;
        else if (a == "-srgbi") o.srgb |= TEX_FILTER_SRGB_IN;
        else if (a == "-srgbo") o.srgb |= TEX_FILTER_SRGB_OUT;
        else if (a == "-wrap") { if (o.filterOpts & TEX_FILTER_MIRROR) { std::fprintf(stderr, "-wrap and -mirror exclude each other\n"); return false; } o.filterOpts |= TEX_FILTER_WRAP; }
        else if (a == "-mirror") { if (o.filterOpts & TEX_FILTER_WRAP) { std::fprintf(stderr, "-wrap and -mirror exclude each other\n"); return false; } o.filterOpts |= TEX_FILTER_MIRROR; }
        else if (a == "-sepalpha") { o.sepalpha = true; o.filterOpts |= TEX_FILTER_SEPARATE_ALPHA; }
        else if (a == "-x2bias") o.convert |= TEX_FILTER_FLOAT_X2BIAS;
        else if (a == "-pow2") o.pow2 = true;
        else if (a == "-pmalpha") o.pmalpha = true;
        else if (a == "-alpha") o.demul = true;
        else if (a == "-keepcoverage") { o.keepCoverage = float(std::atof(next())); if (!(o.keepCoverage >= 0.f && o.keepCoverage <= 1.f)) { std::fprintf(stderr, "-keepcoverage wants a value in [0, 1]\n"); return false; } }
        else if (a == "-at") { o.alphaThreshold = float(std::atof(next())); if (!(o.alphaThreshold >= 0.f)) { std::fprintf(stderr, "-at wants a non-negative value\n"); return false; } }
        else if (a == "-aw") next();                                              // the DirectCompute encoder's alpha weight: no meaning here
        else if (a == "-dword") o.ddsRead |= DDS_FLAGS_LEGACY_DWORD;
        else if (a == "-badtails") o.ddsRead |= DDS_FLAGS_BAD_DXTN_TAILS;
        else if (a == "-permissive") o.ddsRead |= DDS_FLAGS_PERMISSIVE;
        else if (a == "-ignoremips") o.ddsRead |= DDS_FLAGS_IGNORE_MIPS;
        else if (a == "-xlum") o.ddsRead |= DDS_FLAGS_EXPAND_LUMINANCE;
        else if (a == "-tga20") o.tga20 = true;
        else if (a == "-info") o.info = true;
        else if (a == "-dx10") o.dx10 = true;
        else if (a == "-dx9") o.dx9 = true;
        else if (a == "-px") o.prefix = next();
        else if (a == "-sx") o.suffix = next();
        else if (a == "-l") o.lower = true;
        else if (a == "-y") o.overwrite = true;
        else if (a == "-timing") o.timing = true;
        else if (a == "-nologo") o.nologo = true;
        else if (a == "-gpu") o.gpus.assign(1, std::atoi(next()));
        else if (a == "-gpus")
        {
            // "0,1,2,3": the files are dealt out over these GPUs, one host thread and one context each (contexts share nothing)
            o.gpus.clear();
            for (const char* p = next(); *p;)
            {
                char* end = nullptr;
                const long g = std::strtol(p, &end, 10);
                if (end == p || g < 0) { std::fprintf(stderr, "-gpus wants a comma-separated list of device ordinals\n"); return false; }
                o.gpus.push_back(int(g));
                p = (*end == ',') ? end + 1 : end;
                if (*end && *end != ',') { std::fprintf(stderr, "-gpus wants a comma-separated list of device ordinals\n"); return false; }
            }
            if (o.gpus.empty()) { std::fprintf(stderr, "-gpus wants a comma-separated list of device ordinals\n"); return false; }
        }
        else if (a == "-overlap") { o.overlap = std::strtoull(next(), nullptr, 10); if (o.overlap < 1 || o.overlap > 8) { std::fprintf(stderr, "-overlap wants 1 .. 8 workers per GPU\n"); return false; } }
        else if (a == "-o") o.out = next();
        else if (a == "-ft")
        {
            const char* ft = next();
            if (!strcasecmp(ft, "hdr")) o.hdrOut = true;
            else if (!strcasecmp(ft, "tga")) o.tgaOut = true;
            else if (strcasecmp(ft, "dds")) { std::fprintf(stderr, "output file types: dds, hdr, tga\n"); return false; }
        }
        else if (a == "-r" || a == "-nogpu" || a == "-singleproc") { }            // nothing to switch here
        else if (a[0] == '-') { std::fprintf(stderr, "unknown or unsupported option %s\n", a.c_str()); return false; }
        else o.inputs.push_back(argv[i]);
        if (missing) { std::fprintf(stderr, "option %s needs a value\n", a.c_str()); return false; }
    }
    if (o.pmalpha && o.demul) { std::fprintf(stderr, "-pmalpha and -alpha exclude each other\n"); return false; }
    if (o.dx10 && o.dx9) { std::fprintf(stderr, "-dx10 and -dx9 exclude each other\n"); return false; }
    if (o.gpus.empty()) o.gpus.assign(1, 0);
    if (o.info) return !o.inputs.empty();
    return !o.inputs.empty() && !o.out.empty();
}

// <out> names a file when it ends in .dds and there is one input; otherwise a directory that receives <px><name><sx>.dds
std::string OutputName(const Options& o, const std::string& input)
{
    auto endsDDS = [](const std::string& s) { return s.size() > 4 && (!strcasecmp(s.c_str() + s.size() - 4, ".dds") || !strcasecmp(s.c_str() + s.size() - 4, ".hdr") || !strcasecmp(s.c_str() + s.size() - 4, ".tga")); };
    if (o.inputs.size() == 1 && endsDDS(o.out) && o.prefix.empty() && o.suffix.empty()) return o.out;
    std::string base = input.substr(input.find_last_of('/') == std::string::npos ? 0 : input.find_last_of('/') + 1);
    if (base.find_last_of('.') != std::string::npos) base.erase(base.find_last_of('.'));
    std::string name = o.prefix + base + o.suffix + (o.hdrOut ? ".hdr" : o.tgaOut ? ".tga" : ".dds");
    if (o.lower) std::transform(name.begin(), name.end(), name.begin(), [](unsigned char c) { return char(std::tolower(c)); });
    return o.out + "/" + name;
}

struct StepFailed { const char* what; HRESULT hr; };

// PremultiplyAlpha checks the texture's alpha mode (DirectXTexPMAlpha.cpp:283-284); the tool tracks it in `info`, which must be what the
// resident image carries
HRESULT PremultiplyAlphaWithMode(Device& dev, const DeviceScratchImage& image, const TexMetadata& info, TEX_PMALPHA_FLAGS flags, DeviceScratchImage& result)
{
    if (image.GetMetadata().miscFlags2 != info.miscFlags2) return E_FAIL;
    return PremultiplyAlpha(dev, image, flags, result);
}

// One file through the pipeline; throws StepFailed. The texture is uploaded ONCE after the file has been decoded, every step runs on the
// device-resident copy (DeviceScratchImage in, DeviceScratchImage out: the steps queue their kernels on the Device's stream back to back),
// and the final images come back ONCE for the file writer: two PCIe transfers per file whatever the number of steps, where the
// ScratchImage -> ScratchImage calls of the reference tool's sequence would be one round trip per step.
void ConvertOne(Device& dev, const Options& o, const std::string& inFile, const std::string& outFile)
{
    auto check = [](const char* what, HRESULT hr) { if (FAILED(hr)) throw StepFailed{ what, hr }; };
    const TEX_FILTER_FLAGS filter = TEX_FILTER_FLAGS(o.filter | o.filterOpts);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t up0 = 0, down0 = 0;
    GetTransferBytes(dev, up0, down0);

    ScratchImage loaded; TexMetadata info;
    auto hasExt = [](const std::string& f, const char* ext) { const size_t n = std::strlen(ext); return f.size() > n && !strcasecmp(f.c_str() + f.size() - n, ext); };
    if (hasExt(inFile, ".hdr")) check("load", LoadFromHDRFile(inFile.c_str(), &info, loaded));               // Radiance RGBE -> RGBA32F
    else if (hasExt(inFile, ".tga")) check("load", LoadFromTGAFile(inFile.c_str(), TGA_FLAGS_NONE, &info, loaded));
    else check("load", LoadFromDDSFile(inFile.c_str(), DDS_FLAGS(o.ddsRead), &info, loaded));
    std::printf("reading %s (%zux%zu", inFile.c_str(), info.width, info.height);
    if (info.dimension == TEX_DIMENSION_TEXTURE3D) std::printf("x%zu", info.depth);
    std::printf(", %zu mips, %zu items, format %u)\n", info.mipLevels, info.arraySize, unsigned(info.format));
    const DXGI_FORMAT tformat = o.format ? DXGI_FORMAT(o.format) : info.format;
    size_t tMips = (!o.mipLevels && info.mipLevels > 1) ? info.mipLevels : o.mipLevels;           // texconv.cpp:2270

    // --- the one upload
    DeviceScratchImage image;
    check("upload", image.Upload(dev, loaded));
    auto keep = [&](DeviceScratchImage& t)         // a step produced t: it becomes the image, metadata keeps its alpha mode
    {
        const uint32_t misc2 = info.miscFlags2;
        image = std::move(t); info = image.GetMetadata(); info.miscFlags2 = misc2;
    };

    // --- decompress (texconv.cpp:2325-2480). The compressed original is kept (on the host, where it already is, and on the device for the
    // alpha scan): if no step below changes the texels and the target is its format, it is written back as it is instead of being encoded again.
    ScratchImage cimage;
    DeviceScratchImage dcimage;
    bool haveOriginal = false;
    if (IsCompressed(info.format))
    {
        DeviceScratchImage t;
        check("decompress", Decompress(dev, image, DXGI_FORMAT_UNKNOWN, t));
        cimage = std::move(loaded); dcimage = std::move(image); haveOriginal = true;
        keep(t);
    }
    else loaded.Release();

    // --- undo premultiplied alpha
    if (o.demul && HasAlpha(info.format) && info.format != DXGI_FORMAT_A8_UNORM)
    {
        if (info.GetAlphaMode() == TEX_ALPHA_MODE_STRAIGHT) std::printf("WARNING: image is already using straight alpha\n");
        else if (!info.IsPMAlpha()) std::printf("WARNING: image is not using premultiplied alpha\n");
        else
        {
            DeviceScratchImage t;
            // the alpha mode lives in the tool's metadata: the resident image carries the file's, which the step checks (DirectXTexPMAlpha.cpp:283-284)
            check("demultiply alpha", PremultiplyAlphaWithMode(dev, image, info, TEX_PMALPHA_FLAGS(TEX_PMALPHA_REVERSE | o.srgb), t));
            info.miscFlags2 = t.GetMetadata().miscFlags2;
            image = std::move(t);
            haveOriginal = false;
        }
    }

    // --- resize: level 0 of every item / slice; the result has one level (DirectXTexResize.cpp:942-1103)
    size_t tw = o.width ? o.width : info.width, th = o.height ? o.height : info.height;
    if (tw > o.maxSize) { if (!o.width) tw = o.maxSize; else std::printf("WARNING: width exceeds the feature level's %zu\n", o.maxSize); }
    if (th > o.maxSize) { if (!o.height) th = o.maxSize; else std::printf("WARNING: height exceeds the feature level's %zu\n", o.maxSize); }
    if (o.pow2) FitPowerOf2(info.width, info.height, tw, th, o.maxSize);
    if (tw != info.width || th != info.height)
    {
        DeviceScratchImage t;
        check("resize", Resize(dev, image, tw, th, filter, t));
        keep(t);
        haveOriginal = false;
        if (tMips > 0)
        {
            size_t maxMips = 0;
            if (info.depth > 1) CalculateMipLevels3D(info.width, info.height, info.depth, maxMips); else CalculateMipLevels(info.width, info.height, maxMips);
            tMips = std::min(tMips, maxMips);
        }
    }

    // --- convert to the uncompressed target format
    if (!IsCompressed(tformat) && tformat != info.format)
    {
        DeviceScratchImage t;
        check("convert", Convert(dev, image, tformat, TEX_FILTER_FLAGS(filter | o.srgb | o.convert), o.alphaThreshold, t));
        keep(t);
        haveOriginal = false;
    }

    // --- mipmaps
    const bool keepCoverage = o.keepCoverage > 0.f && HasAlpha(info.format) && !IsAlphaAllOpaque(dev, image);
    TEX_FILTER_FLAGS filter3D = filter;
    if (!ispow2(info.width) || !ispow2(info.height) || !ispow2(info.depth))
    {
        if (!tMips || info.mipLevels != 1) std::printf("WARNING: not a power-of-two texture with mips\n");
        if (info.dimension == TEX_DIMENSION_TEXTURE3D) filter3D = TEX_FILTER_FLAGS(TEX_FILTER_TRIANGLE | o.filterOpts);      // the only correct one for such volumes (:3317-3321)
    }
    if ((!tMips || info.mipLevels != tMips || keepCoverage) && info.mipLevels != 1)
    {
        // generation starts from a single level: strip the existing chain
        DeviceScratchImage t;
        check("copy to single level", CopyTopLevels(dev, image, t));
        keep(t);
        if (haveOriginal && tMips == 1)
        {
            // only trimming mips off a compressed texture: its top level stays the encoder's original (:3378-3412)
            ScratchImage c;
            check("copy compressed to single level", TopLevels(cimage, c));
            cimage = std::move(c);
            DeviceScratchImage dc;
            check("copy compressed to single level", CopyTopLevels(dev, dcimage, dc));
            dcimage = std::move(dc);
        }
        else haveOriginal = false;
    }
    if ((!tMips || info.mipLevels != tMips) && (info.width > 1 || info.height > 1 || info.depth > 1))
    {
        DeviceScratchImage t;
        if (info.dimension == TEX_DIMENSION_TEXTURE3D) check("mipmaps", GenerateMipMaps3D(dev, image, filter3D, tMips, t));
        else check("mipmaps", GenerateMipMaps(dev, image, filter, tMips, t));
        keep(t);
        haveOriginal = false;
    }

    // --- keep the alpha-test coverage of level 0 in the smaller levels
    if (keepCoverage && info.mipLevels != 1)
    {
        DeviceScratchImage t;
        check("keepcoverage", ScaleMipMapsAlphaForCoverage(dev, image, o.keepCoverage, t));
        image = std::move(t);
        haveOriginal = false;
    }

    // --- premultiplied alpha
    if (o.pmalpha && HasAlpha(info.format) && info.format != DXGI_FORMAT_A8_UNORM)
    {
        if (info.IsPMAlpha()) std::printf("WARNING: image is already using premultiplied alpha\n");
        else
        {
            DeviceScratchImage t;
            check("premultiply alpha", PremultiplyAlphaWithMode(dev, image, info, TEX_PMALPHA_FLAGS(TEX_PMALPHA_DEFAULT | o.srgb), t));
            info.miscFlags2 = t.GetMetadata().miscFlags2;
            image = std::move(t);
            haveOriginal = false;
        }
    }

    // --- compress
    bool handThrough = false;
    if (IsCompressed(tformat))
    {
        if (haveOriginal && cimage.GetMetadata().format == tformat)
        {
            // nothing touched the texels and the input already has the target format: hand its blocks through (:3566-3574)
            handThrough = true;
            const uint32_t misc2 = info.miscFlags2;
            info = cimage.GetMetadata(); info.miscFlags2 = misc2;
            image = std::move(dcimage);
        }
        else
        {
            if ((info.width % 4) || (info.height % 4)) std::printf("WARNING: block-compressed texture whose size is not a multiple of 4\n");
            DeviceScratchImage t;
            check("compress", Compress(dev, image, tformat, TEX_COMPRESS_FLAGS(o.compress | o.srgb), o.alphaThreshold, t));
            keep(t);
        }
    }

    // --- alpha mode (texconv.cpp:3738-3766): a reduction on the device, compressed results decoded there
    if (HasAlpha(info.format) && info.format != DXGI_FORMAT_A8_UNORM)
    {
        if (IsAlphaAllOpaque(dev, image)) info.SetAlphaMode(TEX_ALPHA_MODE_OPAQUE);
        else if (info.IsPMAlpha()) { }
        else if (o.sepalpha) info.SetAlphaMode(TEX_ALPHA_MODE_CUSTOM);
        else if (info.GetAlphaMode() == TEX_ALPHA_MODE_UNKNOWN) info.SetAlphaMode(TEX_ALPHA_MODE_STRAIGHT);
    }
    else info.SetAlphaMode(TEX_ALPHA_MODE_UNKNOWN);

    // --- save
    if (!o.overwrite)
    {
        struct stat st;
        if (::stat(outFile.c_str(), &st) == 0) { std::printf("skipping %s: it exists (use -y to overwrite)\n", outFile.c_str()); return; }
    }
    // --- the one download (blocks handed through are still on the host)
    ScratchImage result;
    if (handThrough) result = std::move(cimage);
    else check("download", image.Download(result));
    image.Release();
    uint32_t ddsFlags = DDS_FLAGS_NONE;
    if (o.dx10) ddsFlags |= DDS_FLAGS_FORCE_DX10_EXT | DDS_FLAGS_FORCE_DX10_EXT_MISC2;
    else if (o.dx9) ddsFlags |= DDS_FLAGS_FORCE_DX9_LEGACY;
    if (o.hdrOut) check("save", SaveToHDRFile(result.GetImages()[0], outFile.c_str()));                     // level 0 of the first item, like texconv's non-DDS codecs
    else if (o.tgaOut) check("save", SaveToTGAFile(result.GetImages()[0], TGA_FLAGS_NONE, outFile.c_str(), o.tga20 ? &info : nullptr));      // -tga20: with the TGA 2.0 extension area
    else check("save", SaveToDDSFile(result.GetImages(), result.GetImageCount(), info, DDS_FLAGS(ddsFlags), outFile.c_str()));
    std::printf("writing %s (%zux%zu", outFile.c_str(), info.width, info.height);
    if (info.dimension == TEX_DIMENSION_TEXTURE3D) std::printf("x%zu", info.depth);
    std::printf(", %zu mips, %zu items, format %u)\n", info.mipLevels, info.arraySize, unsigned(info.format));
    if (o.timing)
    {
        uint64_t up = 0, down = 0;
        GetTransferBytes(dev, up, down);
        std::printf("  %s: %.3f ms, host -> device %llu bytes, device -> host %llu bytes\n", inFile.c_str(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), (unsigned long long)(up - up0), (unsigned long long)(down - down0));
    }
}
}

// -info: what a file holds (the header only; no GPU involved) - the "info" command of the reference's texdiag in one line per file
int PrintInfo(const Options& o)
{
    int failures = 0;
    for (const std::string& f : o.inputs)
    {
        auto hasExt = [&](const char* ext) { const size_t n = std::strlen(ext); return f.size() > n && !strcasecmp(f.c_str() + f.size() - n, ext); };
        TexMetadata m;
        const HRESULT hr = hasExt(".hdr") ? GetMetadataFromHDRFile(f.c_str(), m) : hasExt(".tga") ? GetMetadataFromTGAFile(f.c_str(), TGA_FLAGS_NONE, m)
                                                                                                   : GetMetadataFromDDSFile(f.c_str(), DDS_FLAGS(o.ddsRead), m);
        if (FAILED(hr)) { std::printf("%s: FAILED (%08X)\n", f.c_str(), unsigned(hr)); ++failures; continue; }
        const char* name = "";
        for (const Name& n : kFormats) if (n.value == uint32_t(m.format)) { name = n.name; break; }
        static const char* const alpha[] = { "unknown", "straight", "premultiplied", "opaque", "custom" };
        size_t images = 0, bytes = 0;
        for (size_t level = 0, w = m.width, h = m.height, d = m.depth; level < m.mipLevels; ++level)
        {
            size_t rp = 0, sp = 0;
            if (SUCCEEDED(ComputePitch(m.format, w, h, rp, sp))) bytes += sp * d * (m.dimension == TEX_DIMENSION_TEXTURE3D ? 1 : m.arraySize);
            images += d * (m.dimension == TEX_DIMENSION_TEXTURE3D ? 1 : m.arraySize);
            if (w > 1) w >>= 1;
            if (h > 1) h >>= 1;
            if (d > 1) d >>= 1;
        }
        std::printf("%s: %zux%zu", f.c_str(), m.width, m.height);
        if (m.dimension == TEX_DIMENSION_TEXTURE3D) std::printf("x%zu", m.depth);
        std::printf(" %s mips %zu items %zu format %u %s bpp %zu alpha %s%s images %zu bytes %zu%s\n",
                    m.dimension == TEX_DIMENSION_TEXTURE1D ? "1D" : m.dimension == TEX_DIMENSION_TEXTURE3D ? "3D" : (m.IsCubemap() ? "cube" : "2D"),
                    m.mipLevels, m.arraySize, unsigned(m.format), name, BitsPerPixel(m.format), alpha[std::min<uint32_t>(m.GetAlphaMode(), 4)],
                    IsSRGB(m.format) ? " sRGB" : "", images, bytes, IsSupportedOnDevice(m.format) ? "" : " (container only: no GPU path for this format)");
    }
    return failures ? 1 : 0;
}

int main(int argc, char** argv)
{
    Options o;
    if (!Parse(argc, argv, o)) return usage();
    if (o.info) return PrintInfo(o);
    if (!o.nologo) std::printf("dxtexconv: DirectXTex pipeline on MI355X (gfx950)\n");

    // Image-per-GPU sharding: files are independent, so worker k takes inputs k, k + n, k + 2n, ... on its own
    // Device; no data moves between GPUs. With one GPU this is a plain loop on the calling thread.
    std::atomic<int> failures{ 0 };
    auto work = [&](size_t k, size_t n)
    {
        Device dev;
        if (FAILED(dev.Create(o.gpus[k]))) { std::fprintf(stderr, "no gfx950 device %d (this tool has no CPU path)\n", o.gpus[k]); failures += int((o.inputs.size() - k + n - 1) / n); return; }
        for (size_t i = k; i < o.inputs.size(); i += n)
        {
            try { ConvertOne(dev, o, o.inputs[i], OutputName(o, o.inputs[i])); }
            catch (const StepFailed& f)
            {
                std::fprintf(stderr, "FAILED [%s] (%08X) %s: %s\n", f.what, unsigned(f.hr), o.inputs[i].c_str(), dev.LastError());
                ++failures;                               // like texconv: report, go on with the next file, exit code 1
            }
        }
    };
    // Two (-overlap n) workers per GPU, each with its own Device: while one worker's kernels run, the other reads, decodes and uploads its next
    // file (or downloads and writes its last one), so the PCIe transfers and the file codecs of file k + 1 overlap the kernels of file k.
    if (o.overlap > 1 && o.inputs.size() > o.gpus.size())
    {
        std::vector<int> expanded;
        for (size_t r = 0; r < o.overlap; ++r) expanded.insert(expanded.end(), o.gpus.begin(), o.gpus.end());
        o.gpus = expanded;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t workers = std::min(o.gpus.size(), o.inputs.size());
    if (workers <= 1) work(0, 1);
    else
    {
        std::vector<std::thread> pool;
        for (size_t k = 0; k < workers; ++k) pool.emplace_back(work, k, workers);
        for (std::thread& t : pool) t.join();
    }
    if (o.timing) std::printf("processing time: %.3f seconds\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return failures.load() ? 1 : 0;
}
