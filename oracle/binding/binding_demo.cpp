// TEST INFRASTRUCTURE (oracle/_ref/binding_demo, run by tests/test_zz_binding_gpu.py on the GPU box): the drop-in claim end to end.
// The reference's OWN types and CPU encoder (DirectX::Image, ScratchImage, DirectX::Compress - compiled in place into
// libdxtex_ref.so) next to the binding INTEGRATION.md documents (DirectXTexCompressMI355X.cpp, the file beside this one), which
// fills the reference's ScratchImage through libdxtex_amd.so. Same image in, the two ScratchImages must hold the same bytes.
#include "DirectXTexP.h"
#include <cstdio>
#include <cstring>
#include <vector>

namespace DirectX
{
    HRESULT CompressMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& image) noexcept;
    // DirectXTexMI355X.cpp: the array overload (one Prepare per mip size, then one dxtex_compress_many) and the other entry points of the path
    HRESULT CompressMI355X(int hipDevice, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                           TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImages) noexcept;
    HRESULT DecompressMI355X(int hipDevice, const Image& cImage, DXGI_FORMAT format, ScratchImage& image) noexcept;
    HRESULT GenerateMipMapsMI355X(int hipDevice, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept;
    HRESULT ResizeMI355X(int hipDevice, const Image& srcImage, size_t width, size_t height, TEX_FILTER_FLAGS filter, ScratchImage& image) noexcept;
    HRESULT ConvertMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, ScratchImage& image) noexcept;
}

namespace
{
    bool SameScratch(const DirectX::ScratchImage& a, const DirectX::ScratchImage& b)
    {
        const DirectX::TexMetadata &ma = a.GetMetadata(), &mb = b.GetMetadata();
        return a.GetPixels() && b.GetPixels() && a.GetImageCount() == b.GetImageCount() && a.GetPixelsSize() == b.GetPixelsSize() &&
               ma.width == mb.width && ma.height == mb.height && ma.format == mb.format && ma.mipLevels == mb.mipLevels && ma.arraySize == mb.arraySize &&
               std::memcmp(a.GetPixels(), b.GetPixels(), a.GetPixelsSize()) == 0;
    }
}

using namespace DirectX;

int main()
{
    const size_t W = 72, H = 40;                       // not a multiple of 4 in neither direction? 72 is, 40 is: add a ragged case below
    int failures = 0;
    for (int variant = 0; variant < 2; ++variant)
    {
        const size_t w = variant ? 70 : W, h = variant ? 38 : H;
        std::vector<uint8_t> px(w * h * 4);
        uint32_t s = 2024u + uint32_t(variant);
        for (size_t y = 0; y < h; ++y)
            for (size_t x = 0; x < w; ++x)
            {
                s = s * 1664525u + 1013904223u;
                uint8_t* p = &px[(y * w + x) * 4];
                p[0] = uint8_t(x * 3 + ((s >> 24) & 15)); p[1] = uint8_t(y * 5 + ((s >> 20) & 15)); p[2] = uint8_t(x + y + ((s >> 16) & 31)); p[3] = uint8_t(255 - ((s >> 8) & 127));
            }
        Image src = {};
        src.width = w; src.height = h; src.format = DXGI_FORMAT_R8G8B8A8_UNORM; src.rowPitch = w * 4; src.slicePitch = w * h * 4; src.pixels = px.data();
        for (DXGI_FORMAT fmt : { DXGI_FORMAT_BC1_UNORM, DXGI_FORMAT_BC3_UNORM, DXGI_FORMAT_BC5_UNORM, DXGI_FORMAT_BC7_UNORM })
        {
            ScratchImage cpu, gpu;
            const HRESULT hrCpu = Compress(src, fmt, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, cpu);
            const HRESULT hrGpu = CompressMI355X(0, src, fmt, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, gpu);
            const bool same = SUCCEEDED(hrCpu) && SUCCEEDED(hrGpu) && cpu.GetPixelsSize() == gpu.GetPixelsSize()
                              && std::memcmp(cpu.GetPixels(), gpu.GetPixels(), cpu.GetPixelsSize()) == 0
                              && gpu.GetMetadata().format == fmt && gpu.GetMetadata().width == w && gpu.GetMetadata().height == h;
            std::printf("%zux%zu format %d: reference CPU %08X, binding over libdxtex_amd.so %08X, %s\n", w, h, int(fmt), unsigned(hrCpu), unsigned(hrGpu),
                        same ? "identical ScratchImages" : "DIFFERENT");
            if (!same) ++failures;
        }
        // the reference's argument checks come back through the binding unchanged
        ScratchImage bad;
        if (CompressMI355X(0, src, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bad) != E_INVALIDARG) { std::puts("expected E_INVALIDARG for an uncompressed target"); ++failures; }
    }
    // ---- the other entry points, each against the reference's own CPU function on the reference's own ScratchImage ---------------------
    {
        auto report = [&](const char* what, HRESULT hrCpu, HRESULT hrGpu, bool same)
        {
            std::printf("%s: reference CPU %08X, binding over libdxtex_amd.so %08X, %s\n", what, unsigned(hrCpu), unsigned(hrGpu), same ? "identical ScratchImages" : "DIFFERENT");
            if (!same) ++failures;
        };
        // a texture array with mips: 2 items x 3 levels of 44 x 28 (levels 22 x 14 and 11 x 7 have partial blocks)
        ScratchImage arr;
        if (FAILED(arr.Initialize2D(DXGI_FORMAT_R8G8B8A8_UNORM, 44, 28, 2, 3))) { std::puts("Initialize2D failed"); return 1; }
        uint32_t s = 77u;
        for (size_t i = 0; i < arr.GetPixelsSize(); ++i) { s = s * 1664525u + 1013904223u; arr.GetPixels()[i] = uint8_t((i * 7u) / 5u + ((s >> 24) & 31u)); }
        for (DXGI_FORMAT fmt : { DXGI_FORMAT_BC1_UNORM, DXGI_FORMAT_BC7_UNORM })
        {
            ScratchImage cpu, gpu;
            const HRESULT hrCpu = Compress(arr.GetImages(), arr.GetImageCount(), arr.GetMetadata(), fmt, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, cpu);
            const HRESULT hrGpu = CompressMI355X(0, arr.GetImages(), arr.GetImageCount(), arr.GetMetadata(), fmt, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, gpu);
            report(fmt == DXGI_FORMAT_BC1_UNORM ? "Compress (array 2 x 3 mips) -> BC1" : "Compress (array 2 x 3 mips) -> BC7", hrCpu, hrGpu,
                   SUCCEEDED(hrCpu) && SUCCEEDED(hrGpu) && SameScratch(cpu, gpu));
            if (fmt == DXGI_FORMAT_BC7_UNORM && SUCCEEDED(hrCpu))
            {
                ScratchImage dcpu, dgpu;
                const Image& c = *cpu.GetImage(0, 1, 0);
                const HRESULT h1 = Decompress(c, DXGI_FORMAT_R8G8B8A8_UNORM, dcpu);
                const HRESULT h2 = DecompressMI355X(0, c, DXGI_FORMAT_R8G8B8A8_UNORM, dgpu);
                report("Decompress BC7 -> RGBA8", h1, h2, SUCCEEDED(h1) && SUCCEEDED(h2) && SameScratch(dcpu, dgpu));
            }
        }
        {
            ScratchImage bad;
            TexMetadata m = arr.GetMetadata();
            if (CompressMI355X(0, arr.GetImages(), arr.GetImageCount() - 1, m, DXGI_FORMAT_BC1_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bad) != E_FAIL || bad.GetPixels())
            { std::puts("expected E_FAIL and a released result for a short image array"); ++failures; }                     // DirectXTexCompressGPU.cpp:360-364
        }
        const Image& base = *arr.GetImage(0, 0, 0);
        for (TEX_FILTER_FLAGS flt : { TEX_FILTER_LINEAR, TEX_FILTER_CUBIC, TEX_FILTER_TRIANGLE })
        {
            ScratchImage cpu, gpu;
            const HRESULT h1 = GenerateMipMaps(base, flt, 0, cpu);
            const HRESULT h2 = GenerateMipMapsMI355X(0, base, flt, 0, gpu);
            report(flt == TEX_FILTER_LINEAR ? "GenerateMipMaps 44 x 28 linear, full chain" : flt == TEX_FILTER_CUBIC ? "GenerateMipMaps 44 x 28 cubic, full chain" : "GenerateMipMaps 44 x 28 triangle, full chain",
                   h1, h2, SUCCEEDED(h1) && SUCCEEDED(h2) && SameScratch(cpu, gpu));
        }
        {
            ScratchImage cpu, gpu;
            const HRESULT h1 = Resize(base, 61, 17, TEX_FILTER_CUBIC, cpu);
            const HRESULT h2 = ResizeMI355X(0, base, 61, 17, TEX_FILTER_CUBIC, gpu);
            report("Resize 44 x 28 -> 61 x 17 cubic", h1, h2, SUCCEEDED(h1) && SUCCEEDED(h2) && SameScratch(cpu, gpu));
        }
        {
            // DirectX::Convert itself (DirectXTexConvert.cpp compiled in place into libdxtex_ref.so since round 6)
            ScratchImage cpu, gpu;
            const HRESULT h1 = Convert(base, DXGI_FORMAT_R16G16B16A16_FLOAT, TEX_FILTER_DEFAULT, TEX_THRESHOLD_DEFAULT, cpu);
            const HRESULT h2 = ConvertMI355X(0, base, DXGI_FORMAT_R16G16B16A16_FLOAT, TEX_FILTER_DEFAULT, TEX_THRESHOLD_DEFAULT, gpu);
            report("Convert RGBA8 -> RGBA16F", h1, h2, SUCCEEDED(h1) && SUCCEEDED(h2) && SameScratch(cpu, gpu));
            ScratchImage bad;
            if (ConvertMI355X(0, base, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_FILTER_DEFAULT, 0.5f, bad) != E_INVALIDARG) { std::puts("expected E_INVALIDARG for Convert to the same format"); ++failures; }
        }
    }
    std::puts(failures ? "binding demo FAILED" : "binding demo OK");
    return failures ? 1 : 0;
}
