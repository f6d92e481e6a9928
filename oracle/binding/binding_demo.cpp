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
    std::puts(failures ? "binding demo FAILED" : "binding demo OK");
    return failures ? 1 : 0;
}
