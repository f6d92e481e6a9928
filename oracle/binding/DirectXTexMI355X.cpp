// DirectXTexMI355X.cpp - the rest of the path's public entry points bound to libdxtex_amd.so, written against the reference's own
// headers. Each function has the shape of the reference function it stands beside, with the ID3D11Device* (where
// there is one) replaced by a HIP device ordinal; the output ScratchImage is created by the reference's own code, the pixels are
// produced by the C ABI, and every failure releases the result as the reference does.
#include "DirectXTexP.h"
#include <dxtex_amd.h>
#include "MI355XContext.h"
#include <vector>

namespace DirectX
{
    // next to Compress(ID3D11Device*, const Image*, size_t, const TexMetadata&, ...) (DirectXTex.h:951-955)
    HRESULT __cdecl CompressMI355X(int hipDevice, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                                   TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImages) noexcept;
    // next to Decompress (DirectXTex.h:965-968)
    HRESULT __cdecl DecompressMI355X(int hipDevice, const Image& cImage, DXGI_FORMAT format, ScratchImage& image) noexcept;
    // next to GenerateMipMaps (DirectXTex.h:841-843)
    HRESULT __cdecl GenerateMipMapsMI355X(int hipDevice, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept;
    // next to Resize (DirectXTex.h:799-801)
    HRESULT __cdecl ResizeMI355X(int hipDevice, const Image& srcImage, size_t width, size_t height, TEX_FILTER_FLAGS filter, ScratchImage& image) noexcept;
    // next to Convert (DirectXTex.h:818-820)
    HRESULT __cdecl ConvertMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, ScratchImage& image) noexcept;
}

namespace
{
    inline dxtex_image View(const DirectX::Image& i) noexcept
    {
        return dxtex_image{ i.width, i.height, int32_t(i.format), i.rowPitch, i.slicePitch, i.pixels };
    }

}

// DirectXTexCompressGPU.cpp:320-470: one Prepare per mip size (GPUCompressBC::Prepare, :392), then the images of that size.
// Here the whole set goes to the GPU in ONE dxtex_compress_many call after the Prepare calls: BC6H / BC7 arrays run as one block list.
_Use_decl_annotations_
HRESULT DirectX::CompressMI355X(int hipDevice, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                                TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImages) noexcept
{
    if (!srcImages || !nimages || !IsValid(metadata.format)) return E_INVALIDARG;                    // :332-333
    if (IsCompressed(metadata.format) || !IsCompressed(format)) return E_INVALIDARG;                  // :335-336
    if (IsTypeless(format) || IsTypeless(metadata.format) || IsPlanar(metadata.format) || IsPalettized(metadata.format))
        return HRESULT_E_NOT_SUPPORTED;                                                               // :338-340
    cImages.Release();
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) return E_FAIL;
    TexMetadata mdata2 = metadata;
    mdata2.format = format;
    HRESULT hr = cImages.Initialize(mdata2);                                                          // :353-358
    if (FAILED(hr)) return hr;
    if (nimages != cImages.GetImageCount()) { cImages.Release(); return E_FAIL; }                     // :360-364
    const Image* dest = cImages.GetImages();
    if (!dest) { cImages.Release(); return E_POINTER; }
    if (metadata.dimension == TEX_DIMENSION_TEXTURE3D) { cImages.Release(); return HRESULT_E_NOT_SUPPORTED; }      // (volumes: per-slice loop, not bound here)

    std::vector<dxtex_image> srcs, dsts;
    size_t w = metadata.width, h = metadata.height;
    for (size_t level = 0; level < metadata.mipLevels; ++level)
    {
        size_t held = 0;
        hr = HRESULT(dxtex_ctx_prepare(ctx, w, h, int32_t(metadata.format), int32_t(format), uint32_t(compress), metadata.arraySize, &held));      // Prepare, :392
        if (FAILED(hr)) { cImages.Release(); return hr; }
        for (size_t item = 0; item < metadata.arraySize; ++item)
        {
            const size_t index = metadata.ComputeIndex(level, item, 0);
            if (index >= nimages) { cImages.Release(); return E_FAIL; }                               // :403-407
            const Image& src = srcImages[index];
            if (src.width != dest[index].width || src.height != dest[index].height) { cImages.Release(); return E_FAIL; }      // :413-417
            srcs.push_back(View(src)); dsts.push_back(View(dest[index]));
        }
        if (h > 1) h >>= 1;
        if (w > 1) w >>= 1;
    }
    hr = HRESULT(dxtex_compress_many(ctx, srcs.data(), dsts.data(), srcs.size(), uint32_t(compress), threshold));
    if (FAILED(hr)) cImages.Release();
    return hr;
}

// DirectXTexCompress.cpp:852-910
_Use_decl_annotations_
HRESULT DirectX::DecompressMI355X(int hipDevice, const Image& cImage, DXGI_FORMAT format, ScratchImage& image) noexcept
{
    if (!IsCompressed(cImage.format) || IsCompressed(format)) return E_INVALIDARG;                    // :858-859
    if (format == DXGI_FORMAT_UNKNOWN)
    {
        // DefaultDecompress (:377-421) is file-local in the reference; a caller that wants its default passes the format it names there
        return E_INVALIDARG;
    }
    if (!IsValid(format)) return E_INVALIDARG;
    if (IsTypeless(format) || IsPlanar(format) || IsPalettized(format)) return HRESULT_E_NOT_SUPPORTED;      // :873-874
    image.Release();
    HRESULT hr = image.Initialize2D(format, cImage.width, cImage.height, 1, 1);                        // :878-880
    if (FAILED(hr)) return hr;
    const Image* img = image.GetImage(0, 0, 0);
    if (!img) { image.Release(); return E_POINTER; }
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) { image.Release(); return E_FAIL; }
    const dxtex_image src = View(cImage), dst = View(*img);
    hr = HRESULT(dxtex_decompress(ctx, &src, &dst));
    if (FAILED(hr)) image.Release();                                                                   // :890-894
    return hr;
}

// DirectXTexMipmaps.cpp:2828-3017 (non-WIC path): Setup2DMips (:851-904) creates the chain and copies the base image, the per-level
// filter loop (Generate2DMips*Filter, :907-1602) becomes ONE dxtex_generate_mips call over the Images of the chain.
_Use_decl_annotations_
HRESULT DirectX::GenerateMipMapsMI355X(int hipDevice, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept
{
    if (!IsValid(baseImage.format)) return E_INVALIDARG;
    if (!baseImage.pixels) return E_POINTER;
    if (!CalculateMipLevels(baseImage.width, baseImage.height, levels)) return E_INVALIDARG;           // :2840-2841
    if (levels <= 1) return E_INVALIDARG;
    if (IsCompressed(baseImage.format) || IsTypeless(baseImage.format) || IsPlanar(baseImage.format) || IsPalettized(baseImage.format))
        return HRESULT_E_NOT_SUPPORTED;                                                                // :2846-2849
    mipChain.Release();
    HRESULT hr = mipChain.Initialize2D(baseImage.format, baseImage.width, baseImage.height, 1, levels);      // Setup2DMips, :863-865
    if (FAILED(hr)) return hr;
    const Image* dest = mipChain.GetImage(0, 0, 0);
    if (!dest) { mipChain.Release(); return E_POINTER; }
    {
        // copy the base image to level 0 row by row (:877-897)
        const uint8_t* pSrc = baseImage.pixels; uint8_t* pDest = dest->pixels;
        const size_t size = std::min<size_t>(baseImage.rowPitch, dest->rowPitch);
        for (size_t y = 0; y < baseImage.height; ++y) { memcpy(pDest, pSrc, size); pSrc += baseImage.rowPitch; pDest += dest->rowPitch; }
    }
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) { mipChain.Release(); return E_FAIL; }
    std::vector<dxtex_image> chain;
    for (size_t level = 0; level < levels; ++level) chain.push_back(View(*mipChain.GetImage(level, 0, 0)));
    hr = HRESULT(dxtex_generate_mips(ctx, chain.data(), chain.size(), uint32_t(filter)));
    if (FAILED(hr)) mipChain.Release();
    return hr;
}

// DirectXTexResize.cpp:854-930 (custom-filter path)
_Use_decl_annotations_
HRESULT DirectX::ResizeMI355X(int hipDevice, const Image& srcImage, size_t width, size_t height, TEX_FILTER_FLAGS filter, ScratchImage& image) noexcept
{
    if (width == 0 || height == 0) return E_INVALIDARG;                                                 // :862-863
    if ((srcImage.width > UINT32_MAX) || (srcImage.height > UINT32_MAX) || (width > UINT32_MAX) || (height > UINT32_MAX)) return E_INVALIDARG;
    if (!srcImage.pixels) return E_POINTER;
    if (IsCompressed(srcImage.format)) return HRESULT_E_NOT_SUPPORTED;                                  // :872-876
    image.Release();
    HRESULT hr = image.Initialize2D(srcImage.format, width, height, 1, 1);                              // :887-889
    if (FAILED(hr)) return hr;
    const Image* rimage = image.GetImage(0, 0, 0);
    if (!rimage) { image.Release(); return E_POINTER; }
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) { image.Release(); return E_FAIL; }
    const dxtex_image src = View(srcImage), dst = View(*rimage);
    hr = HRESULT(dxtex_resize(ctx, &src, &dst, uint32_t(filter)));
    if (FAILED(hr)) image.Release();                                                                    // :922-926
    return hr;
}

// DirectXTexConvert.cpp:5091-5176 (ConvertEx, non-WIC path -> ConvertCustom)
_Use_decl_annotations_
HRESULT DirectX::ConvertMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, ScratchImage& image) noexcept
{
    if ((srcImage.format == format) || !IsValid(format)) return E_INVALIDARG;                           // :5115-5116
    if (!srcImage.pixels) return E_POINTER;
    if (IsCompressed(srcImage.format) || IsCompressed(format) || IsPlanar(srcImage.format) || IsPlanar(format)
        || IsPalettized(srcImage.format) || IsPalettized(format) || IsTypeless(srcImage.format) || IsTypeless(format))
        return HRESULT_E_NOT_SUPPORTED;                                                                 // :5121-5126
    if ((srcImage.width > UINT32_MAX) || (srcImage.height > UINT32_MAX)) return E_INVALIDARG;
    image.Release();
    HRESULT hr = image.Initialize2D(format, srcImage.width, srcImage.height, 1, 1);                     // :5131-5133
    if (FAILED(hr)) return hr;
    const Image* rimage = image.GetImage(0, 0, 0);
    if (!rimage) { image.Release(); return E_POINTER; }
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) { image.Release(); return E_FAIL; }
    const dxtex_image src = View(srcImage), dst = View(*rimage);
    hr = HRESULT(dxtex_convert(ctx, &src, &dst, uint32_t(filter), threshold));
    if (FAILED(hr)) image.Release();                                                                    // :5168-5172
    return hr;
}
