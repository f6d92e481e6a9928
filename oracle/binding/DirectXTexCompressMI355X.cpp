// DirectXTexCompressMI355X.cpp
#include "DirectXTexP.h"
#include <dxtex_amd.h>
#include "MI355XContext.h"

// (this declaration goes into DirectXTex.h, next to the ID3D11Device* overloads at :946-963)
namespace DirectX
{
    HRESULT __cdecl CompressMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress,
                                   float threshold, ScratchImage& image) noexcept;
}

namespace
{
    inline dxtex_image View(const DirectX::Image& i) noexcept
    {
        return dxtex_image{ i.width, i.height, int32_t(i.format), i.rowPitch, i.slicePitch, i.pixels };
    }

}

// Same shape as Compress(ID3D11Device*, const Image&, DXGI_FORMAT, TEX_COMPRESS_FLAGS, float, ScratchImage&)
// (DirectXTex.h:946-950) with the device handle replaced by a HIP device ordinal.
_Use_decl_annotations_
HRESULT DirectX::CompressMI355X(int hipDevice, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress,
                                float threshold, ScratchImage& image) noexcept
{
    if (IsCompressed(srcImage.format) || !IsCompressed(format)) return E_INVALIDARG;          // :671-672
    if (IsTypeless(format) || IsTypeless(srcImage.format) || IsPlanar(srcImage.format) || IsPalettized(srcImage.format))
        return HRESULT_E_NOT_SUPPORTED;                                                       // :674-676
    image.Release();                                                                          // :679
    HRESULT hr = image.Initialize2D(format, srcImage.width, srcImage.height, 1, 1);
    if (FAILED(hr)) return hr;
    dxtex_ctx* const ctx = MI355X::ContextFor(hipDevice);          // one per (thread, device), kept: MI355XContext.h
    if (!ctx) { image.Release(); return E_FAIL; }
    // GPUCompressBC::Prepare's call (DirectXTexCompressGPU.cpp:392): sizes the context's staging and search scratch for this shape; a no-op
    // from the second texture of a size on, because the context - and what it holds - outlives this call
    hr = HRESULT(dxtex_ctx_prepare(ctx, srcImage.width, srcImage.height, int32_t(srcImage.format), int32_t(format), uint32_t(compress), 1, nullptr));
    if (FAILED(hr)) { image.Release(); return hr; }
    const dxtex_image src = View(srcImage), dst = View(*image.GetImage(0, 0, 0));
    hr = HRESULT(dxtex_compress(ctx, &src, &dst, uint32_t(compress), threshold));
    if (FAILED(hr)) image.Release();                                                          // :713-717
    return hr;
}
