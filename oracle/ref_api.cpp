// TEST INFRASTRUCTURE ONLY (oracle/_ref): C ABI around the reference's own block codecs
// (DirectXTex/BC.h:321-343), compiled in place from /root/reference by oracle/Makefile.
// Used by tests/ as the parity checker and by bench.py's cpu_baseline leg (kind "reference").
// Never linked into, imported by, or called from the product library.
#include "DirectXTexP.h"
#include "BC.h"
#include "filters.h"
#include <omp.h>
#include <vector>
#include <algorithm>

using namespace DirectX;

namespace
{
    enum : int
    {
        F_BC1 = 71, F_BC1_SRGB = 72, F_BC2 = 74, F_BC2_SRGB = 75, F_BC3 = 77, F_BC3_SRGB = 78,
        F_BC4U = 80, F_BC4S = 81, F_BC5U = 83, F_BC5S = 84, F_BC6HU = 95, F_BC6HS = 96, F_BC7 = 98, F_BC7_SRGB = 99
    };

    size_t BlockBytes(int fmt)
    {
        switch (fmt)
        {
        case F_BC1: case F_BC1_SRGB: case F_BC4U: case F_BC4S: return 8;
        case F_BC2: case F_BC2_SRGB: case F_BC3: case F_BC3_SRGB: case F_BC5U: case F_BC5S:
        case F_BC6HU: case F_BC6HS: case F_BC7: case F_BC7_SRGB: return 16;
        default: return 0;
        }
    }
}

extern "C"
{
    int dxtex_ref_num_threads() { return omp_get_max_threads(); }

    // Same dispatch as DetermineEncoderSettings (DirectXTexCompress.cpp:46-68) + the BC1 special case
    // (DirectXTexCompress.cpp:191-194). rgba = nblocks x 16 texels x 4 floats (row-major 4x4 tiles).
    int dxtex_ref_encode_blocks(int fmt, uint32_t bcflags, float threshold, const float* rgba, size_t nblocks,
                                uint8_t* out, int threads)
    {
        const size_t bb = BlockBytes(fmt);
        if (!bb || !rgba || !out) return -1;
        if (threads <= 0) threads = omp_get_max_threads();
        #pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
        for (int64_t nb = 0; nb < int64_t(nblocks); ++nb)
        {
            XMVECTOR temp[16];
            memcpy(temp, rgba + size_t(nb) * 64, sizeof(temp));
            uint8_t* p = out + size_t(nb) * bb;
            switch (fmt)
            {
            case F_BC1: case F_BC1_SRGB: D3DXEncodeBC1(p, temp, threshold, bcflags); break;
            case F_BC2: case F_BC2_SRGB: D3DXEncodeBC2(p, temp, bcflags); break;
            case F_BC3: case F_BC3_SRGB: D3DXEncodeBC3(p, temp, bcflags); break;
            case F_BC4U: D3DXEncodeBC4U(p, temp, bcflags); break;
            case F_BC4S: D3DXEncodeBC4S(p, temp, bcflags); break;
            case F_BC5U: D3DXEncodeBC5U(p, temp, bcflags); break;
            case F_BC5S: D3DXEncodeBC5S(p, temp, bcflags); break;
            case F_BC6HU: D3DXEncodeBC6HU(p, temp, bcflags); break;
            case F_BC6HS: D3DXEncodeBC6HS(p, temp, bcflags); break;
            case F_BC7: case F_BC7_SRGB: D3DXEncodeBC7(p, temp, bcflags); break;
            default: break;
            }
        }
        return 0;
    }

    int dxtex_ref_decode_blocks(int fmt, const uint8_t* in, size_t nblocks, float* rgba)
    {
        const size_t bb = BlockBytes(fmt);
        if (!bb || !rgba || !in) return -1;
        #pragma omp parallel for schedule(static)
        for (int64_t nb = 0; nb < int64_t(nblocks); ++nb)
        {
            XMVECTOR temp[16];
            const uint8_t* p = in + size_t(nb) * bb;
            switch (fmt)
            {
            case F_BC1: case F_BC1_SRGB: D3DXDecodeBC1(temp, p); break;
            case F_BC2: case F_BC2_SRGB: D3DXDecodeBC2(temp, p); break;
            case F_BC3: case F_BC3_SRGB: D3DXDecodeBC3(temp, p); break;
            case F_BC4U: D3DXDecodeBC4U(temp, p); break;
            case F_BC4S: D3DXDecodeBC4S(temp, p); break;
            case F_BC5U: D3DXDecodeBC5U(temp, p); break;
            case F_BC5S: D3DXDecodeBC5S(temp, p); break;
            case F_BC6HU: D3DXDecodeBC6HU(temp, p); break;
            case F_BC6HS: D3DXDecodeBC6HS(temp, p); break;
            case F_BC7: case F_BC7_SRGB: D3DXDecodeBC7(temp, p); break;
            default: break;
            }
            memcpy(rgba + size_t(nb) * 64, temp, sizeof(temp));
        }
        return 0;
    }

    // ---- image-level drivers: the reference's own Compress / Decompress / GenerateMipMaps / Resize / ComputeMSE
    // (DirectXTexCompress.cpp, DirectXTexMipmaps.cpp, DirectXTexResize.cpp, DirectXTexMisc.cpp compiled in place) on top
    // of the reference's own scanline layer (DirectXTexConvert.cpp, compiled in place: ref_convert.cpp). Results are copied out level by
    // level with tight pitch.
    static Image make_image(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch)
    {
        Image img;
        img.width = w; img.height = h; img.format = DXGI_FORMAT(fmt);
        size_t rp = 0, sp = 0;
        ComputePitch(img.format, w, h, rp, sp);
        img.rowPitch = rowPitch ? rowPitch : rp;
        const size_t rows = IsCompressed(img.format) ? std::max<size_t>(1, (h + 3) / 4) : h;
        img.slicePitch = img.rowPitch * rows;
        img.pixels = const_cast<uint8_t*>(pixels);
        return img;
    }

    static int64_t copy_out(const ScratchImage& si, uint8_t* out, size_t capacity)
    {
        size_t at = 0;
        for (size_t i = 0; i < si.GetImageCount(); ++i)
        {
            const Image& im = si.GetImages()[i];
            size_t rp = 0, sp = 0;
            ComputePitch(im.format, im.width, im.height, rp, sp);
            if (at + sp > capacity) return -2;
            const size_t rows = sp / rp;
            for (size_t y = 0; y < rows; ++y) memcpy(out + at + y * rp, im.pixels + y * im.rowPitch, rp);
            at += sp;
        }
        return int64_t(at);
    }

    // returns bytes written (>= 0) or the negated HRESULT-ish failure: -1 = call failed (hr in *hrOut), -2 = capacity
    int64_t dxtex_ref_compress(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, int dstFmt, uint32_t flags, float threshold,
                               uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = Compress(make_image(pixels, w, h, fmt, rowPitch), DXGI_FORMAT(dstFmt), TEX_COMPRESS_FLAGS(flags), threshold, si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    int64_t dxtex_ref_decompress(const uint8_t* payload, size_t w, size_t h, int fmt, int dstFmt, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = Decompress(make_image(payload, w, h, fmt, 0), DXGI_FORMAT(dstFmt), si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    int64_t dxtex_ref_generate_mips(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, uint32_t filter, size_t levels,
                                    uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = GenerateMipMaps(make_image(pixels, w, h, fmt, rowPitch), TEX_FILTER_FLAGS(filter), levels, si, false);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    int64_t dxtex_ref_resize(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, size_t nw, size_t nh, uint32_t filter,
                             uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = Resize(make_image(pixels, w, h, fmt, rowPitch), nw, nh, TEX_FILTER_FLAGS(filter), si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    // The reference's own Convert (DirectXTexConvert.cpp:5091-5180 -> ConvertCustom :4804-4913), compiled in place (ref_convert.cpp)
    int64_t dxtex_ref_convert(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, int dstFmt, uint32_t filter, float threshold,
                              uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = Convert(make_image(pixels, w, h, fmt, rowPitch), DXGI_FORMAT(dstFmt), TEX_FILTER_FLAGS(filter), threshold, si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    // One row through the reference's LoadScanline / StoreScanline (DirectXTexConvert.cpp:779-1619, :1643-2533) with fp32 RGBA on the
    // other side: the leaves of oracle/shim/DirectXPackedVector.h under the reference's own case analysis, for the leaf tests.
    int dxtex_ref_load_scanline(const uint8_t* src, size_t size, int fmt, float* rgba, size_t count)
    {
        std::vector<XMVECTOR> row(count);
        if (!Internal::LoadScanline(row.data(), count, src, size, DXGI_FORMAT(fmt))) return -1;
        memcpy(rgba, row.data(), count * sizeof(XMVECTOR));
        return 0;
    }
    int dxtex_ref_store_scanline(uint8_t* dst, size_t size, int fmt, const float* rgba, size_t count, float threshold)
    {
        std::vector<XMVECTOR> row(count);
        memcpy(row.data(), rgba, count * sizeof(XMVECTOR));
        return Internal::StoreScanline(dst, size, DXGI_FORMAT(fmt), row.data(), count, threshold) ? 0 : -1;
    }

    // GenerateMipMaps3D (DirectXTexMipmaps.cpp:3254-3361). `pixels`: the `depth` base slices, tight pitch, consecutive.
    // Output: the volume's images in ScratchImage order (level by level, the level's slices consecutive), tight pitch.
    int64_t dxtex_ref_generate_mips3d(const uint8_t* pixels, size_t w, size_t h, size_t d, int fmt, uint32_t filter, size_t levels,
                                      uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        std::vector<Image> base(d);
        size_t rp = 0, sp = 0;
        ComputePitch(DXGI_FORMAT(fmt), w, h, rp, sp);
        for (size_t z = 0; z < d; ++z) base[z] = make_image(pixels + z * sp, w, h, fmt, 0);
        ScratchImage si;
        const HRESULT hr = GenerateMipMaps3D(base.data(), d, TEX_FILTER_FLAGS(filter), levels, si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    // PremultiplyAlpha (DirectXTexPMAlpha.cpp:214-262); flags = TEX_PMALPHA_*
    int64_t dxtex_ref_premultiply_alpha(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, uint32_t flags,
                                        uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        const HRESULT hr = PremultiplyAlpha(make_image(pixels, w, h, fmt, rowPitch), TEX_PMALPHA_FLAGS(flags), si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    // ScaleMipMapsAlphaForCoverage (DirectXTexMipmaps.cpp:3483-3556) over a tight mip chain of `levels` images
    int64_t dxtex_ref_scale_mips_alpha(const uint8_t* chain, size_t w, size_t h, int fmt, size_t levels, float alphaReference,
                                       uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage in, si;
        HRESULT hr = in.Initialize2D(DXGI_FORMAT(fmt), w, h, 1, levels);
        if (SUCCEEDED(hr)) hr = si.Initialize2D(DXGI_FORMAT(fmt), w, h, 1, levels);
        if (FAILED(hr)) { if (hrOut) *hrOut = int32_t(hr); return -1; }
        size_t at = 0;
        for (size_t l = 0; l < levels; ++l)
        {
            const Image& im = in.GetImages()[l];
            size_t rp = 0, sp = 0;
            ComputePitch(im.format, im.width, im.height, rp, sp);
            for (size_t y = 0; y < im.height; ++y) memcpy(im.pixels + y * im.rowPitch, chain + at + y * rp, rp);
            at += sp;
        }
        hr = ScaleMipMapsAlphaForCoverage(in.GetImages(), in.GetImageCount(), in.GetMetadata(), 0, alphaReference, si);
        if (hrOut) *hrOut = int32_t(hr);
        return FAILED(hr) ? -1 : copy_out(si, out, capacity);
    }

    int dxtex_ref_compute_mse(const uint8_t* a, int fmtA, const uint8_t* b, int fmtB, size_t w, size_t h, float* mse, float* mseV)
    {
        float m = 0.f;
        const HRESULT hr = ComputeMSE(make_image(a, w, h, fmtA, 0), make_image(b, w, h, fmtB, 0), m, mseV, CMSE_DEFAULT);
        if (mse) *mse = m;
        return int(hr);
    }

    // ---- DDS container (DirectXTexDDS.cpp compiled in place): used to pin the host layer's reader / writer -----------------
    // `pixels`: the images of the texture in ScratchImage order, each with its default (tight) pitch.
    int64_t dxtex_ref_save_dds(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t arraySize, size_t mipLevels, uint32_t miscFlags,
                               uint32_t ddsFlags, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        TexMetadata m = {};
        m.width = w; m.height = h; m.depth = 1; m.arraySize = arraySize; m.mipLevels = mipLevels; m.miscFlags = miscFlags;
        m.format = DXGI_FORMAT(fmt); m.dimension = TEX_DIMENSION_TEXTURE2D;
        ScratchImage si;
        HRESULT hr = si.Initialize(m);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        memcpy(si.GetPixels(), pixels, si.GetPixelsSize());
        Blob blob;
        hr = SaveToDDSMemory(si.GetImages(), si.GetImageCount(), si.GetMetadata(), DDS_FLAGS(ddsFlags), blob);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        if (blob.GetBufferSize() > capacity) return -2;
        memcpy(out, blob.GetBufferPointer(), blob.GetBufferSize());
        return int64_t(blob.GetBufferSize());
    }

    // the same for a volume texture: `pixels` in ScratchImage order (level by level, slices consecutive), tight pitches
    int64_t dxtex_ref_save_dds_volume(const uint8_t* pixels, size_t w, size_t h, size_t d, int fmt, size_t mipLevels, uint32_t ddsFlags,
                                      uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si;
        HRESULT hr = si.Initialize3D(DXGI_FORMAT(fmt), w, h, d, mipLevels);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        memcpy(si.GetPixels(), pixels, si.GetPixelsSize());
        Blob blob;
        hr = SaveToDDSMemory(si.GetImages(), si.GetImageCount(), si.GetMetadata(), DDS_FLAGS(ddsFlags), blob);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        if (blob.GetBufferSize() > capacity) return -2;
        memcpy(out, blob.GetBufferPointer(), blob.GetBufferSize());
        return int64_t(blob.GetBufferSize());
    }

    // meta[0..6] = width, height, format, arraySize, mipLevels, miscFlags, miscFlags2
    int64_t dxtex_ref_load_dds(const uint8_t* file, size_t size, uint64_t* meta, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si; TexMetadata m = {};
        const HRESULT hr = LoadFromDDSMemory(file, size, DDS_FLAGS_NONE, &m, si);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        meta[0] = m.width; meta[1] = m.height; meta[2] = uint64_t(m.format); meta[3] = m.arraySize; meta[4] = m.mipLevels; meta[5] = m.miscFlags; meta[6] = m.miscFlags2;
        return copy_out(si, out, capacity);
    }

    // LoadFromDDSMemory with flags. meta[0..8] = width, height, depth, format, arraySize, mipLevels, miscFlags, miscFlags2, dimension.
    // Returns the number of pixel bytes (ScratchImage order, the result's own pitches), -1 on failure (hrOut), -2 if `out` is too small.
    int64_t dxtex_ref_load_dds_ex(const uint8_t* file, size_t size, uint32_t ddsFlags, uint64_t* meta, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si; TexMetadata m = {};
        const HRESULT hr = LoadFromDDSMemory(file, size, DDS_FLAGS(ddsFlags), &m, si);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        meta[0] = m.width; meta[1] = m.height; meta[2] = m.depth; meta[3] = uint64_t(m.format); meta[4] = m.arraySize; meta[5] = m.mipLevels;
        meta[6] = m.miscFlags; meta[7] = m.miscFlags2; meta[8] = uint64_t(m.dimension);
        if (si.GetPixelsSize() > capacity) return -2;
        memcpy(out, si.GetPixels(), si.GetPixelsSize());
        return int64_t(si.GetPixelsSize());
    }

    // SaveToDDSMemory of any texture: `pixels` in ScratchImage order with default pitches; dimension 2 / 3 / 4 as TEX_DIMENSION.
    int64_t dxtex_ref_save_dds_ex(const uint8_t* pixels, size_t w, size_t h, size_t d, int fmt, size_t arraySize, size_t mipLevels, uint32_t miscFlags,
                                  uint32_t miscFlags2, uint32_t dimension, uint32_t ddsFlags, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        TexMetadata m = {};
        m.width = w; m.height = h; m.depth = d; m.arraySize = arraySize; m.mipLevels = mipLevels; m.miscFlags = miscFlags; m.miscFlags2 = miscFlags2;
        m.format = DXGI_FORMAT(fmt); m.dimension = TEX_DIMENSION(dimension);
        ScratchImage si;
        HRESULT hr = si.Initialize(m);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        memcpy(si.GetPixels(), pixels, si.GetPixelsSize());
        Blob blob;
        hr = SaveToDDSMemory(si.GetImages(), si.GetImageCount(), si.GetMetadata(), DDS_FLAGS(ddsFlags), blob);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        if (blob.GetBufferSize() > capacity) return -2;
        memcpy(out, blob.GetBufferPointer(), blob.GetBufferSize());
        return int64_t(blob.GetBufferSize());
    }

    // format facts (DirectXTexUtil.cpp): bits per pixel, pitches under CP_FLAGS, scanline counts, predicates as a bit set
    // (1 compressed, 2 packed, 4 planar, 8 palettised, 16 sRGB, 32 valid, 64 has alpha)
    int dxtex_ref_format_facts(int fmt, size_t* bpp)
    {
        const DXGI_FORMAT f = DXGI_FORMAT(fmt);
        if (bpp) *bpp = BitsPerPixel(f);
        return (IsCompressed(f) ? 1 : 0) | (IsPacked(f) ? 2 : 0) | (IsPlanar(f) ? 4 : 0) | (IsPalettized(f) ? 8 : 0) | (IsSRGB(f) ? 16 : 0) | (IsValid(f) ? 32 : 0) | (HasAlpha(f) ? 64 : 0);
    }
    int dxtex_ref_compute_pitch_ex(int fmt, size_t w, size_t h, uint32_t cpFlags, size_t* rowPitch, size_t* slicePitch, size_t* scanlines)
    {
        if (scanlines) *scanlines = IsValid(DXGI_FORMAT(fmt)) ? ComputeScanlines(DXGI_FORMAT(fmt), h) : h;       // (asserts on ids past 191)
        return int(ComputePitch(DXGI_FORMAT(fmt), w, h, *rowPitch, *slicePitch, CP_FLAGS(cpFlags)));
    }

    // more format facts: out[0..6] = BitsPerColor, BytesPerBlock, MakeSRGB, MakeLinear, MakeTypeless, MakeTypelessUNORM, MakeTypelessFLOAT;
    // returns predicate bits: 1 video, 2 depth-stencil, 4 BGR, 8 typeless (partial counts), 16 typeless (partial does not count)
    int dxtex_ref_format_facts2(int fmt, size_t* out)
    {
        const DXGI_FORMAT f = DXGI_FORMAT(fmt);
        out[0] = BitsPerColor(f); out[1] = BytesPerBlock(f); out[2] = size_t(MakeSRGB(f)); out[3] = size_t(MakeLinear(f)); out[4] = size_t(MakeTypeless(f));
        out[5] = size_t(MakeTypelessUNORM(f)); out[6] = size_t(MakeTypelessFLOAT(f));
        return (IsVideo(f) ? 1 : 0) | (IsDepthStencil(f) ? 2 : 0) | (IsBGR(f) ? 4 : 0) | (IsTypeless(f, true) ? 8 : 0) | (IsTypeless(f, false) ? 16 : 0);
    }

    // ---- Radiance .hdr (DirectXTexHDR.cpp compiled in place) --------------------------------------------------------------------
    // meta[0..3] = width, height, format, miscFlags2; returns the pixel bytes (RGBA32F, tight), -1 on failure
    int64_t dxtex_ref_load_hdr(const uint8_t* file, size_t size, uint64_t* meta, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si; TexMetadata m = {};
        const HRESULT hr = LoadFromHDRMemory(file, size, &m, si);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        meta[0] = m.width; meta[1] = m.height; meta[2] = uint64_t(m.format); meta[3] = m.miscFlags2;
        if (si.GetPixelsSize() > capacity) return -2;
        memcpy(out, si.GetPixels(), si.GetPixelsSize());
        return int64_t(si.GetPixelsSize());
    }
    int64_t dxtex_ref_save_hdr(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        Image img = {};
        img.width = w; img.height = h; img.format = DXGI_FORMAT(fmt); img.rowPitch = rowPitch; img.slicePitch = rowPitch * h; img.pixels = const_cast<uint8_t*>(pixels);
        Blob blob;
        const HRESULT hr = SaveToHDRMemory(img, blob);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        if (blob.GetBufferSize() > capacity) return -2;
        memcpy(out, blob.GetBufferPointer(), blob.GetBufferSize());
        return int64_t(blob.GetBufferSize());
    }

    // ---- TGA (DirectXTexTGA.cpp compiled in place) ---------------------------------------------------------------------------------
    int64_t dxtex_ref_load_tga(const uint8_t* file, size_t size, uint32_t flags, uint64_t* meta, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        ScratchImage si; TexMetadata m = {};
        const HRESULT hr = LoadFromTGAMemory(file, size, TGA_FLAGS(flags), &m, si);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        meta[0] = m.width; meta[1] = m.height; meta[2] = uint64_t(m.format); meta[3] = m.miscFlags2;
        meta[4] = uint64_t(si.GetMetadata().format);           // the image's own label after OverrideFormat
        TexMetadata q = {};
        meta[5] = uint64_t(uint32_t(GetMetadataFromTGAMemory(file, size, TGA_FLAGS(flags), q)));
        meta[6] = uint64_t(q.format); meta[7] = q.miscFlags2;
        if (si.GetPixelsSize() > capacity) return -2;
        memcpy(out, si.GetPixels(), si.GetPixelsSize());
        return int64_t(si.GetPixelsSize());
    }
    // alphaMode < 0: no metadata (no extension area)
    int64_t dxtex_ref_save_tga(const uint8_t* pixels, size_t w, size_t h, int fmt, size_t rowPitch, uint32_t flags, int alphaMode, uint8_t* out, size_t capacity, int32_t* hrOut)
    {
        Image img = {};
        img.width = w; img.height = h; img.format = DXGI_FORMAT(fmt); img.rowPitch = rowPitch; img.slicePitch = rowPitch * h; img.pixels = const_cast<uint8_t*>(pixels);
        TexMetadata m = {};
        m.width = w; m.height = h; m.depth = m.arraySize = m.mipLevels = 1; m.format = DXGI_FORMAT(fmt); m.dimension = TEX_DIMENSION_TEXTURE2D;
        if (alphaMode >= 0) m.SetAlphaMode(TEX_ALPHA_MODE(alphaMode));
        Blob blob;
        const HRESULT hr = SaveToTGAMemory(img, TGA_FLAGS(flags), blob, alphaMode >= 0 ? &m : nullptr);
        if (hrOut) *hrOut = int32_t(hr);
        if (FAILED(hr)) return -1;
        if (blob.GetBufferSize() > capacity) return -2;
        memcpy(out, blob.GetBufferPointer(), blob.GetBufferSize());
        return int64_t(blob.GetBufferSize());
    }

    int dxtex_ref_tile_shape(int fmt, uint32_t dimension, size_t* whd)
    {
        TileShape t = {};
        const HRESULT hr = ComputeTileShape(DXGI_FORMAT(fmt), TEX_DIMENSION(dimension), t);
        whd[0] = t.width; whd[1] = t.height; whd[2] = t.depth;
        return int(hr);
    }

    // The reference's CreateTriangleFilter (filters.h:249-419) flattened: for every source texel, in order, its (destination, weight)
    // entries. Returns the number of entries (-1 on failure, -2 if the arrays are too small).
    int64_t dxtex_ref_triangle_filter(size_t source, size_t dest, int wrap, uint32_t* srcOf, uint32_t* dstOf, float* weightOf, size_t capacity)
    {
        std::unique_ptr<Filters::Filter> tf;
        if (FAILED(Filters::CreateTriangleFilter(source, dest, wrap != 0, tf)) || !tf) return -1;
        size_t n = 0, u = 0;
        const Filters::FilterFrom* pFrom = tf->from;
        auto pEnd = reinterpret_cast<const Filters::FilterFrom*>(reinterpret_cast<const uint8_t*>(tf.get()) + tf->sizeInBytes);
        for (; pFrom < pEnd; ++u)
        {
            for (size_t j = 0; j < pFrom->count; ++j, ++n)
            {
                if (n >= capacity) return -2;
                srcOf[n] = uint32_t(u); dstOf[n] = uint32_t(pFrom->to[j].u); weightOf[n] = pFrom->to[j].weight;
            }
            pFrom = reinterpret_cast<const Filters::FilterFrom*>(reinterpret_cast<const uint8_t*>(pFrom) + pFrom->sizeInBytes);
        }
        return (u == source) ? int64_t(n) : -1;
    }
}
