// Exercises the C++ host layer (directxtex_amd/host/DirectXTexAMD.h) the way a DirectXTex user would.
//   host_api_test cpu            - container / pitch / index rules only, no GPU needed
//   host_api_test gpu <outdir>   - Compress, Decompress, GenerateMipMaps, Resize, Convert, ComputeMSE on device 0;
//                                  writes the results to <outdir> for tests/test_host_api.py to compare with the oracle
#include "../../directxtex_amd/host/DirectXTexAMD.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

using namespace DirectXTexAMD;

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static void dump(const std::string& path, const void* p, size_t n)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (f) { std::fwrite(p, 1, n, f); std::fclose(f); }
}

static int cpu_checks()
{
    size_t rp = 0, sp = 0;
    CHECK(ComputePitch(DXGI_FORMAT_BC1_UNORM, 256, 256, rp, sp) == S_OK && rp == 512 && sp == 512 * 64);
    CHECK(ComputePitch(DXGI_FORMAT_BC7_UNORM, 5, 7, rp, sp) == S_OK && rp == 32 && sp == 64);
    CHECK(ComputePitch(DXGI_FORMAT_R8G8B8A8_UNORM, 13, 3, rp, sp) == S_OK && rp == 52 && sp == 156);
    size_t lv = 0;
    CHECK(CalculateMipLevels(8192, 8192, lv) && lv == 14);
    lv = 20; CHECK(!CalculateMipLevels(64, 64, lv));
    lv = 3; CHECK(CalculateMipLevels(64, 1, lv) && lv == 3);

    ScratchImage si;
    CHECK(si.Initialize2D(DXGI_FORMAT_R8G8B8A8_UNORM, 20, 10, 3, 0) == S_OK);
    CHECK(si.GetMetadata().mipLevels == 5 && si.GetImageCount() == 15);
    // item-major, then mip; tightly packed; zero filled; 16-byte aligned
    CHECK((reinterpret_cast<uintptr_t>(si.GetPixels()) & 15) == 0);
    const Image* a = si.GetImage(0, 0, 0); const Image* b = si.GetImage(1, 0, 0); const Image* c = si.GetImage(0, 1, 0);
    CHECK(a && b && c && a->pixels == si.GetPixels() && b->pixels == a->pixels + a->slicePitch);
    CHECK(b->width == 10 && b->height == 5 && si.GetImage(4, 2, 0)->width == 1 && si.GetImage(4, 2, 0)->height == 1);
    size_t chain = 0; for (size_t l = 0; l < 5; ++l) chain += si.GetImage(l, 0, 0)->slicePitch;
    CHECK(c->pixels == a->pixels + chain && si.GetPixelsSize() == chain * 3);
    for (size_t i = 0; i < si.GetPixelsSize(); ++i) CHECK(si.GetPixels()[i] == 0);
    CHECK(si.GetImage(5, 0, 0) == nullptr && si.GetImage(0, 3, 0) == nullptr && si.GetImage(0, 0, 1) == nullptr);
    CHECK(si.GetMetadata().ComputeIndex(2, 1, 0) == 7);
    CHECK(si.Initialize2D(DXGI_FORMAT_UNKNOWN, 4, 4, 1, 1) == E_INVALIDARG);
    CHECK(si.Initialize2D(DXGI_FORMAT_BC3_UNORM, 0, 4, 1, 1) == E_INVALIDARG);
    CHECK(si.Initialize2D(DXGI_FORMAT_BC3_UNORM, 9, 9, 1, 0) == S_OK && si.GetMetadata().mipLevels == 4 && si.GetImage(3, 0, 0)->slicePitch == 16);

    // volumes: level by level, the level's slices consecutive (DirectXTexImage.cpp:228-262, DirectXTexUtil.cpp:1714-1738)
    ScratchImage v3;
    CHECK(v3.Initialize3D(DXGI_FORMAT_R8G8B8A8_UNORM, 16, 8, 4, 0) == S_OK);
    CHECK(v3.GetMetadata().mipLevels == 5 && v3.GetImageCount() == 4 + 2 + 1 + 1 + 1 && v3.GetMetadata().dimension == TEX_DIMENSION_TEXTURE3D);
    CHECK(v3.GetImage(0, 0, 3)->pixels == v3.GetPixels() + 3 * 16 * 8 * 4 && v3.GetImage(1, 0, 0)->pixels == v3.GetPixels() + 4 * 16 * 8 * 4);
    CHECK(v3.GetImage(1, 0, 1)->width == 8 && v3.GetImage(1, 0, 2) == nullptr && v3.GetImage(4, 0, 0)->width == 1 && v3.GetImage(0, 1, 0) == nullptr);
    CHECK(v3.GetMetadata().ComputeIndex(2, 0, 0) == 6 && v3.GetMetadata().ComputeIndex(3, 0, 0) == 7);
    {
        TexMetadata arr; arr.width = 20; arr.height = 10; arr.depth = 1; arr.arraySize = 3; arr.mipLevels = 5; arr.format = DXGI_FORMAT_R8G8B8A8_UNORM;
        CHECK(arr.CalculateSubresource(2, 1) == 7 && arr.CalculateSubresource(2, 1, 1) == 7 + 15 && arr.CalculateSubresource(5, 0) == uint32_t(-1) && arr.CalculateSubresource(0, 3) == uint32_t(-1));
    }
    CHECK(v3.GetMetadata().CalculateSubresource(3, 0) == 3 && v3.GetMetadata().CalculateSubresource(3, 0, 2) == 13 && v3.GetMetadata().CalculateSubresource(0, 1) == uint32_t(-1));
    lv = 0; CHECK(CalculateMipLevels3D(4, 2, 32, lv) && lv == 6);
    lv = 7; CHECK(!CalculateMipLevels3D(4, 2, 32, lv));
    CHECK(v3.Initialize3D(DXGI_FORMAT_R8G8B8A8_UNORM, 4, 4, 0, 1) == E_INVALIDARG);

    // the container holds any valid DXGI format (GPU work on it is another matter); 1D textures and cubemaps
    {
        ScratchImage any;
        CHECK(any.Initialize2D(DXGI_FORMAT_B5G6R5_UNORM, 7, 3, 1, 1) == S_OK && any.GetImage(0, 0, 0)->rowPitch == 14);
        CHECK(any.Initialize2D(DXGI_FORMAT_R1_UNORM, 9, 2, 1, 1) == S_OK && any.GetImage(0, 0, 0)->rowPitch == 2);
        CHECK(any.Initialize2D(DXGI_FORMAT_P8, 4, 4, 1, 1) == HRESULT_E_NOT_SUPPORTED);
        CHECK(any.Initialize2D(DXGI_FORMAT(150), 4, 4, 1, 1) == E_INVALIDARG && any.Initialize2D(DXGI_FORMAT(192), 4, 4, 1, 1) == E_INVALIDARG);
        CHECK(any.Initialize2D(DXGI_FORMAT_R8G8B8A8_UNORM, 5, 3, 1, 1, CP_FLAGS_PARAGRAPH) == S_OK && any.GetImage(0, 0, 0)->rowPitch == 32);
        CHECK(any.Initialize1D(DXGI_FORMAT_R8_UNORM, 16, 2, 0) == S_OK && any.GetMetadata().dimension == TEX_DIMENSION_TEXTURE1D && any.GetImageCount() == 10);
        CHECK(any.InitializeCube(DXGI_FORMAT_R8G8B8A8_UNORM, 4, 4, 2, 1) == S_OK && any.GetMetadata().IsCubemap() && any.GetImageCount() == 12);
        CHECK(any.InitializeCube(DXGI_FORMAT_R8G8B8A8_UNORM, 4, 4, 0, 1) == E_INVALIDARG);
        TexMetadata odd = any.GetMetadata(); odd.arraySize = 7;
        CHECK(any.Initialize(odd) == E_INVALIDARG);                     // a cubemap needs a multiple of six
        CHECK(!IsSupportedOnDevice(DXGI_FORMAT_A8P8) && IsSupportedOnDevice(DXGI_FORMAT_D16_UNORM) && IsSupportedOnDevice(DXGI_FORMAT_R16_UINT) && IsSupportedOnDevice(DXGI_FORMAT_B5G6R5_UNORM) && IsSupportedOnDevice(DXGI_FORMAT_R8G8B8A8_UNORM) && IsSupportedOnDevice(DXGI_FORMAT_BC7_UNORM));
    }
    // copies of caller images: array, cubemap, volume, relabelling (DirectXTexImage.cpp:534-740)
    {
        std::vector<uint8_t> texels(6 * 4 * 4 * 4);
        for (size_t i = 0; i < texels.size(); ++i) texels[i] = uint8_t(i * 13 + 5);
        std::vector<Image> six(6);
        for (size_t i = 0; i < 6; ++i) { six[i].width = 4; six[i].height = 4; six[i].format = DXGI_FORMAT_R8G8B8A8_UNORM; six[i].rowPitch = 16; six[i].slicePitch = 64; six[i].pixels = texels.data() + i * 64; }
        ScratchImage s;
        CHECK(s.InitializeArrayFromImages(six.data(), 6) == S_OK && s.GetMetadata().arraySize == 6 && !s.GetMetadata().IsCubemap() && std::memcmp(s.GetPixels(), texels.data(), texels.size()) == 0);
        CHECK(s.InitializeCubeFromImages(six.data(), 6) == S_OK && s.GetMetadata().IsCubemap() && s.GetImage(0, 5, 0)->pixels[0] == texels[5 * 64]);
        CHECK(s.InitializeCubeFromImages(six.data(), 5) == E_INVALIDARG);
        CHECK(s.Initialize3DFromImages(six.data(), 4) == S_OK && s.GetMetadata().depth == 4 && s.GetMetadata().dimension == TEX_DIMENSION_TEXTURE3D && s.GetImage(0, 0, 3)->pixels[1] == texels[3 * 64 + 1]);
        six[2].width = 3;
        CHECK(s.InitializeArrayFromImages(six.data(), 6) == E_FAIL);
        six[2].width = 4; six[1].pixels = nullptr;
        CHECK(s.Initialize3DFromImages(six.data(), 4) == E_POINTER);
        Image line = six[0]; line.width = 16; line.height = 1; line.rowPitch = 64;
        CHECK(s.InitializeFromImage(line, true) == S_OK && s.GetMetadata().dimension == TEX_DIMENSION_TEXTURE1D);
        CHECK(s.InitializeFromImage(line) == S_OK && s.GetMetadata().dimension == TEX_DIMENSION_TEXTURE2D);
        CHECK(s.OverrideFormat(DXGI_FORMAT_R8G8B8A8_UNORM_SRGB) && s.GetMetadata().format == DXGI_FORMAT_R8G8B8A8_UNORM_SRGB && s.GetImage(0, 0, 0)->format == DXGI_FORMAT_R8G8B8A8_UNORM_SRGB);
        CHECK(!s.OverrideFormat(DXGI_FORMAT_NV12) && !s.OverrideFormat(DXGI_FORMAT_P8) && !s.OverrideFormat(DXGI_FORMAT_UNKNOWN));
        ScratchImage empty;
        CHECK(!empty.OverrideFormat(DXGI_FORMAT_R8G8B8A8_UNORM));
    }
    // DDS: images with padded rows are written with the file's tight pitch; the one-image overloads; the header query
    {
        ScratchImage tight;
        CHECK(tight.Initialize2D(DXGI_FORMAT_R8G8B8A8_UNORM, 5, 3, 1, 1) == S_OK);
        for (size_t i = 0; i < tight.GetPixelsSize(); ++i) tight.GetPixels()[i] = uint8_t(i * 7 + 1);
        std::vector<uint8_t> padded(3 * 32, 0xEE);
        Image wide = *tight.GetImage(0, 0, 0); wide.rowPitch = 32; wide.slicePitch = 96; wide.pixels = padded.data();
        for (size_t y = 0; y < 3; ++y) std::memcpy(padded.data() + y * 32, tight.GetPixels() + y * 20, 20);
        Blob b1, b2;
        CHECK(SaveToDDSMemory(*tight.GetImage(0, 0, 0), DDS_FLAGS_NONE, b1) == S_OK && SaveToDDSMemory(wide, DDS_FLAGS_NONE, b2) == S_OK);
        CHECK(b1.GetBufferSize() == 128 + 60 && b2.GetBufferSize() == b1.GetBufferSize() && std::memcmp(b1.GetBufferPointer(), b2.GetBufferPointer(), b1.GetBufferSize()) == 0);
        size_t need = 0;
        CHECK(EncodeDDSHeader(tight.GetMetadata(), DDS_FLAGS_FORCE_DX10_EXT, nullptr, 0, need) == S_OK && need == 148);
        uint8_t small[64];
        CHECK(EncodeDDSHeader(tight.GetMetadata(), DDS_FLAGS_NONE, small, sizeof(small), need) == HRESULT(0x8007007A));
        TexMetadata q; DDSMetaData pf;
        CHECK(GetMetadataFromDDSMemoryEx(b1.GetBufferPointer(), b1.GetBufferSize(), DDS_FLAGS_NONE, q, &pf) == S_OK && q.width == 5 && q.format == DXGI_FORMAT_R8G8B8A8_UNORM);
        CHECK(pf.size == 32 && pf.flags == 0x41 && pf.RGBBitCount == 32 && pf.RBitMask == 0xff && pf.ABitMask == 0xff000000u);
        CHECK(GetMetadataFromDDSMemory(nullptr, 10, DDS_FLAGS_NONE, q) == E_INVALIDARG && GetMetadataFromDDSMemory(b1.GetBufferPointer(), 0, DDS_FLAGS_NONE, q) == E_INVALIDARG);
        CHECK(GetMetadataFromDDSFile("/nonexistent/x.dds", DDS_FLAGS_NONE, q) == E_FAIL && LoadFromDDSFile(nullptr, DDS_FLAGS_NONE, nullptr, tight) == E_INVALIDARG);
    }

    // without a device every entry point refuses to work: there is no CPU path
    Device none;
    ScratchImage out;
    Image img = *si.GetImage(0, 0, 0);
    CHECK(Compress(none, img, DXGI_FORMAT_BC1_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, out) == E_POINTER);
    CHECK(none.Prepare(64, 64, DXGI_FORMAT_R8G8B8A8_UNORM, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, 1) == E_POINTER);
    {
        size_t calls = 0;
        const CompressOptions co = { TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, TEX_ALPHA_WEIGHT_DEFAULT };
        const ConvertOptions cv = { TEX_FILTER_DEFAULT, TEX_THRESHOLD_DEFAULT };
        auto count = [&](size_t, size_t) { ++calls; return true; };
        CHECK(CompressEx(none, img, DXGI_FORMAT_BC1_UNORM, co, out, count) == E_POINTER && calls == 0);
        CHECK(ConvertEx(none, img, DXGI_FORMAT_R16G16B16A16_FLOAT, cv, out, count) == E_POINTER && calls == 0);
    }
    std::puts("cpu checks OK");
    return 0;
}

static int gpu_run(const std::string& outdir)
{
    Device dev;
    if (FAILED(dev.Create(0))) { std::fprintf(stderr, "no gfx950 device\n"); return 2; }

    const size_t W = 96, H = 64;
    std::vector<uint8_t> px(W * H * 4);
    uint32_t s = 12345;
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x)
        {
            s = s * 1664525u + 1013904223u;
            uint8_t* p = &px[(y * W + x) * 4];
            p[0] = uint8_t(x * 2 + ((s >> 24) & 15)); p[1] = uint8_t(y * 3 + ((s >> 20) & 15)); p[2] = uint8_t((x + y) + ((s >> 16) & 31)); p[3] = uint8_t(255 - ((s >> 8) & 63));
        }
    Image src; src.width = W; src.height = H; src.format = DXGI_FORMAT_R8G8B8A8_UNORM; src.rowPitch = W * 4; src.slicePitch = W * H * 4; src.pixels = px.data();
    dump(outdir + "/src.bin", px.data(), px.size());

    ScratchImage bc7, bc3, back, mips, resized, conv;
    CHECK(Compress(dev, src, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, bc7) == S_OK);
    dump(outdir + "/bc7.bin", bc7.GetPixels(), bc7.GetPixelsSize());
    CHECK(Decompress(dev, *bc7.GetImage(0, 0, 0), DXGI_FORMAT_UNKNOWN, back) == S_OK);
    CHECK(back.GetMetadata().format == DXGI_FORMAT_R8G8B8A8_UNORM);
    dump(outdir + "/bc7_decoded.bin", back.GetPixels(), back.GetPixelsSize());

    CHECK(GenerateMipMaps(dev, src, TEX_FILTER_CUBIC, 0, mips) == S_OK);
    CHECK(mips.GetMetadata().mipLevels == 7 && mips.GetImageCount() == 7);
    dump(outdir + "/mips_cubic.bin", mips.GetPixels(), mips.GetPixelsSize());
    // mip chain -> BC3, the array overload
    CHECK(Compress(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_BC3_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bc3) == S_OK);
    CHECK(bc3.GetImageCount() == 7 && bc3.GetImage(6, 0, 0)->slicePitch == 16);
    dump(outdir + "/mips_bc3.bin", bc3.GetPixels(), bc3.GetPixelsSize());
    ScratchImage bc7chain;     // BC7 arrays run through the search pipeline as one block list
    CHECK(Compress(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bc7chain) == S_OK);
    dump(outdir + "/mips_bc7.bin", bc7chain.GetPixels(), bc7chain.GetPixelsSize());

    CHECK(Resize(dev, src, 50, 70, TEX_FILTER_TRIANGLE, resized) == S_OK);
    dump(outdir + "/resized_triangle.bin", resized.GetPixels(), resized.GetPixelsSize());
    CHECK(Convert(dev, src, DXGI_FORMAT_R16G16B16A16_FLOAT, TEX_FILTER_DEFAULT, 0.5f, conv) == S_OK);
    dump(outdir + "/converted_f16.bin", conv.GetPixels(), conv.GetPixelsSize());

    // array overloads: the whole mip chain converted, every chain's top level resized
    ScratchImage convAll, resizedAll;
    CHECK(Convert(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_B8G8R8A8_UNORM, TEX_FILTER_DEFAULT, 0.5f, convAll) == S_OK);
    CHECK(convAll.GetImageCount() == 7 && convAll.GetMetadata().format == DXGI_FORMAT_B8G8R8A8_UNORM);
    dump(outdir + "/mips_bgra.bin", convAll.GetPixels(), convAll.GetPixelsSize());
    CHECK(Resize(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), 40, 24, TEX_FILTER_LINEAR, resizedAll) == S_OK);
    CHECK(resizedAll.GetImageCount() == 1 && resizedAll.GetMetadata().mipLevels == 1 && resizedAll.GetMetadata().width == 40);
    dump(outdir + "/resized_linear_array.bin", resizedAll.GetPixels(), resizedAll.GetPixelsSize());

    // one image over three devices (three contexts of this GPU): the single-device bytes (dxtex_compress_multi / dxtex_generate_mips_multi)
    {
        Device d1, d2;
        CHECK(SUCCEEDED(d1.Create(0)) && SUCCEEDED(d2.Create(0)));
        Device* const devs[3] = { &dev, &d1, &d2 };
        ScratchImage multi;
        CHECK(Compress(devs, 3, src, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, multi) == S_OK);
        CHECK(multi.GetPixelsSize() == bc7.GetPixelsSize() && !std::memcmp(multi.GetPixels(), bc7.GetPixels(), bc7.GetPixelsSize()));
        const size_t BW = 512, BH = 1024;
        std::vector<uint8_t> big(BW * BH * 4);
        for (size_t i = 0; i < big.size(); ++i) { s = s * 1664525u + 1013904223u; big[i] = uint8_t((i >> 7) + ((s >> 24) & 31)); }
        Image bsrc; bsrc.width = BW; bsrc.height = BH; bsrc.format = DXGI_FORMAT_R8G8B8A8_UNORM; bsrc.rowPitch = BW * 4; bsrc.slicePitch = big.size(); bsrc.pixels = big.data();
        for (TEX_FILTER_FLAGS f : { TEX_FILTER_CUBIC, TEX_FILTER_BOX })
        {
            ScratchImage one, three;
            CHECK(GenerateMipMaps(dev, bsrc, f, 0, one) == S_OK);
            CHECK(GenerateMipMaps(devs, 3, bsrc, f, 0, three) == S_OK);
            CHECK(one.GetPixelsSize() == three.GetPixelsSize() && !std::memcmp(one.GetPixels(), three.GetPixels(), one.GetPixelsSize()));
        }
        CHECK(Compress(devs, 0, src, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, multi) == E_POINTER);
        std::printf("multi-device OK\n");
    }

    // premultiplied alpha and alpha-to-coverage preserving mips
    ScratchImage pm, cov;
    CHECK(PremultiplyAlpha(dev, src, TEX_PMALPHA_DEFAULT, pm) == S_OK);
    dump(outdir + "/premultiplied.bin", pm.GetPixels(), pm.GetPixelsSize());
    CHECK(cov.Initialize2D(src.format, W, H, 1, mips.GetMetadata().mipLevels) == S_OK);
    CHECK(ScaleMipMapsAlphaForCoverage(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), 0, 0.6f, cov) == S_OK);
    dump(outdir + "/mips_coverage.bin", cov.GetPixels(), cov.GetPixelsSize());
    {
        ScratchImage r8; Image one = src; one.format = DXGI_FORMAT_R8_UNORM; one.rowPitch = W;
        CHECK(PremultiplyAlpha(dev, one, TEX_PMALPHA_DEFAULT, r8) == HRESULT_E_NOT_SUPPORTED);
    }

    // a volume texture: 8 slices cut from the image, GenerateMipMaps3D, DDS round trip
    {
        const size_t VW = 32, VH = 16, VD = 8;
        std::vector<Image> slices(VD);
        for (size_t z = 0; z < VD; ++z)
        {
            slices[z] = src; slices[z].width = VW; slices[z].height = VH;
            slices[z].pixels = px.data() + (z * 4) * src.rowPitch + (z * 8) * 4;           // a shifted window per slice
        }
        ScratchImage vol, volBack;
        CHECK(GenerateMipMaps3D(dev, slices.data(), VD, TEX_FILTER_CUBIC, 0, vol) == S_OK);
        CHECK(vol.GetMetadata().dimension == TEX_DIMENSION_TEXTURE3D && vol.GetMetadata().mipLevels == 6 && vol.GetImageCount() == 8 + 4 + 2 + 1 + 1 + 1);
        CHECK(vol.GetImage(1, 0, 3) != nullptr && vol.GetImage(1, 0, 4) == nullptr && vol.GetImage(5, 0, 0)->width == 1);
        dump(outdir + "/volume_mips.bin", vol.GetPixels(), vol.GetPixelsSize());
        const std::string f = outdir + "/volume.dds";
        CHECK(SaveToDDSFile(vol.GetImages(), vol.GetImageCount(), vol.GetMetadata(), DDS_FLAGS_NONE, f.c_str()) == S_OK);
        TexMetadata vm;
        CHECK(LoadFromDDSFile(f.c_str(), DDS_FLAGS_NONE, &vm, volBack) == S_OK);
        CHECK(vm.dimension == TEX_DIMENSION_TEXTURE3D && vm.depth == VD && vm.mipLevels == 6 && volBack.GetPixelsSize() == vol.GetPixelsSize());
        CHECK(std::memcmp(volBack.GetPixels(), vol.GetPixels(), vol.GetPixelsSize()) == 0);
        ScratchImage bad3;
        CHECK(GenerateMipMaps3D(dev, slices.data(), 7, TEX_FILTER_BOX, 0, bad3) == E_FAIL);          // 7 slices: not a power of two
    }

    // CompressEx / ConvertEx: progress and cancel. One image goes to the GPU in bands of rows (forced small here), a set
    // image by image; either way the bytes are those of the callback-free call (DirectXTexCompress.cpp:664-850).
    {
        dev.SetProgressBands(100, 1000);                         // 24 blocks per row -> 4 block rows per band; 96 texels per row -> 10 rows per band
        const CompressOptions co = { TEX_COMPRESS_DEFAULT, TEX_THRESHOLD_DEFAULT, TEX_ALPHA_WEIGHT_DEFAULT };
        const ConvertOptions cv = { TEX_FILTER_DEFAULT, TEX_THRESHOLD_DEFAULT };
        std::vector<std::pair<size_t, size_t>> calls;
        auto record = [&](size_t a, size_t b) { calls.emplace_back(a, b); return true; };
        ScratchImage ex;
        CHECK(CompressEx(dev, src, DXGI_FORMAT_BC7_UNORM, co, ex, record) == S_OK);
        CHECK(ex.GetPixelsSize() == bc7.GetPixelsSize() && std::memcmp(ex.GetPixels(), bc7.GetPixels(), bc7.GetPixelsSize()) == 0);
        CHECK(calls.size() == 5 && calls[0] == std::make_pair(size_t(0), H) && calls[1] == std::make_pair(size_t(16), H)
              && calls[3] == std::make_pair(size_t(48), H) && calls[4] == std::make_pair(H, H));
        // 62 rows: the last band ends in a partial block row
        Image cut = src; cut.height = 62; cut.slicePitch = cut.rowPitch * 62;
        ScratchImage cutA, cutB;
        calls.clear();
        CHECK(Compress(dev, cut, DXGI_FORMAT_BC3_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, cutA) == S_OK);
        CHECK(CompressEx(dev, cut, DXGI_FORMAT_BC3_UNORM, co, cutB, record) == S_OK);
        CHECK(cutA.GetPixelsSize() == cutB.GetPixelsSize() && std::memcmp(cutA.GetPixels(), cutB.GetPixels(), cutA.GetPixelsSize()) == 0);
        CHECK(calls.size() == 5 && calls.back() == std::make_pair(size_t(62), size_t(62)));
        // cancel in the middle: E_ABORT, result released
        size_t n = 0;
        CHECK(CompressEx(dev, src, DXGI_FORMAT_BC7_UNORM, co, ex, [&](size_t, size_t) { return ++n < 3; }) == E_ABORT);
        CHECK(n == 3 && ex.GetPixels() == nullptr && ex.GetImageCount() == 0);
        // a set reports images: (0,n), then (index,n) after each image, then (n,n)
        calls.clear();
        CHECK(CompressEx(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_BC3_UNORM, co, ex, record) == S_OK);
        CHECK(ex.GetPixelsSize() == bc3.GetPixelsSize() && std::memcmp(ex.GetPixels(), bc3.GetPixels(), bc3.GetPixelsSize()) == 0);
        CHECK(calls.size() == 9 && calls[0] == std::make_pair(size_t(0), size_t(7)) && calls[1] == std::make_pair(size_t(0), size_t(7))
              && calls[7] == std::make_pair(size_t(6), size_t(7)) && calls[8] == std::make_pair(size_t(7), size_t(7)));
        n = 0;
        CHECK(CompressEx(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_BC7_UNORM, co, ex, [&](size_t, size_t) { return ++n < 4; }) == E_ABORT);
        CHECK(n == 4 && ex.GetPixels() == nullptr);
        // a set of one plain 2-D image takes the single-image route and reports rows
        calls.clear();
        TexMetadata one; one.width = W; one.height = H; one.depth = 1; one.arraySize = 1; one.mipLevels = 1; one.format = src.format;
        CHECK(CompressEx(dev, &src, 1, one, DXGI_FORMAT_BC7_UNORM, co, ex, record) == S_OK);
        CHECK(calls.size() == 5 && calls[4] == std::make_pair(H, H) && std::memcmp(ex.GetPixels(), bc7.GetPixels(), bc7.GetPixelsSize()) == 0);

        calls.clear();
        CHECK(ConvertEx(dev, src, DXGI_FORMAT_R16G16B16A16_FLOAT, cv, ex, record) == S_OK);
        CHECK(ex.GetPixelsSize() == conv.GetPixelsSize() && std::memcmp(ex.GetPixels(), conv.GetPixels(), conv.GetPixelsSize()) == 0);
        CHECK(calls.size() == 8 && calls[1] == std::make_pair(size_t(10), H) && calls[6] == std::make_pair(size_t(60), H) && calls[7] == std::make_pair(H, H));
        calls.clear();
        CHECK(ConvertEx(dev, mips.GetImages(), mips.GetImageCount(), mips.GetMetadata(), DXGI_FORMAT_B8G8R8A8_UNORM, cv, ex, record) == S_OK);
        CHECK(ex.GetPixelsSize() == convAll.GetPixelsSize() && std::memcmp(ex.GetPixels(), convAll.GetPixels(), convAll.GetPixelsSize()) == 0);
        CHECK(calls.size() == 9 && calls[8] == std::make_pair(size_t(7), size_t(7)));
        n = 0;
        CHECK(ConvertEx(dev, src, DXGI_FORMAT_R16G16B16A16_FLOAT, cv, ex, [&](size_t, size_t) { return ++n < 2; }) == E_ABORT && ex.GetPixels() == nullptr);
        dev.SetProgressBands(0, 0);
        // default band sizes: a 96x64 image is one band - start and end only
        calls.clear();
        CHECK(CompressEx(dev, src, DXGI_FORMAT_BC1_UNORM, co, ex, record) == S_OK && calls.size() == 2);
    }

    // ---- the device-resident pipeline: the same steps on DeviceScratchImages, one upload, one download, identical bytes ----------------
    {
        auto same = [](const ScratchImage& a, const ScratchImage& b) { return a.GetPixelsSize() == b.GetPixelsSize() && a.GetImageCount() == b.GetImageCount() &&
                                                                          std::memcmp(a.GetPixels(), b.GetPixels(), a.GetPixelsSize()) == 0; };
        ScratchImage srcS, got;
        CHECK(srcS.InitializeFromImage(src) == S_OK);
        uint64_t up = 0, down = 0;
        GetTransferBytes(dev, up, down, true);
        DeviceScratchImage dsrc, dmips, dbc3, dbc7, dback, dresized, dconv, dconvAll, dresizedAll, dpm, dcov, dtop;
        CHECK(dsrc.Upload(dev, srcS) == S_OK);
        CHECK(dsrc.GetImageCount() == 1 && dsrc.GetMetadata().width == W && dsrc.GetImage(0, 0, 0)->rowPitch == W * 4 && dsrc.GetDevice() == &dev);
        // resize -> convert -> mipmaps -> compress, texconv's order, without leaving the device
        CHECK(GenerateMipMaps(dev, dsrc, TEX_FILTER_CUBIC, 0, dmips) == S_OK && dmips.GetMetadata().mipLevels == 7 && dmips.GetImageCount() == 7);
        CHECK(Compress(dev, dmips, DXGI_FORMAT_BC3_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dbc3) == S_OK);
        GetTransferBytes(dev, up, down);
        CHECK(up == srcS.GetPixelsSize() && down == 0);                       // nothing but the source has crossed PCIe so far
        CHECK(dbc3.Download(got) == S_OK && same(got, bc3));
        GetTransferBytes(dev, up, down);
        CHECK(up == srcS.GetPixelsSize() && down == bc3.GetPixelsSize());     // one upload, one download
        CHECK(dmips.Download(got) == S_OK && same(got, mips));
        CHECK(Compress(dev, dmips, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dbc7) == S_OK && dbc7.Download(got) == S_OK && same(got, bc7chain));
        CHECK(Compress(dev, dsrc, DXGI_FORMAT_BC7_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dbc7) == S_OK && dbc7.Download(got) == S_OK && same(got, bc7));
        CHECK(Decompress(dev, dbc7, DXGI_FORMAT_UNKNOWN, dback) == S_OK && dback.GetMetadata().format == DXGI_FORMAT_R8G8B8A8_UNORM && dback.Download(got) == S_OK && same(got, back));
        CHECK(Resize(dev, dsrc, 50, 70, TEX_FILTER_TRIANGLE, dresized) == S_OK && dresized.Download(got) == S_OK && same(got, resized));
        CHECK(Convert(dev, dsrc, DXGI_FORMAT_R16G16B16A16_FLOAT, TEX_FILTER_DEFAULT, 0.5f, dconv) == S_OK && dconv.Download(got) == S_OK && same(got, conv));
        CHECK(Convert(dev, dmips, DXGI_FORMAT_B8G8R8A8_UNORM, TEX_FILTER_DEFAULT, 0.5f, dconvAll) == S_OK && dconvAll.Download(got) == S_OK && same(got, convAll));
        CHECK(Resize(dev, dmips, 40, 24, TEX_FILTER_LINEAR, dresizedAll) == S_OK && dresizedAll.GetMetadata().mipLevels == 1 && dresizedAll.Download(got) == S_OK && same(got, resizedAll));
        CHECK(PremultiplyAlpha(dev, dsrc, TEX_PMALPHA_DEFAULT, dpm) == S_OK && dpm.GetMetadata().IsPMAlpha() && dpm.Download(got) == S_OK && same(got, pm));
        CHECK(ScaleMipMapsAlphaForCoverage(dev, dmips, 0.6f, dcov) == S_OK && dcov.Download(got) == S_OK && same(got, cov));
        CHECK(CopyTopLevels(dev, dmips, dtop) == S_OK && dtop.GetMetadata().mipLevels == 1 && dtop.Download(got) == S_OK && same(got, srcS));
        // alpha scan: this image has alpha 192 .. 255; an opaque copy; the BC forms (decoded on the device, threshold 0.99)
        CHECK(!IsAlphaAllOpaque(dev, dsrc) && !IsAlphaAllOpaque(dev, dbc3));
        {
            std::vector<uint8_t> opq(px);
            for (size_t i = 3; i < opq.size(); i += 4) opq[i] = (i % 8 == 3) ? 255 : 254;           // 254 / 255 = 0.9961 < 0.997: not opaque as RGBA8 ...
            Image o = src; o.pixels = opq.data();
            ScratchImage oS; DeviceScratchImage dO, dOB;
            CHECK(oS.InitializeFromImage(o) == S_OK && dO.Upload(dev, oS) == S_OK && !IsAlphaAllOpaque(dev, dO));
            CHECK(Compress(dev, dO, DXGI_FORMAT_BC3_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dOB) == S_OK && IsAlphaAllOpaque(dev, dOB));    // ... but >= 0.99 once block-compressed
            for (size_t i = 3; i < opq.size(); i += 4) opq[i] = 255;
            CHECK(dO.Upload(dev, &o, 1, oS.GetMetadata()) == S_OK && IsAlphaAllOpaque(dev, dO));
            CHECK(Compress(dev, dO, DXGI_FORMAT_BC1_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dOB) == S_OK && IsAlphaAllOpaque(dev, dOB));
            CHECK(Compress(dev, dO, DXGI_FORMAT_BC5_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dOB) == S_OK && IsAlphaAllOpaque(dev, dOB));    // no alpha channel: opaque
        }
        // a padded source uploaded image by image
        {
            std::vector<uint8_t> wide(H * (W * 4 + 32), 0xEE);
            for (size_t y = 0; y < H; ++y) std::memcpy(&wide[y * (W * 4 + 32)], &px[y * W * 4], W * 4);
            Image wsrc = src; wsrc.rowPitch = W * 4 + 32; wsrc.slicePitch = wide.size(); wsrc.pixels = wide.data();
            DeviceScratchImage dw;
            CHECK(dw.Upload(dev, &wsrc, 1, srcS.GetMetadata()) == S_OK && dw.Download(got) == S_OK && same(got, srcS));
        }
        // a volume
        {
            const size_t VW = 32, VH = 16, VD = 8;
            std::vector<Image> slices(VD);
            for (size_t z = 0; z < VD; ++z) { slices[z] = src; slices[z].width = VW; slices[z].height = VH; slices[z].pixels = px.data() + (z * 4) * src.rowPitch + (z * 8) * 4; }
            ScratchImage vbase, vol;
            CHECK(vbase.Initialize3DFromImages(slices.data(), VD) == S_OK && GenerateMipMaps3D(dev, slices.data(), VD, TEX_FILTER_CUBIC, 0, vol) == S_OK);
            DeviceScratchImage dv, dvm;
            CHECK(dv.Upload(dev, vbase) == S_OK && GenerateMipMaps3D(dev, dv, TEX_FILTER_CUBIC, 0, dvm) == S_OK && dvm.Download(got) == S_OK && same(got, vol));
        }
        // error behaviour matches the host overloads; a failed step leaves its output released
        DeviceScratchImage dbad, dnone;
        CHECK(Compress(dev, dbc7, DXGI_FORMAT_BC1_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dbad) == E_INVALIDARG);
        CHECK(Compress(dev, dsrc, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, dbad) == E_INVALIDARG);
        CHECK(Convert(dev, dsrc, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_FILTER_DEFAULT, 0.5f, dbad) == E_INVALIDARG);
        CHECK(GenerateMipMaps(dev, dsrc, TEX_FILTER_BOX, 0, dbad) == E_FAIL && dbad.GetPixels() == nullptr);      // 96 is not a power of two
        CHECK(Resize(dev, dnone, 8, 8, TEX_FILTER_LINEAR, dbad) == E_INVALIDARG);                                   // nothing uploaded
        CHECK(dnone.Download(got) == E_POINTER);
        Device other;                                                                                               // an image belongs to its Device
        CHECK(Resize(other, dsrc, 8, 8, TEX_FILTER_LINEAR, dbad) == E_INVALIDARG);
        std::puts("resident pipeline OK");
    }

    float mse = 0, v[4];
    CHECK(ComputeMSE(dev, src, *bc7.GetImage(0, 0, 0), mse, v) == S_OK);
    std::printf("mse %.9g %.9g %.9g %.9g %.9g\n", mse, v[0], v[1], v[2], v[3]);

    // error behaviour
    ScratchImage bad;
    CHECK(Compress(dev, *bc7.GetImage(0, 0, 0), DXGI_FORMAT_BC1_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bad) == E_INVALIDARG);
    CHECK(Compress(dev, src, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, bad) == E_INVALIDARG);
    CHECK(Convert(dev, src, DXGI_FORMAT_R8G8B8A8_UNORM, TEX_FILTER_DEFAULT, 0.5f, bad) == E_INVALIDARG);
    CHECK(Resize(dev, src, 50, 32, TEX_FILTER_BOX, bad) == E_FAIL && bad.GetPixels() == nullptr);
    CHECK(GenerateMipMaps(dev, src, TEX_FILTER_BOX, 0, bad) == E_FAIL);          // 96 is not a power of two
    CHECK(GenerateMipMaps(dev, src, TEX_FILTER_DEFAULT, 1, bad) == E_INVALIDARG);
    std::puts("gpu run OK");
    return 0;
}

// dds_save <tight pixels.bin> <w> <h> <format> <arraySize> <mipLevels> <miscFlags> <ddsFlags> <out.dds> [depth] [miscFlags2] [dimension]
// a depth argument > 0 makes it a volume unless a dimension (2, 3, 4) says otherwise
static int dds_save(int n, char** a)
{
    const size_t w = std::strtoull(a[1], nullptr, 10), h = std::strtoull(a[2], nullptr, 10);
    TexMetadata m; m.width = w; m.height = h; m.depth = 1; m.format = DXGI_FORMAT(std::atoi(a[3]));
    m.arraySize = std::strtoull(a[4], nullptr, 10); m.mipLevels = std::strtoull(a[5], nullptr, 10); m.miscFlags = uint32_t(std::strtoul(a[6], nullptr, 0));
    if (n > 9 && std::strtoull(a[9], nullptr, 10) > 0) { m.depth = std::strtoull(a[9], nullptr, 10); m.dimension = TEX_DIMENSION_TEXTURE3D; }
    if (n > 10) m.miscFlags2 = uint32_t(std::strtoul(a[10], nullptr, 0));
    if (n > 11) { m.dimension = TEX_DIMENSION(std::atoi(a[11])); if (m.dimension != TEX_DIMENSION_TEXTURE3D) m.depth = 1; }
    ScratchImage si;
    HRESULT hr = si.Initialize(m);
    if (FAILED(hr)) { std::printf("hr %08x\n", unsigned(hr)); return 3; }
    FILE* f = std::fopen(a[0], "rb");
    if (!f || std::fread(si.GetPixels(), 1, si.GetPixelsSize(), f) != si.GetPixelsSize()) return 4;
    std::fclose(f);
    hr = SaveToDDSFile(si.GetImages(), si.GetImageCount(), si.GetMetadata(), DDS_FLAGS(std::strtoul(a[7], nullptr, 0)), a[8]);
    std::printf("hr %08x\n", unsigned(hr));
    return FAILED(hr) ? 3 : 0;
}

static void print_meta(const TexMetadata& m)
{
    std::printf("meta %zu %zu %zu %u %zu %zu %u %u %u\n", m.width, m.height, m.depth, unsigned(m.format), m.arraySize, m.mipLevels, m.miscFlags, m.miscFlags2, unsigned(m.dimension));
}

// dds_load <in.dds> <out tight pixels.bin> [ddsFlags]: prints the HRESULT and the metadata
static int dds_load(int n, char** a)
{
    ScratchImage si; TexMetadata m;
    const HRESULT hr = LoadFromDDSFile(a[0], DDS_FLAGS(n > 2 ? std::strtoul(a[2], nullptr, 0) : 0), &m, si);
    std::printf("hr %08x\n", unsigned(hr));
    if (FAILED(hr)) return 3;
    print_meta(m);
    dump(a[1], si.GetPixels(), si.GetPixelsSize());
    return 0;
}

// dds_load_many <list.txt>: every line "<in.dds> <ddsFlags>"; prints "<hr>" or "<hr> meta ..." per line, writes <in.dds>.out
static int dds_load_many(char** a)
{
    FILE* list = std::fopen(a[0], "r");
    if (!list) return 4;
    char path[4096]; unsigned long flags;
    while (std::fscanf(list, "%4095s %lu", path, &flags) == 2)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) return 4;
        std::fseek(f, 0, SEEK_END); const long len = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> buf(size_t(len > 0 ? len : 0));
        if (!buf.empty() && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) return 4;
        std::fclose(f);
        ScratchImage si; TexMetadata m;
        const HRESULT hr = LoadFromDDSMemory(buf.data(), buf.size(), DDS_FLAGS(flags), &m, si);
        std::printf("hr %08x ", unsigned(hr));
        if (FAILED(hr)) { std::puts(""); continue; }
        print_meta(m);
        // the header-only query must agree with the loader
        TexMetadata m2;
        if (GetMetadataFromDDSMemory(buf.data(), buf.size(), DDS_FLAGS(flags), m2) != S_OK || m2.width != m.width || m2.format != m.format || m2.mipLevels != m.mipLevels)
        { std::fprintf(stderr, "GetMetadataFromDDSMemory disagrees with LoadFromDDSMemory for %s\n", path); return 5; }
        dump(std::string(path) + ".out", si.GetPixels(), si.GetPixelsSize());
    }
    std::fclose(list);
    return 0;
}

// codec_load_many <hdr|tga> <list.txt>: every line "<file> <flags>"; prints "hr <hr>[ meta w h format miscFlags2]", writes <file>.out
static int codec_load_many(char** a)
{
    const bool hdr = !std::strcmp(a[0], "hdr");
    FILE* list = std::fopen(a[1], "r");
    if (!list) return 4;
    char path[4096]; unsigned long flags;
    while (std::fscanf(list, "%4095s %lu", path, &flags) == 2)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) return 4;
        std::fseek(f, 0, SEEK_END); const long len = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> buf(size_t(len > 0 ? len : 0));
        if (!buf.empty() && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) return 4;
        std::fclose(f);
        ScratchImage si; TexMetadata m, m2;
        const HRESULT hr = hdr ? LoadFromHDRMemory(buf.data(), buf.size(), &m, si) : LoadFromTGAMemory(buf.data(), buf.size(), TGA_FLAGS(flags), &m, si);
        std::printf("hr %08x", unsigned(hr));
        if (FAILED(hr)) { std::puts(""); continue; }
        const HRESULT hr2 = hdr ? GetMetadataFromHDRMemory(buf.data(), buf.size(), m2) : GetMetadataFromTGAMemory(buf.data(), buf.size(), TGA_FLAGS(flags), m2);
        std::printf(" meta %zu %zu %u %u", m.width, m.height, unsigned(m.format), m.miscFlags2);
        if (!hdr) std::printf(" %u %u %u %u", unsigned(si.GetMetadata().format), unsigned(hr2), unsigned(m2.format), m2.miscFlags2);
        std::puts("");
        if (hr2 != S_OK || m2.width != m.width || m2.height != m.height || m2.format != m.format)
        { std::fprintf(stderr, "the header-only query disagrees with the loader for %s\n", path); return 5; }
        dump(std::string(path) + ".out", si.GetPixels(), si.GetPixelsSize());
    }
    std::fclose(list);
    return 0;
}

// codec_save <hdr|tga> <pixels.bin> <w> <h> <format> <rowPitch> <flags> <out> [alphaMode]
static int codec_save(char** a)
{
    const bool hdr = !std::strcmp(a[0], "hdr");
    Image im; im.width = std::strtoull(a[2], nullptr, 10); im.height = std::strtoull(a[3], nullptr, 10); im.format = DXGI_FORMAT(std::atoi(a[4]));
    im.rowPitch = std::strtoull(a[5], nullptr, 10); im.slicePitch = im.rowPitch * im.height;
    std::vector<uint8_t> px(im.slicePitch);
    FILE* f = std::fopen(a[1], "rb");
    if (!f || std::fread(px.data(), 1, px.size(), f) != px.size()) return 4;
    std::fclose(f);
    im.pixels = px.data();
    // TGA: flags, and an optional 9th argument = alpha mode of the metadata to pass (adds the TGA 2.0 extension area)
    TexMetadata md; md.width = im.width; md.height = im.height; md.depth = md.arraySize = md.mipLevels = 1; md.format = im.format;
    if (a[8]) md.SetAlphaMode(TEX_ALPHA_MODE(std::atoi(a[8])));
    const HRESULT hr = hdr ? SaveToHDRFile(im, a[7]) : SaveToTGAFile(im, TGA_FLAGS(std::strtoul(a[6], nullptr, 0)), a[7], a[8] ? &md : nullptr);
    std::printf("hr %08x\n", unsigned(hr));
    return FAILED(hr) ? 3 : 0;
}

// formats: the container-side format tables for every format id 0..200, and pitches for a set of sizes and CP_FLAGS
static int formats()
{
    const size_t dims[][2] = { { 1, 1 }, { 5, 3 }, { 16, 16 }, { 31, 7 }, { 256, 2 }, { 1023, 4 } };
    const uint32_t cps[] = { 0, CP_FLAGS_LEGACY_DWORD, CP_FLAGS_PARAGRAPH, CP_FLAGS_YMM, CP_FLAGS_ZMM, CP_FLAGS_PAGE4K, CP_FLAGS_BAD_DXTN_TAILS,
                             CP_FLAGS_24BPP, CP_FLAGS_16BPP, CP_FLAGS_8BPP, CP_FLAGS_24BPP | CP_FLAGS_LEGACY_DWORD, CP_FLAGS_8BPP | CP_FLAGS_LEGACY_DWORD };
    for (uint32_t f = 0; f <= 200; ++f)
    {
        const DXGI_FORMAT fmt = DXGI_FORMAT(f);
        const int bits = (IsCompressed(fmt) ? 1 : 0) | (IsPacked(fmt) ? 2 : 0) | (IsPlanar(fmt) ? 4 : 0) | (IsPalettized(fmt) ? 8 : 0) | (IsSRGB(fmt) ? 16 : 0) | (IsValid(fmt) ? 32 : 0) | (HasAlpha(fmt) ? 64 : 0);
        std::printf("fmt %u %zu %d\n", f, BitsPerPixel(fmt), bits);
        std::printf("more %u %zu %zu %u %u %u %u %u %d\n", f, BitsPerColor(fmt), BytesPerBlock(fmt), unsigned(MakeSRGB(fmt)), unsigned(MakeLinear(fmt)), unsigned(MakeTypeless(fmt)),
                    unsigned(MakeTypelessUNORM(fmt)), unsigned(MakeTypelessFLOAT(fmt)),
                    (IsVideo(fmt) ? 1 : 0) | (IsDepthStencil(fmt) ? 2 : 0) | (IsBGR(fmt) ? 4 : 0) | (IsTypeless(fmt, true) ? 8 : 0) | (IsTypeless(fmt, false) ? 16 : 0));
        for (uint32_t dim = 1; dim <= 5; ++dim)
        {
            TileShape ts;
            const HRESULT hr = ComputeTileShape(fmt, TEX_DIMENSION(dim), ts);
            std::printf("tile %u %u %08x %zu %zu %zu\n", f, dim, unsigned(hr), ts.width, ts.height, ts.depth);
        }
        for (const auto& d : dims)
            for (uint32_t cp : cps)
            {
                size_t rp = 0, sp = 0;
                const HRESULT hr = ComputePitch(fmt, d[0], d[1], rp, sp, CP_FLAGS(cp));
                std::printf("pitch %u %zu %zu %u %08x %zu %zu %zu\n", f, d[0], d[1], cp, unsigned(hr), FAILED(hr) ? size_t(0) : rp, FAILED(hr) ? size_t(0) : sp, ComputeScanlines(fmt, d[1]));
            }
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 11 && !std::strcmp(argv[1], "dds_save")) return dds_save(argc - 2, argv + 2);
    if (argc >= 4 && !std::strcmp(argv[1], "dds_load")) return dds_load(argc - 2, argv + 2);
    if (argc >= 3 && !std::strcmp(argv[1], "dds_load_many")) return dds_load_many(argv + 2);
    if (argc >= 2 && !std::strcmp(argv[1], "formats")) return formats();
    if (argc >= 4 && !std::strcmp(argv[1], "codec_load_many")) return codec_load_many(argv + 2);
    if (argc >= 10 && !std::strcmp(argv[1], "codec_save")) return codec_save(argv + 2);
    if (argc >= 2 && !std::strcmp(argv[1], "cpu")) return cpu_checks();
    if (argc >= 3 && !std::strcmp(argv[1], "gpu")) return gpu_run(argv[2]);
    std::fprintf(stderr, "usage: host_api_test cpu | gpu <outdir>\n");
    return 64;
}
