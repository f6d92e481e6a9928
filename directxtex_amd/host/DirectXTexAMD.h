// DirectXTexAMD.h - C++ host layer of the MI355X DirectXTex hot path.
//
// Keeps the reference's API surface for the path (DirectXTex.h:187-216 TexMetadata, :437-498 Image / ScratchImage,
// :799-846 Resize / Convert / GenerateMipMaps, :929-968 Compress / Decompress, :1021 ComputeMSE): same type and
// function names, argument order and meaning, HRESULT codes and ownership rules ("ScratchImage in, ScratchImage out",
// the callee Release()s and re-initialises the output and releases it again on failure). Everything below the
// signatures is new: the functions validate like the reference, allocate the output with ScratchImage, and hand
// `dxtex_image` views to the C ABI (include/dxtex_amd.h), i.e. to HIP kernels. There is no CPU compute path.
//
// Where the reference takes an ID3D11Device* (DirectXTex.h:946-963) these take a Device (a dxtex_ctx bound to one
// MI355X). One Device per GPU / host thread; Devices share nothing.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>

struct dxtex_ctx;

namespace DirectXTexAMD
{
using HRESULT = int32_t;
constexpr HRESULT S_OK = 0;
constexpr HRESULT E_FAIL = HRESULT(0x80004005);
constexpr HRESULT E_INVALIDARG = HRESULT(0x80070057);
constexpr HRESULT E_OUTOFMEMORY = HRESULT(0x8007000E);
constexpr HRESULT E_POINTER = HRESULT(0x80004003);
constexpr HRESULT E_NOTIMPL = HRESULT(0x80004001);
constexpr HRESULT E_ABORT = HRESULT(0x80004004);
constexpr HRESULT HRESULT_E_NOT_SUPPORTED = HRESULT(0x80070032);
constexpr HRESULT HRESULT_E_ARITHMETIC_OVERFLOW = HRESULT(0x80070216);
inline bool FAILED(HRESULT hr) noexcept { return hr < 0; }
inline bool SUCCEEDED(HRESULT hr) noexcept { return hr >= 0; }

// DXGI_FORMAT values this layer understands (public DXGI numbering).
// The public DXGI numbering. The container side of the library (ScratchImage, pitches, DDS) knows every format; the GPU
// entry points work on the subset IsSupportedOnDevice() names and return HRESULT_E_NOT_SUPPORTED for the rest.
enum DXGI_FORMAT : uint32_t
{
    DXGI_FORMAT_UNKNOWN = 0,
    DXGI_FORMAT_R32G32B32A32_TYPELESS = 1, DXGI_FORMAT_R32G32B32A32_FLOAT = 2, DXGI_FORMAT_R32G32B32A32_UINT = 3, DXGI_FORMAT_R32G32B32A32_SINT = 4,
    DXGI_FORMAT_R32G32B32_TYPELESS = 5, DXGI_FORMAT_R32G32B32_FLOAT = 6, DXGI_FORMAT_R32G32B32_UINT = 7, DXGI_FORMAT_R32G32B32_SINT = 8,
    DXGI_FORMAT_R16G16B16A16_TYPELESS = 9, DXGI_FORMAT_R16G16B16A16_FLOAT = 10, DXGI_FORMAT_R16G16B16A16_UNORM = 11, DXGI_FORMAT_R16G16B16A16_UINT = 12,
    DXGI_FORMAT_R16G16B16A16_SNORM = 13, DXGI_FORMAT_R16G16B16A16_SINT = 14,
    DXGI_FORMAT_R32G32_TYPELESS = 15, DXGI_FORMAT_R32G32_FLOAT = 16, DXGI_FORMAT_R32G32_UINT = 17, DXGI_FORMAT_R32G32_SINT = 18,
    DXGI_FORMAT_R32G8X24_TYPELESS = 19, DXGI_FORMAT_D32_FLOAT_S8X24_UINT = 20, DXGI_FORMAT_R32_FLOAT_X8X24_TYPELESS = 21, DXGI_FORMAT_X32_TYPELESS_G8X24_UINT = 22,
    DXGI_FORMAT_R10G10B10A2_TYPELESS = 23, DXGI_FORMAT_R10G10B10A2_UNORM = 24, DXGI_FORMAT_R10G10B10A2_UINT = 25, DXGI_FORMAT_R11G11B10_FLOAT = 26,
    DXGI_FORMAT_R8G8B8A8_TYPELESS = 27, DXGI_FORMAT_R8G8B8A8_UNORM = 28, DXGI_FORMAT_R8G8B8A8_UNORM_SRGB = 29, DXGI_FORMAT_R8G8B8A8_UINT = 30,
    DXGI_FORMAT_R8G8B8A8_SNORM = 31, DXGI_FORMAT_R8G8B8A8_SINT = 32,
    DXGI_FORMAT_R16G16_TYPELESS = 33, DXGI_FORMAT_R16G16_FLOAT = 34, DXGI_FORMAT_R16G16_UNORM = 35, DXGI_FORMAT_R16G16_UINT = 36,
    DXGI_FORMAT_R16G16_SNORM = 37, DXGI_FORMAT_R16G16_SINT = 38,
    DXGI_FORMAT_R32_TYPELESS = 39, DXGI_FORMAT_D32_FLOAT = 40, DXGI_FORMAT_R32_FLOAT = 41, DXGI_FORMAT_R32_UINT = 42, DXGI_FORMAT_R32_SINT = 43,
    DXGI_FORMAT_R24G8_TYPELESS = 44, DXGI_FORMAT_D24_UNORM_S8_UINT = 45, DXGI_FORMAT_R24_UNORM_X8_TYPELESS = 46, DXGI_FORMAT_X24_TYPELESS_G8_UINT = 47,
    DXGI_FORMAT_R8G8_TYPELESS = 48, DXGI_FORMAT_R8G8_UNORM = 49, DXGI_FORMAT_R8G8_UINT = 50, DXGI_FORMAT_R8G8_SNORM = 51, DXGI_FORMAT_R8G8_SINT = 52,
    DXGI_FORMAT_R16_TYPELESS = 53, DXGI_FORMAT_R16_FLOAT = 54, DXGI_FORMAT_D16_UNORM = 55, DXGI_FORMAT_R16_UNORM = 56, DXGI_FORMAT_R16_UINT = 57,
    DXGI_FORMAT_R16_SNORM = 58, DXGI_FORMAT_R16_SINT = 59,
    DXGI_FORMAT_R8_TYPELESS = 60, DXGI_FORMAT_R8_UNORM = 61, DXGI_FORMAT_R8_UINT = 62, DXGI_FORMAT_R8_SNORM = 63, DXGI_FORMAT_R8_SINT = 64,
    DXGI_FORMAT_A8_UNORM = 65, DXGI_FORMAT_R1_UNORM = 66, DXGI_FORMAT_R9G9B9E5_SHAREDEXP = 67,
    DXGI_FORMAT_R8G8_B8G8_UNORM = 68, DXGI_FORMAT_G8R8_G8B8_UNORM = 69,
    DXGI_FORMAT_BC1_TYPELESS = 70, DXGI_FORMAT_BC1_UNORM = 71, DXGI_FORMAT_BC1_UNORM_SRGB = 72,
    DXGI_FORMAT_BC2_TYPELESS = 73, DXGI_FORMAT_BC2_UNORM = 74, DXGI_FORMAT_BC2_UNORM_SRGB = 75,
    DXGI_FORMAT_BC3_TYPELESS = 76, DXGI_FORMAT_BC3_UNORM = 77, DXGI_FORMAT_BC3_UNORM_SRGB = 78,
    DXGI_FORMAT_BC4_TYPELESS = 79, DXGI_FORMAT_BC4_UNORM = 80, DXGI_FORMAT_BC4_SNORM = 81,
    DXGI_FORMAT_BC5_TYPELESS = 82, DXGI_FORMAT_BC5_UNORM = 83, DXGI_FORMAT_BC5_SNORM = 84,
    DXGI_FORMAT_B5G6R5_UNORM = 85, DXGI_FORMAT_B5G5R5A1_UNORM = 86, DXGI_FORMAT_B8G8R8A8_UNORM = 87, DXGI_FORMAT_B8G8R8X8_UNORM = 88,
    DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM = 89, DXGI_FORMAT_B8G8R8A8_TYPELESS = 90, DXGI_FORMAT_B8G8R8A8_UNORM_SRGB = 91,
    DXGI_FORMAT_B8G8R8X8_TYPELESS = 92, DXGI_FORMAT_B8G8R8X8_UNORM_SRGB = 93,
    DXGI_FORMAT_BC6H_TYPELESS = 94, DXGI_FORMAT_BC6H_UF16 = 95, DXGI_FORMAT_BC6H_SF16 = 96,
    DXGI_FORMAT_BC7_TYPELESS = 97, DXGI_FORMAT_BC7_UNORM = 98, DXGI_FORMAT_BC7_UNORM_SRGB = 99,
    DXGI_FORMAT_AYUV = 100, DXGI_FORMAT_Y410 = 101, DXGI_FORMAT_Y416 = 102, DXGI_FORMAT_NV12 = 103, DXGI_FORMAT_P010 = 104, DXGI_FORMAT_P016 = 105,
    DXGI_FORMAT_420_OPAQUE = 106, DXGI_FORMAT_YUY2 = 107, DXGI_FORMAT_Y210 = 108, DXGI_FORMAT_Y216 = 109, DXGI_FORMAT_NV11 = 110,
    DXGI_FORMAT_AI44 = 111, DXGI_FORMAT_IA44 = 112, DXGI_FORMAT_P8 = 113, DXGI_FORMAT_A8P8 = 114, DXGI_FORMAT_B4G4R4A4_UNORM = 115,
    DXGI_FORMAT_P208 = 130, DXGI_FORMAT_V208 = 131, DXGI_FORMAT_V408 = 132,
    DXGI_FORMAT_SAMPLER_FEEDBACK_MIN_MIP_OPAQUE = 189, DXGI_FORMAT_SAMPLER_FEEDBACK_MIP_REGION_USED_OPAQUE = 190, DXGI_FORMAT_A4B4G4R4_UNORM = 191,
};

enum TEX_DIMENSION : uint32_t { TEX_DIMENSION_TEXTURE1D = 2, TEX_DIMENSION_TEXTURE2D = 3, TEX_DIMENSION_TEXTURE3D = 4 };

// TEX_COMPRESS_FLAGS / TEX_FILTER_FLAGS: the reference's values (DirectXTex.h:741-793, :887-917)
enum TEX_COMPRESS_FLAGS : uint32_t
{
    TEX_COMPRESS_DEFAULT = 0, TEX_COMPRESS_RGB_DITHER = 0x10000, TEX_COMPRESS_A_DITHER = 0x20000, TEX_COMPRESS_DITHER = 0x30000,
    TEX_COMPRESS_UNIFORM = 0x40000, TEX_COMPRESS_BC7_USE_3SUBSETS = 0x80000, TEX_COMPRESS_BC7_QUICK = 0x100000,
    TEX_COMPRESS_SRGB_IN = 0x1000000, TEX_COMPRESS_SRGB_OUT = 0x2000000, TEX_COMPRESS_SRGB = 0x3000000, TEX_COMPRESS_PARALLEL = 0x10000000,
};
enum TEX_FILTER_FLAGS : uint32_t
{
    TEX_FILTER_DEFAULT = 0, TEX_FILTER_WRAP_U = 0x1, TEX_FILTER_WRAP_V = 0x2, TEX_FILTER_WRAP = 0x7,
    TEX_FILTER_MIRROR_U = 0x10, TEX_FILTER_MIRROR_V = 0x20, TEX_FILTER_MIRROR = 0x70, TEX_FILTER_FLOAT_X2BIAS = 0x200,
    TEX_FILTER_RGB_COPY_RED = 0x1000, TEX_FILTER_RGB_COPY_GREEN = 0x2000, TEX_FILTER_RGB_COPY_BLUE = 0x4000, TEX_FILTER_RGB_COPY_ALPHA = 0x8000,
    TEX_FILTER_POINT = 0x100000, TEX_FILTER_LINEAR = 0x200000, TEX_FILTER_CUBIC = 0x300000, TEX_FILTER_BOX = 0x400000, TEX_FILTER_FANT = 0x400000,
    TEX_FILTER_TRIANGLE = 0x500000, TEX_FILTER_SRGB_IN = 0x1000000, TEX_FILTER_SRGB_OUT = 0x2000000, TEX_FILTER_SRGB = 0x3000000,
};
constexpr float TEX_THRESHOLD_DEFAULT = 0.5f;

// Format facts for every DXGI format (DirectXTex.inl:57-60, DirectXTexUtil.cpp:340-960)
constexpr bool IsValid(DXGI_FORMAT fmt) noexcept { return uint32_t(fmt) >= 1 && uint32_t(fmt) <= 191; }
bool IsCompressed(DXGI_FORMAT fmt) noexcept;
bool IsPacked(DXGI_FORMAT fmt) noexcept;
bool IsPlanar(DXGI_FORMAT fmt) noexcept;
bool IsPalettized(DXGI_FORMAT fmt) noexcept;
bool IsSRGB(DXGI_FORMAT fmt) noexcept;
bool HasAlpha(DXGI_FORMAT fmt) noexcept;
bool IsVideo(DXGI_FORMAT fmt) noexcept;
bool IsDepthStencil(DXGI_FORMAT fmt) noexcept;
bool IsBGR(DXGI_FORMAT fmt) noexcept;
bool IsTypeless(DXGI_FORMAT fmt, bool partialTypeless = true) noexcept;
size_t BitsPerColor(DXGI_FORMAT fmt) noexcept;            // the widest channel; 0 for palettised formats
size_t BytesPerBlock(DXGI_FORMAT fmt) noexcept;           // 8 / 16 for BC formats, else 0
DXGI_FORMAT MakeLinear(DXGI_FORMAT fmt) noexcept;
DXGI_FORMAT MakeTypeless(DXGI_FORMAT fmt) noexcept;
DXGI_FORMAT MakeTypelessUNORM(DXGI_FORMAT fmt) noexcept;
DXGI_FORMAT MakeTypelessFLOAT(DXGI_FORMAT fmt) noexcept;
DXGI_FORMAT MakeSRGB(DXGI_FORMAT fmt) noexcept;
size_t BitsPerPixel(DXGI_FORMAT fmt) noexcept;
// true for the formats the GPU entry points (Compress, Convert, Resize, ...) accept
bool IsSupportedOnDevice(DXGI_FORMAT fmt) noexcept;
// ComputePitch / ComputeScanlines (DirectXTexUtil.cpp:961-1247). CP_FLAGS values as in the reference (DirectXTex.h:104-137).
enum CP_FLAGS : uint32_t
{
    CP_FLAGS_NONE = 0, CP_FLAGS_LEGACY_DWORD = 0x1, CP_FLAGS_PARAGRAPH = 0x2, CP_FLAGS_YMM = 0x4, CP_FLAGS_ZMM = 0x8, CP_FLAGS_PAGE4K = 0x200,
    CP_FLAGS_BAD_DXTN_TAILS = 0x1000, CP_FLAGS_24BPP = 0x10000, CP_FLAGS_16BPP = 0x20000, CP_FLAGS_8BPP = 0x40000,
};
HRESULT ComputePitch(DXGI_FORMAT fmt, size_t width, size_t height, size_t& rowPitch, size_t& slicePitch, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
size_t ComputeScanlines(DXGI_FORMAT fmt, size_t height) noexcept;
// the standard shape of a 64 KiB tile of a tiled resource (DirectXTexUtil.cpp:1251-1405)
struct TileShape { size_t width, height, depth; };
// CalculateMipLevels (DirectXTexMipmaps.cpp:40-69): mipLevels == 0 asks for the full chain
HRESULT ComputeTileShape(DXGI_FORMAT fmt, TEX_DIMENSION dimension, TileShape& tiling) noexcept;
bool CalculateMipLevels(size_t width, size_t height, size_t& mipLevels) noexcept;
bool CalculateMipLevels3D(size_t width, size_t height, size_t depth, size_t& mipLevels) noexcept;

enum TEX_MISC_FLAG : uint32_t { TEX_MISC_TEXTURECUBE = 0x4 };
enum TEX_MISC_FLAG2 : uint32_t { TEX_MISC2_ALPHA_MODE_MASK = 0x7 };
enum TEX_ALPHA_MODE : uint32_t
{
    TEX_ALPHA_MODE_UNKNOWN = 0, TEX_ALPHA_MODE_STRAIGHT = 1, TEX_ALPHA_MODE_PREMULTIPLIED = 2, TEX_ALPHA_MODE_OPAQUE = 3, TEX_ALPHA_MODE_CUSTOM = 4,
};

struct TexMetadata
{
    size_t width = 0, height = 0, depth = 0, arraySize = 0, mipLevels = 0;
    uint32_t miscFlags = 0, miscFlags2 = 0;
    DXGI_FORMAT format = DXGI_FORMAT_UNKNOWN;
    TEX_DIMENSION dimension = TEX_DIMENSION_TEXTURE2D;
    // index = item * mipLevels + mip for 1D / 2D textures (DirectXTexUtil.cpp:1695-1740)
    size_t ComputeIndex(size_t mip, size_t item, size_t slice) const noexcept;
    // D3D11CalcSubresource / D3D12CalcSubresource (DirectXTexUtil.cpp:1744-1806); uint32_t(-1) when out of range
    uint32_t CalculateSubresource(size_t mip, size_t item) const noexcept;
    uint32_t CalculateSubresource(size_t mip, size_t item, size_t plane) const noexcept;
    bool IsVolumemap() const noexcept { return dimension == TEX_DIMENSION_TEXTURE3D; }
    bool IsCubemap() const noexcept { return (miscFlags & TEX_MISC_TEXTURECUBE) != 0; }
    // the alpha mode lives in the low three bits of miscFlags2 (DirectXTex.h:174-207)
    bool IsPMAlpha() const noexcept { return (miscFlags2 & TEX_MISC2_ALPHA_MODE_MASK) == TEX_ALPHA_MODE_PREMULTIPLIED; }
    void SetAlphaMode(TEX_ALPHA_MODE mode) noexcept { miscFlags2 = (miscFlags2 & ~uint32_t(TEX_MISC2_ALPHA_MODE_MASK)) | uint32_t(mode); }
    TEX_ALPHA_MODE GetAlphaMode() const noexcept { return TEX_ALPHA_MODE(miscFlags2 & TEX_MISC2_ALPHA_MODE_MASK); }
};

struct Image
{
    size_t width = 0, height = 0;
    DXGI_FORMAT format = DXGI_FORMAT_UNKNOWN;
    size_t rowPitch = 0, slicePitch = 0;
    uint8_t* pixels = nullptr;
};

// One 16-byte-aligned, zero-filled allocation; images laid out item-major, then by mip (DirectXTexImage.cpp:173-209).
class ScratchImage
{
public:
    ScratchImage() noexcept = default;
    ScratchImage(ScratchImage&& o) noexcept { *this = static_cast<ScratchImage&&>(o); }
    ScratchImage& operator=(ScratchImage&& o) noexcept;
    ScratchImage(const ScratchImage&) = delete;
    ScratchImage& operator=(const ScratchImage&) = delete;
    ~ScratchImage() { Release(); }

    // any valid, non-palettised DXGI format can be held (DirectXTexImage.cpp:300-505); flags select the pitch rule
    HRESULT Initialize(const TexMetadata& mdata, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT Initialize1D(DXGI_FORMAT fmt, size_t length, size_t arraySize, size_t mipLevels, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT Initialize2D(DXGI_FORMAT fmt, size_t width, size_t height, size_t arraySize, size_t mipLevels, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT Initialize3D(DXGI_FORMAT fmt, size_t width, size_t height, size_t depth, size_t mipLevels, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT InitializeCube(DXGI_FORMAT fmt, size_t width, size_t height, size_t nCubes, size_t mipLevels, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    // copies of caller images (DirectXTexImage.cpp:534-723): one image, an array, cubemaps (a multiple of six), a volume
    HRESULT InitializeFromImage(const Image& srcImage, bool allow1D = false, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT InitializeArrayFromImages(const Image* images, size_t nImages, bool allow1D = false, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT InitializeCubeFromImages(const Image* images, size_t nImages, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    HRESULT Initialize3DFromImages(const Image* images, size_t depth, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    bool OverrideFormat(DXGI_FORMAT f) noexcept;               // relabels the pixels (same size per texel is the caller's business)
    void Release() noexcept;

    const TexMetadata& GetMetadata() const noexcept { return m_metadata; }
    const Image* GetImage(size_t mip, size_t item, size_t slice) const noexcept;
    const Image* GetImages() const noexcept { return m_images.get(); }
    size_t GetImageCount() const noexcept { return m_nimages; }
    uint8_t* GetPixels() const noexcept { return m_memory; }
    size_t GetPixelsSize() const noexcept { return m_size; }

private:
    size_t m_nimages = 0, m_size = 0;
    TexMetadata m_metadata;
    std::unique_ptr<Image[]> m_images;
    uint8_t* m_memory = nullptr;
};

// The MI355X counterpart of the ID3D11Device* the reference's GPU overloads take.
class Device
{
public:
    Device() noexcept = default;
    ~Device();
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    HRESULT Create(int hipDevice) noexcept;          // E_FAIL when no gfx950 device is visible
    // GPUCompressBC::Prepare's role (BCDirectCompute.h:28): size the encoder's scratch and staging for `count` images of this
    // shape now, so that the Compress calls that follow allocate nothing. Optional.
    HRESULT Prepare(size_t width, size_t height, DXGI_FORMAT srcFormat, DXGI_FORMAT bcFormat, TEX_COMPRESS_FLAGS flags, size_t count) noexcept;
    explicit operator bool() const noexcept { return m_ctx != nullptr; }
    dxtex_ctx* Get() const noexcept { return m_ctx; }
    const char* LastError() const noexcept;
    // Granularity of the progress / cancel callbacks of CompressEx / ConvertEx on ONE image: the image goes to the GPU as bands of
    // whole (block) rows of about this many blocks / texels, and the callback is asked between bands. 0 keeps the default
    // (2^18 blocks, 2^24 texels: large enough to fill the GPU). The bytes written never depend on it.
    void SetProgressBands(size_t compressBlocks, size_t convertTexels) noexcept { m_bandBlocks = compressBlocks; m_bandTexels = convertTexels; }
    size_t ProgressBandBlocks() const noexcept { return m_bandBlocks; }
    size_t ProgressBandTexels() const noexcept { return m_bandTexels; }
private:
    dxtex_ctx* m_ctx = nullptr;
    size_t m_bandBlocks = 0, m_bandTexels = 0;
};

// ---- the path's entry points (shapes of DirectXTex.h:799-846, :946-968, :1021) ----------------------------------------
HRESULT Compress(Device& device, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImage) noexcept;
HRESULT Compress(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                 TEX_COMPRESS_FLAGS compress, float threshold, ScratchImage& cImages) noexcept;
// CompressEx (DirectXTex.h:922-944). statusCallBack(done, total) -> false cancels: the result is released and E_ABORT
// returned. One image reports rows (the image goes to the GPU as bands of block rows, asked between bands); a set
// reports images. alphaWeight belongs to the reference's DirectCompute BC7 encoder and is not used here either way.
constexpr float TEX_ALPHA_WEIGHT_DEFAULT = 1.0f;
using StatusCallback = std::function<bool(size_t, size_t)>;
struct CompressOptions { TEX_COMPRESS_FLAGS flags; float threshold; float alphaWeight; };
HRESULT CompressEx(Device& device, const Image& srcImage, DXGI_FORMAT format, const CompressOptions& options, ScratchImage& cImage,
                   StatusCallback statusCallBack = nullptr);
HRESULT CompressEx(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                   const CompressOptions& options, ScratchImage& cImages, StatusCallback statusCallBack = nullptr);
// ONE image over several devices (normally one Device per GPU of the node): the image's block rows - for GenerateMipMaps the destination
// rows of its large levels, with the filter's halo of source rows - are dealt out over the devices, each stripe on a thread of its own
// (dxtex_compress_multi / dxtex_generate_mips_multi). The role of CompressBC_Parallel's split over OpenMP threads
// (DirectXTexCompress.cpp:257-281); the result is byte for byte that of the single-device overload.
HRESULT Compress(Device* const* devices, size_t ndevices, const Image& srcImage, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold,
                 ScratchImage& cImage) noexcept;
HRESULT GenerateMipMaps(Device* const* devices, size_t ndevices, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept;
// format == DXGI_FORMAT_UNKNOWN picks the default target (DefaultDecompress, DirectXTexCompress.cpp:377-421)
HRESULT Decompress(Device& device, const Image& cImage, DXGI_FORMAT format, ScratchImage& image) noexcept;
HRESULT Decompress(Device& device, const Image* cImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format, ScratchImage& images) noexcept;
// levels == 0 generates the full chain
HRESULT GenerateMipMaps(Device& device, const Image& baseImage, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept;
HRESULT GenerateMipMaps(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_FILTER_FLAGS filter,
                        size_t levels, ScratchImage& mipChain) noexcept;
// volume textures (DirectXTex.h:853-858): the base slices are `depth` images of one size
HRESULT GenerateMipMaps3D(Device& device, const Image* baseImages, size_t depth, TEX_FILTER_FLAGS filter, size_t levels, ScratchImage& mipChain) noexcept;
HRESULT GenerateMipMaps3D(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_FILTER_FLAGS filter, size_t levels,
                          ScratchImage& mipChain) noexcept;
HRESULT Resize(Device& device, const Image& srcImage, size_t width, size_t height, TEX_FILTER_FLAGS filter, ScratchImage& image) noexcept;
HRESULT Resize(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t width, size_t height,
               TEX_FILTER_FLAGS filter, ScratchImage& result) noexcept;
HRESULT Convert(Device& device, const Image& srcImage, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, ScratchImage& image) noexcept;
HRESULT Convert(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                TEX_FILTER_FLAGS filter, float threshold, ScratchImage& result) noexcept;
// ConvertEx (DirectXTex.h:812-832), callback as for CompressEx
struct ConvertOptions { TEX_FILTER_FLAGS filter; float threshold; };
HRESULT ConvertEx(Device& device, const Image& srcImage, DXGI_FORMAT format, const ConvertOptions& options, ScratchImage& image,
                  StatusCallback statusCallBack = nullptr);
HRESULT ConvertEx(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, DXGI_FORMAT format,
                  const ConvertOptions& options, ScratchImage& result, StatusCallback statusCallBack = nullptr);
// mse = sum of the per-channel values, mseV[4] the per-channel MSE over [0,1] floats
// PremultiplyAlpha (DirectXTex.h:864-884). TEX_PMALPHA_FLAGS values as in the reference.
enum TEX_PMALPHA_FLAGS : uint32_t
{
    TEX_PMALPHA_DEFAULT = 0, TEX_PMALPHA_IGNORE_SRGB = 0x1, TEX_PMALPHA_REVERSE = 0x2,
    TEX_PMALPHA_SRGB_IN = 0x1000000, TEX_PMALPHA_SRGB_OUT = 0x2000000, TEX_PMALPHA_SRGB = 0x3000000,
};
HRESULT PremultiplyAlpha(Device& device, const Image& srcImage, TEX_PMALPHA_FLAGS flags, ScratchImage& image) noexcept;
HRESULT PremultiplyAlpha(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, TEX_PMALPHA_FLAGS flags, ScratchImage& result) noexcept;
// ScaleMipMapsAlphaForCoverage (DirectXTex.h:848-851): mipChain must already be initialised with the chain's layout
HRESULT ScaleMipMapsAlphaForCoverage(Device& device, const Image* srcImages, size_t nimages, const TexMetadata& metadata, size_t item,
                                     float alphaReference, ScratchImage& mipChain) noexcept;

HRESULT ComputeMSE(Device& device, const Image& image1, const Image& image2, float& mse, float* mseV) noexcept;

// ---- the device-resident pipeline ---------------------------------------------------------------------------------------------------
// texconv runs resize -> convert -> mipmaps -> compress (Texconv/texconv.cpp:2609, 3109, 3434, 3711) as four ScratchImage -> ScratchImage calls;
// on a GPU that is four uploads and four downloads around a millisecond of kernels. The reference's own GPU path keeps its intermediate on
// the device (DirectXTexCompressGPU.cpp:34-140 converts on the way in, BCDirectCompute.cpp:373-642 reads the result back once). Here the
// whole chain can stay there: a DeviceScratchImage is a ScratchImage whose blob lives in HBM - same layout (item-major, then mips), same
// pitches, zero-filled - and every step below takes one and produces one with the SAME validation, error codes and kernels as the
// host-memory overload of the same name. Upload() is the chain's one host -> device copy, Download() its one device -> host copy.
class DeviceScratchImage
{
public:
    DeviceScratchImage() noexcept = default;
    DeviceScratchImage(DeviceScratchImage&& o) noexcept { *this = static_cast<DeviceScratchImage&&>(o); }
    DeviceScratchImage& operator=(DeviceScratchImage&& o) noexcept;
    DeviceScratchImage(const DeviceScratchImage&) = delete;
    DeviceScratchImage& operator=(const DeviceScratchImage&) = delete;
    ~DeviceScratchImage() { Release(); }

    // allocates and zero-fills (stream-ordered) device memory laid out like ScratchImage::Initialize(mdata, flags); the Device must outlive the image
    HRESULT Initialize(Device& device, const TexMetadata& mdata, CP_FLAGS flags = CP_FLAGS_NONE) noexcept;
    // Initialize + ONE host -> device copy of the source's blob (same layout on both sides)
    HRESULT Upload(Device& device, const ScratchImage& src) noexcept;
    // caller images of any pitch (nimages must match the metadata's image count): one copy per image when the pitches agree, else per row
    HRESULT Upload(Device& device, const Image* images, size_t nimages, const TexMetadata& metadata) noexcept;
    // dst.Initialize(metadata) + ONE device -> host copy; returns after the copy
    HRESULT Download(ScratchImage& dst) const noexcept;
    bool OverrideFormat(DXGI_FORMAT f) noexcept;
    void Release() noexcept;

    const TexMetadata& GetMetadata() const noexcept { return m_metadata; }
    // Image views whose `pixels` are DEVICE pointers: valid arguments for the *_device entry points of the C ABI, not for host code
    const Image* GetImage(size_t mip, size_t item, size_t slice) const noexcept;
    const Image* GetImages() const noexcept { return m_images.get(); }
    size_t GetImageCount() const noexcept { return m_nimages; }
    uint8_t* GetPixels() const noexcept { return m_memory; }
    size_t GetPixelsSize() const noexcept { return m_size; }
    Device* GetDevice() const noexcept { return m_device; }

private:
    Device* m_device = nullptr;
    size_t m_nimages = 0, m_size = 0;
    TexMetadata m_metadata;
    std::unique_ptr<Image[]> m_images;
    uint8_t* m_memory = nullptr;
};

// The steps, resident: same argument meaning and HRESULTs as the (Image*, nimages, metadata) overloads above; all work is queued on the
// Device's stream and the result may be handed to the next step at once. `src` and the result must belong to `device`.
HRESULT Compress(Device& device, const DeviceScratchImage& src, DXGI_FORMAT format, TEX_COMPRESS_FLAGS compress, float threshold, DeviceScratchImage& cImages) noexcept;
HRESULT Decompress(Device& device, const DeviceScratchImage& cImages, DXGI_FORMAT format, DeviceScratchImage& images) noexcept;
HRESULT GenerateMipMaps(Device& device, const DeviceScratchImage& src, TEX_FILTER_FLAGS filter, size_t levels, DeviceScratchImage& mipChain) noexcept;
HRESULT GenerateMipMaps3D(Device& device, const DeviceScratchImage& src, TEX_FILTER_FLAGS filter, size_t levels, DeviceScratchImage& mipChain) noexcept;
HRESULT Resize(Device& device, const DeviceScratchImage& src, size_t width, size_t height, TEX_FILTER_FLAGS filter, DeviceScratchImage& result) noexcept;
HRESULT Convert(Device& device, const DeviceScratchImage& src, DXGI_FORMAT format, TEX_FILTER_FLAGS filter, float threshold, DeviceScratchImage& result) noexcept;
HRESULT PremultiplyAlpha(Device& device, const DeviceScratchImage& src, TEX_PMALPHA_FLAGS flags, DeviceScratchImage& result) noexcept;
// every array item of a mip chain (the per-item loop texconv runs, texconv.cpp:3470-3490); the 10-step bisection per level reads 8 bytes back per step
HRESULT ScaleMipMapsAlphaForCoverage(Device& device, const DeviceScratchImage& src, float alphaReference, DeviceScratchImage& mipChain) noexcept;
// level 0 of every array item / depth slice as a texture with one mip level (what texconv keeps before it regenerates a chain, texconv.cpp:3324-3380)
HRESULT CopyTopLevels(Device& device, const DeviceScratchImage& src, DeviceScratchImage& result) noexcept;
// ScratchImage::IsAlphaAllOpaque (DirectXTexImage.cpp:800-852) as a device reduction; false on any failure, like the reference. The Image form takes
// device-resident images of one format (e.g. the top levels of a chain).
bool IsAlphaAllOpaque(Device& device, const DeviceScratchImage& image) noexcept;
bool IsAlphaAllOpaque(Device& device, const Image* deviceImages, size_t nimages) noexcept;
// bytes moved between host and device for this Device since the last reset (see dxtex_ctx_transfer_bytes)
void GetTransferBytes(Device& device, uint64_t& hostToDevice, uint64_t& deviceToHost, bool reset = false) noexcept;
} // namespace DirectXTexAMD

// ---- DDS container: the on-disk format either side of the path ---------------------------------
// DirectXTexDDS.cpp's reader and writer: textures, arrays, cubemaps and volumes of every DXGI format; the whole legacy
// (Direct3D 9) pixel-format table incl. the expanding / swizzling conversions (24 bpp RGB, 3:3:2, palettes, luminance,
// bump-map formats, the D3DX 10:10:10:2 reversal), every DDS_FLAGS reader and writer option. File names are UTF-8.
namespace DirectXTexAMD
{
enum DDS_FLAGS : uint32_t
{
    DDS_FLAGS_NONE = 0x0,
    DDS_FLAGS_LEGACY_DWORD = 0x1,                // rows of a legacy file are DWORD aligned
    DDS_FLAGS_NO_LEGACY_EXPANSION = 0x2,         // fail instead of expanding a legacy format
    DDS_FLAGS_NO_R10B10G10A2_FIXUP = 0x4,        // trust the 10:10:10:2 masks instead of assuming D3DX's reversed ones
    DDS_FLAGS_FORCE_RGB = 0x8,                   // BGRA / BGRX -> RGBA
    DDS_FLAGS_NO_16BPP = 0x10,                   // 5:6:5, 5:5:5:1, 4:4:4:4 -> RGBA8
    DDS_FLAGS_EXPAND_LUMINANCE = 0x20,           // L8, A8L8, L16 -> RGBA (replicated) instead of R / RG
    DDS_FLAGS_BAD_DXTN_TAILS = 0x40,             // mips smaller than a block were not written properly
    DDS_FLAGS_PERMISSIVE = 0x80,                 // accept known header variants
    DDS_FLAGS_IGNORE_MIPS = 0x100,               // top level only (non-array files)
    DDS_FLAGS_FORCE_DX10_EXT = 0x10000, DDS_FLAGS_FORCE_DX10_EXT_MISC2 = 0x20000, DDS_FLAGS_FORCE_DX9_LEGACY = 0x40000,
    DDS_FLAGS_FORCE_DXT5_RXGB = 0x80000, DDS_FLAGS_FORCE_24BPP_RGB = 0x100000,
    DDS_FLAGS_ALLOW_LARGE_FILES = 0x1000000,
};
constexpr HRESULT HRESULT_E_INVALID_DATA = HRESULT(0x8007000D), HRESULT_E_HANDLE_EOF = HRESULT(0x80070026),
                  HRESULT_E_CANNOT_MAKE = HRESULT(0x80070052), HRESULT_E_FILE_TOO_LARGE = HRESULT(0x800700DF);

// the file's DDS_PIXELFORMAT as read (DirectXTex.h:297-307)
struct DDSMetaData { uint32_t size, flags, fourCC, RGBBitCount, RBitMask, GBitMask, BBitMask, ABitMask; };

class Blob
{
public:
    Blob() noexcept = default;
    ~Blob() { Release(); }
    Blob(const Blob&) = delete;
    Blob& operator=(const Blob&) = delete;
    HRESULT Initialize(size_t size) noexcept;
    HRESULT Trim(size_t size) noexcept;              // shortens the logical size, keeps the allocation
    void Release() noexcept;
    uint8_t* GetBufferPointer() const noexcept { return m_buffer; }
    size_t GetBufferSize() const noexcept { return m_size; }
private:
    uint8_t* m_buffer = nullptr;
    size_t m_size = 0;
};

HRESULT GetMetadataFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata& metadata) noexcept;
HRESULT GetMetadataFromDDSMemoryEx(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata& metadata, DDSMetaData* ddPixelFormat) noexcept;
HRESULT GetMetadataFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata& metadata) noexcept;
HRESULT GetMetadataFromDDSFileEx(const char* szFile, DDS_FLAGS flags, TexMetadata& metadata, DDSMetaData* ddPixelFormat) noexcept;
HRESULT LoadFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT LoadFromDDSMemoryEx(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata* metadata, DDSMetaData* ddPixelFormat, ScratchImage& image) noexcept;
HRESULT LoadFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT LoadFromDDSFileEx(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, DDSMetaData* ddPixelFormat, ScratchImage& image) noexcept;
// required = header bytes; pDestination may be null to query (DirectXTexDDS.cpp:711-1033)
HRESULT EncodeDDSHeader(const TexMetadata& metadata, DDS_FLAGS flags, uint8_t* pDestination, size_t maxsize, size_t& required) noexcept;
HRESULT SaveToDDSMemory(const Image& image, DDS_FLAGS flags, Blob& blob) noexcept;
HRESULT SaveToDDSMemory(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, Blob& blob) noexcept;
HRESULT SaveToDDSFile(const Image& image, DDS_FLAGS flags, const char* szFile) noexcept;
HRESULT SaveToDDSFile(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, const char* szFile) noexcept;

// ---- Radiance RGBE (.hdr), DirectXTexHDR.cpp: loads to R32G32B32A32_FLOAT; saves RGBA32F / RGB32F / RGBA16F images ---------------
HRESULT GetMetadataFromHDRMemory(const void* pSource, size_t size, TexMetadata& metadata) noexcept;
HRESULT GetMetadataFromHDRFile(const char* szFile, TexMetadata& metadata) noexcept;
HRESULT LoadFromHDRMemory(const void* pSource, size_t size, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT LoadFromHDRFile(const char* szFile, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT SaveToHDRMemory(const Image& image, Blob& blob) noexcept;
HRESULT SaveToHDRFile(const Image& image, const char* szFile) noexcept;

// ---- Truevision TGA, DirectXTexTGA.cpp: 8-bit grey -> R8, 16-bit -> B5G5R5A1, 24-bit -> RGBA8 (opaque), 32-bit -> RGBA8, colour-mapped
// (24-bit palette) -> RGBA8; raw or run-length encoded; TGA 2.0 extension area for alpha mode and gamma ------------------------------
enum TGA_FLAGS : uint32_t
{
    TGA_FLAGS_NONE = 0x0,
    TGA_FLAGS_BGR = 0x1,                       // keep BGR order: 24-bit -> B8G8R8X8, 32-bit -> B8G8R8A8
    TGA_FLAGS_ALLOW_ALL_ZERO_ALPHA = 0x2,      // an alpha channel of all zeros is meant (default: treated as opaque)
    TGA_FLAGS_IGNORE_SRGB = 0x10, TGA_FLAGS_FORCE_SRGB = 0x20, TGA_FLAGS_FORCE_LINEAR = 0x40, TGA_FLAGS_DEFAULT_SRGB = 0x80,
};
HRESULT GetMetadataFromTGAMemory(const void* pSource, size_t size, TGA_FLAGS flags, TexMetadata& metadata) noexcept;
HRESULT GetMetadataFromTGAFile(const char* szFile, TGA_FLAGS flags, TexMetadata& metadata) noexcept;
HRESULT LoadFromTGAMemory(const void* pSource, size_t size, TGA_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT LoadFromTGAFile(const char* szFile, TGA_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
// saves RGBA8 / BGRA8 (32-bit), BGRX8 (24-bit), R8 / A8 (grey), B5G5R5A1 (16-bit); metadata adds the TGA 2.0 extension area
HRESULT SaveToTGAMemory(const Image& image, TGA_FLAGS flags, Blob& blob, const TexMetadata* metadata = nullptr) noexcept;
HRESULT SaveToTGAFile(const Image& image, TGA_FLAGS flags, const char* szFile, const TexMetadata* metadata = nullptr) noexcept;
} // namespace DirectXTexAMD
