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
enum DXGI_FORMAT : uint32_t
{
    DXGI_FORMAT_UNKNOWN = 0,
    DXGI_FORMAT_R32G32B32A32_FLOAT = 2, DXGI_FORMAT_R16G16B16A16_FLOAT = 10, DXGI_FORMAT_R16G16B16A16_UNORM = 11,
    DXGI_FORMAT_R32G32_FLOAT = 16, DXGI_FORMAT_R8G8B8A8_UNORM = 28, DXGI_FORMAT_R8G8B8A8_UNORM_SRGB = 29,
    DXGI_FORMAT_R8G8B8A8_SNORM = 31, DXGI_FORMAT_R16G16_FLOAT = 34, DXGI_FORMAT_R16G16_UNORM = 35, DXGI_FORMAT_R32_FLOAT = 41,
    DXGI_FORMAT_R8G8_UNORM = 49, DXGI_FORMAT_R8G8_SNORM = 51, DXGI_FORMAT_R16_FLOAT = 54, DXGI_FORMAT_R16_UNORM = 56,
    DXGI_FORMAT_R8_UNORM = 61, DXGI_FORMAT_R8_SNORM = 63, DXGI_FORMAT_A8_UNORM = 65,
    DXGI_FORMAT_BC1_UNORM = 71, DXGI_FORMAT_BC1_UNORM_SRGB = 72, DXGI_FORMAT_BC2_UNORM = 74, DXGI_FORMAT_BC2_UNORM_SRGB = 75,
    DXGI_FORMAT_BC3_UNORM = 77, DXGI_FORMAT_BC3_UNORM_SRGB = 78, DXGI_FORMAT_BC4_UNORM = 80, DXGI_FORMAT_BC4_SNORM = 81,
    DXGI_FORMAT_BC5_UNORM = 83, DXGI_FORMAT_BC5_SNORM = 84,
    DXGI_FORMAT_B8G8R8A8_UNORM = 87, DXGI_FORMAT_B8G8R8X8_UNORM = 88, DXGI_FORMAT_B8G8R8A8_UNORM_SRGB = 91, DXGI_FORMAT_B8G8R8X8_UNORM_SRGB = 93,
    DXGI_FORMAT_BC6H_UF16 = 95, DXGI_FORMAT_BC6H_SF16 = 96, DXGI_FORMAT_BC7_UNORM = 98, DXGI_FORMAT_BC7_UNORM_SRGB = 99,
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

bool IsCompressed(DXGI_FORMAT fmt) noexcept;
size_t BitsPerPixel(DXGI_FORMAT fmt) noexcept;
// ComputePitch with CP_FLAGS_NONE (DirectXTexUtil.cpp:961-1186)
HRESULT ComputePitch(DXGI_FORMAT fmt, size_t width, size_t height, size_t& rowPitch, size_t& slicePitch) noexcept;
// CalculateMipLevels (DirectXTexMipmaps.cpp:40-69): mipLevels == 0 asks for the full chain
bool CalculateMipLevels(size_t width, size_t height, size_t& mipLevels) noexcept;
bool CalculateMipLevels3D(size_t width, size_t height, size_t depth, size_t& mipLevels) noexcept;

struct TexMetadata
{
    size_t width = 0, height = 0, depth = 0, arraySize = 0, mipLevels = 0;
    uint32_t miscFlags = 0, miscFlags2 = 0;
    DXGI_FORMAT format = DXGI_FORMAT_UNKNOWN;
    TEX_DIMENSION dimension = TEX_DIMENSION_TEXTURE2D;
    // index = item * mipLevels + mip for 1D / 2D textures (DirectXTexUtil.cpp:1695-1740)
    size_t ComputeIndex(size_t mip, size_t item, size_t slice) const noexcept;
    bool IsVolumemap() const noexcept { return dimension == TEX_DIMENSION_TEXTURE3D; }
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

    HRESULT Initialize(const TexMetadata& mdata) noexcept;
    HRESULT Initialize2D(DXGI_FORMAT fmt, size_t width, size_t height, size_t arraySize, size_t mipLevels) noexcept;
    HRESULT Initialize3D(DXGI_FORMAT fmt, size_t width, size_t height, size_t depth, size_t mipLevels) noexcept;
    HRESULT InitializeFromImage(const Image& srcImage) noexcept;      // copies the pixels
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
    explicit operator bool() const noexcept { return m_ctx != nullptr; }
    dxtex_ctx* Get() const noexcept { return m_ctx; }
    const char* LastError() const noexcept;
private:
    dxtex_ctx* m_ctx = nullptr;
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
} // namespace DirectXTexAMD

// ---- DDS container: the on-disk format either side of the path ---------------------------------
// Subset of DirectXTexDDS.cpp: 1D/2D textures, arrays and cubemaps of the formats this library handles; legacy (DX9)
// pixel formats are read when they map 1:1 onto one of those formats (no expansion / swizzling), and written exactly
// where the reference writes them (EncodeDDSHeader, DirectXTexDDS.cpp:711-1033). File names are UTF-8 char strings.
namespace DirectXTexAMD
{
enum DDS_FLAGS : uint32_t { DDS_FLAGS_NONE = 0x0, DDS_FLAGS_FORCE_DX10_EXT = 0x10000, DDS_FLAGS_FORCE_DX10_EXT_MISC2 = 0x20000 };
enum TEX_MISC_FLAG : uint32_t { TEX_MISC_TEXTURECUBE = 0x4 };

class Blob
{
public:
    Blob() noexcept = default;
    ~Blob() { Release(); }
    Blob(const Blob&) = delete;
    Blob& operator=(const Blob&) = delete;
    HRESULT Initialize(size_t size) noexcept;
    void Release() noexcept;
    uint8_t* GetBufferPointer() const noexcept { return m_buffer; }
    size_t GetBufferSize() const noexcept { return m_size; }
private:
    uint8_t* m_buffer = nullptr;
    size_t m_size = 0;
};

HRESULT GetMetadataFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata& metadata) noexcept;
HRESULT LoadFromDDSMemory(const void* pSource, size_t size, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT LoadFromDDSFile(const char* szFile, DDS_FLAGS flags, TexMetadata* metadata, ScratchImage& image) noexcept;
HRESULT SaveToDDSMemory(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, Blob& blob) noexcept;
HRESULT SaveToDDSFile(const Image* images, size_t nimages, const TexMetadata& metadata, DDS_FLAGS flags, const char* szFile) noexcept;
} // namespace DirectXTexAMD
