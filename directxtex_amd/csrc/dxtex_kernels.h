// Internal launcher prototypes (one per .hip translation unit). Everything is stream-ordered and
// returns the launch status; no launcher synchronises or allocates.
#pragma once
#include "dxtex_device.h"
#include "dxtex_dev.h"

namespace dxtex
{
// Optional per-kernel timing hook: launchers call mark(name) immediately before each kernel they enqueue;
// the context turns consecutive marks into hipEvent pairs on the launch stream.
struct KernelMarks
{
    virtual void mark(const char* kernelName) = 0;
protected:
    ~KernelMarks() = default;
};

hipError_t launch_bc15_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                              uint32_t flags, float threshold, hipStream_t stream);

// BC7: `scratch` must hold bc7_scratch_bytes(total number of 4x4 blocks, flags, number of images) bytes of device memory.
// The _many form runs an array of images (a mip chain, a texture array) through the per-mode pipeline as one block list.
struct BcImage { SrcView src; uint8_t* dst; uint64_t dstRowPitch; };
// BC1-BC5: up to bc15_small_batch_max() images that satisfy bc15_small_image() (at most 256 x 256 texels) in ONE launch - the tail
// of a mip chain, whose levels would otherwise each cost the latency of a kernel's slowest block.
bool bc15_small_image(uint32_t width, uint32_t height);
int bc15_small_batch_max();
hipError_t launch_bc15_encode_small(const BcImage* images, int count, int dstFormat, uint32_t flags, float threshold, hipStream_t stream);
// Two extra streams (and the events that fork / join them) for pipelines that are independent of each other until the last kernel:
// owned by the context (one set per context and device, destroyed with it); nullptr = everything on `stream`.
constexpr int kSideStreams = 3;
struct SideStreams { hipStream_t side[kSideStreams]; hipEvent_t forked; hipEvent_t joined[kSideStreams]; };
size_t bc7_scratch_bytes(uint64_t nblocks, uint32_t flags, size_t nimages = 1);
hipError_t launch_bc7_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, uint32_t flags,
                             void* scratch, hipStream_t stream, KernelMarks* marks, const SideStreams* side = nullptr);
hipError_t launch_bc7_encode_many(const BcImage* images, size_t count, uint32_t flags, void* scratch, hipStream_t stream, KernelMarks* marks,
                                  const SideStreams* side = nullptr);

// BC6H (UF16 / SF16): `scratch` must hold bc6h_scratch_bytes(number of 4x4 blocks) bytes of device memory.
size_t bc6h_scratch_bytes(uint64_t nblocks, size_t nimages = 1);
hipError_t launch_bc6h_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, bool isSigned, void* scratch,
                              hipStream_t stream, KernelMarks* marks, const SideStreams* side = nullptr);
hipError_t launch_bc6h_encode_many(const BcImage* images, size_t count, bool isSigned, void* scratch, hipStream_t stream, KernelMarks* marks,
                                   const SideStreams* side = nullptr);

// BC -> uncompressed (DecompressBC). `plan` = resolve_convert_plan(bc format, target format, TEX_FILTER_DEFAULT).
struct ConvertPlan;
hipError_t launch_bc_decode(const uint8_t* src, uint64_t srcRowPitch, int srcFormat, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                            uint32_t width, uint32_t height, const ConvertPlan& plan, hipStream_t stream);

// Convert (ConvertCustom without dithering): same size, different format.
hipError_t launch_convert(const uint8_t* src, uint64_t srcPitch, int srcFormat, uint8_t* dst, uint64_t dstPitch, int dstFormat,
                          uint32_t width, uint32_t height, const ConvertPlan& plan, float threshold, hipStream_t stream);

// Resize / one mip level. filterMode = TEX_FILTER_POINT..TRIANGLE (already resolved, never 0); filterFlags carries the
// wrap / mirror / sRGB bits. `tri` (device pointers) is required for TEX_FILTER_TRIANGLE: per destination column / row
// ofs[i]..ofs[i+1] indexes (source index, fp32 weight bits) pairs, see triangle_filter.h. `staleLevel` (box mips only):
// the last level of the chain that was 2 texels high, see resize_box_kernel.
struct TriangleTables { const uint32_t* ofsX; const void* entX; const uint32_t* ofsY; const void* entY; };
hipError_t launch_resize(const uint8_t* src, uint64_t srcPitch, uint32_t srcW, uint32_t srcH, uint8_t* dst, uint64_t dstPitch,
                         uint32_t dstW, uint32_t dstH, int format, uint32_t filterMode, uint32_t filterFlags, bool mipAlias,
                         const TriangleTables* tri, hipStream_t stream,
                         const uint8_t* staleLevel = nullptr, uint64_t stalePitch = 0, uint32_t staleW = 0, int dstFormat = -1);
// dstFormat >= 0: the destination rows are written in that format instead of `format` (R32G32B32A32_FLOAT rows for launch_pack_group).

// Formats whose element holds several texels (FC_GROUP): R32G32B32A32_FLOAT rows -> the format, with StoreScanline's pair / bit packing.
hipError_t launch_pack_group(const uint8_t* rows, uint64_t rowsPitch, uint8_t* dst, uint64_t dstPitch, int dstFormat, uint32_t width, uint32_t height, hipStream_t stream);

// The tail of a 2-D mip chain (levels[0] = the first source level, at most 64 x 64; levels[1..] = the levels generated from it) in one
// workgroup: point / linear / cubic / box, the arithmetic of launch_resize with mipAlias. twoHigh = the last level of the chain before
// levels[0] that was at least 2 texels high (the box filter's stale tap, see resize_box_kernel), or nullptr.
struct MipLevel { uint8_t* pixels; uint64_t pitch; uint32_t width, height; };
bool resize_tail_applies(uint32_t srcW, uint32_t srcH, uint32_t filterMode);
// cubic: only the tail of a power-of-two RGBA8 chain with clamp addressing (every level an exact halving) has a one-workgroup form;
// levels[] as for launch_resize_tail
bool resize_cubic_tail_applies(const MipLevel* levels, int nlevels, int format, uint32_t filterFlags);
hipError_t launch_resize_tail(const MipLevel* levels, int nlevels, int format, uint32_t filterMode, uint32_t filterFlags,
                              const MipLevel* twoHigh, hipStream_t stream);

// Volume mips (Generate3DMips*Filter): one level whose source is more than one slice deep. Slices of a level are `slicePitch` apart.
struct VolumeView { const uint8_t* pixels; uint64_t rowPitch, slicePitch; uint32_t width, height, depth; int format; };
struct TriangleTables3 { const uint32_t* ofsX; const void* entX; const uint32_t* ofsY; const void* entY; const uint32_t* ofsZ; const void* entZ; };
hipError_t launch_resize3d(const VolumeView& src, const VolumeView& dst, uint32_t filterMode, uint32_t filterFlags, const TriangleTables3* tri,
                           hipStream_t stream, const uint8_t* staleU = nullptr, const uint8_t* staleV = nullptr, uint64_t stalePitch = 0, uint32_t staleW = 0);

// ComputeMSE: out4 (device) receives the per-channel SUM of squared differences; divide by width * height on the host.
hipError_t launch_mse(const uint8_t* a, uint64_t aPitch, int aFormat, const uint8_t* b, uint64_t bPitch, int bFormat,
                      uint32_t width, uint32_t height, double* out4, hipStream_t stream);
// PremultiplyAlpha / DemultiplyAlpha (DirectXTexPMAlpha.cpp:30-205); pmFlags = TEX_PMALPHA_*
hipError_t launch_pmalpha(const uint8_t* src, uint64_t srcPitch, uint8_t* dst, uint64_t dstPitch, int format, uint32_t width, uint32_t height,
                          uint32_t pmFlags, hipStream_t stream);
// ScaleAlpha and CalculateAlphaCoverage (DirectXTexMipmaps.cpp:143-305); *count receives the number of covered sub-samples
hipError_t launch_scale_alpha(const uint8_t* src, uint64_t srcPitch, uint8_t* dst, uint64_t dstPitch, int format, uint32_t width, uint32_t height,
                              float scale, hipStream_t stream);
hipError_t launch_alpha_coverage(const uint8_t* src, uint64_t srcPitch, int format, uint32_t width, uint32_t height, float scale, float alphaReference,
                                 unsigned long long* count, hipStream_t stream);
// IsAlphaAllOpaque's scan: *count (device, NOT cleared here) is incremented by the number of texels with alpha < threshold
hipError_t launch_alpha_below(const uint8_t* src, uint64_t srcPitch, int format, uint32_t width, uint32_t height, float threshold,
                              unsigned long long* count, hipStream_t stream);
} // namespace dxtex
