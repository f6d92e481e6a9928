// Internal launcher prototypes (one per .hip translation unit). Everything is stream-ordered and
// returns the launch status; no launcher synchronises or allocates.
#pragma once
#include "dxtex_device.h"

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

// BC7: `scratch` must hold bc7_scratch_bytes(number of 4x4 blocks, flags) bytes of device memory.
size_t bc7_scratch_bytes(uint64_t nblocks, uint32_t flags);
hipError_t launch_bc7_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, uint32_t flags,
                             void* scratch, hipStream_t stream, KernelMarks* marks);

// BC -> uncompressed (DecompressBC). `plan` = resolve_convert_plan(bc format, target format, TEX_FILTER_DEFAULT).
struct ConvertPlan;
hipError_t launch_bc_decode(const uint8_t* src, uint64_t srcRowPitch, int srcFormat, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                            uint32_t width, uint32_t height, const ConvertPlan& plan, hipStream_t stream);
} // namespace dxtex
