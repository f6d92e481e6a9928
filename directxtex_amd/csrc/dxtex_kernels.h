// Internal launcher prototypes (one per .hip translation unit). Everything is stream-ordered and
// returns the launch status; no launcher synchronises or allocates.
#pragma once
#include "dxtex_device.h"

namespace dxtex
{
hipError_t launch_bc15_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, int dstFormat,
                              uint32_t flags, float threshold, hipStream_t stream);
} // namespace dxtex
