// C ABI of libdxtex_amd.so (include/dxtex_amd.h): context, validation with the reference's HRESULTs,
// staging for the host-pointer variants, and kernel submission. There is no CPU compute path here: every
// entry point either launches HIP kernels on gfx950 or fails.
#include "../../include/dxtex_amd.h"
#include "dxtex_formats.h"
#include "dxtex_kernels.h"
#include "dxtex_plan.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace dxtex;

struct dxtex_ctx;
namespace
{
// Per-kernel device timing (dxtex_ctx_profile_*): an event before every kernel and one after the last.
struct Marks final : dxtex::KernelMarks
{
    dxtex_ctx* ctx = nullptr;
    std::vector<hipEvent_t> pool;        // reused across calls
    std::vector<const char*> names;      // names[i] labels the interval events[i] -> events[i+1]; nullptr = end of a call
    size_t used = 0;
    void mark(const char* kernelName) override;
    void reset() { names.clear(); used = 0; }
};
}

struct dxtex_ctx
{
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    float lastKernelMs = -1.0f;
    bool timing = false;
    // grow-only device staging for the host-pointer entry points
    void* stageIn = nullptr; size_t stageInBytes = 0;
    void* stageOut = nullptr; size_t stageOutBytes = 0;
    // grow-only device scratch for the multi-kernel BC6H/BC7 search (per-mode candidates)
    void* scratch = nullptr; size_t scratchBytes = 0;
    std::string lastError;
    bool profiling = false;
    Marks marks;
};

void Marks::mark(const char* kernelName)
{
    if (used == pool.size())
    {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        pool.push_back(e);
    }
    (void)hipEventRecord(pool[used++], ctx->stream);
    names.push_back(kernelName);
}

namespace
{
dxtex_hresult fail(dxtex_ctx* ctx, dxtex_hresult hr, const char* what, hipError_t e = hipSuccess)
{
    if (ctx)
    {
        ctx->lastError = what;
        if (e != hipSuccess) { ctx->lastError += ": "; ctx->lastError += hipGetErrorString(e); }
    }
    return hr;
}

#define HIP_TRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(ctx, (e_ == hipErrorOutOfMemory) ? DXTEX_E_OUTOFMEMORY : DXTEX_E_FAIL, #expr, e_); } while (0)

dxtex_hresult ensure(dxtex_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (*have >= need) return DXTEX_S_OK;
    if (*buf) { HIP_TRY(ctx, hipFree(*buf)); *buf = nullptr; *have = 0; }
    const size_t bytes = std::max<size_t>(need, 1u << 20);
    HIP_TRY(ctx, hipMalloc(buf, bytes));
    *have = bytes;
    return DXTEX_S_OK;
}

struct ScopedDevice
{
    int prev = -1;
    explicit ScopedDevice(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~ScopedDevice() { if (prev >= 0) (void)hipSetDevice(prev); }
};

void time_begin(dxtex_ctx* ctx) { (void)hipEventRecord(ctx->evStart, ctx->stream); }
void time_end(dxtex_ctx* ctx) { (void)hipEventRecord(ctx->evStop, ctx->stream); ctx->timing = true; }

// The part of ConvertScanline that Compress reaches (DirectXTexConvert.cpp:3080-3854), resolved once
// per image on the host into the (tcv, tsw) pair the tile loader applies.
dxtex_hresult tile_conversion(const FmtInfo& in, const FmtInfo& out, uint32_t compressFlags, int* tcv, int* tsw)
{
    bool srgbIn = (compressFlags & DXTEX_COMPRESS_SRGB_IN) != 0 || (in.cls & FC_SRGB);
    bool srgbOut = (compressFlags & DXTEX_COMPRESS_SRGB_OUT) != 0 || (out.cls & FC_SRGB);
    if (in.format == FMT_A8_UNORM) srgbIn = false;
    if (srgbIn != srgbOut)
        return DXTEX_E_NOT_SUPPORTED;   // one-sided sRGB needs the pow() path (XMColorSRGBToRGB): not implemented

    *tcv = TCV_NONE; *tsw = TSW_NONE;
    if (out.cls & FC_UNORM)
    {
        if (in.cls & FC_SNORM) *tcv = TCV_SNORM_TO_UNORM;
        else if (in.cls & FC_FLOAT) *tcv = TCV_SATURATE;
    }
    else if (out.cls & FC_SNORM)
    {
        if (in.cls & FC_UNORM) *tcv = TCV_UNORM_TO_SNORM;
        else if (in.cls & FC_FLOAT) *tcv = TCV_CLAMP_SNORM;
    }

    const uint32_t inRGBA = in.cls & (FC_R | FC_G | FC_B | FC_A), outRGB = out.cls & (FC_R | FC_G | FC_B);
    if (inRGBA == FC_A && !(out.cls & FC_A)) *tsw = TSW_A_TO_RGB;
    else if ((in.cls & (FC_R | FC_G | FC_B)) == FC_R)
    {
        if (outRGB == (FC_R | FC_G | FC_B)) *tsw = TSW_R_TO_RGB;
        else if (outRGB == (FC_R | FC_G)) *tsw = TSW_R_TO_RG;
    }
    return DXTEX_S_OK;
}

dxtex_hresult submit_compress(dxtex_ctx* ctx, const uint8_t* dSrc, size_t width, size_t height, int srcFormat, size_t srcRowPitch,
                              uint8_t* dDst, int dstFormat, size_t dstRowPitch, uint32_t flags, float threshold)
{
    const FmtInfo* in = format_info(srcFormat);
    const FmtInfo* out = format_info(dstFormat);
    if (!out || !(out->cls & FC_BC)) return fail(ctx, in && !(in->cls & FC_BC) ? DXTEX_E_NOT_SUPPORTED : DXTEX_E_INVALIDARG, "destination is not a supported BC format");
    if (in && (in->cls & FC_BC)) return fail(ctx, DXTEX_E_INVALIDARG, "source image is already compressed");
    if (!in) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "source format is not supported by the MI355X path");
    if (!width || !height) return fail(ctx, DXTEX_E_INVALIDARG, "empty image");
    if (width > 0xFFFFFFFCull || height > 0xFFFFFFFCull) return fail(ctx, DXTEX_E_INVALIDARG, "image too large");

    SrcView v;
    v.pixels = dSrc; v.width = uint32_t(width); v.height = uint32_t(height); v.rowPitch = srcRowPitch; v.format = srcFormat;
    dxtex_hresult hr = tile_conversion(*in, *out, flags, &v.tcv, &v.tsw);
    if (hr != DXTEX_S_OK) return fail(ctx, hr, "one-sided sRGB conversion is not supported");

    hipError_t e;
    switch (dstFormat)
    {
    case FMT_BC1_UNORM: case FMT_BC1_UNORM_SRGB: case FMT_BC2_UNORM: case FMT_BC2_UNORM_SRGB:
    case FMT_BC3_UNORM: case FMT_BC3_UNORM_SRGB: case FMT_BC4_UNORM: case FMT_BC4_SNORM:
    case FMT_BC5_UNORM: case FMT_BC5_SNORM:
        e = launch_bc15_encode(v, dDst, dstRowPitch, dstFormat, flags, threshold, ctx->stream);
        break;
    case FMT_BC7_UNORM: case FMT_BC7_UNORM_SRGB:
    {
        const uint64_t nblocks = uint64_t((width + 3) / 4) * uint64_t((height + 3) / 4);
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc7_scratch_bytes(nblocks, flags));
        if (hr != DXTEX_S_OK) return hr;
        e = launch_bc7_encode(v, dDst, dstRowPitch, flags, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr);
        break;
    }
    default:
        return fail(ctx, DXTEX_E_NOT_SUPPORTED, "BC format not implemented yet");
    }
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    return DXTEX_S_OK;
}

dxtex_hresult check_pair(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!src || !dst) return fail(ctx, DXTEX_E_INVALIDARG, "null image");
    if (!src->pixels || !dst->pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
    if (src->width != dst->width || src->height != dst->height) return fail(ctx, DXTEX_E_FAIL, "size mismatch");
    return DXTEX_S_OK;
}
} // namespace

extern "C"
{
dxtex_hresult dxtex_ctx_create(int device, dxtex_ctx** out)
{
    if (!out) return DXTEX_E_POINTER;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count)
        return DXTEX_E_FAIL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return DXTEX_E_FAIL;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        std::fprintf(stderr, "dxtex_amd: device %d is %s; this library carries gfx950 code objects only\n", device, prop.gcnArchName);
        return DXTEX_E_FAIL;
    }
    dxtex_ctx* ctx = new (std::nothrow) dxtex_ctx;
    if (!ctx) return DXTEX_E_OUTOFMEMORY;
    ctx->device = device;
    ScopedDevice sd(device);
    if (hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->evStart) != hipSuccess || hipEventCreate(&ctx->evStop) != hipSuccess)
    {
        delete ctx;
        return DXTEX_E_FAIL;
    }
    ctx->stream = ctx->ownStream;
    ctx->marks.ctx = ctx;
    *out = ctx;
    return DXTEX_S_OK;
}

void dxtex_ctx_destroy(dxtex_ctx* ctx)
{
    if (!ctx) return;
    ScopedDevice sd(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stageIn) (void)hipFree(ctx->stageIn);
    if (ctx->stageOut) (void)hipFree(ctx->stageOut);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (hipEvent_t e : ctx->marks.pool) (void)hipEventDestroy(e);
    if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
    if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

dxtex_hresult dxtex_ctx_set_stream(dxtex_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DXTEX_E_POINTER;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->ownStream;
    return DXTEX_S_OK;
}

void* dxtex_ctx_get_stream(dxtex_ctx* ctx) { return ctx ? ctx->stream : nullptr; }

dxtex_hresult dxtex_ctx_synchronize(dxtex_ctx* ctx)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

const char* dxtex_ctx_last_error(dxtex_ctx* ctx) { return ctx ? ctx->lastError.c_str() : "null context"; }

float dxtex_ctx_last_kernel_ms(dxtex_ctx* ctx)
{
    if (!ctx || !ctx->timing) return -1.0f;
    ScopedDevice sd(ctx->device);
    if (hipEventSynchronize(ctx->evStop) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->evStart, ctx->evStop) != hipSuccess) return -1.0f;
    return ms;
}

dxtex_hresult dxtex_ctx_profile_begin(dxtex_ctx* ctx)
{
    if (!ctx) return DXTEX_E_POINTER;
    ctx->marks.reset();
    ctx->profiling = true;
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_ctx_profile_end(dxtex_ctx* ctx, char* names, size_t namesBytes, float* ms, uint32_t* launches, size_t capacity, size_t* count)
{
    if (!ctx || !count) return DXTEX_E_POINTER;
    ctx->profiling = false;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // aggregate by kernel name (pointer identity: names are string literals inside the launchers)
    std::vector<const char*> uniq; std::vector<double> total; std::vector<uint32_t> n;
    Marks& m = ctx->marks;
    for (size_t i = 0; i + 1 < m.used; ++i)
    {
        if (!m.names[i]) continue;
        float t = 0.0f;
        if (hipEventElapsedTime(&t, m.pool[i], m.pool[i + 1]) != hipSuccess) continue;
        size_t k = 0;
        for (; k < uniq.size(); ++k) if (uniq[k] == m.names[i]) break;
        if (k == uniq.size()) { uniq.push_back(m.names[i]); total.push_back(0.0); n.push_back(0); }
        total[k] += t; n[k] += 1;
    }
    *count = uniq.size();
    size_t off = 0;
    for (size_t k = 0; k < uniq.size() && k < capacity; ++k)
    {
        if (ms) ms[k] = float(total[k]);
        if (launches) launches[k] = n[k];
        if (names)
        {
            const size_t len = std::strlen(uniq[k]);
            if (off + len + 1 < namesBytes) { std::memcpy(names + off, uniq[k], len); off += len; names[off++] = '\n'; }
        }
    }
    if (names && namesBytes) names[off < namesBytes ? off : namesBytes - 1] = 0;
    m.reset();
    return DXTEX_S_OK;
}

int dxtex_is_compressed(int32_t format) { return is_bc(format) ? 1 : 0; }

size_t dxtex_bits_per_pixel(int32_t format)
{
    const FmtInfo* f = format_info(format);
    return f ? f->bpp : 0;
}

dxtex_hresult dxtex_compute_pitch(int32_t format, size_t width, size_t height, size_t* rowPitch, size_t* slicePitch)
{
    if (!rowPitch || !slicePitch) return DXTEX_E_POINTER;
    const FmtInfo* f = format_info(format);
    if (!f) return DXTEX_E_INVALIDARG;
    uint64_t pitch, slice;
    if (f->cls & FC_BC)
    {
        // DirectXTexUtil.cpp:972-1029
        const uint64_t nbw = std::max<uint64_t>(1u, (uint64_t(width) + 3u) / 4u);
        const uint64_t nbh = std::max<uint64_t>(1u, (uint64_t(height) + 3u) / 4u);
        pitch = nbw * bc_block_bytes(format);
        slice = pitch * nbh;
    }
    else
    {
        pitch = (uint64_t(width) * f->bpp + 7u) / 8u;   // default byte alignment, :1174-1178
        slice = pitch * uint64_t(height);
    }
    *rowPitch = size_t(pitch); *slicePitch = size_t(slice);
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_compress_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags, float threshold)
{
    dxtex_hresult hr = check_pair(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    time_begin(ctx);
    hr = submit_compress(ctx, src->pixels, src->width, src->height, src->format, src->rowPitch,
                         dst->pixels, dst->format, dst->rowPitch, flags, threshold);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_compress_many_device(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts, size_t count,
                                         uint32_t flags, float threshold)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!srcs || !dsts || !count) return fail(ctx, DXTEX_E_INVALIDARG, "empty batch");
    ScopedDevice sd(ctx->device);
    time_begin(ctx);
    for (size_t i = 0; i < count; ++i)
    {
        dxtex_hresult hr = check_pair(ctx, &srcs[i], &dsts[i]);
        if (hr == DXTEX_S_OK)
            hr = submit_compress(ctx, srcs[i].pixels, srcs[i].width, srcs[i].height, srcs[i].format, srcs[i].rowPitch,
                                 dsts[i].pixels, dsts[i].format, dsts[i].rowPitch, flags, threshold);
        if (hr != DXTEX_S_OK) { time_end(ctx); return hr; }
    }
    time_end(ctx);
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_compress(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags, float threshold)
{
    dxtex_hresult hr = check_pair(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    const FmtInfo* in = format_info(src->format);
    const FmtInfo* out = format_info(dst->format);
    if (in && (in->cls & FC_BC)) return fail(ctx, DXTEX_E_INVALIDARG, "source image is already compressed");
    if (!out || !(out->cls & FC_BC)) return fail(ctx, DXTEX_E_INVALIDARG, "destination is not a BC format");
    if (!in) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "source format is not supported by the MI355X path");

    ScopedDevice sd(ctx->device);
    const size_t srcBytes = src->rowPitch * src->height;
    const size_t nbh = std::max<size_t>(1, (src->height + 3) / 4);
    const size_t dstBytes = dst->rowPitch * nbh;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hr = submit_compress(ctx, static_cast<const uint8_t*>(ctx->stageIn), src->width, src->height, src->format, src->rowPitch,
                         static_cast<uint8_t*>(ctx->stageOut), dst->format, dst->rowPitch, flags, threshold);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_encode_blocks(dxtex_ctx* ctx, int32_t bc_format, uint32_t bc_flags, float threshold,
                                  const float* rgba, size_t nblocks, uint8_t* bc)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!rgba || !bc) return fail(ctx, DXTEX_E_POINTER, "null buffer");
    const size_t bb = bc_block_bytes(bc_format);
    if (!bb) return fail(ctx, DXTEX_E_INVALIDARG, "not a BC format");
    if (!nblocks) return DXTEX_S_OK;
    ScopedDevice sd(ctx->device);
    // nblocks tiles of 16 x float4 == an R32G32B32A32_FLOAT image 4 texels wide and 4*nblocks high.
    const size_t srcBytes = nblocks * 256, dstBytes = nblocks * bb;
    dxtex_hresult hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stageIn, rgba, srcBytes, hipMemcpyHostToDevice, ctx->stream));

    SrcView v;
    v.pixels = static_cast<const uint8_t*>(ctx->stageIn); v.width = 4; v.height = uint32_t(nblocks * 4);
    v.rowPitch = 64; v.format = FMT_R32G32B32A32_FLOAT; v.tcv = TCV_NONE; v.tsw = TSW_NONE;   // raw floats, as BC_ENCODE receives them
    hipError_t e;
    if (bc_format == FMT_BC7_UNORM || bc_format == FMT_BC7_UNORM_SRGB)
    {
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc7_scratch_bytes(nblocks, bc_flags));
        if (hr != DXTEX_S_OK) return hr;
    }
    time_begin(ctx);
    switch (bc_format)
    {
    case FMT_BC6H_UF16: case FMT_BC6H_SF16:
        time_end(ctx);
        return fail(ctx, DXTEX_E_NOT_SUPPORTED, "BC format not implemented yet");
    case FMT_BC7_UNORM: case FMT_BC7_UNORM_SRGB:
        e = launch_bc7_encode(v, static_cast<uint8_t*>(ctx->stageOut), bb, bc_flags, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr);
        break;
    default:
        e = launch_bc15_encode(v, static_cast<uint8_t*>(ctx->stageOut), bb, bc_format, bc_flags, threshold, ctx->stream);
        break;
    }
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    HIP_TRY(ctx, hipMemcpyAsync(bc, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

// DecompressBC (DirectXTexCompress.cpp:425-535): BC image -> uncompressed image of the same size, on device pointers.
static dxtex_hresult submit_decompress(dxtex_ctx* ctx, const uint8_t* dSrc, int srcFormat, size_t srcRowPitch,
                                       uint8_t* dDst, int dstFormat, size_t dstRowPitch, size_t width, size_t height)
{
    const FmtInfo* in = format_info(srcFormat);
    const FmtInfo* out = format_info(dstFormat);
    if (!in || !(in->cls & FC_BC)) return fail(ctx, DXTEX_E_INVALIDARG, "source image is not block compressed");
    if (out && (out->cls & FC_BC)) return fail(ctx, DXTEX_E_INVALIDARG, "destination format is block compressed");
    if (!out) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "destination format is not supported by the MI355X path");
    if (!width || !height) return fail(ctx, DXTEX_E_INVALIDARG, "empty image");
    const ConvertPlan plan = resolve_convert_plan(*in, *out, 0);
    hipError_t e = launch_bc_decode(dSrc, srcRowPitch, srcFormat, dDst, dstRowPitch, dstFormat, uint32_t(width), uint32_t(height), plan, ctx->stream);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_decompress_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst)
{
    dxtex_hresult hr = check_pair(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    time_begin(ctx);
    hr = submit_decompress(ctx, src->pixels, src->format, src->rowPitch, dst->pixels, dst->format, dst->rowPitch, src->width, src->height);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_decompress(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst)
{
    dxtex_hresult hr = check_pair(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    const size_t nbh = std::max<size_t>(1, (src->height + 3) / 4);
    const size_t srcBytes = src->rowPitch * nbh, dstBytes = dst->rowPitch * dst->height;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hr = submit_decompress(ctx, static_cast<const uint8_t*>(ctx->stageIn), src->format, src->rowPitch,
                           static_cast<uint8_t*>(ctx->stageOut), dst->format, dst->rowPitch, src->width, src->height);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_decode_blocks(dxtex_ctx* ctx, int32_t bc_format, const uint8_t* bc, size_t nblocks, float* rgba)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!rgba || !bc) return fail(ctx, DXTEX_E_POINTER, "null buffer");
    const size_t bb = bc_block_bytes(bc_format);
    if (!bb) return fail(ctx, DXTEX_E_INVALIDARG, "not a BC format");
    if (!nblocks) return DXTEX_S_OK;
    ScopedDevice sd(ctx->device);
    // nblocks blocks == a BC image 4 texels wide and 4*nblocks high; the raw decoder output is R32G32B32A32_FLOAT
    const size_t srcBytes = nblocks * bb, dstBytes = nblocks * 256;
    dxtex_hresult hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->stageIn, bc, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    ConvertPlan plan; plan.srgbIn = 0; plan.tcv = TCV_NONE; plan.tsw = TSW_NONE; plan.srgbOut = 0;
    time_begin(ctx);
    hipError_t e = launch_bc_decode(static_cast<const uint8_t*>(ctx->stageIn), bb, bc_format, static_cast<uint8_t*>(ctx->stageOut), 64,
                                    FMT_R32G32B32A32_FLOAT, 4, uint32_t(nblocks * 4), plan, ctx->stream);
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    HIP_TRY(ctx, hipMemcpyAsync(rgba, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_generate_mips(dxtex_ctx* ctx, const dxtex_image*, size_t, uint32_t)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_generate_mips: not implemented yet"); }
dxtex_hresult dxtex_generate_mips_device(dxtex_ctx* ctx, const dxtex_image*, size_t, uint32_t)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_generate_mips_device: not implemented yet"); }
dxtex_hresult dxtex_convert(dxtex_ctx* ctx, const dxtex_image*, const dxtex_image*, uint32_t, float)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_convert: not implemented yet"); }
dxtex_hresult dxtex_convert_device(dxtex_ctx* ctx, const dxtex_image*, const dxtex_image*, uint32_t, float)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_convert_device: not implemented yet"); }
dxtex_hresult dxtex_resize(dxtex_ctx* ctx, const dxtex_image*, const dxtex_image*, uint32_t)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_resize: not implemented yet"); }
dxtex_hresult dxtex_resize_device(dxtex_ctx* ctx, const dxtex_image*, const dxtex_image*, uint32_t)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_resize_device: not implemented yet"); }
dxtex_hresult dxtex_compute_mse_device(dxtex_ctx* ctx, const dxtex_image*, const dxtex_image*, double*)
{ return fail(ctx, DXTEX_E_NOTIMPL, "dxtex_compute_mse_device: not implemented yet"); }

dxtex_hresult dxtex_device_alloc(dxtex_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipMalloc(out, bytes ? bytes : 1));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_device_free(dxtex_ctx* ctx, void* p)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipFree(p));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_memcpy_h2d(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_memcpy_d2h(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
} // extern "C"
