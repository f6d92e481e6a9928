// C ABI of libdxtex_amd.so (include/dxtex_amd.h): context, validation with the reference's HRESULTs,
// staging for the host-pointer variants, and kernel submission. There is no CPU compute path here: every
// entry point either launches HIP kernels on gfx950 or fails.
#include "../../include/dxtex_amd.h"
#include "dxtex_formats.h"
#include "dxtex_kernels.h"
#include "dxtex_plan.h"
#include "triangle_filter.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <new>
#include <string>
#include <thread>
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
    // triangle-filter gather tables (host copies stay alive until the next call: the upload is stream-ordered)
    void* triBuf = nullptr; size_t triBytes = 0;
    std::vector<uint8_t> triHost;
    void* triPinned = nullptr; size_t triPinnedBytes = 0; hipEvent_t triConsumed = nullptr; bool triPending = false;
    void* mseBuf = nullptr; size_t mseBytes = 0;
    // R32G32B32A32_FLOAT rows on their way into a format whose element holds several texels (launch_pack_group)
    void* groupRows = nullptr; size_t groupRowsBytes = 0;
    // dxtex_compress_many (host pointers): double-buffered pinned + device staging, copy streams on either side of ctx->stream
    struct Lane
    {
        void* pinIn = nullptr; size_t pinInBytes = 0; void* pinOut = nullptr; size_t pinOutBytes = 0;
        void* devIn = nullptr; size_t devInBytes = 0; void* devOut = nullptr; size_t devOutBytes = 0;
        hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
    } lane[2];
    hipStream_t h2d = nullptr, d2h = nullptr;
    // side streams of the BC7 pipeline (modes 4 / 5 run next to each other): created on first use, destroyed with the context
    SideStreams side = {};
    bool sideTried = false, sideOk = false;
    uint64_t warmedFormats = 0;        // dxtex_ctx_prepare: destination BC formats (bit = format - 64) whose pipeline has run once on this context
    std::string lastError;
    bool profiling = false;
    Marks marks;
    // host <-> device bytes moved for this context (dxtex_ctx_transfer_bytes)
    uint64_t h2dBytes = 0, d2hBytes = 0;
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

// every host <-> device copy of the library goes through here so that dxtex_ctx_transfer_bytes can account for it
inline hipError_t counted_copy(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream)
{
    if (kind == hipMemcpyHostToDevice) ctx->h2dBytes += bytes;
    else if (kind == hipMemcpyDeviceToHost) ctx->d2hBytes += bytes;
    return hipMemcpyAsync(dst, src, bytes, kind, stream);
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

// lazily creates the context's side streams; nullptr (= serial pipelines) if the runtime refuses
const SideStreams* side_streams(dxtex_ctx* ctx)
{
    if (!ctx->sideTried)
    {
        ctx->sideTried = true;
        bool ok = hipEventCreateWithFlags(&ctx->side.forked, hipEventDisableTiming) == hipSuccess;
        for (int k = 0; k < kSideStreams && ok; ++k)
            ok = hipStreamCreateWithFlags(&ctx->side.side[k], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&ctx->side.joined[k], hipEventDisableTiming) == hipSuccess;
        ctx->sideOk = ok;
    }
    return ctx->sideOk ? &ctx->side : nullptr;
}

static const bool kNoTiming = dev_env("DXTEX_NO_TIMING") != nullptr;
void time_begin(dxtex_ctx* ctx) { if (!kNoTiming) (void)hipEventRecord(ctx->evStart, ctx->stream); }
void time_end(dxtex_ctx* ctx) { if (!kNoTiming) { (void)hipEventRecord(ctx->evStop, ctx->stream); ctx->timing = true; } }

// The part of ConvertScanline that Compress reaches (DirectXTexConvert.cpp:3080-3854), resolved once
// per image on the host into the (tcv, tsw) pair the tile loader applies.
dxtex_hresult tile_conversion(const FmtInfo& in, const FmtInfo& out, uint32_t compressFlags, int* tcv, int* tsw)
{
    bool srgbIn = (compressFlags & DXTEX_COMPRESS_SRGB_IN) != 0 || (in.cls & FC_SRGB);
    bool srgbOut = (compressFlags & DXTEX_COMPRESS_SRGB_OUT) != 0 || (out.cls & FC_SRGB);
    if (in.format == FMT_A8_UNORM || in.format == FMT_R10G10B10_XR_BIAS_A2_UNORM) srgbIn = false;      // :3136-3139
    if (srgbIn && srgbOut) srgbIn = srgbOut = false;       // :3164-3167

    *tcv = TCV_NONE; *tsw = TSW_NONE;
    if (in.cls & FC_DEPTH)
    {
        // a depth source: ConvertScanline's depth branch instead of the range conversion (:3186-3291); sRGB does not apply (:3172, :3845 ask
        // for a non-depth format on the side they convert - the BC side still gets its encode step below)
        *tsw = resolve_depth_steps(in, out, 0) << 8;
        srgbIn = false;
    }
    else if (out.cls & FC_UNORM)
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
    if (srgbIn && (in.cls & (FC_FLOAT | FC_UNORM))) *tcv |= TCV_SRGB_TO_LINEAR;      // :3170-3180
    if (srgbOut && (out.cls & (FC_FLOAT | FC_UNORM))) *tcv |= TCV_LINEAR_TO_SRGB;    // :3843-3853
    return DXTEX_S_OK;
}

// Compress' argument checks (DirectXTexCompress.cpp:671-676, :741-745) and the source view the tile loaders take
dxtex_hresult compress_view(dxtex_ctx* ctx, const uint8_t* dSrc, size_t width, size_t height, int srcFormat, size_t srcRowPitch, int dstFormat,
                            uint32_t flags, SrcView* view)
{
    const FmtInfo* in = format_info(srcFormat);
    const FmtInfo* out = format_info(dstFormat);
    // the reference's order (DirectXTexCompress.cpp:671-676): E_INVALIDARG for a compressed source or an uncompressed target first,
    // HRESULT_E_NOT_SUPPORTED for formats the path cannot take (typeless / planar / palettised there; anything without kernels here)
    const auto bcId = [](int f) { return (f >= 70 && f <= 84) || (f >= 94 && f <= 99); };      // IsCompressed: BC1_TYPELESS .. BC5_SNORM, BC6H_TYPELESS .. BC7_UNORM_SRGB
    if (bcId(srcFormat)) return fail(ctx, DXTEX_E_INVALIDARG, "source image is already compressed");
    if (!bcId(dstFormat)) return fail(ctx, DXTEX_E_INVALIDARG, "destination is not a BC format");
    if (!out || !(out->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "destination BC format is not supported (typeless)");
    if (!in) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "source format is not supported by the MI355X path");
    // R1_UNORM: the reference refuses it (DirectXTexCompress.cpp:228-232, "we don't support compressing from monochrome"). The packed
    // two-texel formats: CompressBC steps through a row with BitsPerPixel / 8 bytes per texel (:224-235, :279), which for them is the
    // size of an ELEMENT of two texels (DirectXTexUtil.cpp:625-670) - block column k reads texels 8k.. instead of 4k.. and the right half
    // of the image reads past its rows (past the image on the last block row). Nothing defined to reproduce: refused here.
    if (in->cls & FC_GROUP) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "Compress does not take R1_UNORM or the packed two-texel formats as a source");
    if (!width || !height) return fail(ctx, DXTEX_E_INVALIDARG, "empty image");
    if (width > 0xFFFFFFFCull || height > 0xFFFFFFFCull) return fail(ctx, DXTEX_E_INVALIDARG, "image too large");
    SrcView v;
    v.pixels = dSrc; v.width = uint32_t(width); v.height = uint32_t(height); v.rowPitch = srcRowPitch; v.format = srcFormat;
    const dxtex_hresult hr = tile_conversion(*in, *out, flags, &v.tcv, &v.tsw);
    if (hr != DXTEX_S_OK) return fail(ctx, hr, "unsupported tile conversion");
    *view = v;
    return DXTEX_S_OK;
}

dxtex_hresult submit_compress(dxtex_ctx* ctx, const uint8_t* dSrc, size_t width, size_t height, int srcFormat, size_t srcRowPitch,
                              uint8_t* dDst, int dstFormat, size_t dstRowPitch, uint32_t flags, float threshold)
{
    SrcView v;
    dxtex_hresult hr = compress_view(ctx, dSrc, width, height, srcFormat, srcRowPitch, dstFormat, flags, &v);
    if (hr != DXTEX_S_OK) return hr;

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
        e = launch_bc7_encode(v, dDst, dstRowPitch, flags, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx));
        break;
    }
    case FMT_BC6H_UF16: case FMT_BC6H_SF16:
    {
        const uint64_t nblocks = uint64_t((width + 3) / 4) * uint64_t((height + 3) / 4);
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc6h_scratch_bytes(nblocks));
        if (hr != DXTEX_S_OK) return hr;
        e = launch_bc6h_encode(v, dDst, dstRowPitch, dstFormat == FMT_BC6H_SF16, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx));
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
    if (ctx->groupRows) (void)hipFree(ctx->groupRows);
    if (ctx->triBuf) (void)hipFree(ctx->triBuf);
    if (ctx->triPinned) (void)hipHostFree(ctx->triPinned);
    if (ctx->triConsumed) (void)hipEventDestroy(ctx->triConsumed);
    if (ctx->mseBuf) (void)hipFree(ctx->mseBuf);
    if (ctx->h2d) { (void)hipStreamSynchronize(ctx->h2d); (void)hipStreamDestroy(ctx->h2d); }
    if (ctx->d2h) { (void)hipStreamSynchronize(ctx->d2h); (void)hipStreamDestroy(ctx->d2h); }
    for (dxtex_ctx::Lane& l : ctx->lane)
    {
        if (l.pinIn) (void)hipHostFree(l.pinIn);
        if (l.pinOut) (void)hipHostFree(l.pinOut);
        if (l.devIn) (void)hipFree(l.devIn);
        if (l.devOut) (void)hipFree(l.devOut);
        if (l.uploaded) (void)hipEventDestroy(l.uploaded);
        if (l.computed) (void)hipEventDestroy(l.computed);
        if (l.downloaded) (void)hipEventDestroy(l.downloaded);
    }
    for (int k = 0; k < kSideStreams; ++k)
    {
        if (ctx->side.side[k]) { (void)hipStreamSynchronize(ctx->side.side[k]); (void)hipStreamDestroy(ctx->side.side[k]); }
        if (ctx->side.joined[k]) (void)hipEventDestroy(ctx->side.joined[k]);
    }
    if (ctx->side.forked) (void)hipEventDestroy(ctx->side.forked);
    for (hipEvent_t e : ctx->marks.pool) (void)hipEventDestroy(e);
    if (ctx->evStart) (void)hipEventDestroy(ctx->evStart);
    if (ctx->evStop) (void)hipEventDestroy(ctx->evStop);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

dxtex_hresult dxtex_ctx_set_stream(dxtex_ctx* ctx, void* hip_stream)
{
    if (!ctx) return DXTEX_E_POINTER;
    hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->ownStream;
    if (next != ctx->stream)
    {
        // the context's scratch, staging and filter tables are ordered by ONE stream: work queued on the old one must be done
        // before kernels on the new one may reuse them. The previous stream must stay alive until this call returns. A handle the
        // runtime no longer knows (the caller destroyed the stream: nothing left to wait for) is tolerated and the context switches;
        // any other error is an asynchronous failure of work that was queued on the old stream - a faulted kernel, an ECC error -
        // whose output the caller must not trust: it is reported and the context keeps its stream.
        ScopedDevice sd(ctx->device);
        const hipError_t drained = hipStreamSynchronize(ctx->stream);
        if (drained != hipSuccess)
        {
            (void)hipGetLastError();
            if (drained != hipErrorInvalidHandle && drained != hipErrorInvalidResourceHandle && drained != hipErrorContextIsDestroyed)
                return fail(ctx, DXTEX_E_FAIL, "work queued on the previous stream failed", drained);
        }
        ctx->stream = next;
    }
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
    else if (f->cls & FC_PACKED)
    {
        // two texels per element: ((width + 1) >> 1) elements of 4 (R8G8_B8G8, G8R8_G8B8, YUY2) or 8 (Y210, Y216) bytes (DirectXTexUtil.cpp:1031-1052)
        pitch = ((uint64_t(width) + 1u) >> 1) * ((f->bpp == 32) ? 8u : 4u);
        slice = pitch * uint64_t(height);
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
    // BC7 / BC6H arrays go through the per-mode pipeline as one block list (launch_bc7_encode_many / launch_bc6h_encode_many): the pipeline's latency floor and
    // tails are paid once per 2^22 blocks instead of once per image
    bool allBc7 = count > 1, allBc6 = count > 1;
    for (size_t i = 0; i < count; ++i)
    {
        allBc7 = allBc7 && (dsts[i].format == FMT_BC7_UNORM || dsts[i].format == FMT_BC7_UNORM_SRGB);
        allBc6 = allBc6 && dsts[i].format == dsts[0].format && (dsts[i].format == FMT_BC6H_UF16 || dsts[i].format == FMT_BC6H_SF16);
    }
    if (allBc7 || allBc6)
    {
        std::vector<BcImage> batch(count);
        uint64_t nblocks = 0;
        for (size_t i = 0; i < count; ++i)
        {
            dxtex_hresult hr = check_pair(ctx, &srcs[i], &dsts[i]);
            if (hr == DXTEX_S_OK)
                hr = compress_view(ctx, srcs[i].pixels, srcs[i].width, srcs[i].height, srcs[i].format, srcs[i].rowPitch, dsts[i].format, flags, &batch[i].src);
            if (hr != DXTEX_S_OK) return hr;
            batch[i].dst = dsts[i].pixels; batch[i].dstRowPitch = dsts[i].rowPitch;
            nblocks += uint64_t((srcs[i].width + 3) / 4) * uint64_t((srcs[i].height + 3) / 4);
        }
        dxtex_hresult hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, allBc7 ? bc7_scratch_bytes(nblocks, flags, count) : bc6h_scratch_bytes(nblocks, count));
        if (hr != DXTEX_S_OK) return hr;
        time_begin(ctx);
        const hipError_t e = allBc7 ? launch_bc7_encode_many(batch.data(), count, flags, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx))
                                    : launch_bc6h_encode_many(batch.data(), count, dsts[0].format == FMT_BC6H_SF16, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx));
        time_end(ctx);
        if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
        return DXTEX_S_OK;
    }
    time_begin(ctx);
    // BC1-BC5: runs of small images of one target format (the tail of a mip chain) share a launch; everything else goes image by image
    std::vector<BcImage> small;
    int smallFormat = 0;
    auto flush_small = [&]() -> dxtex_hresult
    {
        if (small.empty()) return DXTEX_S_OK;
        const hipError_t e = launch_bc15_encode_small(small.data(), int(small.size()), smallFormat, flags, threshold, ctx->stream);
        small.clear();
        return e == hipSuccess ? DXTEX_S_OK : fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    };
    for (size_t i = 0; i < count; ++i)
    {
        dxtex_hresult hr = check_pair(ctx, &srcs[i], &dsts[i]);
        const FmtInfo* out = (hr == DXTEX_S_OK) ? format_info(dsts[i].format) : nullptr;
        const bool bc15 = out && (out->cls & FC_BC) && bc_block_bytes(dsts[i].format) && dsts[i].format != FMT_BC7_UNORM && dsts[i].format != FMT_BC7_UNORM_SRGB &&
                          dsts[i].format != FMT_BC6H_UF16 && dsts[i].format != FMT_BC6H_SF16;
        if (hr == DXTEX_S_OK && bc15 && bc15_small_image(uint32_t(srcs[i].width), uint32_t(srcs[i].height)) && srcs[i].width <= 0xFFFFFFFFull && srcs[i].height <= 0xFFFFFFFFull)
        {
            if (!small.empty() && (smallFormat != dsts[i].format || int(small.size()) == bc15_small_batch_max())) hr = flush_small();
            BcImage im;
            if (hr == DXTEX_S_OK)
                hr = compress_view(ctx, srcs[i].pixels, srcs[i].width, srcs[i].height, srcs[i].format, srcs[i].rowPitch, dsts[i].format, flags, &im.src);
            if (hr == DXTEX_S_OK) { im.dst = dsts[i].pixels; im.dstRowPitch = dsts[i].rowPitch; small.push_back(im); smallFormat = dsts[i].format; }
        }
        else if (hr == DXTEX_S_OK)
        {
            hr = flush_small();                        // keeps the images in submission order on the stream
            if (hr == DXTEX_S_OK)
                hr = submit_compress(ctx, srcs[i].pixels, srcs[i].width, srcs[i].height, srcs[i].format, srcs[i].rowPitch,
                                     dsts[i].pixels, dsts[i].format, dsts[i].rowPitch, flags, threshold);
        }
        if (hr != DXTEX_S_OK) { time_end(ctx); return hr; }
    }
    const dxtex_hresult hrSmall = flush_small();
    time_end(ctx);
    return hrSmall;
}

// GPUCompressBC::Prepare's role (BCDirectCompute.cpp:203-369): size the context for `count` images of one shape up front.
dxtex_hresult dxtex_ctx_prepare(dxtex_ctx* ctx, size_t width, size_t height, int32_t src_format, int32_t dst_format, uint32_t flags, size_t count,
                                size_t* device_bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!count) return fail(ctx, DXTEX_E_INVALIDARG, "empty batch");
    size_t srcRow = 0, srcSlice = 0, dstRow = 0, dstSlice = 0;
    SrcView v;
    dxtex_hresult hr = compress_view(ctx, nullptr, width, height, src_format, 0, dst_format, flags, &v);           // the format / size checks of dxtex_compress
    if (hr != DXTEX_S_OK) return hr;
    if (dxtex_compute_pitch(src_format, width, height, &srcRow, &srcSlice) != DXTEX_S_OK || dxtex_compute_pitch(dst_format, width, height, &dstRow, &dstSlice) != DXTEX_S_OK)
        return fail(ctx, DXTEX_E_INVALIDARG, "image too large");
    ScopedDevice sd(ctx->device);
    const uint64_t nblocks = uint64_t((width + 3) / 4) * uint64_t((height + 3) / 4) * count;
    if (dst_format == FMT_BC7_UNORM || dst_format == FMT_BC7_UNORM_SRGB)
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc7_scratch_bytes(nblocks, flags, count));
    else if (dst_format == FMT_BC6H_UF16 || dst_format == FMT_BC6H_SF16)
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc6h_scratch_bytes(nblocks, count));
    if (hr != DXTEX_S_OK) return hr;
    // staging as dxtex_compress_many lays it out: every image 256-byte aligned
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, ((srcSlice + 255) & ~size_t(255)) * count); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, ((dstSlice + 255) & ~size_t(255)) * count); if (hr != DXTEX_S_OK) return hr;
    if (device_bytes) *device_bytes = ctx->scratchBytes + ctx->stageInBytes + ctx->stageOutBytes;
    // GPUCompressBC::Prepare also binds the shaders the size needs (BCDirectCompute.cpp:203-369). The counterpart here: the HIP runtime loads a
    // kernel's code object on its first launch, and the BC7 / BC6H pipelines are some forty kernels plus three side streams - 19 ms on the first
    // call of a fresh process against 1.2 ms on the second (tools/cold_probe.py: a 64 x 64 image). One block of zeros goes through the same
    // pipeline now, once per destination format and context, so the first real call finds everything loaded. Nothing is returned from it.
    const uint64_t bit = (dst_format >= 64 && dst_format < 128) ? (uint64_t(1) << (dst_format - 64)) : 0;
    if (bit && !(ctx->warmedFormats & bit))
    {
        const size_t ww = std::min<size_t>(width, 4), wh = std::min<size_t>(height, 4);
        size_t wRow = 0, wSlice = 0, oRow = 0, oSlice = 0;
        if (dxtex_compute_pitch(src_format, ww, wh, &wRow, &wSlice) == DXTEX_S_OK && dxtex_compute_pitch(dst_format, ww, wh, &oRow, &oSlice) == DXTEX_S_OK &&
            wSlice <= ctx->stageInBytes && oSlice <= ctx->stageOutBytes)
        {
            HIP_TRY(ctx, hipMemsetAsync(ctx->stageIn, 0, wSlice, ctx->stream));
            hr = submit_compress(ctx, static_cast<const uint8_t*>(ctx->stageIn), ww, wh, src_format, wRow, static_cast<uint8_t*>(ctx->stageOut), dst_format, oRow, flags, 0.5f);
            if (hr != DXTEX_S_OK) return hr;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->warmedFormats |= bit;
        }
    }
    return DXTEX_S_OK;
}

// Array form of dxtex_compress with host pointers (the cfg5 entry point: DirectXTexCompress.cpp:794-833 loops over the images of
// an array). The array is cut into chunks of about DXTEX_MANY_CHUNK_TEXELS texels (default 32 Mi: eight 2048^2 images, a pass
// size at which the BC6H / BC7 search pipeline runs at its large-image rate); chunk k+1 is gathered into pinned memory and
// uploaded on a copy stream while chunk k is searched on the context's stream, and payloads come back on a second copy stream:
//   host gather -> [h2d] pinIn -> devIn -> [ctx->stream] kernels -> devOut -> [d2h] pinOut -> host scatter
// with two lanes of buffers, so PCIe traffic in both directions overlaps the kernels (for BC1-BC5, where a 4096^2 image is
// 0.1-0.5 ms of kernel time and 1.6 ms of PCIe, it is the copies that overlap each other).
namespace
{
dxtex_hresult ensure_pinned(dxtex_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (*have >= need) return DXTEX_S_OK;
    if (*buf) { HIP_TRY(ctx, hipHostFree(*buf)); *buf = nullptr; *have = 0; }
    const size_t bytes = std::max<size_t>(need, 1u << 20);
    HIP_TRY(ctx, hipHostMalloc(buf, bytes, hipHostMallocDefault));
    *have = bytes;
    return DXTEX_S_OK;
}

struct ManyChunk { size_t first, count, inBytes, outBytes; };

// tight size checks for host-pointer images: a pitch below the format's minimum would make the kernels read or write past the staging
dxtex_hresult check_host_pitches(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t* srcBytes, size_t* dstBytes)
{
    size_t minSrcRow = 0, minSrcSlice = 0, minDstRow = 0, minDstSlice = 0;
    if (dxtex_compute_pitch(src->format, src->width, src->height, &minSrcRow, &minSrcSlice) != DXTEX_S_OK ||
        dxtex_compute_pitch(dst->format, dst->width, dst->height, &minDstRow, &minDstSlice) != DXTEX_S_OK)
        return fail(ctx, DXTEX_E_INVALIDARG, "image too large");
    if (src->rowPitch < minSrcRow || dst->rowPitch < minDstRow) return fail(ctx, DXTEX_E_INVALIDARG, "rowPitch is smaller than the format's minimum (ComputePitch)");
    const size_t srcRows = (minSrcRow && minSrcSlice) ? minSrcSlice / minSrcRow : src->height;
    const size_t dstRows = (minDstRow && minDstSlice) ? minDstSlice / minDstRow : dst->height;
    if (src->rowPitch > SIZE_MAX / std::max<size_t>(1, srcRows) || dst->rowPitch > SIZE_MAX / std::max<size_t>(1, dstRows))
        return fail(ctx, DXTEX_E_INVALIDARG, "rowPitch x rows overflows");
    *srcBytes = src->rowPitch * srcRows;
    *dstBytes = dst->rowPitch * dstRows;
    return DXTEX_S_OK;
}
}

namespace
{
dxtex_hresult compress_many_pipelined(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts, size_t count, uint32_t flags, float threshold);
}

dxtex_hresult dxtex_compress_many(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts, size_t count, uint32_t flags, float threshold)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!srcs || !dsts || !count) return fail(ctx, DXTEX_E_INVALIDARG, "empty batch");
    ScopedDevice sd(ctx->device);
    const dxtex_hresult hr = compress_many_pipelined(ctx, srcs, dsts, count, flags, threshold);
    if (hr != DXTEX_S_OK)
    {
        // a failure in the middle leaves copies and kernels of earlier chunks in flight: let them finish before the caller may free
        // or reuse its images (the staging they touch belongs to the context)
        const std::string why = ctx->lastError;
        if (ctx->h2d) (void)hipStreamSynchronize(ctx->h2d);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->d2h) (void)hipStreamSynchronize(ctx->d2h);
        ctx->lastError = why;
    }
    return hr;
}

namespace
{
dxtex_hresult compress_many_pipelined(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts, size_t count, uint32_t flags, float threshold)
{
    static const uint64_t chunkTexels = dev_env("DXTEX_MANY_CHUNK_TEXELS") ? std::max<uint64_t>(1, strtoull(dev_env("DXTEX_MANY_CHUNK_TEXELS"), nullptr, 10)) : (32ull << 20);
    std::vector<size_t> inBytes(count), outBytes(count);
    std::vector<ManyChunk> chunks;
    {
        ManyChunk cur = { 0, 0, 0, 0 };
        uint64_t texels = 0;
        for (size_t i = 0; i < count; ++i)
        {
            dxtex_hresult hr = check_pair(ctx, &srcs[i], &dsts[i]);
            if (hr == DXTEX_S_OK) { SrcView v; hr = compress_view(ctx, nullptr, srcs[i].width, srcs[i].height, srcs[i].format, srcs[i].rowPitch, dsts[i].format, flags, &v); }
            if (hr == DXTEX_S_OK) hr = check_host_pitches(ctx, &srcs[i], &dsts[i], &inBytes[i], &outBytes[i]);
            if (hr != DXTEX_S_OK) return hr;
            const uint64_t t = uint64_t(srcs[i].width) * srcs[i].height;
            if (cur.count && texels + t > chunkTexels) { chunks.push_back(cur); cur = { i, 0, 0, 0 }; texels = 0; }
            ++cur.count; texels += t;
            cur.inBytes += (inBytes[i] + 255) & ~size_t(255);
            cur.outBytes += (outBytes[i] + 255) & ~size_t(255);
        }
        chunks.push_back(cur);
    }
    if (!ctx->h2d) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->h2d, hipStreamNonBlocking));
    if (!ctx->d2h) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->d2h, hipStreamNonBlocking));
    for (dxtex_ctx::Lane& l : ctx->lane)
    {
        if (!l.uploaded) HIP_TRY(ctx, hipEventCreateWithFlags(&l.uploaded, hipEventDisableTiming));
        if (!l.computed) HIP_TRY(ctx, hipEventCreateWithFlags(&l.computed, hipEventDisableTiming));
        if (!l.downloaded) HIP_TRY(ctx, hipEventCreateWithFlags(&l.downloaded, hipEventDisableTiming));
    }

    // payload of chunk c: pinned -> the caller's images (after its download has finished)
    auto scatter = [&](size_t c) -> dxtex_hresult
    {
        dxtex_ctx::Lane& l = ctx->lane[c & 1];
        HIP_TRY(ctx, hipEventSynchronize(l.downloaded));
        size_t at = 0;
        for (size_t i = chunks[c].first; i < chunks[c].first + chunks[c].count; ++i)
        {
            std::memcpy(dsts[i].pixels, static_cast<const uint8_t*>(l.pinOut) + at, outBytes[i]);
            at += (outBytes[i] + 255) & ~size_t(255);
        }
        return DXTEX_S_OK;
    };

    std::vector<dxtex_image> ds, dd;
    for (size_t c = 0; c < chunks.size(); ++c)
    {
        const ManyChunk& ch = chunks[c];
        dxtex_ctx::Lane& l = ctx->lane[c & 1];
        if (c >= 2) { const dxtex_hresult hr = scatter(c - 2); if (hr != DXTEX_S_OK) return hr; }      // frees this lane's pinOut (and, stream-ordered, devOut)
        dxtex_hresult hr = ensure_pinned(ctx, &l.pinIn, &l.pinInBytes, ch.inBytes); if (hr != DXTEX_S_OK) return hr;
        hr = ensure_pinned(ctx, &l.pinOut, &l.pinOutBytes, ch.outBytes); if (hr != DXTEX_S_OK) return hr;
        hr = ensure(ctx, &l.devIn, &l.devInBytes, ch.inBytes); if (hr != DXTEX_S_OK) return hr;
        hr = ensure(ctx, &l.devOut, &l.devOutBytes, ch.outBytes); if (hr != DXTEX_S_OK) return hr;
        // gather (pinIn of this lane was last read by the upload of chunk c - 2, which the kernels of c - 2 waited for, which the download
        // of c - 2 waited for, which scatter(c - 2) has just waited for)
        ds.assign(srcs + ch.first, srcs + ch.first + ch.count);
        dd.assign(dsts + ch.first, dsts + ch.first + ch.count);
        size_t atIn = 0, atOut = 0;
        for (size_t k = 0; k < ch.count; ++k)
        {
            const size_t i = ch.first + k;
            std::memcpy(static_cast<uint8_t*>(l.pinIn) + atIn, srcs[i].pixels, inBytes[i]);
            ds[k].pixels = static_cast<uint8_t*>(l.devIn) + atIn;
            dd[k].pixels = static_cast<uint8_t*>(l.devOut) + atOut;
            atIn += (inBytes[i] + 255) & ~size_t(255);
            atOut += (outBytes[i] + 255) & ~size_t(255);
        }
        HIP_TRY(ctx, counted_copy(ctx, l.devIn, l.pinIn, atIn, hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipEventRecord(l.uploaded, ctx->h2d));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, l.uploaded, 0));
        hr = dxtex_compress_many_device(ctx, ds.data(), dd.data(), ch.count, flags, threshold);
        if (hr != DXTEX_S_OK) return hr;
        HIP_TRY(ctx, hipEventRecord(l.computed, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->d2h, l.computed, 0));
        HIP_TRY(ctx, counted_copy(ctx, l.pinOut, l.devOut, atOut, hipMemcpyDeviceToHost, ctx->d2h));
        HIP_TRY(ctx, hipEventRecord(l.downloaded, ctx->d2h));
        // (no wait of h2d on `computed`: the next upload into this lane's devIn belongs to chunk c + 2, and iteration c + 2 begins with
        // scatter(c), a host wait for downloaded(c), which is stream-ordered after computed(c). The upload of chunk c + 1 - the other
        // lane - therefore runs while these kernels do.)
    }
    for (size_t c = chunks.size() >= 2 ? chunks.size() - 2 : 0; c < chunks.size(); ++c)
    {
        const dxtex_hresult hr = scatter(c);
        if (hr != DXTEX_S_OK) return hr;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
}

dxtex_hresult dxtex_compress(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags, float threshold)
{
    dxtex_hresult hr = check_pair(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    { SrcView v; hr = compress_view(ctx, nullptr, src->width, src->height, src->format, src->rowPitch, dst->format, flags, &v); if (hr != DXTEX_S_OK) return hr; }
    ScopedDevice sd(ctx->device);
    size_t srcBytes = 0, dstBytes = 0;
    hr = check_host_pitches(ctx, src, dst, &srcBytes, &dstBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hr = submit_compress(ctx, static_cast<const uint8_t*>(ctx->stageIn), src->width, src->height, src->format, src->rowPitch,
                         static_cast<uint8_t*>(ctx->stageOut), dst->format, dst->rowPitch, flags, threshold);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
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
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, rgba, srcBytes, hipMemcpyHostToDevice, ctx->stream));

    SrcView v;
    v.pixels = static_cast<const uint8_t*>(ctx->stageIn); v.width = 4; v.height = uint32_t(nblocks * 4);
    v.rowPitch = 64; v.format = FMT_R32G32B32A32_FLOAT; v.tcv = TCV_NONE; v.tsw = TSW_NONE;   // raw floats, as BC_ENCODE receives them
    hipError_t e;
    if (bc_format == FMT_BC7_UNORM || bc_format == FMT_BC7_UNORM_SRGB)
    {
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc7_scratch_bytes(nblocks, bc_flags));
        if (hr != DXTEX_S_OK) return hr;
    }
    if (bc_format == FMT_BC6H_UF16 || bc_format == FMT_BC6H_SF16)
    {
        hr = ensure(ctx, &ctx->scratch, &ctx->scratchBytes, bc6h_scratch_bytes(nblocks));
        if (hr != DXTEX_S_OK) return hr;
    }
    time_begin(ctx);
    switch (bc_format)
    {
    case FMT_BC6H_UF16: case FMT_BC6H_SF16:
        e = launch_bc6h_encode(v, static_cast<uint8_t*>(ctx->stageOut), bb, bc_format == FMT_BC6H_SF16, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx));
        break;
    case FMT_BC7_UNORM: case FMT_BC7_UNORM_SRGB:
        e = launch_bc7_encode(v, static_cast<uint8_t*>(ctx->stageOut), bb, bc_flags, ctx->scratch, ctx->stream, ctx->profiling ? &ctx->marks : nullptr, side_streams(ctx));
        break;
    default:
        e = launch_bc15_encode(v, static_cast<uint8_t*>(ctx->stageOut), bb, bc_format, bc_flags, threshold, ctx->stream);
        break;
    }
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    HIP_TRY(ctx, counted_copy(ctx, bc, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
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
    if (!out || (out->cls & FC_GROUP)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "destination format is not supported by the MI355X path");
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
    size_t srcBytes = 0, dstBytes = 0;
    if (format_info(src->format) && format_info(dst->format)) { hr = check_host_pitches(ctx, src, dst, &srcBytes, &dstBytes); if (hr != DXTEX_S_OK) return hr; }
    else { srcBytes = src->rowPitch * std::max<size_t>(1, (src->height + 3) / 4); dstBytes = dst->rowPitch * dst->height; }      // submit_decompress rejects the formats below
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hr = submit_decompress(ctx, static_cast<const uint8_t*>(ctx->stageIn), src->format, src->rowPitch,
                           static_cast<uint8_t*>(ctx->stageOut), dst->format, dst->rowPitch, src->width, src->height);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
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
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, bc, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    ConvertPlan plan; plan.srgbIn = 0; plan.tcv = TCV_NONE; plan.tsw = TSW_NONE; plan.srgbOut = 0; plan.depth = 0;
    time_begin(ctx);
    hipError_t e = launch_bc_decode(static_cast<const uint8_t*>(ctx->stageIn), bb, bc_format, static_cast<uint8_t*>(ctx->stageOut), 64,
                                    FMT_R32G32B32A32_FLOAT, 4, uint32_t(nblocks * 4), plan, ctx->stream);
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    HIP_TRY(ctx, counted_copy(ctx, rgba, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
// ---- GenerateMipMaps / Resize / Convert ----------------------------------------------------------------------------------
namespace
{
constexpr uint32_t kFilterModeMask = 0xF00000u, kFilterDitherMask = 0xF0000u;
inline bool ispow2(size_t x) { return x != 0 && (x & (x - 1)) == 0; }

struct LevelPair { const uint8_t* src; size_t srcPitch, sw, sh; uint8_t* dst; size_t dstPitch, dw, dh; };

// Builds the triangle tables of every (src -> dst) pair into one device buffer, then launches the filter per pair.
// The triangle filter's gather lists go to the device through a pinned buffer of the context; an event marks when the last
// upload has been consumed, so an asynchronous (_device) call never rewrites host memory under a copy in flight.
dxtex_hresult upload_tables(dxtex_ctx* ctx, const std::vector<uint8_t>& host)
{
    dxtex_hresult hr = ensure(ctx, &ctx->triBuf, &ctx->triBytes, host.size()); if (hr != DXTEX_S_OK) return hr;
    if (ctx->triPending) { HIP_TRY(ctx, hipEventSynchronize(ctx->triConsumed)); ctx->triPending = false; }
    if (ctx->triPinnedBytes < host.size())
    {
        if (ctx->triPinned) { HIP_TRY(ctx, hipHostFree(ctx->triPinned)); ctx->triPinned = nullptr; ctx->triPinnedBytes = 0; }
        const size_t bytes = std::max<size_t>(host.size(), 1u << 16);
        HIP_TRY(ctx, hipHostMalloc(&ctx->triPinned, bytes, hipHostMallocDefault));
        ctx->triPinnedBytes = bytes;
    }
    if (!ctx->triConsumed) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->triConsumed, hipEventDisableTiming));
    std::memcpy(ctx->triPinned, host.data(), host.size());
    HIP_TRY(ctx, counted_copy(ctx, ctx->triBuf, ctx->triPinned, host.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->triConsumed, ctx->stream));
    ctx->triPending = true;
    return DXTEX_S_OK;
}

// A destination whose element holds several texels (FC_GROUP) is written in two steps: the operation leaves R32G32B32A32_FLOAT rows in
// ctx->groupRows (what the reference hands to StoreScanline), launch_pack_group stores them. The stream orders the steps, so one
// buffer serves every level of a chain.
bool is_group_format(int format) { const FmtInfo* f = format_info(format); return f && (f->cls & FC_GROUP); }
dxtex_hresult group_rows(dxtex_ctx* ctx, size_t width, size_t height, uint8_t** rows, size_t* pitch)
{
    *pitch = width * 16;
    const dxtex_hresult hr = ensure(ctx, &ctx->groupRows, &ctx->groupRowsBytes, *pitch * height);
    *rows = static_cast<uint8_t*>(ctx->groupRows);
    return hr;
}

dxtex_hresult submit_resizes(dxtex_ctx* ctx, const std::vector<LevelPair>& pairs, int format, uint32_t mode, uint32_t flags, bool mipAlias)
{
    std::vector<size_t> base(pairs.size(), 0);
    const bool grouped = is_group_format(format);
    // one resize: straight into the destination, or through float rows and the pack kernel
    auto resize_one = [&](const LevelPair& p, const TriangleTables* t, const uint8_t* staleSrc, uint64_t stalePitch, uint32_t staleW) -> dxtex_hresult
    {
        uint8_t* out = p.dst; size_t outPitch = p.dstPitch;
        if (grouped) { const dxtex_hresult hr = group_rows(ctx, p.dw, p.dh, &out, &outPitch); if (hr != DXTEX_S_OK) return hr; }
        hipError_t e = launch_resize(p.src, p.srcPitch, uint32_t(p.sw), uint32_t(p.sh), out, outPitch, uint32_t(p.dw), uint32_t(p.dh),
                                     format, mode, flags, mipAlias, t, ctx->stream, staleSrc, stalePitch, staleW, grouped ? FMT_R32G32B32A32_FLOAT : -1);
        if (e == hipSuccess && grouped) e = launch_pack_group(out, outPitch, p.dst, p.dstPitch, format, uint32_t(p.dw), uint32_t(p.dh), ctx->stream);
        return e == hipSuccess ? DXTEX_S_OK : fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    };
    if (mode == DXTEX_FILTER_TRIANGLE)
    {
        std::vector<uint8_t>& host = ctx->triHost;
        host.clear();
        std::vector<uint32_t> ofs; std::vector<TriEntry> ent;
        struct Slot { size_t ofsX, entX, ofsY, entY; };
        std::vector<Slot> slots(pairs.size());
        auto append = [&](const void* p, size_t bytes) { const size_t at = (host.size() + 15) & ~size_t(15); host.resize(at + bytes); std::memcpy(host.data() + at, p, bytes); return at; };
        for (size_t i = 0; i < pairs.size(); ++i)
        {
            build_triangle_axis(pairs[i].sw, pairs[i].dw, (flags & DXTEX_FILTER_WRAP_U) != 0, ofs, ent);
            slots[i].ofsX = append(ofs.data(), ofs.size() * 4); slots[i].entX = append(ent.data(), std::max<size_t>(1, ent.size()) * 8);
            build_triangle_axis(pairs[i].sh, pairs[i].dh, (flags & DXTEX_FILTER_WRAP_V) != 0, ofs, ent);
            slots[i].ofsY = append(ofs.data(), ofs.size() * 4); slots[i].entY = append(ent.data(), std::max<size_t>(1, ent.size()) * 8);
        }
        host.resize(host.size() + 16);
        dxtex_hresult hr = upload_tables(ctx, host); if (hr != DXTEX_S_OK) return hr;
        const uint8_t* d = static_cast<const uint8_t*>(ctx->triBuf);
        for (size_t i = 0; i < pairs.size(); ++i)
        {
            TriangleTables t;
            t.ofsX = reinterpret_cast<const uint32_t*>(d + slots[i].ofsX); t.entX = d + slots[i].entX;
            t.ofsY = reinterpret_cast<const uint32_t*>(d + slots[i].ofsY); t.entY = d + slots[i].entY;
            const dxtex_hresult hr1 = resize_one(pairs[i], &t, nullptr, 0, 0);
            if (hr1 != DXTEX_S_OK) return hr1;
        }
        return DXTEX_S_OK;
    }
    const LevelPair* twoHigh = nullptr;      // box mips: the last source level that was 2 texels high (resize_box_kernel's stale tap)
    for (size_t i = 0; i < pairs.size(); ++i)
    {
        const LevelPair& p = pairs[i];
        // a mip chain's last levels (source at most 64 x 64, each level the next one's source) run in one workgroup
        const bool cubicTail = mode == DXTEX_FILTER_CUBIC && p.sw <= 64 && p.sh <= 64;
        if (mipAlias && !grouped && pairs.size() - i >= 2 && (cubicTail || resize_tail_applies(uint32_t(p.sw), uint32_t(p.sh), mode)))
        {
            bool chain = true;
            for (size_t k = i + 1; k < pairs.size(); ++k) chain = chain && pairs[k].src == pairs[k - 1].dst && pairs[k].srcPitch == pairs[k - 1].dstPitch;
            std::vector<MipLevel> lv;
            if (chain)
            {
                lv.push_back({ const_cast<uint8_t*>(p.src), p.srcPitch, uint32_t(p.sw), uint32_t(p.sh) });
                for (size_t k = i; k < pairs.size(); ++k) lv.push_back({ pairs[k].dst, pairs[k].dstPitch, uint32_t(pairs[k].dw), uint32_t(pairs[k].dh) });
                if (cubicTail) chain = resize_cubic_tail_applies(lv.data(), int(lv.size()), format, flags);
            }
            if (chain)
            {
                MipLevel th = { nullptr, 0, 0, 0 };
                if (twoHigh) th = { const_cast<uint8_t*>(twoHigh->src), twoHigh->srcPitch, uint32_t(twoHigh->sw), uint32_t(twoHigh->sh) };
                const hipError_t e = launch_resize_tail(lv.data(), int(lv.size()), format, mode, flags, twoHigh ? &th : nullptr, ctx->stream);
                if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
                return DXTEX_S_OK;
            }
        }
        if (mipAlias && p.sh >= 2) twoHigh = &p;
        const bool stale = mipAlias && mode == DXTEX_FILTER_BOX && p.sh == 1 && p.sw > 1 && twoHigh;
        const dxtex_hresult hr1 = resize_one(p, nullptr, stale ? twoHigh->src : nullptr, stale ? twoHigh->srcPitch : 0, stale ? uint32_t(twoHigh->sw) : 0u);
        if (hr1 != DXTEX_S_OK) return hr1;
    }
    return DXTEX_S_OK;
}

// GenerateMipMaps' checks and filter choice (DirectXTexMipmaps.cpp:2828-3017, non-WIC path)
dxtex_hresult check_mips(dxtex_ctx* ctx, const dxtex_image* levels, size_t nlevels, uint32_t filter, uint32_t* mode)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!levels || nlevels <= 1) return fail(ctx, DXTEX_E_INVALIDARG, "need at least two levels");
    const FmtInfo* f = format_info(levels[0].format);
    if (f && (f->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "cannot filter a block-compressed image");
    if (!f) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    size_t w = levels[0].width, h = levels[0].height;
    if (!w || !h) return fail(ctx, DXTEX_E_INVALIDARG, "empty image");
    for (size_t i = 0; i < nlevels; ++i)
    {
        if (!levels[i].pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
        if (levels[i].width != w || levels[i].height != h || levels[i].format != levels[0].format)
            return fail(ctx, DXTEX_E_INVALIDARG, "levels do not form a mip chain");
        if (i + 1 < nlevels && w == 1 && h == 1) return fail(ctx, DXTEX_E_INVALIDARG, "too many levels");     // CalculateMipLevels
        w = std::max<size_t>(1, w >> 1); h = std::max<size_t>(1, h >> 1);
    }
    uint32_t m = filter & kFilterModeMask;
    if (!m) m = (ispow2(levels[0].width) && ispow2(levels[0].height)) ? DXTEX_FILTER_BOX : DXTEX_FILTER_LINEAR;
    if (m != DXTEX_FILTER_POINT && m != DXTEX_FILTER_LINEAR && m != DXTEX_FILTER_CUBIC && m != DXTEX_FILTER_BOX && m != DXTEX_FILTER_TRIANGLE)
        return fail(ctx, DXTEX_E_NOT_SUPPORTED, "unknown filter mode");
    if (m == DXTEX_FILTER_BOX && (!ispow2(levels[0].width) || !ispow2(levels[0].height)))
        return fail(ctx, DXTEX_E_FAIL, "the box filter needs power-of-two dimensions");                         // :1005
    *mode = m;
    return DXTEX_S_OK;
}
} // namespace

dxtex_hresult dxtex_generate_mips_device(dxtex_ctx* ctx, const dxtex_image* levels, size_t nlevels, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_mips(ctx, levels, nlevels, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<LevelPair> pairs;
    for (size_t i = 1; i < nlevels; ++i)
        pairs.push_back({ levels[i - 1].pixels, levels[i - 1].rowPitch, levels[i - 1].width, levels[i - 1].height,
                          levels[i].pixels, levels[i].rowPitch, levels[i].width, levels[i].height });
    time_begin(ctx);
    hr = submit_resizes(ctx, pairs, levels[0].format, mode, filter, true);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_generate_mips(dxtex_ctx* ctx, const dxtex_image* levels, size_t nlevels, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_mips(ctx, levels, nlevels, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    // one device allocation holding the whole chain, 256-byte aligned levels
    std::vector<size_t> at(nlevels);
    size_t total = 0;
    for (size_t i = 0; i < nlevels; ++i) { at[i] = total; total += (levels[i].rowPitch * levels[i].height + 255) & ~size_t(255); }
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, total); if (hr != DXTEX_S_OK) return hr;
    uint8_t* d = static_cast<uint8_t*>(ctx->stageIn);
    HIP_TRY(ctx, counted_copy(ctx, d, levels[0].pixels, levels[0].rowPitch * levels[0].height, hipMemcpyHostToDevice, ctx->stream));
    std::vector<LevelPair> pairs;
    for (size_t i = 1; i < nlevels; ++i)
        pairs.push_back({ d + at[i - 1], levels[i - 1].rowPitch, levels[i - 1].width, levels[i - 1].height,
                          d + at[i], levels[i].rowPitch, levels[i].width, levels[i].height });
    time_begin(ctx);
    hr = submit_resizes(ctx, pairs, levels[0].format, mode, filter, true);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    for (size_t i = 1; i < nlevels; ++i)
        HIP_TRY(ctx, counted_copy(ctx, levels[i].pixels, d + at[i], levels[i].rowPitch * levels[i].height, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

namespace
{
// GenerateMipMaps3D's checks and filter choice (DirectXTexMipmaps.cpp:3254-3305)
dxtex_hresult check_mips3d(dxtex_ctx* ctx, const dxtex_volume* levels, size_t nlevels, uint32_t filter, uint32_t* mode)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!levels || nlevels <= 1) return fail(ctx, DXTEX_E_INVALIDARG, "need at least two levels");
    const FmtInfo* f = format_info(levels[0].format);
    if (f && (f->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "cannot filter a block-compressed volume");
    if (!f || (f->cls & FC_GROUP)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    size_t w = levels[0].width, h = levels[0].height, d = levels[0].depth;
    if (!w || !h || !d || d > 32767) return fail(ctx, DXTEX_E_INVALIDARG, "bad volume dimensions");           // depth > INT16_MAX, :3264
    if (filter & 0x20000000u) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "TEX_FILTER_FORCE_WIC");
    for (size_t i = 0; i < nlevels; ++i)
    {
        if (!levels[i].pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
        if (levels[i].width != w || levels[i].height != h || levels[i].depth != d || levels[i].format != levels[0].format)
            return fail(ctx, DXTEX_E_INVALIDARG, "levels do not form a volume mip chain");
        if (i + 1 < nlevels && w == 1 && h == 1 && d == 1) return fail(ctx, DXTEX_E_INVALIDARG, "too many levels");     // CalculateMipLevels3D
        w = std::max<size_t>(1, w >> 1); h = std::max<size_t>(1, h >> 1); d = std::max<size_t>(1, d >> 1);
    }
    const bool pow2 = ispow2(levels[0].width) && ispow2(levels[0].height) && ispow2(levels[0].depth);
    uint32_t m = filter & kFilterModeMask;
    if (!m) m = pow2 ? DXTEX_FILTER_BOX : DXTEX_FILTER_TRIANGLE;
    if (m != DXTEX_FILTER_POINT && m != DXTEX_FILTER_LINEAR && m != DXTEX_FILTER_CUBIC && m != DXTEX_FILTER_BOX && m != DXTEX_FILTER_TRIANGLE)
        return fail(ctx, DXTEX_E_NOT_SUPPORTED, "unknown filter mode");
    if (m == DXTEX_FILTER_BOX && !pow2) return fail(ctx, DXTEX_E_FAIL, "the box filter needs power-of-two dimensions");   // :1831-1832
    *mode = m;
    return DXTEX_S_OK;
}

// The level loop of Generate3DMips*Filter on device-resident levels: 3-D kernels while the source is more than one slice deep,
// then the reference's 2-D branches (the kernels GenerateMipMaps uses) - except the triangle filter, which has no 2-D branch.
dxtex_hresult submit_mips3d(dxtex_ctx* ctx, const std::vector<VolumeView>& lv, uint32_t mode, uint32_t flags)
{
    const int format = lv[0].format;
    struct Slot { size_t ofsX, entX, ofsY, entY, ofsZ, entZ; };
    std::vector<Slot> slots(lv.size());
    const uint8_t* tri = nullptr;
    if (mode == DXTEX_FILTER_TRIANGLE)
    {
        std::vector<uint8_t>& host = ctx->triHost;
        host.clear();
        std::vector<uint32_t> ofs; std::vector<TriEntry> ent;
        auto append = [&](const void* p, size_t bytes) { const size_t at = (host.size() + 15) & ~size_t(15); host.resize(at + bytes); std::memcpy(host.data() + at, p, bytes); return at; };
        for (size_t i = 1; i < lv.size(); ++i)
        {
            build_triangle_axis(lv[i - 1].width, lv[i].width, (flags & DXTEX_FILTER_WRAP_U) != 0, ofs, ent);
            slots[i].ofsX = append(ofs.data(), ofs.size() * 4); slots[i].entX = append(ent.data(), std::max<size_t>(1, ent.size()) * 8);
            build_triangle_axis(lv[i - 1].height, lv[i].height, (flags & DXTEX_FILTER_WRAP_V) != 0, ofs, ent);
            slots[i].ofsY = append(ofs.data(), ofs.size() * 4); slots[i].entY = append(ent.data(), std::max<size_t>(1, ent.size()) * 8);
            build_triangle_axis(lv[i - 1].depth, lv[i].depth, (flags & 0x4u) != 0, ofs, ent);
            slots[i].ofsZ = append(ofs.data(), ofs.size() * 4); slots[i].entZ = append(ent.data(), std::max<size_t>(1, ent.size()) * 8);
        }
        host.resize(host.size() + 16);
        dxtex_hresult hr = upload_tables(ctx, host); if (hr != DXTEX_S_OK) return hr;
        tri = static_cast<const uint8_t*>(ctx->triBuf);
    }
    const VolumeView* twoHigh = nullptr;     // box: the last SOURCE level that was 2 texels high (what urow1 / vrow1's old buffers hold)
    for (size_t i = 1; i < lv.size(); ++i)
    {
        const VolumeView& s = lv[i - 1]; const VolumeView& d = lv[i];
        if (s.height >= 2) twoHigh = &s;
        // row 1 of the last slice pair loaded into urow1 / vrow1 at that level: slices depth-2 and depth-1 (a one-slice level only has urow1)
        const bool stale = mode == DXTEX_FILTER_BOX && s.height == 1 && s.width > 1 && twoHigh;
        const uint8_t* staleU = stale ? twoHigh->pixels + uint64_t(twoHigh->depth >= 2 ? twoHigh->depth - 2 : 0) * twoHigh->slicePitch : nullptr;
        const uint8_t* staleV = stale ? twoHigh->pixels + uint64_t(twoHigh->depth - 1) * twoHigh->slicePitch : nullptr;
        hipError_t e;
        if (s.depth > 1 || mode == DXTEX_FILTER_TRIANGLE)
        {
            TriangleTables3 t{};
            if (tri)
            {
                t.ofsX = reinterpret_cast<const uint32_t*>(tri + slots[i].ofsX); t.entX = tri + slots[i].entX;
                t.ofsY = reinterpret_cast<const uint32_t*>(tri + slots[i].ofsY); t.entY = tri + slots[i].entY;
                t.ofsZ = reinterpret_cast<const uint32_t*>(tri + slots[i].ofsZ); t.entZ = tri + slots[i].entZ;
            }
            e = launch_resize3d(s, d, mode, flags, tri ? &t : nullptr, ctx->stream, staleU, staleV, stale ? twoHigh->rowPitch : 0, stale ? twoHigh->width : 0u);
        }
        else
            e = launch_resize(s.pixels, s.rowPitch, s.width, s.height, const_cast<uint8_t*>(d.pixels), d.rowPitch, d.width, d.height, format, mode, flags, true,
                              nullptr, ctx->stream, staleU, stale ? twoHigh->rowPitch : 0, stale ? twoHigh->width : 0u);
        if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    }
    return DXTEX_S_OK;
}

VolumeView view_of(const dxtex_volume& v, const uint8_t* pixels)
{
    VolumeView o; o.pixels = pixels; o.rowPitch = v.rowPitch; o.slicePitch = v.slicePitch;
    o.width = uint32_t(v.width); o.height = uint32_t(v.height); o.depth = uint32_t(v.depth); o.format = v.format;
    return o;
}
} // namespace

dxtex_hresult dxtex_generate_mips3d_device(dxtex_ctx* ctx, const dxtex_volume* levels, size_t nlevels, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_mips3d(ctx, levels, nlevels, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<VolumeView> lv(nlevels);
    for (size_t i = 0; i < nlevels; ++i) lv[i] = view_of(levels[i], levels[i].pixels);
    time_begin(ctx);
    hr = submit_mips3d(ctx, lv, mode, filter);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_generate_mips3d(dxtex_ctx* ctx, const dxtex_volume* levels, size_t nlevels, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_mips3d(ctx, levels, nlevels, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<size_t> at(nlevels);
    size_t total = 0;
    for (size_t i = 0; i < nlevels; ++i) { at[i] = total; total += (levels[i].slicePitch * levels[i].depth + 255) & ~size_t(255); }
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, total); if (hr != DXTEX_S_OK) return hr;
    uint8_t* d = static_cast<uint8_t*>(ctx->stageIn);
    HIP_TRY(ctx, counted_copy(ctx, d, levels[0].pixels, levels[0].slicePitch * levels[0].depth, hipMemcpyHostToDevice, ctx->stream));
    std::vector<VolumeView> lv(nlevels);
    for (size_t i = 0; i < nlevels; ++i) lv[i] = view_of(levels[i], d + at[i]);
    time_begin(ctx);
    hr = submit_mips3d(ctx, lv, mode, filter);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    for (size_t i = 1; i < nlevels; ++i)
        HIP_TRY(ctx, counted_copy(ctx, levels[i].pixels, d + at[i], levels[i].slicePitch * levels[i].depth, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

namespace
{
// Resize's checks and filter choice (DirectXTexResize.cpp:807-843, :854-930)
dxtex_hresult check_resize(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, uint32_t* mode)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!src || !dst) return fail(ctx, DXTEX_E_INVALIDARG, "null image");
    if (!dst->width || !dst->height || !src->width || !src->height) return fail(ctx, DXTEX_E_INVALIDARG, "empty image");
    if (src->width > 0xFFFFFFFFull || src->height > 0xFFFFFFFFull || dst->width > 0xFFFFFFFFull || dst->height > 0xFFFFFFFFull)
        return fail(ctx, DXTEX_E_INVALIDARG, "image too large");
    if (!src->pixels || !dst->pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
    const FmtInfo* f = format_info(src->format);
    if (f && (f->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "cannot resize a block-compressed image");
    if (!f) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    if (src->format != dst->format) return fail(ctx, DXTEX_E_INVALIDARG, "Resize keeps the format");
    uint32_t m = filter & kFilterModeMask;
    const bool half = ((dst->width << 1) == src->width) && ((dst->height << 1) == src->height);
    if (!m) m = half ? DXTEX_FILTER_BOX : DXTEX_FILTER_LINEAR;
    if (m != DXTEX_FILTER_POINT && m != DXTEX_FILTER_LINEAR && m != DXTEX_FILTER_CUBIC && m != DXTEX_FILTER_BOX && m != DXTEX_FILTER_TRIANGLE)
        return fail(ctx, DXTEX_E_NOT_SUPPORTED, "unknown filter mode");
    if (m == DXTEX_FILTER_BOX && !half) return fail(ctx, DXTEX_E_FAIL, "the box filter needs an exact 2:1 reduction");   // :319-320
    *mode = m;
    return DXTEX_S_OK;
}
} // namespace

dxtex_hresult dxtex_resize_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_resize(ctx, src, dst, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<LevelPair> pairs{ { src->pixels, src->rowPitch, src->width, src->height, dst->pixels, dst->rowPitch, dst->width, dst->height } };
    time_begin(ctx);
    hr = submit_resizes(ctx, pairs, src->format, mode, filter, false);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_resize(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter)
{
    uint32_t mode = 0;
    dxtex_hresult hr = check_resize(ctx, src, dst, filter, &mode);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    size_t srcBytes = 0, dstBytes = 0;
    hr = check_host_pitches(ctx, src, dst, &srcBytes, &dstBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    std::vector<LevelPair> pairs{ { static_cast<const uint8_t*>(ctx->stageIn), src->rowPitch, src->width, src->height,
                                    static_cast<uint8_t*>(ctx->stageOut), dst->rowPitch, dst->width, dst->height } };
    time_begin(ctx);
    hr = submit_resizes(ctx, pairs, src->format, mode, filter, false);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

namespace
{
// ConvertEx's checks (DirectXTexConvert.cpp:5107-5125); dithering is not implemented on this path
dxtex_hresult check_convert(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, ConvertPlan* plan)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!src || !dst) return fail(ctx, DXTEX_E_INVALIDARG, "null image");
    if (src->format == dst->format) return fail(ctx, DXTEX_E_INVALIDARG, "source and destination formats are the same");
    if (!src->pixels || !dst->pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
    const FmtInfo* in = format_info(src->format);
    const FmtInfo* out = format_info(dst->format);
    if ((in && (in->cls & FC_BC)) || (out && (out->cls & FC_BC))) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "Convert does not take block-compressed formats");
    if (!in || !out) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    if (src->width != dst->width || src->height != dst->height) return fail(ctx, DXTEX_E_FAIL, "size mismatch");
    if (filter & kFilterDitherMask) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "dithered conversion is not implemented on the MI355X path");
    *plan = resolve_convert_plan(*in, *out, filter);
    return DXTEX_S_OK;
}

// the conversion kernel, through float rows + the pack kernel when the destination's element holds several texels
dxtex_hresult submit_convert(dxtex_ctx* ctx, const uint8_t* dSrc, size_t srcPitch, int srcFormat, uint8_t* dDst, size_t dstPitch, int dstFormat,
                             size_t width, size_t height, const ConvertPlan& plan, float threshold)
{
    uint8_t* out = dDst; size_t outPitch = dstPitch; int outFormat = dstFormat;
    const bool grouped = is_group_format(dstFormat);
    if (grouped)
    {
        const dxtex_hresult hr = group_rows(ctx, width, height, &out, &outPitch);
        if (hr != DXTEX_S_OK) return hr;
        outFormat = FMT_R32G32B32A32_FLOAT;
    }
    hipError_t e = launch_convert(dSrc, srcPitch, srcFormat, out, outPitch, outFormat, uint32_t(width), uint32_t(height), plan, threshold, ctx->stream);
    if (e == hipSuccess && grouped) e = launch_pack_group(out, outPitch, dDst, dstPitch, dstFormat, uint32_t(width), uint32_t(height), ctx->stream);
    return e == hipSuccess ? DXTEX_S_OK : fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
}
} // namespace

dxtex_hresult dxtex_convert_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, float threshold)
{
    ConvertPlan plan;
    dxtex_hresult hr = check_convert(ctx, src, dst, filter, &plan);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    time_begin(ctx);
    hr = submit_convert(ctx, src->pixels, src->rowPitch, src->format, dst->pixels, dst->rowPitch, dst->format, src->width, src->height, plan, threshold);
    time_end(ctx);
    return hr;
}

dxtex_hresult dxtex_convert(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, float threshold)
{
    ConvertPlan plan;
    dxtex_hresult hr = check_convert(ctx, src, dst, filter, &plan);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    size_t srcBytes = 0, dstBytes = 0;
    hr = check_host_pitches(ctx, src, dst, &srcBytes, &dstBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hr = submit_convert(ctx, static_cast<const uint8_t*>(ctx->stageIn), src->rowPitch, src->format, static_cast<uint8_t*>(ctx->stageOut),
                        dst->rowPitch, dst->format, src->width, src->height, plan, threshold);
    time_end(ctx);
    if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

namespace
{
// PremultiplyAlpha's checks (DirectXTexPMAlpha.cpp:214-231)
dxtex_hresult check_pmalpha(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!src || !dst) return fail(ctx, DXTEX_E_INVALIDARG, "null image");
    if (!src->pixels || !dst->pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
    const FmtInfo* f = format_info(src->format);
    if (!f || (f->cls & FC_BC) || !(f->cls & FC_A)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "PremultiplyAlpha needs an uncompressed format with alpha");
    if (src->width > 0xFFFFFFFFull || src->height > 0xFFFFFFFFull) return fail(ctx, DXTEX_E_INVALIDARG, "image too large");
    if (src->format != dst->format || src->width != dst->width || src->height != dst->height) return fail(ctx, DXTEX_E_FAIL, "size or format mismatch");
    return DXTEX_S_OK;
}

// EstimateAlphaScaleForCoverage (DirectXTexMipmaps.cpp:310-352) around the device coverage count
dxtex_hresult alpha_coverage(dxtex_ctx* ctx, const uint8_t* d, const dxtex_image& im, float scale, float alphaReference, float* coverage)
{
    dxtex_hresult hr = ensure(ctx, &ctx->mseBuf, &ctx->mseBytes, 4 * sizeof(double)); if (hr != DXTEX_S_OK) return hr;
    hipError_t e = launch_alpha_coverage(d, im.rowPitch, im.format, uint32_t(im.width), uint32_t(im.height), scale, alphaReference,
                                         static_cast<unsigned long long*>(ctx->mseBuf), ctx->stream);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    unsigned long long n = 0;
    HIP_TRY(ctx, counted_copy(ctx, &n, ctx->mseBuf, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const float cscale = static_cast<float>((im.width - 1) * (im.height - 1) * 8 * 8);      // :299-303
    *coverage = (cscale > 0.f) ? static_cast<float>(size_t(n)) / cscale : 0.0f;
    return DXTEX_S_OK;
}

dxtex_hresult check_coverage_chain(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t nlevels)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!src || !dst || !nlevels) return fail(ctx, DXTEX_E_INVALIDARG, "empty mip chain");
    const FmtInfo* f = format_info(src[0].format);
    if (f && (f->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "ScaleMipMapsAlphaForCoverage does not take block-compressed formats");
    if (!f || (f->cls & FC_GROUP)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    for (size_t i = 0; i < nlevels; ++i)
    {
        if (!src[i].pixels || !dst[i].pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
        if (src[i].format != src[0].format || dst[i].format != src[0].format || src[i].width != dst[i].width || src[i].height != dst[i].height)
            return fail(ctx, DXTEX_E_FAIL, "level size or format mismatch");
    }
    return DXTEX_S_OK;
}

// the body of ScaleMipMapsAlphaForCoverage (:3503-3553) on device-resident levels
dxtex_hresult submit_coverage_chain(dxtex_ctx* ctx, const std::vector<const uint8_t*>& s, const std::vector<uint8_t*>& d, const dxtex_image* src,
                                    const dxtex_image* dst, size_t nlevels, float alphaReference)
{
    float target = 0.0f;
    dxtex_hresult hr = alpha_coverage(ctx, s[0], src[0], 1.0f, alphaReference, &target);
    if (hr != DXTEX_S_OK) return hr;
    const FmtInfo* f = format_info(src[0].format);
    const size_t rowBytes = (src[0].width * f->bpp + 7) / 8;
    HIP_TRY(ctx, hipMemcpy2DAsync(d[0], dst[0].rowPitch, s[0], src[0].rowPitch, std::min(rowBytes, std::min(src[0].rowPitch, dst[0].rowPitch)), src[0].height,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    for (size_t level = 1; level < nlevels; ++level)
    {
        float lo = 0.0f, hi = 4.0f, scale = 1.0f;
        for (int i = 0; i < 10; ++i)
        {
            float cov = 0.0f;
            hr = alpha_coverage(ctx, s[level], src[level], scale, alphaReference, &cov);
            if (hr != DXTEX_S_OK) return hr;
            if (cov < target) lo = scale;
            else if (cov > target) hi = scale;
            else break;
            scale = (lo + hi) * 0.5f;
        }
        hipError_t e = launch_scale_alpha(s[level], src[level].rowPitch, d[level], dst[level].rowPitch, src[level].format, uint32_t(src[level].width),
                                          uint32_t(src[level].height), scale, ctx->stream);
        if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    }
    return DXTEX_S_OK;
}
} // namespace

dxtex_hresult dxtex_premultiply_alpha_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags)
{
    dxtex_hresult hr = check_pmalpha(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    time_begin(ctx);
    hipError_t e = launch_pmalpha(src->pixels, src->rowPitch, dst->pixels, dst->rowPitch, src->format, uint32_t(src->width), uint32_t(src->height), flags, ctx->stream);
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_premultiply_alpha(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags)
{
    dxtex_hresult hr = check_pmalpha(ctx, src, dst);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    size_t srcBytes = 0, dstBytes = 0;
    hr = check_host_pitches(ctx, src, dst, &srcBytes, &dstBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, srcBytes); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, dstBytes); if (hr != DXTEX_S_OK) return hr;
    HIP_TRY(ctx, counted_copy(ctx, ctx->stageIn, src->pixels, srcBytes, hipMemcpyHostToDevice, ctx->stream));
    time_begin(ctx);
    hipError_t e = launch_pmalpha(static_cast<const uint8_t*>(ctx->stageIn), src->rowPitch, static_cast<uint8_t*>(ctx->stageOut), dst->rowPitch, src->format,
                                  uint32_t(src->width), uint32_t(src->height), flags, ctx->stream);
    time_end(ctx);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    HIP_TRY(ctx, counted_copy(ctx, dst->pixels, ctx->stageOut, dstBytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_scale_mips_alpha_for_coverage_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t nlevels, float alphaReference)
{
    dxtex_hresult hr = check_coverage_chain(ctx, src, dst, nlevels);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<const uint8_t*> s(nlevels); std::vector<uint8_t*> d(nlevels);
    for (size_t i = 0; i < nlevels; ++i) { s[i] = src[i].pixels; d[i] = dst[i].pixels; }
    return submit_coverage_chain(ctx, s, d, src, dst, nlevels, alphaReference);
}

dxtex_hresult dxtex_scale_mips_alpha_for_coverage(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t nlevels, float alphaReference)
{
    dxtex_hresult hr = check_coverage_chain(ctx, src, dst, nlevels);
    if (hr != DXTEX_S_OK) return hr;
    ScopedDevice sd(ctx->device);
    std::vector<size_t> atS(nlevels), atD(nlevels);
    size_t totalS = 0, totalD = 0;
    for (size_t i = 0; i < nlevels; ++i)
    {
        atS[i] = totalS; totalS += (src[i].rowPitch * src[i].height + 255) & ~size_t(255);
        atD[i] = totalD; totalD += (dst[i].rowPitch * dst[i].height + 255) & ~size_t(255);
    }
    hr = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, totalS); if (hr != DXTEX_S_OK) return hr;
    hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, totalD); if (hr != DXTEX_S_OK) return hr;
    std::vector<const uint8_t*> s(nlevels); std::vector<uint8_t*> d(nlevels);
    for (size_t i = 0; i < nlevels; ++i)
    {
        s[i] = static_cast<const uint8_t*>(ctx->stageIn) + atS[i]; d[i] = static_cast<uint8_t*>(ctx->stageOut) + atD[i];
        HIP_TRY(ctx, counted_copy(ctx, static_cast<uint8_t*>(ctx->stageIn) + atS[i], src[i].pixels, src[i].rowPitch * src[i].height, hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->stageOut, 0, totalD, ctx->stream));
    hr = submit_coverage_chain(ctx, s, d, src, dst, nlevels, alphaReference);
    if (hr != DXTEX_S_OK) return hr;
    for (size_t i = 0; i < nlevels; ++i)
        HIP_TRY(ctx, counted_copy(ctx, dst[i].pixels, d[i], dst[i].rowPitch * dst[i].height, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_compute_mse_device(dxtex_ctx* ctx, const dxtex_image* a, const dxtex_image* b, double mse[4])
{
    dxtex_hresult hr = check_pair(ctx, a, b);
    if (hr != DXTEX_S_OK) return hr;
    if (!mse) return fail(ctx, DXTEX_E_POINTER, "null result");
    const FmtInfo* fa = format_info(a->format);
    const FmtInfo* fb = format_info(b->format);
    if (!fa || !fb || (fa->cls & FC_BC) || (fb->cls & FC_BC)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "ComputeMSE takes uncompressed images (decompress first)");
    ScopedDevice sd(ctx->device);
    hr = ensure(ctx, &ctx->mseBuf, &ctx->mseBytes, 4 * sizeof(double)); if (hr != DXTEX_S_OK) return hr;
    hipError_t e = launch_mse(a->pixels, a->rowPitch, a->format, b->pixels, b->rowPitch, b->format, uint32_t(a->width), uint32_t(a->height),
                              static_cast<double*>(ctx->mseBuf), ctx->stream);
    if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    double sum[4];
    HIP_TRY(ctx, counted_copy(ctx, sum, ctx->mseBuf, sizeof(sum), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const double n = double(a->width) * double(a->height);
    for (int c = 0; c < 4; ++c) mse[c] = sum[c] / n;
    return DXTEX_S_OK;
}

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
    HIP_TRY(ctx, counted_copy(ctx, dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_memcpy_d2h(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, counted_copy(ctx, dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return DXTEX_S_OK;
}

// ---- the device-resident pipeline's helpers -------------------------------------------------------------------------------------------
dxtex_hresult dxtex_device_memset(dxtex_ctx* ctx, void* p, int value, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!p) return fail(ctx, DXTEX_E_POINTER, "null buffer");
    ScopedDevice sd(ctx->device);
    if (bytes) HIP_TRY(ctx, hipMemsetAsync(p, value, bytes, ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_copy_rows_device(dxtex_ctx* ctx, void* dst, size_t dstPitch, const void* src, size_t srcPitch, size_t rowBytes, size_t rows)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!dst || !src) return fail(ctx, DXTEX_E_POINTER, "null buffer");
    if (rowBytes > dstPitch || rowBytes > srcPitch) return fail(ctx, DXTEX_E_INVALIDARG, "row is wider than a pitch");
    ScopedDevice sd(ctx->device);
    if (rowBytes && rows) HIP_TRY(ctx, hipMemcpy2DAsync(dst, dstPitch, src, srcPitch, rowBytes, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_memcpy_h2d_async(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, counted_copy(ctx, dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_memcpy_d2h_async(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, counted_copy(ctx, dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_host_alloc(dxtex_ctx* ctx, size_t bytes, void** out)
{
    if (!ctx || !out) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return DXTEX_S_OK;
}
dxtex_hresult dxtex_host_free(dxtex_ctx* ctx, void* p)
{
    if (!ctx) return DXTEX_E_POINTER;
    ScopedDevice sd(ctx->device);
    HIP_TRY(ctx, hipHostFree(p));
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_alpha_all_opaque_device(dxtex_ctx* ctx, const dxtex_image* images, size_t count, int* opaque)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (!opaque) return fail(ctx, DXTEX_E_POINTER, "null result");
    *opaque = 0;
    if (!images || !count) return fail(ctx, DXTEX_E_INVALIDARG, "no images");
    const FmtInfo* f = format_info(images[0].format);
    if (!f || (f->cls & FC_GROUP)) return fail(ctx, DXTEX_E_NOT_SUPPORTED, "format is not supported by the MI355X path");
    // HasAlpha (DirectXTexUtil.cpp:340-372): of the BC formats BC1 / BC2 / BC3 / BC7 carry alpha; a format without alpha is opaque (:805-806)
    const bool bc = (f->cls & FC_BC) != 0;
    const bool bcAlpha = bc && bc_block_bytes(f->format) && f->format != FMT_BC4_UNORM && f->format != FMT_BC4_SNORM && f->format != FMT_BC5_UNORM &&
                         f->format != FMT_BC5_SNORM && f->format != FMT_BC6H_UF16 && f->format != FMT_BC6H_SF16;
    if ((bc && !bcAlpha) || (!bc && !(f->cls & FC_A))) { *opaque = 1; return DXTEX_S_OK; }
    ScopedDevice sd(ctx->device);
    dxtex_hresult hr = ensure(ctx, &ctx->mseBuf, &ctx->mseBytes, 4 * sizeof(double)); if (hr != DXTEX_S_OK) return hr;
    unsigned long long* counter = static_cast<unsigned long long*>(ctx->mseBuf);
    HIP_TRY(ctx, hipMemsetAsync(counter, 0, sizeof(unsigned long long), ctx->stream));
    for (size_t i = 0; i < count; ++i)
    {
        const dxtex_image& im = images[i];
        if (!im.pixels) return fail(ctx, DXTEX_E_POINTER, "null pixels");
        if (im.format != images[0].format) return fail(ctx, DXTEX_E_FAIL, "format mismatch");
        if (!im.width || !im.height || im.width > 0xFFFFFFFFull || im.height > 0xFFFFFFFFull) return fail(ctx, DXTEX_E_INVALIDARG, "image size");
        hipError_t e;
        if (bc)
        {
            // IsAlphaAllOpaqueBC decodes every block to floats and tests the texels inside the image against 0.99
            const size_t pitch = im.width * 16;
            hr = ensure(ctx, &ctx->stageOut, &ctx->stageOutBytes, pitch * im.height); if (hr != DXTEX_S_OK) return hr;
            hr = submit_decompress(ctx, im.pixels, im.format, im.rowPitch, static_cast<uint8_t*>(ctx->stageOut), FMT_R32G32B32A32_FLOAT, pitch, im.width, im.height);
            if (hr != DXTEX_S_OK) return hr;
            e = launch_alpha_below(static_cast<const uint8_t*>(ctx->stageOut), pitch, FMT_R32G32B32A32_FLOAT, uint32_t(im.width), uint32_t(im.height), 0.99f, counter, ctx->stream);
        }
        else
            e = launch_alpha_below(im.pixels, im.rowPitch, im.format, uint32_t(im.width), uint32_t(im.height), 0.997f, counter, ctx->stream);
        if (e != hipSuccess) return fail(ctx, DXTEX_E_FAIL, "kernel launch failed", e);
    }
    unsigned long long n = 0;
    HIP_TRY(ctx, counted_copy(ctx, &n, counter, sizeof(n), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *opaque = n ? 0 : 1;
    return DXTEX_S_OK;
}

dxtex_hresult dxtex_ctx_transfer_bytes(dxtex_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes, int reset)
{
    if (!ctx) return DXTEX_E_POINTER;
    if (h2d_bytes) *h2d_bytes = ctx->h2dBytes;
    if (d2h_bytes) *d2h_bytes = ctx->d2hBytes;
    if (reset) ctx->h2dBytes = ctx->d2hBytes = 0;
    return DXTEX_S_OK;
}
} // extern "C"

// ---- ONE image over several contexts (one process, N GPUs - or N contexts of one GPU) -------------------------------------------------
// Blocks are independent (DirectXTexCompress.cpp:257-281 is the reference's own split of an image's block rows over OpenMP threads), and a
// destination row of a 2:1 filter needs a bounded window of source rows (cubic: one source row above the pair it covers and one below,
// filters.h:176-179). So an image in host memory is cut into stripes of (block) rows, one per context, each stripe goes through the
// single-context entry point on a thread of its own, and the stripes land in the caller's destination: the bytes of the one-context call.
namespace
{
// Returns the first failing stripe's result and, through *failed, its index (so that the caller can surface that context's error text).
template<class F>
dxtex_hresult run_stripes(size_t n, F&& stripe, size_t* failed = nullptr)
{
    std::vector<dxtex_hresult> hr(n, DXTEX_S_OK);
    std::vector<std::thread> workers;
    // (a stripe that runs out of host memory reports it like any other failure: nothing may leave a worker thread as an exception)
    auto guarded = [&](size_t i) { try { hr[i] = stripe(i); } catch (...) { hr[i] = DXTEX_E_OUTOFMEMORY; } };
    try { for (size_t i = 1; i < n; ++i) workers.emplace_back(guarded, i); }
    catch (...) { for (std::thread& t : workers) t.join(); return DXTEX_E_OUTOFMEMORY; }
    guarded(0);
    for (std::thread& t : workers) t.join();
    for (size_t i = 0; i < n; ++i) if (hr[i] != DXTEX_S_OK) { if (failed) *failed = i; return hr[i]; }
    return DXTEX_S_OK;
}

// Every stripe runs on a context of its own (a context's staging buffers and stream serve one call at a time): the same context twice is
// the caller's error. A failing stripe's text is copied to the first context, where the caller of a multi-context call looks for it.
dxtex_hresult check_distinct(dxtex_ctx* const* ctxs, size_t nctx)
{
    for (size_t i = 0; i < nctx; ++i)
        for (size_t j = i + 1; j < nctx; ++j)
            if (ctxs[i] == ctxs[j]) return fail(ctxs[0], DXTEX_E_INVALIDARG, "the same context listed twice");
    return DXTEX_S_OK;
}
dxtex_hresult surface_stripe_error(dxtex_ctx* const* ctxs, dxtex_hresult hr, size_t failed)
{
    if (hr != DXTEX_S_OK && failed != 0)
    {
        try { ctxs[0]->lastError = "stripe " + std::to_string(failed) + ": " + ctxs[failed]->lastError; } catch (...) { }
    }
    return hr;
}
} // namespace

extern "C" {
dxtex_hresult dxtex_compress_multi(dxtex_ctx* const* ctxs, size_t nctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags, float threshold)
{
    if (!ctxs || !nctx || !ctxs[0]) return DXTEX_E_POINTER;
    for (size_t i = 0; i < nctx; ++i) if (!ctxs[i]) return DXTEX_E_POINTER;
    if (!src || !dst) return fail(ctxs[0], DXTEX_E_INVALIDARG, "null image");
    const size_t nbh = (src->height + 3) / 4;
    const size_t n = std::max<size_t>(1, std::min(nctx, nbh));
    if (n == 1 || src->width != dst->width || src->height != dst->height) return dxtex_compress(ctxs[0], src, dst, flags, threshold);      // (a mismatch is the single call's error to report)
    // the whole image is validated ONCE, before any stripe pointer is formed from it (a stripe's `pixels + y0 * rowPitch` of a null image
    // would pass the stripes' own null test): the checks of the single-context call, on the first context
    dxtex_hresult hr = check_pair(ctxs[0], src, dst);
    if (hr != DXTEX_S_OK) return hr;
    { size_t sb = 0, db = 0; hr = check_host_pitches(ctxs[0], src, dst, &sb, &db); if (hr != DXTEX_S_OK) return hr; }
    hr = check_distinct(ctxs, n); if (hr != DXTEX_S_OK) return hr;
    size_t failed = 0;
    hr = run_stripes(n, [&](size_t i) -> dxtex_hresult
    {
        const size_t b0 = nbh * i / n, b1 = nbh * (i + 1) / n;            // block rows [b0, b1)
        if (b1 <= b0) return DXTEX_S_OK;
        const size_t y0 = b0 * 4, rows = std::min(src->height, b1 * 4) - y0;
        dxtex_image s = *src, d = *dst;
        s.pixels = src->pixels + y0 * src->rowPitch; s.height = rows; s.slicePitch = src->rowPitch * rows;
        d.pixels = dst->pixels + b0 * dst->rowPitch; d.height = rows; d.slicePitch = dst->rowPitch * (b1 - b0);
        return dxtex_compress(ctxs[i], &s, &d, flags, threshold);
    }, &failed);
    return surface_stripe_error(ctxs, hr, failed);
}

dxtex_hresult dxtex_generate_mips_multi(dxtex_ctx* const* ctxs, size_t nctx, const dxtex_image* levels, size_t nlevels, uint32_t filter)
{
    if (!ctxs || !nctx) return DXTEX_E_POINTER;
    for (size_t i = 0; i < nctx; ++i) if (!ctxs[i]) return DXTEX_E_POINTER;
    uint32_t mode = 0;
    const dxtex_hresult hc = check_mips(ctxs[0], levels, nlevels, filter, &mode);
    if (hc != DXTEX_S_OK) return hc;
    { const dxtex_hresult hd = check_distinct(ctxs, nctx); if (hd != DXTEX_S_OK) return hd; }
    const uint32_t explicitFilter = (filter & ~kFilterModeMask) | mode;      // the sub-chains below must not choose again (a stripe is not a power of two high)
    // A level is split while it is an exact halving, large enough to be worth a context of its own, and filtered by a kernel whose taps
    // are a fixed window around the destination row: point / box (the 2 x 2 source texels), linear and cubic with clamp addressing in V
    // (u = (y + 0.5) * 2 - 0.5 is exact in fp32, so a stripe's rows get the weights the whole image's rows get). The triangle filter (a
    // gather over the whole axis) and V wrap / mirror go to the first context whole, as does the chain below kSplitMinRows.
    //
    // Round 6: the stripes stay RESIDENT. A context computes rows of every split level from its own copy of the level above and never
    // exchanges anything: destination rows [a, b) of a level come from source rows [2 a - 2, 2 b + 2) of the level above (cubic reads one
    // row above the pair it covers and one below, filters.h:176-179; a sub-image is resized with one throw-away destination row on either
    // inner side, whose taps run into the sub-image's clamped edge), so working back from the stripe a context owns at the LAST split level
    // gives the rows it needs of every level above - its own stripe plus a halo that doubles per level (3 * 2^k rows k levels up: 48 rows
    // of level 0 for four split levels, against stripes of hundreds). One upload per context (its rows of level 0), the levels chained on
    // the device, one download per context (its stripes of levels 1 ... L): every byte crosses the host link once, and on N GPUs no byte
    // crosses between them. (Round 5 went host -> device -> host once per level and stripe.)
    constexpr size_t kSplitMinRows = 256;
    const bool splittable = nctx > 1 && (mode == DXTEX_FILTER_POINT || mode == DXTEX_FILTER_BOX || mode == DXTEX_FILTER_LINEAR || mode == DXTEX_FILTER_CUBIC) &&
                            !(filter & (DXTEX_FILTER_WRAP_V | DXTEX_FILTER_MIRROR_V));
    size_t lv = 1;      // levels [1, lv) are split
    for (; splittable && lv < nlevels; ++lv)
    {
        const dxtex_image& S = levels[lv - 1];
        const dxtex_image& D = levels[lv];
        if (S.width != 2 * D.width || S.height != 2 * D.height || D.height < kSplitMinRows || D.height < 4 * nctx) break;
    }
    const size_t L = lv - 1;                          // the last split level (0: nothing to split)
    if (L >= 1)
    {
        struct Rows { size_t a, b; };
        size_t failed = 0;
        const dxtex_hresult hr = run_stripes(nctx, [&](size_t i) -> dxtex_hresult
        {
            dxtex_ctx* ctx = ctxs[i];
            auto owned = [&](size_t l) { const size_t H = levels[l].height; return Rows{ H * i / nctx, H * (i + 1) / nctx }; };
            // C[l]: rows of level l this context must hold CORRECT; E[l]: rows it computes (C[l] plus the throw-away rows); S[l - 1] = [2 E[l].a, 2 E[l].b):
            // the sub-image of level l - 1 they are computed from
            std::vector<Rows> C(L + 1), E(L + 1);
            C[L] = owned(L);
            for (size_t l = L; l >= 1; --l)
            {
                const size_t H = levels[l].height;
                E[l] = Rows{ C[l].a ? C[l].a - 1 : 0, std::min(C[l].b + 1, H) };
                C[l - 1] = Rows{ 2 * E[l].a, 2 * E[l].b };
                if (l - 1 >= 1)                       // the level above is an output too: its own stripe must be inside what is held (it is: the halo grows faster than the stripes' rounding)
                {
                    const Rows o = owned(l - 1);
                    C[l - 1].a = std::min(C[l - 1].a, o.a); C[l - 1].b = std::max(C[l - 1].b, o.b);
                }
            }
            E[0] = C[0];
            if (C[L].b <= C[L].a) return DXTEX_S_OK;
            // one arena for the context's rows of every level (the staging buffer of the single-image calls, reused across calls)
            std::vector<size_t> at(L + 1);
            size_t bytes = 0;
            for (size_t l = 0; l <= L; ++l) { at[l] = bytes; bytes += ((E[l].b - E[l].a) * levels[l].rowPitch + 255) & ~size_t(255); }
            ScopedDevice sd(ctx->device);
            dxtex_hresult h = ensure(ctx, &ctx->stageIn, &ctx->stageInBytes, bytes); if (h != DXTEX_S_OK) return h;
            uint8_t* arena = static_cast<uint8_t*>(ctx->stageIn);
            // rows [a, b) of level l as one copy: whole pitches, the last row only as far as its texels go (a host image may end there)
            auto span = [&](size_t l, const Rows& r) -> size_t
            {
                size_t minRow = 0, minSlice = 0;
                (void)dxtex_compute_pitch(levels[l].format, levels[l].width, 1, &minRow, &minSlice);
                return (r.b - r.a - 1) * levels[l].rowPitch + minRow;
            };
            HIP_TRY(ctx, counted_copy(ctx, arena + at[0], levels[0].pixels + E[0].a * levels[0].rowPitch, span(0, E[0]), hipMemcpyHostToDevice, ctx->stream));
            time_begin(ctx);
            for (size_t l = 1; l <= L; ++l)
            {
                const size_t s0 = 2 * E[l].a;         // first row of the source sub-image, inside [E[l - 1].a, E[l - 1].b)
                std::vector<LevelPair> pairs{ { arena + at[l - 1] + (s0 - E[l - 1].a) * levels[l - 1].rowPitch, levels[l - 1].rowPitch, levels[l - 1].width, 2 * (E[l].b - E[l].a),
                                                arena + at[l], levels[l].rowPitch, levels[l].width, E[l].b - E[l].a } };
                h = submit_resizes(ctx, pairs, levels[0].format, mode, explicitFilter, false);
                if (h != DXTEX_S_OK) { time_end(ctx); return h; }
            }
            time_end(ctx);
            for (size_t l = 1; l <= L; ++l)
            {
                const Rows o = owned(l);
                if (o.b <= o.a) continue;
                HIP_TRY(ctx, counted_copy(ctx, levels[l].pixels + o.a * levels[l].rowPitch, arena + at[l] + (o.a - E[l].a) * levels[l].rowPitch, span(l, o), hipMemcpyDeviceToHost, ctx->stream));
            }
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return DXTEX_S_OK;
        }, &failed);
        if (hr != DXTEX_S_OK) return surface_stripe_error(ctxs, hr, failed);
    }
    if (lv >= nlevels) return DXTEX_S_OK;
    return dxtex_generate_mips(ctxs[0], levels + (lv - 1), nlevels - (lv - 1), explicitFilter);       // the rest of the chain from the last level filled
}
} // extern "C"
