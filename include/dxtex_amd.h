/*
 * dxtex_amd.h - C ABI of the MI355X-native DirectXTex hot path (libdxtex_amd.so).
 *
 * Drop-in boundary: these entry points are what a maintainer of microsoft/DirectXTex would bind in place
 * of the library's own GPU plugin (GPUCompressBC, DirectXTex/BCDirectCompute.h:15-68) and of the CPU
 * loops behind Compress / Decompress / GenerateMipMaps / Convert / Resize. Plain pointers and sizes only;
 * every function returns an HRESULT with the reference's values (DirectXTexP.h:210-234).
 *
 *   reference interface                                         replaced by
 *   ----------------------------------------------------------  -----------------------------------------
 *   GPUCompressBC::Initialize(ID3D11Device*)   BCDirectCompute.h:26   dxtex_ctx_create
 *   GPUCompressBC::Prepare(w,h,flags,fmt,aw)   BCDirectCompute.h:28   dxtex_ctx_prepare (optional: sizes the search scratch and the staging now instead of
 *                                                                     on first use; alphaWeight belongs to the D3D11 BC7 shader and has no counterpart in
 *                                                                     the CPU-path encoders reproduced here)
 *   GPUCompressBC::Compress(src,dst)           BCDirectCompute.h:30   dxtex_compress / dxtex_compress_device
 *   CompressBC / CompressBC_Parallel           DirectXTexCompress.cpp:72-372   dxtex_compress
 *   DecompressBC                               DirectXTexCompress.cpp:425-535  dxtex_decompress
 *   BC_ENCODE / BC_DECODE fn-ptr hooks         BC.h:318-343           dxtex_encode_blocks / dxtex_decode_blocks
 *   Generate2DMips{Point,Box,Linear,Cubic,Triangle}Filter  DirectXTexMipmaps.cpp:907-1602  dxtex_generate_mips
 *   ConvertCustom                              DirectXTexConvert.cpp:4804-4913 dxtex_convert
 *   Resize*Filter                              DirectXTexResize.cpp:255-803    dxtex_resize
 *
 * Threading: a context is bound to one GPU and one HIP stream; use one context per GPU (or per host
 * thread). Contexts share nothing. No function retains caller pointers past its return, except the
 * *_device variants, which are asynchronous on the context's stream.
 */
#ifndef DXTEX_AMD_H
#define DXTEX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t dxtex_hresult;

#define DXTEX_S_OK                 ((dxtex_hresult)0)
#define DXTEX_E_FAIL               ((dxtex_hresult)0x80004005)
#define DXTEX_E_INVALIDARG         ((dxtex_hresult)0x80070057)
#define DXTEX_E_OUTOFMEMORY        ((dxtex_hresult)0x8007000E)
#define DXTEX_E_POINTER            ((dxtex_hresult)0x80004003)
#define DXTEX_E_ABORT              ((dxtex_hresult)0x80004004)
#define DXTEX_E_NOTIMPL            ((dxtex_hresult)0x80004001)
#define DXTEX_E_UNEXPECTED         ((dxtex_hresult)0x8000FFFF)
#define DXTEX_E_NOT_SUPPORTED      ((dxtex_hresult)0x80070032)  /* HRESULT_FROM_WIN32(ERROR_NOT_SUPPORTED) */
#define DXTEX_E_ARITHMETIC_OVERFLOW ((dxtex_hresult)0x80070216)

/* TEX_COMPRESS_FLAGS, bit-for-bit (DirectXTex.h:887-917). */
#define DXTEX_COMPRESS_DEFAULT          0u
#define DXTEX_COMPRESS_RGB_DITHER       0x10000u
#define DXTEX_COMPRESS_A_DITHER         0x20000u
#define DXTEX_COMPRESS_DITHER           0x30000u
#define DXTEX_COMPRESS_UNIFORM          0x40000u
#define DXTEX_COMPRESS_BC7_USE_3SUBSETS 0x80000u
#define DXTEX_COMPRESS_BC7_QUICK        0x100000u
#define DXTEX_COMPRESS_SRGB_IN          0x1000000u
#define DXTEX_COMPRESS_SRGB_OUT         0x2000000u
#define DXTEX_COMPRESS_PARALLEL         0x10000000u

/* TEX_FILTER_FLAGS subset (DirectXTex.h:741-793). */
#define DXTEX_FILTER_DEFAULT   0u
#define DXTEX_FILTER_WRAP_U    0x1u
#define DXTEX_FILTER_WRAP_V    0x2u
#define DXTEX_FILTER_MIRROR_U  0x10u
#define DXTEX_FILTER_MIRROR_V  0x20u
#define DXTEX_FILTER_POINT     0x100000u
#define DXTEX_FILTER_LINEAR    0x200000u
#define DXTEX_FILTER_CUBIC     0x300000u
#define DXTEX_FILTER_BOX       0x400000u
#define DXTEX_FILTER_TRIANGLE  0x500000u
#define DXTEX_FILTER_MODE_MASK 0xF00000u

/* Mirrors DirectX::Image (DirectXTex.h:437-445); `format` is a DXGI_FORMAT value. */
typedef struct dxtex_image
{
    size_t   width;
    size_t   height;
    int32_t  format;
    size_t   rowPitch;
    size_t   slicePitch;
    uint8_t* pixels;
} dxtex_image;

typedef struct dxtex_ctx dxtex_ctx;

/* ---- context ---------------------------------------------------------------------------------- */

/* Binds a context to HIP device `device` and creates its stream. Fails with DXTEX_E_FAIL when no
 * gfx950-capable device is visible: there is no CPU fallback anywhere in this library. */
dxtex_hresult dxtex_ctx_create(int device, dxtex_ctx** out);
void          dxtex_ctx_destroy(dxtex_ctx* ctx);
/* Run subsequent work on a caller-owned hipStream_t (e.g. the current PyTorch stream); NULL restores
 * the context's own stream. Waits for the work queued on the previous stream, which must outlive this call (a handle
 * the runtime no longer knows is tolerated); DXTEX_E_FAIL, stream unchanged, if that work failed asynchronously. */
dxtex_hresult dxtex_ctx_set_stream(dxtex_ctx* ctx, void* hip_stream);
void*         dxtex_ctx_get_stream(dxtex_ctx* ctx);
dxtex_hresult dxtex_ctx_synchronize(dxtex_ctx* ctx);
/* Human-readable description of the last failure on this context (never NULL). */
const char*   dxtex_ctx_last_error(dxtex_ctx* ctx);
/* Device time in milliseconds of the kernels launched by the most recent call on this context
 * (hipEvent pair on the context's stream, transfers excluded); -1 if nothing was timed. */
float         dxtex_ctx_last_kernel_ms(dxtex_ctx* ctx);

/* Per-kernel device timing. Between profile_begin and profile_end every kernel this context launches is
 * bracketed by hipEvents on the launch stream. profile_end synchronises the stream and returns, per
 * distinct kernel, the summed duration in ms and the number of launches; `names` receives the kernel
 * names separated by '\n'. Used by bench.py for the roofline object. */
dxtex_hresult dxtex_ctx_profile_begin(dxtex_ctx* ctx);
dxtex_hresult dxtex_ctx_profile_end(dxtex_ctx* ctx, char* names, size_t names_bytes, float* total_ms,
                                    uint32_t* launches, size_t capacity, size_t* count);

/* ---- format utilities (DirectXTexUtil.cpp:340-1186) ------------------------------------------- */

int           dxtex_is_compressed(int32_t format);
size_t        dxtex_bits_per_pixel(int32_t format);
/* ComputePitch (DirectXTexUtil.cpp:961-1186) with CP_FLAGS_NONE. */
dxtex_hresult dxtex_compute_pitch(int32_t format, size_t width, size_t height, size_t* rowPitch, size_t* slicePitch);

/* ---- Compress / Decompress -------------------------------------------------------------------- */

/* The counterpart of GPUCompressBC::Prepare (BCDirectCompute.h:31, BCDirectCompute.cpp:203-369; called once per size from
 * DirectXTexCompressGPU.cpp:392-442): allocates everything a later dxtex_compress* of `count` images of this size and these
 * formats needs - the BC6H / BC7 search scratch and, for the host-pointer entry points, the device staging buffers - so that
 * the compress calls themselves allocate nothing - and, as Prepare binds its shaders, runs one block of zeros through the format's
 * pipeline once per context so that the kernels' code objects and the side streams exist before the first real call (a fresh process:
 * first 64 x 64 BC7 call 19 ms without, about 1 ms with). Optional: without it the buffers grow on first use. Same format checks and
 * error codes as dxtex_compress. Returns the bytes of device memory the context now holds in `device_bytes` (may be NULL). */
dxtex_hresult dxtex_ctx_prepare(dxtex_ctx* ctx, size_t width, size_t height, int32_t src_format, int32_t dst_format,
                                uint32_t compress_flags, size_t count, size_t* device_bytes);

/* Host images in, host image out: H2D copy, kernels, D2H copy, synchronous. `dst` must already describe
 * a BC image of the same width/height (ScratchImage::Initialize2D layout). Error behaviour follows
 * CompressBC (DirectXTexCompress.cpp:72-205): E_POINTER for null pixels, HRESULT_E_NOT_SUPPORTED for an
 * unsupported source/destination format, E_INVALIDARG for a compressed source. */
dxtex_hresult dxtex_compress(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst,
                             uint32_t compress_flags, float threshold);

/* Same, but `src->pixels` / `dst->pixels` are device pointers on the context's GPU; asynchronous on the
 * context's stream. This is the entry point the benchmark times (inputs resident in HBM). */
dxtex_hresult dxtex_compress_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst,
                                    uint32_t compress_flags, float threshold);

/* Batch of `count` independent images (texture array, mip chain, atlas pages: the array overload of Compress,
 * DirectXTexCompress.cpp:722-846) on device memory; one stream-ordered submission. BC6H / BC7 arrays go through the
 * search pipeline as one block list, so many small images cost what one image of the same total size costs. */
dxtex_hresult dxtex_compress_many_device(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts,
                                         size_t count, uint32_t compress_flags, float threshold);
/* Same with host pointers; returns when the payloads are back. The images are cut into chunks of about 32 Mi texels
 * (DXTEX_MANY_CHUNK_TEXELS overrides); two sets of pinned staging + device buffers alternate so that the upload of chunk k+1
 * and the download of chunk k-1 run on copy streams while chunk k's kernels run. The bytes written are those of
 * `count` separate dxtex_compress calls. */
dxtex_hresult dxtex_compress_many(dxtex_ctx* ctx, const dxtex_image* srcs, const dxtex_image* dsts,
                                  size_t count, uint32_t compress_flags, float threshold);

/* BC -> uncompressed (R8G8B8A8_UNORM, R16G16B16A16_FLOAT, R32G32B32A32_FLOAT, R8_UNORM/SNORM, R8G8_*).
 * Host and device variants as above. */
dxtex_hresult dxtex_decompress(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst);
dxtex_hresult dxtex_decompress_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst);

/* Block-level hooks with the reference's BC_ENCODE / BC_DECODE shape (BC.h:318-343): `rgba` holds
 * nblocks x 16 texels x 4 floats (row-major 4x4 tiles) in host memory, `bc` nblocks x 8|16 bytes.
 * `threshold` is only read for BC1. */
dxtex_hresult dxtex_encode_blocks(dxtex_ctx* ctx, int32_t bc_format, uint32_t bc_flags, float threshold,
                                  const float* rgba, size_t nblocks, uint8_t* bc);
dxtex_hresult dxtex_decode_blocks(dxtex_ctx* ctx, int32_t bc_format, const uint8_t* bc, size_t nblocks, float* rgba);

/* ---- GenerateMipMaps / Convert / Resize -------------------------------------------------------- */

/* Fills levels[1..nlevels-1] from levels[0] (each level from the previous *stored* level,
 * DirectXTexMipmaps.cpp:1036-1037). All levels share levels[0].format. `filter` = TEX_FILTER_FLAGS. */
dxtex_hresult dxtex_generate_mips(dxtex_ctx* ctx, const dxtex_image* levels, size_t nlevels, uint32_t filter);
dxtex_hresult dxtex_generate_mips_device(dxtex_ctx* ctx, const dxtex_image* levels, size_t nlevels, uint32_t filter);

/* GenerateMipMaps3D (DirectXTex.h:853-858, DirectXTexMipmaps.cpp:3254-3361): volume textures. A level is `depth` slices of
 * `slicePitch` bytes starting at `pixels` (the layout ScratchImage::Initialize3D gives a level, DirectXTexImage.cpp:228-262);
 * levels[0] holds the base slices, levels[1..] are filled, each from the STORED previous level. Level i is
 * max(1, w>>i) x max(1, h>>i) x max(1, d>>i). filter 0 = box when all three dimensions are powers of two, else triangle. */
typedef struct dxtex_volume
{
    size_t   width, height, depth;
    int32_t  format;
    size_t   rowPitch, slicePitch;
    uint8_t* pixels;
} dxtex_volume;
dxtex_hresult dxtex_generate_mips3d(dxtex_ctx* ctx, const dxtex_volume* levels, size_t nlevels, uint32_t filter);
dxtex_hresult dxtex_generate_mips3d_device(dxtex_ctx* ctx, const dxtex_volume* levels, size_t nlevels, uint32_t filter);

dxtex_hresult dxtex_convert(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, float threshold);
dxtex_hresult dxtex_convert_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter, float threshold);

dxtex_hresult dxtex_resize(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter);
dxtex_hresult dxtex_resize_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t filter);

/* ---- one image over several contexts (in-process strong scaling) --------------------------------
 * The role of the reference's own split of one image over its workers (CompressBC_Parallel hands block rows to OpenMP threads,
 * DirectXTexCompress.cpp:257-281): `ctxs` are nctx contexts - normally one per GPU of the node, several on one GPU work too - and the
 * image in HOST memory is cut into stripes of block rows (Compress) or destination rows with the filter's halo of source rows
 * (GenerateMipMaps: exact-halving levels of at least 256 rows, point / box / linear / cubic, no V wrap / mirror; everything else runs on
 * ctxs[0]). Each stripe goes through the single-context entry point on its own thread; the result is byte for byte that of
 * dxtex_compress / dxtex_generate_mips on one context. Errors: the first failing stripe's code (its text in that context's last_error). */
dxtex_hresult dxtex_compress_multi(dxtex_ctx* const* ctxs, size_t nctx, const dxtex_image* src, const dxtex_image* dst,
                                   uint32_t flags, float threshold);
dxtex_hresult dxtex_generate_mips_multi(dxtex_ctx* const* ctxs, size_t nctx, const dxtex_image* levels, size_t nlevels, uint32_t filter);

/* ComputeMSE (DirectXTexMisc.cpp:27-176) for two same-size images on the device: per-channel MSE in
 * mse[4] over [0,1] floats. Used for PSNR reporting without a D2H round trip. */
dxtex_hresult dxtex_compute_mse_device(dxtex_ctx* ctx, const dxtex_image* a, const dxtex_image* b, double mse[4]);

/* PremultiplyAlpha / its REVERSE (DirectXTex.h:864-884, DirectXTexPMAlpha.cpp:214-262): same size and format on both sides,
 * the format must carry alpha (else DXTEX_E_NOT_SUPPORTED). `flags` = TEX_PMALPHA_FLAGS. */
#define DXTEX_PMALPHA_DEFAULT      0x0u
#define DXTEX_PMALPHA_IGNORE_SRGB  0x1u
#define DXTEX_PMALPHA_REVERSE      0x2u
#define DXTEX_PMALPHA_SRGB_IN      0x1000000u
#define DXTEX_PMALPHA_SRGB_OUT     0x2000000u
dxtex_hresult dxtex_premultiply_alpha(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags);
dxtex_hresult dxtex_premultiply_alpha_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, uint32_t flags);

/* ScaleMipMapsAlphaForCoverage (DirectXTex.h:848-851, DirectXTexMipmaps.cpp:3483-3556) for one mip chain: dst[0] = src[0];
 * every further level gets its alpha scaled so that its coverage at `alphaReference` matches level 0's (10-step bisection,
 * :310-352). src[i] and dst[i] have the same size and format. */
dxtex_hresult dxtex_scale_mips_alpha_for_coverage(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t nlevels, float alphaReference);
dxtex_hresult dxtex_scale_mips_alpha_for_coverage_device(dxtex_ctx* ctx, const dxtex_image* src, const dxtex_image* dst, size_t nlevels, float alphaReference);

/* ---- device memory helpers (so non-HIP hosts can stay resident in HBM) -------------------------- */
dxtex_hresult dxtex_device_alloc(dxtex_ctx* ctx, size_t bytes, void** out);
dxtex_hresult dxtex_device_free(dxtex_ctx* ctx, void* p);
dxtex_hresult dxtex_memcpy_h2d(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes);
dxtex_hresult dxtex_memcpy_d2h(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes);

/* ---- the device-resident pipeline (texconv's resize -> convert -> mipmaps -> compress chain, Texconv/texconv.cpp:2609, 3109, 3434,
 * 3711, with ONE upload of the source and ONE download of the final payload; the reference's own GPU path keeps its intermediate
 * in device memory the same way, DirectXTexCompressGPU.cpp:34-140) -------------------------------------------------------------
 * The *_device entry points above are the steps; these are what a host needs around them. All are stream-ordered on the context's
 * stream unless stated. */
/* hipMemsetAsync on the context's stream: a device image starts zero-filled like ScratchImage's memory (DirectXTexImage.cpp:376). */
dxtex_hresult dxtex_device_memset(dxtex_ctx* ctx, void* p, int value, size_t bytes);
/* Device-to-device copy of `rows` rows of `rowBytes` bytes between two pitched images (Setup2DMips' copy of the base image into
 * the top of the chain, DirectXTexMipmaps.cpp:851-904). */
dxtex_hresult dxtex_copy_rows_device(dxtex_ctx* ctx, void* dst, size_t dstPitch, const void* src, size_t srcPitch, size_t rowBytes, size_t rows);
/* Asynchronous transfers on the context's stream (host memory from dxtex_host_alloc is page-locked, so these overlap with the
 * kernels of other contexts of the same GPU and return at once; with pageable memory they are staged by the runtime). The host
 * buffer must stay valid until dxtex_ctx_synchronize. */
dxtex_hresult dxtex_memcpy_h2d_async(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes);
dxtex_hresult dxtex_memcpy_d2h_async(dxtex_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Page-locked host memory for the two ends of the pipeline. */
dxtex_hresult dxtex_host_alloc(dxtex_ctx* ctx, size_t bytes, void** out);
dxtex_hresult dxtex_host_free(dxtex_ctx* ctx, void* p);
/* ScratchImage::IsAlphaAllOpaque (DirectXTexImage.cpp:800-852) over `count` device images of one format: *opaque = 1 when every
 * texel's alpha is >= 0.997 (uncompressed; LoadScanline's alpha) or >= 0.99 (BC1 / BC2 / BC3 / BC7, decoded as IsAlphaAllOpaqueBC
 * does, DirectXTexCompress.cpp:537-625), or when the format has no alpha. Synchronises the stream (4 bytes come back). */
dxtex_hresult dxtex_alpha_all_opaque_device(dxtex_ctx* ctx, const dxtex_image* images, size_t count, int* opaque);
/* Bytes this context has moved over PCIe since its creation or the last reset (every host <-> device copy the library issues for
 * it, staging and tables included): what tests/ and bench.py use to show that a resident pipeline uploads the source once and
 * downloads the payload once. */
dxtex_hresult dxtex_ctx_transfer_bytes(dxtex_ctx* ctx, uint64_t* h2d_bytes, uint64_t* d2h_bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif /* DXTEX_AMD_H */
