// Scanline-layer kernels for gfx950: Convert, Resize / GenerateMipMaps filters, ComputeMSE.
//
//   reference                                                             here
//   ConvertCustom (DirectXTexConvert.cpp:4804-4913, no-dither branch)      convert_kernel
//   Resize{Point,Box,Linear,Cubic,Triangle}Filter (DirectXTexResize.cpp:255-803) and
//   Generate2DMips{Point,Box,Linear,Cubic,Triangle}Filter (DirectXTexMipmaps.cpp:907-1602)   resize_*_kernel
//   ComputeMSE_ (DirectXTexMisc.cpp:27-176)                                mse_kernel
//
// The reference walks scanlines through a float4 row buffer (LoadScanline -> filter -> StoreScanline). Here every
// lane owns one destination texel and reads the source texels it needs straight from HBM/L2 with LoadScanline's
// per-texel arithmetic; nothing is staged, so a level costs one read of the source footprint (the 2x2 / 4x4
// neighbourhoods of adjacent lanes overlap in L1/L2) and one coalesced write of the destination. The per-texel
// fp32 expressions are evaluated in exactly the reference's order (compile with -ffp-contract=off), so results are
// bit-identical for the non-sRGB formats; sRGB goes through powf and is within 1 ulp per step.
#include "dxtex_kernels.h"
#include "dxtex_store.h"
#include "dxtex_formats.h"
#include "cubic_filter.h"
#include <algorithm>

namespace dxtex
{
namespace
{
struct ImgView
{
    uint8_t* pixels;
    uint64_t rowPitch;
    uint32_t width, height;
    int format;
};

struct ResizeArgs
{
    ImgView src, dst;
    int srgbIn, srgbOut;    // LoadScanlineLinear / StoreScanlineLinear convert sRGB <-> linear around the filter
    int wrapU, wrapV, mirrorU, mirrorV;
    int mipAlias;           // Generate2DMipsBoxFilter: rows / columns alias when the source is 1 high / 1 wide
    ImgView stale;          // ... and what its never-refreshed fourth row pointer still sees (see resize_box_kernel)
    // triangle filter tables (device memory): per destination row / column a run of (source index, weight)
    const uint32_t* triOfsX; const uint2* triX;
    const uint32_t* triOfsY; const uint2* triY;
};

__device__ __forceinline__ Texel load_linear(const ImgView& v, uint32_t x, uint32_t y, int srgb)
{
    Texel t = load_texel(v.pixels + uint64_t(y) * v.rowPitch, x, v.format);
    if (srgb) { t.r = srgb_to_linear1(t.r); t.g = srgb_to_linear1(t.g); t.b = srgb_to_linear1(t.b); }
    return t;
}

__device__ __forceinline__ void store_linear(const ImgView& v, uint32_t x, uint32_t y, int srgb, Texel t)
{
    if (srgb) { t.r = linear_to_srgb1(t.r); t.g = linear_to_srgb1(t.g); t.b = linear_to_srgb1(t.b); }
    store_texel(v.pixels + uint64_t(y) * v.rowPitch, x, v.format, t);
}

#define DXTEX_PER_CHANNEL(EXPR_R, EXPR_G, EXPR_B, EXPR_A) Texel{ (EXPR_R), (EXPR_G), (EXPR_B), (EXPR_A) }

// ---- Convert -------------------------------------------------------------------------------------------------------------
// Rows are the grid's y dimension, which HIP limits to 65535: taller images wrap (grid_rows() caps the launch, the kernels stride).
__host__ __device__ inline uint32_t grid_rows(uint32_t height) { return height < 65535u ? height : 65535u; }

__global__ void __launch_bounds__(256) convert_kernel(ImgView src, ImgView dst, ConvertPlan plan, float threshold)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= src.width) return;
    for (uint32_t y = blockIdx.y; y < src.height; y += gridDim.y)
    {
        const Texel t = load_texel(src.pixels + uint64_t(y) * src.rowPitch, x, src.format);
        store_texel(dst.pixels + uint64_t(y) * dst.rowPitch, x, dst.format, apply_plan(t, plan), threshold);
    }
}

// Formats whose element holds several texels (FC_GROUP: R8G8_B8G8, G8R8_G8B8, YUY2, Y210, Y216, R1): every operation that writes
// one produces R32G32B32A32_FLOAT rows first - exactly the XMVECTOR row the reference hands to StoreScanline - and this kernel
// stores them, an element per lane.
__global__ void __launch_bounds__(256) pack_group_kernel(ImgView rows, ImgView dst)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t per = group_texels(dst.format);
    if (uint64_t(g) * per >= dst.width) return;
    const uint32_t n = min(per, dst.width - g * per);
    for (uint32_t y = blockIdx.y; y < dst.height; y += gridDim.y)
    {
        const float4* in = reinterpret_cast<const float4*>(rows.pixels + uint64_t(y) * rows.rowPitch) + uint64_t(g) * per;
        Texel t[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
        {
            t[k].r = t[k].g = t[k].b = t[k].a = 0.0f;
            if (k < n) { const float4 v = in[k]; t[k].r = v.x; t[k].g = v.y; t[k].b = v.z; t[k].a = v.w; }
        }
        store_group(dst.pixels + uint64_t(y) * dst.rowPitch, g, dst.format, t, n);
    }
}

// Four consecutive texels of a row per lane: the source quad arrives in 1-4 sixteen-byte loads (a wavefront reads 1-4 KiB of
// consecutive bytes per instruction), is decoded / converted / encoded texel by texel with the same load_texel / apply_plan /
// store_texel as above - on a register image of the quad, every index a compile-time constant - and leaves in 1-4 sixteen-byte
// stores. Used when a texel is a whole number of dwords on both sides (>= 32 bpp) and rows are 16-byte aligned.
// every format whose texel is a whole number of dwords (the quad kernel's domain; launch_convert admits no other), with the bytes of a quad
#define DXTEX_QUAD_FORMATS(X) \
    X(FMT_R32G32B32A32_FLOAT, 64) X(FMT_R32G32B32A32_UINT, 64) X(FMT_R32G32B32A32_SINT, 64) X(FMT_R32G32B32_FLOAT, 48) X(FMT_R32G32B32_UINT, 48) \
    X(FMT_R32G32B32_SINT, 48) X(FMT_R16G16B16A16_FLOAT, 32) X(FMT_R16G16B16A16_UNORM, 32) X(FMT_R16G16B16A16_UINT, 32) X(FMT_R16G16B16A16_SNORM, 32) \
    X(FMT_R16G16B16A16_SINT, 32) X(FMT_R32G32_FLOAT, 32) X(FMT_R32G32_UINT, 32) X(FMT_R32G32_SINT, 32) X(FMT_Y416, 32) \
    X(FMT_R10G10B10A2_UNORM, 16) X(FMT_R10G10B10A2_UINT, 16) X(FMT_R11G11B10_FLOAT, 16) X(FMT_R8G8B8A8_UNORM, 16) X(FMT_R8G8B8A8_UNORM_SRGB, 16) \
    X(FMT_R8G8B8A8_UINT, 16) X(FMT_R8G8B8A8_SNORM, 16) X(FMT_R8G8B8A8_SINT, 16) X(FMT_R16G16_FLOAT, 16) X(FMT_R16G16_UNORM, 16) \
    X(FMT_R16G16_UINT, 16) X(FMT_R16G16_SNORM, 16) X(FMT_R16G16_SINT, 16) X(FMT_R32_FLOAT, 16) X(FMT_R32_UINT, 16) \
    X(FMT_R32_SINT, 16) X(FMT_R9G9B9E5_SHAREDEXP, 16) X(FMT_B8G8R8A8_UNORM, 16) X(FMT_B8G8R8X8_UNORM, 16) X(FMT_R10G10B10_XR_BIAS_A2_UNORM, 16) \
    X(FMT_B8G8R8A8_UNORM_SRGB, 16) X(FMT_B8G8R8X8_UNORM_SRGB, 16) X(FMT_AYUV, 16) X(FMT_Y410, 16) X(FMT_D32_FLOAT_S8X24_UINT, 32) \
    X(FMT_D32_FLOAT, 16) X(FMT_D24_UNORM_S8_UINT, 16)

template<int W>
__device__ __forceinline__ void load_quad(uint32_t (&q)[W], const uint8_t* p, uint32_t bytes)
{
    const uint4* v = reinterpret_cast<const uint4*>(p);
    { const uint4 a = v[0]; q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; }
    if constexpr (W >= 8) if (bytes >= 32u) { const uint4 a = v[1]; q[4] = a.x; q[5] = a.y; q[6] = a.z; q[7] = a.w; }
    if constexpr (W >= 12) if (bytes >= 48u) { const uint4 a = v[2]; q[8] = a.x; q[9] = a.y; q[10] = a.z; q[11] = a.w; }
    if constexpr (W >= 16) if (bytes >= 64u) { const uint4 a = v[3]; q[12] = a.x; q[13] = a.y; q[14] = a.z; q[15] = a.w; }
}

template<int W>
__device__ __forceinline__ void store_quad(uint8_t* p, const uint32_t (&q)[W], uint32_t bytes)
{
    uint4* v = reinterpret_cast<uint4*>(p);
    v[0] = make_uint4(q[0], q[1], q[2], q[3]);
    if constexpr (W >= 8) if (bytes >= 32u) v[1] = make_uint4(q[4], q[5], q[6], q[7]);
    if constexpr (W >= 12) if (bytes >= 48u) v[2] = make_uint4(q[8], q[9], q[10], q[11]);
    if constexpr (W >= 16) if (bytes >= 64u) v[3] = make_uint4(q[12], q[13], q[14], q[15]);
}

// SQ / DQ = bytes of a source / destination quad (16, 32 or 64; 0 = taken from the arguments: the 48-byte R32G32B32 quads).
// ROWS quads (of consecutive rows, same columns) are loaded before the first is converted, so that a lane keeps 64 bytes of
// reads in flight: with one 16-byte load per lane the kernel was bound by latency x occupancy (Little's law), not by HBM.
template<int SQ, int DQ, int ROWS>
__global__ void __launch_bounds__(256) convert_quad_kernel(ImgView src, ImgView dst, ConvertPlan plan, float threshold, uint32_t srcQuadBytes, uint32_t dstQuadBytes)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q * 4u >= src.width) return;
    const uint32_t sq = SQ ? uint32_t(SQ) : srcQuadBytes, dq = DQ ? uint32_t(DQ) : dstQuadBytes;
    constexpr int SW = SQ ? SQ / 4 : 16, DW = DQ ? DQ / 4 : 16;
    for (uint32_t y0 = blockIdx.y * uint32_t(ROWS); y0 < src.height; y0 += gridDim.y * uint32_t(ROWS))
    {
        uint32_t in[ROWS][SW];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
        {
#pragma unroll
            for (int k = 0; k < SW; ++k) in[r][k] = 0u;
            const uint32_t y = min(y0 + uint32_t(r), src.height - 1u);       // a short last group re-reads the last row (and does not store it)
            load_quad<SW>(in[r], src.pixels + uint64_t(y) * src.rowPitch + uint64_t(q) * sq, sq);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
        {
            uint32_t out[DW];
#pragma unroll
            for (int k = 0; k < DW; ++k) out[k] = 0u;
            // load_texel / store_texel are called with the format as a compile-time constant (one case per whole-dword format):
            // with a run-time format their switch also holds the byte- and word-addressed formats, whose accesses would force the
            // register image of the quad into memory
            Texel tx[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) tx[k].r = tx[k].g = tx[k].b = tx[k].a = 0.0f;
            switch (src.format)
            {
#define DXTEX_QCASE(F, QB) case F: if constexpr (SQ == QB || (SQ == 0 && QB > 0)) { _Pragma("unroll") for (uint32_t k = 0; k < 4u; ++k) tx[k] = load_texel(reinterpret_cast<const uint8_t*>(in[r]), k, F); } break;
                DXTEX_QUAD_FORMATS(DXTEX_QCASE)
#undef DXTEX_QCASE
            default: break;
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) tx[k] = apply_plan(tx[k], plan);
            switch (dst.format)
            {
#define DXTEX_QCASE(F, QB) case F: if constexpr (DQ == QB || (DQ == 0 && QB > 0)) { _Pragma("unroll") for (uint32_t k = 0; k < 4u; ++k) store_texel(reinterpret_cast<uint8_t*>(out), k, F, tx[k], threshold); } break;
                DXTEX_QUAD_FORMATS(DXTEX_QCASE)
#undef DXTEX_QCASE
            default: break;
            }
            if (y0 + uint32_t(r) < src.height)
                store_quad<DW>(dst.pixels + uint64_t(y0 + uint32_t(r)) * dst.rowPitch + uint64_t(q) * dq, out, dq);
        }
    }
}

// ---- PremultiplyAlpha / DemultiplyAlpha (DirectXTexPMAlpha.cpp:30-205): rgb * a, or rgb / a where a > 0, in linear space ------------
__device__ __forceinline__ void pmalpha_kernel_row(ImgView src, ImgView dst, int srgbIn, int srgbOut, int reverse, const uint32_t x, const uint32_t y)
{
    if (x >= src.width) return;
    Texel t = load_linear(src, x, y, srgbIn);
    if (!reverse) { t.r = t.r * t.a; t.g = t.g * t.a; t.b = t.b * t.a; }
    else if (t.a > 0.0f) { t.r = t.r / t.a; t.g = t.g / t.a; t.b = t.b / t.a; }
    else { t.r = t.g = t.b = t.a; }     // as written (:134-141): with alpha <= 0 the select picks the undivided alpha splat, not the colour
    store_linear(dst, x, y, srgbOut, t);
}
__global__ void __launch_bounds__(256) pmalpha_kernel(ImgView src, ImgView dst, int srgbIn, int srgbOut, int reverse)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < src.height; y += gridDim.y) pmalpha_kernel_row(src, dst, srgbIn, srgbOut, reverse, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// ---- ScaleMipMapsAlphaForCoverage (DirectXTexMipmaps.cpp:143-352) ---------------------------------------------------------------------
// ScaleAlpha: alpha * scale, colour untouched.
__device__ __forceinline__ void scale_alpha_kernel_row(ImgView src, ImgView dst, float scale, const uint32_t x, const uint32_t y)
{
    if (x >= src.width) return;
    Texel t = load_texel(src.pixels + uint64_t(y) * src.rowPitch, x, src.format);
    t.a = t.a * scale;
    store_texel(dst.pixels + uint64_t(y) * dst.rowPitch, x, dst.format, t);
}
__global__ void __launch_bounds__(256) scale_alpha_kernel(ImgView src, ImgView dst, float scale)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < src.height; y += gridDim.y) scale_alpha_kernel_row(src, dst, scale, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// CalculateAlphaCoverage: every 2x2 quad of scaled, saturated alphas is sampled at 8x8 sub-positions with bilinear weights and
// the samples above alphaReference are counted. Reproduced as written, including that the running vector `v` is overwritten
// with the (splatted) sum after every sub-sample (:283), so samples 2..64 of a quad see the previous sum, not the four alphas.
__device__ __forceinline__ void alpha_coverage_kernel_row(ImgView src, float scale, float alphaReference, unsigned long long* count, const uint32_t x, const uint32_t y)
{
    uint32_t n = 0;
    if (x + 1 < src.width)
    {
        const uint8_t* row0 = src.pixels + uint64_t(y) * src.rowPitch;
        const uint8_t* row1 = row0 + src.rowPitch;
        float v[4];
        v[0] = load_texel(row0, x, src.format).a * scale; v[1] = load_texel(row1, x, src.format).a * scale;
        v[2] = load_texel(row0, x + 1, src.format).a * scale; v[3] = load_texel(row1, x + 1, src.format).a * scale;
#pragma unroll
        for (int i = 0; i < 4; ++i) { float m = (v[i] > 0.0f) ? v[i] : 0.0f; v[i] = (m < 1.0f) ? m : 1.0f; }      // XMVectorSaturate
#pragma unroll 1
        for (int sy = 0; sy < 8; ++sy)
        {
            const float fy = (float(sy) + 0.5f) / 8.0f, ify = 1.0f - fy;
#pragma unroll
            for (int sx = 0; sx < 8; ++sx)
            {
                const float fx = (float(sx) + 0.5f) / 8.0f, ifx = 1.0f - fx;
                const float s = (v[0] * (ifx * ify) + v[1] * (ifx * fy)) + (v[2] * (fx * ify) + v[3] * (fx * fy));   // XMVectorSum: (x + y) + (z + w)
                v[0] = v[1] = v[2] = v[3] = s;
                n += (s > alphaReference) ? 1u : 0u;
            }
        }
    }
    // wave total, one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(count, static_cast<unsigned long long>(n));
}
__global__ void __launch_bounds__(256) alpha_coverage_kernel(ImgView src, float scale, float alphaReference, unsigned long long* count)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < src.height - 1u; y += gridDim.y) alpha_coverage_kernel_row(src, scale, alphaReference, count, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// ScratchImage::IsAlphaAllOpaque (DirectXTexImage.cpp:800-852): counts the texels whose alpha is below the threshold
// (XMVector4Less on the splatted alpha: a NaN alpha is not "less" and counts as opaque, as there).
__global__ void __launch_bounds__(256) alpha_below_kernel(ImgView src, float threshold, unsigned long long* count)
{
    uint32_t n = 0;
    for (uint32_t y = blockIdx.y; y < src.height; y += gridDim.y)
        for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < src.width; x += gridDim.x * 256u)
            n += (load_texel(src.pixels + uint64_t(y) * src.rowPitch, x, src.format).a < threshold) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(count, static_cast<unsigned long long>(n));
}

// ---- point (:255-309 / :907-987): 16.16 fixed-point stepping -----------------------------------------------------------------
__device__ __forceinline__ void resize_point_kernel_row(ResizeArgs a, const uint32_t x, const uint32_t y)
{
    if (x >= a.dst.width) return;
    const uint64_t xinc = (uint64_t(a.src.width) << 16) / a.dst.width;
    const uint64_t yinc = (uint64_t(a.src.height) << 16) / a.dst.height;
    const uint32_t sx = uint32_t((uint64_t(x) * xinc) >> 16), sy = uint32_t((uint64_t(y) * yinc) >> 16);
    const Texel t = load_texel(a.src.pixels + uint64_t(sy) * a.src.rowPitch, sx, a.src.format);
    store_texel(a.dst.pixels + uint64_t(y) * a.dst.rowPitch, x, a.dst.format, t);
}
__global__ void __launch_bounds__(256) resize_point_kernel(ResizeArgs a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y) resize_point_kernel_row(a, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// ---- box (filters.h:31-37): (((p0 + p1) + p2) + p3) * 0.25 with p0 = (2x, 2y), p1 = (2x, 2y+1), p2 = (2x+1, 2y), p3 = (2x+1, 2y+1)
__device__ __forceinline__ void resize_box_kernel_row(ResizeArgs a, const uint32_t x, const uint32_t y)
{
    if (x >= a.dst.width) return;
    // Generate2DMipsBoxFilter: a 1-high source reads the same row twice, a 1-wide source the same column (:1024-1033)
    const bool oneRow = a.mipAlias && a.src.height <= 1, oneCol = a.mipAlias && a.src.width <= 1;
    const uint32_t x0 = oneCol ? 0u : 2u * x, x1 = oneCol ? 0u : 2u * x + 1u;
    const uint32_t y0 = oneRow ? 0u : 2u * y, y1 = oneRow ? 0u : 2u * y + 1u;
    const Texel p0 = load_linear(a.src, x0, y0, a.srgbIn), p1 = load_linear(a.src, x0, y1, a.srgbIn);
    const Texel p2 = load_linear(a.src, x1, y0, a.srgbIn);
    // Reference quirk, reproduced: urow3 = urow1 + 1 is computed once, before the level loop (:1017), and is not
    // re-pointed when a 1-high source makes urow1 alias urow0 (:1024-1027). For W x 1 sources with W > 1 the fourth
    // tap therefore still reads the old second-row buffer: row 1 of the last level that was 2 texels high.
    const bool staleTap = oneRow && !oneCol && a.stale.pixels != nullptr;
    const Texel p3 = staleTap ? load_linear(a.stale, x1, 1u, a.srgbIn) : load_linear(a.src, x1, y1, a.srgbIn);
    Texel r;
    r.r = (((p0.r + p1.r) + p2.r) + p3.r) * 0.25f;
    r.g = (((p0.g + p1.g) + p2.g) + p3.g) * 0.25f;
    r.b = (((p0.b + p1.b) + p2.b) + p3.b) * 0.25f;
    r.a = (((p0.a + p1.a) + p2.a) + p3.a) * 0.25f;
    store_linear(a.dst, x, y, a.srgbOut, r);
}
__global__ void __launch_bounds__(256) resize_box_kernel(ResizeArgs a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y) resize_box_kernel_row(a, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// Box, exactly 2:1 on RGBA8 (the upper levels of a power-of-two mip chain, where the bytes are): one lane produces two adjacent
// destination texels from one 16-byte load per source row - consecutive lanes read consecutive 16 bytes - and writes 8 bytes.
// Same expression as resize_box_kernel: (((p00 + p01) + p10) + p11) * 0.25 with p0x the left column's two rows.
__global__ void __launch_bounds__(256) resize_box_half_rgba8_kernel(ResizeArgs a)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;              // pair of destination texels
    if (q * 2u >= a.dst.width) return;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y)
    {
        const uint4 t = reinterpret_cast<const uint4*>(a.src.pixels + uint64_t(2u * y) * a.src.rowPitch)[q];
        const uint4 b = reinterpret_cast<const uint4*>(a.src.pixels + uint64_t(2u * y + 1u) * a.src.rowPitch)[q];
        const uint32_t top[4] = { t.x, t.y, t.z, t.w }, bot[4] = { b.x, b.y, b.z, b.w };
        uint32_t out[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
        {
            uint32_t packed = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
            {
                const float p0 = float((top[2 * k] >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), p1 = float((bot[2 * k] >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
                const float p2 = float((top[2 * k + 1] >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), p3 = float((bot[2 * k + 1] >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
                packed |= store_ubn_biased((((p0 + p1) + p2) + p3) * 0.25f) << (8 * c);
            }
            out[k] = packed;
        }
        reinterpret_cast<uint2*>(a.dst.pixels + uint64_t(y) * a.dst.rowPitch)[q] = make_uint2(out[0], out[1]);
    }
}

// ---- linear (filters.h:57-104) ---------------------------------------------------------------------------------------------
struct Lin { uint32_t u0, u1; float w0, w1; };
__device__ __forceinline__ Lin linear_entry(uint32_t source, uint32_t dest, bool wrap, uint32_t u)
{
    const float scale = float(source) / float(dest);
    const float srcB = (float(u) + 0.5f) * scale + 0.5f;
    long long isrcB = (long long)srcB;
    long long isrcA = isrcB - 1;
    const float weight = 1.0f + float(isrcB) - srcB;
    if (isrcA < 0) isrcA = wrap ? (long long)source - 1 : 0;
    if ((unsigned long long)isrcB >= source) isrcB = wrap ? 0 : (long long)source - 1;
    Lin e; e.u0 = uint32_t(isrcA); e.w0 = weight; e.u1 = uint32_t(isrcB); e.w1 = 1.0f - weight;
    return e;
}

__device__ __forceinline__ void resize_linear_kernel_row(ResizeArgs a, const uint32_t x, const uint32_t y)
{
    if (x >= a.dst.width) return;
    const Lin tx = linear_entry(a.src.width, a.dst.width, a.wrapU != 0, x);
    const Lin ty = linear_entry(a.src.height, a.dst.height, a.wrapV != 0, y);
    const Texel p00 = load_linear(a.src, tx.u0, ty.u0, a.srgbIn), p01 = load_linear(a.src, tx.u1, ty.u0, a.srgbIn);
    const Texel p10 = load_linear(a.src, tx.u0, ty.u1, a.srgbIn), p11 = load_linear(a.src, tx.u1, ty.u1, a.srgbIn);
    // BILINEAR_INTERPOLATE: ((r0[u0]*wx0 + r0[u1]*wx1) * wy0) + ((r1[u0]*wx0 + r1[u1]*wx1) * wy1)
#define DXTEX_BILERP(C) (((p00.C * tx.w0 + p01.C * tx.w1) * ty.w0) + ((p10.C * tx.w0 + p11.C * tx.w1) * ty.w1))
    Texel r;
    r.r = DXTEX_BILERP(r); r.g = DXTEX_BILERP(g); r.b = DXTEX_BILERP(b); r.a = DXTEX_BILERP(a);
#undef DXTEX_BILERP
    store_linear(a.dst, x, y, a.srgbOut, r);
}
__global__ void __launch_bounds__(256) resize_linear_kernel(ResizeArgs a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y) resize_linear_kernel_row(a, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// ---- cubic (filters.h:106-207) -----------------------------------------------------------------------------------------------
__device__ __forceinline__ long long bounduvw(long long u, long long maxu, bool wrap, bool mirror)
{
    if (wrap)
    {
        if (u < 0) u = maxu + u + 1;
        else if (u > maxu) u = u - maxu - 1;
    }
    else if (mirror)
    {
        if (u < 0) u = (-u) - 1;
        else if (u > maxu) u = maxu - (u - maxu - 1);
    }
    u = (u < maxu) ? u : maxu;
    u = (u > 0) ? u : 0;
    return u;
}

struct Cub { uint32_t u0, u1, u2, u3; float x; };
__device__ __forceinline__ Cub cubic_entry(uint32_t source, uint32_t dest, bool wrap, bool mirror, uint32_t u)
{
    const float scale = float(source) / float(dest);
    const float srcB = (float(u) + 0.5f) * scale - 0.5f;
    const long long maxu = (long long)source - 1;
    const long long iB = bounduvw((long long)srcB, maxu, wrap, mirror);
    Cub e;
    e.u0 = uint32_t(bounduvw(iB - 1, maxu, wrap, mirror));
    e.u1 = uint32_t(iB);
    e.u2 = uint32_t(bounduvw(iB + 1, maxu, wrap, mirror));
    e.u3 = uint32_t(bounduvw(iB + 2, maxu, wrap, mirror));
    e.x = srcB - float(iB);
    return e;
}

__device__ __forceinline__ void resize_cubic_kernel_row(ResizeArgs a, const uint32_t x, const uint32_t y)
{
    if (x >= a.dst.width) return;
    const Cub tx = cubic_entry(a.src.width, a.dst.width, a.wrapU != 0, a.mirrorU != 0, x);
    const Cub ty = cubic_entry(a.src.height, a.dst.height, a.wrapV != 0, a.mirrorV != 0, y);
    const uint32_t ys[4] = { ty.u0, ty.u1, ty.u2, ty.u3 };
    Texel c[4];
    // RGBA8 away from the left / right border: the four taps of a row are adjacent texels, one 16-byte load instead of four
    // format-dispatched 4-byte loads (XMLoadUByteN4: byte * (1/255), as load_texel does)
    const bool rowLoad = a.src.format == FMT_R8G8B8A8_UNORM && !a.srgbIn && tx.u1 == tx.u0 + 1u && tx.u2 == tx.u0 + 2u && tx.u3 == tx.u0 + 3u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        Texel p0, p1, p2, p3;
        if (rowLoad)
        {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(a.src.pixels + uint64_t(ys[r]) * a.src.rowPitch) + tx.u0;
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3];        // 4-byte aligned: the compiler merges them into one dwordx4 load
#define DXTEX_UNPACK(P, W) P.r = float(W & 0xFF) * (1.0f / 255.0f); P.g = float((W >> 8) & 0xFF) * (1.0f / 255.0f); \
                           P.b = float((W >> 16) & 0xFF) * (1.0f / 255.0f); P.a = float(W >> 24) * (1.0f / 255.0f)
            DXTEX_UNPACK(p0, w0); DXTEX_UNPACK(p1, w1); DXTEX_UNPACK(p2, w2); DXTEX_UNPACK(p3, w3);
#undef DXTEX_UNPACK
        }
        else
        {
            p0 = load_linear(a.src, tx.u0, ys[r], a.srgbIn); p1 = load_linear(a.src, tx.u1, ys[r], a.srgbIn);
            p2 = load_linear(a.src, tx.u2, ys[r], a.srgbIn); p3 = load_linear(a.src, tx.u3, ys[r], a.srgbIn);
        }
        c[r].r = cubic1(tx.x, p0.r, p1.r, p2.r, p3.r); c[r].g = cubic1(tx.x, p0.g, p1.g, p2.g, p3.g);
        c[r].b = cubic1(tx.x, p0.b, p1.b, p2.b, p3.b); c[r].a = cubic1(tx.x, p0.a, p1.a, p2.a, p3.a);
    }
    Texel o;
    o.r = cubic1(ty.x, c[0].r, c[1].r, c[2].r, c[3].r); o.g = cubic1(ty.x, c[0].g, c[1].g, c[2].g, c[3].g);
    o.b = cubic1(ty.x, c[0].b, c[1].b, c[2].b, c[3].b); o.a = cubic1(ty.x, c[0].a, c[1].a, c[2].a, c[3].a);
    store_linear(a.dst, x, y, a.srgbOut, o);
}
__global__ void __launch_bounds__(256) resize_cubic_kernel(ResizeArgs a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y) resize_cubic_kernel_row(a, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// Cubic, exactly 2:1 in both directions with clamp addressing on RGBA8 (every level of a power-of-two mip chain): srcB = 2u + 0.5, so the
// four taps are texels 2u-1 .. 2u+2 (clamped) and dx = 0.5 for every destination texel. The filter is separable in the reference too
// (CUBIC_INTERPOLATE along x for four source rows, then along y, filters.h:192-207), and neighbouring destination rows share two of their
// four source rows: a lane owns one destination column of a strip of rows and walks down it with the x-filtered values of the last four
// source rows in registers - two row passes per destination texel instead of four (plus two at the top of the strip), every source texel
// unpacked once per row pass, no LDS and no barrier (a tile in LDS held 35 KiB per workgroup and left the kernel waiting on it at four
// waves per SIMD). Same expressions, same order, same bits as resize_cubic_kernel.
__global__ void __launch_bounds__(256) resize_cubic_half_rgba8_kernel(ResizeArgs a, uint32_t stripRows)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    if (x >= a.dst.width) return;
    const int64_t srcW = a.src.width, srcH = a.src.height;
    const int64_t u0 = int64_t(2) * x - 1;
    const bool inside = u0 >= 0 && u0 + 3 < srcW;
    const uint32_t i0 = uint32_t(u0 < 0 ? 0 : u0), i1 = uint32_t(u0 + 1 > srcW - 1 ? srcW - 1 : u0 + 1),
                   i2 = uint32_t(u0 + 2 > srcW - 1 ? srcW - 1 : u0 + 2), i3 = uint32_t(u0 + 3 > srcW - 1 ? srcW - 1 : u0 + 3);
    // the x-filtered texel of source row sy (clamped) at this column
    auto xpass = [&](int64_t sy) -> float4
    {
        sy = sy < 0 ? 0 : (sy > srcH - 1 ? srcH - 1 : sy);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a.src.pixels + uint64_t(sy) * a.src.rowPitch);
        uint32_t w0, w1, w2, w3;
        if (inside) { w0 = q[u0]; w1 = q[u0 + 1]; w2 = q[u0 + 2]; w3 = q[u0 + 3]; }
        else { w0 = q[i0]; w1 = q[i1]; w2 = q[i2]; w3 = q[i3]; }
#define DXTEX_CH(W, S) (float(((W) >> (S)) & 0xFFu) * (1.0f / 255.0f))
        float4 c;
        c.x = cubic_half1(DXTEX_CH(w0, 0), DXTEX_CH(w1, 0), DXTEX_CH(w2, 0), DXTEX_CH(w3, 0));
        c.y = cubic_half1(DXTEX_CH(w0, 8), DXTEX_CH(w1, 8), DXTEX_CH(w2, 8), DXTEX_CH(w3, 8));
        c.z = cubic_half1(DXTEX_CH(w0, 16), DXTEX_CH(w1, 16), DXTEX_CH(w2, 16), DXTEX_CH(w3, 16));
        c.w = cubic_half1(DXTEX_CH(w0, 24), DXTEX_CH(w1, 24), DXTEX_CH(w2, 24), DXTEX_CH(w3, 24));
#undef DXTEX_CH
        return c;
    };
    // the grid's y dimension is capped at 65535 strips (HIP's limit): a taller level is covered by striding over the strips
    for (uint64_t s0 = uint64_t(blockIdx.y) * stripRows; s0 < a.dst.height; s0 += uint64_t(gridDim.y) * stripRows)
    {
    const uint32_t y0 = uint32_t(s0), y1 = uint32_t(min(s0 + stripRows, uint64_t(a.dst.height)));
    // destination row y takes source rows 2y - 1 .. 2y + 2
    float4 c0 = xpass(int64_t(2) * y0 - 1), c1 = xpass(int64_t(2) * y0), c2 = xpass(int64_t(2) * y0 + 1);
#pragma unroll 2
    for (uint32_t y = y0; y < y1; ++y)
    {
        const float4 c3 = xpass(int64_t(2) * y + 2);
        const float4 n2 = xpass(int64_t(2) * y + 3);             // row 2(y+1) + 1 of the next trip, loaded before this trip's store
        Texel o;
        o.r = cubic_half1(c0.x, c1.x, c2.x, c3.x); o.g = cubic_half1(c0.y, c1.y, c2.y, c3.y);
        o.b = cubic_half1(c0.z, c1.z, c2.z, c3.z); o.a = cubic_half1(c0.w, c1.w, c2.w, c3.w);
        reinterpret_cast<uint32_t*>(a.dst.pixels + uint64_t(y) * a.dst.rowPitch)[x] = pack_texel32(FMT_R8G8B8A8_UNORM, o);
        c0 = c2; c1 = c3; c2 = n2;
    }
    }
}

// The same filter with TWO adjacent destination texels per lane (destination width even, rows 16-byte aligned): the six source texels a
// pair needs are one 16-byte load and two neighbouring dwords, every source texel is unpacked once per pair instead of once per destination
// texel, and a wavefront reads 1 KiB runs. Same expressions in the same order as resize_cubic_half_rgba8_kernel, hence the same bits.
__global__ void __launch_bounds__(256) resize_cubic_half_rgba8_x2_kernel(ResizeArgs a, uint32_t stripRows)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;            // pair index: destination texels 2k, 2k + 1
    if (2u * k >= a.dst.width) return;
    const int64_t srcW = a.src.width, srcH = a.src.height;
    const uint32_t iL = (k == 0) ? 0u : 4u * k - 1u;               // clamp addressing: texel 4k - 1 / 4k + 4
    const uint32_t iR = uint32_t(int64_t(4) * k + 4 > srcW - 1 ? srcW - 1 : int64_t(4) * k + 4);
    struct Pair { float4 l, r; };
    auto xpass = [&](int64_t sy) -> Pair
    {
        sy = sy < 0 ? 0 : (sy > srcH - 1 ? srcH - 1 : sy);
        const uint8_t* row = a.src.pixels + uint64_t(sy) * a.src.rowPitch;
        const uint4 m = reinterpret_cast<const uint4*>(row)[k];
        const uint32_t wl = reinterpret_cast<const uint32_t*>(row)[iL], wr = reinterpret_cast<const uint32_t*>(row)[iR];
#define DXTEX_CH(W, S) (float(((W) >> (S)) & 0xFFu) * (1.0f / 255.0f))
        Pair o;
#define DXTEX_ONE(S, FL, FR) { const float t0 = DXTEX_CH(wl, S), t1 = DXTEX_CH(m.x, S), t2 = DXTEX_CH(m.y, S), t3 = DXTEX_CH(m.z, S), t4 = DXTEX_CH(m.w, S), t5 = DXTEX_CH(wr, S); \
                               FL = cubic_half1(t0, t1, t2, t3); FR = cubic_half1(t2, t3, t4, t5); }
        DXTEX_ONE(0, o.l.x, o.r.x) DXTEX_ONE(8, o.l.y, o.r.y) DXTEX_ONE(16, o.l.z, o.r.z) DXTEX_ONE(24, o.l.w, o.r.w)
#undef DXTEX_ONE
#undef DXTEX_CH
        return o;
    };
    for (uint64_t s0 = uint64_t(blockIdx.y) * stripRows; s0 < a.dst.height; s0 += uint64_t(gridDim.y) * stripRows)
    {
        const uint32_t y0 = uint32_t(s0), y1 = uint32_t(min(s0 + stripRows, uint64_t(a.dst.height)));
        Pair c0 = xpass(int64_t(2) * y0 - 1), c1 = xpass(int64_t(2) * y0), c2 = xpass(int64_t(2) * y0 + 1);
#pragma unroll 2
        for (uint32_t y = y0; y < y1; ++y)
        {
            const Pair c3 = xpass(int64_t(2) * y + 2);
            const Pair n2 = xpass(int64_t(2) * y + 3);
            Texel ol, orr;
            ol.r = cubic_half1(c0.l.x, c1.l.x, c2.l.x, c3.l.x); ol.g = cubic_half1(c0.l.y, c1.l.y, c2.l.y, c3.l.y);
            ol.b = cubic_half1(c0.l.z, c1.l.z, c2.l.z, c3.l.z); ol.a = cubic_half1(c0.l.w, c1.l.w, c2.l.w, c3.l.w);
            orr.r = cubic_half1(c0.r.x, c1.r.x, c2.r.x, c3.r.x); orr.g = cubic_half1(c0.r.y, c1.r.y, c2.r.y, c3.r.y);
            orr.b = cubic_half1(c0.r.z, c1.r.z, c2.r.z, c3.r.z); orr.a = cubic_half1(c0.r.w, c1.r.w, c2.r.w, c3.r.w);
            reinterpret_cast<uint2*>(a.dst.pixels + uint64_t(y) * a.dst.rowPitch)[k] =
                make_uint2(pack_texel32(FMT_R8G8B8A8_UNORM, ol), pack_texel32(FMT_R8G8B8A8_UNORM, orr));
            c0 = c2; c1 = c3; c2 = n2;
        }
    }
}

// ---- triangle (filters.h:209-419; accumulation order of DirectXTexMipmaps.cpp:1517-1542 / DirectXTexResize.cpp:730-760) ----------
// The reference scatters every source texel into accumulation rows; the sum a destination texel receives is ordered
// by (source row, source column, row-list entry, column-list entry). The host inverts the filter lists so that a lane
// can gather its texel's contributions in that same order: acc = acc + src * (wy * wx), unfused.
__device__ __forceinline__ void resize_triangle_kernel_row(ResizeArgs a, const uint32_t x, const uint32_t y)
{
    if (x >= a.dst.width) return;
    const uint32_t yb = a.triOfsY[y], ye = a.triOfsY[y + 1];
    const uint32_t xb = a.triOfsX[x], xe = a.triOfsX[x + 1];
    Texel acc; acc.r = acc.g = acc.b = acc.a = 0.0f;
    uint32_t i = yb;
    while (i < ye)
    {
        const uint32_t sy = a.triY[i].x;
        uint32_t iEnd = i + 1;
        while (iEnd < ye && a.triY[iEnd].x == sy) ++iEnd;
        uint32_t k = xb;
        while (k < xe)
        {
            const uint32_t sx = a.triX[k].x;
            uint32_t kEnd = k + 1;
            while (kEnd < xe && a.triX[kEnd].x == sx) ++kEnd;
            const Texel p = load_linear(a.src, sx, sy, a.srgbIn);
            for (uint32_t j = i; j < iEnd; ++j)
            {
                const float wy = __uint_as_float(a.triY[j].y);
                for (uint32_t m = k; m < kEnd; ++m)
                {
                    const float w = wy * __uint_as_float(a.triX[m].y);
                    acc.r = p.r * w + acc.r; acc.g = p.g * w + acc.g; acc.b = p.b * w + acc.b; acc.a = p.a * w + acc.a;
                }
            }
            k = kEnd;
        }
        i = iEnd;
    }
    if (a.dst.format == FMT_R10G10B10A2_UNORM || a.dst.format == FMT_R10G10B10A2_UINT) acc.a = acc.a + 0.1f;       // the reference biases 2-bit alpha against accumulation error (DirectXTexMipmaps.cpp:1560-1579, DirectXTexResize.cpp:768-787)
    store_linear(a.dst, x, y, a.srgbOut, acc);
}
__global__ void __launch_bounds__(256) resize_triangle_kernel(ResizeArgs a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t y = blockIdx.y; y < a.dst.height; y += gridDim.y) resize_triangle_kernel_row(a, x, y);        // grid_rows(): HIP caps grid.y at 65535
}

// ---- volume mips: Generate3DMips{Point,Box,Linear,Cubic,Triangle}Filter (DirectXTexMipmaps.cpp:1666-2826) ----------------------------
// One lane = one destination texel (x, y = blockIdx.y, z = blockIdx.z) of a level whose SOURCE has depth > 1; levels whose source
// is one slice deep take the reference's 2-D branch, i.e. the kernels above.
struct Vol
{
    uint8_t* pixels;                 // slice 0
    uint64_t rowPitch, slicePitch;
    uint32_t width, height, depth;
    int format;
};

struct Resize3Args
{
    Vol src, dst;
    int srgbIn, srgbOut;
    int wrapU, wrapV, wrapW, mirrorU, mirrorV, mirrorW;
    ImgView staleU, staleV;          // box: what the never re-pointed urow3 / vrow3 still see on W x 1 x D levels (see resize3d_box_kernel)
    const uint32_t* triOfsX; const uint2* triX;
    const uint32_t* triOfsY; const uint2* triY;
    const uint32_t* triOfsZ; const uint2* triZ;
};

// ---- the tail of a mip chain in ONE workgroup --------------------------------------------------------------------------------
// The last levels of a chain are a few thousand texels each: a launch per level costs ~5 us of dispatch latency for ~1 us of
// work (ten such launches were 40 % of the 8192^2 box chain). From the first level whose source is at most kTailSide texels on a
// side, one workgroup of 1024 lanes walks the remaining levels, a workgroup barrier (with its release / acquire fences on global
// memory) between levels: level n + 1 reads what the same workgroup stored for level n, exactly as the per-level launches do.
constexpr uint32_t kTailSide = 64;
constexpr int kTailMaxLevels = 8;
struct TailArgs
{
    ResizeArgs a;                       // flags; src = the first source level; stale as launch_resize sets it up
    ImgView level[kTailMaxLevels];      // the destination levels, in order
    int nlevels;
    uint32_t mode;                      // TEX_FILTER_POINT / LINEAR / CUBIC / BOX
    ImgView twoHigh;                    // box: the last level of the whole chain that was 2 texels high BEFORE the tail (or null)
};

__global__ void __launch_bounds__(1024) resize_tail_kernel(TailArgs t)
{
    ResizeArgs a = t.a;
    ImgView twoHigh = t.twoHigh;
    for (int l = 0; l < t.nlevels; ++l)
    {
        a.dst = t.level[l];
        if (a.mipAlias && a.src.height >= 2u) twoHigh = a.src;              // submit_resizes' bookkeeping of the stale tap
        a.stale = (a.mipAlias && t.mode == 0x400000u && a.src.height == 1u && a.src.width > 1u && twoHigh.pixels) ? twoHigh : ImgView{ nullptr, 0, 0, 2, a.src.format };
        const uint32_t n = a.dst.width * a.dst.height;
        for (uint32_t i = threadIdx.x; i < n; i += 1024u)
        {
            const uint32_t y = i / a.dst.width, x = i - y * a.dst.width;
            switch (t.mode)
            {
            case 0x100000u: resize_point_kernel_row(a, x, y); break;
            case 0x200000u: resize_linear_kernel_row(a, x, y); break;
            case 0x300000u: resize_cubic_kernel_row(a, x, y); break;
            default: resize_box_kernel_row(a, x, y); break;
            }
        }
        __syncthreads();                                                     // level l is complete and visible to the whole workgroup
        a.src = a.dst;
    }
}

// The tail of a power-of-two RGBA8 cubic chain in LDS: from a source of at most 64 x 64 texels down to 1 x 1, every level an exact halving in
// both directions (clamp addressing, no sRGB). The generic tail above reads each level back from global memory - sixteen dependent taps per
// texel with nothing to hide their latency, 84 us for the six levels - and a launch per level costs ~6.6 us each. Here the source level is
// staged once, every level is produced from the previous one's packed texels in LDS (and written out), a barrier per level: the arithmetic of
// resize_cubic_half_rgba8_kernel (x-pass over four clamped taps of each of four clamped rows, then y), the same bits.
struct CubicTailArgs
{
    const uint8_t* src; uint64_t srcPitch; uint32_t srcW, srcH; int nlevels;
    uint8_t* dst[kTailMaxLevels]; uint64_t dstPitch[kTailMaxLevels];
};
template<bool CUBIC>
__global__ void __launch_bounds__(1024) resize_half_tail_rgba8_kernel(CubicTailArgs t)
{
    __shared__ uint32_t bufA[kTailSide * kTailSide], bufB[(kTailSide / 2) * (kTailSide / 2)];
    uint32_t w = t.srcW, h = t.srcH;
    for (uint32_t i = threadIdx.x; i < w * h; i += 1024u)
    {
        const uint32_t y = i / w, x = i - y * w;
        bufA[i] = reinterpret_cast<const uint32_t*>(t.src + uint64_t(y) * t.srcPitch)[x];
    }
    __syncthreads();
    uint32_t* s = bufA;
    uint32_t* d = bufB;
    for (int l = 0; l < t.nlevels; ++l)
    {
        const uint32_t dw = w >> 1, dh = h >> 1;
        for (uint32_t i = threadIdx.x; i < dw * dh; i += 1024u)
        {
            const uint32_t y = i / dw, x = i - y * dw;
            if constexpr (!CUBIC)
            {
                // the box filter of resize_box_half_rgba8_kernel: (((p0 + p1) + p2) + p3) * 0.25 over (2x, 2y), (2x, 2y + 1), (2x + 1, 2y), (2x + 1, 2y + 1)
                const uint32_t t0 = s[(2u * y) * w + 2u * x], t1 = s[(2u * y) * w + 2u * x + 1u], b0 = s[(2u * y + 1u) * w + 2u * x], b1 = s[(2u * y + 1u) * w + 2u * x + 1u];
                uint32_t packed = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    const float p0 = float((t0 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), p1 = float((b0 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
                    const float p2 = float((t1 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), p3 = float((b1 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
                    packed |= store_ubn_biased((((p0 + p1) + p2) + p3) * 0.25f) << (8 * c);
                }
                d[i] = packed;
                reinterpret_cast<uint32_t*>(t.dst[l] + uint64_t(y) * t.dstPitch[l])[x] = packed;
                continue;
            }
            const int32_t u0 = int32_t(2u * x) - 1, v0 = int32_t(2u * y) - 1;
            uint32_t xi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int32_t u = u0 + k; xi[k] = uint32_t(u < 0 ? 0 : (u > int32_t(w) - 1 ? int32_t(w) - 1 : u)); }
            float4 c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int32_t v = v0 + r;
                const uint32_t* row = s + uint32_t(v < 0 ? 0 : (v > int32_t(h) - 1 ? int32_t(h) - 1 : v)) * w;
                const uint32_t w0 = row[xi[0]], w1 = row[xi[1]], w2 = row[xi[2]], w3 = row[xi[3]];
#define DXTEX_CH(W, S) (float(((W) >> (S)) & 0xFFu) * (1.0f / 255.0f))
                c[r].x = cubic_half1(DXTEX_CH(w0, 0), DXTEX_CH(w1, 0), DXTEX_CH(w2, 0), DXTEX_CH(w3, 0));
                c[r].y = cubic_half1(DXTEX_CH(w0, 8), DXTEX_CH(w1, 8), DXTEX_CH(w2, 8), DXTEX_CH(w3, 8));
                c[r].z = cubic_half1(DXTEX_CH(w0, 16), DXTEX_CH(w1, 16), DXTEX_CH(w2, 16), DXTEX_CH(w3, 16));
                c[r].w = cubic_half1(DXTEX_CH(w0, 24), DXTEX_CH(w1, 24), DXTEX_CH(w2, 24), DXTEX_CH(w3, 24));
#undef DXTEX_CH
            }
            Texel o;
            o.r = cubic_half1(c[0].x, c[1].x, c[2].x, c[3].x); o.g = cubic_half1(c[0].y, c[1].y, c[2].y, c[3].y);
            o.b = cubic_half1(c[0].z, c[1].z, c[2].z, c[3].z); o.a = cubic_half1(c[0].w, c[1].w, c[2].w, c[3].w);
            const uint32_t packed = pack_texel32(FMT_R8G8B8A8_UNORM, o);
            d[i] = packed;
            reinterpret_cast<uint32_t*>(t.dst[l] + uint64_t(y) * t.dstPitch[l])[x] = packed;
        }
        __syncthreads();
        uint32_t* const tmp = s; s = d; d = tmp;          // the next level (a quarter of this one) fits where this level's source was
        w = dw; h = dh;
    }
}

__device__ __forceinline__ ImgView slice_of(const Vol& v, uint32_t z)
{
    ImgView s; s.pixels = v.pixels + uint64_t(z) * v.slicePitch; s.rowPitch = v.rowPitch; s.width = v.width; s.height = v.height; s.format = v.format;
    return s;
}

// point (:1666-1813): source slice (z * zinc) >> 16, then the 2-D point filter inside it
__global__ void __launch_bounds__(256) resize3d_point_kernel(Resize3Args a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= a.dst.width) return;
    const uint64_t zinc = (uint64_t(a.src.depth) << 16) / a.dst.depth;
    const uint64_t xinc = (uint64_t(a.src.width) << 16) / a.dst.width, yinc = (uint64_t(a.src.height) << 16) / a.dst.height;
    const ImgView src = slice_of(a.src, uint32_t((uint64_t(z) * zinc) >> 16)), dst = slice_of(a.dst, z);
    const uint32_t sx = uint32_t((uint64_t(x) * xinc) >> 16), sy = uint32_t((uint64_t(y) * yinc) >> 16);
    store_texel(dst.pixels + uint64_t(y) * dst.rowPitch, x, dst.format, load_texel(src.pixels + uint64_t(sy) * src.rowPitch, sx, src.format));
}

// box (:1815-1990): AVERAGE8 over slices (2z, 2z+1) x rows x columns, summed in the macro's order, x 0.125.
__global__ void __launch_bounds__(256) resize3d_box_kernel(Resize3Args a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= a.dst.width) return;
    const bool oneRow = a.src.height <= 1, oneCol = a.src.width <= 1;
    const uint32_t za = min(2u * z, a.src.depth - 1u), zb = min(za + 1u, a.src.depth - 1u);
    const ImgView A = slice_of(a.src, za), B = slice_of(a.src, zb);
    const uint32_t x0 = oneCol ? 0u : 2u * x, x1 = oneCol ? 0u : 2u * x + 1u;
    const uint32_t y0 = oneRow ? 0u : 2u * y, y1 = oneRow ? 0u : 2u * y + 1u;
    // Reference quirk, reproduced: urow3 = urow1 + 1 and vrow3 = vrow1 + 1 are set once, before the level loop (:1849-1852); a
    // 1-high source re-points urow1 / vrow1 (:1857-1861) but not urow3 / vrow3 unless the source is also 1 wide (:1863-1869).
    // On W x 1 x D sources (W > 1) the fourth tap of either slice therefore still reads the old second-row buffers: row 1 of the
    // last two slices of the last level that was 2 texels high.
    const bool staleTap = oneRow && !oneCol && a.staleU.pixels != nullptr;
    const Texel p0 = load_linear(A, x0, y0, a.srgbIn), p1 = load_linear(A, x0, y1, a.srgbIn), p2 = load_linear(A, x1, y0, a.srgbIn);
    const Texel p3 = staleTap ? load_linear(a.staleU, x1, 1u, a.srgbIn) : load_linear(A, x1, y1, a.srgbIn);
    const Texel p4 = load_linear(B, x0, y0, a.srgbIn), p5 = load_linear(B, x0, y1, a.srgbIn), p6 = load_linear(B, x1, y0, a.srgbIn);
    const Texel p7 = staleTap ? load_linear(a.staleV, x1, 1u, a.srgbIn) : load_linear(B, x1, y1, a.srgbIn);
#define DXTEX_AVG8(C) ((((((((p0.C + p1.C) + p2.C) + p3.C) + p4.C) + p5.C) + p6.C) + p7.C) * 0.125f)
    Texel r;
    r.r = DXTEX_AVG8(r); r.g = DXTEX_AVG8(g); r.b = DXTEX_AVG8(b); r.a = DXTEX_AVG8(a);
#undef DXTEX_AVG8
    store_linear(slice_of(a.dst, z), x, y, a.srgbOut, r);
}

// linear (:1992-2188): TRILINEAR_INTERPOLATE (filters.h:106-113)
__global__ void __launch_bounds__(256) resize3d_linear_kernel(Resize3Args a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= a.dst.width) return;
    const Lin tx = linear_entry(a.src.width, a.dst.width, a.wrapU != 0, x);
    const Lin ty = linear_entry(a.src.height, a.dst.height, a.wrapV != 0, y);
    const Lin tz = linear_entry(a.src.depth, a.dst.depth, a.wrapW != 0, z);
    const ImgView A = slice_of(a.src, tz.u0), B = slice_of(a.src, tz.u1);
    const Texel a00 = load_linear(A, tx.u0, ty.u0, a.srgbIn), a01 = load_linear(A, tx.u1, ty.u0, a.srgbIn);
    const Texel a10 = load_linear(A, tx.u0, ty.u1, a.srgbIn), a11 = load_linear(A, tx.u1, ty.u1, a.srgbIn);
    const Texel b00 = load_linear(B, tx.u0, ty.u0, a.srgbIn), b01 = load_linear(B, tx.u1, ty.u0, a.srgbIn);
    const Texel b10 = load_linear(B, tx.u0, ty.u1, a.srgbIn), b11 = load_linear(B, tx.u1, ty.u1, a.srgbIn);
#define DXTEX_TRILERP(C) ((((a00.C * tx.w0 + a01.C * tx.w1) * ty.w0 + (a10.C * tx.w0 + a11.C * tx.w1) * ty.w1) * tz.w0) + \
                          (((b00.C * tx.w0 + b01.C * tx.w1) * ty.w0 + (b10.C * tx.w0 + b11.C * tx.w1) * ty.w1) * tz.w1))
    Texel r;
    r.r = DXTEX_TRILERP(r); r.g = DXTEX_TRILERP(g); r.b = DXTEX_TRILERP(b); r.a = DXTEX_TRILERP(a);
#undef DXTEX_TRILERP
    store_linear(slice_of(a.dst, z), x, y, a.srgbOut, r);
}

// cubic (:2190-2572): per source slice the 2-D cubic (rows through toX, then toY), then CUBIC_INTERPOLATE through toZ
__global__ void __launch_bounds__(256) resize3d_cubic_kernel(Resize3Args a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= a.dst.width) return;
    const Cub tx = cubic_entry(a.src.width, a.dst.width, a.wrapU != 0, a.mirrorU != 0, x);
    const Cub ty = cubic_entry(a.src.height, a.dst.height, a.wrapV != 0, a.mirrorV != 0, y);
    const Cub tz = cubic_entry(a.src.depth, a.dst.depth, a.wrapW != 0, a.mirrorW != 0, z);
    const uint32_t ys[4] = { ty.u0, ty.u1, ty.u2, ty.u3 }, zs[4] = { tz.u0, tz.u1, tz.u2, tz.u3 };
    Texel d[4];
#pragma unroll 1
    for (int j = 0; j < 4; ++j)
    {
        const ImgView S = slice_of(a.src, zs[j]);
        Texel c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const Texel p0 = load_linear(S, tx.u0, ys[r], a.srgbIn), p1 = load_linear(S, tx.u1, ys[r], a.srgbIn);
            const Texel p2 = load_linear(S, tx.u2, ys[r], a.srgbIn), p3 = load_linear(S, tx.u3, ys[r], a.srgbIn);
            c[r].r = cubic1(tx.x, p0.r, p1.r, p2.r, p3.r); c[r].g = cubic1(tx.x, p0.g, p1.g, p2.g, p3.g);
            c[r].b = cubic1(tx.x, p0.b, p1.b, p2.b, p3.b); c[r].a = cubic1(tx.x, p0.a, p1.a, p2.a, p3.a);
        }
        Texel o;
        o.r = cubic1(ty.x, c[0].r, c[1].r, c[2].r, c[3].r); o.g = cubic1(ty.x, c[0].g, c[1].g, c[2].g, c[3].g);
        o.b = cubic1(ty.x, c[0].b, c[1].b, c[2].b, c[3].b); o.a = cubic1(ty.x, c[0].a, c[1].a, c[2].a, c[3].a);
        if (j == 0) d[0] = o; else if (j == 1) d[1] = o; else if (j == 2) d[2] = o; else d[3] = o;
    }
    Texel o;
    o.r = cubic1(tz.x, d[0].r, d[1].r, d[2].r, d[3].r); o.g = cubic1(tz.x, d[0].g, d[1].g, d[2].g, d[3].g);
    o.b = cubic1(tz.x, d[0].b, d[1].b, d[2].b, d[3].b); o.a = cubic1(tz.x, d[0].a, d[1].a, d[2].a, d[3].a);
    store_linear(slice_of(a.dst, z), x, y, a.srgbOut, o);
}

// triangle (:2574-2826): acc = src * ((wz * wy) * wx) + acc, contributions ordered by (source slice, source row, source column,
// slice-list entry, row-list entry, column-list entry) - gathered per destination texel from the inverted lists, as in 2-D.
__global__ void __launch_bounds__(256) resize3d_triangle_kernel(Resize3Args a)
{
    const uint32_t x = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y, z = blockIdx.z;
    if (x >= a.dst.width) return;
    const uint32_t zb = a.triOfsZ[z], ze = a.triOfsZ[z + 1];
    const uint32_t yb = a.triOfsY[y], ye = a.triOfsY[y + 1];
    const uint32_t xb = a.triOfsX[x], xe = a.triOfsX[x + 1];
    Texel acc; acc.r = acc.g = acc.b = acc.a = 0.0f;
    uint32_t h = zb;
    while (h < ze)
    {
        const uint32_t sz = a.triZ[h].x;
        uint32_t hEnd = h + 1;
        while (hEnd < ze && a.triZ[hEnd].x == sz) ++hEnd;
        const ImgView S = slice_of(a.src, sz);
        uint32_t i = yb;
        while (i < ye)
        {
            const uint32_t sy = a.triY[i].x;
            uint32_t iEnd = i + 1;
            while (iEnd < ye && a.triY[iEnd].x == sy) ++iEnd;
            uint32_t k = xb;
            while (k < xe)
            {
                const uint32_t sx = a.triX[k].x;
                uint32_t kEnd = k + 1;
                while (kEnd < xe && a.triX[kEnd].x == sx) ++kEnd;
                const Texel p = load_linear(S, sx, sy, a.srgbIn);
                for (uint32_t g = h; g < hEnd; ++g)
                {
                    const float wz = __uint_as_float(a.triZ[g].y);
                    for (uint32_t j = i; j < iEnd; ++j)
                    {
                        const float wzy = wz * __uint_as_float(a.triY[j].y);
                        for (uint32_t m = k; m < kEnd; ++m)
                        {
                            const float w = wzy * __uint_as_float(a.triX[m].y);
                            acc.r = p.r * w + acc.r; acc.g = p.g * w + acc.g; acc.b = p.b * w + acc.b; acc.a = p.a * w + acc.a;
                        }
                    }
                }
                k = kEnd;
            }
            i = iEnd;
        }
        h = hEnd;
    }
    if (a.dst.format == FMT_R10G10B10A2_UNORM || a.dst.format == FMT_R10G10B10A2_UINT) acc.a = acc.a + 0.1f;       // DirectXTexMipmaps.cpp:2767-2786
    store_linear(slice_of(a.dst, z), x, y, a.srgbOut, acc);
}

// ---- ComputeMSE: sum over texels of (v1 - v2)^2 per channel, accumulated in fp64 ----------------------------------------------
__global__ void __launch_bounds__(256) mse_kernel(ImgView a, ImgView b, int srgbA, int srgbB, int ignoreAlpha, double* out)
{
    __shared__ double part[4][4];
    double s[4] = { 0.0, 0.0, 0.0, 0.0 };
    for (uint32_t y = blockIdx.y; y < a.height; y += gridDim.y)
        for (uint32_t x = blockIdx.x * 256u + threadIdx.x; x < a.width; x += gridDim.x * 256u)
        {
            Texel p = load_texel(a.pixels + uint64_t(y) * a.rowPitch, x, a.format);
            Texel q = load_texel(b.pixels + uint64_t(y) * b.rowPitch, x, b.format);
            if (srgbA) { p.r = powf(p.r, 2.2f); p.g = powf(p.g, 2.2f); p.b = powf(p.b, 2.2f); p.a = powf(p.a, 2.2f); }     // XMVectorPow(v, g_Gamma22)
            if (srgbB) { q.r = powf(q.r, 2.2f); q.g = powf(q.g, 2.2f); q.b = powf(q.b, 2.2f); q.a = powf(q.a, 2.2f); }
            const float d[4] = { p.r - q.r, p.g - q.g, p.b - q.b, ignoreAlpha ? 0.0f : p.a - q.a };
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += double(d[c]) * double(d[c]);
        }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        for (int d = 32; d >= 1; d >>= 1) s[c] += __shfl_xor(s[c], d);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { part[wave][0] = s[0]; part[wave][1] = s[1]; part[wave][2] = s[2]; part[wave][3] = s[3]; }
    __syncthreads();
    if (threadIdx.x < 4)
        atomicAdd(&out[threadIdx.x], part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

ImgView make_view(const uint8_t* p, uint64_t pitch, uint32_t w, uint32_t h, int fmt)
{
    ImgView v; v.pixels = const_cast<uint8_t*>(p); v.rowPitch = pitch; v.width = w; v.height = h; v.format = fmt;
    return v;
}

bool srgb_linear_format(int format)
{
    return format == FMT_R8G8B8A8_UNORM_SRGB || format == FMT_B8G8R8A8_UNORM_SRGB || format == FMT_B8G8R8X8_UNORM_SRGB;
}

bool can_srgb(int format)
{
    // LoadScanlineLinear / StoreScanlineLinear: "can't treat A8, XR, Depth, SNORM, UINT, or SINT as sRGB" (:2842-2858)
    switch (format)
    {
    case FMT_R32G32B32A32_FLOAT: case FMT_R16G16B16A16_FLOAT: case FMT_R16G16B16A16_UNORM: case FMT_R32G32_FLOAT:
    case FMT_R8G8B8A8_UNORM: case FMT_R16G16_FLOAT: case FMT_R16G16_UNORM: case FMT_R32_FLOAT: case FMT_R8G8_UNORM:
    case FMT_R16_FLOAT: case FMT_R16_UNORM: case FMT_R8_UNORM: case FMT_B8G8R8A8_UNORM: case FMT_B8G8R8X8_UNORM:
    case FMT_R32G32B32_FLOAT: case FMT_R10G10B10A2_UNORM: case FMT_R11G11B10_FLOAT: case FMT_R9G9B9E5_SHAREDEXP:
    case FMT_R8G8_B8G8_UNORM: case FMT_G8R8_G8B8_UNORM: case FMT_B5G6R5_UNORM: case FMT_B5G5R5A1_UNORM: case FMT_B4G4R4A4_UNORM:
    case FMT_A4B4G4R4_UNORM:         // the whole list of :2825-2849
        return true;
    default:
        return srgb_linear_format(format);
    }
}
} // namespace

hipError_t launch_pack_group(const uint8_t* rows, uint64_t rowsPitch, uint8_t* dst, uint64_t dstPitch, int dstFormat, uint32_t width, uint32_t height, hipStream_t stream)
{
    if (!width || !height) return hipSuccess;
    const uint32_t per = group_texels(dstFormat), groups = (width + per - 1) / per;
    hipLaunchKernelGGL(pack_group_kernel, dim3((groups + 255) / 256, grid_rows(height)), dim3(256), 0, stream,
                       make_view(rows, rowsPitch, width, height, FMT_R32G32B32A32_FLOAT), make_view(dst, dstPitch, width, height, dstFormat));
    return hipGetLastError();
}

hipError_t launch_convert(const uint8_t* src, uint64_t srcPitch, int srcFormat, uint8_t* dst, uint64_t dstPitch, int dstFormat,
                          uint32_t width, uint32_t height, const ConvertPlan& plan, float threshold, hipStream_t stream)
{
    if (!width || !height) return hipSuccess;
    const FmtInfo* in = format_info(srcFormat);
    const FmtInfo* out = format_info(dstFormat);
    // four texels per lane through 16-byte loads / stores where a texel is a whole number of dwords on both sides and rows are 16-byte aligned
    if (in && out && !((in->cls | out->cls) & FC_GROUP) && in->bpp >= 32 && out->bpp >= 32 && (in->bpp % 32) == 0 && (out->bpp % 32) == 0 && (width % 4u) == 0 &&
        ((reinterpret_cast<uintptr_t>(src) | srcPitch | reinterpret_cast<uintptr_t>(dst) | dstPitch) & 15u) == 0)
    {
        const uint32_t quads = width / 4u;
        const uint32_t gx = (quads + 255u) / 256u;
        const uint32_t sq = uint32_t(in->bpp / 8u) * 4u, dq = uint32_t(out->bpp / 8u) * 4u;
        const ImgView sv = make_view(src, srcPitch, width, height, srcFormat), dv = make_view(dst, dstPitch, width, height, dstFormat);
        // row groups per workgroup column: enough workgroups to fill 256 CUs several times over, few enough that a lane streams several groups
#define DXTEX_QUAD(SQ, DQ, ROWS) do { const uint32_t groups = (height + (ROWS) - 1u) / (ROWS); \
            const uint32_t gy = std::min<uint32_t>(groups, std::max<uint32_t>(1u, 8192u / gx)); \
            hipLaunchKernelGGL((convert_quad_kernel<SQ, DQ, ROWS>), dim3(gx, gy), dim3(256), 0, stream, sv, dv, plan, threshold, sq, dq); } while (0)
        if (sq == 16u && dq == 16u) DXTEX_QUAD(16, 16, 4);
        else if (sq == 16u && dq == 32u) DXTEX_QUAD(16, 32, 4);
        else if (sq == 16u && dq == 64u) DXTEX_QUAD(16, 64, 4);
        else if (sq == 32u && dq == 16u) DXTEX_QUAD(32, 16, 2);
        else if (sq == 32u && dq == 32u) DXTEX_QUAD(32, 32, 2);
        else if (sq == 32u && dq == 64u) DXTEX_QUAD(32, 64, 2);
        else if (sq == 64u && dq == 16u) DXTEX_QUAD(64, 16, 1);
        else if (sq == 64u && dq == 32u) DXTEX_QUAD(64, 32, 1);
        else DXTEX_QUAD(0, 0, 1);
#undef DXTEX_QUAD
        return hipGetLastError();
    }
    hipLaunchKernelGGL(convert_kernel, dim3((width + 255) / 256, grid_rows(height)), dim3(256), 0, stream,
                       make_view(src, srcPitch, width, height, srcFormat), make_view(dst, dstPitch, width, height, dstFormat), plan, threshold);
    return hipGetLastError();
}

hipError_t launch_resize(const uint8_t* src, uint64_t srcPitch, uint32_t srcW, uint32_t srcH, uint8_t* dst, uint64_t dstPitch,
                         uint32_t dstW, uint32_t dstH, int format, uint32_t filterMode, uint32_t filterFlags, bool mipAlias,
                         const TriangleTables* tri, hipStream_t stream, const uint8_t* staleLevel, uint64_t stalePitch, uint32_t staleW, int dstFormat)
{
    if (!dstW || !dstH) return hipSuccess;
    ResizeArgs a;
    a.stale = make_view(staleLevel, stalePitch, staleW, 2, format);
    a.src = make_view(src, srcPitch, srcW, srcH, format);
    a.dst = make_view(dst, dstPitch, dstW, dstH, dstFormat >= 0 ? dstFormat : format);
    // sRGB formats filter in linear space; TEX_FILTER_SRGB forces it for the other colour formats (:2803-2945)
    const bool wantIn = srgb_linear_format(format) || (filterFlags & 0x1000000u), wantOut = srgb_linear_format(format) || (filterFlags & 0x2000000u);
    a.srgbIn = (can_srgb(format) && wantIn) ? 1 : 0;
    a.srgbOut = (can_srgb(format) && wantOut) ? 1 : 0;
    a.wrapU = (filterFlags & 0x1u) != 0; a.wrapV = (filterFlags & 0x2u) != 0;
    a.mirrorU = (filterFlags & 0x10u) != 0; a.mirrorV = (filterFlags & 0x20u) != 0;
    a.mipAlias = mipAlias ? 1 : 0;
    a.triOfsX = tri ? tri->ofsX : nullptr; a.triX = tri ? reinterpret_cast<const uint2*>(tri->entX) : nullptr;
    a.triOfsY = tri ? tri->ofsY : nullptr; a.triY = tri ? reinterpret_cast<const uint2*>(tri->entY) : nullptr;
    const dim3 grid((dstW + 255) / 256, grid_rows(dstH)), block(256);
    switch (filterMode)
    {
    case 0x100000u: hipLaunchKernelGGL(resize_point_kernel, grid, block, 0, stream, a); break;
    case 0x200000u: hipLaunchKernelGGL(resize_linear_kernel, grid, block, 0, stream, a); break;
    case 0x300000u:
        // the 2:1 RGBA8 case of a power-of-two mip chain has a separable kernel (a column strip per lane); the small levels take it too (6 us a
        // launch against 10 - 30 us of the general kernel's sixteen dependent taps)
        if (format == FMT_R8G8B8A8_UNORM && !a.srgbIn && !a.srgbOut && srcW == 2 * dstW && srcH == 2 * dstH &&
            !a.wrapU && !a.wrapV && !a.mirrorU && !a.mirrorV && (srcPitch % 4) == 0 && (dstPitch % 4) == 0 &&
            (reinterpret_cast<uintptr_t>(src) % 4) == 0 && (reinterpret_cast<uintptr_t>(dst) % 4) == 0)
        {
            // rows per lane: long strips amortise the two extra row passes at their top; short ones keep a small level spread over the chip
            // (at least ~4096 wavefronts while that leaves 4 rows or more per strip)
            uint32_t strip = 32;
            // two destination texels per lane where the rows allow 16-byte loads and 8-byte stores (every level of a power-of-two chain down to 2 texels)
#if !defined(DXTEX_CUBIC_X2)
#define DXTEX_CUBIC_X2 1               // 0: every level through the one-texel-per-lane kernel (A/B builds)
#endif
            const bool pairs = DXTEX_CUBIC_X2 && (dstW % 2) == 0 && (srcPitch % 16) == 0 && (dstPitch % 8) == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 &&
                               (reinterpret_cast<uintptr_t>(dst) % 8) == 0;
            const uint32_t lanesX = pairs ? dstW / 2 : dstW;
            while (strip > 4 && uint64_t((lanesX + 63) / 64) * ((dstH + strip - 1) / strip) < 4096) strip >>= 1;
            if (pairs) hipLaunchKernelGGL(resize_cubic_half_rgba8_x2_kernel, dim3((lanesX + 255) / 256, grid_rows((dstH + strip - 1) / strip)), block, 0, stream, a, strip);
            else hipLaunchKernelGGL(resize_cubic_half_rgba8_kernel, dim3((dstW + 255) / 256, grid_rows((dstH + strip - 1) / strip)), block, 0, stream, a, strip);
        }
        else
            hipLaunchKernelGGL(resize_cubic_kernel, grid, block, 0, stream, a);
        break;
    case 0x400000u:
        if (format == FMT_R8G8B8A8_UNORM && !a.srgbIn && !a.srgbOut && srcW == 2 * dstW && srcH == 2 * dstH && (dstW % 2) == 0 && dstW >= 256 &&
            (srcPitch % 16) == 0 && (dstPitch % 8) == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 && (reinterpret_cast<uintptr_t>(dst) % 8) == 0)
            hipLaunchKernelGGL(resize_box_half_rgba8_kernel, dim3((dstW / 2 + 255) / 256, grid_rows(dstH)), block, 0, stream, a);
        else
            hipLaunchKernelGGL(resize_box_kernel, grid, block, 0, stream, a);
        break;
    case 0x500000u: hipLaunchKernelGGL(resize_triangle_kernel, grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool resize_half_tail_applies(const MipLevel* levels, int nlevels, int format, uint32_t filterFlags);
hipError_t launch_resize_tail(const MipLevel* levels, int nlevels, int format, uint32_t filterMode, uint32_t filterFlags,
                              const MipLevel* twoHigh, hipStream_t stream)
{
    if (nlevels < 2) return hipSuccess;
    const bool halving = (filterMode == 0x300000u || filterMode == 0x400000u) && resize_half_tail_applies(levels, nlevels, format, filterFlags);
    if (filterMode == 0x300000u && !halving)
    {
        // a cubic chain the one-workgroup form does not cover (not an exact-halving RGBA8 clamp chain): one launch per level, as GenerateMipMaps does above the tail
        for (int k = 1; k < nlevels; ++k)
        {
            const hipError_t e = launch_resize(levels[k - 1].pixels, levels[k - 1].pitch, levels[k - 1].width, levels[k - 1].height, levels[k].pixels, levels[k].pitch,
                                               levels[k].width, levels[k].height, format, filterMode, filterFlags, true, nullptr, stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (halving)
    {
        // the LDS tail: RGBA8, no sRGB, every level an exact halving (cubic: clamp addressing; admitted by the caller, checked again here)
        CubicTailArgs c;
        c.src = levels[0].pixels; c.srcPitch = levels[0].pitch; c.srcW = levels[0].width; c.srcH = levels[0].height; c.nlevels = nlevels - 1;
        if (format != FMT_R8G8B8A8_UNORM || (filterFlags & 0x3000077u) || c.nlevels > kTailMaxLevels || c.srcW > kTailSide || c.srcH > kTailSide) return hipErrorInvalidValue;
        for (int k = 1; k < nlevels; ++k)
        {
            if (levels[k].width * 2u != levels[k - 1].width || levels[k].height * 2u != levels[k - 1].height || (levels[k].pitch % 4) != 0 ||
                (reinterpret_cast<uintptr_t>(levels[k].pixels) % 4) != 0) return hipErrorInvalidValue;
            c.dst[k - 1] = levels[k].pixels; c.dstPitch[k - 1] = levels[k].pitch;
        }
        for (int k = nlevels - 1; k < kTailMaxLevels; ++k) { c.dst[k] = c.dst[0]; c.dstPitch[k] = c.dstPitch[0]; }
        if (filterMode == 0x300000u) hipLaunchKernelGGL(resize_half_tail_rgba8_kernel<true>, dim3(1), dim3(1024), 0, stream, c);
        else hipLaunchKernelGGL(resize_half_tail_rgba8_kernel<false>, dim3(1), dim3(1024), 0, stream, c);
        return hipGetLastError();
    }
    TailArgs t;
    ResizeArgs& a = t.a;
    a.stale = make_view(nullptr, 0, 0, 2, format);
    a.src = make_view(levels[0].pixels, levels[0].pitch, levels[0].width, levels[0].height, format);
    a.dst = a.src;
    const bool wantIn = srgb_linear_format(format) || (filterFlags & 0x1000000u), wantOut = srgb_linear_format(format) || (filterFlags & 0x2000000u);
    a.srgbIn = (can_srgb(format) && wantIn) ? 1 : 0;
    a.srgbOut = (can_srgb(format) && wantOut) ? 1 : 0;
    a.wrapU = (filterFlags & 0x1u) != 0; a.wrapV = (filterFlags & 0x2u) != 0;
    a.mirrorU = (filterFlags & 0x10u) != 0; a.mirrorV = (filterFlags & 0x20u) != 0;
    a.mipAlias = 1;
    a.triOfsX = nullptr; a.triX = nullptr; a.triOfsY = nullptr; a.triY = nullptr;
    t.mode = filterMode;
    t.twoHigh = twoHigh ? make_view(twoHigh->pixels, twoHigh->pitch, twoHigh->width, twoHigh->height, format) : make_view(nullptr, 0, 0, 2, format);
    for (int at = 1; at < nlevels; )
    {
        t.nlevels = std::min(kTailMaxLevels, nlevels - at);
        for (int k = 0; k < t.nlevels; ++k) t.level[k] = make_view(levels[at + k].pixels, levels[at + k].pitch, levels[at + k].width, levels[at + k].height, format);
        for (int k = t.nlevels; k < kTailMaxLevels; ++k) t.level[k] = t.level[0];
        hipLaunchKernelGGL(resize_tail_kernel, dim3(1), dim3(1024), 0, stream, t);
        // a chain with more than kTailMaxLevels tail levels (cannot happen below 64 x 64, kept for safety) continues from the last one written
        a.src = t.level[t.nlevels - 1];
        at += t.nlevels;
    }
    return hipGetLastError();
}

bool resize_half_tail_applies(const MipLevel* levels, int nlevels, int format, uint32_t filterFlags)
{
    // the LDS tail of a power-of-two RGBA8 chain (resize_cubic_tail_rgba8_kernel): every remaining level halves both sides exactly
    if (nlevels < 2 || nlevels - 1 > kTailMaxLevels || format != FMT_R8G8B8A8_UNORM || (filterFlags & 0x3000077u)) return false;       // sRGB, wrap, mirror bits
    if (levels[0].width > kTailSide || levels[0].height > kTailSide || (levels[0].pitch % 4) != 0 || (reinterpret_cast<uintptr_t>(levels[0].pixels) % 4) != 0) return false;
    for (int k = 1; k < nlevels; ++k)
        if (levels[k].width * 2u != levels[k - 1].width || levels[k].height * 2u != levels[k - 1].height || (levels[k].pitch % 4) != 0 ||
            (reinterpret_cast<uintptr_t>(levels[k].pixels) % 4) != 0) return false;
    return true;
}

bool resize_cubic_tail_applies(const MipLevel* levels, int nlevels, int format, uint32_t filterFlags)
{
    return resize_half_tail_applies(levels, nlevels, format, filterFlags);
}

bool resize_tail_applies(uint32_t srcW, uint32_t srcH, uint32_t filterMode)
{
    // measured (8192^2 chain, rocprofv3): the box tail takes 15.8 us in one workgroup against 6 launches x 3.2 us; the cubic tail 84 us
    // against 6 x 7 us (sixteen dependent-latency taps per texel and no other workgroup to hide them) - so cubic keeps its launches
    return srcW <= kTailSide && srcH <= kTailSide && (filterMode == 0x100000u || filterMode == 0x200000u || filterMode == 0x400000u);
}

hipError_t launch_pmalpha(const uint8_t* src, uint64_t srcPitch, uint8_t* dst, uint64_t dstPitch, int format, uint32_t width, uint32_t height,
                          uint32_t pmFlags, hipStream_t stream)
{
    if (!width || !height) return hipSuccess;
    // TEX_PMALPHA_IGNORE_SRGB (0x1): plain Load/StoreScanline; otherwise the *Linear wrappers with the SRGB_IN/OUT bits (:68-112)
    const bool linear = !(pmFlags & 0x1u);
    const bool wantIn = linear && (srgb_linear_format(format) || (pmFlags & 0x1000000u)), wantOut = linear && (srgb_linear_format(format) || (pmFlags & 0x2000000u));
    hipLaunchKernelGGL(pmalpha_kernel, dim3((width + 255) / 256, grid_rows(height)), dim3(256), 0, stream,
                       make_view(src, srcPitch, width, height, format), make_view(dst, dstPitch, width, height, format),
                       (can_srgb(format) && wantIn) ? 1 : 0, (can_srgb(format) && wantOut) ? 1 : 0, (pmFlags & 0x2u) ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_scale_alpha(const uint8_t* src, uint64_t srcPitch, uint8_t* dst, uint64_t dstPitch, int format, uint32_t width, uint32_t height,
                              float scale, hipStream_t stream)
{
    if (!width || !height) return hipSuccess;
    hipLaunchKernelGGL(scale_alpha_kernel, dim3((width + 255) / 256, grid_rows(height)), dim3(256), 0, stream,
                       make_view(src, srcPitch, width, height, format), make_view(dst, dstPitch, width, height, format), scale);
    return hipGetLastError();
}

hipError_t launch_alpha_coverage(const uint8_t* src, uint64_t srcPitch, int format, uint32_t width, uint32_t height, float scale, float alphaReference,
                                 unsigned long long* count, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    if (width < 2 || height < 2) return hipSuccess;
    hipLaunchKernelGGL(alpha_coverage_kernel, dim3((width - 1 + 255) / 256, grid_rows(height - 1)), dim3(256), 0, stream,
                       make_view(src, srcPitch, width, height, format), scale, alphaReference, count);
    return hipGetLastError();
}

hipError_t launch_alpha_below(const uint8_t* src, uint64_t srcPitch, int format, uint32_t width, uint32_t height, float threshold,
                              unsigned long long* count, hipStream_t stream)
{
    if (!width || !height) return hipSuccess;
    const uint32_t gx = std::min<uint32_t>((width + 255) / 256, 64), gy = std::min<uint32_t>(height, 2048);
    hipLaunchKernelGGL(alpha_below_kernel, dim3(gx, gy), dim3(256), 0, stream, make_view(src, srcPitch, width, height, format), threshold, count);
    return hipGetLastError();
}

hipError_t launch_resize3d(const VolumeView& src, const VolumeView& dst, uint32_t filterMode, uint32_t filterFlags, const TriangleTables3* tri,
                           hipStream_t stream, const uint8_t* staleU, const uint8_t* staleV, uint64_t stalePitch, uint32_t staleW)
{
    if (!dst.width || !dst.height || !dst.depth) return hipSuccess;
    Resize3Args a;
    auto vol = [](const VolumeView& v) { Vol o; o.pixels = const_cast<uint8_t*>(v.pixels); o.rowPitch = v.rowPitch; o.slicePitch = v.slicePitch;
                                         o.width = v.width; o.height = v.height; o.depth = v.depth; o.format = v.format; return o; };
    a.src = vol(src); a.dst = vol(dst);
    const int format = src.format;
    const bool wantIn = srgb_linear_format(format) || (filterFlags & 0x1000000u), wantOut = srgb_linear_format(format) || (filterFlags & 0x2000000u);
    a.srgbIn = (can_srgb(format) && wantIn) ? 1 : 0;
    a.srgbOut = (can_srgb(format) && wantOut) ? 1 : 0;
    a.wrapU = (filterFlags & 0x1u) != 0; a.wrapV = (filterFlags & 0x2u) != 0; a.wrapW = (filterFlags & 0x4u) != 0;
    a.mirrorU = (filterFlags & 0x10u) != 0; a.mirrorV = (filterFlags & 0x20u) != 0; a.mirrorW = (filterFlags & 0x40u) != 0;
    a.staleU = make_view(staleU, stalePitch, staleW, 2, format);
    a.staleV = make_view(staleV, stalePitch, staleW, 2, format);
    a.triOfsX = tri ? tri->ofsX : nullptr; a.triX = tri ? reinterpret_cast<const uint2*>(tri->entX) : nullptr;
    a.triOfsY = tri ? tri->ofsY : nullptr; a.triY = tri ? reinterpret_cast<const uint2*>(tri->entY) : nullptr;
    a.triOfsZ = tri ? tri->ofsZ : nullptr; a.triZ = tri ? reinterpret_cast<const uint2*>(tri->entZ) : nullptr;
    const dim3 grid((dst.width + 255) / 256, dst.height, dst.depth), block(256);
    switch (filterMode)
    {
    case 0x100000u: hipLaunchKernelGGL(resize3d_point_kernel, grid, block, 0, stream, a); break;
    case 0x200000u: hipLaunchKernelGGL(resize3d_linear_kernel, grid, block, 0, stream, a); break;
    case 0x300000u: hipLaunchKernelGGL(resize3d_cubic_kernel, grid, block, 0, stream, a); break;
    case 0x400000u: hipLaunchKernelGGL(resize3d_box_kernel, grid, block, 0, stream, a); break;
    case 0x500000u: hipLaunchKernelGGL(resize3d_triangle_kernel, grid, block, 0, stream, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_mse(const uint8_t* a, uint64_t aPitch, int aFormat, const uint8_t* b, uint64_t bPitch, int bFormat,
                      uint32_t width, uint32_t height, double* out4, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(out4, 0, 4 * sizeof(double), stream);
    if (e != hipSuccess) return e;
    if (!width || !height) return hipSuccess;
    const bool ignoreAlpha = aFormat == FMT_B8G8R8X8_UNORM || aFormat == FMT_B8G8R8X8_UNORM_SRGB || bFormat == FMT_B8G8R8X8_UNORM || bFormat == FMT_B8G8R8X8_UNORM_SRGB;
    const uint32_t gx = std::min<uint32_t>((width + 255) / 256, 64), gy = std::min<uint32_t>(height, 1024);
    hipLaunchKernelGGL(mse_kernel, dim3(gx, gy), dim3(256), 0, stream, make_view(a, aPitch, width, height, aFormat),
                       make_view(b, bPitch, width, height, bFormat), srgb_linear_format(aFormat) ? 1 : 0, srgb_linear_format(bFormat) ? 1 : 0,
                       ignoreAlpha ? 1 : 0, out4);
    return hipGetLastError();
}
} // namespace dxtex
