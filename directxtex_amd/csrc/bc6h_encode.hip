// BC6H encoder kernels for gfx950: byte-identical to D3DX_BC6H::Encode (BC6HBC7.cpp:1817-1859) for BC6H_UF16 and
// BC6H_SF16. What one lane computes is in bc6h_core.h; this file spreads it over the machine the same way
// bc7_encode.hip does for BC7:
//   rough   : one wavefront per block, lane = partition shape (32 two-region shapes + the one-region case). Float
//             OptimizeRGB seeds, RoughMSE against the unquantised palette, the reference's partial selection sort
//             (:1836-1848) -> the 8 best shapes and their seeds. The ranking does not depend on the mode, so it is
//             done once for the ten two-region modes.
//   per mode (14, in the encoder's order):
//     pre     lane = (block, candidate shape, region): QuantizeEndPts -> AssignIndices -> SwapIndices ->
//             TransformForward -> EndPointsFit (the two region lanes of a candidate combine with shuffles).
//     bin     counting sort of the live tasks by texel count (search_common.h).
//     perturb persistent wavefronts pulling tasks from a global queue; OptimizeOne (:2145-2194) as straight-line
//             PERTURB macro-ops of 2 * precision candidate evaluations each (BC6H has no exhaustive phase).
//     post    AssignIndices / SwapIndices of the optimised endpoints, org-vs-opt decision (:2401-2424), first minimum
//             over the block's candidates, EmitBlock into the block's running best if it is strictly better - modes run
//             in order on one stream, which reproduces Encode's "fBestErr" sequence.
//   store   running best -> destination image.
// The search is VALU-bound fp32 work (10 ops per texel x palette entry); HBM traffic is 9 B/texel algorithmic.
#include "dxtex_kernels.h"
#include "bc67_tables.h"
#include "bc7_core.h"
#include "bc6h_core.h"
#include "search_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <vector>

namespace dxtex
{
namespace
{
using namespace bc6h;

// A task's quantised endpoints as six 16-bit fields: every mode's endpoint precision is at most 16 bits (ms_aInfo, :1051-1067); Quantize leaves
// unsigned formats in 0 ... 2^prec - 1 and signed ones in -(2^(prec-1) - 1) ... 2^(prec-1) - 1, so the extension on the way back is the format's.
// One exception: PerturbOne tries every value in [0, 2^prec) whatever the signedness (:2112-2118), so with SF16 and the 16-bit one-region mode
// a search can END on a component above 32767. The only 16-bit mode is 16:4 TRANSFORMED (ms_aInfo, :1066): EndPointsFit (:1962-1968) tests A
// against 16 signed bits - A > 32767 never fits (NBits = 17) and the reference drops the optimised endpoints - but B as the DELTA B - A
// against 4 bits, so A <= 32767 with B in 32768 ... A + 7 does fit there. The fields would wrap, so the search kernels mark the record
// instead (ep16_mark -> Rec6::err, which nobody reads as an error after the search: post recomputes it): -1 = A overflowed, the result does
// not fit, the unoptimised endpoints stand; -(2 + m) = only components of B did, m = their bit mask - bc6h_post_kernel adds the 65536 back
// to those components and runs TransformForward / EndPointsFit / AssignIndices on the true values, as the reference does.
// (Round 3 kept them as six ints: 72 bytes of records per task, 1.2 GB written by every mode's pre and read by its post.)
struct Ep16 { uint32_t a01, a2b0, b12; };
__device__ __forceinline__ Ep16 pack_ep16(const int (&A)[3], const int (&B)[3])
{
    Ep16 e;
    e.a01 = (uint32_t(A[0]) & 0xFFFFu) | (uint32_t(A[1]) << 16);
    e.a2b0 = (uint32_t(A[2]) & 0xFFFFu) | (uint32_t(B[0]) << 16);
    e.b12 = (uint32_t(B[1]) & 0xFFFFu) | (uint32_t(B[2]) << 16);
    return e;
}
__device__ __forceinline__ void unpack_ep16(const Ep16& e, bool sg, int (&A)[3], int (&B)[3])
{
    const auto lo = [sg](uint32_t w) { return sg ? int(int16_t(w & 0xFFFFu)) : int(w & 0xFFFFu); };
    const auto hi = [sg](uint32_t w) { return sg ? (int(w) >> 16) : int(w >> 16); };
    A[0] = lo(e.a01); A[1] = hi(e.a01); A[2] = lo(e.a2b0); B[0] = hi(e.a2b0); B[1] = lo(e.b12); B[2] = hi(e.b12);
}
__device__ __forceinline__ float ep16_mark(bool sg, const int (&A)[3], const int (&B)[3])
{
    if (!sg) return 0.0f;
    if (A[0] > 32767 || A[1] > 32767 || A[2] > 32767) return -1.0f;
    const int m = int(B[0] > 32767) | (int(B[1] > 32767) << 1) | (int(B[2] > 32767) << 2);
    return m ? -float(2 + m) : 0.0f;
}
// The filter kernel's per-lane task state (bc6h_core.h: Perturb6, twelve dwords) in SIX: the endpoints as Ep16, err, err0, and new0 / ch / sub /
// do_b in one word. As twelve dwords the compiler kept the struct in SCRATCH (the kernel is compiled for four waves per SIMD) and read it back
// at every PerturbOne call and - three dependent round trips to memory - for every step whose candidates reach the exact list: by the kernel's own
// clock (DXTEX_BC6H_STATS, profiles/r06_bc6h.md) the "list write" that contains those loads was 35 - 44 % of a wavefront's time. The
// two-region modes' endpoints have at most eleven bits (ms_aInfo, :1051-1060), new0 is an endpoint value: sixteen-bit fields hold them.
struct PkState { Ep16 ep; float err, err0; uint32_t misc; };       // misc = new0 (low 16 bits, sign-extended back) | ch << 16 | sub << 18 | do_b << 20
__device__ __forceinline__ PkState pk_pack(const Perturb6& s)
{
    PkState p; p.ep = pack_ep16(s.ep.A, s.ep.B); p.err = s.err; p.err0 = s.err0;
    p.misc = (uint32_t(s.new0) & 0xFFFFu) | (uint32_t(s.ch) << 16) | (uint32_t(s.sub) << 18) | (uint32_t(s.do_b) << 20);
    return p;
}
__device__ __forceinline__ Perturb6 pk_unpack(const PkState& p, bool sg)
{
    Perturb6 s; unpack_ep16(p.ep, sg, s.ep.A, s.ep.B); s.err = p.err; s.err0 = p.err0;
    s.new0 = int(int16_t(p.misc & 0xFFFFu)); s.ch = int((p.misc >> 16) & 3u); s.sub = int((p.misc >> 18) & 3u); s.do_b = int((p.misc >> 20) & 1u);
    return s;
}
struct Rec6 { Ep16 ep; float err; };                               // 16 bytes per task: the search's start, then its result
struct Best6 { float err; uint32_t mode; uint64_t lo, hi; };       // 24 bytes per block; mode = position of the winner's mode in the encoder's order
struct OrgSave { Ep16 ep; float err; uint64_t idx; };              // 24 bytes per task: Refine's unoptimised half, pre -> post
struct OptSave { uint64_t idx; float err; uint32_t swapped; };         // 16 bytes per task: Refine's optimised half of a precision trio's first member, post -> post

enum : int { SEED_INTS = 17 * 6 + 2 };      // 8 shapes x 2 regions + the one-region seed, 6 ints each (+ pad)

struct Bc6hArgs
{
    SegTable seg;           // the images behind this pass (search_common.h)
    uint32_t nblocks;       // blocks in this pass; scratch arrays are indexed by pass-local block number
    int isSigned;
    int prune;              // 0 = search every candidate like the reference does (DXTEX_BC6H_NO_PRUNE, for A/B runs)
    uint32_t taskBase;      // one-region modes: the four modes' tasks share the arrays, mode slot m owns [m * nblocks, (m + 1) * nblocks)
    int prec1[4];           // ... and the search kernel runs them as ONE list: endpoint precision of mode slot m
    float* fpix;            // nblocks x 3 x 16: the block's texels as INTColor values held in floats (r[16], g[16], b[16])
    uint8_t* lists;         // nblocks x 8 shape ids
    int* seeds;             // nblocks x SEED_INTS
    Rec6* recs;
    OrgSave* orgs;          // per task: endpoints (after SwapIndices), error and indices of the unoptimised candidate
    uint2* order; uint32_t* tinfo; uint32_t* counters;
    Best6* best;
    float* bounds;          // nblocks x 17: region_lower_bound6 of the 8 ranked shapes x 2 regions and of the whole block (the same for every mode)
    int boundsReady;        // 0: this launch computes and stores them, 1: it reads them
    int filterStats;        // development build: bc6h_perturb_filter_kernel counts its steps and exact rounds in counters[48..]
    int samePrec;           // two-region modes: the previous mode had the same endpoint precision and its task arrays are still in place
    struct OptSave* opts;   // two-region modes of one precision: post's AssignIndices of the optimised endpoints, first member -> the others (saveOpt)
    int saveOpt;            // 1: the next mode has the same precision - post stores what it assigned
    ModeRt mode;
};

// tinfo bit 23: the task was searched by an earlier mode of the same precision (see bc6h_pre_kernel); bits 24.. = subset size, 0..15 = texel mask
constexpr uint32_t kDoneBit6 = 1u << 23;

// Running order of the two-region modes (positions in the encoder's order, ms_aInfo :1051-1067): the 9-bit mode and the 8-bit trio first,
// the 7-bit mode (which nearly always fits, 83 % of its tasks live when it runs second as in the encoder's order) after them, the 6-bit mode
// last. Measured on the cfg3 image: encoder's order 112.9 ms, this order 101.3 ms (the 7-bit search 21.6 -> 11.4 ms: what the 9- and 8-bit
// modes leave on the table prunes half of it), 9-bit after the 8-bit trio 105.4, 6-bit first 125.9. Same bytes in every order.
#if !defined(DXTEX_BC6H_DEFAULT_ORDER)
#define DXTEX_BC6H_DEFAULT_ORDER "5,6,7,8,0,2,3,4,1,9"
#endif

__device__ __forceinline__ Texels slot_texels(float* slot /* &sSlot[0][0][lane] */, int np)
{
    Texels t; t.r = slot; t.g = slot + 16 * 64; t.b = slot + 32 * 64; t.stride = 64; t.np = np;
    return t;
}

// Copies the texels selected by `mask` from a block's float planes (r[16], g[16], b[16]) into the lane's LDS column
// and returns their packed block positions.
__device__ __forceinline__ int gather_texels(const float* planes, uint32_t mask, float* slot, uint64_t& pos)
{
    int np = 0; pos = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((mask >> i) & 1u)
        {
            slot[np * 64] = planes[i]; slot[(16 + np) * 64] = planes[16 + i]; slot[(32 + np) * 64] = planes[32 + i];
            pos |= uint64_t(i) << (4 * np);
            ++np;
        }
    return np;
}

// The packed block positions of the texels selected by `mask` (no copy: pre / post of the two-region modes read the block's planes in place).
__device__ __forceinline__ int region_positions(uint32_t mask, uint64_t& pos)
{
    int np = 0; pos = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((mask >> i) & 1u) { pos |= uint64_t(i) << (4 * np); ++np; }
    return np;
}

// Stages the planes of the BPW blocks a wavefront of pre / post works on (48 floats each) in LDS: one 16-byte load per lane instead
// of twelve, and no per-lane copy of the block in registers or LDS columns - the two-region kernels were held at three waves per SIMD
// by 48 KiB of columns per workgroup and 136 registers, and waited (issue utilisation 0.37 / 0.59).
template<int BPW>
__device__ __forceinline__ void stage_planes(const Bc6hArgs& a, uint32_t nbFirst, int lane, float* tile)
{
    static_assert(BPW * 12 <= 64, "one float4 per lane");
    if (lane < BPW * 12)
    {
        const uint32_t nb = min(nbFirst + uint32_t(lane) / 12u, a.nblocks - 1);
        const float4 v = reinterpret_cast<const float4*>(a.fpix + uint64_t(nb) * 48)[uint32_t(lane) % 12u];
        reinterpret_cast<float4*>(tile)[lane] = v;
    }
    wave_lds_sync();
}

// ---- rough ------------------------------------------------------------------------------------------------------------------
// six wavefronts per SIMD (94 -> 80 registers, no spill): 7.60 -> 7.37 ms per cfg3 image; 7 / 8: 7.49 / 7.38 (round 6)
#if !defined(DXTEX_ROUGH6_WGS)
#define DXTEX_ROUGH6_WGS 6
#endif
__global__ void __launch_bounds__(256, DXTEX_ROUGH6_WGS) bc6h_rough_kernel(Bc6hArgs a)
{
    __shared__ float sF[4][64];
    __shared__ float sP[4][48];
    __shared__ int sSeedA[4][64][3], sSeedB[4][64][3];     // the fits of the 32 shapes x 2 regions (code = shape * 2 + region)
    __shared__ float sPart[4][64];                         // their rough errors; < 0: a region of one or two texels (adds nothing)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nb = blockIdx.x * 4 + wave;
    const bool inRange = nb < a.nblocks;
    const bool sg = a.isSigned != 0;

    if (inRange && lane < 16)
    {
        // one texel of the block, with the reference's partial-block replication (DirectXTexCompress.cpp:315-341)
        const BcSeg& im = seg_of(a.seg, nb);
        const uint32_t gb = im.nb0 + (nb - im.l0);
        const uint32_t by = gb / im.nbw, bx = gb - by * im.nbw;
        const uint32_t x0 = bx * 4, y0 = by * 4;
        const uint32_t pw = min(4u, im.src.width - x0), ph = min(4u, im.src.height - y0);
        const uint32_t sx = x0 + replicate_src(lane & 3, pw), sy = y0 + replicate_src(lane >> 2, ph);
        const Texel px = convert_texel(load_texel(im.src.pixels + uint64_t(sy) * im.src.rowPitch, sx, im.src.format), im.src.tcv, im.src.tsw);
        sF[wave][lane * 4 + 0] = px.r; sF[wave][lane * 4 + 1] = px.g; sF[wave][lane * 4 + 2] = px.b; sF[wave][lane * 4 + 3] = px.a;
        const float ir = float(float_to_int16f(px.r, sg)), ig = float(float_to_int16f(px.g, sg)), ib = float(float_to_int16f(px.b, sg));
        sP[wave][lane] = ir; sP[wave][16 + lane] = ig; sP[wave][32 + lane] = ib;
        float* gp = a.fpix + uint64_t(nb) * 48;
        gp[lane] = ir; gp[16 + lane] = ig; gp[32 + lane] = ib;
    }
    if (inRange && lane == 0) { Best6 b; b.err = 3.402823466e+38f; b.mode = 0xFFFFFFFFu; b.lo = 0; b.hi = 0; a.best[nb] = b; }
    __syncthreads();

    // Every fit is ONE subset of a two-region shape (the one-region fit has a kernel of its own, bc6h_block_seed_kernel). The 4 x 64
    // fits of the workgroup's four blocks are dealt to the wavefronts BY SUBSET SIZE (kFit2Order32: largest first, four groups of
    // sixteen; wavefront w takes group w of all four blocks, lane = block * 16 + entry), so the texel loops of a wavefront run
    // 13 / 10 / 8 / 4 trips instead of 13 in every wavefront with a lane per (shape, region). The subsets are read in place from the
    // block's planes (no per-lane columns: 50 KiB of LDS held the kernel at 3 waves per SIMD).
    {
        const uint32_t blk = uint32_t(lane) >> 4, ent = uint32_t(lane) & 15u;
        if (blockIdx.x * 4 + blk < a.nblocks)
        {
            const uint32_t code = kFit2Order32[wave * 16 + ent];
            const uint32_t m1 = uint32_t(kPart2Mask[code >> 1]);
            const uint32_t mask = (code & 1u) ? m1 : ((~m1) & 0xFFFFu);
            const float* fpx = sF[blk];
            const float* planes = sP[blk];
            EndPts seed;
            float part = -1.0f;
            uint64_t pos;
            const int np = region_positions(mask, pos);
            const uint32_t p0 = uint32_t(pos) & 15u, p1 = uint32_t(pos >> 4) & 15u;
            if (np == 1)
            {
                seed.A[0] = seed.B[0] = int(planes[p0]); seed.A[1] = seed.B[1] = int(planes[16 + p0]); seed.A[2] = seed.B[2] = int(planes[32 + p0]);
            }
            else if (np == 2)
            {
                seed.A[0] = int(planes[p0]); seed.A[1] = int(planes[16 + p0]); seed.A[2] = int(planes[32 + p0]);
                seed.B[0] = int(planes[p1]); seed.B[1] = int(planes[16 + p1]); seed.B[2] = int(planes[32 + p1]);
            }
            else
            {
                float X[4], Y[4];
                bc7::seed_fit<false>(fpx, mask, X, Y);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                {
                    seed.A[c] = clamp_seed(float_to_int16f(X[c], sg), sg);
                    seed.B[c] = clamp_seed(float_to_int16f(Y[c], sg), sg);
                }
                const TileTexels tx = { planes, np };
                part = rough_error6<8>(tx, seed, pos);       // >= 0
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { sSeedA[blk][code][c] = seed.A[c]; sSeedB[blk][code][c] = seed.B[c]; }
            sPart[blk][code] = part;
        }
    }
    __syncthreads();
    if (!inRange) return;

    // RoughMSE's total (:2523-2556): fError += error of region 0, then of region 1; regions of one or two texels add nothing
    float rough = 0.0f;
    if (lane < 32)
    {
        const float q0 = sPart[wave][lane * 2], q1 = sPart[wave][lane * 2 + 1];
        if (q0 >= 0.0f) rough += q0;
        if (q1 >= 0.0f) rough += q1;
    }
    int key = (lane < 32) ? __float_as_int(rough) : 0x7FFFFFFF;
    uint32_t shp = uint32_t(lane);
    for (int i = 0; i < 8; ++i) selection_pass(key, shp, lane, i);
    // lane i < 8 now knows the i-th best shape
    if (lane < 8)
    {
        int* sd = a.seeds + uint64_t(nb) * SEED_INTS;
        const uint32_t sh = shp & 31u;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
            {
                sd[(lane * 2 + r) * 6 + c] = sSeedA[wave][sh * 2 + r][c];
                sd[(lane * 2 + r) * 6 + 3 + c] = sSeedB[wave][sh * 2 + r][c];
            }
        a.lists[uint64_t(nb) * 8 + lane] = uint8_t(shp);
    }
}

// The one-region seed (:2513-2521 with uPartitions == 0): the whole block fitted once, one lane per block, texels in registers.
__global__ void __launch_bounds__(256) bc6h_block_seed_kernel(Bc6hArgs a)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nblocks) return;
    const bool sg = a.isSigned != 0;
    const BcSeg& im = seg_of(a.seg, nb);
    const uint32_t gb = im.nb0 + (nb - im.l0);
    const uint32_t by = gb / im.nbw, bx = gb - by * im.nbw;
    const uint32_t x0 = bx * 4, y0 = by * 4;
    const uint32_t pw = min(4u, im.src.width - x0), ph = min(4u, im.src.height - y0);
    float f[64];
#pragma unroll
    for (uint32_t t = 0; t < 16; ++t)
    {
        const uint32_t sx = x0 + replicate_src(t & 3, pw), sy = y0 + replicate_src(t >> 2, ph);
        const Texel px = convert_texel(load_texel(im.src.pixels + uint64_t(sy) * im.src.rowPitch, sx, im.src.format), im.src.tcv, im.src.tsw);
        f[t * 4 + 0] = px.r; f[t * 4 + 1] = px.g; f[t * 4 + 2] = px.b; f[t * 4 + 3] = px.a;
    }
    float X[4], Y[4];
    bc7::seed_fit<false, true>(f, 0xFFFFu, X, Y);
    int* sd = a.seeds + uint64_t(nb) * SEED_INTS;
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        sd[16 * 6 + c] = clamp_seed(float_to_int16f(X[c], sg), sg);
        sd[16 * 6 + 3 + c] = clamp_seed(float_to_int16f(Y[c], sg), sg);
    }
}

// ---- pre / post --------------------------------------------------------------------------------------------------------------
// Lane layout of pre / post: two-region modes use 16 lanes per block (rank * 2 + region), one-region modes 1 lane.
template<int REGIONS2> struct Lay6 { enum : int { TPB = REGIONS2 ? 16 : 1, N = REGIONS2 ? 8 : 16, BPW = 64 / TPB }; };

// ---- exact pruning (same argument as bc7_core.h, subset_lower_bound) ---------------------------------------------------------------
// A BC6H palette entry is FinishUnquantize(((64 - w) A + w B + 32) >> 6) per channel (:1930-1940, :2044-2077): within 1.5 of a
// point on the real line through the scaled unquantised endpoints, so within 1.5 sqrt(3) in distance. Hence for ANY endpoints
//     sum_t |texel_t - palette(t)|^2  >=  ( sqrt(tr S - lambda_max(S)) - 3 sqrt(n) )^2     when the bracket is positive,
// S the 3x3 scatter matrix of the region's texels (INTColor values). lambda_max is bounded from above by ||N^16||_F^(1/16),
// N = S / tr S. A candidate (mode, shape) whose bound exceeds the block's running best can never pass Refine's
// "fOptErr < fBestErr" / "fOrgErr < fBestErr" tests (:2412-2424), so it is not searched. The reference accumulates its errors
// in fp32; the margin below is far wider than that rounding.
__device__ __forceinline__ float region_lower_bound6(const float* planes, uint32_t mask)
{
    double n = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, q00 = 0.0, q01 = 0.0, q02 = 0.0, q11 = 0.0, q12 = 0.0, q22 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if ((mask >> i) & 1u)
        {
            const double x = planes[i], y = planes[16 + i], z = planes[32 + i];       // exact integers
            n += 1.0; s0 += x; s1 += y; s2 += z;
            q00 += x * x; q01 += x * y; q02 += x * z; q11 += y * y; q12 += y * z; q22 += z * z;
        }
    if (n < 2.0) return 0.0f;
    // M = n * S (exact in double: |entries| < 2^53)
    const double M00 = n * q00 - s0 * s0, M01 = n * q01 - s0 * s1, M02 = n * q02 - s0 * s2, M11 = n * q11 - s1 * s1, M12 = n * q12 - s1 * s2, M22 = n * q22 - s2 * s2;
    const double T = M00 + M11 + M22;
    if (!(T > 0.0)) return 0.0f;
    const double inv = 1.0 / T;
    double a00 = M00 * inv, a01 = M01 * inv, a02 = M02 * inv, a11 = M11 * inv, a12 = M12 * inv, a22 = M22 * inv;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const double b00 = a00 * a00 + a01 * a01 + a02 * a02;
        const double b01 = a00 * a01 + a01 * a11 + a02 * a12;
        const double b02 = a00 * a02 + a01 * a12 + a02 * a22;
        const double b11 = a01 * a01 + a11 * a11 + a12 * a12;
        const double b12 = a01 * a02 + a11 * a12 + a12 * a22;
        const double b22 = a02 * a02 + a12 * a12 + a22 * a22;
        a00 = b00; a01 = b01; a02 = b02; a11 = b11; a12 = b12; a22 = b22;
    }
    const double f2 = a00 * a00 + a11 * a11 + a22 * a22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12);
    const double lam = sqrt(sqrt(sqrt(sqrt(sqrt(f2))))) * (1.0 + 1e-9);
    if (lam >= 1.0) return 0.0f;
    const double resid = T * (1.0 - lam) / n;
    const double d = sqrt(resid) - 3.0 * sqrt(n) - 1e-3;
    if (d <= 0.0) return 0.0f;
    return float(d * d * 0.9999);
}

struct Org6
{
    EndPts ep;              // quantised, anchor-fixed endpoints of this lane's region
    EndPts epT;             // the same after TransformForward (what EmitBlock would write)
    float err; uint64_t idx;
    bool fit;               // the whole candidate passes EndPointsFit
    uint32_t shape, mask, anchor; uint64_t pos; int np;
};

// Refine's first half (:2386-2393) for this lane's region; needs the partner lane of the candidate for the transform.
template<int REGIONS2>
__device__ __forceinline__ void org_candidate(const Bc6hArgs& a, uint32_t nb, uint32_t r, const float* planes, float* slot, Org6& o, const OrgSave* saved = nullptr,
                                              const OrgSave* loaded = nullptr, const Best6* bestLoaded = nullptr)
{
    typedef Lay6<REGIONS2> L;
    const bool sg = a.isSigned != 0;
    const uint32_t rank = REGIONS2 ? (r >> 1) : 0u, region = REGIONS2 ? (r & 1u) : 0u;
    o.shape = REGIONS2 ? uint32_t(a.lists[uint64_t(nb) * 8 + rank]) : 0u;
    const uint32_t m1 = REGIONS2 ? uint32_t(kPart2Mask[o.shape]) : 0u;
    o.mask = REGIONS2 ? (region ? m1 : ((~m1) & 0xFFFFu)) : 0xFFFFu;
    o.anchor = (REGIONS2 && region) ? uint32_t(kAnchor2[o.shape]) : 0u;
    const int* sd = a.seeds + uint64_t(nb) * SEED_INTS + (REGIONS2 ? (rank * 2 + region) * 6 : 16 * 6);
    if constexpr (REGIONS2) o.np = region_positions(o.mask, o.pos);            // `planes` is the block's tile in LDS, read in place
    else o.np = gather_texels(planes, o.mask, slot, o.pos);
    if (saved)
    {
        // post: the pre kernel of this mode already quantised the seeds and assigned the indices
        const OrgSave sv = loaded ? *loaded : *saved;
        unpack_ep16(sv.ep, sg, o.ep.A, o.ep.B);
        o.err = sv.err; o.idx = sv.idx;
    }
    else
    {
#pragma unroll
        for (int c = 0; c < 3; ++c)
        {
            o.ep.A[c] = quantize(sd[c], a.mode.prec, sg);
            o.ep.B[c] = quantize(sd[3 + c], a.mode.prec, sg);
        }
        if constexpr (REGIONS2) { const TileTexels tx = { planes, o.np }; o.err = assign_indices6<L::N>(tx, o.pos, o.ep, a.mode.prec, sg, o.anchor, o.idx); }
        else o.err = assign_indices6<L::N>(slot_texels(slot, o.np), o.pos, o.ep, a.mode.prec, sg, o.anchor, o.idx);
    }
    int a0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) a0[c] = REGIONS2 ? __shfl(o.ep.A[c], (threadIdx.x & 63) & ~1) : o.ep.A[c];
    o.epT = a.mode.transformed ? transform_forward(o.ep, int(region), a0) : o.ep;
    bool fit = endpoints_fit(o.epT, int(region), a.mode, sg);
    if (REGIONS2) { const int partner = __shfl_xor(int(fit), 1); fit = fit && (partner != 0); }     // no short-circuit around the shuffle
    // Encode() stops at the first candidate that reaches error 0 (:1823, :1851): nothing later can be strictly better
    // ... from an EARLIER mode of the encoder's order that is: the modes may run in another order here (see launch_bc6h_encode_many), and a
    // zero reached by a later mode does not stop an earlier one, which could reach zero too and would then come first
    const Best6 cur = bestLoaded ? *bestLoaded : a.best[nb];
    o.fit = fit && (cur.err > 0.0f || cur.mode > uint32_t(a.mode.index));
}

template<int REGIONS2>
__global__ void __launch_bounds__(256) bc6h_pre_kernel(Bc6hArgs a)
{
    typedef Lay6<REGIONS2> L;
    __shared__ float sSlot[4][REGIONS2 ? L::BPW * 48 : 48 * 64];        // two regions: the blocks' planes; one region: a column per lane
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbFirst = (blockIdx.x * 4 + wave) * L::BPW;
    if (nbFirst >= a.nblocks) return;
    const uint32_t blk = uint32_t(lane) / L::TPB, r = uint32_t(lane) % L::TPB;
    const uint32_t nb = min(nbFirst + blk, a.nblocks - 1);       // out-of-range lanes shadow the last block (shuffles stay defined)
    const bool inRange = (nbFirst + blk) < a.nblocks;
    float regs[REGIONS2 ? 1 : 48];
    const float* planes = regs;
    float* slot = &sSlot[wave][REGIONS2 ? 0 : lane];
    if constexpr (REGIONS2)
    {
        stage_planes<L::BPW>(a, nbFirst, lane, sSlot[wave]);
        planes = &sSlot[wave][blk * 48];
    }
    else
    {
        const float4* gp = reinterpret_cast<const float4*>(a.fpix + uint64_t(nb) * 48);
#pragma unroll
        for (int i = 0; i < 12; ++i) { const float4 v = gp[i]; regs[4 * i] = v.x; regs[4 * i + 1] = v.y; regs[4 * i + 2] = v.z; regs[4 * i + 3] = v.w; }
    }
    Org6 o;
    const uint64_t t = REGIONS2 ? uint64_t(nb) * L::TPB + r : uint64_t(a.taskBase) + nb;
    // Refine's unoptimised half - quantised seeds, AssignIndices, SwapIndices - depends on the endpoint PRECISION, not on the mode's delta
    // bits: the second and third mode of an 8-bit / 11-bit trio read what the first left in orgs[] instead of assigning indices again
    const bool reuseOrg = REGIONS2 && a.samePrec != 0;
    org_candidate<REGIONS2>(a, nb, r, planes, slot, o, reuseOrg ? a.orgs + t : nullptr);
#if defined(DXTEX_BC6H_TRACE)
    if (nb == 0 && a.mode.index == 0) printf("pre r %u shape %u fit %d A %d %d %d B %d %d %d | T A %d %d %d B %d %d %d err %.9g np %d tr %d delta %d %d %d\n", r, o.shape, int(o.fit), o.ep.A[0], o.ep.A[1], o.ep.A[2], o.ep.B[0], o.ep.B[1], o.ep.B[2], o.epT.A[0], o.epT.A[1], o.epT.A[2], o.epT.B[0], o.epT.B[1], o.epT.B[2], o.err, o.np, a.mode.transformed, a.mode.delta[0], a.mode.delta[1], a.mode.delta[2]);
#endif
    // exact pruning: the candidate's lower bound (over its PROPER regions - what Refine scores) against the running best
    bool prune = false;
    if (a.prune)
    {
        // the bound depends on the block, the shape and the region only - not on the mode: the first two-region launch (and the first
        // one-region launch) of a pass computes it (a fp64 power iteration), the others read it back
        float* bslot = a.bounds + uint64_t(nb) * 17 + (REGIONS2 ? r : 16u);
        float lb;
        if (a.boundsReady) lb = *bslot;
        else { lb = region_lower_bound6(planes, o.mask); if (inRange) *bslot = lb; }
        if (REGIONS2) lb += __shfl_xor(lb, 1);
        // what is already on the table: the running best of the earlier modes and, within this mode, the unoptimised error of
        // every candidate that fits (Refine emits at least that, :2412-2424)
        float table = o.fit ? o.err : 3.0e38f;
        if (REGIONS2)
        {
            const float pe = __shfl_xor(o.err, 1);
            table = o.fit ? o.err + pe : 3.0e38f;
#pragma unroll
            for (int d = 2; d < 16; d <<= 1) table = fminf(table, __shfl_xor(table, d));
        }
        table = fminf(table * 1.00001f, a.best[nb].err);
        prune = lb > table;
    }
    if (!inRange) return;
    if (!reuseOrg)
    {
        OrgSave sv;
        sv.ep = pack_ep16(o.ep.A, o.ep.B);
        sv.err = o.err; sv.idx = o.idx;
        a.orgs[t] = sv;
    }
    // OptimizeOne's search (:2145-2194) depends on the block, the shape, the region and the endpoint PRECISION - not on the mode's delta
    // bits, which only decide what fits (EndPointsFit). Modes 11:5:4:4 / 11:4:5:4 / 11:4:4:5 and 8:6:5:5 / 8:5:6:5 / 8:5:5:6 follow each
    // other in the encoder's order, so a task the previous mode of the same precision already searched (or inherited) keeps the
    // optimised endpoints it left in recs[] and is not searched again; post scores them against this mode's bit budget as Refine would
    // (:2395-2424). On the cfg3 image 98 % of the tasks of the second and third 8-bit mode are of that kind (2 x 13.5 ms of search).
    uint32_t done = 0;
    if (REGIONS2 && a.samePrec)
    {
        const uint32_t prev = a.tinfo[t];
        if ((prev >> 24) != 0u || (prev & kDoneBit6) != 0u) done = kDoneBit6;
    }
    // OptimizeEndPoints' quirk (:2215): region 0 is optimised against all sixteen texels. Nothing to search when the
    // candidate does not fit or its error is already 0 (PerturbOne only accepts strictly smaller errors).
    const bool region0 = !REGIONS2 || (r & 1u) == 0;
    const uint32_t smask = region0 ? 0xFFFFu : o.mask;
    uint32_t snp = (o.fit && o.err > 0.0f) ? uint32_t(region0 ? 16 : o.np) : 0u;
    if (prune) snp = 0;
    // one-region tasks are sorted by mode slot instead (13 + slot: the 16-bit mode, the longest search, goes first); all have 16 texels
    if (!REGIONS2 && snp) snp = 13u + a.taskBase / a.nblocks;
    if (done) snp = 0;
    // The search record is the start of a search: only a task that will be searched gets one (a `done` task keeps the earlier mode's;
    // post takes the unoptimised endpoints of everything else from orgs[]) - most tasks of the later modes write nothing here.
    if (snp)
    {
        Rec6 rec;
        rec.ep = pack_ep16(o.ep.A, o.ep.B);
        rec.err = o.err;
        a.recs[t] = rec;
    }
    a.tinfo[t] = smask | done | (snp << 24);
}

template<int REGIONS2>
__global__ void __launch_bounds__(256) bc6h_post_kernel(Bc6hArgs a)
{
    typedef Lay6<REGIONS2> L;
    __shared__ float sSlot[4][REGIONS2 ? L::BPW * 48 : 48 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbFirst = (blockIdx.x * 4 + wave) * L::BPW;
    if (nbFirst >= a.nblocks) return;
    const bool sg = a.isSigned != 0;
    const uint32_t blk = uint32_t(lane) / L::TPB, r = uint32_t(lane) % L::TPB;
    const uint32_t nb = min(nbFirst + blk, a.nblocks - 1);
    const bool inRange = (nbFirst + blk) < a.nblocks;
    const uint32_t rank = REGIONS2 ? (r >> 1) : 0u, region = REGIONS2 ? (r & 1u) : 0u;
    float regs[REGIONS2 ? 1 : 48];
    const float* planes = regs;
    float* slot = &sSlot[wave][REGIONS2 ? 0 : lane];
    // everything this lane reads of its task, requested before the texels are staged: the kernel spent more than half of its time parked on
    // these loads one after the other (orgs -> best -> tinfo -> recs; profiles/r05_kernels.md)
    const uint64_t t = REGIONS2 ? uint64_t(nb) * L::TPB + r : uint64_t(a.taskBase) + nb;
    const uint32_t ti = a.tinfo[t];
    const OrgSave orgLoaded = a.orgs[t];
    const Rec6 recLoaded = a.recs[t];            // (read for every task: the array is there whether or not a search wrote the slot)
    const Best6 bestLoaded = a.best[nb];
    if constexpr (REGIONS2)
    {
        stage_planes<L::BPW>(a, nbFirst, lane, sSlot[wave]);
        planes = &sSlot[wave][blk * 48];
    }
    else
    {
        const float4* gp = reinterpret_cast<const float4*>(a.fpix + uint64_t(nb) * 48);
#pragma unroll
        for (int i = 0; i < 12; ++i) { const float4 v = gp[i]; regs[4 * i] = v.x; regs[4 * i + 1] = v.y; regs[4 * i + 2] = v.z; regs[4 * i + 3] = v.w; }
    }
    Org6 o;
    org_candidate<REGIONS2>(a, nb, r, planes, slot, o, a.orgs + t, &orgLoaded, &bestLoaded);

    // the optimised endpoints: in recs[] where a search ran for this task (this mode's, or an earlier one's of the same precision), the
    // unoptimised ones everywhere else
    const bool ownSearched = (ti >> 24) != 0u || (ti & kDoneBit6) != 0u;
    EndPts opt = o.ep;
    bool optOverflow = false;            // the search ended with A outside the signed 16-bit range (see Ep16): EndPointsFit fails in the reference
    if (ownSearched)
    {
        const Rec6 rec = recLoaded;
        unpack_ep16(rec.ep, sg, opt.A, opt.B);
        optOverflow = rec.err == -1.0f;
        if (rec.err <= -2.0f)            // components of B above 32767 wrapped in their fields: the true values (ep16_mark)
        {
            const int m = int(-rec.err) - 2;
#pragma unroll
            for (int c = 0; c < 3; ++c) if ((m >> c) & 1) opt.B[c] += 65536;
        }
    }
    // A candidate the search never ran for either region (pruned, does not fit, error already 0: subset size 0 in the task list)
    // still has its unoptimised endpoints: it either cannot win (its lower bound exceeds an error on the table, or it is not
    // encodable) or wins with its unoptimised half (error 0, which nothing beats), so it stands with those numbers. A wavefront
    // whose candidates are all of that kind - most wavefronts of the later modes - skips the second AssignIndices.
    bool searched = ownSearched;
    if (REGIONS2) searched = searched || (__shfl_xor(int(searched), 1) != 0);      // the candidate's other region: Refine scores both (:2401-2410)
    uint64_t optIdx = o.idx;
    float optErr = o.err;
    // AssignIndices (+ SwapIndices) of the optimised endpoints depends on the endpoint PRECISION, not on the mode's delta bits (as pre's does,
    // reuseOrg): the second and third mode of an 8-bit / 11-bit trio inherit 98 % of their searches from the first (kDoneBit6), and a
    // candidate neither of whose regions was searched anew in THIS mode has the endpoints the previous member's post scored - it reads back
    // what that post assigned (error, indices, whether the endpoints were swapped) instead of assigning again.
    const bool newly = (ti >> 24) != 0u;
    bool candNewly = newly;
    if (REGIONS2) candNewly = candNewly || (__shfl_xor(int(newly), 1) != 0);
    const bool inherited = REGIONS2 && a.samePrec != 0 && searched && !candNewly;
    bool swapped = false;
    if (inherited)
    {
        const OptSave sv = a.opts[t];
        optErr = sv.err; optIdx = sv.idx; swapped = sv.swapped != 0u;
        if (swapped) { for (int c = 0; c < 3; ++c) { const int tmp = opt.A[c]; opt.A[c] = opt.B[c]; opt.B[c] = tmp; } }
    }
    const bool assign = searched && !inherited;
    if (__any(assign))
    {
        uint64_t ix;
        float e;
        EndPts tmpEp = opt;
        if constexpr (REGIONS2) { const TileTexels tx = { planes, o.np }; e = assign_indices6<L::N>(tx, o.pos, tmpEp, a.mode.prec, sg, o.anchor, ix); }
        else e = assign_indices6<L::N>(slot_texels(slot, o.np), o.pos, tmpEp, a.mode.prec, sg, o.anchor, ix);
        if (assign)
        {
            swapped = (tmpEp.A[0] != opt.A[0]) || (tmpEp.A[1] != opt.A[1]) || (tmpEp.A[2] != opt.A[2]);      // (A == B: a swap changes nothing)
            optErr = e; optIdx = ix; opt = tmpEp;
        }
    }
    if (REGIONS2 && a.saveOpt != 0 && searched && inRange) { OptSave sv; sv.idx = optIdx; sv.err = optErr; sv.swapped = swapped ? 1u : 0u; a.opts[t] = sv; }
    if (!searched) opt = o.ep;
    float orgTot = 0.0f + o.err, optTot = 0.0f + optErr;
    if (REGIONS2)
    {
        // fTot += aErr[0]; fTot += aErr[1] (:2401-2406): region 0 first
        const float oe = __shfl_xor(o.err, 1), pe = __shfl_xor(optErr, 1);
        orgTot = region ? (0.0f + oe) + o.err : (0.0f + o.err) + oe;
        optTot = region ? (0.0f + pe) + optErr : (0.0f + optErr) + pe;
    }
    int b0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) b0[c] = REGIONS2 ? __shfl(opt.A[c], lane & ~1) : opt.A[c];
    const EndPts optT = a.mode.transformed ? transform_forward(opt, int(region), b0) : opt;
    bool fitOpt = endpoints_fit(optT, int(region), a.mode, sg) && !optOverflow;
    if (REGIONS2) { const int partner = __shfl_xor(int(fitOpt), 1); fitOpt = fitOpt && (partner != 0); }
    const bool useOpt = fitOpt && (optTot < orgTot);
    const float err = useOpt ? optTot : orgTot;
    const EndPts fin = useOpt ? optT : o.epT;
    const uint64_t myIdx = useOpt ? optIdx : o.idx;

    // first minimum over the block's candidates, in rank order
    const bool valid = o.fit && inRange;
    uint64_t key = valid ? ((uint64_t(uint32_t(__float_as_int(err))) << 8) | rank) : ~0ull;
    uint64_t bestKey = key;
    if (REGIONS2)
    {
#pragma unroll
        for (int d = 2; d < 16; d <<= 1)
        {
            const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(bestKey)), d)), hi = uint32_t(__shfl_xor(int(uint32_t(bestKey >> 32)), d));
            const uint64_t other = (uint64_t(hi) << 32) | lo;
            bestKey = other < bestKey ? other : bestKey;
        }
    }
    // region 1's endpoints and indices travel to the candidate's region-0 lane
    int ep[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        ep[0][c] = fin.A[c]; ep[1][c] = fin.B[c];
        ep[2][c] = REGIONS2 ? __shfl_down(fin.A[c], 1) : 0;
        ep[3][c] = REGIONS2 ? __shfl_down(fin.B[c], 1) : 0;
    }
    uint64_t idx = myIdx;
    if (REGIONS2)
        idx |= uint64_t(uint32_t(__shfl_down(int(uint32_t(myIdx)), 1))) | (uint64_t(uint32_t(__shfl_down(int(uint32_t(myIdx >> 32)), 1))) << 32);
    if (valid && region == 0 && key == bestKey)
    {
        Best6* b = a.best + nb;
        // Refine only emits when it beats fBestErr (:2412-2424): the block ends up with the FIRST minimum in the encoder's mode order. The modes
        // of a pass run one after the other on one stream, in any order: ties go to the mode that comes first in the encoder's order.
        if (err < bestLoaded.err || (err == bestLoaded.err && uint32_t(a.mode.index) < bestLoaded.mode))      // (only this lane writes the block's slot in this launch)
        {
            Best6 n; n.err = err; n.mode = uint32_t(a.mode.index);
            emit_block6(a.mode, o.shape, ep, idx, REGIONS2 ? uint32_t(kAnchor2[o.shape]) : 0u, n.lo, n.hi);
            *b = n;
        }
    }
}

// ---- perturb ------------------------------------------------------------------------------------------------------------------
template<int N>
__global__ void __launch_bounds__(64) bc6h_perturb_kernel(Bc6hArgs a, uint32_t waveMax)
{
    __shared__ float sSlot[48 * 64];
    const int lane = threadIdx.x;
    const bool sg = a.isSigned != 0;
    const uint32_t live = a.counters[34];
    if (live == 0) return;
    if (N == 16 && live <= waveMax) return;        // short one-region lists: bc6h_perturb_wave_kernel's turn
    uint32_t* head = a.counters + kQueueBase;
    float* slot = &sSlot[lane];
    const int TPB = (N == 8) ? 16 : 1;

    EndPts zero; for (int c = 0; c < 3; ++c) { zero.A[c] = 0; zero.B[c] = 0; }
    Perturb6 st = perturb6_begin(zero, 0.0f);
    Texels tx = slot_texels(slot, 0);
    uint32_t myTask = 0xFFFFFFFFu;
    int prec = a.mode.prec;
    WaveQueue q; q.lo = q.hi = 0; q.drained = false;
    for (;;)
    {
        const unsigned long long idle = __ballot(myTask == 0xFFFFFFFFu);
        if (idle && !(q.drained && q.lo >= q.hi))
        {
            const uint32_t idx = queue_take(q, head, live, idle, lane);
            if (idx != 0xFFFFFFFFu)
            {
                const uint2 task = a.order[idx];
                myTask = task.x;
                const Rec6 rec = a.recs[myTask];
                const uint32_t nb = (N == 8) ? myTask / uint32_t(TPB) : myTask % a.nblocks;
                prec = (N == 8) ? a.mode.prec : a.prec1[myTask / a.nblocks];
                float planes[48];
                const float4* gp = reinterpret_cast<const float4*>(a.fpix + uint64_t(nb) * 48);
#pragma unroll
                for (int i = 0; i < 12; ++i) { const float4 v = gp[i]; planes[4 * i] = v.x; planes[4 * i + 1] = v.y; planes[4 * i + 2] = v.z; planes[4 * i + 3] = v.w; }
                uint64_t pos;
                tx.np = gather_texels(planes, task.y & 0xFFFFu, slot, pos);
                EndPts e;
                unpack_ep16(rec.ep, sg, e.A, e.B);
                st = perturb6_begin(e, rec.err);
            }
        }
        if (__ballot(myTask != 0xFFFFFFFFu) == 0ull)
        {
            if (q.drained && q.lo >= q.hi) break;
            continue;
        }
        if (myTask != 0xFFFFFFFFu)
        {
            float e; int v;
            perturb6_macro<N>(tx, st, prec, sg, e, v);
            st = perturb6_transition(st, e, v);
            if (st.ch >= 3)
            {
                a.recs[myTask].ep = pack_ep16(st.ep.A, st.ep.B);
                { const float mk = ep16_mark(sg, st.ep.A, st.ep.B); if (mk < 0.0f) a.recs[myTask].err = mk; }
                myTask = 0xFFFFFFFFu;
            }
        }
    }
}

// ---- perturb of the two-region modes through a bound filter (round 4) ------------------------------------------------------------
// PerturbOne (:2081-2141) accepts a candidate only when its error is BELOW the best so far. perturb6_bound_pair (bc6h_core.h) bounds the
// two candidates of a step from below at well under half the cost of evaluating them (three FMAs and a maximum per (texel, entry)
// instead of eight ordered fp32 operations and a compare / select pair; the rounding of the reference's fp32 sums and of the bound's own
// arithmetic is covered by an explicit margin). On the cfg3 image 2.6 % of the candidates pass it (none at the large steps, 8 % at step
// 1; DXTEX_BC6H_STATS in the development build, tools/bc6h_debug.cpp -DDXTEX_COUNT_EVALS6 on the host). Those are evaluated exactly -
// operation for operation as MapColorsQuantized does - by the WHOLE wavefront: the owners put the candidates' palettes on a list in LDS,
// sixteen lanes take one list entry, lane k scores texel k (norm3 + scan_min, the functions of the plain kernel), and the per-texel errors
// are summed in texel order for the group's last lane (adds with DPP row_shr sources: the reference's fTotErr += fBestErr, :2074).
constexpr int kFilterSlots = 32;          // list entries per round (a step with more passing candidates takes several rounds)
// Texel k of the lane's region sits kColStride6 16-bit words after texel k - 1: 66 (33 dwords), not 64, so that the sixteen lanes of an exact
// round - same owner column, texels 0 ... 15 - read sixteen different banks (at 64 they all hit one; the bound's loops, where a lane reads
// its own column, are conflict-free either way)
#if !defined(DXTEX_F6_STRIDE)
#define DXTEX_F6_STRIDE 66
#endif
constexpr int kColStride6 = DXTEX_F6_STRIDE;
struct FilterLds
{
    float4 pal[kFilterSlots][6];          // r[8], g[8], b[8] of a passing candidate
    float tot[kFilterSlots];
    uint32_t meta[kFilterSlots];          // owner lane | np << 8
};

// lane l of a row of sixteen gets lane l - D's value, 0.0f where that lane is outside the row (v_mov_b32_dpp row_shr:D, bound_ctrl:0; the
// compiler folds it into the v_add_f32 that consumes it)
template<int D>
__device__ __forceinline__ float row_shr(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xF, 0xF, true));
}
// fTotErr += fBestErr, texel by texel (:2074), for the LAST lane of a group of W lanes (W = 8: half rows, 16: rows): ((e_0 + e_1) + e_2) + ...
// with e_j pulled from W - 1 - j lanes below - one add with a DPP source per texel, the reference's order. The other lanes compute sums of
// shifted windows nobody reads. (Round 4 passed the partial sum along the row: a move, an add and a select per texel.)
template<int W>
__device__ __forceinline__ float row_ordered_sum(float err)
{
    float S = row_shr<W - 1>(err);
    if constexpr (W > 2) S = S + row_shr<W - 2>(err);
    if constexpr (W > 3) S = S + row_shr<W - 3>(err);
    if constexpr (W > 4) S = S + row_shr<W - 4>(err);
    if constexpr (W > 5) S = S + row_shr<W - 5>(err);
    if constexpr (W > 6) S = S + row_shr<W - 6>(err);
    if constexpr (W > 7) S = S + row_shr<W - 7>(err);
    if constexpr (W > 8) S = S + row_shr<W - 8>(err);
    if constexpr (W > 9) S = S + row_shr<W - 9>(err);
    if constexpr (W > 10) S = S + row_shr<W - 10>(err);
    if constexpr (W > 11) S = S + row_shr<W - 11>(err);
    if constexpr (W > 12) S = S + row_shr<W - 12>(err);
    if constexpr (W > 13) S = S + row_shr<W - 13>(err);
    if constexpr (W > 14) S = S + row_shr<W - 14>(err);
    if constexpr (W > 15) S = S + row_shr<W - 15>(err);
    return S + err;
}

template<class WRITE>
__device__ __forceinline__ void exact_by_wave(const int16_t* cols, FilterLds& L, int lane, int np, bool pass0, bool pass1, WRITE writePal, float& e0, float& e1, uint32_t* stats)
{
    const unsigned long long b0 = __ballot(pass0), b1 = __ballot(pass1);
    if ((b0 | b1) == 0ull) return;                                            // wave-uniform: most steps end here
#if defined(DXTEX_DEV)
    if (stats && lane == 0) { atomicAdd(stats + 1, 1u); atomicAdd(stats + 2, uint32_t(__popcll(b0) + __popcll(b1))); }
    const long long t0 = stats ? clock64() : 0;
    long long tList = 0, tRounds = 0;
#endif
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t n0 = uint32_t(__popcll(b0)), total = n0 + uint32_t(__popcll(b1));
    const uint32_t s0 = uint32_t(__popcll(b0 & below)), s1 = n0 + uint32_t(__popcll(b1 & below));
    const int grp = lane >> 4, k = lane & 15;
    // Regions of at most eight texels (the smaller region of most shapes; the task list is sorted by size, so a wavefront's regions are nearly
    // all of one size) take HALF a row of sixteen lanes each: eight list entries per round instead of four, and a sum chain of seven steps
    // instead of fifteen. Wave-uniform per call; the arithmetic per texel and the order of the sum are the same.
#if defined(DXTEX_F6_NO_PACK)
    const bool packed = false;
#else
    const bool packed = __ballot((pass0 || pass1) && np > 8) == 0ull;
#endif
    const int kk = packed ? (k & 7) : k;
    const uint32_t perRound = packed ? 8u : 4u;
    const uint32_t slotInRound = packed ? uint32_t(grp * 2 + (k >> 3)) : uint32_t(grp);
    for (uint32_t first = 0; first < total; first += kFilterSlots)
    {
        const uint32_t cnt = min(total - first, uint32_t(kFilterSlots));
        const bool in0 = pass0 && (s0 - first) < cnt, in1 = pass1 && (s1 - first) < cnt;          // unsigned: s < first wraps above cnt
        // ONE pass of the palette derivation (~250 instructions, run by the few owners) serves both candidates of a step: a lane writes the one it
        // has; the second pass runs only when some lane has both on the list (the kernel is bound by instruction issue: what counts is how
        // often the derivation is issued, not how many lanes take part)
        {
            const bool one = in0 || in1, both = in0 && in1;
            const uint32_t sl = (in0 ? s0 : s1) - first;
            if (one) { writePal(in0 ? 0 : 1, reinterpret_cast<float*>(L.pal[sl])); L.meta[sl] = uint32_t(lane) | (uint32_t(np) << 8); }
            if (__ballot(both) != 0ull)
            {
                if (both) { writePal(1, reinterpret_cast<float*>(L.pal[s1 - first])); L.meta[s1 - first] = uint32_t(lane) | (uint32_t(np) << 8); }
            }
        }
#if defined(DXTEX_DEV)
        const long long tw0 = stats ? clock64() : 0;
#endif
        __syncthreads();                  // one wavefront per workgroup: orders the LDS traffic, costs no barrier
#if defined(DXTEX_DEV)
        const long long tw1 = stats ? clock64() : 0;
        tList += tw1 - (first ? tw0 : t0);
#endif
        for (uint32_t g = 0; g < cnt; g += perRound)
        {
#if defined(DXTEX_DEV)
            if (stats && lane == 0) atomicAdd(stats + 3, 1u);
#endif
            const uint32_t sl = g + slotInRound;
            const uint32_t meta = (sl < cnt) ? L.meta[sl] : 0u;
            const int owner = int(meta & 63u), onp = int(meta >> 8);
            float err = 0.0f;
#if defined(DXTEX_DEV)
            if (stats) { const unsigned long long used = __ballot(kk < onp); if (lane == 0) atomicAdd(stats + 6, uint32_t(__popcll(used))); }
#endif
            if (kk < onp)
            {
                const int16_t* t = cols + owner + kk * kColStride6;
                const float tr = float(t[0]), tg = float(t[16 * kColStride6]), tb = float(t[32 * kColStride6]);
                const float4 r0 = L.pal[sl][0], r1 = L.pal[sl][1], g0 = L.pal[sl][2], g1 = L.pal[sl][3], c0 = L.pal[sl][4], c1 = L.pal[sl][5];
                const float pr[8] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w }, pg[8] = { g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w },
                            pb[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
                float e[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) e[i] = norm3(tr, tg, tb, pr[i], pg[i], pb[i]);
                err = scan_min(e);
            }
            // fTotErr += fBestErr, texel by texel (:2074), in the reference's order; texels past the region's end add +0.0f, which changes
            // nothing, so the group's last lane ends with the total (row_ordered_sum; half rows: lane 7 / 15 of a row pull from their own half)
            const float S = packed ? row_ordered_sum<8>(err) : row_ordered_sum<16>(err);
            if (kk == (packed ? 7 : 15) && sl < cnt) L.tot[sl] = S;
        }
#if defined(DXTEX_DEV)
        if (stats) tRounds += clock64() - tw1;
#endif
        __syncthreads();
        if (in0) e0 = L.tot[s0 - first];
        if (in1) e1 = L.tot[s1 - first];
        __syncthreads();
    }
#if defined(DXTEX_DEV)
    if (stats && lane == 0)
    {
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + 8), (unsigned long long)(clock64() - t0));
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + 14), (unsigned long long)tList);
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + 16), (unsigned long long)tRounds);
    }
#endif
}

// One PerturbOne call for every lane of the wavefront (lanes without a task take part in the exact evaluations only): perturb6_macro with
// the evaluations replaced by bound + shared exact evaluation. Same decisions: a candidate the bound excludes has an error >= the bound >=
// the best so far and `<` would refuse it; the +step candidate is bounded against the best error before the -step candidate's result (a
// superset passes) and accepted against the one after it, as in the reference's loop.
template<bool SG>
__device__ __forceinline__ void perturb6_macro_filter(const int16_t* cols, FilterLds& L, int lane, bool active, const Texels16& tx, const Bound6& bd, const PkState& pk,
                                                      int prec, float& outErr, int& outVal, uint32_t* stats)
{
    constexpr int N = 8;
    const Perturb6 s = pk_unpack(pk, SG);          // short-lived: only ch / do_b (and what is derived below) live across the step loop
#if defined(DXTEX_DEV)
    if (stats) { const unsigned long long act = __ballot(active); if (lane == 0) { atomicAdd(stats + 4, 1u); atomicAdd(stats + 5, uint32_t(__popcll(act))); } }
    const long long tm0 = stats ? clock64() : 0;
    long long tBound = 0;
#endif
    const int fixedQ = (s.ch == 0) ? (s.do_b ? s.ep.A[0] : s.ep.B[0]) : (s.ch == 1) ? (s.do_b ? s.ep.A[1] : s.ep.B[1]) : (s.do_b ? s.ep.A[2] : s.ep.B[2]);
    int cur = (s.ch == 0) ? (s.do_b ? s.ep.B[0] : s.ep.A[0]) : (s.ch == 1) ? (s.do_b ? s.ep.B[1] : s.ep.A[1]) : (s.do_b ? s.ep.B[2] : s.ep.A[2]);
    const int uFixed = unquantize(fixedQ, prec, SG);                 // the other endpoint does not move during this call
    // the palettes of the two channels this call does not walk: (ch + 1) % 3 and (ch + 2) % 3 (the endpoints are selected, not the palettes:
    // three palettes indexed by the channel would live in scratch)
    float fix1[N], fix2[N];
    palette_channel<N>((s.ch == 0) ? s.ep.A[1] : (s.ch == 1) ? s.ep.A[2] : s.ep.A[0], (s.ch == 0) ? s.ep.B[1] : (s.ch == 1) ? s.ep.B[2] : s.ep.B[0], prec, SG, fix1);
    palette_channel<N>((s.ch == 0) ? s.ep.A[2] : (s.ch == 1) ? s.ep.A[0] : s.ep.A[1], (s.ch == 0) ? s.ep.B[2] : (s.ch == 1) ? s.ep.B[0] : s.ep.B[1], prec, SG, fix2);
    const MacroBound6<N, int16_t> mb = bound6_macro<N>(tx, bd, s.ch, fix1, fix2);
    float minErr = s.err;
    // the walked channel's palette of a candidate (GeneratePaletteQuantized with the candidate's endpoint)
    auto var_of = [&](int tmp, float (&var)[N])
    {
        const int uT = unquantize(tmp, prec, SG);
        palette_channel_unq<N>(s.do_b ? uFixed : uT, s.do_b ? uT : uFixed, SG, var);
    };
    auto write_pal = [&](int tmp, float* dst)        // the rare path: all three channels' palettes are derived again rather than kept across the bound's loop
    {
        // the endpoints from the packed words, opaque to the optimiser: it must not keep the six unpacked values alive from the top of the call
        Ep16 e = pk.ep;
        asm volatile("" : "+v"(e.a01), "+v"(e.a2b0), "+v"(e.b12));
        int A[3], B[3];
        unpack_ep16(e, SG, A, B);
#pragma unroll
        for (int c = 0; c < 3; ++c)
        {
            const int qa = (c == s.ch && !s.do_b) ? tmp : A[c], qb = (c == s.ch && s.do_b) ? tmp : B[c];
            float pal[N];
            palette_channel<N>(qa, qb, prec, SG, pal);
#pragma unroll
            for (int i = 0; i < N; ++i) dst[8 * c + i] = pal[i];
        }
    };
    {
        const int half = 1 << (prec - 1);
        const int tmp = (cur >= half) ? cur - half : cur + half;
        const bool valid = (tmp >= 0) && (tmp < (1 << prec));
        float var[N];
        var_of(tmp, var);
        const bool pass = active && valid && perturb6_bound<N>(tx, bd, mb, var, minErr) < minErr;
        float e = 0.0f, none = 0.0f;
        exact_by_wave(cols, L, lane, tx.np, pass, false, [&](int, float* dst) { write_pal(tmp, dst); }, e, none, stats);
        if (pass && e < minErr) { minErr = e; cur = tmp; }
    }
#pragma unroll 1
    for (int step = 1 << (prec - 1) >> 1; step; step >>= 1)
    {
        const int tM = cur - step, tP = cur + step;
#if defined(DXTEX_DEV)
        const long long tb0 = stats ? clock64() : 0;
#endif
        float varM[N], varP[N];
        var_of(tM, varM);
        var_of(tP, varP);
        float lbM, lbP;
        perturb6_bound_pair<N>(tx, bd, mb, varM, varP, minErr, lbM, lbP);
#if defined(DXTEX_DEV)
        if (stats)
        {
            // lanes that still walk texels when the pair's loop ends are not known here: count the lanes with a task, weighted by their share of the longest region
            int npMax = tx.np;
            for (int d = 32; d; d >>= 1) npMax = max(npMax, __shfl_xor(npMax, d));
            int share = npMax ? (tx.np * 64) / npMax : 0;          // 64ths of the wave's loop this lane is busy
            for (int d = 32; d; d >>= 1) share += __shfl_xor(share, d);
            if (lane == 0) atomicAdd(stats + 7, uint32_t(share >> 6));
            tBound += clock64() - tb0;
        }
#endif
        const bool passM = active && (tM >= 0) && (tM < (1 << prec)) && lbM < minErr;
        const bool passP = active && (tP >= 0) && (tP < (1 << prec)) && lbP < minErr;
        float eM = 0.0f, eP = 0.0f;
#if defined(DXTEX_DEV)
        if (stats && lane == 0) atomicAdd(stats + 0, 1u);
#endif
        exact_by_wave(cols, L, lane, tx.np, passM, passP, [&](int which, float* dst) { write_pal(which ? tP : tM, dst); }, eM, eP, stats);
        int beststep = 0;
        if (passM && eM < minErr) { minErr = eM; beststep = -step; }
        if (passP && eP < minErr) { minErr = eP; beststep = step; }
        cur += beststep;
    }
    outErr = minErr; outVal = cur;
#if defined(DXTEX_DEV)
    if (stats && lane == 0)
    {
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + 12), (unsigned long long)tBound);
        atomicAdd(reinterpret_cast<unsigned long long*>(stats + 20), (unsigned long long)(clock64() - tm0));
    }
#endif
}

#if !defined(DXTEX_F6_WAVES)
#define DXTEX_F6_WAVES 4
#endif
#if !defined(DXTEX_F6_PICKUP_MIN)
#define DXTEX_F6_PICKUP_MIN 1
#endif
constexpr int kPickupMin6 = DXTEX_F6_PICKUP_MIN;
template<bool SG>
__global__ void __launch_bounds__(64, DXTEX_F6_WAVES) bc6h_perturb_filter_kernel(Bc6hArgs a)
{
    __shared__ int16_t sCols[48 * kColStride6];        // 6.2 KiB + the list: sixteen wavefronts per CU
    __shared__ FilterLds sList;
    const int lane = threadIdx.x;
    const uint32_t live = a.counters[34];
    if (live == 0) return;
    uint32_t* head = a.counters + kQueueBase;
    int16_t* slot = &sCols[lane];
    EndPts zero; for (int c = 0; c < 3; ++c) { zero.A[c] = 0; zero.B[c] = 0; }
    PkState st = pk_pack(perturb6_begin(zero, 0.0f));
    Texels16 tx; tx.r = slot; tx.g = slot + 16 * kColStride6; tx.b = slot + 32 * kColStride6; tx.stride = kColStride6; tx.np = 0;
    Bound6 bd; bd.o[0] = bd.o[1] = bd.o[2] = 0.0f; bd.pp = 0.0f; bd.pre[0] = bd.pre[1] = bd.pre[2] = 0.0f;
    uint32_t myTask = 0xFFFFFFFFu;
    const int prec = a.mode.prec;
#if defined(DXTEX_DEV)
    uint32_t* stats = a.filterStats ? a.counters + 64 : nullptr;      // see the table printed by the launcher (DXTEX_BC6H_STATS)
#else
    uint32_t* stats = nullptr;
#endif
    WaveQueue q; q.lo = q.hi = 0; q.drained = false;
#if defined(DXTEX_DEV)
    const long long tk0 = stats ? clock64() : 0;
#endif
    for (;;)
    {
        const unsigned long long idle = __ballot(myTask == 0xFFFFFFFFu);
#if defined(DXTEX_DEV)
        const long long tp0 = stats ? clock64() : 0;
#endif
        // (kPickupMin6 > 1 would let idle lanes wait until several can take tasks together, as the BC7 kernels do; measured 1 / 4 / 8 / 16 / 32:
        // 56.0 / 56.0 / 56.2 / 56.8 / 59.7 ms - the kernel is bound by instruction issue, a pick-up's latency is hidden by the other waves, and
        // waiting lanes are lost work: 1 stays)
        if (idle && !(q.drained && q.lo >= q.hi) && (__popcll(idle) >= kPickupMin6 || idle == ~0ull))
        {
            const uint32_t idx = queue_take(q, head, live, idle, lane);
            if (idx != 0xFFFFFFFFu)
            {
                const uint2 task = a.order[idx];
                myTask = task.x;
                const Rec6 rec = a.recs[myTask];
                // the block's 48 texel values (r[16], g[16], b[16], floats holding 16-bit integers): twelve loads in flight at once, then the region's
                // texels into the lane's columns with static register indices (before: a load per texel and channel inside the loop over the mask,
                // i.e. np dependent trips to memory)
                const float4* gp = reinterpret_cast<const float4*>(a.fpix + uint64_t(myTask / 16u) * 48);
                float4 v[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = gp[i];
                const uint32_t mask = task.y & 0xFFFFu;
                int np = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                {
                    if (mask & (1u << i))
                    {
                        const float4 r4 = v[i >> 2], g4 = v[4 + (i >> 2)], b4 = v[8 + (i >> 2)];
                        const float fr = (i & 3) == 0 ? r4.x : (i & 3) == 1 ? r4.y : (i & 3) == 2 ? r4.z : r4.w;
                        const float fg = (i & 3) == 0 ? g4.x : (i & 3) == 1 ? g4.y : (i & 3) == 2 ? g4.z : g4.w;
                        const float fb = (i & 3) == 0 ? b4.x : (i & 3) == 1 ? b4.y : (i & 3) == 2 ? b4.z : b4.w;
                        slot[np * kColStride6] = int16_t(int(fr)); slot[(16 + np) * kColStride6] = int16_t(int(fg)); slot[(32 + np) * kColStride6] = int16_t(int(fb));
                        ++np;
                    }
                }
                tx.np = np;
                bd = bound6_begin(tx);
                EndPts e;
                unpack_ep16(rec.ep, SG, e.A, e.B);
                st = pk_pack(perturb6_begin(e, rec.err));
            }
        }
#if defined(DXTEX_DEV)
        if (stats && lane == 0 && idle) atomicAdd(reinterpret_cast<unsigned long long*>(stats + 18), (unsigned long long)(clock64() - tp0));
#endif
        const bool active = myTask != 0xFFFFFFFFu;
        if (__ballot(active) == 0ull)
        {
            if (q.drained && q.lo >= q.hi) break;
            continue;
        }
        if (!active) tx.np = 0;
        float e; int v;
        perturb6_macro_filter<SG>(sCols, sList, lane, active, tx, bd, st, prec, e, v, stats);
        if (active)
        {
            const Perturb6 nx = perturb6_transition(pk_unpack(st, SG), e, v);
            st = pk_pack(nx);
            if (nx.ch >= 3)
            {
                a.recs[myTask].ep = st.ep;
                { const float mk = ep16_mark(SG, nx.ep.A, nx.ep.B); if (mk < 0.0f) a.recs[myTask].err = mk; }
                myTask = 0xFFFFFFFFu;
            }
        }
    }
#if defined(DXTEX_DEV)
    if (stats && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(stats + 10), (unsigned long long)(clock64() - tk0));
#endif
}

// The one-region modes have one task per block, and after pruning few of them are left (hundreds to a few thousand in a 4096^2
// image) - but each is long: a PerturbOne call of the 16-bit mode walks 32 candidates of 16 texels x 16 palette entries, and the
// alternating loop repeats it as long as it improves. With a lane per task the kernel takes as long as its longest chain
// (8 ms per 4096^2 image, nearly all of the machine idle). Below kWaveTaskMax6 live tasks a WAVEFRONT owns a task instead: lane
// (half, k) evaluates texel k against candidate `cur - step` (half 0) or `cur + step` (half 1), the per-texel errors are summed
// in texel order as MapColorsQuantized does (:2044-2077), and every lane takes the same decisions from the two totals. Same
// functions, same operation order, 1/13 of the chain. Lanes 32..63 mirror lanes 0..31.
constexpr uint32_t kWaveTaskMax6 = 65536;

__device__ __forceinline__ void perturb6_wave_macro(float tr, float tg, float tb, int lane, const Perturb6& s, int prec, bool isSigned, float& outErr, int& outVal)
{
    constexpr int N = 16;
    float base[3][N];
#pragma unroll
    for (int c = 0; c < 3; ++c) palette_channel<N>(s.ep.A[c], s.ep.B[c], prec, isSigned, base[c]);
    const int fixedQ = (s.ch == 0) ? (s.do_b ? s.ep.A[0] : s.ep.B[0]) : (s.ch == 1) ? (s.do_b ? s.ep.A[1] : s.ep.B[1]) : (s.do_b ? s.ep.A[2] : s.ep.B[2]);
    int cur = (s.ch == 0) ? (s.do_b ? s.ep.B[0] : s.ep.A[0]) : (s.ch == 1) ? (s.do_b ? s.ep.B[1] : s.ep.A[1]) : (s.do_b ? s.ep.B[2] : s.ep.A[2]);
    const int half = (lane >> 4) & 1, group = lane & 48;
    float minErr = s.err;
#pragma unroll 1
    for (int step = 1 << (prec - 1); step; step >>= 1)
    {
        const int tmp = cur + (half ? step : -step);
        float var[N];
        palette_channel<N>(s.do_b ? fixedQ : tmp, s.do_b ? tmp : fixedQ, prec, isSigned, var);
        float e[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            e[i] = norm3(tr, tg, tb, (s.ch == 0) ? var[i] : base[0][i], (s.ch == 1) ? var[i] : base[1][i], (s.ch == 2) ? var[i] : base[2][i]);
        const float te = scan_min(e);
        float tot = 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += __shfl(te, group | j);
        const float eMinus = __shfl(tot, 0), ePlus = __shfl(tot, 16);
        const int tMinus = cur - step, tPlus = cur + step;
        int beststep = 0;
        if (tMinus >= 0 && tMinus < (1 << prec) && eMinus < minErr) { minErr = eMinus; beststep = -step; }
        if (tPlus >= 0 && tPlus < (1 << prec) && ePlus < minErr) { minErr = ePlus; beststep = step; }
        cur += beststep;
    }
    outErr = minErr; outVal = cur;
}

__global__ void __launch_bounds__(64) bc6h_perturb_wave_kernel(Bc6hArgs a, uint32_t waveMax)
{
    const int lane = threadIdx.x;
    const bool sg = a.isSigned != 0;
    const uint32_t live = a.counters[34];
    if (live == 0 || live > waveMax) return;
    uint32_t* head = a.counters + kQueueBase;
    for (;;)
    {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(head, 1u);
        idx = uint32_t(__builtin_amdgcn_readfirstlane(int(idx)));
        if (idx >= live) break;
        const uint32_t myTask = a.order[idx].x;
        const Rec6 rec = a.recs[myTask];
        const uint32_t nb = myTask % a.nblocks;
        const int prec = a.prec1[myTask / a.nblocks];
        const float* gp = a.fpix + uint64_t(nb) * 48 + (lane & 15);
        const float tr = gp[0], tg = gp[16], tb = gp[32];
        EndPts e;
        unpack_ep16(rec.ep, sg, e.A, e.B);
        Perturb6 st = perturb6_begin(e, rec.err);
        while (st.ch < 3)
        {
            float err; int v;
            perturb6_wave_macro(tr, tg, tb, lane, st, prec, sg, err, v);
            st = perturb6_transition(st, err, v);
        }
        if (lane == 0)
        {
            a.recs[myTask].ep = pack_ep16(st.ep.A, st.ep.B);
            { const float mk = ep16_mark(sg, st.ep.A, st.ep.B); if (mk < 0.0f) a.recs[myTask].err = mk; }
        }
    }
}

__global__ void __launch_bounds__(256) bc6h_store_kernel(Bc6hArgs a)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nblocks) return;
    const Best6 b = a.best[nb];
    const BcSeg& im = seg_of(a.seg, nb);
    const uint32_t gb = im.nb0 + (nb - im.l0);
    const uint32_t by = gb / im.nbw, bx = gb - by * im.nbw;
    uint64_t* out = reinterpret_cast<uint64_t*>(im.dst + uint64_t(by) * im.dstRowPitch) + 2 * uint64_t(bx);
    out[0] = b.lo; out[1] = b.hi;
}

const uint64_t kMaxBlocksPerPass6 = dev_env("DXTEX_MAX_BLOCKS_PER_PASS") ? std::max<uint64_t>(1, strtoull(dev_env("DXTEX_MAX_BLOCKS_PER_PASS"), nullptr, 10)) : (1u << 22);
struct Scratch6
{
    size_t fpix, lists, seeds, recs, orgs, order, tinfo, counters, best, bounds, recs1, orgs1, order1, tinfo1, counters1, opts, total;
    explicit Scratch6(uint64_t nb)
    {
        auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
        size_t o = 0;
        fpix = o; o = up(o + nb * 48 * sizeof(float));
        lists = o; o = up(o + nb * 8);
        seeds = o; o = up(o + nb * SEED_INTS * sizeof(int));
        recs = o; o = up(o + nb * 16 * sizeof(Rec6));
        orgs = o; o = up(o + nb * 16 * sizeof(OrgSave));
        order = o; o = up(o + nb * 16 * sizeof(uint2));
        tinfo = o; o = up(o + nb * 16 * sizeof(uint32_t));
        counters = o; o = up(o + 128 * sizeof(uint32_t));       // [0, 64): bins, live count, queue heads; [64, 128): the development build's statistics
        best = o; o = up(o + nb * sizeof(Best6));
        bounds = o; o = up(o + nb * 17 * sizeof(float));
        // the one-region section's own task arrays (four mode slots of nb tasks): it runs on a side stream next to the last two-region mode
        recs1 = o; o = up(o + nb * 4 * sizeof(Rec6));
        orgs1 = o; o = up(o + nb * 4 * sizeof(OrgSave));
        order1 = o; o = up(o + nb * 4 * sizeof(uint2));
        tinfo1 = o; o = up(o + nb * 4 * sizeof(uint32_t));
        counters1 = o; o = up(o + 128 * sizeof(uint32_t));
        opts = o; o = up(o + nb * 16 * sizeof(OptSave));
        total = o;
    }
};
} // namespace

size_t bc6h_scratch_bytes(uint64_t nblocks, size_t nimages)
{
    return Scratch6(nblocks < kMaxBlocksPerPass6 ? nblocks : kMaxBlocksPerPass6).total + seg_table_bytes(nblocks, kMaxBlocksPerPass6, nimages);
}

hipError_t launch_bc6h_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, bool isSigned, void* scratch,
                              hipStream_t stream, KernelMarks* marks, const SideStreams* side)
{
    BcImage one; one.src = src; one.dst = dst; one.dstRowPitch = dstRowPitch;
    return launch_bc6h_encode_many(&one, 1, isSigned, scratch, stream, marks, side);
}

hipError_t launch_bc6h_encode_many(const BcImage* images, size_t count, bool isSigned, void* scratch, hipStream_t stream, KernelMarks* marks,
                                   const SideStreams* side)
{
#define DXTEX_MARK(NAME) do { if (marks) marks->mark(NAME); } while (0)
    std::vector<BcSeg> segs;
    std::vector<BcPass> passes;
    uint64_t perPass = 0;
    if (!build_passes(images, count, kMaxBlocksPerPass6, segs, passes, &perPass)) return hipSuccess;
    const Scratch6 L(perPass);
    uint8_t* base = static_cast<uint8_t*>(scratch);
    BcSeg* dSegs = reinterpret_cast<BcSeg*>(base + L.total);
    const hipError_t ce = upload_segments(dSegs, segs, passes, stream);
    if (ce != hipSuccess) return ce;
    for (const BcPass& pass : passes)
    {
        Bc6hArgs a;
        set_pass(a.seg, dSegs, segs, pass);
        a.nblocks = pass.nblocks;
        a.isSigned = isSigned ? 1 : 0;
        static const bool noPrune = dev_env("DXTEX_BC6H_NO_PRUNE") != nullptr;
        a.prune = noPrune ? 0 : 1;
        a.fpix = reinterpret_cast<float*>(base + L.fpix);
        a.lists = base + L.lists;
        a.seeds = reinterpret_cast<int*>(base + L.seeds);
        a.recs = reinterpret_cast<Rec6*>(base + L.recs);
        a.orgs = reinterpret_cast<OrgSave*>(base + L.orgs);
        a.order = reinterpret_cast<uint2*>(base + L.order);
        a.tinfo = reinterpret_cast<uint32_t*>(base + L.tinfo);
        a.counters = reinterpret_cast<uint32_t*>(base + L.counters);
        a.best = reinterpret_cast<Best6*>(base + L.best);
        a.bounds = reinterpret_cast<float*>(base + L.bounds);
        a.opts = reinterpret_cast<OptSave*>(base + L.opts);
        a.saveOpt = 0;
        a.boundsReady = 0;
        a.filterStats = 0;
        a.mode = ModeRt();

        DXTEX_MARK("bc6h_rough");
        hipLaunchKernelGGL(bc6h_rough_kernel, dim3((a.nblocks + 3) / 4), dim3(256), 0, stream, a);
        DXTEX_MARK("bc6h_block_seed");
        hipLaunchKernelGGL(bc6h_block_seed_kernel, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a);
        if (dev_env("DXTEX_BC6H_DUMP"))     // development aid: rank lists and seeds of the first blocks
        {
            (void)hipStreamSynchronize(stream);
            const uint32_t n = std::min<uint32_t>(a.nblocks, 4);
            std::vector<uint8_t> l(n * 8); std::vector<int> sd(n * SEED_INTS);
            (void)hipMemcpy(l.data(), a.lists, l.size(), hipMemcpyDeviceToHost);
            (void)hipMemcpy(sd.data(), a.seeds, sd.size() * 4, hipMemcpyDeviceToHost);
            for (uint32_t b = 0; b < n; ++b)
            {
                fprintf(stderr, "block %u shapes:", b);
                for (int i = 0; i < 8; ++i) fprintf(stderr, " %d", l[b * 8 + i]);
                fprintf(stderr, "\n  seeds rank0:");
                for (int i = 0; i < 12; ++i) fprintf(stderr, " %d", sd[b * SEED_INTS + i]);
                fprintf(stderr, "\n");
            }
        }
        static const int onlyMode = dev_env("DXTEX_BC6H_ONLY_MODE") ? atoi(dev_env("DXTEX_BC6H_ONLY_MODE")) : -1;     // development aid
        static const bool noSearch = dev_env("DXTEX_BC6H_NO_SEARCH") != nullptr;
        static const Bc6hMode kModes[14] = {
            { 0x00, 1, 1, 3, { 10, 10, 10 }, { 5, 5, 5 } }, { 0x01, 1, 1, 3, { 7, 7, 7 }, { 6, 6, 6 } }, { 0x02, 1, 1, 3, { 11, 11, 11 }, { 5, 4, 4 } },
            { 0x06, 1, 1, 3, { 11, 11, 11 }, { 4, 5, 4 } }, { 0x0a, 1, 1, 3, { 11, 11, 11 }, { 4, 4, 5 } }, { 0x0e, 1, 1, 3, { 9, 9, 9 }, { 5, 5, 5 } },
            { 0x12, 1, 1, 3, { 8, 8, 8 }, { 6, 5, 5 } }, { 0x16, 1, 1, 3, { 8, 8, 8 }, { 5, 6, 5 } }, { 0x1a, 1, 1, 3, { 8, 8, 8 }, { 5, 5, 6 } },
            { 0x1e, 1, 0, 3, { 6, 6, 6 }, { 6, 6, 6 } }, { 0x03, 0, 0, 4, { 10, 10, 10 }, { 10, 10, 10 } }, { 0x07, 0, 1, 4, { 11, 11, 11 }, { 9, 9, 9 } },
            { 0x0b, 0, 1, 4, { 12, 12, 12 }, { 8, 8, 8 } }, { 0x0f, 0, 1, 4, { 16, 16, 16 }, { 4, 4, 4 } } };     // == kBc6hModes (device table), host copy
        auto set_mode_on = [&](Bc6hArgs& a, int mi)
        {
            const Bc6hMode& k = kModes[mi];
            a.mode.index = mi; a.mode.code = k.code; a.mode.regions2 = k.regions2; a.mode.transformed = k.transformed; a.mode.prec = k.prec[0];
            for (int c = 0; c < 3; ++c) a.mode.delta[c] = k.delta[c];
        };
        auto set_mode = [&](int mi) { set_mode_on(a, mi); };
        auto sort_tasks = [&](const Bc6hArgs& a, uint32_t ntasks, hipStream_t stream)
        {
            const uint32_t binGroups = std::min<uint32_t>(kBinGroups, (ntasks + 255) / 256);
            (void)hipMemsetAsync(a.counters, 0, 128 * sizeof(uint32_t), stream);
            hipLaunchKernelGGL(bc7_bin_count_kernel, dim3(binGroups), dim3(256), 0, stream, a.tinfo, ntasks, a.counters);
            hipLaunchKernelGGL(bc7_bin_scan_kernel, dim3(1), dim3(1), 0, stream, a.counters);
            hipLaunchKernelGGL(bc7_bin_scatter_kernel, dim3(binGroups), dim3(256), 0, stream, a.tinfo, ntasks, a.counters, a.order);
        };
        a.taskBase = 0;
        for (int m = 0; m < 4; ++m) a.prec1[m] = kModes[10 + m].prec[0];
        // the ten two-region modes, one after the other: pre -> sort -> search -> post (post folds the mode into the running best, keyed by
        // (error, position in the encoder's order), so the order they RUN in is free). What the running order changes is how good an error is
        // on the table when a mode's pre prunes; modes of equal precision stay next to each other (they share one search).
        static const std::vector<int> order6 = []
        {
            std::vector<int> o;
            const char* e = dev_env("DXTEX_BC6H_ORDER");
            const char* p = e ? e : DXTEX_BC6H_DEFAULT_ORDER;
            while (*p) { if (*p >= '0' && *p <= '9') o.push_back(int(strtol(p, const_cast<char**>(&p), 10))); else ++p; }
            return o;
        }();
        int prevPrec = -1;
        a.samePrec = 0;
        std::vector<int> run2;
        for (int mi : order6) if (mi >= 0 && mi <= 9 && (onlyMode < 0 || mi == onlyMode)) run2.push_back(mi);
        // The one-region section (below) is two milliseconds of latency-bound kernels - a few thousand long tasks - that used to run alone after
        // the last two-region mode. Its search now runs on a side stream NEXT TO the last two-region modes (own task arrays; it prunes against the
        // best of the modes before that one - a looser table, never a wrong one: the bound is exact whatever it is compared with - and reads
        // best[].err while that mode's post may be writing it: a 32-bit word, old or new, both upper bounds of the final error; a zero in it
        // comes from a two-region mode either way, which is all org_candidate asks). Its posts fold into best[] after the join, in order.
        static const int forkBack = dev_env("DXTEX_BC6H_FORK_BACK") ? atoi(dev_env("DXTEX_BC6H_FORK_BACK")) : 2;      // 0 = serial, k = next to the last k two-region modes (cfg3 on one box: 54.90 / 54.97 / 54.10 / 54.21 ms for 0 / 1 / 2 / 3 - a persistent search kernel leaves the side stream little room, the gaps between two modes more)
        const SideStreams* fork = (marks || forkBack <= 0 || run2.size() < size_t(forkBack) + 1 || onlyMode >= 0) ? nullptr : side;
        Bc6hArgs a1 = a;
        a1.recs = reinterpret_cast<Rec6*>(base + L.recs1);
        a1.orgs = reinterpret_cast<OrgSave*>(base + L.orgs1);
        a1.order = reinterpret_cast<uint2*>(base + L.order1);
        a1.tinfo = reinterpret_cast<uint32_t*>(base + L.tinfo1);
        a1.counters = reinterpret_cast<uint32_t*>(base + L.counters1);
        a1.samePrec = 0;
        auto one_region_search = [&](hipStream_t stream, KernelMarks* marks)
        {
            const uint32_t gridPP = (a1.nblocks + 255) / 256;
            DXTEX_MARK("bc6h_pre_1region");
            a1.boundsReady = 0;
            for (int m = 0; m < 4; ++m)
            {
                if (onlyMode >= 0 && 10 + m != onlyMode) continue;
                set_mode_on(a1, 10 + m); a1.taskBase = uint32_t(m) * a1.nblocks;
                hipLaunchKernelGGL(bc6h_pre_kernel<0>, dim3(gridPP), dim3(256), 0, stream, a1);
                a1.boundsReady = 1;
            }
            const uint32_t ntasks = a1.nblocks * 4u;
            if (onlyMode >= 0)        // development aid: the slots of the modes that did not run hold no tasks
                for (int m = 0; m < 4; ++m) if (10 + m != onlyMode) (void)hipMemsetAsync(a1.tinfo + uint64_t(m) * a1.nblocks, 0, uint64_t(a1.nblocks) * 4, stream);
            DXTEX_MARK("bc6h_bin_1region");
            sort_tasks(a1, ntasks, stream);
            DXTEX_MARK("bc6h_perturb_1region");
            // both are launched; the live count (known on the device only) decides which one works
            static const uint32_t waveMax = dev_env("DXTEX_BC6H_WAVE_MAX") ? uint32_t(strtoul(dev_env("DXTEX_BC6H_WAVE_MAX"), nullptr, 0)) : kWaveTaskMax6;
            if (!noSearch)
            {
                hipLaunchKernelGGL(bc6h_perturb_kernel<16>, dim3(std::min<uint32_t>(kSearchWaves, (ntasks + 63) / 64)), dim3(64), 0, stream, a1, waveMax);
                hipLaunchKernelGGL(bc6h_perturb_wave_kernel, dim3(std::min<uint32_t>(kSearchWaves, ntasks)), dim3(64), 0, stream, a1, waveMax);
            }
        };
        bool forked = false;
        for (size_t at = 0; at < run2.size(); ++at)
        {
            const int mi = run2[at];
            if (fork && at + size_t(forkBack) == run2.size())
            {
                (void)hipEventRecord(fork->forked, stream);
                (void)hipStreamWaitEvent(fork->side[0], fork->forked, 0);
                one_region_search(fork->side[0], nullptr);
                (void)hipEventRecord(fork->joined[0], fork->side[0]);
                forked = true;
            }
            set_mode(mi);
            static const bool noReuse = dev_env("DXTEX_BC6H_NO_REUSE") != nullptr;      // development A/B: search every mode from scratch
            a.samePrec = (!noReuse && prevPrec == kModes[mi].prec[0]) ? 1 : 0;
            prevPrec = kModes[mi].prec[0];
            a.saveOpt = (!noReuse && at + 1 < run2.size() && kModes[run2[at + 1]].prec[0] == kModes[mi].prec[0]) ? 1 : 0;
            const uint32_t ntasks = a.nblocks * 16u;
            const uint32_t gridPP = (a.nblocks + 15) / 16;
            static const char* const kPre[10] = { "bc6h_pre_m0", "bc6h_pre_m1", "bc6h_pre_m2", "bc6h_pre_m3", "bc6h_pre_m4", "bc6h_pre_m5", "bc6h_pre_m6", "bc6h_pre_m7", "bc6h_pre_m8", "bc6h_pre_m9" };
            static const char* const kPerturb[10] = { "bc6h_perturb_m0", "bc6h_perturb_m1", "bc6h_perturb_m2", "bc6h_perturb_m3", "bc6h_perturb_m4", "bc6h_perturb_m5", "bc6h_perturb_m6", "bc6h_perturb_m7", "bc6h_perturb_m8", "bc6h_perturb_m9" };
            static const char* const kPost[10] = { "bc6h_post_m0", "bc6h_post_m1", "bc6h_post_m2", "bc6h_post_m3", "bc6h_post_m4", "bc6h_post_m5", "bc6h_post_m6", "bc6h_post_m7", "bc6h_post_m8", "bc6h_post_m9" };
            DXTEX_MARK(kPre[mi]);
            hipLaunchKernelGGL(bc6h_pre_kernel<1>, dim3(gridPP), dim3(256), 0, stream, a);
            a.boundsReady = 1;
            DXTEX_MARK("bc6h_bin_2region");
            sort_tasks(a, ntasks, stream);
#if defined(DXTEX_DEV)
            static const bool stats6 = dev_env("DXTEX_BC6H_STATS") != nullptr;       // development statistics: tasks that survive pre, per mode
            a.filterStats = stats6 ? 1 : 0;
            if (stats6)
            {
                uint32_t c[40] = {};
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(c, a.counters, sizeof(c), hipMemcpyDeviceToHost);
                std::fprintf(stderr, "bc6h stats mode %d (prec %d): %u task slots, %u live (16-texel: %u)\n", mi, a.mode.prec, ntasks, c[34], c[16]);
            }
#endif
            DXTEX_MARK(kPerturb[mi]);
            static const bool plainPerturb = dev_env("DXTEX_BC6H_PERTURB_PLAIN") != nullptr;      // development A/B: every candidate evaluated exactly, a lane per task
            if (!noSearch)
            {
                const dim3 grid(std::min<uint32_t>(kSearchWaves, (ntasks + 63) / 64));
                if (plainPerturb) hipLaunchKernelGGL(bc6h_perturb_kernel<8>, grid, dim3(64), 0, stream, a, 0u);
                else if (isSigned) hipLaunchKernelGGL(bc6h_perturb_filter_kernel<true>, grid, dim3(64), 0, stream, a);
                else hipLaunchKernelGGL(bc6h_perturb_filter_kernel<false>, grid, dim3(64), 0, stream, a);
            }
#if defined(DXTEX_DEV)
            if (stats6 && !plainPerturb)
            {
                uint32_t c[24] = {};
                (void)hipStreamSynchronize(stream);
                (void)hipMemcpy(c, a.counters + 64, sizeof(c), hipMemcpyDeviceToHost);
                auto u64 = [&](int i) { unsigned long long v = 0; memcpy(&v, c + i, 8); return double(v); };
                const double tk = u64(10), tx = u64(8), tb = u64(12), tl = u64(14), tr = u64(16), tp = u64(18), tm = u64(20);
                const double lanesMacro = c[4] ? double(c[5]) / c[4] : 0.0, lanesBound = c[0] ? double(c[7]) / c[0] : 0.0, lanesRounds = c[3] ? double(c[6]) / c[3] : 0.0,
                             lanesList = c[1] ? double(c[2]) / c[1] : 0.0;
                std::fprintf(stderr, "bc6h filter mode %d: %u macros (%.1f lanes active), %u pair steps, %u with passing candidates (%.2f per such step), %u list rounds; "
                             "exact evaluation %.1f %% of the waves' time\n", mi, c[4], lanesMacro, c[0], c[1], lanesList, c[3], tk ? 100.0 * tx / tk : 0.0);
                // where the wave time goes and how many lanes work there (time-weighted lanes = what "active lanes per VALU instruction" approximates)
                const double tOtherMacro = tm - tb - tx, tRest = tk - tm - tp, tBack = tx - tl - tr;
                const double weighted = (tb * lanesBound + tl * lanesList + tr * lanesRounds + tBack * lanesList + tOtherMacro * lanesMacro + tp * 32.0 + tRest * lanesMacro) / (tk ? tk : 1.0);
                std::fprintf(stderr, "bc6h filter sections mode %d | share of wave time, lanes busy: bound loops %.1f %% at %.1f | list write %.1f %% at %.1f | exact rounds %.1f %% at %.1f | "
                             "read-back + syncs %.1f %% at %.1f | rest of PerturbOne (fixed palettes, bound setup, transitions) %.1f %% at %.1f | task pickup %.1f %% | outside %.1f %% | "
                             "time-weighted lanes %.1f of 64\n", mi, 100 * tb / tk, lanesBound, 100 * tl / tk, lanesList, 100 * tr / tk, lanesRounds, 100 * tBack / tk, lanesList,
                             100 * tOtherMacro / tk, lanesMacro, 100 * tp / tk, 100 * tRest / tk, weighted);
            }
#endif
            DXTEX_MARK(kPost[mi]);
            hipLaunchKernelGGL(bc6h_post_kernel<1>, dim3(gridPP), dim3(256), 0, stream, a);
        }
        // The four one-region modes have ONE task per block each - a long serial chain on one lane - so a search kernel per mode would
        // only be as fast as its slowest lane. Their pre kernels run first (each prunes against the best of the two-region modes that have
        // finished), the four task lists are searched as one list (4x the tasks per lane, sorted by mode), then the posts fold the modes into
        // the running best in the reference's order.
        if (forked) (void)hipStreamWaitEvent(stream, fork->joined[0], 0);
        else one_region_search(stream, marks);
        {
            const uint32_t gridPP = (a1.nblocks + 255) / 256;
            DXTEX_MARK("bc6h_post_1region");
            for (int m = 0; m < 4; ++m)
            {
                if (onlyMode >= 0 && 10 + m != onlyMode) continue;
                set_mode_on(a1, 10 + m); a1.taskBase = uint32_t(m) * a1.nblocks;
                hipLaunchKernelGGL(bc6h_post_kernel<0>, dim3(gridPP), dim3(256), 0, stream, a1);
            }
        }
        DXTEX_MARK("bc6h_store");
        hipLaunchKernelGGL(bc6h_store_kernel, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a);
    }
    DXTEX_MARK(nullptr);
#undef DXTEX_MARK
    return hipGetLastError();
}
} // namespace dxtex
