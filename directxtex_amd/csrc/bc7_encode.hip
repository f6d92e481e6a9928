// BC7 encoder kernels for gfx950 (see bc7_core.h for what each task computes and why the result is
// byte-identical to D3DXEncodeBC7, BC6HBC7.cpp:3654).
//
// Decomposition (SIMT-friendly restatement of D3DX_BC7::Encode's mode x rotation x index-mode x shape loops):
//   rough   : one wavefront per block, lane = partition shape. Each lane fits the float seed endpoints of
//             its shape's two subsets once, scores them with the 3-bit and the 2-bit palettes (modes 1 / 3,7
//             share the seed), then the wavefront reproduces the reference's partial selection sort
//             (:2855-2865) with prefix-min scans and ballots and publishes the 16 best shapes per list.
//   refine2 : modes 1, 3, 7 - lane = (block, rank, subset): 32 lanes per block, two blocks per wavefront.
//             The two subset lanes of a candidate exchange totals with a lane shuffle; a butterfly
//             min-reduction over the 16 candidates picks the mode's winner, whose lane packs the block.
//   refine1 : modes 4, 5, 6 - lane = (block, rotation[, index mode]); texels live in registers.
//   pick    : lane = block; minimum over the per-mode winners in the reference's evaluation order.
// The per-mode winners travel through a small scratch buffer (24 B per mode per block).
#include "dxtex_kernels.h"
#include "bc67_tables.h"
#include "bc7_core.h"

namespace dxtex
{
namespace
{
using namespace bc7;

// Lanes of one wavefront exchange data through LDS: DS operations of a wave execute in order, so only the
// compiler has to be told not to move accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Cand { uint32_t err; uint32_t ord; uint64_t lo, hi; };   // ord = evaluation order inside D3DX_BC7::Encode

enum : int { SLOT_M0 = 0, SLOT_M1, SLOT_M2, SLOT_M3, SLOT_M4A, SLOT_M4B, SLOT_M5, SLOT_M6, SLOT_M7, NUM_SLOTS };
enum : int { LIST_BYTES = 64 };   // per block: [0..15] 3-bit list, [16..31] 2-bit list, [32] hasAlpha, [33..36] mode-0 list, [40..55] mode-2 list

struct Bc7Args
{
    SrcView src;
    uint8_t* dst;
    uint64_t dstRowPitch;
    uint32_t nbw, nbh, nblocks;
    uint32_t flags;
    uint8_t* lists;
    Cand* cands;
};

// One texel of block `nb` (texel t = y*4+x), with the reference's partial-block replication, as float4
// plus the 8-bit value D3DX_BC7::Encode derives from it (:2792-2799).
__device__ __forceinline__ void load_block_texel(const SrcView& src, uint32_t nbw, uint32_t nb, uint32_t t, float* f4, uint32_t& ldr)
{
    const uint32_t by = nb / nbw, bx = nb - by * nbw;
    const uint32_t x0 = bx * 4, y0 = by * 4;
    const uint32_t pw = min(4u, src.width - x0), ph = min(4u, src.height - y0);
    const uint32_t sx = x0 + replicate_src(t & 3, pw), sy = y0 + replicate_src(t >> 2, ph);
    const Texel px = convert_texel(load_texel(src.pixels + uint64_t(sy) * src.rowPitch, sx, src.format), src.tcv, src.tsw);
    f4[0] = px.r; f4[1] = px.g; f4[2] = px.b; f4[3] = px.a;
    const float c[4] = { px.r, px.g, px.b, px.a };
    ldr = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        float v = c[i] * 255.0f + 0.01f;
        v = (v < 255.0f) ? v : 255.0f;     // std::min<float>(255.0f, v)
        v = (0.0f < v) ? v : 0.0f;         // std::max<float>(0.0f, v)
        ldr |= (uint32_t(v) & 0xFFu) << (8 * i);
    }
}

// ---- rough: seeds + shape ranking -------------------------------------------------------------------------------
// Reproduces "bubble up the first uItems items" (:2855-2865): position i ends up with the first minimum
// of positions i.., and every strict prefix-minimum record along the way shifts to the next record's place.
__device__ __forceinline__ void selection_pass(int& e, uint32_t& s, int lane, int i)
{
    const int v = (lane >= i) ? e : 0x7FFFFFFF;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl = min(incl, o);
    }
    int excl = __shfl_up(incl, 1);
    if (lane <= i) excl = 0x7FFFFFFF;
    const bool isrec = (lane > i) && (e < excl);
    const unsigned long long mask = __ballot(isrec);
    int srcLane = lane;
    if (isrec)
    {
        const unsigned long long below = mask & ((1ull << lane) - 1ull);
        srcLane = below ? (63 - __clzll(below)) : i;
    }
    else if (lane == i && mask)
        srcLane = 63 - __clzll(mask);
    e = __shfl(e, srcLane);
    s = uint32_t(__shfl(int(s), srcLane));
}

__global__ void __launch_bounds__(256) bc7_rough_kernel(Bc7Args a)
{
    __shared__ float sF[4][64];
    __shared__ uint32_t sL[4][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nb = blockIdx.x * 4 + wave;
    if (nb >= a.nblocks) return;       // whole wave exits together

    if (lane < 16)
    {
        uint32_t ldr;
        load_block_texel(a.src, a.nbw, nb, lane, &sF[wave][lane * 4], ldr);
        sL[wave][lane] = ldr;
    }
    wave_lds_sync();
    const float* fpx = sF[wave];
    const uint32_t* pix = sL[wave];

    const bool alphaLane = (lane < 16) && ((pix[lane & 15] >> 24) != 0xFFu);
    const bool hasAlpha = __ballot(alphaLane) != 0ull;

    uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;

    // ---- 2-subset shapes: modes 1, 3, 7 ----
    {
        const uint32_t shape = lane;
        const uint32_t m1 = kPart2Mask[shape], m0 = (~m1) & 0xFFFFu;
        int e3 = 0, e2 = 0;
#pragma unroll 1
        for (int r = 0; r < 2; ++r)
        {
            const uint32_t m = r ? m1 : m0;
            Region rg; region_init(rg, pix, m);
            uint32_t A, B;
            if (rg.np == 1) { A = pix[rg.pos(0)]; B = A; }
            else if (rg.np == 2) { A = pix[rg.pos(0)]; B = pix[rg.pos(1)]; }
            else seed_endpoints<true>(fpx, m, A, B);
            e3 += rough_error<3, 0>(rg, A, B);
            e2 += rough_error<2, 0>(rg, A, B);
        }
        int ea = e3, eb = e2;
        uint32_t sa = shape, sb = shape;
        for (int i = 0; i < 16; ++i)
        {
            selection_pass(ea, sa, lane, i);
            selection_pass(eb, sb, lane, i);
        }
        if (lane < 16) { lst[lane] = uint8_t(sa); lst[16 + lane] = uint8_t(sb); }
        if (lane == 0) lst[32] = hasAlpha ? 1 : 0;
    }

    // ---- 3-subset shapes: modes 0 (first 16 shapes) and 2 (64 shapes) ----
    if (a.flags & BCF_USE_3SUBSETS)
    {
        const uint32_t shape = lane;
        const uint32_t bits = kPart3Bits[shape];
        int e3 = 0, e2 = 0;
#pragma unroll 1
        for (uint32_t r = 0; r < 3; ++r)
        {
            uint32_t m = 0;
            for (int i = 0; i < 16; ++i) if (((bits >> (2 * i)) & 3u) == r) m |= 1u << i;
            Region rg; region_init(rg, pix, m);
            uint32_t A, B;
            if (rg.np == 1) { A = pix[rg.pos(0)]; B = A; }
            else if (rg.np == 2) { A = pix[rg.pos(0)]; B = pix[rg.pos(1)]; }
            else seed_endpoints<true>(fpx, m, A, B);
            e3 += rough_error<3, 0>(rg, A, B);
            e2 += rough_error<2, 0>(rg, A, B);
        }
        // mode 0: 16 shapes, uItems = 4; lanes >= 16 must never win: give them +inf
        int ea = (lane < 16) ? e3 : 0x7FFFFFFF, eb = e2;
        uint32_t sa = shape, sb = shape;
        for (int i = 0; i < 16; ++i)
        {
            if (i < 4) selection_pass(ea, sa, lane, i);
            selection_pass(eb, sb, lane, i);
        }
        if (lane < 4) lst[33 + lane] = uint8_t(sa);
        if (lane < 16) lst[40 + lane] = uint8_t(sb);
    }
}

// ---- refine2: modes 1, 3, 7 (two subsets) ------------------------------------------------------------------------
template<int MODE>
__global__ void __launch_bounds__(256) bc7_refine2_kernel(Bc7Args a)
{
    __shared__ float sF[4][2][64];
    __shared__ uint32_t sL[4][2][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, rank = (lane >> 1) & 15, region = lane & 1;
    const uint32_t nb0 = (blockIdx.x * 4 + wave) * 2;
    if (nb0 >= a.nblocks) return;
    const uint32_t nb = nb0 + half;
    const bool valid = nb < a.nblocks;

    if ((lane & 31) < 16 && valid)
    {
        uint32_t ldr;
        load_block_texel(a.src, a.nbw, nb, lane & 15, &sF[wave][half][(lane & 15) * 4], ldr);
        sL[wave][half][lane & 15] = ldr;
    }
    wave_lds_sync();

    const uint8_t* lst = a.lists + uint64_t(valid ? nb : nb0) * LIST_BYTES;
    const bool active = valid && !(MODE == 7 && lst[32] == 0);
    const int slot = (MODE == 1) ? SLOT_M1 : (MODE == 3) ? SLOT_M3 : SLOT_M7;

    SubsetResult res;
    res.orgErr = 0; res.optErr = 0; res.orgA = res.orgB = res.optA = res.optB = 0;
    res.orgIdx1 = res.orgIdx2 = res.optIdx1 = res.optIdx2 = 0;
    uint32_t shape = 0;
    if (active)
    {
        const float* fpx = sF[wave][half];
        const uint32_t* pix = sL[wave][half];
        shape = lst[(MODE == 1 ? 0 : 16) + rank];
        const uint32_t m1 = kPart2Mask[shape];
        const uint32_t m = region ? m1 : ((~m1) & 0xFFFFu);
        Region rg; region_init(rg, pix, m);
        uint32_t A, B;
        if (rg.np == 1) { A = pix[rg.pos(0)]; B = A; }
        else if (rg.np == 2) { A = pix[rg.pos(0)]; B = pix[rg.pos(1)]; }
        else seed_endpoints<true>(fpx, m, A, B);
        refine_subset<MODE, 0>(rg, A, B, region ? uint32_t(kAnchor2[shape]) : 0u, res);
    }

    // candidate totals over the two subset lanes (fOrgTotErr / fOptTotErr, :3447-3452)
    const int orgTot = res.orgErr + __shfl_xor(res.orgErr, 1);
    const int optTot = res.optErr + __shfl_xor(res.optErr, 1);
    const bool useOpt = optTot < orgTot;
    const int err = useOpt ? optTot : orgTot;
    const uint32_t myA = useOpt ? res.optA : res.orgA, myB = useOpt ? res.optB : res.orgB;
    const uint64_t myIdx = useOpt ? res.optIdx1 : res.orgIdx1;
    const uint32_t otherA = uint32_t(__shfl_xor(int(myA), 1)), otherB = uint32_t(__shfl_xor(int(myB), 1));
    const uint64_t otherIdx = uint64_t(uint32_t(__shfl_xor(int(uint32_t(myIdx)), 1))) |
                              (uint64_t(uint32_t(__shfl_xor(int(uint32_t(myIdx >> 32)), 1))) << 32);

    // first minimum over the 16 candidates of this block, in evaluation order (strict <, :2870)
    uint32_t key = (uint32_t(err) << 4) | uint32_t(rank);
    uint32_t best = key;
#pragma unroll
    for (int d = 2; d < 32; d <<= 1) best = min(best, uint32_t(__shfl_xor(int(best), d)));

    if (active && region == 0 && key == best)
    {
        const uint32_t epA[3] = { myA, otherA, 0 }, epB[3] = { myB, otherB, 0 };
        const uint32_t anchor[3] = { 0, kAnchor2[shape], 0 };
        Cand c;
        c.err = uint32_t(err);
        c.ord = uint32_t(MODE) * 128u + uint32_t(rank);
        emit_block<MODE>(shape, 0, 0, epA, epB, myIdx | otherIdx, 0, anchor, c.lo, c.hi);
        a.cands[uint64_t(nb) * NUM_SLOTS + slot] = c;
    }
    else if (valid && !active && region == 0 && rank == 0)
    {
        Cand c; c.err = 0xFFFFFFFFu; c.ord = 0xFFFFFFFFu; c.lo = 0; c.hi = 0;
        a.cands[uint64_t(nb) * NUM_SLOTS + slot] = c;
    }
}

// ---- refine1: modes 4, 5, 6 (one subset, texels in registers) -----------------------------------------------------
// MODE 4 runs once per index mode (IM) so that a wavefront never mixes the two palette shapes.
template<int MODE, int IM>
__global__ void __launch_bounds__(256) bc7_refine1_kernel(Bc7Args a)
{
    constexpr int T = (MODE == 6) ? 1 : 4;            // candidates (rotations) per block in this launch
    constexpr int BPW = 64 / T;                        // blocks per wavefront
    constexpr int FSTRIDE = 65, LSTRIDE = 17;          // odd strides: lane = block reads stay conflict-free
    __shared__ float sF[4][BPW * FSTRIDE];
    __shared__ uint32_t sL[4][BPW * LSTRIDE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbBase = (blockIdx.x * 4 + wave) * BPW;
    if (nbBase >= a.nblocks) return;

    for (int t = lane; t < BPW * 16; t += 64)
    {
        const uint32_t b = uint32_t(t) >> 4, nbt = nbBase + b;
        if (nbt < a.nblocks)
        {
            float f4[4]; uint32_t ldr;
            load_block_texel(a.src, a.nbw, nbt, t & 15, f4, ldr);
#pragma unroll
            for (int c = 0; c < 4; ++c) sF[wave][b * FSTRIDE + (t & 15) * 4 + c] = f4[c];
            sL[wave][b * LSTRIDE + (t & 15)] = ldr;
        }
    }
    wave_lds_sync();

    const uint32_t b = uint32_t(lane) / T, rot = uint32_t(lane) % T;
    const uint32_t nb = nbBase + b;
    const bool valid = nb < a.nblocks;
    const int slot = (MODE == 6) ? SLOT_M6 : (MODE == 5) ? SLOT_M5 : (IM ? SLOT_M4B : SLOT_M4A);

    SubsetResult res;
    res.orgErr = 0; res.optErr = 0; res.orgA = res.orgB = res.optA = res.optB = 0;
    res.orgIdx1 = res.orgIdx2 = res.optIdx1 = res.optIdx2 = 0;
    if (valid)
    {
        const float* fpx = &sF[wave][b * FSTRIDE];
        Block16 rg;
        block16_init(rg, &sL[wave][b * LSTRIDE], (MODE == 6) ? 0u : rot);
        uint32_t A, B;
        if (MODE == 6)
            seed_endpoints<true>(fpx, 0xFFFFu, A, B);
        else
        {
            // colour endpoints from the *unrotated* float texels, alpha endpoints = min/max of the rotated
            // 8-bit alpha (:3552-3568)
            seed_endpoints<false>(fpx, 0xFFFFu, A, B);
            uint32_t mn = 255, mx = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const uint32_t al = rg.px[i] >> 24; mn = min(mn, al); mx = max(mx, al); }
            A = (A & 0x00FFFFFFu) | (mn << 24);
            B = (B & 0x00FFFFFFu) | (mx << 24);
        }
        refine_subset<MODE, IM>(rg, A, B, 0u, res);
    }

    const bool useOpt = res.optErr < res.orgErr;
    const int err = useOpt ? res.optErr : res.orgErr;
    const uint32_t sub = (MODE == 4) ? (rot * 2 + IM) : rot;
    uint32_t key = (uint32_t(err) << 4) | sub;
    uint32_t best = key;
#pragma unroll
    for (int d = 1; d < T; d <<= 1) best = min(best, uint32_t(__shfl_xor(int(best), d)));

    if (valid && key == best)
    {
        const uint32_t epA[3] = { useOpt ? res.optA : res.orgA, 0, 0 }, epB[3] = { useOpt ? res.optB : res.orgB, 0, 0 };
        const uint32_t anchor[3] = { 0, 0, 0 };
        Cand c;
        c.err = uint32_t(err);
        c.ord = uint32_t(MODE) * 128u + sub * 16u;
        emit_block<MODE>(0, rot, IM, epA, epB, useOpt ? res.optIdx1 : res.orgIdx1, useOpt ? res.optIdx2 : res.orgIdx2, anchor, c.lo, c.hi);
        a.cands[uint64_t(nb) * NUM_SLOTS + slot] = c;
    }
}

// ---- pick: first minimum over the per-mode winners, in D3DX_BC7::Encode's order -----------------------------------
__global__ void __launch_bounds__(256) bc7_pick_kernel(Bc7Args a, uint32_t slotMask)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nblocks) return;
    const Cand* c = a.cands + uint64_t(nb) * NUM_SLOTS;
    uint64_t bestKey = ~0ull;
    uint64_t lo = 0, hi = 0;
#pragma unroll
    for (int s = 0; s < NUM_SLOTS; ++s)
    {
        if (!((slotMask >> s) & 1u)) continue;
        const Cand v = c[s];
        if (v.err == 0xFFFFFFFFu) continue;
        const uint64_t key = (uint64_t(v.err) << 32) | v.ord;
        if (key < bestKey) { bestKey = key; lo = v.lo; hi = v.hi; }
    }
    const uint32_t by = nb / a.nbw, bx = nb - by * a.nbw;
    uint64_t* out = reinterpret_cast<uint64_t*>(a.dst + uint64_t(by) * a.dstRowPitch) + 2 * uint64_t(bx);
    out[0] = lo; out[1] = hi;
}
} // namespace

size_t bc7_scratch_bytes(uint64_t nblocks)
{
    return size_t(nblocks) * (LIST_BYTES + NUM_SLOTS * sizeof(Cand));
}

hipError_t launch_bc7_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, uint32_t flags,
                             void* scratch, hipStream_t stream, KernelMarks* marks)
{
#define DXTEX_MARK(NAME) do { if (marks) marks->mark(NAME); } while (0)
    Bc7Args a;
    a.src = src; a.dst = dst; a.dstRowPitch = dstRowPitch;
    a.nbw = (src.width + 3) / 4; a.nbh = (src.height + 3) / 4;
    a.nblocks = a.nbw * a.nbh;
    a.flags = flags;
    a.lists = static_cast<uint8_t*>(scratch);
    a.cands = reinterpret_cast<Cand*>(static_cast<uint8_t*>(scratch) + size_t(a.nblocks) * LIST_BYTES);
    if (!a.nblocks) return hipSuccess;
    const uint32_t nb = a.nblocks;
    const bool quick = (flags & BCF_BC7_QUICK) != 0;
    uint32_t slotMask = 0;

    if (!quick)
    {
        DXTEX_MARK("bc7_rough");
        hipLaunchKernelGGL(bc7_rough_kernel, dim3((nb + 3) / 4), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine2_mode1");
        hipLaunchKernelGGL(bc7_refine2_kernel<1>, dim3((nb + 7) / 8), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine2_mode3");
        hipLaunchKernelGGL(bc7_refine2_kernel<3>, dim3((nb + 7) / 8), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine2_mode7");
        hipLaunchKernelGGL(bc7_refine2_kernel<7>, dim3((nb + 7) / 8), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine1_mode4_im0");
        hipLaunchKernelGGL((bc7_refine1_kernel<4, 0>), dim3((nb + 63) / 64), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine1_mode4_im1");
        hipLaunchKernelGGL((bc7_refine1_kernel<4, 1>), dim3((nb + 63) / 64), dim3(256), 0, stream, a);
        DXTEX_MARK("bc7_refine1_mode5");
        hipLaunchKernelGGL((bc7_refine1_kernel<5, 0>), dim3((nb + 63) / 64), dim3(256), 0, stream, a);
        slotMask |= (1u << SLOT_M1) | (1u << SLOT_M3) | (1u << SLOT_M7) | (1u << SLOT_M4A) | (1u << SLOT_M4B) | (1u << SLOT_M5);
    }
    DXTEX_MARK("bc7_refine1_mode6");
    hipLaunchKernelGGL((bc7_refine1_kernel<6, 0>), dim3((nb + 255) / 256), dim3(256), 0, stream, a);
    slotMask |= (1u << SLOT_M6);
    DXTEX_MARK("bc7_pick");
    hipLaunchKernelGGL(bc7_pick_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, a, slotMask);
    DXTEX_MARK(nullptr);
#undef DXTEX_MARK
    return hipGetLastError();
}
} // namespace dxtex
