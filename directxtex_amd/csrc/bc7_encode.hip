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

// ---- per-mode search kernels ---------------------------------------------------------------------------------------
// One wavefront owns a run of consecutive blocks and works through all their tasks of one mode in three phases:
//   pre    (lane = task, fixed assignment, uniform cost): seed -> Quantize -> FixEndpointPBits -> AssignIndices;
//          the "org" endpoints and error of every task go to an LDS task table.
//   search (persistent lanes): OptimizeOne as a state machine (bc7_core.h). Every loop iteration scores ONE
//          candidate endpoint pair per lane with a fully converged map_colors; a lane whose search ends stores
//          its result and takes the next task from the wave's table, so lanes stay busy although searches differ
//          in length by an order of magnitude. Tasks are handed out in order of decreasing subset size so that
//          the lanes of a wave loop over similar texel counts.
//   post   (lane = task, fixed assignment): FixEndpointPBits + AssignIndices of the optimised endpoints, org-vs-opt
//          decision over the subsets of a candidate (lane shuffle), first-minimum over the block's candidates
//          (butterfly), EmitBlock by the winning lane.

template<int MODE>
__global__ void __launch_bounds__(64) bc7_subset2_kernel(Bc7Args a)
{
    constexpr int NB = 16;              // blocks per wavefront
    constexpr int T = NB * 32;          // tasks: 16 candidate shapes x 2 subsets per block
    constexpr int ROUNDS = T / 64;
    __shared__ float sF[NB * 64];
    __shared__ uint32_t sL[NB * 16];
    __shared__ uint32_t tA[T], tB[T];
    __shared__ int tErr[T];
    __shared__ uint16_t sOrder[T];
    __shared__ uint8_t sNp[T];
    __shared__ uint32_t sSlot[16 * 64];
    __shared__ uint32_t sBins[20];

    const int lane = threadIdx.x;
    const uint32_t nb0 = blockIdx.x * NB;
    const int slotIdx = (MODE == 1) ? SLOT_M1 : (MODE == 3) ? SLOT_M3 : SLOT_M7;
    const int listOfs = (MODE == 1) ? 0 : 16;

    for (int t = lane; t < NB * 16; t += 64)
    {
        const uint32_t nb = nb0 + (uint32_t(t) >> 4);
        uint32_t ldr = 0;
        if (nb < a.nblocks) load_block_texel(a.src, a.nbw, nb, t & 15, &sF[(t >> 4) * 64 + (t & 15) * 4], ldr);
        sL[t] = ldr;
    }
    if (lane < 20) sBins[lane] = 0;
    wave_lds_sync();

    // ---- pre ----
    auto task_setup = [&](int t, uint32_t& nb, uint32_t& shape, uint32_t& mask, uint32_t& anchor) -> bool
    {
        const int blk = t >> 5, rank = (t >> 1) & 15, region = t & 1;
        nb = nb0 + uint32_t(blk);
        if (nb >= a.nblocks) return false;
        const uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;
        if (MODE == 7 && lst[32] == 0) return false;
        shape = lst[listOfs + rank];
        const uint32_t m1 = kPart2Mask[shape];
        mask = region ? m1 : ((~m1) & 0xFFFFu);
        anchor = region ? uint32_t(kAnchor2[shape]) : 0u;
        return true;
    };
    auto task_pre = [&](int t, uint32_t mask, uint32_t anchor, Region& rg, SubsetResult& res)
    {
        const int blk = t >> 5;
        const float* fpx = &sF[blk * 64];
        const uint32_t* pix = &sL[blk * 16];
        region_init(rg, pix, mask);
        uint32_t A, B;
        if (rg.np == 1) { A = pix[rg.pos(0)]; B = A; }
        else if (rg.np == 2) { A = pix[rg.pos(0)]; B = pix[rg.pos(1)]; }
        else seed_endpoints<true>(fpx, mask, A, B);
        refine_pre<MODE, 0>(rg, A, B, anchor, res);
    };

    for (int r = 0; r < ROUNDS; ++r)
    {
        const int t = r * 64 + lane;
        uint32_t nb, shape = 0, mask = 0, anchor = 0;
        const bool active = task_setup(t, nb, shape, mask, anchor);
        int np = 0;
        if (active)
        {
            Region rg; SubsetResult res;
            task_pre(t, mask, anchor, rg, res);
            tA[t] = res.orgA; tB[t] = res.orgB; tErr[t] = res.orgErr;
            np = rg.np;
            atomicAdd(&sBins[16 - np], 1u);
        }
        sNp[t] = uint8_t(np);
    }
    wave_lds_sync();
    // exclusive prefix over the size bins (largest subsets first)
    if (lane == 0)
    {
        uint32_t run = 0;
        for (int i = 0; i <= 16; ++i) { const uint32_t c = sBins[i]; sBins[i] = run; run += c; }
        sBins[17] = run;
    }
    wave_lds_sync();
    for (int r = 0; r < ROUNDS; ++r)
    {
        const int t = r * 64 + lane;
        const int np = sNp[t];
        if (np > 0) sOrder[atomicAdd(&sBins[16 - np], 1u)] = uint16_t(t);
    }
    wave_lds_sync();
    const int nTasks = int(sBins[17]);

    // ---- search ----
    {
        SearchState st; st.phase = 0;
        SlotRegion rg; rg.base = &sSlot[lane]; rg.np = 0; rg.p2sum = 0;
        int myTask = -1;
        int nextIdx = 0;
        for (;;)
        {
            if (myTask >= 0 && st.phase == 0)
            {
                tA[myTask] = st.optA; tB[myTask] = st.optB;
                myTask = -1;
            }
            const unsigned long long idle = __ballot(myTask < 0);
            if (idle && nextIdx < nTasks)
            {
                const int k = __popcll(idle & ((1ull << lane) - 1ull));
                if (myTask < 0 && nextIdx + k < nTasks)
                {
                    myTask = sOrder[nextIdx + k];
                    const int blk = myTask >> 5, rank = (myTask >> 1) & 15, region = myTask & 1;
                    const uint8_t* lst = a.lists + uint64_t(nb0 + blk) * LIST_BYTES;
                    const uint32_t shape = lst[listOfs + rank];
                    const uint32_t m1 = kPart2Mask[shape];
                    const uint32_t mask = region ? m1 : ((~m1) & 0xFFFFu);
                    const uint32_t* pix = &sL[blk * 16];
                    int np = 0, p2 = 0;
                    for (uint32_t i = 0; i < 16; ++i)
                        if ((mask >> i) & 1u)
                        {
                            const uint32_t p = pix[i];
                            sSlot[np * 64 + lane] = p;
                            p2 += int(udot4(p, p));
                            ++np;
                        }
                    rg.np = np; rg.p2sum = p2;
                    st = ss_begin<MODE>(tA[myTask], tB[myTask], tErr[myTask]);
                }
                nextIdx += __popcll(idle);
            }
            if (__ballot(myTask >= 0) == 0ull) break;
            bool has = false;
            if (myTask >= 0) st = ss_next<MODE>(st, has);
            if (has)
            {
                const int e = map_colors<MODE, 0>(rg, st.candA, st.candB);
                st = ss_consume(st, e);
            }
        }
    }
    wave_lds_sync();

    // ---- post ----
    for (int r = 0; r < ROUNDS; ++r)
    {
        const int t = r * 64 + lane;
        const int rank = (t >> 1) & 15, region = t & 1;
        uint32_t nb, shape = 0, mask = 0, anchor = 0;
        const bool active = task_setup(t, nb, shape, mask, anchor);
        SubsetResult res;
        res.orgErr = 0; res.optErr = 0; res.orgA = res.orgB = res.optA = res.optB = 0;
        res.orgIdx1 = res.orgIdx2 = res.optIdx1 = res.optIdx2 = 0;
        if (active)
        {
            Region rg;
            task_pre(t, mask, anchor, rg, res);
            refine_post<MODE, 0>(rg, tA[t], tB[t], anchor, res);
        }
        const int orgTot = res.orgErr + __shfl_xor(res.orgErr, 1);
        const int optTot = res.optErr + __shfl_xor(res.optErr, 1);
        const bool useOpt = optTot < orgTot;
        const int err = useOpt ? optTot : orgTot;
        const uint32_t myA = useOpt ? res.optA : res.orgA, myB = useOpt ? res.optB : res.orgB;
        const uint64_t myIdx = useOpt ? res.optIdx1 : res.orgIdx1;
        const uint32_t otherA = uint32_t(__shfl_xor(int(myA), 1)), otherB = uint32_t(__shfl_xor(int(myB), 1));
        const uint64_t otherIdx = uint64_t(uint32_t(__shfl_xor(int(uint32_t(myIdx)), 1))) |
                                  (uint64_t(uint32_t(__shfl_xor(int(uint32_t(myIdx >> 32)), 1))) << 32);
        const uint32_t key = (uint32_t(err) << 4) | uint32_t(rank);
        uint32_t best = key;
#pragma unroll
        for (int d = 2; d < 32; d <<= 1) best = min(best, uint32_t(__shfl_xor(int(best), d)));

        if (active && region == 0 && key == best)
        {
            const uint32_t epA[3] = { myA, otherA, 0 }, epB[3] = { myB, otherB, 0 };
            const uint32_t anchors[3] = { 0, kAnchor2[shape], 0 };
            Cand c;
            c.err = uint32_t(err);
            c.ord = uint32_t(MODE) * 128u + uint32_t(rank);
            emit_block<MODE>(shape, 0, 0, epA, epB, myIdx | otherIdx, 0, anchors, c.lo, c.hi);
            a.cands[uint64_t(nb) * NUM_SLOTS + slotIdx] = c;
        }
        else if (!active && region == 0 && rank == 0 && (nb0 + uint32_t(t >> 5)) < a.nblocks)
        {
            Cand c; c.err = 0xFFFFFFFFu; c.ord = 0xFFFFFFFFu; c.lo = 0; c.hi = 0;
            a.cands[uint64_t(nb0 + uint32_t(t >> 5)) * NUM_SLOTS + slotIdx] = c;
        }
    }
}

// Modes 4, 5, 6: one subset, the block's texels live in registers. TB candidates per block in one launch
// (mode 4 runs once per index mode so a wavefront never mixes palette shapes).
template<int MODE, int IM>
__global__ void __launch_bounds__(64) bc7_subset1_kernel(Bc7Args a)
{
    constexpr int TB = (MODE == 6) ? 1 : 4;
    constexpr int T = 512;              // tasks per wavefront
    constexpr int NB = T / TB;
    constexpr int ROUNDS = T / 64;
    __shared__ uint32_t tA[T], tB[T];
    __shared__ int tErr[T];

    const int lane = threadIdx.x;
    const uint32_t nb0 = blockIdx.x * NB;
    const int slotIdx = (MODE == 6) ? SLOT_M6 : (MODE == 5) ? SLOT_M5 : (IM ? SLOT_M4B : SLOT_M4A);

    auto load_regs = [&](uint32_t nb, float (&f)[64], uint32_t (&px)[16])
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) load_block_texel(a.src, a.nbw, nb, i, &f[i * 4], px[i]);
    };
    auto task_pre = [&](uint32_t nb, uint32_t rot, Block16& rg, SubsetResult& res)
    {
        float f[64]; uint32_t px[16];
        load_regs(nb, f, px);
        block16_init(rg, px, (MODE == 6) ? 0u : rot);
        uint32_t A, B;
        if (MODE == 6)
            seed_endpoints<true, true>(f, 0xFFFFu, A, B);
        else
        {
            // colour endpoints from the *unrotated* float texels, alpha endpoints = min/max of the rotated
            // 8-bit alpha (:3552-3568)
            seed_endpoints<false, true>(f, 0xFFFFu, A, B);
            uint32_t mn = 255, mx = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const uint32_t al = rg.px[i] >> 24; mn = min(mn, al); mx = max(mx, al); }
            A = (A & 0x00FFFFFFu) | (mn << 24);
            B = (B & 0x00FFFFFFu) | (mx << 24);
        }
        refine_pre<MODE, IM>(rg, A, B, 0u, res);
    };

    // ---- pre ----
    for (int r = 0; r < ROUNDS; ++r)
    {
        const int t = r * 64 + lane;
        const uint32_t nb = nb0 + uint32_t(t) / TB, rot = uint32_t(t) % TB;
        if (nb < a.nblocks)
        {
            Block16 rg; SubsetResult res;
            task_pre(nb, rot, rg, res);
            tA[t] = res.orgA; tB[t] = res.orgB; tErr[t] = res.orgErr;
        }
    }
    wave_lds_sync();
    const int nTasks = int(min(uint32_t(T), (a.nblocks > nb0 ? (a.nblocks - nb0) : 0u) * TB));

    // ---- search ----
    {
        SearchState st; st.phase = 0;
        Block16 rg;
#pragma unroll
        for (int i = 0; i < 16; ++i) rg.px[i] = 0;
        rg.p2sum = 0;
        int myTask = -1;
        int nextIdx = 0;
        for (;;)
        {
            if (myTask >= 0 && st.phase == 0)
            {
                tA[myTask] = st.optA; tB[myTask] = st.optB;
                myTask = -1;
            }
            const unsigned long long idle = __ballot(myTask < 0);
            if (idle && nextIdx < nTasks)
            {
                const int k = __popcll(idle & ((1ull << lane) - 1ull));
                if (myTask < 0 && nextIdx + k < nTasks)
                {
                    myTask = nextIdx + k;
                    const uint32_t nb = nb0 + uint32_t(myTask) / TB, rot = uint32_t(myTask) % TB;
                    int p2 = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                    {
                        float f4[4]; uint32_t ldr;
                        load_block_texel(a.src, a.nbw, nb, i, f4, ldr);
                        ldr = rotate_pixel(ldr, (MODE == 6) ? 0u : rot);
                        rg.px[i] = ldr;
                        p2 += int(udot4(ldr, ldr));
                    }
                    rg.p2sum = p2;
                    st = ss_begin<MODE>(tA[myTask], tB[myTask], tErr[myTask]);
                }
                nextIdx += __popcll(idle);
            }
            if (__ballot(myTask >= 0) == 0ull) break;
            bool has = false;
            if (myTask >= 0) st = ss_next<MODE>(st, has);
            if (has)
            {
                const int e = map_colors<MODE, IM>(rg, st.candA, st.candB);
                st = ss_consume(st, e);
            }
        }
    }
    wave_lds_sync();

    // ---- post ----
    for (int r = 0; r < ROUNDS; ++r)
    {
        const int t = r * 64 + lane;
        const uint32_t nb = nb0 + uint32_t(t) / TB, rot = uint32_t(t) % TB;
        const bool valid = nb < a.nblocks;
        SubsetResult res;
        res.orgErr = 0; res.optErr = 0; res.orgA = res.orgB = res.optA = res.optB = 0;
        res.orgIdx1 = res.orgIdx2 = res.optIdx1 = res.optIdx2 = 0;
        if (valid)
        {
            Block16 rg;
            task_pre(nb, rot, rg, res);
            refine_post<MODE, IM>(rg, tA[t], tB[t], 0u, res);
        }
        const bool useOpt = res.optErr < res.orgErr;
        const int err = useOpt ? res.optErr : res.orgErr;
        const uint32_t sub = (MODE == 4) ? (rot * 2 + IM) : rot;
        const uint32_t key = (uint32_t(err) << 4) | sub;
        uint32_t best = key;
#pragma unroll
        for (int d = 1; d < TB; d <<= 1) best = min(best, uint32_t(__shfl_xor(int(best), d)));
        if (valid && key == best)
        {
            const uint32_t epA[3] = { useOpt ? res.optA : res.orgA, 0, 0 }, epB[3] = { useOpt ? res.optB : res.orgB, 0, 0 };
            const uint32_t anchors[3] = { 0, 0, 0 };
            Cand c;
            c.err = uint32_t(err);
            c.ord = uint32_t(MODE) * 128u + sub * 16u;
            emit_block<MODE>(0, rot, IM, epA, epB, useOpt ? res.optIdx1 : res.orgIdx1, useOpt ? res.optIdx2 : res.orgIdx2, anchors, c.lo, c.hi);
            a.cands[uint64_t(nb) * NUM_SLOTS + slotIdx] = c;
        }
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
        DXTEX_MARK("bc7_subset2_mode1");
        hipLaunchKernelGGL(bc7_subset2_kernel<1>, dim3((nb + 15) / 16), dim3(64), 0, stream, a);
        DXTEX_MARK("bc7_subset2_mode3");
        hipLaunchKernelGGL(bc7_subset2_kernel<3>, dim3((nb + 15) / 16), dim3(64), 0, stream, a);
        DXTEX_MARK("bc7_subset2_mode7");
        hipLaunchKernelGGL(bc7_subset2_kernel<7>, dim3((nb + 15) / 16), dim3(64), 0, stream, a);
        DXTEX_MARK("bc7_subset1_mode4_im0");
        hipLaunchKernelGGL((bc7_subset1_kernel<4, 0>), dim3((nb + 127) / 128), dim3(64), 0, stream, a);
        DXTEX_MARK("bc7_subset1_mode4_im1");
        hipLaunchKernelGGL((bc7_subset1_kernel<4, 1>), dim3((nb + 127) / 128), dim3(64), 0, stream, a);
        DXTEX_MARK("bc7_subset1_mode5");
        hipLaunchKernelGGL((bc7_subset1_kernel<5, 0>), dim3((nb + 127) / 128), dim3(64), 0, stream, a);
        slotMask |= (1u << SLOT_M1) | (1u << SLOT_M3) | (1u << SLOT_M7) | (1u << SLOT_M4A) | (1u << SLOT_M4B) | (1u << SLOT_M5);
    }
    DXTEX_MARK("bc7_subset1_mode6");
    hipLaunchKernelGGL((bc7_subset1_kernel<6, 0>), dim3((nb + 511) / 512), dim3(64), 0, stream, a);
    slotMask |= (1u << SLOT_M6);
    DXTEX_MARK("bc7_pick");
    hipLaunchKernelGGL(bc7_pick_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, a, slotMask);
    DXTEX_MARK(nullptr);
#undef DXTEX_MARK
    return hipGetLastError();
}
} // namespace dxtex
