// BC7 encoder kernels for gfx950 (see bc7_core.h for what each task computes and why the result is
// byte-identical to D3DXEncodeBC7, BC6HBC7.cpp:3654).
//
// Decomposition (SIMT-friendly restatement of D3DX_BC7::Encode's mode x rotation x index-mode x shape loops):
//   rough   : one wavefront per block, lane = partition shape. Each lane fits the float seed endpoints of
//             its shape's subsets once, scores them with the 3-bit and the 2-bit palettes (modes 1 / 3,7
//             share the seed), then the wavefront reproduces the reference's partial selection sort
//             (:2855-2865) with prefix-min scans and ballots and publishes the best shapes per list. Also
//             stores the block's 16 texels as packed RGBA8 for the search kernels, the fitted endpoints of every
//             shape (Refine starts from the same fit) and two scheduling flags (has alpha; mode 6 first).
//   per mode (1, 3, 7, 4 x 2 index modes, 5, 6; 0 and 2 with BC7_USE_3SUBSETS): pre -> bin -> search -> post,
//             described further down. The per-mode winners travel through a 24 B slot per mode per block.
//             pre drops the candidates that provably cannot win (subset_lower_bound, bc7_core.h); the modes run in
//             an order chosen for that pruning (launch_bc7_encode_many), modes 4 / 5 / 6 possibly in two phases.
//   pick    : lane = block; minimum over the per-mode winners in the reference's evaluation order.
#include "dxtex_kernels.h"
#include "bc67_tables.h"
#include "bc7_core.h"
#include "search_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace dxtex
{
namespace
{
using namespace bc7;

struct Cand { uint32_t err; uint32_t ord; uint64_t lo, hi; };   // ord = evaluation order inside D3DX_BC7::Encode

enum : int { SLOT_M0 = 0, SLOT_M1, SLOT_M2, SLOT_M3, SLOT_M4A, SLOT_M4B, SLOT_M5, SLOT_M6, SLOT_M7, NUM_SLOTS };
enum : int { LIST_BYTES = 64 };   // per block: [0..15] 3-bit list, [16..31] 2-bit list, [32] hasAlpha, [33..36] mode-0 list, [37] mode 6 first, [40..55] mode-2 list,
                                  // [56..59] best 3-bit rough error (int; rough -> flag_count)

enum : int { PHASE_ALL = 0, PHASE_EARLY = 1, PHASE_LATE = 2 };

struct Bc7Args
{
    SegTable seg;            // the images behind this pass (search_common.h)
    uint32_t nblocks;        // blocks in this pass; scratch arrays are indexed by pass-local block number
    uint32_t flags;
    uint8_t* lists;
    Cand* cands;
    uint32_t* px;            // nblocks x 16 packed RGBA8 texels (D3DX_BC7::Encode's aLDRPixels, :2792-2799)
    struct TaskRec* recs;    // per-mode task records (reused by every mode)
    uint2* order;            // live tasks of the current mode, sorted by subset size: (task, tinfo)
    uint32_t* tinfo;         // per task: texel mask | rotation << 16 | subset size << 24 (size 0 = no search needed)
    uint32_t* counters;      // 35 words, see bc7_bin_* kernels
    uint32_t* zeroOrd;       // per block: evaluation-order key of the first candidate (in Encode's order) known to reach error 0;
                             // Encode() returns there (:2803, :2835, :2845), so later candidates are never looked at
    uint2* seeds1;           // per block: the whole-block RGB fit (modes 4, 5) and RGBA fit (mode 6), :3541-3568
    uint2* seeds3;           // BC7_USE_3SUBSETS only: per block 64 shapes x 3 subsets (modes 0 and 2)
    uint2* seeds;            // per block 2 lists (3-bit, 2-bit rough error) x 16 ranked shapes x 2 subsets: the float-fit endpoints RoughMSE
                             // derives (:3526-3552) for the shapes Refine will look at (the fits of the other shapes are never read again)
    int* bestErr;            // per block: smallest error an already finished mode reached (subset_lower_bound prunes against it)
    int prune;               // 0 = search every candidate like the reference does (DXTEX_BC7_NO_PRUNE, for A/B runs)
    const uint32_t* flagged; // [0] = blocks of this pass flagged for an early mode 6, [1] = blocks with alpha (bc7_flag_count_kernel)
    uint32_t early6Min;      // the early phase of mode 6 only exists when at least this many blocks would be in it
    uint32_t earlyAlphaMin;  // ... and the early phases of modes 4 / 5 (blocks with alpha) when at least this many
    int early6Pct;           // rough kernel: mode 6 goes first where 100 * lower bound <= early6Pct * best 3-bit rough error
    int phase;               // which blocks this launch of a mode owns: PHASE_ALL, or the early / late half of a split mode
    const uint32_t* gate;    // a mode (or the early phase of a split mode) that owns no block of the pass - mode 7 on an opaque image, the early phases
    uint32_t gateMin;        // when too few blocks are flagged - is skipped on the device: its kernels return at once when *gate < gateMin (nullptr: always run)
    uint32_t perturbWaveMax; // whole-block modes: lists of at most this many live tasks are searched by bc7_perturb_wave_kernel (0 = never)
    uint32_t exhWaveMax;     // ... and by bc7_exhaustive_wave_kernel
};

// Whole-block tasks (modes 4, 5, 6: one subset of 16 texels) on SHORT lists - a small image, the late phase of mode 6 - are searched by
// groups of lanes instead of a lane each: with fewer tasks than lanes a kernel is as slow as its longest serial chain, and a chain of
// PerturbOne / Exhaustive evaluations on one lane is 0.1 - 0.5 ms. Exhaustive: a wavefront per task (the <= 121 candidates of a window
// on its 64 lanes). PerturbOne: half a wavefront per task (16 texels x the step's two candidates).
// The live count is known on the device only: the launcher passes each kernel pair the limit below which the group kernel works and the
// lane-per-task kernel stands down (0 when the group kernel was not launched because the list cannot be short). Measured on MI355X:
// mode 6's Exhaustive pays up to 2^18 tasks (its sixteen-entry evaluations are dear and its lists are pruned hard); modes 4 / 5 break
// even near 25 K tasks; the half-wave PerturbOne near 32 K (2 x 8192 tasks in flight).
constexpr uint32_t kWaveTaskMax = 262144;
constexpr uint32_t kWaveTaskMax45 = 24576;
constexpr uint32_t kPerturbWaveMax = 32768;

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

// Compiled for 8 wavefronts per SIMD: the kernel waits on LDS reads inside divergent Newton loops, and more waves hide that better
// than the 168 bytes of spill per lane cost (14.9 -> 14.0 ms per 4096^2 image; 6 / 7 waves: 14.4 / 14.2 ms).
#if !defined(DXTEX_ROUGH_WGS)
#define DXTEX_ROUGH_WGS 8
#endif
__global__ void __launch_bounds__(256, DXTEX_ROUGH_WGS) bc7_rough_kernel(Bc7Args a)
{
    __shared__ float sF[4][64];
    __shared__ uint32_t sL[4][16];
    __shared__ uint2 sSeed[4][128];    // the fits of all 64 shapes x 2 subsets; only the ranked ones leave the kernel
    __shared__ int2 sErr[4][128];      // their rough errors with 3-bit / 2-bit indices
    __shared__ uint32_t sOpaque[4];    // every alpha of the block is exactly 1.0f: the fits take the A1 variant (bc7_core.h, fit_setup)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nb = blockIdx.x * 4 + wave;
    const bool inRange = nb < a.nblocks;

    if (inRange && lane < 16)
    {
        uint32_t ldr;
        const BcSeg& sg = seg_of(a.seg, nb);
        load_block_texel(sg.src, sg.nbw, sg.nb0 + (nb - sg.l0), lane, &sF[wave][lane * 4], ldr);
        sL[wave][lane] = ldr;
        a.px[uint64_t(nb) * 16 + lane] = ldr;
        if (lane == 0) { a.zeroOrd[nb] = 0xFFFFFFFFu; a.bestErr[nb] = 0x7FFFFFFF; }
    }
    {
        const bool one = !inRange || lane >= 16 || sF[wave][(lane & 15) * 4 + 3] == 1.0f;       // own write, same lane
        const unsigned long long all = __ballot(one);
        if (lane == 0) sOpaque[wave] = (all == ~0ull) ? 1u : 0u;
    }
    __syncthreads();

    // ---- 2-subset shapes: modes 1, 3, 7 ----
    // The 4 x 128 fits of the workgroup's four blocks are dealt to the wavefronts BY SUBSET SIZE (kFit2Order: 128 subsets, largest
    // first, in eight groups of sixteen): a wavefront takes one group for all four blocks (lane = block * 16 + entry), twice - group
    // w, then group 7 - w. The texel loops of a wavefront then run max-of-group trips (13, 12, 8, 8, 8, 8, 8, 4: 69 per four blocks)
    // instead of max(larger subset) + max(smaller subset) = 13 + 8 per block with a lane per shape: the same fits, 18 % fewer trips.
    // Every fit is independent and the error sums are integers, so who computes what is free.
    {
        const uint32_t blk = uint32_t(lane) >> 4, ent = uint32_t(lane) & 15u;
        const uint32_t nbk = blockIdx.x * 4 + blk;
        const float* fpxk = sF[blk];
        const uint32_t* pixk = sL[blk];
        const bool opaque = sOpaque[blk] != 0u;
#pragma unroll 1
        for (int k = 0; k < 2; ++k)
        {
            const uint32_t group = k ? uint32_t(7 - wave) : uint32_t(wave);
            const uint32_t code = kFit2Order[group * 16 + ent];
            if (nbk < a.nblocks)
            {
                const uint32_t m1 = kPart2Mask[code >> 1];
                const uint32_t m = (code & 1u) ? m1 : ((~m1) & 0xFFFFu);
                Region rg; region_init(rg, pixk, m);
                uint32_t A, B;
                if (rg.np == 1) { A = pixk[rg.pos(0)]; B = A; }
                else if (rg.np == 2) { A = pixk[rg.pos(0)]; B = pixk[rg.pos(1)]; }
                else if (opaque) seed_endpoints<true, false, true>(fpxk, m, A, B);
                else seed_endpoints<true>(fpxk, m, A, B);
                sSeed[blk][code] = make_uint2(A, B);                              // Refine starts from the same fit (:3411-3417)
#if defined(DXTEX_ROUGH_SPLIT_ERR)
                sErr[blk][code] = make_int2(rough_error<3, 0>(rg, A, B), rough_error<2, 0>(rg, A, B));
#else
                int e3, e2;
                rough_error_3_2(rg, A, B, e3, e2);
                sErr[blk][code] = make_int2(e3, e2);
#endif
            }
        }
    }
    __syncthreads();
    if (!inRange) return;              // whole wave exits together
    const float* fpx = sF[wave];
    const uint32_t* pix = sL[wave];

    const bool alphaLane = (lane < 16) && ((pix[lane & 15] >> 24) != 0xFFu);
    const bool hasAlpha = __ballot(alphaLane) != 0ull;

    uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;

    {
        const uint32_t shape = lane;
        const int2 er0 = sErr[wave][shape * 2], er1 = sErr[wave][shape * 2 + 1];
        const int e3 = er0.x + er1.x, e2 = er0.y + er1.y;
        int ea = e3, eb = e2;
        uint32_t sa = shape, sb = shape;
        for (int i = 0; i < 16; ++i)
        {
            selection_pass(ea, sa, lane, i);
            selection_pass(eb, sb, lane, i);
        }
        if (lane < 16) { lst[lane] = uint8_t(sa); lst[16 + lane] = uint8_t(sb); }
        {
            // lanes 0-15: the shapes ranked by 3-bit error (mode 1), lanes 16-31: by 2-bit error (modes 3, 7); 16 bytes per lane, 512 per block
            wave_lds_sync();
            uint32_t ranked2 = uint32_t(__shfl(int(sb), lane & 15));
            asm volatile("" : "+v"(ranked2));      // keeps the exchange out of the lanes >= 16 branch, where its source lanes would be off
            const uint32_t mine = (lane < 16) ? sa : ranked2;
            if (lane < 32)
            {
                const uint2 s0 = sSeed[wave][mine * 2], s1 = sSeed[wave][mine * 2 + 1];
                reinterpret_cast<uint4*>(a.seeds + uint64_t(nb) * 64)[lane] = make_uint4(s0.x, s0.y, s1.x, s1.y);
            }
        }
        if (lane == 0)
        {
            lst[32] = hasAlpha ? 1 : 0;
            // Scheduling hint, not a result: mode 6 (one subset, RGBA on one line) runs BEFORE the two-subset modes for blocks
            // where it has a chance against them - its lower bound does not exceed the best 3-bit rough error - so that its
            // result can prune their candidates; elsewhere it runs last and is mostly pruned itself. See launch order below.
            // (decided by bc7_flag_count_kernel, a lane per block: on one lane of this wavefront the bound's fp64 arithmetic cost 1.2 of the
            // kernel's 11.8 ms per 4096^2 image; what it needs from here is the best 3-bit rough error)
            *reinterpret_cast<int*>(lst + 56) = ea;
        }
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
            a.seeds3[uint64_t(nb) * 192 + shape * 3 + r] = make_uint2(A, B);
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

// ---- per-mode search: pre -> bin/scatter -> search -> post -----------------------------------------------------------
// A *task* is one subset of one candidate (mode, shape | rotation, index mode) of one block: the unit
// OptimizeEndPoints (:3113-3136) hands to OptimizeOne. Tasks of one mode are numbered
//     t = block * TPB + rank * G + subset         (G = lanes per candidate: 1, 2 or 4; rank = candidate number)
//   pre     (lane = task, natural order): seed -> Quantize -> FixEndpointPBits -> AssignIndices; the "org" endpoints,
//           their error and the subset size go to the task record. Tasks that need no search (error already 0,
//           block outside the image, mode 7 on an opaque block) get size 0.
//   bin     counting sort of the live tasks by subset size (17 bins, wave-aggregated atomics), so that a search
//           wavefront only ever sees subsets of (almost) one size and its texel loops have one trip count.
//   search  persistent wavefronts pulling from the sorted list; OptimizeOne cut into lockstep pieces (bc7_core.h):
//           one kernel of PERTURB macro-ops, then one of flattened Exhaustive windows; the colour and alpha
//           channels of the separate-alpha modes get kernels of their own. Endpoints and error of a task travel
//           through its record between kernels; the subset's texels sit in a per-lane LDS column.
//   post    (lane = task, natural order): FixEndpointPBits + AssignIndices of the optimised endpoints, org-vs-opt
//           decision over the subsets of a candidate (lane shuffles), first minimum over the block's candidates
//           (butterfly), EmitBlock by the winning lane -> per-mode candidate slot.
struct TaskRec { uint32_t A, B; int err; int orgErr; };      // 16 bytes: the search's start and then its result; orgErr = error of the unoptimised endpoints
                                                             // (pre -> post, the search kernels leave it alone)

template<int MODE, int IM> struct TaskMap
{
    typedef ModeInfo<MODE> MI;
    enum : int { NS = MI::NS,
                 G = (NS == 1) ? 1 : (NS == 2) ? 2 : 4,
                 RANKS = (NS == 1) ? ((MODE == 6) ? 1 : 4) : ((MODE == 0) ? 4 : 16),     // max(1, shapes >> 2) or rotations
                 TPB = RANKS * G,
                 LIST = (MODE == 1) ? 0 : (MODE == 0) ? 33 : (MODE == 2) ? 40 : 16,
                 SLOT = (MODE == 0) ? SLOT_M0 : (MODE == 1) ? SLOT_M1 : (MODE == 2) ? SLOT_M2 : (MODE == 3) ? SLOT_M3 :
                        (MODE == 4) ? (IM ? SLOT_M4B : SLOT_M4A) : (MODE == 5) ? SLOT_M5 : (MODE == 6) ? SLOT_M6 : SLOT_M7 };
};

// Position of a candidate in D3DX_BC7::Encode's loop nest: modes ascending, then rotation, index mode, shape rank (:2805-2886)
template<int MODE, int IM>
__device__ __forceinline__ uint32_t candidate_sub(uint32_t rank)
{
    return (TaskMap<MODE, IM>::NS == 1) ? ((MODE == 4) ? (rank * 2 + IM) * 16u : rank * 16u) : rank;
}
template<int MODE, int IM>
__device__ __forceinline__ uint32_t candidate_ord(uint32_t rank) { return uint32_t(MODE) * 128u + candidate_sub<MODE, IM>(rank); }

// Modes 4, 5 and 6 are launched twice when the rough pass has run: early (before the two-subset modes) for the blocks they are
// likely to win - blocks with alpha for 4 / 5, blocks flagged by the rough kernel for 6 - and late for the others. A launch
// must not touch the candidate slot of a block it does not own.
template<int MODE>
__device__ __forceinline__ bool phase_owns(const Bc7Args& a, uint32_t nb)
{
    if (a.phase == PHASE_ALL || (MODE != 4 && MODE != 5 && MODE != 6)) return true;
    const uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;
    // A handful of flagged blocks is not worth a phase of its own: the search kernels of a nearly empty phase still take as long
    // as their longest task (milliseconds). Below the threshold the flagged blocks simply stay with the late phase.
    const bool early = (MODE == 6) ? (lst[37] != 0 && a.flagged[0] >= a.early6Min) : (lst[32] != 0 && a.flagged[1] >= a.earlyAlphaMin);
    return early == (a.phase == PHASE_EARLY);
}

__device__ __forceinline__ bool lst_has_alpha(const Bc7Args& a, uint32_t nb) { return a.lists[uint64_t(nb) * LIST_BYTES + 32] != 0; }

// Texel mask, anchor and rotation of task `r` (= t % TPB) of block `nb`; false if the task does not exist.
template<int MODE, int IM>
__device__ __forceinline__ bool task_geometry(const Bc7Args& a, uint32_t nb, uint32_t r, uint32_t& shape, uint32_t& mask, uint32_t& anchor, uint32_t& rot)
{
    typedef TaskMap<MODE, IM> TM;
    shape = 0; mask = 0xFFFFu; anchor = 0; rot = 0;
    if (nb >= a.nblocks) return false;
    if (!phase_owns<MODE>(a, nb)) return false;
    const uint32_t rank = r / TM::G, region = r % TM::G;
    // fMSEBest == 0: the reference returns from Encode, i.e. skips every candidate that comes later in ITS order. The modes
    // may run in any order here, so the comparison is on the evaluation-order key, not on "some mode is done".
    if (a.zeroOrd[nb] < candidate_ord<MODE, IM>(rank)) return false;
    if (TM::NS == 1) { rot = (MODE == 6) ? 0u : r; return true; }
    if (region >= uint32_t(TM::NS)) return false;
    const uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;
    if (MODE == 7 && lst[32] == 0) return false;
    shape = lst[TM::LIST + rank];
    if (TM::NS == 2)
    {
        const uint32_t m1 = kPart2Mask[shape];
        mask = region ? m1 : ((~m1) & 0xFFFFu);
        anchor = region ? uint32_t(kAnchor2[shape]) : 0u;
    }
    else
    {
        const uint32_t bits = kPart3Bits[shape];
        mask = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (((bits >> (2 * i)) & 3u) == region) mask |= 1u << i;
        anchor = (region == 0) ? 0u : (region == 1) ? uint32_t(kAnchor3[shape] & 15) : uint32_t(kAnchor3[shape] >> 4);
    }
    return true;
}

// Refine's first half for one task, from the block's float + 8-bit texels (LDS or registers).
// ASSIGN = true (pre): the unoptimised endpoints with their anchor fix-up and their error, no indices (assign_error). ASSIGN = false (post): the
// endpoints before AssignIndices only (Quantize + FixEndpointPBits) - post takes their error from the task record and only the block's winner,
// if it stands with them, runs AssignIndices for their indices.
template<int MODE, int IM, bool ASSIGN = true>
__device__ __forceinline__ void task_org(const float* fpx, const uint32_t* pix, uint32_t mask, uint32_t anchor, uint32_t rot,
                                         SubsetResult& res, int& np, bool wantRegion, Region& rgOut, Block16& b16Out, const uint2* seed = nullptr)
{
    typedef TaskMap<MODE, IM> TM;
    if (TM::NS == 1)
    {
        block16_init(b16Out, pix, rot);
        uint32_t A, B;
        if (seed) { const uint2 sd = seed[(MODE == 6) ? 1 : 0]; A = sd.x; B = sd.y; }     // bc7_block_seeds_kernel fitted the block once
        else if (MODE == 6) seed_endpoints<true>(fpx, 0xFFFFu, A, B);
        else seed_endpoints<false>(fpx, 0xFFFFu, A, B);
        if (MODE != 6)
        {
            // colour endpoints from the *unrotated* float texels, alpha endpoints = min/max of the rotated 8-bit alpha (:3552-3568)
            uint32_t mn = 255, mx = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const uint32_t al = b16Out.px[i] >> 24; mn = min(mn, al); mx = max(mx, al); }
            A = (A & 0x00FFFFFFu) | (mn << 24);
            B = (B & 0x00FFFFFFu) | (mx << 24);
        }
        if (ASSIGN) refine_pre_err<MODE, IM>(b16Out, A, B, 0u, res);
        else fix_pbits<MODE>(quantize_endpoint<MODE>(A), quantize_endpoint<MODE>(B), res.orgA, res.orgB);
        np = 16;
    }
    else
    {
        region_init(rgOut, pix, mask);
        uint32_t A, B;
        if (seed) { const uint2 sd = *seed; A = sd.x; B = sd.y; }          // the rough pass already fitted this subset
        else if (rgOut.np == 1) { A = pix[rgOut.pos(0)]; B = A; }
        else if (rgOut.np == 2) { A = pix[rgOut.pos(0)]; B = pix[rgOut.pos(1)]; }
        else seed_endpoints<true>(fpx, mask, A, B);
        if (ASSIGN) refine_pre_err<MODE, IM>(rgOut, A, B, anchor, res);
        else fix_pbits<MODE>(quantize_endpoint<MODE>(A), quantize_endpoint<MODE>(B), res.orgA, res.orgB);
        np = rgOut.np;
    }
}

// pre / post need no float texels - every fit they start from is stored by the rough pass or bc7_block_seeds_kernel - and take
// the 8-bit texels of their blocks from the pass's scratch.
template<int BPW>
__device__ __forceinline__ void stage_packed(const Bc7Args& a, uint32_t nbFirst, int lane, uint32_t* sL)
{
    for (int t = lane; t < BPW * 16; t += 64)
    {
        const uint32_t nb = nbFirst + (uint32_t(t) >> 4);
        sL[t] = (nb < a.nblocks) ? a.px[uint64_t(nb) * 16 + (t & 15)] : 0u;
    }
    wave_lds_sync();
}

#if !defined(DXTEX_PP45_WGS)
#define DXTEX_PP45_WGS 1               // workgroups per CU the pre / post kernels of modes 4 / 5 are compiled for (1 = unconstrained: 200 - 256 registers, two waves per SIMD)
#endif
template<int MODE, int IM>
__global__ void __launch_bounds__(256, (MODE == 4 || MODE == 5) ? DXTEX_PP45_WGS : 1) bc7_pre_kernel(Bc7Args a)
{
    typedef TaskMap<MODE, IM> TM;
    constexpr int BPW = (TM::TPB >= 64) ? 1 : 64 / TM::TPB;       // blocks per wavefront
    __shared__ uint32_t sL[4][BPW * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbFirst = (blockIdx.x * 4 + wave) * BPW;
    if (nbFirst >= a.nblocks) return;
    if (a.gate && *a.gate < a.gateMin) return;          // this launch owns no block: no task list at all (the bin kernels stand down with it)
    const uint32_t blk = uint32_t(lane) / TM::TPB, r = uint32_t(lane) % TM::TPB;
    const uint32_t nb = nbFirst + blk;
    uint32_t shape, mask, anchor, rot;
    const bool active = (blk < uint32_t(BPW)) && task_geometry<MODE, IM>(a, nb, r, shape, mask, anchor, rot);
    TaskRec rec; rec.A = 0; rec.B = 0; rec.err = 0; rec.orgErr = 0;
    uint32_t npLive = 0;               // subset size if the task is to be searched, else 0
    if (!__any(active))
    {
        // nothing to do for these blocks (another phase owns them, mode 7 on opaque blocks, Encode already returned): the task
        // list still needs their size-0 entries
        if (nb < a.nblocks && blk < uint32_t(BPW))
        {
            a.recs[uint64_t(nb) * TM::TPB + r] = rec;
            a.tinfo[uint64_t(nb) * TM::TPB + r] = 0;
        }
        return;
    }
    stage_packed<BPW>(a, nbFirst, lane, sL[wave]);
    int lb = 0;
    if (active)
    {
        SubsetResult res; int np; Region rg; Block16 b16;
        task_org<MODE, IM>(nullptr, &sL[wave][blk * 16], mask, anchor, rot, res, np, true, rg, b16,
                           (TM::NS == 2) ? a.seeds + uint64_t(nb) * 64 + (TM::LIST ? 32 : 0) + r
                                         : (TM::NS == 1) ? a.seeds1 + uint64_t(nb) * 2 : a.seeds3 + uint64_t(nb) * 192 + shape * 3 + (r % TM::G));
        rec.A = res.orgA; rec.B = res.orgB; rec.err = res.orgErr; rec.orgErr = res.orgErr;
        npLive = (res.orgErr != 0) ? uint32_t(np) : 0u;        // error 0: OptimizeOne cannot move the endpoints
        if (a.prune && npLive)
        {
            lb = subset_lower_bound(&sL[wave][blk * 16], mask, rot, (MODE >= 6) ? 4 : 3);
            if (MODE == 4 || MODE == 5)
            {
                // the scalar slot: 3-bit indices for mode 4 with index mode 0, 2-bit otherwise
                if (rot != 0 || lst_has_alpha(a, nb))
                    lb += scalar_kmeans_lower_bound<(MODE == 4 && IM == 0) ? 8 : 4>(&sL[wave][blk * 16], rot);
            }
            if (MODE < 4)
            {
                // colour-only modes decode alpha as 255 (Unquantize, :841), whatever the indices: that part of the error is exact
                const uint32_t* pix = &sL[wave][blk * 16];
#pragma unroll
                for (int i = 0; i < 16; ++i) if ((mask >> i) & 1u) { const int da = 255 - int(pix[i] >> 24); lb += da * da; }
            }
        }
    }
    if (a.prune)
    {
        // Exact pruning (bc7_core.h, subset_lower_bound): a candidate whose lower bound exceeds an error that is already on the
        // table for this block - the unoptimised error of another candidate of this mode, or the result of a finished mode -
        // cannot become Encode's first minimum; its subsets keep their seed endpoints and are not searched.
        int candOrg = active ? rec.err : 0, candLb = lb;
#pragma unroll
        for (int d = 1; d < TM::G; d <<= 1) { candOrg += __shfl_xor(candOrg, d); candLb += __shfl_xor(candLb, d); }
        int table = active ? candOrg : 0x7FFFFFFF;
#pragma unroll
        for (int d = TM::G; d < TM::TPB && d < 64; d <<= 1) table = min(table, __shfl_xor(table, d));
        if (nb < a.nblocks && blk < uint32_t(BPW)) table = min(table, a.bestErr[nb]);
        if (candLb > table) npLive = 0;
    }
    if (nb < a.nblocks && blk < uint32_t(BPW))
    {
        a.recs[uint64_t(nb) * TM::TPB + r] = rec;
        a.tinfo[uint64_t(nb) * TM::TPB + r] = (mask & 0xFFFFu) | (rot << 16) | (npLive << 24);
    }
}

// A search lane picks up a task: copies the subset's texels (rotated for modes 4, 5) to its LDS column.
template<int MODE, int IM>
__device__ __forceinline__ void search_pickup(const Bc7Args& a, uint2 task, uint32_t* slotCol, SlotRegion& rg)
{
    typedef TaskMap<MODE, IM> TM;
    const uint32_t nb = task.x / TM::TPB;
    const uint32_t mask = task.y & 0xFFFFu, rot = (task.y >> 16) & 3u;
    const uint4* px4 = reinterpret_cast<const uint4*>(a.px + uint64_t(nb) * 16);
    const uint4 q0 = px4[0], q1 = px4[1], q2 = px4[2], q3 = px4[3];
    const uint32_t px[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
    int np = 0;
    if (TM::NS == 1)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) slotCol[i * kSlotStride] = rotate_pixel(px[i], rot);
        np = 16;
    }
    else
    {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((mask >> i) & 1u) { slotCol[np * kSlotStride] = px[i]; ++np; }
    }
    rg.base = slotCol; rg.np = np; rg.p2sum = 0;
}

// Set bits of a ballot below this lane (v_mbcnt_lo / v_mbcnt_hi).
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return int(__builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)));
}

#if defined(DXTEX_BC7_WAVETIMES)
// Development instrumentation (-DDXTEX_BC7_WAVETIMES, never in the product): per wavefront of every lane-per-task search kernel outside the
// early phases - start, the moment it first saw the queue drained, end (s_memrealtime, 100 MHz), rounds, busy lanes summed over the rounds,
// tasks taken - and a histogram of rounds per finished task; summarised on stderr after the submission (bc7_wavetimes_report).
enum : int { WT_KINDS = 3, WT_KEYS = WT_KINDS * 8 * 2 * 3, WT_FIELDS = 6 };
__device__ unsigned long long g_wt[WT_KEYS][8192][WT_FIELDS];
__device__ unsigned int g_wtHist[WT_KEYS][256];
#define WT_BEGIN(KIND) const int wtKey = (((KIND) * 8 + MODE) * 2 + IM) * 3 + CHSET; const bool wtOn = a.phase != PHASE_EARLY; \
    const unsigned long long wt0 = wall_clock64(); unsigned long long wtDr = 0, wtRounds = 0, wtLanes = 0, wtTasks = 0; int wtMine = 0
#define WT_TAKE(IDX) do { wtTasks += (unsigned long long)__popcll(__ballot((IDX) != 0xFFFFFFFFu)); if (!wtDr && q.drained) wtDr = wall_clock64(); } while (0)
#define WT_ROUND(BUSYL) do { ++wtRounds; wtLanes += (unsigned long long)__popcll(__ballot(BUSYL)); if (BUSYL) ++wtMine; } while (0)
#define WT_DONE() do { wtMine = 0; } while (0)
#define WT_END() do { if (wtOn && lane == 0 && wtRounds) { unsigned long long* w_ = g_wt[wtKey][blockIdx.x & 8191]; w_[0] = wt0; w_[1] = wtDr ? wtDr : wall_clock64(); \
    w_[2] = wall_clock64(); w_[3] = wtRounds; w_[4] = wtLanes; w_[5] = wtTasks; } } while (0)
#else
#define WT_BEGIN(KIND) do { } while (0)
#define WT_TAKE(IDX) do { } while (0)
#define WT_ROUND(BUSYL) do { } while (0)
#define WT_DONE() do { } while (0)
#define WT_END() do { } while (0)
#endif
// Which search kernels look for PerturbOne calls that cannot change anything (bc7_core.h, flat_call): mode 6, whose alpha channel is such a
// channel in every opaque block (2.49 -> 2.30 ms on the benchmark image). Measured and left out: modes 4 / 5 - their colour endpoints start from
// the fit of the UNROTATED texels (:3552-3568), so the swapped-in constant of rotations 1 - 3 is not exact when its calls come, and the test
// costs what the few later calls save; two- and three-subset regions seldom have a constant channel at all.
#if !defined(DXTEX_FLAT_SKIP)
#define DXTEX_FLAT_SKIP 1
#endif
template<int MODE, int IM, int CHSET>
constexpr bool kFlatSkip = DXTEX_FLAT_SKIP != 0 && MODE == 6 && CHSET == CH_ALL;

// The PERTURB phase of OptimizeOne (:3060-3105) for the channels of CHSET, over every live task of the mode.
template<int MODE, int IM, int CHSET>
__global__ void __launch_bounds__(64) bc7_perturb_kernel(Bc7Args a, int loop)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    __shared__ uint32_t sSlot[16 * kSlotStride];
    const int lane = threadIdx.x;
    const uint32_t live = a.counters[34];
    if (live == 0) return;           // nothing survived pre (a phase that owns no block, everything pruned): skip the queue atomics
    if (TaskMap<MODE, IM>::NS == 1 && live <= a.perturbWaveMax) return;      // short list of whole-block tasks: bc7_perturb_wave_kernel's turn
    uint32_t* head = a.counters + kQueueBase + loop;
    uint32_t* slotCol = &sSlot[lane];
    WT_BEGIN(0);

    PerturbState st = perturb_begin<MODE, IM, CHSET>(0, 0, 0);
    SlotRegion rg; rg.base = slotCol; rg.np = 0; rg.p2sum = 0;
    int base = 0;
    uint32_t myTask = 0xFFFFFFFFu;
    uint32_t flatMask = 0, flatVals = 0;            // whole-block modes: the texel channels that are constant over the block (flat_channels)
    WaveQueue q; q.lo = q.hi = 0; q.drained = false;
    for (;;)
    {
        const unsigned long long idle = __ballot(myTask == 0xFFFFFFFFu);
        if (idle && !(q.drained && q.lo >= q.hi))
        {
            const uint32_t idx = queue_take(q, head, live, idle, lane);
            WT_TAKE(idx);
            if (idx != 0xFFFFFFFFu)
            {
                const uint2 task = a.order[idx];
                myTask = task.x;
                const TaskRec r = a.recs[myTask];
                search_pickup<MODE, IM>(a, task, slotCol, rg);
                st = perturb_begin<MODE, IM, CHSET>(r.A, r.B, r.err);
                int other;
                base = loop_base<MODE, IM, CHSET>(rg, st.optA, st.optB, &other);
                if (loop_is_settled<CHSET>(r.err, other)) myTask = 0xFFFFFFFFu;     // scalar slot already exact: the record stays as it is
                if constexpr (kFlatSkip<MODE, IM, CHSET>) flatMask = flat_channels(rg, flatVals);
            }
        }
        // PerturbOne calls on a channel that is constant over the block and already exact cannot change anything (bc7_core.h, flat_call): the
        // lane's state moves past them without the 2 PREC - 1 evaluations - every call on alpha of an opaque block in mode 6 (kFlatSkip: modes
        // 4 / 5 were measured and left out, see above). One nearest-entry evaluation (for all lanes of the wavefront) decides.
        if constexpr (kFlatSkip<MODE, IM, CHSET>)
        {
            const bool fc = myTask != 0xFFFFFFFFu && flat_call<MODE, IM, CHSET>(st, flatMask, flatVals);
            if (__any(fc))
            {
                const int nearest = eval_nearest<MODE, IM, CHSET>(rg, st, base);
                if (fc && nearest == st.optErr)
                {
                    st = skip_flat_calls<MODE, IM, CHSET>(st);
                    if (st.ch >= C::CH1)
                    {
                        TaskRec* r = a.recs + myTask;
                        r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                        myTask = 0xFFFFFFFFu;
                        WT_DONE();
                    }
                }
            }
        }
        if (__ballot(myTask != 0xFFFFFFFFu) == 0ull)
        {
            if (q.drained && q.lo >= q.hi) break;
            continue;
        }
        WT_ROUND(myTask != 0xFFFFFFFFu);
        if (myTask != 0xFFFFFFFFu)
        {
            int e; uint32_t v;
            perturb_macro<MODE, IM, CHSET>(rg, st, base, e, v);
            st = perturb_transition<MODE, IM, CHSET>(st, e, v);
            if (st.ch >= C::CH1)
            {
                TaskRec* r = a.recs + myTask;
                r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                myTask = 0xFFFFFFFFu;
                WT_DONE();
            }
        }
    }
    WT_END();
}

// PerturbOne (:2926-2966) for SHORT lists of whole-block tasks, half a wavefront per task: lane (cand, k) of a half scores texel k
// against candidate `cur - step` (cand 0) or `cur + step` (cand 1); the sixteen per-texel results are added up inside the group
// (integers: any order) and every lane of the half takes the same decisions from the two totals - perturb_macro's candidates and
// decisions, a chain of 2 * PREC - 1 single-texel evaluations instead of that many sixteen-texel ones. Two tasks per wavefront.
template<int MODE, int IM, int CHSET>
__global__ void __launch_bounds__(64) bc7_perturb_wave_kernel(Bc7Args a)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    typedef TaskMap<MODE, IM> TM;
    static_assert(TM::NS == 1, "whole-block tasks only");
    __shared__ uint32_t sTex[16 * kSlotStride];              // column `slot` holds the slot's sixteen texels (SlotRegion's layout)
    const int lane = threadIdx.x, slot = lane >> 5, k = lane & 15, cand = (lane >> 4) & 1;
    const uint32_t live = a.counters[34];
    if (live == 0 || live > a.perturbWaveMax) return;
    for (uint32_t idx0 = blockIdx.x * 2u; idx0 < live; idx0 += gridDim.x * 2u)
    {
        const uint32_t idx = min(idx0 + uint32_t(slot), live - 1u);      // an odd tail: the second half shadows the last task and does not store
        const bool mine = (idx0 + uint32_t(slot)) < live;
        const uint2 task = a.order[idx];
        const TaskRec rec = a.recs[task.x];
        SlotRegion rg;
        wave_lds_sync();                                  // the previous tasks' reads are done
        search_pickup<MODE, IM>(a, task, &sTex[slot], rg);      // every lane of the half writes the same 16 texels
        wave_lds_sync();
        int other;
        const int base = loop_base<MODE, IM, CHSET>(rg, rec.A, rec.B, &other);
        PerturbState st = perturb_begin<MODE, IM, CHSET>(rec.A, rec.B, rec.err);
        if (loop_is_settled<CHSET>(rec.err, other)) st.ch = C::CH1;       // scalar slot already exact: the record stays as it is
        SlotRegion one; one.base = &sTex[slot] + k * kSlotStride; one.np = 1; one.p2sum = 0;      // this lane's texel
        // sum over the sixteen texel lanes of a candidate group
        auto group_sum = [](int v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; };
        while (__ballot(st.ch < C::CH1) != 0ull)
        {
            const bool busy = st.ch < C::CH1;
            const int ch = busy ? st.ch : int(C::CH0);
            VarPal<C::N> vp;
            varpal_init<MODE, IM, CHSET>(vp, st.optA, st.optB, ch);
            const uint32_t fixedU = unq1<C::PREC>(byte_of(st.do_b ? st.optA : st.optB, ch));
            int cur = int(byte_of(st.do_b ? st.optB : st.optA, ch));
            int minErr = st.optErr;
            {
                // the first step is half the range: one legal candidate (perturb_macro), both groups score it
                constexpr int half = 1 << (C::PREC - 1);
                const int tmp = (cur >= half) ? cur - half : cur + half;
                const uint32_t u = unq1<C::PREC>(uint32_t(tmp) & ((1u << C::PREC) - 1u));
                const int e = base + group_sum(eval_var<MODE, IM, CHSET>(one, vp, ch, st.do_b ? fixedU : u, st.do_b ? u : fixedU, 0));
                if (e < minErr) { minErr = e; cur = tmp; }
            }
#pragma unroll 1
            for (int step = 1 << (C::PREC - 2); step; step >>= 1)
            {
                const int tmp = cur + (cand ? step : -step);
                const uint32_t u = unq1<C::PREC>(uint32_t(tmp) & ((1u << C::PREC) - 1u));
                const int tot = base + group_sum(eval_var<MODE, IM, CHSET>(one, vp, ch, st.do_b ? fixedU : u, st.do_b ? u : fixedU, 0));
                const int eMinus = __shfl(tot, slot * 32), ePlus = __shfl(tot, slot * 32 + 16);
                const int tMinus = cur - step, tPlus = cur + step;
                int beststep = 0;
                if (tMinus >= 0 && tMinus < (1 << C::PREC) && eMinus < minErr) { minErr = eMinus; beststep = -step; }
                if (tPlus >= 0 && tPlus < (1 << C::PREC) && ePlus < minErr) { minErr = ePlus; beststep = step; }
                cur += beststep;
            }
            if (busy) st = perturb_transition<MODE, IM, CHSET>(st, minErr, uint32_t(cur));
        }
        if (mine && (lane & 31) == 0 && !loop_is_settled<CHSET>(rec.err, other))
        {
            TaskRec* r = a.recs + task.x;
            r->A = st.optA; r->B = st.optB; r->err = st.optErr;
        }
    }
}

// PerturbOne through the bound filter (see eval_var_bound). A step of the logarithmic search has to be decided before the next one
// starts, so the filter cannot postpone a lane's exact evaluations the way Exhaustive's windows do; what it can do is make them rare
// and share them. Per candidate every lane derives the palette and takes the bound (no second dot product, no compare-select scan);
// the candidates that could still beat the lane's best error - 10 % on average, 2-4 % at the large steps, a third at step 1 - put
// their palettes on a list in LDS, and the wavefront evaluates that list exactly TEXEL BY TEXEL: item g = (list entry g / np, texel
// g % np), so a handful of candidates still fills the 64 lanes; the per-texel first-peak scores are added up with LDS atomics (integer
// sums: any order) and the owner reads its exact error back. Same candidates, same decisions as perturb_macro (bc7_core.h), which the
// plain kernel above runs and DXTEX_BC7_PERTURB_PLAIN=1 selects.
// Mode 1 at 5 waves per SIMD (91 registers instead of 105, no spill): the list handling waits on LDS, 21.3 -> 20.9 ms; 6 waves: no gain.
#if !defined(DXTEX_PF1_WAVES)
#define DXTEX_PF1_WAVES 5
#endif
#if !defined(DXTEX_PF_M4)
#define DXTEX_PF_M4 1
#endif
#if !defined(DXTEX_PF_LCAP)
#define DXTEX_PF_LCAP 32               // entries of the exact-evaluation list: a step with more passing candidates (rare: 6 of 64 on average) takes two rounds
#endif
template<int MODE, int IM, int CHSET>
__global__ void __launch_bounds__(64, (MODE == 1) ? DXTEX_PF1_WAVES : 1) bc7_perturb_filter_kernel(Bc7Args a, int loop)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    static_assert(!C::kAlpha, "colour / combined loops only");
    constexpr int N = C::N;
    constexpr int LCAP = DXTEX_PF_LCAP;
    __shared__ uint32_t sSlot[16 * kSlotStride];    // texel columns, one per lane
    __shared__ uint32_t sCand[LCAP * (2 * N)];  // list entry e: pal[N], then -|q|^2 [N] (16-byte aligned: written and read as ds_*_b128; an odd stride measured 0.8 ms slower)
    __shared__ uint32_t sOwner[LCAP];               // list entry -> owner lane | subset size << 8
    __shared__ uint32_t sSum[LCAP];                 // list entry -> sum over the subset of the first-peak scores
    const int lane = threadIdx.x;
    const uint32_t live = a.counters[34];
    if (live == 0) return;
    if (TaskMap<MODE, IM>::NS == 1 && live <= a.perturbWaveMax) return;      // short list of whole-block tasks: bc7_perturb_wave_kernel's turn
    uint32_t* head = a.counters + kQueueBase + loop;
    uint32_t* slotCol = &sSlot[lane];
    WT_BEGIN(1);

    PerturbState st = perturb_begin<MODE, IM, CHSET>(0, 0, 0);
    SlotRegion rg; rg.base = slotCol; rg.np = 0; rg.p2sum = 0;
    int base = 0;
    uint32_t myTask = 0xFFFFFFFFu;
    uint32_t flatMask = 0, flatVals = 0;
    WaveQueue q; q.lo = q.hi = 0; q.drained = false;
    for (;;)
    {
        const unsigned long long idle = __ballot(myTask == 0xFFFFFFFFu);
        if (idle && !(q.drained && q.lo >= q.hi))
        {
            const uint32_t idx = queue_take(q, head, live, idle, lane);
            WT_TAKE(idx);
            if (idx != 0xFFFFFFFFu)
            {
                const uint2 task = a.order[idx];
                myTask = task.x;
                const TaskRec r = a.recs[myTask];
                search_pickup<MODE, IM>(a, task, slotCol, rg);
                st = perturb_begin<MODE, IM, CHSET>(r.A, r.B, r.err);
                base = loop_base<MODE, IM, CHSET>(rg, st.optA, st.optB);
                if constexpr (kFlatSkip<MODE, IM, CHSET>) flatMask = flat_channels(rg, flatVals);
            }
        }
        if constexpr (kFlatSkip<MODE, IM, CHSET>)         // see bc7_perturb_kernel
        {
            const bool fc = myTask != 0xFFFFFFFFu && flat_call<MODE, IM, CHSET>(st, flatMask, flatVals);
            if (__any(fc))
            {
                const int nearest = eval_nearest<MODE, IM, CHSET>(rg, st, base);
                if (fc && nearest == st.optErr)
                {
                    st = skip_flat_calls<MODE, IM, CHSET>(st);
                    if (st.ch >= C::CH1)
                    {
                        TaskRec* r = a.recs + myTask;
                        r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                        myTask = 0xFFFFFFFFu;
                        WT_DONE();
                    }
                }
            }
        }
        const bool busyL = myTask != 0xFFFFFFFFu;
        if (__ballot(busyL) == 0ull)
        {
            if (q.drained && q.lo >= q.hi) break;
            continue;
        }
        WT_ROUND(busyL);
        wave_lds_sync();             // the texel columns of tasks just taken are visible to every lane
        // widest subset among the busy lanes (the list is sorted by size: nearly always everybody's)
        int npMax = busyL ? rg.np : 0;
#pragma unroll
        for (int d = 32; d; d >>= 1) npMax = max(npMax, __shfl_xor(npMax, d));
        npMax = __builtin_amdgcn_readfirstlane(npMax);
        const uint32_t npMagic = (65536u + uint32_t(npMax) - 1u) / uint32_t(npMax);      // g / npMax == (g * npMagic) >> 16 for g < 4096 (npMax <= 16)

        // ---- one PerturbOne call (:2926-2966) for every busy lane
        VarPal<N> vp;
        varpal_init<MODE, IM, CHSET>(vp, st.optA, st.optB, st.ch);
        const uint32_t fixedU = unq1<C::PREC>(byte_of(st.do_b ? st.optA : st.optB, st.ch));
        int cur = int(byte_of(st.do_b ? st.optB : st.optA, st.ch));
        int minErr = st.optErr;
        const int sh = 8 * st.ch;

        // exact error of the candidate `tmp` if it can still be below minErr, else INT_MAX
        auto candidate = [&](int tmp, bool valid) -> int
        {
            const uint32_t u = unq1<C::PREC>(uint32_t(tmp) & ((1u << C::PREC) - 1u));
            const uint32_t uaC = st.do_b ? fixedU : u, ubC = st.do_b ? u : fixedU;
            uint32_t pal[N], nq2[N];
            const int lb64 = lerp_base(uaC), ld = lerp_delta(uaC, ubC);
#pragma unroll
            for (int i = 0; i < N; ++i)
            {
                const uint32_t v = lerp1(lb64, ld, weight(C::BITS, i));
                pal[i] = vp.palO[i] | (v << sh);
                nq2[i] = vp.nq2O[i] - umul24(v, v);
            }
            int sum = 0;
            for_texels(rg, [&](int k)
            {
                uint32_t p = rg.fetch(k);
                if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
                int m = int(udot4acc(p, pal[0], uint32_t(int(nq2[0]) >> 1)));
#pragma unroll
                for (int i = 1; i < N; ++i) { const int t = int(udot4acc(p, pal[i], uint32_t(int(nq2[i]) >> 1))); m = t > m ? t : m; }
                sum += m;
            });
            const int lb = base - 2 * sum - rg.count();
            const bool pass = busyL && valid && lb < minErr;
            const unsigned long long passMask = __ballot(pass);
            if (passMask == 0ull) return 0x7FFFFFFF;
            const int n = __popcll(passMask), pos = lanes_below(passMask);
            int exact = 0x7FFFFFFF;
            for (int first = 0; first < n; first += LCAP)
            {
                const int cnt = min(n - first, LCAP), e = pos - first;
                const bool mine = pass && uint32_t(e) < uint32_t(cnt);
                if (mine)
                {
#pragma unroll
                    for (int i = 0; i < N; ++i) { sCand[e * (2 * N) + i] = pal[i]; sCand[e * (2 * N) + N + i] = nq2[i]; }
                    sOwner[e] = uint32_t(lane) | (uint32_t(rg.np) << 8);
                    sSum[e] = 0u;
                }
                wave_lds_sync();
                const int items = cnt * npMax;
                for (int g0 = 0; g0 < items; g0 += 64)
                {
                    const int g = g0 + lane;
                    const int ent = int((uint32_t(g) * npMagic) >> 16), k = g - ent * npMax;
                    if (g < items)
                    {
                        const uint32_t own = sOwner[ent];
                        if (k < int(own >> 8))
                        {
                            uint32_t p = sSlot[(own & 63u) + uint32_t(k) * uint32_t(kSlotStride)];
                            if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
                            int sc[N];
#pragma unroll
                            for (int i = 0; i < N; ++i) sc[i] = score(p, sCand[ent * (2 * N) + i], sCand[ent * (2 * N) + N + i]);
                            atomicAdd(&sSum[ent], uint32_t(first_peak(sc)));
                        }
                    }
                }
                wave_lds_sync();
                if (mine) exact = base - int(sSum[e]);
                if (first + LCAP < n) wave_lds_sync();          // the next round rewrites the list
            }
            return exact;
        };

        {
            // the first step is half the range: exactly one direction is a legal endpoint (see perturb_macro)
            constexpr int half = 1 << (C::PREC - 1);
            const int tmp = (cur >= half) ? cur - half : cur + half;
            const int e = candidate(tmp, true);
            if (e < minErr) { minErr = e; cur = tmp; }
        }
#pragma unroll 1
        for (int step = 1 << (C::PREC - 2); step; step >>= 1)
        {
            int beststep = 0;
#pragma unroll 1
            for (int sign = -1; sign <= 1; sign += 2)
            {
                const int tmp = cur + sign * step;
                const bool valid = (tmp >= 0) && (tmp < (1 << C::PREC));
                const int e = candidate(tmp, valid);
                if (e < minErr) { minErr = e; beststep = sign * step; }
            }
            cur += beststep;
        }
        if (busyL)
        {
            st = perturb_transition<MODE, IM, CHSET>(st, minErr, uint32_t(cur));
            if (st.ch >= C::CH1)
            {
                TaskRec* r = a.recs + myTask;
                r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                myTask = 0xFFFFFFFFu;
                WT_DONE();
            }
        }
    }
    WT_END();
}

// Mode 6's short task lists go to bc7_exhaustive_wave_kernel (below): above this many live tasks the lane-per-task kernel is the
// efficient one. Both are launched, each returns when it is not its turn.


// The Exhaustive phase (:2971-3042) for the channels of CHSET, wave-synchronous by window: every lane holds one task; all lanes
// walk their current window (one channel's +-5 x +-5 neighbourhood) together, and a lane that needs a new task takes it at a
// window boundary.
//   * Visiting a candidate costs a bound (eval_var_bound: about half an evaluation). The few per cent of candidates whose key is
//     below the lane's best key (bc7_core.h: exh_key) go to a list shared by the wavefront.
//   * The list is evaluated exactly by ALL lanes, whoever's candidates they are: a helper fetches the owner's palette cache and
//     window geometry with cross-lane reads, reads the owner's texels from the owner's LDS column, and folds the result into the
//     owner's best key with an LDS atomic min (the window's result is an order-independent minimum).
//   * Windows differ in size (55 ... 110 candidates). Once fewer than `tailBelow` lanes still have candidates to visit, the
//     remaining candidates of all lanes are pooled and bounded by all 64 lanes in the same way, so the tail of a round runs at
//     full width instead of at the width of its longest window.
// Exactly the result of evaluating every candidate in the reference's order, at about half the instructions.
#if defined(DXTEX_EXH_STATS)
// development instrumentation: trip counts and lane participation of the mode-1 Exhaustive kernel (printed by bc7_stats_print_kernel)
#define DXTEX_STAT(SLOT, MASK) do { const unsigned long long m_ = (MASK); if (MODE == 1 && lane == 0) { atomicAdd(&stats[2 * (SLOT)], 1ull); atomicAdd(&stats[2 * (SLOT) + 1], (unsigned long long)__popcll(m_)); } } while (0)
__global__ void bc7_stats_print_kernel(unsigned long long* stats)
{
    // 0 round (busy lanes), 1 own-window bound trip, 2 drain (list length), 3 exact trip, 4 pooled bound trip, 5 refill
    for (int i = 0; i < 6; ++i)
        printf("exh-stats %d trips %llu lanes %llu\n", i, stats[2 * i], stats[2 * i + 1]);
}
#else
#define DXTEX_STAT(SLOT, MASK) do { } while (0)
#endif

// Mode 4 with 2-bit colour indices would take 217 registers (2 waves per SIMD): capped at 3 waves (168 registers + 168 B of spill), 4.6 -> 3.9 ms.
#if !defined(DXTEX_EXH45_WGS)
#define DXTEX_EXH45_WGS 3
#endif
// LDS lists of bc7_exhaustive_kernel, per mode: sWork = the pooled candidates of a round (the lanes' shares), sExact = candidates waiting for their exact
// evaluation (drained when fewer than 64 slots are left). Together with the texel columns (4160 B) and the best keys (256 B) they decide how many
// wavefronts a CU holds: 10 752 B (1024 + 512 entries, rounds 2 - 5) were 15, one SIMD of four a wavefront short. Round 6, one box, byte-identical:
// 896 + 512 (10 240 B: 16 per CU) exhaustive<1> 31.5 -> 30.6 ms, <3> 21.2 -> 20.1, the image 126.0 -> 123.8 ms; mode 3 with 512 + 256 (shorter
// pooled shares, earlier drains) 20.1 -> 19.5, the image 122.9; mode 1 prefers 896 (768 / 640 / 512: + 0.1 ... 0.3 ms), the late modes do not care;
// five wavefronts per SIMD (launch bound 5, 640 + 304 entries) change nothing for either.
#if !defined(DXTEX_EXH_WORK1)
#define DXTEX_EXH_WORK1 896
#endif
#if !defined(DXTEX_EXH_EXACT1)
#define DXTEX_EXH_EXACT1 512
#endif
#if !defined(DXTEX_EXH_WORK3)
#define DXTEX_EXH_WORK3 512
#endif
#if !defined(DXTEX_EXH_EXACT3)
#define DXTEX_EXH_EXACT3 256
#endif
#if !defined(DXTEX_EXH_WORK)
#define DXTEX_EXH_WORK 896
#endif
#if !defined(DXTEX_EXH_EXACT)
#define DXTEX_EXH_EXACT 512
#endif
#if !defined(DXTEX_EXH1_WAVES)
#define DXTEX_EXH1_WAVES 1
#endif
#if !defined(DXTEX_EXH3_WAVES)
#define DXTEX_EXH3_WAVES 1
#endif
template<int MODE> struct ExhLists { enum : int { kWork = (MODE == 1) ? DXTEX_EXH_WORK1 : (MODE == 3) ? DXTEX_EXH_WORK3 : DXTEX_EXH_WORK,
                                                  kExact = (MODE == 1) ? DXTEX_EXH_EXACT1 : (MODE == 3) ? DXTEX_EXH_EXACT3 : DXTEX_EXH_EXACT,
                                                  kWaves = (MODE == 4 || MODE == 5) ? DXTEX_EXH45_WGS : (MODE == 1) ? DXTEX_EXH1_WAVES : (MODE == 3) ? DXTEX_EXH3_WAVES : 1 }; };


// What a helper needs of another lane's window.
template<int N> struct ExhCtx { VarPal<N> vp; uint32_t geom; int base; };

template<int MODE, int IM, int CHSET>
__device__ __forceinline__ ExhCtx<LoopCfg<MODE, IM, CHSET>::N> exh_fetch_ctx(const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, uint32_t geom, int base, int owner)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    ExhCtx<C::N> c;
#pragma unroll
    for (int i = 0; i < C::N; ++i)
    {
        c.vp.palO[i] = C::kAlpha ? 0u : uint32_t(__shfl(int(vp.palO[i]), owner));
        c.vp.nq2O[i] = C::kAlpha ? 0u : uint32_t(__shfl(int(vp.nq2O[i]), owner));
    }
    c.geom = uint32_t(__shfl(int(geom), owner));
    c.base = __shfl(base, owner);
    return c;
}

#if !defined(DXTEX_RANGE_TESTS)
#define DXTEX_RANGE_TESTS 5
#endif
template<int MODE, int IM, int CHSET>
__global__ void __launch_bounds__(64, ExhLists<MODE>::kWaves) bc7_exhaustive_kernel(Bc7Args a, int loop, int tailBelow, int rangeTests)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    __shared__ uint32_t sSlot[16 * kSlotStride];             // texel columns, one per lane
    constexpr int kExhWorkMax = ExhLists<MODE>::kWork, kExhExactMax = ExhLists<MODE>::kExact;
    __shared__ uint32_t sWork[kExhWorkMax];         // pooled candidates to bound: (owner << 8) | code
    __shared__ uint32_t sExact[kExhExactMax];       // candidates to evaluate exactly: (owner << 8) | code
    __shared__ uint32_t sBest[64];                  // per lane: best key of its current window
    const int lane = threadIdx.x;
#if defined(DXTEX_EXH_STATS)
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(a.flagged)) + 4;
#endif
    const uint32_t live = a.counters[34];
    if (live == 0) return;           // nothing survived pre (a phase that owns no block, everything pruned): skip the queue atomics
    if (TaskMap<MODE, IM>::NS == 1 && live <= a.exhWaveMax) return;      // short list of whole-block tasks: bc7_exhaustive_wave_kernel has done it
    uint32_t* head = a.counters + kQueueBase + loop;
    uint32_t* slotCol = &sSlot[lane];
    WT_BEGIN(2);

    ExhState st; st.ch = C::CH1; st.optA = st.optB = 0; st.optErr = 0;
    st.o = st.i = st.oEnd = st.iEnd = st.lo = 0; st.iA = 0; st.o0 = 0; st.aleb = 0; st.omin = st.imin = 0; st.best = 0; st.bestCode = -1;
    VarPal<C::N> vp;
#pragma unroll
    for (int i = 0; i < C::N; ++i) { vp.palO[i] = 0; vp.nq2O[i] = 0; }
    SlotRegion rg; rg.base = slotCol; rg.np = 0; rg.p2sum = 0;
    int base = 0;
    uint32_t myTask = 0xFFFFFFFFu;
    WaveQueue q; q.lo = q.hi = 0; q.drained = false;
    int nExact = 0;                                 // entries in sExact (wave-uniform)

    // exact evaluation of everything in sExact, by all lanes; afterwards every lane's bestKey is current
    auto drain = [&](uint32_t geom, uint32_t& bestKey)
    {
        wave_lds_sync();
        const int total = nExact;
        DXTEX_STAT(2, (total >= 64) ? ~0ull : ((1ull << total) - 1ull));
        for (int g0 = 0; g0 < total; g0 += 64)
        {
            const int g = g0 + lane;
            const bool valid = g < total;
            DXTEX_STAT(3, __ballot(valid));
            const uint32_t ent = valid ? sExact[g] : 0u;
            const int owner = int(ent >> 8), code = int(ent & 0xFFu);
            const ExhCtx<C::N> c = exh_fetch_ctx<MODE, IM, CHSET>(vp, geom, base, owner);
            if (valid)
            {
                SlotRegion ro; ro.base = sSlot + owner; ro.np = int((c.geom >> 3) & 31u); ro.p2sum = 0;
                const int e = exh_exact<MODE, IM, CHSET>(ro, c.vp, int(c.geom & 3u), int((c.geom >> 2) & 1u), int((c.geom >> 8) & 0xFFu), int((c.geom >> 16) & 0xFFu), code, c.base);
                atomicMin(&sBest[owner], exh_key(e, code));
            }
        }
        wave_lds_sync();
        nExact = 0;
        bestKey = sBest[lane];
    };

    for (;;)
    {
        // ---- window boundary: lanes without a task take one (its first window opens here)
        const unsigned long long idle = __ballot(myTask == 0xFFFFFFFFu);
        if (idle != 0ull && !(q.drained && q.lo >= q.hi))
        {
            const uint32_t idx = queue_take(q, head, live, idle, lane);
            WT_TAKE(idx);
            DXTEX_STAT(5, __ballot(idx != 0xFFFFFFFFu));
            if (idx != 0xFFFFFFFFu)
            {
                const uint2 task = a.order[idx];
                myTask = task.x;
                const TaskRec r = a.recs[myTask];
                search_pickup<MODE, IM>(a, task, slotCol, rg);
                int other;
                base = loop_base<MODE, IM, CHSET>(rg, r.A, r.B, &other);
                if (loop_is_settled<CHSET>(r.err, other) || !exh_begin<MODE, IM, CHSET>(st, vp, r.A, r.B, r.err)) myTask = 0xFFFFFFFFu;     // nothing to search: endpoints stay
            }
        }
        // Many windows cannot hold an improvement - every window on a channel that is constant over the block, as the swapped-in alpha
        // of an opaque block is, and many others - and one interval bound says so (exh_range_bound over the whole window: 78 % / 55 % / 32 % /
        // 20 % / 14 % of the windows of modes 4 / 5 / 6 / 1 / 3 on the benchmark image, unpruned). Such a window is closed at once and the lane
        // goes on to its next one. A window that stays loses the strips along its four sides that cannot hold an improvement either (exh_peel,
        // bc7_core.h: 30 % / 22 % of the candidates of modes 1 / 3). Both are the same test on a different rectangle, so they share ONE loop: per
        // trip every lane tests the rectangle its own stage calls for - stage 0: the whole window, stages 1 ... 4: a side - and a lane whose
        // window was closed starts over on the next one. `rangeTests` trips (DXTEX_RANGE_TESTS = 5: a lane that never closes a window tests all four sides);
        // whatever has not been tested by then is simply visited. Lanes that end up without candidates help the others through the pooled phase.
        {
            int stage = 0;
            const int trips = rangeTests & 0xFF, lastStage = 1 + 4 * (rangeTests >> 8);      // (rangeTests >> 8 = layers of strips: 1; development knob: 0 = no peeling, 2 = two layers)
#pragma unroll 1
            for (int trip = 0; trip < trips; ++trip)
            {
                const bool testing = (myTask != 0xFFFFFFFFu) && stage < lastStage;
                if (__ballot(testing) == 0ull) break;
                int ro0 = st.o, ro1 = st.oEnd - 1, ri0 = st.i, ri1 = st.iEnd - 1;      // at window open: st.i == the first row's first value
                bool valid = true;
                if (stage > 0) valid = exh_peel_rect(st, (stage - 1) & 3, ro0, ro1, ri0, ri1);
                if (!valid) { ro0 = ro1 = st.o; ri0 = ri1 = st.iEnd - 1; }           // (the call is wave-uniform: a harmless rectangle)
                const int b = exh_range_bound<MODE, IM, CHSET>(rg, vp, st, base, ro0, ro1, ri0, ri1);
                if (testing)
                {
                    const bool out = valid && b >= st.optErr;
                    if (stage == 0)
                    {
                        if (out)
                        {
                            st.o = st.oEnd;              // past the last row: exh_next commits "no change" and opens the next window
                            if (!exh_next<MODE, IM, CHSET>(st, vp))
                            {
                                TaskRec* r = a.recs + myTask;
                                r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                                myTask = 0xFFFFFFFFu;
                                WT_DONE();
                            }
                        }
                        else stage = 1;
                    }
                    else
                    {
                        exh_peel_apply(st, (stage - 1) & 3, out);
                        ++stage;
                    }
                }
            }
        }
        const bool busyL = myTask != 0xFFFFFFFFu;
        const unsigned long long busy = __ballot(busyL);
        DXTEX_STAT(0, busy);
        if (busy == 0ull)
        {
            if (q.drained && q.lo >= q.hi) break;
            continue;
        }
        WT_ROUND(busyL);
        // what a helper needs of this lane's window: channel, loop orientation, subset size, window origin
        const uint32_t geom = uint32_t(st.ch & 3) | (uint32_t(st.aleb & 1) << 2) | (uint32_t(rg.np) << 3) | (uint32_t(st.o0 & 0xFF) << 8) | (uint32_t(st.lo & 0xFF) << 16);
        uint32_t bestKey = busyL ? exh_start_key(st) : 0u;
        sBest[lane] = bestKey;
        int rem = busyL ? exh_remaining(st) : 0;
        wave_lds_sync();             // texel columns and best keys are visible to every lane

        // ---- every lane visits its own window (bounds only) while most lanes have candidates left
        for (;;)
        {
            const unsigned long long moreMask = __ballot(rem > 0);
            if (__popcll(moreMask) < tailBelow) break;
            if (nExact > kExhExactMax - 64) { drain(geom, bestKey); continue; }
            DXTEX_STAT(1, moreMask);
            bool keep = false;
            int code = 0;
            if (rem > 0)
            {
                code = exh_code(st);
                const int av = st.aleb ? st.o : st.i, bv = st.aleb ? st.i : st.o;
                const int lb = eval_var_bound<MODE, IM, CHSET>(rg, vp, st.ch, unq1<C::PREC>(uint32_t(av)), unq1<C::PREC>(uint32_t(bv)), base);
                keep = exh_key(lb, code) < bestKey;
                exh_advance(st); --rem;
            }
            // append the unbeaten candidates to the shared list: positions from the ballot, no atomics
            const unsigned long long keepMask = __ballot(keep);
            if (keep) sExact[nExact + lanes_below(keepMask)] = (uint32_t(lane) << 8) | uint32_t(code);
            nExact += int(__popcll(keepMask));
        }
        // ---- the remaining candidates of all lanes, pooled and bounded by all lanes
        for (;;)
        {
            const unsigned long long owners = __ballot(rem > 0);
            if (owners == 0ull) break;
            const int share = kExhWorkMax / __popcll(owners);            // >= 16
            const int mine = rem < share ? rem : share;
            int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const int first = incl - mine, total = __shfl(incl, 63);
            for (int k = 0; k < mine; ++k)
            {
                sWork[first + k] = (uint32_t(lane) << 8) | uint32_t(exh_code(st));
                exh_advance(st);
            }
            rem -= mine;
            wave_lds_sync();
            for (int g0 = 0; g0 < total; g0 += 64)
            {
                if (nExact > kExhExactMax - 64) drain(geom, bestKey);
                const int g = g0 + lane;
                const bool valid = g < total;
                DXTEX_STAT(4, __ballot(valid));
                const uint32_t ent = valid ? sWork[g] : 0u;
                const int owner = int(ent >> 8), code = int(ent & 0xFFu);
                const ExhCtx<C::N> c = exh_fetch_ctx<MODE, IM, CHSET>(vp, geom, base, owner);
                bool keep = false;
                if (valid)
                {
                    SlotRegion ro; ro.base = sSlot + owner; ro.np = int((c.geom >> 3) & 31u); ro.p2sum = 0;
                    const int lb = exh_bound<MODE, IM, CHSET>(ro, c.vp, int(c.geom & 3u), int((c.geom >> 2) & 1u), int((c.geom >> 8) & 0xFFu), int((c.geom >> 16) & 0xFFu), code, c.base);
                    keep = exh_key(lb, code) < sBest[owner];
                }
                const unsigned long long keepMask = __ballot(keep);
                if (keep) sExact[nExact + lanes_below(keepMask)] = ent;
                nExact += int(__popcll(keepMask));
            }
            wave_lds_sync();
        }
        // ---- exact evaluation of what is left on the list, then every lane commits its window and opens the next (or finishes)
        drain(geom, bestKey);
        if (busyL)
        {
            exh_apply_key(st, bestKey);
            if (!exh_next<MODE, IM, CHSET>(st, vp))
            {
                TaskRec* r = a.recs + myTask;
                r->A = st.optA; r->B = st.optB; r->err = st.optErr;
                myTask = 0xFFFFFFFFu;
                WT_DONE();
            }
        }
    }
    WT_END();
}

// Exhaustive for SHORT task lists, one task per wavefront: the <= 11 x 11 candidates of a channel's window are evaluated by the
// 64 lanes in two rounds and the first minimum in the reference's loop order is taken by a wave reduction - the same candidates,
// the same winner, but the serial chain of a task shrinks from ~480 evaluations to 8. Mode 6 (one task per block) often leaves
// fewer tasks than the machine has lanes; with one task per lane the kernel was as slow as its slowest lane and 85 % idle.

template<int MODE, int IM, int CHSET>
__global__ void __launch_bounds__(64) bc7_exhaustive_wave_kernel(Bc7Args a)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    typedef TaskMap<MODE, IM> TM;
    static_assert(TM::NS == 1, "whole-block tasks only");
    __shared__ uint32_t sTex[16 * kSlotStride];
    const int lane = threadIdx.x;
    const uint32_t live = a.counters[34];
    if (live == 0 || live > a.exhWaveMax) return;
    for (uint32_t idx = blockIdx.x; idx < live; idx += gridDim.x)
    {
        const uint2 task = a.order[idx];
        const TaskRec rec = a.recs[task.x];
        SlotRegion rg;
        wave_lds_sync();                                  // the previous task's reads are done
        search_pickup<MODE, IM>(a, task, &sTex[0], rg);   // every lane writes the same 16 texels to column 0: all lanes then read them back
        wave_lds_sync();
        int other;
        const int base = loop_base<MODE, IM, CHSET>(rg, rec.A, rec.B, &other);
        if (loop_is_settled<CHSET>(rec.err, other)) continue;      // scalar slot already exact (wave-uniform: one task per wavefront)
        ExhState st; st.optA = rec.A; st.optB = rec.B; st.optErr = rec.err;
        st.o = st.i = st.oEnd = st.iEnd = st.lo = 0; st.iA = 0; st.o0 = 0; st.aleb = 0; st.omin = st.imin = 0; st.best = rec.err; st.bestCode = -1; st.ch = C::CH0;
        VarPal<C::N> vp;
#pragma unroll 1
        for (int ch = C::CH0; ch < C::CH1; ++ch)
        {
            st = exh_window<MODE, IM, CHSET>(st, ch);
            if (st.ch >= C::CH1) break;                   // error already zero: Exhaustive returns at once (:2980)
            varpal_init<MODE, IM, CHSET>(vp, st.optA, st.optB, ch);
            // candidate c = (o - oStart) * 11 + (i - lo) over the window's bounding rectangle; the loop nest visits exactly the
            // cells with o < oEnd, max(o, lo) <= i < iEnd, in increasing c
            const int oStart = st.o;
            uint64_t bestKey = ~0ull;
#pragma unroll 1
            for (int round = 0; round < 2; ++round)
            {
                const int c = lane + 64 * round;
                const int o = oStart + c / 11, i = st.lo + c % 11;
                const bool valid = c < 121 && o < st.oEnd && i < st.iEnd && i >= o;
                int e = 0x7FFFFFFF;
                if (valid)
                {
                    const int av = st.aleb ? o : i, bv = st.aleb ? i : o;
                    e = eval_var<MODE, IM, CHSET>(rg, vp, ch, unq1<C::PREC>(uint32_t(av)), unq1<C::PREC>(uint32_t(bv)), base);
                }
                const uint64_t key = valid ? ((uint64_t(uint32_t(e)) << 8) | uint32_t(c)) : ~0ull;
                bestKey = key < bestKey ? key : bestKey;
            }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1)
            {
                const uint32_t lo = uint32_t(__shfl_xor(int(uint32_t(bestKey)), d)), hi = uint32_t(__shfl_xor(int(uint32_t(bestKey >> 32)), d));
                const uint64_t other = (uint64_t(hi) << 32) | lo;
                bestKey = other < bestKey ? other : bestKey;
            }
            if (bestKey != ~0ull)
            {
                const int e = int(uint32_t(bestKey >> 8)), c = int(bestKey & 0xFFu);
                if (e < st.best) { st.best = e; st.omin = oStart + c / 11; st.imin = st.lo + c % 11; }      // strict: first minimum in loop order (:3006)
            }
            st = exh_commit(st);
        }
        if (lane == 0)
        {
            TaskRec* r = a.recs + task.x;
            r->A = st.optA; r->B = st.optB; r->err = st.optErr;
        }
    }
}

template<int MODE, int IM>
__global__ void __launch_bounds__(256, (MODE == 4 || MODE == 5) ? DXTEX_PP45_WGS : 1) bc7_post_kernel(Bc7Args a)
{
    typedef TaskMap<MODE, IM> TM;
    constexpr int BPW = (TM::TPB >= 64) ? 1 : 64 / TM::TPB;
    __shared__ uint32_t sL[4][BPW * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t nbFirst = (blockIdx.x * 4 + wave) * BPW;
    if (nbFirst >= a.nblocks) return;
    // a gated launch that owns no block: an early phase leaves every candidate slot to its late phase; mode 7 leaves its slots unwritten
    // and bc7_pick_kernel skips them on the same condition
    if (a.gate && *a.gate < a.gateMin) return;
    const uint32_t blk = uint32_t(lane) / TM::TPB, r = uint32_t(lane) % TM::TPB;
    const uint32_t nb = nbFirst + blk;
    const uint32_t rank = r / TM::G, region = r % TM::G;
    uint32_t shape, mask, anchor, rot;
    const bool active = (blk < uint32_t(BPW)) && task_geometry<MODE, IM>(a, nb, r, shape, mask, anchor, rot);
    const bool mine = blk < uint32_t(BPW) && nb < a.nblocks && phase_owns<MODE>(a, nb);
    if (!__any(active))
    {
        if (mine && r == 0)
        {
            Cand c; c.err = 0xFFFFFFFFu; c.ord = 0xFFFFFFFFu; c.lo = 0; c.hi = 0;
            a.cands[uint64_t(nb) * NUM_SLOTS + TM::SLOT] = c;
        }
        return;
    }
    stage_packed<BPW>(a, nbFirst, lane, sL[wave]);

    SubsetResult res;
    res.orgErr = 0; res.optErr = 0; res.orgA = res.orgB = res.optA = res.optB = 0;
    res.orgIdx1 = res.orgIdx2 = res.optIdx1 = res.optIdx2 = 0;
    // The unoptimised half of Refine was evaluated by pre (its error is in the task record); here only its endpoints are re-derived, and
    // AssignIndices runs once per task - for the optimised endpoints - instead of twice. The indices of the unoptimised endpoints are needed
    // only where the block's winning candidate stands with them (below).
    int np = 0; Region rg; Block16 b16;
    if (active)
    {
        task_org<MODE, IM, false>(nullptr, &sL[wave][blk * 16], mask, anchor, rot, res, np, true, rg, b16,
                                  (TM::NS == 2) ? a.seeds + uint64_t(nb) * 64 + (TM::LIST ? 32 : 0) + r
                                                : (TM::NS == 1) ? a.seeds1 + uint64_t(nb) * 2 : a.seeds3 + uint64_t(nb) * 192 + shape * 3 + (r % TM::G));
        const TaskRec rec = a.recs[uint64_t(nb) * TM::TPB + r];
        res.orgErr = rec.orgErr;
        if (TM::NS == 1) refine_post<MODE, IM>(b16, rec.A, rec.B, 0u, res);
        else refine_post<MODE, IM>(rg, rec.A, rec.B, anchor, res);
    }
    // totals over the subsets of the candidate (lanes of one candidate are adjacent)
    int orgTot = res.orgErr, optTot = res.optErr;
#pragma unroll
    for (int d = 1; d < TM::G; d <<= 1) { orgTot += __shfl_xor(orgTot, d); optTot += __shfl_xor(optTot, d); }
    const bool useOpt = optTot < orgTot;
    const int err = useOpt ? optTot : orgTot;
    // evaluation order inside D3DX_BC7::Encode: modes ascending, then rotation, index mode, shape rank
    const uint32_t sub = candidate_sub<MODE, IM>(rank);
    const uint32_t key = active ? ((uint32_t(err) << 7) | sub) : 0xFFFFFFFFu;
    uint32_t best = key;
#pragma unroll
    for (int d = 1; d < TM::TPB && d < 64; d <<= 1) best = min(best, uint32_t(__shfl_xor(int(best), d)));
    // the winner keeps its unoptimised endpoints: their indices (and the anchor fix-up of the endpoints) now
    const bool needOrg = active && key == best && !useOpt;
    if (__any(needOrg))
    {
        if (needOrg)
        {
            if (TM::NS == 1) (void)assign_indices<MODE, IM>(b16, res.orgA, res.orgB, 0u, res.orgIdx1, res.orgIdx2);
            else (void)assign_indices<MODE, IM>(rg, res.orgA, res.orgB, anchor, res.orgIdx1, res.orgIdx2);
        }
    }
    const uint32_t myA = useOpt ? res.optA : res.orgA, myB = useOpt ? res.optB : res.orgB;
    const uint64_t myIdx1 = useOpt ? res.optIdx1 : res.orgIdx1, myIdx2 = useOpt ? res.optIdx2 : res.orgIdx2;
    // gather the candidate's endpoints and indices into its subset-0 lane
    uint32_t epA[3] = { myA, 0, 0 }, epB[3] = { myB, 0, 0 };
    uint64_t idx1 = myIdx1;
#pragma unroll
    for (int s = 1; s < TM::NS; ++s)
    {
        epA[s] = uint32_t(__shfl_down(int(myA), s));
        epB[s] = uint32_t(__shfl_down(int(myB), s));
        const uint64_t oi = uint64_t(uint32_t(__shfl_down(int(uint32_t(myIdx1)), s))) |
                            (uint64_t(uint32_t(__shfl_down(int(uint32_t(myIdx1 >> 32)), s))) << 32);
        idx1 |= oi;
    }

    if (blk < uint32_t(BPW) && nb < a.nblocks && region == 0)
    {
        if (active && key == best)
        {
            uint32_t anchors[3] = { 0, 0, 0 };
            if (TM::NS == 2) anchors[1] = kAnchor2[shape];
            if (TM::NS == 3) { anchors[1] = kAnchor3[shape] & 15; anchors[2] = kAnchor3[shape] >> 4; }
            Cand c;
            c.err = uint32_t(err);
            c.ord = uint32_t(MODE) * 128u + sub;
            emit_block<MODE>(shape, rot, IM, epA, epB, idx1, myIdx2, anchors, c.lo, c.hi);
            a.cands[uint64_t(nb) * NUM_SLOTS + TM::SLOT] = c;
            // (atomics: the single-subset modes of one phase run concurrently on separate streams)
            if (err == 0) atomicMin(&a.zeroOrd[nb], c.ord);
            atomicMin(&a.bestErr[nb], err);
        }
        else if (!active && rank == 0 && mine)
        {
            Cand c; c.err = 0xFFFFFFFFu; c.ord = 0xFFFFFFFFu; c.lo = 0; c.hi = 0;
            a.cands[uint64_t(nb) * NUM_SLOTS + TM::SLOT] = c;
        }
    }
}

// The whole-block fits the single-subset modes start from (RoughMSE with one region, :3541-3568): RGB for modes 4 / 5 (every
// rotation and index mode uses the same one - it is taken from the unrotated floats), RGBA for mode 6. One lane per block, one
// launch per fit; the 16 texels sit in registers (FULL = constant trip counts).
template<bool RGBA>
__global__ void __launch_bounds__(256) bc7_block_seeds_kernel(Bc7Args a)
{
    const uint32_t nb = blockIdx.x * 256u + threadIdx.x;
    if (nb >= a.nblocks) return;
    const BcSeg& sg = seg_of(a.seg, nb);
    float f[64];
#pragma unroll
    for (int t = 0; t < 16; ++t) { uint32_t ldr; load_block_texel(sg.src, sg.nbw, sg.nb0 + (nb - sg.l0), uint32_t(t), &f[t * 4], ldr); }
    uint32_t A, B;
    seed_endpoints<RGBA, true>(f, 0xFFFFu, A, B);
    a.seeds1[uint64_t(nb) * 2 + (RGBA ? 1 : 0)] = make_uint2(A, B);
}

// out[0] = number of blocks flagged for the early mode-6 phase (lists[37]), out[1] = number of blocks that have alpha (lists[32]).
__global__ void __launch_bounds__(256) bc7_flag_count_kernel(Bc7Args a, uint32_t* out)
{
    // ... and decides lists[37] first: mode 6 goes early where its lower bound does not exceed early6Pct % of the best 3-bit rough error (or
    // the block has alpha). A scheduling hint, not a result - see the rough kernel.
    uint32_t n = 0, m = 0;
    for (uint32_t nb = blockIdx.x * 256u + threadIdx.x; nb < a.nblocks; nb += gridDim.x * 256u)
    {
        uint8_t* lst = a.lists + uint64_t(nb) * LIST_BYTES;
        const bool hasAlpha = lst[32] != 0;
        bool first = hasAlpha;
        if (!first)
        {
            const uint4* px4 = reinterpret_cast<const uint4*>(a.px + uint64_t(nb) * 16);
            const uint4 q0 = px4[0], q1 = px4[1], q2 = px4[2], q3 = px4[3];
            const uint32_t px[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
            const int ea = *reinterpret_cast<const int*>(lst + 56);
            first = int64_t(subset_lower_bound(px, 0xFFFFu, 0u, 4)) * 100 <= int64_t(ea) * a.early6Pct;
        }
        lst[37] = first ? 1 : 0;
        n += first ? 1u : 0u;
        m += hasAlpha ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n += __shfl_xor(n, o); m += __shfl_xor(m, o); }
    if ((threadIdx.x & 63u) == 0 && n) atomicAdd(out, n);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(out + 1, m);
}

// BC7_QUICK skips the rough pass; the search kernels still need the packed texels.
__global__ void __launch_bounds__(256) bc7_texels_kernel(Bc7Args a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.nblocks * 16u) return;
    float f4[4]; uint32_t ldr;
    const BcSeg& sg = seg_of(a.seg, i >> 4);
    load_block_texel(sg.src, sg.nbw, sg.nb0 + ((i >> 4) - sg.l0), i & 15u, f4, ldr);
    a.px[i] = ldr;
    if ((i & 15u) == 0) { a.zeroOrd[i >> 4] = 0xFFFFFFFFu; a.bestErr[i >> 4] = 0x7FFFFFFF; }
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
        if (s == SLOT_M7 && a.flagged[1] == 0) continue;       // no block has alpha: mode 7's (gated) launch wrote nothing
        const Cand v = c[s];
        if (v.err == 0xFFFFFFFFu) continue;
        const uint64_t key = (uint64_t(v.err) << 32) | v.ord;
        if (key < bestKey) { bestKey = key; lo = v.lo; hi = v.hi; }
    }
    const BcSeg& sg = seg_of(a.seg, nb);
    const uint32_t gb = sg.nb0 + (nb - sg.l0);
    const uint32_t by = gb / sg.nbw, bx = gb - by * sg.nbw;
    uint64_t* out = reinterpret_cast<uint64_t*>(sg.dst + uint64_t(by) * sg.dstRowPitch) + 2 * uint64_t(bx);
    out[0] = lo; out[1] = hi;
}
} // namespace

// Scratch layout for a pass over `nb` blocks (every array 256-byte aligned).
namespace
{
// bounds the scratch (about 1.1 KiB per block); DXTEX_MAX_BLOCKS_PER_PASS shrinks it so tests can exercise the pass loop
const uint64_t kMaxBlocksPerPass = dev_env("DXTEX_MAX_BLOCKS_PER_PASS") ? std::max<uint64_t>(1, strtoull(dev_env("DXTEX_MAX_BLOCKS_PER_PASS"), nullptr, 10)) : (1u << 22);
constexpr int kMaxTasksPerBlock = 64;                 // mode 2: 16 candidates x 4 lanes
// A submission of at most this many blocks (a lone image up to 2048^2, a mip chain) is a LATENCY problem: every kernel of the per-mode
// pipelines has fewer tasks than the machine has lanes and lasts as long as its longest serial chain, so the modes - independent until
// `pick`, see launch_bc7_encode_many - run side by side on the context's side streams, each pipeline with task arrays of its own.
const uint64_t kSmallPassBlocks = dev_env("DXTEX_BC7_SMALL_BLOCKS") ? strtoull(dev_env("DXTEX_BC7_SMALL_BLOCKS"), nullptr, 10) : 262144;     // up to a lone 2048^2 image (1448^2: 22.1 -> 20.4 ms, 2048^2: 39.2 -> 38.7; at 4096^2 the plan loses: 139.5 -> 143.7)
struct ScratchLayout
{
    size_t lists, cands, px, recs, order, tinfo, counters, zeroOrd, bestErr, seeds, seeds1, seeds3, flagcnt, auxRecs, auxOrder, auxTinfo, auxCounters, total;
    size_t auxTpb, auxSlices;     // the extra pipelines' task arrays: slices of nb * auxTpb tasks
    explicit ScratchLayout(uint64_t nb, bool threeSubsets)
    {
        auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
        const size_t tpb = threeSubsets ? kMaxTasksPerBlock : 32;
        // large passes: modes 4 (both index modes) and 5 side by side, 4 tasks per block each; small submissions: any mode on any side stream
        const bool small = nb <= kSmallPassBlocks;
        auxTpb = small ? tpb : 4; auxSlices = small ? size_t(kSideStreams) : 2;
        size_t o = 0;
        lists = o; o = up(o + nb * LIST_BYTES);
        cands = o; o = up(o + nb * NUM_SLOTS * sizeof(Cand));
        px = o; o = up(o + nb * 64);
        recs = o; o = up(o + nb * tpb * sizeof(TaskRec));
        order = o; o = up(o + nb * tpb * sizeof(uint2));
        tinfo = o; o = up(o + nb * tpb * sizeof(uint32_t));
        counters = o; o = up(o + 64 * sizeof(uint32_t));
        zeroOrd = o; o = up(o + nb * sizeof(uint32_t));
        bestErr = o; o = up(o + nb * sizeof(int));
        seeds = o; o = up(o + nb * 64 * sizeof(uint2));
        seeds1 = o; o = up(o + nb * 2 * sizeof(uint2));
        seeds3 = o; o = up(o + (threeSubsets ? nb * 192 * sizeof(uint2) : 0));
        flagcnt = o; o = up(o + 256);
        auxRecs = o; o = up(o + auxSlices * nb * auxTpb * sizeof(TaskRec));
        auxOrder = o; o = up(o + auxSlices * nb * auxTpb * sizeof(uint2));
        auxTinfo = o; o = up(o + auxSlices * nb * auxTpb * sizeof(uint32_t));
        auxCounters = o; o = up(o + size_t(kSideStreams) * 64 * sizeof(uint32_t));
        total = o;
    }
};

template<int MODE, int IM>
void launch_mode(const Bc7Args& a0, hipStream_t stream, KernelMarks* marks, const char* const (&names)[7])
{
    typedef TaskMap<MODE, IM> TM;
    constexpr int BPW = (TM::TPB >= 64) ? 1 : 64 / TM::TPB;
    const uint32_t nb = a0.nblocks;
    const uint32_t ntasks = nb * uint32_t(TM::TPB);
    const uint32_t gridPP = (nb + 4 * BPW - 1) / (4 * BPW);
    // whole-block modes: the group-of-lanes kernels for short lists are launched next to the lane-per-task ones and the live count
    // (known on the device only) decides which of the two works; on lists that cannot be short they are not launched at all
    // (a list of n slots holds at most n live tasks and typically at least a quarter of them; mode 6's lists are pruned to a few per
    // cent on large images, so its pair is always launched: an empty launch costs ~5 us)
    constexpr bool kWhole = TM::NS == 1;
    constexpr uint32_t kExhMax = (MODE == 6) ? kWaveTaskMax : kWaveTaskMax45;
    static const bool noGroup = dev_env("DXTEX_BC7_NO_GROUP") != nullptr;      // development A/B: every list through the lane-per-task kernels
    const bool maybeShortP = kWhole && !noGroup && (MODE == 6 || ntasks <= 4u * kPerturbWaveMax);
    const bool maybeShortE = kWhole && !noGroup && (MODE == 6 || ntasks <= 4u * kExhMax);
    const uint32_t wavesP = std::min<uint32_t>(kSearchWaves, (ntasks + 1) / 2), wavesE = std::min<uint32_t>(kSearchWaves, ntasks);
    Bc7Args a = a0;
    // device-side skip of launches that own no block (flagged[] is written by bc7_flag_count_kernel; BC7_QUICK has neither flags nor phases)
    a.gate = nullptr; a.gateMin = 0;
    if (a0.flagged && !(a0.flags & BCF_BC7_QUICK))
    {
        if (MODE == 7) { a.gate = a0.flagged + 1; a.gateMin = 1; }                                               // blocks with alpha (phase_owns / task_geometry: lst[32])
        else if (a0.phase == PHASE_EARLY && (MODE == 4 || MODE == 5)) { a.gate = a0.flagged + 1; a.gateMin = std::max<uint32_t>(1u, a0.earlyAlphaMin); }
        else if (a0.phase == PHASE_EARLY && MODE == 6) { a.gate = a0.flagged; a.gateMin = std::max<uint32_t>(1u, a0.early6Min); }
    }
    a.perturbWaveMax = maybeShortP ? kPerturbWaveMax : 0u;
    a.exhWaveMax = maybeShortE ? kExhMax : 0u;
    if (marks) marks->mark(names[0]);
    hipLaunchKernelGGL((bc7_pre_kernel<MODE, IM>), dim3(gridPP), dim3(256), 0, stream, a);
    if (marks) marks->mark(names[1]);
    (void)hipMemsetAsync(a.counters, 0, 64 * sizeof(uint32_t), stream);
    const uint32_t binGroups = std::min<uint32_t>(kBinGroups, (ntasks + 255) / 256);
    hipLaunchKernelGGL(bc7_bin_count_kernel, dim3(binGroups), dim3(256), 0, stream, a.tinfo, ntasks, a.counters, a.gate, a.gateMin);
    hipLaunchKernelGGL(bc7_bin_scan_kernel, dim3(1), dim3(1), 0, stream, a.counters);
    hipLaunchKernelGGL(bc7_bin_scatter_kernel, dim3(binGroups), dim3(256), 0, stream, a.tinfo, ntasks, a.counters, a.order, a.gate, a.gateMin);
    if (marks) marks->mark(names[2]);
    const uint32_t waves = std::min<uint32_t>(kSearchWaves, (ntasks + 63) / 64);
    static const int tailBelow = dev_env("DXTEX_BC7_TAIL_BELOW") ? atoi(dev_env("DXTEX_BC7_TAIL_BELOW")) : 48;
    static const bool perturbPlain = dev_env("DXTEX_BC7_PERTURB_PLAIN") != nullptr;      // A/B: PerturbOne without the bound filter
    // Exhaustive's interval tests per round (see the kernel); DXTEX_BC7_NO_PEEL = whole-window tests only (three of them, as before round 5)
    static const int rangeTests = dev_env("DXTEX_BC7_NO_PEEL") ? 3 : ((dev_env("DXTEX_BC7_RANGE_TESTS") ? atoi(dev_env("DXTEX_BC7_RANGE_TESTS")) & 0xFF : DXTEX_RANGE_TESTS) |
                                                                       ((dev_env("DXTEX_BC7_PEEL_LAYERS") ? atoi(dev_env("DXTEX_BC7_PEEL_LAYERS")) : 1) << 8));
    if constexpr (PaletteBits<MODE, IM>::AB == 0)
    {
        // the filter pays where the exact evaluation is dearest - eight palette entries on subsets of ~8 texels (mode 1: 24.5 -> 21.7 ms
        // per 4096^2 image) and sixteen entries on whole blocks (mode 6: 3.8 -> 3.3 ms, BC7_QUICK 12.4 -> 11.6 ms); with four entries
        // (modes 3, 5, 7, mode 4's 2-bit colours) its list handling costs more than it saves
        if constexpr (MODE == 1 || MODE == 6)
        {
            if (perturbPlain) hipLaunchKernelGGL((bc7_perturb_kernel<MODE, IM, CH_ALL>), dim3(waves), dim3(64), 0, stream, a, 0);
            else hipLaunchKernelGGL((bc7_perturb_filter_kernel<MODE, IM, CH_ALL>), dim3(waves), dim3(64), 0, stream, a, 0);
        }
        else
            hipLaunchKernelGGL((bc7_perturb_kernel<MODE, IM, CH_ALL>), dim3(waves), dim3(64), 0, stream, a, 0);
        if constexpr (kWhole) { if (maybeShortP) hipLaunchKernelGGL((bc7_perturb_wave_kernel<MODE, IM, CH_ALL>), dim3(wavesP), dim3(64), 0, stream, a); }
        if (marks) marks->mark(names[4]);
        if constexpr (kWhole) { if (maybeShortE) hipLaunchKernelGGL((bc7_exhaustive_wave_kernel<MODE, IM, CH_ALL>), dim3(wavesE), dim3(64), 0, stream, a); }
        hipLaunchKernelGGL((bc7_exhaustive_kernel<MODE, IM, CH_ALL>), dim3(waves), dim3(64), 0, stream, a, 1, tailBelow, rangeTests);
    }
    else
    {
        // mode 4's 3-bit colours (index mode 1): eight entries on sixteen texels, the dearest exact evaluation after mode 6's - through the
        // filter as well (3.98 -> 3.67 ms on the benchmark image)
        if constexpr (MODE == 4 && IM == 1 && DXTEX_PF_M4 != 0)
        {
            if (perturbPlain) hipLaunchKernelGGL((bc7_perturb_kernel<MODE, IM, CH_COLOR>), dim3(waves), dim3(64), 0, stream, a, 0);
            else hipLaunchKernelGGL((bc7_perturb_filter_kernel<MODE, IM, CH_COLOR>), dim3(waves), dim3(64), 0, stream, a, 0);
        }
        else
            hipLaunchKernelGGL((bc7_perturb_kernel<MODE, IM, CH_COLOR>), dim3(waves), dim3(64), 0, stream, a, 0);
        if (maybeShortP) hipLaunchKernelGGL((bc7_perturb_wave_kernel<MODE, IM, CH_COLOR>), dim3(wavesP), dim3(64), 0, stream, a);
        if (marks) marks->mark(names[3]);
        hipLaunchKernelGGL((bc7_perturb_kernel<MODE, IM, CH_ALPHA>), dim3(waves), dim3(64), 0, stream, a, 1);
        if (maybeShortP) hipLaunchKernelGGL((bc7_perturb_wave_kernel<MODE, IM, CH_ALPHA>), dim3(wavesP), dim3(64), 0, stream, a);
        if (marks) marks->mark(names[4]);
        if (maybeShortE) hipLaunchKernelGGL((bc7_exhaustive_wave_kernel<MODE, IM, CH_COLOR>), dim3(wavesE), dim3(64), 0, stream, a);
        hipLaunchKernelGGL((bc7_exhaustive_kernel<MODE, IM, CH_COLOR>), dim3(waves), dim3(64), 0, stream, a, 2, tailBelow, rangeTests);
        if (marks) marks->mark(names[5]);
        if (maybeShortE) hipLaunchKernelGGL((bc7_exhaustive_wave_kernel<MODE, IM, CH_ALPHA>), dim3(wavesE), dim3(64), 0, stream, a);
        hipLaunchKernelGGL((bc7_exhaustive_kernel<MODE, IM, CH_ALPHA>), dim3(waves), dim3(64), 0, stream, a, 3, tailBelow, rangeTests);
    }
    if (marks) marks->mark(names[6]);
    hipLaunchKernelGGL((bc7_post_kernel<MODE, IM>), dim3(gridPP), dim3(256), 0, stream, a);
#if defined(DXTEX_DEV)
    // development statistics (DXTEX_BC7_STATS): how many of the mode's tasks survived pre, by subset size
    static const bool stats = dev_env("DXTEX_BC7_STATS") != nullptr;
    if (stats)
    {
        uint32_t c[40] = {};
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(c, a.counters, sizeof(c), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "bc7 stats %-22s phase %d: %9u task slots, %9u live; by size 16..1:", names[0], a.phase, ntasks, c[34]);
        for (int b = 16; b >= 1; --b) std::fprintf(stderr, " %u", c[b]);
        std::fprintf(stderr, "\n");
    }
#endif
}
} // namespace

#if defined(DXTEX_BC7_WAVETIMES)
namespace
{
void bc7_wavetimes_report()
{
    (void)hipDeviceSynchronize();
    static int call = 0;
    static const int wanted = dev_env("DXTEX_BC7_WT_CALL") ? atoi(dev_env("DXTEX_BC7_WT_CALL")) : 2;      // which submission of the process is reported
    const bool print = call++ == wanted;
    static std::vector<unsigned long long> h(size_t(WT_KEYS) * 8192 * WT_FIELDS);
    static std::vector<unsigned int> hist(size_t(WT_KEYS) * 256);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_wt), h.size() * 8);
    (void)hipMemcpyFromSymbol(hist.data(), HIP_SYMBOL(g_wtHist), hist.size() * 4);
    unsigned long long tAll = ~0ull;
    for (size_t i = 0; i < size_t(WT_KEYS) * 8192; ++i) if (h[i * WT_FIELDS + 2]) tAll = std::min(tAll, h[i * WT_FIELDS]);
    static const char* const kinds[3] = { "perturb", "perturb_filter", "exhaustive" };
    for (int key = 0; key < WT_KEYS; ++key)
    {
        const unsigned long long* w = h.data() + size_t(key) * 8192 * WT_FIELDS;
        std::vector<unsigned long long> st, dr, en; unsigned long long rounds = 0, lanes = 0, life = 0, maxR = 0, tasks = 0, maxT = 0;
        for (int i = 0; i < 8192; ++i)
            if (w[i * WT_FIELDS + 2])
            {
                st.push_back(w[i * WT_FIELDS]); dr.push_back(w[i * WT_FIELDS + 1]); en.push_back(w[i * WT_FIELDS + 2]);
                rounds += w[i * WT_FIELDS + 3]; lanes += w[i * WT_FIELDS + 4]; tasks += w[i * WT_FIELDS + 5];
                life += w[i * WT_FIELDS + 2] - w[i * WT_FIELDS]; maxR = std::max(maxR, w[i * WT_FIELDS + 3]); maxT = std::max(maxT, w[i * WT_FIELDS + 5]);
            }
        if (st.empty() || !print) continue;
        const unsigned long long t0 = *std::min_element(st.begin(), st.end()), t1 = *std::max_element(en.begin(), en.end());
        std::sort(st.begin(), st.end()); std::sort(dr.begin(), dr.end()); std::sort(en.begin(), en.end());
        const double dur = double(t1 - t0); const size_t n = st.size();
        auto pct = [&](const std::vector<unsigned long long>& v, double p) { return 100.0 * double(v[std::min(n - 1, size_t(p * n))] - t0) / dur; };
        const unsigned int* hh = hist.data() + size_t(key) * 256;
        unsigned long long nt = 0, rt = 0; for (int r = 0; r < 256; ++r) { nt += hh[r]; rt += (unsigned long long)hh[r] * r; }
        auto hp = [&](double p) { unsigned long long c = 0; for (int r = 0; r < 256; ++r) { c += hh[r]; if (double(c) >= p * double(nt)) return r; } return 255; };
        unsigned long long rLong = 0; for (int r = 2 * hp(0.9) + 1; r < 256; ++r) if (r >= 0) rLong += (unsigned long long)hh[r] * r;
        const int chset = key % 3, im = (key / 3) % 2, mode = (key / 6) % 8, kind = key / 48;
        std::fprintf(stderr, "wt %s<%d,%d,%d>: at %.2f ms, span %.3f ms, %zu wavefronts worked; starts p50 %.0f p90 %.0f; saw the queue drained p10 %.0f p50 %.0f p90 %.0f; ends p10 %.0f p50 %.0f p90 %.0f p99 %.0f (%% of span); "
                     "mean lifetime %.0f %%; rounds / wavefront mean %.1f max %llu, busy lanes / round %.1f; tasks / wavefront mean %.0f max %llu; "
                     "finished tasks %llu, rounds / task mean %.2f p50 %d p90 %d p99 %d p99.9 %d, share of task-rounds in tasks above 2 x p90: %.1f %%\n",
                     kinds[kind], mode, im, chset, double(t0 - tAll) / 1e5, dur / 1e5, n, pct(st, 0.5), pct(st, 0.9), pct(dr, 0.1), pct(dr, 0.5), pct(dr, 0.9),
                     pct(en, 0.1), pct(en, 0.5), pct(en, 0.9), pct(en, 0.99), 100.0 * double(life) / (dur * n), double(rounds) / n, maxR, rounds ? double(lanes) / rounds : 0.0,
                     double(tasks) / n, maxT, nt, nt ? double(rt) / nt : 0.0, hp(0.5), hp(0.9), hp(0.99), hp(0.999), rt ? 100.0 * double(rLong) / rt : 0.0);
        std::vector<int> idx; for (int i = 0; i < 8192; ++i) if (w[i * WT_FIELDS + 2]) idx.push_back(i);
        std::sort(idx.begin(), idx.end(), [&](int x, int y) { return w[x * WT_FIELDS + 2] > w[y * WT_FIELDS + 2]; });
        for (size_t k = 0; k < std::min<size_t>(4, idx.size()); ++k)
        {
            const unsigned long long* e = w + idx[k] * WT_FIELDS;
            std::fprintf(stderr, "    late wavefront %d: start %.0f drained %.0f end %.0f %% of span, rounds %llu, busy lanes / round %.1f, tasks %llu\n", idx[k],
                         100.0 * double(e[0] - t0) / dur, 100.0 * double(e[1] - t0) / dur, 100.0 * double(e[2] - t0) / dur, e[3], e[3] ? double(e[4]) / e[3] : 0.0, e[5]);
        }
        {
            // median wavefront by end time, for comparison
            const unsigned long long* e = w + idx[idx.size() / 2] * WT_FIELDS;
            std::fprintf(stderr, "    median wavefront %d: start %.0f drained %.0f end %.0f %% of span, rounds %llu, busy lanes / round %.1f, tasks %llu\n", idx[idx.size() / 2],
                         100.0 * double(e[0] - t0) / dur, 100.0 * double(e[1] - t0) / dur, 100.0 * double(e[2] - t0) / dur, e[3], e[3] ? double(e[4]) / e[3] : 0.0, e[5]);
        }
    }
    std::fill(h.begin(), h.end(), 0ull); std::fill(hist.begin(), hist.end(), 0u);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wt), h.data(), h.size() * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wtHist), hist.data(), hist.size() * 4);
}
} // namespace
#endif

size_t bc7_scratch_bytes(uint64_t nblocks, uint32_t flags, size_t nimages)
{
    // monotone in the number of blocks (a context prepared for a larger submission never reallocates for a smaller one): the small
    // layout's side pipelines have full-size task arrays, so a lone 2048^2 image needs more than a slightly larger submission would
    const bool three = (flags & BCF_USE_3SUBSETS) != 0;
    size_t need = ScratchLayout(nblocks < kMaxBlocksPerPass ? nblocks : kMaxBlocksPerPass, three).total;
    if (nblocks > kSmallPassBlocks) need = std::max(need, ScratchLayout(kSmallPassBlocks, three).total);
    return need + seg_table_bytes(nblocks, kMaxBlocksPerPass, nimages);
}

hipError_t launch_bc7_encode(const SrcView& src, uint8_t* dst, uint64_t dstRowPitch, uint32_t flags,
                             void* scratch, hipStream_t stream, KernelMarks* marks, const SideStreams* side)
{
    BcImage one; one.src = src; one.dst = dst; one.dstRowPitch = dstRowPitch;
    return launch_bc7_encode_many(&one, 1, flags, scratch, stream, marks, side);
}

hipError_t launch_bc7_encode_many(const BcImage* images, size_t count, uint32_t flags, void* scratch, hipStream_t stream, KernelMarks* marks,
                                  const SideStreams* side)
{
#define DXTEX_MARK(NAME) do { if (marks) marks->mark(NAME); } while (0)
#define DXTEX_MODE(MODE, IM, TAG) do { static const char* const n_[7] = { "bc7_pre_" TAG, "bc7_bin_" TAG, "bc7_perturb_" TAG, "bc7_perturb_alpha_" TAG, \
                                           "bc7_exhaustive_" TAG, "bc7_exhaustive_alpha_" TAG, "bc7_post_" TAG }; \
                                       launch_mode<MODE, IM>(a, stream, marks, n_); } while (0)
    std::vector<BcSeg> segs;
    std::vector<BcPass> passes;
    uint64_t perPass = 0;
    if (!build_passes(images, count, kMaxBlocksPerPass, segs, passes, &perPass)) return hipSuccess;
    const bool three = (flags & BCF_USE_3SUBSETS) != 0;
    const ScratchLayout L(perPass, three);
    uint8_t* base = static_cast<uint8_t*>(scratch);
    BcSeg* dSegs = reinterpret_cast<BcSeg*>(base + L.total);
    const hipError_t ce = upload_segments(dSegs, segs, passes, stream);
    if (ce != hipSuccess) return ce;
    const bool quick = (flags & BCF_BC7_QUICK) != 0;

    for (const BcPass& pass : passes)
    {
        Bc7Args a;
        set_pass(a.seg, dSegs, segs, pass);
        a.nblocks = pass.nblocks;
        a.flags = flags;
        a.perturbWaveMax = 0; a.exhWaveMax = 0;       // set per mode by launch_mode
        a.gate = nullptr; a.gateMin = 0;
        a.lists = base + L.lists;
        a.cands = reinterpret_cast<Cand*>(base + L.cands);
        a.px = reinterpret_cast<uint32_t*>(base + L.px);
        a.recs = reinterpret_cast<TaskRec*>(base + L.recs);
        a.order = reinterpret_cast<uint2*>(base + L.order);
        a.tinfo = reinterpret_cast<uint32_t*>(base + L.tinfo);
        a.counters = reinterpret_cast<uint32_t*>(base + L.counters);
        a.zeroOrd = reinterpret_cast<uint32_t*>(base + L.zeroOrd);
        a.bestErr = reinterpret_cast<int*>(base + L.bestErr);
        a.seeds = reinterpret_cast<uint2*>(base + L.seeds);
        a.seeds1 = reinterpret_cast<uint2*>(base + L.seeds1);
        a.seeds3 = reinterpret_cast<uint2*>(base + L.seeds3);
        static const bool noPrune = dev_env("DXTEX_BC7_NO_PRUNE") != nullptr;
        a.prune = noPrune ? 0 : 1;
        static const int early6 = dev_env("DXTEX_BC7_EARLY6_PCT") ? atoi(dev_env("DXTEX_BC7_EARLY6_PCT")) : 100;
        a.early6Pct = early6;
        uint32_t slotMask = 0;

        uint32_t* flagCount = reinterpret_cast<uint32_t*>(base + L.flagcnt);
        a.flagged = flagCount;
        // Mode 6's early phase costs its own search (5 ms for the 42 % of the benchmark image's blocks it takes) and buys pruning in modes 1 / 3; since
        // round 5's cheaper Exhaustive it only pays when at least half of the blocks are in it (131.8 -> 130.7 ms without it on the benchmark image)
        // (a small submission runs the early phase on a side stream of the plan below, next to modes 1 / 3: there it costs nothing and a quarter
        // of the blocks is enough - lone 1448^2 / 2048^2 images 20.2 / 37.7 -> 18.9 / 36.2 ms)
        static const int early6MinPct = dev_env("DXTEX_BC7_EARLY6_MIN_PCT") ? atoi(dev_env("DXTEX_BC7_EARLY6_MIN_PCT")) : -1;
        static const int earlyAlphaMinPct = dev_env("DXTEX_BC7_EARLYA_MIN_PCT") ? atoi(dev_env("DXTEX_BC7_EARLYA_MIN_PCT")) : 25;
        const bool smallPlanRuns = side && !marks && !quick && dev_env("DXTEX_BC7_SERIAL") == nullptr && dev_env("DXTEX_BC7_NO_SMALL_PLAN") == nullptr &&
                                   L.auxTpb >= 32 && perPass <= kSmallPassBlocks && dev_env("DXTEX_BC7_ORDER") == nullptr;
        a.early6Min = uint32_t(uint64_t(a.nblocks) * uint32_t(early6MinPct >= 0 ? early6MinPct : (smallPlanRuns ? 25 : 50)) / 100u);
        a.earlyAlphaMin = uint32_t(uint64_t(a.nblocks) * uint32_t(earlyAlphaMinPct) / 100u);
        if (!quick)
        {
            DXTEX_MARK("bc7_rough");
            hipLaunchKernelGGL(bc7_rough_kernel, dim3((a.nblocks + 3) / 4), dim3(256), 0, stream, a);
            (void)hipMemsetAsync(flagCount, 0, 256, stream);
            hipLaunchKernelGGL(bc7_flag_count_kernel, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a, flagCount);
        }
        else
        {
            DXTEX_MARK("bc7_texels");
            hipLaunchKernelGGL(bc7_texels_kernel, dim3((a.nblocks * 16 + 255) / 256), dim3(256), 0, stream, a);
        }
        DXTEX_MARK("bc7_block_seeds");
        if (!quick) hipLaunchKernelGGL(bc7_block_seeds_kernel<false>, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(bc7_block_seeds_kernel<true>, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a);
        // The modes are independent until `pick` (which restores Encode's order through the candidates' keys), so they may run
        // in any order; what the order changes is how early a good error is on the table for subset_lower_bound to prune with.
        // Step codes: mode number (8 = mode 4 with index mode 1), +10 = the early half of a split mode, +20 = its late half.
        // Default: mode 6 for the blocks the rough pass flagged, then the alpha-carrying modes for blocks that have alpha, then
        // the reference's order for everything else. With BC7_QUICK only mode 6 exists and nothing is split.
        static const std::vector<int> order = []
        {
            std::vector<int> o;
            const char* e = dev_env("DXTEX_BC7_ORDER");
            const char* p = e ? e : "16,7,14,18,15,0,1,2,3,24,28,25,26";
            while (*p)
            {
                if (*p >= '0' && *p <= '9') { o.push_back(int(strtol(p, const_cast<char**>(&p), 10))); }
                else ++p;
            }
            return o;
        }();
        a.phase = PHASE_ALL;
        if (quick) { DXTEX_MODE(6, 0, "mode6"); slotMask |= 1u << SLOT_M6; }
        auto run_step = [&](int step, Bc7Args a, hipStream_t stream, KernelMarks* marks)
        {
            a.phase = (step >= 20) ? PHASE_LATE : (step >= 10) ? PHASE_EARLY : PHASE_ALL;
            switch (step)
            {
            case 0: DXTEX_MODE(0, 0, "mode0"); slotMask |= 1u << SLOT_M0; break;
            case 1: DXTEX_MODE(1, 0, "mode1"); slotMask |= 1u << SLOT_M1; break;
            case 2: DXTEX_MODE(2, 0, "mode2"); slotMask |= 1u << SLOT_M2; break;
            case 3: DXTEX_MODE(3, 0, "mode3"); slotMask |= 1u << SLOT_M3; break;
            case 4: DXTEX_MODE(4, 0, "mode4_im0"); slotMask |= 1u << SLOT_M4A; break;
            case 14: DXTEX_MODE(4, 0, "mode4_im0_early"); slotMask |= 1u << SLOT_M4A; break;
            case 24: DXTEX_MODE(4, 0, "mode4_im0_late"); slotMask |= 1u << SLOT_M4A; break;
            case 8: DXTEX_MODE(4, 1, "mode4_im1"); slotMask |= 1u << SLOT_M4B; break;
            case 18: DXTEX_MODE(4, 1, "mode4_im1_early"); slotMask |= 1u << SLOT_M4B; break;
            case 28: DXTEX_MODE(4, 1, "mode4_im1_late"); slotMask |= 1u << SLOT_M4B; break;
            case 5: DXTEX_MODE(5, 0, "mode5"); slotMask |= 1u << SLOT_M5; break;
            case 15: DXTEX_MODE(5, 0, "mode5_early"); slotMask |= 1u << SLOT_M5; break;
            case 25: DXTEX_MODE(5, 0, "mode5_late"); slotMask |= 1u << SLOT_M5; break;
            case 6: DXTEX_MODE(6, 0, "mode6"); slotMask |= 1u << SLOT_M6; break;
            case 16: DXTEX_MODE(6, 0, "mode6_early"); slotMask |= 1u << SLOT_M6; break;
            case 26: DXTEX_MODE(6, 0, "mode6_late"); slotMask |= 1u << SLOT_M6; break;
            case 7: DXTEX_MODE(7, 0, "mode7"); slotMask |= 1u << SLOT_M7; break;
            default: break;
            }
        };
        // Modes 4 (index modes 0 and 1) and 5 of one phase have four tasks per block each, long search chains and search kernels that end
        // with tails of a few busy wavefronts: the three pipelines are independent until `pick` (each has its own candidate slot; the
        // per-block best error / first-zero key they share are updated with atomics and only ever prune work that cannot matter), so
        // consecutive steps of that family run side by side on two extra streams, each pipeline on its own slice of the task arrays.
        // Per-kernel timing (marks) needs one stream and keeps them serial.
        auto family45 = [](int step) { const int m = step % 10; return m == 4 || m == 8 || m == 5; };
        static const bool serial45 = dev_env("DXTEX_BC7_SERIAL") != nullptr;
        const SideStreams* fork = (marks || serial45 || quick) ? nullptr : side;
        // the task arrays of side pipeline k (0-based)
        auto side_args = [&](const Bc7Args& a0, size_t k)
        {
            Bc7Args b = a0;
            b.recs = reinterpret_cast<TaskRec*>(base + L.auxRecs) + k * size_t(a0.nblocks) * L.auxTpb;
            b.order = reinterpret_cast<uint2*>(base + L.auxOrder) + k * size_t(a0.nblocks) * L.auxTpb;
            b.tinfo = reinterpret_cast<uint32_t*>(base + L.auxTinfo) + k * size_t(a0.nblocks) * L.auxTpb;
            b.counters = reinterpret_cast<uint32_t*>(base + L.auxCounters) + k * 64;
            return b;
        };
        // Small submissions: the whole schedule as four concurrent pipelines (the context's stream + three side streams), joined before
        // `pick`. What an earlier mode leaves on the table for a later one's pruning is given up where the two now run at the same time -
        // never the result: every bound is exact, `pick` restores Encode's order, bestErr / zeroOrd are folded with atomicMin - and the
        // search work that adds is free on a machine that a small submission cannot fill. Mode 6's late phase, which lives off the
        // table (97 % pruned), still runs last on the main stream; mode 3 gets a stream to itself next to mode 1.
        // The plan: stages separated by '|' (joined on the main stream), pipelines of a stage by '/', steps by ','. Measured on MI355X (rocprofv3
        // timeline of a lone 512^2 image): modes 1 and 3 next to each other take 2.0 ms instead of 1.6 + 1.1 one after the other; a FOURTH
        // hardware queue does not start before the first three drain, so a stage has at most three pipelines; the late single-subset modes
        // keep a three-way stage of their own after modes 1 / 3 (next to them their kernels starved behind the persistent search waves:
        // 1.8 ms instead of 0.8). Of the dozen plans tried this one was the best at all three sizes - lone 256^2 / 512^2 / 1024^2 images:
        // 2.48 / 4.82 / 12.46 ms serial -> 1.73 / 3.78 / 11.39 ms (mode 6's early phase beside modes 1 / 3 instead of before them, its late
        // phase right after mode 1 on the main stream).
        static const std::vector<std::vector<std::vector<int>>> smallPlan = []
        {
            const char* e = dev_env("DXTEX_BC7_SMALL_PLAN");
            const char* p = e ? e : "1,26/3,2/16,14,15,18,7,0|24/28/25";
            std::vector<std::vector<std::vector<int>>> stages(1, std::vector<std::vector<int>>(1));
            while (*p)
            {
                if (*p >= '0' && *p <= '9') stages.back().back().push_back(int(strtol(p, const_cast<char**>(&p), 10)));
                else
                {
                    if (*p == '|') stages.emplace_back(1);
                    else if (*p == '/' && stages.back().size() < size_t(kSideStreams) + 1) stages.back().emplace_back();
                    ++p;
                }
            }
            return stages;
        }();
        static const bool noSmall = dev_env("DXTEX_BC7_NO_SMALL_PLAN") != nullptr;
        if (fork && !noSmall && smallPlanRuns)
        {
            for (const std::vector<std::vector<int>>& stage : smallPlan)
            {
                if (stage.size() > 1) (void)hipEventRecord(fork->forked, stream);
                for (size_t l = 1; l < stage.size(); ++l)
                {
                    hipStream_t s = fork->side[l - 1];
                    (void)hipStreamWaitEvent(s, fork->forked, 0);
                    const Bc7Args al = side_args(a, l - 1);
                    for (int step : stage[l])
                        if (three || (step != 0 && step != 2)) run_step(step, al, s, nullptr);
                    (void)hipEventRecord(fork->joined[l - 1], s);
                }
                for (int step : stage[0])
                    if (three || (step != 0 && step != 2)) run_step(step, a, stream, nullptr);
                for (size_t l = 1; l < stage.size(); ++l) (void)hipStreamWaitEvent(stream, fork->joined[l - 1], 0);
            }
        }
        else
        for (size_t at = 0; at < order.size() && !quick; )
        {
            const int step = order[at];
            if (!three && (step == 0 || step == 2)) { ++at; continue; }
            size_t run = 1;
            if (fork && family45(step))
                while (at + run < order.size() && run < 3 && family45(order[at + run]) && order[at + run] / 10 == step / 10) ++run;
            if (run == 1) { run_step(step, a, stream, marks); ++at; continue; }
            (void)hipEventRecord(fork->forked, stream);
            for (size_t k = 1; k < run; ++k)
            {
                const Bc7Args b = side_args(a, k - 1);
                (void)hipStreamWaitEvent(fork->side[k - 1], fork->forked, 0);
                run_step(order[at + k], b, fork->side[k - 1], nullptr);
                (void)hipEventRecord(fork->joined[k - 1], fork->side[k - 1]);
            }
            run_step(step, a, stream, nullptr);
            for (size_t k = 1; k < run; ++k) (void)hipStreamWaitEvent(stream, fork->joined[k - 1], 0);
            at += run;
        }
#if defined(DXTEX_EXH_STATS)
        hipLaunchKernelGGL(bc7_stats_print_kernel, dim3(1), dim3(1), 0, stream, reinterpret_cast<unsigned long long*>(flagCount) + 4);
#endif
        DXTEX_MARK("bc7_pick");
        hipLaunchKernelGGL(bc7_pick_kernel, dim3((a.nblocks + 255) / 256), dim3(256), 0, stream, a, slotMask);
    }
    DXTEX_MARK(nullptr);
#undef DXTEX_MODE
#undef DXTEX_MARK
#if defined(DXTEX_BC7_WAVETIMES)
    bc7_wavetimes_report();
#endif
    return hipGetLastError();
}
} // namespace dxtex
