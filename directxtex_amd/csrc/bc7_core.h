// BC7 encoder core for gfx950: everything one "task" (one region of one candidate shape, or one
// single-subset candidate) computes in a single lane. The kernels in bc7_encode.hip spread the tasks of
// a block over the lanes of a wavefront and reduce the results with cross-lane operations.
//
// What is computed is exactly the search of the reference CPU encoder (D3DX_BC7::Encode,
// BC6HBC7.cpp:2783-2889, with RoughMSE :3492-3597, Refine :3399-3463, FixEndpointPBits :3311-3396,
// AssignIndices :3139-3218, OptimizeEndPoints/OptimizeOne :3113-3136/:3045-3110, PerturbOne :2926-2966,
// Exhaustive :2971-3042, MapColors :3466-3489, ComputeError :1559-1635, EmitBlock :3221-3308), including
// its quirks, so the emitted blocks are byte-identical. How it is computed is not:
//   * errors are exact integers, so they are carried as int32 instead of fp32 sums; |p - q|^2 is expanded
//     to |p|^2 + |q|^2 - 2 p.q and evaluated with v_dot4_u32_u8 on packed RGBA bytes;
//   * MapColors' "running total exceeded the best so far" early-out only ever turns a losing candidate
//     into FLT_MAX, so it is dropped: a candidate wins iff its exact total is smaller;
//   * palettes are interpolated two channels at a time in 16-bit halves of a 32-bit lane register;
//   * the float Newton seed (OptimizeRGB/OptimizeRGBA, :1198-1555) is shared by every mode that uses the
//     same partition, since it does not depend on the mode.
// Only the seed uses floating point; it must round like the reference (compile with -ffp-contract=off).
#pragma once
#include <stdint.h>
#include <type_traits>

#if defined(DXTEX_HOST_DEBUG)
#define DXTEX_HD __host__ __device__ inline
#else
#define DXTEX_HD __device__ __forceinline__
#endif

namespace dxtex
{
namespace bc7
{
#if defined(DXTEX_HOST_DEBUG)
// host mirror only: lockstep_perturb_loop takes the flat-call shortcut in EVERY mode (so that the CPU suite checks the argument on two- and
// three-subset regions as well) unless this is false - then only where the kernels take it (kFlatSkip in bc7_encode.hip: mode 6's combined
// loop), which makes the mirror run the path the kernels run for modes 0 - 5. tools/bc7_debug.cpp sets it from DXTEX_HOST_FLAT_SKIP=kernel.
static bool g_hostFlatSkipEveryMode = true;
#endif
#if defined(DXTEX_COUNT_EVALS)
static long g_evalCount[8], g_evalTexels[8], g_macroCount[8], g_boundCount[8], g_pendCount[8], g_drainCount[8], g_pfTotal[8], g_pfPass[8], g_pfImprove[8], g_pfStepTotal[8][8], g_pfStepPass[8][8], g_tabWin[8], g_tabWinOut[8], g_tabWinOutPrev[8];
static int g_statTable = 0x7FFFFFFF, g_statTablePrev = 0x7FFFFFFF, g_statOther = 0;
static long g_rowCand[8], g_rowWin[8], g_rowTests[8][4], g_rowOut[8][4], g_rowDistN[8][6], g_rowDistOut[8][6], g_peelTests[8][4][3], g_peelOut[8][4][3];
#endif
// ---- per-mode constants (BC6HBC7.cpp:1106-1124) ---------------------------------------------------
template<int MODE> struct ModeInfo;
#define DXTEX_BC7_MODE(M, NS_, PARTBITS_, PB_, ROTBITS_, IMBITS_, IB_, IB2_, CP_, AP_, CPP_, APP_) \
    template<> struct ModeInfo<M> { enum : int { NS = NS_, PARTBITS = PARTBITS_, PB = PB_, ROTBITS = ROTBITS_, IMBITS = IMBITS_, \
        IB = IB_, IB2 = IB2_, CP = CP_, AP = AP_, CPP = CPP_, APP = APP_ }; }
// PB: 0 = no p-bits, 1 = one per endpoint, 2 = one shared per subset
DXTEX_BC7_MODE(0, 3, 4, 1, 0, 0, 3, 0, 4, 0, 5, 0);
DXTEX_BC7_MODE(1, 2, 6, 2, 0, 0, 3, 0, 6, 0, 7, 0);
DXTEX_BC7_MODE(2, 3, 6, 0, 0, 0, 2, 0, 5, 0, 5, 0);
DXTEX_BC7_MODE(3, 2, 6, 1, 0, 0, 2, 0, 7, 0, 8, 0);
DXTEX_BC7_MODE(4, 1, 0, 0, 2, 1, 2, 3, 5, 6, 5, 6);
DXTEX_BC7_MODE(5, 1, 0, 0, 2, 0, 2, 2, 7, 8, 7, 8);
DXTEX_BC7_MODE(6, 1, 0, 1, 0, 0, 4, 0, 7, 7, 8, 8);
DXTEX_BC7_MODE(7, 2, 6, 1, 0, 0, 2, 0, 5, 5, 6, 6);
#undef DXTEX_BC7_MODE

DXTEX_HD constexpr int weight(int bits, int i)
{
    // g_aWeights2/3/4 (BC6HBC7.cpp:327-329)
    return bits == 2 ? (i == 0 ? 0 : i == 1 ? 21 : i == 2 ? 43 : 64)
         : bits == 3 ? (i == 0 ? 0 : i == 1 ? 9 : i == 2 ? 18 : i == 3 ? 27 : i == 4 ? 37 : i == 5 ? 46 : i == 6 ? 55 : 64)
         : (i == 0 ? 0 : i == 1 ? 4 : i == 2 ? 9 : i == 3 ? 13 : i == 4 ? 17 : i == 5 ? 21 : i == 6 ? 26 : i == 7 ? 30
            : i == 8 ? 34 : i == 9 ? 38 : i == 10 ? 43 : i == 11 ? 47 : i == 12 ? 51 : i == 13 ? 55 : i == 14 ? 60 : 64);
}

DXTEX_HD uint32_t udot4(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, 0u, false);
#else
    return (a & 0xFF) * (b & 0xFF) + ((a >> 8) & 0xFF) * ((b >> 8) & 0xFF) + ((a >> 16) & 0xFF) * ((b >> 16) & 0xFF) + (a >> 24) * (b >> 24);
#endif
}

// 24-bit multiplies. Everything the palette arithmetic multiplies is a byte, a weight <= 64 or two bytes 16 bits apart - but the
// compiler cannot see that and emits full 32-bit multiplies (v_mul_lo_u32, v_mad_u64_u32: quarter rate on gfx950, ~16 issue cycles
// per wave64 against 4 for v_mul_u32_u24 / v_mad_i32_i24). Spelling the width out took ~100 cycles off every candidate of the
// search kernels. Operands must fit 24 bits (signed: 23 bits + sign); the low 32 bits of the product are returned.
DXTEX_HD uint32_t umul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
DXTEX_HD int mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}
// a * b + c with 24-bit factors (v_mad_i32_i24) and the middle of three (v_med3_i32: a clamp in one instruction)
DXTEX_HD int mad24(int a, int b, int c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b) + c;
#else
    return a * b + c;
#endif
}
DXTEX_HD int med3(int v, int lo, int hi)      // lo <= hi
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
#else
    return v < lo ? lo : (v > hi ? hi : v);
#endif
}
// One interpolated byte of LDRColorA::Interpolate (:384-416): (ua (64 - w) + ub w + 32) >> 6 = (64 ua + 32 + w (ub - ua)) >> 6.
// lerp_base / lerp_delta are per candidate, lerp1 is one v_mad_i32_i24 (w is an inline constant) and a shift per entry; the
// result is <= 255 without masking.
DXTEX_HD int lerp_base(uint32_t ua) { return int(ua << 6) + 32; }
DXTEX_HD int lerp_delta(uint32_t ua, uint32_t ub) { return int(ub) - int(ua); }
DXTEX_HD uint32_t lerp1(int base64, int delta, int w) { return uint32_t(mul24(delta, w) + base64) >> 6; }

DXTEX_HD uint32_t byte_of(uint32_t v, int ch) { return (v >> (8 * ch)) & 0xFFu; }
DXTEX_HD uint32_t with_byte(uint32_t v, int ch, uint32_t b) { return (v & ~(0xFFu << (8 * ch))) | (b << (8 * ch)); }

// swap channel (rot-1) with alpha: the BC7 rotation (BC6HBC7.cpp:2837-2843)
DXTEX_HD uint32_t rotate_pixel(uint32_t p, uint32_t rot)
{
    if (rot == 0) return p;
    const int sh = 8 * int(rot - 1);
    const uint32_t c = (p >> sh) & 0xFFu, a = p >> 24;
    return (p & ~((0xFFu << sh) | 0xFF000000u)) | (a << sh) | (c << 24);
}

// D3DX_BC7::Unquantize for one component at compile-time precision (:826-831). PREC == 0 -> 255.
template<int PREC>
DXTEX_HD uint32_t unq1(uint32_t c)
{
    if (PREC == 0) return 255u;
    if (PREC == 8) return c;
    // c is a quantised value, c < 2^PREC: the reference's "& 0xFF" never clears anything and (c << (8 - PREC)) >> PREC is
    // c >> (2 PREC - 8) - two instructions (shift, shift-or) instead of four, in every candidate of the search kernels
    static_assert(PREC == 0 || PREC >= 4, "endpoint precisions of BC7 incl. the p-bit are 5 ... 8");
    return (c << (8 - PREC)) | (c >> (2 * PREC - 8));
}

template<int MODE>
DXTEX_HD uint32_t unquantize(uint32_t ep)
{
    typedef ModeInfo<MODE> MI;
    const uint32_t r = unq1<MI::CPP>(ep & 0xFF), g = unq1<MI::CPP>((ep >> 8) & 0xFF), b = unq1<MI::CPP>((ep >> 16) & 0xFF), a = unq1<MI::APP>(ep >> 24);
    return r | (g << 8) | (b << 16) | (a << 24);
}

// One palette entry, two channels per 16-bit half (LDRColorA::Interpolate, :384-416):
// (c0 * (64 - w) + c1 * w + 32) >> 6 for each of the four bytes.
DXTEX_HD uint32_t lerp_bytes(uint32_t a, uint32_t b, int w)
{
    const uint32_t arb = a & 0x00FF00FFu, aga = (a >> 8) & 0x00FF00FFu;
    const uint32_t brb = b & 0x00FF00FFu, bga = (b >> 8) & 0x00FF00FFu;
    const uint32_t rb = ((umul24(arb, uint32_t(64 - w)) + umul24(brb, uint32_t(w)) + 0x00200020u) >> 6) & 0x00FF00FFu;
    const uint32_t ga = ((umul24(aga, uint32_t(64 - w)) + umul24(bga, uint32_t(w)) + 0x00200020u) >> 6) & 0x00FF00FFu;
    return rb | (ga << 8);
}

// Scores. Squared distances are compared through their negation with the per-texel constant |p|^2 dropped:
//   score_i = 2 p.q_i - |q_i|^2        (so |p - q_i|^2 = |p|^2 - score_i)
// which is two chained v_dot4_u32_u8 (the second accumulates onto the first; the accumulator starts at
// -|q_i|^2), all in wrapping 32-bit arithmetic.
DXTEX_HD uint32_t udot4acc(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return udot4(a, b) + c;
#endif
}
DXTEX_HD int score(uint32_t p, uint32_t q, uint32_t negq2) { return int(udot4acc(p, q, udot4acc(p, q, negq2))); }

// ComputeError scans the palette and stops at the first entry whose error is larger than the best so far
// (:1581-1595): it finds the first local minimum, i.e. (in scores) s[j-1] for the first j with
// s[j] < s[j-1], or s[N-1] if the scores never drop. Evaluated from the far end so that the N-1
// comparisons are independent of each other and of the selects (no serial compare -> mask-or -> select
// chain through the scalar unit).
template<int N>
DXTEX_HD int first_peak(const int (&sc)[N])
{
    int res = sc[N - 1];
#pragma unroll
    for (int i = N - 1; i >= 1; --i) res = (sc[i] < sc[i - 1]) ? sc[i - 1] : res;
    return res;
}

template<int N>
DXTEX_HD int first_peak_idx(const int (&sc)[N], uint32_t& idx)
{
    // the scan keeps the FIRST entry of a plateau (strict '<' on errors to replace the best). The scores do not fall before the scan stops, so
    // its result v = first_peak(sc) is the largest score of that stretch, the entries equal to v inside it are its last ones, and every entry
    // before them is smaller: the scan's index is the SMALLEST index whose score equals v (a later entry that climbs back to v has a larger
    // index). Two chains of independent compare / selects instead of one serial scan with a "done" flag.
    const int v = first_peak(sc);
    idx = uint32_t(N - 1);
#pragma unroll
    for (int i = N - 2; i >= 0; --i) idx = (sc[i] == v) ? uint32_t(i) : idx;
    return v;
}

// ---- the pixels one task sees ------------------------------------------------------------------------
// Region: a subset of a block whose packed RGBA8 texels live in LDS (`pix`); `order` lists the texel
// positions of the subset, 4 bits each, in increasing position order; `np` of them are valid. Used by the
// multi-subset modes, where the subset size differs from lane to lane.
struct Region
{
    enum : bool { kStatic = false };
    const uint32_t* pix;
    uint64_t order;
    int np;
    int p2sum;       // sum over the region of |p|^2 (all four channels)

    DXTEX_HD int count() const { return np; }
    DXTEX_HD uint32_t pos(int k) const { return uint32_t(order >> (4 * k)) & 15u; }
    DXTEX_HD uint32_t fetch(int k) const { return pix[pos(k)]; }
    DXTEX_HD uint32_t fetch_pos(uint32_t p) const { return pix[p]; }        // the texel at block position p (a member of the region)
};

DXTEX_HD void region_init(Region& r, const uint32_t* pix, uint32_t mask16)
{
    r.pix = pix; r.order = 0; r.np = 0; r.p2sum = 0;
    for (uint32_t i = 0; i < 16; ++i)
        if ((mask16 >> i) & 1u)
        {
            r.order |= uint64_t(i) << (4 * r.np);
            const uint32_t p = pix[i];
            r.p2sum += int(udot4(p, p));
            ++r.np;
        }
}

// SlotRegion: the subset's texels copied, in order, to a per-lane column of an LDS array laid out
// [k][lane], so that a wavefront's fetch(k) is one conflict-free ds_read_b32. Used by the persistent search
// loop, where every lane works on a different (block, shape, subset). Rows are kSlotStride = 65 dwords apart:
// with 64 a column would sit in ONE bank, and the lanes that score the texels of ANOTHER lane's column side by
// side (the exact-evaluation list of bc7_perturb_filter_kernel: eight lanes, eight texels of one owner) would
// collide eight ways - 40 % of that kernel's LDS cycles in round 4's counters.
#if !defined(DXTEX_SLOT_STRIDE)
#define DXTEX_SLOT_STRIDE 65
#endif
enum : int { kSlotStride = DXTEX_SLOT_STRIDE };
struct SlotRegion
{
    enum : bool { kStatic = false };
    const uint32_t* base;   // &slots[lane]
    int np;
    int p2sum;
    DXTEX_HD int count() const { return np; }
    DXTEX_HD uint32_t pos(int k) const { return uint32_t(k); }
    DXTEX_HD uint32_t fetch(int k) const { return base[k * kSlotStride]; }
};

// Block16: all 16 texels of a block held in the lane's registers (already rotated); used by the
// single-subset modes 4, 5, 6 where every task sees the whole block.
struct Block16
{
    enum : bool { kStatic = true };
    uint32_t px[16];
    int p2sum;
    DXTEX_HD int count() const { return 16; }
    DXTEX_HD uint32_t pos(int k) const { return uint32_t(k); }
    DXTEX_HD uint32_t fetch(int k) const { return px[k]; }
    DXTEX_HD uint32_t fetch_pos(uint32_t p) const { return px[p]; }
};

template<class PIX>
DXTEX_HD void block16_init(Block16& r, const PIX* pix, uint32_t rot)
{
    r.p2sum = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        r.px[i] = rotate_pixel(pix[i], rot);
        r.p2sum += int(udot4(r.px[i], r.px[i]));
    }
}

// Texel loops: fully unrolled for register-resident blocks, counted for LDS-resident regions.
template<class RG, class F>
DXTEX_HD void for_texels(const RG& rg, F&& f)
{
    if constexpr (RG::kStatic)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) f(k);
    }
    else
    {
        // two texels per trip: their compare/select chains interleave and hide the VCC write->read hazard
        const int n = rg.count();
        int k = 0;
        for (; k + 1 < n; k += 2) { f(k); f(k + 1); }
        if (k < n) f(k);
    }
}

// ---- palette + error of one endpoint pair over a region (GeneratePaletteQuantized + MapColors) --------
// Combined-index modes (IB2 == 0): 4-channel error. Separate-alpha modes: RGB error with the colour
// index precision plus alpha error with the alpha index precision (uIndexMode swaps the two).
template<int MODE, int IM>
struct PaletteBits
{
    typedef ModeInfo<MODE> MI;
    enum : int { CB = (MI::IB2 == 0) ? MI::IB : (IM ? MI::IB2 : MI::IB),      // colour (or combined) index bits
                 AB = (MI::IB2 == 0) ? 0 : (IM ? MI::IB : MI::IB2),           // alpha index bits (0 = combined)
                 NC = 1 << CB, NA = (AB ? (1 << AB) : 1) };
};

template<int MODE, int IM, class RG>
DXTEX_HD int map_colors(const RG& rg, uint32_t epA, uint32_t epB)
{
    typedef PaletteBits<MODE, IM> PB;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    int total = rg.p2sum;

    uint32_t pal[PB::NC], nq2[PB::NC];
#pragma unroll
    for (int i = 0; i < PB::NC; ++i)
    {
        pal[i] = lerp_bytes(ua, ub, weight(PB::CB, i));
        if (PB::AB != 0) pal[i] &= 0x00FFFFFFu;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    if (PB::AB == 0)
    {
        for_texels(rg, [&](int k)
        {
            const uint32_t p = rg.fetch(k);
            int sc[PB::NC];
#pragma unroll
            for (int i = 0; i < PB::NC; ++i) sc[i] = score(p, pal[i], nq2[i]);
            total -= first_peak(sc);
        });
    }
    else
    {
        int pa[PB::NA], npa2[PB::NA];
        const int a0 = int(ua >> 24), a1 = int(ub >> 24);
#pragma unroll
        for (int i = 0; i < PB::NA; ++i)
        {
            pa[i] = int(lerp1(lerp_base(uint32_t(a0)), a1 - a0, weight(PB::AB, i)));
            npa2[i] = -(pa[i] * pa[i]);
        }
        for_texels(rg, [&](int k)
        {
            const uint32_t p = rg.fetch(k);
            const uint32_t prgb = p & 0x00FFFFFFu;
            const int al2 = int(p >> 24) * 2;
            int sc[PB::NC];
#pragma unroll
            for (int i = 0; i < PB::NC; ++i) sc[i] = score(prgb, pal[i], nq2[i]);
            int su[PB::NA];
#pragma unroll
            for (int i = 0; i < PB::NA; ++i) su[i] = al2 * pa[i] + npa2[i];
            total -= first_peak(sc) + first_peak(su);
        });
    }
    return total;
}

// AssignIndices for one region (:3139-3218): error, per-texel indices (4 bits per texel *position*), and
// the anchor fix-up that swaps the endpoints when the anchor's index has its top bit set.
template<int MODE, int IM, class RG>
DXTEX_HD int assign_indices(const RG& rg, uint32_t& epA, uint32_t& epB, uint32_t anchorPos, uint64_t& idx1, uint64_t& idx2)
{
    typedef PaletteBits<MODE, IM> PB;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    int total = rg.p2sum;
    idx1 = 0; idx2 = 0;

    uint32_t pal[PB::NC], nq2[PB::NC];
#pragma unroll
    for (int i = 0; i < PB::NC; ++i)
    {
        pal[i] = lerp_bytes(ua, ub, weight(PB::CB, i));
        if (PB::AB != 0) pal[i] &= 0x00FFFFFFu;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    int pa[PB::NA];
    if (PB::AB != 0)
    {
        const int a0 = int(ua >> 24), a1 = int(ub >> 24);
#pragma unroll
        for (int i = 0; i < PB::NA; ++i)
            pa[i] = int(lerp1(lerp_base(uint32_t(a0)), a1 - a0, weight(PB::AB, i)));
    }

    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k);
        const uint32_t pc = (PB::AB != 0) ? (p & 0x00FFFFFFu) : p;
        int sc[PB::NC];
#pragma unroll
        for (int i = 0; i < PB::NC; ++i) sc[i] = score(pc, pal[i], nq2[i]);
        uint32_t i1;
        total -= first_peak_idx(sc, i1);
        idx1 |= uint64_t(i1) << (4 * rg.pos(k));
        if (PB::AB != 0)
        {
            const int al = int(p >> 24);
            int su[PB::NA];
#pragma unroll
            for (int i = 0; i < PB::NA; ++i) su[i] = 2 * al * pa[i] - pa[i] * pa[i];
            uint32_t i2;
            total -= first_peak_idx(su, i2);
            idx2 |= uint64_t(i2) << (4 * rg.pos(k));
        }
    });

    // texel positions of this region, 0xF per member
    uint64_t member = 0;
    for_texels(rg, [&](int k) { member |= uint64_t(0xF) << (4 * rg.pos(k)); });

    if ((idx1 >> (4 * anchorPos)) & uint64_t(PB::NC >> 1))
    {
        if (PB::AB == 0) { const uint32_t t = epA; epA = epB; epB = t; }
        else
        {
            const uint32_t a = epA, b = epB;
            epA = (b & 0x00FFFFFFu) | (a & 0xFF000000u);
            epB = (a & 0x00FFFFFFu) | (b & 0xFF000000u);
        }
        // idx = (N - 1) - idx for every member texel == xor with N-1
        idx1 ^= member & (uint64_t(PB::NC - 1) * 0x1111111111111111ull);
    }
    if (PB::AB != 0)
    {
        if (idx2 & uint64_t(PB::NA >> 1))      // aIndices2[0]: separate-alpha modes have one region, anchor texel 0
        {
            const uint32_t a = epA, b = epB;
            epA = (a & 0x00FFFFFFu) | (b & 0xFF000000u);
            epB = (b & 0x00FFFFFFu) | (a & 0xFF000000u);
            idx2 ^= uint64_t(PB::NA - 1) * 0x1111111111111111ull;
        }
    }
    return total;
}

// AssignIndices for a caller that needs the region's ERROR and the anchor fix-up of the endpoints but not the indices (pre: the unoptimised
// half of Refine, whose indices are only ever wanted for a block's winner, and post derives those itself): the texel loop takes the
// first-peak VALUE (first_peak: independent compares) instead of tracking the index (first_peak_idx: a serial scan plus the 64-bit index
// word) - the same sum - and only the anchor texel is scanned for its index. Same error, same endpoints as assign_indices.
template<int MODE, int IM, class RG>
DXTEX_HD int assign_error(const RG& rg, uint32_t& epA, uint32_t& epB, uint32_t anchorPos)
{
    typedef PaletteBits<MODE, IM> PB;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    int total = rg.p2sum;
    uint32_t pal[PB::NC], nq2[PB::NC];
#pragma unroll
    for (int i = 0; i < PB::NC; ++i)
    {
        pal[i] = lerp_bytes(ua, ub, weight(PB::CB, i));
        if (PB::AB != 0) pal[i] &= 0x00FFFFFFu;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    int pa[PB::NA];
    if (PB::AB != 0)
    {
        const int a0 = int(ua >> 24), a1 = int(ub >> 24);
#pragma unroll
        for (int i = 0; i < PB::NA; ++i)
            pa[i] = int(lerp1(lerp_base(uint32_t(a0)), a1 - a0, weight(PB::AB, i)));
    }
    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k);
        const uint32_t pc = (PB::AB != 0) ? (p & 0x00FFFFFFu) : p;
        int sc[PB::NC];
#pragma unroll
        for (int i = 0; i < PB::NC; ++i) sc[i] = score(pc, pal[i], nq2[i]);
        total -= first_peak(sc);
        if (PB::AB != 0)
        {
            const int al = int(p >> 24);
            int su[PB::NA];
#pragma unroll
            for (int i = 0; i < PB::NA; ++i) su[i] = 2 * al * pa[i] - pa[i] * pa[i];
            total -= first_peak(su);
        }
    });
    // the anchor texel's index decides the swap (:3195-3215)
    {
        const uint32_t p = rg.fetch_pos(anchorPos);
        const uint32_t pc = (PB::AB != 0) ? (p & 0x00FFFFFFu) : p;
        int sc[PB::NC];
#pragma unroll
        for (int i = 0; i < PB::NC; ++i) sc[i] = score(pc, pal[i], nq2[i]);
        uint32_t i1;
        (void)first_peak_idx(sc, i1);
        if (i1 & uint32_t(PB::NC >> 1))
        {
            if (PB::AB == 0) { const uint32_t t = epA; epA = epB; epB = t; }
            else
            {
                const uint32_t a = epA, b = epB;
                epA = (b & 0x00FFFFFFu) | (a & 0xFF000000u);
                epB = (a & 0x00FFFFFFu) | (b & 0xFF000000u);
            }
        }
    }
    if (PB::AB != 0)
    {
        const int al = int(rg.fetch_pos(0u) >> 24);      // aIndices2[0]: separate-alpha modes have one region, anchor texel 0
        int su[PB::NA];
#pragma unroll
        for (int i = 0; i < PB::NA; ++i) su[i] = 2 * al * pa[i] - pa[i] * pa[i];
        uint32_t i2;
        (void)first_peak_idx(su, i2);
        if (i2 & uint32_t(PB::NA >> 1))
        {
            const uint32_t a = epA, b = epB;
            epA = (a & 0x00FFFFFFu) | (b & 0xFF000000u);
            epB = (b & 0x00FFFFFFu) | (a & 0xFF000000u);
        }
    }
    return total;
}

// FixEndpointPBits for the endpoint pair of one subset (:3311-3396). Inputs are RGBAPrecWithP-bit values.
template<int MODE>
DXTEX_HD void fix_pbits(uint32_t inA, uint32_t inB, uint32_t& outA, uint32_t& outB)
{
    typedef ModeInfo<MODE> MI;
    if (MI::PB == 0) { outA = inA; outB = inB; return; }
    constexpr int NCH = (MI::AP != MI::APP) ? 4 : 3;      // channels that carry a p-bit
    uint32_t voteA = 0, voteB = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { voteA += byte_of(inA, ch) & 1u; voteB += byte_of(inB, ch) & 1u; }
    uint32_t pA, pB;
    if (MI::PB == 2)
    {
        const uint32_t p = (voteA + voteB) > uint32_t((2 * NCH) >> 1) ? 1u : 0u;
        pA = p; pB = p;
    }
    else
    {
        pA = voteA > uint32_t(NCH >> 1) ? 1u : 0u;
        pB = voteB > uint32_t(NCH >> 1) ? 1u : 0u;
    }
    // every byte: ((v >> 1) << 1) | p for p-bit channels; the unencoded alpha byte (255) becomes
    // uint8((255 << 1) | p), exactly as the reference's second loop does to all four channels.
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
    {
        const uint32_t va = byte_of(inA, ch), vb = byte_of(inB, ch);
        const uint32_t ha = (ch < NCH) ? (va >> 1) : va, hb = (ch < NCH) ? (vb >> 1) : vb;
        a |= (((ha << 1) | pA) & 0xFFu) << (8 * ch);
        b |= (((hb << 1) | pB) & 0xFFu) << (8 * ch);
    }
    outA = a; outB = b;
}

// Quantize the 8-bit seed endpoints to RGBAPrecWithP (:3428-3429) with D3DX_BC7::Quantize (:806-811):
//   rnd = min<uint8_t>(255, uint8_t(comp + (1 << (7 - prec))));  q = rnd >> (8 - prec)
// The cast to uint8_t happens before the min, so the rounding offset WRAPS for bright components (e.g.
// prec 5: 250 + 4 -> 254 is fine, 253 + 4 -> 1); the min is a no-op. Reproduced as is. For prec == 8 the
// offset is 1u << -1, which x86 evaluates as 1u << 31: the low byte is unchanged.
template<int PREC>
DXTEX_HD uint32_t quantize1(uint32_t v)
{
    if (PREC == 0) return 255u;
    if (PREC == 8) return v;
    return ((v + (1u << (7 - (PREC & 7)))) & 0xFFu) >> (8 - PREC);
}

template<int MODE>
DXTEX_HD uint32_t quantize_endpoint(uint32_t c)
{
    typedef ModeInfo<MODE> MI;
    return quantize1<MI::CPP>(c & 0xFF) | (quantize1<MI::CPP>((c >> 8) & 0xFF) << 8) |
           (quantize1<MI::CPP>((c >> 16) & 0xFF) << 16) | (quantize1<MI::APP>(c >> 24) << 24);
}

// ---- endpoint search ---------------------------------------------------------------------------------
template<int MODE, int IM, int CH, class RG>
DXTEX_HD int perturb_one(const RG& rg, uint32_t oldA, uint32_t oldB, int oldErr, int do_b, uint32_t& newVal)
{
    typedef ModeInfo<MODE> MI;
    constexpr int prec = (CH == 3) ? MI::APP : MI::CPP;
    int minErr = oldErr;
    int cur = int(byte_of(do_b ? oldB : oldA, CH));
    for (int step = 1 << (prec - 1); step; step >>= 1)
    {
        bool improved = false;
        int beststep = 0;
#pragma unroll
        for (int sign = -1; sign <= 1; sign += 2)
        {
            const int tmp = cur + sign * step;
            if (tmp < 0 || tmp >= (1 << prec)) continue;
            const uint32_t a = do_b ? oldA : with_byte(oldA, CH, uint32_t(tmp));
            const uint32_t b = do_b ? with_byte(oldB, CH, uint32_t(tmp)) : oldB;
            const int e = map_colors<MODE, IM>(rg, a, b);
            if (e < minErr) { improved = true; minErr = e; beststep = sign * step; }
        }
        if (improved) cur += beststep;
    }
    newVal = uint32_t(cur);
    return minErr;
}

template<int MODE, int IM, int CH, class RG>
DXTEX_HD void exhaustive(const RG& rg, int& orgErr, uint32_t& optA, uint32_t& optB)
{
    typedef ModeInfo<MODE> MI;
    constexpr int prec = (CH == 3) ? MI::APP : MI::CPP;
    if (orgErr == 0) return;
    constexpr int delta = 5;
    const int ca = int(byte_of(optA, CH)), cb = int(byte_of(optB, CH));
    const int hi = (1 << prec) - 1;                        // prec == 0 -> 0: the loops below do not run
    const int alow = (ca - delta) > 0 ? (ca - delta) : 0;
    const int ahigh = (ca + delta) < hi ? (ca + delta) : hi;
    const int blow = (cb - delta) > 0 ? (cb - delta) : 0;
    const int bhigh = (cb + delta) < hi ? (cb + delta) : hi;
    int amin = 0, bmin = 0;
    int best = orgErr;
    if (ca <= cb)
    {
        for (int a = alow; a <= ahigh; ++a)
            for (int b = (a > blow ? a : blow); b < bhigh; ++b)
            {
                const int e = map_colors<MODE, IM>(rg, with_byte(optA, CH, uint32_t(a)), with_byte(optB, CH, uint32_t(b)));
                if (e < best) { amin = a; bmin = b; best = e; }
            }
    }
    else
    {
        for (int b = blow; b < bhigh; ++b)
            for (int a = (b > alow ? b : alow); a <= ahigh; ++a)
            {
                const int e = map_colors<MODE, IM>(rg, with_byte(optA, CH, uint32_t(a)), with_byte(optB, CH, uint32_t(b)));
                if (e < best) { amin = a; bmin = b; best = e; }
            }
    }
    if (best < orgErr)
    {
        optA = with_byte(optA, CH, uint32_t(amin));
        optB = with_byte(optB, CH, uint32_t(bmin));
        orgErr = best;
    }
}

// One channel of OptimizeOne's coordinate descent (:3060-3105), quirks included: the B endpoint is
// never moved by this phase (cnew_b aliases new_a.B[ch], which still holds the old value), only its
// claimed error is adopted, and the alternating loop re-applies the first A perturbation.
template<int MODE, int IM, int CH, class RG>
DXTEX_HD void optimize_channel(const RG& rg, uint32_t& optA, uint32_t& optB, int& optErr)
{
    typedef ModeInfo<MODE> MI;
    constexpr int prec = (CH == 3) ? MI::APP : MI::CPP;
    if (prec == 0) return;
    uint32_t newA_val, newB_val, dummy;
    const int err0 = perturb_one<MODE, IM, CH>(rg, optA, optB, optErr, 0, newA_val);
    const int err1 = perturb_one<MODE, IM, CH>(rg, optA, optB, optErr, 1, newB_val);
    (void)newB_val;
    int do_b;
    if (err0 < err1)
    {
        if (err0 >= optErr) return;
        optA = with_byte(optA, CH, newA_val);
        optErr = err0;
        do_b = 1;
    }
    else
    {
        if (err1 >= optErr) return;
        // copt_b = cnew_b == new_a.B[ch] == the current B value: no change
        optErr = err1;
        do_b = 0;
    }
    for (;;)
    {
        const int e = perturb_one<MODE, IM, CH>(rg, optA, optB, optErr, do_b, dummy);
        if (e >= optErr) break;
        if (do_b == 0) optA = with_byte(optA, CH, newA_val);     // copt_a = cnew_a (new_a.A[ch] from the first perturbation)
        // do_b == 1: copt_b = cnew_b (unchanged value)
        optErr = e;
        do_b = 1 - do_b;
    }
}

template<int MODE, int IM, class RG>
DXTEX_HD void optimize_one(const RG& rg, int orgErr, uint32_t orgA, uint32_t orgB, uint32_t& optA, uint32_t& optB)
{
    int optErr = orgErr;
    optA = orgA; optB = orgB;
    optimize_channel<MODE, IM, 0>(rg, optA, optB, optErr);
    optimize_channel<MODE, IM, 1>(rg, optA, optB, optErr);
    optimize_channel<MODE, IM, 2>(rg, optA, optB, optErr);
    optimize_channel<MODE, IM, 3>(rg, optA, optB, optErr);
    exhaustive<MODE, IM, 0>(rg, optErr, optA, optB);
    exhaustive<MODE, IM, 1>(rg, optErr, optA, optB);
    exhaustive<MODE, IM, 2>(rg, optErr, optA, optB);
    exhaustive<MODE, IM, 3>(rg, optErr, optA, optB);
}

// ---- OptimizeOne restated for lockstep execution ------------------------------------------------------------
// The same search as optimize_one() above, cut into pieces that every lane of a wavefront can execute in
// lockstep although each lane works on a different (block, shape, subset):
//   * a PERTURB macro-op = one PerturbOne call (:2926-2966): exactly 2 * PREC candidate evaluations, the same
//     count for every lane, so the macro-op is straight-line code; what differs between lanes (channel, which
//     endpoint moves, where in OptimizeOne's per-channel logic the lane is) is data, updated by a few selects
//     between macro-ops (perturb_transition);
//   * the Exhaustive windows (:2971-3042) flattened into one candidate per loop trip (exh_* functions).
// While one channel of the endpoints varies, the palette bytes of the other channels and their contribution
// to |q|^2 do not: they are computed once per macro-op / window (VarPal) and each candidate only re-derives
// the varying byte. In the separate-alpha modes (4, 5) the colour and alpha errors are independent sums, so a
// loop that walks colour channels carries the alpha error as a constant and vice versa (CH_COLOR / CH_ALPHA);
// the combined-index modes walk all their channels in one loop (CH_ALL).
enum : int { CH_ALL = 0, CH_COLOR = 1, CH_ALPHA = 2 };

template<int MODE, int IM, int CHSET>
struct LoopCfg
{
    typedef ModeInfo<MODE> MI;
    typedef PaletteBits<MODE, IM> PB;
    enum : int { kAlpha = (CHSET == CH_ALPHA) ? 1 : 0,
                 N = kAlpha ? PB::NA : PB::NC,                        // palette entries scored per texel
                 BITS = kAlpha ? PB::AB : PB::CB,
                 PREC = kAlpha ? MI::APP : MI::CPP,                   // endpoint precision of the channels this loop walks
                 CH0 = kAlpha ? 3 : 0,
                 CH1 = (CHSET == CH_ALL) ? (MI::APP ? 4 : 3) : (CHSET == CH_COLOR ? 3 : 4) };   // one past the last channel
    static_assert(CHSET != CH_ALL || MI::APP == 0 || MI::APP == MI::CPP, "combined loops need one precision");
    static_assert((CHSET == CH_ALL) == (PB::AB == 0), "CH_ALL <=> combined colour+alpha indices");
};

template<int N> struct VarPal { uint32_t palO[N]; uint32_t nq2O[N]; };

// Palette of (epA, epB) with channel `ch` blanked, and minus the squared length of what is left.
template<int MODE, int IM, int CHSET>
DXTEX_HD void varpal_init(VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, uint32_t epA, uint32_t epB, int ch)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    if (C::kAlpha) return;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    const uint32_t keep = ~(0xFFu << (8 * ch)) & (CHSET == CH_COLOR ? 0x00FFFFFFu : 0xFFFFFFFFu);
#pragma unroll
    for (int i = 0; i < C::N; ++i)
    {
        vp.palO[i] = lerp_bytes(ua, ub, weight(C::BITS, i)) & keep;
        vp.nq2O[i] = 0u - udot4(vp.palO[i], vp.palO[i]);
    }
}

// Error of the region when channel `ch` of the endpoints unquantises to (uaC, ubC) and everything else is as
// in `vp`. `base` = sum of |p|^2 over the part of the texel this loop scores + the constant error of the other part.
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int eval_var(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, int ch, uint32_t uaC, uint32_t ubC, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    int total = base;
#if defined(DXTEX_COUNT_EVALS)
    ++g_evalCount[MODE]; g_evalTexels[MODE] += rg.count();
#endif
    if (C::kAlpha)
    {
        int pa[C::N], npa2[C::N];
        const int lb64 = lerp_base(uaC), ld = lerp_delta(uaC, ubC);
#pragma unroll
        for (int i = 0; i < C::N; ++i)
        {
            pa[i] = int(lerp1(lb64, ld, weight(C::BITS, i)));
            npa2[i] = -mul24(pa[i], pa[i]);
        }
        for_texels(rg, [&](int k)
        {
            const int al2 = int((rg.fetch(k) >> 23) & 0x1FEu);
            int su[C::N];
#pragma unroll
            for (int i = 0; i < C::N; ++i) su[i] = mul24(al2, pa[i]) + npa2[i];
            total -= first_peak(su);
        });
    }
    else
    {
        uint32_t pal[C::N], nq2[C::N];
        const int sh = 8 * ch;
        const int lb64 = lerp_base(uaC), ld = lerp_delta(uaC, ubC);
#pragma unroll
        for (int i = 0; i < C::N; ++i)
        {
            const uint32_t v = lerp1(lb64, ld, weight(C::BITS, i));
            pal[i] = vp.palO[i] | (v << sh);
            nq2[i] = vp.nq2O[i] - umul24(v, v);
        }
        for_texels(rg, [&](int k)
        {
            uint32_t p = rg.fetch(k);
            if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
            int sc[C::N];
#pragma unroll
            for (int i = 0; i < C::N; ++i) sc[i] = score(p, pal[i], nq2[i]);
            total -= first_peak(sc);
        });
    }
    return total;
}

// A LOWER BOUND on eval_var's result at well under half its cost. ComputeError's "first local minimum" can only be worse
// than the nearest palette entry, so the error with the best entry per texel bounds the candidate from below; and with the
// accumulator of ONE v_dot4_u32_u8 set to floor(-|q_i|^2 / 2) the dot product is t_i = p.q_i - ceil(|q_i|^2 / 2), for which
// 2 p.q_i - |q_i|^2 <= 2 t_i + 1. So   error >= base - sum_t (2 max_i t_i + 1):   N dot4 + N/2 v_max3 per texel instead of
// 2N dot4 + (N - 1) compare/select pairs (v_dot4, v_cmp, v_cndmask and v_max3 all issue at 4 cycles per wave64 on gfx950,
// profiles/r02_valu_rates.md). The alpha loops of modes 4 / 5 need no halving: their bound is the exact nearest-entry error.
// A candidate whose bound is not below the best error so far can never satisfy Exhaustive's `fErr < fBestErr` (:3006) and
// needs no exact evaluation; the others are evaluated exactly, in loop order, a little later (ExhPending).
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int eval_var_bound(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, int ch, uint32_t uaC, uint32_t ubC, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
#if defined(DXTEX_COUNT_EVALS)
    ++g_boundCount[MODE];
#endif
    if (C::kAlpha)
    {
        int pa[C::N], npa2[C::N];
        const int lb64 = lerp_base(uaC), ld = lerp_delta(uaC, ubC);
#pragma unroll
        for (int i = 0; i < C::N; ++i)
        {
            pa[i] = int(lerp1(lb64, ld, weight(C::BITS, i)));
            npa2[i] = -mul24(pa[i], pa[i]);
        }
        int total = base;
        for_texels(rg, [&](int k)
        {
            const int al2 = int((rg.fetch(k) >> 23) & 0x1FEu);
            int m = mul24(al2, pa[0]) + npa2[0];
#pragma unroll
            for (int i = 1; i < C::N; ++i) { const int su = mul24(al2, pa[i]) + npa2[i]; m = su > m ? su : m; }
            total -= m;
        });
        return total;
    }
    else
    {
        uint32_t pal[C::N], acc[C::N];
        const int sh = 8 * ch;
        const int lb64 = lerp_base(uaC), ld = lerp_delta(uaC, ubC);
#pragma unroll
        for (int i = 0; i < C::N; ++i)
        {
            const uint32_t v = lerp1(lb64, ld, weight(C::BITS, i));
            pal[i] = vp.palO[i] | (v << sh);
            acc[i] = uint32_t(int(vp.nq2O[i] - umul24(v, v)) >> 1);        // floor(-|q|^2 / 2)
        }
        int sum = 0;
        for_texels(rg, [&](int k)
        {
            uint32_t p = rg.fetch(k);
            if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
            int m = int(udot4acc(p, pal[0], acc[0]));
#pragma unroll
            for (int i = 1; i < C::N; ++i) { const int t = int(udot4acc(p, pal[i], acc[i])); m = t > m ? t : m; }
            sum += m;
        });
        return base - 2 * sum - rg.count();
    }
}

// The two independent error sums of a separate-alpha mode, without the |p|^2 terms folded in.
template<int MODE, int IM, class RG>
DXTEX_HD int color_part_error(const RG& rg, uint32_t epA, uint32_t epB)
{
    typedef PaletteBits<MODE, IM> PB;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    uint32_t pal[PB::NC], nq2[PB::NC];
#pragma unroll
    for (int i = 0; i < PB::NC; ++i)
    {
        pal[i] = lerp_bytes(ua, ub, weight(PB::CB, i)) & 0x00FFFFFFu;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    int total = 0;
    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k) & 0x00FFFFFFu;
        int sc[PB::NC];
#pragma unroll
        for (int i = 0; i < PB::NC; ++i) sc[i] = score(p, pal[i], nq2[i]);
        total += int(udot4(p, p)) - first_peak(sc);
    });
    return total;
}

template<int MODE, int IM, class RG>
DXTEX_HD int alpha_part_error(const RG& rg, uint32_t epA, uint32_t epB)
{
    typedef PaletteBits<MODE, IM> PB;
    const uint32_t ua = unquantize<MODE>(epA), ub = unquantize<MODE>(epB);
    constexpr int NA = PB::NA, AB = PB::AB ? PB::AB : 2;
    int pa[NA];
    const int a0 = int(ua >> 24), a1 = int(ub >> 24);
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = int(lerp1(lerp_base(uint32_t(a0)), a1 - a0, weight(AB, i)));
    int total = 0;
    for_texels(rg, [&](int k)
    {
        const int al = int(rg.fetch(k) >> 24);
        int su[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) su[i] = 2 * al * pa[i] - pa[i] * pa[i];
        total += al * al - first_peak(su);
    });
    return total;
}

// `base` of eval_var for a loop over CHSET, given the endpoints whose other part stays fixed during the loop.
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int loop_base(const RG& rg, uint32_t epA, uint32_t epB, int* otherPart = nullptr)
{
    int own = 0;
    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k);
        const uint32_t q = (CHSET == CH_COLOR) ? (p & 0x00FFFFFFu) : (CHSET == CH_ALPHA) ? (p >> 24) : p;
        own += int(udot4(q, q));
    });
    int other = 0;
    if (CHSET == CH_COLOR) other = alpha_part_error<MODE, IM>(rg, epA, epB);
    if (CHSET == CH_ALPHA) other = color_part_error<MODE, IM>(rg, epA, epB);
    if (otherPart) *otherPart = other;
    return own + other;
}

// A loop over the scalar slot of modes 4 / 5 cannot change anything when the slot's error is already 0: every candidate's error is
// (error of the colour part, constant in this loop) + (scalar error >= 0), and PerturbOne / Exhaustive only accept strictly smaller
// errors (:2954, :3006). True for the alpha slot of every opaque block. `err` is the task's current total error.
template<int CHSET>
DXTEX_HD bool loop_is_settled(int err, int otherPart) { return CHSET == CH_ALPHA && err == otherPart; }

// Where a lane stands inside OptimizeOne's per-channel logic (:3060-3105).
struct PerturbState
{
    uint32_t optA, optB;
    int optErr;
    int ch;         // current channel; >= CH1 when the loop's channels are exhausted
    int sub;        // 0 = first pass on A, 1 = first pass on B, 2 = alternating loop
    int do_b;
    int err0;       // result of the first pass on A
    uint32_t newA;  // new_a.A[ch] of the first pass on A
};

template<int MODE, int IM, int CHSET>
DXTEX_HD PerturbState perturb_begin(uint32_t optA, uint32_t optB, int optErr)
{
    PerturbState s;
    s.optA = optA; s.optB = optB; s.optErr = optErr;
    s.ch = LoopCfg<MODE, IM, CHSET>::CH0; s.sub = 0; s.do_b = 0; s.err0 = 0; s.newA = 0;
    return s;
}

// One PerturbOne call (:2926-2966) on channel s.ch of endpoint A (s.do_b == 0) or B: returns the best error
// found (fMinErr) and the channel value it belongs to. Straight-line: 2 * PREC - 1 evaluations for every lane.
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD void perturb_macro(const RG& rg, const PerturbState& s, int base, int& outErr, uint32_t& outVal)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    VarPal<C::N> vp;
#if defined(DXTEX_COUNT_EVALS)
    ++g_macroCount[MODE];
#endif
    varpal_init<MODE, IM, CHSET>(vp, s.optA, s.optB, s.ch);
    const uint32_t fixedU = unq1<C::PREC>(byte_of(s.do_b ? s.optA : s.optB, s.ch));
    int cur = int(byte_of(s.do_b ? s.optB : s.optA, s.ch));
    int minErr = s.optErr;
    // The first step is half the range, so exactly one of cur - step / cur + step is a legal endpoint (cur >= step or cur < step)
    // and the other is skipped by the reference (:2937): one evaluation instead of two.
    {
        constexpr int half = 1 << (C::PREC - 1);
        const int tmp = (cur >= half) ? cur - half : cur + half;
        const uint32_t u = unq1<C::PREC>(uint32_t(tmp) & ((1u << C::PREC) - 1u));
        const int e = eval_var<MODE, IM, CHSET>(rg, vp, s.ch, s.do_b ? fixedU : u, s.do_b ? u : fixedU, base);
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
            const uint32_t u = unq1<C::PREC>(uint32_t(tmp) & ((1u << C::PREC) - 1u));
            const int e = eval_var<MODE, IM, CHSET>(rg, vp, s.ch, s.do_b ? fixedU : u, s.do_b ? u : fixedU, base);
#if defined(DXTEX_COUNT_EVALS) && defined(DXTEX_COUNT_PERTURB_FILTER)
            { const int lb = eval_var_bound<MODE, IM, CHSET>(rg, vp, s.ch, s.do_b ? fixedU : u, s.do_b ? u : fixedU, base); --g_boundCount[MODE];
              ++g_pfTotal[MODE]; if (valid && lb < minErr) ++g_pfPass[MODE]; if (valid && e < minErr) ++g_pfImprove[MODE];
              { int si = 0; for (int t = step; t > 1; t >>= 1) ++si; ++g_pfStepTotal[MODE][si]; if (valid && lb < minErr) ++g_pfStepPass[MODE][si]; } }
#endif
#if defined(DXTEX_COUNT_EVALS)
            ++g_pfStepTotal[MODE][7]; if (!valid) ++g_pfStepPass[MODE][7];
#endif
            if (valid && e < minErr) { minErr = e; beststep = sign * step; }
        }
        cur += beststep;
    }
    outErr = minErr; outVal = uint32_t(cur);
}

// OptimizeOne's bookkeeping between PerturbOne calls, quirks included (see optimize_channel above).
// Returns the next state; `s.ch >= CH1` afterwards means the loop's channels are done.
template<int MODE, int IM, int CHSET>
DXTEX_HD PerturbState perturb_transition(const PerturbState& in, int e, uint32_t val)
{
    PerturbState s = in;
    const bool sub0 = (in.sub == 0), sub1 = (in.sub == 1);
    const bool takeA = sub1 ? (in.err0 < e) : (in.do_b == 0);
    const int claimed = (sub1 && takeA) ? in.err0 : e;
    const bool giveUp = !sub0 && (claimed >= in.optErr);
    // sub 0: remember the A result, perturb B next. Otherwise adopt the claimed error; only endpoint A ever moves
    s.err0 = sub0 ? e : in.err0;
    s.newA = sub0 ? val : in.newA;
    s.optA = (!sub0 && !giveUp && takeA) ? with_byte(in.optA, in.ch, in.newA) : in.optA;
    s.optErr = (sub0 || giveUp) ? in.optErr : claimed;
    s.sub = giveUp ? 0 : (sub0 ? 1 : 2);
    s.do_b = giveUp ? 0 : (sub0 ? 1 : (sub1 ? (takeA ? 1 : 0) : (1 - in.do_b)));
    s.ch = giveUp ? in.ch + 1 : in.ch;
    return s;
}

// ---- PerturbOne calls that cannot change anything ---------------------------------------------------------------------------------
// A channel that is CONSTANT over the region's texels (value v) and whose two endpoints both unquantise to exactly v contributes nothing to any
// palette entry's error (every entry's value on it is ((64 - w) v + w v + 32) >> 6 = v). A PerturbOne call on that channel moves one endpoint
// away from v, which adds a non-negative term to EVERY entry's error for every texel and leaves the other channels alone: whatever entry
// ComputeError's first-local-minimum scan then stops at, the candidate's error is at least the sum over the texels of the smallest entry
// error of the CURRENT palette (G, eval_nearest). The scan itself can do better with the extra terms than without (they can push it past an
// early local minimum), so G, not the current error, is what bounds the candidates - but where G equals the current error (the scan already
// finds every texel's nearest entry: the usual case) no candidate can be strictly below it and the call returns (optErr, current value):
// its 2 PREC - 1 evaluations are skipped and the state moves on as it would have. On an opaque image that is every call on the alpha channel in
// mode 6 and on the swapped-in constant of rotations 1 - 3 in modes 4 / 5: a quarter of their PerturbOne calls.
template<class RG>
DXTEX_HD uint32_t flat_channels(const RG& rg, uint32_t& vals)
{
    const uint32_t first = rg.fetch(0);
    uint32_t diff = 0;
    for_texels(rg, [&](int k) { diff |= rg.fetch(k) ^ first; });
    vals = first;
    return ((diff & 0xFFu) ? 0u : 1u) | ((diff & 0xFF00u) ? 0u : 2u) | ((diff & 0xFF0000u) ? 0u : 4u) | ((diff & 0xFF000000u) ? 0u : 8u);
}

template<int MODE, int IM, int CHSET>
DXTEX_HD bool flat_call(const PerturbState& s, uint32_t flatMask, uint32_t flatVals)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    if (C::kAlpha || s.ch >= C::CH1) return false;
    const uint32_t a = byte_of(s.optA, s.ch), b = byte_of(s.optB, s.ch);
    return ((flatMask >> s.ch) & 1u) != 0u && a == b && unq1<C::PREC>(a) == byte_of(flatVals, s.ch);
}

// G: the error of the state's endpoints with every texel on its NEAREST palette entry (<= the first-local-minimum error the state carries)
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int eval_nearest(const RG& rg, const PerturbState& s, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    if (C::kAlpha) return 0x7FFFFFFF;
    const uint32_t ua = unquantize<MODE>(s.optA), ub = unquantize<MODE>(s.optB);
    const uint32_t keep = (CHSET == CH_COLOR) ? 0x00FFFFFFu : 0xFFFFFFFFu;
    uint32_t pal[C::N], nq2[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i)
    {
        pal[i] = lerp_bytes(ua, ub, weight(C::BITS, i)) & keep;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    int total = base;
    for_texels(rg, [&](int k)
    {
        uint32_t p = rg.fetch(k);
        if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
        int m = score(p, pal[0], nq2[0]);
#pragma unroll
        for (int i = 1; i < C::N; ++i) { const int t = score(p, pal[i], nq2[i]); m = t > m ? t : m; }
        total -= m;
    });
    return total;
}

// The state after the calls on the current channel that flat_call / eval_nearest have shown to find nothing (one call, or the A and the B
// call of a channel just entered): perturb_transition with the "nothing found" result (optErr, the endpoint's current value).
template<int MODE, int IM, int CHSET>
DXTEX_HD PerturbState skip_flat_calls(const PerturbState& s)
{
    const int ch = s.ch;
    PerturbState t = perturb_transition<MODE, IM, CHSET>(s, s.optErr, byte_of(s.do_b ? s.optB : s.optA, ch));
    if (t.ch == ch) t = perturb_transition<MODE, IM, CHSET>(t, t.optErr, byte_of(t.do_b ? t.optB : t.optA, ch));
    return t;
}

// Exhaustive (:2971-3042), one candidate per step. (o, i) are the outer / inner loop variables: (a, b) when
// the channel starts with a <= b, else (b, a). Note the reference's asymmetric bounds: a <= ahigh, b < bhigh.
struct ExhState
{
    uint32_t optA, optB;
    int optErr;
    int ch;                 // >= CH1: finished
    int o, i, oEnd, iEnd, lo;   // lo: lowest value of the inner variable in the window as the reference opens it = origin of the candidate codes
    int iA;                 // lowest value of the inner variable still to be visited (>= lo: exh_peel may have taken columns off)
    int o0;                 // first value of the outer variable in the window as opened = origin of the candidate codes (s.o may start above it after exh_peel)
    int aleb;
    int omin, imin, best;
    int bestCode;           // queue code of (omin, imin) in the current window, -1 while the window has not improved on optErr
};

template<int MODE, int IM, int CHSET>
DXTEX_HD ExhState exh_window(const ExhState& in, int ch)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    ExhState s = in;
    s.ch = (in.optErr == 0) ? int(C::CH1) : ch;          // Exhaustive returns at once when the error is already zero (:2980)
    const int c = (s.ch >= C::CH1) ? int(C::CH0) : s.ch;
    constexpr int delta = 5, hi = (1 << C::PREC) - 1;
    const int ca = int(byte_of(in.optA, c)), cb = int(byte_of(in.optB, c));
    const int alow = (ca - delta) > 0 ? (ca - delta) : 0, ahigh = (ca + delta) < hi ? (ca + delta) : hi;
    const int blow = (cb - delta) > 0 ? (cb - delta) : 0, bhigh = (cb + delta) < hi ? (cb + delta) : hi;
    s.aleb = ca <= cb;
    s.o = s.aleb ? alow : blow;
    s.o0 = s.o;
    s.oEnd = s.aleb ? ahigh + 1 : bhigh;
    s.lo = s.aleb ? blow : alow;
    s.iA = s.lo;
    s.iEnd = s.aleb ? bhigh : ahigh + 1;
    s.i = s.o > s.lo ? s.o : s.lo;
    s.omin = 0; s.imin = 0; s.best = in.optErr; s.bestCode = -1;
    return s;
}

// Skip empty inner ranges; returns false when the window is used up.
DXTEX_HD bool exh_settle(ExhState& s)
{
    while (s.o < s.oEnd && s.i >= s.iEnd) { ++s.o; s.i = s.o > s.iA ? s.o : s.iA; }
    return s.o < s.oEnd;
}

// One candidate forward, branch-free. Rows that hold no candidate (max(o, lo) >= iEnd) only occur at the end of a window - the first
// value of the inner variable never decreases from row to row - so while candidates remain (exh_remaining) one step lands on the
// next one; after the last candidate the state is merely "somewhere past it", which exh_settle / exh_next turn into "window used up".
DXTEX_HD void exh_advance(ExhState& s)
{
    ++s.i;
    const bool wrap = s.i >= s.iEnd;
    s.o += wrap ? 1 : 0;
    const int first = s.o > s.iA ? s.o : s.iA;
    s.i = wrap ? first : s.i;
}

DXTEX_HD ExhState exh_commit(const ExhState& in)
{
    ExhState s = in;
    if (in.best < in.optErr)
    {
        const int a = in.aleb ? in.omin : in.imin, b = in.aleb ? in.imin : in.omin;
        s.optA = with_byte(in.optA, in.ch, uint32_t(a));
        s.optB = with_byte(in.optB, in.ch, uint32_t(b));
        s.optErr = in.best;
    }
    return s;
}

// Advance to the next candidate, opening the following channels' windows as needed. `vp` is rebuilt whenever a
// new window opens. Returns false when every channel of the loop has been searched.
template<int MODE, int IM, int CHSET>
DXTEX_HD bool exh_next(ExhState& s, VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    while (s.ch < C::CH1 && !exh_settle(s))
    {
        s = exh_commit(s);
        s = exh_window<MODE, IM, CHSET>(s, s.ch + 1);
        if (s.ch < C::CH1) varpal_init<MODE, IM, CHSET>(vp, s.optA, s.optB, s.ch);
    }
    return s.ch < C::CH1;
}

template<int MODE, int IM, int CHSET>
DXTEX_HD bool exh_begin(ExhState& s, VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, uint32_t optA, uint32_t optB, int optErr)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    s.optA = optA; s.optB = optB; s.optErr = optErr;
    s.ch = 0; s.o = s.i = s.oEnd = s.iEnd = s.lo = 0; s.iA = 0; s.o0 = 0; s.aleb = 0; s.omin = s.imin = 0; s.best = optErr; s.bestCode = -1;
    s = exh_window<MODE, IM, CHSET>(s, C::CH0);
    if (s.ch < C::CH1) varpal_init<MODE, IM, CHSET>(vp, s.optA, s.optB, s.ch);
    return exh_next<MODE, IM, CHSET>(s, vp);
}

template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD void exh_step(const RG& rg, ExhState& s, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    const int a = s.aleb ? s.o : s.i, b = s.aleb ? s.i : s.o;
    const int e = eval_var<MODE, IM, CHSET>(rg, vp, s.ch, unq1<C::PREC>(uint32_t(a)), unq1<C::PREC>(uint32_t(b)), base);
    if (e < s.best) { s.omin = s.o; s.imin = s.i; s.best = e; }       // strict: the first minimum in loop order wins (:3006)
    ++s.i;
}

// The same step split in two (see eval_var_bound): visiting a candidate only bounds it; the candidates that might still become the
// window's result are set aside and evaluated exactly later - in the search kernel by ALL lanes of the wavefront together, whoever's
// candidates they are (bc7_exhaustive_kernel in bc7_encode.hip).
// Exhaustive's loop keeps the FIRST candidate, in loop order, of those with the smallest error below the starting error (:3006 is a
// strict '<'), so the result of a window is the minimum of (error, loop position) over its candidates - independent of the order in
// which they are evaluated - or "no change" if no error is below the start. Both are packed into one key,
//     key = (error << 8) | (code + 1),  code = ((o - o0) << 4) | (i - lo)   (a window is at most 11 x 11; increasing code = loop order)
// with key 0 in the low byte for "the starting endpoints" (error optErr): the window's result is the MINIMUM KEY, an unsigned
// min that lanes can take with LDS atomics, and a candidate is out of the race ("beaten") as soon as the key of its bound is not below
// the best key known.
DXTEX_HD uint32_t exh_key(int err, int code) { return (uint32_t(err > 0 ? err : 0) << 8) | uint32_t(code + 1); }     // errors are < 2^23
DXTEX_HD uint32_t exh_start_key(const ExhState& s) { return uint32_t(s.optErr) << 8; }
DXTEX_HD int exh_code(const ExhState& s) { return ((s.o - s.o0) << 4) | (s.i - s.lo); }

// The window's minimum key -> the state exh_commit expects.
DXTEX_HD void exh_apply_key(ExhState& s, uint32_t key)
{
    if ((key & 0xFFu) == 0u) return;              // nothing beat the starting endpoints
    const int code = int(key & 0xFFu) - 1;
    s.best = int(key >> 8); s.omin = s.o0 + (code >> 4); s.imin = s.lo + (code & 0xF);
}

// Candidates of the window that are still to be visited, from (s.o, s.i) on.
DXTEX_HD int exh_remaining(const ExhState& s)
{
    int n = 0;
    for (int o = s.o; o < s.oEnd; ++o)
    {
        const int from = (o == s.o) ? s.i : (o > s.iA ? o : s.iA);
        n += (s.iEnd > from) ? (s.iEnd - from) : 0;
    }
    return n;
}

template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int exh_bound(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, int ch, int aleb, int o0, int lo, int code, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    const int o = o0 + (code >> 4), i = lo + (code & 0xF);
    const int a = aleb ? o : i, b = aleb ? i : o;
    return eval_var_bound<MODE, IM, CHSET>(rg, vp, ch, unq1<C::PREC>(uint32_t(a)), unq1<C::PREC>(uint32_t(b)), base);
}

template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int exh_exact(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, int ch, int aleb, int o0, int lo, int code, int base)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    const int o = o0 + (code >> 4), i = lo + (code & 0xF);
    const int a = aleb ? o : i, b = aleb ? i : o;
    return eval_var<MODE, IM, CHSET>(rg, vp, ch, unq1<C::PREC>(uint32_t(a)), unq1<C::PREC>(uint32_t(b)), base);
}

// Can ANY candidate of the window that has just been opened beat the error the window starts from? Every palette entry's value on
// the channel being searched is monotone in both endpoints, so over the window it stays inside [v(lowest a, lowest b), v(highest a,
// highest b)]; with the entry anywhere in that interval a texel's error against it is at least (error on the other channels) +
// (distance of the texel's channel value to the interval)^2, and a candidate's error is at least the sum over the texels of the
// smallest such value (ComputeError's first-local-minimum scan can only do worse than the nearest entry). If that bound is not below
// the starting error, Exhaustive's strict '<' (:3006) accepts nothing in this window: the ~100 candidates need not be visited.
// One evaluation of about four candidate bounds. (Measured on the benchmark image, unpruned: 78 % of mode 4's windows, 55 % of mode
// 5's, 20 % / 14 % / 32 % of the windows of modes 1 / 3 / 6 are of that kind - e.g. every window on a channel that is constant.)
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int exh_range_bound(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, const ExhState& s, int base, int oLo, int oHi, int iLo, int iHi);

template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int exh_window_bound(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, const ExhState& s, int base)
{
    return exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, s.o, s.oEnd - 1, s.i, s.iEnd - 1);      // at window open: s.o == s.o0, s.i == the first row's first value
}

// "nothing in this window can beat the error it starts from": the bound is not below that error
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD bool exh_window_excluded(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, const ExhState& s, int base)
{
    return exh_window_bound<MODE, IM, CHSET>(rg, vp, s, base) >= s.optErr;
}

// The same bound over any rectangle of (outer, inner) endpoint values of the window `s` describes.
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD int exh_range_bound(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, const ExhState& s, int base, int oLo, int oHi, int iLo, int iHi)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    const uint32_t uoL = unq1<C::PREC>(uint32_t(oLo)), uoH = unq1<C::PREC>(uint32_t(oHi)), uiL = unq1<C::PREC>(uint32_t(iLo)), uiH = unq1<C::PREC>(uint32_t(iHi));
    const uint32_t aL = s.aleb ? uoL : uiL, bL = s.aleb ? uiL : uoL, aH = s.aleb ? uoH : uiH, bH = s.aleb ? uiH : uoH;
    int lo[C::N], hi[C::N];
    const int lbL = lerp_base(aL), ldL = lerp_delta(aL, bL), lbH = lerp_base(aH), ldH = lerp_delta(aH, bH);
#pragma unroll
    for (int i = 0; i < C::N; ++i)
    {
        lo[i] = int(lerp1(lbL, ldL, weight(C::BITS, i)));
        hi[i] = int(lerp1(lbH, ldH, weight(C::BITS, i)));
    }
    if (C::kAlpha)
    {
        // the scalar slot of modes 4 / 5: the same statement in one dimension
        int sumA = 0;
        for_texels(rg, [&](int k)
        {
            const int al = int(rg.fetch(k) >> 24);
            int d2 = 0x7FFFFFFF;
#pragma unroll
            for (int i = 0; i < C::N; ++i)
            {
                const int c = al < lo[i] ? lo[i] : (al > hi[i] ? hi[i] : al);
                const int d = al - c;
                const int dd = mul24(d, d);
                d2 = dd < d2 ? dd : d2;
            }
            sumA += mul24(al, al) - d2;
        });
        return base - sumA;
    }
    // vp.palO has the searched channel blanked: h = p.q' + floor(-|q'|^2 / 2) is the other channels' share of the score, halved and rounded
    // down as in eval_var_bound (2 h <= 2 p.q' - |q'|^2 <= 2 h + 1: ONE dot product); the searched channel adds 2 pc v - v^2 = pc^2 - (pc - v)^2
    // <= pc^2 - d^2 with d the distance of pc to the entry's interval (v_med3). So score <= 2 h - d^2 + pc^2 + 1 per entry.
    const int sh = 8 * (s.ch & 3);           // (idle lanes of the kernels call this with s.ch == CH1 and discard the result: no shift by 32)
    uint32_t accO[C::N];
#pragma unroll
    for (int i = 0; i < C::N; ++i) accO[i] = uint32_t(int(vp.nq2O[i]) >> 1);
    int sum = 0;
    for_texels(rg, [&](int k)
    {
        uint32_t p = rg.fetch(k);
        if (CHSET == CH_COLOR) p &= 0x00FFFFFFu;
        const int pc = int((p >> sh) & 0xFFu);
        int m = -0x7FFFFFFF;
#pragma unroll
        for (int i = 0; i < C::N; ++i)
        {
            const int h = int(udot4acc(p, vp.palO[i], accO[i]));
            const int c = med3(pc, lo[i], hi[i]);
            const int t = mad24(pc - c, c - pc, h + h);
            m = t > m ? t : m;
        }
        sum += m + mul24(pc, pc);
    });
    return base - sum - rg.count();
}

// Strips of kPeelRows rows / columns are taken off the four sides of a freshly opened window's rectangle while their interval bound says
// that nothing in them can beat the error the window starts from (the argument of exh_window_excluded, applied to a part of the window:
// Exhaustive's strict '<' (:3006) accepts none of the strip's candidates whatever the rest of the window finds, because the best error
// only goes down). On the benchmark image the four tests remove 30 % / 22 % / 35 % / 26 % of the candidates of the windows of modes
// 1 / 3 / 4 / 5 that survive the whole-window test, for the price of about seven candidate bounds (tools/bc7_debug.cpp, -DDXTEX_ROW_STATS);
// single rows anywhere in the window, or wider strips, pay less. The codes of the remaining candidates keep their origin (s.o0, s.lo).
enum : int { kPeelRows = 2 };
// The strip on `side` of what is left of the window (0 / 1: the lowest / highest rows = outer values, 2 / 3: the lowest / highest columns =
// inner values): its rectangle, and whether there is one to test (the side must keep at least one row / column).
DXTEX_HD bool exh_peel_rect(const ExhState& s, int side, int& ro0, int& ro1, int& ri0, int& ri1)
{
    const int oA = s.o, oB = s.oEnd - 1, iA = s.iA, iB = s.iEnd - 1;      // candidates have inner >= outer
    const bool rows = side < 2, high = (side & 1) != 0;
    const int extent = rows ? (oB - oA + 1) : (iB - iA + 1);
    ro0 = (rows && high) ? oB - kPeelRows + 1 : oA;
    ro1 = (rows && !high) ? oA + kPeelRows - 1 : oB;
    ri0 = (!rows && high) ? iB - kPeelRows + 1 : iA;
    ri1 = (!rows && !high) ? iA + kPeelRows - 1 : iB;
    ro1 = ro1 < ri1 ? ro1 : ri1;             // outer <= inner
    ri0 = ri0 > ro0 ? ri0 : ro0;             // inner >= outer
    return extent > kPeelRows && ro0 <= ro1 && ri0 <= ri1;
}
DXTEX_HD void exh_peel_apply(ExhState& s, int side, bool take = true)
{
    // (selects, not branches on `side`: indexed through a per-lane side the four bounds would be moved to scratch memory)
    const int k = take ? int(kPeelRows) : 0;
    s.o += (side == 0) ? k : 0;
    s.oEnd -= (side == 1) ? k : 0;
    s.iA += (side == 2) ? k : 0;
    s.iEnd -= (side == 3) ? k : 0;
    s.i = s.o > s.iA ? s.o : s.iA;
}
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD void exh_peel(const RG& rg, const VarPal<LoopCfg<MODE, IM, CHSET>::N>& vp, ExhState& s, int base)
{
#pragma unroll 1
    for (int side = 0; side < 4; ++side)
    {
        int ro0, ro1, ri0, ri1;
        if (exh_peel_rect(s, side, ro0, ro1, ri0, ri1) && exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, ro0, ro1, ri0, ri1) >= s.optErr) exh_peel_apply(s, side);
    }
}

// optimize_one() through the lockstep pieces, one lane's worth (host-side equivalence check, and the
// definition of what the search kernels compute).
template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD void lockstep_perturb_loop(const RG& rg, uint32_t& optA, uint32_t& optB, int& optErr)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    if (C::PREC == 0) return;
    int other;
    const int base = loop_base<MODE, IM, CHSET>(rg, optA, optB, &other);
    if (loop_is_settled<CHSET>(optErr, other)) return;
    PerturbState s = perturb_begin<MODE, IM, CHSET>(optA, optB, optErr);
    uint32_t flatVals = 0;
    const uint32_t flatMask = flat_channels(rg, flatVals);
    while (s.ch < C::CH1)
    {
        // (the kernels apply this to mode 6's combined loop only - kFlatSkip, bc7_encode.hip; here every mode takes it by default, so the CPU
        // suite checks the argument against the reference on two- and three-subset regions as well, and once more the way the kernels run)
        bool take = MODE == 6 && CHSET == CH_ALL;
#if defined(DXTEX_HOST_DEBUG)
        take = take || g_hostFlatSkipEveryMode;
#endif
        if (take && flat_call<MODE, IM, CHSET>(s, flatMask, flatVals) && eval_nearest<MODE, IM, CHSET>(rg, s, base) == s.optErr)
        {
            s = skip_flat_calls<MODE, IM, CHSET>(s);
            continue;
        }
        int e; uint32_t v;
        perturb_macro<MODE, IM, CHSET>(rg, s, base, e, v);
        s = perturb_transition<MODE, IM, CHSET>(s, e, v);
    }
    optA = s.optA; optB = s.optB; optErr = s.optErr;
}

template<int MODE, int IM, int CHSET, class RG>
DXTEX_HD void lockstep_exhaustive_loop(const RG& rg, uint32_t& optA, uint32_t& optB, int& optErr)
{
    typedef LoopCfg<MODE, IM, CHSET> C;
    if (C::PREC == 0) return;
    int other;
    const int base = loop_base<MODE, IM, CHSET>(rg, optA, optB, &other);
    if (loop_is_settled<CHSET>(optErr, other)) return;
    ExhState s; VarPal<C::N> vp;
    bool has = exh_begin<MODE, IM, CHSET>(s, vp, optA, optB, optErr);
    // what the kernel does, one lane's worth: bound every candidate of the window (counting them down, stepping with exh_advance), set
    // the unbeaten ones aside, evaluate those exactly (here newest first, to exercise the order independence), take the minimum key
    int queue[128], n = 0;
    // a window that cannot hold an improvement is closed at once (exh_window_excluded): state "past the last row", nothing visited
    auto skip_excluded = [&]()
    {
#if defined(DXTEX_TABLE_STATS)
        // development statistics: windows that survive the test above but whose candidate could not win the block even in the best
        // case (window bound + lower bounds of the candidate's other subsets > an error already achieved for the block)
        if (has && !exh_window_excluded<MODE, IM, CHSET>(rg, vp, s, base))
        {
            const int lbw = exh_window_bound<MODE, IM, CHSET>(rg, vp, s, base);
            ++g_tabWin[MODE];
            if (long(lbw) + g_statOther > long(g_statTable)) ++g_tabWinOut[MODE];
            if (long(lbw) + g_statOther > long(g_statTablePrev)) ++g_tabWinOutPrev[MODE];
        }
#endif
        while (has && exh_window_excluded<MODE, IM, CHSET>(rg, vp, s, base))
        {
            s.o = s.oEnd;
            has = exh_next<MODE, IM, CHSET>(s, vp);
        }
    };
    skip_excluded();
    if (has) exh_peel<MODE, IM, CHSET>(rg, vp, s, base);
#if defined(DXTEX_ROW_STATS)
    // development statistics: candidates of the surviving windows that sit in strips of H rows (outer endpoint values) whose interval
    // bound is not below the error the window starts from
    auto row_stats = [&]()
    {
        if (!has) return;
        g_rowCand[MODE] += exh_remaining(s); ++g_rowWin[MODE];
        for (int H = 1; H <= 3; ++H)
            for (int o = s.o; o < s.oEnd; o += H)
            {
                const int oHi = (o + H - 1 < s.oEnd - 1) ? o + H - 1 : s.oEnd - 1;
                int n = 0;
                for (int oo = o; oo <= oHi; ++oo) { const int from = oo > s.lo ? oo : s.lo; n += (s.iEnd > from) ? s.iEnd - from : 0; }
                if (!n) continue;
                const int from = o > s.lo ? o : s.lo;
                ++g_rowTests[MODE][H];
                const bool out = exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, o, oHi, from, s.iEnd - 1) >= s.optErr;
                if (out) g_rowOut[MODE][H] += n;
                if (H == 2)
                {
                    // by distance of the strip from the row of the starting endpoints (the outer variable's current value)
                    const int cur = int(byte_of(s.aleb ? s.optA : s.optB, s.ch));
                    int dist = (cur < o) ? o - cur : (cur > oHi ? cur - oHi : 0);
                    dist = dist > 5 ? 5 : dist;
                    g_rowDistN[MODE][dist] += n; if (out) g_rowDistOut[MODE][dist] += n;
                }
            }
    };
    row_stats();
    // peel statistics: strips of H rows / columns taken off the four sides of the window's rectangle while their interval bound is not
    // below the starting error, L layers deep
    auto peel_stats = [&]()
    {
        if (!has) return;
        for (int H = 1; H <= 3; ++H)
            for (int L = 1; L <= 2; ++L)
            {
                int oA = s.o, oB = s.oEnd - 1, iA = s.lo, iB = s.iEnd - 1;     // rectangle of (outer, inner), inclusive; cells need inner >= outer
                auto count = [&](int a, int b, int c, int d) { int n = 0; for (int o = a; o <= b; ++o) { const int f = o > c ? o : c; n += (d >= f) ? d - f + 1 : 0; } return n; };
                const int before = count(oA, oB, iA, iB);
                int tests = 0;
                for (int l = 0; l < L; ++l)
                {
                    // low rows, high rows, low columns, high columns
                    if (oB - oA + 1 > H) { ++tests; const int f = oA > iA ? oA : iA; if (count(oA, oA + H - 1, iA, iB) == 0 || exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, oA, oA + H - 1, f, iB) >= s.optErr) oA += H; }
                    if (oB - oA + 1 > H) { ++tests; const int f = (oB - H + 1) > iA ? (oB - H + 1) : iA; if (count(oB - H + 1, oB, iA, iB) == 0 || exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, oB - H + 1, oB, f, iB) >= s.optErr) oB -= H; }
                    if (iB - iA + 1 > H) { ++tests; const int hi = (iA + H - 1) < oB ? oB : oB; const int oh = oB < iA + H - 1 ? oB : iA + H - 1; (void)hi;
                                           if (count(oA, oB, iA, iA + H - 1) == 0 || exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, oA, oh, iA > oA ? iA : oA, iA + H - 1) >= s.optErr) iA += H; }
                    if (iB - iA + 1 > H) { ++tests; const int oh = oB < iB ? oB : iB;
                                           if (count(oA, oB, iB - H + 1, iB) == 0 || exh_range_bound<MODE, IM, CHSET>(rg, vp, s, base, oA, oh, (iB - H + 1) > oA ? (iB - H + 1) : oA, iB) >= s.optErr) iB -= H; }
                }
                const int after = count(oA, oB, iA, iB);
                g_peelTests[MODE][H][L] += tests; g_peelOut[MODE][H][L] += before - after;
            }
    };
    peel_stats();
#endif
    uint32_t bestKey = has ? exh_start_key(s) : 0u;
    int rem = has ? exh_remaining(s) : 0;
    while (has)
    {
        const int code = exh_code(s);
        const int lb = exh_bound<MODE, IM, CHSET>(rg, vp, s.ch, s.aleb, s.o0, s.lo, code, base);
        if (exh_key(lb, code) < bestKey)
        {
            queue[n++] = code;
#if defined(DXTEX_COUNT_EVALS)
            ++g_pendCount[MODE];
#endif
        }
        exh_advance(s); --rem;
        const bool ended = rem == 0;
        if (n == 8 || ended)
        {
#if defined(DXTEX_COUNT_EVALS)
            if (n) ++g_drainCount[MODE];
#endif
            for (int j = n - 1; j >= 0; --j)
            {
                const uint32_t k = exh_key(exh_exact<MODE, IM, CHSET>(rg, vp, s.ch, s.aleb, s.o0, s.lo, queue[j], base), queue[j]);
                bestKey = k < bestKey ? k : bestKey;
            }
            n = 0;
        }
        if (ended)
        {
            exh_apply_key(s, bestKey);
            has = exh_next<MODE, IM, CHSET>(s, vp);          // the window is used up: commit, open the next one
            skip_excluded();
            if (has) exh_peel<MODE, IM, CHSET>(rg, vp, s, base);
#if defined(DXTEX_ROW_STATS)
            row_stats(); peel_stats();
#endif
            if (has) { bestKey = exh_start_key(s); rem = exh_remaining(s); }
        }
    }
    optA = s.optA; optB = s.optB; optErr = s.optErr;
}

template<int MODE, int IM, class RG>
DXTEX_HD void optimize_one_lockstep(const RG& rg, int orgErr, uint32_t orgA, uint32_t orgB, uint32_t& optA, uint32_t& optB)
{
    optA = orgA; optB = orgB;
    int err = orgErr;
    if constexpr (PaletteBits<MODE, IM>::AB == 0)
    {
        lockstep_perturb_loop<MODE, IM, CH_ALL>(rg, optA, optB, err);
        lockstep_exhaustive_loop<MODE, IM, CH_ALL>(rg, optA, optB, err);
    }
    else
    {
        lockstep_perturb_loop<MODE, IM, CH_COLOR>(rg, optA, optB, err);
        lockstep_perturb_loop<MODE, IM, CH_ALPHA>(rg, optA, optB, err);
        lockstep_exhaustive_loop<MODE, IM, CH_COLOR>(rg, optA, optB, err);
        lockstep_exhaustive_loop<MODE, IM, CH_ALPHA>(rg, optA, optB, err);
    }
}

// Everything Refine does for one subset (:3399-3463) up to, but excluding, the org-vs-opt decision,
// which needs the totals over all subsets of the candidate.
struct SubsetResult
{
    uint32_t orgA, orgB, optA, optB;
    uint64_t orgIdx1, orgIdx2, optIdx1, optIdx2;
    int orgErr, optErr;
};

// Refine, first half: quantise the seed, settle p-bits, assign indices (org candidate).
template<int MODE, int IM, class RG>
DXTEX_HD void refine_pre(const RG& rg, uint32_t seedA, uint32_t seedB, uint32_t anchorPos, SubsetResult& out)
{
    const uint32_t qa = quantize_endpoint<MODE>(seedA), qb = quantize_endpoint<MODE>(seedB);
    fix_pbits<MODE>(qa, qb, out.orgA, out.orgB);
    out.orgErr = assign_indices<MODE, IM>(rg, out.orgA, out.orgB, anchorPos, out.orgIdx1, out.orgIdx2);
}

// ... the same without the indices (what bc7_pre_kernel needs: endpoints after the anchor fix-up, and the error)
template<int MODE, int IM, class RG>
DXTEX_HD void refine_pre_err(const RG& rg, uint32_t seedA, uint32_t seedB, uint32_t anchorPos, SubsetResult& out)
{
    const uint32_t qa = quantize_endpoint<MODE>(seedA), qb = quantize_endpoint<MODE>(seedB);
    fix_pbits<MODE>(qa, qb, out.orgA, out.orgB);
    out.orgErr = assign_error<MODE, IM>(rg, out.orgA, out.orgB, anchorPos);
}

// Refine, second half: the optimised endpoints (from OptimizeOne) get their p-bits and indices (opt candidate).
template<int MODE, int IM, class RG>
DXTEX_HD void refine_post(const RG& rg, uint32_t oa, uint32_t ob, uint32_t anchorPos, SubsetResult& out)
{
    fix_pbits<MODE>(oa, ob, out.optA, out.optB);
    out.optErr = assign_indices<MODE, IM>(rg, out.optA, out.optB, anchorPos, out.optIdx1, out.optIdx2);
}

template<int MODE, int IM, class RG>
DXTEX_HD void refine_subset(const RG& rg, uint32_t seedA, uint32_t seedB, uint32_t anchorPos, SubsetResult& out)
{
    refine_pre<MODE, IM>(rg, seedA, seedB, anchorPos, out);
    uint32_t oa, ob;
#if defined(DXTEX_BC7_USE_LOCKSTEP)
    optimize_one_lockstep<MODE, IM>(rg, out.orgErr, out.orgA, out.orgB, oa, ob);
#else
    optimize_one<MODE, IM>(rg, out.orgErr, out.orgA, out.orgB, oa, ob);
#endif
    refine_post<MODE, IM>(rg, oa, ob, anchorPos, out);
}

// ---- float seed (OptimizeRGB / OptimizeRGBA with cSteps == 4, :1198-1555) -----------------------------
// `fpx` = the block's 16 float texels (r,g,b,a); only texels in `mask16` take part, in increasing order.
// Returns the endpoints clamped to [0,1], scaled by 255 and truncated with the +0.01 bias (:3543-3548).
// FULL: all 16 texels take part and `fpx` may be a register array (every texel loop is unrolled).
template<bool FULL, class F>
DXTEX_HD void for_masked(uint32_t mask16, F&& f)
{
    if constexpr (FULL)
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) f(i);
    }
    else
    {
        // set bits in ascending order: the trip count is the subset size, not 16 (the texels live in LDS, any index is fine)
        for (uint32_t m = mask16 & 0xFFFFu; m; m &= m - 1u) f(int(__builtin_ctz(m)));
    }
}

// The raw fit: X / Y are the end points OptimizeRGB / OptimizeRGBA return (pX, pY), before any clamping. In two pieces so that a
// kernel can schedule the Newton iterations of many fits itself (bc7_rough_kernel): fit_setup = everything up to the iteration
// loop (:3461-3531 for RGBA: bounding box, choice of the diagonal), fit_iterate = one trip of that loop (:3533-3606).

// Returns true when the loop has to run (false: X / Y are final - a degenerate box, :3478 and :3527).
// A1 (with RGBA): the caller has checked that the alpha of every texel of the block is exactly 1.0f (an opaque block - the common
// case). Then X[3] = Y[3] = 1 from the bounding box on, and every alpha term of the RGBA formulas is an exact zero: Dir[3] = 0,
// (p[3] - X[3]) * Dir[3] = 0, Diff[3] = (X[3] pc + Y[3] pd) - p[3] = (pc + pd) - 1 = 0 (float(2/3) + float(1/3) rounds to 1.0f), so
// dX[3] = dY[3] = 0 and the end points' alpha never moves; x + 0.0f = x. The A1 instantiation drops those terms (a quarter of the
// Newton loop) and keeps everything else of the RGBA path - the eight-direction choice (directions 2j and 2j + 1 tie, the strict
// compare keeps 2j, so bit 0 of iDirMax is clear) and the summed convergence test - bit for bit.
template<bool RGBA, bool FULL = false, bool A1 = false>
DXTEX_HD bool fit_setup(const float* fpx, uint32_t mask16, float (&X)[4], float (&Y)[4])
{
    static_assert(RGBA || !A1, "A1 is a variant of the RGBA fit");
    constexpr int NC = (RGBA && !A1) ? 4 : 3;
    if (RGBA) { X[0] = X[1] = X[2] = X[3] = 1.0f; Y[0] = Y[1] = Y[2] = Y[3] = 0.0f; }
    else { X[0] = X[1] = X[2] = 3.402823466e+38f; Y[0] = Y[1] = Y[2] = -3.402823466e+38f; X[3] = 0.0f; Y[3] = 0.0f; }
    if (A1) Y[3] = 1.0f;

    for_masked<FULL>(mask16, [&](int i)
        {
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                const float v = fpx[i * 4 + c];
                if (v < X[c]) X[c] = v;
                if (v > Y[c]) Y[c] = v;
            }
        });

    float AB[4];
#pragma unroll
    for (int c = 0; c < NC; ++c) AB[c] = Y[c] - X[c];
    float fAB = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
    if (RGBA && !A1) fAB = fAB + AB[3] * AB[3];

    if (fAB < 1.175494351e-38f) return false;

    const float fABInv = 1.0f / fAB;
    float Dir[4], Mid[4];
#pragma unroll
    for (int c = 0; c < NC; ++c) { Dir[c] = AB[c] * fABInv; Mid[c] = (X[c] + Y[c]) * 0.5f; }

    float fDir[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) fDir[d] = 0.0f;
    for_masked<FULL>(mask16, [&](int i)
        {
            float Pt[4];
#pragma unroll
            for (int c = 0; c < NC; ++c) Pt[c] = (fpx[i * 4 + c] - Mid[c]) * Dir[c];
            float f;
            if (RGBA && A1)
            {
                // Pt[3] = 0: directions 2j and 2j + 1 of the RGBA list coincide; j in the order (+,+) (+,-) (-,+) (-,-)
                f = Pt[0] + Pt[1] + Pt[2]; fDir[0] += f * f;
                f = Pt[0] + Pt[1] - Pt[2]; fDir[1] += f * f;
                f = Pt[0] - Pt[1] + Pt[2]; fDir[2] += f * f;
                f = Pt[0] - Pt[1] - Pt[2]; fDir[3] += f * f;
            }
            else if (RGBA)
            {
                f = Pt[0] + Pt[1] + Pt[2] + Pt[3]; fDir[0] += f * f;
                f = Pt[0] + Pt[1] + Pt[2] - Pt[3]; fDir[1] += f * f;
                f = Pt[0] + Pt[1] - Pt[2] + Pt[3]; fDir[2] += f * f;
                f = Pt[0] + Pt[1] - Pt[2] - Pt[3]; fDir[3] += f * f;
                f = Pt[0] - Pt[1] + Pt[2] + Pt[3]; fDir[4] += f * f;
                f = Pt[0] - Pt[1] + Pt[2] - Pt[3]; fDir[5] += f * f;
                f = Pt[0] - Pt[1] - Pt[2] + Pt[3]; fDir[6] += f * f;
                f = Pt[0] - Pt[1] - Pt[2] - Pt[3]; fDir[7] += f * f;
            }
            else
            {
                f = Pt[0] + Pt[1] + Pt[2]; fDir[0] += f * f;
                f = Pt[0] + Pt[1] - Pt[2]; fDir[1] += f * f;
                f = Pt[0] - Pt[1] + Pt[2]; fDir[2] += f * f;
                f = Pt[0] - Pt[1] - Pt[2]; fDir[3] += f * f;
            }
        });

    float fDirMax = fDir[0];
    int iDirMax = 0;
#pragma unroll
    for (int d = 1; d < ((RGBA && !A1) ? 8 : 4); ++d)
        if (fDir[d] > fDirMax) { fDirMax = fDir[d]; iDirMax = d; }

    if (RGBA && A1)
    {
        if (iDirMax & 2) { const float t = X[1]; X[1] = Y[1]; Y[1] = t; }
        if (iDirMax & 1) { const float t = X[2]; X[2] = Y[2]; Y[2] = t; }
    }
    else if (RGBA)
    {
        if (iDirMax & 4) { const float t = X[1]; X[1] = Y[1]; Y[1] = t; }
        if (iDirMax & 2) { const float t = X[2]; X[2] = Y[2]; Y[2] = t; }
        if (iDirMax & 1) { const float t = X[3]; X[3] = Y[3]; Y[3] = t; }
    }
    else
    {
        if (iDirMax & 2) { const float t = X[1]; X[1] = Y[1]; Y[1] = t; }
        if (iDirMax & 1) { const float t = X[2]; X[2] = Y[2]; Y[2] = t; }
    }

    return !(fAB < 1.0f / 4096.0f);
}

// One trip of the Newton loop; returns true when the loop ends with this trip (either break). The caller runs at most 8 trips.
template<bool RGBA, bool FULL = false, bool A1 = false>
DXTEX_HD bool fit_iterate(const float* fpx, uint32_t mask16, float (&X)[4], float (&Y)[4])
{
    constexpr float fEpsilon = (0.25f / 64.0f) * (0.25f / 64.0f);
    constexpr int NC = (RGBA && !A1) ? 4 : 3;
    const float fSteps = 3.0f;
#if defined(DXTEX_FIT_STATS)
    ++g_iters;
#endif
    float Dir[4];
#pragma unroll
    for (int c = 0; c < NC; ++c) Dir[c] = Y[c] - X[c];
    float fLen = Dir[0] * Dir[0] + Dir[1] * Dir[1] + Dir[2] * Dir[2];
    if (RGBA && !A1) fLen = fLen + Dir[3] * Dir[3];
    if (fLen < (1.0f / 4096.0f)) return true;

    const float fScale = fSteps / fLen;
#pragma unroll
    for (int c = 0; c < NC; ++c) Dir[c] *= fScale;

    float d2X = 0.0f, d2Y = 0.0f;
    float dX[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, dY[4] = { 0.0f, 0.0f, 0.0f, 0.0f };

    for_masked<FULL>(mask16, [&](int i)
        {
            float p[4];
#pragma unroll
            for (int c = 0; c < NC; ++c) p[c] = fpx[i * 4 + c];
            float fDot = (p[0] - X[0]) * Dir[0] + (p[1] - X[1]) * Dir[1] + (p[2] - X[2]) * Dir[2];
            if (RGBA && !A1) fDot = fDot + (p[3] - X[3]) * Dir[3];

            // fDot <= 0 -> step 0, fDot >= fSteps -> step 3, else uint32(fDot + 0.5f) (:1300-1306): the clamp maps the outer cases
            // onto the same conversion (the sum is in [0.5, 3.5]: truncation = floor, one instruction instead of two conversions).
            // The table lookups pC4 = {1, 2/3, 1/3, 0}, pD4 = {0, 1/3, 2/3, 1} (BC.cpp:26-31) are ONE multiplication each, as in
            // bc15_encode.hip's StepCoef: with r = float(1/3) = 0x3EAAAAAB, 0 r = 0, 1 r = r, 2 r = 0x3F2AAAAB = float(2/3) (a
            // power-of-two scaling) and 3 r = 1.00000003 rounds to 1.0f - every entry exact. (Until round 6 three float
            // compare-selects per coefficient: 9 of the loop's ~40 instructions per texel, all at the 4-cycle rate.)
            const float kStep = floorf(fminf(fmaxf(fDot, 0.0f), fSteps) + 0.5f);
            const float pc = (fSteps - kStep) * (1.0f / 3.0f);
            const float pd = kStep * (1.0f / 3.0f);
            const float fC = pc * (1.0f / 8.0f);
            const float fD = pd * (1.0f / 8.0f);
            d2X += fC * pc;
            d2Y += fD * pd;
#pragma unroll
            for (int c = 0; c < NC; ++c)
            {
                const float Diff = (X[c] * pc + Y[c] * pd) - p[c];
                dX[c] += Diff * fC;
                dY[c] += Diff * fD;
            }
        });

    if (d2X > 0.0f)
    {
        const float f = -1.0f / d2X;
#pragma unroll
        for (int c = 0; c < NC; ++c) X[c] += dX[c] * f;
    }
    if (d2Y > 0.0f)
    {
        const float f = -1.0f / d2Y;
#pragma unroll
        for (int c = 0; c < NC; ++c) Y[c] += dY[c] * f;
    }

    if (RGBA && A1)
    {
        const float ex = dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2];
        const float ey = dY[0] * dY[0] + dY[1] * dY[1] + dY[2] * dY[2];
        return (ex < fEpsilon) && (ey < fEpsilon);
    }
    if (RGBA)
    {
        const float ex = dX[0] * dX[0] + dX[1] * dX[1] + dX[2] * dX[2] + dX[3] * dX[3];
        const float ey = dY[0] * dY[0] + dY[1] * dY[1] + dY[2] * dY[2] + dY[3] * dY[3];
        return (ex < fEpsilon) && (ey < fEpsilon);
    }
    return (dX[0] * dX[0] < fEpsilon) && (dX[1] * dX[1] < fEpsilon) && (dX[2] * dX[2] < fEpsilon) &&
           (dY[0] * dY[0] < fEpsilon) && (dY[1] * dY[1] < fEpsilon) && (dY[2] * dY[2] < fEpsilon);
}

template<bool RGBA, bool FULL = false, bool A1 = false>
DXTEX_HD void seed_fit(const float* fpx, uint32_t mask16, float (&X)[4], float (&Y)[4])
{
    if (!fit_setup<RGBA, FULL, A1>(fpx, mask16, X, Y)) return;
    for (int iter = 0; iter < 8; ++iter)
        if (fit_iterate<RGBA, FULL, A1>(fpx, mask16, X, Y)) break;
}

// X / Y of a fit -> the 8-bit end points Refine and RoughMSE start from: clamped to [0,1], scaled by 255 and truncated with the +0.01
// bias (:3543-3548).
template<bool RGBA>
DXTEX_HD void fit_to_bytes(const float (&X)[4], const float (&Y)[4], uint32_t& outA, uint32_t& outB)
{
    constexpr int NC = RGBA ? 4 : 3;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
    {
        float x = X[c], y = Y[c];
        x = x > 0.0f ? x : 0.0f; x = x < 1.0f ? x : 1.0f;       // std::min(fMax, std::max(fMin, v))
        y = y > 0.0f ? y : 0.0f; y = y < 1.0f ? y : 1.0f;
        x *= 255.0f; y *= 255.0f;
        a |= (uint32_t(x + 0.01f) & 0xFFu) << (8 * c);
        b |= (uint32_t(y + 0.01f) & 0xFFu) << (8 * c);
    }
    outA = a; outB = b;
}

template<bool RGBA, bool FULL = false, bool A1 = false>
DXTEX_HD void seed_endpoints(const float* fpx, uint32_t mask16, uint32_t& outA, uint32_t& outB)
{
    float X[4], Y[4];
    seed_fit<RGBA, FULL, A1>(fpx, mask16, X, Y);
    fit_to_bytes<RGBA>(X, Y, outA, outB);
}

// RoughMSE's palette error from *unquantised* 8-bit endpoints (:3572-3596) for one region.
template<int CB, int AB, class RG>
DXTEX_HD int rough_error(const RG& rg, uint32_t epA, uint32_t epB)
{
    constexpr int NC = 1 << CB, NA = AB ? (1 << AB) : 1;
    int total = rg.p2sum;
    uint32_t pal[NC], nq2[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i)
    {
        pal[i] = lerp_bytes(epA, epB, weight(CB, i));
        if (AB != 0) pal[i] &= 0x00FFFFFFu;
        nq2[i] = 0u - udot4(pal[i], pal[i]);
    }
    int pa[NA];
    if (AB != 0)
    {
        const int a0 = int(epA >> 24), a1 = int(epB >> 24);
#pragma unroll
        for (int i = 0; i < NA; ++i) pa[i] = int(lerp1(lerp_base(uint32_t(a0)), a1 - a0, weight(AB ? AB : 2, i)));
    }
    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k);
        const uint32_t pc = (AB != 0) ? (p & 0x00FFFFFFu) : p;
        int sc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) sc[i] = score(pc, pal[i], nq2[i]);
        total -= first_peak(sc);
        if (AB != 0)
        {
            const int al = int(p >> 24);
            int su[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) su[i] = 2 * al * pa[i] - pa[i] * pa[i];
            total -= first_peak(su);
        }
    });
    return total;
}

// RoughMSE for the 3-bit and the 2-bit palette of the same endpoints in ONE pass over the texels (what bc7_rough_kernel needs of every
// two-subset fit: mode 1 ranks the shapes by the first, modes 3 / 7 by the second). Both palettes start at A and end at B (weights 0 and 64),
// so the scores of those two entries are shared: ten scores per texel instead of twelve, one texel fetch instead of two.
template<class RG>
DXTEX_HD void rough_error_3_2(const RG& rg, uint32_t epA, uint32_t epB, int& e3, int& e2)
{
    uint32_t pal3[8], nq3[8], pal2[2], nq2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { pal3[i] = lerp_bytes(epA, epB, weight(3, i)); nq3[i] = 0u - udot4(pal3[i], pal3[i]); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { pal2[i] = lerp_bytes(epA, epB, weight(2, i + 1)); nq2[i] = 0u - udot4(pal2[i], pal2[i]); }
    int t3 = rg.p2sum, t2 = rg.p2sum;
    for_texels(rg, [&](int k)
    {
        const uint32_t p = rg.fetch(k);
        int sc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) sc[i] = score(p, pal3[i], nq3[i]);
        const int sd[4] = { sc[0], score(p, pal2[0], nq2[0]), score(p, pal2[1], nq2[1]), sc[7] };
        t3 -= first_peak(sc);
        t2 -= first_peak(sd);
    });
    e3 = t3; e2 = t2;
}

// ---- bit packing (EmitBlock, :3221-3308) ------------------------------------------------------------------
struct Bits128
{
    uint64_t lo, hi;
    uint32_t pos;
    DXTEX_HD void init() { lo = 0; hi = 0; pos = 0; }
    DXTEX_HD void put(uint32_t nbits, uint32_t value)
    {
        if (nbits == 0) return;
        const uint64_t v = uint64_t(value) & ((uint64_t(1) << nbits) - 1);
        if (pos < 64)
        {
            lo |= v << pos;
            if (pos + nbits > 64) hi |= v >> (64 - pos);
        }
        else hi |= v << (pos - 64);
        pos += nbits;
    }
};

// ---- exact pruning ------------------------------------------------------------------------------------------------------
// A lower bound on the error that ANY endpoints and indices of a BC7 mode can reach on one subset. Every palette entry is
// ((64 - w) e0 + w e1 + 32) >> 6 per channel, i.e. within 0.5 per channel (0.5 sqrt(C) in distance) of a point on the real
// line through the unquantised endpoints e0, e1. So a texel's distance to its palette entry is at least its distance to that
// line minus 0.5 sqrt(C), and over the subset (triangle inequality in l2)
//     error >= ( sqrt(sum_t dist(p_t, line)^2) - 0.5 sqrt(C n) )^2   whenever the bracket is positive,
// and sum_t dist^2 is at least what the best-fit line leaves: tr(S) - lambda_max(S), S the scatter matrix of the n texels over
// the C channels the line lives in (3 colour channels, or 4 when alpha is interpolated with them). lambda_max is bounded from
// ABOVE by ||N^16||_F^(1/16) (N = S / tr S, four squarings: (sum lambda_i^32)^(1/32) >= lambda_max, and tight unless the
// two largest eigenvalues are within a few per cent). A candidate partition whose bound exceeds an error some other
// candidate or mode of the block has already reached can never win D3DX_BC7::Encode's "first minimum" (:2835-2848), so
// its OptimizeOne search (:3045-3110) is skipped - the output does not change, only work that cannot matter is dropped.
// Returned rounded down (and shaved) so that float rounding can only weaken the bound.
template<int C>
DXTEX_HD int subset_lower_bound_c(const uint32_t* pix, uint32_t mask16, uint32_t rot)
{
    // C == 3: the fourth channel does not take part; every term it would contribute is an exact zero (0 * finite, x + 0), so the
    // three-channel form below is the same number with 18 instead of 40 products per squaring
    constexpr bool Q = (C == 4);
    uint32_t n = 0, s[4] = { 0, 0, 0, 0 }, ss[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };    // ss: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
    for (uint32_t i = 0; i < 16; ++i)
        if ((mask16 >> i) & 1u)
        {
            const uint32_t p = rotate_pixel(pix[i], rot);
            const uint32_t c0 = p & 0xFFu, c1 = (p >> 8) & 0xFFu, c2 = (p >> 16) & 0xFFu, c3 = Q ? (p >> 24) : 0u;
            ++n; s[0] += c0; s[1] += c1; s[2] += c2;
            ss[0] += c0 * c0; ss[1] += c0 * c1; ss[2] += c0 * c2;
            ss[4] += c1 * c1; ss[5] += c1 * c2;
            ss[7] += c2 * c2;
            if (Q) { s[3] += c3; ss[3] += c0 * c3; ss[6] += c1 * c3; ss[8] += c2 * c3; ss[9] += c3 * c3; }
        }
    if (n < 2) return 0;
    // M = n * S, exact integers (|entries| <= 16 * 16 * 255^2)
    const int M00 = int(n * ss[0]) - int(s[0] * s[0]), M01 = int(n * ss[1]) - int(s[0] * s[1]), M02 = int(n * ss[2]) - int(s[0] * s[2]), M03 = Q ? int(n * ss[3]) - int(s[0] * s[3]) : 0;
    const int M11 = int(n * ss[4]) - int(s[1] * s[1]), M12 = int(n * ss[5]) - int(s[1] * s[2]), M13 = Q ? int(n * ss[6]) - int(s[1] * s[3]) : 0;
    const int M22 = int(n * ss[7]) - int(s[2] * s[2]), M23 = Q ? int(n * ss[8]) - int(s[2] * s[3]) : 0, M33 = Q ? int(n * ss[9]) - int(s[3] * s[3]) : 0;
    const int T = M00 + M11 + M22 + M33;
    if (T <= 0) return 0;
    const double inv = 1.0 / double(T);
    double a00 = M00 * inv, a01 = M01 * inv, a02 = M02 * inv, a03 = M03 * inv, a11 = M11 * inv, a12 = M12 * inv, a13 = M13 * inv,
           a22 = M22 * inv, a23 = M23 * inv, a33 = M33 * inv;
    for (int k = 0; k < 4; ++k)
    {
        double b00 = a00 * a00 + a01 * a01 + a02 * a02;
        double b01 = a00 * a01 + a01 * a11 + a02 * a12;
        double b02 = a00 * a02 + a01 * a12 + a02 * a22;
        double b11 = a01 * a01 + a11 * a11 + a12 * a12;
        double b12 = a01 * a02 + a11 * a12 + a12 * a22;
        double b22 = a02 * a02 + a12 * a12 + a22 * a22;
        double b03 = 0.0, b13 = 0.0, b23 = 0.0, b33 = 0.0;
        if (Q)
        {
            b00 += a03 * a03; b01 += a03 * a13; b02 += a03 * a23; b11 += a13 * a13; b12 += a13 * a23; b22 += a23 * a23;
            b03 = a00 * a03 + a01 * a13 + a02 * a23 + a03 * a33;
            b13 = a01 * a03 + a11 * a13 + a12 * a23 + a13 * a33;
            b23 = a02 * a03 + a12 * a13 + a22 * a23 + a23 * a33;
            b33 = a03 * a03 + a13 * a13 + a23 * a23 + a33 * a33;
        }
        a00 = b00; a01 = b01; a02 = b02; a03 = b03; a11 = b11; a12 = b12; a13 = b13; a22 = b22; a23 = b23; a33 = b33;
    }
    double f2 = a00 * a00 + a11 * a11 + a22 * a22;
    if (Q) f2 += a33 * a33;
    double off = a01 * a01 + a02 * a02;
    if (Q) off += a03 * a03;
    off += a12 * a12;
    if (Q) { off += a13 * a13; off += a23 * a23; }
    f2 = f2 + 2.0 * off;
    double lam = sqrt(sqrt(sqrt(sqrt(sqrt(f2)))));          // ||N^16||_F ^ (1/16)
    lam = lam * (1.0 + 1e-9);
    if (lam >= 1.0) return 0;
    const double resid = double(T) * (1.0 - lam) / double(n);       // tr(S) - lambda_max(S), S = M / n
    // The rounding slack is per channel that takes part. A channel that is CONSTANT over the subset has no variance and no covariance: the
    // scatter matrix, its largest eigenvalue and the residual are those of the other channels alone, and the error on those channels alone
    // (the constant channel's share is >= 0) is bounded by the same argument one dimension lower - the projection of the palette's line is a
    // line there, its entries within 0.5 per channel of it. So only the channels that vary pay slack: alpha of an opaque block in mode 6, the
    // swapped-in constant of rotations 1 - 3 in modes 4 / 5, grey or single-hue subsets anywhere (round 5).
    const int varying = (M00 > 0 ? 1 : 0) + (M11 > 0 ? 1 : 0) + (M22 > 0 ? 1 : 0) + ((Q && M33 > 0) ? 1 : 0);
    const double d = sqrt(resid) - 0.5 * sqrt(double(varying) * double(n)) - 1e-3;
    if (d <= 0.0) return 0;
    const double lb = d * d * 0.99999 - 1.0;
    return (lb > 0.0) ? int(lb) : 0;
}
DXTEX_HD int subset_lower_bound(const uint32_t* pix, uint32_t mask16, uint32_t rot, int C)
{
    return (C == 4) ? subset_lower_bound_c<4>(pix, mask16, rot) : subset_lower_bound_c<3>(pix, mask16, rot);
}

// The separate-alpha modes (4, 5) give the fourth slot of the (rotated) texel a scalar palette of K = 2^bits entries of its own.
// Whatever those K values are, the slot's error over the 16 texels is at least the optimum of 1-D K-means on the 16 values:
// exact dynamic programme over the sorted values (clusters are runs), D_k(j) = min_i D_{k-1}(i-1) + SSE(i..j), fully unrolled
// so that every array index is a constant. fp32 with a downward margin (values <= 16 * 255^2).
template<int K>
DXTEX_HD int scalar_kmeans_lower_bound(const uint32_t* pix, uint32_t rot)
{
    int v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = int(rotate_pixel(pix[i], rot) >> 24);
    // bitonic sorting network, ascending
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i)
            {
                const int l = i ^ j;
                if (l > i)
                {
                    const int lo = (v[i] < v[l]) ? v[i] : v[l], hi = (v[i] < v[l]) ? v[l] : v[i];
                    if ((i & k) == 0) { v[i] = lo; v[l] = hi; } else { v[i] = hi; v[l] = lo; }
                }
            }
    if (v[0] == v[15]) return 0;
    float s1[17], s2[17];
    s1[0] = 0.0f; s2[0] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[i + 1] = s1[i] + float(v[i]); s2[i + 1] = s2[i] + float(v[i] * v[i]); }      // exact: < 2^24
    float d[16], e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float m = s1[j + 1]; d[j] = s2[j + 1] - m * m * (1.0f / float(j + 1)); }
#pragma unroll
    for (int k = 2; k <= K; ++k)
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            float best = (j < k) ? 0.0f : 3.0e38f;          // at most k points: one cluster each
            if (j >= k)
            {
#pragma unroll
                for (int i = k - 1; i <= j; ++i)
                {
                    const float m = s1[j + 1] - s1[i];
                    const float c = (s2[j + 1] - s2[i]) - m * m * (1.0f / float(j - i + 1));
                    const float t = d[i - 1] + ((c > 0.0f) ? c : 0.0f);
                    best = (t < best) ? t : best;
                }
            }
            e[j] = best;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = e[j];
    }
    const float lb = d[15] * 0.9999f - 4.0f;
    return (lb > 0.0f) ? int(lb) : 0;
}

// eps[s] = (A, B) of subset s in RGBAPrecWithP units; idx1/idx2 = 4 bits per texel position;
// anchors = texel positions of the subset anchors (anchor[0] == 0).
template<int MODE>
DXTEX_HD void emit_block(uint32_t shape, uint32_t rot, uint32_t im, const uint32_t (&epA)[3], const uint32_t (&epB)[3],
                         uint64_t idx1, uint64_t idx2, const uint32_t (&anchor)[3], uint64_t& outLo, uint64_t& outHi)
{
    typedef ModeInfo<MODE> MI;
    Bits128 w; w.init();
    w.put(MODE, 0); w.put(1, 1);
    w.put(MI::ROTBITS, rot);
    w.put(MI::IMBITS, im);
    w.put(MI::PARTBITS, shape);
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
    {
        constexpr int dummy = 0; (void)dummy;
        const int prec = (ch == 3) ? MI::AP : MI::CP;
        const int precP = (ch == 3) ? MI::APP : MI::CPP;
#pragma unroll
        for (int s = 0; s < MI::NS; ++s)
        {
            const uint32_t a = byte_of(epA[s], ch), b = byte_of(epB[s], ch);
            if (prec == precP) { w.put(prec, a); w.put(prec, b); }
            else { w.put(prec, a >> 1); w.put(prec, b >> 1); }
        }
    }
    if (MI::PB == 1)
    {
#pragma unroll
        for (int s = 0; s < MI::NS; ++s) { w.put(1, epA[s] & 1u); w.put(1, epB[s] & 1u); }
    }
    else if (MI::PB == 2)
    {
        // shared p-bit: majority over the six LSBs of the subset, which fix_pbits has made unanimous
#pragma unroll
        for (int s = 0; s < MI::NS; ++s) w.put(1, epA[s] & 1u);
    }
    // first index set (colour / combined), anchors one bit short; uIndexMode selects which set goes first
    const uint64_t i1 = im ? idx2 : idx1;
    const uint64_t i2 = im ? idx1 : idx2;
    for (uint32_t i = 0; i < 16; ++i)
    {
        bool isAnchor = (i == 0);
#pragma unroll
        for (int s = 1; s < MI::NS; ++s) isAnchor = isAnchor || (i == anchor[s]);
        w.put(isAnchor ? MI::IB - 1 : MI::IB, uint32_t(i1 >> (4 * i)) & 15u);
    }
    if (MI::IB2)
        for (uint32_t i = 0; i < 16; ++i)
            w.put(i ? MI::IB2 : MI::IB2 - 1, uint32_t(i2 >> (4 * i)) & 15u);
    outLo = w.lo; outHi = w.hi;
}

} // namespace bc7
} // namespace dxtex
