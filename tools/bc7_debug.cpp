// TEST INFRASTRUCTURE / DEVELOPMENT TOOL (not part of the product; built by oracle/Makefile into oracle/_ref/bc7_core_check and
// run by tests/test_bc7_core_cpu.py): runs the BC7 core of
// directxtex_amd/csrc/bc7_core.h on the HOST, next to the reference's D3DX_BC7 class compiled in place
// with its private members exposed, and compares them stage by stage (seed / RoughMSE / Refine / final
// block). Build + run:  tools/run_bc7_debug.sh [ntiles] [seed]
#define DXTEX_HOST_DEBUG 1
#define private public
#define protected public
#include "BC6HBC7.cpp"     // the reference, in place (-I/root/reference/DirectXTex), against oracle/shim
#undef private
#undef protected

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include "../directxtex_amd/csrc/bc67_tables.h"
#include "../directxtex_amd/csrc/bc7_core.h"

using namespace dxtex;
using namespace dxtex::bc7;

static uint32_t g_rng = 12345;
static uint32_t rnd() { g_rng = g_rng * 1664525u + 1013904223u; return g_rng >> 8; }

struct HostBlock
{
    float f[64];
    uint32_t ldr[16];
    bool hasAlpha;
};

static void make_block(HostBlock& b, const uint8_t* px)
{
    b.hasAlpha = false;
    for (int i = 0; i < 16; ++i)
    {
        uint32_t l = 0;
        for (int c = 0; c < 4; ++c)
        {
            const float v = float(px[i * 4 + c]) * (1.0f / 255.0f);
            b.f[i * 4 + c] = v;
            float q = v * 255.0f + 0.01f;
            q = (q < 255.0f) ? q : 255.0f; q = (0.0f < q) ? q : 0.0f;
            l |= (uint32_t(q) & 0xFF) << (8 * c);
        }
        b.ldr[i] = l;
        if ((l >> 24) != 0xFF) b.hasAlpha = true;
    }
}

struct Cand { uint32_t err, ord; uint64_t lo, hi; bool valid; };

static void seeds2(const HostBlock& b, uint32_t shape, int region, Region& rg, uint32_t& A, uint32_t& B, uint32_t& anchor)
{
    const uint32_t m1 = kPart2Mask[shape];
    const uint32_t m = region ? m1 : ((~m1) & 0xFFFF);
    region_init(rg, b.ldr, m);
    if (rg.np == 1) { A = b.ldr[rg.pos(0)]; B = A; }
    else if (rg.np == 2) { A = b.ldr[rg.pos(0)]; B = b.ldr[rg.pos(1)]; }
    else seed_endpoints<true>(b.f, m, A, B);
    anchor = region ? kAnchor2[shape] : 0;
}

template<int MODE>
static Cand refine2_one(const HostBlock& b, uint32_t shape, int rank)
{
    SubsetResult r[2];
    for (int region = 0; region < 2; ++region)
    {
        Region rg; uint32_t A, B, anchor;
        seeds2(b, shape, region, rg, A, B, anchor);
#if defined(DXTEX_TABLE_STATS)
        { const uint32_t m1 = kPart2Mask[shape]; const uint32_t mo = region ? ((~m1) & 0xFFFF) : m1;      // the OTHER region
          g_statOther = subset_lower_bound(b.ldr, mo, 0u, (MODE >= 6) ? 4 : 3); }
#endif
        refine_subset<MODE, 0>(rg, A, B, anchor, r[region]);
    }
    const int orgTot = r[0].orgErr + r[1].orgErr, optTot = r[0].optErr + r[1].optErr;
    const bool useOpt = optTot < orgTot;
    Cand c; c.valid = true;
    c.err = uint32_t(useOpt ? optTot : orgTot);
    c.ord = MODE * 128 + rank;
    const uint32_t epA[3] = { useOpt ? r[0].optA : r[0].orgA, useOpt ? r[1].optA : r[1].orgA, 0 };
    const uint32_t epB[3] = { useOpt ? r[0].optB : r[0].orgB, useOpt ? r[1].optB : r[1].orgB, 0 };
    const uint64_t idx = useOpt ? (r[0].optIdx1 | r[1].optIdx1) : (r[0].orgIdx1 | r[1].orgIdx1);
    const uint32_t anchor[3] = { 0, kAnchor2[shape], 0 };
    emit_block<MODE>(shape, 0, 0, epA, epB, idx, 0, anchor, c.lo, c.hi);
    return c;
}

// modes 0 and 2: three subsets (D3DX_BC7::Encode with BC_FLAGS_USE_3SUBSETS, BC6HBC7.cpp:2805-2815)
static uint32_t mask3(uint32_t shape, uint32_t region)
{
    const uint32_t bits = kPart3Bits[shape];
    uint32_t m = 0;
    for (int i = 0; i < 16; ++i) if (((bits >> (2 * i)) & 3u) == region) m |= 1u << i;
    return m;
}

static void seeds3(const HostBlock& b, uint32_t shape, uint32_t region, Region& rg, uint32_t& A, uint32_t& B)
{
    const uint32_t m = mask3(shape, region);
    region_init(rg, b.ldr, m);
    if (rg.np == 1) { A = b.ldr[rg.pos(0)]; B = A; }
    else if (rg.np == 2) { A = b.ldr[rg.pos(0)]; B = b.ldr[rg.pos(1)]; }
    else seed_endpoints<true>(b.f, m, A, B);
}

template<int MODE>
static Cand refine3_one(const HostBlock& b, uint32_t shape, int rank)
{
    SubsetResult r[3];
    const uint32_t anchor[3] = { 0, uint32_t(kAnchor3[shape] & 15), uint32_t(kAnchor3[shape] >> 4) };
    int orgTot = 0, optTot = 0;
    for (uint32_t region = 0; region < 3; ++region)
    {
        Region rg; uint32_t A, B;
        seeds3(b, shape, region, rg, A, B);
        refine_subset<MODE, 0>(rg, A, B, anchor[region], r[region]);
        orgTot += r[region].orgErr; optTot += r[region].optErr;
    }
    const bool useOpt = optTot < orgTot;
    Cand c; c.valid = true;
    c.err = uint32_t(useOpt ? optTot : orgTot);
    c.ord = MODE * 128 + rank;
    uint32_t epA[3], epB[3]; uint64_t idx = 0;
    for (int s = 0; s < 3; ++s) { epA[s] = useOpt ? r[s].optA : r[s].orgA; epB[s] = useOpt ? r[s].optB : r[s].orgB; idx |= useOpt ? r[s].optIdx1 : r[s].orgIdx1; }
    emit_block<MODE>(shape, 0, 0, epA, epB, idx, 0, anchor, c.lo, c.hi);
    return c;
}

// the `count` best of the first `nshapes` 3-subset shapes by rough error with IDXBITS-bit indices (stable: ties keep shape order)
template<int IDXBITS>
static void rough_list3(const HostBlock& b, int nshapes, int count, uint32_t* list)
{
    int e[64]; uint32_t sh[64];
    for (int s = 0; s < nshapes; ++s)
    {
        e[s] = 0; sh[s] = uint32_t(s);
        for (uint32_t region = 0; region < 3; ++region)
        {
            Region rg; uint32_t A, B;
            seeds3(b, uint32_t(s), region, rg, A, B);
            e[s] += rough_error<IDXBITS, 0>(rg, A, B);
        }
    }
    for (int i = 0; i < count; ++i)
        for (int j = i + 1; j < nshapes; ++j)
            if (e[i] > e[j]) { std::swap(e[i], e[j]); std::swap(sh[i], sh[j]); }
    for (int i = 0; i < count; ++i) list[i] = sh[i];
}

template<int MODE, int IM>
static Cand refine1_one(const HostBlock& b, uint32_t rot)
{
    Block16 rg;
    block16_init(rg, b.ldr, MODE == 6 ? 0u : rot);
    uint32_t A, B;
    if (MODE == 6) seed_endpoints<true>(b.f, 0xFFFF, A, B);
    else
    {
        seed_endpoints<false>(b.f, 0xFFFF, A, B);
        uint32_t mn = 255, mx = 0;
        for (int i = 0; i < 16; ++i) { const uint32_t al = rg.px[i] >> 24; mn = al < mn ? al : mn; mx = al > mx ? al : mx; }
        A = (A & 0xFFFFFF) | (mn << 24); B = (B & 0xFFFFFF) | (mx << 24);
    }
    SubsetResult r;
    refine_subset<MODE, IM>(rg, A, B, 0, r);
    const bool useOpt = r.optErr < r.orgErr;
    Cand c; c.valid = true;
    c.err = uint32_t(useOpt ? r.optErr : r.orgErr);
    const uint32_t sub = (MODE == 4) ? rot * 2 + IM : rot;
    c.ord = MODE * 128 + sub * 16;
    const uint32_t epA[3] = { useOpt ? r.optA : r.orgA, 0, 0 }, epB[3] = { useOpt ? r.optB : r.orgB, 0, 0 };
    const uint32_t anchor[3] = { 0, 0, 0 };
    emit_block<MODE>(0, rot, IM, epA, epB, useOpt ? r.optIdx1 : r.orgIdx1, useOpt ? r.optIdx2 : r.orgIdx2, anchor, c.lo, c.hi);
    return c;
}

static void rough_lists(const HostBlock& b, int* e3, int* e2, uint32_t* l3, uint32_t* l2)
{
    for (uint32_t s = 0; s < 64; ++s)
    {
        e3[s] = e2[s] = 0;
        for (int region = 0; region < 2; ++region)
        {
            Region rg; uint32_t A, B, anchor;
            seeds2(b, s, region, rg, A, B, anchor);
            e3[s] += rough_error<3, 0>(rg, A, B);
            e2[s] += rough_error<2, 0>(rg, A, B);
        }
    }
    for (int pass = 0; pass < 2; ++pass)
    {
        int e[64]; uint32_t sh[64];
        for (int s = 0; s < 64; ++s) { e[s] = pass ? e2[s] : e3[s]; sh[s] = s; }
        for (int i = 0; i < 16; ++i)
            for (int j = i + 1; j < 64; ++j)
                if (e[i] > e[j]) { std::swap(e[i], e[j]); std::swap(sh[i], sh[j]); }
        for (int i = 0; i < 16; ++i) (pass ? l2 : l3)[i] = sh[i];
    }
}

static void better(Cand& best, const Cand& c)
{
    if (!c.valid) return;
    const uint64_t kb = best.valid ? ((uint64_t(best.err) << 32) | best.ord) : ~0ull;
    const uint64_t kc = (uint64_t(c.err) << 32) | c.ord;
    if (kc < kb) best = c;
}

static void ref_setup(D3DX_BC7::EncodeParams& EP, const HDRColorA* pIn, int mode, int rot)
{
    for (size_t i = 0; i < 16; ++i)
    {
        EP.aLDRPixels[i].r = uint8_t(std::max<float>(0.0f, std::min<float>(255.0f, pIn[i].r * 255.0f + 0.01f)));
        EP.aLDRPixels[i].g = uint8_t(std::max<float>(0.0f, std::min<float>(255.0f, pIn[i].g * 255.0f + 0.01f)));
        EP.aLDRPixels[i].b = uint8_t(std::max<float>(0.0f, std::min<float>(255.0f, pIn[i].b * 255.0f + 0.01f)));
        EP.aLDRPixels[i].a = uint8_t(std::max<float>(0.0f, std::min<float>(255.0f, pIn[i].a * 255.0f + 0.01f)));
        switch (rot)
        {
        case 1: std::swap(EP.aLDRPixels[i].r, EP.aLDRPixels[i].a); break;
        case 2: std::swap(EP.aLDRPixels[i].g, EP.aLDRPixels[i].a); break;
        case 3: std::swap(EP.aLDRPixels[i].b, EP.aLDRPixels[i].a); break;
        default: break;
        }
    }
    EP.uMode = uint8_t(mode);
}

int main(int argc, char** argv)
{
    if (const char* fs = getenv("DXTEX_HOST_FLAT_SKIP")) dxtex::bc7::g_hostFlatSkipEveryMode = std::string(fs) != "kernel";
    const int ntiles = argc > 1 ? atoi(argv[1]) : 200;
    g_rng = argc > 2 ? uint32_t(atoi(argv[2])) : 12345u;
    // optional third argument: BC_FLAGS (0x80000 BC7_USE_3SUBSETS, 0x100000 BC7_QUICK)
    const uint32_t bcFlags = argc > 3 ? uint32_t(strtoul(argv[3], nullptr, 0)) : 0u;
    const bool use3 = (bcFlags & 0x80000u) != 0, quick = (bcFlags & 0x100000u) != 0;
    // optional fourth argument: a file of raw RGBA8 tiles (64 bytes each, row-major 4x4) to use instead of random tiles
    std::vector<uint8_t> tileFile;
    if (argc > 4)
    {
        FILE* f = fopen(argv[4], "rb");
        if (!f) { printf("cannot open %s\n", argv[4]); return 2; }
        tileFile.resize(size_t(ntiles) * 64);
        const size_t got = fread(tileFile.data(), 1, tileFile.size(), f);
        fclose(f);
        if (got != tileFile.size()) { printf("%s: short read\n", argv[4]); return 2; }
    }
    int nbad = 0;
    for (int t = 0; t < ntiles; ++t)
    {
        uint8_t px[64];
        const int spread = (int[]){ 0, 1, 3, 10, 40, 120 }[rnd() % 6];
        uint8_t base[4] = { uint8_t(rnd()), uint8_t(rnd()), uint8_t(rnd()), uint8_t(rnd()) };
        const bool opaque = rnd() & 1;
        for (int i = 0; i < 16; ++i)
            for (int c = 0; c < 4; ++c)
            {
                int v = base[c] + (spread ? int(rnd() % (2 * spread + 1)) - spread : 0);
                v = v < 0 ? 0 : v > 255 ? 255 : v;
                px[i * 4 + c] = (c == 3 && opaque) ? 255 : uint8_t(v);
            }
        if (!tileFile.empty()) memcpy(px, tileFile.data() + size_t(t) * 64, 64);
        HostBlock hb; make_block(hb, px);

        // reference
        alignas(16) HDRColorA pIn[16];
        memcpy(pIn, hb.f, sizeof(pIn));
        alignas(16) uint8_t refBlk[16];
        reinterpret_cast<D3DX_BC7*>(refBlk)->Encode(bcFlags, pIn);

        // ours
        int e3[64], e2[64]; uint32_t l3[16], l2[16];
        rough_lists(hb, e3, e2, l3, l2);
        Cand best; best.valid = false;
        Cand perMode[8]; for (auto& c : perMode) c.valid = false;
        if (!quick)
        {
            if (use3)
            {
                uint32_t m0[4], m2[16];
                rough_list3<3>(hb, 16, 4, m0);            // mode 0: 16 shapes, 3-bit indices, the best max(1, 16 / 4)
                rough_list3<2>(hb, 64, 16, m2);           // mode 2: 64 shapes, 2-bit indices, the best 16
                for (int i = 0; i < 4; ++i) better(perMode[0], refine3_one<0>(hb, m0[i], i));
                for (int i = 0; i < 16; ++i) better(perMode[2], refine3_one<2>(hb, m2[i], i));
            }
#if defined(DXTEX_TABLE_STATS)
            {
                // first pass without statistics: what the block ends up with (the optimistic table), and mode 6 / mode 1 alone (what is
                // on the table when mode 1 / mode 3 start in the kernels' order)
                g_statTable = g_statTablePrev = 0x7FFFFFFF;
                long sv[3][8]; memcpy(sv[0], g_tabWin, sizeof(g_tabWin)); memcpy(sv[1], g_tabWinOut, sizeof(g_tabWin)); memcpy(sv[2], g_tabWinOutPrev, sizeof(g_tabWin));
                Cand b6 = refine1_one<6, 0>(hb, 0), b1; b1.valid = false; Cand b3; b3.valid = false;
                for (int i = 0; i < 16; ++i) better(b1, refine2_one<1>(hb, l3[i], i));
                for (int i = 0; i < 16; ++i) better(b3, refine2_one<3>(hb, l2[i], i));
                memcpy(g_tabWin, sv[0], sizeof(g_tabWin)); memcpy(g_tabWinOut, sv[1], sizeof(g_tabWin)); memcpy(g_tabWinOutPrev, sv[2], sizeof(g_tabWin));
                const int fin = int(std::min(b6.err, std::min(b1.err, b3.err)));
                g_statTable = fin; g_statTablePrev = int(b6.err);
                for (int i = 0; i < 16; ++i) better(perMode[1], refine2_one<1>(hb, l3[i], i));
                g_statTablePrev = int(std::min(b6.err, b1.err));
                for (int i = 0; i < 16; ++i) better(perMode[3], refine2_one<3>(hb, l2[i], i));
                g_statTable = g_statTablePrev = 0x7FFFFFFF;
            }
#else
            for (int i = 0; i < 16; ++i) better(perMode[1], refine2_one<1>(hb, l3[i], i));
            for (int i = 0; i < 16; ++i) better(perMode[3], refine2_one<3>(hb, l2[i], i));
#endif
            for (uint32_t r = 0; r < 4; ++r) { better(perMode[4], refine1_one<4, 0>(hb, r)); better(perMode[4], refine1_one<4, 1>(hb, r)); }
            for (uint32_t r = 0; r < 4; ++r) better(perMode[5], refine1_one<5, 0>(hb, r));
        }
        better(perMode[6], refine1_one<6, 0>(hb, 0));
        if (!quick && hb.hasAlpha) for (int i = 0; i < 16; ++i) better(perMode[7], refine2_one<7>(hb, l2[i], i));
        for (int m = 0; m < 8; ++m) better(best, perMode[m]);

        uint64_t rlo, rhi; memcpy(&rlo, refBlk, 8); memcpy(&rhi, refBlk + 8, 8);
        if (rlo == best.lo && rhi == best.hi) continue;
        ++nbad;
        if (nbad > 3) continue;
        int refMode = 0; while (refMode < 8 && !((refBlk[0] >> refMode) & 1)) ++refMode;
        printf("tile %d MISMATCH: ref mode %d, ours mode %u (err %u) spread %d opaque %d\n", t, refMode, best.ord / 128, best.err, spread, int(opaque));

        // stage-by-stage comparison against the reference's own member functions
        D3DX_BC7 obj;
        for (int mode : { 1, 3, 7, 4, 5, 6 })
        {
            if (mode == 7 && !hb.hasAlpha) continue;
            const int nrot = (mode == 4 || mode == 5) ? 4 : 1, nim = (mode == 4) ? 2 : 1;
            const int nshapes = (mode == 1 || mode == 3 || mode == 7) ? 64 : 1;
            float bestRefErr = FLT_MAX; int bestRefIdx = -1;
            for (int rot = 0; rot < nrot; ++rot)
                for (int im = 0; im < nim; ++im)
                {
                    D3DX_BC7::EncodeParams EP(pIn);
                    ref_setup(EP, pIn, mode, rot);
                    float rough[64]; size_t shp[64];
                    for (int s = 0; s < nshapes; ++s) { rough[s] = D3DX_BC7::RoughMSE(&EP, s, im); shp[s] = s; }
                    if (nshapes == 64)
                    {
                        const int* mine = (mode == 1) ? e3 : e2;
                        for (int s = 0; s < 64; ++s)
                            if (float(mine[s]) != rough[s]) { printf("  mode %d shape %d: rough ref %.0f ours %d\n", mode, s, rough[s], mine[s]); break; }
                    }
                    const int items = nshapes == 64 ? 16 : 1;
                    for (int i = 0; i < items; ++i)
                        for (int j = i + 1; j < nshapes; ++j)
                            if (rough[i] > rough[j]) { std::swap(rough[i], rough[j]); std::swap(shp[i], shp[j]); }
                    for (int i = 0; i < items; ++i)
                    {
                        const float e = obj.Refine(&EP, shp[i], rot, im);
                        uint64_t lo, hi; memcpy(&lo, &obj, 8); memcpy(&hi, reinterpret_cast<uint8_t*>(&obj) + 8, 8);
                        Cand c;
                        if (mode == 1) c = refine2_one<1>(hb, uint32_t(shp[i]), i);
                        else if (mode == 3) c = refine2_one<3>(hb, uint32_t(shp[i]), i);
                        else if (mode == 7) c = refine2_one<7>(hb, uint32_t(shp[i]), i);
                        else if (mode == 4) c = im ? refine1_one<4, 1>(hb, rot) : refine1_one<4, 0>(hb, rot);
                        else if (mode == 5) c = refine1_one<5, 0>(hb, rot);
                        else c = refine1_one<6, 0>(hb, 0);
                        if (float(c.err) != e || c.lo != lo || c.hi != hi)
                            printf("  mode %d rot %d im %d shape %zu: Refine ref err %.0f blk %016llx%016llx | ours err %u blk %016llx%016llx\n",
                                   mode, rot, im, shp[i], e, (unsigned long long)hi, (unsigned long long)lo, c.err, (unsigned long long)c.hi, (unsigned long long)c.lo);
                        if (e < bestRefErr) { bestRefErr = e; bestRefIdx = i; }
                    }
                }
            printf("  mode %d: ref best err %.0f, ours %s %u\n", mode, bestRefErr, perMode[mode].valid ? "err" : "n/a", perMode[mode].valid ? perMode[mode].err : 0);
        }
    }
#if defined(DXTEX_COUNT_EVALS)
    for (int m = 0; m < 8; ++m)
        if (g_evalCount[m]) printf("mode %d: %.1f evals/tile, %.1f macro-ops/tile, %.2f texels/eval; exhaustive: %.1f bounds/tile, %.1f passed the filter (%.1f %%), %.1f drains/tile\n", m, double(g_evalCount[m]) / ntiles, double(g_macroCount[m]) / ntiles, double(g_evalTexels[m]) / g_evalCount[m],
                                    double(g_boundCount[m]) / ntiles, double(g_pendCount[m]) / ntiles, 100.0 * double(g_pendCount[m]) / double(g_boundCount[m] ? g_boundCount[m] : 1), double(g_drainCount[m]) / ntiles);
#endif
#if defined(DXTEX_COUNT_EVALS)
    for (int m = 0; m < 8; ++m) if (g_pfStepTotal[m][7]) printf("mode %d perturb: %.1f %% of the candidates below the first step are out of range (skipped by the reference)\n", m, 100.0 * g_pfStepPass[m][7] / g_pfStepTotal[m][7]);
#endif
#if defined(DXTEX_TABLE_STATS)
    for (int m = 0; m < 8; ++m) if (g_tabWin[m]) printf("mode %d: of the windows the interval test leaves, %.1f %% belong to candidates that cannot win against the block's final error (%.1f %% against what earlier modes have put on the table)\n", m, 100.0 * g_tabWinOut[m] / g_tabWin[m], 100.0 * g_tabWinOutPrev[m] / g_tabWin[m]);
#endif
#if defined(DXTEX_COUNT_PERTURB_FILTER)
    for (int m = 0; m < 8; ++m)
    {
        if (g_pfTotal[m]) { printf("mode %d perturb pass rate by step (1,2,4,...):", m); for (int si = 0; si < 8; ++si) if (g_pfStepTotal[m][si]) printf(" %.2f%%", 100.0 * g_pfStepPass[m][si] / g_pfStepTotal[m][si]); printf("\n"); }
        if (g_pfTotal[m]) printf("mode %d perturb: %.1f candidates/tile, %.1f %% pass the bound filter, %.1f %% improve\n", m, double(g_pfTotal[m]) / ntiles, 100.0 * g_pfPass[m] / g_pfTotal[m], 100.0 * g_pfImprove[m] / g_pfTotal[m]);
    }
#endif
#if defined(DXTEX_ROW_STATS)
    for (int m = 0; m < 8; ++m)
        if (g_rowCand[m])
            for (int H = 1; H <= 3; ++H)
                printf("mode %d: %.1f candidates per surviving window; strips of %d rows: %.2f tests per window, %.1f %% of the candidates in excluded strips\n", m,
                       double(g_rowCand[m]) / g_rowWin[m], H, double(g_rowTests[m][H]) / g_rowWin[m], 100.0 * g_rowOut[m][H] / g_rowCand[m]);
    for (int m = 0; m < 8; ++m)
        if (g_rowCand[m])
        {
            for (int H = 1; H <= 3; ++H) for (int L = 1; L <= 2; ++L)
                printf("mode %d peel H=%d L=%d: %.2f tests per window, %.1f %% of the candidates removed\n", m, H, L, double(g_peelTests[m][H][L]) / g_rowWin[m], 100.0 * g_peelOut[m][H][L] / g_rowCand[m]);
            printf("mode %d, strips of 2 rows by distance from the starting row (share of candidates : excluded):", m);
            for (int d = 0; d < 6; ++d) printf("  d%d %.1f%% : %.1f%%", d, 100.0 * g_rowDistN[m][d] / g_rowCand[m], g_rowDistN[m][d] ? 100.0 * g_rowDistOut[m][d] / g_rowDistN[m][d] : 0.0);
            printf("\n");
        }
#endif
    printf("%d of %d tiles differ\n", nbad, ntiles);
    return nbad ? 1 : 0;
}
