// Pieces shared by the BC6H and BC7 search pipelines (bc6h_encode.hip, bc7_encode.hip): wavefront-level LDS
// synchronisation, the partial selection sort both encoders' Encode() loops perform on the rough errors, the counting
// sort of tasks by subset size, and the work queue of the persistent search kernels. Everything lives in an anonymous
// namespace: each translation unit gets its own copy of the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include "dxtex_kernels.h"

namespace dxtex
{
namespace
{
// ---- passes and segments ---------------------------------------------------------------------------------------------
// A pass covers up to a fixed number of blocks taken from one or more images (an array / mip chain goes through the
// per-mode pipeline as ONE block list, so small images do not each pay the pipeline's latency floor and tails). A segment
// is the run of one image's blocks inside a pass; only the kernels that touch pixels or the payload look segments up,
// everything between works on pass-local block numbers.
struct BcSeg
{
    SrcView src;
    uint8_t* dst;
    uint64_t dstRowPitch;
    uint32_t nbw;            // blocks per row of the image
    uint32_t nb0;            // first block of the image in this segment
    uint32_t l0;             // pass-local number of that block
    uint32_t pad;
};

struct SegTable
{
    BcSeg inl[2];            // nseg <= 2 (a single image, the usual case): the segments travel in the kernel arguments
    const BcSeg* segs;       // otherwise: the pass's segments in device memory, ascending l0
    uint32_t nseg;
};

__device__ __forceinline__ const BcSeg& seg_of(const SegTable& t, uint32_t local)
{
    if (t.nseg <= 2) return (t.nseg == 2 && t.inl[1].l0 <= local) ? t.inl[1] : t.inl[0];
    uint32_t lo = 0, hi = t.nseg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (t.segs[mid].l0 <= local) lo = mid; else hi = mid; }
    return t.segs[lo];
}

struct BcPass { uint32_t seg0, nseg, nblocks; };

// Cuts the concatenated block list of `count` images into passes of at most `perPass` blocks and per-image segments.
inline uint64_t build_passes(const BcImage* images, size_t count, uint64_t maxPerPass, std::vector<BcSeg>& segs, std::vector<BcPass>& passes, uint64_t* perPassOut)
{
    uint64_t total = 0;
    for (size_t i = 0; i < count; ++i) total += uint64_t((images[i].src.width + 3) / 4) * ((images[i].src.height + 3) / 4);
    const uint64_t perPass = total < maxPerPass ? total : maxPerPass;
    *perPassOut = perPass;
    if (!total) return 0;
    BcPass cur = { 0, 0, 0 };
    for (size_t i = 0; i < count; ++i)
    {
        const uint32_t nbw = (images[i].src.width + 3) / 4, nbh = (images[i].src.height + 3) / 4;
        uint64_t left = uint64_t(nbw) * nbh, at = 0;
        while (left)
        {
            const uint64_t take = std::min<uint64_t>(left, perPass - cur.nblocks);
            BcSeg sg; sg.src = images[i].src; sg.dst = images[i].dst; sg.dstRowPitch = images[i].dstRowPitch;
            sg.nbw = nbw; sg.nb0 = uint32_t(at); sg.l0 = cur.nblocks; sg.pad = 0;
            segs.push_back(sg);
            ++cur.nseg; cur.nblocks += uint32_t(take); at += take; left -= take;
            if (cur.nblocks == perPass) { passes.push_back(cur); cur = { uint32_t(segs.size()), 0, 0 }; }
        }
    }
    if (cur.nblocks) passes.push_back(cur);
    return total;
}

// Uploads the segment table when some pass needs it (more than two segments). Stream-ordered: the previous call's kernels may
// still be reading the old table. The host copy sits in a pinned buffer owned by the calling thread; an event marks when the
// last upload from it has been consumed, so the buffer is never rewritten (or reallocated) under a copy in flight.
inline hipError_t upload_segments(BcSeg* dSegs, std::vector<BcSeg>& segs, const std::vector<BcPass>& passes, hipStream_t stream)
{
    bool needTable = false;
    for (const BcPass& pass : passes) needTable |= pass.nseg > 2;
    if (!needTable) return hipSuccess;
    struct Staging { void* host = nullptr; size_t capacity = 0; hipEvent_t consumed = nullptr; bool pending = false; int device = -1; };
    static thread_local Staging st;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (st.pending) { e = hipEventSynchronize(st.consumed); if (e != hipSuccess) return e; st.pending = false; }
    const size_t bytes = segs.size() * sizeof(BcSeg);
    if (st.device != device || st.capacity < bytes)
    {
        if (st.host) (void)hipHostFree(st.host);
        if (st.consumed) (void)hipEventDestroy(st.consumed);
        st = Staging();
        e = hipHostMalloc(&st.host, std::max<size_t>(bytes, 4096), hipHostMallocDefault);
        if (e != hipSuccess) return e;
        e = hipEventCreateWithFlags(&st.consumed, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        st.capacity = std::max<size_t>(bytes, 4096); st.device = device;
    }
    std::memcpy(st.host, segs.data(), bytes);
    e = hipMemcpyAsync(dSegs, st.host, bytes, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipEventRecord(st.consumed, stream);
    st.pending = (e == hipSuccess);
    return e;
}

inline void set_pass(SegTable& t, const BcSeg* dSegs, const std::vector<BcSeg>& segs, const BcPass& pass)
{
    t.segs = dSegs + pass.seg0; t.nseg = pass.nseg;
    t.inl[0] = segs[pass.seg0]; t.inl[1] = segs[pass.seg0 + (pass.nseg > 1 ? 1 : 0)];
}

inline size_t seg_table_bytes(uint64_t nblocks, uint64_t maxPerPass, size_t nimages)
{
    const uint64_t perPass = nblocks < maxPerPass ? nblocks : maxPerPass;
    const uint64_t passes = perPass ? (nblocks + perPass - 1) / perPass : 1;
    return ((nimages + passes + 1) * sizeof(BcSeg) + 255) & ~size_t(255);
}

// Lanes of one wavefront exchange data through LDS: DS operations of a wave execute in order, so only the
// compiler has to be told not to move accesses across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Reproduces "bubble up the first uItems items" (:2855-2865): position i ends up with the first minimum
// of positions i.., and every strict prefix-minimum record along the way shifts to the next record's place.
// Inclusive minimum over lanes 0 ... l of a wavefront, on the data-parallel-primitive path of gfx9 (row_shr within the rows of sixteen
// lanes, then row_bcast:15 / row_bcast:31 across them): six v_min_i32 with a DPP operand instead of six ds_bpermute + compare + select -
// the rough passes run 8 - 48 selection passes per block and were waiting on the LDS crossbar. A lane without a source keeps its value.
__device__ __forceinline__ int wave_prefix_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xF, 0xF, false));      // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xF, 0xF, false));      // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xF, 0xF, false));      // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xF, 0xF, false));      // row_shr:8
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));      // row_bcast:15 into rows 1 and 3
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));      // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ void selection_pass(int& e, uint32_t& s, int lane, int i)
{
    const int v = (lane >= i) ? e : 0x7FFFFFFF;
    const int incl = wave_prefix_min(v);
    int excl = __builtin_amdgcn_update_dpp(0x7FFFFFFF, incl, 0x138, 0xF, 0xF, false);      // wave_shr:1; lane 0 keeps +inf
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

// counters: [1..16] histogram by subset size, [18..33] scatter cursors, [34] number of live tasks.
// Both passes run kBinGroups workgroups over contiguous slices of the task list, histogram in LDS, and touch
// the 16 global counters once per workgroup.
constexpr int kBinGroups = 2048;

// gate / gateMin: a launch whose mode (or phase of a mode) owns no block of the pass - known on the device only - is skipped when
// *gate < gateMin (the task list was not written then; the zeroed counters leave counters[34] = 0 live tasks). nullptr = always run.
__global__ void __launch_bounds__(256) bc7_bin_count_kernel(const uint32_t* tinfo, uint32_t ntasks, uint32_t* counters, const uint32_t* gate = nullptr, uint32_t gateMin = 0)
{
    if (gate && *gate < gateMin) return;
    __shared__ uint32_t hist[17];
    if (threadIdx.x < 17) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t per = (ntasks + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = min(ntasks, t0 + per);
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += 256)
    {
        const uint32_t np = tinfo[t] >> 24;
        if (np) atomicAdd(&hist[np], 1u);
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= 16 && hist[threadIdx.x]) atomicAdd(&counters[threadIdx.x], hist[threadIdx.x]);
}

__global__ void bc7_bin_scan_kernel(uint32_t* counters)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t run = 0;
    for (int b = 16; b >= 1; --b) { counters[17 + b] = run; run += counters[b]; }    // largest subsets first
    counters[34] = run;
}

__global__ void __launch_bounds__(256) bc7_bin_scatter_kernel(const uint32_t* tinfo, uint32_t ntasks, uint32_t* counters, uint2* order, const uint32_t* gate = nullptr, uint32_t gateMin = 0)
{
    if (gate && *gate < gateMin) return;
    __shared__ uint32_t hist[17], cursor[17];
    if (threadIdx.x < 17) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t per = (ntasks + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = min(ntasks, t0 + per);
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += 256)
    {
        const uint32_t np = tinfo[t] >> 24;
        if (np) atomicAdd(&hist[np], 1u);
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x <= 16)
        cursor[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&counters[17 + threadIdx.x], hist[threadIdx.x]) : 0u;
    __syncthreads();
    for (uint32_t t = t0 + threadIdx.x; t < t1; t += 256)
    {
        const uint32_t ti = tinfo[t], np = ti >> 24;
        if (np) order[atomicAdd(&cursor[np], 1u)] = make_uint2(t, ti);
    }
}

// Work distribution of the search kernels: a fixed number of persistent wavefronts pull task indices from
// one global counter (counters[kQueueBase + loop]); a lane that finishes its task takes the next one, so lanes
// stay busy although searches differ in length by an order of magnitude, and there is no per-chunk tail.
constexpr int kQueueBase = 36;          // counters[36..39]: queue heads of the (up to 4) loops of a mode
constexpr int kSearchWaves = 8192;      // 256 CUs x 4 SIMDs x 8 wave slots

constexpr uint32_t kQueueBatch = 128;   // indices a wavefront reserves per atomic (same-address atomics serialise in L2: 11 ns per take machine-wide,
                                        // tools/atomic_ubench.hip; round 5 measured batches that shrink with the remaining list: the extra takes cost more than
                                        // the better balance saves, 139.4 -> 147.0 ms per 4096^2 image)

struct WaveQueue
{
    uint32_t lo, hi;     // reserved, not yet handed out (wave-uniform)
    bool drained;        // the global counter has passed the end of the list
};

// Hands the idle lanes (mask `idle`) indices into the sorted task list; returns 0xFFFFFFFF for a lane that gets
// none this time. One global atomic per kQueueBatch tasks.
__device__ __forceinline__ uint32_t queue_take(WaveQueue& q, uint32_t* head, uint32_t live, unsigned long long idle, int lane)
{
    if (q.lo >= q.hi && !q.drained)
    {
        // short lists (small images): one task per lane, so that the serial length of a task is paid once, not twice
        const uint32_t batch = (live > uint32_t(kSearchWaves) * 64u) ? kQueueBatch : 64u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, batch);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        q.lo = base;
        q.hi = min(base + batch, live);
        if (base >= live) { q.drained = true; q.hi = q.lo; }
    }
    const uint32_t k = uint32_t(__popcll(idle & ((1ull << lane) - 1ull)));
    const uint32_t avail = q.hi - q.lo;
    const uint32_t mine = (((idle >> lane) & 1ull) && k < avail) ? (q.lo + k) : 0xFFFFFFFFu;
    q.lo += min(uint32_t(__popcll(idle)), avail);
    return mine;
}

} // namespace
} // namespace dxtex
