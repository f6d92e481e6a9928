// Pieces shared by the BC6H and BC7 search pipelines (bc6h_encode.hip, bc7_encode.hip): wavefront-level LDS
// synchronisation, the partial selection sort both encoders' Encode() loops perform on the rough errors, the counting
// sort of tasks by subset size, and the work queue of the persistent search kernels. Everything lives in an anonymous
// namespace: each translation unit gets its own copy of the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dxtex
{
namespace
{
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

// counters: [1..16] histogram by subset size, [18..33] scatter cursors, [34] number of live tasks.
// Both passes run kBinGroups workgroups over contiguous slices of the task list, histogram in LDS, and touch
// the 16 global counters once per workgroup.
constexpr int kBinGroups = 2048;

__global__ void __launch_bounds__(256) bc7_bin_count_kernel(const uint32_t* tinfo, uint32_t ntasks, uint32_t* counters)
{
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

__global__ void __launch_bounds__(256) bc7_bin_scatter_kernel(const uint32_t* tinfo, uint32_t ntasks, uint32_t* counters, uint2* order)
{
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

constexpr uint32_t kQueueBatch = 128;   // indices a wavefront reserves per atomic (same-address atomics serialise in L2)

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
