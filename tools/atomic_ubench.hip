// DEVELOPMENT TOOL: how fast can 8192 persistent wavefronts pull batches from ONE global counter (the search kernels' task queue,
// search_common.h: queue_take)? Prints ns per atomic for one shared counter, one counter per XCD-sized group, one per wavefront.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_ubench.hip -o /tmp/atomic_ubench && /tmp/atomic_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(64) take_kernel(uint32_t* heads, uint32_t nheads, int iters, uint32_t* sink, int work)
{
    uint32_t* head = heads + (blockIdx.x % nheads) * 64;       // 256 B apart
    uint32_t acc = threadIdx.x;
    for (int i = 0; i < iters; ++i)
    {
        uint32_t base = 0;
        if (threadIdx.x == 0) base = atomicAdd(head, 128u);
        base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
        acc += base;
        for (int k = 0; k < work; ++k) acc = acc * 1664525u + 1013904223u;      // dependent ALU chain between takes
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    uint32_t* heads; uint32_t* sink;
    hipMalloc(&heads, 8192 * 256); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int waves = 8192;
    for (int work : { 0, 1000, 10000 })
        for (uint32_t nheads : { 1u, 8u, 64u, 8192u })
            for (int iters : { 4, 32 })
            {
                hipMemset(heads, 0, 8192 * 256);
                take_kernel<<<waves, 64>>>(heads, nheads, iters, sink, work);
                hipDeviceSynchronize();
                hipMemset(heads, 0, 8192 * 256);
                hipEventRecord(e0);
                take_kernel<<<waves, 64>>>(heads, nheads, iters, sink, work);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                printf("work %5d  counters %4u  takes/wave %2d : %8.3f ms  = %7.1f ns per take (machine-wide)\n", work, nheads, iters, ms, ms * 1e6 / (double(waves) * iters));
            }
    return 0;
}
