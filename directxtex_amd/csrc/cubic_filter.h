// cubic_filter.h - the reference's CUBIC_INTERPOLATE (filters.h:192-207) for one channel, shared by the kernels of scanline.hip and by the
// host check that pins the folded form to the plain one (tests/cpp/cubic_check.cpp, run by tests/test_bounds_cpu.py).
// Compile with -ffp-contract=off (csrc/Makefile): the plain form must round every product on its own.
#pragma once
#if defined(__HIPCC__)
#define DXTEX_CUBIC_FN __host__ __device__ __forceinline__
#else
#define DXTEX_CUBIC_FN inline
#endif

namespace dxtex
{
// CUBIC_INTERPOLATE for one channel, operation for operation
DXTEX_CUBIC_FN float cubic1(float dx, float p0, float p1, float p2, float p3)
{
    const float a0 = p1;
    const float d0 = p0 - a0, d2 = p2 - a0, d3 = p3 - a0;
    float a1 = d2 - (1.0f / 3.0f) * d0;
    a1 = a1 - (1.0f / 6.0f) * d3;
    const float a2 = (1.0f / 2.0f) * d0 + (1.0f / 2.0f) * d2;
    float a3 = (1.0f / 6.0f) * d3 - (1.0f / 6.0f) * d0;
    a3 = a3 - (1.0f / 2.0f) * d2;
    const float dx2 = dx * dx;
    const float dx3 = dx2 * dx;
    return ((a0 + a1 * dx) + a2 * dx2) + a3 * dx3;
}


// CUBIC_INTERPOLATE (filters.h:192-207) at dx = 0.5, with every operation whose result does not depend on how it is issued folded: dx^2 = 0.25
// and dx^3 = 0.125 are exact, a multiplication by a power of two is exact, so a2 = 0.5 d0 + 0.5 d2 is exactly 0.5 (d0 + d2), and
// "x + (power of two) * y" rounds once whether the product is formed first or inside an FMA. The products with 1/3 and 1/6 keep their own
// rounding (a multiplication, then the subtraction). 14 operations instead of 21, the same bits as cubic1(0.5f, ...) for every input.
DXTEX_CUBIC_FN float cubic_half1(float p0, float p1, float p2, float p3)
{
    const float d0 = p0 - p1, d2 = p2 - p1, d3 = p3 - p1;
    const float sixthD3 = (1.0f / 6.0f) * d3;                    // the reference forms this product twice
    float a1 = d2 - (1.0f / 3.0f) * d0;
    a1 = a1 - sixthD3;
    const float s02 = d0 + d2;                                   // a2 = 0.5 * s02, a2 * dx^2 = 0.125 * s02
    float a3 = sixthD3 - (1.0f / 6.0f) * d0;
    a3 = __builtin_fmaf(-0.5f, d2, a3);                          // a3 - 0.5 * d2
    float r = __builtin_fmaf(0.5f, a1, p1);                      // a0 + a1 * dx
    r = __builtin_fmaf(0.125f, s02, r);                          // + a2 * dx^2
    return __builtin_fmaf(0.125f, a3, r);                        // + a3 * dx^3
}

} // namespace dxtex
