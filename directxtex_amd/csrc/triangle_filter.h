// Host side of the triangle filter. The reference describes the filter as SCATTER lists - per source texel, the destination texels it
// feeds and with what weight (CreateTriangleFilter, filters.h:249-419) - and walks them source by source. The kernels want the transpose:
// per destination texel, the (source, weight) pairs in the order the reference would have accumulated them (ascending source, then the
// entry's position in that source's list). This file computes the pairs with the reference's fp32 expressions, operation for operation
// (compile with -ffp-contract=off; the weights must be bit-identical, tests/cpp/triangle_check.cpp compares them with the reference's
// own lists on 330 (source, dest, wrap) triples), and files them straight into per-destination buckets.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dxtex
{
struct TriEntry { uint32_t src; uint32_t weightBits; };     // weight as the raw bits of the fp32 value

namespace tri_detail
{
// Consecutive cells of one source texel that land on the same destination texel are summed before they are filed (the reference merges
// them the same way, and the sum's rounding depends on it): one running (destination, sum) pair, flushed when the destination changes.
struct Pending
{
    size_t dst = 0;
    float sum = 0.f;
    void flush(std::vector<std::vector<TriEntry>>& buckets, size_t srcTexel, size_t ndst)
    {
        constexpr float kNegligible = 0.00001f;                 // TF_EPSILON
        if (sum > kNegligible && dst < ndst)
        {
            TriEntry e; e.src = uint32_t(srcTexel);
            std::memcpy(&e.weightBits, &sum, 4);
            buckets[dst].push_back(e);
        }
        sum = 0.f;
    }
};

// One of the two halves of a source texel's tent: the unit interval that starts at `centre` (= texel index - 0.5 for the rising half,
// + 0.5 for the falling one), mapped to destination space and cut into destination cells.
inline void spread_half(size_t srcTexel, bool falling, size_t nsrc, size_t ndst, bool wrap, float ratio, float halfInvRatio,
                        Pending& acc, std::vector<std::vector<TriEntry>>& buckets)
{
    const float centre = float(srcTexel + (falling ? 1 : 0)) - 0.5f;
    float lo = centre * ratio;
    float hi = lo + ratio;
    if (!wrap)
    {
        if (lo < 0.f) lo = 0.f;
        if (hi > float(ndst)) hi = float(ndst);
    }
    const ptrdiff_t n = ptrdiff_t(ndst);
    for (ptrdiff_t cell = ptrdiff_t(floorf(lo)); float(cell) < hi; ++cell)
    {
        const size_t target = size_t(cell < 0 ? cell + n : (cell >= n ? cell - n : cell));        // wrap addressing: one period either way
        if (target != acc.dst)
        {
            acc.flush(buckets, srcTexel, ndst);
            acc.dst = target;
        }
        float c0 = float(cell);
        float c1 = c0 + 1.f;
        if (c0 < lo) c0 = lo;
        if (c1 > hi) c1 = hi;
        // height of the tent's rising edge at the middle of the cell; clamped addressing pins it at the image's two edges
        float rise;
        if (!wrap && centre < 0.f) rise = 1.f;
        else if (!wrap && ((centre + 1.f) >= float(nsrc))) rise = 0.f;
        else rise = (c0 + c1) * halfInvRatio - centre;
        acc.sum += (c1 - c0) * (falling ? (1.f - rise) : rise);
    }
}
} // namespace tri_detail

// ofs[d] .. ofs[d + 1]: destination texel d's entries in `ent`
inline void build_triangle_axis(size_t source, size_t dest, bool wrap, std::vector<uint32_t>& ofs, std::vector<TriEntry>& ent)
{
    std::vector<std::vector<TriEntry>> buckets(dest);
    const float ratio = float(dest) / float(source);
    const float halfInvRatio = 0.5f / ratio;
    tri_detail::Pending acc;
    for (size_t s = 0; s < source; ++s)
    {
        tri_detail::spread_half(s, false, source, dest, wrap, ratio, halfInvRatio, acc, buckets);
        tri_detail::spread_half(s, true, source, dest, wrap, ratio, halfInvRatio, acc, buckets);
        acc.flush(buckets, s, dest);
    }
    ofs.assign(dest + 1, 0);
    ent.clear();
    for (size_t d = 0; d < dest; ++d)
    {
        ofs[d] = uint32_t(ent.size());
        ent.insert(ent.end(), buckets[d].begin(), buckets[d].end());
    }
    ofs[dest] = uint32_t(ent.size());
}
} // namespace dxtex
