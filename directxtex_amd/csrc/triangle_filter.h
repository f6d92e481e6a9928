// Host side of the triangle filter: the reference builds, per source texel, the list of destination texels it
// contributes to (CreateTriangleFilter, filters.h:249-419). This computes the same lists with the same fp32
// arithmetic (compile with -ffp-contract=off) and inverts them into per-destination gather lists that keep the
// reference's accumulation order: ascending source index, then the entry's position in its source list.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dxtex
{
struct TriEntry { uint32_t src; uint32_t weightBits; };     // weight as the raw bits of the fp32 value

inline void build_triangle_axis(size_t source, size_t dest, bool wrap, std::vector<uint32_t>& ofs, std::vector<TriEntry>& ent)
{
    constexpr float TF_EPSILON = 0.00001f;
    std::vector<std::vector<TriEntry>> to(dest);
    const float scale = float(dest) / float(source);
    const float scaleInv = 0.5f / scale;
    size_t accumU = 0;
    float accumWeight = 0.f;
    auto emit = [&](size_t srcIndex)
    {
        if (accumWeight > TF_EPSILON && accumU < dest)
        {
            TriEntry e; e.src = uint32_t(srcIndex);
            std::memcpy(&e.weightBits, &accumWeight, 4);
            to[accumU].push_back(e);
        }
    };
    for (size_t u = 0; u < source; ++u)
    {
        for (size_t j = 0; j < 2; ++j)
        {
            const float src = float(u + j) - 0.5f;
            float destMin = src * scale;
            float destMax = destMin + scale;
            if (!wrap)
            {
                if (destMin < 0.f) destMin = 0.f;
                if (destMax > float(dest)) destMax = float(dest);
            }
            for (auto k = static_cast<ptrdiff_t>(floorf(destMin)); float(k) < destMax; ++k)
            {
                float d0 = float(k);
                float d1 = d0 + 1.f;
                size_t u0;
                if (k < 0) u0 = size_t(k + ptrdiff_t(dest));
                else if (k >= ptrdiff_t(dest)) u0 = size_t(k - ptrdiff_t(dest));
                else u0 = size_t(k);
                if (u0 != accumU)
                {
                    emit(u);
                    accumWeight = 0.f;
                    accumU = u0;
                }
                if (d0 < destMin) d0 = destMin;
                if (d1 > destMax) d1 = destMax;
                float weight;
                if (!wrap && src < 0.f) weight = 1.f;
                else if (!wrap && ((src + 1.f) >= float(source))) weight = 0.f;
                else weight = (d0 + d1) * scaleInv - src;
                accumWeight += (d1 - d0) * (j ? (1.f - weight) : weight);
            }
        }
        emit(u);
        accumWeight = 0.f;
    }
    ofs.assign(dest + 1, 0);
    ent.clear();
    for (size_t d = 0; d < dest; ++d)
    {
        ofs[d] = uint32_t(ent.size());
        ent.insert(ent.end(), to[d].begin(), to[d].end());
    }
    ofs[dest] = uint32_t(ent.size());
}
} // namespace dxtex
