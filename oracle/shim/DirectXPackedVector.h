// TEST INFRASTRUCTURE ONLY (oracle). Stand-in for DirectXMath's <DirectXPackedVector.h> subset used by
// the reference's block codecs (see DirectXMath.h in this directory for provenance and caveats).
#pragma once
#include "DirectXMath.h"

namespace DirectX
{
    namespace PackedVector
    {
        typedef uint16_t HALF;
        struct XMHALF4 { HALF x, y, z, w; };
        struct XMU565 { uint16_t v; };
        struct XMUBYTE4 { union { struct { uint8_t x, y, z, w; }; uint32_t v; }; };
        struct XMUBYTEN4 { union { struct { uint8_t x, y, z, w; }; uint32_t v; }; };

        // binary16 -> binary32: exact.
        inline float XMConvertHalfToFloat(HALF h) noexcept
        {
            const uint32_t sign = (uint32_t(h) & 0x8000u) << 16;
            uint32_t exp = (h >> 10) & 0x1Fu;
            uint32_t man = h & 0x3FFu;
            uint32_t bits;
            if (exp == 0x1F) bits = sign | 0x7F800000u | (man << 13);
            else if (exp != 0) bits = sign | ((exp + 112u) << 23) | (man << 13);
            else if (man != 0)
            {
                exp = 113;
                while (!(man & 0x400u)) { man <<= 1; --exp; }
                bits = sign | (exp << 23) | ((man & 0x3FFu) << 13);
            }
            else bits = sign;
            float f; memcpy(&f, &bits, 4); return f;
        }

        // binary32 -> binary16, round-to-nearest-even, overflow -> infinity, NaN preserved (the F16C /
        // current software convention). Inputs used by the tests are finite and <= 65504.
        inline HALF XMConvertFloatToHalf(float f) noexcept
        {
            uint32_t x; memcpy(&x, &f, 4);
            const uint32_t sign = (x >> 16) & 0x8000u;
            x &= 0x7FFFFFFFu;
            uint32_t r;
            if (x >= 0x7F800000u) r = (x > 0x7F800000u) ? (0x7E00u | ((x >> 13) & 0x3FFu)) : 0x7C00u;
            else if (x >= 0x47800000u) r = 0x7C00u;
            else if (x < 0x38800000u)
            {
                if (x < 0x33000000u) r = 0;     // below half the smallest subnormal: rounds to zero
                else
                {
                    const uint32_t e = x >> 23;
                    const uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
                    const uint32_t shift = 126u - e;             // 14..24
                    const uint32_t q = m >> shift;
                    const uint32_t rem = m & ((1u << shift) - 1u);
                    const uint32_t half = 1u << (shift - 1u);
                    r = q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
                }
            }
            else
            {
                const uint32_t y = x - 0x38000000u;
                r = (y + 0x0FFFu + ((y >> 13) & 1u)) >> 13;
            }
            return HALF(sign | r);
        }

        inline XMVECTOR XMLoadHalf4(const XMHALF4* p) noexcept
        {
            return XMVECTOR{ { XMConvertHalfToFloat(p->x), XMConvertHalfToFloat(p->y), XMConvertHalfToFloat(p->z), XMConvertHalfToFloat(p->w) } };
        }
        inline void XMStoreHalf4(XMHALF4* p, FXMVECTOR V) noexcept
        {
            p->x = XMConvertFloatToHalf(V.f[0]); p->y = XMConvertFloatToHalf(V.f[1]);
            p->z = XMConvertFloatToHalf(V.f[2]); p->w = XMConvertFloatToHalf(V.f[3]);
        }
        inline XMVECTOR XMLoadU565(const XMU565* p) noexcept
        {
            return XMVECTOR{ { float(p->v & 0x1F), float((p->v >> 5) & 0x3F), float((p->v >> 11) & 0x1F), 0.0f } };
        }
        inline XMVECTOR XMLoadUByte4(const XMUBYTE4* p) noexcept
        {
            return XMVECTOR{ { float(p->x), float(p->y), float(p->z), float(p->w) } };
        }
        inline XMVECTOR XMLoadUByteN4(const XMUBYTEN4* p) noexcept
        {
            const float s = 1.0f / 255.0f;
            return XMVECTOR{ { float(p->x) * s, float(p->y) * s, float(p->z) * s, float(p->w) * s } };
        }
    }
}
