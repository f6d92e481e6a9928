// TEST INFRASTRUCTURE ONLY (oracle). Stand-in for DirectXMath's <DirectXPackedVector.h> subset used by
// the reference's block codecs and (round 6) by every packed load / store DirectXTexConvert.cpp's scanline
// layer calls (see DirectXMath.h in this directory for provenance and caveats). Every leaf names the
// DirectXPackedVector.inl code path it restates; the conventions that decide bytes are:
//   * normalised loads multiply by a reciprocal CONSTANT (mulps by 1/255, 1/65535, 1/32767, 1/127, 1/1023, 1/511, 1/510, 1/3 - the
//     SSE2 paths carry the constant pre-divided by the field's power-of-two position, which commutes with rounding), signed ones
//     then maxps with -1;
//   * stores clamp with maxps(V, lo) then minps(., hi) (NaN -> lo), scale with mulps, and convert with cvtps2dq (round to nearest
//     even) - EXCEPT XMStoreUByteN4, XMStoreUDecN4, XMStoreUDec4, XMStoreUDecN4_XR, which convert with cvttps2dq (truncate; the
//     reference knows: DirectXTexConvert.cpp:198-199 adds 0.5/255 before XMStoreUByteN4), the 32-bit integer stores (cvttps2dq
//     with explicit overflow lanes), and the generic-bodied two-component 8-bit stores (XMStoreUByteN2: v*255+0.5 truncated;
//     XMStoreByteN2 / XMStoreByte2 / XMStoreUByte2: XMVectorRound).
#pragma once
#include "DirectXMath.h"

namespace DirectX
{
    namespace PackedVector
    {
        typedef uint16_t HALF;
        struct XMHALF4 { HALF x, y, z, w; };
        struct XMHALF2 { HALF x, y; };
        struct XMU565 { union { struct { uint16_t x : 5; uint16_t y : 6; uint16_t z : 5; }; uint16_t v; }; };
        struct XMU555 { union { struct { uint16_t x : 5; uint16_t y : 5; uint16_t z : 5; uint16_t w : 1; }; uint16_t v; }; };
        struct XMUNIBBLE4 { union { struct { uint16_t x : 4; uint16_t y : 4; uint16_t z : 4; uint16_t w : 4; }; uint16_t v; }; };
        struct XMUBYTE4 { union { struct { uint8_t x, y, z, w; }; uint32_t v; }; };
        struct XMUBYTEN4 { union { struct { uint8_t x, y, z, w; }; uint32_t v; }; };
        struct XMBYTE4 { union { struct { int8_t x, y, z, w; }; uint32_t v; }; };
        struct XMBYTEN4 { union { struct { int8_t x, y, z, w; }; uint32_t v; }; };
        struct XMUBYTE2 { union { struct { uint8_t x, y; }; uint16_t v; }; };
        struct XMUBYTEN2 { union { struct { uint8_t x, y; }; uint16_t v; }; };
        struct XMBYTE2 { union { struct { int8_t x, y; }; uint16_t v; }; };
        struct XMBYTEN2 { union { struct { int8_t x, y; }; uint16_t v; }; };
        struct XMUSHORT2 { union { struct { uint16_t x, y; }; uint32_t v; }; };
        struct XMUSHORTN2 { union { struct { uint16_t x, y; }; uint32_t v; }; };
        struct XMSHORT2 { union { struct { int16_t x, y; }; uint32_t v; }; };
        struct XMSHORTN2 { union { struct { int16_t x, y; }; uint32_t v; }; };
        struct XMUSHORT4 { union { struct { uint16_t x, y, z, w; }; uint64_t v; }; };
        struct XMUSHORTN4 { union { struct { uint16_t x, y, z, w; }; uint64_t v; }; };
        struct XMSHORT4 { union { struct { int16_t x, y, z, w; }; uint64_t v; }; };
        struct XMSHORTN4 { union { struct { int16_t x, y, z, w; }; uint64_t v; }; };
        struct XMUDECN4 { union { struct { uint32_t x : 10; uint32_t y : 10; uint32_t z : 10; uint32_t w : 2; }; uint32_t v; }; };
        struct XMUDEC4 { union { struct { uint32_t x : 10; uint32_t y : 10; uint32_t z : 10; uint32_t w : 2; }; uint32_t v; }; };
        struct XMXDECN4 { union { struct { int32_t x : 10; int32_t y : 10; int32_t z : 10; uint32_t w : 2; }; uint32_t v; }; };
        struct XMFLOAT3PK { union { struct { uint32_t xm : 6; uint32_t xe : 5; uint32_t ym : 6; uint32_t ye : 5; uint32_t zm : 5; uint32_t ze : 5; }; uint32_t v; }; };
        struct XMFLOAT3SE { union { struct { uint32_t xm : 9; uint32_t ym : 9; uint32_t zm : 9; uint32_t e : 5; }; uint32_t v; }; };

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
        // ---- round 6: the rest of what DirectXTexConvert.cpp calls -------------------------------------------------------------------
        // XMConvertHalfToFloatStream / XMConvertFloatToHalfStream: without F16C, a loop of the scalar conversions over the strides.
        inline float* XMConvertHalfToFloatStream(float* pOutputStream, size_t OutputStride, const HALF* pInputStream, size_t InputStride, size_t HalfCount) noexcept
        {
            auto pHalf = reinterpret_cast<const uint8_t*>(pInputStream);
            auto pFloat = reinterpret_cast<uint8_t*>(pOutputStream);
            for (size_t i = 0; i < HalfCount; ++i, pHalf += InputStride, pFloat += OutputStride)
            {
                HALF h; memcpy(&h, pHalf, 2);
                const float f = XMConvertHalfToFloat(h); memcpy(pFloat, &f, 4);
            }
            return pOutputStream;
        }
        inline HALF* XMConvertFloatToHalfStream(HALF* pOutputStream, size_t OutputStride, const float* pInputStream, size_t InputStride, size_t FloatCount) noexcept
        {
            auto pFloat = reinterpret_cast<const uint8_t*>(pInputStream);
            auto pHalf = reinterpret_cast<uint8_t*>(pOutputStream);
            for (size_t i = 0; i < FloatCount; ++i, pFloat += InputStride, pHalf += OutputStride)
            {
                float f; memcpy(&f, pFloat, 4);
                const HALF h = XMConvertFloatToHalf(f); memcpy(pHalf, &h, 2);
            }
            return pOutputStream;
        }
        inline XMVECTOR XMLoadHalf2(const XMHALF2* p) noexcept { return XMVECTOR{ { XMConvertHalfToFloat(p->x), XMConvertHalfToFloat(p->y), 0.f, 0.f } }; }
        inline void XMStoreHalf2(XMHALF2* p, FXMVECTOR V) noexcept { p->x = XMConvertFloatToHalf(V.f[0]); p->y = XMConvertFloatToHalf(V.f[1]); }

        namespace ShimPV
        {
            using namespace ShimSSE;
            // clamp the way every SSE2 store does: maxps(V, lo) then minps(., hi)
            inline float clamp_store(float v, float lo, float hi) noexcept { return minps(maxps(v, lo), hi); }
            // normalised load: cvtdq2ps, mulps by the reciprocal constant, (signed) maxps with -1
            inline float unorm(uint32_t i, float recip) noexcept { return cvtdq(int32_t(i)) * recip; }
            inline float snorm(int32_t i, float recip) noexcept { return maxps(cvtdq(i) * recip, -1.0f); }
            // normalised store with cvtps2dq: clamp, mulps by the scale, round to nearest even
            inline int32_t store_round(float v, float lo, float hi, float scale) noexcept { return cvtps(clamp_store(v, lo, hi) * scale); }
            // normalised store with cvttps2dq (XMStoreUByteN4 and the 10:10:10:2 family)
            inline int32_t store_trunc(float v, float lo, float hi, float scale) noexcept { return cvttps(clamp_store(v, lo, hi) * scale); }
        }

        // -- 16-bit x 4 / x 2 (XMLoadShortN4 ... XMStoreUShort2: SSE2 paths) --
        inline XMVECTOR XMLoadShortN4(const XMSHORTN4* p) noexcept
        {
            const float k = 1.0f / 32767.0f;
            return XMVECTOR{ { ShimPV::snorm(p->x, k), ShimPV::snorm(p->y, k), ShimPV::snorm(p->z, k), ShimPV::snorm(p->w, k) } };
        }
        inline XMVECTOR XMLoadShortN2(const XMSHORTN2* p) noexcept { const float k = 1.0f / 32767.0f; return XMVECTOR{ { ShimPV::snorm(p->x, k), ShimPV::snorm(p->y, k), 0.f, 0.f } }; }
        inline XMVECTOR XMLoadShort4(const XMSHORT4* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), float(p->z), float(p->w) } }; }
        inline XMVECTOR XMLoadShort2(const XMSHORT2* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), 0.f, 0.f } }; }
        inline XMVECTOR XMLoadUShortN4(const XMUSHORTN4* p) noexcept
        {
            const float k = 1.0f / 65535.0f;
            return XMVECTOR{ { ShimPV::unorm(p->x, k), ShimPV::unorm(p->y, k), ShimPV::unorm(p->z, k), ShimPV::unorm(p->w, k) } };
        }
        inline XMVECTOR XMLoadUShortN2(const XMUSHORTN2* p) noexcept { const float k = 1.0f / 65535.0f; return XMVECTOR{ { ShimPV::unorm(p->x, k), ShimPV::unorm(p->y, k), 0.f, 0.f } }; }
        inline XMVECTOR XMLoadUShort4(const XMUSHORT4* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), float(p->z), float(p->w) } }; }
        inline XMVECTOR XMLoadUShort2(const XMUSHORT2* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), 0.f, 0.f } }; }
        // stores: clamp, (scale,) cvtps2dq; packssdw / extraction cannot saturate after the clamp
        inline void XMStoreShortN4(XMSHORTN4* p, FXMVECTOR V) noexcept
        {
            p->x = int16_t(ShimPV::store_round(V.f[0], -1.f, 1.f, 32767.f)); p->y = int16_t(ShimPV::store_round(V.f[1], -1.f, 1.f, 32767.f));
            p->z = int16_t(ShimPV::store_round(V.f[2], -1.f, 1.f, 32767.f)); p->w = int16_t(ShimPV::store_round(V.f[3], -1.f, 1.f, 32767.f));
        }
        inline void XMStoreShortN2(XMSHORTN2* p, FXMVECTOR V) noexcept
        {
            p->x = int16_t(ShimPV::store_round(V.f[0], -1.f, 1.f, 32767.f)); p->y = int16_t(ShimPV::store_round(V.f[1], -1.f, 1.f, 32767.f));
        }
        inline void XMStoreShort4(XMSHORT4* p, FXMVECTOR V) noexcept
        {
            p->x = int16_t(ShimPV::store_round(V.f[0], -32767.f, 32767.f, 1.f)); p->y = int16_t(ShimPV::store_round(V.f[1], -32767.f, 32767.f, 1.f));
            p->z = int16_t(ShimPV::store_round(V.f[2], -32767.f, 32767.f, 1.f)); p->w = int16_t(ShimPV::store_round(V.f[3], -32767.f, 32767.f, 1.f));
        }
        inline void XMStoreShort2(XMSHORT2* p, FXMVECTOR V) noexcept
        {
            p->x = int16_t(ShimPV::store_round(V.f[0], -32767.f, 32767.f, 1.f)); p->y = int16_t(ShimPV::store_round(V.f[1], -32767.f, 32767.f, 1.f));
        }
        inline void XMStoreUShortN4(XMUSHORTN4* p, FXMVECTOR V) noexcept
        {
            p->x = uint16_t(ShimPV::store_round(V.f[0], 0.f, 1.f, 65535.f)); p->y = uint16_t(ShimPV::store_round(V.f[1], 0.f, 1.f, 65535.f));
            p->z = uint16_t(ShimPV::store_round(V.f[2], 0.f, 1.f, 65535.f)); p->w = uint16_t(ShimPV::store_round(V.f[3], 0.f, 1.f, 65535.f));
        }
        inline void XMStoreUShortN2(XMUSHORTN2* p, FXMVECTOR V) noexcept
        {
            p->x = uint16_t(ShimPV::store_round(V.f[0], 0.f, 1.f, 65535.f)); p->y = uint16_t(ShimPV::store_round(V.f[1], 0.f, 1.f, 65535.f));
        }
        inline void XMStoreUShort4(XMUSHORT4* p, FXMVECTOR V) noexcept
        {
            p->x = uint16_t(ShimPV::store_round(V.f[0], 0.f, 65535.f, 1.f)); p->y = uint16_t(ShimPV::store_round(V.f[1], 0.f, 65535.f, 1.f));
            p->z = uint16_t(ShimPV::store_round(V.f[2], 0.f, 65535.f, 1.f)); p->w = uint16_t(ShimPV::store_round(V.f[3], 0.f, 65535.f, 1.f));
        }
        inline void XMStoreUShort2(XMUSHORT2* p, FXMVECTOR V) noexcept
        {
            p->x = uint16_t(ShimPV::store_round(V.f[0], 0.f, 65535.f, 1.f)); p->y = uint16_t(ShimPV::store_round(V.f[1], 0.f, 65535.f, 1.f));
        }

        // -- 8-bit x 4 (SSE2 paths) --
        inline XMVECTOR XMLoadByteN4(const XMBYTEN4* p) noexcept
        {
            const float k = 1.0f / 127.0f;
            return XMVECTOR{ { ShimPV::snorm(p->x, k), ShimPV::snorm(p->y, k), ShimPV::snorm(p->z, k), ShimPV::snorm(p->w, k) } };
        }
        inline XMVECTOR XMLoadByte4(const XMBYTE4* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), float(p->z), float(p->w) } }; }
        // XMStoreUByteN4: maxps / minps to [0, 1], mulps by 255, cvtTps2dq - truncation (see the header; the reference compensates)
        inline void XMStoreUByteN4(XMUBYTEN4* p, FXMVECTOR V) noexcept
        {
            p->x = uint8_t(ShimPV::store_trunc(V.f[0], 0.f, 1.f, 255.f)); p->y = uint8_t(ShimPV::store_trunc(V.f[1], 0.f, 1.f, 255.f));
            p->z = uint8_t(ShimPV::store_trunc(V.f[2], 0.f, 1.f, 255.f)); p->w = uint8_t(ShimPV::store_trunc(V.f[3], 0.f, 1.f, 255.f));
        }
        inline void XMStoreUByte4(XMUBYTE4* p, FXMVECTOR V) noexcept
        {
            p->x = uint8_t(ShimPV::store_round(V.f[0], 0.f, 255.f, 1.f)); p->y = uint8_t(ShimPV::store_round(V.f[1], 0.f, 255.f, 1.f));
            p->z = uint8_t(ShimPV::store_round(V.f[2], 0.f, 255.f, 1.f)); p->w = uint8_t(ShimPV::store_round(V.f[3], 0.f, 255.f, 1.f));
        }
        inline void XMStoreByteN4(XMBYTEN4* p, FXMVECTOR V) noexcept
        {
            p->x = int8_t(ShimPV::store_round(V.f[0], -1.f, 1.f, 127.f)); p->y = int8_t(ShimPV::store_round(V.f[1], -1.f, 1.f, 127.f));
            p->z = int8_t(ShimPV::store_round(V.f[2], -1.f, 1.f, 127.f)); p->w = int8_t(ShimPV::store_round(V.f[3], -1.f, 1.f, 127.f));
        }
        inline void XMStoreByte4(XMBYTE4* p, FXMVECTOR V) noexcept
        {
            p->x = int8_t(ShimPV::store_round(V.f[0], -127.f, 127.f, 1.f)); p->y = int8_t(ShimPV::store_round(V.f[1], -127.f, 127.f, 1.f));
            p->z = int8_t(ShimPV::store_round(V.f[2], -127.f, 127.f, 1.f)); p->w = int8_t(ShimPV::store_round(V.f[3], -127.f, 127.f, 1.f));
        }

        // -- 8-bit x 2: one generic body each (no intrinsic paths), written with the vector ops of DirectXMath.h --
        inline XMVECTOR XMLoadByteN2(const XMBYTEN2* p) noexcept
        {
            return XMVectorSet((p->x == -128) ? -1.f : (float(p->x) * (1.0f / 127.0f)), (p->y == -128) ? -1.f : (float(p->y) * (1.0f / 127.0f)), 0.f, 0.f);
        }
        inline XMVECTOR XMLoadByte2(const XMBYTE2* p) noexcept { return XMVectorSet(float(p->x), float(p->y), 0.f, 0.f); }
        inline XMVECTOR XMLoadUByteN2(const XMUBYTEN2* p) noexcept { return XMVectorSet(float(p->x) * (1.0f / 255.0f), float(p->y) * (1.0f / 255.0f), 0.f, 0.f); }
        inline XMVECTOR XMLoadUByte2(const XMUBYTE2* p) noexcept { return XMVectorSet(float(p->x), float(p->y), 0.f, 0.f); }
        inline void XMStoreByteN2(XMBYTEN2* p, FXMVECTOR V) noexcept
        {
            XMVECTOR N = XMVectorClamp(V, g_XMNegativeOne, g_XMOne);
            N = XMVectorRound(XMVectorMultiply(N, XMVectorReplicate(127.0f)));
            p->x = int8_t(ShimSSE::cvttps(N.f[0])); p->y = int8_t(ShimSSE::cvttps(N.f[1]));
        }
        inline void XMStoreByte2(XMBYTE2* p, FXMVECTOR V) noexcept
        {
            const XMVECTOR N = XMVectorRound(XMVectorClamp(V, XMVectorReplicate(-127.0f), XMVectorReplicate(127.0f)));
            p->x = int8_t(ShimSSE::cvttps(N.f[0])); p->y = int8_t(ShimSSE::cvttps(N.f[1]));
        }
        // XMStoreUByteN2: XMVectorSaturate, XMVectorMultiplyAdd(N, 255, 0.5) (unfused), XMVectorTruncate
        inline void XMStoreUByteN2(XMUBYTEN2* p, FXMVECTOR V) noexcept
        {
            XMVECTOR N = XMVectorSaturate(V);
            N = XMVectorTruncate(XMVectorMultiplyAdd(N, XMVectorReplicate(255.0f), g_XMOneHalf));
            p->x = uint8_t(ShimSSE::cvttps(N.f[0])); p->y = uint8_t(ShimSSE::cvttps(N.f[1]));
        }
        inline void XMStoreUByte2(XMUBYTE2* p, FXMVECTOR V) noexcept
        {
            const XMVECTOR N = XMVectorRound(XMVectorClamp(V, XMVectorZero(), XMVectorReplicate(255.0f)));
            p->x = uint8_t(ShimSSE::cvttps(N.f[0])); p->y = uint8_t(ShimSSE::cvttps(N.f[1]));
        }

        // -- 5:6:5, 5:5:5:1, 4:4:4:4 (SSE2 paths; un-normalised: the reference scales by 31 / 63 / 15 itself) --
        inline XMVECTOR XMLoadU555(const XMU555* p) noexcept
        {
            return XMVECTOR{ { float(p->v & 0x1F), float((p->v >> 5) & 0x1F), float((p->v >> 10) & 0x1F), float((p->v >> 15) & 1) } };
        }
        inline XMVECTOR XMLoadUNibble4(const XMUNIBBLE4* p) noexcept
        {
            return XMVECTOR{ { float(p->v & 0xF), float((p->v >> 4) & 0xF), float((p->v >> 8) & 0xF), float((p->v >> 12) & 0xF) } };
        }
        inline void XMStoreU565(XMU565* p, FXMVECTOR V) noexcept
        {
            const int x = ShimPV::store_round(V.f[0], 0.f, 31.f, 1.f), y = ShimPV::store_round(V.f[1], 0.f, 63.f, 1.f), z = ShimPV::store_round(V.f[2], 0.f, 31.f, 1.f);
            p->v = uint16_t(((z & 0x1F) << 11) | ((y & 0x3F) << 5) | (x & 0x1F));
        }
        inline void XMStoreU555(XMU555* p, FXMVECTOR V) noexcept
        {
            const int x = ShimPV::store_round(V.f[0], 0.f, 31.f, 1.f), y = ShimPV::store_round(V.f[1], 0.f, 31.f, 1.f), z = ShimPV::store_round(V.f[2], 0.f, 31.f, 1.f),
                      w = ShimPV::store_round(V.f[3], 0.f, 1.f, 1.f);
            p->v = uint16_t((w ? 0x8000 : 0) | ((z & 0x1F) << 10) | ((y & 0x1F) << 5) | (x & 0x1F));
        }
        inline void XMStoreUNibble4(XMUNIBBLE4* p, FXMVECTOR V) noexcept
        {
            const int x = ShimPV::store_round(V.f[0], 0.f, 15.f, 1.f), y = ShimPV::store_round(V.f[1], 0.f, 15.f, 1.f), z = ShimPV::store_round(V.f[2], 0.f, 15.f, 1.f),
                      w = ShimPV::store_round(V.f[3], 0.f, 15.f, 1.f);
            p->v = uint16_t(((w & 0xF) << 12) | ((z & 0xF) << 8) | ((y & 0xF) << 4) | (x & 0xF));
        }

        // -- 10:10:10:2 (SSE2 paths: masked fields converted in place, reciprocal constants carry the field position) --
        inline XMVECTOR XMLoadUDecN4(const XMUDECN4* p) noexcept
        {
            const float k = 1.0f / 1023.0f, k2 = 1.0f / 3.0f;
            return XMVECTOR{ { ShimPV::unorm(p->v & 0x3FF, k), ShimPV::unorm((p->v >> 10) & 0x3FF, k), ShimPV::unorm((p->v >> 20) & 0x3FF, k), ShimPV::unorm(p->v >> 30, k2) } };
        }
        inline XMVECTOR XMLoadUDec4(const XMUDEC4* p) noexcept
        {
            return XMVECTOR{ { float(p->v & 0x3FF), float((p->v >> 10) & 0x3FF), float((p->v >> 20) & 0x3FF), float(p->v >> 30) } };
        }
        // XMLoadUDecN4_XR: the bias 0x180 is subtracted as an INTEGER (psubd) before the conversion, then mulps by 1/510 (w: 1/3)
        inline XMVECTOR XMLoadUDecN4_XR(const XMUDECN4* p) noexcept
        {
            const float k = 1.0f / 510.0f, k2 = 1.0f / 3.0f;
            return XMVECTOR{ { ShimSSE::cvtdq(int32_t(p->v & 0x3FF) - 0x180) * k, ShimSSE::cvtdq(int32_t((p->v >> 10) & 0x3FF) - 0x180) * k,
                               ShimSSE::cvtdq(int32_t((p->v >> 20) & 0x3FF) - 0x180) * k, ShimPV::unorm(p->v >> 30, k2) } };
        }
        inline XMVECTOR XMLoadXDecN4(const XMXDECN4* p) noexcept
        {
            const float k = 1.0f / 511.0f, k2 = 1.0f / 3.0f;
            auto sx = [](uint32_t f) { return int32_t(f << 22) >> 22; };
            return XMVECTOR{ { ShimPV::snorm(sx(p->v & 0x3FF), k), ShimPV::snorm(sx((p->v >> 10) & 0x3FF), k), ShimPV::snorm(sx((p->v >> 20) & 0x3FF), k),
                               ShimPV::unorm(p->v >> 30, k2) } };
        }
        // XMStoreUDecN4 / XMStoreUDec4 / XMStoreUDecN4_XR: scale constants carry the field position (y and w at half position so the
        // product stays below 2^31), cvtTps2dq, mask - the mask drops the fraction, i.e. truncation of v * 1023 (w: v * 3)
        inline void XMStoreUDecN4(XMUDECN4* p, FXMVECTOR V) noexcept
        {
            const uint32_t x = uint32_t(ShimPV::store_trunc(V.f[0], 0.f, 1.f, 1023.f)), y = uint32_t(ShimPV::store_trunc(V.f[1], 0.f, 1.f, 1023.f)),
                           z = uint32_t(ShimPV::store_trunc(V.f[2], 0.f, 1.f, 1023.f)), w = uint32_t(ShimPV::store_trunc(V.f[3], 0.f, 1.f, 3.f));
            p->v = (x & 0x3FF) | ((y & 0x3FF) << 10) | ((z & 0x3FF) << 20) | ((w & 3) << 30);
        }
        inline void XMStoreUDec4(XMUDEC4* p, FXMVECTOR V) noexcept
        {
            const uint32_t x = uint32_t(ShimPV::store_trunc(V.f[0], 0.f, 1023.f, 1.f)), y = uint32_t(ShimPV::store_trunc(V.f[1], 0.f, 1023.f, 1.f)),
                           z = uint32_t(ShimPV::store_trunc(V.f[2], 0.f, 1023.f, 1.f)), w = uint32_t(ShimPV::store_trunc(V.f[3], 0.f, 3.f, 1.f));
            p->v = (x & 0x3FF) | ((y & 0x3FF) << 10) | ((z & 0x3FF) << 20) | ((w & 3) << 30);
        }
        // XMStoreUDecN4_XR: V * 510 + 384 (w: V * 3), unfused on SSE2, clamp to [0, 1023] ([0, 3]), truncate
        inline void XMStoreUDecN4_XR(XMUDECN4* p, FXMVECTOR V) noexcept
        {
            auto q = [](float v, float scale, float bias, float hi) { return uint32_t(ShimSSE::cvttps(ShimPV::clamp_store(v * scale + bias, 0.f, hi))); };
            p->v = (q(V.f[0], 510.f, 384.f, 1023.f) & 0x3FF) | ((q(V.f[1], 510.f, 384.f, 1023.f) & 0x3FF) << 10) | ((q(V.f[2], 510.f, 384.f, 1023.f) & 0x3FF) << 20) |
                   ((q(V.f[3], 3.f, 0.f, 3.f) & 3) << 30);
        }
        // XMStoreXDecN4 (only the Xbox-only R10G10B10_SNORM_A2_UNORM reaches it; not a format the MI355X path claims): clamp to
        // [-1, 1] ([0, 1] for w), scale by 511 (3), cvtps2dq
        inline void XMStoreXDecN4(XMXDECN4* p, FXMVECTOR V) noexcept
        {
            const uint32_t x = uint32_t(ShimPV::store_round(V.f[0], -1.f, 1.f, 511.f)), y = uint32_t(ShimPV::store_round(V.f[1], -1.f, 1.f, 511.f)),
                           z = uint32_t(ShimPV::store_round(V.f[2], -1.f, 1.f, 511.f)), w = uint32_t(ShimPV::store_round(V.f[3], 0.f, 1.f, 3.f));
            p->v = (x & 0x3FF) | ((y & 0x3FF) << 10) | ((z & 0x3FF) << 20) | ((w & 3) << 30);
        }

        // -- R11G11B10_FLOAT / R9G9B9E5 (scalar bodies on every path) --
        namespace ShimPV
        {
            // one unsigned small float (5-bit exponent, mbits of mantissa) -> binary32; denormals are normalised, exponent 31 -> Inf / NaN
            inline float load_small(uint32_t exponent, uint32_t mantissa, int mbits) noexcept
            {
                uint32_t result;
                if (exponent == 0x1f) result = 0x7f800000u | (mantissa << (23 - mbits));
                else
                {
                    if (exponent != 0) { }
                    else if (mantissa != 0)
                    {
                        exponent = 1;
                        do { exponent--; mantissa <<= 1; } while ((mantissa & (1u << mbits)) == 0);
                        mantissa &= (1u << mbits) - 1u;
                    }
                    else exponent = uint32_t(-112);
                    result = ((exponent + 112) << 23) | (mantissa << (23 - mbits));
                }
                return from_bits(result);
            }
            // binary32 -> unsigned small float: negative and tiny -> 0, -Inf -> 0, too large -> the largest finite value, NaN -> all ones,
            // otherwise round to nearest even on the dropped mantissa bits (denormals by shifting the significand first)
            inline uint32_t store_small(float value, int mbits) noexcept
            {
                const uint32_t iv = bits(value);
                const bool sign = (iv & 0x80000000u) != 0;
                uint32_t I = iv & 0x7FFFFFFFu;
                const uint32_t expMask = 0x1Fu << mbits, allOnes = expMask | ((1u << mbits) - 1u);
                const uint32_t drop = uint32_t(23 - mbits);
                if ((I & 0x7F800000u) == 0x7F800000u) return ((I & 0x7FFFFFu) != 0) ? allOnes : (sign ? 0u : expMask);
                if (sign || I < (mbits == 6 ? 0x35800000u : 0x36000000u)) return 0;
                if (I > (mbits == 6 ? 0x477E0000u : 0x477C0000u)) return expMask - 1u;
                if (I < 0x38800000u) { const uint32_t shift = 113u - (I >> 23u); I = (0x800000u | (I & 0x7FFFFFu)) >> shift; }
                else I += 0xC8000000u;
                return ((I + ((1u << (drop - 1)) - 1u) + ((I >> drop) & 1u)) >> drop) & allOnes;
            }
            // Internal::round_to_nearest of DirectXMath (ties to even), as used by XMStoreFloat3SE
            inline float round_to_nearest(float x) noexcept
            {
                float i = floorf(x);
                x -= i;
                if (x < 0.5f) return i;
                if (x > 0.5f) return i + 1.f;
                float ip; (void)modff(i / 2.f, &ip);
                return ((2.f * ip) == i) ? i : i + 1.f;
            }
        }
        inline XMVECTOR XMLoadFloat3PK(const XMFLOAT3PK* p) noexcept
        {
            return XMVECTOR{ { ShimPV::load_small(p->xe, p->xm, 6), ShimPV::load_small(p->ye, p->ym, 6), ShimPV::load_small(p->ze, p->zm, 5), 0.f } };
        }
        inline void XMStoreFloat3PK(XMFLOAT3PK* p, FXMVECTOR V) noexcept
        {
            p->v = (ShimPV::store_small(V.f[0], 6) & 0x7FF) | ((ShimPV::store_small(V.f[1], 6) & 0x7FF) << 11) | ((ShimPV::store_small(V.f[2], 5) & 0x3FF) << 22);
        }
        inline XMVECTOR XMLoadFloat3SE(const XMFLOAT3SE* p) noexcept
        {
            const float scale = ShimSSE::from_bits(0x33800000u + (uint32_t(p->e) << 23));
            return XMVECTOR{ { scale * float(p->xm), scale * float(p->ym), scale * float(p->zm), 1.0f } };
        }
        // XMStoreFloat3SE (DirectXMath >= 3.10): clamp to [0, 65408], the shared exponent from the largest channel rounded up at nine
        // significand bits, mantissas by round-to-nearest-even of v * 2^(9 - e)
        inline void XMStoreFloat3SE(XMFLOAT3SE* p, FXMVECTOR V) noexcept
        {
            constexpr float maxf9 = float(0x1FF << 7);
            constexpr float minf9 = float(1.f / (1 << 16));
            const float x = (V.f[0] >= 0.f) ? ((V.f[0] > maxf9) ? maxf9 : V.f[0]) : 0.f;
            const float y = (V.f[1] >= 0.f) ? ((V.f[1] > maxf9) ? maxf9 : V.f[1]) : 0.f;
            const float z = (V.f[2] >= 0.f) ? ((V.f[2] > maxf9) ? maxf9 : V.f[2]) : 0.f;
            const float max_xy = (x > y) ? x : y;
            const float max_xyz = (max_xy > z) ? max_xy : z;
            const float maxColor = (max_xyz > minf9) ? max_xyz : minf9;
            const uint32_t fi = ShimSSE::bits(maxColor) + 0x00004000u;
            const uint32_t exp = fi >> 23;
            p->e = exp - 0x6f;
            const float ScaleR = ShimSSE::from_bits(0x83000000u - (exp << 23));
            p->xm = uint32_t(ShimPV::round_to_nearest(x * ScaleR));
            p->ym = uint32_t(ShimPV::round_to_nearest(y * ScaleR));
            p->zm = uint32_t(ShimPV::round_to_nearest(z * ScaleR));
        }
    }
}
