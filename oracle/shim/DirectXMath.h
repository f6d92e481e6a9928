// TEST INFRASTRUCTURE ONLY (oracle). Scalar stand-in for the subset of Microsoft DirectXMath that the
// reference's block codecs and filters use. DirectXMath is an un-vendored, un-pinned dependency of
// the reference (CMakeLists.txt:384, build/vcpkg.json:4) and is absent from this image, so the
// conventions below restate its published x86-64 SSE2 code path (the path a Linux g++ -msse2 build of
// the reference takes): per-component IEEE fp32 mul/add/sub without FMA, Dot3 = (x*x' + y*y') + z*z',
// Dot4 = (x*x' + z*z') + (y*y' + w*w'), XMVectorLerp = V0 + (V1 - V0) * t, XMVectorMultiplyAdd = a*b + c
// unfused, half conversion round-to-nearest-even. PARITY UNPINNED at this boundary: no reference test
// or golden vector in /root/reference fixes these conventions.
// Round 6: grown to every DirectXMath symbol DirectXTexConvert.cpp uses, so that the reference's scanline layer
// (LoadScanline / StoreScanline / ConvertScanline / Convert, DirectXTexConvert.cpp) is compiled in place against it
// (oracle/ref_convert.cpp) instead of being restated: the control flow above these leaves is the reference's own.
// Each leaf says which DirectXMath code path it restates (the _XM_SSE_INTRINSICS_ one unless the function has a
// single generic body): maxps / minps operand order (NaN behaviour), cvtps2dq (round to nearest even) versus
// cvttps2dq (truncate), mulps by a reciprocal constant versus divps.
// Written from scratch for this repo; nothing here is copied from DirectXMath.
#pragma once
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>
#include <cmath>
using std::isnan;   // BC4BC5.cpp:162 uses unqualified isnan

#define DIRECTX_MATH_VERSION 320
#define XM_CALLCONV
#define XM_ALIGNED_DATA(x) alignas(x)
#define XM_ALIGNED_STRUCT(x) struct alignas(x)
#define XMGLOBALCONST static const
#define XM_CONSTEXPR constexpr
#define XM_CONST constexpr

namespace DirectX
{
    struct alignas(16) XMVECTOR { float f[4]; };
    typedef const XMVECTOR FXMVECTOR;
    typedef const XMVECTOR GXMVECTOR;
    typedef const XMVECTOR HXMVECTOR;
    typedef const XMVECTOR& CXMVECTOR;

    struct alignas(16) XMVECTORF32
    {
        union { float f[4]; XMVECTOR v; };
        operator XMVECTOR() const noexcept { return v; }
        operator const float*() const noexcept { return f; }
    };
    struct alignas(16) XMVECTORU32
    {
        union { uint32_t u[4]; XMVECTOR v; };
        operator XMVECTOR() const noexcept { return v; }
    };
    struct alignas(16) XMVECTORI32
    {
        union { int32_t i[4]; XMVECTOR v; };
        operator XMVECTOR() const noexcept { return v; }
    };

    struct XMFLOAT2 { float x, y; };
    struct XMFLOAT3 { float x, y, z; };
    struct alignas(16) XMFLOAT3A : public XMFLOAT3 {};
    struct XMFLOAT4 { float x, y, z, w; };
    struct alignas(16) XMFLOAT4A : public XMFLOAT4 {};
    struct XMINT2 { int32_t x, y; };
    struct XMINT3 { int32_t x, y, z; };
    struct XMINT4 { int32_t x, y, z, w; };
    struct XMUINT2 { uint32_t x, y; };
    struct XMUINT3 { uint32_t x, y, z; };
    struct XMUINT4 { uint32_t x, y, z, w; };

    constexpr uint32_t XM_SELECT_0 = 0x00000000u;
    constexpr uint32_t XM_SELECT_1 = 0xFFFFFFFFu;

    XMGLOBALCONST XMVECTORF32 g_XMIdentityR3 = { { 0.0f, 0.0f, 0.0f, 1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMZero = { { 0.0f, 0.0f, 0.0f, 0.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMOne = { { 1.0f, 1.0f, 1.0f, 1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMNegativeOne = { { -1.0f, -1.0f, -1.0f, -1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMOneHalf = { { 0.5f, 0.5f, 0.5f, 0.5f } };
    XMGLOBALCONST XMVECTORF32 g_XMTwo = { { 2.0f, 2.0f, 2.0f, 2.0f } };
    XMGLOBALCONST XMVECTORU32 g_XMSelect1110 = { { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMSelect1100 = { { 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMSelect1000 = { { 0xFFFFFFFFu, 0u, 0u, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMMaskX = { { 0xFFFFFFFFu, 0u, 0u, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMMaskY = { { 0u, 0xFFFFFFFFu, 0u, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMMaskZ = { { 0u, 0u, 0xFFFFFFFFu, 0u } };
    XMGLOBALCONST XMVECTORU32 g_XMMaskW = { { 0u, 0u, 0u, 0xFFFFFFFFu } };

    inline XMVECTOR XMVectorZero() noexcept { return XMVECTOR{ {0, 0, 0, 0} }; }
    inline XMVECTOR XMVectorSet(float x, float y, float z, float w) noexcept { return XMVECTOR{ {x, y, z, w} }; }
    inline XMVECTOR XMVectorReplicate(float v) noexcept { return XMVECTOR{ {v, v, v, v} }; }
    inline float XMVectorGetX(FXMVECTOR V) noexcept { return V.f[0]; }
    inline float XMVectorGetY(FXMVECTOR V) noexcept { return V.f[1]; }
    inline float XMVectorGetZ(FXMVECTOR V) noexcept { return V.f[2]; }
    inline float XMVectorGetW(FXMVECTOR V) noexcept { return V.f[3]; }
    inline XMVECTOR XMVectorSetW(FXMVECTOR V, float w) noexcept { XMVECTOR r = V; r.f[3] = w; return r; }
    inline XMVECTOR XMVectorSplatW(FXMVECTOR V) noexcept { return XMVectorReplicate(V.f[3]); }
    inline XMVECTOR XMVectorSplatX(FXMVECTOR V) noexcept { return XMVectorReplicate(V.f[0]); }
    inline XMVECTOR XMVectorSplatY(FXMVECTOR V) noexcept { return XMVectorReplicate(V.f[1]); }
    inline XMVECTOR XMVectorSplatZ(FXMVECTOR V) noexcept { return XMVectorReplicate(V.f[2]); }

    // The x86 instructions the SSE2 paths are made of, one lane each.
    namespace ShimSSE
    {
        inline float maxps(float a, float b) noexcept { return (a > b) ? a : b; }      // NaN in either operand -> b
        inline float minps(float a, float b) noexcept { return (a < b) ? a : b; }      // NaN in either operand -> b
        inline uint32_t bits(float f) noexcept { uint32_t u; memcpy(&u, &f, 4); return u; }
        inline float from_bits(uint32_t u) noexcept { float f; memcpy(&f, &u, 4); return f; }
        // cvtps2dq under the default MXCSR: round to nearest even; NaN / out of range -> the integer indefinite 0x80000000
        inline int32_t cvtps(float f) noexcept
        {
            if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
            const float r = nearbyintf(f);                                                // FE_TONEAREST is never changed by the reference
            if (!(r < 2147483648.0f)) return INT32_MIN;
            return int32_t(r);
        }
        // cvttps2dq: truncate; NaN / out of range -> 0x80000000
        inline int32_t cvttps(float f) noexcept
        {
            if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
            return int32_t(f);
        }
        // cvtdq2ps
        inline float cvtdq(int32_t i) noexcept { return float(i); }
    }

    inline XMVECTOR XMLoadFloat4(const XMFLOAT4* p) noexcept { return XMVECTOR{ {p->x, p->y, p->z, p->w} }; }
    inline XMVECTOR XMLoadFloat4A(const XMFLOAT4A* p) noexcept { return XMVECTOR{ {p->x, p->y, p->z, p->w} }; }
    inline void XMStoreFloat4(XMFLOAT4* p, FXMVECTOR V) noexcept { p->x = V.f[0]; p->y = V.f[1]; p->z = V.f[2]; p->w = V.f[3]; }
    inline void XMStoreFloat4A(XMFLOAT4A* p, FXMVECTOR V) noexcept { p->x = V.f[0]; p->y = V.f[1]; p->z = V.f[2]; p->w = V.f[3]; }
    inline XMVECTOR XMLoadSInt4(const XMINT4* p) noexcept
    {
        return XMVECTOR{ { float(p->x), float(p->y), float(p->z), float(p->w) } };
    }

#define DXM_BINOP(NAME, OP) \
    inline XMVECTOR NAME(FXMVECTOR A, FXMVECTOR B) noexcept \
    { return XMVECTOR{ { A.f[0] OP B.f[0], A.f[1] OP B.f[1], A.f[2] OP B.f[2], A.f[3] OP B.f[3] } }; }
    DXM_BINOP(XMVectorAdd, +)
    DXM_BINOP(XMVectorSubtract, -)
    DXM_BINOP(XMVectorMultiply, *)
    DXM_BINOP(XMVectorDivide, /)
#undef DXM_BINOP

    inline XMVECTOR XMVectorScale(FXMVECTOR V, float s) noexcept { return XMVectorMultiply(V, XMVectorReplicate(s)); }
    inline XMVECTOR XMVectorMultiplyAdd(FXMVECTOR A, FXMVECTOR B, FXMVECTOR C) noexcept
    {
        // SSE2 path without FMA3: mulps then addps.
        return XMVectorAdd(XMVectorMultiply(A, B), C);
    }
    inline XMVECTOR XMVectorLerp(FXMVECTOR V0, FXMVECTOR V1, float t) noexcept
    {
        return XMVectorMultiplyAdd(XMVectorSubtract(V1, V0), XMVectorReplicate(t), V0);
    }
    inline XMVECTOR XMVectorSaturate(FXMVECTOR V) noexcept
    {
        XMVECTOR r;
        for (int i = 0; i < 4; ++i) { float v = V.f[i]; v = (v > 0.0f) ? v : 0.0f; r.f[i] = (v < 1.0f) ? v : 1.0f; }
        return r;
    }
    inline XMVECTOR XMVectorSelect(FXMVECTOR V1, FXMVECTOR V2, FXMVECTOR Control) noexcept
    {
        XMVECTOR r;
        for (int i = 0; i < 4; ++i)
        {
            uint32_t a, b, c; memcpy(&a, &V1.f[i], 4); memcpy(&b, &V2.f[i], 4); memcpy(&c, &Control.f[i], 4);
            uint32_t o = (a & ~c) | (b & c); memcpy(&r.f[i], &o, 4);
        }
        return r;
    }
    template<uint32_t E0, uint32_t E1, uint32_t E2, uint32_t E3>
    inline XMVECTOR XMVectorSwizzle(FXMVECTOR V) noexcept { return XMVECTOR{ { V.f[E0], V.f[E1], V.f[E2], V.f[E3] } }; }

    inline XMVECTOR XMVector3Dot(FXMVECTOR A, FXMVECTOR B) noexcept
    {
        const float d = (A.f[0] * B.f[0] + A.f[1] * B.f[1]) + A.f[2] * B.f[2];
        return XMVectorReplicate(d);
    }
    inline XMVECTOR XMVector4Dot(FXMVECTOR A, FXMVECTOR B) noexcept
    {
        const float d = (A.f[0] * B.f[0] + A.f[2] * B.f[2]) + (A.f[1] * B.f[1] + A.f[3] * B.f[3]);
        return XMVectorReplicate(d);
    }
    // XMVectorSum: x+y+z+w in every lane; SSE2 shape (x+y) + (z+w)
    inline XMVECTOR XMVectorSum(FXMVECTOR V) noexcept { return XMVectorReplicate((V.f[0] + V.f[1]) + (V.f[2] + V.f[3])); }
    inline XMVECTOR XMVectorMergeXY(FXMVECTOR A, FXMVECTOR B) noexcept { return XMVECTOR{ { A.f[0], B.f[0], A.f[1], B.f[1] } }; }
    template<uint32_t P0, uint32_t P1, uint32_t P2, uint32_t P3>
    inline XMVECTOR XMVectorPermute(FXMVECTOR A, FXMVECTOR B) noexcept
    {
        const float* s[2] = { A.f, B.f };
        return XMVECTOR{ { s[P0 >> 2][P0 & 3], s[P1 >> 2][P1 & 3], s[P2 >> 2][P2 & 3], s[P3 >> 2][P3 & 3] } };
    }
    inline XMVECTOR XMVectorPow(FXMVECTOR A, FXMVECTOR B) noexcept
    {
        return XMVECTOR{ { powf(A.f[0], B.f[0]), powf(A.f[1], B.f[1]), powf(A.f[2], B.f[2]), powf(A.f[3], B.f[3]) } };
    }
    inline bool XMVector4Less(FXMVECTOR A, FXMVECTOR B) noexcept
    {
        return A.f[0] < B.f[0] && A.f[1] < B.f[1] && A.f[2] < B.f[2] && A.f[3] < B.f[3];
    }

    // ---- round 6: what DirectXTexConvert.cpp needs on top ------------------------------------------------------------------------
#define DXM_LANES(EXPR) XMVECTOR r; for (int i = 0; i < 4; ++i) { r.f[i] = (EXPR); } return r

    // XMVectorMin / XMVectorMax: minps(V1, V2) / maxps(V1, V2)
    inline XMVECTOR XMVectorMin(FXMVECTOR A, FXMVECTOR B) noexcept { DXM_LANES(ShimSSE::minps(A.f[i], B.f[i])); }
    inline XMVECTOR XMVectorMax(FXMVECTOR A, FXMVECTOR B) noexcept { DXM_LANES(ShimSSE::maxps(A.f[i], B.f[i])); }
    // XMVectorClamp: maxps(Min, V) then minps(Max, .) - V is the SECOND operand, so a NaN in V survives (XMVectorSaturate has V
    // first: NaN -> 0)
    inline XMVECTOR XMVectorClamp(FXMVECTOR V, FXMVECTOR Min, FXMVECTOR Max) noexcept
    {
        DXM_LANES(ShimSSE::minps(Max.f[i], ShimSSE::maxps(Min.f[i], V.f[i])));
    }
    // XMVectorNegate: subps(0, V)
    inline XMVECTOR XMVectorNegate(FXMVECTOR V) noexcept { DXM_LANES(0.0f - V.f[i]); }
    // XMVectorRound, SSE2 (no SSE4.1 roundps): add and subtract 2^23 carrying V's sign where |V| <= 2^23, V itself elsewhere (NaN
    // included) - round to nearest even; the SSE4.1 path (roundps, nearest) gives the same values.
    inline XMVECTOR XMVectorRound(FXMVECTOR V) noexcept
    {
        XMVECTOR r;
        for (int i = 0; i < 4; ++i)
        {
            const uint32_t u = ShimSSE::bits(V.f[i]);
            const float magic = ShimSSE::from_bits(0x4B000000u | (u & 0x80000000u));
            volatile float t = V.f[i] + magic;                                            // volatile: the two roundings must both happen
            const float r1 = t - magic;
            const float a = ShimSSE::from_bits(u & 0x7FFFFFFFu);
            r.f[i] = (a <= 8388608.0f) ? r1 : V.f[i];
        }
        return r;
    }
    // XMVectorTruncate (used by the generic packed stores below): cvttps2dq / cvtdq2ps where |V| < 2^23, V elsewhere
    inline XMVECTOR XMVectorTruncate(FXMVECTOR V) noexcept
    {
        XMVECTOR r;
        for (int i = 0; i < 4; ++i)
        {
            const float a = ShimSSE::from_bits(ShimSSE::bits(V.f[i]) & 0x7FFFFFFFu);
            r.f[i] = (a < 8388608.0f) ? ShimSSE::cvtdq(ShimSSE::cvttps(V.f[i])) : V.f[i];
        }
        return r;
    }
    inline XMVECTOR XMVectorGreater(FXMVECTOR A, FXMVECTOR B) noexcept { DXM_LANES(ShimSSE::from_bits(A.f[i] > B.f[i] ? 0xFFFFFFFFu : 0u)); }
    inline XMVECTOR XMVectorLess(FXMVECTOR A, FXMVECTOR B) noexcept { DXM_LANES(ShimSSE::from_bits(A.f[i] < B.f[i] ? 0xFFFFFFFFu : 0u)); }

    // loads / stores of plain floats and raw 32-bit patterns (movss / movq / movups shapes: the lanes not loaded are zero)
    inline XMVECTOR XMLoadInt(const uint32_t* p) noexcept { XMVECTOR r = XMVectorZero(); memcpy(&r.f[0], p, 4); return r; }
    inline void XMStoreInt(uint32_t* p, FXMVECTOR V) noexcept { memcpy(p, &V.f[0], 4); }
    inline XMVECTOR XMLoadFloat(const float* p) noexcept { return XMVECTOR{ { *p, 0.f, 0.f, 0.f } }; }
    inline XMVECTOR XMLoadFloat2(const XMFLOAT2* p) noexcept { return XMVECTOR{ { p->x, p->y, 0.f, 0.f } }; }
    inline XMVECTOR XMLoadFloat3(const XMFLOAT3* p) noexcept { return XMVECTOR{ { p->x, p->y, p->z, 0.f } }; }
    inline void XMStoreFloat(float* p, FXMVECTOR V) noexcept { *p = V.f[0]; }
    inline void XMStoreFloat2(XMFLOAT2* p, FXMVECTOR V) noexcept { p->x = V.f[0]; p->y = V.f[1]; }
    inline void XMStoreFloat3(XMFLOAT3* p, FXMVECTOR V) noexcept { p->x = V.f[0]; p->y = V.f[1]; p->z = V.f[2]; }
    inline void XMStoreFloat3A(XMFLOAT3A* p, FXMVECTOR V) noexcept { p->x = V.f[0]; p->y = V.f[1]; p->z = V.f[2]; }

    // XMConvertVectorIntToFloat: cvtdq2ps, then mulps by 2^-DivExponent
    inline XMVECTOR XMConvertVectorIntToFloat(FXMVECTOR VInt, uint32_t DivExponent) noexcept
    {
        const float scale = ShimSSE::from_bits(0x3F800000u - (DivExponent << 23));
        DXM_LANES(ShimSSE::cvtdq(int32_t(ShimSSE::bits(VInt.f[i]))) * scale);
    }
    // XMConvertVectorUIntToFloat / XMLoadUInt2/3/4: the top bit is masked off, cvtdq2ps, and 2^31 (g_XMFixUnsigned) added back
    // where it was set - two roundings for values >= 2^31 that are not multiples of 256.
    namespace ShimSSE
    {
        inline float uint_to_float(uint32_t v) noexcept
        {
            const float lo = cvtdq(int32_t(v & 0x7FFFFFFFu));
            return (v & 0x80000000u) ? lo + 2147483648.0f : lo + 0.0f;
        }
        // XMConvertVectorFloatToUInt / XMStoreUInt2/3/4: maxps(V, 0); overflow where > g_XMMaxUInt (65536*65536 - 256 = 4294967040);
        // where >= 2^31 (g_XMUnsignedFix) subtract it, cvttps2dq, flip the top bit back; overflow lanes -> 0xFFFFFFFF
        inline uint32_t float_to_uint(float v) noexcept
        {
            const float s = maxps(v, 0.0f);
            if (s > 4294967040.0f) return 0xFFFFFFFFu;
            const bool big = s >= 2147483648.0f;
            const float t = big ? s - 2147483648.0f : s;
            return uint32_t(cvttps(t)) ^ (big ? 0x80000000u : 0u);
        }
        // XMConvertVectorFloatToInt / XMStoreSInt2/3/4: overflow where > g_XMMaxInt (65536*32768 - 128) -> 0x7FFFFFFF, else cvttps2dq
        inline uint32_t float_to_sint(float v) noexcept { return (v > 2147483520.0f) ? 0x7FFFFFFFu : uint32_t(cvttps(v)); }
    }
    inline XMVECTOR XMConvertVectorUIntToFloat(FXMVECTOR VUInt, uint32_t DivExponent) noexcept
    {
        const float scale = ShimSSE::from_bits(0x3F800000u - (DivExponent << 23));
        DXM_LANES(ShimSSE::uint_to_float(ShimSSE::bits(VUInt.f[i])) * scale);
    }
    inline XMVECTOR XMConvertVectorFloatToInt(FXMVECTOR VFloat, uint32_t MulExponent) noexcept
    {
        const float scale = float(1u << MulExponent);
        DXM_LANES(ShimSSE::from_bits(ShimSSE::float_to_sint(scale * VFloat.f[i])));
    }
    inline XMVECTOR XMConvertVectorFloatToUInt(FXMVECTOR VFloat, uint32_t MulExponent) noexcept
    {
        const float scale = float(1u << MulExponent);
        DXM_LANES(ShimSSE::from_bits(ShimSSE::float_to_uint(scale * VFloat.f[i])));
    }
    inline XMVECTOR XMLoadSInt2(const XMINT2* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), 0.f, 0.f } }; }
    inline XMVECTOR XMLoadSInt3(const XMINT3* p) noexcept { return XMVECTOR{ { float(p->x), float(p->y), float(p->z), 0.f } }; }
    inline XMVECTOR XMLoadUInt2(const XMUINT2* p) noexcept { return XMVECTOR{ { ShimSSE::uint_to_float(p->x), ShimSSE::uint_to_float(p->y), 0.f, 0.f } }; }
    inline XMVECTOR XMLoadUInt3(const XMUINT3* p) noexcept
    {
        return XMVECTOR{ { ShimSSE::uint_to_float(p->x), ShimSSE::uint_to_float(p->y), ShimSSE::uint_to_float(p->z), 0.f } };
    }
    inline XMVECTOR XMLoadUInt4(const XMUINT4* p) noexcept
    {
        return XMVECTOR{ { ShimSSE::uint_to_float(p->x), ShimSSE::uint_to_float(p->y), ShimSSE::uint_to_float(p->z), ShimSSE::uint_to_float(p->w) } };
    }
    inline void XMStoreSInt2(XMINT2* p, FXMVECTOR V) noexcept { p->x = int32_t(ShimSSE::float_to_sint(V.f[0])); p->y = int32_t(ShimSSE::float_to_sint(V.f[1])); }
    inline void XMStoreSInt3(XMINT3* p, FXMVECTOR V) noexcept
    {
        p->x = int32_t(ShimSSE::float_to_sint(V.f[0])); p->y = int32_t(ShimSSE::float_to_sint(V.f[1])); p->z = int32_t(ShimSSE::float_to_sint(V.f[2]));
    }
    inline void XMStoreSInt4(XMINT4* p, FXMVECTOR V) noexcept
    {
        p->x = int32_t(ShimSSE::float_to_sint(V.f[0])); p->y = int32_t(ShimSSE::float_to_sint(V.f[1]));
        p->z = int32_t(ShimSSE::float_to_sint(V.f[2])); p->w = int32_t(ShimSSE::float_to_sint(V.f[3]));
    }
    inline void XMStoreUInt2(XMUINT2* p, FXMVECTOR V) noexcept { p->x = ShimSSE::float_to_uint(V.f[0]); p->y = ShimSSE::float_to_uint(V.f[1]); }
    inline void XMStoreUInt3(XMUINT3* p, FXMVECTOR V) noexcept
    {
        p->x = ShimSSE::float_to_uint(V.f[0]); p->y = ShimSSE::float_to_uint(V.f[1]); p->z = ShimSSE::float_to_uint(V.f[2]);
    }
    inline void XMStoreUInt4(XMUINT4* p, FXMVECTOR V) noexcept
    {
        p->x = ShimSSE::float_to_uint(V.f[0]); p->y = ShimSSE::float_to_uint(V.f[1]); p->z = ShimSSE::float_to_uint(V.f[2]); p->w = ShimSSE::float_to_uint(V.f[3]);
    }

    // XMColorSRGBToRGB / XMColorRGBToSRGB (DirectXMathMisc.inl; one generic body each, built from the vector ops above): saturate,
    // the linear toe by MULTIPLICATION with the constants 1/12.92 and 12.92, the power segment as pow((V + 0.055) * (1/1.055), 2.4)
    // and 1.055 * pow(V, 1/2.4) - 0.055, selected where V > 0.04045 / where V < 0.0031308 picks the toe; w passes through.
    inline XMVECTOR XMColorSRGBToRGB(FXMVECTOR srgb) noexcept
    {
        static const XMVECTORF32 Cutoff = { { 0.04045f, 0.04045f, 0.04045f, 1.f } };
        static const XMVECTORF32 ILinear = { { 1.f / 12.92f, 1.f / 12.92f, 1.f / 12.92f, 1.f } };
        static const XMVECTORF32 Scale = { { 1.f / 1.055f, 1.f / 1.055f, 1.f / 1.055f, 1.f } };
        static const XMVECTORF32 Bias = { { 0.055f, 0.055f, 0.055f, 0.f } };
        static const XMVECTORF32 Gamma = { { 2.4f, 2.4f, 2.4f, 1.f } };
        const XMVECTOR V = XMVectorSaturate(srgb);
        XMVECTOR V0 = XMVectorMultiply(V, ILinear);
        const XMVECTOR V1 = XMVectorPow(XMVectorMultiply(XMVectorAdd(V, Bias), Scale), Gamma);
        V0 = XMVectorSelect(V0, V1, XMVectorGreater(V, Cutoff));
        return XMVectorSelect(srgb, V0, g_XMSelect1110);
    }
    inline XMVECTOR XMColorRGBToSRGB(FXMVECTOR rgb) noexcept
    {
        static const XMVECTORF32 Cutoff = { { 0.0031308f, 0.0031308f, 0.0031308f, 1.f } };
        static const XMVECTORF32 Linear = { { 12.92f, 12.92f, 12.92f, 1.f } };
        static const XMVECTORF32 Scale = { { 1.055f, 1.055f, 1.055f, 1.f } };
        static const XMVECTORF32 Bias = { { 0.055f, 0.055f, 0.055f, 0.f } };
        static const XMVECTORF32 InvGamma = { { 1.0f / 2.4f, 1.0f / 2.4f, 1.0f / 2.4f, 1.f } };
        const XMVECTOR V = XMVectorSaturate(rgb);
        const XMVECTOR V0 = XMVectorMultiply(V, Linear);
        XMVECTOR V1 = XMVectorSubtract(XMVectorMultiply(Scale, XMVectorPow(V, InvGamma)), Bias);
        V1 = XMVectorSelect(V1, V0, XMVectorLess(V, Cutoff));
        return XMVectorSelect(rgb, V1, g_XMSelect1110);
    }
#undef DXM_LANES
}
