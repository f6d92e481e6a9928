// TEST INFRASTRUCTURE ONLY (oracle). Scalar stand-in for the subset of Microsoft DirectXMath that the
// reference's block codecs and filters use. DirectXMath is an un-vendored, un-pinned dependency of
// the reference (CMakeLists.txt:384, build/vcpkg.json:4) and is absent from this image, so the
// conventions below restate its published x86-64 SSE2 code path (the path a Linux g++ -msse2 build of
// the reference takes): per-component IEEE fp32 mul/add/sub without FMA, Dot3 = (x*x' + y*y') + z*z',
// Dot4 = (x*x' + z*z') + (y*y' + w*w'), XMVectorLerp = V0 + (V1 - V0) * t, XMVectorMultiplyAdd = a*b + c
// unfused, half conversion round-to-nearest-even. PARITY UNPINNED at this boundary: no reference test
// or golden vector in /root/reference fixes these conventions.
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
    struct XMFLOAT4 { float x, y, z, w; };
    struct alignas(16) XMFLOAT4A : public XMFLOAT4 {};
    struct XMINT4 { int32_t x, y, z, w; };
    struct XMUINT4 { uint32_t x, y, z, w; };

    XMGLOBALCONST XMVECTORF32 g_XMIdentityR3 = { { 0.0f, 0.0f, 0.0f, 1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMZero = { { 0.0f, 0.0f, 0.0f, 0.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMOne = { { 1.0f, 1.0f, 1.0f, 1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMNegativeOne = { { -1.0f, -1.0f, -1.0f, -1.0f } };
    XMGLOBALCONST XMVECTORF32 g_XMOneHalf = { { 0.5f, 0.5f, 0.5f, 0.5f } };
    XMGLOBALCONST XMVECTORU32 g_XMSelect1110 = { { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u } };
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
}
