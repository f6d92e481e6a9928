// TEST INFRASTRUCTURE ONLY. Pins the scalar stand-ins of oracle/shim against the x86 INSTRUCTIONS they stand for: every ShimSSE
// primitive (maxps / minps operand order, cvtps2dq, cvttps2dq, cvtdq2ps) and every vector op built from an instruction sequence
// (XMVectorRound's 2^23 trick, XMVectorTruncate, XMVectorClamp / XMVectorSaturate, the unsigned <-> float detours) is run next to the
// same sequence written with <emmintrin.h> on the host CPU, over edge values and random bit patterns. What this cannot pin is that the
// sequences are the ones DirectXMath uses (DirectXMath is absent from the image): that remains a statement, made leaf by leaf in the shim.
// Built by oracle/Makefile into oracle/_ref/shim_sse_check; run by tests/test_shim_leaves_cpu.py. Exit code 0 = all equal.
#include <emmintrin.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "DirectXMath.h"
#include "DirectXPackedVector.h"

using namespace DirectX;

static uint32_t rng_state = 0x9E3779B9u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }

static std::vector<float> values()
{
    std::vector<float> v;
    const float edges[] = { 0.f, -0.f, 0.5f, -0.5f, 1.5f, -1.5f, 2.5f, -2.5f, 0.49999997f, 0.50000006f, 1.f, -1.f, 127.5f, -127.5f, 254.5f, 255.f, 255.5f, 32767.5f, -32767.5f,
                            65534.5f, 65535.5f, 8388607.5f, 8388608.f, 8388609.f, -8388607.5f, -8388608.f, 16777216.f, 2147483520.f, 2147483648.f, 2147483904.f, -2147483648.f,
                            -2147483904.f, 4294967040.f, 4294967296.f, 8e9f, -9e9f, 1e-38f, -1e-38f, 1e-45f, 3.4e38f, -3.4e38f, 1.0f / 255.f, 0.5f / 255.f, 0.0031308f, 0.04045f };
    for (float e : edges) v.push_back(e);
    const uint32_t special[] = { 0x7F800000u, 0xFF800000u, 0x7FC00000u, 0xFFC00000u, 0x7F800001u, 0x00000001u, 0x80000001u, 0x007FFFFFu, 0x4B000000u, 0xCB000000u, 0x4AFFFFFFu };
    for (uint32_t s : special) v.push_back(ShimSSE::from_bits(s));
    for (int i = 0; i < 200000; ++i) v.push_back(ShimSSE::from_bits(rnd()));                        // every exponent, NaNs included
    for (int i = 0; i < 200000; ++i) v.push_back(float(int32_t(rnd() >> 8)) * (1.0f / 16.0f) - 500000.0f);  // halves and quarters around integers
    for (int i = 0; i < 100000; ++i) v.push_back(float(rnd() & 0xFFFF) * 0.5f - 16384.f);                 // exact .5 ties
    return v;
}

static int failures = 0;
static void expect(bool ok, const char* what, float a, float b = 0.f)
{
    if (!ok && failures++ < 20) std::printf("MISMATCH %s at %.9g (0x%08X), %.9g\n", what, a, ShimSSE::bits(a), b);
}
static uint32_t lane0(__m128 m) { float f[4]; _mm_storeu_ps(f, m); return ShimSSE::bits(f[0]); }
static uint32_t lane0i(__m128i m) { return uint32_t(_mm_cvtsi128_si32(m)); }

int main()
{
    const std::vector<float> v = values();
    const __m128 zero = _mm_setzero_ps(), one = _mm_set1_ps(1.f);
    const __m128 negZero = _mm_castsi128_ps(_mm_set1_epi32(int(0x80000000u))), absMask = _mm_castsi128_ps(_mm_set1_epi32(0x7FFFFFFF));
    const __m128 noFraction = _mm_set1_ps(8388608.f);
    for (size_t i = 0; i < v.size(); ++i)
    {
        const float a = v[i], b = v[(i * 7 + 3) % v.size()];
        const __m128 A = _mm_set1_ps(a), B = _mm_set1_ps(b);
        const XMVECTOR VA = XMVectorReplicate(a), VB = XMVectorReplicate(b);
        // primitives
        expect(ShimSSE::bits(ShimSSE::maxps(a, b)) == lane0(_mm_max_ps(A, B)), "maxps", a, b);
        expect(ShimSSE::bits(ShimSSE::minps(a, b)) == lane0(_mm_min_ps(A, B)), "minps", a, b);
        expect(uint32_t(ShimSSE::cvtps(a)) == lane0i(_mm_cvtps_epi32(A)), "cvtps2dq", a);
        expect(uint32_t(ShimSSE::cvttps(a)) == lane0i(_mm_cvttps_epi32(A)), "cvttps2dq", a);
        const int32_t ia = int32_t(ShimSSE::bits(a));
        expect(ShimSSE::bits(ShimSSE::cvtdq(ia)) == lane0(_mm_cvtepi32_ps(_mm_set1_epi32(ia))), "cvtdq2ps", a);
        // XMVectorMin / Max / Clamp / Saturate / Negate as the instruction orders the shim states
        expect(ShimSSE::bits(XMVectorMin(VA, VB).f[0]) == lane0(_mm_min_ps(A, B)), "XMVectorMin", a, b);
        expect(ShimSSE::bits(XMVectorMax(VA, VB).f[0]) == lane0(_mm_max_ps(A, B)), "XMVectorMax", a, b);
        expect(ShimSSE::bits(XMVectorClamp(VA, g_XMNegativeOne, g_XMOne).f[0]) == lane0(_mm_min_ps(one, _mm_max_ps(_mm_set1_ps(-1.f), A))), "XMVectorClamp", a);
        expect(ShimSSE::bits(XMVectorSaturate(VA).f[0]) == lane0(_mm_min_ps(_mm_max_ps(A, zero), one)), "XMVectorSaturate", a);
        expect(ShimSSE::bits(XMVectorNegate(VA).f[0]) == lane0(_mm_sub_ps(zero, A)), "XMVectorNegate", a);
        {   // XMVectorRound, SSE2 sequence
            const __m128 sign = _mm_and_ps(A, negZero);
            const __m128 sMagic = _mm_or_ps(noFraction, sign);
            __m128 R1 = _mm_sub_ps(_mm_add_ps(A, sMagic), sMagic);
            __m128 R2 = _mm_and_ps(A, absMask);
            const __m128 mask = _mm_cmple_ps(R2, noFraction);
            R2 = _mm_andnot_ps(mask, A);
            R1 = _mm_and_ps(R1, mask);
            expect(ShimSSE::bits(XMVectorRound(VA).f[0]) == lane0(_mm_xor_ps(R1, R2)), "XMVectorRound", a);
        }
        {   // XMVectorTruncate, SSE2 sequence
            __m128i vTest = _mm_and_si128(_mm_castps_si128(A), _mm_set1_epi32(0x7FFFFFFF));
            vTest = _mm_cmplt_epi32(vTest, _mm_castps_si128(noFraction));
            const __m128i vInt = _mm_cvttps_epi32(A);
            __m128 vResult = _mm_cvtepi32_ps(vInt);
            vResult = _mm_and_ps(vResult, _mm_castsi128_ps(vTest));
            vTest = _mm_andnot_si128(vTest, _mm_castps_si128(A));
            expect(ShimSSE::bits(XMVectorTruncate(VA).f[0]) == lane0(_mm_or_ps(vResult, _mm_castsi128_ps(vTest))), "XMVectorTruncate", a);
        }
        {   // unsigned -> float: mask the top bit, cvtdq2ps, add 2^31 back
            const __m128i U = _mm_set1_epi32(ia);
            const __m128 vMask = _mm_and_ps(_mm_castsi128_ps(U), negZero);
            __m128 vResult = _mm_cvtepi32_ps(_mm_castps_si128(_mm_xor_ps(_mm_castsi128_ps(U), vMask)));
            const __m128i iMask = _mm_srai_epi32(_mm_castps_si128(vMask), 31);
            vResult = _mm_add_ps(vResult, _mm_and_ps(_mm_castsi128_ps(iMask), _mm_set1_ps(2147483648.f)));
            expect(ShimSSE::bits(ShimSSE::uint_to_float(uint32_t(ia))) == lane0(vResult), "uint -> float", a);
        }
        {   // float -> unsigned
            __m128 vResult = _mm_max_ps(A, zero);
            const __m128 vOverflow = _mm_cmpgt_ps(vResult, _mm_set1_ps(65536.0f * 65536.0f - 256.0f));
            __m128 vValue = _mm_set1_ps(2147483648.f);
            __m128 vMask = _mm_cmpge_ps(vResult, vValue);
            vValue = _mm_and_ps(vValue, vMask);
            vResult = _mm_sub_ps(vResult, vValue);
            const __m128i vResulti = _mm_cvttps_epi32(vResult);
            vMask = _mm_and_ps(vMask, negZero);
            vResult = _mm_or_ps(_mm_xor_ps(_mm_castsi128_ps(vResulti), vMask), vOverflow);
            expect(ShimSSE::float_to_uint(a) == lane0(vResult), "float -> uint", a);
        }
        {   // float -> signed
            const __m128 vOverflow = _mm_cmpgt_ps(A, _mm_set1_ps(65536.0f * 32768.0f - 128.0f));
            const __m128i vResulti = _mm_cvttps_epi32(A);
            const __m128 r = _mm_or_ps(_mm_and_ps(vOverflow, absMask), _mm_andnot_ps(vOverflow, _mm_castsi128_ps(vResulti)));
            expect(ShimSSE::float_to_sint(a) == lane0(r), "float -> sint", a);
        }
        {   // a clamp / scale / cvtps2dq store (XMStoreUShortN4's shape) and the truncating one (XMStoreUByteN4's shape)
            PackedVector::XMUSHORTN4 us; PackedVector::XMStoreUShortN4(&us, VA);
            expect(us.x == uint16_t(lane0i(_mm_cvtps_epi32(_mm_mul_ps(_mm_min_ps(_mm_max_ps(A, zero), one), _mm_set1_ps(65535.f))))), "XMStoreUShortN4 shape", a);
            PackedVector::XMUBYTEN4 ub; PackedVector::XMStoreUByteN4(&ub, VA);
            expect(ub.x == uint8_t(lane0i(_mm_cvttps_epi32(_mm_mul_ps(_mm_min_ps(_mm_max_ps(A, zero), one), _mm_set1_ps(255.f))))), "XMStoreUByteN4 shape", a);
        }
    }
    std::printf("shim_sse_check: %zu values, %d mismatches\n", v.size(), failures);
    return failures ? 1 : 0;
}
