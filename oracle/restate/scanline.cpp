// TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's scanline layer - LoadScanline / StoreScanline /
// LoadScanlineLinear / StoreScanlineLinear / ConvertScanline of DirectXTexConvert.cpp - for the formats the MI355X
// path handles, with the reference's own signatures (DirectXTexP.h:401-441) so that the reference's real drivers
// (DirectXTexMipmaps.cpp, DirectXTexResize.cpp, DirectXTexMisc.cpp, DirectXTexCompress.cpp, compiled in place into
// oracle/_ref) link against it. DirectXTexConvert.cpp itself cannot be compiled here: it is 163 DirectXMath symbols
// deep and DirectXMath is not vendored in /root/reference.
//
// PARITY UNPINNED at this layer: the packed loads / stores are DirectXMath's; what is written below is its SSE2
// behaviour as published (see directxtex_amd/csrc/dxtex_store.h for the list), and the reference holds no test
// vectors for it. Every case cites the reference lines it follows.
#include "DirectXTexP.h"

#include <cmath>
#include <cstring>

using namespace DirectX;

namespace
{
    inline float clampf(float v, float lo, float hi) { v = (v > lo) ? v : lo; return (v < hi) ? v : hi; }   // maxps, minps

    // XMConvertHalfToFloat / XMConvertFloatToHalf come from the shim (IEEE, round to nearest even).
    inline uint8_t store_ubn_biased(float v)
    {
        // XMVectorAdd(v, g_8BitBiasV) then XMStoreUByteN4: clamp, * 255, truncate (DirectXTexConvert.cpp:1759-1772)
        float s = v + (0.5f / 255.f);
        s = clampf(s, 0.f, 1.f);
        return uint8_t(uint32_t(s * 255.0f));
    }
    inline uint8_t store_ubn2(float v) { const float s = clampf(v, 0.f, 1.f); return uint8_t(uint32_t(s * 255.0f + 0.5f)); }
    inline int8_t store_bn(float v) { const float s = clampf(v, -1.f, 1.f); return int8_t(int32_t(nearbyintf(s * 127.0f))); }
    inline uint16_t store_usn(float v) { const float s = clampf(v, 0.f, 1.f); return uint16_t(uint32_t(nearbyintf(s * 65535.0f))); }
    inline uint16_t store_half(float v) { return PackedVector::XMConvertFloatToHalf(clampf(v, -65504.f, 65504.f)); }
    inline float loadh(uint16_t h) { return PackedVector::XMConvertHalfToFloat(h); }

    inline float srgb_to_rgb(float v)
    {
        // XMColorSRGBToRGB
        const float s = clampf(v, 0.f, 1.f);
        const float lo = s / 12.92f;
        const float hi = powf((s + 0.055f) / 1.055f, 2.4f);
        return (s > 0.04045f) ? hi : lo;
    }
    inline float rgb_to_srgb(float v)
    {
        // XMColorRGBToSRGB
        const float s = clampf(v, 0.f, 1.f);
        const float lo = s * 12.92f;
        const float hi = 1.055f * powf(s, 1.0f / 2.4f) - 0.055f;
        return (s > 0.0031308f) ? hi : lo;
    }

    // ---- DirectXMath packed types the long-tail formats go through (DirectXPackedVector.inl, SSE2 / scalar behaviour as published) ----
    // XMLoadFloat3PK: one unsigned small float (5-bit exponent, `mbits`-bit mantissa) -> fp32
    inline float load_small_float(uint32_t exponent, uint32_t mantissa, int mbits)
    {
        uint32_t result;
        if (exponent == 0x1f) result = 0x7f800000u | (mantissa << (23 - mbits));          // INF or NAN
        else
        {
            if (exponent != 0) { /* normalised */ }
            else if (mantissa != 0)
            {
                exponent = 1;                                                              // denormal: normalise in the resulting float
                do { exponent--; mantissa <<= 1; } while ((mantissa & (1u << mbits)) == 0);
                mantissa &= (1u << mbits) - 1u;
            }
            else exponent = uint32_t(-112);
            result = ((exponent + 112) << 23) | (mantissa << (23 - mbits));
        }
        float f; memcpy(&f, &result, 4); return f;
    }
    // XMStoreFloat3PK, one channel
    inline uint32_t store_small_float(float value, int mbits)
    {
        uint32_t iv; memcpy(&iv, &value, 4);
        const bool sign = (iv & 0x80000000u) != 0;
        uint32_t I = iv & 0x7FFFFFFFu;
        const uint32_t expMask = 0x1Fu << mbits, allOnes = expMask | ((1u << mbits) - 1u);
        const uint32_t shift = uint32_t(23 - mbits);
        if ((I & 0x7F800000u) == 0x7F800000u)
        {
            uint32_t r = expMask;                                      // INF
            if ((I & 0x7FFFFFu) != 0) r = allOnes;                      // NAN
            else if (sign) r = 0;                                       // -INF is clamped to 0 since 3PK is positive only
            return r;
        }
        if (sign || I < (mbits == 6 ? 0x35800000u : 0x36000000u)) return 0;                 // positive only, or too small
        if (I > (mbits == 6 ? 0x477E0000u : 0x477C0000u)) return expMask - 1u;              // too large: the largest finite value
        if (I < 0x38800000u)
        {
            const uint32_t Shift = 113u - (I >> 23u);                  // too small for a normalised value: make it a denormal
            I = (0x800000u | (I & 0x7FFFFFu)) >> Shift;
        }
        else I += 0xC8000000u;                                          // rebias the exponent
        return ((I + ((1u << (shift - 1)) - 1u) + ((I >> shift) & 1u)) >> shift) & allOnes;
    }
    inline uint16_t store_snorm16(float v) { const float s = clampf(v, -1.f, 1.f); return uint16_t(int16_t(int32_t(nearbyintf(s * 32767.0f)))); }   // XMStoreShortN4/N2: cvtps, pack
    inline uint32_t store_scaled(float v, float scale) { return uint32_t(int32_t(nearbyintf(clampf(v * scale, 0.f, scale)))); }                   // XMStoreU565/U555/UNibble4 after the reference's multiply

    enum : uint32_t { C_UNORM = 1, C_SNORM = 2, C_FLOAT = 4, C_R = 0x10, C_G = 0x20, C_B = 0x40, C_A = 0x80, C_BC = 8, C_POS_ONLY = 0x200,
                      C_UINT = 0x400, C_SINT = 0x800, C_XR = 0x1000, C_YUV = 0x2000, C_DEPTH = 0x4000, C_STENCIL = 0x8000, C_PACKED = 0x10000 };

    // The integer formats (value = the integer itself), as a table: channels, bits per channel, signedness. `scalar` marks the single-channel
    // 8- / 16-bit formats, which the reference loads and stores with its own scalar code (C++ casts: truncation, :1067-1156, :1913-2016); the
    // others go through DirectXMath's XMLoad* / XMStore* (restated from its SSE2 paths: 32-bit stores truncate, 8- / 16-bit stores round to
    // nearest even after a clamp to [0, max] or [-max, max]).
    struct IntFormat { int format; int channels; int bits; bool isSigned; bool scalar; };
    const IntFormat* int_format(DXGI_FORMAT f)
    {
        static const IntFormat table[] = {
            { DXGI_FORMAT_R32G32B32A32_UINT, 4, 32, false, false }, { DXGI_FORMAT_R32G32B32A32_SINT, 4, 32, true, false },
            { DXGI_FORMAT_R32G32B32_UINT, 3, 32, false, false }, { DXGI_FORMAT_R32G32B32_SINT, 3, 32, true, false },
            { DXGI_FORMAT_R16G16B16A16_UINT, 4, 16, false, false }, { DXGI_FORMAT_R16G16B16A16_SINT, 4, 16, true, false },
            { DXGI_FORMAT_R32G32_UINT, 2, 32, false, false }, { DXGI_FORMAT_R32G32_SINT, 2, 32, true, false },
            { DXGI_FORMAT_R8G8B8A8_UINT, 4, 8, false, false }, { DXGI_FORMAT_R8G8B8A8_SINT, 4, 8, true, false },
            { DXGI_FORMAT_R16G16_UINT, 2, 16, false, false }, { DXGI_FORMAT_R16G16_SINT, 2, 16, true, false },
            { DXGI_FORMAT_R32_UINT, 1, 32, false, false }, { DXGI_FORMAT_R32_SINT, 1, 32, true, false },
            { DXGI_FORMAT_R8G8_UINT, 2, 8, false, false }, { DXGI_FORMAT_R8G8_SINT, 2, 8, true, false },
            { DXGI_FORMAT_R16_UINT, 1, 16, false, true }, { DXGI_FORMAT_R16_SINT, 1, 16, true, true },
            { DXGI_FORMAT_R8_UINT, 1, 8, false, true }, { DXGI_FORMAT_R8_SINT, 1, 8, true, true },
        };
        for (const IntFormat& e : table) if (e.format == int(f)) return &e;
        return nullptr;
    }
    // XMLoadUInt* / XMConvertVectorUIntToFloat: cvtdq2ps of (v & 0x7FFFFFFF), + 2^31 if the top bit was set
    inline float uint_to_float(uint32_t v) { const float lo = float(int32_t(v & 0x7FFFFFFFu)); return (v & 0x80000000u) ? lo + 2147483648.0f : lo; }
    // XMStoreUInt* / XMConvertVectorFloatToUInt
    inline uint32_t float_to_uint(float v)
    {
        const float s = (v > 0.0f) ? v : 0.0f;                                  // maxps(v, 0): NaN -> 0
        if (s > 4294967295.0f) return 0xFFFFFFFFu;
        if (s < 2147483648.0f) return uint32_t(int32_t(s));
        const float t = s - 2147483648.0f;                                      // exactly 2^32 passes the overflow test (MaxUInt rounds to 2^32): t = 2^31
        return ((t >= 2147483648.0f) ? 0x80000000u : uint32_t(int32_t(t))) ^ 0x80000000u;       // -> cvttps2dq's 0x80000000 -> 0 after the XOR
    }
    // XMStoreSInt* / XMConvertVectorFloatToInt
    inline uint32_t float_to_sint(float v)
    {
        if (v > 2147483520.0f) return 0x7FFFFFFFu;
        if (!(v >= -2147483648.0f)) return 0x80000000u;                         // cvttps2dq's integer indefinite (also for NaN)
        return uint32_t(int32_t(v));
    }
    inline int32_t clamp_round_even(float v, float lo, float hi) { float s = (v > lo) ? v : lo; s = (s < hi) ? s : hi; return int32_t(nearbyintf(s)); }
    inline int32_t clamp_truncate_std(float v, float lo, float hi)
    {
        const float s = std::max<float>(std::min<float>(v, hi), lo);
        return (s != s) ? 0 : int32_t(s);                                       // the reference's cast of a NaN is undefined; x86-64 leaves 0 in the narrow field
    }
    // the CONVF_* words of g_ConvertTable (DirectXTexConvert.cpp:2960-3047), reduced to what the supported formats use
    uint32_t conv_flags(DXGI_FORMAT f)
    {
        switch (int(f))
        {
        case DXGI_FORMAT_R32G32B32A32_FLOAT: case DXGI_FORMAT_R16G16B16A16_FLOAT: return C_FLOAT | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_R32G32B32_FLOAT: return C_FLOAT | C_R | C_G | C_B;
        case DXGI_FORMAT_R11G11B10_FLOAT: case DXGI_FORMAT_R9G9B9E5_SHAREDEXP: return C_FLOAT | C_POS_ONLY | C_R | C_G | C_B;       // :2979, :3011
        case DXGI_FORMAT_R16G16B16A16_SNORM: return C_SNORM | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_R16G16_SNORM: return C_SNORM | C_R | C_G;
        case DXGI_FORMAT_R16_SNORM: return C_SNORM | C_R;
        case DXGI_FORMAT_R10G10B10A2_UNORM: case DXGI_FORMAT_B5G5R5A1_UNORM: case DXGI_FORMAT_B4G4R4A4_UNORM: case WIN11_DXGI_FORMAT_A4B4G4R4_UNORM: return C_UNORM | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_B5G6R5_UNORM: return C_UNORM | C_R | C_G | C_B;
        case DXGI_FORMAT_R16G16B16A16_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB:
        case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: return C_UNORM | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: return C_UNORM | C_R | C_G | C_B;
        case DXGI_FORMAT_R8G8B8A8_SNORM: return C_SNORM | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_R32G32_FLOAT: case DXGI_FORMAT_R16G16_FLOAT: return C_FLOAT | C_R | C_G;
        case DXGI_FORMAT_R16G16_UNORM: case DXGI_FORMAT_R8G8_UNORM: return C_UNORM | C_R | C_G;
        case DXGI_FORMAT_R8G8_SNORM: return C_SNORM | C_R | C_G;
        case DXGI_FORMAT_R32_FLOAT: case DXGI_FORMAT_R16_FLOAT: return C_FLOAT | C_R;
        case DXGI_FORMAT_R16_UNORM: case DXGI_FORMAT_R8_UNORM: return C_UNORM | C_R;
        case DXGI_FORMAT_R8_SNORM: return C_SNORM | C_R;
        case DXGI_FORMAT_A8_UNORM: return C_UNORM | C_A;
        case DXGI_FORMAT_BC1_UNORM: case DXGI_FORMAT_BC1_UNORM_SRGB: case DXGI_FORMAT_BC2_UNORM: case DXGI_FORMAT_BC2_UNORM_SRGB:
        case DXGI_FORMAT_BC3_UNORM: case DXGI_FORMAT_BC3_UNORM_SRGB: case DXGI_FORMAT_BC7_UNORM: case DXGI_FORMAT_BC7_UNORM_SRGB:
            return C_UNORM | C_BC | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_BC4_UNORM: return C_UNORM | C_BC | C_R;
        case DXGI_FORMAT_BC4_SNORM: return C_SNORM | C_BC | C_R;
        case DXGI_FORMAT_BC5_UNORM: return C_UNORM | C_BC | C_R | C_G;
        case DXGI_FORMAT_BC5_SNORM: return C_SNORM | C_BC | C_R | C_G;
        case DXGI_FORMAT_BC6H_UF16: case DXGI_FORMAT_BC6H_SF16: return C_FLOAT | C_BC | C_R | C_G | C_B | C_A;
        case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM: return C_UNORM | C_XR | C_R | C_G | C_B | C_A;                                        // :3030
        case DXGI_FORMAT_AYUV: case DXGI_FORMAT_Y410: case DXGI_FORMAT_Y416: return C_UNORM | C_YUV | C_R | C_G | C_B | C_A;                  // :3037-3039
        case DXGI_FORMAT_R10G10B10A2_UINT: return C_UINT | C_R | C_G | C_B | C_A;                                                         // :2978
        case DXGI_FORMAT_D32_FLOAT_S8X24_UINT: return C_FLOAT | C_DEPTH | C_STENCIL;                                                      // :2976
        case DXGI_FORMAT_D32_FLOAT: return C_FLOAT | C_DEPTH;                                                                           // :2990
        case DXGI_FORMAT_D24_UNORM_S8_UINT: return C_UNORM | C_DEPTH | C_STENCIL;                                                         // :2994
        case DXGI_FORMAT_D16_UNORM: return C_UNORM | C_DEPTH;                                                                           // :3000
        case DXGI_FORMAT_R1_UNORM: return C_UNORM | C_R;                                                                                 // :3010
        case DXGI_FORMAT_R8G8_B8G8_UNORM: case DXGI_FORMAT_G8R8_G8B8_UNORM: return C_UNORM | C_PACKED | C_R | C_G | C_B;                   // :3012-3013
        case DXGI_FORMAT_YUY2: case DXGI_FORMAT_Y210: case DXGI_FORMAT_Y216: return C_UNORM | C_YUV | C_PACKED | C_R | C_G | C_B;          // :3038-3040
        default:
            if (const IntFormat* e = int_format(f))
                return (e->isSigned ? C_SINT : C_UINT) | C_R | (e->channels > 1 ? C_G : 0) | (e->channels > 2 ? C_B : 0) | (e->channels > 3 ? C_A : 0);
            return 0;
        }
    }

    bool is_srgb_format(DXGI_FORMAT f)
    {
        switch (int(f))
        {
        case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: case DXGI_FORMAT_BC1_UNORM_SRGB: case DXGI_FORMAT_BC2_UNORM_SRGB: case DXGI_FORMAT_BC3_UNORM_SRGB:
        case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB: case DXGI_FORMAT_BC7_UNORM_SRGB: return true;
        default: return false;
        }
    }

    // the format list of LoadScanlineLinear / StoreScanlineLinear (:2813-2858, :2890-2928), supported subset
    bool linear_filter_srgb_ok(DXGI_FORMAT f)
    {
        switch (int(f))
        {
        case DXGI_FORMAT_R32G32B32A32_FLOAT: case DXGI_FORMAT_R16G16B16A16_FLOAT: case DXGI_FORMAT_R16G16B16A16_UNORM: case DXGI_FORMAT_R32G32_FLOAT:
        case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R16G16_FLOAT: case DXGI_FORMAT_R16G16_UNORM: case DXGI_FORMAT_R32_FLOAT:
        case DXGI_FORMAT_R8G8_UNORM: case DXGI_FORMAT_R16_FLOAT: case DXGI_FORMAT_R16_UNORM: case DXGI_FORMAT_R8_UNORM:
        case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM: return true;
        case DXGI_FORMAT_R32G32B32_FLOAT: case DXGI_FORMAT_R10G10B10A2_UNORM: case DXGI_FORMAT_R11G11B10_FLOAT: case DXGI_FORMAT_R9G9B9E5_SHAREDEXP:
        case DXGI_FORMAT_B5G6R5_UNORM: case DXGI_FORMAT_B5G5R5A1_UNORM: case DXGI_FORMAT_B4G4R4A4_UNORM: return true;     // :2826-2847
        case DXGI_FORMAT_R8G8_B8G8_UNORM: case DXGI_FORMAT_G8R8_G8B8_UNORM: case WIN11_DXGI_FORMAT_A4B4G4R4_UNORM: return true;   // :2841-2842, :2848
        default: return false;
        }
    }
}

// ---- LoadScanline (DirectXTexConvert.cpp:779-1619) ---------------------------------------------------------------------------
bool DirectX::Internal::LoadScanline(XMVECTOR* pDestination, size_t count, const void* pSource, size_t size, DXGI_FORMAT format) noexcept
{
    if (!pDestination || !count || !pSource || !size) return false;
    const uint8_t* s = static_cast<const uint8_t*>(pSource);
    auto texels = [&](size_t bytes) { const size_t n = size / bytes; return n < count ? n : count; };
    if (const IntFormat* e = int_format(format))
    {
        // :805-809, :814-818, :826-833, :838-842, :913-920, :928-935, :952-981, :1031-1038, :1067-1156; absent channels from (0, 0, 0, 1)
        const size_t bytes = size_t(e->channels * e->bits / 8);
        for (size_t i = 0, n = texels(bytes); i < n; ++i)
        {
            float v[4] = { 0.f, 0.f, 0.f, 1.f };
            for (int c = 0; c < e->channels; ++c)
            {
                const uint8_t* p = s + i * bytes + size_t(c * e->bits / 8);
                if (e->bits == 32) { uint32_t u; memcpy(&u, p, 4); v[c] = e->isSigned ? float(int32_t(u)) : uint_to_float(u); }
                else if (e->bits == 16) { uint16_t u; memcpy(&u, p, 2); v[c] = e->isSigned ? float(int16_t(u)) : float(u); }
                else v[c] = e->isSigned ? float(int8_t(*p)) : float(*p);
            }
            pDestination[i] = XMVectorSet(v[0], v[1], v[2], v[3]);
        }
        return true;
    }
    switch (int(format))
    {
    case DXGI_FORMAT_R10G10B10A2_UINT:          // XMLoadUDec4, :903-904
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            uint32_t u; memcpy(&u, s + i * 4, 4);
            pDestination[i] = XMVectorSet(float(u & 0x3FF), float((u >> 10) & 0x3FF), float((u >> 20) & 0x3FF), float(u >> 30));
        }
        return true;
    case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM:    // XMLoadUDecN4_XR, :900-901
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            uint32_t u; memcpy(&u, s + i * 4, 4);
            // the SSE2 path multiplies by XRMul = { 1/510, 1/(510*2^10), 1/(510*2^20), 1/(3*2^30) } after subtracting the bias in place
            const float m = 1.0f / 510.0f;
            pDestination[i] = XMVectorSet(float(int32_t(u & 0x3FF) - 0x180) * m, float(int32_t((u >> 10) & 0x3FF) - 0x180) * m,
                                          float(int32_t((u >> 20) & 0x3FF) - 0x180) * m, float(u >> 30) * (1.0f / 3.0f));
        }
        return true;
    case DXGI_FORMAT_AYUV:                      // :1291-1326
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint8_t* b = s + i * 4;
            const int v = int(b[0]) - 128, u = int(b[1]) - 128, y = int(b[2]) - 16;
            const int r = (298 * y + 409 * v + 128) >> 8, g = (298 * y - 100 * u - 208 * v + 128) >> 8, bl = (298 * y + 516 * u + 128) >> 8;
            pDestination[i] = XMVectorSet(float(std::min<int>(std::max<int>(r, 0), 255)) / 255.f, float(std::min<int>(std::max<int>(g, 0), 255)) / 255.f,
                                          float(std::min<int>(std::max<int>(bl, 0), 255)) / 255.f, float(b[3]) / 255.f);
        }
        return true;
    case DXGI_FORMAT_Y410:                      // :1328-1360
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            uint32_t w; memcpy(&w, s + i * 4, 4);
            const int64_t u = int(w & 0x3FF) - 512, y = int((w >> 10) & 0x3FF) - 64, v = int((w >> 20) & 0x3FF) - 512;
            const int r = int((76533 * y + 104905 * v + 32768) >> 16), g = int((76533 * y - 25747 * u - 53425 * v + 32768) >> 16), bl = int((76533 * y + 132590 * u + 32768) >> 16);
            pDestination[i] = XMVectorSet(float(std::min<int>(std::max<int>(r, 0), 1023)) / 1023.f, float(std::min<int>(std::max<int>(g, 0), 1023)) / 1023.f,
                                          float(std::min<int>(std::max<int>(bl, 0), 1023)) / 1023.f, float(w >> 30) / 3.f);
        }
        return true;
    case DXGI_FORMAT_Y416:                      // :1362-1394
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            uint16_t h[4]; memcpy(h, s + i * 8, 8);
            const int64_t u = int64_t(h[0]) - 32768, y = int64_t(h[1]) - 4096, v = int64_t(h[2]) - 32768;
            const int a = int(h[3]);
            const int r = int((76607 * y + 105006 * v + 32768) >> 16), g = int((76607 * y - 25772 * u - 53477 * v + 32768) >> 16), bl = int((76607 * y + 132718 * u + 32768) >> 16);
            pDestination[i] = XMVectorSet(float(std::min<int>(std::max<int>(r, 0), 65535)) / 65535.f, float(std::min<int>(std::max<int>(g, 0), 65535)) / 65535.f,
                                          float(std::min<int>(std::max<int>(bl, 0), 65535)) / 65535.f, float(std::min<int>(std::max<int>(a, 0), 65535)) / 65535.f);
        }
        return true;
    // ---- formats whose element holds several texels: the reference's element loops, two (eight) destinations per element ----
    case DXGI_FORMAT_R1_UNORM:                  // :1171-1188
    {
        size_t o = 0;
        for (size_t icount = 0; icount < size; ++icount)
            for (size_t bcount = 8; bcount > 0; --bcount)
            {
                if (o >= count) break;
                pDestination[o++] = XMVectorSet(((s[icount] >> (bcount - 1)) & 0x1) ? 1.f : 0.f, 0.f, 0.f, 1.f);
            }
        return true;
    }
    case DXGI_FORMAT_R8G8_B8G8_UNORM:           // :1192-1207: XMLoadUByteN4, then (x, y, z, 1) and (x, w, z, 1)
    case DXGI_FORMAT_G8R8_G8B8_UNORM:           // :1209-1225: (y, x, w, 1) and (y, z, w, 1)
    {
        if (size < 4) return false;
        size_t o = 0;
        for (size_t icount = 0; icount < (size - 4 + 1); icount += 4)
        {
            const float v[4] = { float(s[icount]) * (1.0f / 255.0f), float(s[icount + 1]) * (1.0f / 255.0f), float(s[icount + 2]) * (1.0f / 255.0f), float(s[icount + 3]) * (1.0f / 255.0f) };
            const bool rg = format == DXGI_FORMAT_R8G8_B8G8_UNORM;
            if (o >= count) break;
            pDestination[o++] = rg ? XMVectorSet(v[0], v[1], v[2], 1.f) : XMVectorSet(v[1], v[0], v[3], 1.f);
            if (o >= count) break;
            pDestination[o++] = rg ? XMVectorSet(v[0], v[3], v[2], 1.f) : XMVectorSet(v[1], v[2], v[3], 1.f);
        }
        return true;
    }
    case DXGI_FORMAT_YUY2:                      // :1399-1434
    {
        if (size < 4) return false;
        size_t o = 0;
        for (size_t icount = 0; icount < (size - 4 + 1); icount += 4)
        {
            const int y0 = int(s[icount]) - 16, u = int(s[icount + 1]) - 128, y1 = int(s[icount + 2]) - 16, v = int(s[icount + 3]) - 128;
            for (int half = 0; half < 2; ++half)
            {
                const int y = half ? y1 : y0;
                const int r = (298 * y + 409 * v + 128) >> 8, g = (298 * y - 100 * u - 208 * v + 128) >> 8, b = (298 * y + 516 * u + 128) >> 8;
                if (o >= count) break;
                pDestination[o++] = XMVectorSet(float(std::min<int>(std::max<int>(r, 0), 255)) / 255.f, float(std::min<int>(std::max<int>(g, 0), 255)) / 255.f,
                                                float(std::min<int>(std::max<int>(b, 0), 255)) / 255.f, 1.f);
            }
        }
        return true;
    }
    case DXGI_FORMAT_Y210:                      // :1436-1472
    case DXGI_FORMAT_Y216:                      // :1474-1510
    {
        if (size < 8) return false;
        const bool ten = format == DXGI_FORMAT_Y210;
        size_t o = 0;
        for (size_t icount = 0; icount < (size - 8 + 1); icount += 8)
        {
            uint16_t h[4]; memcpy(h, s + icount, 8);
            const int64_t y0 = ten ? int64_t(h[0] >> 6) - 64 : int64_t(h[0]) - 4096, u = ten ? int64_t(h[1] >> 6) - 512 : int64_t(h[1]) - 32768;
            const int64_t y1 = ten ? int64_t(h[2] >> 6) - 64 : int64_t(h[2]) - 4096, v = ten ? int64_t(h[3] >> 6) - 512 : int64_t(h[3]) - 32768;
            for (int half = 0; half < 2; ++half)
            {
                const int64_t y = half ? y1 : y0;
                int r, g, b; float top;
                if (ten) { r = int((76533 * y + 104905 * v + 32768) >> 16); g = int((76533 * y - 25747 * u - 53425 * v + 32768) >> 16); b = int((76533 * y + 132590 * u + 32768) >> 16); top = 1023.f; }
                else { r = int((76607 * y + 105006 * v + 32768) >> 16); g = int((76607 * y - 25772 * u - 53477 * v + 32768) >> 16); b = int((76607 * y + 132718 * u + 32768) >> 16); top = 65535.f; }
                if (o >= count) break;
                pDestination[o++] = XMVectorSet(float(std::min<int>(std::max<int>(r, 0), int(top))) / top, float(std::min<int>(std::max<int>(g, 0), int(top))) / top,
                                                float(std::min<int>(std::max<int>(b, 0), int(top))) / top, 1.f);
            }
        }
        return true;
    }
    case DXGI_FORMAT_R32G32B32A32_FLOAT:        // :798-803
    {
        const size_t n = texels(16);
        memcpy(pDestination, s, n * 16);
        return true;
    }
    case DXGI_FORMAT_R16G16B16A16_FLOAT:        // XMLoadHalf4, :820-821
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            const uint16_t* h = reinterpret_cast<const uint16_t*>(s + i * 8);
            pDestination[i] = XMVectorSet(loadh(h[0]), loadh(h[1]), loadh(h[2]), loadh(h[3]));
        }
        return true;
    case DXGI_FORMAT_R16G16B16A16_UNORM:        // XMLoadUShortN4
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            const uint16_t* h = reinterpret_cast<const uint16_t*>(s + i * 8);
            pDestination[i] = XMVectorSet(float(h[0]) * (1.0f / 65535.0f), float(h[1]) * (1.0f / 65535.0f), float(h[2]) * (1.0f / 65535.0f), float(h[3]) * (1.0f / 65535.0f));
        }
        return true;
    case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB:      // XMLoadUByteN4, :909-911
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint8_t* b = s + i * 4;
            pDestination[i] = XMVectorSet(float(b[0]) * (1.0f / 255.0f), float(b[1]) * (1.0f / 255.0f), float(b[2]) * (1.0f / 255.0f), float(b[3]) * (1.0f / 255.0f));
        }
        return true;
    case DXGI_FORMAT_R8G8B8A8_SNORM:            // XMLoadByteN4, :916-917
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const int8_t* b = reinterpret_cast<const int8_t*>(s + i * 4);
            float v[4];
            for (int c = 0; c < 4; ++c) { v[c] = float(b[c]) * (1.0f / 127.0f); v[c] = (v[c] > -1.0f) ? v[c] : -1.0f; }
            pDestination[i] = XMVectorSet(v[0], v[1], v[2], v[3]);
        }
        return true;
    case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB:      // :1260-1273
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint8_t* b = s + i * 4;
            pDestination[i] = XMVectorSet(float(b[2]) * (1.0f / 255.0f), float(b[1]) * (1.0f / 255.0f), float(b[0]) * (1.0f / 255.0f), float(b[3]) * (1.0f / 255.0f));
        }
        return true;
    case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB:      // :1275-1289
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint8_t* b = s + i * 4;
            pDestination[i] = XMVectorSet(float(b[2]) * (1.0f / 255.0f), float(b[1]) * (1.0f / 255.0f), float(b[0]) * (1.0f / 255.0f), 1.0f);
        }
        return true;
    case DXGI_FORMAT_R32G32_FLOAT:              // LOAD_SCANLINE2(XMFLOAT2, ..., g_XMIdentityR3)
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            const float* f = reinterpret_cast<const float*>(s + i * 8);
            pDestination[i] = XMVectorSet(f[0], f[1], 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_R16G16_FLOAT:              // :922-923
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint16_t* h = reinterpret_cast<const uint16_t*>(s + i * 4);
            pDestination[i] = XMVectorSet(loadh(h[0]), loadh(h[1]), 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_R16G16_UNORM:              // :925-926
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint16_t* h = reinterpret_cast<const uint16_t*>(s + i * 4);
            pDestination[i] = XMVectorSet(float(h[0]) * (1.0f / 65535.0f), float(h[1]) * (1.0f / 65535.0f), 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_D32_FLOAT:                 // :937
    case DXGI_FORMAT_R32_FLOAT:                 // :938-950
        for (size_t i = 0, n = texels(4); i < n; ++i) pDestination[i] = XMVectorSet(reinterpret_cast<const float*>(s)[i], 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_D32_FLOAT_S8X24_UINT:      // :844-860: (depth, float(stencil byte), 0, 1)
        for (size_t i = 0, n = texels(8); i < n; ++i) pDestination[i] = XMVectorSet(reinterpret_cast<const float*>(s)[i * 2], float(s[i * 8 + 4]), 0.f, 1.f);
        return true;
    case DXGI_FORMAT_D24_UNORM_S8_UINT:         // :982-997
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint32_t v = reinterpret_cast<const uint32_t*>(s)[i];
            const auto dd = static_cast<float>(v & 0xFFFFFF) / 16777215.f;
            const auto ss = static_cast<float>((v & 0xFF000000) >> 24);
            pDestination[i] = XMVectorSet(dd, ss, 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_R8G8_UNORM:                // XMLoadUByteN2, :1028-1029
        for (size_t i = 0, n = texels(2); i < n; ++i)
            pDestination[i] = XMVectorSet(float(s[i * 2]) * (1.0f / 255.0f), float(s[i * 2 + 1]) * (1.0f / 255.0f), 0.f, 1.f);
        return true;
    case DXGI_FORMAT_R8G8_SNORM:                // XMLoadByteN2, :1034-1035
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            float x = float(int8_t(s[i * 2])) * (1.0f / 127.0f), y = float(int8_t(s[i * 2 + 1])) * (1.0f / 127.0f);
            x = (x > -1.0f) ? x : -1.0f; y = (y > -1.0f) ? y : -1.0f;
            pDestination[i] = XMVectorSet(x, y, 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_R16_FLOAT:                 // :1040-1051
        for (size_t i = 0, n = texels(2); i < n; ++i) pDestination[i] = XMVectorSet(loadh(reinterpret_cast<const uint16_t*>(s)[i]), 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_D16_UNORM:                 // :1053
    case DXGI_FORMAT_R16_UNORM:                 // :1054-1065
        for (size_t i = 0, n = texels(2); i < n; ++i) pDestination[i] = XMVectorSet(float(reinterpret_cast<const uint16_t*>(s)[i]) / 65535.f, 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_R8_UNORM:                  // :1106-1117
        for (size_t i = 0, n = texels(1); i < n; ++i) pDestination[i] = XMVectorSet(float(s[i]) / 255.f, 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_R8_SNORM:                  // :1132-1143
        for (size_t i = 0, n = texels(1); i < n; ++i) pDestination[i] = XMVectorSet(float(int8_t(s[i])) / 127.f, 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_A8_UNORM:                  // :1158-1169
        for (size_t i = 0, n = texels(1); i < n; ++i) pDestination[i] = XMVectorSet(0.f, 0.f, 0.f, float(s[i]) / 255.f);
        return true;
    case DXGI_FORMAT_R32G32B32_FLOAT:           // LOAD_SCANLINE3(XMFLOAT3, XMLoadFloat3, g_XMIdentityR3), :811-812
        for (size_t i = 0, n = texels(12); i < n; ++i)
        {
            const float* f = reinterpret_cast<const float*>(s + i * 12);
            pDestination[i] = XMVectorSet(f[0], f[1], f[2], 1.f);
        }
        return true;
    case DXGI_FORMAT_R16G16B16A16_SNORM:        // XMLoadShortN4, :829-830
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            const int16_t* h = reinterpret_cast<const int16_t*>(s + i * 8);
            float c[4];
            for (int k = 0; k < 4; ++k) { c[k] = float(h[k]) * (1.0f / 32767.0f); c[k] = (c[k] > -1.0f) ? c[k] : -1.0f; }
            pDestination[i] = XMVectorSet(c[0], c[1], c[2], c[3]);
        }
        return true;
    case DXGI_FORMAT_R16G16_SNORM:              // XMLoadShortN2, :931-932
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const int16_t* h = reinterpret_cast<const int16_t*>(s + i * 4);
            float x = float(h[0]) * (1.0f / 32767.0f), y = float(h[1]) * (1.0f / 32767.0f);
            x = (x > -1.0f) ? x : -1.0f; y = (y > -1.0f) ? y : -1.0f;
            pDestination[i] = XMVectorSet(x, y, 0.f, 1.f);
        }
        return true;
    case DXGI_FORMAT_R16_SNORM:                 // :1080-1091
        for (size_t i = 0, n = texels(2); i < n; ++i) pDestination[i] = XMVectorSet(static_cast<float>(reinterpret_cast<const int16_t*>(s)[i]) / 32767.f, 0.f, 0.f, 1.f);
        return true;
    case DXGI_FORMAT_R10G10B10A2_UNORM:         // XMLoadUDecN4, :897-898
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint32_t v = reinterpret_cast<const uint32_t*>(s)[i];
            pDestination[i] = XMVectorSet(float(v & 0x3FF) * (1.0f / 1023.0f), float((v >> 10) & 0x3FF) * (1.0f / 1023.0f),
                                          float((v >> 20) & 0x3FF) * (1.0f / 1023.0f), float(v >> 30) * (1.0f / 3.0f));
        }
        return true;
    case DXGI_FORMAT_R11G11B10_FLOAT:           // XMLoadFloat3PK, :906-907
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint32_t v = reinterpret_cast<const uint32_t*>(s)[i];
            pDestination[i] = XMVectorSet(load_small_float((v >> 6) & 0x1F, v & 0x3F, 6), load_small_float((v >> 17) & 0x1F, (v >> 11) & 0x3F, 6),
                                          load_small_float((v >> 27) & 0x1F, (v >> 22) & 0x1F, 5), 1.f);
        }
        return true;
    case DXGI_FORMAT_R9G9B9E5_SHAREDEXP:        // XMLoadFloat3SE, :1189-1190
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const uint32_t v = reinterpret_cast<const uint32_t*>(s)[i];
            union { float f; int32_t i; } fi;
            fi.i = 0x33800000 + int32_t((v >> 27) << 23);
            const float Scale = fi.f;
            pDestination[i] = XMVectorSet(Scale * float(v & 0x1FF), Scale * float((v >> 9) & 0x1FF), Scale * float((v >> 18) & 0x1FF), 1.f);
        }
        return true;
    case DXGI_FORMAT_B5G6R5_UNORM:              // :1227-1242
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const uint16_t v = reinterpret_cast<const uint16_t*>(s)[i];
            const float x = float(v & 0x1F) * (1.f / 31.f), y = float((v >> 5) & 0x3F) * (1.f / 63.f), z = float((v >> 11) & 0x1F) * (1.f / 31.f);
            pDestination[i] = XMVectorSet(z, y, x, 1.f);        // XMVectorSwizzle<2, 1, 0, 3>, w from g_XMIdentityR3
        }
        return true;
    case DXGI_FORMAT_B5G5R5A1_UNORM:            // :1244-1258
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const uint16_t v = reinterpret_cast<const uint16_t*>(s)[i];
            const float x = float(v & 0x1F) * (1.f / 31.f), y = float((v >> 5) & 0x1F) * (1.f / 31.f), z = float((v >> 10) & 0x1F) * (1.f / 31.f);
            pDestination[i] = XMVectorSet(z, y, x, float(v >> 15) * 1.f);
        }
        return true;
    case DXGI_FORMAT_B4G4R4A4_UNORM:            // :1511-1525
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const uint16_t v = reinterpret_cast<const uint16_t*>(s)[i];
            pDestination[i] = XMVectorSet(float((v >> 8) & 0xF) * (1.f / 15.f), float((v >> 4) & 0xF) * (1.f / 15.f), float(v & 0xF) * (1.f / 15.f), float(v >> 12) * (1.f / 15.f));
        }
        return true;
    case WIN11_DXGI_FORMAT_A4B4G4R4_UNORM:      // :1527-1541: XMLoadUNibble4 * 1/15, swizzle <3, 2, 1, 0>
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const uint16_t v = reinterpret_cast<const uint16_t*>(s)[i];
            const float nib[4] = { float(v & 0xF) * (1.f / 15.f), float((v >> 4) & 0xF) * (1.f / 15.f), float((v >> 8) & 0xF) * (1.f / 15.f), float(v >> 12) * (1.f / 15.f) };
            pDestination[i] = XMVectorSet(nib[3], nib[2], nib[1], nib[0]);
        }
        return true;
    default:
        return false;
    }
}

// ---- StoreScanline (:1629-2533) --------------------------------------------------------------------------------------------------
bool DirectX::Internal::StoreScanline(void* pDestination, size_t size, DXGI_FORMAT format, const XMVECTOR* pSource, size_t count, float threshold) noexcept
{
    if (!pDestination || !size || !pSource || !count) return false;
    uint8_t* d = static_cast<uint8_t*>(pDestination);
    auto texels = [&](size_t bytes) { const size_t n = size / bytes; return n < count ? n : count; };
    if (const IntFormat* e = int_format(format))
    {
        // :1674-1687, :1707-1714, :1719-1723, :1774-1781, :1801-1808, :1824-1850, :1873-1880, :1913-2016
        const size_t bytes = size_t(e->channels * e->bits / 8);
        for (size_t i = 0, n = texels(bytes); i < n; ++i)
            for (int c = 0; c < e->channels; ++c)
            {
                const float v = pSource[i].f[c];
                uint8_t* p = d + i * bytes + size_t(c * e->bits / 8);
                if (e->bits == 32) { const uint32_t u = e->isSigned ? float_to_sint(v) : float_to_uint(v); memcpy(p, &u, 4); }
                else
                {
                    const float hi = e->isSigned ? (e->bits == 16 ? 32767.f : 127.f) : (e->bits == 16 ? 65535.f : 255.f), lo = e->isSigned ? -hi : 0.f;
                    const int32_t q = e->scalar ? clamp_truncate_std(v, lo, hi) : clamp_round_even(v, lo, hi);
                    if (e->bits == 16) { const uint16_t u = uint16_t(q); memcpy(p, &u, 2); } else *p = uint8_t(q);
                }
            }
        return true;
    }
    switch (int(format))
    {
    case DXGI_FORMAT_R10G10B10A2_UINT:          // XMStoreUDec4, :1753-1754: clamp (maxps / minps), truncate
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            auto q = [](float v, float hi) { float t = (v > 0.f) ? v : 0.f; t = (t < hi) ? t : hi; return uint32_t(int32_t(t)); };
            const uint32_t u = q(pSource[i].f[0], 1023.f) | (q(pSource[i].f[1], 1023.f) << 10) | (q(pSource[i].f[2], 1023.f) << 20) | (q(pSource[i].f[3], 3.f) << 30);
            memcpy(d + i * 4, &u, 4);
        }
        return true;
    case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM:    // XMStoreUDecN4_XR, :1750-1751
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            auto q = [](float v, float scale, float bias, float hi) { float t = v * scale + bias; t = (t > 0.f) ? t : 0.f; t = (t < hi) ? t : hi; return uint32_t(t); };
            const uint32_t u = (q(pSource[i].f[0], 510.f, 384.f, 1023.f) & 0x3FF) | ((q(pSource[i].f[1], 510.f, 384.f, 1023.f) & 0x3FF) << 10) |
                               ((q(pSource[i].f[2], 510.f, 384.f, 1023.f) & 0x3FF) << 20) | (q(pSource[i].f[3], 3.f, 0.f, 3.f) << 30);
            memcpy(d + i * 4, &u, 4);
        }
        return true;
    case DXGI_FORMAT_AYUV:                      // :2173-2202 (XMStoreUByteN4 on the raw vector: no bias here)
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            auto ubn = [](float v) { const float t = clampf(v, 0.f, 1.f); return int(uint32_t(t * 255.0f)); };
            const int r = ubn(pSource[i].f[0]), g = ubn(pSource[i].f[1]), b = ubn(pSource[i].f[2]);
            const int y = ((66 * r + 129 * g + 25 * b + 128) >> 8) + 16, u = ((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128, v = ((112 * r - 94 * g - 18 * b + 128) >> 8) + 128;
            d[i * 4 + 0] = uint8_t(std::min<int>(std::max<int>(v, 0), 255)); d[i * 4 + 1] = uint8_t(std::min<int>(std::max<int>(u, 0), 255));
            d[i * 4 + 2] = uint8_t(std::min<int>(std::max<int>(y, 0), 255)); d[i * 4 + 3] = uint8_t(ubn(pSource[i].f[3]));
        }
        return true;
    case DXGI_FORMAT_Y410:                      // :2204-2237 (XMStoreUDecN4: saturate, scale, truncate)
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            auto dec = [](float v, float scale) { const float t = clampf(v, 0.f, 1.f); return int64_t(uint32_t(t * scale)); };
            const int64_t r = dec(pSource[i].f[0], 1023.f), g = dec(pSource[i].f[1], 1023.f), b = dec(pSource[i].f[2], 1023.f), a = dec(pSource[i].f[3], 3.f);
            const int y = int((16780 * r + 32942 * g + 6544 * b + 32768) >> 16) + 64, u = int((-9683 * r - 19017 * g + 28700 * b + 32768) >> 16) + 512,
                      v = int((28700 * r - 24033 * g - 4667 * b + 32768) >> 16) + 512;
            const uint32_t w = uint32_t(std::min<int>(std::max<int>(u, 0), 1023)) | (uint32_t(std::min<int>(std::max<int>(y, 0), 1023)) << 10) |
                               (uint32_t(std::min<int>(std::max<int>(v, 0), 1023)) << 20) | (uint32_t(a) << 30);
            memcpy(d + i * 4, &w, 4);
        }
        return true;
    case DXGI_FORMAT_Y416:                      // :2239-2272 (XMStoreUShortN4)
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            const int64_t r = store_usn(pSource[i].f[0]), g = store_usn(pSource[i].f[1]), b = store_usn(pSource[i].f[2]);
            const int y = int((16763 * r + 32910 * g + 6537 * b + 32768) >> 16) + 4096, u = int((-9674 * r - 18998 * g + 28672 * b + 32768) >> 16) + 32768,
                      v = int((28672 * r - 24010 * g - 4662 * b + 32768) >> 16) + 32768;
            uint16_t h[4] = { uint16_t(std::min<int>(std::max<int>(u, 0), 65535)), uint16_t(std::min<int>(std::max<int>(y, 0), 65535)),
                              uint16_t(std::min<int>(std::max<int>(v, 0), 65535)), store_usn(pSource[i].f[3]) };
            memcpy(d + i * 8, h, 8);
        }
        return true;
    // ---- formats whose element holds several texels: the reference's element loops, two (eight) sources per element ----
    case DXGI_FORMAT_R1_UNORM:                  // :2033-2055
    {
        size_t i = 0;
        for (size_t icount = 0; icount < size; ++icount)
        {
            uint8_t pixels = 0;
            for (size_t bcount = 8; bcount > 0; --bcount)
            {
                if (i >= count) break;
                if (pSource[i++].f[0] > 0.25f) pixels |= uint8_t(1 << (bcount - 1));
            }
            d[icount] = pixels;
        }
        return true;
    }
    case DXGI_FORMAT_R8G8_B8G8_UNORM:           // :2060-2075
    case DXGI_FORMAT_G8R8_G8B8_UNORM:           // :2077-2094
    {
        if (size < 4) return false;
        size_t i = 0;
        for (size_t icount = 0; icount < (size - 4 + 1); icount += 4)
        {
            if (i >= count) break;
            const float* v0 = pSource[i++].f;
            const float g1 = (i < count) ? pSource[i++].f[1] : 0.f;          // XMVectorSplatY of the second texel, or zero
            if (format == DXGI_FORMAT_R8G8_B8G8_UNORM) { d[icount] = store_ubn_biased(v0[0]); d[icount + 1] = store_ubn_biased(v0[1]); d[icount + 2] = store_ubn_biased(v0[2]); d[icount + 3] = store_ubn_biased(g1); }
            else { d[icount] = store_ubn_biased(v0[1]); d[icount + 1] = store_ubn_biased(v0[0]); d[icount + 2] = store_ubn_biased(g1); d[icount + 3] = store_ubn_biased(v0[2]); }
        }
        return true;
    }
    case DXGI_FORMAT_YUY2:                      // :2274-2308 (XMStoreUByteN4 on the raw vectors: no bias)
    {
        if (size < 4) return false;
        auto ubn = [](float v) { const float t = clampf(v, 0.f, 1.f); return int(uint32_t(t * 255.0f)); };
        size_t i = 0;
        for (size_t icount = 0; icount < (size - 4 + 1); icount += 4)
        {
            if (i >= count) break;
            int rgb[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
            for (int c = 0; c < 3; ++c) rgb[0][c] = ubn(pSource[i].f[c]);
            ++i;
            if (i < count) { for (int c = 0; c < 3; ++c) rgb[1][c] = ubn(pSource[i].f[c]); ++i; }
            int y[2], u[2], v[2];
            for (int k = 0; k < 2; ++k)
            {
                y[k] = ((66 * rgb[k][0] + 129 * rgb[k][1] + 25 * rgb[k][2] + 128) >> 8) + 16;
                u[k] = ((-38 * rgb[k][0] - 74 * rgb[k][1] + 112 * rgb[k][2] + 128) >> 8) + 128;
                v[k] = ((112 * rgb[k][0] - 94 * rgb[k][1] - 18 * rgb[k][2] + 128) >> 8) + 128;
            }
            d[icount] = uint8_t(std::min<int>(std::max<int>(y[0], 0), 255)); d[icount + 1] = uint8_t(std::min<int>(std::max<int>((u[0] + u[1]) >> 1, 0), 255));
            d[icount + 2] = uint8_t(std::min<int>(std::max<int>(y[1], 0), 255)); d[icount + 3] = uint8_t(std::min<int>(std::max<int>((v[0] + v[1]) >> 1, 0), 255));
        }
        return true;
    }
    case DXGI_FORMAT_Y210:                      // :2310-2353 (XMStoreUDecN4: saturate, * 1023, truncate)
    case DXGI_FORMAT_Y216:                      // :2355-2397 (XMStoreUShortN4)
    {
        if (size < 8) return false;
        const bool ten = format == DXGI_FORMAT_Y210;
        auto q = [&](float v) -> int64_t { return ten ? int64_t(uint32_t(clampf(v, 0.f, 1.f) * 1023.0f)) : int64_t(store_usn(v)); };
        size_t i = 0;
        for (size_t icount = 0; icount < (size - 8 + 1); icount += 8)
        {
            if (i >= count) break;
            int64_t rgb[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } };
            for (int c = 0; c < 3; ++c) rgb[0][c] = q(pSource[i].f[c]);
            ++i;
            if (i < count) { for (int c = 0; c < 3; ++c) rgb[1][c] = q(pSource[i].f[c]); ++i; }
            int y[2], u[2], v[2];
            for (int k = 0; k < 2; ++k)
            {
                const int64_t r = rgb[k][0], g = rgb[k][1], b = rgb[k][2];
                if (ten) { y[k] = int((16780 * r + 32942 * g + 6544 * b + 32768) >> 16) + 64; u[k] = int((-9683 * r - 19017 * g + 28700 * b + 32768) >> 16) + 512; v[k] = int((28700 * r - 24033 * g - 4667 * b + 32768) >> 16) + 512; }
                else { y[k] = int((16763 * r + 32910 * g + 6537 * b + 32768) >> 16) + 4096; u[k] = int((-9674 * r - 18998 * g + 28672 * b + 32768) >> 16) + 32768; v[k] = int((28672 * r - 24010 * g - 4662 * b + 32768) >> 16) + 32768; }
            }
            const int top = ten ? 1023 : 65535, sh = ten ? 6 : 0;
            uint16_t h[4] = { uint16_t(std::min<int>(std::max<int>(y[0], 0), top) << sh), uint16_t(std::min<int>(std::max<int>((u[0] + u[1]) >> 1, 0), top) << sh),
                              uint16_t(std::min<int>(std::max<int>(y[1], 0), top) << sh), uint16_t(std::min<int>(std::max<int>((v[0] + v[1]) >> 1, 0), top) << sh) };
            memcpy(d + icount, h, 8);
        }
        return true;
    }
    case DXGI_FORMAT_R32G32B32A32_FLOAT:
        memcpy(d, pSource, texels(16) * 16);
        return true;
    case DXGI_FORMAT_R16G16B16A16_FLOAT:        // clamp to +-65504, XMStoreHalf4 (:1689-1702)
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 8);
            for (int c = 0; c < 4; ++c) h[c] = store_half(pSource[i].f[c]);
        }
        return true;
    case DXGI_FORMAT_R16G16B16A16_UNORM:
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 8);
            for (int c = 0; c < 4; ++c) h[c] = store_usn(pSource[i].f[c]);
        }
        return true;
    case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB:      // :1759-1772
        for (size_t i = 0, n = texels(4); i < n; ++i)
            for (int c = 0; c < 4; ++c) d[i * 4 + c] = store_ubn_biased(pSource[i].f[c]);
        return true;
    case DXGI_FORMAT_R8G8B8A8_SNORM:
        for (size_t i = 0, n = texels(4); i < n; ++i)
            for (int c = 0; c < 4; ++c) d[i * 4 + c] = uint8_t(store_bn(pSource[i].f[c]));
        return true;
    case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB:      // :2141-2155
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            d[i * 4 + 0] = store_ubn_biased(pSource[i].f[2]); d[i * 4 + 1] = store_ubn_biased(pSource[i].f[1]);
            d[i * 4 + 2] = store_ubn_biased(pSource[i].f[0]); d[i * 4 + 3] = store_ubn_biased(pSource[i].f[3]);
        }
        return true;
    case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB:      // :2157-2171
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            d[i * 4 + 0] = store_ubn_biased(pSource[i].f[2]); d[i * 4 + 1] = store_ubn_biased(pSource[i].f[1]);
            d[i * 4 + 2] = store_ubn_biased(pSource[i].f[0]); d[i * 4 + 3] = store_ubn_biased(1.0f);
        }
        return true;
    case DXGI_FORMAT_R32G32_FLOAT:
        for (size_t i = 0, n = texels(8); i < n; ++i) { float* f = reinterpret_cast<float*>(d + i * 8); f[0] = pSource[i].f[0]; f[1] = pSource[i].f[1]; }
        return true;
    case DXGI_FORMAT_R16G16_FLOAT:              // :1783-1796
        for (size_t i = 0, n = texels(4); i < n; ++i) { uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 4); h[0] = store_half(pSource[i].f[0]); h[1] = store_half(pSource[i].f[1]); }
        return true;
    case DXGI_FORMAT_R16G16_UNORM:
        for (size_t i = 0, n = texels(4); i < n; ++i) { uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 4); h[0] = store_usn(pSource[i].f[0]); h[1] = store_usn(pSource[i].f[1]); }
        return true;
    case DXGI_FORMAT_D32_FLOAT:                 // :1810
    case DXGI_FORMAT_R32_FLOAT:                 // :1811-1823
        for (size_t i = 0, n = texels(4); i < n; ++i) reinterpret_cast<float*>(d)[i] = pSource[i].f[0];
        return true;
    case DXGI_FORMAT_D32_FLOAT_S8X24_UINT:      // :1725-1744
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            reinterpret_cast<float*>(d)[i * 2] = pSource[i].f[0];
            d[i * 8 + 4] = static_cast<uint8_t>(std::min<float>(255.f, std::max<float>(0.f, pSource[i].f[1])));
            d[i * 8 + 5] = d[i * 8 + 6] = d[i * 8 + 7] = 0;
        }
        return true;
    case DXGI_FORMAT_D24_UNORM_S8_UINT:         // :1852-1869: XMVectorClamp(v, 0, (1, 255, 0, 0))
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const float fx = clampf(pSource[i].f[0], 0.f, 1.f), fy = clampf(pSource[i].f[1], 0.f, 255.f);
            reinterpret_cast<uint32_t*>(d)[i] = (static_cast<uint32_t>(fx * 16777215.f) & 0xFFFFFF) | ((static_cast<uint32_t>(fy) & 0xFF) << 24);
        }
        return true;
    case DXGI_FORMAT_R8G8_UNORM:                // XMStoreUByteN2, :1870-1871
        for (size_t i = 0, n = texels(2); i < n; ++i) { d[i * 2] = store_ubn2(pSource[i].f[0]); d[i * 2 + 1] = store_ubn2(pSource[i].f[1]); }
        return true;
    case DXGI_FORMAT_R8G8_SNORM:                // XMStoreByteN2, :1876-1877
        for (size_t i = 0, n = texels(2); i < n; ++i) { d[i * 2] = uint8_t(store_bn(pSource[i].f[0])); d[i * 2 + 1] = uint8_t(store_bn(pSource[i].f[1])); }
        return true;
    case DXGI_FORMAT_R16_FLOAT:                 // :1882-1896
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            float v = pSource[i].f[0];
            v = std::max<float>(std::min<float>(v, 65504.f), -65504.f);
            reinterpret_cast<uint16_t*>(d)[i] = PackedVector::XMConvertFloatToHalf(v);
        }
        return true;
    case DXGI_FORMAT_D16_UNORM:                 // :1897
    case DXGI_FORMAT_R16_UNORM:                 // :1898-1912
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            float v = pSource[i].f[0];
            v = std::max<float>(std::min<float>(v, 1.f), 0.f);
            reinterpret_cast<uint16_t*>(d)[i] = static_cast<uint16_t>(v * 65535.f + 0.5f);
        }
        return true;
    case DXGI_FORMAT_R8_UNORM:                  // :1958-1971
        for (size_t i = 0, n = texels(1); i < n; ++i)
        {
            float v = pSource[i].f[0] + (0.5f / 255.f);
            v = std::max<float>(std::min<float>(v, 1.f), 0.f);
            d[i] = static_cast<uint8_t>(v * 255.f);
        }
        return true;
    case DXGI_FORMAT_R8_SNORM:                  // :1988-2001
        for (size_t i = 0, n = texels(1); i < n; ++i)
        {
            float v = pSource[i].f[0];
            v = std::max<float>(std::min<float>(v, 1.f), -1.f);
            d[i] = uint8_t(static_cast<int8_t>(lroundf(v * 127.f)));
        }
        return true;
    case DXGI_FORMAT_A8_UNORM:                  // :2018-2031
        for (size_t i = 0, n = texels(1); i < n; ++i)
        {
            float v = pSource[i].f[3] + (0.5f / 255.f);
            v = std::max<float>(std::min<float>(v, 1.f), 0.f);
            d[i] = static_cast<uint8_t>(v * 255.f);
        }
        return true;
    case DXGI_FORMAT_R32G32B32_FLOAT:           // XMStoreFloat3, :1680-1681
        for (size_t i = 0, n = texels(12); i < n; ++i) memcpy(d + i * 12, pSource[i].f, 12);
        return true;
    case DXGI_FORMAT_R16G16B16A16_SNORM:        // XMStoreShortN4, :1710-1711
        for (size_t i = 0, n = texels(8); i < n; ++i)
        {
            uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 8);
            for (int c = 0; c < 4; ++c) h[c] = store_snorm16(pSource[i].f[c]);
        }
        return true;
    case DXGI_FORMAT_R16G16_SNORM:              // XMStoreShortN2, :1804-1805
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            uint16_t* h = reinterpret_cast<uint16_t*>(d + i * 4);
            h[0] = store_snorm16(pSource[i].f[0]); h[1] = store_snorm16(pSource[i].f[1]);
        }
        return true;
    case DXGI_FORMAT_R16_SNORM:                 // :1928-1941
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            float v = pSource[i].f[0];
            v = std::max<float>(std::min<float>(v, 1.f), -1.f);
            reinterpret_cast<int16_t*>(d)[i] = static_cast<int16_t>(lroundf(v * 32767.f));
        }
        return true;
    case DXGI_FORMAT_R10G10B10A2_UNORM:         // XMStoreUDecN4 (saturate, scale, truncate), :1747-1748
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const float* v = pSource[i].f;
            reinterpret_cast<uint32_t*>(d)[i] = (uint32_t(clampf(v[0], 0.f, 1.f) * 1023.0f) & 0x3FF) | ((uint32_t(clampf(v[1], 0.f, 1.f) * 1023.0f) & 0x3FF) << 10) |
                                                ((uint32_t(clampf(v[2], 0.f, 1.f) * 1023.0f) & 0x3FF) << 20) | ((uint32_t(clampf(v[3], 0.f, 1.f) * 3.0f) & 0x3) << 30);
        }
        return true;
    case DXGI_FORMAT_R11G11B10_FLOAT:           // XMStoreFloat3PK, :1756-1757
        for (size_t i = 0, n = texels(4); i < n; ++i)
            reinterpret_cast<uint32_t*>(d)[i] = store_small_float(pSource[i].f[0], 6) | (store_small_float(pSource[i].f[1], 6) << 11) | (store_small_float(pSource[i].f[2], 5) << 22);
        return true;
    case DXGI_FORMAT_R9G9B9E5_SHAREDEXP:        // StoreFloat3SE, :158-191 / :2057-2058
        for (size_t i = 0, n = texels(4); i < n; ++i)
        {
            const float* t = pSource[i].f;
            constexpr float maxf9 = float(0x1FF << 7);
            constexpr float minf9 = float(1.f / (1 << 16));
            float x = (t[0] >= 0.f) ? ((t[0] > maxf9) ? maxf9 : t[0]) : 0.f;
            float y = (t[1] >= 0.f) ? ((t[1] > maxf9) ? maxf9 : t[1]) : 0.f;
            float z = (t[2] >= 0.f) ? ((t[2] > maxf9) ? maxf9 : t[2]) : 0.f;
            const float max_xy = (x > y) ? x : y;
            const float max_xyz = (max_xy > z) ? max_xy : z;
            const float maxColor = (max_xyz > minf9) ? max_xyz : minf9;
            union { float f; int32_t i; } fi;
            fi.f = maxColor;
            fi.i += 0x00004000;
            const uint32_t exp = uint32_t(fi.i) >> 23;
            const uint32_t e = (exp - 0x6f) & 0x1F;
            fi.i = int32_t(0x83000000u - (exp << 23));
            const float ScaleR = fi.f;
            reinterpret_cast<uint32_t*>(d)[i] = (uint32_t(lroundf(x * ScaleR)) & 0x1FF) | ((uint32_t(lroundf(y * ScaleR)) & 0x1FF) << 9) | ((uint32_t(lroundf(z * ScaleR)) & 0x1FF) << 18) | (e << 27);
        }
        return true;
    case DXGI_FORMAT_B5G6R5_UNORM:              // :2096-2114 (x64: multiply, XMStoreU565 rounds to nearest)
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const float* v = pSource[i].f;
            reinterpret_cast<uint16_t*>(d)[i] = uint16_t((store_scaled(v[2], 31.f) & 0x1F) | ((store_scaled(v[1], 63.f) & 0x3F) << 5) | ((store_scaled(v[0], 31.f) & 0x1F) << 11));
        }
        return true;
    case DXGI_FORMAT_B5G5R5A1_UNORM:            // :2116-2139
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const float* v = pSource[i].f;
            reinterpret_cast<uint16_t*>(d)[i] = uint16_t((store_scaled(v[2], 31.f) & 0x1F) | ((store_scaled(v[1], 31.f) & 0x1F) << 5) | ((store_scaled(v[0], 31.f) & 0x1F) << 10) |
                                                         ((v[3] > threshold) ? 0x8000u : 0u));
        }
        return true;
    case DXGI_FORMAT_B4G4R4A4_UNORM:            // :2399-2417
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const float* v = pSource[i].f;
            reinterpret_cast<uint16_t*>(d)[i] = uint16_t((store_scaled(v[2], 15.f) & 0xF) | ((store_scaled(v[1], 15.f) & 0xF) << 4) | ((store_scaled(v[0], 15.f) & 0xF) << 8) | ((store_scaled(v[3], 15.f) & 0xF) << 12));
        }
        return true;
    case WIN11_DXGI_FORMAT_A4B4G4R4_UNORM:      // :2419-2437: swizzle <3, 2, 1, 0>, * 15, XMStoreUNibble4
        for (size_t i = 0, n = texels(2); i < n; ++i)
        {
            const float sw[4] = { pSource[i].f[3], pSource[i].f[2], pSource[i].f[1], pSource[i].f[0] };
            reinterpret_cast<uint16_t*>(d)[i] = uint16_t((store_scaled(sw[0], 15.f) & 0xF) | ((store_scaled(sw[1], 15.f) & 0xF) << 4) | ((store_scaled(sw[2], 15.f) & 0xF) << 8) | ((store_scaled(sw[3], 15.f) & 0xF) << 12));
        }
        return true;
    default:
        return false;
    }
}

bool DirectX::Internal::StoreScanlineDither(void*, size_t, DXGI_FORMAT, XMVECTOR*, size_t, float, size_t, size_t, XMVECTOR*) noexcept
{
    return false;       // dithered stores are out of scope
}

// ---- LoadScanlineLinear / StoreScanlineLinear (:2803-2945) -----------------------------------------------------------------------
bool DirectX::Internal::LoadScanlineLinear(XMVECTOR* pDestination, size_t count, const void* pSource, size_t size, DXGI_FORMAT format, TEX_FILTER_FLAGS flags) noexcept
{
    uint32_t fl = uint32_t(flags);
    if (format == DXGI_FORMAT_R8G8B8A8_UNORM_SRGB || format == DXGI_FORMAT_B8G8R8A8_UNORM_SRGB || format == DXGI_FORMAT_B8G8R8X8_UNORM_SRGB) fl |= TEX_FILTER_SRGB;
    else if (!linear_filter_srgb_ok(format)) fl &= ~uint32_t(TEX_FILTER_SRGB);
    if (!LoadScanline(pDestination, count, pSource, size, format)) return false;
    if (fl & TEX_FILTER_SRGB_IN)
        for (size_t i = 0; i < count; ++i)
            for (int c = 0; c < 3; ++c) pDestination[i].f[c] = srgb_to_rgb(pDestination[i].f[c]);
    return true;
}

bool DirectX::Internal::StoreScanlineLinear(void* pDestination, size_t size, DXGI_FORMAT format, XMVECTOR* pSource, size_t count, TEX_FILTER_FLAGS flags, float threshold) noexcept
{
    if (!pSource || !count) return false;
    uint32_t fl = uint32_t(flags);
    if (format == DXGI_FORMAT_R8G8B8A8_UNORM_SRGB || format == DXGI_FORMAT_B8G8R8A8_UNORM_SRGB || format == DXGI_FORMAT_B8G8R8X8_UNORM_SRGB) fl |= TEX_FILTER_SRGB;
    else if (!linear_filter_srgb_ok(format)) fl &= ~uint32_t(TEX_FILTER_SRGB);
    if (fl & TEX_FILTER_SRGB_OUT)
        for (size_t i = 0; i < count; ++i)
            for (int c = 0; c < 3; ++c) pSource[i].f[c] = rgb_to_srgb(pSource[i].f[c]);
    return StoreScanline(pDestination, size, format, pSource, count, threshold);
}

// ---- ConvertScanline (:3080-3854) -----------------------------------------------------------------------------
void DirectX::Internal::ConvertScanline(XMVECTOR* pBuffer, size_t count, DXGI_FORMAT outFormat, DXGI_FORMAT inFormat, TEX_FILTER_FLAGS tflags) noexcept
{
    if (!pBuffer) return;
    const uint32_t in = conv_flags(inFormat), out = conv_flags(outFormat);
    if (!in || !out) return;
    uint32_t flags = uint32_t(tflags);
    if (is_srgb_format(inFormat)) flags |= TEX_FILTER_SRGB_IN;
    if (inFormat == DXGI_FORMAT_A8_UNORM || inFormat == DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM) flags &= ~uint32_t(TEX_FILTER_SRGB_IN);        // :3136-3139
    if (is_srgb_format(outFormat)) flags |= TEX_FILTER_SRGB_OUT;
    if (outFormat == DXGI_FORMAT_A8_UNORM || outFormat == DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM) flags &= ~uint32_t(TEX_FILTER_SRGB_OUT);     // :3156-3159
    if ((flags & TEX_FILTER_SRGB) == TEX_FILTER_SRGB) flags &= ~uint32_t(TEX_FILTER_SRGB);

    auto each = [&](auto&& fn) { for (size_t i = 0; i < count; ++i) fn(pBuffer[i].f); };

    if ((flags & TEX_FILTER_SRGB_IN) && !(in & C_DEPTH) && (in & (C_FLOAT | C_UNORM)))                                              // :3170-3180
        each([](float* v) { for (int c = 0; c < 3; ++c) v[c] = srgb_to_rgb(v[c]); });

    const uint32_t diff = in ^ out;
    if (diff != 0)
    {
        const bool x2bias = (flags & TEX_FILTER_FLOAT_X2BIAS) != 0;
        const uint32_t copyAny = flags & (TEX_FILTER_RGB_COPY_RED | TEX_FILTER_RGB_COPY_GREEN | TEX_FILTER_RGB_COPY_BLUE | TEX_FILTER_RGB_COPY_ALPHA);
        if (diff & C_DEPTH)
        {
            if (in & C_DEPTH)
            {
                // depth -> colour (:3189-3291): stencil to alpha, then depth to RGB
                if (in & C_STENCIL)
                {
                    if (out & C_UNORM) each([](float* v) { v[3] = clampf(v[1], 0.f, 255.f) / 255.f; });                              // :3196-3209
                    else if (out & C_SNORM) each([](float* v) { v[3] = (clampf(v[1], 0.f, 255.f) / 255.f) * 2.0f + -1.0f; });         // :3210-3224
                    else each([](float* v) { v[3] = v[1]; });                                                                     // :3225-3235
                }
                if ((out & C_UNORM) && (in & C_FLOAT)) each([](float* v) { v[0] = v[1] = v[2] = clampf(v[0], 0.f, 1.f); });          // :3239-3250
                else if (out & C_SNORM)
                {
                    if (in & C_UNORM) each([](float* v) { v[0] = v[1] = v[2] = v[0] * 2.0f + -1.0f; });                             // :3253-3265
                    else each([](float* v) { v[0] = v[1] = v[2] = clampf(v[0], -1.f, 1.f); });                                      // :3266-3278
                }
                else each([](float* v) { v[1] = v[2] = v[0]; });                                                                  // :3280-3290
            }
            else
            {
                // colour -> depth (:3293-3434): one channel to x, its range conversion, alpha to stencil
                if (copyAny == TEX_FILTER_RGB_COPY_GREEN) each([](float* v) { v[0] = v[1]; });
                else if (copyAny == TEX_FILTER_RGB_COPY_BLUE) each([](float* v) { v[0] = v[2]; });
                else if (copyAny == TEX_FILTER_RGB_COPY_ALPHA) each([](float* v) { v[0] = v[3]; });
                else if (copyAny != TEX_FILTER_RGB_COPY_RED && (in & C_UNORM) && ((in & (C_R | C_G | C_B)) == (C_R | C_G | C_B)))
                    each([](float* v) { v[0] = (v[0] * 0.2125f + v[1] * 0.7154f) + v[2] * 0.0721f; });                               // :3330-3343
                if (out & C_UNORM)
                {
                    if (in & C_SNORM) each([](float* v) { v[0] = v[0] * 0.5f + 0.5f; });                                             // :3368-3379
                    else if (in & C_FLOAT) each([](float* v) { v[0] = clampf(v[0], 0.f, 1.f); });                                    // :3380-3391
                }
                if (out & C_STENCIL)
                {
                    if (in & C_UNORM) each([](float* v) { v[1] = v[3] * 255.f; });                                                   // :3396-3408
                    else if (in & C_SNORM) each([](float* v) { v[1] = (v[3] * 0.5f + 0.5f) * 255.f; });                              // :3409-3422
                    else each([](float* v) { v[1] = v[3]; });                                                                      // :3423-3433
                }
            }
        }
        else if (out & C_DEPTH)
        {
            if ((diff & C_FLOAT) && (in & C_FLOAT)) each([](float* v) { v[0] = clampf(v[0], 0.f, 1.f); });                             // :3435-3451
        }
        else if (out & C_UNORM)
        {
            if (in & C_SNORM) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = v[c] * 0.5f + 0.5f; });                       // :3457-3463
            else if (in & C_FLOAT)
            {
                if (!(in & C_POS_ONLY) && x2bias) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], -1.f, 1.f) * 0.5f + 0.5f; });   // :3469-3477
                else each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], 0.f, 1.f); });                               // :3481-3486
            }
        }
        else if (out & C_SNORM)
        {
            if (in & C_UNORM) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = v[c] * 2.0f + -1.0f; });                       // :3495-3501
            else if (in & C_FLOAT)
            {
                if ((in & C_POS_ONLY) && x2bias) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], 0.f, 1.f) * 2.0f + -1.0f; });    // :3506-3515
                else each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], -1.f, 1.f); });                             // :3519-3526
            }
        }
        else if (diff & C_UNORM)
        {
            if ((out & C_FLOAT) && !(out & C_POS_ONLY) && x2bias)
                each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = v[c] * 2.0f + -1.0f; });                                     // :3533-3542
        }
        else if ((diff & C_POS_ONLY) && x2bias)
        {
            // :3545-3587
            if (in & C_POS_ONLY)
            {
                if (out & C_FLOAT) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], 0.f, 1.f) * 2.0f + -1.0f; });
            }
            else if (out & C_POS_ONLY)
            {
                if (in & C_FLOAT) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = clampf(v[c], -1.f, 1.f) * 0.5f + 0.5f; });
                else if (in & C_SNORM) each([](float* v) { for (int c = 0; c < 4; ++c) v[c] = v[c] * 0.5f + 0.5f; });
            }
        }

        const uint32_t RGBA = C_R | C_G | C_B | C_A, RGB = C_R | C_G | C_B;
        auto gray = [](const float* v) { return (v[0] * 0.2125f + v[1] * 0.7154f) + v[2] * 0.0721f; };         // XMVector3Dot(v, g_Grayscale)
        const uint32_t copy3 = flags & (TEX_FILTER_RGB_COPY_RED | TEX_FILTER_RGB_COPY_GREEN | TEX_FILTER_RGB_COPY_BLUE);
        const uint32_t copy4 = flags & (TEX_FILTER_RGB_COPY_RED | TEX_FILTER_RGB_COPY_GREEN | TEX_FILTER_RGB_COPY_BLUE | TEX_FILTER_RGB_COPY_ALPHA);
        if (((out & RGBA) == C_A) && !(in & C_A))
        {
            // :3596-3652
            if (copy3 == TEX_FILTER_RGB_COPY_GREEN) each([](float* v) { v[0] = v[2] = v[3] = v[1]; });
            else if (copy3 == TEX_FILTER_RGB_COPY_BLUE) each([](float* v) { v[0] = v[1] = v[3] = v[2]; });
            else if (copy3 != TEX_FILTER_RGB_COPY_RED && (in & C_UNORM) && ((in & RGB) == RGB)) each([&](float* v) { const float g = gray(v); v[0] = v[1] = v[2] = v[3] = g; });
            else each([](float* v) { v[1] = v[2] = v[3] = v[0]; });
        }
        else if (((in & RGBA) == C_A) && !(out & C_A)) each([](float* v) { v[0] = v[1] = v[2] = v[3]; });                          // :3654-3664
        else if ((in & RGB) == C_R)
        {
            if ((out & RGB) == RGB) each([](float* v) { v[1] = v[2] = v[0]; });                                                     // :3667-3679
            else if ((out & RGB) == (C_R | C_G)) each([](float* v) { v[1] = v[0]; });                                               // :3680-3691
        }
        else if ((in & RGB) == RGB)
        {
            if ((out & RGB) == C_R)
            {
                // :3696-3771
                if (copy4 == TEX_FILTER_RGB_COPY_GREEN) each([](float* v) { v[0] = v[2] = v[1]; });
                else if (copy4 == TEX_FILTER_RGB_COPY_BLUE) each([](float* v) { v[0] = v[1] = v[2]; });
                else if (copy4 == TEX_FILTER_RGB_COPY_ALPHA) each([](float* v) { v[0] = v[1] = v[2] = v[3]; });
                else if (copy4 != TEX_FILTER_RGB_COPY_RED && (in & C_UNORM)) each([&](float* v) { const float g = gray(v); v[0] = v[1] = v[2] = g; });
            }
            else if ((out & RGB) == (C_R | C_G))
            {
                // :3773-3838
                if ((flags & TEX_FILTER_RGB_COPY_ALPHA) && (in & C_A))
                {
                    if (copy4 == (TEX_FILTER_RGB_COPY_GREEN | TEX_FILTER_RGB_COPY_ALPHA)) each([](float* v) { v[0] = v[1]; v[1] = v[3]; });
                    else if (copy4 == (TEX_FILTER_RGB_COPY_BLUE | TEX_FILTER_RGB_COPY_ALPHA)) each([](float* v) { v[0] = v[2]; v[1] = v[3]; });
                    else each([](float* v) { v[1] = v[3]; });
                }
                else
                {
                    if (copy3 == (TEX_FILTER_RGB_COPY_RED | TEX_FILTER_RGB_COPY_BLUE)) each([](float* v) { v[1] = v[2]; });
                    else if (copy3 == (TEX_FILTER_RGB_COPY_GREEN | TEX_FILTER_RGB_COPY_BLUE)) each([](float* v) { v[0] = v[1]; v[1] = v[2]; });
                }
            }
        }
    }

    if ((flags & TEX_FILTER_SRGB_OUT) && !(out & C_DEPTH) && (out & (C_FLOAT | C_UNORM)))                                           // :3843-3853
        each([](float* v) { for (int c = 0; c < 3; ++c) v[c] = rgb_to_srgb(v[c]); });
}

// ---- the three scanline helpers the DDS reader links against (DirectXTexConvert.cpp:207-735) ---------------------------------
// Integer bit manipulation only (no DirectXMath), restated texel by texel with the reference's own masks and shifts so that
// DirectXTexDDS.cpp - compiled in place - behaves as it does in the reference. The product's reader
// (directxtex_amd/host/DirectXTexAMD_DDS.cpp) is written differently (generic bit replication) and is checked against this.
namespace
{
    template<typename T> inline T peek(const void* p, size_t byteOffset) { T v; memcpy(&v, static_cast<const uint8_t*>(p) + byteOffset, sizeof(T)); return v; }
    template<typename T> inline void poke(void* p, size_t byteOffset, T v) { memcpy(static_cast<uint8_t*>(p) + byteOffset, &v, sizeof(T)); }
}

// DirectXTexConvert.cpp:207-430. With TEXP_SCANLINE_SETALPHA the alpha channel of the listed formats is overwritten with
// "opaque"; everything else is a memcpy of min(outSize, inSize) bytes (nothing at all when used in place).
void DirectX::Internal::CopyScanline(void* pDestination, size_t outSize, const void* pSource, size_t inSize, DXGI_FORMAT format, uint32_t tflags) noexcept
{
    const size_t n = outSize < inSize ? outSize : inSize;
    if (tflags & TEXP_SCANLINE_SETALPHA)
    {
        switch (int(format))
        {
        case DXGI_FORMAT_R32G32B32A32_TYPELESS: case DXGI_FORMAT_R32G32B32A32_FLOAT: case DXGI_FORMAT_R32G32B32A32_UINT: case DXGI_FORMAT_R32G32B32A32_SINT:   // :223-260
            if (inSize >= 16 && outSize >= 16)
            {
                const uint32_t alpha = (format == DXGI_FORMAT_R32G32B32A32_FLOAT) ? 0x3f800000u : (format == DXGI_FORMAT_R32G32B32A32_SINT) ? 0x7fffffffu : 0xffffffffu;
                for (size_t at = 0; at < n - 15; at += 16)
                {
                    for (size_t c = 0; c < 12; c += 4) poke<uint32_t>(pDestination, at + c, peek<uint32_t>(pSource, at + c));
                    poke<uint32_t>(pDestination, at + 12, alpha);
                }
            }
            return;
        case DXGI_FORMAT_R16G16B16A16_TYPELESS: case DXGI_FORMAT_R16G16B16A16_FLOAT: case DXGI_FORMAT_R16G16B16A16_UNORM: case DXGI_FORMAT_R16G16B16A16_UINT:   // :263-304
        case DXGI_FORMAT_R16G16B16A16_SNORM: case DXGI_FORMAT_R16G16B16A16_SINT: case DXGI_FORMAT_Y416:
            if (inSize >= 8 && outSize >= 8)
            {
                const uint16_t alpha = (format == DXGI_FORMAT_R16G16B16A16_FLOAT) ? uint16_t(0x3c00)
                                     : (format == DXGI_FORMAT_R16G16B16A16_SNORM || format == DXGI_FORMAT_R16G16B16A16_SINT) ? uint16_t(0x7fff) : uint16_t(0xffff);
                for (size_t at = 0; at < n - 7; at += 8)
                {
                    for (size_t c = 0; c < 6; c += 2) poke<uint16_t>(pDestination, at + c, peek<uint16_t>(pSource, at + c));
                    poke<uint16_t>(pDestination, at + 6, alpha);
                }
            }
            return;
        case DXGI_FORMAT_R10G10B10A2_TYPELESS: case DXGI_FORMAT_R10G10B10A2_UNORM: case DXGI_FORMAT_R10G10B10A2_UINT: case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM:   // :307-339
        case DXGI_FORMAT_Y410: case XBOX_DXGI_FORMAT_R10G10B10_7E3_A2_FLOAT: case XBOX_DXGI_FORMAT_R10G10B10_6E4_A2_FLOAT: case XBOX_DXGI_FORMAT_R10G10B10_SNORM_A2_UNORM:
            if (inSize >= 4 && outSize >= 4)
                for (size_t at = 0; at < n - 3; at += 4) poke<uint32_t>(pDestination, at, peek<uint32_t>(pSource, at) | 0xC0000000u);
            return;
        case DXGI_FORMAT_R8G8B8A8_TYPELESS: case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: case DXGI_FORMAT_R8G8B8A8_UINT:   // :342-380
        case DXGI_FORMAT_R8G8B8A8_SNORM: case DXGI_FORMAT_R8G8B8A8_SINT: case DXGI_FORMAT_B8G8R8A8_UNORM: case DXGI_FORMAT_B8G8R8A8_TYPELESS:
        case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: case DXGI_FORMAT_AYUV:
            if (inSize >= 4 && outSize >= 4)
            {
                const uint32_t alpha = (format == DXGI_FORMAT_R8G8B8A8_SNORM || format == DXGI_FORMAT_R8G8B8A8_SINT) ? 0x7f000000u : 0xff000000u;
                for (size_t at = 0; at < n - 3; at += 4) poke<uint32_t>(pDestination, at, (peek<uint32_t>(pSource, at) & 0xFFFFFFu) | alpha);
            }
            return;
        case DXGI_FORMAT_B5G5R5A1_UNORM: case DXGI_FORMAT_B4G4R4A4_UNORM: case WIN11_DXGI_FORMAT_A4B4G4R4_UNORM:   // :383-415
            if (inSize >= 2 && outSize >= 2)
            {
                const uint16_t alpha = (format == DXGI_FORMAT_B4G4R4A4_UNORM) ? uint16_t(0xF000) : (format == WIN11_DXGI_FORMAT_A4B4G4R4_UNORM) ? uint16_t(0x000F) : uint16_t(0x8000);
                for (size_t at = 0; at < n - 1; at += 2) poke<uint16_t>(pDestination, at, uint16_t(peek<uint16_t>(pSource, at) | alpha));
            }
            return;
        case DXGI_FORMAT_A8_UNORM:   // :418-420
            memset(pDestination, 0xff, outSize);
            return;
        default:
            break;
        }
    }
    if (pDestination != pSource) memcpy(pDestination, pSource, n);   // :426-430
}

// DirectXTexConvert.cpp:440-605. Red <-> blue for the 10:10:10:2 family (only for legacy sources) and the 8:8:8:8 family,
// UYVY -> YUY2 for legacy sources; anything else is copied.
void DirectX::Internal::SwizzleScanline(void* pDestination, size_t outSize, const void* pSource, size_t inSize, DXGI_FORMAT format, uint32_t tflags) noexcept
{
    const size_t n = outSize < inSize ? outSize : inSize;
    const bool opaque = (tflags & TEXP_SCANLINE_SETALPHA) != 0, legacy = (tflags & TEXP_SCANLINE_LEGACY) != 0;
    switch (int(format))
    {
    case DXGI_FORMAT_R10G10B10A2_TYPELESS: case DXGI_FORMAT_R10G10B10A2_UNORM: case DXGI_FORMAT_R10G10B10A2_UINT: case DXGI_FORMAT_R10G10B10_XR_BIAS_A2_UNORM:   // :455-497
    case XBOX_DXGI_FORMAT_R10G10B10_SNORM_A2_UNORM:
        if (inSize >= 4 && outSize >= 4 && legacy)
        {
            for (size_t at = 0; at < n - 3; at += 4)
            {
                const uint32_t t = peek<uint32_t>(pSource, at);
                const uint32_t t1 = (t & 0x3ff00000u) >> 20, t2 = (t & 0x000003ffu) << 20, t3 = t & 0x000ffc00u;
                poke<uint32_t>(pDestination, at, t1 | t2 | t3 | (opaque ? 0xC0000000u : (t & 0xC0000000u)));
            }
            return;
        }
        break;
    case DXGI_FORMAT_R8G8B8A8_TYPELESS: case DXGI_FORMAT_R8G8B8A8_UNORM: case DXGI_FORMAT_R8G8B8A8_UNORM_SRGB: case DXGI_FORMAT_B8G8R8A8_UNORM:   // :500-545
    case DXGI_FORMAT_B8G8R8X8_UNORM: case DXGI_FORMAT_B8G8R8A8_TYPELESS: case DXGI_FORMAT_B8G8R8A8_UNORM_SRGB: case DXGI_FORMAT_B8G8R8X8_TYPELESS:
    case DXGI_FORMAT_B8G8R8X8_UNORM_SRGB:
        if (inSize >= 4 && outSize >= 4)
        {
            for (size_t at = 0; at < n - 3; at += 4)
            {
                const uint32_t t = peek<uint32_t>(pSource, at);
                const uint32_t t1 = (t & 0x00ff0000u) >> 16, t2 = (t & 0x000000ffu) << 16, t3 = t & 0x0000ff00u;
                poke<uint32_t>(pDestination, at, t1 | t2 | t3 | (opaque ? 0xff000000u : (t & 0xFF000000u)));
            }
            return;
        }
        break;
    case DXGI_FORMAT_YUY2:   // :548-594
        if (inSize >= 4 && outSize >= 4 && legacy)
        {
            for (size_t at = 0; at < n - 3; at += 4)
            {
                const uint32_t t = peek<uint32_t>(pSource, at);
                poke<uint32_t>(pDestination, at, ((t & 0x000000ffu) << 8) | ((t & 0x0000ff00u) >> 8) | ((t & 0x00ff0000u) << 8) | ((t & 0xff000000u) >> 8));
            }
            return;
        }
        break;
    default:
        break;
    }
    if (pDestination != pSource) memcpy(pDestination, pSource, n);   // :600-605
}

// DirectXTexConvert.cpp:613-735: the 16-bit formats widened to R8G8B8A8_UNORM (the only target), bits replicated downwards.
bool DirectX::Internal::ExpandScanline(void* pDestination, size_t outSize, DXGI_FORMAT outFormat, const void* pSource, size_t inSize, DXGI_FORMAT inFormat, uint32_t tflags) noexcept
{
    const bool opaque = (tflags & TEXP_SCANLINE_SETALPHA) != 0;
    const int in = int(inFormat);
    if (in != DXGI_FORMAT_B5G6R5_UNORM && in != DXGI_FORMAT_B5G5R5A1_UNORM && in != DXGI_FORMAT_B4G4R4A4_UNORM && in != int(WIN11_DXGI_FORMAT_A4B4G4R4_UNORM)) return false;
    if (outFormat != DXGI_FORMAT_R8G8B8A8_UNORM) return false;
    if (!(inSize >= 2 && outSize >= 4)) return false;
    for (size_t ocount = 0, icount = 0; (icount < (inSize - 1)) && (ocount < (outSize - 3)); icount += 2, ocount += 4)
    {
        const uint32_t t = peek<uint16_t>(pSource, icount);
        uint32_t t1, t2, t3, ta;
        if (in == DXGI_FORMAT_B5G6R5_UNORM)   // :631-652
        {
            t1 = ((t & 0xf800u) >> 8) | ((t & 0xe000u) >> 13);
            t2 = ((t & 0x07e0u) << 5) | ((t & 0x0600u) >> 5);
            t3 = ((t & 0x001fu) << 19) | ((t & 0x001cu) << 14);
            ta = 0xff000000u;
        }
        else if (in == DXGI_FORMAT_B5G5R5A1_UNORM)   // :654-677
        {
            t1 = ((t & 0x7c00u) >> 7) | ((t & 0x7000u) >> 12);
            t2 = ((t & 0x03e0u) << 6) | ((t & 0x0380u) << 1);
            t3 = ((t & 0x001fu) << 19) | ((t & 0x001cu) << 14);
            ta = opaque ? 0xff000000u : ((t & 0x8000u) ? 0xff000000u : 0u);
        }
        else if (in == DXGI_FORMAT_B4G4R4A4_UNORM)   // :679-702
        {
            t1 = ((t & 0x0f00u) >> 4) | ((t & 0x0f00u) >> 8);
            t2 = ((t & 0x00f0u) << 8) | ((t & 0x00f0u) << 4);
            t3 = ((t & 0x000fu) << 20) | ((t & 0x000fu) << 16);
            ta = opaque ? 0xff000000u : (((t & 0xf000u) << 16) | ((t & 0xf000u) << 12));
        }
        else   // A4B4G4R4, :704-727
        {
            t1 = ((t & 0xf000u) >> 8) | ((t & 0xf000u) >> 12);
            t2 = (t & 0x0f00u) | ((t & 0x0f00u) << 4);
            t3 = ((t & 0x00f0u) << 16) | ((t & 0x00f0u) << 12);
            ta = opaque ? 0xff000000u : (((t & 0x000fu) << 28) | ((t & 0x000fu) << 24));
        }
        poke<uint32_t>(pDestination, ocount, t1 | t2 | t3 | ta);
    }
    return true;
}
