// Radiance RGBE (.hdr) reader / writer of the host layer - the usual source of BC6H textures. Behaviour follows
// DirectXTexHDR.cpp: the header grammar and its error codes (DecodeHDRHeader, :60-322), both run-length schemes on input
// (:700-860), the exposure-scaled float conversion (:862-875), and on output FloatToRGBE / HalfToRGBE (:324-392) with the
// per-channel run-length encoding of EncodeRLE (:394-588). Files come out byte-identical to the reference's, pixels bit-identical
// (tests/test_hdr_tga_cpu.py checks both against oracle/_ref, including damaged and mutated files). Host code: a container
// either side of the GPU path.
#include "DirectXTexAMD.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

namespace DirectXTexAMD
{
namespace
{
    const char kSignature[] = "#?RADIANCE", kAltSignature[] = "#?RGBE", kFormat[] = "FORMAT=", kExposure[] = "EXPOSURE=";
    const char kRGBE[] = "32-bit_rle_rgbe", kXYZE[] = "32-bit_rle_xyze";
    constexpr size_t kNone = size_t(-1);

    // Where the line ends: index of '\n' within the first maxlen characters; kNone if a NUL comes first; 0 if there is neither -
    // which callers cannot tell from an empty line, and treat as an error (FindEOL, DirectXTexHDR.cpp:42-57)
    size_t LineLength(const char* s, size_t maxlen) noexcept
    {
        for (size_t i = 0; i < maxlen; ++i)
        {
            if (s[i] == '\n') return i;
            if (s[i] == '\0') return kNone;
        }
        return 0;
    }

    HRESULT DecodeHeader(const uint8_t* pSource, size_t size, TexMetadata& metadata, size_t& offset, float& exposure) noexcept
    {
        if (!pSource) return E_POINTER;
        metadata = TexMetadata();
        metadata.dimension = TEX_DIMENSION(0);
        exposure = 1.f;
        if (size < sizeof(kSignature)) return HRESULT_E_INVALID_DATA;
        if (std::memcmp(pSource, kSignature, sizeof(kSignature) - 1) != 0 && std::memcmp(pSource, kAltSignature, sizeof(kAltSignature) - 1) != 0) return E_FAIL;

        // header lines up to the blank one; FORMAT= is required, EXPOSURE= lines multiply up
        bool formatFound = false;
        const char* info = reinterpret_cast<const char*>(pSource);
        auto skipBlanks = [&]() -> bool
        {
            while (*info == ' ' || *info == '\t')
            {
                if (--size == 0) return false;
                ++info;
            }
            return true;
        };
        auto nextLine = [&]() -> bool
        {
            const size_t len = LineLength(info, size);
            if (len == kNone || len < 1) return false;
            info += len + 1; size -= len + 1;
            return true;
        };
        while (size > 0)
        {
            if (*info == '\n') { ++info; --size; break; }
            constexpr size_t formatLen = sizeof(kFormat) - 1, exposureLen = sizeof(kExposure) - 1, encodingLen = sizeof(kRGBE) - 1;
            if (size > formatLen && std::memcmp(info, kFormat, formatLen) == 0)
            {
                info += formatLen; size -= formatLen;
                if (!skipBlanks()) return E_FAIL;
                if (size < encodingLen) return E_FAIL;
                if (std::memcmp(info, kRGBE, encodingLen) != 0 && std::memcmp(info, kXYZE, encodingLen) != 0) return HRESULT_E_NOT_SUPPORTED;
                formatFound = true;
                if (!nextLine()) return E_FAIL;
            }
            else if (size > exposureLen && std::memcmp(info, kExposure, exposureLen) == 0)
            {
                info += exposureLen; size -= exposureLen;
                if (!skipBlanks()) return E_FAIL;
                const size_t len = LineLength(info, size);
                if (len == kNone || len < 1) return E_FAIL;
                char number[32] = {};
                std::memcpy(number, info, std::min<size_t>(31, len));
                const float e = float(std::atof(number));
                if (e >= 1e-12f && e <= 1e12f) exposure *= e;
                info += len + 1; size -= len + 1;
            }
            else if (!nextLine()) return E_FAIL;
        }
        if (!formatFound || size < 3) return E_FAIL;

        // the resolution line: only "-Y <height> +X <width>" (top to bottom, left to right) is read
        char orientation[256] = {};
        const size_t len = LineLength(info, std::min<size_t>(sizeof(orientation) - 1, size));
        if (len == kNone || len <= 2) return E_FAIL;
        std::memcpy(orientation, info, len);
        if (orientation[0] != '-' || orientation[1] != 'Y')
            return ((orientation[0] == '+' || orientation[0] == '-') && (orientation[1] == 'X' || orientation[1] == 'Y')) ? HRESULT_E_NOT_SUPPORTED : HRESULT_E_INVALID_DATA;
        unsigned height = 0;
        if (std::sscanf(orientation + 2, "%u", &height) != 1) return E_FAIL;
        if (height > UINT16_MAX) return HRESULT_E_NOT_SUPPORTED;
        const char* p = orientation + 2;
        while (*p != 0 && *p != '-' && *p != '+') ++p;
        if (*p == 0) return E_FAIL;
        if (*p != '+') return HRESULT_E_NOT_SUPPORTED;
        ++p;
        if (*p == 0 || (*p != 'X' && *p != 'Y')) return E_FAIL;
        if (*p != 'X') return HRESULT_E_NOT_SUPPORTED;
        ++p;
        unsigned width = 0;
        if (std::sscanf(p, "%u", &width) != 1) return E_FAIL;
        if (width > UINT16_MAX) return HRESULT_E_NOT_SUPPORTED;
        info += len + 1; size -= len + 1;
        if (!width || !height) return HRESULT_E_INVALID_DATA;
        if (uint64_t(width) * uint64_t(height) * 16u > UINT32_MAX) return HRESULT_E_ARITHMETIC_OVERFLOW;
        if (size == 0) return E_FAIL;

        offset = size_t(info - reinterpret_cast<const char*>(pSource));
        metadata.width = width; metadata.height = height;
        metadata.depth = metadata.arraySize = metadata.mipLevels = 1;
        metadata.format = DXGI_FORMAT_R32G32B32A32_FLOAT;
        metadata.dimension = TEX_DIMENSION_TEXTURE2D;
        metadata.SetAlphaMode(TEX_ALPHA_MODE_OPAQUE);
        return S_OK;
    }

    inline float HalfToFloat(uint16_t h) noexcept
    {
        const uint32_t sign = uint32_t(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
        uint32_t bits;
        if (e == 0)
        {
            if (!m) bits = sign;
            else { int shift = 0; uint32_t mm = m; while (!(mm & 0x400)) { mm <<= 1; ++shift; } bits = sign | (uint32_t(113 - shift) << 23) | ((mm & 0x3ff) << 13); }
        }
        else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
        else bits = sign | ((e + 112) << 23) | (m << 13);
        float f; std::memcpy(&f, &bits, 4);
        return f;
    }

    // one texel to shared-exponent bytes: the largest channel's exponent, mantissas truncated (DirectXTexHDR.cpp:324-392)
    inline void ToRGBE(uint8_t* d, float r, float g, float b) noexcept
    {
        r = (r >= 0.f) ? r : 0.f; g = (g >= 0.f) ? g : 0.f; b = (b >= 0.f) ? b : 0.f;         // negatives (and NaN) count as 0
        const float rg = (r > g) ? r : g;
        float top = (rg > b) ? rg : b;
        if (top > 1e-32f)
        {
            int e;
            top = std::frexp(top, &e) * 256.f / top;
            e += 128;
            const uint8_t red = uint8_t(r * top), green = uint8_t(g * top), blue = uint8_t(b * top);
            d[0] = red; d[1] = green; d[2] = blue;
            d[3] = (red || green || blue) ? uint8_t(e & 0xff) : 0u;
        }
        else d[0] = d[1] = d[2] = d[3] = 0;
    }

    // The "new" run-length scheme: 2 2 <width hi> <width lo>, then each of the four channels as runs (128 + n, value) and
    // literals (n, n bytes), n <= 127. 0 = not encodable within rowPitch bytes (the row is then written raw). :394-588
    size_t EncodeRow(uint8_t* enc, const uint8_t* rgbe, size_t rowPitch, size_t width) noexcept
    {
        if (width < 8 || width > INT16_MAX) return 0;
        enc[0] = 2; enc[1] = 2; enc[2] = uint8_t(width >> 8); enc[3] = uint8_t(width & 0xff);
        enc += 4;
        size_t used = 4;
        for (int channel = 0; channel < 4; ++channel)
        {
            const uint8_t* v = rgbe + channel;              // this channel's values are 4 bytes apart
            for (size_t x = 0; x < width;)
            {
                size_t run = 1;
                while (x + run < width && run < 127 && v[run * 4] == v[0]) ++run;
                if (run > 1)
                {
                    if (used + 2 > rowPitch) return 0;
                    enc[0] = uint8_t(128u + run); enc[1] = v[0];
                    enc += 2; used += 2;
                    v += run * 4; x += run;
                    continue;
                }
                // a literal: up to the value that starts repeating
                uint8_t literal[128];
                size_t n = 1;
                literal[0] = v[0];
                while (x + n < width && n < 127 && v[(n - 1) * 4] != v[n * 4]) { literal[n] = v[n * 4]; ++n; }
                if (used + n + 1 > rowPitch) return 0;
                *enc++ = uint8_t(n);
                std::memcpy(enc, literal, n);
                enc += n; used += n + 1;
                v += n * 4; x += n;
            }
        }
        return used;
    }

    HRESULT ReadAll(const char* szFile, std::vector<uint8_t>& buf, size_t minimum) noexcept
    {
        FILE* f = std::fopen(szFile, "rb");
        if (!f) return E_FAIL;
        std::fseek(f, 0, SEEK_END); const long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        if (n < 0) { std::fclose(f); return E_FAIL; }
        if (uint64_t(n) > UINT32_MAX) { std::fclose(f); return HRESULT_E_FILE_TOO_LARGE; }
        if (size_t(n) < minimum) { std::fclose(f); return E_FAIL; }
        try { buf.resize(size_t(n)); } catch (...) { std::fclose(f); return E_OUTOFMEMORY; }
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
        std::fclose(f);
        return got == buf.size() ? S_OK : E_FAIL;
    }
}

HRESULT Blob::Trim(size_t size) noexcept
{
    if (!size) return E_INVALIDARG;
    if (!m_buffer) return HRESULT(0x8000FFFF);         // E_UNEXPECTED
    if (size > m_size) return E_INVALIDARG;
    m_size = size;
    return S_OK;
}

HRESULT GetMetadataFromHDRMemory(const void* pSource, size_t size, TexMetadata& metadata) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    size_t offset; float exposure;
    return DecodeHeader(static_cast<const uint8_t*>(pSource), size, metadata, offset, exposure);
}

HRESULT GetMetadataFromHDRFile(const char* szFile, TexMetadata& metadata) noexcept
{
    if (!szFile) return E_INVALIDARG;
    FILE* f = std::fopen(szFile, "rb");
    if (!f) return E_FAIL;
    uint8_t header[8192] = {};
    const size_t got = std::fread(header, 1, sizeof(header), f);
    std::fclose(f);
    if (got < sizeof(kSignature)) return E_FAIL;
    size_t offset; float exposure;
    return DecodeHeader(header, got, metadata, offset, exposure);
}

// LoadFromHDRMemory (DirectXTexHDR.cpp:688-880): always R32G32B32A32_FLOAT, alpha 1
HRESULT LoadFromHDRMemory(const void* pSource, size_t size, TexMetadata* metadata, ScratchImage& image) noexcept
{
    if (!pSource || size == 0) return E_INVALIDARG;
    image.Release();
    size_t offset; float exposure;
    TexMetadata mdata;
    HRESULT hr = DecodeHeader(static_cast<const uint8_t*>(pSource), size, mdata, offset, exposure);
    if (FAILED(hr)) return hr;
    if (offset > size) return E_FAIL;
    size_t left = size - offset;
    if (left == 0) return E_FAIL;
    hr = image.Initialize2D(mdata.format, mdata.width, mdata.height, 1, 1);
    if (FAILED(hr)) return hr;
    const Image* img = image.GetImage(0, 0, 0);
    if (!img) { image.Release(); return E_POINTER; }
    const uint8_t* src = static_cast<const uint8_t*>(pSource) + offset;
    const size_t width = mdata.width;
    auto bad = [&]() { image.Release(); return E_FAIL; };

    // pass 1: the four bytes of every texel, as floats, into the image
    for (size_t y = 0; y < mdata.height; ++y)
    {
        if (left < 4) return bad();
        uint8_t in[4];
        std::memcpy(in, src, 4); src += 4; left -= 4;
        float* row = reinterpret_cast<float*>(img->pixels + y * img->rowPitch);
        if (in[0] == 2 && in[1] == 2 && in[2] < 128)
        {
            // new scheme: the row's width, then channel after channel
            if (((size_t(in[2]) << 8) + in[3]) != width) return bad();
            for (int channel = 0; channel < 4; ++channel)
            {
                float* out = row + channel;
                for (size_t x = 0; x < width;)
                {
                    if (left < 2) return bad();
                    size_t n = *src;
                    if (n > 128)
                    {
                        n &= 127;
                        if (x + n > width) return bad();
                        const float value = float(src[1]);
                        for (size_t j = 0; j < n; ++j, out += 4) *out = value;
                        src += 2; left -= 2;
                    }
                    else
                    {
                        if (left < n + 1 || x + n > width) return bad();
                        ++src;
                        for (size_t j = 0; j < n; ++j, out += 4) *out = float(*src++);
                        left -= n + 1;
                    }
                    x += n;
                }
            }
            continue;
        }
        // old scheme: texels, where (1, 1, 1, n) repeats the previous one n times - shifted left 8 bits for each such marker in a row
        float prev[4] = { float(in[0]), float(in[1]), float(in[2]), float(in[3]) };
        float* out = row;
        int shift = 0;
        for (size_t x = 0; x < width;)
        {
            if (in[0] == 1 && in[1] == 1 && in[2] == 1)
            {
                if (shift > 24) return bad();
                const size_t n = size_t(in[3]) << shift;
                if (n + x > width) return bad();
                for (size_t j = 0; j < n; ++j, out += 4) { out[0] = prev[0]; out[1] = prev[1]; out[2] = prev[2]; out[3] = prev[3]; }
                x += n;
                shift += 8;
            }
            else
            {
                for (int c = 0; c < 4; ++c) out[c] = prev[c] = float(in[c]);
                shift = 0;
                ++x; out += 4;
            }
            if (x >= width) break;
            if (left < 4) return bad();
            std::memcpy(in, src, 4); src += 4; left -= 4;
        }
    }
    // pass 2: mantissa bytes and the shared exponent to floats, (m + 0.5) * 2^(e - 136) / exposure, alpha 1
    float* f = reinterpret_cast<float*>(image.GetPixels());
    const float scale = 1.0f / exposure;
    for (size_t i = 0; i < image.GetPixelsSize(); i += 16, f += 4)
    {
        const int e = int(f[3]) - (128 + 8);
        f[0] = scale * std::ldexp(f[0] + 0.5f, e);
        f[1] = scale * std::ldexp(f[1] + 0.5f, e);
        f[2] = scale * std::ldexp(f[2] + 0.5f, e);
        f[3] = 1.f;
    }
    if (metadata) *metadata = mdata;
    return S_OK;
}

HRESULT LoadFromHDRFile(const char* szFile, TexMetadata* metadata, ScratchImage& image) noexcept
{
    if (!szFile) return E_INVALIDARG;
    image.Release();
    std::vector<uint8_t> buf;
    const HRESULT hr = ReadAll(szFile, buf, sizeof(kSignature));
    if (FAILED(hr)) return hr;
    return LoadFromHDRMemory(buf.data(), buf.size(), metadata, image);
}

// SaveToHDRMemory (DirectXTexHDR.cpp:1001-1110): RGBA32F, RGB32F or RGBA16F in, run-length encoded RGBE out
HRESULT SaveToHDRMemory(const Image& image, Blob& blob) noexcept
{
    if (!image.pixels) return E_POINTER;
    if (image.width > INT16_MAX || image.height > INT16_MAX) return HRESULT_E_NOT_SUPPORTED;
    size_t channels;
    switch (image.format)
    {
    case DXGI_FORMAT_R32G32B32A32_FLOAT: case DXGI_FORMAT_R16G16B16A16_FLOAT: channels = 4; break;
    case DXGI_FORMAT_R32G32B32_FLOAT: channels = 3; break;
    default: return HRESULT_E_NOT_SUPPORTED;
    }
    blob.Release();
    char header[256] = {};
    std::snprintf(header, sizeof(header), "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", unsigned(image.height), unsigned(image.width));
    const size_t headerLen = std::strlen(header), rowPitch = image.width * 4;
    HRESULT hr = blob.Initialize(headerLen + image.height * rowPitch);
    if (FAILED(hr)) return hr;
    uint8_t* d = blob.GetBufferPointer();
    std::memcpy(d, header, headerLen);
    d += headerLen;
    std::unique_ptr<uint8_t[]> temp(new (std::nothrow) uint8_t[rowPitch * 2]);
    if (!temp) { blob.Release(); return E_OUTOFMEMORY; }
    uint8_t* rgbe = temp.get(); uint8_t* enc = temp.get() + rowPitch;
    for (size_t y = 0; y < image.height; ++y)
    {
        const uint8_t* s = image.pixels + y * image.rowPitch;
        if (image.format == DXGI_FORMAT_R16G16B16A16_FLOAT)
            for (size_t x = 0; x < image.width; ++x)
            {
                uint16_t h[3]; std::memcpy(h, s + x * 8, 6);
                ToRGBE(rgbe + x * 4, HalfToFloat(h[0]), HalfToFloat(h[1]), HalfToFloat(h[2]));
            }
        else
            for (size_t x = 0; x < image.width; ++x)
            {
                float v[3]; std::memcpy(v, s + x * channels * 4, 12);
                ToRGBE(rgbe + x * 4, v[0], v[1], v[2]);
            }
        const size_t n = EncodeRow(enc, rgbe, rowPitch, image.width);
        if (n > 0) { std::memcpy(d, enc, n); d += n; }
        else { std::memcpy(d, rgbe, rowPitch); d += rowPitch; }
    }
    hr = blob.Trim(size_t(d - blob.GetBufferPointer()));
    if (FAILED(hr)) blob.Release();
    return hr;
}

HRESULT SaveToHDRFile(const Image& image, const char* szFile) noexcept
{
    if (!szFile) return E_INVALIDARG;
    Blob blob;
    const HRESULT hr = SaveToHDRMemory(image, blob);
    if (FAILED(hr)) return hr;
    FILE* f = std::fopen(szFile, "wb");
    if (!f) return E_FAIL;
    const size_t n = std::fwrite(blob.GetBufferPointer(), 1, blob.GetBufferSize(), f);
    const bool closed = std::fclose(f) == 0;
    if (n != blob.GetBufferSize() || !closed) { std::remove(szFile); return E_FAIL; }
    return S_OK;
}
} // namespace DirectXTexAMD
