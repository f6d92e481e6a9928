// TEST INFRASTRUCTURE ONLY (oracle/_ref): C ABI around the reference's own block codecs
// (DirectXTex/BC.h:321-343), compiled in place from /root/reference by oracle/Makefile.
// Used by tests/ as the parity checker and by bench.py's cpu_baseline leg (kind "reference").
// Never linked into, imported by, or called from the product library.
#include "DirectXTexP.h"
#include "BC.h"
#include <omp.h>

using namespace DirectX;

namespace
{
    enum : int
    {
        F_BC1 = 71, F_BC1_SRGB = 72, F_BC2 = 74, F_BC2_SRGB = 75, F_BC3 = 77, F_BC3_SRGB = 78,
        F_BC4U = 80, F_BC4S = 81, F_BC5U = 83, F_BC5S = 84, F_BC6HU = 95, F_BC6HS = 96, F_BC7 = 98, F_BC7_SRGB = 99
    };

    size_t BlockBytes(int fmt)
    {
        switch (fmt)
        {
        case F_BC1: case F_BC1_SRGB: case F_BC4U: case F_BC4S: return 8;
        case F_BC2: case F_BC2_SRGB: case F_BC3: case F_BC3_SRGB: case F_BC5U: case F_BC5S:
        case F_BC6HU: case F_BC6HS: case F_BC7: case F_BC7_SRGB: return 16;
        default: return 0;
        }
    }
}

extern "C"
{
    int dxtex_ref_num_threads() { return omp_get_max_threads(); }

    // Same dispatch as DetermineEncoderSettings (DirectXTexCompress.cpp:46-68) + the BC1 special case
    // (DirectXTexCompress.cpp:191-194). rgba = nblocks x 16 texels x 4 floats (row-major 4x4 tiles).
    int dxtex_ref_encode_blocks(int fmt, uint32_t bcflags, float threshold, const float* rgba, size_t nblocks,
                                uint8_t* out, int threads)
    {
        const size_t bb = BlockBytes(fmt);
        if (!bb || !rgba || !out) return -1;
        if (threads <= 0) threads = omp_get_max_threads();
        #pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
        for (int64_t nb = 0; nb < int64_t(nblocks); ++nb)
        {
            XMVECTOR temp[16];
            memcpy(temp, rgba + size_t(nb) * 64, sizeof(temp));
            uint8_t* p = out + size_t(nb) * bb;
            switch (fmt)
            {
            case F_BC1: case F_BC1_SRGB: D3DXEncodeBC1(p, temp, threshold, bcflags); break;
            case F_BC2: case F_BC2_SRGB: D3DXEncodeBC2(p, temp, bcflags); break;
            case F_BC3: case F_BC3_SRGB: D3DXEncodeBC3(p, temp, bcflags); break;
            case F_BC4U: D3DXEncodeBC4U(p, temp, bcflags); break;
            case F_BC4S: D3DXEncodeBC4S(p, temp, bcflags); break;
            case F_BC5U: D3DXEncodeBC5U(p, temp, bcflags); break;
            case F_BC5S: D3DXEncodeBC5S(p, temp, bcflags); break;
            case F_BC6HU: D3DXEncodeBC6HU(p, temp, bcflags); break;
            case F_BC6HS: D3DXEncodeBC6HS(p, temp, bcflags); break;
            case F_BC7: case F_BC7_SRGB: D3DXEncodeBC7(p, temp, bcflags); break;
            default: break;
            }
        }
        return 0;
    }

    int dxtex_ref_decode_blocks(int fmt, const uint8_t* in, size_t nblocks, float* rgba)
    {
        const size_t bb = BlockBytes(fmt);
        if (!bb || !rgba || !in) return -1;
        #pragma omp parallel for schedule(static)
        for (int64_t nb = 0; nb < int64_t(nblocks); ++nb)
        {
            XMVECTOR temp[16];
            const uint8_t* p = in + size_t(nb) * bb;
            switch (fmt)
            {
            case F_BC1: case F_BC1_SRGB: D3DXDecodeBC1(temp, p); break;
            case F_BC2: case F_BC2_SRGB: D3DXDecodeBC2(temp, p); break;
            case F_BC3: case F_BC3_SRGB: D3DXDecodeBC3(temp, p); break;
            case F_BC4U: D3DXDecodeBC4U(temp, p); break;
            case F_BC4S: D3DXDecodeBC4S(temp, p); break;
            case F_BC5U: D3DXDecodeBC5U(temp, p); break;
            case F_BC5S: D3DXDecodeBC5S(temp, p); break;
            case F_BC6HU: D3DXDecodeBC6HU(temp, p); break;
            case F_BC6HS: D3DXDecodeBC6HS(temp, p); break;
            case F_BC7: case F_BC7_SRGB: D3DXDecodeBC7(temp, p); break;
            default: break;
            }
            memcpy(rgba + size_t(nb) * 64, temp, sizeof(temp));
        }
        return 0;
    }
}
