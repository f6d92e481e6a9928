// BC6H integer pipeline shared by the decoder and the encoder (BC6HBC7.cpp: INTColor :452-575, Quantize :1864-1889,
// Unquantize :1893-1926, FinishUnquantize :1930-1940, TransformInverse :1153-1165).
// Texel components are half-float bit patterns carried as ints; unsigned formats drop negatives to 0.
#pragma once
#include <stdint.h>

#if defined(DXTEX_HOST_DEBUG)
#define DXTEX_HD6 __host__ __device__ inline
#else
#define DXTEX_HD6 __device__ __forceinline__
#endif

namespace dxtex
{
namespace bc6h
{
enum : int { F16S_MASK = 0x8000, F16EM_MASK = 0x7FFF, F16MAX = 0x7BFF };

// SIGN_EXTEND(x, nb) (BC6HBC7.cpp:47)
DXTEX_HD6 int sign_extend(int x, int nb) { return ((x & (1 << (nb - 1))) ? ((~0) ^ ((1 << nb) - 1)) : 0) | x; }

// INTColor::F16ToINT (:533-551): half bits -> int
DXTEX_HD6 int f16_to_int(uint32_t h, bool isSigned)
{
    if (isSigned)
    {
        const int s = int(h & F16S_MASK);
        int m = int(h & F16EM_MASK);
        if (m > F16MAX) m = F16MAX;
        return s ? -m : m;
    }
    return (h & F16S_MASK) ? 0 : int(h);
}

// INTColor::INT2F16 (:553-575): int -> half bits
DXTEX_HD6 uint32_t int_to_f16(int v, bool isSigned)
{
    if (isSigned)
    {
        int s = 0;
        if (v < 0) { s = F16S_MASK; v = -v; }
        return uint32_t(s | v) & 0xFFFFu;
    }
    return uint32_t(v) & 0xFFFFu;
}

DXTEX_HD6 int quantize(int v, int prec, bool isSigned)
{
    int q;
    if (isSigned)
    {
        int s = 0;
        if (v < 0) { s = 1; v = -v; }
        q = (prec >= 16) ? v : (v << (prec - 1)) / (F16MAX + 1);
        if (s) q = -q;
    }
    else
        q = (prec >= 15) ? v : (v << prec) / (F16MAX + 1);
    return q;
}

DXTEX_HD6 int unquantize(int comp, int bits, bool isSigned)
{
    int unq;
    if (isSigned)
    {
        if (bits >= 16) unq = comp;
        else
        {
            int s = 0;
            if (comp < 0) { s = 1; comp = -comp; }
            if (comp == 0) unq = 0;
            else if (comp >= ((1 << (bits - 1)) - 1)) unq = 0x7FFF;
            else unq = ((comp << 15) + 0x4000) >> (bits - 1);
            if (s) unq = -unq;
        }
    }
    else
    {
        if (bits >= 15) unq = comp;
        else if (comp == 0) unq = 0;
        else if (comp == ((1 << bits) - 1)) unq = 0xFFFF;
        else unq = ((comp << 16) + 0x8000) >> bits;
    }
    return unq;
}

DXTEX_HD6 int finish_unquantize(int comp, bool isSigned)
{
    if (isSigned) return (comp < 0) ? -(((-comp) * 31) >> 5) : (comp * 31) >> 5;
    return (comp * 31) >> 6;
}

} // namespace bc6h
} // namespace dxtex
