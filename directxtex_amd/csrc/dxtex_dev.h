// Development knobs. The shipped library (libdxtex_amd.so) reads NO environment variable: dev_env() is a constant nullptr there,
// so nothing in a user's shell can change what the encoders search, in which order, or how work is cut into passes. The same
// sources compiled with -DDXTEX_DEV (libdxtex_amd_dev.so, built next to the product by the same Makefile, loaded only by the
// tests and the tools under tools/) honour them: A/B switches of equivalent search strategies (the tests assert the bytes do
// not change), pass / chunk sizes small enough to exercise the multi-pass machinery on tiny images, and the BC6H bring-up aids.
#pragma once
#include <cstdlib>

namespace dxtex
{
inline const char* dev_env(const char* name)
{
#if defined(DXTEX_DEV)
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
} // namespace dxtex
