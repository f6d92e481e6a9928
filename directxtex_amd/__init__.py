"""directxtex_amd - MI355X-native DirectXTex hot path.

This Python package is only a thin ctypes binding over the C ABI in ``include/dxtex_amd.h``
(``lib/libdxtex_amd.so``: hand-written HIP kernels for gfx950 + the C++ host layer). It exists so that the
parity tests and ``bench.py`` can drive the library; the product is the shared library. There is no CPU
or PyTorch fallback: if the library is missing, importing :mod:`directxtex_amd.capi` raises.
"""
from .formats import *  # noqa: F401,F403
from . import capi  # noqa: F401
from .capi import (Context, DxtexError, Image, compute_pitch, device_image, is_compressed, bits_per_pixel, library_path)  # noqa: F401
