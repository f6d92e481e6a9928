"""ctypes binding of include/dxtex_amd.h. One-to-one with the C ABI; numpy arrays carry host pixels,
integers carry device pointers (e.g. ``torch.Tensor.data_ptr()``)."""
import ctypes
import os

import numpy as np

from . import formats as F

_HERE = os.path.dirname(os.path.abspath(__file__))


_loaded = {"path": None, "dev": False}


def library_path(dev=None):
    """Path of the loaded library (dev=None), of the product library (dev=False) or of the development build (dev=True)."""
    if dev is None:
        return _loaded["path"]
    return os.path.join(_HERE, "lib", "libdxtex_amd_dev.so" if dev else "libdxtex_amd.so")


def _open(dev):
    path = library_path(dev)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the MI355X path)")
    try:
        # Share PyTorch's HIP runtime when PyTorch is in the process (same SONAME libamdhip64.so.7), so that
        # torch device pointers and streams are valid in this library.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    # RTLD_LOCAL: the product and the development build export the same C++ symbols (dxtex::launch_*); with both in the global scope the
    # second one's internal calls would bind to the first one's definitions and a development knob would silently do nothing. (The libraries
    # are also linked with -Bsymbolic-functions, so each binds its own calls whatever the loader is told.) libamdhip64 is shared with
    # PyTorch by its SONAME, not through this handle's visibility.
    return ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), path


class Volume(ctypes.Structure):
    """dxtex_volume (include/dxtex_amd.h): one mip level of a volume texture."""
    _fields_ = [("width", ctypes.c_size_t), ("height", ctypes.c_size_t), ("depth", ctypes.c_size_t), ("format", ctypes.c_int32),
                ("rowPitch", ctypes.c_size_t), ("slicePitch", ctypes.c_size_t), ("pixels", ctypes.c_void_p)]


class Image(ctypes.Structure):
    """Mirrors ``dxtex_image`` / ``DirectX::Image`` (DirectXTex.h:437-445)."""
    _fields_ = [("width", ctypes.c_size_t), ("height", ctypes.c_size_t), ("format", ctypes.c_int32),
                ("rowPitch", ctypes.c_size_t), ("slicePitch", ctypes.c_size_t), ("pixels", ctypes.c_void_p)]


_lib = None
_P = ctypes.POINTER
_ctx_p = ctypes.c_void_p

_SIGS = {
    "dxtex_ctx_create": (ctypes.c_int32, [ctypes.c_int, _P(_ctx_p)]),
    "dxtex_ctx_destroy": (None, [_ctx_p]),
    "dxtex_ctx_set_stream": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p]),
    "dxtex_ctx_get_stream": (ctypes.c_void_p, [_ctx_p]),
    "dxtex_ctx_synchronize": (ctypes.c_int32, [_ctx_p]),
    "dxtex_ctx_last_error": (ctypes.c_char_p, [_ctx_p]),
    "dxtex_ctx_last_kernel_ms": (ctypes.c_float, [_ctx_p]),
    "dxtex_ctx_profile_begin": (ctypes.c_int32, [_ctx_p]),
    "dxtex_ctx_profile_end": (ctypes.c_int32, [_ctx_p, ctypes.c_char_p, ctypes.c_size_t, _P(ctypes.c_float), _P(ctypes.c_uint32), ctypes.c_size_t, _P(ctypes.c_size_t)]),
    "dxtex_is_compressed": (ctypes.c_int, [ctypes.c_int32]),
    "dxtex_bits_per_pixel": (ctypes.c_size_t, [ctypes.c_int32]),
    "dxtex_compute_pitch": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_size_t, ctypes.c_size_t, _P(ctypes.c_size_t), _P(ctypes.c_size_t)]),
    "dxtex_ctx_prepare": (ctypes.c_int32, [_ctx_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32, ctypes.c_size_t, _P(ctypes.c_size_t)]),
    "dxtex_compress": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32, ctypes.c_float]),
    "dxtex_compress_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32, ctypes.c_float]),
    "dxtex_compress_many_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_size_t, ctypes.c_uint32, ctypes.c_float]),
    "dxtex_decompress": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image)]),
    "dxtex_decompress_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image)]),
    "dxtex_encode_blocks": (ctypes.c_int32, [_ctx_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dxtex_decode_blocks": (ctypes.c_int32, [_ctx_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "dxtex_generate_mips": (ctypes.c_int32, [_ctx_p, _P(Image), ctypes.c_size_t, ctypes.c_uint32]),
    "dxtex_compress_multi": (ctypes.c_int32, [_P(_ctx_p), ctypes.c_size_t, _P(Image), _P(Image), ctypes.c_uint32, ctypes.c_float]),
    "dxtex_generate_mips_multi": (ctypes.c_int32, [_P(_ctx_p), ctypes.c_size_t, _P(Image), ctypes.c_size_t, ctypes.c_uint32]),
    "dxtex_generate_mips_device": (ctypes.c_int32, [_ctx_p, _P(Image), ctypes.c_size_t, ctypes.c_uint32]),
    "dxtex_compress_many": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_size_t, ctypes.c_uint32, ctypes.c_float]),
    "dxtex_convert": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32, ctypes.c_float]),
    "dxtex_convert_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32, ctypes.c_float]),
    "dxtex_generate_mips3d": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]),
    "dxtex_generate_mips3d_device": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]),
    "dxtex_premultiply_alpha": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32]),
    "dxtex_premultiply_alpha_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32]),
    "dxtex_scale_mips_alpha_for_coverage": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_size_t, ctypes.c_float]),
    "dxtex_scale_mips_alpha_for_coverage_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_size_t, ctypes.c_float]),
    "dxtex_resize": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32]),
    "dxtex_resize_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), ctypes.c_uint32]),
    "dxtex_compute_mse_device": (ctypes.c_int32, [_ctx_p, _P(Image), _P(Image), _P(ctypes.c_double)]),
    "dxtex_device_alloc": (ctypes.c_int32, [_ctx_p, ctypes.c_size_t, _P(ctypes.c_void_p)]),
    "dxtex_device_free": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p]),
    "dxtex_memcpy_h2d": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "dxtex_memcpy_d2h": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "dxtex_device_memset": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]),
    "dxtex_copy_rows_device": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]),
    "dxtex_memcpy_h2d_async": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "dxtex_memcpy_d2h_async": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "dxtex_host_alloc": (ctypes.c_int32, [_ctx_p, ctypes.c_size_t, _P(ctypes.c_void_p)]),
    "dxtex_host_free": (ctypes.c_int32, [_ctx_p, ctypes.c_void_p]),
    "dxtex_alpha_all_opaque_device": (ctypes.c_int32, [_ctx_p, _P(Image), ctypes.c_size_t, _P(ctypes.c_int)]),
    "dxtex_ctx_transfer_bytes": (ctypes.c_int32, [_ctx_p, _P(ctypes.c_uint64), _P(ctypes.c_uint64), ctypes.c_int]),
}
def load(dev=False):
    """Binds this module to the product library (the default, done at import) or - dev=True - to libdxtex_amd_dev.so: the same sources
    compiled with -DDXTEX_DEV, the only build that reads DXTEX_* development knobs from the environment. The choice is this explicit
    call, made by the tests and tools that need knobs before they create a Context; no environment variable selects a library, so
    nothing in a user's shell changes what `import directxtex_amd` runs. A Context keeps the library that created it (its handle belongs to
    that library's statics), so contexts of both builds can live side by side; module-level helpers follow the latest load()."""
    global _lib
    lib, path = _open(dev)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)     # AttributeError here == the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _loaded["path"], _loaded["dev"] = path, bool(dev)
    return path


load(False)

EXPORTED_SYMBOLS = tuple(_SIGS)


class DxtexError(RuntimeError):
    def __init__(self, hr, msg=""):
        self.hresult = hr
        super().__init__(f"HRESULT 0x{hr & 0xFFFFFFFF:08X} {msg}".strip())


def compute_pitch(fmt, width, height):
    rp, sp = ctypes.c_size_t(), ctypes.c_size_t()
    hr = _lib.dxtex_compute_pitch(fmt, width, height, ctypes.byref(rp), ctypes.byref(sp))
    if hr != 0:
        raise DxtexError(hr, "compute_pitch")
    return rp.value, sp.value


def is_compressed(fmt):
    return bool(_lib.dxtex_is_compressed(fmt))


def bits_per_pixel(fmt):
    return int(_lib.dxtex_bits_per_pixel(fmt))


def _host_image(arr, width, height, fmt, row_pitch=None):
    """numpy buffer -> dxtex_image (host pointer). `arr` must be C-contiguous."""
    assert arr.flags["C_CONTIGUOUS"]
    rp, sp = compute_pitch(fmt, width, height)
    if row_pitch is not None:
        rows = sp // rp
        rp, sp = row_pitch, row_pitch * rows
    assert arr.nbytes >= sp, (arr.nbytes, sp)
    return Image(width, height, fmt, rp, sp, arr.ctypes.data)


def device_image(ptr, width, height, fmt, row_pitch=None):
    rp, sp = compute_pitch(fmt, width, height)
    if row_pitch is not None:
        rows = sp // rp
        rp, sp = row_pitch, row_pitch * rows
    return Image(width, height, fmt, rp, sp, ptr)


def _ctx_array(contexts):
    libs = {id(c._lib) for c in contexts}
    assert len(libs) == 1, "the contexts of one call must come from one library"
    return (_ctx_p * len(contexts))(*[c._h for c in contexts])


def compress_multi(contexts, pixels, width, height, src_format, dst_format, flags=0, threshold=0.5, out=None):
    """ONE host image over several contexts (dxtex_compress_multi: stripes of block rows, a thread per context) -> the BC payload,
    byte for byte what contexts[0].compress returns."""
    pixels = np.ascontiguousarray(pixels)
    src = _host_image(pixels, width, height, src_format)
    rp, sp = compute_pitch(dst_format, width, height)
    if out is None:
        out = np.zeros(sp, np.uint8)
    dst = Image(width, height, dst_format, rp, sp, out.ctypes.data)
    c0 = contexts[0]
    c0._check(c0._lib.dxtex_compress_multi(_ctx_array(contexts), len(contexts), ctypes.byref(src), ctypes.byref(dst), flags, threshold), "compress_multi")
    return out


def generate_mips_multi(contexts, level0, width, height, fmt, nlevels, filter_flags=0):
    """dxtex_generate_mips_multi: the chain of a host image with its large levels cut into stripes of rows over the contexts; returns the
    levels as uint8 arrays (tight pitch), identical to contexts[0].generate_mips."""
    bufs, levels = [], []
    w, h = width, height
    for i in range(nlevels):
        rp, sp = compute_pitch(fmt, w, h)
        buf = np.zeros(sp, np.uint8)
        if i == 0:
            buf[:] = np.ascontiguousarray(level0).view(np.uint8).reshape(-1)[:sp]
        bufs.append(buf)
        levels.append(Image(w, h, fmt, rp, sp, buf.ctypes.data))
        w, h = max(1, w >> 1), max(1, h >> 1)
    arr = (Image * nlevels)(*levels)
    c0 = contexts[0]
    c0._check(c0._lib.dxtex_generate_mips_multi(_ctx_array(contexts), len(contexts), arr, nlevels, filter_flags), "generate_mips_multi")
    return bufs


class Context:
    """One context per GPU (``dxtex_ctx``)."""

    def __init__(self, device=0):
        self._h = _ctx_p()
        self._lib = _lib            # the library that creates the handle serves it for life, whatever load() binds the module to later
        hr = self._lib.dxtex_ctx_create(device, ctypes.byref(self._h))
        if hr != 0:
            raise DxtexError(hr, "dxtex_ctx_create: no usable gfx950 device (this library has no CPU path)")
        self.device = device

    def close(self):
        if self._h:
            self._lib.dxtex_ctx_destroy(self._h)
            self._h = _ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, hr, what):
        if hr != 0:
            raise DxtexError(hr, f"{what}: {self._lib.dxtex_ctx_last_error(self._h).decode()}")

    # -- plumbing ---------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._check(self._lib.dxtex_ctx_set_stream(self._h, stream_ptr), "set_stream")

    def stream(self):
        return self._lib.dxtex_ctx_get_stream(self._h)

    def synchronize(self):
        self._check(self._lib.dxtex_ctx_synchronize(self._h), "synchronize")

    def last_kernel_ms(self):
        return float(self._lib.dxtex_ctx_last_kernel_ms(self._h))

    def profile_begin(self):
        self._check(self._lib.dxtex_ctx_profile_begin(self._h), "profile_begin")

    def profile_end(self):
        """-> {kernel name: (total ms, launches)} since profile_begin (synchronises the stream)."""
        cap = 256
        names = ctypes.create_string_buffer(16384)
        ms = (ctypes.c_float * cap)()
        n = (ctypes.c_uint32 * cap)()
        cnt = ctypes.c_size_t()
        self._check(self._lib.dxtex_ctx_profile_end(self._h, names, 16384, ms, n, cap, ctypes.byref(cnt)), "profile_end")
        keys = names.value.decode().split("\n")[:cnt.value]
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(keys)}

    # -- device-resident pipeline helpers -----------------------------------------------------------
    def device_alloc(self, nbytes, zero=False):
        p = ctypes.c_void_p()
        self._check(self._lib.dxtex_device_alloc(self._h, nbytes, ctypes.byref(p)), "device_alloc")
        if zero:
            self._check(self._lib.dxtex_device_memset(self._h, p, 0, nbytes), "device_memset")
        return p.value

    def device_free(self, ptr):
        self._check(self._lib.dxtex_device_free(self._h, ptr), "device_free")

    def host_alloc(self, nbytes):
        """Page-locked host memory as a numpy uint8 array (free with host_free(arr.ctypes.data) after the last use)."""
        p = ctypes.c_void_p()
        self._check(self._lib.dxtex_host_alloc(self._h, nbytes, ctypes.byref(p)), "host_alloc")
        return np.ctypeslib.as_array(ctypes.cast(p, _P(ctypes.c_uint8)), shape=(nbytes,))

    def host_free(self, ptr):
        self._check(self._lib.dxtex_host_free(self._h, ptr), "host_free")

    def upload(self, dst_ptr, arr, nbytes=None, sync=False):
        """Stream-ordered host -> device copy of a C-contiguous numpy buffer (keep it alive until synchronize())."""
        n = arr.nbytes if nbytes is None else nbytes
        fn = self._lib.dxtex_memcpy_h2d if sync else self._lib.dxtex_memcpy_h2d_async
        self._check(fn(self._h, dst_ptr, arr.ctypes.data, n), "upload")

    def download(self, arr, src_ptr, nbytes=None, sync=False):
        n = arr.nbytes if nbytes is None else nbytes
        fn = self._lib.dxtex_memcpy_d2h if sync else self._lib.dxtex_memcpy_d2h_async
        self._check(fn(self._h, arr.ctypes.data, src_ptr, n), "download")

    def copy_rows_device(self, dst_ptr, dst_pitch, src_ptr, src_pitch, row_bytes, rows):
        self._check(self._lib.dxtex_copy_rows_device(self._h, dst_ptr, dst_pitch, src_ptr, src_pitch, row_bytes, rows), "copy_rows_device")

    def alpha_all_opaque_device(self, images):
        arr = (Image * len(images))(*images)
        out = ctypes.c_int(0)
        self._check(self._lib.dxtex_alpha_all_opaque_device(self._h, arr, len(images), ctypes.byref(out)), "alpha_all_opaque_device")
        return bool(out.value)

    def transfer_bytes(self, reset=False):
        """(host -> device bytes, device -> host bytes) this context has moved since creation / the last reset."""
        up, down = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.dxtex_ctx_transfer_bytes(self._h, ctypes.byref(up), ctypes.byref(down), 1 if reset else 0), "transfer_bytes")
        return int(up.value), int(down.value)

    # -- Compress -----------------------------------------------------------------------------------
    def prepare(self, width, height, src_format, dst_format, flags=0, count=1):
        """GPUCompressBC::Prepare's role: allocate the search scratch and staging for `count` images of this shape now.
        Returns the bytes of device memory the context holds afterwards."""
        held = ctypes.c_size_t(0)
        self._check(self._lib.dxtex_ctx_prepare(self._h, width, height, src_format, dst_format, flags, count, ctypes.byref(held)), "prepare")
        return held.value

    def compress(self, pixels, width, height, src_format, dst_format, flags=0, threshold=0.5, src_row_pitch=None):
        """Host image (numpy, any dtype, C-contiguous rows) -> numpy uint8 BC payload (tight pitch)."""
        pixels = np.ascontiguousarray(pixels)
        src = _host_image(pixels, width, height, src_format, src_row_pitch)
        rp, sp = compute_pitch(dst_format, width, height)
        out = np.zeros(sp, np.uint8)
        dst = Image(width, height, dst_format, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_compress(self._h, ctypes.byref(src), ctypes.byref(dst), flags, threshold), "compress")
        return out

    def compress_into(self, pixels, width, height, src_format, out, dst_format, flags=0, threshold=0.5):
        """dxtex_compress with caller-owned buffers on both sides (e.g. pinned memory): `out` receives the tight BC payload."""
        src = _host_image(pixels, width, height, src_format)
        rp, sp = compute_pitch(dst_format, width, height)
        assert out.flags["C_CONTIGUOUS"] and out.nbytes >= sp
        dst = Image(width, height, dst_format, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_compress(self._h, ctypes.byref(src), ctypes.byref(dst), flags, threshold), "compress")

    def compress_device(self, src_ptr, width, height, src_format, dst_ptr, dst_format, flags=0, threshold=0.5, src_row_pitch=None):
        src = device_image(src_ptr, width, height, src_format, src_row_pitch)
        dst = device_image(dst_ptr, width, height, dst_format)
        self._check(self._lib.dxtex_compress_device(self._h, ctypes.byref(src), ctypes.byref(dst), flags, threshold), "compress_device")

    def compress_many_device(self, srcs, dsts, flags=0, threshold=0.5):
        n = len(srcs)
        a = (Image * n)(*srcs)
        b = (Image * n)(*dsts)
        self._check(self._lib.dxtex_compress_many_device(self._h, a, b, n, flags, threshold), "compress_many_device")

    def compress_many(self, images, width, height, src_format, dst_format, flags=0, threshold=0.5):
        """Array of same-sized host images -> list of BC payloads, through dxtex_compress_many (the cfg5 entry point: chunks of the
        array are staged through pinned memory on copy streams while the previous chunk is searched)."""
        images = [np.ascontiguousarray(im) for im in images]
        n = len(images)
        rp, sp = compute_pitch(dst_format, width, height)
        outs = [np.zeros(sp, np.uint8) for _ in range(n)]
        srcs = (Image * n)(*[_host_image(im, width, height, src_format) for im in images])
        dsts = (Image * n)(*[Image(width, height, dst_format, rp, sp, o.ctypes.data) for o in outs])
        self._check(self._lib.dxtex_compress_many(self._h, srcs, dsts, n, flags, threshold), "compress_many")
        return outs

    def compress_array(self, items, dst_format, flags=0, threshold=0.5):
        """General form of dxtex_compress_many: items = [(pixels, width, height, src_format, src_row_pitch or None), ...] - any mix
        of sizes and source formats, as the images of a DirectX::Compress array call may be. Returns the list of tight BC payloads."""
        keep = [np.ascontiguousarray(it[0]) for it in items]
        n = len(items)
        outs, srcs, dsts = [], [], []
        for arr, (_, w, h, fmt, pitch) in zip(keep, items):
            srcs.append(_host_image(arr, w, h, fmt, pitch))
            rp, sp = compute_pitch(dst_format, w, h)
            o = np.zeros(sp, np.uint8); outs.append(o)
            dsts.append(Image(w, h, dst_format, rp, sp, o.ctypes.data))
        a = (Image * n)(*srcs); b = (Image * n)(*dsts)
        self._check(self._lib.dxtex_compress_many(self._h, a, b, n, flags, threshold), "compress_many")
        return outs

    def encode_blocks(self, bc_format, rgba, flags=0, threshold=0.5):
        rgba = np.ascontiguousarray(rgba, np.float32).reshape(-1, 16, 4)
        n = rgba.shape[0]
        out = np.zeros((n, F.BC_BLOCK_BYTES[bc_format]), np.uint8)
        self._check(self._lib.dxtex_encode_blocks(self._h, bc_format, flags, threshold, rgba.ctypes.data, n, out.ctypes.data), "encode_blocks")
        return out

    def decode_blocks(self, bc_format, blocks):
        blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1, F.BC_BLOCK_BYTES[bc_format])
        n = blocks.shape[0]
        out = np.zeros((n, 16, 4), np.float32)
        self._check(self._lib.dxtex_decode_blocks(self._h, bc_format, blocks.ctypes.data, n, out.ctypes.data), "decode_blocks")
        return out

    def decompress(self, payload, width, height, bc_format, dst_format):
        payload = np.ascontiguousarray(payload, np.uint8)
        src = _host_image(payload, width, height, bc_format)
        rp, sp = compute_pitch(dst_format, width, height)
        out = np.zeros(sp, np.uint8)
        dst = Image(width, height, dst_format, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_decompress(self._h, ctypes.byref(src), ctypes.byref(dst)), "decompress")
        return out

    def decompress_device(self, src_ptr, width, height, bc_format, dst_ptr, dst_format):
        src = device_image(src_ptr, width, height, bc_format)
        dst = device_image(dst_ptr, width, height, dst_format)
        self._check(self._lib.dxtex_decompress_device(self._h, ctypes.byref(src), ctypes.byref(dst)), "decompress_device")

    def compute_mse_device(self, a_ptr, a_format, b_ptr, b_format, width, height):
        """Per-channel MSE (4 doubles) of two device images of the same size."""
        a = device_image(a_ptr, width, height, a_format)
        b = device_image(b_ptr, width, height, b_format)
        out = (ctypes.c_double * 4)()
        self._check(self._lib.dxtex_compute_mse_device(self._h, ctypes.byref(a), ctypes.byref(b), out), "compute_mse_device")
        return np.array(list(out), np.float64)

    # -- GenerateMipMaps / Convert / Resize ---------------------------------------------------------
    def generate_mips(self, level0, width, height, fmt, nlevels, filter_flags):
        """Returns [level0, level1, ...] as numpy uint8 buffers with tight pitch."""
        levels, bufs = [], []
        w, h = width, height
        for i in range(nlevels):
            rp, sp = compute_pitch(fmt, w, h)
            buf = np.zeros(sp, np.uint8)
            if i == 0:
                src = np.ascontiguousarray(level0).view(np.uint8).reshape(-1)
                buf[:] = src[:sp]
            bufs.append(buf)
            levels.append(Image(w, h, fmt, rp, sp, buf.ctypes.data))
            w, h = max(1, w >> 1), max(1, h >> 1)
        arr = (Image * nlevels)(*levels)
        self._check(self._lib.dxtex_generate_mips(self._h, arr, nlevels, filter_flags), "generate_mips")
        return bufs

    def generate_mips_device(self, levels, filter_flags):
        """levels: list of device Images (capi.device_image) forming a mip chain; fills levels[1:] from levels[0]."""
        arr = (Image * len(levels))(*levels)
        self._check(self._lib.dxtex_generate_mips_device(self._h, arr, len(levels), filter_flags), "generate_mips_device")

    def convert(self, pixels, width, height, src_format, dst_format, filter_flags=0, threshold=0.5):
        pixels = np.ascontiguousarray(pixels)
        src = _host_image(pixels, width, height, src_format)
        rp, sp = compute_pitch(dst_format, width, height)
        out = np.zeros(sp, np.uint8)
        dst = Image(width, height, dst_format, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_convert(self._h, ctypes.byref(src), ctypes.byref(dst), filter_flags, threshold), "convert")
        return out

    def generate_mips3d(self, volume, width, height, depth, fmt, nlevels, filter_flags):
        """DirectX::GenerateMipMaps3D: `volume` = the base slices (tight, consecutive). Returns one uint8 buffer per level."""
        bufs, vols = [], []
        w, h, d = width, height, depth
        for i in range(nlevels):
            rp, sp = compute_pitch(fmt, w, h)
            buf = np.zeros(sp * d, np.uint8)
            if i == 0:
                buf[:] = np.ascontiguousarray(volume).view(np.uint8).reshape(-1)[:sp * d]
            bufs.append(buf)
            vols.append(Volume(w, h, d, fmt, rp, sp, buf.ctypes.data))
            w, h, d = max(1, w >> 1), max(1, h >> 1), max(1, d >> 1)
        arr = (Volume * nlevels)(*vols)
        self._check(self._lib.dxtex_generate_mips3d(self._h, arr, nlevels, filter_flags), "generate_mips3d")
        return bufs

    def premultiply_alpha(self, pixels, width, height, fmt, flags=0):
        """DirectX::PremultiplyAlpha (flags = TEX_PMALPHA_*: 0x1 IGNORE_SRGB, 0x2 REVERSE, 0x1000000 / 0x2000000 SRGB_IN / OUT)."""
        pixels = np.ascontiguousarray(pixels)
        src = _host_image(pixels, width, height, fmt)
        rp, sp = compute_pitch(fmt, width, height)
        out = np.zeros(sp, np.uint8)
        dst = Image(width, height, fmt, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_premultiply_alpha(self._h, ctypes.byref(src), ctypes.byref(dst), flags), "premultiply_alpha")
        return out

    def scale_mips_alpha_for_coverage(self, levels, width, height, fmt, alpha_reference):
        """DirectX::ScaleMipMapsAlphaForCoverage on a list of tight per-level buffers; returns the new levels."""
        levels = [np.ascontiguousarray(l).view(np.uint8).reshape(-1) for l in levels]
        n = len(levels)
        srcs, dsts, outs = [], [], []
        w, h = width, height
        for l in levels:
            srcs.append(_host_image(l, w, h, fmt))
            rp, sp = compute_pitch(fmt, w, h)
            o = np.zeros(sp, np.uint8); outs.append(o)
            dsts.append(Image(w, h, fmt, rp, sp, o.ctypes.data))
            w, h = max(1, w >> 1), max(1, h >> 1)
        a = (Image * n)(*srcs); b = (Image * n)(*dsts)
        self._check(self._lib.dxtex_scale_mips_alpha_for_coverage(self._h, a, b, n, alpha_reference), "scale_mips_alpha_for_coverage")
        return outs

    def resize(self, pixels, width, height, fmt, new_width, new_height, filter_flags=0):
        pixels = np.ascontiguousarray(pixels)
        src = _host_image(pixels, width, height, fmt)
        rp, sp = compute_pitch(fmt, new_width, new_height)
        out = np.zeros(sp, np.uint8)
        dst = Image(new_width, new_height, fmt, rp, sp, out.ctypes.data)
        self._check(self._lib.dxtex_resize(self._h, ctypes.byref(src), ctypes.byref(dst), filter_flags), "resize")
        return out
