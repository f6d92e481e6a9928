"""TEST INFRASTRUCTURE ONLY - the parity oracle for the MI355X DirectXTex hot path.

Nothing under ``oracle/`` is part of the product. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.

Two layers:
  * ``oracle/_ref/libdxtex_ref.so`` - the reference's own block codecs (BC.cpp, BC4BC5.cpp, BC6HBC7.cpp),
    image drivers and scanline layer (DirectXTexConvert.cpp) compiled unmodified and in place from
    /root/reference by ``oracle/Makefile`` against the DirectXMath leaf shim in ``oracle/shim``
    (git-ignored build output; travels to the GPU box with the snapshot).
  * this package's numpy restatement of the image-level driver around them: LoadScanline for the
    supported source formats, 4x4 tile gather with partial-block replication, the ConvertScanline
    branches Compress reaches, ComputeMSE - each function cites the reference lines it follows.
"""
from .dxtex_oracle import *  # noqa: F401,F403
