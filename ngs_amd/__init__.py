"""ngs_amd -- MI355X-native `ngs qc` record-scanning hot path.

The product is ``libngsq.so`` (hand-written HIP kernels for gfx950 + C++ host
code behind the C ABI of ``include/ngsq.h``).  This package holds its sources
(``csrc/``), the build recipe (``build.py``) and a ctypes harness used by the
tests and ``bench.py``.
"""
from . import ffi  # noqa: F401

__all__ = ["ffi"]
