"""pandepth_amd — MI355X-native per-base depth engine (PanDepth-compatible).

The product is the C-ABI shared library `libpandepth_amd.so` (include/pandepth_amd.h) built from
hand-written gfx950 HIP kernels, plus the C++ host side (`pandepth` CLI).  This package is only
the thin ctypes mirror of that ABI used by tests and bench.py.  It never falls back to a CPU
implementation: if the library is missing or no gfx950 device is present, it raises.
"""
from .capi import Comm, comm_unique_id, Engine, PdError, lib_path, load, PD_PUSH_SORTED, PD_PUSH_DEFAULT, PD_PUSH_DISORDER, PD_PUSH_MORE  # noqa: F401

__version__ = "0.1.0"
