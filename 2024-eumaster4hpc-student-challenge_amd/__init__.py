"""MI355X-native dense Conjugate-Gradient hot path (gfx950 HIP kernels behind a C ABI).

Python is only a thin ctypes binding over ``liblam_hip.so`` (see include/lam_hip.h) used by the
tests and bench.py; the host-side drop-in for the reference's C++ classes lives in ``LAM/`` and
``test/`` next to this file.  There is no CPU fallback: without the built library, or without a
gfx950 GPU, every compute call raises.
"""
from ._capi import (  # noqa: F401
    LamHipError, Stats, Solver, build, lib, lib_path, device_count, get_unique_id, partition, rccl_version, symv_plan_check,
    F64, F32, BF16, TUNING_LIB,
)
from ._rendezvous import Rendezvous, launched_with_ranks  # noqa: F401,E402
