"""ctypes front-end of the CPU parity oracle (oracle/cg_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See cg_oracle.c for the reference
file:line each function restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_DIR = os.path.join(_HERE, "_ref")


class OracleStats(C.Structure):
    _fields_ = [("num_iters", C.c_int), ("converged", C.c_int), ("rel_err", C.c_double),
                ("t_total", C.c_double), ("t_gemv", C.c_double)]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cg_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        for suf, ct in (("f64", C.c_double), ("f32", C.c_float)):
            p = C.c_void_p
            getattr(L, f"oracle_gemv_{suf}").argtypes = [ct, p, p, ct, p, C.c_size_t, C.c_size_t, C.c_int]
            getattr(L, f"oracle_gemv_{suf}").restype = None
            getattr(L, f"oracle_dot_{suf}").argtypes = [p, p, C.c_size_t, C.c_int]
            getattr(L, f"oracle_dot_{suf}").restype = ct
            getattr(L, f"oracle_axpby_{suf}").argtypes = [ct, p, ct, p, C.c_size_t, C.c_int]
            getattr(L, f"oracle_axpby_{suf}").restype = None
            getattr(L, f"oracle_generate_tridiag_{suf}").argtypes = [p, C.c_size_t, C.c_size_t, C.c_size_t]
            getattr(L, f"oracle_generate_tridiag_{suf}").restype = None
            getattr(L, f"oracle_cg_solve_{suf}").argtypes = [p, p, p, C.c_size_t, C.c_int, ct, C.c_int,
                                                            C.POINTER(OracleStats)]
            getattr(L, f"oracle_cg_solve_sharded_{suf}").argtypes = [p, p, p, C.c_size_t, C.c_int, C.c_int, ct,
                                                                    C.POINTER(OracleStats)]
            getattr(L, f"oracle_cpu_baseline_{suf}").argtypes = [C.c_size_t, C.c_int, C.c_int,
                                                                C.POINTER(OracleStats)]
        L.oracle_partition.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.oracle_partition.restype = None
        _lib = L
    return _lib


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64", C.c_double
    if dtype == np.float32:
        return "f32", C.c_float
    raise TypeError(dtype)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def gemv(A, x, alpha=1.0, beta=0.0, y=None, threads=1):
    A = np.ascontiguousarray(A)
    suf, ct = _suf(A.dtype)
    x = np.ascontiguousarray(x, dtype=A.dtype)
    rows, cols = A.shape
    y = np.zeros(rows, dtype=A.dtype) if y is None else np.ascontiguousarray(y, dtype=A.dtype).copy()
    getattr(lib(), f"oracle_gemv_{suf}")(ct(alpha), _ptr(A), _ptr(x), ct(beta), _ptr(y), rows, cols, threads)
    return y


def dot(x, y, threads=1):
    x = np.ascontiguousarray(x)
    suf, _ = _suf(x.dtype)
    y = np.ascontiguousarray(y, dtype=x.dtype)
    return getattr(lib(), f"oracle_dot_{suf}")(_ptr(x), _ptr(y), x.size, threads)


def axpby(alpha, x, beta, y, threads=1):
    x = np.ascontiguousarray(x)
    suf, ct = _suf(x.dtype)
    y = np.ascontiguousarray(y, dtype=x.dtype).copy()
    getattr(lib(), f"oracle_axpby_{suf}")(ct(alpha), _ptr(x), ct(beta), _ptr(y), x.size, threads)
    return y


def tridiag(n, row0=0, nrows=None, dtype=np.float64):
    nrows = n if nrows is None else nrows
    suf, _ = _suf(dtype)
    A = np.empty((nrows, n), dtype=dtype)
    getattr(lib(), f"oracle_generate_tridiag_{suf}")(_ptr(A), row0, nrows, n)
    return A


def partition(n, P, q):
    r0, nr = C.c_size_t(), C.c_size_t()
    lib().oracle_partition(n, P, q, C.byref(r0), C.byref(nr))
    return r0.value, nr.value


def cg_solve(A, b, max_iters, rel_error, threads=1, P=None):
    """Returns (x, stats dict).  P=None: single-process recurrence; P>=1: emulated MPI ranks."""
    A = np.ascontiguousarray(A)
    suf, ct = _suf(A.dtype)
    b = np.ascontiguousarray(b, dtype=A.dtype).reshape(-1)
    n = b.size
    assert A.shape == (n, n)
    x = np.zeros(n, dtype=A.dtype)
    st = OracleStats()
    if P is None:
        rc = getattr(lib(), f"oracle_cg_solve_{suf}")(_ptr(A), _ptr(b), _ptr(x), n, max_iters, ct(rel_error),
                                                     threads, C.byref(st))
    else:
        rc = getattr(lib(), f"oracle_cg_solve_sharded_{suf}")(_ptr(A), _ptr(b), _ptr(x), n, P, max_iters,
                                                             ct(rel_error), C.byref(st))
    if rc < 0:
        raise MemoryError("oracle allocation failed")
    return x, dict(num_iters=st.num_iters, converged=bool(st.converged), rel_err=st.rel_err,
                   t_total=st.t_total, t_gemv=st.t_gemv)


def cpu_baseline(n, iters, threads, dtype=np.float64):
    suf, _ = _suf(dtype)
    st = OracleStats()
    rc = getattr(lib(), f"oracle_cpu_baseline_{suf}")(n, iters, threads, C.byref(st))
    if rc < 0:
        raise MemoryError("oracle allocation failed")
    return dict(num_iters=st.num_iters, rel_err=st.rel_err, t_total=st.t_total, t_gemv=st.t_gemv)


# ---- file format helpers (numpy side; format: see cg_oracle.c oracle_read_header) ----
def read_bin(path, dtype=np.float64):
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.uint64)
        rows, cols = int(hdr[0]), int(hdr[1]) & 0xFFFFFFFF
        data = np.frombuffer(f.read(rows * cols * np.dtype(dtype).itemsize), dtype=dtype)
    return data.reshape(rows, cols).copy()


def write_bin(path, arr):
    arr = np.ascontiguousarray(arr)
    if arr.ndim == 1:
        arr = arr.reshape(-1, 1)
    with open(path, "wb") as f:
        f.write(np.array(arr.shape, dtype=np.uint64).tobytes())
        f.write(arr.tobytes())
