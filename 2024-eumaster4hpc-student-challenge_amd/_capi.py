"""ctypes binding of include/lam_hip.h.  Mirrors the reference's solver interface
(LAM::ConjugateGradient<T>: solve / load_matrix_from_file / load_rhs_from_file /
save_result_to_file, plus generate_matrix / generate_rhs of the distributed classes --
/root/reference/challenge/main/LAM/src/ConjugateGradient.hpp:23-27,
LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:31-35) on top of the C ABI."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LAM_HIP_LIB: load another build of the same ABI instead (the tuning build liblam_hip_tuning.so, tools/gemv_probe.py)
_LIB = os.environ.get("LAM_HIP_LIB") or os.path.join(_HERE, "liblam_hip.so")
TUNING_LIB = os.path.join(_HERE, "liblam_hip_tuning.so")

F64, F32, BF16 = 0, 1, 2
ABI_VERSION = 4     # include/lam_hip.h LAM_HIP_ABI_VERSION
_VEC_DTYPE = {F64: np.float64, F32: np.float32, BF16: np.float32}
_HOST_MAT_DTYPE = {F64: np.float64, F32: np.float32, BF16: np.float32}


class LamHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lam_hip error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("num_iters", C.c_int32), ("converged", C.c_int32), ("rel_err", C.c_double),
                ("t_gemv", C.c_double), ("t_iter", C.c_double), ("t_total", C.c_double),
                ("t_comm_init", C.c_double), ("gemv_bytes", C.c_double), ("t_exchange", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def lib_path():
    return _LIB


def build(force=False):
    """Compile liblam_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = _source_files()
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building liblam_hip.so failed:\n" + r.stdout + r.stderr)
    return _LIB


_lib = None


def _source_files():
    """csrc/lam_hip.hip, every csrc/*.h it includes (sorted, as the Makefile's $(sort $(wildcard ...))), include/lam_hip.h."""
    import glob
    return ([os.path.join(_HERE, "csrc", "lam_hip.hip")] + sorted(glob.glob(os.path.join(_HERE, "csrc", "*.h")))
            + [os.path.join(os.path.dirname(_HERE), "include", "lam_hip.h")])


def source_id():
    """sha256 prefix of the sources the library is built from (same recipe as the Makefile's SRC_ID), or None when the
    sources are not next to the package."""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in _source_files():
            with open(f, "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


def lib():
    """Load the HIP library.  Fails loudly if it is missing: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} is not built (run `make -C {_HERE}` or __graft_entry__.build()); "
                              "the product has no CPU fallback")
        L = C.CDLL(_LIB, mode=C.RTLD_GLOBAL)
        vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
        sig = {
            "lam_hip_abi_version": ([], i32),
            "lam_hip_build_id": ([], C.c_char_p),
            "lam_hip_device_count": ([C.POINTER(i32)], i32),
            "lam_hip_create": ([C.POINTER(vp), i32, i32, C.POINTER(i32)], i32),
            "lam_hip_get_unique_id": ([vp], i32),
            "lam_hip_create_rank": ([C.POINTER(vp), i32, i32, i32, i32, vp], i32),
            "lam_hip_destroy": ([vp], None),
            "lam_hip_last_error": ([vp], C.c_char_p),
            "lam_hip_set_problem": ([vp, u64], i32),
            "lam_hip_partition": ([u64, i32, i32, C.POINTER(u64), C.POINTER(u64)], i32),
            "lam_hip_n": ([vp, C.POINTER(u64)], i32),
            "lam_hip_num_shards": ([vp, C.POINTER(i32), C.POINTER(i32)], i32),
            "lam_hip_get_partition": ([vp, i32, C.POINTER(u64), C.POINTER(u64)], i32),
            "lam_hip_upload_rows": ([vp, u64, u64, vp], i32),
            "lam_hip_download_rows": ([vp, u64, u64, vp], i32),
            "lam_hip_generate_tridiag": ([vp], i32),
            "lam_hip_generate_random_spd": ([vp, u64, C.c_double], i32),
            "lam_hip_generate_spectrum_spd": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_double), i32], i32),
            "lam_hip_set_rhs": ([vp, vp], i32),
            "lam_hip_get_rhs": ([vp, vp], i32),
            "lam_hip_generate_rhs": ([vp, C.c_double], i32),
            "lam_hip_generate_random_rhs": ([vp, u64], i32),
            "lam_hip_solve": ([vp, i32, C.c_double, C.POINTER(Stats)], i32),
            "lam_hip_cg_init": ([vp], i32),
            "lam_hip_cg_iterate": ([vp, i32, C.c_double, C.POINTER(Stats)], i32),
            "lam_hip_get_solution": ([vp, vp], i32),
            "lam_hip_true_residual": ([vp, C.POINTER(C.c_double)], i32),
            "lam_hip_gemv": ([vp, vp, vp], i32),
            "lam_hip_gemv_only": ([vp, i32, C.POINTER(C.c_double)], i32),
            "lam_hip_check_symmetry": ([vp, C.POINTER(C.c_double)], i32),
            "lam_hip_debug_symv_plan": ([u64, i32, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)], i32),
            "lam_hip_dot": ([vp, vp, vp, u64, C.POINTER(C.c_double)], i32),
            "lam_hip_axpby": ([vp, C.c_double, vp, C.c_double, vp, u64], i32),
            "lam_hip_all_ok": ([vp, i32, C.POINTER(i32)], i32),
            "lam_hip_rccl_version": ([C.POINTER(i32)], i32),
            "lam_hip_gemv_kernel_name": ([vp, C.c_char_p, C.c_size_t], i32),
            "lam_hip_set_option": ([vp, C.c_char_p, C.c_int64], i32),
            "lam_hip_get_option": ([vp, C.c_char_p, C.POINTER(C.c_int64)], i32),
        }
        for name, (args, res) in sig.items():
            fn = getattr(L, name)      # AttributeError if the library does not export it
            fn.argtypes, fn.restype = args, res
        L._lam_symbols = tuple(sig)
        if L.lam_hip_abi_version() != ABI_VERSION:      # the Stats struct above belongs to one ABI version
            raise ImportError(f"{_LIB} has ABI {L.lam_hip_abi_version()}, this binding was written for ABI {ABI_VERSION}")
        built, want = L.lam_hip_build_id().decode(), source_id()
        if want is not None and built != want and not os.environ.get("LAM_HIP_ALLOW_STALE"):
            raise ImportError(f"{_LIB} was built from other sources (library {built}, sources {want}): rebuild it "
                              f"(`make -C {_HERE} all tuning` or __graft_entry__.build())")
        _lib = L
    return _lib


def device_count():
    n = C.c_int(0)
    rc = lib().lam_hip_device_count(C.byref(n))
    if rc != 0:
        raise LamHipError(rc, (lib().lam_hip_last_error(None) or b"").decode())
    return n.value


def get_unique_id():
    buf = C.create_string_buffer(128)
    rc = lib().lam_hip_get_unique_id(buf)
    if rc != 0:
        raise LamHipError(rc, (lib().lam_hip_last_error(None) or b"").decode())
    return buf.raw


def rccl_version():
    v = C.c_int(0)
    rc = lib().lam_hip_rccl_version(C.byref(v))
    if rc != 0:
        raise LamHipError(rc, (lib().lam_hip_last_error(None) or b"").decode())
    return v.value


def symv_plan_check(n, shards, dtype=0):
    """Host-only check of the symmetric product's task plan (lam_hip_debug_symv_plan): (bad_pairs, bad_interior, tasks)."""
    bp, bi, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib().lam_hip_debug_symv_plan(n, shards, dtype, C.byref(bp), C.byref(bi), C.byref(nt))
    if rc != 0:
        raise LamHipError(rc, "lam_hip_debug_symv_plan")
    return bp.value, bi.value, nt.value


def partition(n, num_shards, shard):
    """Rows (row0, nrows) of shard `shard` of `num_shards` -- the reference's block-row rule."""
    r0, nr = C.c_uint64(), C.c_uint64()
    rc = lib().lam_hip_partition(n, num_shards, shard, C.byref(r0), C.byref(nr))
    if rc != 0:
        raise LamHipError(rc, "bad partition arguments")
    return r0.value, nr.value


def _read_bin(path, dtype):
    """(rows, cols) of a matrix / vector file.  Same rule as LAM::parse_bin_header (the C++ loaders): the
    reference's writers store an `int` with sizeof(size_t) -- cols in ConjugateGradient_CPU_OMP.hpp:206-210,
    both words in ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:754-757 -- leaving garbage in the upper 32 bits, so
    take the words as they are if the file is long enough for them, else their low halves if those fit."""
    size = os.path.getsize(path)
    esz = np.dtype(dtype).itemsize
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.uint64)
    if hdr.size != 2:
        raise IOError("short header")
    for rows, cols in ((int(hdr[0]), int(hdr[1])), (int(hdr[0]) & 0xFFFFFFFF, int(hdr[1]) & 0xFFFFFFFF)):
        if rows > 0 and cols > 0 and rows * cols * esz <= size - 16:
            return rows, cols
    raise IOError("file is shorter than its header says")


class Solver:
    """Python face of the reference's solver classes.

    Solver(dtype, n_shards=1, device_ids=None)            one process, one or more shards
    Solver(dtype, rank=r, nranks=P, device_id=d, unique_id=..)  one process per GPU (RCCL)
    """

    def __init__(self, dtype=F64, n_shards=1, device_ids=None, rank=None, nranks=None, device_id=0,
                 unique_id=None):
        self._L = lib()
        self._h = C.c_void_p()
        self.dtype = dtype
        self.rank = 0 if rank is None else rank
        if rank is None:
            ids = None
            if device_ids is not None:
                ids = (C.c_int * len(device_ids))(*device_ids)
                n_shards = len(device_ids)
            rc = self._L.lam_hip_create(C.byref(self._h), dtype, n_shards, ids)
        else:
            rc = self._L.lam_hip_create_rank(C.byref(self._h), dtype, device_id, rank, nranks, unique_id)
        if rc != 0:
            raise LamHipError(rc, (self._L.lam_hip_last_error(None) or b"").decode())
        self.vec_dtype = _VEC_DTYPE[dtype]
        self.mat_host_dtype = _HOST_MAT_DTYPE[dtype]

    # -- plumbing -------------------------------------------------------------------------------
    def _chk(self, rc):
        if rc != 0:
            raise LamHipError(rc, (self._L.lam_hip_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.lam_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- problem --------------------------------------------------------------------------------
    def set_problem(self, n):
        self._chk(self._L.lam_hip_set_problem(self._h, n))
        self.n = n

    def num_shards(self):
        t, l = C.c_int(), C.c_int()
        self._chk(self._L.lam_hip_num_shards(self._h, C.byref(t), C.byref(l)))
        return t.value, l.value

    def partition(self, shard):
        r0, nr = C.c_uint64(), C.c_uint64()
        self._chk(self._L.lam_hip_get_partition(self._h, shard, C.byref(r0), C.byref(nr)))
        return r0.value, nr.value

    def upload_rows(self, row0, rows):
        rows = np.ascontiguousarray(rows, dtype=self.mat_host_dtype)
        self._chk(self._L.lam_hip_upload_rows(self._h, row0, rows.shape[0], rows.ctypes.data_as(C.c_void_p)))

    def download_rows(self, row0, nrows):
        out = np.empty((nrows, self.n), dtype=self.mat_host_dtype)
        self._chk(self._L.lam_hip_download_rows(self._h, row0, nrows, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_matrix(self, A):
        """Upload a full host matrix (single-process contexts)."""
        A = np.ascontiguousarray(A, dtype=self.mat_host_dtype)
        assert A.ndim == 2 and A.shape[0] == A.shape[1]
        self.set_problem(A.shape[0])
        self.upload_rows(0, A)

    def generate_matrix(self, rows, cols=None):
        """generate_matrix(rows, cols): dense tridiag(1,2,1) (CPU_MPI_OMP.hpp:167-256)."""
        if cols is not None and cols != rows:
            raise ValueError("Matrix has to be square")
        self.set_problem(rows)
        self._chk(self._L.lam_hip_generate_tridiag(self._h))
        return True

    def generate_random_spd(self, n, seed, cond, keep_problem=False):
        """keep_problem: a context that already holds a problem of this size keeps its allocations (no lam_hip_set_problem: its
        hipFree waits for the whole device, which deadlocks rank contexts that are THREADS of one process -- the test harness's
        shape -- as soon as another rank has a collective in flight that waits for this one)."""
        if not (keep_problem and getattr(self, "n", None) == n):
            self.set_problem(n)
        self._chk(self._L.lam_hip_generate_random_spd(self._h, seed, cond))

    def generate_spectrum_spd(self, eig, reflectors):
        """A = H_k..H_1 diag(eig) H_1..H_k (the reference generator's law, random_spd_system.cpp:66-97, with Householder
        reflectors for Q); `reflectors`: k x N array (k may be 0: the diagonal matrix itself)."""
        eig = np.ascontiguousarray(eig, dtype=np.float64).reshape(-1)
        V = np.ascontiguousarray(reflectors, dtype=np.float64).reshape(-1, eig.size) if np.size(reflectors) else np.zeros((0, eig.size))
        self.set_problem(eig.size)
        dp = C.POINTER(C.c_double)
        self._chk(self._L.lam_hip_generate_spectrum_spd(self._h, eig.ctypes.data_as(dp), V.ctypes.data_as(dp), V.shape[0]))

    def generate_rhs(self, value=1.0):
        self._chk(self._L.lam_hip_generate_rhs(self._h, value))
        return True

    def generate_random_rhs(self, seed):
        self._chk(self._L.lam_hip_generate_random_rhs(self._h, seed))

    def set_rhs(self, b):
        b = np.ascontiguousarray(b, dtype=self.vec_dtype).reshape(-1)
        assert b.size == self.n
        self._chk(self._L.lam_hip_set_rhs(self._h, b.ctypes.data_as(C.c_void_p)))

    def rhs(self):
        b = np.empty(self.n, dtype=self.vec_dtype)
        self._chk(self._L.lam_hip_get_rhs(self._h, b.ctypes.data_as(C.c_void_p)))
        return b

    # -- file mode (format: random_spd_system.cpp:105-121) ------------------------------------------
    def load_matrix_from_file(self, filename, chunk_bytes=256 << 20):
        try:
            rows, cols = _read_bin(filename, np.float64 if self.dtype == F64 else np.float32)
        except OSError:
            return False
        if rows != cols:
            return False            # "Matrix has to be square" (CPU_OMP.hpp:151-155)
        self.set_problem(rows)
        total, local = self.num_shards()
        es = 8 if self.dtype == F64 else 4
        file_dtype = np.float64 if self.dtype == F64 else np.float32
        mm = np.memmap(filename, dtype=file_dtype, mode="r", offset=16, shape=(rows, cols))
        owned = [self.partition(q) for q in range(total)] if local == total else [self.partition(self.rank)]
        step = max(1, chunk_bytes // (cols * es))
        for r0, nr in owned:
            for s in range(r0, r0 + nr, step):
                e = min(s + step, r0 + nr)
                self.upload_rows(s, np.asarray(mm[s:e]))
        return True

    def load_rhs_from_file(self, filename):
        try:
            rows, cols = _read_bin(filename, np.float64 if self.dtype == F64 else np.float32)
        except OSError:
            return False
        if cols != 1 or rows != self.n:
            return False            # CPU_MPI_OMP.hpp:278-287
        file_dtype = np.float64 if self.dtype == F64 else np.float32
        b = np.fromfile(filename, dtype=file_dtype, offset=16, count=rows)
        self.set_rhs(b)
        return True

    def save_result_to_file(self, filename):
        x = self.solution()
        try:
            with open(filename, "wb") as f:
                f.write(np.array([x.size, 1], dtype=np.uint64).tobytes())   # clean cols word
                f.write(x.tobytes())
        except OSError:
            return False
        return True

    # -- hot path -------------------------------------------------------------------------------
    def solve(self, max_iters, rel_error):
        st = Stats()
        self._chk(self._L.lam_hip_solve(self._h, max_iters, rel_error, C.byref(st)))
        self.stats = st.asdict()
        return bool(st.converged)

    def cg_init(self):
        self._chk(self._L.lam_hip_cg_init(self._h))

    def cg_iterate(self, iters, rel_error=0.0):
        st = Stats()
        self._chk(self._L.lam_hip_cg_iterate(self._h, iters, rel_error, C.byref(st)))
        self.stats = st.asdict()
        return self.stats

    def solution(self):
        x = np.empty(self.n, dtype=self.vec_dtype)
        self._chk(self._L.lam_hip_get_solution(self._h, x.ctypes.data_as(C.c_void_p)))
        return x

    def true_residual(self):
        v = C.c_double()
        self._chk(self._L.lam_hip_true_residual(self._h, C.byref(v)))
        return v.value

    # -- single operators -----------------------------------------------------------------------
    def gemv(self, x):
        x = np.ascontiguousarray(x, dtype=self.vec_dtype).reshape(-1)
        assert x.size == self.n
        y = np.empty(self.n, dtype=self.vec_dtype)
        self._chk(self._L.lam_hip_gemv(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
        return y

    def gemv_only(self, reps):
        v = C.c_double()
        self._chk(self._L.lam_hip_gemv_only(self._h, reps, C.byref(v)))
        return v.value

    def dot(self, x, y):
        x = np.ascontiguousarray(x, dtype=self.vec_dtype).reshape(-1)
        y = np.ascontiguousarray(y, dtype=self.vec_dtype).reshape(-1)
        v = C.c_double()
        self._chk(self._L.lam_hip_dot(self._h, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size,
                                      C.byref(v)))
        return v.value

    def axpby(self, alpha, x, beta, y):
        x = np.ascontiguousarray(x, dtype=self.vec_dtype).reshape(-1)
        y = np.ascontiguousarray(y, dtype=self.vec_dtype).reshape(-1).copy()
        self._chk(self._L.lam_hip_axpby(self._h, alpha, x.ctypes.data_as(C.c_void_p), beta,
                                        y.ctypes.data_as(C.c_void_p), x.size))
        return y

    def check_symmetry(self):
        v = C.c_double()
        self._chk(self._L.lam_hip_check_symmetry(self._h, C.byref(v)))
        return v.value

    def all_ok(self, ok):
        g = C.c_int(0)
        self._chk(self._L.lam_hip_all_ok(self._h, 1 if ok else 0, C.byref(g)))
        return bool(g.value)

    def gemv_kernel_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self._L.lam_hip_gemv_kernel_name(self._h, buf, 256))
        return buf.value.decode()

    def get_option(self, name):
        v = C.c_int64()
        self._chk(self._L.lam_hip_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def set_option(self, name, value):
        self._chk(self._L.lam_hip_set_option(self._h, name.encode(), value))
