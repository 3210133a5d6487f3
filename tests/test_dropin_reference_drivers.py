"""The reference's OWN GPU driver sources (test_CG_MultiGPUS_CUDA_NCCL.cpp, _CUDA_MPI.cpp,
test_CG_single_GPU.cpp, test_CG_MultiGPUS_CUDA.cpp under /root/reference/challenge/main/test/) compiled
UNCHANGED from where they lie against this repository's LAM headers (-DUSE_HIP) and liblam_hip.so:
oracle/Makefile target `dropin`, outputs in oracle/_ref/dropin/ (git-ignored; they travel to the GPU box).
What it pins: the drop-in claim of LAM/src/ConjugateGradient.hpp -- class names, default constructors,
method signatures, PRINT_RANK0 macros, CSV fragments printed by the classes -- is compiled and run, not
asserted.  The CPU half builds them (needs /root/reference); the GPU half runs what was built."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

DROPIN = os.path.join(ROOT, "oracle", "_ref", "dropin")
NAMES = ["test_CG_MultiGPUS_CUDA_NCCL.out", "test_CG_MultiGPUS_CUDA_MPI.out", "test_CG_single_GPU.out", "test_CG_MultiGPUS_CUDA.out"]
REF_TESTS = "/root/reference/challenge/main/test"


def test_reference_driver_sources_compile_against_hip_headers(lam):
    if not os.path.isdir(REF_TESTS):
        pytest.skip("the reference checkout is not present (GPU box): nothing to compile")
    lam.build()
    for n in NAMES:                       # force a rebuild so that a header regression shows up
        try:
            os.remove(os.path.join(DROPIN, n))
        except OSError:
            pass
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "dropin"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for n in NAMES:
        assert os.path.exists(os.path.join(DROPIN, n)), n


def _need(name):
    exe = os.path.join(DROPIN, name)
    if not os.path.exists(exe):
        pytest.skip(f"{name} was not built (make -C oracle dropin needs the reference checkout)")
    return exe


def _env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    return e


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test_CG_single_GPU.out", "test_CG_MultiGPUS_CUDA.out"])
def test_reference_positional_driver_runs_on_hip_classes(name, golden, oracle, tmp_path):
    """`ConjugateGradient_GPU_CUDA<double> CG_P;` / `ConjugateGradient_MultiGPUS_CUDA<double> CG_P;` in the
    reference's driver are the HIP classes here.  (Its PRINT_RANK0 lines depend on an uninitialised `rank`,
    test_CG_single_GPU.cpp:13-15, so only the files and the exit code are checked.)"""
    exe = _need(name)
    g = next(x for x in golden["file_mode"] if x["n"] == 256)
    sol = tmp_path / "sol.bin"
    r = subprocess.run([exe, os.path.join(GOLDEN, g["name"] + ".matrix.bin"), os.path.join(GOLDEN, g["name"] + ".rhs.bin"),
                        str(sol), str(g["max_iters"]), repr(g["tol"])], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    x = oracle.read_bin(str(sol)).reshape(-1)
    x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) <= 10 * g["tol"]   # 1e-8 at tol 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test_CG_MultiGPUS_CUDA_NCCL.out", "test_CG_MultiGPUS_CUDA_MPI.out"])
def test_reference_getopt_driver_runs_on_hip_classes(name, tmp_path):
    """The reference's distributed driver: MPI_Init in ITS main, a default-constructed class, CSV assembled
    from the driver's and the class's prints.  One rank (MPI singleton), generate mode known answer."""
    exe = _need(name)
    r = subprocess.run([exe, "-s", "4096", "-i", "15", "-o", str(tmp_path / "sol.bin")], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    f = r.stdout.replace("\n", "").strip().split(",")
    # N, procs, omp_threads, gen_s, comm_init_s, avg_gemv, avg_iter, iters, err, cg_s(truncated to whole seconds)
    assert len(f) == 10 and f[0] == "4096" and f[1] == "1", f
    assert int(f[7]) == 16 and abs(float(f[8]) / 0.000368282 - 1) < 1e-5
    hdr = np.fromfile(tmp_path / "sol.bin", dtype=np.uint64, count=2)
    assert hdr.tolist() == [4096, 1]


@pytest.mark.gpu
def test_reference_getopt_driver_two_ranks_under_mpiexec(tmp_path, mock_mp_lib):
    """The same unchanged reference driver on 2 ranks: rank/size/id travel through the MPI its main()
    initialised (lam_bootstrap::attach), file-free; RCCL is the multi-process test double (one GPU)."""
    exe = _need("test_CG_MultiGPUS_CUDA_NCCL.out")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec")
    r = subprocess.run([mpiexec, "-n", "2", "-genv", "LD_PRELOAD", mock_mp_lib, exe, "-s", "2048", "-o", str(tmp_path / "sol.bin")],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    f = r.stdout.replace("\n", "").strip().split(",")
    assert f[0] == "2048" and f[1] == "2" and int(f[7]) == 1024 and float(f[8]) < 1e-9, f
