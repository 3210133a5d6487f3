"""The product's HOST-ONLY logic under AddressSanitizer + UBSan.  The symmetric product's task planner, the
reference's row partition and the device row pitch (csrc/lam_host_plan.h -- the very header the product build includes), the file
loaders of the solver classes (LAM/src/HIP/ConjugateGradient_HIP_base.hpp) and the launcher glue (lam_bootstrap.hpp) are plain
host code whose bugs become out-of-bounds accesses on the DEVICE, where no sanitizer is available on this pool.  tests/host_asan
builds them with g++ -fsanitize=address,undefined against a host-memory fake of the C ABI and runs: the exhaustive plan check, the
loaders on good / truncated / oversized / garbage-header / non-square / empty files in every topology and precision (with a 4-KiB
chunk so that the chunk loop is walked), and the rendezvous of two processes.  Reference behaviour mirrored by the loaders:
/root/reference/challenge/main/LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:307-417 (whose `int` counts overflow past 2^31)."""
import os
import subprocess

import pytest

from conftest import ROOT

HERE = os.path.join(ROOT, "tests", "host_asan")
EXE = os.path.join(HERE, "host_asan.out")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def host_asan():
    r = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return EXE


def _clean(r):
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]


def test_the_product_build_includes_the_same_planner_header():
    """One definition: csrc/lam_hip.hip reaches lam_host_plan.h through lam_kernels.h, the launcher and the C ABI's plan check use
    its symv_plan / symv_plan_check, and nothing in the header needs HIP."""
    src = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "csrc")
    assert '#include "lam_host_plan.h"' in open(os.path.join(src, "lam_kernels.h")).read()
    assert "symv_plan(n, ncv, SS, s.row0, s.nrows, cyc, &plan)" in open(os.path.join(src, "lam_launch.h")).read()
    assert "lam::symv_plan_check(" in open(os.path.join(src, "lam_hip.hip")).read()
    header = open(os.path.join(src, "lam_host_plan.h")).read()
    assert "hip_runtime" not in header and "#include <hip" not in header
    for other in ("lam_launch.h", "lam_ctx.h", "lam_hip.hip", "lam_kernels.h"):
        assert "void symv_plan(" not in open(os.path.join(src, other)).read(), other       # no second copy of the planner


def test_planner_partition_and_pitch_under_sanitizers(host_asan):
    r = subprocess.run([host_asan, "plan"], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok plan"), r.stdout[-2000:] + r.stderr[-3000:]


def test_file_loaders_under_sanitizers(host_asan, tmp_path):
    r = subprocess.run([host_asan, "loaders", str(tmp_path)], capture_output=True, text=True, timeout=600, env=ENV)
    _clean(r)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok loaders"), r.stdout[-2000:] + r.stderr[-3000:]
    # the reference's wording on stderr for the refused files
    for msg in ("Matrix has to be square", "Size of right hand side does not match the matrix", "Cannot open output file"):
        assert msg in r.stderr


def test_launcher_glue_under_sanitizers(host_asan, tmp_path):
    """lam_bootstrap::init in three launches: alone; two ranks that agree on the unique id through the rendezvous file (rank 1 starts
    first and has to wait for it); and a rank whose rank 0 never shows up is not left hanging for ever (bounded wait -- not run
    here: 60 s)."""
    env = {k: v for k, v in ENV.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE")}
    r = subprocess.run([host_asan, "bootstrap"], capture_output=True, text=True, timeout=60, env=env)
    _clean(r)
    assert r.returncode == 0 and r.stdout.split()[:3] == ["0", "1", "0"], r.stdout + r.stderr
    idf = str(tmp_path / "id")
    procs = []
    for rank in (1, 0):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), LAM_RCCL_ID_FILE=idf)
        procs.append(subprocess.Popen([host_asan, "bootstrap"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o + e
        assert "ERROR: AddressSanitizer" not in e and "runtime error" not in e, e
    f1, f0 = outs[0][0].split(), outs[1][0].split()
    assert f1[:3] == ["1", "2", "1"] and f0[:3] == ["0", "2", "0"]
    want = "".join(f"{(0xA5 ^ (i * 7)) & 0xff:02x}" for i in range(128))          # the fake's lam_hip_get_unique_id
    assert f0[3] == f1[3] == want
    assert not os.path.exists(idf)                      # rank 0 removed the rendezvous file


def test_heat_assembler_under_sanitizers(host_asan, tmp_path):
    """apps/heat_system.cpp (BASELINE configs[4]: the 2-D heat problem as a dense SPD system, file format of the reference) is host
    code that indexes an n x n matrix with n = (nx-2)(ny-2): the sanitized build must write byte for byte what the committed
    12 x 12 fixture holds, handle a non-square grid, and refuse bad arguments without touching memory it does not own."""
    from conftest import GOLDEN
    exe = os.path.join(HERE, "heat_asan.out")
    m, b = str(tmp_path / "m.bin"), str(tmp_path / "b.bin")
    r = subprocess.run([exe, "assemble", "12", "12", m, b], capture_output=True, text=True, timeout=120, env=ENV)
    _clean(r)
    assert r.returncode == 0, r.stdout + r.stderr
    assert open(m, "rb").read() == open(os.path.join(GOLDEN, "heat_12x12.matrix.bin"), "rb").read()
    assert open(b, "rb").read() == open(os.path.join(GOLDEN, "heat_12x12.rhs.bin"), "rb").read()
    r = subprocess.run([exe, "assemble", "7", "19", m, b], capture_output=True, text=True, timeout=120, env=ENV)
    _clean(r)
    assert r.returncode == 0 and os.path.getsize(m) == 16 + 8 * (5 * 17) ** 2 and os.path.getsize(b) == 16 + 8 * 5 * 17
    # a solution vector back on the grid (the reference's output format), then the refusals
    r = subprocess.run([exe, "field", "7", "19", b, str(tmp_path / "heat.bin")], capture_output=True, text=True, timeout=120, env=ENV)
    _clean(r)
    assert r.returncode == 0, r.stdout + r.stderr
    for bad in (["assemble", "2", "12", m, b], ["assemble", "12"], ["field", "7", "19", str(tmp_path / "missing.bin"), str(tmp_path / "h.bin")], ["nonsense"]):
        r = subprocess.run([exe] + bad, capture_output=True, text=True, timeout=60, env=ENV)
        _clean(r)
        assert r.returncode != 0, bad
