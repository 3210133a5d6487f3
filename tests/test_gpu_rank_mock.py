"""Rank mode (one context per GPU rank, RCCL exchange) with P > 1 on a ONE-GPU box.

RCCL refuses two ranks on one device, so the P ranks run as P threads of one process on GPU 0 with
tests/mock_rccl/libmock_rccl.so LD_PRELOADed in front of librccl.so: a host-synchronous stand-in
that implements the six collectives' data semantics and verifies that every rank issues the same
call sequence with the same counts.  What this pins down is liblam_hip.so's own rank-mode logic:
partition by rank, in-place all-gather offsets, the grouped broadcasts of the uneven split, the
own-slice GEMV panel + accumulate (overlap path), identical stop decisions, the collective gathers
of x.  (The real RCCL calls are exercised with a 1-rank communicator in test_gpu_drivers.py.)"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
MOCK_DIR = os.path.join(ROOT, "tests", "mock_rccl")
MOCK = os.path.join(MOCK_DIR, "libmock_rccl.so")


@pytest.fixture(scope="module")
def mock_lib():
    src = os.path.join(MOCK_DIR, "mock_rccl.cpp")
    if not os.path.exists(MOCK) or os.path.getmtime(MOCK) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", src, "-o", MOCK], check=True)
    return MOCK


@pytest.mark.parametrize("P,n,mode,overlap,exchange", [
    (2, 1024, "tridiag", 1, 0),     # even split, aligned panels
    (3, 1001, "tridiag", 1, 0),     # odd N: generic kernel; uneven split: grouped broadcasts
    (4, 4096, "spd", 1, 0),
    (4, 4096, "spd", 0, 0),         # all-gather on the compute stream
    (3, 4098, "spd", 1, 0),         # 1366 rows per rank
    (4, 4102, "spd", 1, 0),         # remainder on the last rank, odd row offsets -> no panel split
    (8, 8192, "spd", 1, 0),         # the node shape
    # exchange = 1: ONE all-gather of [Ap slice | p.Ap partial] per iteration, full-length r and p per rank
    (2, 1024, "tridiag", 1, 1),
    (4, 4096, "spd", 1, 1),
    (8, 8192, "spd", 1, 1),
    (3, 1001, "tridiag", 1, 1),     # uneven split: falls back to exchange 0
])
def test_rank_mode_multi_rank_on_mock_rccl(mock_lib, P, n, mode, overlap, exchange):
    env = dict(os.environ, LD_PRELOAD=mock_lib)
    r = subprocess.run([sys.executable, os.path.join(MOCK_DIR, "run_ranks.py"), str(P), str(n), mode, str(overlap),
                        str(exchange)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ranks_identical"], out          # every rank holds the same x, iteration count and residual
    assert out["converged"]
    assert abs(out["iters"] - out["iters_single"]) <= max(3, 0.02 * out["iters_single"]), out
    assert out["true_residual"] <= 2 * (1e-9 if mode == "tridiag" else 1e-10) + 1e-13
    assert out["x_vs_single"] < (1e-5 if mode == "tridiag" else 1e-8), out   # tridiag(1,2,1): cond ~ N^2/2
    assert out["gemv_vs_single"] < 1e-13
    base = n // P
    assert out["partition"] == [[q * base, base + (n % P if q == P - 1 else 0)] for q in range(P)]


MOCK_MP = os.path.join(MOCK_DIR, "libmock_rccl_mp.so")


@pytest.fixture(scope="module")
def mock_mp_lib():
    src = os.path.join(MOCK_DIR, "mock_rccl_mp.cpp")
    if not os.path.exists(MOCK_MP) or os.path.getmtime(MOCK_MP) < os.path.getmtime(src):
        # g++ and NOT linked against libamdhip64: the HIP symbols bind at first use to whatever runtime
        # the process already holds (torch's, in the torchrun path) instead of dragging in a second one
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        src, "-o", MOCK_MP, "-lrt", "-lpthread"], check=True)
    return MOCK_MP


@pytest.mark.parametrize("nproc", [2, 4])
def test_bench_torchrun_path_end_to_end_on_mock_rccl(mock_mp_lib, nproc):
    """The exact command line the driver uses for N > 1 GPUs -- torch.distributed.run, one process per
    rank, gloo control plane, unique-id broadcast, both exchange modes, max-over-ranks timing, one JSON
    line from rank 0 -- with every rank on GPU 0 and the multi-process mock in front of librccl."""
    env = dict(os.environ, LD_PRELOAD=mock_mp_lib)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    port = 29700 + nproc + os.getpid() % 100
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "20",
           "--warmup", "3", "--order", "8192"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                          # exactly one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == nproc and out["steps"] == 20 and out["scaling"] == "strong" and out["dtype"] == "f64"
    assert out["metric"] == "cg_iterations_per_sec" and out["value"] > 0
    assert set(out["exchange_modes"]) == {"allreduce_x2+allgather_p", "allreduce_x2+allgather_p, no overlap", "allgather_Ap"}
    for m in out["exchange_modes"].values():
        assert m["value"] > 0
    # both exchanges solved the same problem: true residuals agree (different rounding only)
    res = [m["rel_residual_true"] for m in out["exchange_modes"].values()]
    assert all(abs(r_ / res[0] - 1) < 1e-6 for r_ in res)
    assert abs(out["rel_residual_true"] / out["rel_residual_recursive"] - 1) < 1e-6
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["algorithmic_bytes_per_launch"] - (8.0 * 8192 * 8192 / nproc + 8.0 * (8192 + 8192 / nproc))) < 1
