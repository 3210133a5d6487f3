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
