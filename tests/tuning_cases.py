#!/usr/bin/env python3
"""Parity cases of the experiments that live in the TUNING build of the library only (liblam_hip_tuning.so, `make tuning`):
the MFMA-fed bf16 GEMV, separate reduction launches (finalize = 0), per-shard enqueue threads, the hub join, the
whole-iteration persistent launch.  The product library refuses these options (tests/test_gpu_parity.py::
test_tuning_build_variants); the GPU tests run this script as a child process with LAM_HIP_LIB pointing at the tuning
build, one case per call:   tuning_cases.py <case> [args ...]   -> exit code 0 and a last line "ok <case>"."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def mfma(variant, n):
    """Asymmetric random data, so a wrong fragment/diagonal map cannot cancel.  Variant 21 feeds p as three exact bf16 terms
    (fp32-faithful); variant 20 rounds p to bf16 (8 significant bits)."""
    tol = 2.0 ** -8 if variant == 20 else 32 * 2.0 ** -24
    A = np.random.default_rng(n + 3).uniform(-1, 1, (n, n)).astype(np.float32)
    x = np.random.default_rng(n + 4).uniform(-1, 1, n).astype(np.float32)
    with lam.Solver(lam.BF16) as s:
        s.set_matrix(A)
        A_dev = s.download_rows(0, n).astype(np.float64)
        y_valu = s.gemv(x)
        s.set_option("gemv_variant", variant)
        assert "mfma" in s.gemv_kernel_name()
        y = s.gemv(x)
    y64 = A_dev @ x.astype(np.float64)
    scale = np.abs(A_dev) @ np.abs(x.astype(np.float64))
    assert np.max(np.abs(y.astype(np.float64) - y64) / scale) <= tol
    assert np.max(np.abs(y_valu.astype(np.float64) - y64) / scale) <= 32 * 2.0 ** -24


def launch_chain(dtype_name, n, shards):
    """Fused update / two kernels with the reducer workgroup / the round-1 chain with separate reduction launches
    (finalize = 0): the same arithmetic, identical bits -- on both event-ordered exchanges of a multi-shard context."""
    dt = getattr(lam, dtype_name)
    for exchange in ((0, 1) if shards > 1 else (0,)):
        res = []
        for fuse, fin in ((1, 1), (0, 1), (0, 0)):
            with lam.Solver(dt, n_shards=shards, device_ids=[0] * shards) as s:
                s.generate_random_spd(n, 5, 300.0)
                s.generate_random_rhs(6)
                if shards > 1:
                    s.set_option("exchange", exchange)
                s.set_option("fuse_update", fuse)
                s.set_option("finalize", fin)
                s.solve(400, 1e-9 if dtype_name == "F64" else 1e-5)
                res.append((s.stats["num_iters"], s.stats["rel_err"], s.solution().tobytes(), s.true_residual()))
                s.cg_init()
                for _ in range(3):
                    s.cg_iterate(7, 0.0)
                res[-1] += (s.solution().tobytes(),)
        assert res[0] == res[1] == res[2], exchange


def host_enqueue(shards, n):
    """Three ways of ordering the shards' streams for the three-join exchange: all-to-all stream waits (product), the hub,
    one enqueue thread per shard (the reference's OpenMP-thread-per-device shape, MultiGPUS_CUDA.cu:337-378): same bits,
    also when the solve stops early and when it is continued in chunks."""
    res = []
    for threads, hub in ((0, 1), (0, 0), (1, 0), (1, 1)):
        with lam.Solver(lam.F64, device_ids=[0] * shards) as s:
            s.generate_random_spd(n, 7, 200.0)
            s.generate_random_rhs(8)
            s.set_option("exchange", 0)
            s.set_option("host_threads", threads)
            s.set_option("exchange_hub", hub)
            s.solve(500, 1e-9)
            assert s.stats["converged"]
            out = (s.stats["num_iters"], s.stats["rel_err"], s.solution().tobytes(), s.true_residual())
            s.cg_init()
            for chunk in (1, 2, 9, 30):
                s.cg_iterate(chunk, 0.0)
            res.append(out + (s.solution().tobytes(), s.stats["rel_err"]))
    assert res[0] == res[1] == res[2] == res[3]


def persistent(dtype_name, n):
    """Whole iterations inside ONE launch must be the same arithmetic as the two-launch chain: identical iteration counts,
    residuals and solution bits, whatever the number of iterations per launch and however the solve is cut into calls."""
    dt = getattr(lam, dtype_name)
    tol = 1e-9 if dtype_name == "F64" else 1e-5
    res = []
    for pers, chunk in ((0, 32), (1, 32), (1, 1), (1, 5)):
        with lam.Solver(dt) as s:
            s.generate_random_spd(n, 5, 300.0)
            s.generate_random_rhs(6)
            s.set_option("gemv_variant", 10)        # the shape whose tile body the persistent launch shares (fp64 default is 13)
            s.set_option("persistent", pers)
            s.set_option("persist_chunk", chunk)
            s.solve(400, tol)
            assert s.get_option("persistent_effective") == pers
            assert s.stats["converged"]
            out = (s.stats["num_iters"], s.stats["rel_err"], s.solution().tobytes(), s.true_residual())
            s.cg_init()
            for _ in range(3):
                st = s.cg_iterate(7, 0.0)
            assert st["t_gemv"] > 0
            res.append(out + (s.solution().tobytes(), st["rel_err"]))
    assert res[0] == res[1] == res[2] == res[3]


CASES = {"mfma": (mfma, (int, int)), "launch_chain": (launch_chain, (str, int, int)), "host_enqueue": (host_enqueue, (int, int)),
         "persistent": (persistent, (str, int))}


def main():
    with lam.Solver(lam.F64) as s:
        assert s.get_option("tuning_variants") == 1, "not the tuning build: set LAM_HIP_LIB to liblam_hip_tuning.so"
    name = sys.argv[1]
    fn, types = CASES[name]
    fn(*(t(a) for t, a in zip(types, sys.argv[2:])))
    print("ok", name)


if __name__ == "__main__":
    main()
