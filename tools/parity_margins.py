#!/usr/bin/env python3
"""The HIP path's parity margins against the reference's own outputs, on record: for every converged golden fixture
(tests/golden: matrices in the reference generator's recipe, iteration counts / solution vectors produced by the reference's
CPU drivers at OMP_NUM_THREADS=1) x every topology the library has --

    one shard;  one process with 2 / 3 shards on the three-join event exchange and on the gather-Ap exchange;
    one process per GPU (rank mode) with 2 / 3 ranks on the stream-ordered RCCL double, exchanges 0 / 1 / 2;
    and the opt-in symmetric product on one shard and on the gather-Ap exchange of both multi-shard topologies

-- it prints  iters_hip - iters_ref,  ||x - x_ref|| / ||x_ref||,  the residual recomputed with numpy, and at the end the
largest |iters_hip - iters_ref| (what the iteration gate of the parity tests has to admit; SURVEY 8c proposes max(2, 1 %)).
--precision f32: the same for the fp32 path against the `file_mode_f32` fixtures (f32_table below).  bf16 storage has no
reference-produced fixture (the reference has no bf16): its margins against the fp64 oracle on the bf16-rounded matrix are
measured by tests/margins_bf16.py (a test-side script: only tests may call the oracle) and appended to the same profile.
No oracle is involved here: the references are the committed fixtures.   usage: parity_margins.py [--precision f64|f32] [--out file]"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl_async.so")
RUN_RANKS = os.path.join(ROOT, "tests", "mock_rccl", "run_ranks.py")
sys.path.insert(0, ROOT)


def read_bin(path):
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.uint64)
        rows, cols = int(hdr[0]), int(hdr[1]) & 0xFFFFFFFF
        return np.frombuffer(f.read(rows * cols * 8), dtype=np.float64).reshape(rows, cols).copy()


def read_bin32(path):
    with open(path, "rb") as f:
        hdr = np.frombuffer(f.read(16), dtype=np.uint64)
        rows, cols = int(hdr[0]), int(hdr[1]) & 0xFFFFFFFF
        return np.frombuffer(f.read(rows * cols * 4), dtype=np.float32).reshape(rows, cols).copy()


def f32_table(lam, golden, out_path):
    """--precision f32: the four `file_mode_f32` fixtures (outputs of the reference's own ConjugateGradient_CPU_OMP<float>,
    oracle/ref_float_harness.cpp) x {one shard, 2 shards gather-Ap, 3 shards gather-Ap (uneven), 2 ranks on the RCCL double
    (exchanges 0 / 1), the symmetric product on one shard and on 2 shards}: iteration difference, |x - x_ref| / |x_ref|, the
    residual recomputed in fp64 with numpy, and the printed recursive residual against the reference's.  What the gates of
    tests/test_gpu_parity.py::test_cg_file_mode_golden_f32 are derived from."""
    rows, stats = [], []
    have_mock = os.path.exists(MOCK)

    def record(g, topo, iters, rel_err, x, A, b, x_ref):
        # (a run that stops at the iteration cap: the reference's float harness prints max_iters, lam_hip_stats the loop counter on
        # exit, max_iters + 1 -- the same iteration)
        d = iters - g["iters_printed"] - (0 if g["converged"] else 1)
        xe = float(np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref))
        res = float(np.linalg.norm(b - A @ x) / np.linalg.norm(b))
        rr = abs(rel_err / g["rel_err_printed"] - 1)
        rows.append(f"{g['tag']:30s} ref {g['iters_printed']:4d}  {topo:44s} iters {iters:4d} ({d:+d})  |x-x_ref|/|x_ref| {xe:9.2e}  "
                    f"residual(fp64) {res:9.2e} (tol {g['tol']:.0e})  rel_err {rel_err:.6e} (ref {g['rel_err_printed']:.6e}, off by {rr:.2e})")
        stats.append(dict(tag=g["tag"], converged=g["converged"], max_iters=g["max_iters"], d=abs(d), rel_d=abs(d) / max(1, g["iters_printed"]),
                          xe=xe, res_over_tol=res / g["tol"], rr=rr))

    for g in golden["file_mode_f32"]:
        mpath, bpath = os.path.join(GOLDEN, g["name"] + ".f32.matrix.bin"), os.path.join(GOLDEN, g["name"] + ".f32.rhs.bin")
        A, b = read_bin32(mpath).astype(np.float64), read_bin32(bpath).reshape(-1).astype(np.float64)
        x_ref = read_bin32(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1).astype(np.float64)
        for shards, exchange, sym in ((1, None, 0), (1, None, 2), (2, 0, 0), (2, 1, 0), (3, 1, 0), (2, 1, 2)):
            with lam.Solver(lam.F32, device_ids=[0] * shards) as s:
                assert s.load_matrix_from_file(mpath) and s.load_rhs_from_file(bpath)
                if exchange is not None:
                    s.set_option("exchange", exchange)
                s.set_option("symmetric", sym)
                assert s.get_option("symmetric_effective") == (1 if sym else 0)
                s.solve(g["max_iters"], g["tol"])
                topo = "one shard" if shards == 1 else f"one process, {shards} shards, {'events x3' if exchange == 0 else 'gather-Ap'}"
                record(g, topo + (" + symmetric" if sym else ""), s.stats["num_iters"], s.stats["rel_err"], s.solution().astype(np.float64), A, b, x_ref)
        if not have_mock:
            continue
        for P, exchange, sym in ((2, 0, 0), (2, 1, 0), (2, 1, 2)):
            with tempfile.TemporaryDirectory() as tmp:
                xf = os.path.join(tmp, "x.npy")
                env = dict(os.environ, LD_PRELOAD=MOCK, GPU_MAX_HW_QUEUES=str(2 * P + 4), MOCK_RCCL_TIMEOUT_MS="20000")
                r = subprocess.run([sys.executable, RUN_RANKS, str(P), str(g["n"]), "file", "--matrix", mpath, "--rhs", bpath, "--dtype", "f32", "--exchange", str(exchange),
                                    "--symmetric", str(sym), "--iters", str(g["max_iters"]), "--tol", repr(g["tol"]), "--no-single", "--save-x", xf],
                                   env=env, capture_output=True, text=True, timeout=300)
                if r.returncode != 0:
                    rows.append(f"{g['tag']:30s} rank mode P={P} exchange {exchange}: FAILED {r.stdout[-300:]} {r.stderr[-300:]}")
                    stats.append(dict(tag=g["tag"], converged=g["converged"], max_iters=g["max_iters"], d=99, rel_d=99, xe=99, res_over_tol=99, rr=99))
                    continue
                out = json.loads(r.stdout.strip().splitlines()[-1])
                assert out["ranks_identical"] and out["exchange_effective"] == [exchange] * P, out
                record(g, f"rank mode (RCCL double), {P} ranks, exchange {exchange}" + (" + symmetric" if sym else ""), out["iters"], out["rel_err"],
                       np.load(xf).astype(np.float64), A, b, x_ref)
    conv = [s_ for s_ in stats if s_["converged"]]
    short = [s_ for s_ in stats if not s_["converged"] and s_["max_iters"] <= 5]
    mid = [s_ for s_ in stats if not s_["converged"] and s_["max_iters"] > 5]
    summary = {"precision": "f32", "runs": len(stats),
               "converged": {"max_abs_delta_iters": max(s_["d"] for s_ in conv), "max_rel_delta_iters": max(s_["rel_d"] for s_ in conv),
                             "max_x_err": max(s_["xe"] for s_ in conv), "max_residual_over_tol": max(s_["res_over_tol"] for s_ in conv)},
               "fixed_5_iterations": {"max_x_err": max(s_["xe"] for s_ in short), "max_rel_err_off": max(s_["rr"] for s_ in short)},
               "fixed_40_iterations": {"max_x_err": max(s_["xe"] for s_ in mid), "max_rel_err_off": max(s_["rr"] for s_ in mid)}}
    lines = ["# tools/parity_margins.py --precision f32 -- the fp32 HIP path against outputs of the reference's own class instantiated with float",
             "# (tests/golden file_mode_f32, from oracle/ref_float_harness.cpp -> ConjugateGradient_CPU_OMP<float>), every topology",
             "# columns: fixture, reference iterations, topology, HIP iterations (difference), solution error, residual recomputed in fp64, printed residual"]
    lines += rows
    lines.append("# summary: " + json.dumps(summary))
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    open(out_path, "w").write(text)
    sys.stdout.write(text)
    print(json.dumps(summary))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, "gpurun_out", "r05_parity_margins.txt" if a.precision == "f64" else "r05_parity_margins_f32.txt")
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    golden = json.load(open(os.path.join(GOLDEN, "golden.json")))
    if a.precision == "f32":
        return f32_table(lam, golden, a.out)
    rows, worst = [], {}

    def record(g, topo, iters, x, A, b, x_ref):
        d = iters - g["iters_printed"]
        xe = float(np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref))
        res = float(np.linalg.norm(b - A @ x) / np.linalg.norm(b))
        rows.append(f"{g['tag']:28s} ref {g['iters_printed']:4d}  {topo:46s} iters {iters:4d} ({d:+d})  |x-x_ref|/|x_ref| {xe:9.2e}  "
                    f"residual {res:9.2e} (tol {g['tol']:.0e})")
        key = topo.split(",")[0] + (" + symmetric" if topo.endswith("symmetric") and "," in topo else "")
        worst[key] = max(worst.get(key, 0), abs(d))
        return abs(d), xe / g["tol"], res / g["tol"]

    stats = []
    have_mock = os.path.exists(MOCK)
    for g in golden["file_mode"]:
        if not g["converged"]:
            continue
        mpath, bpath = os.path.join(GOLDEN, g["name"] + ".matrix.bin"), os.path.join(GOLDEN, g["name"] + ".rhs.bin")
        A, b = read_bin(mpath), read_bin(bpath).reshape(-1)
        x_ref = read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
        n = g["n"]
        for shards in (1, 2, 3):
            # (exchange, symmetric): the opt-in symmetric product reads every pair {A_ij, A_ji} once -- the fixtures are the reference
            # generator's Q D Q^T, symmetric to rounding --, on one shard and on the gather-Ap exchange
            for exchange, sym in (((None, 0), (None, 2)) if shards == 1 else ((0, 0), (1, 0), (1, 2))):
                with lam.Solver(lam.F64, device_ids=[0] * shards) as s:
                    assert s.load_matrix_from_file(mpath) and s.load_rhs_from_file(bpath)
                    if exchange is not None:
                        s.set_option("exchange", exchange)
                    s.set_option("symmetric", sym)
                    assert s.get_option("symmetric_effective") == (1 if sym else 0)
                    s.solve(g["max_iters"], g["tol"])
                    topo = "one shard" if shards == 1 else f"one process, {shards} shards, {'events x3' if exchange == 0 else 'gather-Ap'}"
                    if sym:
                        topo += " + symmetric"
                    stats.append(record(g, topo, s.stats["num_iters"], s.solution(), A, b, x_ref))
        if not have_mock:
            continue
        for P in (2, 3):
            for exchange, sym in ((0, 0), (1, 0), (1, 2), (2, 0)):
                with tempfile.TemporaryDirectory() as tmp:
                    xf = os.path.join(tmp, "x.npy")
                    env = dict(os.environ, LD_PRELOAD=MOCK, GPU_MAX_HW_QUEUES=str(2 * P + 4), MOCK_RCCL_TIMEOUT_MS="20000")
                    r = subprocess.run([sys.executable, RUN_RANKS, str(P), str(n), "file", "--matrix", mpath, "--rhs", bpath, "--exchange", str(exchange), "--symmetric", str(sym),
                                        "--iters", str(g["max_iters"]), "--tol", repr(g["tol"]), "--no-single", "--save-x", xf],
                                       env=env, capture_output=True, text=True, timeout=300)
                    if r.returncode != 0:
                        rows.append(f"{g['tag']:28s} rank mode P={P} exchange {exchange}: FAILED {r.stdout[-300:]} {r.stderr[-300:]}")
                        stats.append((99, 0, 0))
                        continue
                    out = json.loads(r.stdout.strip().splitlines()[-1])
                    assert out["ranks_identical"] and out["exchange_effective"] == [exchange] * P, out
                    stats.append(record(g, f"rank mode (RCCL double), {P} ranks, exchange {exchange}" + (" + symmetric" if sym else ""), out["iters"], np.load(xf), A, b, x_ref))
    lines = ["# tools/parity_margins.py -- HIP path against the reference's own fixtures (tests/golden), every topology",
             "# columns: fixture, reference iterations, topology, HIP iterations (difference), solution error, residual recomputed with numpy"]
    lines += rows
    dmax = max(s_[0] for s_ in stats)
    lines.append(f"# {len(stats)} runs: max |iters_hip - iters_ref| = {dmax};  max |x-x_ref|/|x_ref| / tol = {max(s_[1] for s_ in stats):.3f};  "
                 f"max residual / tol = {max(s_[2] for s_ in stats):.3f}")
    lines.append("# per topology, max |iters_hip - iters_ref|: " + "; ".join(f"{k}: {v}" for k, v in sorted(worst.items())))
    text = "\n".join(lines) + "\n"
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write(text)
    sys.stdout.write(text)
    print(json.dumps({"runs": len(stats), "max_abs_delta_iters": dmax}))


if __name__ == "__main__":
    main()
