#!/usr/bin/env python3
"""The HIP path's parity margins against the reference's own outputs, on record: for every converged golden fixture
(tests/golden: matrices in the reference generator's recipe, iteration counts / solution vectors produced by the reference's
CPU drivers at OMP_NUM_THREADS=1) x every topology the library has --

    one shard;  one process with 2 / 3 shards on the three-join event exchange and on the gather-Ap exchange;
    one process per GPU (rank mode) with 2 / 3 ranks on the stream-ordered RCCL double, exchanges 0 / 1 / 2;
    and the opt-in symmetric product on one shard and on the gather-Ap exchange of both multi-shard topologies

-- it prints  iters_hip - iters_ref,  ||x - x_ref|| / ||x_ref||,  the residual recomputed with numpy, and at the end the
largest |iters_hip - iters_ref| (what the iteration gate of the parity tests has to admit; SURVEY 8c proposes max(2, 1 %)).
No oracle is involved: the references are the committed fixtures.   usage: parity_margins.py [--out file]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_parity_margins.txt"))
    a = ap.parse_args()
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    golden = json.load(open(os.path.join(GOLDEN, "golden.json")))
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
                if exchange == 1 and n % shards != 0:
                    continue
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
                if exchange == 1 and n % P != 0:
                    continue
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
