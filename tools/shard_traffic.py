#!/usr/bin/env python3
"""HBM traffic of ONE rank's GEMV on the shard shapes of a P-way split of N=65536 (rows = N/P, all N columns), from
the PMC counters -- what `roofline.traffic` of a multi-GPU bench line refers to (profiles/traffic.json, keys
n65536_p2 / _p4 / _p8), collected on one GPU because a launcher run cannot profile itself.

The parent (no GPU use) runs this script twice under rocprofv3 as child processes, `--pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` in SEPARATE passes (MI355X_MICROARCH.md, HBM: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; on gfx950
FETCH_SIZE counts a 128-B request as 64 B).  The child launches, for every P, the rank mode's GEMV of a shard:
unsplit, and in the split form the default exchange uses (own-slice column panel, then the remaining columns
accumulated on top -- two launches per GEMV, their counters are added).

    usage: shard_traffic.py [--n 65536] [--out gpurun_out/r03_shard_traffic.json]
"""
import argparse
import collections
import csv
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = 3


def child(n, shards):
    sys.path.insert(0, ROOT)
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    with lam.Solver(lam.F64) as s:
        s.generate_random_spd(n, 1234, 1e6)
        s.generate_random_rhs(1235)
        s.cg_init()
        for P in shards:
            rows = n // P
            s.set_option("probe_rows", rows)
            for lo, hi in ((0, 0), (0, rows)):          # unsplit; split at rank 0's own slice
                s.set_option("panel_lo", lo)
                s.set_option("panel_hi", hi)
                s.gemv_only(REPS)                        # 1 warm-up + REPS GEMVs
        print("kernel", s.gemv_kernel_name())


def collect(n, shards, counter, work):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(work, counter)
    cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
           "--child", "--n", str(n), "--shards", ",".join(map(str, shards))]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=work, env=dict(os.environ, TMPDIR=work))
    files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        sys.exit(f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-500:]}")
    # launches in dispatch order: per P -> (1 + REPS) unsplit GEMVs, then (1 + REPS) split GEMVs of two launches each
    rows = sorted((x for x in csv.DictReader(open(files[0])) if "gemv_" in x["Kernel_Name"] and x["Counter_Name"] == counter),
                  key=lambda x: int(x["Dispatch_Id"]))
    res, i, name = {}, 0, None
    for P in shards:
        grids = {(n // P) // 2 * 256, (n // P) // 2 * 512}   # production shapes: 2 rows per 4-wave (round 1-3) / 8-wave (round 4) workgroup
        uns = rows[i:i + 1 + REPS]; i += 1 + REPS
        spl = rows[i:i + 2 * (1 + REPS)]; i += 2 * (1 + REPS)
        assert all(int(x["Grid_Size"]) in grids for x in uns + spl), (P, grids, [x["Grid_Size"] for x in uns + spl])
        name = uns[0]["Kernel_Name"]
        res[P] = {"unsplit": sum(float(x["Counter_Value"]) for x in uns[1:]) / REPS,
                  "split": sum(float(x["Counter_Value"]) for x in spl[2:]) / REPS}
    assert i == len(rows), (i, len(rows))
    return res, name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--shards", default="2,4,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_shard_traffic.json"))
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    shards = [int(x) for x in a.shards.split(",")]
    if a.child:
        return child(a.n, shards)
    work = tempfile.mkdtemp(prefix="lam_shard_pmc_")
    try:
        fetch, name = collect(a.n, shards, "FETCH_SIZE", work)
        write, _ = collect(a.n, shards, "WRITE_SIZE", work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:   # noqa: BLE001
        commit = None
    out = collections.OrderedDict()
    for P in shards:
        rows = a.n // P
        alg = 8.0 * rows * a.n + 8.0 * (a.n + rows)
        e = {"shape": f"{rows} x {a.n} fp64 (one rank of a {P}-way split)", "algorithmic_bytes": alg, "kernel": name, "commit": commit,
             "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
             "source": "tools/shard_traffic.py (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one GPU, "
                       "lam_hip_gemv_only with probe_rows)"}
        for form in ("unsplit", "split"):
            b = (2.0 * fetch[P][form] + write[P][form]) * 1024.0
            e[form] = {"hbm_bytes_per_gemv": b, "FETCH_SIZE_KiB": fetch[P][form], "WRITE_SIZE_KiB": write[P][form], "over_algorithmic": b / alg}
        # the default exchange (overlap 1) runs the split form
        e["hbm_bytes_per_launch"] = e["split"]["hbm_bytes_per_gemv"]
        e["note"] = "hbm_bytes_per_launch = the split form (own-slice panel + the rest: the two launches of one GEMV added)"
        out[f"n{a.n}_p{P}"] = e
        print(f"P={P}: algorithmic {alg / 1e9:.4f} GB, unsplit {e['unsplit']['hbm_bytes_per_gemv'] / 1e9:.4f} GB "
              f"(x{e['unsplit']['over_algorithmic']:.4f}), split {e['split']['hbm_bytes_per_gemv'] / 1e9:.4f} GB (x{e['split']['over_algorithmic']:.4f})")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
