#!/usr/bin/env python3
"""HBM traffic (PMC) of the BASELINE configs[3] GEMV kernels at N=131072: fp32, bf16 storage (VALU kernel) and the
MFMA-fed bf16 variants -- is any of them re-reading the matrix?  Parent runs itself twice under rocprofv3
(--pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, MI355X_MICROARCH.md).
    usage: config4_traffic.py [--n 131072] [--out gpurun_out/r03_config4_traffic.json]"""
import argparse
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
CASES = (("f32", -1), ("bf16", -1), ("bf16", 21), ("bf16", 20))
REPS = 3


def child(n):
    sys.path.insert(0, ROOT)
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    for dname in ("f32", "bf16"):
        with lam.Solver({"f32": lam.F32, "bf16": lam.BF16}[dname]) as s:
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            for d, v in CASES:
                if d != dname:
                    continue
                s.set_option("gemv_variant", v)
                s.gemv_only(REPS)
                print("case", d, v, s.gemv_kernel_name(), flush=True)


def collect(n, counter, work):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(work, counter)
    r = subprocess.run([prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                        os.path.abspath(__file__), "--child", "--n", str(n)], capture_output=True, text=True, timeout=900, cwd=work,
                       env=dict(os.environ, TMPDIR=work))
    files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        sys.exit(f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-500:]}")
    names = [l.split(None, 3)[3] for l in r.stdout.splitlines() if l.startswith("case ")]
    rows = sorted((x for x in csv.DictReader(open(files[0])) if "gemv_" in x["Kernel_Name"] and x["Counter_Name"] == counter),
                  key=lambda x: int(x["Dispatch_Id"]))
    assert len(rows) == len(CASES) * (1 + REPS), (len(rows), len(CASES))
    res = []
    for i in range(len(CASES)):
        grp = rows[i * (1 + REPS):(i + 1) * (1 + REPS)]
        res.append((sum(float(x["Counter_Value"]) for x in grp[1:]) / REPS, grp[1]["Kernel_Name"]))
    return res, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_config4_traffic.json"))
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.n)
    work = tempfile.mkdtemp(prefix="lam_c4_pmc_")
    try:
        fetch, names = collect(a.n, "FETCH_SIZE", work)
        write, _ = collect(a.n, "WRITE_SIZE", work)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = {}
    for (d, v), (f, kname), (w, _), nm in zip(CASES, fetch, write, names):
        es = 4 if d == "f32" else 2
        alg = es * float(a.n) * a.n + 4.0 * 2 * a.n
        b = (2.0 * f + w) * 1024.0
        out[f"n{a.n}_{d}_v{v}"] = {"kernel": nm, "algorithmic_bytes": alg, "hbm_bytes_per_launch": b, "over_algorithmic": b / alg,
                                   "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}
        print(f"{d} variant {v} [{nm}]: algorithmic {alg / 1e9:.4f} GB, HBM {b / 1e9:.4f} GB (x{b / alg:.4f})")
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
