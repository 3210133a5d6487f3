#!/usr/bin/env python3
"""PMC counters of the GEMV kernels, case by case: what separates the bf16 production kernel (0.84 of peak) from the
fp32 / fp64 ones (0.90), and the short-row sizes (N=10000: 0.80) from the headline?  The parent runs itself under
rocprofv3 once per counter group (--kernel-trace --pmc only: separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots")
and prints one row per case; durations come from the kernel trace of the same passes.

    usage: gemv_counters.py [--cases f64:65536,f64:10000,f32:131072,bf16:131072[:variant]] [--out file.json]
Load another build with LAM_HIP_LIB (the tuning build for non-production variants)."""
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
REPS = 5
# counter groups that fit one pass each (SQ: 8 slots, TCC: 4 with FETCH_SIZE = 3 / WRITE_SIZE = 2, GRBM: 2)
GROUPS = (
    ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"),
    ("SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"),
    ("GRBM_GUI_ACTIVE", "FETCH_SIZE"),
    ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum"),
)
DT = {"f64": 0, "f32": 1, "bf16": 2}
ES = {"f64": 8, "f32": 4, "bf16": 2}


def parse_cases(text):
    cases = []
    for tok in text.split(","):
        f = tok.split(":")
        # a third field "sym": option symmetric = 1 (the upper-triangle product; its first pass, symv_task_kernel, is the row)
        cases.append((f[0], int(f[1]), (-2 if f[2] == "sym" else int(f[2])) if len(f) > 2 else -1))
    return cases


def child(cases):
    sys.path.insert(0, ROOT)
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    for d, n, v in cases:
        with lam.Solver(DT[d]) as s:
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            if v == -2:
                s.set_option("symmetric", 2)
            else:
                s.set_option("gemv_variant", v)
            s.gemv_only(REPS)               # 1 warm-up + REPS launches
            print("case", d, n, v, s.gemv_kernel_name(), flush=True)


def collect(cases_text, counters, work, tag):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(work, tag)
    r = subprocess.run([prof, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--", sys.executable,
                        os.path.abspath(__file__), "--child", "--cases", cases_text], capture_output=True, text=True, timeout=1200, cwd=work,
                       env=dict(os.environ, TMPDIR=work))
    cfiles = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
    kfiles = glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True)
    if r.returncode != 0 or not cfiles:
        return None, f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-400:]}"
    dur = {}
    if kfiles:
        for x in csv.DictReader(open(kfiles[0])):
            dur[int(x["Dispatch_Id"])] = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) * 1e-3
    per = {}
    for x in csv.DictReader(open(cfiles[0])):
        if "gemv_" not in x["Kernel_Name"] and "symv_task" not in x["Kernel_Name"]:
            continue
        d = per.setdefault(int(x["Dispatch_Id"]), {"kernel": x["Kernel_Name"], "vgpr": x.get("VGPR_Count") or x.get("Arch_VGPR_Count"),
                                                      "lds": x.get("LDS_Block_Size"), "grid": x.get("Grid_Size")})
        d[x["Counter_Name"]] = d.get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
    ids = sorted(per)
    return (ids, per, dur), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="f64:65536,f64:10000,f32:131072,bf16:131072")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "gemv_counters.json"))
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    cases = parse_cases(a.cases)
    if a.child:
        return child(cases)
    work = tempfile.mkdtemp(prefix="lam_gc_")
    table = [{"dtype": d, "n": n, "variant": v, "algorithmic_bytes": (ES[d] * float(n) * (n + 1) / 2 if v == -2 else ES[d] * float(n) * n) +
              (8.0 if d == "f64" else 4.0) * 2 * n} for d, n, v in cases]
    try:
        for gi, group in enumerate(GROUPS):
            res, err = collect(a.cases, group, work, f"g{gi}")
            if res is None:
                # a counter name this ROCm does not know fails the whole pass: try its counters one by one
                print(f"# group {gi}: {err}", flush=True)
                for c in group:
                    res1, err1 = collect(a.cases, (c,), work, f"g{gi}_{c}")
                    if res1 is None:
                        print(f"# counter {c}: not collected ({err1[-160:]})", flush=True)
                        continue
                    merge(table, res1, (c,))
                continue
            merge(table, res, group)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for row in table:
        t = row.get("duration_us")
        if t:
            row["gbps"] = row["algorithmic_bytes"] / t / 1e3
            row["frac_of_8TBps"] = row["gbps"] / 8000.0
        wc, wv = row.get("SQ_WAVE_CYCLES"), row.get("SQ_WAVES")
        if wc and wv:
            row["wave_cycles_per_wave"] = wc / wv
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if row.get(k) is not None:
                    row[k + "/WAVE_CYCLES"] = row[k] / wc
        if row.get("SQ_INSTS_VALU") and wv:
            row["valu_insts_per_wave"] = row["SQ_INSTS_VALU"] / wv
        if row.get("GRBM_GUI_ACTIVE") and t:
            row["effective_clock_GHz"] = row["GRBM_GUI_ACTIVE"] / 8.0 / t / 1e3
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(table, open(a.out, "w"), indent=1)


def merge(table, res, counters):
    ids, per, dur = res
    assert len(ids) == len(table) * (1 + REPS), (len(ids), len(table))
    for i, row in enumerate(table):
        grp = ids[i * (1 + REPS) + 1:(i + 1) * (1 + REPS)]            # drop the warm-up launch
        row.setdefault("kernel", per[grp[0]]["kernel"])
        for k in ("vgpr", "lds", "grid"):
            row.setdefault(k, per[grp[0]].get(k))
        for c in counters:
            vals = [per[j][c] for j in grp if c in per[j]]
            if vals:
                row[c] = sum(vals) / len(vals)
        ds = [dur[j] for j in grp if j in dur]
        if ds:
            row["duration_us"] = sum(ds) / len(ds)      # under the LAST pass's counters (PMC passes perturb little)


if __name__ == "__main__":
    main()
