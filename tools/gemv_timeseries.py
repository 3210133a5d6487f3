#!/usr/bin/env python3
"""GEMV rate over time in ONE process: does the number bench.py reports for configs[3] (a median of three gemv_only(10)
calls 0.3 s after the first touch) sit on the device's sustained level?  Prints the rate of consecutive gemv_only(reps)
calls for `seconds`, then the same after an idle pause.
    usage: gemv_timeseries.py [--dtype bf16] [--n 131072] [--seconds 6] [--reps 10] [--variant -1]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--n", type=int, default=131072)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--pause", type=float, default=2.0)
    a = ap.parse_args()
    dt = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    es = {"f64": 8, "f32": 4, "bf16": 2}[a.dtype]
    with lam.Solver(dt) as s:
        s.generate_random_spd(a.n, 1234, 1e4)
        s.generate_random_rhs(1235)
        s.cg_init()
        s.set_option("gemv_variant", a.variant)
        gb = (es * float(a.n) * a.n + (8.0 if a.dtype == "f64" else 4.0) * 2 * a.n) / 1e9
        print(f"# {a.dtype} N={a.n} {s.gemv_kernel_name()}: GB/s of consecutive gemv_only({a.reps}) calls; t = seconds since the first launch")
        for phase in ("first touch", f"after {a.pause} s idle"):
            t0 = time.perf_counter()
            rows = []
            while time.perf_counter() - t0 < a.seconds:
                t = time.perf_counter() - t0
                rows.append((t, gb / s.gemv_only(a.reps)))
            print(f"# {phase}: {len(rows)} samples")
            step = max(1, len(rows) // 40)
            for i in range(0, len(rows), step):
                print(f"t={rows[i][0]:6.3f}s {rows[i][1]:8.1f} GB/s ({rows[i][1] / 80:5.2f}% of 8 TB/s)")
            tail = sorted(r for _, r in rows[len(rows) // 2:])
            print(f"# median of the second half: {tail[len(tail) // 2]:.1f} GB/s; first three calls: " + ", ".join(f"{r:.1f}" for _, r in rows[:3]), flush=True)
            time.sleep(a.pause)


if __name__ == "__main__":
    main()
