#!/usr/bin/env python3
"""Where BENCH_r03's bf16 figure (0.842 of peak, against 0.883 for the same kernel in a fresh process) comes from: the
configs[3] side run of bench.py measured bf16 right after CLOSING the fp32 context (a 68.7 GB hipFree).  This probe
measures the bf16 GEMV at N=131072 (a) in a fresh context, (b) right after a 68.7 GB context was freed, (c) with that
context still alive -- each as a short time series, in ONE process.
    usage: bf16_gap_probe.py [--n 131072]"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def series(s, gb, seconds=1.5):
    t0, out = time.perf_counter(), []
    while time.perf_counter() - t0 < seconds:
        out.append(gb / s.gemv_only(10))
    return out


def make(dt, n):
    s = lam.Solver(dt)
    s.generate_random_spd(n, 1234, 1e4)
    s.generate_random_rhs(1235)
    s.cg_init()
    return s


def show(label, r):
    r2 = sorted(r)
    print(f"{label:58s} first {r[0]:7.1f}  median {r2[len(r2) // 2]:7.1f}  min {r2[0]:7.1f}  max {r2[-1]:7.1f} GB/s  ({r2[len(r2) // 2] / 80:.2f} % of 8 TB/s, {len(r)} calls of 10 launches)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=131072)
    n = ap.parse_args().n
    gb16, gb32 = (2.0 * n * n + 8.0 * n) / 1e9, (4.0 * n * n + 8.0 * n) / 1e9
    b = make(lam.BF16, n)
    show("bf16, fresh process", series(b, gb16))
    b.close()
    f = make(lam.F32, n)
    show("fp32, after a 34 GB free", series(f, gb32))
    f.close()                                               # the 68.7 GB hipFree bench.py (round 3) did here
    b = make(lam.BF16, n)
    show("bf16, right after closing the fp32 context (68.7 GB free)", series(b, gb16))
    time.sleep(2.0)
    show("bf16, same context 2 s later", series(b, gb16))
    f = make(lam.F32, n)                                    # both alive, nothing freed
    show("fp32, bf16 context still alive (no free)", series(f, gb32))
    show("bf16, fp32 context still alive (no free)", series(b, gb16))
    f.close()
    b.close()


if __name__ == "__main__":
    main()
