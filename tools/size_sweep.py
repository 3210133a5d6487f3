#!/usr/bin/env python3
"""GEMV rate of the production kernel over the matrix order N (one process, one grow-only context, largest first so nothing
is freed between sizes): is the fraction of peak a smooth function of the row length, or does it follow the alignment of
the row pitch?   usage: size_sweep.py [--dtype f64] [N ...]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int)
    ap.add_argument("--dtype", default="f64")
    a = ap.parse_args()
    sizes = a.sizes or [65536, 61440, 57344, 53248, 49152, 45056, 40960, 40000, 36864, 32768, 30000, 28672, 24576, 20480, 20000, 16384, 12288, 10000, 8192]
    dt = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    es = {"f64": 8, "f32": 4, "bf16": 2}[a.dtype]
    with lam.Solver(dt) as s:
        for n in sorted(sizes, reverse=True):
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            reps = max(10, min(400, int(0.15 / (es * n * n / 7e12))))
            ts = sorted(s.gemv_only(reps) for _ in range(5))
            gb = es * float(n) * n / 1e9
            t = ts[2]
            print(f"N={n:6d} pitch {n * es:7d} B (mod 4096 = {n * es % 4096:4d}) tiles/row {n / 4096:6.2f}: {t * 1e6:9.1f} us  {gb / t:7.1f} GB/s "
                  f"({gb / t / 80:5.2f} % of 8 TB/s); minus a 9.5 us launch: {gb / (t - 9.5e-6) / 80:5.2f} %", flush=True)


if __name__ == "__main__":
    main()
