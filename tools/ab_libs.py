#!/usr/bin/env python3
"""A/B of two builds of the library on ONE box: alternating child processes, each timing the GEMV alone (lam_hip_gemv_only) and a
short CG run.     usage: ab_libs.py libA.so libB.so [libC.so ...] [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys
sys.path.insert(0, %r)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
out = {}
for name, dt, n in [x for x in (("f64", lam.F64, 65536), ("f32", lam.F32, 131072), ("bf16", lam.BF16, 131072), ("f64_40000", lam.F64, 40000)) if not os.environ.get("AB_ONLY") or x[0] in os.environ["AB_ONLY"].split(",")]:
    with lam.Solver(dt) as s:
        s.generate_random_spd(n, 5, 1e3)
        s.generate_random_rhs(6)
        s.gemv_only(150)
        ms = [s.gemv_only(100) for _ in range(3)]
        s.set_option("gemv_timing", 1)
        s.solve(100, 1e-30)
        st = s.stats
        out[name] = dict(gemv_only_ms=round(1e3 * min(ms), 4), cg_gemv_ms=round(1e3 * st["t_gemv"], 4))
print(json.dumps(out))
''' % ROOT


def main():
    libs = [a for a in sys.argv[1:] if not a.isdigit()]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, LAM_HIP_LIB=os.path.abspath(lib), LAM_HIP_ALLOW_STALE="1")
            p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
            print(r, os.path.basename(lib), line, flush=True)


if __name__ == "__main__":
    main()
