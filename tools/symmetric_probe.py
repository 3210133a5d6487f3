#!/usr/bin/env python3
"""Option "symmetric" (upper-triangle product, lam_kernels.h) next to the general GEMV, ONE context per size, modes interleaved:
the product alone (lam_hip_gemv_only: launches back to back) and inside CG (cg_iterate).  Bytes: the general product streams
s*N^2, the symmetric one s*N(N+1)/2; both rates are quoted on their own bytes and against the 8 TB/s HBM peak.

usage: symmetric_probe.py [f64|f32|bf16] [N ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
args = sys.argv[1:]
dt_name = args.pop(0) if args and args[0] in ("f64", "f32", "bf16") else "f64"
dt, esz = {"f64": (lam.F64, 8), "f32": (lam.F32, 4), "bf16": (lam.BF16, 2)}[dt_name]
for n in [int(a) for a in args] or [65536, 40000, 32768, 20000, 10000, 4096]:
    with lam.Solver(dt) as s:
        s.generate_random_spd(n, 1234, 1e6)
        s.generate_random_rhs(1235)
        reps = max(10, min(2000, int(2e10 / (esz * n * n))))
        res = {0: {"gemv": [], "iter": []}, 1: {"gemv": [], "iter": []}}
        for rnd in range(4):
            for sym in (0, 1):
                s.set_option("symmetric", 2 * sym)      # 2: at every size (1 leaves matrices below 192 MiB on the general GEMV)
                assert s.get_option("symmetric_effective") == sym
                s.gemv_only(3)
                t = s.gemv_only(reps)
                s.cg_init(); s.cg_iterate(5, 0.0)
                st = s.cg_iterate(reps, 0.0)
                if rnd:                      # first round: warm-up (task list, partial buffers, clocks)
                    res[sym]["gemv"].append(t)       # seconds per product (average of `reps` launches)
                    res[sym]["iter"].append(st["t_iter"])
        kern = {}
        for sym in (0, 1):
            s.set_option("symmetric", 2 * sym)
            kern[sym] = s.gemv_kernel_name()
        med = lambda v: sorted(v)[len(v) // 2]
        g0, g1, i0, i1 = med(res[0]["gemv"]), med(res[1]["gemv"]), med(res[0]["iter"]), med(res[1]["iter"])
        b0, b1 = esz * n * n, esz * n * (n + 1) / 2
        print(f"N={n} {dt_name}: general {g0*1e3:8.4f} ms = {b0/g0/1e9:7.1f} GB/s ({b0/g0/8e12*100:4.1f} % of peak) | symmetric {g1*1e3:8.4f} ms = "
              f"{b1/g1/1e9:7.1f} GB/s on the triangle ({b1/g1/8e12*100:4.1f} %) = {g0/g1:.3f}x | CG iteration {i0*1e3:8.4f} -> {i1*1e3:8.4f} ms "
              f"({1/i0:8.1f} -> {1/i1:8.1f} it/s, {i0/i1:.3f}x)   [{kern[1]}]", flush=True)
