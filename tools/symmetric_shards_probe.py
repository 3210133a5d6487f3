#!/usr/bin/env python3
"""Option "symmetric" with several row shards (cyclic half windows, lam_kernels.h symv_use), emulated: P shards of ONE process on
device 0, so an iteration is the P shards' work back to back (plus the join) -- per-shard time ~ t / P.  Next to the general
GEMV on the same shards and to the single-shard numbers.  A multi-GPU measurement it is not.

usage: symmetric_shards_probe.py [N ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
for n in [int(a) for a in sys.argv[1:]] or [65536, 32768]:
    for P in (1, 2, 4, 8):
        with lam.Solver(lam.F64, device_ids=[0] * P) as s:
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            res = {}
            for rnd in range(3):
                for sym in (0, 1):
                    s.set_option("symmetric", 2 * sym)
                    assert s.get_option("symmetric_effective") == sym
                    s.cg_init(); s.cg_iterate(5, 0.0)
                    l0 = s.get_option("hip_calls_launch")
                    st = s.cg_iterate(60, 0.0)
                    launches = (s.get_option("hip_calls_launch") - l0) / 60
                    if rnd:
                        res.setdefault(sym, []).append((st["t_iter"], st["t_gemv"], st["rel_err"], launches))
            g, y = sorted(res[0])[0], sorted(res[1])[0]
            tri = 8.0 * n * (n + 1) / 2
            print(f"N={n} P={P} (one device): general {g[0]*1e3:8.4f} ms/iter (product on shard 0 {g[1]*1e3:7.4f} ms, {g[3]:.0f} launches) | symmetric "
                  f"{y[0]*1e3:8.4f} ms/iter (product on shard 0 {y[1]*1e3:7.4f} ms, {y[3]:.0f} launches) = {g[0]/y[0]:.3f}x; "
                  f"triangle bytes / iteration time = {tri/y[0]/1e9:7.1f} GB/s; residuals {g[2]:.6e} / {y[2]:.6e}", flush=True)
