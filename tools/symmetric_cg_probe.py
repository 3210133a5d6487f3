#!/usr/bin/env python3
"""CG iteration time with option "symmetric" (upper-triangle product) vs the general GEMV, same context."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
for n in [int(a) for a in sys.argv[1:]] or [32768, 65536]:
    out = {}
    for sym in (0, 1):                      # a fresh context per mode (allocation history affects the rate)
        with lam.Solver(lam.F64) as s:
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            s.set_option("symmetric", sym)
            ts = []
            for _ in range(3):
                s.cg_init(); s.cg_iterate(5, 0.0)
                st = s.cg_iterate(100, 0.0)
                ts.append((st["t_iter"], st["t_gemv"], st["rel_err"]))
            ts.sort()
            out[sym] = ts[1]
            print(f"N={n} symmetric={sym}: {ts[1][0]*1e3:.4f} ms/iter ({1/ts[1][0]:.1f} it/s), product {ts[1][1]*1e3:.4f} ms, "
                  f"true residual {s.true_residual():.6e} vs recursive {ts[1][2]:.6e}", flush=True)
    print(f"N={n}: speed-up {out[0][0]/out[1][0]:.2f}x", flush=True)
