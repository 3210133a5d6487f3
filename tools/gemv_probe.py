#!/usr/bin/env python3
"""GEMV roofline probe: GB/s of the production GEMV kernel (lam_hip_gemv_only) for a few sizes/options."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [8192, 32768]
    for n in sizes:
        with lam.Solver(lam.F64) as s:
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            for nt in (1, 0):
                s.set_option("nt_loads", nt)
                best = min(s.gemv_only(20) for _ in range(3))
                print(f"N={n} nt={nt} gemv {best*1e3:.4f} ms  {8.0*n*n/best/1e9:.1f} GB/s  ({8.0*n*n/best/8e12*100:.1f}% of 8 TB/s)", flush=True)
            s.cg_init()
            st = s.cg_iterate(100, 0.0)
            print(f"N={n} CG 100 iters: {st['t_iter']*1e3:.4f} ms/iter, gemv {st['t_gemv']*1e3:.4f} ms, rel_err {st['rel_err']:.3e}", flush=True)

if __name__ == "__main__":
    main()
