#!/usr/bin/env python3
"""the GEMV inside a CG iteration at small N (N = 10000: 0.82 of peak against 0.89 at N = 65536) -- is there a
launch SHAPE chosen by size that reaches 0.85?  Candidates, interleaved in one context per N (tuning build): the production shape
(cooperative rows, 2 rows per 8-wave workgroup), the same with 4 waves, ONE row per workgroup (twice as many, half as long),
3 rows, a 2048-column tile, two row pairs per workgroup sharing every staged tile, 4 rows per wave, and the COLUMN-SPLIT form
(two launches: columns [0, n/2) first, the rest accumulated on top -- the existing panel path, i.e. twice as many, half as long
workgroups at the price of a second launch).  For each: the GEMV alone (back to back) and the CG iteration it gives.
    usage: small_n_probe.py [N ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LAM_HIP_LIB", os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
SHAPES = [("production: coop R2 W8", dict(gemv_variant=13)), ("coop R2 W4", dict(gemv_variant=10)), ("coop R1 W4 (one row per workgroup)", dict(gemv_variant=9)),
          ("coop R3 W4", dict(gemv_variant=18)), ("coop R2 W4 T2048", dict(gemv_variant=14)), ("coop R2 x2 pairs per workgroup", dict(gemv_variant=23)),
          ("4 rows per wave (tile kernel)", dict(gemv_variant=0)), ("coop R4 W8", dict(gemv_variant=17)),
          ("column split: 2 launches of coop R2 W8", dict(gemv_variant=13, split=True))]


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [12000, 10000, 8192]
    with lam.Solver(lam.F64) as s:
        assert s.get_option("tuning_variants") == 1, "run on the tuning build (make tuning)"
        for n in sorted(sizes, reverse=True):
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            gb = 8.0 * n * n / 1e9
            gemv = {name: [] for name, _ in SHAPES}
            cg = {name: [] for name, _ in SHAPES}
            for _ in range(5):
                for name, o in SHAPES:
                    s.set_option("gemv_variant", o["gemv_variant"])
                    half = (n // 2) // 2 * 2
                    s.set_option("panel_lo", 0)
                    s.set_option("panel_hi", half if o.get("split") else 0)
                    gemv[name].append(s.gemv_only(200))
                    s.cg_init()
                    s.cg_iterate(10, 0.0)
                    cg[name].append(s.cg_iterate(200, 0.0)["t_iter"])
            s.set_option("panel_hi", 0)
            print(f"N={n} fp64: matrix {gb:.2f} GB; medians of 5 interleaved rounds")
            for name, _ in SHAPES:
                tg, tc = sorted(gemv[name])[2], sorted(cg[name])[2]
                print(f"  {name:42s} GEMV alone {tg * 1e6:7.1f} us = {gb / tg / 80:5.1f} % of 8 TB/s    CG iteration {tc * 1e6:7.1f} us "
                      f"(GEMV bytes / iteration time = {gb / tc / 80:5.1f} %)", flush=True)


if __name__ == "__main__":
    main()
