#!/usr/bin/env python3
"""Two (or more) GEMV shapes against each other over N, interleaved, in ONE grow-only context (largest N first: nothing is freed
between sizes).  Tuning build.   usage: variant_vs_size.py [--variants 10,13] [--dtype f64] [N ...]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LAM_HIP_LIB", os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int)
    ap.add_argument("--variants", default="10,13")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--cg", type=int, default=0)
    a = ap.parse_args()
    sizes = a.sizes or [65536, 61440, 49152, 40000, 32768, 30000, 20000, 16384, 10000, 8192]
    variants = [int(v) for v in a.variants.split(",")]
    dt = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    es = {"f64": 8, "f32": 4, "bf16": 2}[a.dtype]
    with lam.Solver(dt) as s:
        for n in sorted(sizes, reverse=True):
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            reps = max(10, min(400, int(0.1 / (es * n * n / 7e12))))
            res = {v: [] for v in variants}
            for _ in range(5):
                for v in variants:
                    s.set_option("gemv_variant", v)
                    res[v].append(s.gemv_only(reps))
            gb = es * float(n) * n / 1e9
            line = f"N={n:6d}:"
            for v in variants:
                t = sorted(res[v])[2]
                line += f"  v{v} {gb / t / 80:6.2f} %"
            if a.cg:
                for v in variants:
                    s.set_option("gemv_variant", v)
                    ts = []
                    for _ in range(3):
                        s.cg_init(); s.cg_iterate(5, 0.0)
                        ts.append(s.cg_iterate(a.cg, 0.0)["t_iter"])
                    line += f"  CG v{v} {sorted(ts)[1] * 1e6:8.1f} us/it"
            print(line, flush=True)


if __name__ == "__main__":
    main()
