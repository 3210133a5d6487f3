#!/usr/bin/env python3
"""GEMV roofline probe: GB/s of the GEMV kernel variants (lam_hip_gemv_only), interleaved rounds in
one process (cdna_hip_programming.md rule 24).   usage: gemv_probe.py [N ...] [--variants 0,1,..] [--dtype f64|f32|bf16]"""
import argparse, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the tuning shapes live in the tuning build of the library (`make tuning`), not in the product library
os.environ.setdefault("LAM_HIP_LIB", os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
NAMES = {0: "R4 T4096 lds rot (default)", 1: "R2 T4096", 2: "R8 T4096", 3: "R4 T2048", 4: "R4 T8192", 5: "R2 T8192",
         6: "R4 T4096 p-from-L2", 7: "R4 T4096 no-rot", 8: "R1 T4096", 9: "coop R1", 10: "coop R2", 11: "coop R4",
         12: "coop R8", 13: "coop R2 W8", 14: "coop R2 T2048", 15: "coop R2 W8 T8192 u8", 16: "coop R2 W4 T8192 u8",
         17: "coop R4 W8", 18: "coop R3", 25: "coop R4 T8192 W4", 26: "coop R4 T8192 W8", 27: "coop R8 T8192 W8", 19: "MFMA bf16 R2 split3", 20: "MFMA bf16 R2 split1 (p->bf16)",
         21: "MFMA bf16 R4 split3", 22: "MFMA bf16 R1 split3", 23: "coop R2 x2 pairs/WG", 24: "coop R2 x4 pairs/WG"}

def check(a):
    import numpy as np
    dt = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    tol = {"f64": 1e-13, "f32": 32 * 2.0 ** -24, "bf16": 32 * 2.0 ** -24}[a.dtype]
    variants = [int(v) for v in a.variants.split(",")]
    bad = 0
    for n in a.sizes:
        rng = np.random.default_rng(n)
        A = rng.uniform(-1, 1, (n, n))
        x = rng.uniform(-1, 1, n)
        with lam.Solver(dt) as s:
            assert s.get_option("tuning_variants") == 1, "this is not the tuning build (make tuning)"
            s.set_matrix(A)
            Ad = s.download_rows(0, n).astype(np.float64)
            xs = x.astype(s.vec_dtype)
            y64 = Ad @ xs.astype(np.float64)
            scale = np.abs(Ad) @ np.abs(xs.astype(np.float64))
            for v in variants:
                s.set_option("gemv_variant", v)
                err = float(np.max(np.abs(s.gemv(xs).astype(np.float64) - y64) / scale))
                t = 2.0 ** -8 if v == 20 else tol            # variant 20 rounds p to bf16
                ok = err <= t
                bad += not ok
                print(f"N={n} {a.dtype} v{v} [{NAMES.get(v, '?')}] kernel {s.gemv_kernel_name()}: max err/scale {err:.3e} "
                      f"(tolerance {t:.1e}) {'ok' if ok else 'FAIL'}", flush=True)
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int, default=[32768])
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--nt", default="1")
    ap.add_argument("--rows", type=int, default=0, help="use only the first ROWS rows (a shard of a P-way split)")
    ap.add_argument("--cg", type=int, default=0, help="also time this many CG iterations per variant")
    ap.add_argument("--check", action="store_true", help="no timing: y = A x of every variant against an fp64 product of the "
                    "stored matrix (tests/test_gpu_parity.py runs this on the tuning build)")
    a = ap.parse_args()
    if a.check:
        return check(a)
    dt = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    es = {"f64": 8, "f32": 4, "bf16": 2}[a.dtype]
    variants = [int(v) for v in a.variants.split(",")]
    nts = [int(v) for v in a.nt.split(",")]
    for n in a.sizes:
        with lam.Solver(dt) as s:
            s.generate_random_spd(n, 1234, 1e4)
            s.generate_random_rhs(1235)
            s.cg_init()
            res = {}
            rows = a.rows if a.rows > 0 else n
            s.set_option("probe_rows", a.rows)
            for _ in range(a.rounds):
                for v in variants:
                    for nt in nts:
                        s.set_option("gemv_variant", v)
                        s.set_option("nt_loads", nt)
                        res.setdefault((v, nt), []).append(s.gemv_only(a.reps))
            for (v, nt), ts in sorted(res.items()):
                ts = sorted(ts)
                med, best = ts[len(ts) // 2], ts[0]
                gb = es * rows * n / 1e9
                print(f"N={n} rows={rows} {a.dtype} v{v} nt={nt} [{NAMES.get(v,'?'):26s}] median {med*1e3:8.4f} ms {gb/med:7.1f} GB/s "
                      f"({gb/med/80:5.1f}% of 8 TB/s)  best {gb/best:7.1f} GB/s", flush=True)
            if a.cg > 0:
                for v in variants:
                    s.set_option("gemv_variant", v)
                    s.set_option("nt_loads", nts[0])
                    ts = []
                    for _ in range(3):
                        s.cg_init()
                        s.cg_iterate(5, 0.0)
                        st = s.cg_iterate(a.cg, 0.0)
                        ts.append((st["t_iter"], st["t_gemv"]))
                    ts.sort()
                    print(f"N={n} {a.dtype} v{v} CG {a.cg} iters: {ts[1][0]*1e3:.4f} ms/iter (gemv {ts[1][1]*1e3:.4f} ms, "
                          f"other {(ts[1][0]-ts[1][1])*1e6:.1f} us)", flush=True)

if __name__ == "__main__":
    sys.exit(main() or 0)
