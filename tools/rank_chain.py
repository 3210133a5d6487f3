#!/usr/bin/env python3
"""The per-iteration launch chain of the one-process-per-GPU mode through REAL RCCL with one rank
(LAM_HIP_FORCE_RCCL=1), on the shard shape of an 8-way split of N=65536 (4.29 GB of matrix: N=23168 on
one rank streams the same bytes per GEMV).  Run it under `rocprofv3 --kernel-trace` and feed the trace to
tools/summarize_chain.py; run it bare for the wall-clock table.

    variants: exchange 0 with finalize 0 (round-1 chain: 5 launches + 3 collectives), exchange 0 with
    finalize 1 (3 launches + 3 collectives), exchange 1 with finalize 0 / 1, each with overlap 1 and 0."""
import importlib, os, sys
os.environ["LAM_HIP_FORCE_RCCL"] = "1"
# the experiments this tool measures (host_threads / exchange_hub / persistent / finalize = 0) live in the tuning build of the library
os.environ.setdefault("LAM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 23168
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    with lam.Solver(lam.F64, rank=0, nranks=1, device_id=0, unique_id=None) as s:
        s.generate_random_spd(n, 11, 1e6)
        s.generate_random_rhs(12)
        print(f"rccl version {lam.rccl_version()}  N={n}  kernel {s.gemv_kernel_name()}")
        for exchange, finalize, overlap, fuse, sym in ((0, 0, 1, 0, 0), (0, 1, 1, 0, 0), (0, 1, 0, 0, 0), (1, 0, 1, 0, 0), (1, 1, 1, 0, 0), (1, 1, 1, 1, 0),
                                                       (2, 1, 1, 0, 0), (2, 1, 0, 0, 0), (2, 1, 1, 1, 0), (2, 1, 0, 1, 0), (1, 1, 1, 1, 1)):
            s.set_option("exchange", exchange); s.set_option("finalize", finalize); s.set_option("overlap", overlap)
            s.set_option("fuse_update", fuse)
            s.set_option("symmetric", sym)      # 1: the symmetric product on the rank-mode path (cyclic half windows; here one rank = the whole matrix)
            best = None
            for _ in range(3):
                s.cg_init()
                s.cg_iterate(10, 0.0)
                st = s.cg_iterate(iters, 0.0)
                if best is None or st["t_iter"] < best["t_iter"]:
                    best = st
            print(f"exchange={exchange} finalize={finalize} overlap={overlap} fuse={fuse} symmetric={sym}: {best['t_iter']*1e3:.4f} ms/iter, gemv {best['t_gemv']*1e3:.4f} ms, "
                  f"other {(best['t_iter']-best['t_gemv'])*1e6:.1f} us", flush=True)
        s.set_option("symmetric", 0)
    # single-shard chain (no RCCL) for reference
    os.environ.pop("LAM_HIP_FORCE_RCCL")
    with lam.Solver(lam.F64) as s:
        s.generate_random_spd(n, 11, 1e6); s.generate_random_rhs(12)
        for fuse in (0, 1, 0, 1):
            s.set_option("fuse_update", fuse)
            s.cg_init(); s.cg_iterate(10, 0.0); st = s.cg_iterate(iters, 0.0)
            print(f"single shard (no exchange) fuse={fuse}: {st['t_iter']*1e3:.4f} ms/iter, gemv {st['t_gemv']*1e3:.4f} ms, other {(st['t_iter']-st['t_gemv'])*1e6:.1f} us")


if __name__ == "__main__":
    main()
