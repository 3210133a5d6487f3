#!/usr/bin/env python3
"""What the per-iteration HIP events cost a single-shard CG iteration: option "gemv_timing" = 1 (an event pair
around every GEMV, round 2's loop also recorded a third event per iteration for the host's lag wait), T (every
T-th iteration) and 0 (none).  The host now follows the iteration through a progress word in pinned memory, so
with timing off there is no marker packet between the iteration's kernels at all.  Interleaved rounds, one
process.     usage: event_cost.py [N ...]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [10000, 20000, 32768]
    with lam.Solver(lam.F64) as s:
        for n in sorted(sizes, reverse=True):
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            res = {}
            for _ in range(5):
                for fuse in (1, 0):
                    for timing in (1, 4, 8, 0):
                        s.set_option("fuse_update", fuse)
                        s.set_option("gemv_timing", timing)
                        s.cg_init()
                        s.cg_iterate(20, 0.0)
                        st = s.cg_iterate(200, 0.0)
                        res.setdefault((fuse, timing), []).append((st["t_iter"], st["t_gemv"]))
            for (fuse, timing), v in sorted(res.items(), reverse=True):
                v.sort()
                t_iter, t_gemv = v[len(v) // 2]
                print(f"N={n} fuse_update={fuse} gemv_timing={timing}: {t_iter*1e6:9.2f} us/iteration (median of 5), "
                      f"sampled gemv {t_gemv*1e6:9.2f} us", flush=True)


if __name__ == "__main__":
    main()
