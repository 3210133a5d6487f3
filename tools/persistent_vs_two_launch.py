#!/usr/bin/env python3
"""The whole-iteration persistent launch (option "persistent", cg_persist_kernel) against the two-launch chain:
wall time per iteration and the GEMV phase as each form times it, interleaved rounds in one process, same bits checked.
    usage: persistent_vs_two_launch.py [N ...]"""
import importlib
import os
import sys

import numpy as np

# the experiments this tool measures (host_threads / exchange_hub / persistent / finalize = 0) live in the tuning build of the library
os.environ.setdefault("LAM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [65536, 32768, 20000, 10000, 8192, 4096]
    with lam.Solver(lam.F64) as s:
        for n in sorted(sizes, reverse=True):
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            for _ in range(6):
                s.gemv_only(10)
            res, xs = {}, {}
            for _ in range(5):
                for persistent, chunk in ((0, 32), (1, 32), (1, 8)):
                    s.set_option("persistent", persistent)
                    s.set_option("persist_chunk", chunk)
                    s.set_option("gemv_timing", 0 if persistent else 8)
                    s.cg_init()
                    s.cg_iterate(32, 0.0)
                    st = s.cg_iterate(192, 0.0)
                    res.setdefault((persistent, chunk), []).append((st["t_iter"], st["t_gemv"]))
                    xs[(persistent, chunk)] = s.solution()
            w = s.get_option("persistent_workers")
            for key, v in sorted(res.items()):
                v.sort()
                t_iter, t_gemv = v[len(v) // 2]
                same = bool(np.array_equal(xs[key], xs[(0, 32)]))
                print(f"N={n} persistent={key[0]} iterations/launch={key[1] if key[0] else 1}: {t_iter*1e6:9.2f} us/iteration  "
                      f"(GEMV phase {t_gemv*1e6:9.2f} us, rest {(t_iter-t_gemv)*1e6:6.2f})  workers {w if key[0] else '-'}  same bits as two-launch: {same}", flush=True)


if __name__ == "__main__":
    main()
