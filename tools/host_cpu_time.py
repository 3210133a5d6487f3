#!/usr/bin/env python3
"""CPU time the host spends on a solve: lam_hip_cg_iterate sleeps between its polls of the iteration's progress word
(round 4; rounds 1-3 spun a core per solve -- in rank mode one spinning core per GPU next to RCCL's proxy threads).
Prints wall time and thread CPU time (CLOCK_THREAD_CPUTIME_ID, library option host_cpu_ns) per iteration.
    usage: host_cpu_time.py [N ...]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [65536, 32768, 10000, 4096]
    with lam.Solver(lam.F64) as s:
        for n in sizes:
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            iters = max(50, min(2000, int(0.6 / (8.0 * n * n / 7e12 + 10e-6))))
            s.cg_init()
            s.cg_iterate(10, 0.0)
            c0, p0 = s.get_option("host_cpu_ns"), time.process_time()
            st = s.cg_iterate(iters, 0.0)
            c1, p1 = s.get_option("host_cpu_ns"), time.process_time()
            wall, cpu, proc = st["t_iter"] * 1e6, (c1 - c0) / iters * 1e-3, (p1 - p0) / iters * 1e6
            print(f"N={n:6d} {iters:5d} iterations: wall {wall:9.1f} us/iteration, calling thread's CPU {cpu:7.1f} us/iteration ({100 * cpu / wall:5.1f} % of wall), "
                  f"whole process {proc:7.1f} us/iteration ({100 * proc / wall:5.1f} %)", flush=True)


if __name__ == "__main__":
    main()
