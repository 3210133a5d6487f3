#!/usr/bin/env python3
"""Context re-use fuzz: ONE long-lived context per topology goes through a random sequence of problems -- new sizes (growing and
shrinking), new matrices of the same size, another exchange, the symmetric product on and off, the launch-chain options -- and
after every solve its solution must be, bit for bit, what a FRESH context gives for that problem alone.  Anything that outlives
its problem (a task list, a gather buffer, a partial buffer sized for the old N, a row pitch, a flag) shows up as a difference.
    usage: reuse_fuzz.py [steps] [seed]"""
import importlib
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def apply(s, prob, new_size):
    dt, n, gen, seed, opts, iters = prob
    if gen == "spd":
        s.generate_random_spd(n, seed, 400.0, keep_problem=not new_size)
        s.generate_random_rhs(seed + 1)
    elif gen == "tridiag":
        s.generate_matrix(n)
        s.generate_rhs()
    else:
        rng = np.random.default_rng(seed)
        q, _ = np.linalg.qr(rng.uniform(-1, 1, (n, n)))
        A = (q * np.exp(1.5 * rng.uniform(-1, 1, n))) @ q.T
        s.set_matrix(0.5 * (A + A.T))
        s.set_rhs(rng.uniform(-1, 1, n))
    for k, v in opts.items():
        s.set_option(k, v)
    s.cg_init()
    st = s.cg_iterate(iters, 0.0)
    eff = (s.get_option("exchange_effective"), s.get_option("symmetric_effective"), s.get_option("fuse_effective"))
    return s.solution().tobytes(), st["rel_err"], eff


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bad = 0
    for dt_name, shards in (("F64", 1), ("F64", 3), ("F32", 2), ("BF16", 1), ("F64", 8), ("F64", 5)):
        dt = getattr(lam, dt_name)
        long_lived = lam.Solver(dt, device_ids=[0] * shards)
        last_n = None
        for step in range(steps):
            if last_n is not None and rng.random() < 0.35:
                n = last_n                                        # a new matrix (or only new options) at the same size
            else:
                n = rng.choice((rng.randint(shards, 200), rng.randint(200, 3000), rng.choice((1024, 4096, 4100, 6000, 8192, 2049))))
            gen = rng.choice(("spd", "spd", "tridiag", "upload" if n <= 1500 else "spd"))
            opts = {"symmetric": rng.choice((0, 0, 2)), "fuse_update": rng.choice((1, 1, 0)), "gemv_timing": rng.choice((8, 0, 1))}
            if shards > 1:
                opts["exchange"] = rng.choice((1, 1, 0))
                opts["exchange_join"] = rng.choice((1, 0))
            prob = (dt, n, gen, rng.randint(1, 10 ** 6), opts, min(rng.randint(1, 40), max(1, n // 3)))
            got = apply(long_lived, prob, new_size=n != last_n)
            with lam.Solver(dt, device_ids=[0] * shards) as fresh:
                want = apply(fresh, prob, new_size=True)
            ok = got == want
            bad += not ok
            if not ok or step % 20 == 19:
                print(f"{'ok  ' if ok else 'FAIL'} {dt_name} x{shards} step {step}: N={n} {gen} {opts} iters={prob[5]} effective={got[2]} rel_err={got[1]:.3e}"
                      + ("" if ok else f"   fresh context: effective={want[2]} rel_err={want[1]:.3e}"), flush=True)
            last_n = n
        long_lived.close()
    print(f"# reuse_fuzz: 6 contexts x {steps} problems each, {bad} differences from a fresh context")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
