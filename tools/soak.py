#!/usr/bin/env python3
"""Long soak of the in-launch hand-overs (reducer workgroup, fused-step broadcast, progress word, peer stores + event joins): for a
list of (dtype, N, shards, exchange, symmetric) configurations, solve the same seeded systems over and over for `seconds` and
compare every solution's bits with the first pass.  A hand-over that ever delivered a stale value, or a bounded wait that ever
expired, shows up as a different hash or an error.  Prints a progress line every ~20 s.
    usage: soak.py [seconds] [many]      many: the configurations with 17 ... 64 shards, the symmetric product and fp32 / bf16 among them"""
import hashlib
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")

CONFIGS = [  # dtype, n, shards, exchange, symmetric, iterations per solve
    ("F64", 2048, 1, None, 0, 400), ("F64", 10000, 1, None, 0, 300), ("F64", 4096, 3, 1, 0, 300), ("F64", 4100, 4, 0, 0, 200),
    ("F32", 8192, 2, 1, 0, 200), ("BF16", 6000, 3, 1, 0, 100), ("F64", 8192, 8, 1, 2, 200), ("F64", 1039, 33, 1, 0, 200),
    ("F64", 32768, 1, None, 2, 60), ("F64", 32768, 8, 1, 0, 60), ("F64", 65536, 1, None, 0, 30),
]


CONFIGS_MANY = [
    ("F64", 4099, 33, 1, 2, 150), ("F64", 4100, 64, 1, 0, 150), ("F32", 8192, 64, 1, 2, 100), ("F64", 2051, 17, 0, 0, 150), ("BF16", 6000, 24, 1, 2, 80),
    ("F64", 64, 64, 1, 0, 60), ("F32", 3001, 40, 0, 0, 100), ("F64", 16384, 16, 1, 2, 60),
]


def one(s, cfg, seed):
    dt, n, shards, exchange, sym, iters = cfg
    s.generate_random_spd(n, seed, 1e7)
    s.generate_random_rhs(seed + 1)
    s.cg_init()
    st = s.cg_iterate(iters, 0.0)
    x = s.solution()
    assert st["num_iters"] == iters + 1 and np.all(np.isfinite(x)) and 0 < st["rel_err"] < 10.0, (cfg, st)
    h = hashlib.sha256(x.tobytes())
    h.update(np.float64(st["rel_err"]).tobytes())
    return h.hexdigest()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    configs = CONFIGS_MANY if len(sys.argv) > 2 and sys.argv[2] == "many" else CONFIGS
    t0 = time.time()
    ctx, ref, solves, bad = [], {}, 0, 0
    for cfg in configs:
        dt, n, shards, exchange, sym, iters = cfg
        s = lam.Solver(getattr(lam, dt), device_ids=[0] * shards)
        if exchange is not None:
            s.set_option("exchange", exchange)
        s.set_problem(n)
        s.set_option("symmetric", sym)
        ctx.append(s)
    last = t0
    rnd = 0
    while time.time() - t0 < seconds:
        for i, (s, cfg) in enumerate(zip(ctx, configs)):
            for seed in (21, 57):
                # alternate the launch-chain options that must not change a bit
                s.set_option("fuse_update", (rnd + i) % 2 if rnd % 3 == 2 else 1)
                s.set_option("gemv_timing", (8, 1, 0)[rnd % 3])
                h = one(s, cfg, seed)
                key = (i, seed)
                if key not in ref:
                    ref[key] = h
                elif ref[key] != h:
                    bad += 1
                    print(f"MISMATCH round {rnd} config {cfg} seed {seed}", flush=True)
                solves += 1
        rnd += 1
        if time.time() - last > 20:
            last = time.time()
            print(f"# {time.time() - t0:6.0f} s: {rnd} rounds, {solves} solves, {bad} mismatches", flush=True)
    iters_total = sum(c[5] for c in configs) * 2 * rnd
    print(f"# soak: {time.time() - t0:.0f} s, {rnd} rounds x {len(configs)} configurations x 2 systems = {solves} solves, {iters_total} CG iterations, {bad} mismatches")
    for s in ctx:
        s.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
