#!/usr/bin/env python3
"""Randomised cross-check of the launch / enqueue options that must not change a single bit: for random (dtype, N,
shards, seed, exchange) the solve with default options against the solve with a random combination of the options
that only change HOW an iteration is launched and enqueued, and with the solve cut into random cg_iterate chunks.
Product library: fuse_update, gemv_timing, exchange_join.  Tuning build (LAM_HIP_LIB=.../liblam_hip_tuning.so) adds
the experiments that live there: finalize, host_threads, exchange_hub, persistent / persist_chunk.
    usage: fuzz_options.py [cases] [seed] [many]      many: 9 ... 64 shards per case instead of 1 ... 5"""
import importlib
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def run(dt, n, shards, seed, opts, chunks):
    with lam.Solver(dt, device_ids=[0] * shards) as s:
        s.generate_random_spd(n, seed, 500.0)
        s.generate_random_rhs(seed + 1)
        for k, v in opts.items():
            s.set_option(k, v)
        s.cg_init()
        st = None
        for c in chunks:
            st = s.cg_iterate(c, 0.0)
        eff = {k: s.get_option(k) for k in ("fuse_effective", "persistent_effective", "exchange_effective", "symmetric_effective")}
        return s.solution(), st["rel_err"], st["num_iters"], eff, s.true_residual()


def main():
    with lam.Solver(lam.F64) as s0:
        TUNING = s0.get_option("tuning_variants") == 1      # noqa: N806
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    many = len(sys.argv) > 3 and sys.argv[3] == "many"
    # LAM_HIP_DIRECT_SAME_DEVICE=1 (+ GPU_MAX_HW_QUEUES >= 2 x shards) in the environment: multi-shard cases also try exchange 2
    direct = os.environ.get("LAM_HIP_DIRECT_SAME_DEVICE", "0") not in ("", "0")
    bad = 0
    for case in range(cases):
        dt, dname = rng.choice(((lam.F64, "f64"), (lam.F64, "f64"), (lam.F32, "f32"), (lam.BF16, "bf16")))
        shards = rng.randint(9, 64) if many else rng.choice((1, 1, 1, 2, 3, 5))
        n = rng.choice((rng.randint(shards * 2, 300), rng.randint(300, 9000), rng.choice((256, 1024, 4096, 4098, 8192, 10000))))
        n = max(n, shards)
        total = min(rng.randint(1, 60), max(1, n // 3))      # stay short of exact convergence (r = 0 gives 0/0, as in the reference)
        seed = rng.randint(1, 10 ** 6)
        # the exchange is part of the CASE (gather-Ap sums r.r over full-length partials: other bits than the sliced form)
        exchange = rng.choice((0, 1)) if shards > 1 else 0
        base = {"exchange": exchange} if shards > 1 else {}
        # the symmetric product is part of the CASE too (another summation order): one case in three runs on it (one shard: the
        # upper triangle; several on the gather-Ap exchange: cyclic half windows; elsewhere the option has no effect)
        if rng.random() < 0.34:
            base["symmetric"] = 2
        ref = run(dt, n, shards, seed, base, [total])
        opts = dict(base, fuse_update=rng.choice((0, 1)), gemv_timing=rng.choice((0, 1, 3, 8)), exchange_join=rng.choice((0, 1)))
        if TUNING:
            opts.update({"finalize": rng.choice((1, 1, 0)), "host_threads": rng.choice((0, 1)), "exchange_hub": rng.choice((0, 1)),
                         "persistent": 0 if base.get("symmetric") else rng.choice((0, 1)), "persist_chunk": rng.choice((1, 2, 7, 32))})
            if opts["persistent"] and not base.get("symmetric"):
                # the persistent launch shares the tile body of GEMV shape 10 (fp64's default is 13): same shape on both sides
                base = dict(base, gemv_variant=10)
                opts["gemv_variant"] = 10
                ref = run(dt, n, shards, seed, base, [total])
        if direct and shards > 1 and exchange == 0 and not base.get("symmetric"):
            # the in-kernel flag exchange between the local shards, one GEMV launch per shard (the own-slice panel of
            # overlap = 1 adds a row's products in another order): same bits as the event exchange; needs finalize = 1
            opts.update({"exchange": 2, "overlap": 0, "finalize": 1})
        chunks, left = [], total
        while left > 0:
            c = rng.randint(1, left)
            chunks.append(c)
            left -= c
        got = run(dt, n, shards, seed, opts, chunks)
        same = ref[0].tobytes() == got[0].tobytes() and ref[1] == got[1] and ref[2] == got[2]
        finite = bool(np.all(np.isfinite(got[0])))
        res_ok = abs(got[4] - got[1]) <= 1e-6 * max(got[1], 1e-30) + (1e-12 if dname == "f64" else 1e-4)
        ok = same and finite and res_ok
        bad += not ok
        print(f"{'ok  ' if ok else 'FAIL'} case {case}: {dname} N={n} shards={shards} iters={total} chunks={chunks} opts={opts} "
              f"effective={got[3]} rel_err={got[1]:.3e} true={got[4]:.3e}", flush=True)
    print(f"# {cases - bad} of {cases} cases bit-identical to the default options")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
