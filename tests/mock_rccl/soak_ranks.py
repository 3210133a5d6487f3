#!/usr/bin/env python3
"""Soak of the rank mode under asynchronous collectives: P rank contexts as P threads of this process (all on GPU 0, the
stream-ordered RCCL double LD_PRELOADed in front of librccl.so, see run_ranks.py), each solving the same seeded systems TO
CONVERGENCE over and over -- the stop has to be agreed on by all ranks every time, at an iteration nobody knows in advance -- on
the three exchanges in rotation.  Every rank's solution must be the bits of rank 0's, and every pass the bits of the first.
    usage: soak_ranks.py P N rounds [f64|f32|bf16]      (every other gather-Ap round runs the symmetric product)"""
import hashlib
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    P, n, rounds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dname = sys.argv[4] if len(sys.argv) > 4 else "f64"
    DT = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[dname]
    tol = 1e-10 if dname == "f64" else 2e-5
    uid = lam.get_unique_id()
    assert uid.startswith(b"/lam_mock_") or uid.startswith(b"mock-rccl-"), "the mock is not in front of librccl"
    hashes = [[None] * rounds for _ in range(P)]
    # SOAK_PATTERN="0:1,1:0": the (exchange:overlap) pairs to rotate through instead of the default rotation (debugging)
    pattern = [tuple(int(v) for v in item.split(":")) for item in os.environ.get("SOAK_PATTERN", "").split(",") if item]
    errs = []
    t0 = time.time()

    def rank_main(r):
        try:
            with lam.Solver(DT, rank=r, nranks=P, device_id=0, unique_id=uid) as s:
                s.set_problem(n)
                for name, val in ((kv.split("=")[0], int(kv.split("=")[1])) for kv in os.environ.get("SOAK_OPTIONS", "").split(",") if kv):
                    s.set_option(name, val)               # e.g. SOAK_OPTIONS="fuse_update=0,finalize=0" (debugging)
                for k in range(rounds):
                    exchange, overlap = pattern[k % len(pattern)] if pattern else ((0, 1, 2)[k % 3], (k // 3) % 2)
                    s.set_option("exchange", exchange)
                    s.set_option("overlap", overlap)
                    sym = 2 if exchange == 1 and (k // 3) % 2 == 1 else 0
                    s.set_option("symmetric", sym)
                    seed = 31 + (k % 3) + 7 * ((k // 3) % 4)          # 12 systems, each met on every exchange in turn
                    if r == 0 and os.environ.get("SOAK_VERBOSE"):
                        print(f"round {k}: exchange {exchange} overlap {overlap} seed {seed}", flush=True)
                    s.generate_random_spd(n, seed, 300.0 + 50.0 * (seed % 5), keep_problem=True)     # (no hipFree between solves: see _capi.py)
                    s.generate_random_rhs(seed + 1)
                    assert s.solve(2000, tol), f"round {k}: not converged"
                    assert s.get_option("symmetric_effective") == (1 if sym else 0)
                    x = s.solution()                  # collective
                    h = hashlib.sha256(x.tobytes())
                    h.update(str(s.stats["num_iters"]).encode())
                    hashes[r][k] = (seed, exchange, h.hexdigest(), s.stats["num_iters"], s.get_option("exchange_effective"), s.get_option("direct_fallbacks"), sym)
                    if r == 0 and k % 25 == 24:
                        print(f"# {time.time() - t0:6.0f} s: {k + 1} solves per rank", flush=True)
        except Exception as e:                   # noqa: BLE001
            errs.append(f"rank {r}: {e!r}")

    if os.environ.get("SOAK_DUMP_AFTER"):          # debugging: where is every rank's thread after this many seconds?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SOAK_DUMP_AFTER"]), exit=False)
    th = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(3000)
    if errs or any(h is None for hs in hashes for h in hs):
        print(json.dumps({"error": errs or "a rank did not finish"}))
        return 1
    bad = 0
    first = {}
    iters = 0
    for k in range(rounds):
        if any(hashes[r][k] != hashes[0][k] for r in range(P)):
            bad += 1
            print(f"MISMATCH between ranks in round {k}: {[hashes[r][k] for r in range(P)]}", flush=True)
        seed, exchange, hx, it, eff, fb, sym = hashes[0][k]
        iters += it
        if eff != exchange or fb:
            bad += 1
            print(f"round {k}: exchange {exchange} ran as {eff} (fallbacks {fb})", flush=True)
        # exchanges 0 and 2 add in the same order (bit-identical to each other); gather-Ap sums r.r over full-length partials
        key = (seed, 1 if exchange == 1 else 0, sym)
        if pattern:
            key = (seed, exchange, k % len(pattern), sym)
        if first.setdefault(key, hx) != hx:
            bad += 1
            print(f"MISMATCH with the first pass in round {k}: seed {seed} exchange {exchange}", flush=True)
    print(f"# soak_ranks: P={P} N={n} {dname}: {rounds} solves per rank to convergence, {iters} CG iterations, {time.time() - t0:.0f} s, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
