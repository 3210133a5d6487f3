#!/usr/bin/env python3
"""Rank mode at the smallest sizes -- one row per rank, one or a few extra rows on the last rank -- on the stream-ordered RCCL double,
every exchange, the symmetric product on gather-Ap: ranks identical, the oracle with the same number of ranks, the host-recomputed
residual.  (Under tests/: the comparison runs the oracle, which only tests may.)
    usage: LD_PRELOAD=tests/mock_rccl/libmock_rccl_async.so python tests/mock_rccl/tiny_ranks.py"""
import json
import os
import subprocess
import sys

RUN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_ranks.py")


def main():
    bad = 0
    for P, n in ((4, 4), (4, 5), (3, 3), (8, 8), (8, 15), (2, 2), (5, 9), (7, 50), (2, 3)):
        for ex, sym in ((0, 0), (1, 0), (1, 2), (2, 0)):
            r = subprocess.run([sys.executable, RUN, str(P), str(n), "spd", "--exchange", str(ex), "--symmetric", str(sym), "--iters", "60", "--tol", "1e-12", "--cond", "20"],
                               capture_output=True, text=True, timeout=120)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception:   # noqa: BLE001
                d = {"error": (r.stdout + r.stderr)[-300:]}
            ok = ("error" not in d and d["ranks_identical"] and d["converged"] and d["converged_oracle"] and abs(d["iters"] - d["iters_oracle"]) <= 2
                  and d["residual_numpy"] < 1e-11 and d["x_vs_oracle"] < 1e-10 and d["exchange_effective"] == [ex] * P
                  and d["symmetric_effective"] == [1 if sym else 0] * P)
            bad += not ok
            print(("ok   " if ok else "FAIL ") + f"P={P} N={n} exchange {ex} symmetric {sym}: " +
                  str({k: d.get(k) for k in ("iters", "iters_oracle", "x_vs_oracle", "residual_numpy", "exchange_effective", "symmetric_effective", "error") if k in d})[:400], flush=True)
    print(f"# tiny_ranks: {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
