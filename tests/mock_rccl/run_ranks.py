#!/usr/bin/env python3
"""Runs P rank-mode contexts as P threads of this process (all on GPU 0) on top of the mock RCCL that
is LD_PRELOADed in front of librccl.so (tests/mock_rccl/mock_rccl.cpp).  Prints one JSON line.
usage: run_ranks.py P N mode [overlap [exchange]]   with mode in {tridiag, spd}"""
import importlib, json, os, sys, threading
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    P, n, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    overlap = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    exchange = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    uid = lam.get_unique_id()
    assert uid.startswith(b"mock-rccl-"), "the mock is not in front of librccl"
    out = [None] * P
    errs = []

    def rank_main(r):
        try:
            with lam.Solver(lam.F64, rank=r, nranks=P, device_id=0, unique_id=uid) as s:
                if mode == "tridiag":
                    s.generate_matrix(n)
                    s.generate_rhs()
                    tol, iters = 1e-9, 10000
                else:
                    s.generate_random_spd(n, 99, 200.0)
                    s.generate_random_rhs(100)
                    tol, iters = 1e-10, 2000
                s.set_option("overlap", overlap)
                s.set_option("exchange", exchange)
                conv = s.solve(iters, tol)
                x = s.solution()                 # collective
                res = s.true_residual()          # collective
                y = s.gemv(np.arange(n, dtype=np.float64) / n)   # collective
                out[r] = dict(conv=conv, iters=s.stats["num_iters"], err=s.stats["rel_err"], x=x, res=res, y=y,
                              part=s.partition(r))
        except Exception as e:                   # noqa: BLE001
            errs.append(f"rank {r}: {e!r}")

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(240)
    if errs or any(o is None for o in out):
        print(json.dumps({"error": errs or "a rank did not finish"}))
        sys.exit(1)
    # reference: the same system on one shard, no RCCL
    with lam.Solver(lam.F64) as s:
        if mode == "tridiag":
            s.generate_matrix(n); s.generate_rhs(); s.solve(10000, 1e-9)
        else:
            s.generate_random_spd(n, 99, 200.0); s.generate_random_rhs(100); s.solve(2000, 1e-10)
        x1, it1 = s.solution(), s.stats["num_iters"]
        y1 = s.gemv(np.arange(n, dtype=np.float64) / n)
    same = all(np.array_equal(out[0]["x"], o["x"]) and o["iters"] == out[0]["iters"] and o["err"] == out[0]["err"]
               and np.array_equal(out[0]["y"], o["y"]) for o in out)
    print(json.dumps({
        "P": P, "n": n, "ranks_identical": bool(same), "iters": out[0]["iters"], "iters_single": it1,
        "converged": bool(out[0]["conv"]), "true_residual": out[0]["res"], "rel_err": out[0]["err"],
        "x_vs_single": float(np.linalg.norm(out[0]["x"] - x1) / np.linalg.norm(x1)),
        "gemv_vs_single": float(np.max(np.abs(out[0]["y"] - y1)) / np.max(np.abs(y1))),
        "partition": [list(o["part"]) for o in out]}))


if __name__ == "__main__":
    main()
