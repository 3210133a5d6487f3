#!/usr/bin/env python3
"""Runs P rank-mode contexts as P threads of this process (all on GPU 0) on top of a mock RCCL that is
LD_PRELOADed in front of librccl.so (tests/mock_rccl/mock_rccl_async.hip: stream-ordered, nothing
synchronises hosts or streams -- the semantics of the real library).  Prints one JSON line.

usage: run_ranks.py P N mode [--overlap 0|1] [--exchange 0|1|2] [--finalize 0|1] [--symmetric 0|1|2] [--iters K] [--tol T]
                             [--chunk C]    with mode in {tridiag, spd, file}
--chunk C runs the solve as repeated lam_hip_cg_iterate(C) calls (the stop has to be noticed across
calls, and every rank must leave the loop after the same call).
mode file: --matrix / --rhs name files in the reference's format (every rank reads its own row block), N is ignored.
Unless --no-single is given the same system is also solved on ONE shard and -- n <= 5000 -- by the CPU ORACLE with the
same number of (emulated) ranks (oracle.cg_solve(..., P=P): the reference's MPI recurrence, CPU_MPI_OMP.hpp:71-142), and
the residual is recomputed with numpy: every multi-rank case is tied to the reference algorithm, not only to another
run of the HIP path."""
import argparse, hashlib, importlib, json, os, sys, threading
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("P", type=int)
    ap.add_argument("n", type=int)
    ap.add_argument("mode", choices=["tridiag", "spd", "file"])
    ap.add_argument("--matrix", default=None)
    ap.add_argument("--rhs", default=None)
    ap.add_argument("--save-x", default=None, help="write rank 0's solution vector here (.npy)")
    ap.add_argument("--overlap", type=int, default=1)
    ap.add_argument("--exchange", type=int, default=0)
    ap.add_argument("--finalize", type=int, default=1)
    ap.add_argument("--fuse", type=int, default=1)
    ap.add_argument("--symmetric", type=int, default=0)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--tol", type=float, default=None)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--cond", type=float, default=200.0)
    ap.add_argument("--dtype", choices=["f64", "f32", "bf16"], default="f64")
    ap.add_argument("--no-single", action="store_true", help="skip the single-shard comparison run")
    ap.add_argument("--quick-destroy", type=int, default=0, help="K > 0: cg_init + cg_iterate(K) and destroy the context at once, "
                    "nothing collective in between (the direct exchange must have quiesced by itself); prints {ok: true}")
    a = ap.parse_args()
    P, n, mode = a.P, a.n, a.mode
    if mode == "file":
        with open(a.matrix, "rb") as fh:
            n = int(np.frombuffer(fh.read(8), dtype=np.uint64)[0])
    DT = {"f64": lam.F64, "f32": lam.F32, "bf16": lam.BF16}[a.dtype]
    vdt = np.float64 if a.dtype == "f64" else np.float32
    tol = a.tol if a.tol is not None else (1e-9 if mode == "tridiag" else 1e-10)
    iters = a.iters if a.iters is not None else (10000 if mode == "tridiag" else 2000)
    uid = lam.get_unique_id()
    assert uid.startswith(b"/lam_mock_") or uid.startswith(b"mock-rccl-"), "the mock is not in front of librccl"
    out = [None] * P
    errs = []
    xprobe = (np.arange(n, dtype=np.float64) / n).astype(vdt)

    def setup(s):
        if mode == "file":
            assert s.load_matrix_from_file(a.matrix) and s.load_rhs_from_file(a.rhs)
        elif mode == "tridiag":
            s.generate_matrix(n)
            s.generate_rhs()
        else:
            s.generate_random_spd(n, 99, a.cond)
            s.generate_random_rhs(100)

    def rank_main(r):
        try:
            with lam.Solver(DT, rank=r, nranks=P, device_id=0, unique_id=uid) as s:
                setup(s)
                s.set_option("overlap", a.overlap)
                s.set_option("exchange", a.exchange)
                if a.finalize != 1:
                    s.set_option("finalize", a.finalize)     # 0 exists in the tuning build only
                s.set_option("fuse_update", a.fuse)
                s.set_option("symmetric", a.symmetric)
                if a.quick_destroy > 0:
                    s.cg_init()
                    st = s.cg_iterate(a.quick_destroy, 0.0)
                    out[r] = dict(iters=st["num_iters"], err=st["rel_err"], eff=s.get_option("exchange_effective"))
                    return                                   # leaves the `with`: lam_hip_destroy right behind the last iteration
                if a.chunk > 0:
                    s.cg_init()
                    done, conv, calls = 0, False, 0
                    while done < iters and not conv:
                        st = s.cg_iterate(min(a.chunk, iters - done), tol)
                        done += a.chunk
                        calls += 1
                        conv = bool(st["converged"])
                else:
                    conv, calls = s.solve(iters, tol), 1
                x = s.solution()                 # collective
                res = s.true_residual()          # collective
                y = s.gemv(xprobe)               # collective
                out[r] = dict(conv=conv, iters=s.stats["num_iters"], err=s.stats["rel_err"], x=x, res=res, y=y,
                              part=s.partition(r), ncoll=s.get_option("collectives_enqueued"), calls=calls,
                              eff=s.get_option("exchange_effective"), fallbacks=s.get_option("direct_fallbacks"),
                              sym=s.get_option("symmetric_effective"), t_exchange=s.stats["t_exchange"], t_gemv=s.stats["t_gemv"],
                              on_dev=s.get_option("ranks_on_device"), fused=s.get_option("fuse_effective"))
        except Exception as e:                   # noqa: BLE001
            errs.append(f"rank {r}: {e!r}")

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    if errs or any(o is None for o in out):
        print(json.dumps({"error": errs or "a rank did not finish",
                          "ncoll": [o.get("ncoll") if o else None for o in out]}))
        sys.exit(1)
    if a.quick_destroy > 0:
        print(json.dumps({"ok": True, "iters": [o["iters"] for o in out], "rel_err": [o["err"] for o in out],
                          "exchange_effective": [o["eff"] for o in out]}))
        return
    res = {
        "P": P, "n": n, "iters": out[0]["iters"], "converged": bool(out[0]["conv"]), "true_residual": out[0]["res"],
        "rel_err": out[0]["err"], "partition": [list(o["part"]) for o in out],
        "collectives_enqueued": [o["ncoll"] for o in out], "iterate_calls": [o["calls"] for o in out],
        "exchange_effective": [o["eff"] for o in out], "direct_fallbacks": [o["fallbacks"] for o in out],
        "symmetric_effective": [o["sym"] for o in out], "x_sha": hashlib.sha256(out[0]["x"].tobytes()).hexdigest(),
        "t_exchange": [o["t_exchange"] for o in out], "t_gemv": [o["t_gemv"] for o in out],
        "ranks_on_device": [o["on_dev"] for o in out], "fuse_effective": [o["fused"] for o in out],
        "ranks_identical": bool(all(np.array_equal(out[0]["x"], o["x"]) and o["iters"] == out[0]["iters"]
                                    and o["err"] == out[0]["err"] and np.array_equal(out[0]["y"], o["y"]) for o in out)),
    }
    if a.save_x:
        np.save(a.save_x, out[0]["x"])
    if not a.no_single:
        # the same system on one shard, no RCCL
        with lam.Solver(DT) as s:
            setup(s)
            s.solve(iters, tol)
            x1, it1 = s.solution(), s.stats["num_iters"]
            y1 = s.gemv(xprobe)
            if n <= 5000:
                A_host, b_host = s.download_rows(0, n), s.rhs()       # what the device holds (bf16: the rounded matrix)
        res.update(iters_single=it1, x_vs_single=float(np.linalg.norm(out[0]["x"] - x1) / np.linalg.norm(x1)),
                   gemv_vs_single=float(np.max(np.abs(out[0]["y"] - y1)) / np.max(np.abs(y1))))
        if n <= 5000:
            # ... and the reference algorithm itself on the same system, with the same number of ranks
            from oracle import pyoracle
            x_or, st_or = pyoracle.cg_solve(A_host.astype(vdt), b_host.astype(vdt), iters, tol, P=P)
            A64, b64, x64 = A_host.astype(np.float64), b_host.astype(np.float64), out[0]["x"].astype(np.float64)
            res.update(iters_oracle=st_or["num_iters"], converged_oracle=bool(st_or["converged"]), rel_err_oracle=st_or["rel_err"],
                       x_vs_oracle=float(np.linalg.norm(x64 - x_or) / np.linalg.norm(x_or)),
                       residual_numpy=float(np.linalg.norm(b64 - A64 @ x64) / np.linalg.norm(b64)),
                       residual_numpy_oracle=float(np.linalg.norm(b64 - A64 @ x_or.astype(np.float64)) / np.linalg.norm(b64)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
