#!/usr/bin/env python3
"""The reference's parameter grids (TESTS/GPU_SCRIPTS/*.sh, TESTS/CPU_SCRIPTS/*_gen.sh) run through this package's
drop-in driver, one process per point like the SLURM scripts do, emitting the reference's CSV columns
(test/test_CG_CPU_MPI_OMP.cpp:201-203 + the NCCL variant's comm-init column):

    N, procs, threads, load_or_gen_s, comm_init_s, avg_gemv_s, avg_iter_s, iters, rel_err, cg_total_s

  --grid gen   generate mode `-s N -i 15` for N = 80000 ... 200000 (TESTS/CPU_SCRIPTS/CPU_8_NODE_gen.sh:24-32).  The
               `iters, err` columns are compared with what the reference itself printed for the same (N, -i 15)
               (tests/golden/reference_gen_grid.json, extracted from TESTS/BEST_RESULTS:173-215 with line numbers).
               N = 200000 in fp64 is 320 GB and does not fit one 288 GB MI355X: it is run in fp32 storage (-t f32,
               160 GB) and marked so, never silently shrunk.
  --grid file  the file-mode sizes of TESTS/GPU_SCRIPTS/GPU_1_NODE.sh:41-47 (N = 10000 ... 70000, tol 1e-9, at most
               10000 iterations).  The reference's matrix files were never published (io/ is git-ignored), but they came
               from its generator, and what its runs printed for them -- 358-360 iterations at EVERY size
               (tests/golden/reference_file_grid.json <- TESTS/BEST_RESULTS:93-135) -- is a property of the generator's matrix
               law (spectrum exp(3.5 U[-1,1]), random rhs), which this package's generator reproduces (round 4): the systems
               are drawn from that law on the device (`-s N -R seed`), `--files DIR` writes them with
               apps/random_spd_system.out first (reference generator CLI) and runs real file mode (-A/-b) for sizes up to
               --files-max-n, and the iteration count must land within 3 % of the reference's.

  --grid strong  the reference's STRONG-SCALING series (TESTS/results/STRONG_SCALABILITY_GPU_MPI.txt:15-43 <- TESTS/GPU_SCRIPTS/
               GPU_2_NODE.sh:17-40): N = 20000 / 40000 / 50000 (systems from the generator's law, `-s N -R 42`) at P = 1, 2, 3, 4, 6, 8 --
               P = 3 and 6 do not divide these sizes: the reference's uneven partition.  Through the ONE-PROCESS topology (`-P shards`:
               one process drives P row shards, dealt round-robin over the visible GPUs -- several per GPU when there are fewer) and/or
               through the RANK MODE (`--launcher mpiexec`: `mpiexec -n P` on the MPI-bootstrapped driver, one process per GPU on RCCL).
               Output: the reference's CSV columns + `speedup_iter, speedup_cg` over the series' own P = 1 line + the reference's
               published speed-ups for the same (N, P) (tests/golden/reference_scaling.json).  Checked: the iteration count (the
               reference: 359 / 360 / 360) and the tolerance; speed-ups are REPORTED, not gated (they belong to the node).
  --grid weak    its WEAK-SCALING series (WEAK_SCALABILITY_GPU_MPI.txt:15-17): (N, P) = (10000, 1), (20000, 4), (40000, 8).
               --scale 0.1 runs either grid at a tenth of the sizes (tests on one GPU).

    usage: sweep.py [--grid gen|file|strong|weak|scaling|all] [--csv out.csv] [--exe path] [--files DIR] [--launcher one-process|mpiexec|both]
    the full scaling command for an 8-GPU node:  python tools/sweep.py --grid scaling --launcher both --csv scaling.csv
Exit code 0 iff every checked point matches.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd")
EXE = os.path.join(PKG, "test", "test_CG_MultiGPUS_HIP_RCCL.out")
GEN = os.path.join(PKG, "apps", "random_spd_system.out")
GOLD = os.path.join(ROOT, "tests", "golden", "reference_gen_grid.json")
GOLD_FILE = os.path.join(ROOT, "tests", "golden", "reference_file_grid.json")
GOLD_SCALING = os.path.join(ROOT, "tests", "golden", "reference_scaling.json")
MPI_EXE = os.path.join(PKG, "test", "test_CG_MultiGPUS_HIP_RCCL_mpi.out")
MPIEXEC = "/opt/conda/bin/mpiexec"
FILE_GRID = (10000, 20000, 30000, 40000, 50000, 60000, 70000)
STRONG_SIZES, STRONG_PROCS = (20000, 40000, 50000), (1, 2, 3, 4, 6, 8)
WEAK_POINTS = ((10000, 1), (20000, 4), (40000, 8))
SCALING_COLUMNS = "topology," + "N,procs,threads,load_or_gen_s,comm_init_s,avg_gemv_s,avg_iter_s,iters,rel_err,cg_total_s" + \
                  ",speedup_iter,speedup_cg,reference_speedup_iter,reference_speedup_cg,reference_source"
HBM_BYTES = 288e9
COLUMNS = "N,procs,threads,load_or_gen_s,comm_init_s,avg_gemv_s,avg_iter_s,iters,rel_err,cg_total_s"


def run_point(exe, args, timeout=900):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=timeout)
    line = r.stdout.replace("\n", "").strip()
    return r.returncode, line, r.stderr, time.time() - t0


def gen_grid(exe, sol, out, sizes=None, extra=True):
    gold = json.load(open(GOLD))
    ok = True
    for e in gold["entries"] + (gold.get("entries_extra", []) if extra else []):      # extra: `-s 80000 -i 1000` of the GPU weak-scaling series
        n = e["n"]
        if sizes and n not in sizes:
            continue
        fits = 8.0 * n * n + 64.0 * n < 0.97 * HBM_BYTES
        prec = "f64" if fits else "f32"
        rc, line, err, wall = run_point(exe, ["-s", str(n), "-i", str(e["max_iters"]), "-o", sol, "-t", prec])
        f = line.split(",")
        rec = {"grid": "gen", "n": n, "precision": prec, "csv": line, "reference_iters": e["iters_printed"],
               "reference_err": e["err_printed"], "reference_source": e["sources"][0], "wall_s": round(wall, 2)}
        if rc != 0 or len(f) != 10:
            rec["match"] = False
            rec["error"] = f"rc {rc}: {err.strip()[-300:]}"
        else:
            ours = float(f[8])
            rel = abs(ours / float(e["err_printed"]) - 1.0)
            # the reference prints 6 significant digits: agreement "to the printed digits" = within one unit of the last
            # one (fp32 storage: within 1e-5, the run is there because fp64 does not fit, and says so)
            tol = 2e-6 if prec == "f64" else 1e-5
            rec.update({"iters": int(f[7]), "err": f[8], "rel_diff": rel, "same_printed_digits": f[8] == e["err_printed"],
                        "match": int(f[7]) == e["iters_printed"] and rel <= tol})
            if not fits:
                rec["note"] = f"fp64 needs {8.0 * n * n / 1e9:.0f} GB > one MI355X: run with -t f32"
        ok &= rec["match"]
        out.append(rec)
        print(("ok   " if rec["match"] else "FAIL ") + f"gen  N={n:6d} {prec}: {line}   [reference: {e['iters_printed']}, {e['err_printed']} "
              f"({e['sources'][0]})]" + (f"  {rec.get('note', '')}" if not fits else ""), flush=True)
    return ok


def file_grid(exe, sol, out, files_dir, files_max_n, sizes):
    ok = True
    ref = {e["n"]: e for e in json.load(open(GOLD_FILE))["entries"]}
    ref_lo, ref_hi = min(e["iters_min"] for e in ref.values()), max(e["iters_max"] for e in ref.values())
    for n in sizes:
        args = ["-s", str(n), "-R", "42", "-o", sol]
        mode = "reference generator's law, built on the device (-s N -R 42)"
        if files_dir and n <= files_max_n:
            m, b = os.path.join(files_dir, f"matrix{n}.bin"), os.path.join(files_dir, f"rhs{n}.bin")
            if not (os.path.exists(m) and os.path.exists(b)):
                g = subprocess.run([GEN, str(n), m, b, "42"], capture_output=True, text=True, timeout=1800)
                if g.returncode != 0:
                    out.append({"grid": "file", "n": n, "match": False, "error": "generator failed: " + g.stderr[-300:]})
                    ok = False
                    continue
            args = ["-A", m, "-b", b, "-o", sol]
            mode = f"file mode (-A {os.path.basename(m)} -b {os.path.basename(b)})"
        rc, line, err, wall = run_point(exe, args)
        f = line.split(",")
        rec = {"grid": "file", "n": n, "mode": mode, "csv": line, "wall_s": round(wall, 2)}
        if "Option symmetric" in err:            # LAM_HIP_SYMMETRIC in the environment: the driver says whether the solve ran on it
            rec["symmetric"] = "effective" if ": effective" in err else "NOT effective"
        if rc != 0 or len(f) != 10:
            rec["match"] = False
            rec["error"] = f"rc {rc}: {err.strip()[-300:]}"
        else:
            # known answer: the reference's own count at this N where it published one, else the N-independent range of its runs
            lo, hi = (ref[n]["iters_min"], ref[n]["iters_max"]) if n in ref else (ref_lo, ref_hi)
            it = int(f[7])
            rec.update({"iters": it, "err": f[8], "reference_iters": [lo, hi], "reference_published_at_this_n": n in ref,
                        "match": float(f[8]) < 1e-9 and 0.97 * lo <= it <= 1.03 * hi})
        ok &= rec["match"]
        out.append(rec)
        print(("ok   " if rec["match"] else "FAIL ") + f"file N={n:6d}: {line}   [{mode}; reference: {rec.get('reference_iters')} iterations"
              + ("" if n in ref else " (its N-independent range; no published run at this N)") + "]", flush=True)
    return ok


def scaling_grid(exe, sol, out, which, launchers, scale, mpiexec, mpi_exe, preload, max_ranks):
    """--grid strong / weak: see the module docstring.  One driver process (or one mpiexec launch) per point."""
    gold = json.load(open(GOLD_SCALING))
    ref = {(e["n"], e["procs"]): e for e in gold[which]}
    file_ref = {e["n"]: e for e in json.load(open(GOLD_FILE))["entries"]}
    lo_all, hi_all = min(e["iters_min"] for e in file_ref.values()), max(e["iters_max"] for e in file_ref.values())
    points = [(n, p) for n in STRONG_SIZES for p in STRONG_PROCS] if which == "strong" else list(WEAK_POINTS)
    ok = True
    ndev = None
    for topo in launchers:
        base = {}          # (n) -> the P = 1 record of this topology
        for n_ref, procs in points:
            if topo == "mpiexec" and procs > max_ranks:
                continue                       # (a one-GPU test box admits only a few GPU processes at a time)
            n = max(procs, int(round(n_ref * scale)))
            args = ["-s", str(n), "-R", "42", "-o", sol, "-g"]          # -g: the GEMV column includes the exchange, like the reference's
            env = dict(os.environ)
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            if topo == "one-process":
                cmd = [exe, "-P", str(procs)] + args
            else:
                cmd = [mpiexec, "-n", str(procs)]
                if preload:     # tests: the RCCL double in front of librccl in the RANKS only (hydra's -genv), one hardware queue per stream
                    cmd += ["-genv", "LD_PRELOAD", preload, "-genv", "GPU_MAX_HW_QUEUES", str(2 * procs + 4), "-genv", "MOCK_RCCL_TIMEOUT_MS", "20000"]
                cmd += [mpi_exe] + args
            t0 = time.time()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1800)
                rc, line, err = r.returncode, r.stdout.replace("\n", "").strip(), r.stderr
            except Exception as e:   # noqa: BLE001
                rc, line, err = -1, "", str(e)
            f = line.split(",")
            e_ref = ref.get((n_ref, procs))
            rec = {"grid": which, "topology": topo, "n": n, "n_reference": n_ref, "procs": procs, "csv": line, "wall_s": round(time.time() - t0, 2),
                   "reference": e_ref}
            if rc != 0 or len(f) != 10:
                rec["match"] = False
                rec["error"] = f"rc {rc}: {err.strip()[-300:]}"
            else:
                it, t_iter, t_cg = int(f[7]), float(f[6]), float(f[9])
                lo, hi = (file_ref[n_ref]["iters_min"], file_ref[n_ref]["iters_max"]) if scale == 1.0 and n_ref in file_ref else (lo_all, hi_all)
                if procs == 1 or n_ref not in base:
                    base.setdefault(n_ref, (t_iter, t_cg))
                b_iter, b_cg = base[n_ref]
                rec.update({"iters": it, "err": f[8], "t_gemv_plus_comm": float(f[5]), "t_iter": t_iter, "t_cg": t_cg,
                            "speedup_iter": b_iter / t_iter if which == "strong" else None, "speedup_cg": b_cg / t_cg if which == "strong" else None,
                            # (at a tenth of the sizes the law's count sags a little: 339 at N = 1000, 352 at 2000, 357-358 from 4000 on)
                            "reference_iters": [lo, hi], "match": int(f[1]) == procs and float(f[8]) < 1e-9 and (0.97 if scale >= 1.0 else 0.93) * lo <= it <= 1.03 * hi})
            ok &= rec["match"]
            out.append(rec)
            rs = (f"{e_ref['speedup_iter_vs_p1']}x iter / " + (f"{e_ref['speedup_cg_vs_p1']}x" if e_ref.get('speedup_cg_vs_p1') is not None else "not published") +
                  f" cg ({e_ref['source']})") if e_ref and "speedup_iter_vs_p1" in e_ref else \
                 (f"t_iter {e_ref['t_iter']} s ({e_ref['source']})" if e_ref else "no published line")
            print(("ok   " if rec["match"] else "FAIL ") + f"{which:6s} {topo:11s} N={n:6d} P={procs}: {line}"
                  + (f"   speed-up {rec['speedup_iter']:.2f}x iter / {rec['speedup_cg']:.2f}x cg" if rec.get("speedup_iter") else "")
                  + f"   [reference: {rs}]" + (f"  {rec.get('error', '')}" if not rec["match"] else ""), flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", choices=("gen", "file", "strong", "weak", "scaling", "all"), default="all",
                    help="scaling = strong + weak; all = gen + file (the single-GPU grids)")
    ap.add_argument("--launcher", choices=("one-process", "mpiexec", "both"), default="one-process", help="scaling grids: which multi-GPU topology")
    ap.add_argument("--scale", type=float, default=1.0, help="scaling grids: multiply the sizes (0.1 for a quick run on one GPU)")
    ap.add_argument("--mpiexec", default=MPIEXEC)
    ap.add_argument("--mpi-exe", default=MPI_EXE)
    ap.add_argument("--max-ranks", type=int, default=16, help="scaling grids, mpiexec launcher: skip points with more ranks than this")
    ap.add_argument("--preload", help="LD_PRELOAD for the mpiexec launches (tests: the RCCL double, several ranks on one GPU)")
    ap.add_argument("--csv", help="write the CSV lines (reference column order) here")
    ap.add_argument("--json", help="write the per-point records here")
    ap.add_argument("--exe", default=EXE)
    ap.add_argument("--files", help="directory for real matrix/rhs files (file grid)")
    ap.add_argument("--files-max-n", type=int, default=30000)
    ap.add_argument("--file-sizes", default=",".join(str(x) for x in FILE_GRID))
    ap.add_argument("--gen-sizes", default="", help="gen grid: only these sizes (default: every published one)")
    ap.add_argument("--no-gen-extra", action="store_true", help="gen grid: skip the 1000-iteration point of the weak-scaling series")
    a = ap.parse_args()
    sol = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"lam_sweep_sol_{os.getpid()}.bin")
    out, ok = [], True
    print("# " + COLUMNS, flush=True)
    if a.grid in ("gen", "all"):
        ok &= gen_grid(a.exe, sol, out, [int(x) for x in a.gen_sizes.split(",") if x], not a.no_gen_extra)
    if a.grid in ("file", "all"):
        ok &= file_grid(a.exe, sol, out, a.files, a.files_max_n, [int(x) for x in a.file_sizes.split(",") if x])
    scaling = [g for g in ("strong", "weak") if a.grid in (g, "scaling")]
    if scaling:
        launchers = ["one-process", "mpiexec"] if a.launcher == "both" else [a.launcher]
        if "mpiexec" in launchers and not (os.path.exists(a.mpiexec) and os.path.exists(a.mpi_exe)):
            print(f"# mpiexec launcher not available ({a.mpiexec}, {a.mpi_exe}: `make -C {PKG} mpi`): skipped", flush=True)
            launchers.remove("mpiexec")
        for g in scaling:
            ok &= scaling_grid(a.exe, sol, out, g, launchers, a.scale, a.mpiexec, a.mpi_exe, a.preload, a.max_ranks)
    if os.path.exists(sol):
        os.remove(sol)
    if a.csv:
        with open(a.csv, "w") as f:
            f.write((SCALING_COLUMNS if scaling else COLUMNS) + "\n")
            for r in out:
                if not r.get("csv"):
                    continue
                if scaling:
                    e = r.get("reference") or {}
                    f.write(",".join([r["topology"], r["csv"], f"{r.get('speedup_iter') or ''}", f"{r.get('speedup_cg') or ''}", f"{e.get('speedup_iter_vs_p1') or ''}",
                                      f"{e.get('speedup_cg_vs_p1') or ''}", e.get("source", "")]) + "\n")
                else:
                    f.write(r["csv"] + "\n")
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
    print("# all points match" if ok else "# MISMATCH", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
