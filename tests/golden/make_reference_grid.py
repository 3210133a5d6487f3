#!/usr/bin/env python3
"""Extracts the reference's own published generate-mode results (`-s N -i 15`, the parameter grid of
TESTS/CPU_SCRIPTS/CPU_*_NODE_gen.sh:24-32) from /root/reference/TESTS/BEST_RESULTS into a small data fixture:
tests/golden/reference_gen_grid.json.  Each entry keeps the printed `iters, err` columns as the reference printed
them and the line number it came from.  Run in the build container (the reference does not travel)."""
import json
import os
import re

SRC = "/root/reference/TESTS/BEST_RESULTS"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_gen_grid.json")


def main():
    rows, section = [], None
    for ln, line in enumerate(open(SRC), 1):
        line = line.strip()
        if line.startswith("---") and "gen" in line:
            section = line.strip("-")
        m = re.fullmatch(r"(\d+),(\d+),(\d+),([\d.e+-]+),([\d.e+-]+),([\d.e+-]+),(\d+),([\d.e+-]+),([\d.e+-]+)", line)
        if not m or section is None or "CPU_MPI_OMP gen" not in section:
            continue
        n, p, thr, _, _, _, iters, err, _ = m.groups()
        if int(n) >= 80000 and iters == "16":
            rows.append({"n": int(n), "mpi_ranks": int(p), "threads": int(thr), "max_iters": 15, "iters_printed": int(iters),
                         "err_printed": err, "source": f"TESTS/BEST_RESULTS:{ln}", "section": section})
    # one known answer per N (they are identical across rank counts, which the fixture keeps as evidence)
    by_n = {}
    for r in rows:
        by_n.setdefault(r["n"], []).append(r)
    for n, rs in by_n.items():
        assert len({r["err_printed"] for r in rs}) == 1, (n, rs)
    out = {"what": "generate mode (-s N -i 15): A = tridiag(1,2,1), b = 1; printed iters and err of test_CPU_MPI_OMP.out on MeluXina",
           "grid_script": "TESTS/CPU_SCRIPTS/CPU_8_NODE_gen.sh:24-32",
           "entries": [{"n": n, "max_iters": 15, "iters_printed": rs[0]["iters_printed"], "err_printed": rs[0]["err_printed"],
                        "sources": [r["source"] for r in rs], "rank_counts": sorted({r["mpi_ranks"] for r in rs})}
                       for n, rs in sorted(by_n.items())]}
    # one more generate-mode known answer, from the GPU weak-scaling series: `-s 80000 -i 1000` on 64 GPUs printed 1001, 1.25e-06
    # (= 1 / (1000 sqrt(8 * 80000)): the closed form of tridiag(1,2,1), b = 1)
    extra = []
    for ln, line in enumerate(open("/root/reference/TESTS/results/WEAK_SCALABILITY_GPU_MPI.txt"), 1):
        m = re.fullmatch(r"(\d+),(\d+),(\d+),([\d.e+-]+),([\d.e+-]+),([\d.e+-]+),(\d+),([\d.e+-]+),([\d.e+-]+)", line.strip())
        if m and m.group(7) == "1001":
            extra.append({"n": int(m.group(1)), "max_iters": 1000, "iters_printed": 1001, "err_printed": m.group(8),
                          "sources": [f"TESTS/results/WEAK_SCALABILITY_GPU_MPI.txt:{ln}"], "rank_counts": [int(m.group(2))]})
    assert len(extra) == 1, extra
    out["entries_extra"] = extra
    json.dump(out, open(OUT, "w"), indent=1)
    print(f"{len(out['entries'])} sizes (+{len(extra)} extra) -> {OUT}")
    file_grid()


def file_grid():
    """File mode on the matrices of the reference's generator (matrix{N}.bin, never published): iteration counts and residuals
    the reference's CPU MPI+OMP path printed for N = 10000 ... 50000, tol 1e-9 (TESTS/BEST_RESULTS, the two CPU_MPI_OMP
    file-mode sections).  The count is a property of the generator's matrix LAW (spectrum exp(3.5 U[-1,1]), random rhs), not
    of N: 358-360 everywhere -- a known answer for systems drawn from that law (apps/random_spd_system.out, driver option -R)."""
    rows, section = [], None
    for ln, line in enumerate(open(SRC), 1):
        line = line.strip()
        if line.startswith("---") and line.strip("-"):
            section = line.strip("-")
        m = re.fullmatch(r"(\d+),(\d+),(\d+),([\d.e+-]+),([\d.e+-]+),([\d.e+-]+),(\d+),([\d.e+-]+),([\d.e+-]+)", line)
        if not m or section is None or not section.startswith("CPU_MPI_OMP") or "gen" in section:
            continue
        n, p, thr, _, _, _, iters, err, _ = m.groups()
        if int(n) <= 70000 and int(iters) < 10001:
            rows.append({"n": int(n), "mpi_ranks": int(p), "iters_printed": int(iters), "err_printed": err, "source": f"TESTS/BEST_RESULTS:{ln}"})
    by_n = {}
    for r in rows:
        by_n.setdefault(r["n"], []).append(r)
    out = {"what": "file mode, tol 1e-9, matrices of the reference generator (random_spd_system.cpp: A = Q diag(exp(3.5 U[-1,1])) Q^T, rhs "
                   "U[-1,1]): printed iters and err of test_CPU_MPI_OMP.out on MeluXina, all rank counts",
           "grid_script": "TESTS/CPU_SCRIPTS/CPU_1_NODE.sh:23-27",
           "entries": [{"n": n, "iters_min": min(r["iters_printed"] for r in rs), "iters_max": max(r["iters_printed"] for r in rs),
                        "err_printed": sorted({r["err_printed"] for r in rs}), "sources": [r["source"] for r in rs]}
                       for n, rs in sorted(by_n.items())]}
    path = os.path.join(os.path.dirname(OUT), "reference_file_grid.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"{len(out['entries'])} file-mode sizes -> {path}: " + ", ".join(f"{e['n']}: {e['iters_min']}-{e['iters_max']}" for e in out["entries"]))


def scaling_tables():
    """The reference's published GPU scaling series (TESTS/results/STRONG_SCALABILITY_GPU_MPI.txt:15-43: N = 20000 / 40000 / 50000 at
    1 ... 16 GPUs, file mode on its generator's matrices; WEAK_SCALABILITY_GPU_MPI.txt:15-20) as data: every CSV line with the line
    number it came from, plus the speed-up of the whole CG (column 9, and of the iteration, column 6) over the series' own P = 1 line.
    tools/sweep.py --grid strong | weak prints this package's numbers next to them."""
    pat = re.compile(r"(\d+),(\d+),(\d+),([\d.e+-]+),([\d.e+-]+),([\d.e+-]+),(\d+),([\d.e+na-]+),([\d.e+-]+)")
    out = {"what": "GPU runs of the reference (test_CG_MultiGPUS_CUDA_MPI.out, A100 40 GB, 4 GPUs per node, MeluXina), file mode on its generator's "
                   "matrices, tol 1e-9: CSV columns N,procs,threads,t_load,t_gemv(incl. bcast+gather),t_iter,iters,err,t_cg",
           "grid_script": "TESTS/GPU_SCRIPTS/GPU_2_NODE.sh:17-40", "strong": [], "weak": []}
    for key, path in (("strong", "/root/reference/TESTS/results/STRONG_SCALABILITY_GPU_MPI.txt"), ("weak", "/root/reference/TESTS/results/WEAK_SCALABILITY_GPU_MPI.txt")):
        for ln, line in enumerate(open(path), 1):
            m = pat.match(line.strip())
            if not m:
                continue
            n, p_, _, t_load, t_gemv, t_iter, iters, err, t_cg = m.groups()
            out[key].append({"n": int(n), "procs": int(p_), "t_gemv": float(t_gemv), "t_iter": float(t_iter), "iters": int(iters), "err": err,
                             "t_cg": float(t_cg), "converged": int(iters) < 10001, "source": f"TESTS/results/{os.path.basename(path)}:{ln}"})
    for n in sorted({e["n"] for e in out["strong"]}):
        series = [e for e in out["strong"] if e["n"] == n]
        base = next((e for e in series if e["procs"] == 1), None)
        for e in series:
            # the N = 50000 series has no converged P = 1 run (it printed -nan): its per-iteration time is still a valid base
            e["speedup_iter_vs_p1"] = round(base["t_iter"] / e["t_iter"], 3) if base else None
            e["speedup_cg_vs_p1"] = round(base["t_cg"] / e["t_cg"], 3) if base and base["converged"] and e["converged"] else None
    path = os.path.join(os.path.dirname(OUT), "reference_scaling.json")
    json.dump(out, open(path, "w"), indent=1)
    print(f"{len(out['strong'])} strong + {len(out['weak'])} weak scaling lines -> {path}")


if __name__ == "__main__":
    main()
    scaling_tables()
