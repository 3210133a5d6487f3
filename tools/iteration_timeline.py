#!/usr/bin/env python3
"""Where the time of ONE single-GPU CG iteration goes outside the GEMV: the two launches of an iteration (GEMV with
its reducer workgroup, fused vector step) with durations and the gaps between them, from a rocprofv3 kernel trace,
next to the wall time per iteration of an un-profiled run of the same configurations.

    python tools/iteration_timeline.py            parent: un-profiled table, then itself under rocprofv3 --kernel-trace
    (child: --child runs the configurations)
Prints the table kept as profiles/r03_iteration_timeline.txt."""
import csv
import glob
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (10000, 20000, 32768)
ITERS = 120


def run(verbose):
    sys.path.insert(0, ROOT)
    lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
    with lam.Solver(lam.F64) as s:
        for n in SIZES:
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            for fuse in (1, 0):
                s.set_option("fuse_update", fuse)
                s.set_option("gemv_timing", 0)
                s.cg_init()
                s.cg_iterate(30, 0.0)
                st = s.cg_iterate(ITERS, 0.0)
                if verbose:
                    print(f"N={n} fuse_update={fuse}: {st['t_iter'] * 1e6:8.2f} us per iteration (wall, no events in the loop)", flush=True)


def main():
    if "--child" in sys.argv:
        return run(False)
    print("# un-profiled: wall time per iteration")
    run(True)
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    work = tempfile.mkdtemp(prefix="lam_tl_")
    try:
        r = subprocess.run([prof, "--kernel-trace", "--output-format", "csv", "-d", work, "--", sys.executable, os.path.abspath(__file__), "--child"],
                           capture_output=True, text=True, timeout=600, cwd=work, env=dict(os.environ, TMPDIR=work))
        f = glob.glob(os.path.join(work, "**", "*_kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not f:
            sys.exit("rocprofv3 failed: " + r.stderr[-400:])
        rows = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"], int(x["Grid_Size_X"])) for x in csv.DictReader(open(f[0])))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    # iterations = runs of [gemv, update...] delimited by GEMV launches; group by (gemv grid, kernel-name signature)
    its, cur = [], []
    for st, en, nm, g in rows:
        if "gemv_" in nm:
            if cur:
                its.append(cur)
            cur = [(st, en, nm, g)]
        elif cur and ("update_" in nm):
            cur.append((st, en, nm, g))
        else:
            if cur:
                its.append(cur)
            cur = []
    groups = {}
    for a, b in zip(its, its[1:]):
        if b[0][3] != a[0][3]:
            continue
        sig = (a[0][3], tuple(x[2].split("<")[0].replace("void lam::", "") for x in a))
        groups.setdefault(sig, []).append((a, b[0][0]))
    print("# under rocprofv3 --kernel-trace (gaps are inflated by the tracer; durations are not): median over the steady-state iterations")
    print("# N      launches of one iteration: kernel duration_us [gap_to_next_us] ...                                   sum outside GEMV")
    for (grid, names), lst in sorted(groups.items()):
        if len(lst) < 40:
            continue
        n = {(x // 2 + 1) * 256: x for x in SIZES}.get(grid) or next((x for x in SIZES if abs(grid - (x // 2 + 1) * 256) <= 512), grid)
        med = lambda v: sorted(v)[len(v) // 2]
        parts, outside = [], 0.0
        for i, nm in enumerate(names):
            dur = med([(it[i][1] - it[i][0]) / 1e3 for it, _ in lst])
            nxt = med([((it[i + 1][0] if i + 1 < len(it) else nx) - it[i][1]) / 1e3 for it, nx in lst])
            parts.append(f"{nm} {dur:.1f} [{nxt:.1f}]")
            outside += (0.0 if i == 0 else dur) + nxt
        print(f"N={n:6d}  " + "  ".join(parts) + f"   -> {outside:.1f} us")


if __name__ == "__main__":
    main()
