"""tools/sweep.py without a GPU: the grid, the comparison with the reference's published lines
(tests/golden/reference_gen_grid.json) and the verdict logic, driven by a stand-in driver that prints what the
closed form 1/(k*sqrt(8N)) gives -- and by one that prints a wrong digit."""
import json
import os
import stat
import subprocess
import sys

from conftest import GOLDEN, ROOT

SWEEP = os.path.join(ROOT, "tools", "sweep.py")

FAKE = r'''#!/usr/bin/env python3
import math, sys
a = sys.argv[1:]
n = int(a[a.index("-s") + 1]); k = int(a[a.index("-i") + 1]) if "-i" in a else 10000
prec = a[a.index("-t") + 1] if "-t" in a else "f64"
err = 1.0 / (k * math.sqrt(8.0 * n)) * (1.0 + SKEW)
print(f"{n},1,1,0.5,0,0.01,0.01,{k + 1},{err:.6g},0.2")
'''


def _fake(tmp_path, skew):
    exe = tmp_path / f"fake_{skew}.py"
    exe.write_text(FAKE.replace("SKEW", repr(skew)))
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    return str(exe)


def test_fixture_is_the_reference_grid():
    g = json.load(open(os.path.join(GOLDEN, "reference_gen_grid.json")))
    assert [e["n"] for e in g["entries"]] == [80000, 90000, 100000, 110000, 120000, 140000, 160000, 180000, 200000]
    assert all(e["iters_printed"] == 16 and e["max_iters"] == 15 and e["sources"][0].startswith("TESTS/BEST_RESULTS:") for e in g["entries"])
    assert g["entries"][0]["err_printed"] == "8.33333e-05" and g["entries"][7]["err_printed"] == "5.55555e-05"
    # + the `-s 80000 -i 1000` line of the GPU weak-scaling series
    assert g["entries_extra"] == [{"n": 80000, "max_iters": 1000, "iters_printed": 1001, "err_printed": "1.25e-06",
                                   "sources": ["TESTS/results/WEAK_SCALABILITY_GPU_MPI.txt:20"], "rank_counts": [64]}]
    f = json.load(open(os.path.join(GOLDEN, "reference_file_grid.json")))
    assert [(e["n"], e["iters_min"], e["iters_max"]) for e in f["entries"]] == [(10000, 358, 359), (20000, 359, 359), (30000, 360, 360),
                                                                               (40000, 360, 360), (50000, 360, 360)]


def test_sweep_accepts_the_closed_form_and_rejects_a_wrong_digit(tmp_path):
    js = tmp_path / "ok.json"
    r = subprocess.run([sys.executable, SWEEP, "--grid", "gen", "--exe", _fake(tmp_path, 0.0), "--json", str(js)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = json.load(open(js))
    assert len(recs) == 10 and all(x["match"] for x in recs)
    assert [x["precision"] for x in recs] == ["f64"] * 8 + ["f32", "f64"]   # 200000 x 200000 doubles do not fit one GPU
    assert recs[9]["n"] == 80000 and recs[9]["iters"] == 1001               # the -i 1000 line
    assert sum(x["same_printed_digits"] for x in recs) >= 7                 # 5.55555e-05 is a rounding borderline
    r = subprocess.run([sys.executable, SWEEP, "--grid", "gen", "--exe", _fake(tmp_path, 1e-4), "--json", str(js)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "MISMATCH" in r.stdout
    assert not any(x["match"] for x in json.load(open(js)))


FAKE_SCALING = r'''#!/usr/bin/env python3
import sys
a = sys.argv[1:]
n = int(a[a.index("-s") + 1]); P = int(a[a.index("-P") + 1]) if "-P" in a else 1
assert "-R" in a and "-g" in a
t_iter = 1e-3 * (n / 1000.0) ** 2 / P + 2e-5          # a made-up machine that scales
print(f"{n},{P},1,0.5,0,{t_iter * 0.9:.6g},{t_iter:.6g},{359 + WRONG},9.5e-10,{t_iter * 359:.6g}")
'''


def test_scaling_grids_have_the_reference_procs_axis(tmp_path):
    """--grid strong / weak without a GPU: the (N, P) points are the reference's (STRONG_SCALABILITY_GPU_MPI.txt: N = 20000 / 40000 /
    50000 at 1, 2, 3, 4, 6, 8 devices; WEAK: (10000, 1), (20000, 4), (40000, 8)), every line carries this run's speed-up over its own
    P = 1 line next to the reference's published one with its source line, the iteration count is held to the reference's, and a
    wrong count fails the sweep."""
    g = json.load(open(os.path.join(GOLDEN, "reference_scaling.json")))
    assert [(e["n"], e["procs"]) for e in g["strong"][:8]] == [(20000, p) for p in (1, 2, 3, 4, 6, 8, 12, 16)]
    assert g["strong"][2]["source"] == "TESTS/results/STRONG_SCALABILITY_GPU_MPI.txt:18" and g["strong"][2]["speedup_cg_vs_p1"] == 2.293
    assert [(e["n"], e["procs"], e["iters"]) for e in g["weak"][:3]] == [(10000, 1, 358), (20000, 4, 359), (40000, 8, 360)]

    def fake(wrong):
        exe = tmp_path / f"fake_scaling_{wrong}.py"
        exe.write_text(FAKE_SCALING.replace("WRONG", str(wrong)))
        exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
        return str(exe)

    js, csv = tmp_path / "s.json", tmp_path / "s.csv"
    r = subprocess.run([sys.executable, SWEEP, "--grid", "scaling", "--exe", fake(0), "--json", str(js), "--csv", str(csv)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    recs = json.load(open(js))
    strong, weak = [x for x in recs if x["grid"] == "strong"], [x for x in recs if x["grid"] == "weak"]
    assert [(x["n"], x["procs"]) for x in strong] == [(n, p) for n in (20000, 40000, 50000) for p in (1, 2, 3, 4, 6, 8)]
    assert [(x["n"], x["procs"]) for x in weak] == [(10000, 1), (20000, 4), (40000, 8)]
    assert all(x["match"] and x["topology"] == "one-process" for x in recs)
    s8 = next(x for x in strong if (x["n"], x["procs"]) == (40000, 8))
    assert 7.5 < s8["speedup_iter"] < 8.0 and s8["reference"]["speedup_cg_vs_p1"] == 5.32 and s8["reference"]["source"].endswith(":31")
    lines = open(csv).read().splitlines()
    assert lines[0].startswith("topology,N,procs,threads,") and lines[0].endswith("reference_speedup_iter,reference_speedup_cg,reference_source")
    assert len(lines) == 1 + 18 + 3 and lines[1].startswith("one-process,20000,1,1,")
    r = subprocess.run([sys.executable, SWEEP, "--grid", "weak", "--exe", fake(40), "--json", str(js)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "MISMATCH" in r.stdout
