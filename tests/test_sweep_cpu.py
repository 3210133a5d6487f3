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
