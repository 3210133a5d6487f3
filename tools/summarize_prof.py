#!/usr/bin/env python3
"""Turn rocprofv3 output directories (under gpurun_out/) into the small summaries kept in profiles/.

    python tools/summarize_prof.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>] [--n N --p P]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats table),
profiles/<tag>_gemv_by_grid.csv (GEMV launches grouped by grid size, from the kernel trace) and, when
PMC directories are given, profiles/<tag>_hbm_counters.csv plus an entry in profiles/traffic.json.
HBM bytes follow MI355X_MICROARCH.md "HBM": bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16 B/lane coalesced streaming read."""
import argparse, collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def one(pattern):
    f = glob.glob(pattern, recursive=True)
    if not f:
        sys.exit(f"no file matches {pattern}")
    return max(f, key=os.path.getmtime)      # gpurun_out/ accumulates runs: take the newest

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag"); ap.add_argument("stats_dir")
    ap.add_argument("fetch_dir", nargs="?"); ap.add_argument("write_dir", nargs="?")
    ap.add_argument("--n", type=int, default=65536); ap.add_argument("--p", type=int, default=1)
    ap.add_argument("--kernel", default="gemv_coop_kernel")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles"); os.makedirs(out, exist_ok=True)
    shutil.copy(one(os.path.join(a.stats_dir, "**", "*_kernel_stats.csv")), os.path.join(out, f"{a.tag}_kernel_stats.csv"))
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(one(os.path.join(a.stats_dir, "**", "*_kernel_trace.csv")))):
        if a.kernel in r["Kernel_Name"]:
            groups[(r["Kernel_Name"], r["Grid_Size_X"], r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, f"{a.tag}_gemv_by_grid.csv"), "w") as f:
        f.write("kernel,grid_threads,vgpr,lds_bytes,calls,avg_ns,min_ns,max_ns\n")
        for (k, g, v, l), d in sorted(groups.items()):
            f.write(f"\"{k}\",{g},{v},{l},{len(d)},{sum(d)/len(d):.1f},{min(d)},{max(d)}\n")
    if a.fetch_dir and a.write_dir:
        ctr = collections.defaultdict(list)
        for d in (a.fetch_dir, a.write_dir):
            for r in csv.DictReader(open(one(os.path.join(d, "**", "*_counter_collection.csv")))):
                if a.kernel in r["Kernel_Name"]:
                    ctr[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        with open(os.path.join(out, f"{a.tag}_hbm_counters.csv"), "w") as f:
            f.write("kernel,grid_threads,counter,launches,avg_value_KiB,min,max\n")
            for (g, c), v in sorted(ctr.items()):
                f.write(f"{a.kernel},{g},{c},{len(v)},{sum(v)/len(v):.3f},{min(v):.3f},{max(v):.3f}\n")
        grid = str(max(int(g) for g, _ in ctr))          # the main workload has the largest grid
        fetch = sum(ctr[(grid, "FETCH_SIZE")]) / len(ctr[(grid, "FETCH_SIZE")])
        write = sum(ctr[(grid, "WRITE_SIZE")]) / len(ctr[(grid, "WRITE_SIZE")])
        tpath = os.path.join(out, "traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        try:
            commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            commit = None
        full_name = next((r["Kernel_Name"] for r in csv.DictReader(open(one(os.path.join(a.fetch_dir, "**", "*_counter_collection.csv"))))
                          if a.kernel in r["Kernel_Name"] and r["Grid_Size"] == grid), a.kernel)
        tj[f"n{a.n}_p{a.p}"] = {"hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "FETCH_SIZE_KiB": fetch,
                               "commit": commit, "kernel": full_name,
                               "WRITE_SIZE_KiB": write, "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
                               "source": f"profiles/{a.tag}_hbm_counters.csv", "grid_threads": int(grid)}
        json.dump(tj, open(tpath, "w"), indent=1)
        print(json.dumps(tj[f"n{a.n}_p{a.p}"]))

if __name__ == "__main__":
    main()
