#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of tools/rank_chain.py -> profiles/<tag>_rank_mode_chain.csv: for each variant the
kernel / collective sequence of ONE steady-state iteration (the median-length one) with every kernel's start
offset, duration and the gap to its predecessor.  Iterations are delimited by the GEMV launches; variants by
the order rank_chain.py runs them (a pause separates them: cg_init + host sync).

    python tools/summarize_chain.py <tag> <trace_dir>"""
import csv, glob, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for k in ("gemv_coop_kernel", "gemv_tile_kernel", "gemv_generic_kernel", "update_xr_full_kernel", "update_p_full_kernel",
              "update_xr_kernel", "update_p_kernel", "finalize_sum_kernel", "cg_init", "ncclDevKernel", "rccl", "nccl"):
        if k in name:
            return name[:90] if k in ("ncclDevKernel", "rccl", "nccl") else k
    return name[:60]


def main():
    tag, d = sys.argv[1], sys.argv[2]
    f = max(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    # split into iterations at every GEMV launch
    its, cur = [], []
    for st, en, nm in rows:
        if "gemv_" in nm and cur and any("gemv_" in x[2] for x in cur):
            its.append(cur); cur = []
        cur.append((st, en, nm))
    if cur:
        its.append(cur)
    # group consecutive iterations with the same kernel-name signature (= one variant's steady state)
    groups = []
    for it in its:
        sig = tuple(short(x[2]) for x in it)
        if groups and groups[-1][0] == sig:
            groups[-1][1].append(it)
        else:
            groups.append((sig, [it]))
    out = os.path.join(ROOT, "profiles", f"{tag}_rank_mode_chain.csv")
    with open(out, "w") as w:
        w.write("variant,iterations_seen,iter_us_median,step,kernel,start_offset_us,duration_us,gap_before_us\n")
        v = 0
        for sig, lst in groups:
            if len(lst) < 20:
                continue
            v += 1
            spans = sorted(lst, key=lambda it: it[-1][1] - it[0][0])
            it = spans[len(spans) // 2]
            # iteration length = start of this GEMV to start of the next one: approximate by median of consecutive GEMV starts
            starts = [x[0][0] for x in lst]
            deltas = sorted(b - a for a, b in zip(starts, starts[1:]))
            med = deltas[len(deltas) // 2] / 1e3 if deltas else 0.0
            t0, prev_end = it[0][0], None
            for i, (st, en, nm) in enumerate(it):
                gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:.2f}"
                w.write(f"{v},{len(lst)},{med:.2f},{i},\"{short(nm)}\",{(st - t0) / 1e3:.2f},{(en - st) / 1e3:.2f},{gap}\n")
                prev_end = en
    print(open(out).read())


if __name__ == "__main__":
    main()
