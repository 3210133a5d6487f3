#!/usr/bin/env python3
"""Short view of a bench.py line (a file holding the JSON line): the headline, the roofline, and whichever of the side records the
line carries -- the symmetric option and configs[3] (1 GPU), the exchange modes and the other topology (N > 1).
    usage: show_bench.py bench.json"""
import json
import sys


def short(m):
    keys = ("value", "ms_per_step", "gemv_ms", "exchange_us", "exchange_us_min_over_ranks", "other_us", "vs_one_gpu", "error")
    return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items() if k in keys}


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = d.get("roofline") or {}
    print(f"{d['metric']} = {d['value']} on {d['n_gpus']} GPU(s): {d['config'].get('parallelism')}")
    print(f"  ms_per_step {d['ms_per_step']:.4f}  gemv_ms {d['gemv_ms']:.4f}  roofline.frac {rf.get('frac')}  traffic {rf.get('traffic')}")
    for k in ("headline_from", "own_topology_error", "comparison_error", "error"):
        if d.get(k):
            print(f"  {k}: {d[k]}")
    if "symmetric_option" in d:
        print("  symmetric_option:", {k: v for k, v in d["symmetric_option"].items() if k != "what"})
    for r in d.get("config4_gemv") or []:
        print("  config4:", r["dtype"], r["path"][:40], round(r["gemv_ms"], 3), round(r["roofline_frac"], 4))
    for a in d.get("also") or []:
        print("  also:", a["n"], round(a["value"], 1), round(a["roofline_frac"], 4))
    modes = d.get("exchange_modes") or {}
    for k, m in modes.items():
        print(f"  mode {k}:", m if isinstance(m, str) else short(m))
    for key in ("rank_mode_rccl", "one_process_topology"):
        if key in d:
            o = d[key]
            print(f"  {key}:", short(o) if "value" in o else str(o)[:300])
            for k, m in (o.get("exchange_modes") or {}).items():
                print(f"    mode {k}:", m if isinstance(m, str) else short(m))


if __name__ == "__main__":
    main()
