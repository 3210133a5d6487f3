import importlib, os, sys
sys.path.insert(0, os.getcwd())
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
n = 65536
with lam.Solver(lam.F64) as s:
    s.generate_random_spd(n, 1, 1e4); s.generate_random_rhs(2); s.cg_init()
    s.set_option("probe_rows", 8192)
    for lo, hi in ((0, 0), (8192, 16384), (0, 8192), (57344, 65536)):
        s.set_option("panel_lo", lo); s.set_option("panel_hi", hi)
        ts = sorted(s.gemv_only(50) for _ in range(7))
        print(f"shard 8192x65536 panel [{lo},{hi}): median {ts[3]*1e3:.4f} ms  {8*8192*n/ts[3]/1e9:.0f} GB/s")
