import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(")+1:f.rindex(")")]
            rest = f[f.rindex(")")+2:].split()
            out[t] = (name, (int(rest[11]) + int(rest[12])) / os.sysconf("SC_CLK_TCK"))
        except Exception:
            pass
    return out
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
with lam.Solver(lam.F64) as s:
    s.generate_random_spd(n, 1, 1e6); s.generate_random_rhs(2)
    for timing in (8, 0):
        s.set_option("gemv_timing", timing)
        s.cg_init(); s.cg_iterate(10, 0.0)
        a = threads(); t0 = time.time()
        st = s.cg_iterate(150, 0.0)
        dt = time.time() - t0; b = threads()
        print(f"N={n} gemv_timing={timing}: wall {dt:.3f} s")
        for t, (name, cpu) in b.items():
            d = cpu - a.get(t, (name, 0.0))[1]
            if d > 0.005: print(f"   thread {t} ({name}): {d:.3f} s CPU ({100*d/dt:.0f} % of wall)")
