#!/usr/bin/env python3
"""What makes the same kernel slower later in a process's life?  (VERDICT r01, weak spot 3: the N=32768
GEMV measured ~2 % slower, the symmetric product 23 % slower, in a process that had already run the
N=65536 problem.)  Separates the candidates in ONE process:

  S0  fresh process, fresh hipMalloc                                    baseline
  S1  same allocation, after a 34 GB neighbour was allocated and exercised (still alive)   -> clock/power/thermal state
  S2  same allocation, after the neighbour was freed                    -> the free itself
  S3  NEW allocation made after the free (fresh hipMalloc out of the freed range)         -> allocation history
  S4  a context that shrinks 65536 -> 32768 re-using its matrix allocation ("reuse_matrix", the default)
  S5  the same with reuse_matrix = 0 (free + hipMalloc inside one context)

For each state: GEMV ms (median of 7 x 30 launches) and the symmetric product ms, plus the shader / memory
clock levels the driver reports in sysfs (read as plain files: no child process after the GPU is up)."""
import glob, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")


def clocks():
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))[:1] + sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk"))[:1]:
        try:
            cur = [l.split(":")[1].strip() for l in open(f).read().splitlines() if l.strip().endswith("*")]
            out.append(os.path.basename(f)[7:] + "=" + ("/".join(cur) or "?"))
        except Exception as e:   # noqa: BLE001
            out.append(f"{os.path.basename(f)}: {type(e).__name__}")
    return " ".join(out) or "clocks n/a"


def measure(s, label, n):
    gb = 8.0 * n * n / 1e9
    s.set_option("symmetric", 0)
    ts = sorted(s.gemv_only(30) for _ in range(7))
    c1 = clocks()
    s.set_option("symmetric", 1)
    tsym = sorted(s.gemv_only(30) for _ in range(7)) if s.get_option("symmetric_effective") else [float("nan")] * 7
    s.set_option("symmetric", 0)
    print(f"{label:58s} N={n} gemv {ts[3]*1e3:7.4f} ms ({gb/ts[3]:6.0f} GB/s, min {ts[0]*1e3:.4f})  symv {tsym[3]*1e3:7.4f} ms  [{c1}]", flush=True)
    return ts[3], tsym[3]


def main():
    n, big = 32768, 65536
    b = lam.Solver(lam.F64)
    b.generate_random_spd(n, 1, 1e4); b.generate_random_rhs(2); b.cg_init()
    r0 = measure(b, "S0 fresh process, fresh allocation", n)
    r0b = measure(b, "S0 again (repeatability)", n)
    a = lam.Solver(lam.F64)
    a.generate_random_spd(big, 3, 1e4); a.generate_random_rhs(4); a.cg_init()
    measure(a, "   (neighbour: N=65536, 34 GB, exercised)", big)
    r1 = measure(b, "S1 same allocation, 34 GB neighbour alive + exercised", n)
    a.close()
    r2 = measure(b, "S2 same allocation, neighbour freed", n)
    c = lam.Solver(lam.F64)
    c.generate_random_spd(n, 1, 1e4); c.generate_random_rhs(2); c.cg_init()
    r3 = measure(c, "S3 NEW allocation after the 34 GB free", n)
    r2b = measure(b, "S2 again (first allocation, for drift)", n)
    c.close(); b.close()
    d = lam.Solver(lam.F64)
    d.generate_random_spd(big, 3, 1e4); d.generate_random_rhs(4); d.cg_init(); d.cg_iterate(20)
    d.generate_random_spd(n, 1, 1e4); d.generate_random_rhs(2); d.cg_init()
    r4 = measure(d, "S4 one context 65536 -> 32768, matrix allocation re-used", n)
    d.set_option("reuse_matrix", 0)
    d.generate_random_spd(big, 3, 1e4); d.generate_random_rhs(4); d.cg_init(); d.cg_iterate(20)
    d.generate_random_spd(n, 1, 1e4); d.generate_random_rhs(2); d.cg_init()
    r5 = measure(d, "S5 one context 65536 -> 32768, free + hipMalloc", n)
    d.close()
    # S6: how long does the slow period after a large free last?  One context (32768), a 34 GB neighbour is
    # created and destroyed, then the GEMV is sampled every ~60 ms.
    e = lam.Solver(lam.F64)
    e.generate_random_spd(n, 1, 1e4); e.generate_random_rhs(2); e.cg_init()
    base = sorted(e.gemv_only(20) for _ in range(5))[2]
    f = lam.Solver(lam.F64)
    f.generate_random_spd(big, 3, 1e4)
    f.close()
    t0 = time.perf_counter()
    series = []
    while time.perf_counter() - t0 < 4.0:
        t = time.perf_counter() - t0
        series.append((t, e.gemv_only(20)))
        time.sleep(0.03)
    print(f"S6 GEMV N={n} before the free: {base*1e3:.4f} ms; after hipFree of 34 GB (t = seconds since the free returned):")
    print("   " + "  ".join(f"{t:.2f}s:{v/base:.3f}" for t, v in series[::2]))
    e.close()
    print("relative to S0 (gemv, symv): " + "  ".join(f"{k} {v[0]/r0[0]:.3f}/{v[1]/r0[1]:.3f}" for k, v in
          (("S0'", r0b), ("S1", r1), ("S2", r2), ("S3", r3), ("S2'", r2b), ("S4", r4), ("S5", r5))))


if __name__ == "__main__":
    main()
