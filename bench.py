#!/usr/bin/env python3
"""bench.py -- CG iterations/s and GEMV GB/s of the MI355X-native dense CG hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE Conjugate-Gradient iteration (row-sharded GEMV Ap=A.p + the two dot products +
the x, r, p updates + the per-iteration exchange) on a synthetic dense SPD system that is already
resident in HBM when the timed region starts.  Workload for every GPU count: BASELINE.json
configs[2], N=65536 fp64 (34.4 GB, fits one MI355X) -- strong scaling, so the driver's 1/2/4/8
series is one problem.  The matrix is the device-generated random dense SPD system
(lam_hip_generate_random_spd, cond=1e6 so CG is still iterating at the end of the run).
At N=1 GPU the line also carries, under "also", configs[1] (N=32768) and the sizes the reference
published numbers for (N=10000/20000/40000, TESTS/BEST_RESULTS:362-372) with the per-iteration cost
outside the GEMV ("other_us"); they run after the headline in the same process and context.  Under
"config4_gemv": BASELINE configs[3], the N=131072 fp32 / bf16-storage GEMV (VALU kernel and MFMA variant).

N > 1 GPUs: ONE command measures BOTH multi-GPU topologies the product has, whichever way it is started.
  * started by a launcher (RANK / WORLD_SIZE in the environment: the driver's torchrun line): the HEADLINE is the rank mode --
    one process per GPU, the exchange is RCCL inside liblam_hip.so, the ranks find each other through the package's own socket
    rendezvous (no torch in the process) -- and rank 0 also runs the one-process topology in a child process
    ("one_process_topology");
  * started plainly (`python bench.py --gpus N`): the HEADLINE is one process driving N shards (peer stores over xGMI ordered
    by HIP events, the reference's ConjugateGradient_MultiGPUS_CUDA shape) and the rank mode runs as N child processes
    ("rank_mode_rccl": value, exchange modes 0 and 1, rccl_version, rccl_ranks = the communicator size RCCL reports,
    rccl_calls_enqueued, its own self-check).
Every child ("leg") is started BEFORE its parent touches the GPU (a process that has initialised the GPU must not start
children on this pool) and hands its record back through a file.  The EXPERIMENTAL exchange (option exchange = 2: in-kernel
flags over peer-mapped memory, never yet run on separate GPUs) runs only in legs of its own: a fault there costs that leg's
record (`{"error": ...}`), never the line.  Every N > 1 record carries `exchange_us` (HIP-event pairs around the
collective(s) / event join(s), lam_hip_stats.t_exchange) next to `gemv_ms`: gemv + exchange is the reference's `t_gemv` column
(ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:352-377).  The headline is always the product's DEFAULT exchange of its topology (the
other modes are recorded under "exchange_modes" only), and every line checks itself: true == recursive residual, and for N > 1
the residual of the one-GPU solve of the same system; a failed check prints "value": null with the reason and exits non-zero.
N > 1: the process's OWN topology runs in a worker thread under a supervisor (the main thread: --headline-timeout, SIGTERM from a
launcher that is tearing the job down).  If it raises, hangs or is torn down, rank 0 still prints ONE line: the headline it had
already measured (when only the comparison modes behind it failed: "comparison_error"), or else the OTHER topology's leg record as
the headline ("headline_from", "own_topology_error"; exit code 4), or -- with no usable leg -- "value": null with the reason (4).

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel (its name comes from the
library): achieved = algorithmic bytes of one launch / its average duration measured with HIP events on
the launch stream inside the timed steps; `roofline.traffic` = HBM bytes per launch from two short rocprofv3
PMC passes run as child processes at the start (N=1; otherwise the committed profile's value, tagged).
`cpu_baseline` (rank 0, N=1 only) times the reference's
own CPU driver (oracle/_ref, built from /root/reference in the build container) -- or, if that
binary is missing, the oracle port -- on a bounded sample.
"""
import argparse
import copy
import importlib
import json
import os
import signal
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "2024-eumaster4hpc-student-challenge_amd"
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
MFMA_VARIANTS = ((21, "MFMA v_mfma_f32_32x32x16_bf16, p as 3 bf16 terms"), (20, "MFMA, p rounded to bf16"))
ALSO_SIZES = (32768, 10000, 20000, 40000)
RANK_LABELS = {0: "allgather_x2+allgather_p", 1: "allgather_Ap", 2: "direct_mailboxes"}
LOCAL_LABELS = {0: "events_x3 (p.Ap, r.r, p slices: three joins per iteration)", 1: "gather_Ap (one join per iteration)", 2: "direct_flags"}


def under_profiler():
    """rocprofv3 preloads its tool library, which initialises the GPU before main(): a process in that
    state must not start children (gpurun's exec guard), so the side runs are dropped."""
    env = os.environ
    return ("rocprofiler" in env.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in env)
            or "rocprofiler" in env.get("HSA_TOOLS_LIB", ""))


def cpu_baseline(sample_n, iters):
    """Reference CPU path timed on this host's cores (reported baseline, not a target)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:   # container CPU share (cgroup v2 quota): the GPU box exposes 256 CPUs but grants 16
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    ref = os.path.join(ROOT, "oracle", "_ref", "test_CPU_MPI_OMP.out")
    env = dict(os.environ, OMP_NUM_THREADS=str(cores), OMP_PROC_BIND="close", OMP_PLACES="cores")
    if os.path.exists(ref):
        try:
            t0 = time.time()
            out = subprocess.run([ref, "-s", str(sample_n), "-i", str(iters), "-o", "/tmp/_bench_sol.bin"],
                                 env=env, capture_output=True, text=True, timeout=600, cwd="/tmp")
            line = out.stdout.replace("\n", "").strip()
            f = line.split(",")
            # CSV: N,P,threads,gen_s,avg_gemv,avg_iter,iters,err,total  -- the two averages are divided
            # by num_iters twice (ConjugateGradient_CPU_MPI_OMP.hpp:119-124) and num_iters is
            # max_iters+1 on exit, so true seconds/iteration = printed * num_iters^2 / max_iters
            ni = int(f[6])
            t_iter = float(f[5]) * ni * ni / iters
            t_gemv = float(f[4]) * ni * ni / iters
            return {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": int(f[2]), "kind": "reference",
                    "sample": f"oracle/_ref/test_CPU_MPI_OMP.out -s {sample_n} -i {iters} (generate mode, fp64, "
                              f"1 MPI rank x {f[2]} OpenMP threads; {sample_n}x{sample_n} matrix = "
                              f"{8.0 * sample_n * sample_n / 1e9:.1f} GB/iter)",
                    "gemv_gbps": 8.0 * sample_n * sample_n / t_gemv / 1e9, "sample_n": sample_n,
                    "wall_s": time.time() - t0}
        except Exception as e:   # fall through to the port
            sys.stderr.write(f"[bench] reference CPU driver failed ({e}); using the oracle port\n")
    from oracle import pyoracle
    st = pyoracle.cpu_baseline(sample_n, iters, cores)
    t_iter = st["t_total"] / iters
    return {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle port (cg_oracle.c, OpenMP x{cores}), generate mode N={sample_n} fp64, {iters} iterations",
            "gemv_gbps": 8.0 * sample_n * sample_n / (st["t_gemv"] / iters) / 1e9, "sample_n": sample_n}


def settle(s, seconds):
    """Untimed set-up, like the matrix generation: keep the device busy with GEMV launches for `seconds` before the
    warm-up steps.  What it waits out is NOT a clock ramp (round 3's reading): a fresh process on an idle device runs
    its very first launches at full rate (profiles/r04_bf16_gap_probe.txt).  It is the driver wiping VRAM that another
    process (or context) has just RELEASED: the child processes that run in front of this one (CPU baseline, two PMC
    passes, the legs of an N > 1 run) each free tens of GB at exit, the wipe runs in the background and takes HBM bandwidth
    from whatever runs next -- the same GEMV measures 3-4 % slower for a second or two after a multi-GB hipFree (same file:
    bf16 85.5 % of peak right after a 68.7 GB free, 88.4 % two seconds later and in a fresh process).  A timed region that
    starts inside that window would measure the neighbour's clean-up, not the rate a solve runs at.  The cold number is
    still reported (`value_cold`)."""
    t0 = time.perf_counter()
    rates = []
    while True:
        rates.append(1.0 / s.gemv_only(10))
        el = time.perf_counter() - t0
        # at least `seconds`; then until two consecutive windows of four samples agree to 0.3 % and the last one is flat to
        # 0.4 % (the wipe of 34 GB takes one to three seconds, depending on what else the node is doing); at most 5 s
        if el >= seconds and len(rates) >= 8:
            a, b = rates[-8:-4], rates[-4:]
            if abs(sum(b) / sum(a) - 1.0) < 0.003 and max(b) / min(b) - 1.0 < 0.004:
                break
        if el >= max(seconds, 5.0):
            break
    return time.perf_counter() - t0


def mfma_child(n4):
    """BASELINE configs[3]'s MFMA comparison: the MFMA-fed bf16 GEMV shapes are experiments that lost (slower than the VALU
    kernel) and live in the tuning build of the library only, so they are measured by a child process that loads
    liblam_hip_tuning.so -- started before this process touches the GPU, like the other children.  Prints JSON rows."""
    lam = importlib.import_module(PKG)
    rows = []
    with lam.Solver(lam.BF16) as s4:
        s4.generate_random_spd(n4, 1234, 1e4)
        s4.generate_random_rhs(1235)
        s4.cg_init()
        settle(s4, 0.3)
        gb = (2.0 * float(n4) * n4 + 4.0 * 2 * n4) / 1e9
        for v4, what4 in ((-1, "VALU (production), measured in the child next to the MFMA shapes"),) + MFMA_VARIANTS:
            s4.set_option("gemv_variant", v4)
            ts = sorted(s4.gemv_only(10) for _ in range(5))
            rows.append({"n": n4, "dtype": "bf16", "path": what4, "kernel": s4.gemv_kernel_name(), "gemv_ms": ts[2] * 1e3, "gemv_gbps": gb / ts[2],
                         "roofline_frac": gb / ts[2] / HBM_PEAK_GBPS, "algorithmic_bytes": gb * 1e9, "library": "liblam_hip_tuning.so (child process)"})
    print(json.dumps(rows))


def run_mfma_child(n4):
    lam = importlib.import_module(PKG)
    if not os.path.exists(lam.TUNING_LIB):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--mfma-child", "--config4-n", str(n4)], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, LAM_HIP_LIB=lam.TUNING_LIB))
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        sys.stderr.write(f"[bench] MFMA comparison child failed: {e}\n")
        return None


def run_config(s, n, warmup, steps, barrier, seed=1234, cond=1e6, symmetric=False, generate=True, ramp_s=0.0):
    """W untimed + K timed CG iterations on solver `s`.  The context keeps its matrix allocation when N
    shrinks (grow-only), so several configurations can follow each other in one process without a large
    hipFree in between ("allocation history")."""
    if generate:
        s.generate_random_spd(n, seed, cond)
        s.generate_random_rhs(seed + 1)
    s.set_option("symmetric", 1 if symmetric else 0)
    if symmetric and s.get_option("symmetric_effective") != 1:
        raise RuntimeError("option 'symmetric' is not available for this configuration")
    cold = None
    if ramp_s > 0:
        # for the record: the same W + K steps measured WITHOUT the ramp, i.e. what round 2's bench measured (the device's
        # first ~0.3 s of load after idling); reported next to the headline as "cold_start", never as `value`
        s.cg_init()
        if warmup > 0:
            s.cg_iterate(warmup, 0.0)
        barrier()
        t0c = time.perf_counter()
        st_c = s.cg_iterate(steps, 0.0)
        barrier()
        cold = {"value": steps / (time.perf_counter() - t0c), "gemv_ms": st_c["t_gemv"] * 1e3,
                "what": "the same warm-up + timed steps right after the matrix generation, before the settling launches "
                        "(inside the window in which the driver still wipes the VRAM the child processes released)"}
        cold["settled_for_s"] = settle(s, ramp_s)
    run_config.cold = cold
    s.cg_init()
    if warmup > 0:
        s.cg_iterate(warmup, 0.0)
    barrier()
    t0 = time.perf_counter()
    st = s.cg_iterate(steps, 0.0)          # returns after its streams are synchronised
    barrier()
    dt = time.perf_counter() - t0
    return st, dt


def measure_traffic(n):
    """HBM bytes per launch of the dominant kernel, measured NOW: two short rocprofv3 PMC passes over this very
    script (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; bytes = (2*FETCH_SIZE
    + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE counts a 128-B request as 64 B), run as child processes before this
    process touches the GPU.  Returns (bytes, source-dict) or (None, reason)."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    vals, kernel = {}, None
    work = tempfile.mkdtemp(prefix="lam_pmc_")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.abspath(__file__), "--order", str(n), "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-also",
                   "--no-traffic"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=work, env=dict(os.environ, TMPDIR=work))
            files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode})"
            rows = [x for x in csv.DictReader(open(files[0])) if "gemv_" in x["Kernel_Name"] and x["Counter_Name"] == ctr]
            if not rows:
                return None, f"no GEMV launch in the {ctr} pass"
            grid = max(int(x["Grid_Size"]) for x in rows)
            sel = [float(x["Counter_Value"]) for x in rows if int(x["Grid_Size"]) == grid]
            vals[ctr] = sum(sel) / len(sel)
            kernel = next(x["Kernel_Name"] for x in rows if int(x["Grid_Size"]) == grid)
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, {
            "measured_in_this_run": True, "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 5 steps each) "
                                                 "on this script as child processes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
            "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "kernel": kernel}
    except Exception as e:   # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


def traffic_record(n, n_gpus):
    """HBM bytes per launch of the dominant kernel from the PMC counters.  They cannot be collected inside
    an un-profiled run, so this is the value of the committed rocprofv3 passes (profiles/traffic.json), tagged
    with where it came from."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        e = json.load(open(tpath))[f"n{n}_p{n_gpus}"]
        return e["hbm_bytes_per_launch"], {"measured_in_this_run": False, "file": "profiles/traffic.json", "counters": e.get("source"),
                                           "library_commit": e.get("commit"), "kernel": e.get("kernel")}
    except Exception:
        return None, None


# ---------------------------------------------------------------------------------------------------------------------
# clocks / power during the measurement ("clocks/power state noted")
# ---------------------------------------------------------------------------------------------------------------------
_SAMPLER = r"""
import ctypes, json, subprocess, sys, time
ctypes.CDLL(None).prctl(1, 9)      # PR_SET_PDEATHSIG: die with the bench process
out = open(sys.argv[1], "a")
while True:
    t = time.time()
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out.write(json.dumps({"t": t, "smi": json.loads(line)}) + "\n")
        out.flush()
    except Exception:
        pass
    time.sleep(0.2)
"""


class DeviceStateSampler:
    """A child process (started BEFORE this process touches the GPU; it only reads sysfs through rocm-smi) that notes the device's
    clocks and power every ~0.3 s; summary() reports what it saw inside a time window -- the headline's settle + warm-up + timed
    steps.  Best effort: no rocm-smi, no record."""

    def __init__(self):
        import shutil
        self.proc, self.path = None, None
        if shutil.which("rocm-smi") is None:
            return
        fd, self.path = tempfile.mkstemp(prefix="lam_smi_", suffix=".jsonl")
        os.close(fd)
        try:
            self.proc = subprocess.Popen([sys.executable, "-c", _SAMPLER, self.path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        except Exception:   # noqa: BLE001
            self.proc = None

    def stop(self):
        if self.proc is not None and self.proc.poll() is None:
            try:
                os.killpg(self.proc.pid, signal.SIGKILL)
                self.proc.wait(timeout=5)
            except Exception:   # noqa: BLE001
                pass
        self.proc = None

    def summary(self, t0, t1):
        self.stop()
        if self.path is None:
            return None
        try:
            rows = [json.loads(l) for l in open(self.path) if l.strip()]
            os.unlink(self.path)
        except Exception:   # noqa: BLE001
            return None
        inside = [r for r in rows if t0 <= r["t"] <= t1] or rows[-1:]

        def stat(key, unit):
            vals = []
            for r in inside:
                for card in r["smi"].values():
                    v = str(card.get(key, "")).strip("()").lower().replace(unit, "")
                    try:
                        vals.append(float(v))
                    except ValueError:
                        pass
            vals.sort()
            return {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]} if vals else None

        return {"samples_in_window": len([r for r in rows if t0 <= r["t"] <= t1]), "window_s": t1 - t0,
                "sclk_mhz": stat("sclk clock speed:", "mhz"), "mclk_mhz": stat("mclk clock speed:", "mhz"), "fclk_mhz": stat("fclk clock speed:", "mhz"),
                "socket_power_w": stat("Current Socket Graphics Package Power (W)", "w"),
                "how": "rocm-smi --showclocks --showpower sampled every ~0.3 s by a child process during the headline's settling launches, warm-up and timed steps"}


# ---------------------------------------------------------------------------------------------------------------------
# legs: measurements that run in child processes of their own
# ---------------------------------------------------------------------------------------------------------------------
def maybe_die(leg, where):
    """Test hook (tests/test_gpu_rank_mock.py): LAM_BENCH_KILL_LEG=<leg>[:<where>] makes every process of that leg kill itself
    at that point -- what a GPU memory fault on a peer mapping amounts to (the process is gone, nothing is written)."""
    spec = os.environ.get("LAM_BENCH_KILL_LEG", "")
    if not spec:
        return
    name, _, at = spec.partition(":")
    if name == leg and (at or "timed") == where:
        sys.stderr.write(f"[bench] leg {leg}: killing myself at '{where}' (LAM_BENCH_KILL_LEG)\n")
        sys.stderr.flush()
        os.kill(os.getpid(), signal.SIGKILL)


def leg_command(args, leg, out_path):
    return [sys.executable, os.path.abspath(__file__), "--leg", leg, "--leg-out", out_path, "--gpus", str(max(1, args.gpus)), "--steps", str(args.steps),
            "--warmup", str(args.warmup), "--order", str(args.n), "--gemv-timing", str(args.gemv_timing)]


def run_leg(args, leg, nprocs, timeout, rank_env=None, rdzv_file=None, expect_record=True):
    """Start the `nprocs` processes of one leg (this script with --leg), wait for them, return the record rank 0 of the leg
    wrote -- or {"error": ...} when a process died, timed out or left no record: a leg can never cost the caller its own
    line.  rank_env(i) = extra environment of process i (None: one plain process).  Must be called BEFORE the caller touches the
    GPU.  expect_record = False: this caller's processes are not the ones that write the record (a rank > 0 under a launcher)."""
    work = tempfile.mkdtemp(prefix=f"lam_leg_{leg}_")
    out_path = os.path.join(work, "record.json")
    t0 = time.time()
    procs = []
    try:
        for i in range(nprocs):
            env = dict(os.environ)
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LAM_RDZV_FILE", "LAM_JOB_ID"):
                env.pop(k, None)
            if rank_env is not None:
                env.update(rank_env(i))
                env["LAM_RDZV_FILE"] = rdzv_file or os.path.join(work, "rdzv")
            # stderr goes to a file of its own (a pipe nobody drains would block a chatty child) and is echoed afterwards
            errf = open(os.path.join(work, f"stderr.{i}"), "w+")
            procs.append(subprocess.Popen(leg_command(args, leg, out_path), env=env, stdout=subprocess.DEVNULL, stderr=errf, start_new_session=True))
            procs[-1].errf = errf
        deadline = t0 + timeout
        errs = []
        for i, p in enumerate(procs):
            try:
                p.wait(timeout=max(1.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                errs.append(f"process {i} did not finish within {timeout:.0f} s")
                continue
            p.errf.seek(0)
            err = p.errf.read()
            if err.strip():
                sys.stderr.write("".join(f"[leg {leg}.{i}] {l}\n" for l in err.strip().splitlines()[-12:]))
            if p.returncode != 0:
                tail = (err or "").strip().splitlines()[-3:]
                errs.append(f"process {i} exited with {'signal ' + str(-p.returncode) if p.returncode < 0 else 'code ' + str(p.returncode)}"
                            + (": " + " | ".join(tail)[-400:] if tail else ""))
        rec = None
        if os.path.exists(out_path):
            try:
                rec = json.load(open(out_path))
            except Exception as e:   # noqa: BLE001
                errs.append(f"unreadable record: {e}")
        if errs:
            rec = dict(rec or {}, error="; ".join(errs)[:900])
        elif rec is None:
            rec = {"error": "the leg left no record"} if expect_record else {}
        rec["leg_wall_s"] = time.time() - t0
        return rec
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:400], "leg_wall_s": time.time() - t0}
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)       # its own session: the process and whatever it started
                except OSError:
                    pass
                try:
                    p.wait(timeout=10)
                except Exception:   # noqa: BLE001
                    pass
            p.errf.close()
        import shutil
        shutil.rmtree(work, ignore_errors=True)


def write_leg_record(path, rec):
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump(rec, f)
    os.replace(tmp, path)


# ---------------------------------------------------------------------------------------------------------------------
# the two multi-GPU topologies
# ---------------------------------------------------------------------------------------------------------------------
def maybe_fail_main(leg_name, where):
    """Test hook (tests/test_gpu_rank_mock.py): LAM_BENCH_FAIL_MAIN=<raise|hang>[:<start|late>] makes the process's OWN topology
    (not a leg) raise or hang at its start or behind its headline measurement (in the comparison modes)."""
    spec = os.environ.get("LAM_BENCH_FAIL_MAIN", "")
    if not spec or leg_name is not None:
        return
    how, _, at = spec.partition(":")
    if (at or "start") != where:
        return
    only = os.environ.get("LAM_BENCH_FAIL_MAIN_RANK", "")        # under a launcher: only this rank fails
    if only != "" and only != os.environ.get("RANK", "0"):
        return
    sys.stderr.write(f"[bench] own topology: '{how}' at '{where}' (LAM_BENCH_FAIL_MAIN)\n")
    if how == "hang":
        time.sleep(3600)
    raise RuntimeError(f"LAM_BENCH_FAIL_MAIN={spec}")


class Supervised:
    """Runs fn(on_headline) in a worker thread; the calling (main) thread waits for it, for `timeout` seconds or for SIGTERM.
    on_headline(rec) is what fn calls once its headline measurement is complete (what follows it only adds comparison modes)."""

    def __init__(self, fn, timeout):
        self.box = {}
        self.terminated = False
        self.timeout = timeout

        def work():
            try:
                self.box["rec"] = fn(lambda rec: self.box.__setitem__("headline", rec))
            except BaseException as e:   # noqa: BLE001
                self.box["err"] = f"{type(e).__name__}: {e}"[:600]

        self.thread = threading.Thread(target=work, daemon=True, name="own-topology")

    def run(self):
        old = None
        try:
            old = signal.signal(signal.SIGTERM, lambda *_: setattr(self, "terminated", True))
        except ValueError:       # not the main thread (never in this script)
            pass
        t0 = time.time()
        self.thread.start()
        while self.thread.is_alive() and time.time() - t0 < self.timeout and not self.terminated:
            self.thread.join(0.2)
        if old is not None and not self.terminated:
            signal.signal(signal.SIGTERM, old)
        if "rec" in self.box:
            return self.box["rec"], None
        err = self.box.get("err") or ("torn down by SIGTERM (the launcher lost another rank?)" if self.terminated
                                      else f"no result within {self.timeout:.0f} s (--headline-timeout)")
        return None, err


def record_from_leg(leg, steps):
    """The other topology's leg record (as public() wrote it) in the shape the measure functions return: the fallback headline."""
    st = {"t_gemv": leg["gemv_ms"] * 1e-3, "t_exchange": leg.get("exchange_us", 0.0) * 1e-6, "gemv_bytes": leg["gemv_bytes_per_launch"],
          "rel_err": leg["rel_residual_recursive"], "t_comm_init": leg.get("rccl_init_s") or 0.0}
    if "exchange_us_min_over_ranks" in leg:
        st["t_exchange_min"] = leg["exchange_us_min_over_ranks"] * 1e-6
    return {"st": st, "dt": leg["ms_per_step"] * 1e-3 * steps, "true_res": leg["rel_residual_true"], "self_check": leg.get("self_check"),
            "failures": [leg["error"]] if leg.get("error") else [], "kernel": leg.get("kernel"), "cold_start": None,
            "parallelism": leg.get("parallelism"), "host_enqueue_us_per_step": leg.get("host_enqueue_us_per_step"),
            "exchange_effective": leg.get("exchange_effective"), "exchange_modes": leg.get("exchange_modes"),
            "device_ids": leg.get("device_ids"), "rccl_version": leg.get("rccl_version"), "rccl_ranks": leg.get("rccl_ranks"),
            "rccl_calls_enqueued": leg.get("rccl_calls_enqueued")}


def mode_record(steps, dt_, st_, res_, **extra):
    rec = {"value": steps / dt_, "ms_per_step": dt_ / steps * 1e3, "gemv_ms": st_["t_gemv"] * 1e3, "exchange_us": st_["t_exchange"] * 1e6,
           "gemv_plus_comm_ms": (st_["t_gemv"] + st_["t_exchange"]) * 1e3,
           # what is left of the iteration: the vector step, launch gaps, and whatever of the exchange is NOT on this stream's critical path
           "other_us": (dt_ / steps - st_["t_gemv"] - st_["t_exchange"]) * 1e6,
           "rel_residual_true": res_, "rel_residual_recursive": st_["rel_err"]}
    if "t_exchange_min" in st_:
        rec["exchange_us_min_over_ranks"] = st_["t_exchange_min"] * 1e6
    rec.update(extra)
    return rec


def one_gpu_reference(lam, n, iters, device):
    """The same solve on ONE GPU (N=65536 fits one MI355X): what every multi-GPU line is checked against."""
    with lam.Solver(lam.F64, device_ids=[device]) as ref:
        # LAM_BENCH_SELFTEST_FAIL=1 (tests only): the reference solves a DIFFERENT system, so the check must fail
        ref.generate_random_spd(n, 1234 + (1 if os.environ.get("LAM_BENCH_SELFTEST_FAIL") else 0), 1e6)
        ref.generate_random_rhs(1235)
        ref.cg_init()
        return ref.cg_iterate(iters, 0.0)["rel_err"]


def self_check(lam, args, n_gpus, st, true_res, rank, rdzv, symmetric=False):
    """true == recursive residual, and with more than one GPU the residual must agree with the SAME solve on one GPU (rank 0 runs
    it on its own device right here).  Returns (check dict, failures)."""
    check = {"true_vs_recursive": abs(true_res / st["rel_err"] - 1) if st["rel_err"] > 0 else float("inf"), "tolerance_true_vs_recursive": 1e-6}
    failures = []
    if not check["true_vs_recursive"] < 1e-6:
        failures.append(f"true residual {true_res:.15e} != recursive residual {st['rel_err']:.15e}")
    if n_gpus > 1 and not symmetric:
        ref_err, ref_fail = None, None
        if rank == 0:
            try:
                ref_err = one_gpu_reference(lam, args.n, args.warmup + args.steps, rdzv.local_rank % max(1, lam.device_count()) if rdzv else 0)
            except Exception as e:   # noqa: BLE001
                ref_fail = f"one-GPU reference solve failed: {e}"[:300]
            check["one_gpu_reference_residual"] = ref_err
            check["tolerance_vs_one_gpu"] = 1e-9
            if ref_err is None:
                failures.append(ref_fail or "no one-GPU reference")
            else:
                check["vs_one_gpu"] = abs(st["rel_err"] / ref_err - 1)
                if not check["vs_one_gpu"] < 1e-9:
                    failures.append(f"residual after {args.warmup + args.steps} iterations {st['rel_err']:.15e} differs from the one-GPU solve {ref_err:.15e}")
        if rdzv is not None:
            rdzv.barrier()
    check["passed"] = not failures
    return check, failures


def rank_mode_measure(lam, args, rdzv, part, leg_name=None, on_headline=None):
    """One process per GPU (this process is one rank; RCCL inside liblam_hip.so).  part = "main": the default exchange (the
    headline of this topology) + the other product exchanges + the symmetric option; part = "direct": the EXPERIMENTAL direct
    exchange only (runs in a leg of its own).  Returns this rank's record (rank 0's is the one that counts)."""
    rank, world = rdzv.rank, rdzv.size
    maybe_fail_main(leg_name, "start")
    uid = rdzv.broadcast(lam.get_unique_id() if rank == 0 else b"")
    ndev = lam.device_count()          # counting devices does not initialise the GPU
    barrier = rdzv.barrier
    n = args.n

    def max_over_ranks(dt_, st_):
        # the collective cannot complete before the SLOWEST rank has contributed: the rank whose GEMV ends last sees the collective's
        # own latency, every other one that latency plus its wait for the last -- so the minimum over the ranks is the wire + launch
        # latency L of DESIGN.md's model, the maximum is L + the ranks' skew (both are recorded)
        t_min = -rdzv.max([-st_["t_exchange"]])[0]
        dt_, st_["t_gemv"], st_["t_exchange"] = rdzv.max([dt_, st_["t_gemv"], st_["t_exchange"]])
        st_["t_exchange_min"] = t_min
        return dt_, st_

    s = lam.Solver(lam.F64, rank=rank, nranks=world, device_id=rdzv.local_rank % max(1, ndev), unique_id=uid)
    try:
        s.set_option("gemv_timing", args.gemv_timing)
        default_exchange = s.get_option("exchange")
        what = {0: "RCCL all-gather x2 (8 B/rank) + all-gather(p) per iteration", 1: "ONE RCCL all-gather of [Ap slice | p.Ap part] per iteration (gather-Ap)",
                2: "direct stores into peer-mapped mailboxes"}.get(default_exchange, f"exchange {default_exchange}")
        rec = {"n_gpus": world, "rccl_version": lam.rccl_version(), "rccl_ranks": s.get_option("rccl_ranks"), "rccl_init_s": None,
               "parallelism": f"row-sharded x{world}, 1 process/GPU, {what}"}
        if part == "main":
            host_ns0 = s.get_option("host_enqueue_ns")
            st, dt = run_config(s, n, args.warmup, args.steps, barrier, ramp_s=args.ramp if leg_name is None else min(args.ramp, 0.3))
            maybe_die(leg_name, "main")
            rec["kernel"] = s.gemv_kernel_name()
            rec["cold_start"] = run_config.cold
            rec["exchange_effective"] = s.get_option("exchange_effective")
            rec["host_enqueue_us_per_step"] = (s.get_option("host_enqueue_ns") - host_ns0) * 1e-3 / ((args.warmup + args.steps) * (2 if args.ramp > 0 else 1))
            dt, st = max_over_ranks(dt, st)
            true_res = s.true_residual()
            check, failures = self_check(lam, args, world, st, true_res, rank, rdzv)
            rec.update(dt=dt, st=st, true_res=true_res, self_check=check, failures=failures, rccl_init_s=st.get("t_comm_init", 0.0))
            default_label = RANK_LABELS.get(default_exchange, str(default_exchange))
            modes = {"default": default_label, default_label: mode_record(args.steps, dt, st, true_res)}
            # the headline is complete: whatever fails behind this point costs comparison modes, not the line (see Supervised)
            rec["exchange_modes"] = modes
            rec["rccl_calls_enqueued"] = s.get_option("collectives_enqueued")
            if on_headline is not None:
                on_headline(copy.deepcopy(rec))
            maybe_fail_main(leg_name, "late")

            def timed(label, **opts):
                for k_, v_ in opts.items():
                    s.set_option(k_, v_)
                s.cg_init()
                if args.warmup > 0:
                    s.cg_iterate(args.warmup, 0.0)
                barrier()
                t0_ = time.perf_counter()
                st_ = s.cg_iterate(args.steps, 0.0)
                barrier()
                dt_, st_ = max_over_ranks(time.perf_counter() - t0_, st_)
                modes[label] = mode_record(args.steps, dt_, st_, s.true_residual())

            if world > 1 or os.environ.get("LAM_HIP_FORCE_RCCL", "0") not in ("", "0"):
                # Same problem, same context, the other exchanges of the rank mode (all product paths under the same parity
                # tests) -- recorded for comparison, never the headline.
                if default_exchange != 0:
                    timed("allgather_x2+allgather_p", exchange=0, overlap=1)
                timed("allgather_x2+allgather_p, no overlap", exchange=0, overlap=0)
                if default_exchange != 1:
                    timed("allgather_Ap", exchange=1, overlap=1)
                # the opt-in symmetric product on row shards (every pair {i, j} read once, cyclic half windows; each rank gathers the
                # others' full-length contributions): a different algorithm, never the headline
                sym_label = "allgather_Ap + symmetric product (option, not the headline)"
                try:
                    s.set_option("exchange", 1)
                    s.set_option("symmetric", 1)
                    if s.get_option("symmetric_effective") != 1:       # the same answer on every rank (it depends on N and the rank count only)
                        modes[sym_label] = {"error": "option not effective for this configuration"}
                    else:
                        timed(sym_label, exchange=1, overlap=1, symmetric=1)
                except Exception as e:   # noqa: BLE001  -- recorded, never fatal for the headline
                    modes[sym_label] = {"error": str(e)[:300]}
                finally:
                    s.set_option("symmetric", 0)
                s.set_option("exchange", default_exchange)
                s.set_option("overlap", 1)
            rec["exchange_modes"] = modes
        else:
            # The direct exchange (peer-mapped mailboxes, no collective call inside the iteration; EXPERIMENTAL until it has run on
            # real peers).  Every step ends with an agreement over the control plane, so all ranks take the same path: a rank that
            # cannot map its peers (all ranks then fall back to exchange 0) or a bounded wait that expires ends the attempt on ALL
            # ranks, and nothing collective on the device follows it.
            modes = {}
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)

            def agree(ok_):
                return all(x == b"1" for x in rdzv.allgather(b"1" if ok_ else b"0"))

            def attempt(fn):
                try:
                    return True, fn(), None
                except Exception as e:   # noqa: BLE001
                    return False, None, str(e)[:300]

            def init_direct(overlap):
                s.set_option("exchange", 2)
                s.set_option("overlap", overlap)
                s.cg_init()
                if s.get_option("exchange_effective") != 2:
                    raise RuntimeError("peer mappings not available: fell back to the RCCL exchange")

            def try_direct(label, overlap):
                """One timed run on the direct exchange; records it under `label`; True if it produced a number."""
                ok_, _, err_ = attempt(lambda: init_direct(overlap))
                if not agree(ok_):
                    modes[label] = {"error": err_ or "another rank could not set up the direct exchange"}
                    return False
                maybe_die(leg_name, "init")
                ok_, _, err_ = attempt(lambda: s.cg_iterate(args.warmup, 0.0) if args.warmup > 0 else None)
                if not agree(ok_):                          # doubles as the barrier in front of the timed region
                    modes[label] = {"error": err_ or "another rank failed in the warm-up"}
                    return False
                maybe_die(leg_name, "timed")
                t0_ = time.perf_counter()
                ok_, st_, err_ = attempt(lambda: s.cg_iterate(args.steps, 0.0))
                all_ok_ = agree(ok_)                        # doubles as the closing barrier
                dt_ = time.perf_counter() - t0_
                if not all_ok_:
                    modes[label] = {"error": err_ or "another rank failed in the timed iterations"}
                    return False
                dt_, st_ = max_over_ranks(dt_, st_)
                res_ = s.true_residual()
                modes[label] = mode_record(args.steps, dt_, st_, res_, experimental=True)
                if not abs(res_ / st_["rel_err"] - 1) < 1e-6:
                    modes[label]["error"] = "recomputed residual differs from the recursive one: WRONG RESULT on this hardware"
                return True

            if try_direct("direct_mailboxes", 1):
                try_direct("direct_mailboxes, no split", 0)
            rec["exchange_modes"] = modes
        rec["rccl_calls_enqueued"] = s.get_option("collectives_enqueued")
        return rec
    finally:
        s.close()


def one_process_measure(lam, args, n_gpus, part, leg_name=None, on_headline=None):
    """ONE process driving all shards (the reference's ConjugateGradient_MultiGPUS_CUDA topology).  part = "main": the default
    exchange (gather-Ap) + every other product exchange of this topology + the symmetric option, each with the residual check
    against the one-GPU solve and the host time it takes to enqueue an iteration; part = "direct": the EXPERIMENTAL in-kernel flag
    exchange only."""
    n = args.n
    maybe_fail_main(leg_name, "start")
    # LAM_BENCH_DEVICE_IDS="0,0" (tests on a one-GPU box): put the shards of the one-process topology on these devices
    dev_override = [int(x) for x in os.environ.get("LAM_BENCH_DEVICE_IDS", "").split(",") if x.strip() != ""]
    ndev = max(1, lam.device_count())
    device_ids = dev_override if len(dev_override) == n_gpus else [q % ndev for q in range(n_gpus)]     # fewer devices than shards: shared (an emulation)
    s = lam.Solver(lam.F64, n_shards=n_gpus, device_ids=device_ids)
    try:
        s.set_option("gemv_timing", args.gemv_timing)
        rec = {"n_gpus": n_gpus, "parallelism": f"row-sharded x{n_gpus}, 1 process, direct xGMI peer stores ordered by HIP events", "device_ids": device_ids}
        if len(set(device_ids)) < n_gpus:
            rec["parallelism"] += f" -- EMULATION: {n_gpus} shards on {len(set(device_ids))} device(s), not a multi-GPU measurement"
        default_exchange, default_join = s.get_option("exchange"), s.get_option("exchange_join")

        def nobarrier():
            pass

        if part == "main":
            host_ns0 = s.get_option("host_enqueue_ns")
            st, dt = run_config(s, n, args.warmup, args.steps, nobarrier, ramp_s=args.ramp if leg_name is None else min(args.ramp, 0.3))
            rec["kernel"] = s.gemv_kernel_name()
            rec["cold_start"] = run_config.cold
            gemv_fastest = s.get_option("gemv_ns_min_shard") * 1e-6      # ms; st["t_gemv"] is the slowest shard's: the skew between them
            rec["gemv_ms_fastest_shard"] = gemv_fastest
            effective = s.get_option("exchange_effective")
            rec["exchange_effective"] = effective
            host_us = (s.get_option("host_enqueue_ns") - host_ns0) * 1e-3 / ((args.warmup + args.steps) * (2 if args.ramp > 0 else 1))
            rec["host_enqueue_us_per_step"] = host_us
            true_res = s.true_residual()
            check, failures = self_check(lam, args, n_gpus, st, true_res, 0, None)
            rec.update(dt=dt, st=st, true_res=true_res, self_check=check, failures=failures)
            eff_label = LOCAL_LABELS.get(effective, str(effective))
            if effective == 1:
                eff_label += ", join through shard 0" if default_join else ", all-to-all join"
            ref_err = check.get("one_gpu_reference_residual")

            def vs_ref(err_):
                return abs(err_ / ref_err - 1) if ref_err else None

            modes = {"default": eff_label,
                     eff_label: mode_record(args.steps, dt, st, true_res, host_enqueue_us_per_step=host_us, vs_one_gpu=vs_ref(st["rel_err"]),
                                            gemv_ms_fastest_shard=gemv_fastest)}
            rec["exchange_modes"] = modes
            # the headline is complete: whatever fails behind this point costs comparison modes, not the line (see Supervised)
            if on_headline is not None:
                on_headline(copy.deepcopy(rec))
            maybe_fail_main(leg_name, "late")
            timed_local = make_timed_local(s, args, modes, vs_ref)
            timed_local(LOCAL_LABELS[1] + ", join through shard 0", exchange=1, exchange_join=1)
            timed_local(LOCAL_LABELS[1] + ", all-to-all join", exchange=1, exchange_join=0)
            timed_local(LOCAL_LABELS[0], exchange=0)
            # the opt-in symmetric product on row shards (cyclic half windows, every shard contributes a full-length vector per
            # iteration): a different algorithm -- other rounding, hence the wider gate against the one-GPU solve --, never the headline
            try:
                timed_local(LOCAL_LABELS[1] + " + symmetric product (option, not the headline)", tol_vs=1e-6, exchange=1, exchange_join=default_join, symmetric=1)
            finally:
                s.set_option("symmetric", 0)
            try:
                s.set_option("exchange", default_exchange)
                s.set_option("exchange_join", default_join)
                s.set_option("overlap", 1)
            except Exception:   # noqa: BLE001
                pass
        else:
            modes = {}
            rec["exchange_modes"] = modes
            s.generate_random_spd(n, 1234, 1e6)
            s.generate_random_rhs(1235)
            ref_err = None
            try:        # this leg's own one-GPU reference: its residual gate does not depend on the parent
                ref_err = one_gpu_reference(lam, n, args.warmup + args.steps, device_ids[0])
            except Exception as e:   # noqa: BLE001
                rec["one_gpu_reference_error"] = str(e)[:300]

            def vs_ref(err_):
                return abs(err_ / ref_err - 1) if ref_err else None

            timed_local = make_timed_local(s, args, modes, vs_ref, leg_name=leg_name)
            # an expired in-kernel wait leaves the context unusable (the error is recorded)
            timed_local(LOCAL_LABELS[2] + ", own-slice panel first", experimental=True, exchange=2, overlap=1)
            if "error" not in modes[LOCAL_LABELS[2] + ", own-slice panel first"]:
                timed_local(LOCAL_LABELS[2] + ", no split", experimental=True, exchange=2, overlap=0)
        return rec
    finally:
        s.close()


def make_timed_local(s, args, modes, vs_ref, leg_name=None):
    def timed_local(label, experimental=False, tol_vs=1e-9, **opts):
        if label in modes:
            return
        try:
            for k_, v_ in opts.items():
                s.set_option(k_, v_)
            s.cg_init()
            want = opts.get("exchange")
            if want is not None and s.get_option("exchange_effective") != want:
                raise RuntimeError(f"exchange {want} is not available for this configuration (shared devices, peer mappings)")
            if opts.get("symmetric") and s.get_option("symmetric_effective") != 1:
                raise RuntimeError("option symmetric is not effective for this configuration")
            if experimental:
                maybe_die(leg_name, "init")
            if args.warmup > 0:
                s.cg_iterate(args.warmup, 0.0)
            if experimental:
                maybe_die(leg_name, "timed")
            h0_ = s.get_option("host_enqueue_ns")
            t0_ = time.perf_counter()
            st_ = s.cg_iterate(args.steps, 0.0)
            dt_ = time.perf_counter() - t0_
            h1_ = s.get_option("host_enqueue_ns")
            res_ = s.true_residual()
            rec = mode_record(args.steps, dt_, st_, res_, host_enqueue_us_per_step=(h1_ - h0_) / args.steps * 1e-3, vs_one_gpu=vs_ref(st_["rel_err"]),
                              gemv_ms_fastest_shard=s.get_option("gemv_ns_min_shard") * 1e-6)       # gemv_ms is the SLOWEST shard's
            if experimental:
                rec["experimental"] = True
            if not (abs(res_ / st_["rel_err"] - 1) < 1e-6 and (rec["vs_one_gpu"] is None or rec["vs_one_gpu"] < tol_vs)):
                rec["error"] = "residual differs from the one-GPU solve: WRONG RESULT on this hardware"
            modes[label] = rec
        except Exception as e:   # noqa: BLE001
            modes[label] = {"error": str(e)[:300]}
    return timed_local


def public(rec):
    """A topology record as it goes into the JSON line (the raw stats and timers stay inside)."""
    if rec is None:
        return None
    out = {k: v for k, v in rec.items() if k not in ("dt", "st", "true_res", "failures", "cold_start")}
    if "st" in rec:
        st, dt = rec["st"], rec["dt"]
        steps = rec.get("steps")
        out.update(mode_record(steps, dt, st, rec["true_res"]) if steps else {})
        out["gemv_bytes_per_launch"] = st.get("gemv_bytes")
        if rec.get("failures"):
            out["error"] = "; ".join(rec["failures"])
            out["value_unchecked"], out["value"] = out.get("value"), None
    return out


def merge_direct(target_modes, leg_rec, target_res):
    """Merge an EXPERIMENTAL leg's exchange modes into the modes of the topology it belongs to, checked against that topology's
    residual (all exchanges are deterministic and agree to rounding); a leg without a record becomes one error entry."""
    if target_modes is None or leg_rec is None:
        return
    got = leg_rec.get("exchange_modes") or {}
    for label, m in got.items():
        if isinstance(m, dict) and "rel_residual_true" in m and target_res and not abs(m["rel_residual_true"] / target_res - 1) < 1e-6 and "error" not in m:
            m["error"] = "residual differs from the default exchange: WRONG RESULT on this hardware"
        target_modes[label] = m
    if not got:
        target_modes["direct (experimental leg)"] = {"error": leg_rec.get("error", "the leg produced no record"), "experimental": True}
    elif leg_rec.get("error"):
        target_modes["direct (experimental leg)"] = {"error": leg_rec["error"], "experimental": True}


def leg_main(args):
    """Child-process mode: run one leg and leave its record in --leg-out (rank 0 of the leg writes it)."""
    try:        # a leg lives in a session of its own (so that its parent can kill all of it): make sure it also DIES with its parent
        import ctypes
        ctypes.CDLL(None).prctl(1, signal.SIGKILL)          # PR_SET_PDEATHSIG
    except Exception:   # noqa: BLE001
        pass
    if args.leg.startswith("_selftest"):
        # no GPU, no library: what tests/test_bench_cpu.py drives run_leg with (a record, a failure, a hang, one bad rank of several)
        rank = int(os.environ.get("RANK", "0"))
        if args.leg == "_selftest_hang":
            time.sleep(600)
        if args.leg == "_selftest_fail" or (args.leg == "_selftest_rank1_fails" and rank == 1):
            sys.stderr.write(f"selftest: rank {rank} fails on purpose\n")
            sys.exit(7)
        if rank == 0:
            write_leg_record(args.leg_out, {"value": 1.0, "world": int(os.environ.get("WORLD_SIZE", "1")), "rdzv_file": os.environ.get("LAM_RDZV_FILE")})
        return 0
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.dup2(2, 1)                      # nothing of a leg goes to the caller's stdout
    lam = importlib.import_module(PKG)
    lam.lib()
    leg = args.leg
    n_gpus = max(1, args.gpus)
    rdzv = lam.Rendezvous() if leg.startswith("rank") else None
    try:
        if leg == "rank_mode":
            rec = rank_mode_measure(lam, args, rdzv, "main", leg_name=leg)
        elif leg == "rank_direct":
            rec = rank_mode_measure(lam, args, rdzv, "direct", leg_name=leg)
        elif leg == "one_process":
            rec = one_process_measure(lam, args, n_gpus, "main", leg_name=leg)
        elif leg == "one_direct":
            rec = one_process_measure(lam, args, n_gpus, "direct", leg_name=leg)
        else:
            raise SystemExit(f"unknown leg '{leg}'")
        rec["steps"] = args.steps
        if rdzv is None or rdzv.rank == 0:
            write_leg_record(args.leg_out, public(rec))
    finally:
        if rdzv is not None:
            try:
                rdzv.barrier()
            except Exception:   # noqa: BLE001
                pass
            rdzv.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    # under torch.distributed.run use --order: the launcher's own parser treats "--n" as an ambiguous
    # abbreviation of its --nnodes/--nproc-per-node/... options even after the script name
    ap.add_argument("--n", "--order", dest="n", type=int, default=65536, help="matrix order (default: BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra sizes and the symmetric-option run")
    ap.add_argument("--symmetric", action="store_true",
                    help="(1 GPU) use the upper-triangle product instead of the general GEMV; never the headline")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 PMC passes (roofline.traffic then "
                    "comes from profiles/traffic.json, tagged as such)")
    ap.add_argument("--no-legs", action="store_true", help="N > 1: do not start the child-process legs (the other topology, the experimental exchange)")
    ap.add_argument("--gemv-timing", type=int, default=4, help="time the GEMV (and the exchange) of every T-th iteration with HIP-event pairs")
    ap.add_argument("--ramp", type=float, default=0.6, help="seconds of untimed GEMV launches before the warm-up steps of the headline "
                    "(waits out the driver's wipe of VRAM released by the child processes, see settle); 0 = none")
    ap.add_argument("--config4-n", type=int, default=131072, help="matrix order of the configs[3] GEMV-only side run")
    ap.add_argument("--cpu-sample-n", type=int, default=0, help="matrix order of the CPU baseline sample (0 = the workload's own N: "
                    "no extrapolation; the reference driver needs 8*N^2 bytes of host memory)")
    ap.add_argument("--cpu-sample-iters", type=int, default=20)
    ap.add_argument("--leg-timeout", type=float, default=120.0, help="seconds a child-process leg may take (a healthy one needs 10-40 s at N=65536)")
    ap.add_argument("--headline-timeout", type=float, default=240.0, help="N > 1: seconds this process's own topology may take before rank 0 "
                    "prints the line without it (a healthy one needs 20-60 s at N=65536)")
    ap.add_argument("--mfma-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--leg", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--leg-out", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.mfma_child:
        return mfma_child(args.config4_n)
    if args.leg:
        return leg_main(args)

    # stdout carries ONE line, the JSON: native libraries print there too (RCCL writes a five-line version banner to
    # stdout when a communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the JSON
    # line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # the host driver on these nodes only supports dmabuf IPC; RCCL across processes needs this
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # one node (the contract: N GPUs of ONE node, rendezvous on 127.0.0.1): RCCL's bootstrap sockets go over
    # the loopback interface instead of whatever interface the container happens to expose first
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    lam = importlib.import_module(PKG)
    lam.lib()
    use_dist = lam.launched_with_ranks()
    rdzv = lam.Rendezvous(timeout=max(180.0, 3 * args.leg_timeout + 60.0)) if use_dist else None
    world = rdzv.size if rdzv else 1
    rank = rdzv.rank if rdzv else 0
    profiled = under_profiler()
    n_gpus = world if use_dist else max(1, args.gpus)
    solo = rank == 0 and not use_dist and n_gpus == 1
    n = args.n

    # Children go FIRST: nothing has touched the GPU yet (a process that has initialised the GPU must not fork+exec on this
    # pool), and never under a profiler (its tool library initialises the GPU before main()).
    cb, mfma_rows = None, None
    if solo and not args.no_also and not profiled and not args.symmetric:
        mfma_rows = run_mfma_child(args.config4_n)      # first: its 34 GB are wiped while the CPU baseline runs
    if solo and not args.no_cpu_baseline and not profiled:
        cb = cpu_baseline(args.cpu_sample_n or args.n, args.cpu_sample_iters)
    live_traffic = (None, None)
    if solo and not args.no_traffic and not profiled and not args.symmetric:
        live_traffic = measure_traffic(args.n)
        if live_traffic[0] is None:
            sys.stderr.write(f"[bench] live PMC traffic measurement not available: {live_traffic[1]}\n")

    sampler = DeviceStateSampler() if solo and not profiled and os.environ.get("LAM_BENCH_NO_SAMPLER", "0") in ("", "0") else None
    t_window0 = time.time()

    # N > 1: the legs (see the module docstring).  `other` = the topology this process is NOT; `direct_*` = the EXPERIMENTAL
    # exchange of either topology, each in processes of its own.
    legs = {}
    forced_rccl = use_dist and os.environ.get("LAM_HIP_FORCE_RCCL", "0") not in ("", "0")     # a 1-rank communicator (tests on one GPU)
    want_legs = (n_gpus > 1 or forced_rccl) and not args.no_legs and not profiled and not args.symmetric
    want_direct = want_legs and os.environ.get("LAM_BENCH_DIRECT", "1") != "0"
    if want_legs and not use_dist:
        def rank_env(i):
            return {"RANK": str(i), "WORLD_SIZE": str(n_gpus), "LOCAL_RANK": str(i)}
        legs["rank_mode"] = run_leg(args, "rank_mode", n_gpus, args.leg_timeout, rank_env=rank_env)
        if want_direct:
            legs["rank_direct"] = run_leg(args, "rank_direct", n_gpus, args.leg_timeout, rank_env=rank_env)
            legs["one_direct"] = run_leg(args, "one_direct", 1, args.leg_timeout)
    if want_legs and use_dist:
        # rank 0 alone runs the one-process topology (the other ranks hold no GPU state yet and wait at the barrier below);
        # the direct exchange of the rank mode needs a fresh process PER RANK: every rank starts its own child, and the children
        # find each other through a rendezvous file of their own
        if rank == 0 and n_gpus > 1:
            legs["one_process"] = run_leg(args, "one_process", 1, args.leg_timeout)
            if want_direct:
                legs["one_direct"] = run_leg(args, "one_direct", 1, args.leg_timeout)
        token = rdzv.broadcast(os.urandom(8).hex().encode() if rank == 0 else b"").decode()
        if want_direct:
            my_env = {"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rdzv.local_rank)}
            rec = run_leg(args, "rank_direct", 1, args.leg_timeout, rank_env=lambda i: my_env, rdzv_file=f"/tmp/lam_rdzv.leg.{token}",
                          expect_record=rank == 0)      # the leg's rank 0 writes the record
            errs = [e.decode() for e in rdzv.allgather((rec.get("error") or "").encode())]
            if rank == 0:
                others = "; ".join(f"rank {q}: {e}" for q, e in enumerate(errs) if e and q != 0)
                if others:
                    rec["error"] = (rec.get("error", "") + "; " if rec.get("error") else "") + others[:600]
                legs["rank_direct"] = rec
        rdzv.barrier()

    # ---- this process's own topology ------------------------------------------------------------------------------------
    exchange_modes, effective_exchange, rccl_info, shard_devices, device_state = None, None, None, None, None
    headline_from, own_error, comparison_error = None, None, None
    if use_dist or n_gpus > 1:
        # in a worker thread under a supervisor: an exception, a hang or a tear-down by the launcher must not cost rank 0 its line
        def own(on_headline):
            if use_dist:
                return rank_mode_measure(lam, args, rdzv, "main", on_headline=on_headline)
            return one_process_measure(lam, args, n_gpus, "main", on_headline=on_headline)
        sup = Supervised(own, args.headline_timeout)
        rec, err = sup.run()
        if rec is None and "headline" in sup.box:
            rec, comparison_error = sup.box["headline"], err       # the headline itself had been measured
            sys.stderr.write(f"[bench] the comparison modes behind the headline failed: {err}\n")
        elif rec is None:
            own_error = err
            sys.stderr.write(f"[bench] this process's own topology failed: {err}\n")
            if rank != 0:
                time.sleep(1.5)          # rank 0 prints first (a launcher tears everything down at the first non-zero exit)
                os._exit(4)
            leg = legs.get("one_process" if use_dist else "rank_mode") or {}
            if isinstance(leg.get("gemv_ms"), (int, float)) and isinstance(leg.get("ms_per_step"), (int, float)) and leg.get("gemv_bytes_per_launch"):
                rec = record_from_leg(leg, args.steps)
                headline_from = ("the one-process topology's leg" if use_dist else "the rank mode's leg (RCCL)") + " -- this process's own topology failed"
            else:
                out = {"metric": "cg_iterations_per_sec", "value": None, "unit": "iterations/s", "n_gpus": n_gpus, "steps": args.steps,
                       "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                       "data": "synthetic", "config": {"workload": f"dense SPD CG, N={n} fp64 (BASELINE configs[2])", "n": n},
                       "error": f"own topology failed: {err}; no usable leg of the other topology: {leg.get('error', 'not run')}"[:900]}
                os.write(json_fd, (json.dumps(out) + "\n").encode())
                os._exit(4)
        st, dt, true_res, check, failures = rec["st"], rec["dt"], rec["true_res"], rec["self_check"], rec["failures"]
        kernel_name, cold_start, parallelism = rec["kernel"], rec["cold_start"], rec["parallelism"]
        host_us_per_step, effective_exchange = rec["host_enqueue_us_per_step"], rec["exchange_effective"]
        from_rank_mode = use_dist != (headline_from is not None)          # which topology the headline record describes
        if from_rank_mode:
            exchange_modes = rec["exchange_modes"] if world > 1 or forced_rccl or headline_from else None
            rccl_info = {"rccl_version": rec["rccl_version"], "rccl_ranks": rec["rccl_ranks"], "rccl_calls_enqueued_rank0": rec["rccl_calls_enqueued"]}
        else:
            exchange_modes = rec["exchange_modes"]
            shard_devices = rec["device_ids"]
        s = None
    else:
        s = lam.Solver(lam.F64)
        # t_gemv (roofline.achieved) = average of HIP-event pairs around the GEMV launch of every 4th iteration of the timed steps
        # (library default: every 8th; each record is a marker packet in the stream, ~2 us per iteration at this rate)
        s.set_option("gemv_timing", args.gemv_timing)
        host_ns0 = s.get_option("host_enqueue_ns")

        def barrier():
            pass
        st, dt = run_config(s, n, args.warmup, args.steps, barrier, symmetric=args.symmetric, ramp_s=args.ramp)
        kernel_name = s.gemv_kernel_name()
        cold_start = run_config.cold
        parallelism = "1 GPU"
        # host time the library spent issuing one iteration of the headline configuration (its waits for the device excluded)
        host_us_per_step = (s.get_option("host_enqueue_ns") - host_ns0) * 1e-3 / ((args.warmup + args.steps) * (2 if args.ramp > 0 else 1))
        device_state = sampler.summary(t_window0, time.time()) if sampler is not None else None
        true_res = s.true_residual()
        check, failures = self_check(lam, args, 1, st, true_res, 0, None, symmetric=args.symmetric)

    # (the experimental legs are merged into the exchange modes of the topology they belong to: merge_direct)
    other_topology = None
    if rank == 0 and (n_gpus > 1 or forced_rccl) and headline_from is None:
        if use_dist:
            merge_direct(exchange_modes, legs.get("rank_direct"), true_res)
            other_topology = legs.get("one_process")
            if other_topology is not None:
                merge_direct(other_topology.setdefault("exchange_modes", {}), legs.get("one_direct"), other_topology.get("rel_residual_true"))
        else:
            merge_direct(exchange_modes, legs.get("one_direct"), true_res)
            other_topology = legs.get("rank_mode")
            if other_topology is not None:
                merge_direct(other_topology.setdefault("exchange_modes", {}), legs.get("rank_direct"), other_topology.get("rel_residual_true"))
    elif rank == 0 and headline_from is not None:
        # the headline IS the other topology's leg: its experimental leg goes with it; the failed topology is reported as such
        merge_direct(exchange_modes, legs.get("one_direct" if use_dist else "rank_direct"), true_res)
        other_topology = {"error": own_error}

    # Side measurements, same process, same context, AFTER the headline (N=1 only, not under a profiler):
    # the opt-in symmetric product on the same system, then configs[1] (N=32768) and the sizes the reference
    # published numbers for (TESTS/BEST_RESULTS:362-372), largest first so the matrix allocation is re-used.
    also, sym = None, None
    keep_alive = []
    if solo and not args.no_also and not profiled and not args.symmetric:
        def barrier():
            pass
        try:
            st_s, dt_s = run_config(s, n, args.warmup, args.steps, barrier, symmetric=True, generate=False)
            sym = {"what": "lam_hip_set_option('symmetric', 1): the product reads only the upper triangle of the SPD matrix "
                           "(two-pass, deterministic); same system, same context; NOT the headline (different algorithm, "
                           "single GPU only)", "value": args.steps / dt_s, "ms_per_step": dt_s / args.steps * 1e3,
                   "product_ms": st_s["t_gemv"] * 1e3, "rel_residual_true": s.true_residual()}
            # its own roofline: the triangle's bytes (s*N(N+1)/2 + the vectors) over the product's two launches
            tri_bytes = 8.0 * n * (n + 1) / 2 + 8.0 * 2 * n
            sym["kernel"] = s.gemv_kernel_name()
            sym["product_gbps_on_triangle"] = tri_bytes / st_s["t_gemv"] / 1e9
            sym["roofline_frac_on_triangle"] = sym["product_gbps_on_triangle"] / HBM_PEAK_GBPS
            sym["algorithmic_bytes_per_product"] = tri_bytes
        except Exception as e:   # noqa: BLE001
            sys.stderr.write(f"[bench] symmetric-option side run failed: {e}\n")
        also = []
        for n_also in sorted((x for x in ALSO_SIZES if x != n), reverse=True):
            try:
                st_a, dt_a = run_config(s, n_also, args.warmup, args.steps, barrier)
                ms = dt_a / args.steps * 1e3
                gbps = st_a["gemv_bytes"] / st_a["t_gemv"] / 1e9
                also.append({"n": n_also, "value": args.steps / dt_a, "ms_per_step": ms, "gemv_ms": st_a["t_gemv"] * 1e3,
                             "other_us": (ms - st_a["t_gemv"] * 1e3) * 1e3, "gemv_gbps": gbps, "roofline_frac": gbps / HBM_PEAK_GBPS,
                             "kernel": s.gemv_kernel_name()})
                try:    # the same system with option symmetric (not the headline's algorithm): iterations/s and the product on its own bytes
                    st_y, dt_y = run_config(s, n_also, args.warmup, args.steps, barrier, symmetric=True, generate=False)
                    tri = 8.0 * n_also * (n_also + 1) / 2 + 8.0 * 2 * n_also
                    also[-1]["symmetric_option"] = {"value": args.steps / dt_y, "product_ms": st_y["t_gemv"] * 1e3,
                                                    "roofline_frac_on_triangle": tri / st_y["t_gemv"] / 1e9 / HBM_PEAK_GBPS,
                                                    "speedup": dt_a / dt_y}
                except Exception as e:   # noqa: BLE001
                    also[-1]["symmetric_option"] = {"error": str(e)[:200]}
                finally:
                    s.set_option("symmetric", 0)
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] side run N={n_also} failed: {e}\n")
    # BASELINE configs[3]: N=131072 in fp32 and in bf16 storage (fp32 accumulate), GEMV only -- the production VALU
    # kernel of each dtype and, for bf16, the MFMA-fed variant beside it (kept as an option: it is slower).  Own
    # contexts, after the fp64 context has been closed; yardstick is still HBM GB/s (GEMV has no contraction for MFMA).
    config4 = None
    if solo and not args.no_also and not profiled and not args.symmetric:
        config4 = []
        n4 = args.config4_n
        # NO context is closed before the last timed region: the driver wipes released VRAM in the background, and a
        # GEMV that runs next to a 69 GB wipe measures 3 % low (round 3's bf16 figure, 0.842, was exactly that: it followed
        # the close of the fp32 context; profiles/r04_bf16_gap_probe.txt).  fp64 + fp32 + bf16 = 137 GB of 288 GB.
        for dname, dt4, es4, variants in (("f32", lam.F32, 4, ((-1, "VALU (production)"),)),
                                          ("bf16", lam.BF16, 2, ((-1, "VALU (production)"),) + MFMA_VARIANTS)):
            try:
                s4 = lam.Solver(dt4)
                keep_alive.append(s4)
                s4.generate_random_spd(n4, 1234, 1e4)
                s4.generate_random_rhs(1235)
                s4.cg_init()                              # p = b: a real vector in the GEMV's p replica
                settle(s4, 0.3)
                for v4, what4 in variants:
                    try:
                        s4.set_option("gemv_variant", v4)
                    except Exception:   # noqa: BLE001  (the MFMA shapes live in the tuning build: mfma_child below)
                        continue
                    ts = sorted(s4.gemv_only(10) for _ in range(5))
                    gb = (es4 * float(n4) * n4 + 4.0 * 2 * n4) / 1e9
                    config4.append({"n": n4, "dtype": dname, "path": what4, "kernel": s4.gemv_kernel_name(), "gemv_ms": ts[2] * 1e3,
                                    "gemv_gbps": gb / ts[2], "roofline_frac": gb / ts[2] / HBM_PEAK_GBPS, "algorithmic_bytes": gb * 1e9})
                # the opt-in symmetric product of the same storage type (every pair read once: half the bytes), priced on ITS bytes
                try:
                    s4.set_option("gemv_variant", -1)
                    s4.set_option("symmetric", 1)
                    if s4.get_option("symmetric_effective") == 1:
                        s4.gemv_only(3)
                        ts = sorted(s4.gemv_only(10) for _ in range(5))
                        gb = (es4 * float(n4) * (n4 + 1) / 2 + 4.0 * 2 * n4) / 1e9
                        config4.append({"n": n4, "dtype": dname, "path": "option symmetric (upper-triangle product, two passes; not the GEMV the config names)",
                                        "kernel": s4.gemv_kernel_name(), "gemv_ms": ts[2] * 1e3, "gemv_gbps": gb / ts[2],
                                        "roofline_frac": gb / ts[2] / HBM_PEAK_GBPS, "algorithmic_bytes": gb * 1e9})
                    s4.set_option("symmetric", 0)
                except Exception as e:   # noqa: BLE001
                    sys.stderr.write(f"[bench] configs[3] symmetric run ({dname}) failed: {e}\n")
            except Exception as e:   # noqa: BLE001
                sys.stderr.write(f"[bench] configs[3] run ({dname}) failed: {e}\n")
        if mfma_rows:
            config4.extend(mfma_rows)
    for ctx in keep_alive:
        ctx.close()
    if s is not None:
        s.close()

    ms_per_step = dt / args.steps * 1e3
    gemv_bytes = st["gemv_bytes"]                 # algorithmic bytes of ONE launch on one GPU
    achieved = gemv_bytes / st["t_gemv"] / 1e9 if st["t_gemv"] > 0 else 0.0
    if live_traffic[0] is not None:
        traffic, traffic_src = live_traffic
    else:
        traffic, traffic_src = traffic_record(n, n_gpus)
        if traffic_src is not None and live_traffic[1]:
            traffic_src["live_measurement"] = live_traffic[1]

    out = {
        "metric": "cg_iterations_per_sec", "value": (args.steps / dt) if not failures else None, "unit": "iterations/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"dense SPD CG, N={n} fp64 (BASELINE configs[2]), device-generated random "
                               f"SPD matrix (cond 1e6) + random rhs, {args.steps} fixed iterations",
                   "n": n, "parallelism": parallelism,
                   "untimed_setup": f"matrix generated on the device; {args.ramp} s of GEMV launches before the warm-up steps (waits out "
                                    "the driver's background wipe of the VRAM the child processes released: profiles/r04_bf16_gap_probe.txt; "
                                    "`value_cold` is the same measurement without it -- rounds 1-2 and BASELINE-style figures are cold)",
                   "matrix_bytes_per_gpu": 8.0 * n * n / n_gpus, **({"shard_devices": shard_devices} if shard_devices else {})},
        "gemv_ms": st["t_gemv"] * 1e3,
        **({"exchange_us": st["t_exchange"] * 1e6, "gemv_plus_comm_ms": (st["t_gemv"] + st["t_exchange"]) * 1e3} if n_gpus > 1 or use_dist else {}),
        **({"exchange_us_min_over_ranks": st["t_exchange_min"] * 1e6} if "t_exchange_min" in st else {}),
        **({"exchange_effective": effective_exchange} if effective_exchange is not None else {}),
        "other_us": (ms_per_step - st["t_gemv"] * 1e3) * 1e3,
        "host_enqueue_us_per_step": host_us_per_step,
        "gemv_gbps_per_gpu": achieved,
        "gemv_gbps_aggregate": achieved * n_gpus,
        "rel_residual_recursive": st["rel_err"], "rel_residual_true": true_res, "rccl_init_s": st.get("t_comm_init", 0.0),
        **({"cold_start": cold_start, "value_cold": cold_start["value"]} if cold_start else {}),
        "self_check": check, **({"error": "; ".join(failures), "value_unchecked": args.steps / dt} if failures else {}),
        **({"exchange_modes": exchange_modes} if exchange_modes else {}),
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "timing": f"HIP-event pair around the GEMV launch of every {args.gemv_timing}th iteration of the timed steps, on the launch stream",
                     "algorithmic_bytes_per_launch": gemv_bytes},
        "host_plumbing": {"torch_imported": "torch" in sys.modules,
                          "rendezvous": "package socket rendezvous (_rendezvous.py)" if rdzv else None,
                          "rccl_version": rccl_info["rccl_version"] if rccl_info else None,
                          "rccl_ranks": rccl_info["rccl_ranks"] if rccl_info else None,
                          "rccl_calls_enqueued_rank0": rccl_info["rccl_calls_enqueued_rank0"] if rccl_info else None,
                          "profiler_detected": profiled},
        **({"device_state": device_state} if device_state else {}),
    }
    if n_gpus > 1 and rank == 0:
        # the topology the headline is not: measured by child processes in front of the headline (None: legs were switched off) --
        # or, when the headline had to come from that leg, this process's own topology with the reason it failed
        key = "one_process_topology" if use_dist != (headline_from is not None) else "rank_mode_rccl"
        out[key] = other_topology if other_topology is not None else ({"skipped": "--no-legs, a profiler, or --symmetric"} if not want_legs else {"error": "leg not run"})
    if headline_from is not None:
        out["headline_from"], out["own_topology_error"] = headline_from, own_error
    if comparison_error is not None:
        out["comparison_error"] = comparison_error

    if also:
        out["also"] = also
    if config4:
        out["config4_gemv"] = config4
    if sym is not None:
        sym["speedup_vs_headline"] = sym["value"] / (args.steps / dt)
        out["symmetric_option"] = sym
    if args.symmetric:
        out["config"]["workload"] += "; option symmetric=1 (upper-triangle product)"
        out["roofline"] = None     # the GEMV roofline does not describe this algorithm
    if solo and cb is not None:
        # same unit as `value`, scaled to the workload's N (bytes per iteration scale with N^2)
        if cb["sample_n"] != n:
            cb["value_at_sample_n"] = cb["value"]
            cb["value"] = cb["value"] * (cb["sample_n"] / float(n)) ** 2
            cb["sample"] += f"; value = measured it/s x ({cb['sample_n']}/{n})^2 to the workload's N"
        else:
            cb["sample"] += "; measured at the workload's own N (bounded in iterations, not in size)"
        out["cpu_baseline"] = cb
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if own_error is not None or comparison_error is not None:
        # the worker thread may still sit in a native call (and the other ranks may be gone): no barrier, no destructors
        sys.stderr.write(f"[bench] line printed without a complete run of this process's own topology: {own_error or comparison_error}\n")
        os._exit(4 if own_error is not None else 0)       # comparison modes only: the headline is complete and checked
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()
    if failures:
        sys.stderr.write("[bench] SELF-CHECK FAILED: " + "; ".join(failures) + "\n")
        sys.exit(3)


if __name__ == "__main__":
    main()
