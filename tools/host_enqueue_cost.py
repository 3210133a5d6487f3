#!/usr/bin/env python3
"""Host cost of ONE CG iteration in the one-process, P-shard mode (lam_hip_create with n_shards = P).

All P shards sit on device 0 and N is small (default 4096: the whole matrix is 134 MB, ~20 us of GEMV), so the wall
time per iteration is what the HOST needs to enqueue it -- the number that has to stay well below a shard's GEMV on a
real 8-GPU node (0.61 ms at N=65536, P=8).  Compares round 2's loop (one thread, every stream waits for every other stream: exchange_hub 0), one enqueue
thread per shard (host_threads 1, the shape of the reference's OpenMP-thread-per-device loop,
ConjugateGradient_MultiGPUS_CUDA.cu:337-378) and the hub (exchange_hub 1: the streams meet at one join event per
exchange), prints the runtime calls per iteration, and checks that all give the same bits.

    usage: host_enqueue_cost.py [N] [iters] [P ...]
"""
import importlib
import os
import sys

import numpy as np

# all shards share one device here: give every stream a hardware queue of its own (the default is 4 per device, and a
# stream wait parked in a shared queue holds back the other streams mapped onto it)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
os.environ.setdefault("LAM_HIP_DIRECT_SAME_DEVICE", "1")      # exchange 2 with all shards on one device (one hardware queue per stream above)

# the experiments this tool measures (host_threads / exchange_hub / persistent / finalize = 0) live in the tuning build of the library
os.environ.setdefault("LAM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2024-eumaster4hpc-student-challenge_amd", "liblam_hip_tuning.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")

CALLS = ("launch", "record", "wait", "setdevice")


def calls(s):
    return {k: s.get_option("hip_calls_" + k) for k in CALLS}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    shard_counts = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
    print(f"# N={n} fp64, {iters} iterations per run, all shards on device 0, GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}")
    print("# HOST = time the enqueueing thread spends issuing one iteration (lam_hip option host_enqueue_ns; waits for the device excluded);")
    print("# wall = max(host, device) per iteration -- with 8 streams on ONE device the device side is mostly event hops between streams")
    for P in shard_counts:
        ref = None
        with lam.Solver(lam.F64, device_ids=[0] * P) as s:
            s.generate_random_spd(n, 11, 1e6)
            s.generate_random_rhs(12)
            tuning = s.get_option("tuning_variants") == 1       # host_threads / exchange_hub exist in the tuning build only
            for threads, hub, timing, exchange, overlap in ((0, 0, 1, 0, 1), (0, 0, 0, 0, 1), (1, 0, 0, 0, 1), (0, 1, 0, 0, 1), (1, 1, 0, 0, 1),
                                                            (0, 0, 0, 1, 1), (0, 0, 0, 1, 0), (0, 0, 0, 2, 1), (0, 0, 0, 2, 0)):
                if P == 1 and (threads == 1 or hub == 1 or exchange != 0):
                    continue
                if (threads or hub) and not tuning:
                    continue
                if exchange == 1 and n % P != 0:
                    continue
                if tuning:
                    s.set_option("host_threads", threads)
                    s.set_option("exchange_hub", hub)
                s.set_option("gemv_timing", timing)
                s.set_option("exchange", exchange)
                if exchange == 1:
                    s.set_option("exchange_join", overlap)      # gather-Ap: 1 = join through shard 0's stream, 0 = all-to-all waits
                else:
                    s.set_option("overlap", overlap)
                best, per = None, None
                for _ in range(3):
                    s.cg_init()
                    s.cg_iterate(20, 0.0)
                    c0, h0, u0 = calls(s), s.get_option("host_enqueue_ns"), s.get_option("host_cpu_ns")
                    st = s.cg_iterate(iters, 0.0)
                    c1, h1, u1 = calls(s), s.get_option("host_enqueue_ns"), s.get_option("host_cpu_ns")
                    if best is None or st["t_iter"] < best:
                        best = st["t_iter"]
                        per = {k: (c1[k] - c0[k]) / iters for k in CALLS}
                        host = (h1 - h0) / iters * 1e-3
                        cpu = (u1 - u0) / iters * 1e-3
                x = s.solution()
                if ref is None:
                    ref = x
                # the own-slice GEMV panel of "exchange 2 split" adds a row's products in another order, gather-Ap sums r.r over
                # full-length partials: equal to rounding
                same = bool(np.array_equal(x, ref)) or ((exchange == 1 or (exchange == 2 and overlap == 1)) and np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref))
                kind = {0: "", 1: " gather-Ap join-via-shard0" if overlap else " gather-Ap all-to-all", 2: " split" if overlap else " nosplit"}[exchange]
                print(f"P={P} exchange={s.get_option('exchange_effective')}{kind} host_threads={threads} exchange_hub={hub} gemv_timing={timing}: HOST {host:7.1f} us/iteration to enqueue, wall {best*1e6:7.1f} us/iteration, thread CPU {cpu:7.1f} us/iteration   calls/iteration: "
                      + " ".join(f"{k}={per[k]:.1f}" for k in CALLS) + f"   same result as first variant: {same}", flush=True)
                assert same


if __name__ == "__main__":
    main()
