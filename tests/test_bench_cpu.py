"""bench.py's host logic without a GPU: the child-process legs (round 5).  An N > 1 line measures the topology it is not -- and
every EXPERIMENTAL exchange -- in child processes started before the parent touches the GPU; whatever happens to a leg (a non-zero
exit, a signal, a hang, one bad rank of several, no record) must come back as `{"error": ...}` and never cost the caller its own
line.  run_leg is driven here with bench.py's GPU-free self-test legs; the records' arithmetic (mode_record, public, the merge of an
experimental leg into its topology's exchange modes) with synthetic numbers.  The same paths with real measurements:
tests/test_gpu_rank_mock.py."""
import argparse
import importlib.util
import os
import time

from conftest import ROOT

spec = importlib.util.spec_from_file_location("lam_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

ARGS = argparse.Namespace(gpus=2, steps=20, warmup=3, n=8192, gemv_timing=4)


def test_a_leg_hands_its_record_back_through_a_file():
    rec = bench.run_leg(ARGS, "_selftest_ok", 1, timeout=60)
    assert rec["value"] == 1.0 and rec["world"] == 1 and "error" not in rec and rec["leg_wall_s"] < 30


def test_a_multi_rank_leg_gets_ranks_and_a_rendezvous_file_of_its_own(monkeypatch):
    monkeypatch.setenv("RANK", "5")                 # the caller's own launcher variables must not leak into a leg
    monkeypatch.setenv("WORLD_SIZE", "9")
    rec = bench.run_leg(ARGS, "_selftest_ok", 3, timeout=60, rank_env=lambda i: {"RANK": str(i), "WORLD_SIZE": "3", "LOCAL_RANK": str(i)})
    assert rec["value"] == 1.0 and rec["world"] == 3 and "error" not in rec and rec["rdzv_file"].endswith("/rdzv")
    rec = bench.run_leg(ARGS, "_selftest_ok", 1, timeout=60)
    assert rec["world"] == 1 and rec["rdzv_file"] is None


def test_a_failing_leg_becomes_an_error_record():
    rec = bench.run_leg(ARGS, "_selftest_fail", 1, timeout=60)
    assert "value" not in rec and "exited with code 7" in rec["error"] and "fails on purpose" in rec["error"]
    # one bad rank of three: the record rank 0 wrote survives, flagged
    rec = bench.run_leg(ARGS, "_selftest_rank1_fails", 3, timeout=60, rank_env=lambda i: {"RANK": str(i), "WORLD_SIZE": "3", "LOCAL_RANK": str(i)})
    assert rec["value"] == 1.0 and "process 1 exited with code 7" in rec["error"]
    # a caller whose processes are not the ones that write the record (rank > 0 under a launcher) gets an empty record, not an error
    rec = bench.run_leg(ARGS, "_selftest_ok", 1, timeout=60, rank_env=lambda i: {"RANK": "2", "WORLD_SIZE": "3"}, expect_record=False)
    assert "error" not in rec and "value" not in rec


def test_a_hanging_leg_is_killed_at_its_timeout():
    t0 = time.time()
    rec = bench.run_leg(ARGS, "_selftest_hang", 2, timeout=3, rank_env=lambda i: {"RANK": str(i), "WORLD_SIZE": "2"})
    assert time.time() - t0 < 30 and "did not finish within 3 s" in rec["error"] and "value" not in rec


def test_records_and_the_merge_of_an_experimental_leg():
    st = {"t_gemv": 4.0e-3, "t_exchange": 20e-6, "rel_err": 0.5, "t_exchange_min": 12e-6}
    m = bench.mode_record(20, 0.1, st, 0.5, experimental=True)
    assert m["value"] == 200.0 and abs(m["ms_per_step"] - 5.0) < 1e-12 and abs(m["exchange_us"] - 20.0) < 1e-9 and abs(m["exchange_us_min_over_ranks"] - 12.0) < 1e-9
    assert abs(m["gemv_plus_comm_ms"] - 4.02) < 1e-9 and abs(m["other_us"] - 980.0) < 1e-6 and m["experimental"] is True
    # a topology record as a leg hands it back: the raw timers stay inside, a failed self-check nulls the value
    rec = {"dt": 0.1, "st": st, "true_res": 0.5, "failures": [], "steps": 20, "self_check": {"passed": True}, "exchange_modes": {"default": "x"}, "cold_start": None}
    pub = bench.public(rec)
    assert pub["value"] == 200.0 and "st" not in pub and "dt" not in pub and pub["self_check"]["passed"]
    bad = bench.public(dict(rec, failures=["residual differs"]))
    assert bad["value"] is None and bad["value_unchecked"] == 200.0 and "residual differs" in bad["error"]


def test_merge_of_an_experimental_leg_into_its_topology():
    modes = {"default": "a", "a": {"value": 100.0, "rel_residual_true": 0.5}}
    bench.merge_direct(modes, {"exchange_modes": {"direct": {"value": 110.0, "rel_residual_true": 0.5 * (1 + 1e-9)}}}, 0.5)
    assert modes["direct"]["value"] == 110.0 and "error" not in modes["direct"]
    bench.merge_direct(modes, {"exchange_modes": {"direct2": {"value": 120.0, "rel_residual_true": 0.7}}}, 0.5)
    assert "WRONG RESULT" in modes["direct2"]["error"]                       # another residual than the default exchange's
    bench.merge_direct(modes, {"error": "process 0 exited with signal 9", "leg_wall_s": 1.0}, 0.5)
    assert modes["direct (experimental leg)"] == {"error": "process 0 exited with signal 9", "experimental": True}
    assert modes["a"]["value"] == 100.0                                      # the headline's record is untouched
    bench.merge_direct(None, {"error": "x"}, 0.5)                            # no modes (one GPU) / no leg: nothing to do
    bench.merge_direct(modes, None, 0.5)


def test_supervised_own_topology_and_the_fallback_record():
    """N > 1: the process's own topology runs in a worker thread; the supervisor hands back its record, or -- when it raises,
    hangs past the timeout or the launcher sends SIGTERM -- the reason, with whatever headline the worker had already published."""
    import signal
    import threading
    rec, err = bench.Supervised(lambda pub: {"value": 3.0}, 5.0).run()
    assert rec == {"value": 3.0} and err is None

    def raises(pub):
        raise RuntimeError("no peer access")
    sup = bench.Supervised(raises, 5.0)
    rec, err = sup.run()
    assert rec is None and err == "RuntimeError: no peer access" and "headline" not in sup.box

    def hangs_late(pub):
        pub({"value": 7.0})
        time.sleep(30)
    sup = bench.Supervised(hangs_late, 1.0)
    t0 = time.time()
    rec, err = sup.run()
    assert rec is None and "--headline-timeout" in err and sup.box["headline"] == {"value": 7.0} and time.time() - t0 < 5

    sup = bench.Supervised(lambda pub: time.sleep(30), 20.0)
    threading.Timer(0.5, lambda: os.kill(os.getpid(), signal.SIGTERM)).start()
    t0 = time.time()
    rec, err = sup.run()
    assert rec is None and "SIGTERM" in err and time.time() - t0 < 5
    signal.signal(signal.SIGTERM, signal.SIG_DFL)

    # the other topology's leg record (what public() writes) as the fallback headline: same arithmetic as a measured one
    st = {"t_gemv": 0.6e-3, "t_exchange": 20e-6, "t_exchange_min": 12e-6, "gemv_bytes": 4.3e9, "rel_err": 1e-3, "t_comm_init": 2.5}
    leg = bench.public({"st": st, "dt": 0.013, "true_res": 1.0000001e-3, "steps": 20, "failures": [], "kernel": "k", "parallelism": "p", "self_check": {"passed": True},
                        "exchange_effective": 1, "host_enqueue_us_per_step": 9.0, "exchange_modes": {"default": "x"}, "rccl_ranks": 8, "rccl_version": "2.x",
                        "rccl_calls_enqueued": 99, "rccl_init_s": 2.5, "n_gpus": 8})
    assert leg["gemv_bytes_per_launch"] == 4.3e9
    back = bench.record_from_leg(leg, 20)
    assert abs(back["dt"] - 0.013) < 1e-12 and back["failures"] == [] and back["true_res"] == 1.0000001e-3 and back["rccl_ranks"] == 8
    for k in ("t_gemv", "t_exchange", "t_exchange_min", "gemv_bytes", "rel_err", "t_comm_init"):
        assert abs(back["st"][k] - st[k]) <= 1e-12 * abs(st[k]), k
    bad = bench.record_from_leg(dict(leg, error="process 3 exited with signal 9"), 20)
    assert bad["failures"] == ["process 3 exited with signal 9"]
