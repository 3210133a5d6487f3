"""GPU tests of the C++ host side: the drop-in classes (LAM/src/HIP/*.hpp) and the drivers that
mirror the reference executables, run as real processes and compared with the reference's own
outputs (tests/golden) -- so these read like running the reference's drivers."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

TEST_DIR = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "test")
RCCL_EXE = os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP_RCCL.out")
ONE_EXE = os.path.join(TEST_DIR, "test_CG_single_GPU.out")
MULTI_EXE = os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP.out")


def _run(cmd, env=None, **kw):
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None)
    if env:
        e.update(env)
    return subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=300, **kw)


def _csv(stdout):
    return stdout.replace("\n", "").strip().split(",")


def test_getopt_driver_generate_mode_matches_reference_csv(golden, tmp_path):
    """Same flags, same CSV columns (+ the NCCL variant's comm-init column), same iters/err."""
    for g in golden["gen_mode"]:
        if g["P"] != 1 or g["n"] > 2048 and "-i" not in g["args"]:
            continue
        r = _run([RCCL_EXE, "-s", str(g["n"]), "-o", str(tmp_path / "sol.bin")] + g["args"])
        assert r.returncode == 0, r.stderr
        f = _csv(r.stdout)
        ref = g["csv"].split(",")
        assert len(f) == len(ref) + 1          # N,P,threads,t_gen,[t_comm_init],t_gemv,t_iter,iters,err,t_total
        assert f[0] == ref[0] and f[1] == "1"
        assert int(f[7]) == g["iters_printed"]
        if g["rel_err_printed"] > 1e-10:
            assert abs(float(f[8]) / g["rel_err_printed"] - 1) < 1e-5
        # the solution file holds x (clean header), not the rhs the reference's MPI class writes
        hdr = np.fromfile(tmp_path / "sol.bin", dtype=np.uint64, count=2)
        assert hdr.tolist() == [g["n"], 1]


def test_getopt_driver_file_mode_against_golden_solution(golden, oracle, tmp_path):
    g = next(x for x in golden["file_mode"] if x["n"] == 256)
    sol = tmp_path / "sol.bin"
    r = _run([RCCL_EXE, "-A", os.path.join(GOLDEN, g["name"] + ".matrix.bin"), "-b",
              os.path.join(GOLDEN, g["name"] + ".rhs.bin"), "-o", str(sol), "-e", repr(g["tol"])])
    assert r.returncode == 0, r.stderr
    f = _csv(r.stdout)
    assert int(f[0]) == g["n"]
    assert abs(int(f[7]) - g["iters_printed"]) <= max(3, 0.02 * g["iters_printed"])
    assert float(f[8]) < g["tol"]
    x = oracle.read_bin(str(sol)).reshape(-1)
    x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) <= 10 * g["tol"]   # 1e-8 at tol 1e-9


@pytest.mark.parametrize("exe,env", [(ONE_EXE, {}), (MULTI_EXE, {"LAM_NUM_SHARDS": "3"}),
                                     # the one-process class on the in-kernel flag exchange (shards share GPU 0 here: on request)
                                     pytest.param(MULTI_EXE, {"LAM_NUM_SHARDS": "4", "LAM_HIP_EXCHANGE": "2", "LAM_HIP_EXPERIMENTAL_DIRECT": "1",
                                                               "LAM_HIP_DIRECT_SAME_DEVICE": "1", "GPU_MAX_HW_QUEUES": "12"}, marks=pytest.mark.slow),
                                     # ... and on the three-join event exchange (the default is gather-Ap)
                                     pytest.param(MULTI_EXE, {"LAM_NUM_SHARDS": "2", "LAM_HIP_EXCHANGE": "0"}, marks=pytest.mark.slow),
                                     # option "symmetric" from the environment (the drivers have no flag for it): one shard -- the
                                     # upper triangle --, and row shards -- cyclic half windows on the gather-Ap exchange
                                     (ONE_EXE, {"LAM_HIP_SYMMETRIC": "2"}), (MULTI_EXE, {"LAM_NUM_SHARDS": "2", "LAM_HIP_SYMMETRIC": "2"})])
def test_positional_drivers(golden, oracle, tmp_path, exe, env):
    """matrix rhs sol max_iters rel_error; prints the reference's 'Converged in K iterations' line."""
    for g in golden["file_mode"]:
        if g["n"] < 16:
            continue
        sol = tmp_path / "sol.bin"
        r = _run([exe, os.path.join(GOLDEN, g["name"] + ".matrix.bin"), os.path.join(GOLDEN, g["name"] + ".rhs.bin"),
                  str(sol), str(g["max_iters"]), repr(g["tol"])], env=env)
        assert r.returncode == 0, r.stderr
        assert "Finished successfully" in r.stdout
        if "LAM_HIP_SYMMETRIC" in env:          # the option must be APPLIED, not only requested
            assert "Option symmetric (LAM_HIP_SYMMETRIC=2): effective" in r.stdout and "refused" not in r.stderr, r.stdout[-600:] + r.stderr[-600:]
        x = oracle.read_bin(str(sol)).reshape(-1)
        x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
        if g["converged"]:
            line = next(l for l in r.stdout.splitlines() if l.startswith("Converged in"))
            k = int(line.split()[2])
            assert abs(k - g["iters_printed"]) <= max(3, 0.02 * g["iters_printed"])
            assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) <= 10 * g["tol"]   # 1e-8 at tol 1e-9
        else:
            assert f"Did not converge in {g['max_iters']} iterations" in r.stdout
            assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < 1e-12


def test_driver_exit_codes(tmp_path, golden):
    """1 = matrix problem, 2 = rhs problem (test_CG_CPU_MPI_OMP.cpp:55-59,72-76); usage -> 1."""
    g = golden["file_mode"][0]
    m = os.path.join(GOLDEN, g["name"] + ".matrix.bin")
    assert _run([ONE_EXE, str(tmp_path / "nope.bin")]).returncode == 1
    other = next(x for x in golden["file_mode"] if x["n"] != g["n"])
    r = _run([ONE_EXE, m, os.path.join(GOLDEN, other["name"] + ".rhs.bin"), str(tmp_path / "s.bin")])
    assert r.returncode == 2 and "does not match" in r.stderr
    assert _run([RCCL_EXE]).returncode == 1
    assert _run([RCCL_EXE, "-s", "64", "-A", "x"]).returncode == 1     # -s and -A are exclusive
    assert _run([RCCL_EXE, "-h"]).returncode == 0


def test_rank_mode_collectives_single_rank(lam, oracle, monkeypatch):
    """LAM_HIP_FORCE_RCCL=1 runs every RCCL call of the one-process-per-GPU path (the 8-byte ncclAllGather
    x2 + ncclAllGather / grouped ncclBroadcast of p per iteration, the gathers of x, the all-reduce of the
    residual check) on a 1-rank communicator of the REAL library.
    The result must be bit-identical to the plain single-shard path (same reduction order)."""
    n = 1000
    rng = np.random.default_rng(3)
    q, _ = np.linalg.qr(rng.uniform(-1, 1, (n, n)))
    A = (q * np.exp(2.0 * rng.uniform(-1, 1, n))) @ q.T
    A = 0.5 * (A + A.T)
    b = rng.uniform(-1, 1, n)
    with lam.Solver(lam.F64) as s:
        s.set_matrix(A); s.set_rhs(b); s.solve(500, 1e-10)
        x0, st0 = s.solution(), s.stats
    monkeypatch.setenv("LAM_HIP_FORCE_RCCL", "1")
    with lam.Solver(lam.F64, rank=0, nranks=1, device_id=0, unique_id=None) as s:
        assert s.get_option("exchange") == 1          # the rank mode's default: gather-Ap (round 5)
        s.set_option("exchange", 0)                   # the sliced-vector exchange sums in the single-shard order: same bits
        s.set_matrix(A); s.set_rhs(b); s.solve(500, 1e-10)
        x1, st1 = s.solution(), s.stats
        assert st1["t_comm_init"] > 0
        res = s.true_residual()
        y = s.gemv(b)
    assert st0["num_iters"] == st1["num_iters"] and st0["rel_err"] == st1["rel_err"]
    assert np.array_equal(x0, x1)
    assert res < 2e-10
    # the single-collective exchange (one ncclAllGather of [Ap | p.Ap] per iteration) through real RCCL
    with lam.Solver(lam.F64, rank=0, nranks=1, device_id=0, unique_id=None) as s:
        s.set_matrix(A); s.set_rhs(b)
        s.set_option("exchange", 1)
        s.solve(500, 1e-10)
        x2, st2 = s.solution(), s.stats
        assert s.true_residual() < 2e-10
    assert abs(st2["num_iters"] - st0["num_iters"]) <= 1
    assert np.linalg.norm(x2 - x0) / np.linalg.norm(x0) < 1e-9
    assert np.max(np.abs(y - oracle.gemv(A, b))) <= 1e-13 * np.max(np.abs(A) @ np.abs(b))


def test_driver_csv_is_clean_with_real_rccl(tmp_path):
    """LAM_HIP_FORCE_RCCL=1: the getopt driver through the REAL librccl (1-rank communicator).  RCCL prints a version
    banner to stdout at communicator creation; the driver's stdout must still be exactly the reference's CSV line."""
    r = _run([RCCL_EXE, "-s", "4096", "-i", "15", "-o", str(tmp_path / "sol.bin")], env={"LAM_HIP_FORCE_RCCL": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    f = lines[0].split(",")
    assert f[0] == "4096" and int(f[7]) == 16 and abs(float(f[8]) / 0.000368282 - 1) < 1e-5 and float(f[4]) > 0   # comm-init column


def test_library_leaves_the_hosts_stdout_alone_by_default(tmp_path):
    """Redirecting file descriptor 1 around ncclCommInitRank is OPT-IN (LAM_HIP_QUIET_RCCL=1, set by this package's
    drivers): a host application that links liblam_hip.so keeps its stdout.  With the variable the banner RCCL
    prints at communicator creation goes to stderr and stdout is exactly what the host wrote; without it the
    library touches nothing (whatever RCCL prints lands between the host's own lines, in order)."""
    code = ("import importlib, sys, os\n"
            f"sys.path.insert(0, {ROOT!r})\n"
            "lam = importlib.import_module('2024-eumaster4hpc-student-challenge_amd')\n"
            "os.write(1, b'HOST-BEFORE\\n')\n"
            "s = lam.Solver(lam.F64, rank=0, nranks=1, device_id=0, unique_id=None)\n"
            "os.write(1, b'HOST-AFTER\\n')\n"
            "s.generate_matrix(256); s.generate_rhs(); s.solve(5, 1e-9); s.close()\n")
    outs = {}
    for quiet in ("1", None):
        env = dict(os.environ, LAM_HIP_FORCE_RCCL="1")
        env.pop("LAM_HIP_QUIET_RCCL", None)
        if quiet:
            env["LAM_HIP_QUIET_RCCL"] = quiet
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[quiet] = r.stdout
    assert outs["1"] == "HOST-BEFORE\nHOST-AFTER\n"
    lines = outs[None].splitlines()
    assert lines[0] == "HOST-BEFORE" and lines[-1] == "HOST-AFTER"       # nothing of the host's was diverted


def test_mpi_bootstrapped_driver_single_rank(tmp_path):
    """Optional build (`make mpi`): MPI_Init + MPI_Bcast of the RCCL id, the reference NCCL variant's
    bootstrap (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:320-327).  One rank under mpiexec."""
    exe = os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP_RCCL_mpi.out")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("MPI driver not built (make mpi)")
    r = _run([mpiexec, "-n", "1", exe, "-s", "4096", "-i", "15", "-o", str(tmp_path / "sol.bin")])
    assert r.returncode == 0, r.stdout + r.stderr
    f = _csv(r.stdout)
    assert f[0] == "4096" and int(f[7]) == 16 and abs(float(f[8]) / 0.000368282 - 1) < 1e-5


def _mock_env(kind, mock_mp_lib, mock_async, tmp_path):
    lib = mock_mp_lib if kind == "mp" else mock_async
    return lib, {"GPU_MAX_HW_QUEUES": "8", "MOCK_RCCL_STATS_FILE": str(tmp_path / "mock_stats.jsonl"),
                 "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def _skip_if_no_ipc(r):
    if "HIP IPC is not available" in r.stderr or "hipIpcOpenMemHandle failed" in r.stderr:
        pytest.skip("HIP IPC between processes is not available on this box")


@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("kind", ["mp", "async"])
def test_mpi_bootstrapped_driver_multi_rank(tmp_path, mock_mp_lib, mock_async, nranks, kind):
    """`mpiexec -n P test_CG_MultiGPUS_HIP_RCCL_mpi.out`: MPI_Init + MPI_Bcast of the unique id + one rank per
    process, the launch of the reference's NCCL variant (NCCL.cu:320-327; TESTS/GPU_SCRIPTS use srun -n).
    All ranks share GPU 0, so librccl is replaced by a test double: "mp" = host-synchronous, "async" =
    stream-ordered with the slot ring shared through HIP IPC.  Generate mode, run to convergence: the
    known answer (2048 iterations at N=4096) and a stop that every rank must act on in the same iteration."""
    exe = os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP_RCCL_mpi.out")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("MPI driver not built (make mpi)")
    lib, env = _mock_env(kind, mock_mp_lib, mock_async, tmp_path)
    genv = ["-genv", "LD_PRELOAD", lib]
    for k, v in env.items():
        genv += ["-genv", k, v]
    sol = tmp_path / "sol.bin"
    r = _run([mpiexec, "-n", str(nranks)] + genv + [exe, "-s", "4096", "-o", str(sol)])
    _skip_if_no_ipc(r)
    assert r.returncode == 0, r.stdout + r.stderr
    f = _csv(r.stdout)
    assert f[0] == "4096" and f[1] == str(nranks) and int(f[7]) == 2048 and float(f[8]) < 1e-9, f
    x = np.fromfile(sol, dtype=np.float64, offset=16)
    A_x = 2 * x; A_x[1:] += x[:-1]; A_x[:-1] += x[1:]                  # tridiag(1,2,1) x
    assert np.linalg.norm(A_x - 1.0) / np.sqrt(4096) < 1e-8
    if kind == "async":
        import json
        lines = [json.loads(l) for l in open(tmp_path / "mock_stats.jsonl")]
        assert len(lines) == nranks and all(l["abort"] == 0 for l in lines) and len({l["calls"] for l in lines}) == 1, lines


@pytest.mark.parametrize("exchange", ["0", "1", "2", "1+symmetric"])
def test_mpi_driver_other_exchanges_across_processes(tmp_path, mock_async, exchange):
    """LAM_HIP_EXCHANGE=1 (one all-gather per iteration) and =2 (direct: p replicas and mailboxes of the other
    PROCESSES mapped through HIP IPC, no collective inside the iteration) under `mpiexec -n 4`, run to
    convergence; set-up collectives go through the stream-ordered test double.  "1+symmetric": exchange 1 with
    LAM_HIP_SYMMETRIC=2 -- the symmetric product on row shards across four processes (each gathers the others' full-length
    contributions), the reference's 2048-iteration known answer of the tridiagonal system all the same."""
    import json
    exe = os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP_RCCL_mpi.out")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        pytest.skip("MPI driver not built (make mpi)")
    env = {"LD_PRELOAD": mock_async, "GPU_MAX_HW_QUEUES": "8", "MOCK_RCCL_STATS_FILE": str(tmp_path / "st.jsonl"),
           "HSA_ENABLE_IPC_MODE_LEGACY": "0", "LAM_HIP_EXCHANGE": exchange[0], "LAM_HIP_EXPERIMENTAL_DIRECT": "1"}
    if exchange.endswith("symmetric"):
        env["LAM_HIP_SYMMETRIC"] = "2"
    genv = []
    for k, v in env.items():
        genv += ["-genv", k, v]
    sol = tmp_path / "sol.bin"
    r = _run([mpiexec, "-n", "4"] + genv + [exe, "-s", "4096", "-o", str(sol)])
    _skip_if_no_ipc(r)
    assert r.returncode == 0, r.stdout + r.stderr
    f = _csv(r.stdout)
    assert f[0] == "4096" and f[1] == "4" and int(f[7]) == 2048 and float(f[8]) < 1e-9, f
    x = np.fromfile(sol, dtype=np.float64, offset=16)
    A_x = 2 * x; A_x[1:] += x[:-1]; A_x[:-1] += x[1:]
    assert np.linalg.norm(A_x - 1.0) / np.sqrt(4096) < 1e-8
    lines = [json.loads(l) for l in open(tmp_path / "st.jsonl")]
    assert len(lines) == 4 and all(l["abort"] == 0 for l in lines) and len({l["calls"] for l in lines}) == 1, lines
    if exchange == "2":
        assert lines[0]["calls"] < 40        # set-up + solution gather only: nothing per iteration


@pytest.mark.parametrize("exe_name", ["test_CG_MultiGPUS_HIP_RCCL.out", "test_CG_MultiGPUS_CUDA_NCCL.out"])
def test_env_launched_driver_multi_rank(tmp_path, mock_mp_lib, golden, oracle, exe_name):
    """No MPI: P processes that only get RANK / WORLD_SIZE / LOCAL_RANK from their launcher (what
    `torchrun --no-python`, srun or mpiexec export) and agree on the unique id through the rendezvous file.
    File mode: every rank reads its own row block; the solution must match the reference's golden x.
    Run under the reference's executable name too (Makefile ALIASES)."""
    import json
    g = next(x for x in golden["file_mode"] if x["n"] == 256)
    P = 3
    sol = tmp_path / "sol.bin"
    procs = []
    for rank in range(P):
        e = dict(os.environ, LD_PRELOAD=mock_mp_lib, RANK=str(rank), WORLD_SIZE=str(P), LOCAL_RANK="0",
                 LAM_RCCL_ID_FILE=str(tmp_path / "id"))
        procs.append(subprocess.Popen([os.path.join(TEST_DIR, exe_name), "-A", os.path.join(GOLDEN, g["name"] + ".matrix.bin"),
                                       "-b", os.path.join(GOLDEN, g["name"] + ".rhs.bin"), "-o", str(sol), "-e", repr(g["tol"])],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), outs
    f = _csv(outs[0][0])
    assert int(f[0]) == g["n"] and f[1] == str(P)
    assert abs(int(f[7]) - g["iters_printed"]) <= max(3, 0.02 * g["iters_printed"]) and float(f[8]) < g["tol"]
    assert all(o[0].strip() == "" for o in outs[1:])                    # only rank 0 prints the CSV
    x = oracle.read_bin(str(sol)).reshape(-1)
    x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) <= 10 * g["tol"]   # 1e-8 at tol 1e-9
    assert not os.path.exists(tmp_path / "id")                          # rank 0 removed the rendezvous file


def test_driver_csv_gemv_column_can_include_the_exchange(tmp_path, mock_mp_lib, golden):
    """-g (or LAM_CSV_GEMV_PLUS_COMM=1 for drivers compiled from the reference's own sources): the CSV's GEMV column is
    t_gemv + t_exchange -- the reference's convention, whose t_gemv brackets broadcast + kernel + gather
    (ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:352-377); without it the column is the GEMV kernel alone.  Two ranks, three ways:
    plain, -g, and the environment variable; -v prints both numbers."""
    g = next(x for x in golden["file_mode"] if x["n"] == 256)

    def launch(extra_args, extra_env):
        procs = []
        for rank in range(2):
            e = dict(os.environ, LD_PRELOAD=mock_mp_lib, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", LAM_RCCL_ID_FILE=str(tmp_path / "id"), **extra_env)
            procs.append(subprocess.Popen([RCCL_EXE, "-A", os.path.join(GOLDEN, g["name"] + ".matrix.bin"), "-b", os.path.join(GOLDEN, g["name"] + ".rhs.bin"),
                                           "-o", str(tmp_path / "sol.bin"), "-e", repr(g["tol"])] + extra_args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        try:
            outs = [p.communicate(timeout=300) for p in procs]
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        assert all(p.returncode == 0 for p in procs), outs
        return outs[0][0]

    plain, plus, env = _csv(launch([], {})), _csv(launch(["-g"], {})), _csv(launch([], {"LAM_CSV_GEMV_PLUS_COMM": "1"}))
    assert len(plain) == len(plus) == len(env) and plain[7] == plus[7] == env[7]           # same columns, same solve
    assert 0 < float(plain[5]) and 0 < float(plus[5]) and 0 < float(env[5])
    verbose = launch(["-v"], {})
    import re
    m = re.search(r"GEMV ([0-9.]+) ms = .* exchange ([0-9.]+) ms", verbose)
    assert m and float(m.group(1)) > 0 and float(m.group(2)) >= 0, verbose      # (> 0 on the stream-ordered double: test_gpu_rank_mock.py)
    # (the runs are separate solves, so the columns are compared through -v's two numbers: gemv + exchange > gemv)


def test_rank_local_load_failure_fails_on_every_rank(tmp_path, mock_mp_lib, golden):
    """One rank cannot read the matrix: the loaders agree across ranks (lam_hip_all_ok), so EVERY rank
    leaves with exit code 1 instead of the others waiting in the first collective of the solve."""
    g = next(x for x in golden["file_mode"] if x["n"] == 128)
    good = os.path.join(GOLDEN, g["name"] + ".matrix.bin")
    # rank 1 is pointed at a truncated copy: same header, the tail of its row block is missing
    bad = tmp_path / "truncated.bin"
    data = open(good, "rb").read()
    bad.write_bytes(data[: len(data) - 4096])
    procs = []
    for rank in range(2):
        e = dict(os.environ, LD_PRELOAD=mock_mp_lib, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                 LAM_RCCL_ID_FILE=str(tmp_path / "id"))
        procs.append(subprocess.Popen([RCCL_EXE, "-A", str(bad) if rank == 1 else good, "-b",
                                       os.path.join(GOLDEN, g["name"] + ".rhs.bin"), "-o", str(tmp_path / "s.bin")],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    try:
        outs = [p.communicate(timeout=60) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert [p.returncode for p in procs] == [1, 1], outs


@pytest.mark.parametrize("prec,tol", [("f32", 2e-4), ("bf16", 2e-4)])
def test_getopt_driver_runtime_precision(tmp_path, prec, tol):
    """-t f32 / bf16: the reference hard-codes <double>; here the class template's float instantiation
    (and bf16 matrix storage) is a run-time choice.  Generate mode known answer: tridiag entries and
    b = 1 are exact in every precision, so 15 iterations still give 1/(15 sqrt(8N)) to fp32 rounding."""
    r = _run([RCCL_EXE, "-s", "4096", "-i", "15", "-t", prec, "-o", str(tmp_path / "sol.bin")])
    assert r.returncode == 0, r.stderr
    f = _csv(r.stdout)
    assert int(f[7]) == 16 and abs(float(f[8]) / 0.000368282 - 1) < tol
    x = np.fromfile(tmp_path / "sol.bin", dtype=np.float32, offset=16)
    assert x.size == 4096 and np.all(np.isfinite(x))
    assert _run([RCCL_EXE, "-s", "64", "-t", "fp8"]).returncode == 1


def test_file_mode_past_2_31_elements(lam, tmp_path):
    """A matrix file with MORE than 2^31 elements (N=46342 fp64: 2 147 580 964 elements, 17.2 GB), the size at
    which the reference's loader breaks: its MPI_File_read count is an `int` (ConjugateGradient_CPU_MPI_OMP.hpp:
    408; the 0.08 s "loads" and -nan results at N=50000 in TESTS/BEST_RESULTS:114,154,375,441).  The system is
    generated on the device, written in the reference's file format, loaded back by the C++ drivers (mmap +
    <= 1 GiB uploads with 64-bit counts) and solved; the solution must equal the in-memory solve bit for bit
    (same kernels, same data), which it cannot if any row block was dropped or wrapped."""
    import shutil
    n = 46342
    assert n * n > 2 ** 31
    need = 8 * n * n + (4 << 30)
    base = os.environ.get("LAM_BIG_TMP", "/tmp")
    if shutil.disk_usage(base).free < need:
        pytest.skip(f"not enough room under {base} for a 17 GB matrix file")
    import tempfile
    work = tempfile.mkdtemp(prefix="lam_big_", dir=base)
    try:
        mat, rhs, sol = (os.path.join(work, f) for f in ("matrix.bin", "rhs.bin", "sol.bin"))
        with lam.Solver(lam.F64) as s:
            s.generate_random_spd(n, 7, 50.0)
            s.generate_random_rhs(8)
            with open(mat, "wb") as f:
                f.write(np.array([n, n], dtype=np.uint64).tobytes())
                step = 2048
                for r0 in range(0, n, step):
                    f.write(s.download_rows(r0, min(step, n - r0)).tobytes())
            # b is not readable through the ABI: b = A x0 for a known x0 instead, set and saved explicitly
            b = s.gemv(np.linspace(-1.0, 1.0, n))
            s.set_rhs(b)
            with open(rhs, "wb") as f:
                f.write(np.array([n, 1], dtype=np.uint64).tobytes())
                f.write(b.tobytes())
            conv = s.solve(300, 1e-10)
            x_mem, it_mem = s.solution(), s.stats["num_iters"]
            assert conv
        assert os.path.getsize(mat) == 16 + 8 * n * n
        import time
        t0 = time.time()
        r = _run([ONE_EXE, mat, rhs, sol, "300", "1e-10"])
        t_one = time.time() - t0
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        line = next(l for l in r.stdout.splitlines() if l.startswith("Converged in"))
        assert int(line.split()[2]) == it_mem
        x = np.fromfile(sol, dtype=np.float64, offset=16)
        assert np.array_equal(x, x_mem)
        assert np.linalg.norm(x - np.linspace(-1.0, 1.0, n)) / np.linalg.norm(x) < 1e-8
        # the getopt driver, 3 row shards on one device (each shard reads its own block of the file)
        r = _run([MULTI_EXE, mat, rhs, sol, "300", "1e-10"], env={"LAM_NUM_SHARDS": "3"})
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        x3 = np.fromfile(sol, dtype=np.float64, offset=16)
        assert np.linalg.norm(x3 - x_mem) / np.linalg.norm(x_mem) < 1e-9
        print(f"17.2 GB file: load + solve + save with test_CG_single_GPU.out took {t_one:.1f} s")
    finally:
        shutil.rmtree(work, ignore_errors=True)


def test_generator_tool_with_the_reference_cli(tmp_path, oracle):
    """apps/random_spd_system.out: the reference generator's command line (`size matrix rhs seed`,
    challenge/main/random_spd_system.cpp:127-196), its prints and exit codes; the system is generated on the
    device and written in the reference's format.  The files must be a symmetric positive definite system that the
    positional driver solves and that the CPU oracle solves to the same solution."""
    gen = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "apps", "random_spd_system.out")
    mat, rhs, sol = (str(tmp_path / f) for f in ("matrix.bin", "rhs.bin", "sol.bin"))
    r = _run([gen, "300", mat, rhs, "42"])
    assert r.returncode == 0 and "Finished successfully" in r.stdout, r.stdout + r.stderr
    assert _run([gen, "0", mat, rhs, "1"]).returncode == 1                       # "Wrong argument value"
    assert _run([gen, "16", str(tmp_path / "no" / "dir.bin"), rhs, "1"]).returncode == 2
    A = oracle.read_bin(mat)
    b = oracle.read_bin(rhs).reshape(-1)
    assert A.shape == (300, 300) and b.shape == (300,)
    assert np.array_equal(A, A.T) and np.linalg.eigvalsh(A).min() > 0
    # THE REFERENCE'S LAW AND ITS RANDOM NUMBERS (random_spd_system.cpp:27-37,83-87,166): eigenvalues exp(3.5 u) with u from
    # srand(seed - 10) / rand(), rhs from srand(seed + 10) / rand() -- drawn here from the same libc
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.rand.restype = ctypes.c_int

    def stream(seed, count):
        libc.srand(ctypes.c_uint(seed & 0xFFFFFFFF))
        return np.array([(2.0 * libc.rand()) / 2147483647 - 1.0 for _ in range(count)])

    eig = np.exp(3.5 * stream(42 - 10, 300))
    assert np.array_equal(b, stream(42 + 10, 300))
    w = np.linalg.eigvalsh(A)
    assert np.max(np.abs(w - np.sort(eig))) <= 1e-12 * eig.max()
    assert w.max() / w.min() > 100                                                # the spectrum spans decades (cond -> e^7 ~ 1.1e3)
    assert np.count_nonzero(A) == A.size                                          # dense
    # the round-1..3 law is still there on request (what bench.py generates in place)
    r3 = _run([gen, "300", str(tmp_path / "m3.bin"), str(tmp_path / "r3.bin"), "42", "dominant"])
    assert r3.returncode == 0 and not np.array_equal(oracle.read_bin(str(tmp_path / "m3.bin")), A)
    r2 = _run([gen, "300", str(tmp_path / "m2.bin"), str(tmp_path / "r2.bin"), "42"])   # seeded: reproducible
    assert r2.returncode == 0 and open(mat, "rb").read() == open(tmp_path / "m2.bin", "rb").read()
    r = _run([ONE_EXE, mat, rhs, sol, "2000", "1e-10"])
    assert r.returncode == 0 and "Converged in" in r.stdout, r.stdout + r.stderr
    x = oracle.read_bin(sol).reshape(-1)
    x_ref, st = oracle.cg_solve(A, b, 2000, 1e-10)
    assert st["converged"] and np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < 1e-7


# ------------------------------------------------------------------------------------------------
# the reference's own parameter grids and published known answers
# ------------------------------------------------------------------------------------------------
SWEEP = os.path.join(ROOT, "tools", "sweep.py")


def test_reference_generate_grid_known_answers(tmp_path):
    """`-s N -i 15` for N = 80000 ... 180000 through the drop-in driver, one process per point like the reference's
    SLURM scripts (TESTS/CPU_SCRIPTS/CPU_8_NODE_gen.sh:24-32), against the `iters, err` columns the reference itself
    printed for the same parameters (tests/golden/reference_gen_grid.json <- TESTS/BEST_RESULTS:173-215): 16 and
    8.33333e-05 ... 5.55555e-05, to the printed digits.  N = 200000 needs 320 GB in fp64 -- more than one MI355X has --
    so that point runs in fp32 storage and is marked; it is not shrunk."""
    # the whole grid takes 45 s (ten processes, 51-259 GB each): the default run keeps its ends -- 80000, 180000 and the fp32 point
    # 200000 --; LAM_RUN_SLOW=1 runs every published size and the 1000-iteration point
    full = os.environ.get("LAM_RUN_SLOW", "0") not in ("", "0")
    want = [80000, 90000, 100000, 110000, 120000, 140000, 160000, 180000, 200000] if full else [80000, 180000, 200000]
    js = tmp_path / "gen.json"
    r = subprocess.run([sys.executable, SWEEP, "--grid", "gen", "--json", str(js), "--csv", str(tmp_path / "gen.csv")]
                       + ([] if full else ["--gen-sizes", ",".join(map(str, want)), "--no-gen-extra"]), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    import json
    recs = json.load(open(js))
    gold_all = json.load(open(os.path.join(GOLDEN, "reference_gen_grid.json")))
    gold = [e for e in gold_all["entries"] if e["n"] in want]
    if full:
        # one more published line: `-s 80000 -i 1000` of the GPU weak-scaling series printed 1001, 1.25e-06
        # (TESTS/results/WEAK_SCALABILITY_GPU_MPI.txt:20) -- 1000 iterations of a 51 GB GEMV, 7.3 s on one MI355X
        extra, recs_extra = gold_all["entries_extra"], recs[len(gold):]
        assert len(extra) == len(recs_extra) == 1 and recs_extra[0]["n"] == 80000
        assert recs_extra[0]["match"] and recs_extra[0]["iters"] == 1001 and recs_extra[0]["rel_diff"] <= 2e-6, recs_extra
    recs = recs[:len(gold)]
    assert [x["n"] for x in recs] == [e["n"] for e in gold] == want
    for x, e in zip(recs, gold):
        assert x["match"] and x["iters"] == 16 == e["iters_printed"], x
        assert x["precision"] == ("f64" if x["n"] <= 180000 else "f32")
        assert x["rel_diff"] <= (2e-6 if x["precision"] == "f64" else 1e-5), x
        assert len(x["csv"].split(",")) == 10 and x["csv"].split(",")[0] == str(x["n"])
    assert "does not" not in recs[0].get("note", "") and "f32" in recs[-1]["note"]
    # most points agree with the reference's printed digits character for character
    assert sum(x["same_printed_digits"] for x in recs[:-1]) >= (6 if full else 1)


@pytest.mark.parametrize("symmetric", [0, pytest.param(1, marks=pytest.mark.slow)])
def test_reference_file_grid_sizes(tmp_path, symmetric):
    """(symmetric = 1: the same grid with LAM_HIP_SYMMETRIC=1 in the drivers' environment -- the generator's matrices are
    symmetric bit for bit, every pair is read once, and the reference's published iteration counts must come out all the same.)
    The file-mode sizes of TESTS/GPU_SCRIPTS/GPU_1_NODE.sh:41-47 (10000 ... 70000, default tolerance 1e-9) on systems drawn
    from the reference GENERATOR's law (round 4: apps/random_spd_system.out and the driver's -R option reproduce it -- spectrum
    exp(3.5 U[-1,1]), rhs U[-1,1], the reference's srand/rand streams): the reference's own runs on its generator's matrices
    took 358-360 iterations at every size (tests/golden/reference_file_grid.json <- TESTS/BEST_RESULTS:93-135), and so must
    these, to 3 % -- the first FILE-mode known answer of the reference the package is held to.  The smallest size goes through
    real files written by the generator tool and the -A/-b loaders."""
    js = tmp_path / "file.json"
    env = dict(os.environ, LAM_HIP_SYMMETRIC="1") if symmetric else dict(os.environ)
    r = subprocess.run([sys.executable, SWEEP, "--grid", "file", "--json", str(js), "--files", str(tmp_path), "--files-max-n", "10000"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    import json
    recs = json.load(open(js))
    assert [x["n"] for x in recs] == [10000, 20000, 30000, 40000, 50000, 60000, 70000]
    assert "file mode" in recs[0]["mode"] and all("built on the device" in x["mode"] for x in recs[1:])
    for x in recs:
        f = x["csv"].split(",")
        assert x["match"] and len(f) == 10 and f[0] == str(x["n"]) and f[1] == "1" and float(f[8]) < 1e-9 and float(f[5]) > 0
        assert 0.97 * 358 <= x["iters"] <= 1.03 * 360, x
        if symmetric:                            # applied, not only requested (the driver says so on stderr)
            assert x.get("symmetric") == "effective", x

def test_reference_scaling_grids_at_reduced_size(tmp_path, mock_async):
    """tools/sweep.py --grid scaling: the reference's strong-scaling (N = 20000 / 40000 / 50000 at P = 1, 2, 3, 4, 6, 8 --
    TESTS/results/STRONG_SCALABILITY_GPU_MPI.txt:15-43; P = 3 and 6 are its uneven partition) and weak-scaling series
    (WEAK_SCALABILITY_GPU_MPI.txt:15-17) at a tenth of the sizes on ONE device: through the one-process driver (`-P shards`, all
    shards on GPU 0) for every P, and through `mpiexec -n P` on the MPI-bootstrapped driver (ranks on the stream-ordered RCCL
    double) up to 4 ranks.  Every line must land on the reference's iteration count for its generator's law (358-360, +-3 %) at
    tolerance 1e-9; speed-ups are reported next to the reference's published ones, not gated (one device).  The full command for
    an 8-GPU node is in README.md."""
    import json
    js, csv = tmp_path / "scaling.json", tmp_path / "scaling.csv"
    have_mpi = os.path.exists(os.path.join(TEST_DIR, "test_CG_MultiGPUS_HIP_RCCL_mpi.out")) and os.path.exists("/opt/conda/bin/mpiexec")
    cmd = [sys.executable, SWEEP, "--grid", "scaling", "--scale", "0.1", "--json", str(js), "--csv", str(csv), "--launcher", "both" if have_mpi else "one-process",
           "--preload", mock_async, "--max-ranks", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, LAM_HIP_QUIET_RCCL="1"))
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    recs = json.load(open(js))
    one = [x for x in recs if x["topology"] == "one-process"]
    assert [(x["n_reference"], x["procs"]) for x in one if x["grid"] == "strong"] == [(n, p) for n in (20000, 40000, 50000) for p in (1, 2, 3, 4, 6, 8)]
    assert [(x["n_reference"], x["procs"]) for x in one if x["grid"] == "weak"] == [(10000, 1), (20000, 4), (40000, 8)]
    assert all(x["match"] and (347 if x["n"] >= 4000 else 333) <= x["iters"] <= 371 and float(x["err"]) < 1e-9 for x in recs), [x for x in recs if not x["match"]]
    assert all(x["speedup_iter"] > 0 and x["reference"]["speedup_iter_vs_p1"] > 0 for x in one if x["grid"] == "strong")
    if have_mpi:
        mpi = [x for x in recs if x["topology"] == "mpiexec"]
        assert {x["procs"] for x in mpi} == {1, 2, 3, 4} and len(mpi) == 3 * 4 + 2
        # same systems, same partition: both topologies land on the same count to the summation order (+-3)
        for x in mpi:
            y = next(z for z in one if (z["grid"], z["n"], z["procs"]) == (x["grid"], x["n"], x["procs"]))
            assert abs(x["iters"] - y["iters"]) <= 3, (x, y)
    assert open(csv).read().splitlines()[0].startswith("topology,N,procs,")


def test_getopt_driver_one_process_shards_option(tmp_path):
    """-P <shards> (round 5): the getopt driver as ONE process driving P row shards (the reference's test_CG_MultiGPUS_CUDA topology
    with this driver's flags and CSV): the generate-mode known answer (N = 4096: 2048 iterations) with the procs column = P, for an
    even and an uneven split; refused for bf16 storage, for shard counts outside 1 ... 64 and under a multi-rank launcher."""
    for P in (2, 3, 7):
        r = _run([RCCL_EXE, "-s", "4096", "-o", str(tmp_path / "sol.bin"), "-P", str(P), "-g"])
        assert r.returncode == 0, r.stderr
        f = _csv(r.stdout)
        assert len(f) == 10 and f[0] == "4096" and f[1] == str(P) and int(f[7]) == 2048 and float(f[8]) < 1e-9 and float(f[4]) == 0.0, f
        x = np.fromfile(tmp_path / "sol.bin", dtype=np.float64, offset=16)
        A_x = 2 * x; A_x[1:] += x[:-1]; A_x[:-1] += x[1:]
        assert np.linalg.norm(A_x - 1.0) / np.sqrt(4096) < 1e-8
    for bad in (["-P", "2", "-t", "bf16"], ["-P", "65"], ["-P", "-1"]):
        r = _run([RCCL_EXE, "-s", "1024", "-o", str(tmp_path / "sol.bin")] + bad)
        assert r.returncode == 1 and "Option -P" in r.stderr, (bad, r.stderr)
    r = _run([RCCL_EXE, "-s", "1024", "-o", str(tmp_path / "sol.bin"), "-P", "2"], env=dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0",
                                                                                             LAM_RCCL_ID_FILE=str(tmp_path / "id")))
    assert r.returncode == 1 and "Option -P" in r.stderr, r.stderr
