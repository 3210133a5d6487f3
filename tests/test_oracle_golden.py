"""Pin the CPU oracle (oracle/cg_oracle.c) against outputs of the reference itself.

Fixtures in tests/golden/ were produced by tests/golden/make_golden.py running the reference's
two CPU drivers (built from /root/reference by oracle/Makefile) at OMP_NUM_THREADS=1.
Bar: the single-thread oracle is BIT-IDENTICAL to the reference on the solution vector, and
reproduces the printed iteration count and the printed residual to its printed precision.
"""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _printed_equal(value, printed, sig):
    """`printed` is `value` as the reference printed it with `sig` significant digits."""
    if printed == 0.0:
        return value == 0.0
    return abs(value - printed) <= 0.5000001 * 10 ** (math.floor(math.log10(abs(printed))) - sig + 1)


def test_file_mode_bit_identical(oracle, golden):
    assert golden["file_mode"]
    for g in golden["file_mode"]:
        A = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".matrix.bin"))
        b = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".rhs.bin"))
        x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
        x, st = oracle.cg_solve(A, b, g["max_iters"], g["tol"], threads=1)
        assert st["converged"] == g["converged"], g["tag"]
        # reference prints num_iters when converged, max_iters otherwise (CPU_OMP.hpp:83,88)
        printed_iters = st["num_iters"] if st["converged"] else g["max_iters"]
        assert printed_iters == g["iters_printed"], g["tag"]
        assert _printed_equal(st["rel_err"], g["rel_err_printed"], 7), (g["tag"], st["rel_err"])  # %e
        assert np.array_equal(x.view(np.uint64), x_ref.view(np.uint64)), g["tag"]  # bit-exact


def test_file_mode_float_bit_identical(oracle, golden):
    """The fp32 oracle (oracle_cg_solve_f32: the same generic body as the fp64 one) against the REFERENCE's own class
    instantiated with float (oracle/ref_float_harness.cpp -> ConjugateGradient_CPU_OMP<float>, CPU_OMP.hpp:49-91; the
    reference instantiates float for its GPU classes, ...CUDA_NCCL.cu:767): same iteration counts, printed residuals and
    every bit of the solution vector.  This is what pins the oracle the fp32 / bf16 GPU parity tests compare with."""
    assert len(golden["file_mode_f32"]) >= 4
    for g in golden["file_mode_f32"]:
        A = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".f32.matrix.bin"), dtype=np.float32)
        b = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".f32.rhs.bin"), dtype=np.float32)
        x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin"), dtype=np.float32).reshape(-1)
        x, st = oracle.cg_solve(A, b, g["max_iters"], g["tol"], threads=1)
        assert x.dtype == np.float32
        assert st["converged"] == g["converged"], g["tag"]
        printed_iters = st["num_iters"] if st["converged"] else g["max_iters"]
        assert printed_iters == g["iters_printed"], (g["tag"], st)
        assert _printed_equal(st["rel_err"], g["rel_err_printed"], 7), (g["tag"], st["rel_err"])
        assert np.array_equal(x.view(np.uint32), x_ref.view(np.uint32)), g["tag"]  # bit-exact


def test_solution_file_header_quirk(oracle, golden):
    """The reference writes `int num_cols=1` through sizeof(size_t) (CPU_OMP.hpp:208-210): the low
    32 bits of the cols word are 1; readers must mask."""
    g = golden["file_mode"][0]
    raw = np.fromfile(os.path.join(GOLDEN, g["tag"] + ".sol.bin"), dtype=np.uint64, count=2)
    assert int(raw[0]) == g["n"]
    assert int(raw[1]) & 0xFFFFFFFF == 1


def test_gen_mode_matches_reference_csv(oracle, golden):
    for g in golden["gen_mode"]:
        n, P = g["n"], g["P"]
        if n * n * g["iters_printed"] > 5e9 and not os.environ.get("ORACLE_SLOW"):
            continue   # N=4096 to convergence takes ~30 s single-threaded; run with ORACLE_SLOW=1
        max_iters, tol = 10000, 1e-9   # driver defaults, test_CG_CPU_MPI_OMP.cpp:22-23
        a = g["args"]
        if "-i" in a:
            max_iters = int(a[a.index("-i") + 1])
        if "-e" in a:
            tol = float(a[a.index("-e") + 1])
        A = oracle.tridiag(n)
        b = np.ones(n)
        x, st = oracle.cg_solve(A, b, max_iters, tol, P=P)
        assert st["num_iters"] == g["iters_printed"], g
        if g["rel_err_printed"] > 1e-10:
            # far from convergence the value is insensitive to the reduction tree
            assert _printed_equal(st["rel_err"], g["rel_err_printed"], 6), (g, st["rel_err"])  # cout, 6 sig
        else:
            # at the terminal iteration rr is rounding noise: only its size is meaningful
            assert st["rel_err"] < 1e-9


def test_gen_mode_P1_bit_level(oracle, golden):
    """With one rank the emulated-MPI recurrence and the single-process one coincide exactly."""
    A = oracle.tridiag(513)
    b = np.ones(513)
    x1, s1 = oracle.cg_solve(A, b, 10000, 1e-9)
    x2, s2 = oracle.cg_solve(A, b, 10000, 1e-9, P=1)
    assert s1["num_iters"] == s2["num_iters"]
    assert np.array_equal(x1, x2)


@pytest.mark.parametrize("n,k", [(4096, 15), (4096, 1000), (32768, 200), (65536, 200), (80000, 15),
                                 (560000, 10)])
def test_closed_form_known_answers(n, k):
    """Known-answer table embedded in the reference's own result files: after k < ceil(N/2)
    iterations of generate mode the printed error is 1/(k*sqrt(8N)).
    /root/reference/TESTS/results/MERGE_CPU_MPI_OMP_FP_gen.txt:1 -> 80000 ... 16, 8.33333e-05
    /root/reference/TESTS/results/STRESS_TEST_GPU_MPI.txt:17 -> 560000 ... 11, 4.72456e-05"""
    table = {(80000, 15): 8.33333e-05, (560000, 10): 4.72456e-05, (4096, 15): 0.000368282,
             (4096, 1000): 5.52427e-06}
    val = 1.0 / (k * math.sqrt(8.0 * n))
    if (n, k) in table:
        # the closed form is asymptotic in N: good to ~1e-5 relative at N=4096, exact digits at 80000
        assert abs(val / table[(n, k)] - 1.0) < 2e-5


def test_closed_form_vs_oracle(oracle):
    n = 2048
    A = oracle.tridiag(n)
    for k in (1, 7, 100):
        _, st = oracle.cg_solve(A, np.ones(n), k, 1e-9)
        assert st["num_iters"] == k + 1
        assert abs(st["rel_err"] * k * math.sqrt(8.0 * n) - 1.0) < 5e-4   # asymptotic formula


def test_ops_against_numpy(oracle):
    rng = np.random.default_rng(0)
    A = rng.standard_normal((37, 53))
    x = rng.standard_normal(53)
    y = rng.standard_normal(37)
    np.testing.assert_allclose(oracle.gemv(A, x, 1.5, 0.5, y), 1.5 * A @ x + 0.5 * y, rtol=1e-13)
    np.testing.assert_allclose(oracle.dot(x, x), x @ x, rtol=1e-14)
    np.testing.assert_allclose(oracle.axpby(2.0, x, -1.0, x), x, rtol=1e-15)
    # threaded variants agree to rounding
    np.testing.assert_allclose(oracle.gemv(A, x, threads=4), A @ x, rtol=1e-13)
    assert abs(oracle.dot(x, x, threads=4) - x @ x) < 1e-12


def test_partition_matches_reference_rule(oracle):
    # CPU_MPI_OMP.hpp:176-184: n/P rows each, remainder on the LAST rank
    assert [oracle.partition(1001, 4, q) for q in range(4)] == [(0, 250), (250, 250), (500, 250), (750, 251)]
    assert oracle.partition(65536, 8, 3) == (3 * 8192, 8192)
