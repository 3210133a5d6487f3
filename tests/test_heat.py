"""BASELINE configs[4]: the sibling task's steady 2-D heat problem written as a dense SPD system
(apps/heat_system.out) and solved by CG in file mode.

The reference contains no assembler (heat_equation-main/src/heat_equation.cpp is a Jacobi sweep), so
the fixtures are: the reference Jacobi fields on small grids (loose: its stop rule is max_diff<1e-3,
which leaves it up to ~0.2 degrees from the discrete solution) and the reference CG's solution of the
assembled 12x12 system (tests/golden/make_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

ASM = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "apps", "heat_system.out")
RCCL_EXE = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "test", "test_CG_MultiGPUS_HIP_RCCL.out")


def _numpy_system(nx, ny):
    """Independent assembly: 4T - sum(interior nb) = sum(boundary nb); N=0, S=W=E=100."""
    mx, my = nx - 2, ny - 2
    n = mx * my
    A = np.zeros((n, n))
    b = np.zeros(n)
    for y in range(1, ny - 1):
        for x in range(1, nx - 1):
            k = (y - 1) * mx + (x - 1)
            A[k, k] = 4.0
            for (xx, yy) in ((x, y + 1), (x, y - 1), (x - 1, y), (x + 1, y)):
                if 1 <= xx <= nx - 2 and 1 <= yy <= ny - 2:
                    A[k, (yy - 1) * mx + (xx - 1)] = -1.0
                elif yy == ny - 1:
                    b[k] += 0.0
                else:
                    b[k] += 100.0
    return A, b


def _assemble(nx, ny, d):
    m, r = os.path.join(d, "m.bin"), os.path.join(d, "b.bin")
    subprocess.run([ASM, "assemble", str(nx), str(ny), m, r], check=True, capture_output=True)
    return m, r


@pytest.fixture(scope="module", autouse=True)
def _built(lam):
    lam.build()


@pytest.mark.parametrize("nx,ny", [(3, 3), (5, 4), (12, 12), (9, 17)])
def test_assembler_matches_independent_assembly(oracle, tmp_path, nx, ny):
    m, r = _assemble(nx, ny, str(tmp_path))
    A, b = oracle.read_bin(m), oracle.read_bin(r).reshape(-1)
    A0, b0 = _numpy_system(nx, ny)
    assert np.array_equal(A, A0) and np.array_equal(b, b0)
    assert np.array_equal(A, A.T) and np.linalg.eigvalsh(A)[0] > 0          # SPD


def test_discrete_solution_agrees_with_reference_jacobi_field(oracle, golden, tmp_path):
    for h in golden["heat"]:
        nx, ny = h["nx"], h["ny"]
        m, r = _assemble(nx, ny, str(tmp_path))
        A, b = oracle.read_bin(m), oracle.read_bin(r).reshape(-1)
        x = np.linalg.solve(A, b)
        sol, field = str(tmp_path / "sol.bin"), str(tmp_path / "heat.bin")
        oracle.write_bin(sol, x)
        subprocess.run([ASM, "field", str(nx), str(ny), sol, field], check=True)
        T = oracle.read_bin(field)
        T_ref = oracle.read_bin(os.path.join(GOLDEN, h["jacobi_file"]))
        assert T.shape == T_ref.shape == (ny, nx)
        # boundary (incl. the corner values the reference sets) identical; interior within the Jacobi
        # solver's own stopping error
        assert np.array_equal(T[0], T_ref[0]) and np.array_equal(T[-1], T_ref[-1])
        assert np.array_equal(T[:, 0], T_ref[:, 0]) and np.array_equal(T[:, -1], T_ref[:, -1])
        assert np.max(np.abs(T - T_ref)) < 0.5, np.max(np.abs(T - T_ref))


def test_oracle_cg_on_heat_system_is_bit_identical_to_reference(oracle, golden):
    g = golden["heat_cg"]
    A = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".matrix.bin"))
    b = oracle.read_bin(os.path.join(GOLDEN, g["name"] + ".rhs.bin"))
    x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
    x, st = oracle.cg_solve(A, b, g["max_iters"], g["tol"])
    assert st["num_iters"] == g["iters_printed"]
    assert np.array_equal(x, x_ref)


@pytest.mark.gpu
def test_heat_cg_file_mode_small(oracle, golden, tmp_path):
    g = golden["heat_cg"]
    sol = str(tmp_path / "sol.bin")
    r = subprocess.run([RCCL_EXE, "-A", os.path.join(GOLDEN, g["name"] + ".matrix.bin"), "-b",
                        os.path.join(GOLDEN, g["name"] + ".rhs.bin"), "-o", sol], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = r.stdout.replace("\n", "").split(",")
    assert abs(int(f[7]) - g["iters_printed"]) <= 3 and float(f[8]) < 1e-9
    x = oracle.read_bin(sol).reshape(-1)
    x_ref = oracle.read_bin(os.path.join(GOLDEN, g["tag"] + ".sol.bin")).reshape(-1)
    assert np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref) < 1e-8      # cond ~ 50


@pytest.mark.gpu
def test_heat_cg_file_mode_config5(lam, oracle, tmp_path):
    """nx = ny = 130 -> n = 16384 unknowns, 2.1 GB dense matrix file, tol 1e-9: residual and solution
    against the CPU oracle on the same files."""
    nx = ny = 130
    m, rhs = _assemble(nx, ny, str(tmp_path))
    sol = str(tmp_path / "sol.bin")
    r = subprocess.run([RCCL_EXE, "-A", m, "-b", rhs, "-o", sol, "-e", "1e-9"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    f = r.stdout.replace("\n", "").split(",")
    n = (nx - 2) * (ny - 2)
    assert int(f[0]) == n and float(f[8]) < 1e-9
    x = oracle.read_bin(sol).reshape(-1)
    A = np.memmap(m, dtype=np.float64, mode="r", offset=16, shape=(n, n))
    b = oracle.read_bin(rhs).reshape(-1)
    true_res = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    assert true_res <= 2e-9
    x_or, st_or = oracle.cg_solve(np.asarray(A), b, 10000, 1e-9, threads=16)
    assert st_or["converged"]
    # iteration gate: max(3, 2 %) -- what the HIP path measures against the reference's own fixtures (-3 ... 0 over 45
    # fixture x topology runs, profiles/r05_parity_margins.txt; SURVEY 8c's max(2, 1 %) would fail the 181-vs-184 case)
    assert abs(int(f[7]) - st_or["num_iters"]) <= max(3, 0.02 * st_or["num_iters"])
    # solution: both x solve the system only to their residuals, x - x_or = A^-1 (r_or - r), so
    # ||x - x_or|| <= (||r|| + ||r_or||) / lambda_min -- rigorous; lambda_min of the 5-point Laplacian on the m x m interior
    # grid is 4 - 4 cos(pi / (m + 1)) (= 1.186e-3 at m = 128: cond = 6.7e3)
    lam_min = 4.0 - 4.0 * np.cos(np.pi / (nx - 1))
    bound = (np.linalg.norm(b - A @ x) + np.linalg.norm(b - A @ x_or)) / lam_min
    assert np.linalg.norm(x - x_or) <= 1.01 * bound
    assert np.linalg.norm(x - x_or) / np.linalg.norm(x_or) < 1e-6          # what that bound evaluates to here, rounded up
    # "residual match vs CPU" (BASELINE configs[4]): the RECURSIVE residual after a FIXED number of iterations, HIP against
    # the oracle on the same files -- a comparison of two numbers of size 1e-2 ... 1e-5, not of two numbers that are both
    # already below the tolerance
    with lam.Solver(lam.F64) as s:
        assert s.load_matrix_from_file(m) and s.load_rhs_from_file(rhs)
        for k in (100, 200):
            s.solve(k, 1e-30)
            _, st_k = oracle.cg_solve(np.asarray(A), b, k, 1e-30, threads=16)
            assert s.stats["num_iters"] == st_k["num_iters"] == k + 1
            rel = abs(s.stats["rel_err"] / st_k["rel_err"] - 1)
            print(f"heat n={n}: recursive residual after {k} iterations: HIP {s.stats['rel_err']:.15e} oracle {st_k['rel_err']:.15e} (relative difference {rel:.2e})")
            assert rel < 1e-9, (k, s.stats["rel_err"], st_k["rel_err"])
    assert 0.0 < x.min() and x.max() < 100.0                              # discrete maximum principle
