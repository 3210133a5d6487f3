#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

What it captures (all with OMP_NUM_THREADS=1, fp64, so the reference is deterministic --
SURVEY.md section 4 finding 2):

1. File mode, single process -- oracle/_ref/test_CPU_OMP.out
   (reference: challenge/main/test/test_CG_CPU_OMP.cpp, solver
   LAM/src/CPU/ConjugateGradient_CPU_OMP.hpp:49-91).  Inputs are small dense SPD systems
   made HERE with numpy in the reference generator's recipe (A = Q diag(exp(3.5 U[-1,1])) Q^T,
   rhs U[-1,1]; challenge/main/random_spd_system.cpp:66-97,166 -- that program itself needs
   <mkl.h>, absent from the image, so it is not built) and written in the reference's on-disk
   format (random_spd_system.cpp:105-121).  Outputs: stdout line "Converged in K iterations,
   relative error is E" and the solution file the reference wrote.
2. Generate mode, P = 1..4 MPI ranks -- oracle/_ref/test_CPU_MPI_OMP.out -s N -i K [-e tol]
   (reference: challenge/main/test/test_CG_CPU_MPI_OMP.cpp, solver
   LAM/src/CPU/ConjugateGradient_CPU_MPI_OMP.hpp:71-142).  Output: the CSV line.

3. (round 4) File mode in FLOAT -- oracle/_ref/ref_float_harness.out: the reference's ConjugateGradient_CPU_OMP class
   instantiated with <float> by a 20-line main() of ours (oracle/ref_float_harness.cpp; the reference's drivers hard-code
   <double>, its GPU classes are instantiated with float: ...CUDA_NCCL.cu:767).  Inputs: the fp64 fixtures' systems rounded to
   float (<name>.f32.matrix.bin / .f32.rhs.bin).  `make_golden.py --float-only` adds just these to an existing golden.json.

Fixtures written: <name>.matrix.bin / .rhs.bin (inputs), <name>.sol.bin (reference output,
byte-for-byte, including the garbage upper half of its cols word), golden.json (numbers).
These are data; no reference source text is stored.
"""
import json
import os
import re
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_OMP = os.path.join(ROOT, "oracle", "_ref", "test_CPU_OMP.out")
REF_MPI = os.path.join(ROOT, "oracle", "_ref", "test_CPU_MPI_OMP.out")
MPIEXEC = "/opt/conda/bin/mpiexec"


def write_bin(path, arr2d):
    arr2d = np.ascontiguousarray(arr2d, dtype=np.float64)
    with open(path, "wb") as f:
        f.write(struct.pack("=QQ", arr2d.shape[0], arr2d.shape[1]))
        f.write(arr2d.tobytes())


def make_spd(n, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.uniform(-1.0, 1.0, size=(n, n)))
    eig = np.exp(3.5 * rng.uniform(-1.0, 1.0, size=n))
    qd = q * np.sqrt(eig)[None, :]
    a = qd @ qd.T
    a = 0.5 * (a + a.T)  # exactly symmetric, like the reference generator's output
    b = rng.uniform(-1.0, 1.0, size=(n, 1))
    return a, b


def run(cmd, **kw):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, check=True, **kw)
    return out.stdout


REF_F32 = os.path.join(ROOT, "oracle", "_ref", "ref_float_harness.out")


def float_fixtures(golden):
    """The reference's CPU class with <float> (see the module docstring, 3.)."""
    if not os.path.exists(REF_F32):
        sys.exit("build the reference first: make -C oracle ref")
    golden["file_mode_f32"] = []
    for n, seed, max_iters, tol in [(64, 7, 10000, 1e-5), (128, 42, 10000, 1e-5), (64, 7, 5, 1e-5), (128, 42, 40, 1e-9)]:
        name = f"spd_n{n}_s{seed}"
        a, b = make_spd(n, seed)                      # the very system of the fp64 fixture, rounded to float
        mpath, bpath = os.path.join(HERE, name + ".f32.matrix.bin"), os.path.join(HERE, name + ".f32.rhs.bin")
        for path, arr in ((mpath, a), (bpath, b)):
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            with open(path, "wb") as f:
                f.write(struct.pack("=QQ", arr.shape[0], arr.shape[1]))
                f.write(arr.tobytes())
        tag = f"{name}_f32_i{max_iters}_e{tol:g}"
        spath = os.path.join(HERE, tag + ".sol.bin")
        out = run([REF_F32, mpath, bpath, spath, str(max_iters), repr(tol)])
        m = re.search(r"(Converged in|Did not converge in) (\d+) iterations, relative error is (\S+)", out)
        assert m, out
        golden["file_mode_f32"].append({"name": name, "tag": tag, "n": n, "seed": seed, "max_iters": max_iters, "tol": tol,
                                        "converged": m.group(1).startswith("Converged"), "iters_printed": int(m.group(2)),
                                        "rel_err_printed": float(m.group(3))})
        print(tag, m.group(0))


def main():
    if "--float-only" in sys.argv:
        with open(os.path.join(HERE, "golden.json")) as f:
            golden = json.load(f)
        float_fixtures(golden)
        with open(os.path.join(HERE, "golden.json"), "w") as f:
            json.dump(golden, f, indent=1)
        return
    if not (os.path.exists(REF_OMP) and os.path.exists(REF_MPI)):
        sys.exit("build the reference first: make -C oracle ref")
    golden = {"file_mode": [], "gen_mode": []}

    # ---- 1. file mode ----
    for n, seed, max_iters, tol in [(16, 1, 10000, 1e-9), (64, 7, 10000, 1e-9),
                                    (128, 42, 10000, 1e-9), (256, 3, 10000, 1e-9),
                                    (64, 7, 5, 1e-9), (100, 11, 10000, 1e-6)]:
        name = f"spd_n{n}_s{seed}"
        mpath = os.path.join(HERE, name + ".matrix.bin")
        bpath = os.path.join(HERE, name + ".rhs.bin")
        if not os.path.exists(mpath):
            a, b = make_spd(n, seed)
            write_bin(mpath, a)
            write_bin(bpath, b)
        tag = f"{name}_i{max_iters}_e{tol:g}"
        spath = os.path.join(HERE, tag + ".sol.bin")
        out = run([REF_OMP, mpath, bpath, spath, str(max_iters), repr(tol)])
        m = re.search(r"(Converged in|Did not converge in) (\d+) iterations, relative error is (\S+)", out)
        assert m, out
        golden["file_mode"].append({
            "name": name, "tag": tag, "n": n, "seed": seed, "max_iters": max_iters, "tol": tol,
            "converged": m.group(1).startswith("Converged"),
            "iters_printed": int(m.group(2)), "rel_err_printed": float(m.group(3)),
        })
        print(tag, m.group(0))

    # ---- 2. generate mode (tridiag(1,2,1), b = 1) ----
    sol = "/tmp/_golden_sol.bin"
    for n, P, args in [(4096, 1, ["-i", "15"]), (4096, 2, ["-i", "15"]), (2048, 1, ["-i", "1000"]),
                       (4096, 1, []), (1024, 1, []), (1024, 4, []), (1001, 1, []), (1001, 3, []),
                       (1001, 4, ["-i", "100"]), (2048, 1, ["-e", "1e-4"]), (2048, 3, ["-e", "1e-4"]),
                       (513, 2, [])]:
        cmd = [REF_MPI, "-s", str(n), "-o", sol] + args
        if P > 1:
            cmd = [MPIEXEC, "-n", str(P)] + cmd
        # ranks > 0 print bare newlines (test_CG_CPU_MPI_OMP.cpp:287) which mpiexec may interleave
        # into rank 0's CSV line (it is flushed in pieces): drop every newline
        line = run(cmd).replace("\n", "").strip()
        f = line.split(",")
        # N, procs, threads, gen_s, avg_gemv, avg_iter, iters, err, total_s
        golden["gen_mode"].append({
            "n": n, "P": P, "args": args, "csv": line,
            "iters_printed": int(f[6]), "rel_err_printed": float(f[7]),
        })
        print(P, line)

    # ---- 3. heat problem (BASELINE configs[4]) ----
    # (a) the reference's own Jacobi solver output on small grids (loose sanity fixtures: its stop rule
    #     is max_diff < 1e-3, so the field is only ~0.1 degree from the discrete solution);
    # (b) the reference CG (test_CPU_OMP.out) on the dense system written by OUR assembler
    #     (apps/heat_system.out) for a 12x12 grid -> n = 100.
    golden["heat"] = []
    ref_heat = os.path.join(ROOT, "oracle", "_ref", "heat_equation.out")
    asm = os.path.join(ROOT, "2024-eumaster4hpc-student-challenge_amd", "apps", "heat_system.out")
    for nx, ny in [(12, 12), (18, 18), (34, 34)]:   # square only: on nx != ny the reference misplaces two corner values (heat_equation.cpp:36-37), one of which lands on a boundary point
        hp = os.path.join(HERE, f"heat_{nx}x{ny}.jacobi.bin")
        out = run([ref_heat, str(nx), str(ny), hp])
        m = re.search(r"converged in (\d+) iterations with max_diff=(\S+)", out)
        assert m, out
        golden["heat"].append({"nx": nx, "ny": ny, "jacobi_file": os.path.basename(hp),
                               "jacobi_iters": int(m.group(1)), "jacobi_max_diff": float(m.group(2))})
        print("heat", nx, ny, m.group(0))
    nx = ny = 12
    mpath = os.path.join(HERE, "heat_12x12.matrix.bin")
    bpath = os.path.join(HERE, "heat_12x12.rhs.bin")
    spath = os.path.join(HERE, "heat_12x12_i10000_e1e-09.sol.bin")
    run([asm, "assemble", str(nx), str(ny), mpath, bpath])
    out = run([REF_OMP, mpath, bpath, spath, "10000", "1e-9"])
    m = re.search(r"Converged in (\d+) iterations, relative error is (\S+)", out)
    assert m, out
    golden["heat_cg"] = {"nx": nx, "ny": ny, "n": (nx - 2) * (ny - 2), "name": "heat_12x12",
                         "tag": "heat_12x12_i10000_e1e-09", "max_iters": 10000, "tol": 1e-9,
                         "iters_printed": int(m.group(1)), "rel_err_printed": float(m.group(2))}
    print("heat cg", m.group(0))

    float_fixtures(golden)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1)


if __name__ == "__main__":
    main()
