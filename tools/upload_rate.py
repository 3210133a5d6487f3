#!/usr/bin/env python3
"""Host->device rate of lam_hip_upload_rows (pageable numpy memory), for the PCIe-inclusive note in DESIGN.md."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
n, rows = 32768, 16384
A = np.random.default_rng(0).uniform(-1, 1, (rows, n))
with lam.Solver(lam.F64) as s:
    s.set_problem(n)
    s.upload_rows(0, A[:1024])
    t0 = time.perf_counter(); s.upload_rows(0, A); dt = time.perf_counter() - t0
    print(f"upload_rows {A.nbytes/1e9:.2f} GB pageable: {dt:.3f} s = {A.nbytes/dt/1e9:.1f} GB/s")
    s.upload_rows(rows, A)
    s.generate_random_rhs(1); s.cg_init(); st = s.cg_iterate(50)
    it_gpu = st["t_iter"]
    print(f"one CG iteration streams {8*n*n/1e9:.2f} GB in {it_gpu*1e3:.3f} ms; uploading the matrix once costs {8*n*n/(A.nbytes/dt)/it_gpu:.0f} iterations")
