#!/usr/bin/env python3
"""Host->device rate of lam_hip_upload_rows, for the PCIe-inclusive note in DESIGN.md (section 6, f1):
  (a) pageable numpy memory, default path (the runtime pins the caller's pages and DMAs from them),
  (b) the same through option "upload_staging" (two pinned 64 MiB buffers, host memcpy overlapped with the DMA),
  (c) a read-only mmap of a matrix FILE in the page cache, default path -- what the C++ loaders do."""
import importlib, mmap, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
n, rows = 32768, 16384
A = np.random.default_rng(0).uniform(-1, 1, (rows, n))
with lam.Solver(lam.F64) as s:
    s.set_problem(n)
    s.upload_rows(0, A[:1024])
    for label, staging in (("pageable, runtime pins the pages (default)", 0), ("pageable, option upload_staging (2 x 64 MiB pinned)", 1)):
        s.set_option("upload_staging", staging)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); s.upload_rows(0, A); dt = time.perf_counter() - t0
            best = max(best, A.nbytes / dt / 1e9)
        print(f"upload_rows {A.nbytes/1e9:.2f} GB {label}: {best:.1f} GB/s")
    s.set_option("upload_staging", 0)
    path = os.path.join(tempfile.gettempdir(), "lam_upload_rate.bin")
    A.tofile(path)
    fd = os.open(path, os.O_RDONLY)
    mm = np.frombuffer(mmap.mmap(fd, 0, prot=mmap.PROT_READ), dtype=np.float64).reshape(rows, n)
    for rep in ("first touch of the mapping", "again"):
        t0 = time.perf_counter(); s.upload_rows(0, mm); dt = time.perf_counter() - t0
        print(f"upload_rows {A.nbytes/1e9:.2f} GB from a read-only file mapping (page cache), {rep}: {A.nbytes/dt/1e9:.1f} GB/s")
    os.close(fd); os.unlink(path)
    s.upload_rows(rows, A)
    s.generate_random_rhs(1); s.cg_init(); st = s.cg_iterate(50)
    print(f"one CG iteration streams {8*n*n/1e9:.2f} GB in {st['t_iter']*1e3:.3f} ms")
