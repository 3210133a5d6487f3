#!/usr/bin/env python3
"""Host->device rate of lam_hip_upload_rows when the device rows are PADDED (N not a multiple of a 4-KiB page: the rows travel
through a dense staging buffer and a layout kernel) against the direct copy of an aligned N -- in ONE call and in the 256-MiB
chunks the file loaders use (round 5: the staging buffer is kept by the context, grow-only; rounds 2-4 allocated and freed up to
1 GiB of it per call, a device-wide synchronisation each time) -- and the download direction."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lam = importlib.import_module("2024-eumaster4hpc-student-challenge_amd")
for n in (32768, 30000, 30001):
    rows = 16384
    A = np.random.default_rng(0).uniform(-1, 1, (rows, n))
    step = max(1, (256 << 20) // (n * 8))
    with lam.Solver(lam.F64) as s:
        s.set_problem(n)
        s.upload_rows(0, A[:512])

        def one_call():
            t0 = time.perf_counter()
            s.upload_rows(0, A)
            return time.perf_counter() - t0

        def chunked():
            t0 = time.perf_counter()
            for r in range(0, rows, step):
                s.upload_rows(r, A[r:r + step])
            return time.perf_counter() - t0

        up = A.nbytes / min(one_call() for _ in range(3)) / 1e9
        upc = A.nbytes / min(chunked() for _ in range(3)) / 1e9
        t0 = time.perf_counter(); B = s.download_rows(0, rows); down = A.nbytes / (time.perf_counter() - t0) / 1e9
        assert np.array_equal(A, B)
        print(f"N={n} (row pitch {'padded' if n * 8 % 4096 else 'exact'}): upload {up:.1f} GB/s in one call, {upc:.1f} GB/s in {-(-rows // step)} chunks of "
              f"{step} rows (the file loaders' shape), download {down:.1f} GB/s of {A.nbytes / 1e9:.2f} GB, round trip exact", flush=True)
