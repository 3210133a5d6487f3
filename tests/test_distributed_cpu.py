"""world_size-2 (and 3) `gloo` test of the N>1 protocol, on CPU.

The HIP kernels need a GPU, so this test runs the *exchange protocol* of the rank mode
(csrc/lam_hip.hip: row partition -> local GEMV on the own row block -> exchange of the partial
p.Ap (summed in rank order) -> x, r updates on the own slice -> exchange of the partial r.r -> p update
on the own slice -> all-gather of p) with the product's partition function (lam_hip_partition through the C ABI) and
torch.distributed collectives, the oracle's operators standing in for the device kernels.  It pins
that the sharded recurrence reproduces the reference's MPI path: bit-identical to the emulated-rank
oracle at P=2 (a+b is order-independent), to rounding at P=3."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, PKG_NAME


def _worker(rank, world, port, n, max_iters, tol, outdir, parts):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as o

    # control plane used by bench.py: rank 0's opaque id reaches everyone
    blob = [os.urandom(128) if rank == 0 else None]
    dist.broadcast_object_list(blob, src=0)
    ids = [None] * world
    dist.all_gather_object(ids, blob[0])
    assert all(i == ids[0] for i in ids) and len(ids[0]) == 128

    row0, nrows = parts[rank]                          # the product's partition (computed by the parent)
    A_loc = o.tridiag(n, row0, nrows)                 # generate mode: rows by global index
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, n)                          # same on every rank (seeded)

    def allreduce(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t[0])

    def allgather_p(p_slice):
        if n % world == 0:
            out = [torch.empty(n // world, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(out, torch.from_numpy(p_slice.copy()))
            return np.concatenate([t.numpy() for t in out])
        full = np.empty(n)                             # uneven last block: one broadcast per owner
        for q, (r0, nr) in enumerate(parts):
            t = torch.from_numpy(p_slice.copy()) if q == rank else torch.empty(nr, dtype=torch.float64)
            dist.broadcast(t, src=q)
            full[r0:r0 + nr] = t.numpy()
        return full

    sl = slice(row0, row0 + nrows)
    x = np.zeros(nrows); r = b[sl].copy(); p = b.copy()
    bb = allreduce(o.dot(b[sl], b[sl])); rr = bb
    k = 1
    while k <= max_iters:
        Ap = o.gemv(A_loc, p)
        alpha = rr / allreduce(o.dot(p[sl], Ap))
        x = o.axpby(alpha, p[sl], 1.0, x)
        r = o.axpby(-alpha, Ap, 1.0, r)
        rr_new = allreduce(o.dot(r, r))
        beta = rr_new / rr
        rr = rr_new
        if np.sqrt(rr / bb) < tol:
            break
        p = allgather_p(o.axpby(1.0, r, beta, p[sl]))
        k += 1
    xs = [None] * world
    dist.all_gather_object(xs, x)
    if rank == 0:
        np.save(os.path.join(outdir, "x.npy"), np.concatenate(xs))
        np.save(os.path.join(outdir, "meta.npy"), np.array([k, np.sqrt(rr / bb)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 256), (3, 301)])
def test_sharded_protocol_matches_reference_mpi_path(world, n, oracle, tmp_path):
    # The row partition comes from the product (lam_hip_partition through the C ABI) in THIS process; the
    # gloo workers are separate processes that hold torch and the oracle but never liblam_hip.so -- no
    # process maps torch's bundled ROCm runtime and /opt/rocm's together.
    import importlib
    import multiprocessing
    lam = importlib.import_module(PKG_NAME)
    parts = [lam.partition(n, world, q) for q in range(world)]
    ctx = multiprocessing.get_context("spawn")
    port = 29600 + world + (os.getpid() % 200)
    max_iters, tol = 10000, 1e-9
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, max_iters, tol, str(tmp_path), parts)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, f"worker exit code {pr.exitcode}"
    x = np.load(tmp_path / "x.npy")
    k, err = np.load(tmp_path / "meta.npy")
    A = oracle.tridiag(n)
    b = np.random.default_rng(5).uniform(-1, 1, n)
    x_emul, st_emul = oracle.cg_solve(A, b, max_iters, tol, P=world)
    x_one, st_one = oracle.cg_solve(A, b, max_iters, tol)
    assert int(k) == st_emul["num_iters"]
    assert abs(int(k) - st_one["num_iters"]) <= max(3, 0.02 * st_one["num_iters"])
    if world == 2:
        assert np.array_equal(x, x_emul) and err == st_emul["rel_err"]
    else:
        np.testing.assert_allclose(x, x_emul, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(A @ x, b, atol=1e-7)


def _gather_ap_worker(rank, world, port, n, max_iters, tol, outdir, parts, vec_dtype):
    """The gather-Ap exchange (option exchange = 1, the default of both multi-GPU topologies since round 5) as a protocol: ONE
    equal-count all-gather of byte records per iteration.  Record of rank q = room for the LONGEST slice (n // P + n % P values,
    rank q < P-1 fills the first n // P), padded to 8 bytes, then the rank's part of p.Ap as a double at stride - 8
    (csrc/lam_ctx.h ex1_stride_bytes; kernels: lam_kernels.h gathered_ap).  r and p are full-length on every rank and updated
    redundantly (the reference CPU path's layout, ConjugateGradient_CPU_MPI_OMP.hpp:476,505), so r.r needs no exchange."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as o
    vdt = np.dtype(vec_dtype)
    row0, nrows = parts[rank]
    base, maxrows = n // world, n // world + n % world
    stride = (maxrows * vdt.itemsize + 7) // 8 * 8 + 8
    A_loc = o.tridiag(n, row0, nrows).astype(vdt)
    b = np.random.default_rng(5).uniform(-1, 1, n).astype(vdt)
    sl = slice(row0, row0 + nrows)
    x = np.zeros(nrows, dtype=vdt); r = b.copy(); p = b.copy()          # FULL r and p on every rank
    bb = float(np.dot(b.astype(np.float64), b.astype(np.float64))); rr = bb
    k = 1
    while k <= max_iters:
        Ap_loc = o.gemv(A_loc, p)
        rec = np.zeros(stride, dtype=np.uint8)
        rec[:nrows * vdt.itemsize] = Ap_loc.view(np.uint8)
        rec[stride - 8:] = np.array([np.dot(p[sl].astype(np.float64), Ap_loc.astype(np.float64))]).view(np.uint8)
        out = [torch.empty(stride, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(out, torch.from_numpy(rec))                      # the iteration's ONLY collective
        recs = [t.numpy() for t in out]
        pAp = 0.0
        for q in range(world):                                           # rank order, the same on every rank
            pAp += float(recs[q][stride - 8:].view(np.float64)[0])
        Ap = np.empty(n, dtype=vdt)
        for i0 in range(0, n, 97):                                       # element i lives in record min(i // base, P - 1)
            for i in range(i0, min(i0 + 97, n)):
                q = min(i // base, world - 1)
                Ap[i] = recs[q][:maxrows * vdt.itemsize].view(vdt)[i - q * base]
        alpha = vdt.type(rr / pAp)
        x = o.axpby(alpha, p[sl], 1.0, x)
        r = o.axpby(-alpha, Ap, 1.0, r)
        rr_new = float(np.dot(r.astype(np.float64), r.astype(np.float64)))
        beta = vdt.type(rr_new / rr)
        rr = rr_new
        if np.sqrt(rr / bb) < tol:
            break
        p = o.axpby(1.0, r, beta, p)
        k += 1
    xs = [None] * world
    dist.all_gather_object(xs, x)
    if rank == 0:
        np.save(os.path.join(outdir, "x.npy"), np.concatenate(xs))
        np.save(os.path.join(outdir, "meta.npy"), np.array([k, np.sqrt(rr / bb)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,vec_dtype,tol", [(2, 256, "float64", 1e-9), (3, 301, "float64", 1e-9), (4, 1001, "float64", 1e-9), (3, 301, "float32", 2e-4)])
def test_gather_ap_protocol_with_the_reference_uneven_partition(world, n, vec_dtype, tol, oracle, tmp_path):
    """world_size 2 / 3 / 4 gloo run of the gather-Ap record protocol, uneven partitions (301 = 100 + 100 + 101, 1001 = 250 x 3 + 251)
    and a 4-byte vector type whose odd-length slice needs the 8-byte padding in front of the double: the solve must reproduce the
    reference's MPI recurrence (oracle with the same number of emulated ranks) to the summation order."""
    import importlib
    import multiprocessing
    lam = importlib.import_module(PKG_NAME)
    parts = [lam.partition(n, world, q) for q in range(world)]
    ctx = multiprocessing.get_context("spawn")
    port = 29650 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_gather_ap_worker, args=(r, world, port, n, 10000, tol, str(tmp_path), parts, vec_dtype)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, f"worker exit code {pr.exitcode}"
    x = np.load(tmp_path / "x.npy").astype(np.float64)
    k, err = np.load(tmp_path / "meta.npy")
    A = oracle.tridiag(n)
    b = np.random.default_rng(5).uniform(-1, 1, n)
    x_emul, st_emul = oracle.cg_solve(A.astype(vec_dtype), b.astype(vec_dtype), 10000, tol, P=world)
    assert err < tol and abs(int(k) - st_emul["num_iters"]) <= max(3, 0.05 * st_emul["num_iters"]), (k, st_emul)
    if vec_dtype == "float64":
        np.testing.assert_allclose(x, x_emul, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(A @ x, b, atol=1e-6)
    else:
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 10 * tol


def test_partition_function_matches_reference_rule(lam, oracle):
    for n, P in [(1001, 4), (65536, 8), (7, 7), (513, 2), (10, 3)]:
        got = [lam.partition(n, P, q) for q in range(P)]
        assert got == [oracle.partition(n, P, q) for q in range(P)]
        assert sum(nr for _, nr in got) == n and got[0][0] == 0
        assert all(got[q][0] + got[q][1] == got[q + 1][0] for q in range(P - 1))
        assert got[-1][1] == n // P + n % P          # remainder on the LAST shard


# ------------------------------------------------------------------------------------------------
# option "symmetric" on row shards: the exchange protocol (gather of full-length contributions), world size 2 and 3
# ------------------------------------------------------------------------------------------------
def _window_masks(n, rows):
    """The kernel's use rule (csrc/lam_kernels.h, symv_use<CYC = true>) restated for whole rows: row r uses column c on its row
    side if d = (c - r) mod n is 0 or lies in its window, on its column side if it lies in the window; the window is
    1 <= d <= (n - 1) // 2 plus, for even n, the antipode d = n / 2 for the rows of the upper half."""
    r = np.asarray(rows)[:, None]
    d = (np.arange(n)[None, :] - r) % n
    win = ((d >= 1) & (d <= (n - 1) // 2)) | ((n % 2 == 0) & (d == n // 2) & (r < n // 2))
    return (d == 0) | win, win


def _sym_worker(rank, world, port, n, iters, outdir, parts):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(11)
    M = rng.uniform(-1, 1, (n, n))
    A = 0.5 * (M + M.T) + n * np.eye(n)                 # the same SPD matrix on every rank (seeded); a rank USES its rows only
    b = rng.uniform(-1, 1, n)
    row0, nrows = parts[rank]
    A_loc = A[row0:row0 + nrows]
    row_side, col_side = _window_masks(n, range(row0, row0 + nrows))

    def product(p):
        """This rank's full-length contribution to A p and its part of p.Ap; then the iteration's ONE exchange: all ranks gather
        every rank's record and add them in rank order (lam_exchange.h, symmetric product on several shards)."""
        contrib = np.zeros(n)
        contrib[row0:row0 + nrows] = (A_loc * row_side) @ p                         # y_r += A_rc p_c over the row's window (+ diagonal)
        contrib += (A_loc * col_side).T @ p[row0:row0 + nrows]                       # y_c += A_rc p_r for the same elements
        rec = torch.from_numpy(np.concatenate([contrib, [float(p @ contrib)]]))
        out = [torch.empty(n + 1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(out, rec)
        Ap, pAp = out[0][:n].numpy().copy(), float(out[0][n])
        for q in range(1, world):
            Ap += out[q][:n].numpy()
            pAp += float(out[q][n])
        return Ap, pAp

    # gather-Ap iteration: full-length r and p on every rank, x on the own slice; r.r needs no exchange
    x = np.zeros(nrows); r = b.copy(); p = b.copy(); rr = float(r @ r)
    Ap0 = None
    for k in range(iters):
        Ap, pAp = product(p)
        if k == 0:
            Ap0 = Ap.copy()
        alpha = rr / pAp
        x += alpha * p[row0:row0 + nrows]
        r -= alpha * Ap
        rr_new = float(r @ r)
        p = r + (rr_new / rr) * p
        rr = rr_new
    xs = [None] * world
    dist.all_gather_object(xs, (x, r, Ap0))
    if rank == 0:
        assert all(np.array_equal(xs[0][1], t[1]) and np.array_equal(xs[0][2], t[2]) for t in xs)     # identical r, A p on every rank
        np.save(os.path.join(outdir, "x.npy"), np.concatenate([t[0] for t in xs]))
        np.save(os.path.join(outdir, "Ap0.npy"), Ap0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (3, 99), (2, 51)])
def test_symmetric_row_shard_protocol(world, n, tmp_path):
    """Every rank contributes, from ITS rows only, the products of each row's cyclic half window and their mirror images; the
    gathered records added in rank order are A p -- every pair once, odd and even n, uneven last block --, identical on all
    ranks, and the gather-Ap recurrence on top of it follows plain CG."""
    import importlib
    import multiprocessing
    lam = importlib.import_module(PKG_NAME)
    parts = [lam.partition(n, world, q) for q in range(world)]
    assert lam.symv_plan_check(n, world)[:2] == (0, 0)                 # the product's own plan for this split covers every pair once
    ctx = multiprocessing.get_context("spawn")
    port = 29700 + world + (os.getpid() % 200)
    iters = 12
    procs = [ctx.Process(target=_sym_worker, args=(r, world, port, n, iters, str(tmp_path), parts)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(300)
        assert pr.exitcode == 0, f"worker exit code {pr.exitcode}"
    rng = np.random.default_rng(11)
    M = rng.uniform(-1, 1, (n, n))
    A = 0.5 * (M + M.T) + n * np.eye(n)
    b = rng.uniform(-1, 1, n)
    np.testing.assert_allclose(np.load(tmp_path / "Ap0.npy"), A @ b, rtol=1e-13, atol=1e-13)
    x = np.zeros(n); r = b.copy(); p = b.copy(); rr = r @ r          # plain CG, same number of iterations
    for _ in range(iters):
        Ap = A @ p
        alpha = rr / (p @ Ap)
        x += alpha * p; r -= alpha * Ap
        rr_new = r @ r
        p = r + (rr_new / rr) * p
        rr = rr_new
    np.testing.assert_allclose(np.load(tmp_path / "x.npy"), x, rtol=1e-10, atol=1e-14)
