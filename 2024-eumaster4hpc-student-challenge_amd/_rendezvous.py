"""Launcher glue for one-process-per-GPU runs from Python (bench.py, tests): who am I, and a tiny
control plane (broadcast of the 128-byte RCCL unique id, barrier, max over ranks).  Python twin of
LAM/src/HIP/lam_bootstrap.hpp; the reference does the same job with MPI_Comm_rank/size + MPI_Bcast
(/root/reference/challenge/main/LAM/src/GPU/distributed/ConjugateGradient_MultiGPUS_CUDA_NCCL.cu:320-327).

Deliberately NOT torch.distributed: torch ships its own copies of the ROCm runtime and of RCCL under
the same sonames as /opt/rocm's, and a process that maps both sets aborts at exit.  With this module a
rank process loads exactly one HIP runtime and one RCCL (the ones liblam_hip.so is linked against).

Single node only (what `python -m torch.distributed.run --nnodes=1 ...`, `mpiexec -n P` and `srun -N1`
give): rank 0 listens on a 127.0.0.1 socket whose port it publishes in a file keyed by the launch --
MASTER_PORT (or LAM_JOB_ID) plus the launcher's pid, which all local ranks share as their parent -- so
two launches never see each other's file and a stale one is overwritten before it can be read twice.
"""
import os
import socket
import struct
import time

_RANK_VARS = ("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", "SLURM_PROCID")
_SIZE_VARS = ("WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", "SLURM_NTASKS")
_LOCAL_VARS = ("LOCAL_RANK", "MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID")


def _env_int(names, default):
    for n in names:
        v = os.environ.get(n)
        if v:
            return int(v)
    return default


def launched_with_ranks():
    """True when a launcher exported a rank for this process."""
    return any(os.environ.get(n) for n in _RANK_VARS)


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > (1 << 30):
        raise ValueError("rendezvous: implausible message length")
    return _recv_exact(sock, n)


class Rendezvous:
    """rank / size / local_rank from the launcher's environment + all-gather of byte strings."""

    def __init__(self, timeout=180.0):
        self.rank = _env_int(_RANK_VARS, 0)
        self.size = _env_int(_SIZE_VARS, 1)
        self.local_rank = _env_int(_LOCAL_VARS, self.rank)
        self._peers = []          # rank 0: sockets of ranks 1..size-1 (index rank-1)
        self._sock = None         # other ranks: socket to rank 0
        self._file = None
        if self.size <= 1:
            return
        key = os.environ.get("LAM_JOB_ID") or os.environ.get("MASTER_PORT") or os.environ.get("SLURM_JOB_ID") or "default"
        path = os.environ.get("LAM_RDZV_FILE") or f"/tmp/lam_rdzv.{key}.{os.getppid()}"
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.size)
            nonce = os.urandom(8).hex()
            tmp = f"{path}.{os.getpid()}.tmp"
            # created exclusively, never through a symlink somebody planted at the predictable name, readable by us only
            try:
                os.unlink(tmp)
            except OSError:
                pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as f:
                f.write(f"{srv.getsockname()[1]} {nonce}\n")
            os.replace(tmp, path)             # atomic: readers see the old file or the new one, never half
            self._file = path
            peers = {}
            srv.settimeout(1.0)
            while len(peers) < self.size - 1:
                if time.time() > deadline:
                    srv.close()
                    for c in peers.values():
                        c.close()
                    os.unlink(path)           # do not leave a file behind that nobody listens to
                    self._file = None
                    raise TimeoutError(f"rendezvous: only {len(peers) + 1} of {self.size} ranks arrived")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                # the hello gets a short timeout of its own: a stray local connection that says nothing must not hold
                # the accept loop (and its deadline check) for the whole rendezvous timeout
                conn.settimeout(2.0)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    hello = _recv(conn).decode("ascii", "replace").split()
                    peer = int(hello[0]) if len(hello) == 2 and hello[1] == nonce else -1
                except (OSError, ValueError, ConnectionError, struct.error):
                    peer = -1                 # garbage, a timeout, or a peer that hung up: not one of ours
                if not (0 < peer < self.size) or peer in peers:
                    conn.close()              # somebody holding a stale file, or not a rank at all
                    continue
                conn.settimeout(timeout)
                peers[peer] = conn
                _send(conn, b"ok")
            srv.close()
            self._peers = [peers[r] for r in range(1, self.size)]
            os.unlink(path)                   # everyone is connected: the file has done its job
            self._file = None
        else:
            while True:
                if time.time() > deadline:
                    raise TimeoutError(f"rendezvous: rank {self.rank} could not reach rank 0 through {path}")
                try:
                    port, nonce = open(path).read().split()
                    s = socket.create_connection(("127.0.0.1", int(port)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(timeout)
                    _send(s, f"{self.rank} {nonce}".encode())
                    if _recv(s) == b"ok":
                        self._sock = s
                        break
                    s.close()
                except (OSError, ValueError, ConnectionError):
                    pass                      # no file yet, a stale file, or rank 0 not listening yet
                time.sleep(0.02)

    # -- collectives over the control plane ---------------------------------------------------------
    def allgather(self, payload=b""):
        """Every rank contributes a byte string; every rank gets the list ordered by rank."""
        if self.size <= 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self._peers]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for c in self._peers:
                _send(c, blob)
            return parts
        _send(self._sock, payload)
        blob = _recv(self._sock)
        parts, off = [], 0
        while off < len(blob):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def broadcast(self, payload, src=0):
        return self.allgather(payload if self.rank == src else b"")[src]

    def barrier(self):
        self.allgather(b"")

    def max(self, values):
        """Element-wise maximum over the ranks of a short list of floats."""
        parts = self.allgather(struct.pack(f"<{len(values)}d", *values))
        cols = [struct.unpack(f"<{len(values)}d", p) for p in parts]
        return [max(c[i] for c in cols) for i in range(len(values))]

    def close(self):
        for c in self._peers:
            c.close()
        self._peers = []
        if self._sock is not None:
            self._sock.close()
            self._sock = None
        if self._file:
            try:
                os.unlink(self._file)
            except OSError:
                pass
            self._file = None
