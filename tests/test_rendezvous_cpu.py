"""The torch-free control plane bench.py uses under a launcher (package _rendezvous.py; Python twin of
LAM/src/HIP/lam_bootstrap.hpp): world_size 2 and 3 as real processes on CPU.  Pins rank/size discovery from
the launcher's environment, the unique-id broadcast, barrier, max-over-ranks, that a stale rendezvous
file of a dead launch is not mistaken for the live one, and that no rank process imports torch."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, PKG_NAME

WORKER = r"""
import importlib, json, os, sys
sys.path.insert(0, %r)
lam = importlib.import_module(%r)
r = lam.Rendezvous(timeout=60)
blob = r.broadcast(bytes([7]) * 128 if r.rank == 0 else b"")        # the 128-byte unique id
parts = r.allgather(("rank%%d" %% r.rank).encode() * (r.rank + 1))     # ragged payloads
r.barrier()
mx = r.max([float(r.rank), 10.0 - r.rank])
r.barrier()
r.close()
print(json.dumps({"rank": r.rank, "size": r.size, "local": r.local_rank, "blob_ok": blob == bytes([7]) * 128,
                  "parts": [p.decode() for p in parts], "max": mx, "torch": "torch" in sys.modules,
                  "launched": lam.launched_with_ranks()}))
""" % (ROOT, PKG_NAME)


def _launch(world, env_names, tmp_path, stale=False):
    path = str(tmp_path / "rdzv")
    if stale:
        open(path, "w").write("1 deadbeefdeadbeef\n")         # a dead launch's file: nobody listens on port 1
    procs = []
    for rank in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE")}
        env.update({env_names[0]: str(rank), env_names[1]: str(world), env_names[2]: str(rank), "LAM_RDZV_FILE": path})
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs, path


@pytest.mark.parametrize("world,names,stale", [
    (2, ("RANK", "WORLD_SIZE", "LOCAL_RANK"), False),              # torchrun
    (3, ("PMI_RANK", "PMI_SIZE", "MPI_LOCALRANKID"), False),       # mpiexec (MPICH / hydra)
    (3, ("RANK", "WORLD_SIZE", "LOCAL_RANK"), True),               # a stale file lies where the new one goes
])
def test_rendezvous_world(world, names, stale, tmp_path):
    outs, path = _launch(world, names, tmp_path, stale)
    assert sorted(o["rank"] for o in outs) == list(range(world))
    for o in outs:
        assert o["size"] == world and o["local"] == o["rank"] and o["blob_ok"] and o["launched"]
        assert o["parts"] == ["rank%d" % q * (q + 1) for q in range(world)]
        assert o["max"] == [float(world - 1), 10.0]
        assert o["torch"] is False
    assert not os.path.exists(path)                                # rank 0 removed it once everyone had connected


def test_single_process_needs_no_rendezvous(lam):
    env_backup = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "PMI_RANK", "PMI_SIZE") if k in os.environ}
    try:
        assert not lam.launched_with_ranks()
        r = lam.Rendezvous()
        assert (r.rank, r.size) == (0, 1) and r.allgather(b"x") == [b"x"] and r.max([1.5]) == [1.5]
        r.barrier(); r.close()
    finally:
        os.environ.update(env_backup)


def test_rendezvous_times_out_when_a_rank_never_arrives(tmp_path):
    """A launch that loses a rank must fail with an error, not wait for ever: rank 0 of a 2-rank world whose
    peer never starts gives up after the timeout and removes its rendezvous file."""
    path = str(tmp_path / "rdzv")
    code = ("import importlib, sys; sys.path.insert(0, %r); lam = importlib.import_module(%r)\n"
            "try:\n    lam.Rendezvous(timeout=2.0); print('CONNECTED')\n"
            "except TimeoutError as e:\n    print('TIMEOUT', e)\n" % (ROOT, PKG_NAME))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE")}
    env.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", LAM_RDZV_FILE=path)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert "TIMEOUT" in r.stdout and "1 of 2" in r.stdout, r.stdout + r.stderr
    assert not os.path.exists(path)
    env.update(RANK="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert "TIMEOUT" in r.stdout, r.stdout + r.stderr


def test_rendezvous_survives_strangers(tmp_path):
    """Connections that are not ranks -- one that sends a non-numeric hello with the right framing, one that sends
    an absurd length, one that says nothing -- must not take rank 0 down or hold its accept loop: the real rank
    that arrives afterwards still gets in, well before the rendezvous timeout."""
    import socket
    import struct
    import time
    path = str(tmp_path / "rdzv")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PMI_RANK", "PMI_SIZE")}
    env.update(WORLD_SIZE="2", LAM_RDZV_FILE=path)
    p0 = subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 30 and p0.poll() is None
        time.sleep(0.02)
    assert oct(os.stat(path).st_mode & 0o777) == "0o600"           # created for this user only
    port, nonce = open(path).read().split()
    strangers = []
    for payload in (struct.pack("<Q", 9) + b"abc " + nonce[:5].encode(), struct.pack("<Q", 1 << 60), b""):
        c = socket.create_connection(("127.0.0.1", int(port)), timeout=5)
        if payload:
            c.sendall(payload)
        strangers.append(c)
    bad = f"x {nonce}".encode()                                   # right nonce, rank field not a number
    c = socket.create_connection(("127.0.0.1", int(port)), timeout=5)
    c.sendall(struct.pack("<Q", len(bad)) + bad)
    strangers.append(c)
    p1 = subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK="1", LOCAL_RANK="1"), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True)
    outs = []
    for p in (p0, p1):
        so, se = p.communicate(timeout=60)
        assert p.returncode == 0, se
        outs.append(json.loads(so.strip().splitlines()[-1]))
    for c in strangers:
        c.close()
    assert time.time() - t0 < 40
    assert [o["rank"] for o in outs] == [0, 1] and all(o["blob_ok"] and o["size"] == 2 for o in outs)
