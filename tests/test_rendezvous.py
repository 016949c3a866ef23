"""The torch-free rendezvous that carries the RCCL unique id between the ranks of a
node, exercised with real processes under the launcher the driver uses
(`python -m torch.distributed.run`, world size 2 and 3, loopback)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import REPO

SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
from tls_amd import rendezvous
rank, world, local, addr, port = rendezvous.env_layout()
blob = rendezvous.share_unique_id(rank, world, addr, port, lambda: bytes(range(128)), timeout=60)
assert blob == bytes(range(128)), blob
open(os.path.join(%r, "ok%%d" %% rank), "w").write(str(local))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_unique_id_reaches_every_rank(tmp_path, world):
    script = tmp_path / "w.py"
    script.write_text(SCRIPT % (REPO, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), str(script)]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    for r in range(world):
        assert (tmp_path / ("ok%d" % r)).exists()


def test_single_rank_needs_no_socket():
    from tls_amd import rendezvous
    assert rendezvous.share_unique_id(0, 1, "127.0.0.1", 1, lambda: b"x" * 128) == b"x" * 128


CHANNEL_SCRIPT = r"""
import os, sys, struct
sys.path.insert(0, %r)
from tls_amd import rendezvous
rank, world, local, addr, port = rendezvous.env_layout()
ch = rendezvous.HostChannel(rank, world, addr, port, timeout=60)
parts = ch.allgather_bytes(bytes([rank]) * (rank + 1))
assert parts == [bytes([r]) * (r + 1) for r in range(world)], parts
assert ch.max(10.0 + rank) == 10.0 + world - 1
assert ch.all_true(True) and not ch.all_true(rank != 1)
ch.barrier()
big = os.urandom(300000) if rank == 0 else b""
got = ch.allgather_bytes(big)
assert len(got[0]) == 300000
ch.close()
open(os.path.join(%r, "ch%%d" %% rank), "w").write("ok")
"""


@pytest.mark.parametrize("world", [2, 4])
def test_host_channel_collectives(tmp_path, world):
    script = tmp_path / "c.py"
    script.write_text(CHANNEL_SCRIPT % (REPO, str(tmp_path)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), str(script)]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    for r in range(world):
        assert (tmp_path / ("ch%d" % r)).exists()
