"""Host-side rendezvous for the multi-GPU mode: hands rank 0's 128-byte RCCL unique
id to the other ranks of ONE node over a loopback TCP socket.

Launch contract: one process per GPU with RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR,
MASTER_PORT in the environment (what `python -m torch.distributed.run` sets).  The
launcher's own store owns MASTER_PORT, so rank 0 listens on the first free port of
MASTER_PORT+1 .. +32 and the other ranks probe that range; a fixed handshake tag
rejects anything else that may be listening there.  Everything after this exchange
(barriers, reductions, the result all-gather) goes through RCCL.
"""
import os
import socket
import struct
import time

_TAG = b"TLS-AMD-RCCL-ID1"
_SPAN = 32


def env_layout():
    """(rank, world_size, local_rank, master_addr, master_port) from the environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")))


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return buf


def share_unique_id(rank, world_size, master_addr, master_port, make_id, timeout=120.0):
    """Rank 0 calls make_id() and serves the bytes; every rank returns them."""
    if world_size == 1:
        return make_id()
    ports = [master_port + 1 + k for k in range(_SPAN)]
    if rank == 0:
        payload = make_id()
        server = None
        for port in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((master_addr, port))
                s.listen(world_size)
                server = s
                break
            except OSError:
                s.close()
        if server is None:
            raise RuntimeError("rendezvous: no free port in %d..%d" % (ports[0], ports[-1]))
        server.settimeout(timeout)
        served = set()
        while len(served) < world_size - 1:
            conn, _ = server.accept()
            try:
                conn.settimeout(5.0)
                hello = _recv_exact(conn, len(_TAG) + 8)
                peer_world, peer_rank = struct.unpack("<ii", hello[len(_TAG):])
                if hello[:len(_TAG)] == _TAG and peer_world == world_size:
                    conn.sendall(_TAG + struct.pack("<i", len(payload)) + payload)
                    served.add(peer_rank)
            except (OSError, ConnectionError, struct.error):
                pass
            finally:
                conn.close()
        server.close()
        return payload
    deadline = time.time() + timeout
    hello = _TAG + struct.pack("<ii", world_size, rank)
    while time.time() < deadline:
        for port in ports:
            try:
                with socket.create_connection((master_addr, port), timeout=2.0) as conn:
                    conn.settimeout(5.0)
                    conn.sendall(hello)
                    head = _recv_exact(conn, len(_TAG) + 4)
                    if head[:len(_TAG)] != _TAG:
                        continue
                    (size,) = struct.unpack("<i", head[len(_TAG):])
                    return _recv_exact(conn, size)
            except (OSError, ConnectionError, struct.error):
                continue
        time.sleep(0.2)
    raise RuntimeError("rendezvous: rank %d could not reach rank 0 on %s:%d..%d"
                       % (rank, master_addr, ports[0], ports[-1]))
