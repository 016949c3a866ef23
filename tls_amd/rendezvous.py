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


class HostChannel(object):
    """Star-shaped host channel between the ranks of one node (rank 0 is the hub), over
    persistent loopback TCP sockets.  It carries the RCCL unique id and the small control
    values of the launcher (agreement flags); it is also the stand-in collective if RCCL
    cannot be initialised on a box -- the benchmark then says so in its output.

    Ports MASTER_PORT+40 .. +72 are probed like in share_unique_id().
    """

    _HELLO = b"TLS-AMD-CHANNEL-1"

    def __init__(self, rank, world_size, master_addr, master_port, timeout=120.0):
        self.rank, self.world = int(rank), int(world_size)
        self.peers = {}   # rank 0: {rank: socket}; others: {0: socket}
        if self.world == 1:
            return
        ports = [master_port + 40 + k for k in range(_SPAN)]
        if self.rank == 0:
            server = None
            for port in ports:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((master_addr, port))
                    s.listen(self.world)
                    server = s
                    break
                except OSError:
                    s.close()
            if server is None:
                raise RuntimeError("host channel: no free port in %d..%d" % (ports[0], ports[-1]))
            server.settimeout(timeout)
            while len(self.peers) < self.world - 1:
                conn, _ = server.accept()
                try:
                    conn.settimeout(10.0)
                    hello = _recv_exact(conn, len(self._HELLO) + 8)
                    peer_world, peer_rank = struct.unpack("<ii", hello[len(self._HELLO):])
                    if hello[:len(self._HELLO)] != self._HELLO or peer_world != self.world \
                            or peer_rank in self.peers or not 0 < peer_rank < self.world:
                        conn.close()
                        continue
                    conn.sendall(self._HELLO)
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[peer_rank] = conn
                except (OSError, ConnectionError, struct.error):
                    conn.close()
            server.close()
        else:
            deadline = time.time() + timeout
            hello = self._HELLO + struct.pack("<ii", self.world, self.rank)
            while 0 not in self.peers:
                if time.time() > deadline:
                    raise RuntimeError("host channel: rank %d could not reach rank 0" % self.rank)
                for port in ports:
                    try:
                        conn = socket.create_connection((master_addr, port), timeout=2.0)
                        conn.settimeout(10.0)
                        conn.sendall(hello)
                        if _recv_exact(conn, len(self._HELLO)) == self._HELLO:
                            conn.settimeout(timeout)
                            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.peers[0] = conn
                            break
                        conn.close()
                    except (OSError, ConnectionError):
                        continue
                else:
                    time.sleep(0.2)

    @staticmethod
    def _send(conn, payload):
        conn.sendall(struct.pack("<q", len(payload)) + payload)

    @staticmethod
    def _recv(conn):
        (size,) = struct.unpack("<q", _recv_exact(conn, 8))
        return _recv_exact(conn, size)

    def allgather_bytes(self, payload):
        """Every rank contributes `payload`; every rank gets the list ordered by rank."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [None] * (self.world - 1)
            for r, conn in self.peers.items():
                parts[r] = self._recv(conn)
            blob = b"".join(struct.pack("<q", len(p)) + p for p in parts)
            for conn in self.peers.values():
                self._send(conn, blob)
            return parts
        conn = self.peers[0]
        self._send(conn, payload)
        blob = self._recv(conn)
        parts, pos = [], 0
        for _ in range(self.world):
            (size,) = struct.unpack("<q", blob[pos:pos + 8])
            parts.append(blob[pos + 8: pos + 8 + size])
            pos += 8 + size
        return parts

    def barrier(self):
        self.allgather_bytes(b"")

    def max(self, value):
        parts = self.allgather_bytes(struct.pack("<d", float(value)))
        return max(struct.unpack("<d", p)[0] for p in parts)

    def all_true(self, flag):
        return all(p == b"1" for p in self.allgather_bytes(b"1" if flag else b"0"))

    def close(self):
        for conn in self.peers.values():
            try:
                conn.close()
            except OSError:
                pass
        self.peers = {}
