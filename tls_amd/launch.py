"""Self-launch of the one-process-per-GPU layout on ONE node, without an external launcher.

The reference parallelises over periods with `multiprocessing.Pool(processes=use_threads)`
(main.py:140-163): a user types one command and the workers appear.  The counterpart here:
a program that finds itself started plainly (no RANK / WORLD_SIZE in the environment) but asked
for N GPUs re-executes itself N times with the launch contract `tls_amd.rendezvous.env_layout`
reads -- RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port --
exactly what `python -m torch.distributed.run` would have set, so both ways of starting the job
run the same code.  Rank 0's stdout is the job's stdout; the other ranks' stdout goes to
stderr (prefixed), so a consumer that reads "the last line" sees rank 0's.
"""
import os
import socket
import subprocess
import sys
import threading


def launched_by_a_launcher(environ=None):
    """True when RANK and WORLD_SIZE are already set (torch.distributed.run, or spawn_ranks)."""
    env = os.environ if environ is None else environ
    return "RANK" in env and "WORLD_SIZE" in env


def free_port(span=33):
    """A loopback port p such that p .. p+span-1 were all free a moment ago (the rendezvous
    listens on MASTER_PORT+1 .. +32)."""
    for _ in range(64):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        if port + span >= 65535:
            continue
        ok = True
        for k in range(1, span):
            probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                probe.bind(("127.0.0.1", port + k))
            except OSError:
                ok = False
            finally:
                probe.close()
            if not ok:
                break
        if ok:
            return port
    raise RuntimeError("no free loopback port range")


def rank_environment(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # the host driver only supports dmabuf IPC: RCCL's intra-node transport needs this
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _pump(stream, prefix, sink):
    for line in iter(stream.readline, b""):
        sink.write(prefix + line.decode("utf-8", "replace"))
        sink.flush()
    stream.close()


def spawn_ranks(argv, world, timeout=None):
    """Run `argv` (a full command line) as `world` rank processes and wait for all of them.
    Returns the worst exit code.  A rank that fails takes the others down (they would wait for
    it in the rendezvous until their own deadline)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs, pumps = [], []
    for rank in range(world):
        out = None if rank == 0 else subprocess.PIPE
        p = subprocess.Popen(list(argv), env=rank_environment(rank, world, port), stdout=out)
        procs.append(p)
        if out is not None:
            th = threading.Thread(target=_pump, args=(p.stdout, "[rank %d] " % rank, sys.stderr), daemon=True)
            th.start()
            pumps.append(th)
    worst = 0
    try:
        pending = set(range(world))
        import time
        deadline = None if timeout is None else time.time() + timeout
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for q in pending:          # exact PIDs we started, nothing else
                        procs[q].terminate()
            if deadline is not None and time.time() > deadline:
                worst = worst or 124
                for q in pending:
                    procs[q].terminate()
                deadline = None
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
        for th in pumps:
            th.join(2.0)
    return worst
