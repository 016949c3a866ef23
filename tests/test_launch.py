"""`python bench.py --gpus N` started plainly spawns its own rank processes (tls_amd/launch.py), the way the
reference's `power()` starts its own Pool (main.py:140-163).  Real processes, loopback, no GPU."""
import os
import subprocess
import sys

from conftest import REPO

from tls_amd import launch

WORKER = r"""
import os, sys
sys.path.insert(0, %r)
from tls_amd import rendezvous
rank, world, local, addr, port = rendezvous.env_layout()
assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
ch = rendezvous.HostChannel(rank, world, addr, port, timeout=60)
parts = ch.allgather_bytes(bytes([rank]))
assert parts == [bytes([r]) for r in range(world)]
assert ch.max(float(rank)) == world - 1.0
ch.barrier()
ch.close()
print("rank %%d of %%d on local %%d" %% (rank, world, local), flush=True)
if len(sys.argv) > 1 and int(sys.argv[1]) == rank:
    sys.exit(7)
"""


def _run(tmp_path, world, *extra):
    script = tmp_path / "w.py"
    script.write_text(WORKER % REPO)
    driver = ("import sys; sys.path.insert(0, %r); from tls_amd import launch; "
              "sys.exit(launch.spawn_ranks([sys.executable, %r] + sys.argv[1:], %d, timeout=120))"
              % (REPO, str(script), world))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run([sys.executable, "-c", driver] + list(extra), capture_output=True, text=True,
                          timeout=300, env=env)


def test_spawned_ranks_meet_and_rank0_owns_stdout(tmp_path):
    proc = _run(tmp_path, 3)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert proc.stdout.strip().splitlines() == ["rank 0 of 3 on local 0"]
    assert "[rank 1] rank 1 of 3 on local 1" in proc.stderr
    assert "[rank 2] rank 2 of 3 on local 2" in proc.stderr


def test_a_failing_rank_fails_the_job(tmp_path):
    proc = _run(tmp_path, 2, "1")
    assert proc.returncode == 7


def test_launcher_detection_and_environment():
    assert not launch.launched_by_a_launcher({})
    assert launch.launched_by_a_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    env = launch.rank_environment(1, 4, 12345, base={})
    assert env["RANK"] == "1" and env["LOCAL_RANK"] == "1" and env["WORLD_SIZE"] == "4"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "12345"


def test_bench_started_plainly_becomes_the_launcher(tmp_path):
    """No GPU here: every rank must get as far as creating its context and fail THERE (the product has no
    CPU path), not at a 'use torch.distributed.run' refusal."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1",
                           "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    if proc.returncode == 0:      # a GPU box: the job ran (host-channel fallback on one device) and printed its line
        assert '"n_gpus": 2' in proc.stdout.strip().splitlines()[-1]
    else:
        assert "torch.distributed.run" not in proc.stderr
        assert "no ROCm-capable device" in proc.stderr or "no usable GPU" in proc.stderr
        assert "[rank 1]" not in proc.stdout
