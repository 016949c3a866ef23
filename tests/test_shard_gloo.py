"""Multi-GPU host logic on CPU: world_size-2 `gloo` process group.  Each rank plans
the same cost-balanced period blocks, searches ITS block (with the CPU oracle
standing in for the GPU -- test infrastructure only), all-gathers padded shards and
assembles; every rank must end up with the single-process result."""
import os
import socket
import sys

import numpy
import pytest

from tls_amd import shard
from conftest import REPO


def test_partition_by_cost_properties():
    rng = numpy.random.RandomState(0)
    costs = rng.randint(1, 1000, size=997)
    for n_ranks in (1, 2, 3, 4, 8):
        b = shard.partition_by_cost(costs, n_ranks)
        assert b[0] == 0 and b[-1] == len(costs) and len(b) == n_ranks + 1
        assert numpy.all(numpy.diff(b) >= 0)
        block = [costs[b[r]:b[r + 1]].sum() for r in range(n_ranks)]
        assert max(block) <= costs.sum() / n_ranks + costs.max()
    # degenerate: more ranks than periods, empty shards allowed
    b = shard.partition_by_cost(numpy.array([5, 5, 5]), 8)
    assert b[0] == 0 and b[-1] == 3 and numpy.all(numpy.diff(b) >= 0)
    # assemble undoes the padding
    bounds = numpy.array([0, 2, 2, 5])
    padded = numpy.array([1, 2, 0, 0, 0, 0, 3, 4, 5])
    numpy.testing.assert_array_equal(shard.assemble(padded, bounds, 3), [1, 2, 3, 4, 5])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, layout="cyclic"):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from tls_amd import synthetic, shard as sh
    import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    t, f = synthetic.light_curve(20.0, 24, 3e-4, per=3.3, rp=0.05, a=10)
    inp = synthetic.search_inputs(t, f, period_min=1.0, period_max=6.0)
    periods, p = inp["periods"], inp["params"]
    job = sh.ShardedSearch(rank, world, layout=layout)

    def gloo_digests(mine):   # the ranks compare the shares they derived before anything is searched
        parts = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, torch.frombuffer(bytearray(mine), dtype=torch.uint8))
        return [bytes(x.numpy().tobytes()) for x in parts]

    idx = job.plan(inp["t"], periods, inp["table"], p, y=inp["y"], allgather_digests=gloo_digests)
    chi2, row, depth, _ = oracle.search(inp["t"], inp["y"], inp["dy"], numpy.ascontiguousarray(periods[idx]), inp["table"],
                                        p["transit_depth_min"], p["R_star_min"], p["R_star_max"],
                                        p["M_star_min"], p["M_star_max"], p["T0_fit_margin"],
                                        n_threads=2)

    def gloo_allgather(count_per_rank):
        out = []
        for arr, dt in ((chi2, torch.float64), (row, torch.int64), (depth, torch.float64)):
            mine = torch.zeros(count_per_rank, dtype=dt)
            mine[: len(arr)] = torch.from_numpy(arr)
            parts = [torch.zeros(count_per_rank, dtype=dt) for _ in range(world)]
            dist.all_gather(parts, mine)
            out.append(torch.cat(parts).numpy())
        return tuple(out)

    full = job.gather(gloo_allgather)
    numpy.savez(os.path.join(out_dir, "rank%d.npz" % rank), chi2=full[0], row=full[1],
                depth=full[2], idx=idx, cells=job.my_cells(), time=float(numpy.sum(job.times[idx])),
                span=sh.block_makespan(job.times[idx], job.slots), slots=job.slots, times=job.times)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,layout", [(2, "cyclic"), (3, "cyclic"), (3, "blocks")])
def test_sharded_search_equals_single_process(tmp_path, oracle_lib, world, layout):
    import torch.multiprocessing as mp
    from tls_amd import synthetic
    from conftest import oracle_search

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), layout), nprocs=world, join=True)

    t, f = synthetic.light_curve(20.0, 24, 3e-4, per=3.3, rp=0.05, a=10)
    inp = synthetic.search_inputs(t, f, period_min=1.0, period_max=6.0)
    want = oracle_search(oracle_lib, inp)
    ranks = [numpy.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in ranks:  # every rank holds the complete, ordered result
        numpy.testing.assert_array_equal(r["chi2"], want[0])
        numpy.testing.assert_array_equal(r["row"], want[1])
        numpy.testing.assert_array_equal(r["depth"], want[2])
    # the shares tile the grid: every period searched by exactly one rank
    taken = numpy.sort(numpy.concatenate([r["idx"] for r in ranks]))
    numpy.testing.assert_array_equal(taken, numpy.arange(len(inp["periods"])))
    spans = numpy.array([float(r["span"]) for r in ranks])
    times, slots = ranks[0]["times"], int(ranks[0]["slots"])
    if layout == "cyclic":
        # rank r holds periods r, r + world, ...: the same mix of short and long periods on every rank, whatever the model
        for r, rec in enumerate(ranks):
            numpy.testing.assert_array_equal(rec["idx"], numpy.arange(r, len(inp["periods"]), world))
        model = numpy.array([float(r["time"]) for r in ranks])
        assert model.max() / model.mean() < 1.02
    else:
        # contiguous blocks placed by the modelled time at which a rank's LAST round of periods ends (tls_period_costs,
        # block_makespan): the slowest rank is never slower than with blocks of equal summed time
        for a, b in zip(ranks[:-1], ranks[1:]):
            assert a["idx"][-1] + 1 == b["idx"][0]
        by_sum = shard.partition_by_cost(times, world)
        assert spans.max() <= max(shard.block_makespan(times[by_sum[r]:by_sum[r + 1]], slots) for r in range(world)) * (1 + 1e-12)
    assert sum(int(r["cells"]) for r in ranks) == int(numpy.sum(shard._lib.grid_cells(inp["t"], inp["periods"], inp["table"], inp["params"])))


def test_cyclic_shares_are_balanced_in_measured_cycles_without_any_model():
    """The cyclic layout against MEASURED per-period shader cycles (tests/golden/period_cycles_*.npz, tls_debug_period_cycles
    on an MI355X): summed cycles of the eight shares within 1.5 % of each other on every configuration -- the contiguous
    blocks of rounds 2-5 needed a fitted time model to reach 5-8 % (next test)."""
    for fixture in ("k2_90d_slim", "tess_27d", "kepler_4yr_8_fast", "k2_90d_500"):
        g = numpy.load(os.path.join(os.path.dirname(__file__), "golden", "period_cycles_%s.npz" % fixture))
        cycles = g["cycles"].astype(float)
        for ranks in (2, 4, 8):
            sums = numpy.array([cycles[shard.cyclic_indices(len(cycles), ranks, r)].sum() for r in range(ranks)])
            assert sums.max() / sums.mean() < 1.015, (fixture, ranks, sums.max() / sums.mean())
    got = shard.assemble_cyclic(numpy.concatenate([[0, 3, 6, 9], [1, 4, 7, -1], [2, 5, 8, -1]]), 10, 3, 4)
    numpy.testing.assert_array_equal(got, numpy.arange(10))


@pytest.mark.parametrize("fixture,limit", [("k2_90d", 1.05), ("tess_27d", 1.06), ("kepler_4yr_8", 1.06), ("k2_90d_500", 1.05),
                                           ("kepler_4yr_8_fast", 1.07), ("k2_90d_slim", 1.085)])
def test_time_model_balances_measured_period_cycles(fixture, limit, monkeypatch):
    """The shard boundaries against MEASURED per-period shader cycles (tls_debug_period_cycles on an MI355X,
    tools/gpu_cost_model.py; committed as tests/golden/period_cycles_*.npz): blocks placed by the time model of
    tls_period_costs are balanced in measured time at 2, 4 and 8 ranks; blocks placed by trial cells alone (round 2)
    are not.  The round-3 fixtures of the long series were measured in exact prefix-sum mode (TLS_FAST_SLAB=0, which
    the model honours like the search: `options`); `kepler_4yr_8_fast` is the same grid in round 4's default: fast mode, second
    attempts of the periods that hit the undecided band included, the expected hitters in exact mode from the start.
    `k2_90d` is the classic LDS-resident kernel (`slim = 0`), `k2_90d_slim` round 5's default for that series: four workgroups
    per CU share its SIMDs, a period's cycles depend on its three neighbours (fit residual 26 % rms), and cells alone already
    balance it to 1.11."""
    from tls_amd import synthetic
    options = {"fast_slab": 0} if fixture in ("tess_27d", "kepler_4yr_8") else {"slim": 0} if fixture == "k2_90d" else None
    g = numpy.load(os.path.join(os.path.dirname(__file__), "golden", "period_cycles_%s.npz" % fixture))
    sigma = float(g["sigma_ppm"]) * 1e-6 or None
    t, f, kw = synthetic.config(str(g["config"]), sigma=sigma)
    inp = synthetic.search_inputs(t, f, **kw)
    periods = inp["periods"][::int(g["stride"])]
    cycles = g["cycles"].astype(float)
    assert len(cycles) == len(periods)
    job = shard.ShardedSearch(0, 1)
    job.plan(inp["t"], periods, inp["table"], inp["params"], y=inp["y"], options=options)
    assert job.slots in (256, 512, 1024)
    assert (job.slots == 1024) == (fixture == "k2_90d_slim")
    worst_model = worst_cells = 0.0
    for ranks in (2, 4, 8):
        for cost, label in ((job.times, "model"), (job.costs.astype(float), "cells")):
            b = shard.partition_by_cost(cost, ranks)
            blocks = numpy.array([cycles[b[r]:b[r + 1]].sum() for r in range(ranks)])
            imb = blocks.max() / blocks.mean()
            if label == "model":
                worst_model = max(worst_model, imb)
            else:
                worst_cells = max(worst_cells, imb)
    assert worst_model <= limit, (worst_model, worst_cells)
    assert worst_cells >= (1.10 if fixture == "k2_90d_slim" else 1.15), worst_cells   # what balancing cells alone leaves on the table


def test_partition_by_makespan_fills_whole_rounds():
    """A rank's GPU searches `slots` periods side by side: a block of 307 periods on 256 slots takes two rounds.  The
    boundaries are placed by the time at which a block's LAST round ends (block_makespan), not by summed time."""
    times = numpy.full(2459, 1.0)                      # TESS: 2459 periods of (here) equal cost, 8 ranks, 256 slots
    b = shard.partition_by_makespan(times, 8, 256)
    assert b[0] == 0 and b[-1] == 2459 and numpy.all(numpy.diff(b) >= 0)
    spans = [shard.block_makespan(times[b[r]:b[r + 1]], 256) for r in range(8)]
    by_sum = shard.partition_by_cost(times, 8)
    spans_sum = [shard.block_makespan(times[by_sum[r]:by_sum[r + 1]], 256) for r in range(8)]
    assert max(spans) <= max(spans_sum)                # never worse than balancing the sums
    assert max(spans) == 2.0                            # 2459 / 8 = 307 > 256: two rounds somewhere, but never three
    assert shard.block_makespan(numpy.full(307, 1.0), 256) == 2.0
    assert shard.block_makespan(numpy.full(256, 1.0), 256) == 1.0
    # many rounds per rank: the same as balancing sums
    big = numpy.random.RandomState(1).uniform(1, 3, 200000)
    numpy.testing.assert_array_equal(shard.partition_by_makespan(big, 4, 8), shard.partition_by_cost(big, 4))
    # monotone costs, few rounds: the makespan of the slowest block does not exceed the sum-balanced one
    ramp = numpy.linspace(1.0, 2.5, 9679)
    for ranks in (2, 4, 8):
        bm = shard.partition_by_makespan(ramp, ranks, 512)
        bs = shard.partition_by_cost(ramp, ranks)
        worst_m = max(shard.block_makespan(ramp[bm[r]:bm[r + 1]], 512) for r in range(ranks))
        worst_s = max(shard.block_makespan(ramp[bs[r]:bs[r + 1]], 512) for r in range(ranks))
        assert worst_m <= worst_s * (1 + 1e-12)


def test_block_makespan_never_shrinks_with_one_more_period_and_every_partition_covers_the_grid():
    """The advisor's counter-example ([10, 1, 1, 1] on four slots takes 10; one more cheap period must not make it 3.35),
    random blocks with spikes, and the partition on such times: non-decreasing bounds that end at n, equal digests."""
    from tls_amd import shard
    assert shard.block_makespan([10, 1, 1, 1], 4) == 10.0
    assert shard.block_makespan([10, 1, 1, 1, 0.1], 4) >= 10.0
    # round 4's counter-example: five periods of 10 and any number of free ones on four slots take 20 (the fifth runs behind
    # one of the first four), at EVERY block length -- the value is the largest over all round boundaries, not the last one's
    for zeros in range(0, 12):
        assert shard.block_makespan([10] * 5 + [0] * zeros, 4) == 20.0, zeros
    # ... and the value stays within a factor two of the longest-first schedule it models (simulated here: the model lets a
    # round's longest period start at the AVERAGE end of the rounds before it), exact for equal times
    import heapq
    def simulate(times, slots):
        free = [0.0] * slots
        heapq.heapify(free)
        end = 0.0
        for c in sorted(times, reverse=True):
            s0 = heapq.heappop(free)
            heapq.heappush(free, s0 + c)
            end = max(end, s0 + c)
        return end
    rng = numpy.random.RandomState(5)
    for trial in range(30):
        slots = int(rng.choice([2, 4, 16]))
        times = rng.uniform(0.5, 1.5, int(rng.randint(1, 6 * slots)))
        times[rng.randint(0, len(times), 2)] *= rng.uniform(5, 60)
        assert simulate(times, slots) / 2.0 <= shard.block_makespan(times, slots) <= simulate(times, slots) * 2.0
    assert shard.block_makespan(numpy.full(40, 1.5), 8) == simulate(numpy.full(40, 1.5), 8)
    for trial in range(40):
        slots = int(rng.choice([2, 4, 16, 256]))
        n = int(rng.randint(1, 6 * slots))
        times = rng.uniform(0.5, 1.5, n)
        times[rng.randint(0, n, max(1, n // 20))] *= rng.uniform(5, 60)     # commensurate-period spikes
        prev = 0.0
        for k in range(1, n + 1):
            now = shard.block_makespan(times[:k], slots)
            assert now >= prev - 1e-12, (trial, slots, k, prev, now)
            prev = now
        for ranks in (2, 3, 8):
            b = shard.partition_by_makespan(times, ranks, slots)
            assert len(b) == ranks + 1 and b[0] == 0 and b[-1] == n and numpy.all(numpy.diff(b) >= 0)
            assert shard.bounds_digest(b) == shard.bounds_digest(b.copy())
            assert shard.bounds_digest(b) != shard.bounds_digest(b + 1)
