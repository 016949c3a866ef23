"""power(devices=[...]) / tls_amd.search.DeviceGroup: the period grid of one search over several GPUs of one process
(reference: the use_threads pool over periods, main.py:140-163, validate.py:81).  Host logic here, with stand-in
contexts that search through the CPU oracle and record what was called: the GPU tests run the real contexts."""
import threading

import numpy
import pytest

from tls_amd import search as tsearch, shard, synthetic
from conftest import oracle_search


class RecordingContext(object):
    """The methods of _lib.Context a DeviceGroup uses; the search itself is the oracle's (test infrastructure)."""
    gather_board = None   # shared by the ranks of one communicator: rank -> padded block
    barrier = None
    calls = None

    def __init__(self, device, oracle_lib, inp):
        self.device, self.oracle_lib, self.inp = device, oracle_lib, inp
        self.block = None
        self.log = []
        self.rank = self.n_ranks = None

    def get_options(self):
        return {}

    def prepare(self, t, y, dy, periods, table, params):
        self.log.append(("prepare", len(periods)))
        self.block = numpy.array(periods)

    def execute(self):
        self.log.append(("execute",))

    def synchronize(self):
        self.log.append(("synchronize",))

    def _results(self):
        if len(self.block) == 0:
            return numpy.zeros(0), numpy.zeros(0, dtype=numpy.int64), numpy.zeros(0)
        got = oracle_search(self.oracle_lib, self.inp, periods=self.block)
        return numpy.asarray(got[0]), numpy.asarray(got[1], dtype=numpy.int64), numpy.asarray(got[2])

    def fetch(self):
        self.log.append(("fetch",))
        return self._results()

    def search(self, t, y, dy, periods, table, params):
        self.prepare(t, y, dy, periods, table, params)
        return self._results() + ({},)

    # -- the RCCL entry points (tls_comm_*): an in-process board stands in for ncclAllGather
    def comm_unique_id(self):
        self.log.append(("comm_unique_id",))
        return b"u" * 128

    def comm_init(self, n_ranks, rank, unique_id):
        assert unique_id == b"u" * 128
        self.log.append(("comm_init", n_ranks, rank))
        self.rank, self.n_ranks = rank, n_ranks

    def comm_allgather_results(self, count_per_rank, n_ranks):
        self.log.append(("comm_allgather_results", count_per_rank, n_ranks))
        chi2, row, depth = self._results()
        pad = lambda a: numpy.concatenate([a, numpy.zeros(count_per_rank - len(a), dtype=a.dtype)])
        type(self).gather_board[self.rank] = (pad(chi2), pad(row), pad(depth))
        type(self).barrier.wait(timeout=60)     # every rank takes part, as in the collective
        board = type(self).gather_board
        return tuple(numpy.concatenate([board[r][k] for r in range(n_ranks)]) for k in range(3))

    def comm_destroy(self):
        pass

    def close(self):
        pass


def _inputs():
    t, f = synthetic.light_curve(40.0, 24, 2e-4, per=4.321, rp=0.05, a=12)
    return synthetic.search_inputs(t, f, period_min=1.0, period_max=9.0, oversampling_factor=2)


@pytest.mark.parametrize("devices,collective", [([0, 1, 2], "rccl_allgather"), ([0, 1], "rccl_allgather"),
                                                ([0, 0], "host_concatenate"), ([2, 0, 2], "host_concatenate")])
def test_distinct_devices_take_the_rccl_all_gather_and_repeated_ones_the_host_copy(oracle_lib, devices, collective):
    inp = _inputs()
    RecordingContext.gather_board = {}
    RecordingContext.barrier = threading.Barrier(len(devices))
    group = tsearch.DeviceGroup(devices, context_factory=lambda d: RecordingContext(d, oracle_lib, inp))
    got = group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp)
    for a, b in zip(got, want[:3]):
        numpy.testing.assert_array_equal(a, numpy.asarray(b))   # the blocks come back in period order, nothing padded
    assert group.last_collective == collective
    b = group.last_blocks   # periods per rank: the grid dealt out cyclically
    n_per = len(inp["periods"])
    assert len(b) == len(devices) and int(b.sum()) == n_per and b.max() - b.min() <= 1
    for r, ctx in enumerate(group.contexts):
        names = [c[0] for c in ctx.log]
        assert names.count("prepare") == 1 and ("prepare", int(b[r])) in ctx.log
        numpy.testing.assert_array_equal(ctx.block, inp["periods"][r::len(devices)])
        if collective == "rccl_allgather":
            assert ("comm_init", len(devices), r) in ctx.log and "fetch" not in names
            assert ("comm_allgather_results", int(b.max()), len(devices)) in ctx.log
        else:
            assert "fetch" in names and not any(nm.startswith("comm_") for nm in names)
    # a second search reuses the communicator (one ncclCommInitRank per context and group)
    RecordingContext.gather_board = {}
    RecordingContext.barrier = threading.Barrier(len(devices))
    group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    for ctx in group.contexts:
        assert [c[0] for c in ctx.log].count("comm_init") == (1 if collective == "rccl_allgather" else 0)


def test_a_failing_device_fails_the_search(oracle_lib):
    inp = _inputs()

    class Broken(RecordingContext):
        def fetch(self):
            raise RuntimeError("tls_amd error -2: hipErrorLaunchFailure")

    made = []

    def factory(d):
        made.append((Broken if len(made) == 1 else RecordingContext)(d, oracle_lib, inp))
        return made[-1]

    group = tsearch.DeviceGroup([0, 0], context_factory=factory)
    with pytest.raises(RuntimeError, match="rank 1 of 2"):
        group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])


def test_a_rank_that_fails_before_the_collective_does_not_leave_the_others_waiting(oracle_lib):
    """distinct devices (the RCCL branch).  One rank fails in prepare / execute: nobody may enter the
    all-gather (the stand-in's collective is a barrier over ALL ranks: entering it would block for its 60 s timeout),
    the error names the rank, and the group -- communicator untouched -- serves the next search."""
    import time
    inp = _inputs()

    class FailsOnce(RecordingContext):
        fail_next = True

        def execute(self):
            if type(self).fail_next:
                type(self).fail_next = False
                raise RuntimeError("tls_amd error -2: hipErrorOutOfMemory")
            super().execute()

    made = []

    def factory(d):
        made.append((FailsOnce if len(made) == 2 else RecordingContext)(d, oracle_lib, inp))
        return made[-1]

    RecordingContext.gather_board = {}
    RecordingContext.barrier = threading.Barrier(3)
    group = tsearch.DeviceGroup([0, 1, 2], context_factory=factory)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="rank 2 of 3"):
        group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    assert time.perf_counter() - t0 < 20.0
    assert not any(c[0] == "comm_allgather_results" for ctx in made for c in ctx.log)
    # the communicator was never entered: the same group searches again
    got = group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    numpy.testing.assert_array_equal(got[0], numpy.asarray(oracle_search(oracle_lib, inp)[0]))
    assert [c[0] for c in made[0].log].count("comm_init") == 1


def test_a_failure_inside_the_collective_drops_the_cached_group(oracle_lib, monkeypatch):
    inp = _inputs()

    class BrokenGather(RecordingContext):
        def comm_allgather_results(self, count_per_rank, n_ranks):
            raise RuntimeError("tls_amd error -3: ncclSystemError")

    monkeypatch.setattr(tsearch, "_device_groups", {})
    group = tsearch.DeviceGroup([0, 1], context_factory=lambda d: BrokenGather(d, oracle_lib, inp))
    tsearch._device_groups[(0, 1)] = group
    with pytest.raises(RuntimeError, match="rank 0 of 2"):
        group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    assert (0, 1) not in tsearch._device_groups and group.contexts == []
    with pytest.raises(RuntimeError, match="closed"):
        group.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])


def test_close_takes_the_group_out_of_the_cache(oracle_lib, monkeypatch):
    inp = _inputs()
    monkeypatch.setattr(tsearch, "_device_groups", {})
    monkeypatch.setattr(tsearch, "DeviceGroup", lambda key: _Plain(key, oracle_lib, inp))
    a = tsearch.device_group([0, 0])
    assert tsearch.device_group([0, 0]) is a
    a.close()
    b = tsearch.device_group([0, 0])
    assert b is not a and b.contexts


class _Plain(tsearch.DeviceGroup):
    def __init__(self, key, oracle_lib, inp):
        super().__init__(key, context_factory=lambda d: RecordingContext(d, oracle_lib, inp))


@pytest.mark.parametrize("bad", [[], (), [0.5], ["a"], [-1], "all"])
def test_devices_are_validated_once_for_every_entry_point(bad):
    """devices=[] reached list(devices)[0] (IndexError); now one helper, one ValueError."""
    from tls_amd import survey
    inp = _inputs()
    with pytest.raises(ValueError, match="devices"):
        tsearch.resolve_devices(bad)
    with pytest.raises(ValueError, match="devices"):
        tsearch.search_periods(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], devices=bad, **inp["params"])
    with pytest.raises(ValueError, match="devices"):
        survey.search_batch(inp["t"], numpy.stack([inp["y"], inp["y"]]), devices=bad, period_min=1.0, period_max=9.0,
                            oversampling_factor=2)
    with pytest.raises(ValueError):   # a list AND a single device: refused, not silently ignored
        tsearch.resolve_devices([0], device=1)
    assert tsearch.resolve_devices([3]) == ("one", 3)
    assert tsearch.resolve_devices("auto", device=2) == ("one", 2)      # an explicit device wins over the default
    assert tsearch.resolve_devices("auto") == ("auto", None)
    assert tsearch.resolve_devices(None, device=None) == ("one", None)


def test_auto_uses_every_gpu_when_the_search_is_long_enough_and_one_otherwise(oracle_lib, monkeypatch):
    """power()'s default devices="auto" (reference: use_threads = cpu_count(), validate.py:81): both branches, with the
    number of visible GPUs injected (the planning call tls_period_costs needs no GPU)."""
    inp = _inputs()
    args = (inp["t"], inp["y"], inp["periods"], inp["table"], inp["params"])
    assert tsearch.auto_devices(*args, n_visible=1) is None
    # 40 d at 24 per day: 0.28 ms modelled on one GPU against 50 us per extra rank: three GPUs, not all eight
    few = tsearch.auto_devices(*args, n_visible=8)
    assert few is not None and 2 <= len(few) < 8 and few == list(range(len(few)))
    t2, f2, kw2 = synthetic.config("tess_27d")
    big = synthetic.search_inputs(t2, f2, **kw2)
    assert tsearch.auto_devices(big["t"], big["y"], big["periods"], big["table"], big["params"], n_visible=8) == list(range(8))
    monkeypatch.setattr(tsearch, "AUTO_OVERHEAD_S_PER_RANK", 1.0)
    assert tsearch.auto_devices(*args, n_visible=8) is None          # sharding would cost more than it saves
    assert tsearch.auto_devices(inp["t"], inp["y"], inp["periods"][:5], inp["table"], inp["params"], n_visible=8) is None
    # and the call: auto -> the group of all visible devices
    monkeypatch.setattr(tsearch, "AUTO_OVERHEAD_S_PER_RANK", 50e-6)
    monkeypatch.setattr(tsearch._lib, "device_count", lambda: 2)
    RecordingContext.gather_board = {}
    RecordingContext.barrier = threading.Barrier(2)
    group = tsearch.DeviceGroup([0, 1], context_factory=lambda d: RecordingContext(d, oracle_lib, inp))
    asked = []
    monkeypatch.setattr(tsearch, "device_group", lambda devices: asked.append(list(devices)) or group)
    used = {}
    got = tsearch.search_periods(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], devices="auto", used=used, **inp["params"])
    assert asked == [[0, 1]] and used["devices"] == [0, 1] and used["context"] is group.contexts[0]
    numpy.testing.assert_array_equal(got[0], numpy.asarray(oracle_search(oracle_lib, inp)[0]))


def test_power_hands_its_devices_to_the_group(oracle_lib, monkeypatch):
    """power(devices=[...]) reaches DeviceGroup.search with the whole period grid and uses the group's first context
    for what follows the search; one device id is the plain single-device call."""
    import tls_amd
    inp = _inputs()
    seen = {}

    class Group(tsearch.DeviceGroup):
        def search(self, t, y, dy, periods, table, params):
            seen["n_periods"] = len(periods)
            return super().search(t, y, dy, periods, table, params)

    RecordingContext.gather_board = {}
    RecordingContext.barrier = threading.Barrier(2)
    group = Group([0, 1], context_factory=lambda d: RecordingContext(d, oracle_lib, inp))
    monkeypatch.setattr(tsearch, "device_group", lambda devices: group)
    # (spectra and the T0 fit of power() would go to the device: the search is what is checked here)
    got = tsearch.search_periods(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], devices=[0, 1], **inp["params"])
    assert seen["n_periods"] == len(inp["periods"]) and group.last_collective == "rccl_allgather"
    want = oracle_search(oracle_lib, inp)
    numpy.testing.assert_array_equal(got[0], numpy.asarray(want[0]))
    with pytest.raises(ValueError):
        tsearch.search_periods(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], devices=[0, 1],
                               context=object(), **inp["params"])
    assert "devices" in tls_amd.constants.EXTRA_PARAMETERS
