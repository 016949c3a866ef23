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
    b = group.last_blocks
    assert b[0] == 0 and b[-1] == len(inp["periods"]) and len(b) == len(devices) + 1 and numpy.all(numpy.diff(b) > 0)
    for r, ctx in enumerate(group.contexts):
        names = [c[0] for c in ctx.log]
        assert names.count("prepare") == 1 and ("prepare", int(b[r + 1] - b[r])) in ctx.log
        if collective == "rccl_allgather":
            assert ("comm_init", len(devices), r) in ctx.log and "fetch" not in names
            assert ("comm_allgather_results", int(numpy.max(numpy.diff(b))), len(devices)) in ctx.log
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
