"""Multi-GPU mode: the period grid block-partitioned over ranks (one process per
GPU), each rank searching its contiguous block, ONE all-gather of the per-period
(chi2, row, depth) triples at the end.

The reference's only parallelism is exactly this data parallelism over periods
(a multiprocessing pool, main.py:140-163); periods are independent, so there is
no exchange inside the search.  Block boundaries are placed by cumulative trial-
cell cost, not by count: cells per period vary 4x across the grid.

The collective is pluggable so the host logic can be exercised without GPUs:
`RcclGather` (product: ncclAllGather through the C ABI) or any callable with the
same signature (tests use torch.distributed/gloo).
"""
import numpy

from . import _lib


def partition_by_cost(costs, n_ranks):
    """Boundaries b[0..n_ranks] of contiguous blocks with near-equal summed cost:
    rank r owns periods[b[r]:b[r+1]].  Deterministic, identical on every rank."""
    costs = numpy.asarray(costs, dtype=numpy.float64)
    n = len(costs)
    csum = numpy.concatenate([[0.0], numpy.cumsum(costs)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, n_ranks):
        target = total * r / n_ranks
        k = int(numpy.searchsorted(csum, target, side="left"))
        # pick the neighbour closer to the target, keep blocks non-decreasing
        if k > 0 and (k > n or abs(csum[k - 1] - target) <= abs(csum[min(k, n)] - target)):
            k -= 1
        bounds.append(min(max(k, bounds[-1]), n))
    bounds.append(n)
    return numpy.asarray(bounds, dtype=numpy.int64)


def block_makespan(times, slots):
    """Time one GPU needs for a block of periods with the given modelled times when `slots` of them are searched
    side by side, most expensive first (the kernel's work queue): the full rounds share their work evenly, the
    last, partly filled round adds its most expensive period.  (A block of 307 periods on 256 workgroup slots takes
    two rounds, not 1.2.)"""
    n = len(times)
    if n == 0:
        return 0.0
    c = numpy.sort(numpy.asarray(times, dtype=numpy.float64))[::-1]
    rounds = -(-n // slots)
    full = (rounds - 1) * slots
    value = float(numpy.sum(c[:full]) / slots + c[full])
    # No schedule beats its longest period, and one more (cheap) period never shortens a block: without these two the
    # model is not monotone in the block length where a round boundary is crossed ([10, 1, 1, 1] on four slots takes
    # 10, and [10, 1, 1, 1, 0.1] does not take 3.35) -- and the bisection of partition_by_makespan relies on monotony.
    value = max(value, float(c[0]))
    if full >= slots and n > full:      # the block as it was when its last round was still empty
        value = max(value, float(numpy.sum(c[:full - slots]) / slots + c[full - slots]))
    return value


def partition_by_makespan(times, n_ranks, slots):
    """Contiguous blocks whose block_makespan is as equal as a bisection on the target makes it; falls back to
    equal summed time where blocks are many rounds long (the two agree there)."""
    times = numpy.asarray(times, dtype=numpy.float64)
    n = len(times)
    if n_ranks <= 1 or n == 0:
        return numpy.asarray([0] + [n] * max(n_ranks, 1), dtype=numpy.int64)[: n_ranks + 1]
    if n >= 64 * slots * n_ranks:      # >= 64 rounds per rank: the last round is noise
        return partition_by_cost(times, n_ranks)

    def fill(target):
        """Greedy: every block as long as its makespan stays <= target.  Returns the bounds (n reached or not)."""
        bounds, lo = [0], 0
        for _ in range(n_ranks):
            a, b = lo, n                      # largest hi in [lo, n] with makespan(lo:hi) <= target
            while a < b:
                mid = (a + b + 1) // 2
                if block_makespan(times[lo:mid], slots) <= target:
                    a = mid
                else:
                    b = mid - 1
            lo = a
            bounds.append(lo)
        return bounds

    lo_t, hi_t = 0.0, block_makespan(times, slots)
    for _ in range(40):
        mid_t = 0.5 * (lo_t + hi_t)
        if fill(mid_t)[-1] >= n:
            hi_t = mid_t
        else:
            lo_t = mid_t
    bounds = fill(hi_t)
    if bounds[-1] < n:
        # the greedy fill did not reach the end at the bisection's upper bracket (the model is only nearly monotone):
        # spread what is left over the ranks instead of handing all of it to the last one
        rest = n - bounds[-1]
        for r in range(1, n_ranks + 1):
            bounds[r] += (rest * r) // n_ranks
    bounds[-1] = n
    bounds = numpy.maximum.accumulate(numpy.asarray(bounds, dtype=numpy.int64))
    return numpy.minimum(bounds, n)


def bounds_digest(bounds):
    """What the ranks compare before they search: every rank derives the block boundaries itself (floating-point
    bisection on modelled times), and two ranks that disagreed would assemble mismatched blocks without any error."""
    import hashlib
    return hashlib.sha256(numpy.asarray(bounds, dtype=numpy.int64).tobytes()).digest()


def assemble(gathered, bounds, count_per_rank):
    """Undo the padding of an all-gather: `gathered` has n_ranks blocks of
    count_per_rank entries; block r carries bounds[r+1]-bounds[r] valid ones."""
    parts = []
    for r in range(len(bounds) - 1):
        size = int(bounds[r + 1] - bounds[r])
        parts.append(gathered[r * count_per_rank: r * count_per_rank + size])
    return numpy.concatenate(parts) if parts else gathered[:0]


class RcclGather(object):
    """All-gather of device-resident results over RCCL (tls_comm_allgather_results)."""

    def __init__(self, context, n_ranks):
        self.context, self.n_ranks = context, n_ranks

    def __call__(self, count_per_rank):
        return self.context.comm_allgather_results(count_per_rank, self.n_ranks)


class ShardedSearch(object):
    """One rank's view of a period-sharded search.

    plan(...)     decide the blocks (same on every rank), prepare this rank's block
    run()         execute on this rank's GPU (asynchronous)
    gather(fn)    all-gather + assemble -> full chi2/row/depth on every rank
    """

    def __init__(self, rank, n_ranks):
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self.bounds = None
        self.count_per_rank = 0
        self.costs = None    # trial cells per period
        self.taps = None     # expected template taps per period
        self.times = None    # modelled search time per period: what the boundaries balance

    def plan(self, t, periods, table, params, y=None):
        """Block boundaries by cumulative MODELLED TIME (tls_period_costs): per period a fixed part (fold, sort,
        prefix sum), a part per trial cell and a part per expected template tap.  Balancing trial cells alone
        leaves the blocks of short periods -- many cheap periods, each with the full fixed cost -- 1.2x (90 d),
        2.1x (TESS) and 4x (Kepler 4 yr) slower than the mean at 8 ranks (profiles/r03_cost_model_fit.json).
        y (the flux) only sets the noise level the tap estimate assumes; every rank must pass the same."""
        sigma = 0.0 if y is None else float(numpy.std(numpy.asarray(y, dtype=numpy.float64)))
        self.costs, self.taps, self.times, self.slots = _lib.period_costs(t, periods, table, params, sigma, with_slots=True)
        # (a rank's GPU searches `slots` periods side by side: what is balanced is the time of its LAST round's end)
        self.bounds = partition_by_makespan(self.times, self.n_ranks, self.slots)
        self.count_per_rank = max(1, int(numpy.max(numpy.diff(self.bounds))))
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        return int(lo), int(hi)

    def my_cells(self):
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        return int(numpy.sum(self.costs[lo:hi]))

    def gather(self, allgather):
        chi2, row, depth = allgather(self.count_per_rank)
        c = self.count_per_rank
        return (assemble(chi2, self.bounds, c), assemble(row, self.bounds, c),
                assemble(depth, self.bounds, c))
