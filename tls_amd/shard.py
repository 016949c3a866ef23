"""Multi-GPU mode: the period grid partitioned over ranks (one process per GPU, or one context per GPU of one
process: tls_amd.search.DeviceGroup), each rank searching its share, ONE all-gather of the per-period (chi2, row,
depth) triples at the end.

The reference's only parallelism is exactly this data parallelism over periods (a multiprocessing pool handing out
ONE period at a time, main.py:140-163); periods are independent, so there is no exchange inside the search.

Layout (round 6): **cyclic** -- rank r takes periods r, r + R, r + 2R, ... of the ascending grid (a block-cyclic
partition with blocks of one period).  Cells per period vary 4x across the grid and a period's cost with its kernel,
its noise level and its neighbours on the CU; rounds 2-5 placed CONTIGUOUS blocks by a fitted time model
(`partition_by_makespan`, kept below: bench.py still reports it) and were as good as the model: 4.2 (90 d), 5.2
(TESS) and 7.3 (Kepler) of 8 on the one-GPU projection.  Dealing the periods out in turn gives every rank the same mix of
short and long periods whatever the model says -- what the reference's pool does by handing out single periods --, and
each rank's own queue then packs its share longest-first (Kepler 7.8 projected; TESS 6.0 with the two-role kernel
that short launches of a slab series take).  A period's result does not depend on which rank searches it.

The collective is pluggable so the host logic can be exercised without GPUs: `RcclGather` (product: ncclAllGather
through the C ABI) or any callable with the same signature (tests use torch.distributed/gloo).
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
    # The block as it stood at EVERY round boundary r: the r full rounds before it share their work evenly, the round's
    # most expensive period runs on top -- and the block takes the largest of these.  The top-k sum and the k-th largest
    # of a multiset never shrink when an element is added, so the value is monotone in the block (the bisection of
    # partition_by_makespan relies on that; a formula for the last boundary alone is not: five periods of 10 and eight
    # of 0 on four slots take 20, not the 12.5 of "three full rounds + 0").  r = 0 says no schedule beats its longest period.
    starts = numpy.arange(0, n, slots)
    before = numpy.concatenate([[0.0], numpy.cumsum(c)])[starts]
    return float(numpy.max(before / slots + c[starts]))


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
    """Undo the padding of an all-gather of CONTIGUOUS blocks: `gathered` has n_ranks blocks of
    count_per_rank entries; block r carries bounds[r+1]-bounds[r] valid ones."""
    parts = []
    for r in range(len(bounds) - 1):
        size = int(bounds[r + 1] - bounds[r])
        parts.append(gathered[r * count_per_rank: r * count_per_rank + size])
    return numpy.concatenate(parts) if parts else gathered[:0]


def cyclic_indices(n_periods, n_ranks, rank):
    """Periods of rank `rank` under the cyclic layout: rank, rank + n_ranks, ..."""
    return numpy.arange(int(rank), int(n_periods), int(n_ranks), dtype=numpy.int64)


def assemble_cyclic(gathered, n_periods, n_ranks, count_per_rank):
    """Undo the padding AND the interleaving of an all-gather of cyclic shares: entry j of rank r's block is period
    r + j * n_ranks."""
    gathered = numpy.asarray(gathered)
    out = numpy.empty(int(n_periods), dtype=gathered.dtype)
    for r in range(int(n_ranks)):
        idx = cyclic_indices(n_periods, n_ranks, r)
        out[idx] = gathered[r * count_per_rank: r * count_per_rank + len(idx)]
    return out


class RcclGather(object):
    """All-gather of device-resident results over RCCL (tls_comm_allgather_results)."""

    def __init__(self, context, n_ranks):
        self.context, self.n_ranks = context, n_ranks

    def __call__(self, count_per_rank):
        return self.context.comm_allgather_results(count_per_rank, self.n_ranks)


class ShardedSearch(object):
    """One rank's view of a period-sharded search.

    plan(...)     decide the shares (same on every rank); returns this rank's period indices
    gather(fn)    all-gather + assemble -> full chi2/row/depth on every rank

    layout "cyclic" (default): rank r searches periods[r::n_ranks].  layout "blocks": contiguous blocks placed by the
    modelled time of tls_period_costs (rounds 2-5; `bounds`)."""

    def __init__(self, rank, n_ranks, layout="cyclic"):
        if layout not in ("cyclic", "blocks"):
            raise ValueError("layout must be 'cyclic' or 'blocks'")
        self.rank, self.n_ranks, self.layout = int(rank), int(n_ranks), layout
        self.n_periods = 0
        self.bounds = None   # layout "blocks": boundaries of the contiguous blocks
        self.count_per_rank = 0
        self.costs = None    # trial cells per period
        self.taps = None     # expected template taps per period
        self.times = None    # modelled search time per period
        self.slots = 0

    def indices(self, rank=None):
        """Period indices (into the ascending grid) of a rank's share."""
        r = self.rank if rank is None else int(rank)
        if self.layout == "cyclic":
            return cyclic_indices(self.n_periods, self.n_ranks, r)
        return numpy.arange(int(self.bounds[r]), int(self.bounds[r + 1]), dtype=numpy.int64)

    def plan(self, t, periods, table, params, y=None, options=None, allgather_digests=None, with_costs=True):
        """The shares of every rank.  Cyclic layout: nothing to compute (the cost model is evaluated all the same when
        `with_costs`: bench.py and the tests report modelled against measured balance).  Block layout: boundaries by
        cumulative MODELLED TIME (tls_period_costs): per period a fixed part (fold, sort, prefix sum), a part per trial
        cell and a part per expected template tap.  y (the flux) only sets the noise level the tap estimate assumes;
        every rank must pass the same.  options: the switches of the searching context (Context.get_options()).
        allgather_digests: a callable (32-byte digest) -> list of every rank's digest; when given, the ranks compare the
        shares they derived before anything is searched (ranks that disagreed -- different grids, different switches --
        would assemble mismatched results without any error).  Returns this rank's period indices."""
        self.n_periods = len(periods)
        if with_costs or self.layout == "blocks":
            sigma = 0.0 if y is None else float(numpy.std(numpy.asarray(y, dtype=numpy.float64)))
            self.costs, self.taps, self.times, self.slots = _lib.period_costs(t, periods, table, params, sigma, with_slots=True,
                                                                              options=options)
        if self.layout == "blocks":
            # (a rank's GPU searches `slots` periods side by side: what is balanced is the time of its LAST round's end)
            self.bounds = partition_by_makespan(self.times, self.n_ranks, self.slots)
            self.count_per_rank = max(1, int(numpy.max(numpy.diff(self.bounds))))
            mine = bounds_digest(self.bounds)
        else:
            self.count_per_rank = max(1, -(-self.n_periods // self.n_ranks))
            mine = bounds_digest(numpy.asarray([self.n_periods, self.n_ranks, self.count_per_rank]))
        if allgather_digests is not None:
            theirs = [bytes(d) for d in allgather_digests(mine)]
            if len(theirs) != self.n_ranks or any(d != mine for d in theirs):
                raise RuntimeError("tls_amd: the ranks derived different period shares (rank %d of %d): same inputs, "
                                   "switches and device type on every rank?" % (self.rank, self.n_ranks))
        return self.indices()

    def my_cells(self):
        return int(numpy.sum(self.costs[self.indices()]))

    def gather(self, allgather):
        chi2, row, depth = allgather(self.count_per_rank)
        return self.assemble(chi2), self.assemble(row), self.assemble(depth)

    def assemble(self, gathered):
        """One gathered array (n_ranks blocks of count_per_rank entries) back in grid order."""
        if self.layout == "cyclic":
            return assemble_cyclic(gathered, self.n_periods, self.n_ranks, self.count_per_rank)
        return assemble(gathered, self.bounds, self.count_per_rank)
