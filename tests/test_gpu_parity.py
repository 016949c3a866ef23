"""Parity of the HIP search (through the C ABI) with the CPU oracle and with golden
outputs of the unmodified reference.  Tolerance from BASELINE.json `north_star`:
chi2 within 1e-6 relative, argmin period index exact.  In practice the kernels
agree to ~1e-13; the asserts below use 1e-9 so that a real regression is caught,
and the 1e-6 contract is asserted separately."""
import os

import numpy
import pytest

from tls_amd import synthetic, _lib
from conftest import SEARCH_GOLDENS, load_search_golden, oracle_search

pytestmark = pytest.mark.gpu

RTOL_CONTRACT = 1e-6   # north_star
RTOL_TIGHT = 1e-9


def assert_parity(got, want, n_points, tight=RTOL_TIGHT, allow_tie=False):
    chi2, row, depth = got[:3]
    ochi2, orow, odepth = want[:3]
    finite = numpy.isfinite(ochi2)
    numpy.testing.assert_array_equal(numpy.isfinite(chi2), finite)
    numpy.testing.assert_allclose(chi2[finite], ochi2[finite], rtol=RTOL_CONTRACT, atol=0)
    numpy.testing.assert_allclose(chi2[finite], ochi2[finite], rtol=tight, atol=0)
    numpy.testing.assert_array_equal(row, orow)
    # (fast prefix-sum mode moves a window's mean depth by up to 2^-53 (N + W) max|flux|: DESIGN.md section 3)
    numpy.testing.assert_allclose(depth, odepth, rtol=0, atol=max(1e-12, 3e-16 * n_points))
    if finite.any():
        # the argmin period index is exact (north_star).  Only the randomised sweep (allow_tie) accepts
        # the one case its 160-seed run met: two near-identical trial periods at the end of a very
        # short grid whose chi2 tie to within the 1e-13 agreement of the two arithmetics
        ia, ib = int(numpy.nanargmin(numpy.where(finite, chi2, numpy.inf))), int(numpy.nanargmin(numpy.where(finite, ochi2, numpy.inf)))
        if allow_tie and ia != ib:
            assert abs(ochi2[ia] - ochi2[ib]) <= 1e-11 * abs(ochi2[ib]), (ia, ib, ochi2[ia], ochi2[ib])
        else:
            assert ia == ib, (ia, ib, ochi2[ia], ochi2[ib])
    # exactly N where nothing beat the straight line (core.py:46)
    numpy.testing.assert_array_equal(chi2 == n_points, ochi2 == n_points)


@pytest.mark.parametrize("name", SEARCH_GOLDENS)
def test_hip_matches_reference_goldens(gpu, name):
    """Direct comparison with search_period outputs of the unmodified reference."""
    g, table, params = load_search_golden(name)
    chi2, row, depth, counters = gpu.search(g["t"], g["y"], g["dy"], g["periods"], table, params,
                                            count_work=True)
    assert_parity((chi2, row, depth), (g["chi2"], g["row"], g["depth"]), len(g["t"]))
    plain = gpu.search(g["t"], g["y"], g["dy"], g["periods"], table, params)   # may take the pruning kernel
    assert_parity(plain[:3], (g["chi2"], g["row"], g["depth"]), len(g["t"]))


def _inputs(name, **over):
    t, f, kw = synthetic.config(name, **over)
    return synthetic.search_inputs(t, f, **kw)


def test_k2_90d_full_grid_vs_oracle(gpu, oracle_lib):
    """BASELINE config 2 at full size: 9679 periods x 46 durations x T0."""
    inp = _inputs("k2_90d")
    got = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"],
                     count_work=True)
    want = oracle_search(oracle_lib, inp)
    assert_parity(got, want, len(inp["t"]))
    # SURVEY.md Appendix D anchors for this configuration
    assert len(inp["periods"]) == 9679
    assert int(numpy.argmin(got[0])) == 7738
    numpy.testing.assert_allclose(got[0].min(), 4101.6439079674, rtol=1e-10)
    # identical work: same cells passed the depth predicate, same template samples summed
    assert got[3]["grid_cells"] == int(want[3][0])
    assert got[3]["evaluated_cells"] == int(want[3][1])
    assert got[3]["inner_steps"] == int(want[3][2])


def test_tutorial01_vs_oracle(gpu, oracle_lib):
    """BASELINE config 1 (100 d): N=4800 needs the 1-workgroup-per-CU LDS layout."""
    inp = _inputs("tutorial01")
    got = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp)
    assert_parity(got, want, len(inp["t"]))
    assert len(inp["periods"]) == 10870 and int(numpy.argmin(got[0])) == 8598
    numpy.testing.assert_allclose(got[0].min(), 4562.1426757310, rtol=1e-10)


def test_noisy_500ppm_vs_oracle(gpu, oracle_lib):
    """Config 2 at 500 ppm: 55 % of the cells pass the predicate."""
    inp = _inputs("k2_90d", sigma=500e-6)
    sel = inp["periods"][::5]
    got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert_parity(got, want, len(inp["t"]))


def test_per_point_uncertainties_vs_oracle(gpu, oracle_lib):
    """Non-uniform dy: the weighted (A, B) kernel variant."""
    t, f, kw = synthetic.config("k2_90d")
    rng = numpy.random.RandomState(7)
    dy = rng.uniform(4e-5, 9e-5, len(f))
    inp = synthetic.search_inputs(t, f, dy, **kw)
    sel = inp["periods"][::4]
    got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert_parity(got, want, len(inp["t"]))


def test_tess_2min_full_grid_vs_oracle(gpu, oracle_lib):
    """BASELINE config 4 (N=19440): folded series does not fit LDS -> HBM-slab variant.  The WHOLE
    2459-period grid against the oracle, evaluated-cell counts included."""
    inp = _inputs("tess_27d")
    got = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    assert len(inp["periods"]) == 2459
    want = oracle_search(oracle_lib, inp)
    assert_parity(got, want, len(inp["t"]))
    assert got[3]["grid_cells"] == int(want[3][0])
    assert got[3]["evaluated_cells"] == int(want[3][1])
    assert got[3]["inner_steps"] == int(want[3][2])
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    for x, y in zip(got[:3], plain[:3]):
        numpy.testing.assert_array_equal(x, y)


def test_kepler_4yr_full_grid_vs_oracle(gpu, oracle_lib):
    """BASELINE config 3 (N=70128, W=8416, 182 388 periods, 6e10 trial cells) at FULL size.
    (i) the live oracle on >= 1000 periods spread over the grid; (ii) every period against the
    oracle's whole-grid output kept in tests/golden/oracle_kepler_4yr_grid.npz
    (tools/gen_oracle_grid.py; ~2 core-hours, so not recomputed here) after that fixture has been
    re-checked on the same >= 1000 periods; (iii) the whole-grid argmin exact, with the live oracle
    re-evaluated on +-64 periods around it."""
    inp = _inputs("kepler_4yr")
    periods = inp["periods"]
    n_per = len(periods)
    assert n_per == 182388
    got = gpu.search(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    assert not gpu.plan_info()["resident"]
    sel = numpy.arange(0, n_per, int(os.environ.get("TLS_KEPLER_STRIDE", 181)))
    assert len(sel) >= 1000
    want = oracle_search(oracle_lib, inp, periods=periods[sel])
    # a spread sample has its own argmin; the exact-argmin assertion holds for it as well
    assert_parity(tuple(a[sel] for a in got[:3]), want, len(inp["t"]))
    fix = numpy.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_kepler_4yr_grid.npz"))
    assert len(fix["chi2"]) == n_per
    numpy.testing.assert_array_equal(fix["periods_first_last"], periods[[0, -1]])
    numpy.testing.assert_allclose(fix["chi2"][sel], want[0], rtol=1e-13, atol=0)   # the fixture is this oracle's output
    numpy.testing.assert_array_equal(fix["row"][sel], want[1])
    assert_parity(got, (fix["chi2"], fix["row"].astype(numpy.int64), fix["depth"]), len(inp["t"]))
    best = int(numpy.argmin(got[0]))
    assert best == int(numpy.argmin(fix["chi2"]))
    lo, hi = max(0, best - 64), min(n_per, best + 65)
    near = oracle_search(oracle_lib, inp, periods=periods[lo:hi])
    assert_parity(tuple(a[lo:hi] for a in got[:3]), near, len(inp["t"]))
    assert lo + int(numpy.argmin(near[0])) == best
    assert abs(periods[best] - 10.123) < 0.01      # the injected planet (tls_amd/synthetic.py)


# ---- size-independent properties at full size ------------------------------------------
def test_period_order_invariance_and_determinism(gpu):
    """Each period is independent (core.py:96-109): any order, any subset, any repeat
    of the call gives bit-identical per-period results."""
    inp = _inputs("k2_90d")
    p = inp["periods"]
    a = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"])
    b = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"])
    for x, y in zip(a[:3], b[:3]):
        numpy.testing.assert_array_equal(x, y)
    perm = numpy.random.RandomState(1).permutation(len(p))
    c = gpu.search(inp["t"], inp["y"], inp["dy"], p[perm], inp["table"], inp["params"])
    for x, y in zip(a[:3], c[:3]):
        numpy.testing.assert_array_equal(x[perm], y)
    lo, hi = 3000, 5200  # a shard, as the multi-GPU mode would cut it
    d = gpu.search(inp["t"], inp["y"], inp["dy"], p[lo:hi], inp["table"], inp["params"])
    for x, y in zip(a[:3], d[:3]):
        numpy.testing.assert_array_equal(x[lo:hi], y)


@pytest.mark.parametrize("name,stride", [("k2_90d", 1), ("tess_27d", 8), ("kepler_4yr", 900)])
def test_plain_and_counting_kernels_agree_bit_for_bit(gpu, name, stride):
    """A search that counts its work (count_work: evaluated cells, template taps) runs an instantiation of the kernel of
    its own; what it reports is what the plain instantiation reports, bit for bit (LDS-resident and HBM-slab variants)."""
    inp = _inputs(name)
    p = inp["periods"][::stride]
    counted = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"], count_work=True)
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"])
    for x, y in zip(counted[:3], plain[:3]):
        numpy.testing.assert_array_equal(x, y)
    assert counted[3]["evaluated_cells"] > 0 and counted[3]["inner_steps"] > counted[3]["evaluated_cells"]


def test_counting_search_runs_the_kernel_family_of_the_plain_search(gpu):
    """Which kernel a search takes is the plan's choice (series, noise level) -- not changed by asking for the work counters:
    at 200 ppm the host takes the fp32 screen of the classic kernel, the counting search its counting instantiation (not the
    four-slot kernel, whose values differ from the classic family's in the last bits), and both return the same bits."""
    inp = _inputs("k2_90d", sigma=200e-6)
    p = inp["periods"][::7]
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"])
    assert gpu.last_kernel() in ("resident+screen32", "resident+prune")
    counted = gpu.search(inp["t"], inp["y"], inp["dy"], p, inp["table"], inp["params"], count_work=True)
    assert gpu.last_kernel() == "resident"
    for x, y in zip(counted[:3], plain[:3]):
        numpy.testing.assert_array_equal(x, y)


def test_chi2_bounds_and_flat_light_curve(gpu):
    """chi2 <= N everywhere; a light curve with nothing deeper than transit_depth_min
    returns exactly N, depth 0 (core.py:46-48; tests/test_transit_depth_min.py:62-70)."""
    inp = _inputs("k2_90d")
    chi2, row, depth, _ = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"],
                                     inp["params"])
    n = len(inp["t"])
    assert numpy.all(chi2 <= n) and numpy.all(chi2 > 0)
    assert numpy.all((depth > 0.99) | (depth == 0))
    params = dict(inp["params"], transit_depth_min=0.01)
    chi2, row, depth, _ = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"],
                                     params)
    assert numpy.all(chi2 == n) and numpy.all(depth == 0)


def test_update_flux_equals_fresh_prepare(gpu):
    """Survey mode: swapping the flux of a prepared plan == preparing from scratch."""
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    _, f1, _ = synthetic.config("k2_90d", seed=1)
    i0 = synthetic.search_inputs(t, f0, **kw)
    i1 = synthetic.search_inputs(t, f1, **kw)
    sel = i0["periods"][::7]
    fresh = gpu.search(i1["t"], i1["y"], i1["dy"], sel, i1["table"], i1["params"])
    gpu.prepare(i0["t"], i0["y"], i0["dy"], sel, i0["table"], i0["params"])
    gpu.update_flux(i1["y"], i1["dy"])
    gpu.execute()
    swapped = gpu.fetch()
    for x, y in zip(fresh[:3], swapped):
        numpy.testing.assert_array_equal(x, y)
    # a much noisier light curve through the same plan: the host re-decides between the plain and
    # the pruning kernel from the new noise level, results as from a fresh prepare
    _, f2, _ = synthetic.config("k2_90d", seed=2, sigma=800e-6)
    i2 = synthetic.search_inputs(t, f2, **kw)
    fresh = gpu.search(i2["t"], i2["y"], i2["dy"], sel, i2["table"], i2["params"])
    gpu.prepare(i0["t"], i0["y"], i0["dy"], sel, i0["table"], i0["params"])
    gpu.update_flux(i2["y"], i2["dy"])
    gpu.execute()
    for x, y in zip(fresh[:3], gpu.fetch()):
        numpy.testing.assert_array_equal(x, y)


def test_resident_chi2_is_only_trusted_while_the_device_still_holds_it(gpu):
    """spectra(resident=True) reads the chi2 the search left in HBM -- only while that is still what the device holds.
    Every call that launches a search, swaps the flux or reuses the result buffers ends that (Context.holds), and so does
    an in-place edit of the fetched array; the spectra then come from the uploaded array and equal a fresh upload."""
    from tls_amd import search as search_mod
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    _, f1, _ = synthetic.config("k2_90d", seed=1)
    i0 = synthetic.search_inputs(t, f0, **kw)
    i1 = synthetic.search_inputs(t, f1, **kw)
    sel = i0["periods"][::5]

    def fresh():
        gpu.prepare(i0["t"], i0["y"], i0["dy"], sel, i0["table"], i0["params"])
        gpu.execute()
        return gpu.fetch()[0]

    chi2 = fresh()
    assert gpu.holds(chi2)
    want = search_mod.spectra(chi2.copy(), 3, context=gpu)                  # uploaded copy
    got = search_mod.spectra(chi2, 3, context=gpu, resident=True)          # read in place
    for a, b in zip(want, got):
        numpy.testing.assert_array_equal(a, b)
    chi2 = fresh()
    gpu.update_flux(i1["y"], i1["dy"])
    assert not gpu.holds(chi2)
    gpu.execute_timed(1)                                                    # the device now holds seed 1's chi2
    assert not gpu.holds(chi2)
    got = search_mod.spectra(chi2, 3, context=gpu, resident=True)          # ... and seed 0's array is uploaded
    for a, b in zip(want, got):
        numpy.testing.assert_array_equal(a, b)
    chi2 = fresh()
    gpu.search_batch(i0["t"], numpy.stack([i0["y"], i1["y"]]), numpy.stack([i0["dy"], i1["dy"]]), sel, i0["table"], i0["params"])
    assert not gpu.holds(chi2)
    got = search_mod.spectra(chi2, 3, context=gpu, resident=True)          # (used to fail: "needs a finished search")
    for a, b in zip(want, got):
        numpy.testing.assert_array_equal(a, b)
    chi2 = fresh()
    chi2[3] += 1.0                                                          # an in-place edit
    assert not gpu.holds(chi2)


# ---- edge cases ---------------------------------------------------------------------------
def test_edge_cases(gpu, oracle_lib):
    inp = _inputs("k2_90d")
    args = (inp["t"], inp["y"], inp["dy"])
    # empty period list
    chi2, row, depth, cnt = gpu.search(*args, numpy.zeros(0), inp["table"], inp["params"])
    assert len(chi2) == 0 and cnt["grid_cells"] == 0
    # a single period
    one = inp["periods"][1234:1235]
    got = gpu.search(*args, one, inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp, periods=one)
    assert_parity(got, want, len(inp["t"]))
    # no trial duration in range for the period -> inf, row 0, depth 0 (core.py:139-140,188):
    # a table with narrow rows only, at a period whose shortest plausible transit is wider
    from conftest import SimpleTable
    tab = inp["table"]
    narrow = SimpleTable(tab.values, tab.offset[:10], tab.length[:10], tab.width[:10],
                         tab.overshoot[:10])
    short_p = numpy.array([0.05, 0.08, 45.0])
    p = inp["params"]
    got = gpu.search(*args, short_p, narrow, p)
    want = oracle_lib.search(*args, short_p, narrow, p["transit_depth_min"], p["R_star_min"],
                             p["R_star_max"], p["M_star_min"], p["M_star_max"], p["T0_fit_margin"])
    assert numpy.isinf(want[0][0]) and numpy.isinf(want[0][1]) and numpy.isfinite(want[0][2])
    assert_parity(got, want, len(inp["t"]))
    # tiny ragged light curves: widths down to 1 sample, duplicate widths in the table
    for n_small in (150, 211):
        rng = numpy.random.RandomState(0)
        t = numpy.sort(rng.uniform(0, 20, n_small))
        y = 1 + rng.normal(0, 1e-3, n_small)
        small = synthetic.search_inputs(t, y, period_min=1.0, period_max=5.0)
        assert small["table"].width[0] == 1
        got = gpu.search(small["t"], small["y"], small["dy"], small["periods"], small["table"],
                         small["params"])
        want = oracle_search(oracle_lib, small)
        assert_parity(got, want, n_small)


def test_unsorted_and_duplicate_times(gpu, oracle_lib):
    """The fold must be a STABLE sort on arbitrary input order (core.py:120)."""
    rng = numpy.random.RandomState(11)
    t, f, kw = synthetic.config("k2_90d")
    shuffle = rng.permutation(len(t))
    t, f = t[shuffle], f[shuffle]
    t[100:110] = t[100]          # exact ties
    t[2000] = t[77]
    inp = synthetic.search_inputs(t, f, **kw)
    sel = inp["periods"][::9]
    got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert_parity(got, want, len(inp["t"]))


def test_commensurate_periods_cluster_the_phases(gpu, oracle_lib):
    """Periods that are exact multiples of the cadence pile every phase into a few
    buckets of the device sort; results must not change."""
    n = 4320
    t = 3.0 + numpy.arange(n) / 48.0  # exact binary cadence
    rng = numpy.random.RandomState(5)
    y = 1 + rng.normal(0, 5e-5, n)
    inp = synthetic.search_inputs(t, y)
    periods = numpy.array([30 / 48.0, 1.0, 2.5, 96 / 48.0, 10.0, 45.0])
    got = gpu.search(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    want = oracle_search(oracle_lib, inp, periods=periods)
    assert_parity(got, want, n)


def test_tiled_sort_ties_order_and_clustered_phases(gpu, oracle_lib):
    """The HBM-slab variant sorts in two levels (fold_and_sort_tiled) on 32-bit phase keys: unsorted input,
    exact ties, equal keys with different phases, and its fallback (phases piled into a few bins by periods
    commensurate with the cadence) must all give the stable order of the reference."""
    n = 19440
    t = 3.0 + numpy.arange(n) / 720.0            # exact binary-friendly 2-min cadence
    rng = numpy.random.RandomState(3)
    y = 1 + rng.normal(0, 2e-4, n)
    y[(t % 3.7) < 0.08] -= 1.5e-3
    shuffle = rng.permutation(n)
    t, y = t[shuffle], y[shuffle]
    t[500:520] = t[500]                           # exact ties
    t[9000] = t[42]
    # near ties: phases closer than 2^-32, i.e. equal 32-bit sort keys with different exact phases, stored in an
    # order that is neither their phase order nor their index order
    t[1300:1310] = t[77] + numpy.array([7, 2, 9, 1, 4, 8, 3, 6, 5, 0]) * 2.0 ** -36
    t[15000] = t[1300] - 2.0 ** -37
    inp = synthetic.search_inputs(t, y, period_max=9.0)
    periods = numpy.sort(numpy.concatenate([inp["periods"][::120], [0.025, 0.05, 0.75, 1.0, 2.5, 3.7, 8.0]]))  # 0.025, 0.05: 18 and 36 distinct phases -> the fallback
    got = gpu.search(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    want = oracle_search(oracle_lib, inp, periods=periods)
    assert_parity(got, want, len(inp["t"]))
    assert got[3]["evaluated_cells"] == int(want[3][1])


def _folded_reference(t, y, period):
    """core.py:15-18 + 120-123: phases by IEEE division, stable argsort, gather."""
    x = t / period
    phases = x - numpy.floor(x)
    return y[numpy.argsort(phases, kind="mergesort")]


@pytest.mark.parametrize("path", ["resident", "slab", "slab_weighted"])
def test_sort_order_is_the_stable_argsort_bit_for_bit(gpu, path, monkeypatch):
    """The folded flux as the kernel's sort leaves it (tls_debug_folded) against numpy's stable argsort of the
    same phases, and the prefix sum the predicate reads (tls_debug_prefix) against numpy.cumsum, element by element: unsorted input, exact ties, phases closer than the 2^-32 resolution of the
    slab sort's keys, periods that pile the phases into a few bins (the general fallback path)."""
    rng = numpy.random.RandomState(17)
    n = 4320 if path == "resident" else 19440
    t = 3.0 + numpy.arange(n) / (48.0 if path == "resident" else 720.0)
    y = 1 + rng.normal(0, 2e-4, n)
    shuffle = rng.permutation(n)
    t, y = t[shuffle], y[shuffle]
    t[500:520] = t[500]                                                    # exact ties
    t[1300:1310] = t[77] + numpy.array([7, 2, 9, 1, 4, 8, 3, 6, 5, 0]) * 2.0 ** -36   # equal keys, different phases
    t[n // 2] = t[1300] - 2.0 ** -37
    t[n // 3] = t[42]
    dy = rng.uniform(1e-4, 3e-4, n) if path == "slab_weighted" else None
    inp = synthetic.search_inputs(t, y, dy=dy, period_max=9.0)
    periods = numpy.sort(numpy.concatenate([inp["periods"][::300], [0.025, 0.05, 0.75, 1.0, 2.5, 3.7, 8.0]]))
    gpu.prepare(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    assert gpu.plan_info()["resident"] == (path == "resident")
    got = gpu.folded(len(periods), n)
    for k, period in enumerate(periods):
        numpy.testing.assert_array_equal(got[k], _folded_reference(inp["t"], inp["y"], period), err_msg="period %r" % period)
    # ... and the prefix sum the depth predicate reads: numpy.cumsum of the patched series (core.py:126, helpers.py:72)
    C = gpu.prefix_sums(len(periods))
    W = C.shape[1] - 1 - n
    assert W > 0 and W % 2 == 0
    for k, period in enumerate(periods):
        f = _folded_reference(inp["t"], inp["y"], period)
        want = numpy.cumsum(numpy.insert(numpy.append(f, f[:W]), 0, 0))
        numpy.testing.assert_array_equal(C[k], want, err_msg="prefix sum, period %r" % period)


def test_bad_arguments_raise(gpu):
    inp = _inputs("k2_90d")
    with pytest.raises(RuntimeError):
        gpu.search(inp["t"], inp["y"], inp["dy"], numpy.array([1.0, -2.0]), inp["table"],
                   inp["params"])
    with pytest.raises(ValueError):
        gpu.search(inp["t"], inp["y"][:-1], inp["dy"], inp["periods"], inp["table"], inp["params"])
    with pytest.raises(RuntimeError):
        gpu.search(inp["t"][:2], inp["y"][:2], inp["dy"][:2], inp["periods"], inp["table"],
                   inp["params"])
    fresh = _lib.Context(0)
    with pytest.raises(RuntimeError):
        fresh.execute()  # nothing prepared
    fresh.close()


def test_single_rank_rccl_allgather(gpu):
    """The RCCL path with a 1-rank communicator (all a 1-GPU box can run):
    all-gather of a padded shard returns the shard."""
    inp = _inputs("k2_90d")
    sel = inp["periods"][:1000]
    ref = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    gpu.comm_init(1, 0, gpu.comm_unique_id())
    try:
        chi2, row, depth = gpu.comm_allgather_results(1024, 1)
        numpy.testing.assert_array_equal(chi2[:1000], ref[0])
        numpy.testing.assert_array_equal(row[:1000], ref[1])
        numpy.testing.assert_array_equal(depth[:1000], ref[2])
        assert numpy.all(chi2[1000:] == 0)
        # two-step form: gather stays on the device, fetched later
        gpu.execute()
        gpu.comm_allgather_device(1024)
        gpu.execute()                      # the next search may be enqueued right behind it
        chi2b, rowb, depthb = gpu.comm_fetch_gathered(1024, 1)
        numpy.testing.assert_array_equal(chi2b, chi2)
        numpy.testing.assert_array_equal(rowb, row)
        assert gpu.comm_max(3.5) == 3.5
        gpu.comm_barrier()
        # survey form: results of several searches parked in slots, ONE all-gather at the end
        other = dict(inp["params"], transit_depth_min=3e-5)
        ref2 = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], other)
        gpu.comm_stage_results(1024, 1, 3)
        gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        gpu.comm_stage_results(1024, 0, 3)
        gpu.comm_stage_results(1024, 2, 3)
        gpu.comm_allgather_staged(1024, 3)
        for slot, want in ((0, ref), (1, ref2), (2, ref)):
            got = gpu.comm_fetch_staged(1024, 3, slot, 1)
            for x, y in zip(got, want[:3]):
                numpy.testing.assert_array_equal(x[:1000], y)
                assert numpy.all(x[1000:] == 0)
        with pytest.raises(RuntimeError):
            gpu.comm_fetch_gathered(1024, 1)   # the last gather was a staged one
    finally:
        gpu.comm_destroy()


def test_exact_parallel_cumsum_is_numpy_cumsum(gpu):
    """The workgroup scan must reproduce the SEQUENTIAL fp64 sum bit for bit
    (numpy.cumsum, helpers.py:72), whatever the values: ties, binade crossings,
    wide dynamic range, subnormals, large addends."""
    rng = numpy.random.RandomState(42)
    cases = {
        "flux": 1 + rng.normal(0, 5e-5, 4838),
        "flux_long": 1 + rng.normal(0, 2e-4, 78544),
        "ties": numpy.round(rng.uniform(0.5, 1.5, 6000) * 2 ** 12) / 2 ** 12,  # exact halves abound
        "halves": numpy.full(5000, 0.5) + (rng.randint(0, 2, 5000) * 2.0 ** -40),
        "wide": 10.0 ** rng.uniform(-12, 3, 5000),
        "tiny": rng.uniform(0, 1, 3000) * 1e-310,               # subnormal partial sums
        "growing": numpy.cumsum(rng.uniform(0, 1, 2000)) ** 3,   # an addend can exceed the sum
        "zeros": numpy.concatenate([numpy.zeros(10), rng.uniform(0, 2, 500), numpy.zeros(7)]),
        "unnormalised": 1000.0 + rng.normal(0, 1, 5000),
        "one": numpy.array([0.75]),
        "empty": numpy.zeros(0),
    }
    # The result is exact by construction (every block verifies its own start values and is redone by
    # the per-binade routine otherwise), so a broken fast path would only show up as run time: count
    # the fallbacks.  Ordinary flux series must never need one; series that cross more binades per
    # block than the tables hold ("wide", "tiny", "growing") may.
    must_be_fast = ("flux", "flux_long", "ties", "halves", "zeros", "unnormalised", "one")
    for name, v in cases.items():
        want = numpy.concatenate([[0.0], numpy.cumsum(v)])
        for threads in (64, 512, 1024):
            got = gpu.debug_cumsum(v, threads=threads)
            assert numpy.array_equal(got.view(numpy.uint64), want.view(numpy.uint64)), (name, threads)
            stats = gpu.phase_cycles()
            assert stats["cumsum_blocks"] == (len(v) + 16 * threads - 1) // (16 * threads) or len(v) <= 16 * threads, (name, threads)
            if name in must_be_fast:
                assert stats["cumsum_fallbacks"] == 0, (name, threads, stats["cumsum_blocks"], stats["cumsum_fallbacks"])


def test_t0_fit_kernel_vs_oracle_and_reference_residuals(gpu, oracle_lib):
    """tls_t0_fit (batched final T0 fit, stats.py:135-204) against (i) the per-epoch residuals the
    UNMODIFIED reference computes (tests/golden/t0fit_*.npz, tools/gen_golden_t0fit.py) and (ii) the
    oracle's C restatement of the loop at benchmark sizes: same residual per trial epoch, same first
    minimum."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "t0fit_*.npz")))
    assert len(files) >= 4
    for f in files:
        g = numpy.load(f)
        roll = int(len(g["signal"]) / 2) + 1                      # stats.py:189
        got = gpu.t0_fit_residuals(g["t"], g["y"], float(g["period"]), g["scaled_signal"], g["T0_array"], roll)
        numpy.testing.assert_allclose(got, g["residuals"], rtol=1e-12, atol=0, err_msg=f)
        assert g["T0_array"][int(numpy.argmin(got))] == float(g["T0"]), f
    for name, n_epochs in (("k2_90d", 1500), ("tess_27d", 64)):
        t, f, kw = synthetic.config(name)
        inp = synthetic.search_inputs(t, f, **kw)
        row = 30
        signal = 1 - (1 - inp["rows"][row]) * 0.002  # a template row scaled to a shallow depth
        period = 10.12452
        epochs = numpy.linspace(t.min(), t.min() + period, n_epochs)
        roll = int(len(signal) / 2) + 1
        want = oracle_lib.t0_residuals(inp["t"], inp["y"], period, signal, epochs, roll)
        got = gpu.t0_fit_residuals(inp["t"], inp["y"], period, signal, epochs, roll)
        numpy.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
        assert int(numpy.argmin(got)) == int(numpy.argmin(want))
    # ties in t and an unsorted series: the stable order matters for the rolled weights
    rng = numpy.random.RandomState(3)
    t, f, kw = synthetic.config("k2_90d")
    p = rng.permutation(len(t))
    t, f = t[p], f[p]
    t[50:60] = t[50]
    signal = numpy.linspace(0.999, 0.9995, 77)
    epochs = numpy.linspace(t.min(), t.min() + 3.3, 200)
    want = oracle_lib.t0_residuals(t, f, 3.3, signal, epochs, 39)
    got = gpu.t0_fit_residuals(t, f, 3.3, signal, epochs, 39)
    numpy.testing.assert_allclose(got, want, rtol=1e-12, atol=0)


def test_t0_fit_rotation_path_equals_the_general_kernel(gpu, oracle_lib):
    """Final T0 fit, round 6: the epochs of a fit as rotations of one sorted order (tls_t0fit_rot: one wavefront an epoch, the
    out-of-transit terms summed once per fit) against the general kernel (switch t0_rot = 0: every epoch checks every pair of
    neighbours, every term divided again) -- the same order per epoch, so residuals agree to the rounding of a sum taken in
    another order (1e-13) and the first minimum is the same epoch -- at the FULL epoch counts of the three sizes (the series
    in LDS and in HBM scratch), on unevenly sampled times, and where the rotation path must hand the fit back: tied phases
    (duplicate time stamps, a period commensurate with the cadence)."""
    rng = numpy.random.RandomState(11)
    cases = []
    for name, period in (("k2_90d", 10.12452), ("tess_27d", 3.377), ("kepler_4yr", 10.12452)):
        t, f, kw = synthetic.config(name)
        inp = synthetic.search_inputs(t, f, **kw)
        row = len(inp["rows"]) // 3
        signal = 1 - (1 - inp["rows"][row]) * 0.002
        points = min(len(t), int(len(t) / (0.01 * len(signal))))
        cases.append((name, inp["t"], inp["y"], period, signal, numpy.linspace(t.min(), t.min() + period, points)))
    t = numpy.sort(5.0 + rng.uniform(0, 40.0, 3000))
    f = 1 + rng.normal(0, 3e-4, len(t))
    cases.append(("uneven", t, f, 2.7183, numpy.linspace(0.9990, 0.9996, 41), numpy.linspace(t.min(), t.min() + 2.7183, 3000)))
    t2 = t.copy(); t2[100:104] = t2[100]
    cases.append(("ties", t2, f, 2.7183, numpy.linspace(0.9990, 0.9996, 41), numpy.linspace(t.min(), t.min() + 2.7183, 500)))
    tk, fk, _ = synthetic.config("k2_90d")
    cases.append(("commensurate", tk, fk, 78 / 48.0, numpy.linspace(0.9990, 0.9996, 9), numpy.linspace(tk.min(), tk.min() + 78 / 48.0, 400)))
    for k in range(24):   # random sizes, periods, template lengths; epochs anywhere (tls_t0_fit takes the caller's)
        n = int(rng.choice([17, 64, 500, 2000, 6000, 12000]))
        t = numpy.sort(rng.uniform(0, 30.0, n)) if k % 2 else numpy.linspace(2.0, 32.0, n)
        f = 1 + rng.normal(0, float(rng.choice([1e-5, 1e-3])), n)
        period = float(rng.uniform(0.4, 14.0))
        dur = int(rng.randint(1, max(2, min(n, 400))))
        lo = float(rng.choice([t.min(), t.min() - 50.0, t.max()]))
        epochs = numpy.sort(rng.uniform(lo, lo + 2 * period, int(rng.randint(1, 1500))))
        cases.append(("random%d" % k, t, f, period, rng.uniform(0.99, 1.0, dur), epochs))
    # edges: the template as long as the series (tls_t0_fit refuses a longer one), one epoch, equal epochs, a period far beyond the span (every
    # phase tiny) and one below the cadence (the fold wraps many times between neighbours), the smallest series the path takes
    te = numpy.linspace(1.0, 21.0, 960); fe = 1 + rng.normal(0, 1e-3, len(te))
    cases.append(("dur=n", te, fe, 3.3, rng.uniform(0.99, 1.0, len(te)), numpy.linspace(1.0, 4.3, 50)))
    cases.append(("one epoch", te, fe, 3.3, rng.uniform(0.99, 1.0, 30), numpy.array([2.5])))
    cases.append(("equal epochs", te, fe, 3.3, rng.uniform(0.99, 1.0, 30), numpy.full(70, 2.5)))
    cases.append(("long period", te, fe, 1000.0, rng.uniform(0.99, 1.0, 30), numpy.linspace(1.0, 1001.0, 300)))
    cases.append(("short period", te, fe, 0.0123, rng.uniform(0.99, 1.0, 30), numpy.linspace(1.0, 1.0123, 300)))
    cases.append(("16 points", te[:16], fe[:16], 0.11, rng.uniform(0.99, 1.0, 5), numpy.linspace(1.0, 1.11, 40)))
    try:
        for name, t, y, period, signal, epochs in cases:
            roll = int(len(signal) / 2) + 1
            gpu.set_options(t0_rot=None)
            got = gpu.t0_fit_residuals(t, y, period, signal, epochs, roll)
            gpu.set_options(t0_rot=0)
            want = gpu.t0_fit_residuals(t, y, period, signal, epochs, roll)
            numpy.testing.assert_allclose(got, want, rtol=1e-13, atol=0, err_msg=name)
            assert int(numpy.argmin(got)) == int(numpy.argmin(want)), name
            if len(epochs) <= 3000 and len(t) <= 5000:
                ref = oracle_lib.t0_residuals(t, y, period, signal, epochs, roll)
                numpy.testing.assert_allclose(got, ref, rtol=1e-12, atol=0, err_msg=name)
                assert int(numpy.argmin(got)) == int(numpy.argmin(ref)), name
    finally:
        gpu.set_options(t0_rot=None)


def test_spectra_kernel_vs_reference_and_oracle(gpu, oracle_lib):
    """tls_spectra (stats.py:105-132 + helpers.py:93-108 on the device) against outputs of the unmodified
    reference (tests/golden/spectra_*.npz) and against the oracle at benchmark size, both for a chi^2
    array handed in and for the one still resident after a search."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "spectra_*.npz")))
    assert len(files) >= 3
    for f in files:
        g = numpy.load(f)
        SR, praw, power, sde_raw, sde = gpu.spectra(int(g["oversampling_factor"]) * 30, g["chi2"])
        numpy.testing.assert_allclose(SR, g["SR"], rtol=1e-13, err_msg=f)
        numpy.testing.assert_allclose(praw, g["power_raw"], rtol=1e-10, atol=1e-11, err_msg=f)
        numpy.testing.assert_allclose(power, g["power"], rtol=1e-10, atol=1e-11, err_msg=f)
        numpy.testing.assert_allclose([sde_raw, sde], [float(g["SDE_raw"]), float(g["SDE"])], rtol=1e-11)
        assert int(numpy.argmax(power)) == int(numpy.argmax(g["power"]))
    inp = _inputs("k2_90d")
    chi2 = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])[0]
    for kernel in (90, 150, 31):
        want = oracle_lib.spectra(chi2, kernel)
        for got in (gpu.spectra(kernel, chi2), gpu.spectra(kernel)):      # handed in / resident
            # (power_raw = (SR - mean SR) / std SR: the two implementations sum the 9679 SR values in different
            # orders, and a difference of n * 2^-53 in the mean is 5e-10 after the division by std SR = 2e-3;
            # measured offsets: 0.5e-11 .. 2e-11, the same for every element)
            for x, y in zip(got[:3], want[:3]):
                numpy.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-10)
            numpy.testing.assert_allclose(got[3:], want[3:], rtol=1e-11)
            assert int(numpy.argmax(got[2])) == int(numpy.argmax(want[2]))
    numpy.testing.assert_allclose(oracle_lib.spectra(chi2, 90)[4], 25.71647715, rtol=1e-7)


def test_too_short_series_is_rejected(gpu):
    from conftest import SimpleTable
    t = numpy.linspace(0.0, 1.0, 9)
    y = numpy.ones(9)
    table = SimpleTable(numpy.array([0.5, 0.6]), numpy.array([0, 1]), numpy.array([1, 1]),
                        numpy.array([1, 2]), numpy.array([1.0, 1.0]))
    params = dict(transit_depth_min=1e-5, R_star_min=0.13, R_star_max=3.5, M_star_min=0.1,
                  M_star_max=1.0, T0_fit_margin=0.01)
    with pytest.raises(RuntimeError, match="too short"):
        gpu.search(t, y, numpy.full(9, 0.01), numpy.array([0.3]), table, params)


def test_survey_batch_equals_individual_searches(gpu):
    """BASELINE config 5 in miniature: a batch on shared grids == one search per light curve."""
    from tls_amd import survey
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(4)])
    periods, chi2, row, depth = survey.search_batch(t, fluxes, context=gpu, **kw)
    assert chi2.shape == (4, 9679)
    for k in (0, 3):
        inp = synthetic.search_inputs(t, fluxes[k], **kw)
        one = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        numpy.testing.assert_array_equal(chi2[k], one[0])
        numpy.testing.assert_array_equal(row[k], one[1])
        numpy.testing.assert_array_equal(depth[k], one[2])
    assert int(numpy.argmin(chi2[0])) == 7738


def test_survey_1024_curves_vs_oracle(gpu, oracle_lib):
    """BASELINE config 5 at FULL size: 1024 light curves (seeds 0..1023 of config 2) through ONE
    tls_search_batch call on the full 9679-period grid; curves from different 32-curve launch groups
    (first/last of a group, first and last group) are compared with the ORACLE, period by period."""
    from tls_amd import survey
    n_curves = int(os.environ.get("TLS_SURVEY_CURVES", 1024))
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(n_curves)])
    periods, chi2, row, depth = survey.search_batch(t, fluxes, context=gpu, **kw)
    assert chi2.shape == (n_curves, 9679)
    assert int(numpy.argmin(chi2[0])) == 7738
    numpy.testing.assert_allclose(chi2[0].min(), 4101.6439079674, rtol=1e-10)
    checked = [s for s in (0, 1, 31, 32, 33, 511, 512, 777, 1000, 1023) if s < n_curves]
    assert len([s for s in checked if s > 0]) >= 8 or n_curves < 1024
    for s in checked:
        inp = synthetic.search_inputs(t, fluxes[s], **kw)
        want = oracle_search(oracle_lib, inp)
        assert_parity((chi2[s], row[s], depth[s]), want, len(t))
    # every curve carries the injected planet: the best period of each is near 10.123 d or an alias
    best = periods[numpy.argmin(chi2, axis=1)]
    ratio = best / 10.123
    near = numpy.min(numpy.abs(ratio[:, None] - numpy.array([1 / 3, 0.5, 1.0, 2.0, 3.0, 4.0])[None, :]), axis=1) < 0.01
    assert near.mean() > 0.95, near.mean()
    assert numpy.all(chi2 <= len(t)) and numpy.all(chi2 > 0)


@pytest.mark.parametrize("name,n_curves,stride,per_point", [("k2_90d", 35, 40, False), ("k2_90d", 5, 40, True),
                                                             ("tess_27d", 3, 100, False), ("tess_27d", 2, 100, True)])
def test_search_batch_groups_weights_and_tiled_layout(gpu, name, n_curves, stride, per_point, monkeypatch):
    """tls_search_batch shares the fold + sort of a period between the curves of a launch group:
    more curves than one group holds, per-point weights, and the HBM-slab layout all return what a
    search per light curve returns, bit for bit.  (Same kernel shape on both sides: a batch runs the
    one-workgroup-per-period kernel, so the single searches are kept from the two-role kernel, which a few dozen
    periods of a long series would otherwise take and which runs exact prefix-sum mode only -- 1e-10 apart.  Likewise the
    four-slot kernel of short series is kept out: the curves here differ in noise, so a single search and its group may
    take different kernels, and that one's values differ from the classic family's in the last bits; its own batch test is
    test_four_slot_kernel_batches_ties_and_the_series_it_does_not_fit.)"""
    gpu.set_options(split="0", slim="0")
    t, f0, kw = synthetic.config(name, seed=0)
    rng = numpy.random.RandomState(5)
    inputs = []
    for s in range(n_curves):
        f = synthetic.config(name, seed=s, sigma=synthetic.CONFIGS[name][2] * (1 + s % 3))[1]
        dy = rng.uniform(0.6, 1.7, len(f)) * synthetic.CONFIGS[name][2] if per_point else None
        inputs.append(synthetic.search_inputs(t, f, dy, **kw))
    first = inputs[0]
    sel = first["periods"][::stride]
    if name == "tess_27d":
        # ... and a period of 40 cadences: the phases pile up on 40 values of 486 points, the slab sort's bins overflow a
        # wavefront's window (384) and go to the workgroup's network, which writes the PERMUTATION in a batch
        sel = numpy.sort(numpy.append(sel, 40 * (t[1] - t[0])))
    y = numpy.stack([i["y"] for i in inputs])
    dy = numpy.stack([i["dy"] for i in inputs])
    chi2, row, depth = gpu.search_batch(first["t"], y, dy, sel, first["table"], first["params"])
    assert chi2.shape == (n_curves, len(sel))
    for k in sorted({0, 1, n_curves // 2, n_curves - 1}):
        one = gpu.search(first["t"], inputs[k]["y"], inputs[k]["dy"], sel, first["table"], first["params"])
        numpy.testing.assert_array_equal(chi2[k], one[0])
        numpy.testing.assert_array_equal(row[k], one[1])
        numpy.testing.assert_array_equal(depth[k], one[2])


# TLS_FUZZ_SEEDS="100-140" (or "5,6,7") adds seeds for a one-off longer sweep
def _extra_seeds():
    spec = os.environ.get("TLS_FUZZ_SEEDS", "")
    out = []
    for part in filter(None, spec.split(",")):
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


@pytest.mark.parametrize("seed", [2024, 7, 99] + _extra_seeds())
def test_randomised_configurations_vs_oracle(gpu, oracle_lib, seed):
    """Seeded sweep over sizes, cadences, noise levels, weights, T0 strides, depth thresholds and
    duration-grid steps: resident and tiled kernel variants, dense and strided T0 grids, uniform
    and per-point weights, all against the oracle."""
    rng = numpy.random.RandomState(seed)
    extreme = bool(os.environ.get("TLS_FUZZ_EXTREME"))   # one-off sweeps: unusual parameter values too
    n_cases = 0
    for case in range(24):
        span = float(rng.choice([8.0, 20.0, 45.0, 120.0]))
        cadence = int(rng.choice([24, 48, 96, 200]))
        n = int(span * cadence)
        if n < 400 or n > 30000:
            continue
        sigma = float(rng.choice([3e-5, 2e-4, 1e-3]))
        t = numpy.sort(3.0 + rng.uniform(0, span, n)) if case % 3 == 0 else numpy.linspace(3.0, 3.0 + span, n)
        if case % 4 == 1:  # a gap
            keep = numpy.ones(n, dtype=bool)
            keep[n // 3: n // 3 + n // 10] = False
            t = t[keep]
        per = float(rng.uniform(1.5, span / 3))
        from tls_amd import transit_model
        flux = transit_model.light_curve(t, t[0] + 0.3 * per, per, float(rng.uniform(0.01, 0.08)), 15, 89.5,
                                         0, 90, [0.4, 0.3], "quadratic")
        flux = flux + rng.normal(0, sigma, len(t))
        dy = None
        if case % 2 == 1:
            dy = rng.uniform(0.5, 2.0, len(t)) * sigma
        kwargs = dict(period_min=float(rng.uniform(0.7, 2.0)), period_max=float(rng.uniform(span / 4, span / 2)),
                      oversampling_factor=int(rng.choice([1, 2, 3])),
                      duration_grid_step=float(rng.choice([1.05, 1.1, 1.3] + ([1.02] if extreme else []))),
                      T0_fit_margin=float(rng.choice([0.0, 0.01, 0.05, 0.1] + ([0.3, 0.7, 1.0] if extreme else []))),
                      transit_depth_min=float(rng.choice([1e-6, 1e-5, 2e-4] + ([0.0, 1e-8] if extreme else []))))
        if kwargs["period_min"] >= kwargs["period_max"]:
            continue
        try:
            inp = synthetic.search_inputs(t, flux, dy, **kwargs)
        except ValueError:
            continue  # degenerate template table for this size (the reference raises too)
        sel = inp["periods"][:: max(1, len(inp["periods"]) // 150)]
        if os.environ.get("TLS_FUZZ_VERBOSE"):
            print("fuzz", seed, case, len(t), dy is not None, kwargs, int(inp["table"].width.max()), flush=True)
        got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
        if os.environ.get("TLS_FUZZ_VERBOSE"):
            print("fuzz kernel", gpu.last_kernel(), flush=True)
        want = oracle_search(oracle_lib, inp, periods=sel)
        assert_parity(got, want, len(inp["t"]), allow_tie=True)
        assert got[3]["evaluated_cells"] == int(want[3][1]), (case, kwargs)
        assert got[3]["inner_steps"] == int(want[3][2]), (case, kwargs)
        # the uncounted call may take the pruning kernel (noisy cases; TLS_PRUNE=1 forces it)
        plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        assert_parity(plain, want, len(inp["t"]), allow_tie=True)
        n_cases += 1
    assert n_cases >= 12


@pytest.mark.parametrize("name,stride", [("tess_27d", 40), ("kepler_4yr", 700)])
def test_slab_sort_paths_vs_oracle(gpu, oracle_lib, monkeypatch, name, stride):
    """The HBM-slab variant's two sort paths (the two-level sort with per-wave bins: the default; the general bucket sort
    through global memory: `sort2 = 0`, and the fallback of a period whose phases pile up beyond what the two-level sort
    stages) against the oracle on both sizes, evaluated-cell counts included."""
    inp = _inputs(name)
    sel = inp["periods"][::stride]
    want = oracle_search(oracle_lib, inp, periods=sel)
    for sort2 in (None, 0):
        gpu.set_options(sort2=sort2)
        got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
        assert not gpu.plan_info()["resident"]
        assert_parity(got, want, len(inp["t"]))
        assert got[3]["evaluated_cells"] == int(want[3][1])
        assert got[3]["inner_steps"] == int(want[3][2])


@pytest.mark.parametrize("name,stride,per_point,batch", [("tess_27d", 3, False, None), ("tess_27d", 7, True, "97"),
                                                         ("kepler_4yr", 300, False, "200"), ("kepler_4yr", 1500, True, None)])
def test_two_role_slab_kernel_equals_the_one_kernel_path_bit_for_bit(gpu, oracle_lib, monkeypatch, name, stride, per_point, batch):
    """Series in the HBM slab: the two-role kernel (every workgroup folds periods into per-period slabs, then searches
    (period, position tile) items; a tile's winner is published and the last item of a period compares them) against the
    one-workgroup-per-period kernel -- the same cells, values and tie rule, so chi2, row and depth must be the same BITS and
    the evaluated-cell and tap counts equal; more rounds than workgroups, several batches of slabs (`split_batch`), uniform
    and per-point weights; and both against the oracle.  Round 6: in the DEFAULT prefix-sum mode too (uniform weights) -- a
    period takes fast or exact mode by (light curve, period) alone in both kernels, a tile's X is formed from the tile's
    staged flux in both, a work item whose tile noted band windows forms the period's exact prefix sum itself and decides
    them (2 % of the TESS-size and 7 % of the Kepler-size periods).  Per-point weights: the two roles run exact mode only, so that comparison pins the
    one-kernel path to exact mode (`fast_slab = 0`) and checks its default against it to 1e-10."""
    t, f, kw = synthetic.config(name)
    dy = None
    if per_point:
        dy = numpy.random.RandomState(23).uniform(0.7, 1.6, len(f)) * synthetic.CONFIGS[name][2]
        gpu.set_options(fast_slab="0")
    inp = synthetic.search_inputs(t, f, dy, **kw)
    sel = inp["periods"][::stride]
    if batch:
        gpu.set_options(split_batch=batch)
    results = {}
    for mode, parts in (("0", None), ("1", None)):
        gpu.set_options(split=mode)
        counted = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
        assert not gpu.plan_info()["resident"]
        assert gpu.last_kernel() == ("slab+split" if mode == "1" else "slab")
        plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        again = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])   # flags and queues rewound
        for a, b in zip(plain[:3], again[:3]):
            numpy.testing.assert_array_equal(a, b)
        for a, b in zip(plain[:3], counted[:3]):
            numpy.testing.assert_array_equal(a, b)
        results[(mode, parts)] = counted
    for key in (("1", None),):
        for a, b in zip(results[("0", None)][:3], results[key][:3]):
            numpy.testing.assert_array_equal(a, b)
        assert results[("0", None)][3]["evaluated_cells"] == results[key][3]["evaluated_cells"]
        assert results[("0", None)][3]["inner_steps"] == results[key][3]["inner_steps"]
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert_parity(results[("1", None)], want, len(inp["t"]))
    assert results[("1", None)][3]["evaluated_cells"] == int(want[3][1])
    if per_point:
        # the one-kernel path as it runs by default (fast prefix-sum mode)
        gpu.set_options(fast_slab=None, split="0")
        fast = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
        numpy.testing.assert_array_equal(fast[1], results[("1", None)][1])
        numpy.testing.assert_allclose(fast[0], results[("1", None)][0], rtol=1e-10, atol=0)
        assert fast[3]["evaluated_cells"] == results[("1", None)][3]["evaluated_cells"]
        assert fast[3]["inner_steps"] == results[("1", None)][3]["inner_steps"]
    else:
        # ... and the all-exact plan through both kernels (what the two roles ran before round 6)
        gpu.set_options(fast_slab="0")
        exact = {}
        for mode in ("0", "1"):
            gpu.set_options(split=mode)
            exact[mode] = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        for a, b in zip(exact["0"][:3], exact["1"][:3]):
            numpy.testing.assert_array_equal(a, b)
        numpy.testing.assert_array_equal(exact["1"][1], results[("1", None)][1])
        numpy.testing.assert_allclose(exact["1"][0], results[("1", None)][0], rtol=1e-10, atol=0)


def test_a_short_slab_launch_takes_the_two_roles_by_itself_and_returns_the_full_grids_bits(gpu):
    """a launch whose last round of periods is partly filled (the share of one of eight ranks: 307 of
    2459 TESS-size periods on 256 workgroups) goes through (period, tile) work items without being asked, in the default
    prefix-sum mode, and every period comes out with the bits the full-grid search gives it; a launch of whole rounds and
    the full grid stay with the one-workgroup kernel."""
    inp = _inputs("tess_27d")
    whole = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
    assert gpu.last_kernel() == "slab"
    n_cu = gpu.plan_info()["n_blocks"]
    for share, kernel in ((slice(3, None, 8), "slab+split"), (slice(100, 100 + n_cu), "slab"), (slice(0, n_cu + n_cu // 2), "slab+split")):
        got = gpu.search(inp["t"], inp["y"], inp["dy"], numpy.ascontiguousarray(inp["periods"][share]), inp["table"], inp["params"])
        assert gpu.last_kernel() == kernel, (share, gpu.last_kernel())
        for a, b in zip(got[:3], whole[:3]):
            numpy.testing.assert_array_equal(a, b[share])


@pytest.mark.parametrize("name,stride,sigma,weights", [("tess_27d", 3, None, False), ("tess_27d", 11, 1000e-6, True),
                                                       ("kepler_4yr", 250, None, False), ("kepler_4yr", 2000, None, True)])
def test_fast_prefix_mode_in_the_slab_decides_the_reference_cells(gpu, oracle_lib, monkeypatch, name, stride, sigma, weights):
    """HBM-slab variant, fast prefix-sum mode (the default of the one-workgroup-per-period kernel since round 4): the same
    contract as in the LDS-resident kernel -- evaluated-cell and tap counts equal the oracle's and exact mode's
    (TLS_FAST_SLAB=0), rows identical, chi^2 within 1e-10 of exact mode; a period with a window inside the undecided band
    goes through the prefix sum and phase 3 again in exact mode, on the folded flux it kept (the band is hit by 2-10 % of
    the periods at these sizes: the second attempts are part of what is checked here); plain and counting kernels and a
    repeat of the call give the same bits."""
    t, f, kw = synthetic.config(name, sigma=sigma)
    dy = None
    if weights:
        dy = numpy.random.RandomState(7).uniform(0.7, 1.5, len(f)) * synthetic.CONFIGS[name][2]
    inp = synthetic.search_inputs(t, f, dy, **kw)
    sel = inp["periods"][::stride]
    gpu.set_options(prune="0")
    gpu.set_options(split="0")
    gpu.set_options(fast_slab="0")
    exact = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    gpu.set_options(fast_slab=None)
    fast = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    again = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    for a, b, c in zip(fast[:3], plain[:3], again[:3]):
        numpy.testing.assert_array_equal(a, b)
        numpy.testing.assert_array_equal(b, c)
    assert fast[3]["evaluated_cells"] == exact[3]["evaluated_cells"]
    assert fast[3]["inner_steps"] == exact[3]["inner_steps"]
    numpy.testing.assert_array_equal(fast[1], exact[1])
    numpy.testing.assert_allclose(fast[0], exact[0], rtol=1e-10, atol=0)
    numpy.testing.assert_allclose(fast[2], exact[2], rtol=0, atol=1e-12)
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert fast[3]["evaluated_cells"] == int(want[3][1])
    assert_parity(exact, want, len(inp["t"]))
    assert_parity(fast, want, len(inp["t"]))
    gpu.execute(phase_clock=True)
    retries = gpu.phase_cycles()["stat_exact_retries"]
    assert 0 <= retries <= max(8, len(sel) // 4), retries


@pytest.mark.parametrize("per_point", [False, True])
def test_quarter_million_points_vs_oracle(gpu, oracle_lib, per_point):
    """N = 250 560 (174 d at 1-min cadence): the widest trial windows (30 068 samples) are longer than an LDS
    tile can hold, so those rows are evaluated straight from the HBM slab ("oversize" rows) while the
    others go through the LDS tiles.  The reference has no size limit (core.py:96-188)."""
    t, f = synthetic.light_curve(174.0, 1440, 3e-4, per=7.77, rp=0.03, a=15)
    dy = None
    if per_point:
        dy = numpy.random.RandomState(4).uniform(0.7, 1.5, len(f)) * 3e-4
    inp = synthetic.search_inputs(t, f, dy)
    assert len(inp["t"]) == 250560 and int(inp["table"].width.max()) > 30000
    periods = inp["periods"]
    near = numpy.argsort(numpy.abs(periods - 7.77))[:6]
    sel = numpy.unique(numpy.concatenate([numpy.arange(0, len(periods), 450), near, [len(periods) - 1]]))
    got = gpu.search(inp["t"], inp["y"], inp["dy"], periods[sel], inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    want = oracle_search(oracle_lib, inp, periods=periods[sel])
    assert_parity(got, want, len(inp["t"]))
    assert got[3]["grid_cells"] == int(want[3][0])
    assert got[3]["evaluated_cells"] == int(want[3][1])
    assert got[3]["inner_steps"] == int(want[3][2])
    assert got[0].min() < len(inp["t"]) - 100          # the injected planet is found
    assert abs(periods[sel][int(numpy.argmin(got[0]))] - 7.77) < 0.02


@pytest.mark.parametrize("name,stride", [("tess_27d", 60), ("kepler_4yr", 9000)])
def test_large_series_with_per_point_weights(gpu, oracle_lib, name, stride):
    """Tiled (non-resident) variant with per-point dy: e*w and w staged per tile; for the
    Kepler-size window the prefix sum no longer fits next to them and is read from the slab."""
    t, f, kw = synthetic.config(name)
    rng = numpy.random.RandomState(17)
    dy = rng.uniform(0.7, 1.6, len(f)) * synthetic.CONFIGS[name][2]
    inp = synthetic.search_inputs(t, f, dy, **kw)
    sel = inp["periods"][::stride]
    got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert_parity(got, want, len(inp["t"]))
    assert got[3]["evaluated_cells"] == int(want[3][1])


@pytest.mark.parametrize("name,sigma,stride", [("k2_90d", None, 3), ("k2_90d", 500e-6, 3), ("k2_90d", 3000e-6, 7),
                                               ("tutorial01", 300e-6, 5), ("tess_27d", 1000e-6, 40)])
def test_pruning_kernel_is_exact(gpu, name, sigma, stride, monkeypatch):
    """The branch-and-bound variant (cell_bound in tls_kernels.hip.h) only skips cells that cannot
    win: per period it must return the same bits as the plain kernel, whatever the noise level and
    however aggressively it is switched on (TLS_PRUNE / TLS_PRUNE_MIN_LIVE are read at prepare time)."""
    inp = _inputs(name, sigma=sigma)
    sel = inp["periods"][::stride]
    gpu.set_options(prune="0")
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    gpu.set_options(prune="1")
    for min_live in ("0", "4000"):
        gpu.set_options(prune_min_live=min_live)
        pruned = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        # (a series in the HBM slab has no pruning instantiation -- round 6 dropped it: forced there, it never paid --:
        # the switch is ignored and the plain kernel runs)
        assert gpu.last_kernel() in (("resident+prune",) if gpu.plan_info()["resident"] else ("slab", "slab+split"))
        for x, y in zip(plain[:3], pruned[:3]):
            numpy.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("name,sigma,stride", [("k2_90d", None, 2), ("k2_90d", 100e-6, 3), ("k2_90d", 500e-6, 5),
                                               ("k2_90d", 3000e-6, 11), ("tutorial01", None, 3), ("tutorial01", 300e-6, 5)])
def test_fp32_screen_kernel_is_exact(gpu, name, sigma, stride, monkeypatch):
    """The fp32 screen (screen_cells in tls_kernels.hip.h: dot products of all cells in packed fp32 on the high halves of the
    samples, fp64 valuation -- by the plain kernel's own additions -- of the cells whose error interval can still hold the
    period's minimum) must return the plain kernel's bits, period by period, whatever the noise level; the host takes it
    only where it pays (TLS_SCREEN32=1/0 forces it on/off, read at prepare time).  The statistics of the phase-clock
    buffer say that the variant ran: every period values at least its winner."""
    inp = _inputs(name, sigma=sigma)
    sel = inp["periods"][::stride]
    gpu.set_options(prune="0")
    gpu.set_options(screen32="0")
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    gpu.set_options(screen32="1")
    screened = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    assert gpu.plan_info()["resident"]
    for x, y in zip(plain[:3], screened[:3]):
        numpy.testing.assert_array_equal(x, y)
    gpu.execute(phase_clock=True)
    stats = gpu.phase_cycles()
    assert stats["stat_screen_valued"] >= numpy.count_nonzero(plain[0] < len(inp["t"])) // 2, stats["stat_screen_valued"]
    again = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    for x, y in zip(screened[:3], again[:3]):
        numpy.testing.assert_array_equal(x, y)


def test_host_picks_the_kernel_variant_by_noise_level(gpu, monkeypatch):
    """Which of the three bit-identical variants of the LDS-resident kernel a launch takes is the host's choice from the
    expected passing fraction of the depth predicate (screen_pays / pruning_pays in tls_amd.hip; round-4 measurements in
    PERF_LOG.md): plain at 50 ppm, the fp32 screen at 100 ppm, pruning at 500 ppm.  The statistics slots of the phase-clock
    buffer tell which one ran."""
    gpu.set_options(prune=None)
    gpu.set_options(screen32=None)
    seen = {}
    for ppm in (50, 100, 500):
        inp = _inputs("k2_90d", sigma=ppm * 1e-6)
        sel = inp["periods"][::9]
        gpu.prepare(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
        gpu.execute(phase_clock=True)
        stats = gpu.phase_cycles()
        seen[ppm] = ("screen" if stats["stat_screen_valued"] > 0 else "pruning" if stats["stat_pruned_periods"] > 0 else "plain")
    assert seen == {50: "plain", 100: "screen", 500: "pruning"}, seen


def test_fp32_screen_on_ties_deep_transits_and_inadmissible_flux(gpu, monkeypatch):
    """Edge cases of the fp32 screen against the plain kernel, bit for bit: (i) a light curve built from a few repeated
    values -- thousands of trial cells with EQUAL statistics, so the workgroup's list of parked cells overflows and the
    overflow path (a lane values the cell on the spot) decides, by the reference's tie rule; (ii) a 2.9 % deep transit
    (|1 - flux| just below the 2^-5 the exact split allows: the error bound is at its widest); (iii) a 4 % deep one and
    (iv) a flux outside [0.5, 2]: the screen is not admissible and the launch takes the plain kernel even when forced;
    (v) a survey batch."""
    rng = numpy.random.RandomState(3)
    n = 2400
    t = numpy.linspace(1.0, 51.0, n)
    kw = dict(period_min=1.0, period_max=12.0, oversampling_factor=2)
    cases = []
    y = 1.0 - 1e-4 * ((numpy.arange(n) % 7 == 0).astype(float) + (numpy.arange(n) % 11 == 0))          # (i) ties
    cases.append(("ties", y))
    for depth, label in ((0.029, "deep"), (0.04, "too deep")):
        y = 1.0 + rng.normal(0, 1e-4, n)
        y[(t % 3.7) < 0.15] -= depth
        cases.append((label, y))
    y = 1.0 + rng.normal(0, 1e-4, n); y[100] = 2.5
    cases.append(("outlier", y))
    gpu.set_options(prune="0")
    for label, y in cases:
        inp = synthetic.search_inputs(t, y, **kw)
        gpu.set_options(screen32="0")
        plain = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        gpu.set_options(screen32="1")
        screened = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        for x, z in zip(plain[:3], screened[:3]):
            numpy.testing.assert_array_equal(x, z, err_msg=label)
        gpu.execute(phase_clock=True)
        valued = gpu.phase_cycles()["stat_screen_valued"]
        assert (valued > 0) == (label in ("ties", "deep")), (label, valued)
    # (v) a batch: every curve of a launch group through the screen
    ys = numpy.stack([1.0 + rng.normal(0, s, n) for s in (8e-5, 1.5e-4, 3e-4, 1e-4, 2e-4)])
    for k in range(len(ys)):
        ys[k][(t % (2.0 + k)) < 0.12] -= 0.002
    inp = synthetic.search_inputs(t, ys[0], **kw)
    assert len(inp["t"]) == n
    dy = numpy.stack([numpy.full(n, numpy.std(v)) for v in ys])
    gpu.set_options(screen32="0")
    plain = gpu.search_batch(inp["t"], ys, dy, inp["periods"], inp["table"], inp["params"])
    gpu.set_options(screen32="1")
    screened = gpu.search_batch(inp["t"], ys, dy, inp["periods"], inp["table"], inp["params"])
    for x, z in zip(plain, screened):
        numpy.testing.assert_array_equal(x, z)


@pytest.mark.parametrize("name,sigma,weights", [("k2_90d", None, False), ("k2_90d", 500e-6, False),
                                                ("tutorial01", None, True)])
def test_fast_prefix_mode_decides_the_reference_cells(gpu, oracle_lib, monkeypatch, name, sigma, weights):
    """LDS-resident kernel, fast prefix-sum mode: X = plain prefix sum of 1 - f instead of
    k - numpy.cumsum.  The SET of evaluated cells must still be the reference's, cell for cell (a period with a window
    inside the undecided band is searched again in exact mode): evaluated-cell and inner-step counts equal the oracle's
    and the exact mode's (TLS_EXACT_PREFIX=1), rows identical, chi^2 within 1e-10 of exact mode (its depth scale moves
    by <= 2^-52 * N), and exact mode itself is what every earlier round shipped (1e-13 from the oracle)."""
    t, f, kw = synthetic.config(name, sigma=sigma)
    dy = None
    if weights:
        dy = numpy.random.RandomState(5).uniform(0.7, 1.5, len(f)) * synthetic.CONFIGS[name][2]
    inp = synthetic.search_inputs(t, f, dy, **kw)
    sel = inp["periods"] if not weights else inp["periods"][::3]
    gpu.set_options(prune="0")
    gpu.set_options(exact_prefix="1")
    exact = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    gpu.set_options(exact_prefix="0")
    fast = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    assert gpu.plan_info()["resident"]
    assert fast[3]["evaluated_cells"] == exact[3]["evaluated_cells"]
    assert fast[3]["inner_steps"] == exact[3]["inner_steps"]
    numpy.testing.assert_array_equal(fast[1], exact[1])
    numpy.testing.assert_allclose(fast[0], exact[0], rtol=1e-10, atol=0)
    numpy.testing.assert_allclose(fast[2], exact[2], rtol=0, atol=1e-12)
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert fast[3]["evaluated_cells"] == int(want[3][1])
    assert_parity(exact, want, len(inp["t"]), tight=1e-12)
    assert_parity(fast, want, len(inp["t"]))
    # the phase clock's work statistics say how often the undecided band was hit
    gpu.execute(phase_clock=True)
    retries = gpu.phase_cycles()["stat_exact_retries"]
    assert 0 <= retries <= max(8, len(sel) // 20), retries


@pytest.mark.parametrize("n,cadences_per_day,commensurate", [
    (4320, 48, [30 / 48.0, 1.0, 2.5, 2.0, 10.0, 45.0]),            # LDS-resident: the six periods of the parity test above
    (7200, 48, [30 / 48.0, 1.0, 2.5, 2.0, 10.0, 45.0]),            # ... and in the four-slot kernel's 512-thread shape (piles of 240)
    (70128, 48, [78 / 48.0, 66.5 / 48.0, 1.0, 80 / 48.0, 131 / 48.0]),   # Kepler size: the slab sort's fallback
])
def test_commensurate_periods_cost_no_more_than_their_neighbours(gpu, oracle_lib, n, cadences_per_day, commensurate):
    """A trial period that is a multiple of the cadence folds a regularly sampled series onto a few dozen phase values;
    the bucket sort's in-bucket ranking by counting was quadratic in the pile (one such period of the Kepler-size grid:
    29 ms in one workgroup, 85x its neighbours).  Piled-up buckets now go through the workgroup's bitonic sort
    (sort_big_bucket): every commensurate period of the LDS-resident configuration within 3x the median period of the
    same search (4x under the four-slot kernel, whose pile path is quadratic in the pile), a Kepler-size one below 5 ms and within 10x; results unchanged (oracle)."""
    t = 3.0 + numpy.arange(n) / float(cadences_per_day)       # exact binary cadence
    y = 1 + numpy.random.RandomState(5).normal(0, 5e-5, n)
    kw = dict(period_min=0.5, period_max=400) if n > 10000 else {}
    inp = synthetic.search_inputs(t, y, **kw)
    ordinary = inp["periods"][:: max(1, len(inp["periods"]) // 300)]
    periods = numpy.sort(numpy.concatenate([ordinary, commensurate]))
    got = gpu.search(inp["t"], inp["y"], inp["dy"], periods, inp["table"], inp["params"])
    cycles = gpu.period_cycles().astype(float)
    special = numpy.isin(periods, commensurate)
    median = numpy.median(cycles[~special])
    worst = cycles[special].max()
    if n > 10000:
        # series in HBM: the piled-up bins of the two-level sort go to the workgroup's bitonic network in the staging area of
        # its first pass (round 5; before, such a period left for the general bucket sort through global memory: 2.4 ms,
        # 12x the median of fast-mode periods).  Measured now: 2.2-4.5 M cycles for a series that is ALL piles (48-133 phase
        # values: every point goes through the network, which is bound by the LDS pipe), 4-8x an ordinary period.
        assert worst < 4e-3 * 2.4e9            # shader cycles at <= 2.4 GHz: below 4 ms (round 1: 29 ms)
        assert worst <= 10.0 * median, (worst, median, periods[special][numpy.argmax(cycles[special])])
    else:
        # the four-slot kernel ranks a pile (a bucket beyond 8 points whose records tie on their key bits) on 64-bit keys formed
        # once per member, up to 16 piles side by side: quadratic in the pile -- 30 piles of 144 points (30 cadences) cost
        # 2.1-2.8 x an ordinary period, 120 piles of 36 1.4 x; before the pile path: 16 x.  (The classic kernel's bitonic
        # network, `slim = 0`: 1.7 x.)
        print("commensurate: worst %.0f cycles = %.2f x the median %.0f (%s)" % (worst, worst / median, median, gpu.last_kernel()))
        assert worst <= (4.0 if gpu.last_kernel().startswith("slim") else 3.0) * median, (worst, median, periods[special][numpy.argmax(cycles[special])])
    sel = numpy.nonzero(special)[0]
    want = oracle_search(oracle_lib, inp, periods=periods[sel])
    assert_parity(tuple(a[sel] for a in got[:3]), want, n)


def test_post_search_kernels_at_kepler_and_tess_size(gpu, oracle_lib):
    """The two post-search kernels at the sizes of BASELINE configs 3 and 4 (round 2 checked them at K2 size only):
    tls_t0_fit vs the oracle at N = 70 128 (250 trial epochs: the non-resident T0-fit path with its HBM slabs, a
    period commensurate with the cadence among them) and at N = 19 440; tls_spectra vs the oracle on the
    182 388-period Kepler chi^2 array, handed in and resident after the search."""
    for name, n_epochs, periods in (("kepler_4yr", 250, (10.12452, 78 / 48.0)), ("tess_27d", 300, (3.377,))):
        t, f, kw = synthetic.config(name)
        inp = synthetic.search_inputs(t, f, **kw)
        row = len(inp["rows"]) // 2
        signal = 1 - (1 - inp["rows"][row]) * 0.002
        roll = int(len(signal) / 2) + 1
        for period in periods:
            epochs = numpy.linspace(t.min(), t.min() + period, n_epochs)
            want = oracle_lib.t0_residuals(inp["t"], inp["y"], period, signal, epochs, roll)
            got = gpu.t0_fit_residuals(inp["t"], inp["y"], period, signal, epochs, roll)
            numpy.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
            assert int(numpy.argmin(got)) == int(numpy.argmin(want))
    # spectra on the Kepler-size chi^2 (kernel 90 = oversampling 3 x 30): the reductions of tls_spectra_head/tail run as
    # ONE workgroup over 182 388 values, the running median as 11 394 workgroups
    inp = _inputs("kepler_4yr")
    chi2 = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])[0]
    assert len(chi2) == 182388
    want = oracle_lib.spectra(chi2, 90)
    # numpy restatement of stats.py:105-132 (numpy sums pairwise, as the reference does; the oracle sums left to right,
    # and over 182 388 values that moves the mean of SR by ~5e-14 -- 3e-9 after the division by std SR)
    from tls_amd.helpers import running_median
    SR = numpy.min(chi2) / chi2
    sde_raw = (1 - numpy.mean(SR)) / numpy.std(SR)
    praw = SR - numpy.mean(SR)
    praw = praw * (sde_raw / numpy.max(praw))
    power = praw - running_median(praw, 91)
    power = power - numpy.mean(power)
    sde = numpy.max(power / numpy.std(power))
    power = power * (sde / numpy.max(power))
    for got in (gpu.spectra(90, chi2), gpu.spectra(90)):
        numpy.testing.assert_allclose(got[0], SR, rtol=1e-13)
        numpy.testing.assert_allclose(got[1], praw, rtol=1e-10, atol=2e-11)
        numpy.testing.assert_allclose(got[2], power, rtol=1e-10, atol=2e-11)
        numpy.testing.assert_allclose(got[3:], [sde_raw, sde], rtol=1e-11)
        assert int(numpy.argmax(got[2])) == int(numpy.argmax(power))
        numpy.testing.assert_allclose(got[0], want[0], rtol=1e-13)
        for x, y in zip(got[1:3], want[1:3]):
            numpy.testing.assert_allclose(x, y, rtol=1e-8, atol=1e-8)
        numpy.testing.assert_allclose(got[3:], want[3:], rtol=1e-9)
        assert int(numpy.argmax(got[2])) == int(numpy.argmax(want[2]))


@pytest.mark.parametrize("outlier,weights", [(3.0e4, False), (3.0e4, True), (2.0e6, False)])
def test_noted_band_windows_are_decided_on_the_exact_prefix_sum(gpu, oracle_lib, outlier, weights):
    """Series in the HBM slab, fast prefix-sum mode: a window whose mean depth is too close to transit_depth_min for the
    plain scan to call is NOTED and decided after the attempt on the period's exact prefix sum (band_window /
    resolve_band).  Forced here: the band's half-width is 1.25 * 2^-53 * (N + W) * max|flux|, so ONE
    wild flux value widens it until white noise puts windows inside -- at 3e4 a few dozen per period (noted, resolved by
    one prefix pass), at 2e6 more than the list holds (256: the period is searched again in exact mode).  Either way the
    evaluated cells, rows and winners are exact mode's and the oracle's; uniform and per-point weights; a batch of two
    light curves equals its single searches bit for bit."""
    n = 12000
    rng = numpy.random.RandomState(11)
    t = numpy.linspace(2.0, 62.0, n)
    y = 1.0 + rng.normal(0, 2e-4, n)
    y[(t % 7.3) < 0.2] -= 1.5e-3
    y[137] = outlier
    dy = rng.uniform(0.8, 1.3, n) * 2e-4 if weights else None
    inp = synthetic.search_inputs(t, y, dy, period_min=5.0, period_max=30.0, oversampling_factor=1, transit_depth_min=1e-4)
    sel = inp["periods"][:: max(1, len(inp["periods"]) // 60)]
    gpu.set_options(prune=0, split=0, exact_prefix=1)
    exact = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    gpu.set_options(exact_prefix=None, band_max=1e9)      # (no period starts in exact mode for the hits it expects)
    fast = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    assert not gpu.plan_info()["resident"]
    plain = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    for a, b in zip(fast[:3], plain[:3]):
        numpy.testing.assert_array_equal(a, b)
    assert fast[3]["evaluated_cells"] == exact[3]["evaluated_cells"]
    assert fast[3]["inner_steps"] == exact[3]["inner_steps"]
    numpy.testing.assert_array_equal(fast[1], exact[1])
    numpy.testing.assert_allclose(fast[0], exact[0], rtol=1e-9, atol=0)    # (the band, and with it the modes' distance, is wide here)
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert fast[3]["evaluated_cells"] == int(want[3][1])
    assert_parity(exact, want, n)
    numpy.testing.assert_array_equal(fast[1], want[1])
    numpy.testing.assert_allclose(fast[0], want[0], rtol=1e-9, atol=0)
    gpu.execute(phase_clock=True)
    stats = gpu.phase_cycles()
    assert stats["stat_exact_retries"] >= len(sel) // 2          # periods that went through a second pass ...
    if outlier < 1e6:
        assert 0 < stats["stat_screen_parked"] <= 256 * len(sel)   # ... for their noted windows (this slot counts them)
    else:
        assert stats["stat_screen_parked"] < 256 * len(sel) // 2   # ... or, the list overflowing, through a second search
    # a batch of two light curves: the same bits as the single searches
    y2 = inp["y"] + rng.normal(0, 1e-5, n)
    ys = numpy.stack([inp["y"], y2])
    dys = numpy.stack([inp["dy"], inp["dy"]])
    chi2, row, dep = gpu.search_batch(inp["t"], ys, dys, sel, inp["table"], inp["params"])
    numpy.testing.assert_array_equal(chi2[0], fast[0])
    numpy.testing.assert_array_equal(row[0], fast[1])
    one = gpu.search(inp["t"], y2, inp["dy"], sel, inp["table"], inp["params"])
    numpy.testing.assert_array_equal(chi2[1], one[0])
    numpy.testing.assert_array_equal(row[1], one[1])
    numpy.testing.assert_array_equal(dep[1], one[2])


@pytest.mark.parametrize("sigma,stride,outlier,name", [(None, 1, None, "k2_90d"), (300e-6, 11, None, "k2_90d"), (None, 7, 3.0e4, "k2_90d"),
                                                       (100e-6, 13, 2.0e6, "k2_90d"), (None, 3, None, "tutorial01"),
                                                       (None, 7, None, "lc_150d"), (None, 23, 3.0e4, "lc_150d")])
def test_four_slot_kernel_searches_the_reference_cells(gpu, oracle_lib, sigma, stride, outlier, name):
    """Short LDS-resident series, uniform weights: tls_slim_kernel (four 256-thread workgroups per CU, phase 3 on the
    prefix sum X alone, dot products by summation by parts) against the classic LDS-resident kernel
    in exact prefix-sum mode and the oracle.  Same evaluated cells and template taps, same rows; chi^2 and depth within
    what the two prefix-sum modes differ by.  A window the plain scan cannot decide is noted and decided on the exact
    prefix sum: ONE wild flux value widens the undecided band (1.25 * 2^-53 * (N + W) * max|flux|) until white noise puts
    windows inside -- a few dozen per period at 3e4, more than the list of 256 holds at 2e6 (the period is searched again
    in exact mode).  Tutorial 01 (100 d): three workgroups per CU.  150 d (7200 points; round 6): the kernel's 512-thread
    shape, two workgroups per CU, 14 index bits in a sort record -- where the classic kernel runs one 1024-thread workgroup."""
    slim_name = "slim512" if name == "lc_150d" else "slim"
    inp = _inputs(name) if sigma is None else _inputs(name, sigma=sigma)
    if outlier is not None:
        y = inp["y"].copy()
        y[137] = outlier
        t, _, kw = synthetic.config(name)
        inp = synthetic.search_inputs(inp["t"], y, None, **kw)
    sel = inp["periods"][::stride]
    args = (inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"])
    gpu.set_options(slim=0, prune=0, screen32=0, exact_prefix=1)
    classic = gpu.search(*args, count_work=True)
    assert gpu.last_kernel() == "resident"
    gpu.set_options(slim=1, exact_prefix=None)
    slim = gpu.search(*args, count_work=True)
    assert gpu.last_kernel() == slim_name, gpu.plan_info()
    plain = gpu.search(*args)
    assert gpu.last_kernel() == slim_name
    for a, b in zip(slim[:3], plain[:3]):
        numpy.testing.assert_array_equal(a, b)       # counting and plain instantiation: the same bits
    assert slim[3]["evaluated_cells"] == classic[3]["evaluated_cells"]
    assert slim[3]["inner_steps"] == classic[3]["inner_steps"]
    numpy.testing.assert_array_equal(slim[1], classic[1])
    wide = outlier is not None     # (the band, and with it the modes' distance, is wide then)
    numpy.testing.assert_allclose(slim[0], classic[0], rtol=1e-9 if wide else 1e-10, atol=0)
    numpy.testing.assert_allclose(slim[2], classic[2], rtol=0, atol=1e-9 if wide else 1e-12)
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert slim[3]["evaluated_cells"] == int(want[3][1])
    if not wide:
        assert_parity(slim, want, len(inp["t"]))
    else:
        numpy.testing.assert_array_equal(slim[1], want[1])
        numpy.testing.assert_allclose(slim[0], want[0], rtol=1e-9, atol=0)


def test_four_slot_kernel_batches_ties_and_the_series_it_does_not_fit(gpu, oracle_lib):
    """tls_slim_kernel: a survey batch equals its single searches bit for bit (the sort of a period is shared, a light curve
    with an undecided window goes through exact mode alone); duplicate and unsorted time stamps and a commensurate period
    keep numpy's stable order (32-bit sort keys, ties by the exact phase and the index); per-point weights and a series
    beyond a quarter of the LDS take the classic kernel."""
    from tls_amd import survey
    gpu.set_options(slim=1, prune=0, screen32=0)      # (at 300 ppm the host would take the fp32 screen of the classic kernel)
    t, f0, kw = synthetic.config("k2_90d", seed=0)
    fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(5)])
    periods, chi2, row, depth = survey.search_batch(t, fluxes, context=gpu, **kw)
    assert gpu.last_kernel() == "slim"
    for k in (0, 4):
        inp = synthetic.search_inputs(t, fluxes[k], **kw)
        one = gpu.search(inp["t"], inp["y"], inp["dy"], inp["periods"], inp["table"], inp["params"])
        assert gpu.last_kernel() == "slim"
        numpy.testing.assert_array_equal(chi2[k], one[0])
        numpy.testing.assert_array_equal(row[k], one[1])
        numpy.testing.assert_array_equal(depth[k], one[2])
    # ties: every time stamp twice, shuffled; periods that pile the phases up (cadence 0.02 d: 1.0 d = 50 cadences)
    rng = numpy.random.RandomState(3)
    tt = numpy.repeat(numpy.arange(1500) * 0.02 + 1.0, 2)
    yy = 1.0 + rng.normal(0, 3e-4, tt.size)
    yy[(tt % 2.7) < 0.12] -= 2e-3
    sh = rng.permutation(tt.size)
    inp = synthetic.search_inputs(tt[sh], yy[sh], None, period_min=0.9, period_max=9.0, oversampling_factor=2)
    sel = numpy.concatenate([inp["periods"][::5], [1.0, 2.0, 3.0, 1.5, 1.0 + 1e-9]])
    sel = sel[(sel >= inp["periods"].min()) & (sel <= inp["periods"].max())]
    got = gpu.search(inp["t"], inp["y"], inp["dy"], sel, inp["table"], inp["params"], count_work=True)
    assert gpu.last_kernel() == "slim"
    want = oracle_search(oracle_lib, inp, periods=sel)
    assert got[3]["evaluated_cells"] == int(want[3][1])
    assert_parity(got, want, tt.size)
    # per-point weights: the classic kernel
    dy = rng.uniform(0.7, 1.4, tt.size) * 3e-4
    inp_w = synthetic.search_inputs(tt[sh], yy[sh], dy, period_min=0.9, period_max=9.0, oversampling_factor=1)
    gpu.search(inp_w["t"], inp_w["y"], inp_w["dy"], inp_w["periods"][::9], inp_w["table"], inp_w["params"])
    assert gpu.last_kernel() == "resident"
    # 120 days at 30 min: one region no longer fits a quarter of the LDS -- the 512-thread shape, two workgroups per CU (round 6);
    # its survey batch equals the single searches, and ties / piles keep numpy's stable order with 14 index bits in a record
    n = 5760
    t2 = 1.0 + numpy.arange(n) / 48.0
    y2 = numpy.stack([1.0 + numpy.random.RandomState(40 + s).normal(0, 1e-4, n) for s in range(3)])
    per2, chi2_b, row_b, depth_b = survey.search_batch(t2, y2, context=gpu, period_min=1.0, period_max=30.0, oversampling_factor=1)
    assert gpu.last_kernel() == "slim512"
    for k in (0, 2):
        inp2 = synthetic.search_inputs(t2, y2[k], None, period_min=1.0, period_max=30.0, oversampling_factor=1)
        one2 = gpu.search(inp2["t"], inp2["y"], inp2["dy"], inp2["periods"], inp2["table"], inp2["params"])
        assert gpu.last_kernel() == "slim512"
        numpy.testing.assert_array_equal(chi2_b[k], one2[0])
        numpy.testing.assert_array_equal(row_b[k], one2[1])
        numpy.testing.assert_array_equal(depth_b[k], one2[2])
    tt3 = numpy.repeat(numpy.arange(3500) * 0.02 + 1.0, 2)          # 7000 points, every time stamp twice
    yy3 = 1.0 + rng.normal(0, 3e-4, tt3.size)
    yy3[(tt3 % 2.7) < 0.12] -= 2e-3
    sh3 = rng.permutation(tt3.size)
    inp3 = synthetic.search_inputs(tt3[sh3], yy3[sh3], None, period_min=0.9, period_max=9.0, oversampling_factor=1)
    sel3 = numpy.concatenate([inp3["periods"][::9], [1.0, 2.0, 3.0, 1.5, 1.0 + 1e-9]])
    sel3 = sel3[(sel3 >= inp3["periods"].min()) & (sel3 <= inp3["periods"].max())]
    got3 = gpu.search(inp3["t"], inp3["y"], inp3["dy"], sel3, inp3["table"], inp3["params"], count_work=True)
    assert gpu.last_kernel() == "slim512"
    want3 = oracle_search(oracle_lib, inp3, periods=sel3)
    assert got3[3]["evaluated_cells"] == int(want3[3][1])
    assert_parity(got3, want3, tt3.size)
    # a series whose region does not fit half the LDS either (N + W beyond ~9.9 k): the classic kernel / the slab
    n4 = 9500
    t4 = numpy.arange(n4) / 48.0
    inp4 = synthetic.search_inputs(t4, 1.0 + rng.normal(0, 1e-4, n4), None, period_min=1.0, period_max=30.0, oversampling_factor=1)
    gpu.search(inp4["t"], inp4["y"], inp4["dy"], inp4["periods"][::40], inp4["table"], inp4["params"])
    assert gpu.last_kernel() in ("resident", "slab", "slab+split")


def test_results_do_not_depend_on_what_the_lds_held_before(gpu):
    """No kernel may read LDS it has not written: on a fresh box the LDS holds whatever the previous tenant left, later
    processes mostly find their own data there -- which is how an unset entry (the weights' entry M of the classic
    LDS-resident kernel, read against a zero tap: 0 * NaN) survived four rounds and failed one test run in thirty.
    tls_debug_poison_lds fills every CU's LDS with NaN / -inf / all-ones words; searches before and after agree bit for
    bit -- every kernel variant, uniform and per-point weights, the post-search kernels through power()."""
    import tls_amd
    cases = []
    for name in SEARCH_GOLDENS:
        g, table, params = load_search_golden(name)
        cases.append((name, (g["t"], g["y"], g["dy"], g["periods"], table, params)))
    for name, sigma, stride, weights in (("k2_90d", None, 40, False), ("k2_90d", 200e-6, 40, False), ("k2_90d", 500e-6, 40, False),
                                         ("k2_90d", None, 40, True), ("tutorial01", None, 60, False), ("lc_150d", None, 90, False), ("tess_27d", None, 120, False),
                                         ("tess_27d", None, 120, True), ("kepler_4yr", None, 9000, False)):
        t, f, kw = synthetic.config(name, sigma=sigma)
        dy = numpy.random.RandomState(5).uniform(0.7, 1.5, len(f)) * synthetic.CONFIGS[name][2] if weights else None
        inp = synthetic.search_inputs(t, f, dy, **kw)
        cases.append(("%s %s %s" % (name, sigma, weights), (inp["t"], inp["y"], inp["dy"], inp["periods"][::stride], inp["table"], inp["params"])))
    for label, args in cases:
        for word in (0x7ff80000, 0xfff00000, 0xffffffff):
            for count in (False, True):
                before = gpu.search(*args, count_work=count)
                gpu.poison_lds(word)
                after = gpu.search(*args, count_work=count)
                for x, y in zip(before[:3], after[:3]):
                    numpy.testing.assert_array_equal(x, y, err_msg="%s after LDS words %#x (%s)" % (label, word, gpu.last_kernel()))
    t, f, kw = synthetic.config("k2_90d")
    model = tls_amd.transitleastsquares(t, f, verbose=False)
    first = model.power(verbose=False, show_progress_bar=False, context=gpu, **kw)
    gpu.poison_lds(0x7ff80000)
    second = model.power(verbose=False, show_progress_bar=False, context=gpu, **kw)
    for key in ("SDE", "period", "T0", "depth", "duration", "snr", "chi2_min"):
        assert first[key] == second[key], key
    numpy.testing.assert_array_equal(first["power"], second["power"])
    from tls_amd import survey
    fluxes = numpy.stack([synthetic.config("k2_90d", seed=s)[1] for s in range(3)])
    one = survey.power_batch(t, fluxes, context=gpu, **kw)
    gpu.poison_lds(0xfff00000)
    two = survey.power_batch(t, fluxes, context=gpu, **kw)
    for x, y in zip(one, two):
        numpy.testing.assert_array_equal(x, y)
