/*
 * tls_amd.h -- C ABI of the MI355X transit-least-squares search engine (libtls_amd.so).
 *
 * The reference (hippke/tls v1.0.31) has no FFI layer: its seam for this path is the
 * Python function
 *     search_period(period, t, y, dy, transit_depth_min, R_star_min, R_star_max,
 *                   M_star_min, M_star_max, lc_arr, lc_cache_overview, T0_fit_margin)
 *         -> [period, chi2, row, depth]                 (transitleastsquares/core.py:96-188)
 * mapped over all trial periods by power()              (transitleastsquares/main.py:140-185)
 * and re-sorted by period                               (transitleastsquares/main.py:190-196).
 * This library replaces exactly that: ONE batched call searches every period of one
 * light curve on the GPU and returns chi2/row/depth per period in the order of `periods`.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types; every array is contiguous float64 / int64, owned by the caller;
 *     the library copies host->device->host internally and keeps no caller pointer.
 *   - every function returning int returns TLS_OK (0) or a negative TLS_E_* code; the
 *     message is available from tls_last_error(ctx) (ctx may be NULL for create errors).
 *   - a tls_ctx is bound to one GPU and one HIP stream; it is not re-entrant.  Use one
 *     context per GPU (one process per GPU in multi-GPU runs).
 *   - there is NO CPU fallback: without a usable GPU tls_ctx_create fails.
 */
#ifndef TLS_AMD_H
#define TLS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes its layout or an entry point its signature (3: tls_counters has
 * five fields, tls_period_costs / tls_power_batch exist; 4: tls_options, tls_get_options / tls_set_options; 5: tls_options
 * keeps the two caller-facing switches, the developer switches moved behind tls_debug_set_switch, tls_period_costs takes
 * them as text).  A binding compares it with tls_abi_version(). */
#define TLS_AMD_ABI_VERSION 5

#define TLS_OK 0
#define TLS_E_ARG (-1)      /* invalid argument */
#define TLS_E_HIP (-2)      /* HIP runtime error (message has the HIP error string) */
#define TLS_E_NOMEM (-3)    /* host or device allocation failed */
#define TLS_E_STATE (-4)    /* call order violated (e.g. execute before prepare) */
#define TLS_E_RCCL (-5)     /* RCCL error */

typedef struct tls_ctx tls_ctx;

/* Optional work counters of one search (all per light curve). */
typedef struct tls_counters {
    int64_t grid_cells;      /* (period, duration, T0) cells enumerated: data independent */
    int64_t evaluated_cells; /* cells that passed mean > transit_depth_min (core.py:58) */
    int64_t inner_steps;     /* template samples multiplied (core.py:67-69) */
    int64_t pd_pairs;        /* (period, duration) pairs searched */
    int64_t issued_fma;      /* lane-FMAs the chi^2 phase issued for them (chunk padding, idle lanes
                                and unroll slack included): inner_steps / issued_fma = lane efficiency */
} tls_counters;

/* Template table = the reference's (lc_cache_overview, lc_arr) of transit.py:98-160,
 * flattened: row r is values[offset[r] .. offset[r]+length[r]), trial width width[r]
 * samples, depth-normalisation factor overshoot[r]. */
typedef struct tls_template {
    const double *values;
    const int64_t *offset;
    const int64_t *length;
    const int64_t *width;
    const double *overshoot;
    int64_t n_rows;
} tls_template;

/* Scalar parameters of search_period (core.py:96-109). */
typedef struct tls_params {
    double transit_depth_min;
    double R_star_min, R_star_max, M_star_min, M_star_max;
    double T0_fit_margin;
} tls_params;

/* Switches of a context a caller may set: -1 = the library decides.  They are part of what a prepared plan is keyed by
 * (tls_prepare plans again when they change).  Neither changes a result beyond the distance between the two prefix-sum
 * modes that DESIGN.md section 3 describes (1e-10 relative on chi^2 for a normalised flux; it scales with
 * max|flux| (N + W) 2^-53 / transit depth otherwise).  The developer / test switches that select kernel variants and
 * launch shapes for A/B runs are NOT part of this struct: tls_debug_set_switch below. */
typedef struct tls_options {
    int32_t exact_prefix;   /* 1: every period in exact prefix-sum mode (X = k - numpy.cumsum, bit for bit) */
    int32_t slim;           /* 0: never the four-slots-per-CU kernel of short LDS-resident series (the classic kernel instead) */
} tls_options;

/* ---- context ------------------------------------------------------------------- */
int tls_device_count(void);                 /* number of visible GPUs, <0 on error */
tls_ctx *tls_ctx_create(int device_id);     /* NULL on failure, see tls_last_error(NULL) */
void tls_ctx_destroy(tls_ctx *ctx);
const char *tls_last_error(const tls_ctx *ctx);
const char *tls_version(void);
int tls_abi_version(void);                  /* TLS_AMD_ABI_VERSION the library was built with */
/* "gfx950 ..." style description of the context's device (valid until ctx destroy) */
const char *tls_device_name(const tls_ctx *ctx);
/* the context's switches (see tls_options); tls_set_options drops a prepared plan (the next tls_prepare plans again) */
int tls_get_options(const tls_ctx *ctx, tls_options *out);
int tls_set_options(tls_ctx *ctx, const tls_options *opt);
/* Developer / test switches, by name (not part of the stable ABI; negative value = the library decides): exact_prefix, slim
 * (the two of tls_options), prune, screen32, no_screen (kernel variant of an LDS-resident series), fast_slab, x_staged,
 * sort2, split, split_batch (series in the HBM slab: prefix-sum mode, sort, two-role kernel), threads, blocks, plan_threads
 * (launch shape, host planning), t0_rot (0: the final T0 fit checks every pair of every epoch), prune_min_live, band_max.  A context starts with the values
 * of the TLS_<NAME> environment variables, read once per process; no call reads the environment after that.
 * tls_debug_get_switches writes all of them as "name=value,name=value" (ctx NULL: the process's) -- the text
 * tls_period_costs takes, so that the planning call prices the kernel the searching context will run. */
int tls_debug_set_switch(tls_ctx *ctx, const char *name, double value);
int tls_debug_get_switches(const tls_ctx *ctx, char *out, int64_t capacity);

/* ---- one-shot search: replaces main.py:140-196 for one light curve -------------- */
/* out_chi2/out_row/out_depth have n_periods entries, index i belongs to periods[i].
 * counters may be NULL. */
int tls_search(tls_ctx *ctx, const double *t, const double *y, const double *dy, int64_t n,
               const double *periods, int64_t n_periods, const tls_template *tmpl,
               const tls_params *params, double *out_chi2, int64_t *out_row,
               double *out_depth, tls_counters *counters);

/* ---- survey mode: many light curves on the SAME time stamps, grids and template ---- */
/* y and dy hold n_curves rows of n values (row-major); the outputs n_curves rows of n_periods.
 * The plan is prepared once (tls_prepare with the first curve) and every further curve only
 * replaces the flux and weights (tls_update_flux), exactly as the staged calls below would.
 * All curves must have the same weight structure (all uniform dy or all per-point dy).
 * Replaces one main.py:140-196 pass per light curve (BASELINE config 5). */
int tls_search_batch(tls_ctx *ctx, const double *t, const double *y, const double *dy, int64_t n,
                     int64_t n_curves, const double *periods, int64_t n_periods,
                     const tls_template *tmpl, const tls_params *params, double *out_chi2,
                     int64_t *out_row, double *out_depth);

/* ---- staged search: same result, inputs resident in HBM between the stages ------ */
/* prepare: validate, build the device-side work list, upload everything (one pinned staging buffer, one
 * asynchronous copy; the call does not wait for the device).  A call with the same t, periods, template and
 * parameters as the plan the context already holds (compared byte for byte) only replaces the flux: that is
 * what a survey and repeated power() calls do, and it costs two passes over y and one upload. */
int tls_prepare(tls_ctx *ctx, const double *t, const double *y, const double *dy, int64_t n,
                const double *periods, int64_t n_periods, const tls_template *tmpl,
                const tls_params *params);
/* replace only the light-curve values of a prepared search (same t, n, grids, template):
 * survey mode streams many light curves through one prepared plan. */
int tls_update_flux(tls_ctx *ctx, const double *y, const double *dy);
/* execute: enqueue the search kernels on the context's stream (asynchronous).
 * count_work & 1 also accumulates evaluated_cells/inner_steps (slower: a kernel instantiation of its own keeps
 * the counters -- the plain one does not carry them --, and counting means evaluating every cell that passes
 * the depth predicate, so the pruning kernel variant is not used). */
int tls_execute(tls_ctx *ctx, int count_work);
/* developer instrumentation: tls_execute(ctx, 2) makes thread 0 of every workgroup stamp
 * the shader clock at phase boundaries -- in the instrumented and the checked library (make clocks / make debug;
 * the shipped library compiles the clock marks out and reports zeros for them, the statistics slots stay);
 * this returns the per-phase cycle sums (up to 26 slots:
 * fold+count, scan, scatter, rank, gather+patch, cumsum, batch prefix, chi2, e-convert,
 * strided predicate, two event counters (cumsum blocks, cumsum fallbacks), tile staging,
 * dense predicate, six cumsum sub-phases, two barrier waits, four pruning steps; names in
 * tls_amd/_lib.py::phase_cycles). */
int tls_debug_phase_cycles(tls_ctx *ctx, uint64_t *cycles, int n);
/* developer/test entry: the kernel's exact parallel evaluation of the sequential fp64 prefix
 * sum (numpy.cumsum order, helpers.py:72) on an arbitrary series of non-negative values;
 * out has count + 1 entries, out[0] = 0. */
int tls_debug_cumsum(tls_ctx *ctx, const double *f, int64_t count, double *out, int threads);
/* developer/test entry: runs the prepared search once more and returns, for every period of the plan, the
 * folded flux y[argsort(phase, stable)] exactly as the kernel's sort left it (core.py:120-123) -- the direct
 * check of the sort order, ties included; out holds n_periods * n doubles (capacity in doubles). */
int tls_debug_folded(tls_ctx *ctx, double *out, int64_t capacity);
/* developer/test entry: likewise the prefix sum C[0..M] of the patched folded flux of every period
 * (helpers.py:72 numpy.cumsum order, core.py:126 patch), M = n + widest window; *row_length = M + 1 (out may be
 * NULL to query it), out holds n_periods * row_length doubles. */
int tls_debug_prefix(tls_ctx *ctx, double *out, int64_t capacity, int64_t *row_length);
/* developer instrumentation: runs the prepared search once more and returns the shader cycles the workgroup that
 * searched period p spent on it (one entry per period of the plan, in `periods` order): the per-period cost the
 * shard cost model of tls_amd/shard.py is fitted to (tools/gpu_cost_model.py). */
int tls_debug_period_cycles(tls_ctx *ctx, uint64_t *cycles, int64_t capacity);
/* developer instrumentation: the debug build (make -C tls_amd/csrc debug) tests every hand-computed
 * bound of the search kernel on the device and counts violations per check (names in
 * tls_amd/_lib.py::check_counts); returns 1 from a checked build, 0 (all counts zero) otherwise. */
int tls_debug_check_counts(tls_ctx *ctx, uint64_t *counts, int n);
/* Test entry: fills the LDS of every CU with `word` (0x7ff80000 pairs read as fp64 NaNs) and the context's per-workgroup
 * scratch in HBM (slabs, lists, stashed orders) with all-ones bytes, on the context's stream -- what a previous tenant of
 * the GPU may have left there.  A search after it must return the bits of a search before it
 * (tests/test_gpu_parity.py: no kernel reads memory it has not written). */
int tls_debug_poison_lds(tls_ctx *ctx, uint32_t word);
/* developer instrumentation: host wall time (ms) of every group of 32 light curves of the context's last tls_power_batch /
 * tls_search_batch call (a stall in one group -- a transfer, a wait, a T0-fit launch -- shows here; bench.py prints the
 * largest and the median).  Returns the number of groups; out may be NULL. */
int tls_debug_batch_group_ms(const tls_ctx *ctx, double *out, int64_t capacity);
/* block until the stream is idle */
int tls_synchronize(tls_ctx *ctx);
/* fetch: copy results (and counters, may be NULL) back; synchronises. */
int tls_fetch(tls_ctx *ctx, double *out_chi2, int64_t *out_row, double *out_depth,
              tls_counters *counters);
/* run `reps` executes back to back and report the mean duration of ONE execute in
 * milliseconds, measured with HIP events on the context's stream. */
int tls_execute_timed(tls_ctx *ctx, int reps, double *ms_per_execute);
/* Sum of the search-kernel durations of all executes since the last reset, from HIP
 * events recorded around each launch on the context's stream; synchronises.  The events live in
 * a ring of 64 pairs: with more launches since the last reset, the most recent 64 are summed and
 * *launches says how many that was. */
int tls_kernel_timing(tls_ctx *ctx, int reset, double *total_ms, int64_t *launches);
/* data-independent work of the prepared search (no device work needed). */
int tls_plan_info(const tls_ctx *ctx, tls_counters *counters, int64_t *lds_bytes,
                  int64_t *n_blocks, int64_t *resident /* 1: folded series kept in LDS */);
/* Which search kernel the context's last tls_execute launched: "resident" (LDS-resident series, two or one workgroups
 * per CU), "resident+prune", "resident+screen32", "slim" (LDS-resident, four 256-thread workgroups per CU), "slim512"
 * (the same kernel as two 512-thread workgroups per CU: series of 5280-8640 points), "slab", "slab+split"; "" before the
 * first launch.  The string is static. */
const char *tls_last_kernel(const tls_ctx *ctx);

/* ---- final T0 fit: the batched counterpart of stats.py:135-204 ------------------------ */
/* For every trial epoch: fold (t, y) at (period, epoch), stable sort by phase, roll the folded
 * flux by `roll` cadences, and return the chi^2 of `signal` (the template row already scaled to
 * the fitted depth, `dur` samples) over the first `dur` samples plus the out-of-transit
 * residuals, both weighted by 1/flux^2 of the twice-rolled flux (the reference's own weighting,
 * stats.py:183-195).  The caller takes the FIRST minimum of out_residuals (stats.py:199-201). */
int tls_t0_fit(tls_ctx *ctx, const double *t, const double *y, int64_t n, double period,
               const double *signal, int64_t dur, const double *epochs, int64_t n_epochs,
               int64_t roll, double *out_residuals);

/* ---- pink noise of the out-of-transit flux: the counterpart of stats.py:72-77 (pink_noise) ---- */
/* mean over all n - width + 1 windows of `width` consecutive points of numpy.std(window) / width ** 0.5, with the reference's
 * roundings: every window's two sums in numpy's pairwise association, the running total added window by window from the left.
 * `root_width` = width ** 0.5 as the caller's language forms it (Python's float power; sqrt(width) otherwise).  data finite;
 * 1 <= width <= n.  (Called per power() by the statistics layer: 3 ms of numpy at TESS size.) */
int tls_pink_noise(tls_ctx *ctx, const double *data, int64_t n, int64_t width, double root_width, double *out);

/* ---- SDE spectra: the counterpart of stats.py:105-132 (spectra) with helpers.py:93-108 ---- */
/* SR, power_raw (scaled to SDE_raw) and power (running-median detrended, scaled to SDE) for the
 * chi^2 of every period; out_sde[0] = SDE_raw, out_sde[1] = SDE.  chi2 == NULL takes the chi^2 array
 * that is still resident on the device from the last tls_execute / tls_search (n is then ignored);
 * `kernel` = oversampling_factor * SDE_MEDIAN_KERNEL_SIZE as the reference forms it (made odd here,
 * stats.py:114-117). */
int tls_spectra(tls_ctx *ctx, const double *chi2, int64_t n, int64_t kernel, double *out_SR,
                double *out_power_raw, double *out_power, double *out_sde);

/* ---- survey-mode power(): search + spectra + final T0 fit of many light curves, all on the device --------- */
/* What main.py:198-283 derives for ONE light curve from the search results -- SDE and SDE_raw (stats.py:105-132),
 * the period and depth at the peak of the detrended power, the template row at the chi^2 minimum, the mid-transit
 * time of the final T0 fit (stats.py:135-204) -- for every light curve of a batch that shares t, the grids and the
 * template (the contract of tls_search_batch).  80 bytes come back per light curve instead of three arrays of
 * n_periods entries; the per-period arrays (chi2/row/depth together, and the detrended power) on request.
 * `median_kernel` = oversampling_factor * SDE_MEDIAN_KERNEL_SIZE (as tls_spectra). */
typedef struct tls_power_summary {
    double SDE, SDE_raw, chi2_min, period, T0, depth;
    int64_t index_best;    /* numpy.argmin(chi2)  (main.py:198) */
    int64_t index_power;   /* numpy.argmax(power) (main.py:270) */
    int64_t best_row;      /* template row at index_best: lc_cache_overview["duration"][best_row] is the duration */
    int64_t no_fit;        /* 1: max(chi2) == min(chi2), "no transit was fit" (main.py:203): SDE 0, depth 1, period NaN */
} tls_power_summary;
int tls_power_batch(tls_ctx *ctx, const double *t, const double *y, const double *dy, int64_t n,
                    int64_t n_curves, const double *periods, int64_t n_periods,
                    const tls_template *tmpl, const tls_params *params, int64_t median_kernel,
                    tls_power_summary *out_summary,
                    double *out_chi2 /* [n_curves][n_periods] or NULL */, int64_t *out_row /* with out_chi2 */,
                    double *out_depth /* with out_chi2 */, double *out_power /* [n_curves][n_periods] or NULL */,
                    double *out_SR /* [n_curves][n_periods] or NULL */, double *out_power_raw /* likewise */);
/* (ABI 5: out_SR / out_power_raw.  With n_curves = 1 and all arrays this is the device part of the drop-in power() call:
 * search, spectra, pick, the final T0 fit's trial epochs and scaled template formed on the device, all fits of a group in one
 * launch, ONE wait per group of 32 light curves.) */

/* ---- host-only planning (no GPU needed) ------------------------------------------ */
/* Trial cells (duration x T0 positions) each period will enumerate: the data-independent
 * cost used to place shard boundaries and to report cells/s.  Mirrors core.py:50-57,143-156. */
int tls_grid_cells(const double *t, int64_t n, const double *periods, int64_t n_periods,
                   const tls_template *tmpl, const tls_params *params,
                   int64_t *cells_per_period);
/* The cost of every period for the placement of shard boundaries (tls_amd/shard.py): its trial cells (as
 * tls_grid_cells), its expected template taps -- per in-range duration: trial positions x template length x the
 * fraction of white-noise windows of scatter `sigma` whose mean depth exceeds transit_depth_min (core.py:58), the
 * cells the sliding chi^2 of core.py:59-74 is evaluated for (sigma <= 0: all of them) -- and (time_per_period, may be
 * NULL) the modelled search time of the period in shader cycles of the kernel variant tls_prepare would choose:
 * a fixed part per period (fold, sort, prefix sum: O(n) whatever the duration window) + a part per trial cell
 * (depth predicate) + a part per expected tap, coefficients measured on an MI355X. */
int tls_period_costs(const double *t, int64_t n, const double *periods, int64_t n_periods,
                     const tls_template *tmpl, const tls_params *params, double sigma,
                     int64_t *cells_per_period, double *taps_per_period, double *time_per_period,
                     int64_t *workgroups_in_flight /* periods one MI355X searches side by side, may be NULL */,
                     const char *switches /* "name=value,..." of the context that will search (tls_debug_get_switches):
                                             the kernel variant and prefix-sum mode follow them; NULL: the process's */);

/* ---- multi-GPU: period grid sharded over ranks, one RCCL all-gather at the end --- */
/* rank 0 creates the 128-byte id and hands it to the other ranks by any host channel */
int tls_comm_unique_id(char id_out[128]);
int tls_comm_init(tls_ctx *ctx, int n_ranks, int rank, const char id[128]);
int tls_comm_destroy(tls_ctx *ctx);
/* What RCCL itself reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): an N-rank job
 * prints these, so that a run that silently used one rank cannot pass for an N-GPU run.  No communicator: 0, -1, -1. */
int tls_comm_info(tls_ctx *ctx, int *n_ranks, int *rank, int *device);
/* All-gather of the prepared search's device-resident results: every rank contributes its
 * shard (count_per_rank entries, zero padded) and receives n_ranks*count_per_rank entries of
 * chi2/row/depth in rank order.  One ncclAllGather over a packed 24 B/period buffer. */
int tls_comm_allgather_results(tls_ctx *ctx, int64_t count_per_rank, double *all_chi2,
                               int64_t *all_row, double *all_depth);
/* The same in two halves: _device enqueues pack + ncclAllGather on the context's stream and
 * returns (the gathered batch stays in HBM on every rank, the next search may be enqueued
 * right behind it); _fetch_gathered copies the most recent gather to the host and unpacks it. */
int tls_comm_allgather_device(tls_ctx *ctx, int64_t count_per_rank);
int tls_comm_fetch_gathered(tls_ctx *ctx, int64_t count_per_rank, double *all_chi2,
                            int64_t *all_row, double *all_depth);
/* Survey mode (every rank searches its own light curves): stage the latest results as slot
 * `slot` of `n_slots` with device copies on the stream -- no communication, no rank waits for
 * another -- and exchange all slots with ONE ncclAllGather at the end; _fetch_staged copies one
 * slot of that gather (n_ranks*count_per_rank entries, rank order) to the host. */
int tls_comm_stage_results(tls_ctx *ctx, int64_t count_per_rank, int64_t slot, int64_t n_slots);
int tls_comm_allgather_staged(tls_ctx *ctx, int64_t count_per_rank, int64_t n_slots);
int tls_comm_fetch_staged(tls_ctx *ctx, int64_t count_per_rank, int64_t n_slots, int64_t slot,
                          double *all_chi2, int64_t *all_row, double *all_depth);
/* small host-value collectives used by the bench harness (barrier, max over ranks) */
int tls_comm_barrier(tls_ctx *ctx);
int tls_comm_max(tls_ctx *ctx, double *value_inout);

#ifdef __cplusplus
}
#endif
#endif /* TLS_AMD_H */
