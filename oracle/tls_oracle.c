/*
 * tls_oracle.c -- CPU restatement of the reference's per-period transit search.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker or
 * the timed CPU baseline.  Nothing under tls_amd/ imports, links or executes it.
 *
 * What it restates (all paths relative to /root/reference/transitleastsquares):
 *   core.py:96-188  search_period                      -> oracle_search_one()
 *   core.py:15-18   foldfast                           -> fold step
 *   core.py:120     numpy.argsort(kind="mergesort")    -> stable_argsort()
 *   core.py:21-25   edge_effect_correction             -> edge_effect()
 *   helpers.py:70-73 running_mean                      -> window_mean()
 *   core.py:79-93   out_of_transit_residuals           -> ootr()
 *   core.py:28-76   lowest_residuals_in_this_duration  -> lowest_residuals()
 *   grid.py:9-32    T14                                -> t14()
 *   main.py:140-196 dispatch over periods + ordered gather -> tls_oracle_search()
 *   stats.py:135-204 final_T0_fit (+ core.py:9-12 fold)  -> tls_oracle_final_t0_fit()
 *   stats.py:105-132 spectra, helpers.py:93-108 running_median -> tls_oracle_spectra()
 * It follows the reference statement by statement: sequential summation order,
 * strict '<' comparisons (first trial wins ties), the same predicate, the same
 * 'datapoints' baseline.  Known deviations, all at the 1e-13 level: the reference
 * is compiled by numba with fastmath=True (not bit-reproducible itself, SURVEY.md
 * Appendix C.18) and numpy.sum is pairwise where this file sums left to right.
 *
 * Parity pin: checked against (a) per-period outputs of the unmodified reference
 * run in the build container (tests/golden/search_*.npz, made by
 * tools/gen_golden.py) and (b) the reference's own known-answer tests
 * (tests/test_synthetic.py:50 chi2_min = 8831.654060613922 etc.), see
 * tests/test_oracle_golden.py and tests/test_power_host.py (constants in tests/pins.py).
 * The T0 fit is pinned by tests/golden/t0fit_*.npz: the per-epoch residuals the unmodified
 * reference's final_T0_fit computes, captured by tools/gen_golden_t0fit.py.
 *
 * Build: see oracle/Makefile  (gcc -O3 -ffp-contract=off -fopenmp -shared -fPIC; `make fast`: -O3 -ffast-math, timing only).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* physical constants, tls_constants.py:20-25 */
#define TLS_G 6.673e-11
#define TLS_R_SUN 695508000.0
#define TLS_R_JUP 69911000.0
#define TLS_M_SUN (1.989 * 1e30)
#define TLS_SECONDS_PER_DAY 86400.0
#define TLS_SIGNAL_DEPTH 0.5                 /* tls_constants.py:71 */
#define TLS_FRACTIONAL_TRANSIT_DURATION_MAX 0.12 /* tls_constants.py:78 */

typedef struct {
    int64_t grid_cells;      /* trial (period, duration, T0) cells enumerated */
    int64_t evaluated_cells; /* cells that passed the depth predicate, core.py:58 */
    int64_t inner_steps;     /* template samples summed, core.py:67-69 */
} oracle_counters;

/* grid.py:9-32 */
static double t14(double R_s, double M_s, double P, int small)
{
    double T14max, result;
    P = P * TLS_SECONDS_PER_DAY;
    R_s = TLS_R_SUN * R_s;
    M_s = TLS_M_SUN * M_s;
    if (small)
        T14max = R_s * pow((4 * P) / (M_PI * TLS_G * M_s), 1.0 / 3);
    else
        T14max = (R_s + 2 * TLS_R_JUP) * pow((4 * P) / (M_PI * TLS_G * M_s), 1.0 / 3);
    result = T14max / P;
    if (result > TLS_FRACTIONAL_TRANSIT_DURATION_MAX)
        result = TLS_FRACTIONAL_TRANSIT_DURATION_MAX;
    return result;
}

/* core.py:120 -- stable ascending argsort (top-down merge sort; ties keep input order) */
static void merge_sort_idx(const double *key, int64_t *idx, int64_t *tmp, int64_t lo, int64_t hi)
{
    int64_t mid, i, j, k;
    if (hi - lo < 2)
        return;
    mid = lo + (hi - lo) / 2;
    merge_sort_idx(key, idx, tmp, lo, mid);
    merge_sort_idx(key, idx, tmp, mid, hi);
    i = lo; j = mid; k = lo;
    while (i < mid && j < hi) {
        if (key[idx[j]] < key[idx[i]]) tmp[k++] = idx[j++];
        else tmp[k++] = idx[i++];
    }
    while (i < mid) tmp[k++] = idx[i++];
    while (j < hi) tmp[k++] = idx[j++];
    memcpy(idx + lo, tmp + lo, (size_t)(hi - lo) * sizeof(int64_t));
}

/* core.py:21-25 */
static double edge_effect(const double *flux, const double *patched, const double *dy,
                          const double *inv_sq_patched_dy, int64_t n, int64_t np)
{
    double regular = 0.0, patched_sum = 0.0;
    int64_t i;
    for (i = 0; i < n; i++)
        regular += ((1 - flux[i]) * (1 - flux[i])) * 1 / (dy[i] * dy[i]);
    for (i = 0; i < np; i++)
        patched_sum += ((1 - patched[i]) * (1 - patched[i])) * inv_sq_patched_dy[i];
    return patched_sum - regular;
}

/* core.py:79-93 */
static void ootr(const double *data, int64_t width, const double *w, int64_t np, double *chi2)
{
    double fullsum = 0.0, window = 0.0;
    int64_t i, n_out = np - width + 1;
    for (i = 0; i < np; i++)
        fullsum += ((1 - data[i]) * (1 - data[i])) * w[i];
    for (i = 0; i < width; i++)
        window += ((1 - data[i]) * (1 - data[i])) * w[i];
    chi2[0] = fullsum - window;
    for (i = 1; i < n_out; i++) {
        int64_t vis = i - 1, invis = i - 1 + width;
        double add_left = (1 - data[vis]) * (1 - data[vis]) * w[vis];
        double remove_right = (1 - data[invis]) * (1 - data[invis]) * w[invis];
        chi2[i] = chi2[i - 1] + add_left - remove_right;
    }
}

/* core.py:28-76; returns the lowest statistic, sets *depth (row is the caller's) */
static double lowest_residuals(const double *mean, int64_t n_mean, double transit_depth_min,
                               const double *patched, int64_t duration, const double *signal,
                               int64_t signal_len, const double *w, double overshoot,
                               const double *ootr_arr, double edge_corr, int64_t datapoints,
                               double T0_fit_margin, double *depth_out, oracle_counters *cnt)
{
    double best = (double)datapoints; /* straight-line fit baseline, core.py:46 */
    double best_depth = 0.0;
    int64_t xth_point = 1, i, j;
    if (T0_fit_margin > 0 && (double)duration > T0_fit_margin) {
        T0_fit_margin = 1 / T0_fit_margin;
        xth_point = (int64_t)((double)duration / T0_fit_margin);
        if (xth_point < 1)
            xth_point = 1;
    }
    for (i = 0; i < n_mean; i++) {
        if (i % xth_point == 0)
            cnt->grid_cells++;
        if (mean[i] > transit_depth_min && i % xth_point == 0) {
            const double *data = patched + i;
            const double *dyw = w + i;
            double target_depth = mean[i] * overshoot;
            double scale = TLS_SIGNAL_DEPTH / target_depth;
            double reverse_scale = 1 / scale;
            double intransit = 0.0, stat;
            for (j = 0; j < signal_len; j++) {
                double sigi = (1 - signal[j]) * reverse_scale;
                double r = data[j] - (1 - sigi);
                intransit += (r * r) * dyw[j];
            }
            cnt->evaluated_cells++;
            cnt->inner_steps += signal_len;
            stat = intransit + ootr_arr[i] - edge_corr;
            if (stat < best) {
                best = stat;
                best_depth = 1 - target_depth;
            }
        }
    }
    *depth_out = best_depth;
    return best;
}

typedef struct {
    double *phase, *flux, *dys, *patched, *patched_dy, *w, *cumsum, *mean, *ootr;
    int64_t *idx, *tmp;
} scratch;

static int scratch_alloc(scratch *s, int64_t n, int64_t np)
{
    s->phase = (double *)malloc((size_t)n * sizeof(double));
    s->flux = (double *)malloc((size_t)n * sizeof(double));
    s->dys = (double *)malloc((size_t)n * sizeof(double));
    s->patched = (double *)malloc((size_t)np * sizeof(double));
    s->patched_dy = (double *)malloc((size_t)np * sizeof(double));
    s->w = (double *)malloc((size_t)np * sizeof(double));
    s->cumsum = (double *)malloc((size_t)(np + 1) * sizeof(double));
    s->mean = (double *)malloc((size_t)(np + 1) * sizeof(double));
    s->ootr = (double *)malloc((size_t)(np + 1) * sizeof(double));
    s->idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    s->tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    return s->phase && s->flux && s->dys && s->patched && s->patched_dy && s->w && s->cumsum &&
           s->mean && s->ootr && s->idx && s->tmp;
}

static void scratch_free(scratch *s)
{
    free(s->phase); free(s->flux); free(s->dys); free(s->patched); free(s->patched_dy);
    free(s->w); free(s->cumsum); free(s->mean); free(s->ootr); free(s->idx); free(s->tmp);
}

/* core.py:96-188 for ONE period.  uniq_width: ascending unique widths (numpy.unique,
 * core.py:113); first_row[k]: first table row with uniq_width[k] (core.py:163-165). */
static void oracle_search_one(double period, const double *t, const double *y, const double *dy,
                              int64_t n, double t_min, double t_max, const double *tmpl_values,
                              const int64_t *tmpl_offset, const int64_t *tmpl_length,
                              const double *tmpl_overshoot, const int64_t *uniq_width,
                              const int64_t *first_row, int64_t n_uniq, int64_t maxwidth,
                              double transit_depth_min, double R_star_min, double R_star_max,
                              double M_star_min, double M_star_max, double T0_fit_margin,
                              scratch *s, double *out_chi2, int64_t *out_row, double *out_depth,
                              oracle_counters *cnt)
{
    int64_t np = n + maxwidth, i, k;
    double edge_corr, duration_max, duration_min, length, naive, worst, correction;
    int64_t dmin_samples, dmax_samples;
    double summed_residual_in_rows = INFINITY, best_depth = 0.0;
    int64_t best_row = 0;

    /* fold (core.py:18,119) and stable sort (core.py:120-123) */
    for (i = 0; i < n; i++) {
        double x = t[i] / period;
        s->phase[i] = x - floor(x);
        s->idx[i] = i;
    }
    merge_sort_idx(s->phase, s->idx, s->tmp, 0, n);
    for (i = 0; i < n; i++) {
        s->flux[i] = y[s->idx[i]];
        s->dys[i] = dy[s->idx[i]];
    }
    /* patching (core.py:126-132) */
    for (i = 0; i < n; i++) {
        s->patched[i] = s->flux[i];
        s->patched_dy[i] = s->dys[i];
    }
    for (i = 0; i < maxwidth; i++) {
        s->patched[n + i] = s->flux[i];
        s->patched_dy[n + i] = s->dys[i];
    }
    for (i = 0; i < np; i++)
        s->w[i] = 1 / (s->patched_dy[i] * s->patched_dy[i]);

    edge_corr = edge_effect(s->flux, s->patched, s->dys, s->w, n, np);

    /* duration window (core.py:143-156) */
    duration_max = t14(R_star_max, M_star_max, period, 0);
    duration_min = t14(R_star_min, M_star_min, period, 1);
    length = t_max - t_min;
    naive = length / period;
    worst = naive + 1;
    correction = worst / naive;
    dmin_samples = (int64_t)floor(duration_min * (double)n);
    dmax_samples = (int64_t)ceil(duration_max * (double)n * correction);

    /* prefix sum shared by every running_mean call: numpy.cumsum(insert(data,0,0)) is
     * sequential and does not depend on the width (helpers.py:72) */
    s->cumsum[0] = 0.0;
    for (i = 0; i < np; i++)
        s->cumsum[i + 1] = s->cumsum[i] + s->patched[i];

    for (k = 0; k < n_uniq; k++) {
        int64_t duration = uniq_width[k], row = first_row[k], n_mean;
        double this_residual, this_depth;
        if (duration < dmin_samples || duration > dmax_samples)
            continue;
        n_mean = np - duration + 1;
        for (i = 0; i < n_mean; i++) /* mean = 1 - running_mean (core.py:167) */
            s->mean[i] = 1 - (s->cumsum[i + duration] - s->cumsum[i]) / (double)duration;
        ootr(s->patched, duration, s->w, np, s->ootr);
        this_residual = lowest_residuals(s->mean, n_mean, transit_depth_min, s->patched, duration,
                                         tmpl_values + tmpl_offset[row], tmpl_length[row], s->w,
                                         tmpl_overshoot[row], s->ootr, edge_corr, n,
                                         T0_fit_margin, &this_depth, cnt);
        if (this_residual < summed_residual_in_rows) { /* core.py:183-186 */
            summed_residual_in_rows = this_residual;
            best_row = row;
            best_depth = this_depth;
        }
    }
    *out_chi2 = summed_residual_in_rows;
    *out_row = best_row;
    *out_depth = best_depth;
}

/*
 * Search all periods.  Outputs are written at the index of the period in
 * `periods` (the reference returns them re-sorted by period, main.py:190-196; the
 * caller passes ascending or descending periods and indexes accordingly).
 * counters: int64[3] {grid_cells, evaluated_cells, inner_steps} or NULL.
 * n_threads <= 0: all OpenMP threads.  Returns 0, or -1 on allocation failure,
 * -2 on bad arguments.
 */
int tls_oracle_search(const double *t, const double *y, const double *dy, int64_t n,
                      const double *periods, int64_t n_periods, const double *tmpl_values,
                      const int64_t *tmpl_offset, const int64_t *tmpl_length,
                      const int64_t *tmpl_width, const double *tmpl_overshoot, int64_t n_rows,
                      double transit_depth_min, double R_star_min, double R_star_max,
                      double M_star_min, double M_star_max, double T0_fit_margin,
                      double *out_chi2, int64_t *out_row, double *out_depth, int64_t *counters,
                      int n_threads)
{
    int64_t *uniq_width, *first_row, n_uniq = 0, maxwidth = 0, r, k, p;
    double t_min, t_max;
    int status = 0;
    int64_t c_grid = 0, c_eval = 0, c_steps = 0;

    if (n < 1 || n_periods < 0 || n_rows < 1)
        return -2;

    /* numpy.unique(width_in_samples) ascending + first row per width */
    uniq_width = (int64_t *)malloc((size_t)n_rows * sizeof(int64_t));
    first_row = (int64_t *)malloc((size_t)n_rows * sizeof(int64_t));
    if (!uniq_width || !first_row)
        return -1;
    for (r = 0; r < n_rows; r++) {
        int seen = 0;
        for (k = 0; k < n_uniq; k++)
            if (uniq_width[k] == tmpl_width[r]) { seen = 1; break; }
        if (!seen) {
            uniq_width[n_uniq] = tmpl_width[r];
            first_row[n_uniq] = r;
            n_uniq++;
        }
    }
    for (k = 1; k < n_uniq; k++) { /* insertion sort by width */
        int64_t wv = uniq_width[k], rv = first_row[k], m = k - 1;
        while (m >= 0 && uniq_width[m] > wv) {
            uniq_width[m + 1] = uniq_width[m];
            first_row[m + 1] = first_row[m];
            m--;
        }
        uniq_width[m + 1] = wv;
        first_row[m + 1] = rv;
    }
    maxwidth = uniq_width[n_uniq - 1]; /* core.py:114-116 */
    if (maxwidth % 2 != 0)
        maxwidth += 1;

    t_min = t[0]; t_max = t[0];
    for (k = 1; k < n; k++) {
        if (t[k] < t_min) t_min = t[k];
        if (t[k] > t_max) t_max = t[k];
    }

#ifdef _OPENMP
    /* the thread count is a process-wide OpenMP setting: always set it, so that a 1-thread
     * call does not leak into the next "all threads" call */
    omp_set_num_threads(n_threads > 0 ? n_threads : omp_get_num_procs());
#else
    (void)n_threads;
#endif

#pragma omp parallel reduction(+ : c_grid, c_eval, c_steps)
    {
        scratch s;
        oracle_counters cnt = {0, 0, 0};
        int ok = scratch_alloc(&s, n, n + maxwidth);
        if (!ok) {
#pragma omp atomic write
            status = -1;
        }
#pragma omp for schedule(dynamic, 4)
        for (p = 0; p < n_periods; p++) {
            if (ok)
                oracle_search_one(periods[p], t, y, dy, n, t_min, t_max, tmpl_values, tmpl_offset,
                                  tmpl_length, tmpl_overshoot, uniq_width, first_row, n_uniq,
                                  maxwidth, transit_depth_min, R_star_min, R_star_max, M_star_min,
                                  M_star_max, T0_fit_margin, &s, &out_chi2[p], &out_row[p],
                                  &out_depth[p], &cnt);
        }
        scratch_free(&s);
        c_grid += cnt.grid_cells;
        c_eval += cnt.evaluated_cells;
        c_steps += cnt.inner_steps;
    }
    if (counters) {
        counters[0] = c_grid;
        counters[1] = c_eval;
        counters[2] = c_steps;
    }
    free(uniq_width);
    free(first_row);
    return status;
}

/* Exposed for unit tests of the pieces (tests/test_oracle_golden.py). */
double tls_oracle_t14(double R_s, double M_s, double P, int small) { return t14(R_s, M_s, P, small); }

void tls_oracle_fold_sort(const double *t, int64_t n, double period, double *phase_sorted,
                          int64_t *sort_index)
{
    int64_t i, *tmp = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    double *ph = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (i = 0; i < n; i++) {
        double x = t[i] / period;
        ph[i] = x - floor(x);
        sort_index[i] = i;
    }
    merge_sort_idx(ph, sort_index, tmp, 0, n);
    for (i = 0; i < n; i++)
        phase_sorted[i] = ph[sort_index[i]];
    free(tmp);
    free(ph);
}


/* ---- final T0 fit, stats.py:135-204 ------------------------------------------------------- */
/* numpy.linspace(start, stop, num): arange(num) * step + start with step = (stop-start)/(num-1),
 * last element set to stop. */
static void linspace(double start, double stop, int64_t num, double *out)
{
    int64_t k;
    double delta = stop - start, step;
    if (num <= 0) return;
    if (num == 1) { out[0] = start; return; }
    step = delta / (double)(num - 1);
    for (k = 0; k < num; k++)
        out[k] = (step == 0.0 ? ((double)k / (double)(num - 1)) * delta : (double)k * step) + start;
    out[num - 1] = stop;
}

/* Loop body of stats.py:178-195 for every trial epoch: fold, stable sort, roll twice, residuals.
 * `signal` is already scaled to the fitted depth (stats.py:143). */
static void t0_residuals(const double *t, const double *y, int64_t n, double period, const double *signal,
                         int64_t dur, const double *epochs, int64_t points, int64_t roll_cadences, double *res,
                         int n_threads)
{
    if (roll_cadences >= n) roll_cadences = 0;                 /* flux[-r:] is everything, flux[:-r] nothing */
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)n_threads;
#pragma omp parallel
    {
        double *ph = (double *)malloc((size_t)n * sizeof(double));
        double *flux = (double *)malloc((size_t)n * sizeof(double));
        double *rolled = (double *)malloc((size_t)n * sizeof(double));
        double *dy = (double *)malloc((size_t)n * sizeof(double));
        int64_t *idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        int64_t *tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        int64_t e, j;
#pragma omp for schedule(dynamic, 4)
        for (e = 0; e < points; e++) {
            double Tx = epochs[e], r_in = 0.0, r_out = 0.0;
            for (j = 0; j < n; j++) {                          /* core.py:9-12 fold */
                double x = (t[j] - Tx) / period;
                ph[j] = x - floor(x);
                idx[j] = j;
            }
            merge_sort_idx(ph, idx, tmp, 0, n);                /* stats.py:179 */
            for (j = 0; j < n; j++) flux[j] = y[idx[j]];       /* stats.py:181 */
            /* stats.py:190: flux = concatenate([flux[-r:], flux[:-r]]) */
            for (j = 0; j < n; j++) rolled[j] = flux[(j - roll_cadences + n) % n];
            /* stats.py:191: dy = concatenate([flux[-r:], flux[:-r]]) of the ROLLED flux (kept quirk) */
            for (j = 0; j < n; j++) dy[j] = rolled[(j - roll_cadences + n) % n];
            for (j = 0; j < dur; j++)                          /* stats.py:193 */
                r_in += ((rolled[j] - signal[j]) * (rolled[j] - signal[j])) / (dy[j] * dy[j]);
            for (j = dur; j < n; j++)                          /* stats.py:194 */
                r_out += ((rolled[j] - 1.0) * (rolled[j] - 1.0)) / (dy[j] * dy[j]);
            res[e] = r_in + r_out;
        }
        free(ph); free(flux); free(rolled); free(dy); free(idx); free(tmp);
    }
}

/* The loop alone, for a caller-supplied trial grid and depth-scaled signal (what the device kernel
 * tls_t0_fit computes). */
int tls_oracle_t0_residuals(const double *t, const double *y, int64_t n, double period, const double *signal,
                            int64_t dur, const double *epochs, int64_t n_epochs, int64_t roll, double *out,
                            int n_threads)
{
    if (n < 1 || dur < 1 || dur > n || n_epochs < 0 || roll < 0) return -1;
    t0_residuals(t, y, n, period, signal, dur, epochs, n_epochs, roll, out, n_threads);
    return 0;
}

/* signal_in: the chosen template row (depth SIGNAL_DEPTH), `dur` samples; depth: the fitted depth.
 * Returns the number of trial epochs; *T0_out the first epoch with the strictly smallest residual
 * (0 when none is finite).  epochs_out / residuals_out (each `n` entries, may be NULL) receive the
 * trial grid and the residual of every epoch. */
int64_t tls_oracle_final_t0_fit(const double *signal_in, int64_t dur, double depth, const double *t,
                                const double *y, int64_t n, double period, double T0_fit_margin,
                                double *T0_out, double *epochs_out, double *residuals_out, int n_threads)
{
    int64_t points, k, i;
    double *signal, *T0_array, *res, t_min;
    double scale = TLS_SIGNAL_DEPTH / (1 - depth);             /* stats.py:142 */
    double residuals_lowest = INFINITY, T0 = 0;
    if (n < 1 || dur < 1 || dur > n) return -1;
    signal = (double *)malloc((size_t)dur * sizeof(double));
    for (k = 0; k < dur; k++) signal[k] = 1 - ((1 - signal_in[k]) / scale);   /* stats.py:143 */
    if (T0_fit_margin == 0) points = n;                        /* stats.py:146-152 */
    else points = (int64_t)((double)n / (T0_fit_margin * (double)dur));
    if (points > n) points = n;
    if (points < 0) points = 0;
    T0_array = (double *)malloc((size_t)(points > 0 ? points : 1) * sizeof(double));
    res = (double *)malloc((size_t)(points > 0 ? points : 1) * sizeof(double));
    t_min = t[0];
    for (i = 1; i < n; i++) if (t[i] < t_min) t_min = t[i];
    linspace(t_min, t_min + period, points, T0_array);        /* stats.py:155-157 */
    t0_residuals(t, y, n, period, signal, dur, T0_array, points, dur / 2 + 1 /* stats.py:189 */, res, n_threads);
    for (k = 0; k < points; k++) {                             /* stats.py:199-201 */
        if (res[k] < residuals_lowest) { residuals_lowest = res[k]; T0 = T0_array[k]; }
        if (epochs_out) epochs_out[k] = T0_array[k];
        if (residuals_out) residuals_out[k] = res[k];
    }
    *T0_out = T0;
    free(signal); free(T0_array); free(res);
    return points;
}

/* ---- SDE spectra, stats.py:105-132 with helpers.py:93-108 ----------------------------------- */
static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* helpers.py:93-108 for an integer kernel <= n: numpy.median of every window, edges padded */
static void running_median(const double *data, int64_t n, int64_t kernel, double *out)
{
    int64_t n_med = n - kernel + 1, i, missing, front;
    double *win = (double *)malloc((size_t)kernel * sizeof(double));
    double *med = (double *)malloc((size_t)n_med * sizeof(double));
    for (i = 0; i < n_med; i++) {
        memcpy(win, data + i, (size_t)kernel * sizeof(double));
        qsort(win, (size_t)kernel, sizeof(double), cmp_double);
        med[i] = kernel % 2 ? win[kernel / 2] : 0.5 * (win[kernel / 2 - 1] + win[kernel / 2]);
    }
    missing = n - n_med;
    front = (int64_t)((double)missing * 0.5);
    for (i = 0; i < front; i++) out[i] = med[0];
    for (i = 0; i < n_med; i++) out[front + i] = med[i];
    for (i = front + n_med; i < n; i++) out[i] = med[n_med - 1];
    free(win); free(med);
}

static double mean_of(const double *x, int64_t n)
{
    double s = 0.0; int64_t i;
    for (i = 0; i < n; i++) s += x[i];
    return s / (double)n;
}

static double std_of(const double *x, int64_t n)   /* numpy.std: population, two-pass */
{
    double m = mean_of(x, n), s = 0.0; int64_t i;
    for (i = 0; i < n; i++) s += (x[i] - m) * (x[i] - m);
    return sqrt(s / (double)n);
}

/* chi2[n] -> SR, power_raw, power (n each); sde[0] = SDE_raw, sde[1] = SDE.  `kernel` is
 * oversampling_factor * SDE_MEDIAN_KERNEL_SIZE as an integer (stats.py:114). */
int tls_oracle_spectra(const double *chi2, int64_t n, int64_t kernel, double *SR, double *power_raw,
                       double *power, double *sde)
{
    int64_t i;
    double cmin, m, SDE_raw, mx, scale, SDE;
    if (n < 1) return -1;
    cmin = chi2[0];
    for (i = 1; i < n; i++) if (chi2[i] < cmin) cmin = chi2[i];
    for (i = 0; i < n; i++) SR[i] = cmin / chi2[i];                        /* :106 */
    m = mean_of(SR, n);
    SDE_raw = (1 - m) / std_of(SR, n);                                     /* :107 */
    mx = -INFINITY;
    for (i = 0; i < n; i++) { power_raw[i] = SR[i] - m; if (power_raw[i] > mx) mx = power_raw[i]; }
    scale = SDE_raw / mx;                                                  /* :111 */
    for (i = 0; i < n; i++) power_raw[i] = power_raw[i] * scale;
    if (kernel % 2 == 0) kernel = kernel + 1;                              /* :115-117 */
    if (n > 2 * kernel) {
        double *med = (double *)malloc((size_t)n * sizeof(double));
        running_median(power_raw, n, kernel, med);
        for (i = 0; i < n; i++) power[i] = power_raw[i] - med[i];          /* :120 */
        m = mean_of(power, n);
        for (i = 0; i < n; i++) power[i] = power[i] - m;                   /* :123 */
        {
            double sd = std_of(power, n);
            SDE = -INFINITY; mx = -INFINITY;
            for (i = 0; i < n; i++) {
                if (power[i] / sd > SDE) SDE = power[i] / sd;              /* :124 */
                if (power[i] > mx) mx = power[i];
            }
        }
        scale = SDE / mx;                                                  /* :126 */
        for (i = 0; i < n; i++) power[i] = power[i] * scale;
        free(med);
    } else {
        for (i = 0; i < n; i++) power[i] = power_raw[i];
        SDE = SDE_raw;
    }
    sde[0] = SDE_raw; sde[1] = SDE;
    return 0;
}
