// tls_amd.hip -- host side of libtls_amd.so: the C ABI of include/tls_amd.h.
//
// Builds the device work list for one light curve (distinct trial widths, per-period
// duration windows, cost-ordered period queue), keeps every buffer resident in HBM between
// calls, launches the search kernel of tls_kernels.hip.h and gathers results; optional RCCL
// all-gather for the period-sharded multi-GPU mode.
//
// Reference mapping: main.py:140-196 (dispatch + ordered gather), core.py:113-116,143-156
// (width list, duration window per period), grid.py:9-32 (T14).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tls_amd.h"
#include "tls_kernels.hip.h"

namespace {

// physical constants of the duration window (reference tls_constants.py:20-25,78)
constexpr double kG = 6.673e-11;
constexpr double kRsun = 695508000.0;
constexpr double kRjup = 69911000.0;
constexpr double kMsun = 1.989 * 1e30;
constexpr double kSecondsPerDay = 86400.0;
constexpr double kFracDurationMax = 0.12;
constexpr double kPi = 3.141592653589793;

constexpr size_t kLdsPerCU = 160 * 1024;
// The four-slot kernel is taken when at least this many periods fit a CU's LDS.  Three (Tutorial 01: 100 d, 43.8 KB a period)
// already beat the classic kernel's ONE 1024-thread workgroup per CU by 18 % (1.47 against 1.79 ms, same box); in the narrow
// band where the classic kernel still fits two workgroups and this one only three, the classic one is 4 % faster (a 42-day
// probe: 0.371 against 0.386 ms) -- a series length of one day in a hundred, not special-cased.
#ifndef TLS_SLIM_MIN_SLOTS
#define TLS_SLIM_MIN_SLOTS 3
#endif
constexpr size_t kSlimMinSlots = TLS_SLIM_MIN_SLOTS;
constexpr size_t kEventRing = 64;   // launch-timing event pairs kept per context

std::string g_create_error;  // tls_last_error(NULL)

// grid.py:9-32 with the reference's operation order (libm pow, as CPython does)
double t14(double R_s, double M_s, double P, bool small) {
    P = P * kSecondsPerDay;
    R_s = kRsun * R_s;
    M_s = kMsun * M_s;
    const double chord = std::pow((4 * P) / (kPi * kG * M_s), 1.0 / 3);
    const double T14max = small ? R_s * chord : (R_s + 2 * kRjup) * chord;
    double result = T14max / P;
    if (result > kFracDurationMax) result = kFracDurationMax;
    return result;
}

template <typename T>
struct DevBuf {
    T* ptr = nullptr;
    size_t cap = 0;  // elements
    hipError_t reserve(size_t n_elem) {
        if (n_elem <= cap) return hipSuccess;
        if (ptr) { hipError_t e = hipFree(ptr); ptr = nullptr; cap = 0; if (e != hipSuccess) return e; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(n_elem, 1) * sizeof(T));
        if (e == hipSuccess) cap = n_elem;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// a typed window into the context's one plan allocation (d_plan): same `.ptr` as a DevBuf, not owned
template <typename T>
struct View {
    T* ptr = nullptr;
};

// The switches of a context.  Two are public (tls_options: exact_prefix, slim); the others are developer / test switches
// that select kernel variants and launch shapes for A/B runs, reached by name through tls_debug_set_switch (never part
// of the stable ABI).  -1 (band_max, prune_min_live: negative) = the library decides.
struct Switches {
    int32_t exact_prefix, slim;
    int32_t prune, screen32, no_screen, fast_slab, x_staged, split, split_batch, sort2, threads, blocks, plan_threads, t0_rot;
    int64_t prune_min_live;
    double band_max;
};
struct SwitchName { const char* name; const char* env; size_t offset; int kind; };   // kind 0: int32, 1: int64, 2: double
#define TLS_SW(field, env, kind) { #field, env, offsetof(Switches, field), kind }
const SwitchName kSwitchNames[] = {
    TLS_SW(exact_prefix, "TLS_EXACT_PREFIX", 0), TLS_SW(slim, "TLS_SLIM", 0), TLS_SW(prune, "TLS_PRUNE", 0),
    TLS_SW(screen32, "TLS_SCREEN32", 0), TLS_SW(no_screen, "TLS_NO_SCREEN", 0), TLS_SW(fast_slab, "TLS_FAST_SLAB", 0),
    TLS_SW(x_staged, "TLS_X_STAGED", 0), TLS_SW(split, "TLS_SPLIT", 0), TLS_SW(split_batch, "TLS_SPLIT_BATCH", 0),
    TLS_SW(sort2, "TLS_SORT2", 0), TLS_SW(threads, "TLS_THREADS", 0), TLS_SW(blocks, "TLS_BLOCKS", 0),
    TLS_SW(plan_threads, "TLS_PLAN_THREADS", 0), TLS_SW(t0_rot, "TLS_T0_ROT", 0), TLS_SW(prune_min_live, "TLS_PRUNE_MIN_LIVE", 1),
    TLS_SW(band_max, "TLS_BAND_MAX", 2),
};
#undef TLS_SW
const SwitchName* find_switch(const char* name) {
    for (const auto& sw : kSwitchNames) if (name && std::strcmp(sw.name, name) == 0) return &sw;
    return nullptr;
}
void switch_store(Switches& o, const SwitchName& sw, double value) {
    unsigned char* at = reinterpret_cast<unsigned char*>(&o) + sw.offset;
    if (sw.kind == 0) { const int32_t v = value < 0 ? -1 : (int32_t)value; std::memcpy(at, &v, sizeof v); }
    else if (sw.kind == 1) { const int64_t v = value < 0 ? -1 : (int64_t)value; std::memcpy(at, &v, sizeof v); }
    else { const double v = value < 0 ? -1.0 : value; std::memcpy(at, &v, sizeof v); }
}
double switch_load(const Switches& o, const SwitchName& sw) {
    const unsigned char* at = reinterpret_cast<const unsigned char*>(&o) + sw.offset;
    if (sw.kind == 0) { int32_t v; std::memcpy(&v, at, sizeof v); return (double)v; }
    if (sw.kind == 1) { int64_t v; std::memcpy(&v, at, sizeof v); return (double)v; }
    double v; std::memcpy(&v, at, sizeof v); return v;
}
// every switch "the library decides" (all bytes defined: plans are keyed by memcmp over the struct)
Switches default_switches() {
    Switches o;
    std::memset(&o, 0, sizeof o);
    for (const auto& sw : kSwitchNames) switch_store(o, sw, -1.0);
    return o;
}
// "name=value,name=value" (what tls_debug_get_switches writes) over a set of switches; false on an unknown name
bool switches_parse(Switches& o, const char* spec) {
    if (!spec) return true;
    std::string text(spec);
    size_t at = 0;
    while (at < text.size()) {
        size_t end = text.find(',', at);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(at, end - at);
        at = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) return false;
        const SwitchName* sw = find_switch(item.substr(0, eq).c_str());
        if (!sw) return false;
        switch_store(o, *sw, std::atof(item.c_str() + eq + 1));
    }
    return true;
}

// The TLS_* environment variables (one per switch, kSwitchNames), read ONCE per process: what a new context starts with and
// what the context-free planning call (tls_period_costs) uses when it is given no switches.  Nothing reads the environment
// after this.
const Switches& process_options() {
    static const Switches cached = [] {
        Switches o = default_switches();
        for (const auto& sw : kSwitchNames)
            if (const char* v = std::getenv(sw.env)) switch_store(o, sw, (*v == 0 && sw.kind == 0) ? 1.0 : std::atof(v));
        return o;
    }();
    return cached;
}

// what a prepared plan was built from: a second tls_prepare with the same time stamps, period list, template table,
// parameters and developer switches only replaces the flux (the search call of a survey, or of repeated power()
// calls).  Compared byte for byte (memcmp runs at ~10 GB/s; a cfg2 key is 120 KB).
struct PlanLayout {   // byte offsets of the plan arrays inside d_plan / h_stage (256-byte aligned)
    size_t t = 0, y = 0, w = 0, periods = 0, order = 0, rows = 0, widths = 0, screens = 0, q = 0, q2 = 0, g = 0, tile_prefix = 0, total = 0;
};

struct PlanKey {
    bool valid = false;
    int64_t n = 0, n_periods = 0, n_rows = 0;
    std::vector<double> t, periods, values, overshoot;
    std::vector<int64_t> offset, length, width;
    tls_params params = {0, 0, 0, 0, 0, 0};
    Switches opt;
};

}  // namespace

struct tls_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    std::string name;
    int n_cu = 0;

    // device-resident plan
    // ONE allocation holds every plan array (views below); it is filled from ONE pinned staging buffer by ONE
    // asynchronous copy, and tls_prepare does not wait for it
    DevBuf<unsigned char> d_plan;
    View<double> d_t, d_y, d_w, d_periods, d_q, d_q2, d_g;
    View<int> d_order;
    View<tlsdev::PeriodRows> d_rows;
    View<tlsdev::WidthEntry> d_widths;
    View<tlsdev::RowScreen> d_screens;
    unsigned char* h_stage = nullptr; size_t h_stage_cap = 0;   // pinned: the plan (tls_prepare) / the flux (tls_update_flux)
    hipEvent_t ev_stage = nullptr; bool stage_pending = false;  // its last upload
    // results: [chi2 | row | depth | counters[4]] in one allocation, fetched by one copy into pinned memory
    DevBuf<double> d_out;
    View<double> d_chi2, d_depth;
    View<long long> d_row;
    View<unsigned long long> d_counters;
    double* h_out = nullptr; size_t h_out_cap = 0;
    PlanKey key;
    PlanLayout layout;
    int64_t plan_reuses = 0;   // tls_prepare calls answered from the held plan
    std::vector<double> batch_group_ms;   // wall time of every group of 32 light curves of the last tls_power_batch / tls_search_batch (tls_debug_batch_group_ms)
    std::vector<double> batch_group_wait_ms;   // ... of which the group's one wait for the device (tls_power_batch)
    DevBuf<double> d_scratch, d_pack, d_gather, d_scalar, d_stage;
    DevBuf<unsigned long long> d_phase, d_check;
    DevBuf<unsigned int> d_queue, d_squeue, d_lists, d_perm, d_pqueues;   // d_pqueues: tls_power_batch's T0-fit queues   // d_squeue: the search kernel's self-rewinding queue
    DevBuf<double> d_curve_S0, d_curve_w0;   // survey batches
    // survey batches: two slots of device + pinned host buffers, a second stream for the transfers
    struct BatchSlot {
        DevBuf<double> d_y, d_w, d_S0, d_w0, d_chi2, d_depth;
        DevBuf<long long> d_row;
        double* h_in = nullptr; size_t h_in_cap = 0;     // pinned: y | w | S0 | w0 of one group
        double* h_out = nullptr; size_t h_out_cap = 0;   // pinned: chi2 | row | depth of one group
        hipEvent_t ev_in = nullptr, ev_kernel = nullptr, ev_out = nullptr;
    } slot[2];
    hipStream_t copy_stream = nullptr;
    // enqueue() reads these when set (a batch slot); otherwise the context's own buffers
    const double* over_y = nullptr; const double* over_w = nullptr;
    const double* over_S0 = nullptr; const double* over_w0 = nullptr;
    double* over_chi2 = nullptr; long long* over_row = nullptr; double* over_depth = nullptr;
    bool sort2 = false;                      // tiled variant: two-level sort
    int batch_curves = 1;                    // light curves the next launch searches (tls_search_batch)
    DevBuf<double> d_ft, d_fy, d_fsig, d_fep, d_fres, d_fscratch;  // final T0 fit
    DevBuf<double> d_pink;          // tls_pink_noise: data | terms | running sums
    DevBuf<double> d_frot;          // ... its rotation path: per fit flux | phases | quotients of the base order, state
    DevBuf<int> d_frperm;           // ... and the base order itself
    DevBuf<double> d_spec;                                         // SDE spectra: SR | power_raw | power | sde[2] | chi2 copy
    size_t list_stride = 0;
    // two-kernel slab path (series in HBM, one light curve): fold kernel + search kernel per batch of periods
    bool split = false;                      // the plan supports it (enqueue uses it for single-curve launches)
    int split_blocks = 0;                    // workgroups of its launches (not capped by the number of periods: tiles are items too)
    int split_batch = 0;                     // periods per batch: as many slabs are held in HBM
    int64_t split_max_items = 0;             // most (period, tile, row part) items of any batch
    bool split_fast = false;                 // the two roles may run fast prefix-sum mode (uniform weights, X at staging, dot products on X)
    View<unsigned int> d_tile_prefix;        // [n_periods + 1] tiles in front of work item w (queue order)
    std::vector<unsigned int> host_tile_prefix;
    DevBuf<double> d_partials;               // [split_max_items][3] a tile's winner
    DevBuf<unsigned int> d_tiles_done;       // [split_batch] tiles of the period that are done (zero between launches)
    int hdr_bytes = 0, tile_len = 0, tile_halo = 0, region_pad = 0, p2_shift = 4;
    bool prune_kernel = false;        // launch the pruning variant (pruning_pays)
    std::vector<tlsdev::WidthEntry> host_widths;  // kept for tls_update_flux's pruning decision
    long long prune_min_live = 256;   // live units per period (tile) from which pruning pays; switch prune_min_live overrides
    Switches opt;                     // the context's switches (tls_set_options / tls_debug_set_switch; initially the process's TLS_* environment)

    // host-side plan
    bool prepared = false, executed = false;
    bool uniform_w = true, resident = true;
    int n = 0, W = 0, M = 0, n_periods = 0, n_widths = 0, nb = 0;
    int threads = 512, blocks = 0;
    const char* last_kernel = "";    // tls_last_kernel
    int slim_blocks = 0;              // > 0: the plan fits the four-slots-per-CU kernel (tls_slim_kernel): its workgroups in flight
    size_t slim_lds = 0;              // ... and its dynamic LDS
    int slim_threads = 256;           // ... and its workgroup size: 256 (four or three to a CU) or 512 (two to a CU: series of 5-10 k points)
    int cumsum_round = 2 * tlsdev::kCumsumChunk;
    size_t lds_bytes = 0;
    double S0 = 0, w0 = 1, depth_min = 0;
    double y_abs_max = 1.0;   // largest |flux| of the light curve(s) of the next launch: bounds the prefix sum (fast mode's eps)
    bool screen_kernel = false;   // the next launch takes the fp32-screen variant (screen_pays)
    double e_abs_max = INFINITY;   // largest |1 - flux| of the same (inf: a sample outside [0.5, 2]): admits the fp32 screen
    DevBuf<float> d_split;    // fp32 screen: low halves of the folded samples, one region per workgroup
    DevBuf<double> d_park;    // fp32 screen: parked cells, kParkCap per workgroup
    DevBuf<double> d_band;    // slab variant, fast mode: band_prefix of the launches (two slots, like h_band)
    double* d_band_now = nullptr;   // the slot the next launch reads
    double* h_band = nullptr; size_t h_band_cap = 0;   // pinned: two slots of (n_widths + 1) doubles
    hipEvent_t ev_band[2] = {nullptr, nullptr}; bool band_used[2] = {false, false}; int band_slot = 0;
    double flux_sigma = 0.0;  // scatter of the flux of the next launch (mean over the curves of a batch)
    double band_sigma = -1.0, band_eps = -1.0;   // what d_band was computed for
    long long q_count = 0;    // elements of the padded template rows (the fp32 screen's second copy starts there)
    tls_counters plan_counters = {0, 0, 0, 0, 0};
    bool counted = false;

    // per-launch kernel timing (HIP events on the context's stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;

    // RCCL
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0;
    int64_t gathered_count = 0;
};

namespace {

int update_flux_impl(tls_ctx* ctx, const double* y, const double* dy);
constexpr int kWeightsDiffer = 1;

int fail(tls_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define TLS_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(ctx, TLS_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define TLS_NCCL(ctx, call)                                                             \
    do {                                                                                \
        ncclResult_t r_ = (call);                                                       \
        if (r_ != ncclSuccess)                                                          \
            return fail(ctx, TLS_E_RCCL, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)

// the pinned staging buffer, at least `bytes` long and no longer read by an upload in flight
int stage_reserve(tls_ctx* ctx, size_t bytes) {
    if (!ctx->ev_stage) TLS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stage, hipEventDisableTiming));
    if (ctx->stage_pending) { TLS_HIP(ctx, hipEventSynchronize(ctx->ev_stage)); ctx->stage_pending = false; }
    if (ctx->h_stage_cap < bytes) {
        if (ctx->h_stage) TLS_HIP(ctx, hipHostFree(ctx->h_stage));
        ctx->h_stage = nullptr; ctx->h_stage_cap = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), cap, hipHostMallocDefault));
        ctx->h_stage_cap = cap;
    }
    return TLS_OK;
}

size_t template_values(const tls_template* tmpl) {
    int64_t total = 0;
    for (int64_t r = 0; r < tmpl->n_rows; ++r) total = std::max(total, tmpl->offset[r] + std::max<int64_t>(tmpl->length[r], 0));
    return (size_t)std::max<int64_t>(total, 0);
}

bool key_matches(const PlanKey& k, const Switches& opt, const double* t, int64_t n, const double* periods, int64_t n_periods,
                 const tls_template* tmpl, const tls_params* params) {
    if (!k.valid || k.n != n || k.n_periods != n_periods || k.n_rows != tmpl->n_rows) return false;
    if (std::memcmp(&k.params, params, sizeof(tls_params)) != 0) return false;
    const size_t rows = (size_t)tmpl->n_rows;
    if (std::memcmp(k.offset.data(), tmpl->offset, rows * 8) || std::memcmp(k.length.data(), tmpl->length, rows * 8) ||
        std::memcmp(k.width.data(), tmpl->width, rows * 8) || std::memcmp(k.overshoot.data(), tmpl->overshoot, rows * 8))
        return false;
    if (k.values.size() != template_values(tmpl) || std::memcmp(k.values.data(), tmpl->values, k.values.size() * 8)) return false;
    if (std::memcmp(k.t.data(), t, (size_t)n * 8) || std::memcmp(k.periods.data(), periods, (size_t)n_periods * 8)) return false;
    return std::memcmp(&k.opt, &opt, sizeof(Switches)) == 0;
}

void key_store(PlanKey& k, const Switches& opt, const double* t, int64_t n, const double* periods, int64_t n_periods,
               const tls_template* tmpl, const tls_params* params) {
    k.n = n; k.n_periods = n_periods; k.n_rows = tmpl->n_rows; k.params = *params;
    k.t.assign(t, t + n); k.periods.assign(periods, periods + n_periods);
    const size_t rows = (size_t)tmpl->n_rows;
    k.offset.assign(tmpl->offset, tmpl->offset + rows); k.length.assign(tmpl->length, tmpl->length + rows);
    k.width.assign(tmpl->width, tmpl->width + rows); k.overshoot.assign(tmpl->overshoot, tmpl->overshoot + rows);
    k.values.assign(tmpl->values, tmpl->values + template_values(tmpl));
    k.opt = opt;
    k.valid = true;
}

template <typename T>
int upload(tls_ctx* ctx, DevBuf<T>& buf, const T* host, size_t count) {
    TLS_HIP(ctx, buf.reserve(count));
    if (count)
        TLS_HIP(ctx, hipMemcpyAsync(buf.ptr, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return TLS_OK;
}

// weights and the period-independent constant S0 = sum (1-y)^2 / dy^2
void weights_from(const double* y, const double* dy, int64_t n, bool& uniform, double& w0,
                  std::vector<double>& w, double& S0, double* y_abs_max = nullptr, double* e_abs_max = nullptr) {
    if (y_abs_max) {
        double m = 0.0;
        for (int64_t i = 0; i < n; ++i) m = std::max(m, std::fabs(y[i]));
        *y_abs_max = std::max(*y_abs_max, m);
    }
    if (e_abs_max) {
        // 1 - y is exact for y in [0.5, 2] (Sterbenz) and a multiple of 2^-53 there: what the fp32 screen's split needs
        double m = 0.0;
        for (int64_t i = 0; i < n; ++i) m = (y[i] >= 0.5 && y[i] <= 2.0) ? std::max(m, std::fabs(1 - y[i])) : INFINITY;
        *e_abs_max = std::max(*e_abs_max, m);
    }
    uniform = true;
    for (int64_t i = 1; i < n; ++i)
        if (dy[i] != dy[0]) { uniform = false; break; }
    long double acc = 0.0L;
    if (uniform) {
        w0 = 1 / (dy[0] * dy[0]);  // core.py:127
        w.clear();
        for (int64_t i = 0; i < n; ++i) acc += (long double)((1 - y[i]) * (1 - y[i])) * w0;
    } else {
        w0 = 1.0;
        w.resize((size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            w[(size_t)i] = 1 / (dy[i] * dy[i]);
            acc += (long double)((1 - y[i]) * (1 - y[i])) * w[(size_t)i];
        }
    }
    S0 = (double)acc;
}

// Distinct trial widths ascending with the first template row of each (core.py:113,163-165),
// their T0 stride (core.py:50-55) and, optionally, the q_j = 1 - signal_j rows.
int build_widths(tls_ctx* ctx, const tls_template* tmpl, const tls_params* params, int64_t n,
                 std::vector<tlsdev::WidthEntry>& widths, std::vector<double>* q) {
    if (!tmpl || !params) return fail(ctx, TLS_E_ARG, "null argument");
    if (tmpl->n_rows < 1 || !tmpl->values || !tmpl->offset || !tmpl->length || !tmpl->width || !tmpl->overshoot)
        return fail(ctx, TLS_E_ARG, "empty template table");
    std::vector<int64_t> rows((size_t)tmpl->n_rows);
    std::iota(rows.begin(), rows.end(), 0);
    std::stable_sort(rows.begin(), rows.end(), [&](int64_t a, int64_t b) { return tmpl->width[a] < tmpl->width[b]; });
    const double margin = params->T0_fit_margin;
    size_t q_count = 0;
    for (int64_t r : rows) {
        const int64_t wd = tmpl->width[r];
        if (!widths.empty() && widths.back().width == wd) continue;  // later duplicates never used
        if (wd < 1 || wd > n) return fail(ctx, TLS_E_ARG, "template width out of range [1, n]");
        const int64_t len = tmpl->length[r];
        if (len < 1 || len > wd) return fail(ctx, TLS_E_ARG, "template row longer than its width");
        tlsdev::WidthEntry we;
        we.width = (int)wd; we.row = (int)r; we.q_len = (int)len;
        we.n_pos = 0; we.n_chunks = 0; we.list_base = 0; we.inv_d = 1.0 / (double)wd;
        we.xth = 1;
        if (margin > 0 && (double)wd > margin) {  // core.py:50-55
            const double inv = 1 / margin;
            int xth = (int)((double)wd / inv);
            we.xth = xth < 1 ? 1 : xth;
        }
        we.tiled = tlsdev::row_is_tiled(we.width, we.xth) ? 1 : 0;
        we.oversize = 0; we.pad_ = 0;
        // rows are stored zero padded (pad_front before, pad_back after, then up to a multiple
        // of 8 doubles) so that the unrolled dot product needs no edge handling; the pads grow
        // with the stride of a tiled row (its kR windows reach (kR-1)*xth samples further)
        const size_t front = (size_t)tlsdev::pad_front(we.tiled ? we.xth : 1);
        const size_t back = (size_t)tlsdev::pad_back(we.tiled ? we.xth : 1);
        we.q_offset = (int)(q_count + front);
        we.overshoot = tmpl->overshoot[r];
        double s2 = 0.0, s1 = 0.0, s_abs = 0.0;
        if (q) q->insert(q->end(), front, 0.0);
        for (int64_t j = 0; j < len; ++j) {
            const double qj = 1 - tmpl->values[tmpl->offset[r] + j];  // core.py:68
            if (q) q->push_back(qj);
            s2 += qj * qj;
            s1 += qj;
            s_abs += std::fabs(qj);
        }
        // constants of the pruning bound (cell_bound in tls_kernels.hip.h), rounded outwards
        double vq = 0.0;
        for (int64_t j = 0; j < len; ++j) {
            const double dq = (1 - tmpl->values[tmpl->offset[r] + j]) - s1 / (double)len;
            vq += dq * dq;
        }
        we.var_q = vq * (1 + 1e-12);
        we.k_mono = s1 - we.overshoot * s2;
        we.prunable = (len == wd && we.k_mono >= 0 && we.overshoot > 0 && params->transit_depth_min >= 0) ? 1 : 0;
        we.k_mono *= (1 + 1e-12);
        we.c_proxy = 4 * we.overshoot * we.k_mono;
        {   // fp32 screen (tlsdev::screen_cells): |B32 - B| <= screen_c * max|e|, doubled for the statistic's 2 rs B
            const int S = we.tiled ? (tlsdev::kR - 1) * we.xth : 0;
            we.screen_c = 2.0 * (1 + 1e-6) * ((double)(len + S) / 2 + 8) * 5.9604644775390625e-08 * 1.001 * s_abs;
        }
        size_t row_total = front + (size_t)len + 1 + back;   // (+1: the row of difference taps, one tap longer, shares the layout)
        row_total = (row_total + 7) / 8 * 8;
        if (q) q->resize(q_count + row_total, 0.0);
        q_count += row_total;
        we.sum_q2 = s2;
        widths.push_back(we);
    }
    return TLS_OK;
}

// Piecewise-constant images of the template rows for the pruning bound (tlsdev::window_bound).  All rows are the
// same transit shape resampled to their width (transit.py:98-160), so the segment boundaries are chosen ONCE, as
// fractions of the row length, on the widest row: the kSeg-segment partition with the smallest sum of squared
// deviations from the segment means (dynamic programme over <= 96 groups of taps).  Levels, telescoped
// differences and the squared remainder are then exact sums over each row's own taps (long double, remainder
// rounded up): the bound is rigorous for ANY boundaries, good ones only make it tight.
void build_screens(const std::vector<tlsdev::WidthEntry>& widths, const std::vector<double>& q,
                   std::vector<tlsdev::RowScreen>& screens, bool one_segment_only) {
    constexpr int K = tlsdev::kSeg;
    screens.assign(widths.size(), tlsdev::RowScreen());
    for (auto& sc : screens) { std::memset(&sc, 0, sizeof sc); }
    if (widths.empty()) return;
    double frac[K + 1];
    {
        const auto& we = widths.back();
        const int L = we.q_len, G = std::min(L, 96);
        const double* qr = q.data() + we.q_offset;
        std::vector<long double> s1((size_t)G + 1, 0.0L), s2((size_t)G + 1, 0.0L);
        std::vector<int> edge((size_t)G + 1);
        for (int g = 0; g <= G; ++g) edge[(size_t)g] = (int)((long long)g * L / G);
        for (int g = 0; g < G; ++g) {
            long double a1 = 0, a2 = 0;
            for (int j = edge[(size_t)g]; j < edge[(size_t)g + 1]; ++j) { a1 += qr[j]; a2 += (long double)qr[j] * qr[j]; }
            s1[(size_t)g + 1] = s1[(size_t)g] + a1; s2[(size_t)g + 1] = s2[(size_t)g] + a2;
        }
        auto sse = [&](int a, int b) -> double {
            const long double m = (long double)(edge[(size_t)b] - edge[(size_t)a]);
            const long double d1 = s1[(size_t)b] - s1[(size_t)a];
            return (double)((s2[(size_t)b] - s2[(size_t)a]) - d1 * d1 / m);
        };
        const int Ke = std::min(K, G);
        std::vector<double> cost((size_t)(Ke + 1) * (G + 1), 1e300);
        std::vector<int> arg((size_t)(Ke + 1) * (G + 1), 0);
        cost[0] = 0.0;
        for (int k = 1; k <= Ke; ++k)
            for (int j = k; j <= G; ++j) {
                double best = 1e300; int bi = k - 1;
                for (int i = k - 1; i < j; ++i) {
                    const double c = cost[(size_t)(k - 1) * (G + 1) + i] + sse(i, j);
                    if (c < best) { best = c; bi = i; }
                }
                cost[(size_t)k * (G + 1) + j] = best; arg[(size_t)k * (G + 1) + j] = bi;
            }
        int at = G;
        for (int k = K; k >= 0; --k) frac[k] = 1.0;
        for (int k = Ke; k >= 1; --k) { frac[k] = (double)edge[(size_t)at] / L; at = arg[(size_t)k * (G + 1) + at]; }
        frac[0] = 0.0;
        for (int k = Ke + 1; k <= K; ++k) frac[k] = 1.0;
    }
    for (size_t w = 0; w < widths.size(); ++w) {
        const auto& we = widths[w];
        tlsdev::RowScreen& sc = screens[w];
        const int L = we.q_len;
        if (!we.prunable || L < tlsdev::kScreenMinLen || L < 2 * K) continue;
        const double* qr = q.data() + we.q_offset;
        int b[K + 1];
        b[0] = 0;
        for (int k = 1; k < K; ++k) b[k] = std::max(b[k - 1] + 1, (int)std::lround(frac[k] * L));
        b[K] = L;
        for (int k = K - 1; k >= 1; --k) b[k] = std::min(b[k], b[k + 1] - 1);
        long double lev[K], r2 = 0.0L, sq = 0.0L;
        for (int k = 0; k < K; ++k) {
            long double a1 = 0.0L;
            for (int j = b[k]; j < b[k + 1]; ++j) a1 += qr[j];
            lev[k] = a1 / (long double)(b[k + 1] - b[k]);
        }
        // the device works with the levels rounded to double: remainder and sum are those of the ROUNDED levels
        double levd[K];
        for (int k = 0; k < K; ++k) levd[k] = (double)lev[k];
        long double dsum = 0.0L;   // sum of the differences q - q~: zero for exact means, ~1e-17 L after rounding
        for (int k = 0; k < K; ++k)
            for (int j = b[k]; j < b[k + 1]; ++j) {
                const long double dq = (long double)qr[j] - (long double)levd[k];
                r2 += dq * dq; dsum += dq; sq += (long double)levd[k];
            }
        for (int k = 0; k <= K; ++k) sc.b[k] = b[k];
        sc.g[0] = -levd[0];
        for (int k = 1; k < K; ++k) sc.g[k] = levd[k - 1] - levd[k];
        sc.g[K] = levd[K - 1];
        // (g_k is the rounded difference of two doubles: the telescoped sum then equals sum_j q~'_j e_j for levels
        // q~' within 1 ulp of levd -- absorbed by inflating the remainder; |dsum| * mean enters the same way)
        sc.sq = (double)sq;
        sc.r2 = (double)(r2 * (1.0L + 1e-9L)) + 1e-24 + 1e-12 * (double)fabsl(dsum);
        sc.valid = one_segment_only ? 0 : 1;   // developer switch (switch no_screen): the one-segment bound (cell_bound) for every row
    }
}

// Expected fraction of trial cells that pass the depth predicate (core.py:58) on a flat, white light curve, averaged
// over the trial widths: large when the noise of a window mean, sigma/sqrt(d), is large against transit_depth_min.  It
// says how much of a period is dot products -- what the pruning variant and the fp32 screen save (pick below).
double passing_fraction(const std::vector<tlsdev::WidthEntry>& widths, double sigma, double depth_min) {
    if (!(sigma > 0) || widths.empty()) return 0.0;
    double acc = 0.0;
    for (const auto& we : widths) acc += 0.5 * std::erfc(depth_min * std::sqrt((double)we.width) / sigma / std::sqrt(2.0));
    return acc / (double)widths.size();
}
// fp32 screen of the dot products (tlsdev::screen_cells) admissible: LDS-resident series, uniform weights, every sample
// e = 1 - flux the exact sum of two fp32 halves (flux in [0.5, 2] and |e| < 2^-5)
bool screen_admissible(bool resident, bool uniform, double e_abs_max) {
    return resident && uniform && e_abs_max < 0.03125;
}
// Which variant of the LDS-resident search kernel a launch takes, from the expected passing fraction f of the depth
// predicate.  Round 4, 90-day configuration, same box, ms (plain / fp32 screen / pruning): 50 ppm (f = 0.09) 1.19 / 1.23 /
// 1.61; 75 ppm (0.16) 1.67 / 1.62 / 1.87; 100 ppm (0.20) 2.10 / 1.94 / 2.14; 150 ppm (0.28) 2.62 / 2.34 / 2.37; 200 ppm
// (0.32) 2.94 / 2.54 / 2.47; 300 ppm (0.38) 3.22 / 2.76 / 2.58; 500 ppm (0.42) 3.61 / 2.93 / 2.75.  The
// screen halves the FMA instructions of the dot products but adds a split pass and a valuation pass per period (DESIGN
// section 4): it pays where the dot products dominate and the pruning passes do not pay yet.
// switch prune = 0/1 and ::screen32 = 0/1 force either choice (tests run all three variants).
constexpr double kScreenFromFraction = 0.13, kPruneFromFraction = 0.24, kPruneFromFractionBesideScreen = 0.30;
bool pruning_pays(const Switches& opt, const std::vector<tlsdev::WidthEntry>& widths, double sigma, double depth_min, bool resident,
                  bool screen_ok = false) {
    if (!resident) return false;   // (the bound's look-ups in X want the series in LDS: no slab instantiation)
    if (opt.prune >= 0) return opt.prune != 0;
    if (!(sigma > 0) || widths.empty()) return false;
    for (const auto& we : widths) if (!we.prunable) return false;
    if (opt.screen32 >= 0) screen_ok = screen_ok && opt.screen32 != 0;
    return passing_fraction(widths, sigma, depth_min) >= (screen_ok ? kPruneFromFractionBesideScreen : kPruneFromFraction);
}
bool screen_pays(const Switches& opt, const std::vector<tlsdev::WidthEntry>& widths, double sigma, double depth_min, bool admissible) {
    if (!admissible) return false;
    if (opt.screen32 >= 0) return opt.screen32 != 0;
    return passing_fraction(widths, sigma, depth_min) >= kScreenFromFraction;
}

// scatter of the flux itself: the noise estimate behind pruning_pays (a caller's dy may be in
// arbitrary units -- validate.py:18 normalises it by its mean -- so it says nothing about the noise)
double flux_scatter(const double* y, int64_t n) {
    long double m = 0.0L, v = 0.0L;
    for (int64_t i = 0; i < n; ++i) m += y[i];
    m /= (long double)n;
    for (int64_t i = 0; i < n; ++i) v += (y[i] - m) * (y[i] - m);
    return (double)std::sqrt((double)(v / (long double)n));
}

// Fast prefix-sum mode: half-width of the band around transit_depth_min inside which the plain scan
// cannot decide a window -- 1.25 x the bound 2^-53 c_max on |dX/d - mean_reference| (c_max = (n + W) max|flux| bounds the
// reference's running sum), plus 1e-14 for what the bound leaves out (the plain scan's own rounding, <= ~20 * 2^-53 *
// max|X| / d, and the reference's division; rounds 3 and early 4 shipped 2 x: twice the second attempts for no safety).
// (round 4, when a band hit cost a second search of the period -- Kepler full grid, same box: 0.35 -> 244 ms, 0.1 -> 241,
// 0.01 -> 240, never -> 249.  Round 5: a hit costs one exact prefix pass (band_window in tls_kernels.hip.h) -- every 16th
// Kepler period: 0 (all exact) 16.97 ms, 0.1 14.22, 1 14.02, 10 14.04, never 14.02; TESS 3.01 / 2.72 / 2.72 / 2.72 / 2.73.)
constexpr double kBandMax = 1.0;
constexpr double kBandHitCost = 0.15;   // of a period: the exact prefix pass and the few windows it decides
double fast_mode_eps(int64_t M, double y_abs_max) {
    return 1.25 * (1.1102230246251565e-16 * ((double)M * y_abs_max)) + 1e-14;
}
// Expected number of windows of width row k inside that band, as a prefix over the width table: n_pos * 2 eps * density of
// the window mean at depth_min (a flat, white light curve: mean of 1 - flux ~ N(0, sigma^2 / d)).  A period's expectation
// is pre[k_hi] - pre[k_lo]; above band_max the period starts in exact mode (kernel), and it weighs on the queue order (host).
// n_pos of a row is (M - width) / xth + 1, the same for every period of the plan.
void band_prefix_for(const std::vector<tlsdev::WidthEntry>& widths, double sigma, double depth_min, double eps,
                     std::vector<double>& pre, int64_t M = -1) {
    pre.assign(widths.size() + 1, 0.0);
    for (size_t k = 0; k < widths.size(); ++k) {
        const auto& we = widths[k];
        const double n_pos = M >= 0 ? (double)((M - we.width) / we.xth + 1) : (double)we.n_pos;
        const double sd = sigma / std::sqrt((double)we.width), z = depth_min / sd;
        pre[k + 1] = pre[k] + n_pos * 2.0 * eps * std::exp(-0.5 * z * z) / (sd * 2.5066282746310002);
    }
}

// In-range width window of every period (core.py:143-156) and its trial-cell count.
// The same for every period of a grid (what tls_prepare and tls_grid_cells need): the in-range rows [k_lo, k_hi)
// of the ascending width table by binary search, the dense rows [k_lo, k_x), and the trial-cell count from a
// prefix sum over the table -- per period two pow() calls (t14, kept in the reference's operation order) and a few
// dozen instructions instead of a walk over all widths.  Long grids are cut into slices for a few host threads
// (a Kepler-size grid of 182 388 periods: 21 ms on one core).  Returns false on a non-positive or non-finite period.
struct GridPlan {
    int64_t cells = 0, pairs = 0;
};
bool plan_periods(const std::vector<tlsdev::WidthEntry>& widths, const tls_params* params, const double* periods,
                  int64_t n_periods, double length, int64_t n, int64_t M, tlsdev::PeriodRows* prow, int64_t* cost,
                  GridPlan* total, int plan_threads = -1) {
    const int nw = (int)widths.size();
    std::vector<int> wd((size_t)nw);
    std::vector<int64_t> prefix((size_t)nw + 1, 0);
    int first_strided = nw;   // xth = int(width * margin) never decreases with the width (core.py:50-55)
    for (int k = 0; k < nw; ++k) {
        wd[(size_t)k] = widths[(size_t)k].width;
        prefix[(size_t)k + 1] = prefix[(size_t)k] + ((M - widths[(size_t)k].width) / widths[(size_t)k].xth + 1);
        if (widths[(size_t)k].xth != 1 && first_strided == nw) first_strided = k;
    }
    for (int k = first_strided; k < nw; ++k)
        if (widths[(size_t)k].xth == 1) first_strided = -1;   // not monotone (cannot happen): per-row walk below
    auto slice = [&](int64_t p0, int64_t p1, GridPlan* out, bool* ok) {
        GridPlan g;
        for (int64_t p = p0; p < p1; ++p) {
            const double P = periods[p];
            if (!(P > 0) || !std::isfinite(P)) { *ok = false; return; }
            const double duration_max = t14(params->R_star_max, params->M_star_max, P, false);
            const double duration_min = t14(params->R_star_min, params->M_star_min, P, true);
            const double naive = length / P;
            const double correction = (naive + 1) / naive;
            const double lo = std::floor(duration_min * (double)n);
            const double hi = std::ceil(duration_max * (double)n * correction);
            const int dlo = (int)std::max(-2.0e9, std::min(2.0e9, lo));
            const int dhi = (int)std::max(-2.0e9, std::min(2.0e9, hi));
            const int k_lo = (int)(std::lower_bound(wd.begin(), wd.end(), dlo) - wd.begin());
            const int k_hi = std::max(k_lo, (int)(std::upper_bound(wd.begin(), wd.end(), dhi) - wd.begin()));
            int k_x = std::min(k_hi, std::max(k_lo, first_strided));
            if (first_strided < 0) {
                k_x = k_lo;
                for (int k = k_lo; k < k_hi; ++k) if (widths[(size_t)k].xth == 1) k_x = k + 1;
            }
            const int64_t c = prefix[(size_t)k_hi] - prefix[(size_t)k_lo];
            if (prow) { prow[p].k_lo = k_lo; prow[p].k_hi = k_hi; prow[p].k_x = k_x; prow[p].pad = 0; }
            if (cost) cost[p] = c;
            g.cells += c; g.pairs += k_hi - k_lo;
        }
        *out = g;
    };
    unsigned n_threads = 1;
    if (n_periods >= 4096) {
        n_threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 8u);
        n_threads = (unsigned)std::min<int64_t>(n_threads, n_periods / 2048);
        if (plan_threads > 0) n_threads = (unsigned)std::max(1, std::min(64, plan_threads));
    }
    std::vector<GridPlan> part(n_threads);
    std::vector<char> ok(n_threads, 1);
    if (n_threads <= 1) {
        bool good = true;
        slice(0, n_periods, &part[0], &good);
        ok[0] = good;
    } else {
        std::vector<std::thread> pool;
        std::vector<bool*> flags;
        std::unique_ptr<bool[]> good(new bool[n_threads]);
        for (unsigned i = 0; i < n_threads; ++i) {
            good[i] = true;
            const int64_t p0 = n_periods * i / n_threads, p1 = n_periods * (i + 1) / n_threads;
            pool.emplace_back(slice, p0, p1, &part[i], &good[i]);
        }
        for (auto& th : pool) th.join();
        for (unsigned i = 0; i < n_threads; ++i) ok[i] = good[i];
    }
    for (unsigned i = 0; i < n_threads; ++i) {
        if (!ok[i]) return false;
        total->cells += part[i].cells; total->pairs += part[i].pairs;
    }
    return true;
}

// sort buckets of the general fold_and_sort for a series of n points (the resident kernel uses n; the slab variant
// what its LDS holds): only the order of magnitude matters to the caller
int64_t ctx_nb_for(int64_t n, size_t n_widths) {
    const size_t hdr = ((size_t)tlsdev::kFixedHeader + 4 * (3 * n_widths + 2) + 15) / 16 * 16;
    return std::min<int64_t>(n, (int64_t)((kLdsPerCU - hdr) / 4));
}

// work order of the period queue: most expensive first (longest-processing-time first), ties in grid order --
// a stable LSD radix sort of the 32-bit key (max cost - cost), three passes of 11 bits
void order_by_cost(const std::vector<int64_t>& cost, std::vector<int>& order) {
    const size_t np = cost.size();
    order.resize(np);
    int64_t cmax = 0;
    for (size_t p = 0; p < np; ++p) cmax = std::max(cmax, cost[p]);
    if (cmax >= (1LL << 33)) {   // (absurdly long series: comparison sort)
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
        return;
    }
    std::vector<unsigned long long> key(np), tmp(np);
    for (size_t p = 0; p < np; ++p) key[p] = ((unsigned long long)(cmax - cost[p]) << 31) | (unsigned long long)p;   // p < 2^31
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = 31 + 11 * pass;
        if (pass > 0 && (cmax >> (11 * pass)) == 0) break;
        size_t hist[2049] = {0};
        for (size_t p = 0; p < np; ++p) ++hist[((key[p] >> shift) & 2047u) + 1];
        for (int b = 0; b < 2048; ++b) hist[b + 1] += hist[b];
        for (size_t p = 0; p < np; ++p) tmp[hist[(key[p] >> shift) & 2047u]++] = key[p];
        key.swap(tmp);
    }
    for (size_t p = 0; p < np; ++p) order[p] = (int)(key[p] & 0x7fffffffull);
}

// the two-role kernel of the slab path (fold role, then search role over (period, tile) items)
template <bool UNI, bool COUNTING = false>
hipError_t launch_split(tls_ctx* ctx, const tlsdev::SearchArgs& args, int blocks) {
    auto kernel = tlsdev::tls_fold_search_kernel<UNI, COUNTING>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3((unsigned)ctx->threads), ctx->lds_bytes,
                       ctx->stream, args);
    return hipGetLastError();
}

template <bool RES, bool UNI, typename IdxT, bool PRUNING = false, bool COUNTING = false, bool SCREEN = false>
hipError_t launch_variant(tls_ctx* ctx, const tlsdev::SearchArgs& args, int blocks) {
    auto kernel = tlsdev::tls_search_kernel<RES, UNI, IdxT, PRUNING, COUNTING, SCREEN>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3((unsigned)ctx->threads), ctx->lds_bytes,
                       ctx->stream, args);
    return hipGetLastError();
}

// the event pair around one search (tls_kernel_timing adds them up): one kernel, or the fold + search kernels of all
// batches of the two-kernel slab path
hipError_t timing_pair(tls_ctx* ctx, std::pair<hipEvent_t, hipEvent_t>** out) {
    if (ctx->ev_pool.size() < kEventRing) {   // a ring: a long-lived survey process never grows it
        hipEvent_t a, b;
        hipError_t e;
        if ((e = hipEventCreate(&a)) != hipSuccess || (e = hipEventCreate(&b)) != hipSuccess) return e;
        ctx->ev_pool.emplace_back(a, b);
    }
    *out = &ctx->ev_pool[ctx->ev_used++ % kEventRing];
    return hipSuccess;
}

int enqueue(tls_ctx* ctx, bool count_work, bool phase_clock = false, double* debug_folded = nullptr, double* debug_prefix = nullptr,
            unsigned long long* period_cycles = nullptr) {
    if (count_work)
        TLS_HIP(ctx, hipMemsetAsync(ctx->d_counters.ptr, 0, 3 * sizeof(unsigned long long), ctx->stream));
    tlsdev::SearchArgs a;
    a.t = ctx->d_t.ptr; a.y = ctx->over_y ? ctx->over_y : ctx->d_y.ptr;
    a.w = ctx->uniform_w ? nullptr : (ctx->over_w ? ctx->over_w : ctx->d_w.ptr);
    a.periods = ctx->d_periods.ptr; a.order = ctx->d_order.ptr; a.rows = ctx->d_rows.ptr;
    a.widths = ctx->d_widths.ptr; a.q = ctx->d_q.ptr; a.q2 = ctx->uniform_w ? nullptr : ctx->d_q2.ptr;
    a.q32 = ctx->uniform_w ? reinterpret_cast<const float*>(ctx->d_q2.ptr) : nullptr;
    a.g = ((ctx->resident && ctx->slim_blocks == 0) || (!ctx->resident && ctx->opt.x_staged == 2)) ? nullptr : ctx->d_g.ptr;   // (x_staged = 2: A/B switch, X at staging time but the dot products on the re-staged samples)
    a.split_lo = nullptr; a.park_cells = nullptr; a.e_abs_max = ctx->e_abs_max; a.q32_shifted = ctx->q_count;
    a.screens = ctx->d_screens.ptr;
    a.out_chi2 = ctx->over_chi2 ? ctx->over_chi2 : ctx->d_chi2.ptr;
    a.out_row = ctx->over_row ? ctx->over_row : ctx->d_row.ptr;
    a.out_depth = ctx->over_depth ? ctx->over_depth : ctx->d_depth.ptr;
    a.counters = count_work ? ctx->d_counters.ptr : nullptr;
    a.phase_cycles = nullptr;
    if (phase_clock) {
        TLS_HIP(ctx, ctx->d_phase.reserve(tlsdev::kPhases));
        TLS_HIP(ctx, hipMemsetAsync(ctx->d_phase.ptr, 0, tlsdev::kPhases * sizeof(unsigned long long), ctx->stream));
        a.phase_cycles = ctx->d_phase.ptr;
    }
    a.debug_folded = debug_folded; a.debug_prefix = debug_prefix; a.period_cycles = period_cycles;
    a.check = nullptr; a.lds_bytes = (long long)ctx->lds_bytes;
#ifdef TLS_DEBUG_CHECKS
    if (!ctx->d_check.ptr) {
        TLS_HIP(ctx, ctx->d_check.reserve(tlsdev::kChecks));
        TLS_HIP(ctx, hipMemsetAsync(ctx->d_check.ptr, 0, tlsdev::kChecks * sizeof(unsigned long long), ctx->stream));
    }
    a.check = ctx->d_check.ptr;
#endif
    a.queue = ctx->d_squeue.ptr;
    a.scratch = ctx->d_scratch.ptr;
    a.scratch_stride = (long long)(ctx->uniform_w ? 2 : 3) * ((ctx->M + 1 + ctx->region_pad + 1) & ~1);   // even regions (kernel: RS)
    a.region_pad = ctx->region_pad;
    a.chunk_lists = ctx->d_lists.ptr; a.list_stride = 3 * (long long)ctx->list_stride; a.list_cap = (long long)ctx->list_stride;
    a.prune_min_live = ctx->prune_min_live; a.p2_shift = ctx->p2_shift; a.hdr_bytes = ctx->hdr_bytes; a.tile_len = ctx->tile_len; a.tile_halo = ctx->tile_halo;
    a.depth_min = ctx->depth_min; a.S0 = ctx->S0; a.w0 = ctx->w0;
    a.cumsum_round = ctx->cumsum_round;
    {
        // the sequential cumsum C of the reference (helpers.py:72) rounds by at most half an ulp of its running value
        // per step, and C <= (n + W) * max|flux|: the two constants below follow from that (tls_kernels.hip.h,
        // depth_pass and window_bound)
        const double c_max = (double)ctx->M * ctx->y_abs_max;
        // (band half-width: 1.25 x the bound 2^-53 c_max on |dX/d - mean_reference|, plus 1e-14 for what the bound leaves out --
        // the plain scan's own rounding, <= ~20 * 2^-53 * max|X| / d, and the reference's division; rounds 3 and early 4
        // shipped 2 x: twice the second attempts for no additional safety)
        a.eps_fast = fast_mode_eps(ctx->M, ctx->y_abs_max);
        a.slack_unit = 2.5e-16 * c_max;
        a.exact_prefix = ctx->opt.exact_prefix == 1 ? 1 : 0;
        // Series in the HBM slab, one-workgroup-per-period kernel: fast mode too (switch fast_slab = 0: exact mode).
        // The band grows with the series (eps ~ 2^-52 (n + W) max|flux|) while the noise of a window mean shrinks: 2.6 % of
        // the TESS-size and 10 % of the Kepler-size periods hit it and go through the prefix sum and phase 3 a second time
        // (the folded flux is kept).  Round 4, same box: Kepler full grid 275.8 -> 254.8 ms, TESS 2.97 -> 2.90 ms.
        // WHICH mode a period takes depends on the light curve and the period alone (the expectation below) -- never on how
        // many other periods the launch holds or on the device: a shard of a multi-GPU search returns the bits of the
        // full-grid search.  (Round 4 kept launches of <= 4 rounds exact to spare them a late second attempt; a period's
        // bits then depended on the launch.  The queue order now sends the periods most likely to need one first.)
        a.fast_slab = ctx->opt.fast_slab == 0 ? 0 : 1;
    }
    a.x_at_staging = 0;
    if (!ctx->resident && a.fast_slab) {
        bool any_oversize = false;   // (rows evaluated straight from the slab list their cells with the first tile: they need all of X)
        for (const auto& we : ctx->host_widths) any_oversize = any_oversize || we.oversize != 0;
        a.x_at_staging = any_oversize ? 0 : 1;
        if (ctx->opt.x_staged == 0) a.x_at_staging = 0;
    }
    a.band_prefix = nullptr;
    a.band_max = ctx->opt.band_max >= 0 ? ctx->opt.band_max : kBandMax;
    if (!ctx->resident && a.fast_slab && ctx->flux_sigma > 0 &&
        !(ctx->band_sigma == ctx->flux_sigma && ctx->band_eps == a.eps_fast && ctx->d_band_now)) {
        // the band expectation of every width row (band_prefix_for), as a prefix over the width table: the kernel forms a
        // period's expectation from its duration window [k_lo, k_hi).  Uploaded from one of two pinned slots, nothing is
        // waited for (a survey changes sigma with every group of light curves: the launch in flight keeps reading its own slot).
        const size_t cnt = ctx->host_widths.size() + 1;
        if (ctx->h_band_cap < 2 * cnt) {
            TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->h_band) TLS_HIP(ctx, hipHostFree(ctx->h_band));
            ctx->h_band = nullptr; ctx->h_band_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_band), 2 * cnt * sizeof(double), hipHostMallocDefault));
            ctx->h_band_cap = 2 * cnt;
            TLS_HIP(ctx, ctx->d_band.reserve(2 * cnt));
            for (auto& ev : ctx->ev_band) if (!ev) TLS_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->band_used[0] = ctx->band_used[1] = false;
        }
        const int slot = ctx->band_slot ^= 1;
        if (ctx->band_used[slot]) TLS_HIP(ctx, hipEventSynchronize(ctx->ev_band[slot]));   // (two uploads ago: long done)
        double* h = ctx->h_band + (size_t)slot * cnt;
        std::vector<double> pre;
        band_prefix_for(ctx->host_widths, ctx->flux_sigma, ctx->depth_min, a.eps_fast, pre);
        std::memcpy(h, pre.data(), cnt * sizeof(double));
        ctx->d_band_now = ctx->d_band.ptr + (size_t)slot * cnt;
        TLS_HIP(ctx, hipMemcpyAsync(ctx->d_band_now, h, cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TLS_HIP(ctx, hipEventRecord(ctx->ev_band[slot], ctx->stream));
        ctx->band_used[slot] = true;
        ctx->band_sigma = ctx->flux_sigma; ctx->band_eps = a.eps_fast;
    }
    if (!ctx->resident && a.fast_slab && ctx->flux_sigma > 0) a.band_prefix = ctx->d_band_now;
    a.sort2 = ctx->sort2 ? 1 : 0;
    a.n_curves = ctx->batch_curves;
    a.curve_S0 = ctx->over_S0 ? ctx->over_S0 : ctx->d_curve_S0.ptr;
    a.curve_w0 = ctx->over_w0 ? ctx->over_w0 : ctx->d_curve_w0.ptr;
    a.perm_scratch = ctx->d_perm.ptr;
    a.n = ctx->n; a.W = ctx->W; a.M = ctx->M;
    a.n_periods = ctx->n_periods; a.n_widths = ctx->n_widths; a.nb = ctx->nb;
    a.batch_lo = 0; a.batch_n = 0; a.tile_prefix = ctx->d_tile_prefix.ptr;
    a.partials = ctx->d_partials.ptr; a.tiles_done = ctx->d_tiles_done.ptr;
    a.fold_ready = ctx->d_tiles_done.ptr ? ctx->d_tiles_done.ptr + ctx->split_batch : nullptr;   // (only the two-role plan has them)
    a.split_fast = ctx->split_fast && a.x_at_staging && a.g != nullptr ? 1 : 0;
    hipError_t e;
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    if ((e = timing_pair(ctx, &evp)) != hipSuccess) return fail(ctx, TLS_E_HIP, std::string("timing events: ") + hipGetErrorString(e));
    if ((e = hipEventRecord(evp->first, ctx->stream)) != hipSuccess) {
        --ctx->ev_used;
        return fail(ctx, TLS_E_HIP, std::string("timing events: ") + hipGetErrorString(e));
    }
    // pruning variant: LDS-resident series, uniform weights, noisy enough that most trial cells pass the depth predicate,
    // and not while the evaluated cells are being counted (counting means evaluating all of them)
    const bool prune = ctx->resident && ctx->uniform_w && ctx->prune_kernel && !count_work;
    // (counting has an instantiation of its own: the plain kernels do not keep the counters)
#define TLS_LAUNCH_RESIDENT(BLOCKS)                                                                                          \
    (!ctx->uniform_w ? (count_work ? launch_variant<true, false, unsigned short, false, true>(ctx, a, BLOCKS)                   \
                                   : launch_variant<true, false, unsigned short, false, false>(ctx, a, BLOCKS))                  \
     : prune         ? launch_variant<true, true, unsigned short, true, false>(ctx, a, BLOCKS)                                   \
     : count_work    ? launch_variant<true, true, unsigned short, false, true>(ctx, a, BLOCKS)                                   \
                     : launch_variant<true, true, unsigned short, false, false>(ctx, a, BLOCKS))
#define TLS_LAUNCH_SLAB(BLOCKS)                                                                                              \
    (!ctx->uniform_w ? (count_work ? launch_variant<false, false, unsigned int, false, true>(ctx, a, BLOCKS)                    \
                                   : launch_variant<false, false, unsigned int, false, false>(ctx, a, BLOCKS))                   \
     : count_work    ? launch_variant<false, true, unsigned int, false, true>(ctx, a, BLOCKS)                                    \
                     : launch_variant<false, true, unsigned int, false, false>(ctx, a, BLOCKS))
#define TLS_LAUNCH_SPLIT(BLOCKS)                                                                                             \
    (!ctx->uniform_w ? (count_work ? launch_split<false, true>(ctx, a, BLOCKS) : launch_split<false, false>(ctx, a, BLOCKS))    \
     : count_work    ? launch_split<true, true>(ctx, a, BLOCKS)                                                              \
                     : launch_split<true, false>(ctx, a, BLOCKS))
    const bool split = !ctx->resident && ctx->split && ctx->batch_curves == 1;
    // fp32 screen of the dot products (tlsdev::screen_cells): where the host expects it to pay (screen_pays); counting the
    // work and the debug entries run the plain variant, whose bits it returns anyway
    const bool screen = ctx->screen_kernel && screen_admissible(ctx->resident, ctx->uniform_w, ctx->e_abs_max) && !prune &&
                        !count_work && !debug_folded && !debug_prefix;
    const char* kernel_name = ctx->resident ? (prune ? "resident+prune" : "resident") : split ? "slab+split" : "slab";
    if (screen) {
        kernel_name = "resident+screen32";
        const size_t region = (size_t)ctx->M + 1 + (size_t)ctx->region_pad;
        hipError_t er = ctx->d_split.reserve((size_t)ctx->blocks * region);
        if (er != hipSuccess) { --ctx->ev_used; return fail(ctx, TLS_E_HIP, std::string("fp32 screen scratch: ") + hipGetErrorString(er)); }
        a.split_lo = ctx->d_split.ptr;
        er = ctx->d_park.reserve((size_t)ctx->blocks * tlsdev::kParkCap * 2);   // (a ParkedCell is two doubles wide)
        if (er != hipSuccess) { --ctx->ev_used; return fail(ctx, TLS_E_HIP, std::string("fp32 screen scratch: ") + hipGetErrorString(er)); }
        a.park_cells = ctx->d_park.ptr;
        e = launch_variant<true, true, unsigned short, false, false, true>(ctx, a, ctx->blocks);
    } else if (ctx->slim_blocks > 0 && ctx->uniform_w && !ctx->prune_kernel &&
               !(ctx->screen_kernel && screen_admissible(ctx->resident, ctx->uniform_w, ctx->e_abs_max)) && !debug_folded && !debug_prefix) {
        // (by the plan's choice, not this launch's: a search that counts its work runs the counting instantiation of the kernel
        // the plain search takes -- the classic family where pruning or the screen is the host's choice -- and returns its bits)
        // four period slots per CU (tls_slim_kernel.hip.h): plain variant, uniform weights, 256-thread workgroups
        kernel_name = "slim";
        a.lds_bytes = (long long)ctx->slim_lds;
        const bool wide = ctx->slim_threads == tlsdev::kSlimThreadsWide;
        auto kernel = wide ? (count_work ? tlsdev::tls_slim_kernel<true, tlsdev::kSlimThreadsWide> : tlsdev::tls_slim_kernel<false, tlsdev::kSlimThreadsWide>)
                           : (count_work ? tlsdev::tls_slim_kernel<true, tlsdev::kSlimThreads> : tlsdev::tls_slim_kernel<false, tlsdev::kSlimThreads>);
        if (wide) kernel_name = "slim512";
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->slim_lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(kernel, dim3((unsigned)ctx->slim_blocks), dim3((unsigned)ctx->slim_threads), ctx->slim_lds, ctx->stream, a);
            e = hipGetLastError();
        }
    } else if (ctx->resident) e = TLS_LAUNCH_RESIDENT(ctx->blocks);
    else if (!split) e = TLS_LAUNCH_SLAB(ctx->blocks);
    else {
        // series in the HBM slab, one light curve: per batch of periods ONE launch of the two-role kernel -- every
        // workgroup folds periods of the batch until none is left (one slab per period), then searches (period, tile) items
        e = hipSuccess;
        for (int lo = 0; lo < ctx->n_periods && e == hipSuccess; lo += ctx->split_batch) {
            a.batch_lo = lo; a.batch_n = std::min(ctx->split_batch, ctx->n_periods - lo);
            a.queue = ctx->d_squeue.ptr;   // [0..1] the fold role's queue, [2..3] the search role's
            const int64_t items = (int64_t)ctx->host_tile_prefix[(size_t)(lo + a.batch_n)] - (int64_t)ctx->host_tile_prefix[(size_t)lo];
            const int blocks = (int)std::min<int64_t>(ctx->split_blocks, std::max<int64_t>(items, 1));
            e = TLS_LAUNCH_SPLIT(blocks);
        }
    }
#undef TLS_LAUNCH_SPLIT
#undef TLS_LAUNCH_SLAB
#undef TLS_LAUNCH_RESIDENT
    // (a failure from here on gives the event pair back: tls_kernel_timing must not meet a pair whose second event was
    // never recorded)
    if (e != hipSuccess) { --ctx->ev_used; return fail(ctx, TLS_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e)); }
    if ((e = hipEventRecord(evp->second, ctx->stream)) != hipSuccess) {
        --ctx->ev_used;
        return fail(ctx, TLS_E_HIP, std::string("timing events: ") + hipGetErrorString(e));
    }
    ctx->executed = true;
    ctx->counted = count_work;
    ctx->last_kernel = kernel_name;
    return TLS_OK;
}

// T0-fit launch shared by tls_t0_fit and tls_power_batch: every pointer on the device, nothing waited for.
// One fit (d_params == nullptr: period, dur, roll, n_epochs from the host) or `n_fits` fits in ONE launch (blockIdx.y = fit;
// their parameters, epochs and signals written on the device by tls_power_prep, the arrays `*_stride` doubles apart).
// Three launches (round 6): the base order of every fit (one workgroup a fit), the epochs as rotations of it (one wavefront
// an epoch: tls_t0fit_rot), and the general kernel for the fits the rotation path handed back (ties, gaps the folds' rounding
// could close: its workgroups leave at once otherwise).  Switch t0_rot = 0: the general kernel alone.
int launch_t0_fit(tls_ctx* ctx, const double* d_t, const double* d_y, const double* d_signal, const double* d_epochs,
                  double* d_residuals, unsigned int* d_queue, int64_t n, double period, int64_t dur, int64_t n_epochs,
                  int64_t roll, double t_lo, double t_hi, const tlsdev::T0FitParams* d_params = nullptr, int64_t n_fits = 1,
                  int64_t y_stride = 0, int64_t signal_stride = 0, int64_t epoch_stride = 0) {
    tlsdev::T0FitArgs a;
    a.t = d_t; a.y = d_y; a.signal = d_signal; a.epochs = d_epochs;
    a.residuals = d_residuals; a.queue = d_queue; a.scratch = nullptr; a.scratch_stride = 0;
    a.period = period; a.n = (int)n; a.dur = (int)dur; a.roll = (int)(roll % n); a.n_epochs = (int)n_epochs;
    a.params = d_params; a.y_stride = y_stride; a.signal_stride = signal_stride; a.epoch_stride = epoch_stride;
    a.mode = 0; a.rot = nullptr; a.rot_perm = nullptr; a.rot_stride = 0; a.t_lo = t_lo; a.t_hi = t_hi;
    const size_t hdr = 272;
    const size_t resident_bytes = hdr + 16 * (size_t)n;
    const bool resident = resident_bytes <= kLdsPerCU && n <= 65535;
    size_t lds; int threads, blocks;
    // (a batched launch does not know its fits' epoch counts on the host: every fit gets the full set of workgroups, those
    // beyond its epochs leave at once)
    const int64_t epochs_cap = d_params ? n : n_epochs;
    if (resident) {
        a.nb = (int)n; lds = resident_bytes;
        const size_t per_cu = kLdsPerCU / resident_bytes;
        threads = per_cu >= 2 ? 512 : 1024;
        const size_t wg_per_cu = std::min<size_t>(per_cu, 2048 / (size_t)threads);
        blocks = (int)std::min<int64_t>(epochs_cap, (int64_t)wg_per_cu * ctx->n_cu);
    } else {
        a.nb = (int)std::min<int64_t>(n, 16384); lds = hdr + 4 * (size_t)a.nb;
        threads = 512; blocks = (int)std::min<int64_t>(epochs_cap, (int64_t)2 * ctx->n_cu);
        a.scratch_stride = 3 * n;
        TLS_HIP(ctx, ctx->d_fscratch.reserve((size_t)blocks * (size_t)n_fits * (size_t)a.scratch_stride));
        a.scratch = ctx->d_fscratch.ptr;
    }
    if (blocks < 1) return TLS_OK;
    const unsigned fits = (unsigned)std::max<int64_t>(n_fits, 1);
    const bool rotation = ctx->opt.t0_rot != 0 && n >= tlsdev::kT0RotMinPoints;
    if (rotation) {
        a.rot_stride = 3 * (long long)n + 4;
        TLS_HIP(ctx, ctx->d_frot.reserve((size_t)fits * (size_t)a.rot_stride));
        TLS_HIP(ctx, ctx->d_frperm.reserve((size_t)fits * (size_t)n));
        a.rot = ctx->d_frot.ptr; a.rot_perm = ctx->d_frperm.ptr;
    }
    auto launch = [&](int mode, unsigned grid_x) -> hipError_t {
        tlsdev::T0FitArgs b = a;
        b.mode = mode;
        const dim3 grid(grid_x, fits);
        hipError_t e;
        if (resident) {
            auto kernel = tlsdev::tls_t0fit_kernel<true, unsigned short>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) { hipLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), lds, ctx->stream, b); e = hipGetLastError(); }
        } else {
            auto kernel = tlsdev::tls_t0fit_kernel<false, unsigned int>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess) { hipLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), lds, ctx->stream, b); e = hipGetLastError(); }
        }
        return e;
    };
    hipError_t e = hipSuccess;
    if (rotation) {
        e = launch(1, 1u);
        if (e == hipSuccess) {
            const unsigned waves_per_wg = 4;
            const dim3 grid((unsigned)((epochs_cap + waves_per_wg - 1) / waves_per_wg), fits);
            hipLaunchKernelGGL(tlsdev::tls_t0fit_rot, grid, dim3(waves_per_wg * 64), 0, ctx->stream, a);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = launch(2, (unsigned)blocks);
    } else {
        e = launch(0, (unsigned)blocks);
    }
    if (e != hipSuccess) return fail(ctx, TLS_E_HIP, std::string("t0 fit launch: ") + hipGetErrorString(e));
    return TLS_OK;
}

// compute units of the first visible device; 256 (MI355X) where no device can be asked (host-only planning)
int visible_compute_units() {
    static int cached = 0;
    if (cached > 0) return cached;
    int count = 0, cus = 0;
    if (hipGetDeviceCount(&count) == hipSuccess && count > 0 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0) == hipSuccess && cus > 0) cached = cus;
    else { (void)hipGetLastError(); cached = 256; }
    return cached;
}

}  // namespace

extern "C" {

const char* tls_version(void) { return "tls_amd 0.4 (gfx950)"; }

int tls_abi_version(void) { return TLS_AMD_ABI_VERSION; }

int tls_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return TLS_E_HIP; }
    return n;
}

const char* tls_last_error(const tls_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

const char* tls_device_name(const tls_ctx* ctx) { return ctx ? ctx->name.c_str() : ""; }

tls_ctx* tls_ctx_create(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no usable GPU (hipGetDeviceCount: ") + hipGetErrorString(e) +
                         "); this library has no CPU fallback";
        return nullptr;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device_id " + std::to_string(device_id) + " out of range (" + std::to_string(n) + " GPUs)";
        return nullptr;
    }
    tls_ctx* ctx = new (std::nothrow) tls_ctx();
    if (!ctx) { g_create_error = "out of host memory"; return nullptr; }
    ctx->device = device_id;
    ctx->opt = process_options();
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev0)) != hipSuccess || (e = hipEventCreate(&ctx->ev1)) != hipSuccess) {
        g_create_error = std::string("context setup: ") + hipGetErrorString(e);
        delete ctx;
        return nullptr;
    }
    ctx->n_cu = prop.multiProcessorCount;
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s %s, %d CUs, %.0f GiB", prop.gcnArchName, prop.name, ctx->n_cu,
                  (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    ctx->name = buf;
    return ctx;
}

int tls_get_options(const tls_ctx* ctx, tls_options* out) {
    if (!ctx || !out) return TLS_E_ARG;
    out->exact_prefix = ctx->opt.exact_prefix; out->slim = ctx->opt.slim;
    return TLS_OK;
}

namespace {
int adopt_switches(tls_ctx* ctx, const Switches& o) {
    if (std::memcmp(&o, &ctx->opt, sizeof o) == 0) return TLS_OK;
    ctx->opt = o;
    // a prepared plan was built for the old switches: the next tls_prepare plans again (the key holds them too)
    ctx->prepared = false; ctx->executed = false;
    return TLS_OK;
}
}  // namespace

int tls_set_options(tls_ctx* ctx, const tls_options* opt) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!opt) return fail(ctx, TLS_E_ARG, "null options");
    Switches o = ctx->opt;
    o.exact_prefix = opt->exact_prefix < 0 ? -1 : opt->exact_prefix;
    o.slim = opt->slim < 0 ? -1 : opt->slim;
    return adopt_switches(ctx, o);
}

int tls_debug_set_switch(tls_ctx* ctx, const char* name, double value) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    const SwitchName* sw = find_switch(name);
    if (!sw) return fail(ctx, TLS_E_ARG, std::string("unknown switch: ") + (name ? name : "(null)"));
    Switches o = ctx->opt;
    switch_store(o, *sw, value);
    return adopt_switches(ctx, o);
}

int tls_debug_get_switches(const tls_ctx* ctx, char* out, int64_t capacity) {
    if (!out || capacity < 1) return TLS_E_ARG;
    const Switches& o = ctx ? ctx->opt : process_options();
    std::string text;
    for (const auto& sw : kSwitchNames) {
        char item[96];
        std::snprintf(item, sizeof item, "%s%s=%.17g", text.empty() ? "" : ",", sw.name, switch_load(o, sw));
        text += item;
    }
    if ((int64_t)text.size() + 1 > capacity) return TLS_E_ARG;
    std::memcpy(out, text.c_str(), text.size() + 1);
    return TLS_OK;
}

void tls_ctx_destroy(tls_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) (void)ncclCommDestroy(ctx->comm);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->d_plan.release(); ctx->d_out.release();
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    if (ctx->ev_stage) (void)hipEventDestroy(ctx->ev_stage);
    ctx->d_scratch.release(); ctx->d_pack.release();
    ctx->d_gather.release(); ctx->d_scalar.release(); ctx->d_stage.release();
    ctx->d_partials.release(); ctx->d_tiles_done.release(); ctx->d_check.release(); ctx->d_spec.release(); ctx->d_queue.release(); ctx->d_squeue.release(); ctx->d_pqueues.release(); ctx->d_phase.release(); ctx->d_lists.release(); ctx->d_perm.release(); ctx->d_curve_S0.release(); ctx->d_curve_w0.release();
    ctx->d_ft.release(); ctx->d_fy.release(); ctx->d_fsig.release(); ctx->d_fep.release(); ctx->d_fres.release(); ctx->d_fscratch.release(); ctx->d_frot.release(); ctx->d_frperm.release(); ctx->d_pink.release();
    ctx->d_split.release(); ctx->d_park.release(); ctx->d_band.release();
    if (ctx->h_band) (void)hipHostFree(ctx->h_band);
    for (auto& ev : ctx->ev_band) if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : ctx->slot) {
        sl.d_y.release(); sl.d_w.release(); sl.d_S0.release(); sl.d_w0.release(); sl.d_chi2.release(); sl.d_depth.release(); sl.d_row.release();
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_kernel) (void)hipEventDestroy(sl.ev_kernel);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (auto& evp : ctx->ev_pool) { (void)hipEventDestroy(evp.first); (void)hipEventDestroy(evp.second); }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tls_prepare(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n,
                const double* periods, int64_t n_periods, const tls_template* tmpl, const tls_params* params) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    ctx->prepared = false; ctx->executed = false;
    if (!t || !y || !dy || !periods || !tmpl || !params) return fail(ctx, TLS_E_ARG, "null argument");
    if (n < 3 || n > 50000000) return fail(ctx, TLS_E_ARG, "n out of range (need 3 <= n <= 5e7)");
    if (n_periods < 0 || n_periods > 100000000) return fail(ctx, TLS_E_ARG, "n_periods out of range");
    if (tmpl->n_rows < 1 || !tmpl->values || !tmpl->offset || !tmpl->length || !tmpl->width || !tmpl->overshoot)
        return fail(ctx, TLS_E_ARG, "empty template table");
    TLS_HIP(ctx, hipSetDevice(ctx->device));

    // The same time stamps, periods, template and parameters as the plan this context already holds (a survey, or
    // repeated power() calls): only the flux is new.  tls_prepare then costs two passes over y and one upload.
    if (ctx->key.valid && key_matches(ctx->key, ctx->opt, t, n, periods, n_periods, tmpl, params)) {
        const int rcu = update_flux_impl(ctx, y, dy);
        if (rcu == TLS_OK) { ctx->prepared = true; ++ctx->plan_reuses; return TLS_OK; }
        if (rcu != kWeightsDiffer) return rcu;
        // uniform dy after per-point dy (or the reverse): a different kernel variant and layout -- plan again
    }
    ctx->key.valid = false;

    std::vector<tlsdev::WidthEntry> widths;
    std::vector<double> q;
    { int rcw = build_widths(ctx, tmpl, params, n, widths, &q); if (rcw) return rcw; }
    int64_t W = widths.back().width;  // core.py:114-116
    if (W % 2 != 0) W += 1;
    const int64_t M = n + W;
    if (M + 1 > 0x7fffffff / 4) return fail(ctx, TLS_E_ARG, "series too long");
    if (M < 4 * tlsdev::kR) return fail(ctx, TLS_E_ARG, "series too short (need n + widest width >= 20 samples)");

    // per-period duration window (core.py:143-156) and cost
    double t_min = t[0], t_max = t[0];
    for (int64_t i = 1; i < n; ++i) { t_min = std::min(t_min, t[i]); t_max = std::max(t_max, t[i]); }
    std::vector<int> order;
    std::vector<tlsdev::PeriodRows> prow((size_t)n_periods);
    std::vector<int64_t> cost((size_t)n_periods);
    tls_counters pc = {0, 0, 0, 0, 0};
    {
        GridPlan gp;
        if (!plan_periods(widths, params, periods, n_periods, t_max - t_min, n, M, prow.data(), cost.data(), &gp, ctx->opt.plan_threads))
            return fail(ctx, TLS_E_ARG, "periods must be positive and finite");
        pc.grid_cells = gp.cells; pc.pd_pairs = gp.pairs;
    }
    // weights
    std::vector<double> w;
    bool uniform; double w0, S0;
    double y_abs_max = 0.0, e_abs_max = 0.0;
    weights_from(y, dy, n, uniform, w0, w, S0, &y_abs_max, &e_abs_max);
    ctx->y_abs_max = y_abs_max; ctx->e_abs_max = e_abs_max;
    const double flux_sigma = flux_scatter(y, n);

    // Work order: most expensive first.  A period commensurate with the cadence of a regularly sampled series piles
    // the phases onto a few values and its sort costs several ordinary periods: such a period
    // goes to the head of the queue, where its long run overlaps everything else instead of ending the launch.
    {
        std::vector<int64_t> queue_cost(cost);
        bool regular = n >= 64;
        double dt = 0.0;
        if (regular) {
            dt = (t[n - 1] - t[0]) / (double)(n - 1);
            regular = dt > 0;
            for (int64_t i = 1; i < n && regular; ++i) regular = std::fabs((t[i] - t[i - 1]) - dt) <= 1e-3 * dt;
        }
        if (regular) {
            // (the four-slot kernel ranks piles from 9 points on by themselves, 2-3 x an ordinary period: a short series
            // looks for smaller piles and higher resonances -- flagging too many only reorders the queue)
            const bool fine_piles = uniform && n <= (int64_t)tlsdev::kSlimThreadsWide * tlsdev::kSlimPer;
            const int k_max = fine_piles ? 8 : 4;
            const double a_max = (double)n / (fine_piles ? 9.0 : 48.0);       // `a` distinct phase values: piles of n / a points
            const double n_buckets = fine_piles ? 0.5 * (double)n : (double)ctx_nb_for(n, widths.size());
            const double drift = (fine_piles ? 8.0 : 4.0) / ((double)n * n_buckets);   // a pile's phase range, in buckets, over the series
            const double inv_dt = 1.0 / dt;
            for (int64_t p = 0; p < n_periods; ++p) {
                const double r = periods[p] * inv_dt;             // samples per period
                for (int k = 1; k <= k_max; ++k) {
                    const double rk = r * k, a = std::floor(rk + 0.5);   // r ~ a / k
                    if (a > a_max) break;
                    if (a < 1) continue;
                    // n |k/a - 1/r| n_buckets < limit  <=>  |r k - a| < limit a r / (n n_buckets)
                    if (std::fabs(rk - a) < drift * a * r) { queue_cost[(size_t)p] += 50 * cost[(size_t)p]; break; }
                }
            }
        }
        // Fast prefix-sum mode: a period whose windows are likely to meet the undecided band pays a second attempt
        // pass (kBandHitCost of itself); among periods of similar cost the likelier ones start first, so that the extra
        // passes fall into the body of the launch and not into its last round.  (Only the order: which mode a period takes
        // never depends on it.)  The expectation is band_prefix_for's, as in enqueue; the LDS-resident kernel has no
        // per-period expectation (every period starts in fast mode): the same weight orders its queue.
        if (flux_sigma > 0 && ctx->opt.exact_prefix != 1) {
            std::vector<double> pre;
            band_prefix_for(widths, flux_sigma, params->transit_depth_min, fast_mode_eps(M, y_abs_max), pre, M);
            const double band_max = ctx->opt.band_max >= 0 ? ctx->opt.band_max : kBandMax;
            for (int64_t p = 0; p < n_periods; ++p) {
                const double lambda = pre[(size_t)prow[(size_t)p].k_hi] - pre[(size_t)prow[(size_t)p].k_lo];
                if (lambda > band_max) continue;                                   // (starts in exact mode: no second attempt)
                queue_cost[(size_t)p] += (int64_t)(kBandHitCost * std::min(1.0, lambda) * (double)cost[(size_t)p]);
            }
        }
        order_by_cost(queue_cost, order);
    }

    // launch geometry
    const size_t regions = uniform ? 2 : 3;
    int widest_stride = 1;  // of the tiled rows: sizes the pads behind the folded series and the tile halo
    for (const auto& we : widths) if (we.tiled) widest_stride = std::max(widest_stride, we.xth);
    ctx->region_pad = tlsdev::region_pad_for(widest_stride);
    const size_t region_doubles = (size_t)(M + 1 + ctx->region_pad);
    // LDS header: fixed part + per-row live counters and batch prefix (+ the batch counter)
    const size_t hdr = ((size_t)tlsdev::kFixedHeader + 4 * (3 * widths.size() + 2) + 15) / 16 * 16;
    const size_t resident_bytes = hdr + regions * 8 * region_doubles;
    ctx->hdr_bytes = (int)hdr;
    ctx->resident = resident_bytes <= kLdsPerCU && n <= 65535;
    ctx->slim_blocks = 0; ctx->slim_lds = 0;
    if (ctx->resident) {
        ctx->nb = (int)n;
        ctx->tile_len = 0; ctx->tile_halo = 0; ctx->sort2 = false;
        ctx->lds_bytes = resident_bytes;
        const size_t per_cu = kLdsPerCU / resident_bytes;
        ctx->threads = per_cu >= 2 ? 512 : 1024;
        if (ctx->opt.threads > 0) ctx->threads = std::max(64, std::min(1024, ctx->opt.threads / 64 * 64));   // developer switch
        const size_t wg_per_cu = std::min<size_t>(per_cu, 2048 / (size_t)ctx->threads);
        ctx->blocks = (int)std::min<int64_t>(std::max<int64_t>(n_periods, 1), (int64_t)wg_per_cu * ctx->n_cu);
        if (ctx->opt.blocks > 0) ctx->blocks = std::max(1, std::min(ctx->blocks, ctx->opt.blocks));   // developer switch
        // Four (at least three) 256-thread workgroups per CU, phase 3 on X alone (tls_slim_kernel.hip.h): uniform weights, and
        // the period's one region + header within a quarter (a third) of the LDS.  (switch slim = 0: never.)
        // (auto: only while the library also decides between the classic kernel's variants -- an explicit switch prune
        // or ::screen32 selects among THOSE; slim = 1 forces this kernel wherever neither pruning nor the screen is taken)
        // (exact prefix-sum mode throughout is the classic kernel's: this one values its cells on the plain scan, and keeps
        // the exact prefix sum for the windows the plain scan cannot decide)
        const bool slim_wanted = ctx->opt.exact_prefix != 1 && (ctx->opt.slim == 1 || (ctx->opt.slim < 0 && ctx->opt.prune < 0 && ctx->opt.screen32 < 0));
        if (uniform && slim_wanted && ctx->opt.threads <= 0) {
            const long long need = tlsdev::slim_lds_bytes((int)n, (int)M, ctx->region_pad, (int)widths.size());
            // (a series beyond 5120 points -- 107-200 d at 30 min, two TESS sectors at 10 min --: the same kernel with 512-thread
            // workgroups, two to a CU, where the classic kernel runs ONE 1024-thread workgroup per CU; round 6)
            const long long need_wide = tlsdev::slim_lds_bytes((int)n, (int)M, ctx->region_pad, (int)widths.size(), tlsdev::kSlimThreadsWide);
            if (need > 0 && kSlimMinSlots * (size_t)need <= kLdsPerCU) {
                ctx->slim_lds = (size_t)need; ctx->slim_threads = tlsdev::kSlimThreads;
                ctx->slim_blocks = (int)std::min<int64_t>(std::max<int64_t>(n_periods, 1), (int64_t)std::min<size_t>(4, kLdsPerCU / (size_t)need) * ctx->n_cu);
                if (ctx->opt.blocks > 0) ctx->slim_blocks = std::max(1, std::min(ctx->slim_blocks, ctx->opt.blocks));
            } else if (need == 0 && need_wide > 0 && 2 * (size_t)need_wide <= kLdsPerCU) {
                ctx->slim_lds = (size_t)need_wide; ctx->slim_threads = tlsdev::kSlimThreadsWide;
                ctx->slim_blocks = (int)std::min<int64_t>(std::max<int64_t>(n_periods, 1), (int64_t)2 * ctx->n_cu);
                if (ctx->opt.blocks > 0) ctx->slim_blocks = std::max(1, std::min(ctx->slim_blocks, ctx->opt.blocks));
            }
        }
    } else {
        // the folded series lives in a per-workgroup HBM slab; phase 3 stages it through LDS in
        // tiles of `tile_len` window-start positions plus a halo of the widest window
        // sort histogram: one bucket per point while the counters fit the LDS (fewer points per
        // bucket = fewer comparisons in the in-bucket ranking)
        // (as fine as the LDS allows: a NEARLY commensurate period spreads its piles over neighbouring buckets, and fine
        // buckets keep them below the size from which the counting rank is left; what LDS remains behind the counters
        // stages piled-up buckets for the workgroup sort, fold_and_sort)
        // One 1024-thread workgroup per CU with all of its LDS (two 512-thread ones with half each were measured in rounds 3
        // and 4 and lost: more tiles, more halo staged; the switch is gone).
        const size_t lds_budget = kLdsPerCU;
        ctx->nb = (int)std::min<int64_t>(n, (int64_t)((lds_budget - hdr) / 4));
        size_t halo = (size_t)W + (size_t)(tlsdev::kR - 1) * (size_t)std::max(widest_stride, tlsdev::kMaxTiledStride) + 2 * tlsdev::kU + 4;
        const size_t unit = (size_t)tlsdev::kR * tlsdev::kWave;  // tile bounds: multiples of 320
        {
            // Very long series (N beyond ~150 k with the default duration grid): the widest windows are longer
            // than an LDS tile.  Rows wider than half the tile capacity are marked `oversize`: their (few,
            // widely strided) trial positions are listed and evaluated straight from the slab, one window per
            // wavefront, and the tile halo only has to cover the other rows.  The reference has no size
            // limit (core.py:96-188).
            const size_t cap1 = (lds_budget - hdr) / 8 / ((uniform ? 1 : 2));
            if (cap1 < halo + 4 * unit) {
                const size_t halo_cap = cap1 / 2;
                size_t widest_fit = 1;
                int stride_fit = 1;
                for (auto& we : widths) {
                    const size_t need = (size_t)we.width + (size_t)(tlsdev::kR - 1) * (size_t)std::max(we.tiled ? we.xth : 1, tlsdev::kMaxTiledStride) + 2 * tlsdev::kU + 4;
                    if (need > halo_cap) { we.oversize = 1; we.tiled = 0; we.prunable = 0; }
                    else { widest_fit = std::max(widest_fit, (size_t)we.width); if (we.tiled) stride_fit = std::max(stride_fit, we.xth); }
                }
                widest_stride = stride_fit;
                ctx->region_pad = tlsdev::region_pad_for(widest_stride);
                halo = widest_fit + (widest_fit & 1) + (size_t)(tlsdev::kR - 1) * (size_t)std::max(widest_stride, tlsdev::kMaxTiledStride) + 2 * tlsdev::kU + 4;
            }
        }
        // staged per tile: e (or e*w), and w for per-point weights.  The prefix sum takes the samples' place for the
        // predicate pass (or is formed in place from the staged flux: fast mode), the samples follow for the dot products --
        // two stagings per tile, but fewer and larger tiles than with X staged beside the samples (TESS: 2 instead of 3,
        // -8 %; Kepler-size: 7 instead of 82, most of a tile is halo; that variant was dropped in round 6).
        const size_t buffers = uniform ? 1 : 2;
        if ((lds_budget - hdr) / 8 / buffers < halo + unit) return fail(ctx, TLS_E_ARG, "widest transit window does not fit the LDS tile");
        const size_t cap_doubles = (lds_budget - hdr) / 8 / buffers;
        const size_t cap_tile = (cap_doubles - halo) / unit * unit;
        const size_t n_tiles = ((size_t)M + cap_tile - 1) / cap_tile;
        size_t tile = (((size_t)M + n_tiles - 1) / n_tiles + unit - 1) / unit * unit;
        if (tile > cap_tile) tile = cap_tile;
        ctx->tile_len = (int)tile; ctx->tile_halo = (int)halo;
        // Per period the halo only has to cover the widest IN-RANGE window (core.py:148-156): long periods try
        // narrow windows only, so their tiles can be longer (fewer tiles, less of the slab staged twice).  The
        // LDS tile stays `tile + halo` doubles; PeriodRows::pad carries the period's own tile length.
        {
            const size_t staged = tile + halo;
            // (widths ascend and strides never decrease with them: the widest in-range window and the largest stride
            // of a period are those of its last in-range row that is not oversize -- one table over k_hi, one look-up
            // per period instead of a walk over its rows)
            const size_t nw = widths.size();
            std::vector<int> tile_for_khi(nw + 1, 0);
            {
                size_t wmax = 1; int stride_p = 1;
                for (size_t k = 0; k < nw; ++k) {
                    const auto& we = widths[k];
                    if (!we.oversize) {
                        wmax = std::max(wmax, (size_t)we.width);
                        if (we.tiled) stride_p = std::max(stride_p, we.xth);
                    }
                    const size_t halo_p = wmax + (wmax & 1) + (size_t)(tlsdev::kR - 1) * (size_t)std::max(stride_p, tlsdev::kMaxTiledStride) + 2 * tlsdev::kU + 4;
                    if (halo_p >= halo) continue;   // (0: the plan's tile length)
                    const size_t cap_p = (staged - halo_p) / unit * unit;
                    const size_t tiles_p = ((size_t)M + cap_p - 1) / cap_p;
                    size_t tile_p = (((size_t)M + tiles_p - 1) / tiles_p + unit - 1) / unit * unit;
                    if (tile_p > cap_p) tile_p = cap_p;
                    if (tile_p > tile) tile_for_khi[k + 1] = (int)tile_p;
                }
            }
            for (int64_t p = 0; p < n_periods; ++p) {
                tlsdev::PeriodRows& pr = prow[(size_t)p];
                // (the running maxima above start at row 0, the period's at k_lo: the same whenever the period has a row)
                pr.pad = pr.k_hi > pr.k_lo ? tile_for_khi[(size_t)pr.k_hi] : 0;
            }
        }
        ctx->cumsum_round = 2 * tlsdev::kCumsumChunk;
        const size_t cumsum_bytes = 8 * ((size_t)ctx->cumsum_round + 4);
        ctx->lds_bytes = hdr + std::max<size_t>(std::max<size_t>(4 * (size_t)ctx->nb, cumsum_bytes),
                                                buffers * 8 * (tile + halo));
        ctx->threads = 1024;
        if (ctx->opt.threads > 0) ctx->threads = std::max(64, std::min(1024, ctx->opt.threads / 64 * 64));   // developer switch
        ctx->blocks = (int)std::min<int64_t>(std::max<int64_t>(n_periods, 1), (int64_t)ctx->n_cu);
        if (ctx->opt.blocks > 0)   // developer switch: workgroups in flight (memory-system experiments)
            ctx->blocks = std::max(1, std::min(ctx->blocks, ctx->opt.blocks));
        // two-level sort with sequential HBM accesses (fold_and_sort_tiled) when its LDS windows fit
        const size_t sort2_bytes = hdr + (size_t)tlsdev::sort2_lds_bytes((int)n, ctx->threads);
        ctx->sort2 = sort2_bytes <= lds_budget && ctx->opt.sort2 != 0;
        if (ctx->sort2) ctx->lds_bytes = std::max(ctx->lds_bytes, sort2_bytes);
        // Two-role slab kernel: every workgroup folds periods into per-period slabs, then searches
        // (period, tile) items; the periods go through it in batches that hold one slab per period in HBM (as many periods as
        // fit `kSplitSlabBytes`, at least four rounds of workgroups; all of them when the grid is small).
        // WHEN it is used (measured on one MI355X, same box, against the one-workgroup-per-period kernel): it wins where a
        // GPU holds few periods of a long series -- the shard of a multi-GPU job -- because a period is then searched by
        // several workgroups (260 periods of N = 70 128: 0.61 vs 0.79 ms); on a full grid the one-kernel path keeps every
        // CU in a different phase and needs no hand-off (TESS 2.99 vs 3.50 ms, Kepler sample 5.37 vs 5.43 ms; 713 periods of
        // N = 70 128, 2.8 rounds: 1.37 vs 1.63 ms; 308 periods of the TESS-size series: 0.53 ms both ways).  Hence: up to one
        // and a half rounds of periods -> two-role kernel.  TLS_SPLIT=0/1 forces the choice (A/B, tests).
        ctx->split_blocks = ctx->n_cu;
        if (ctx->opt.blocks > 0) ctx->split_blocks = std::max(1, std::min(ctx->split_blocks, ctx->opt.blocks));
        {
            // WHICH launches take it.  The mode of a period never depends on the launch shape (enqueue), so the two roles must
            // be able to run a period in the mode the one-workgroup kernel gives it: fast mode with X formed at tile-staging
            // time and the dot products on X (`split_fast`: uniform weights, no row wider than an LDS tile, an even number of
            // points), or a plan that is exact throughout.  Where that holds, a SHORT launch whose last round of periods would
            // be partly filled -- the share of a rank of a multi-GPU search, a few hundred periods of a long series -- goes
            // through the two roles: 3.6 items per workgroup instead of 1.2 periods, and the launch ends within a tile's work
            // instead of a whole period's.  Measured (round 6, TESS-size series, same box, one-workgroup kernel / two roles):
            // 307 periods 0.583 / 0.479 ms, 411 periods 0.522 / 0.489; but 256 periods (one full round) 0.567 / 0.623 and 512
            // 0.484 / 0.509 -- a launch of whole rounds has no partly filled round to repair and pays the hand-off (slabs read
            // across XCDs, the ready flags) for nothing; a full grid stays with the one-workgroup kernel (every CU in a
            // different phase).  Hence: up to four rounds, and the last one filled to between 1 and 70 %.
            // switch split = 0 / 1 forces the choice (A/B, tests).
            bool any_oversize = false;
            for (const auto& we : widths) any_oversize = any_oversize || we.oversize != 0;
            const bool all_exact = ctx->opt.exact_prefix == 1 || ctx->opt.fast_slab == 0;
            ctx->split_fast = uniform && !any_oversize && (n & 1) == 0 && !all_exact && ctx->opt.x_staged != 0;
            const int64_t last_round = n_periods % (int64_t)ctx->split_blocks;
            const bool short_launch = n_periods <= 4 * (int64_t)ctx->split_blocks && last_round > 0 && 10 * last_round <= 7 * (int64_t)ctx->split_blocks;
            ctx->split = n_periods > 0 && (ctx->opt.split >= 0 ? ctx->opt.split != 0 : (all_exact || ctx->split_fast) && short_launch);
            const size_t slab_bytes = regions * ((region_doubles + 1) & ~(size_t)1) * 8;
            constexpr size_t kSplitSlabBytes = (size_t)12 << 30;
            int64_t batch = std::max<int64_t>((int64_t)(kSplitSlabBytes / slab_bytes), (int64_t)4 * ctx->split_blocks);
            if (ctx->opt.split_batch > 0) batch = ctx->opt.split_batch;
            ctx->split_batch = (int)std::min<int64_t>(std::max<int64_t>(n_periods, 1), batch);
            ctx->host_tile_prefix.assign((size_t)n_periods + 1, 0u);
            ctx->split_max_items = 0;
            if (ctx->split) {
                for (int64_t wk = 0; wk < n_periods; ++wk) {
                    const tlsdev::PeriodRows& pr = prow[(size_t)order[(size_t)wk]];
                    const size_t tl = pr.pad > 0 ? (size_t)pr.pad : tile;      // the kernel's tile length of this period
                    ctx->host_tile_prefix[(size_t)wk + 1] = ctx->host_tile_prefix[(size_t)wk] + (unsigned int)(((size_t)M + tl - 1) / tl);
                }
                for (int64_t lo = 0; lo < n_periods; lo += ctx->split_batch) {
                    const int64_t hi = std::min<int64_t>(n_periods, lo + ctx->split_batch);
                    ctx->split_max_items = std::max<int64_t>(ctx->split_max_items, (int64_t)ctx->host_tile_prefix[(size_t)hi] - (int64_t)ctx->host_tile_prefix[(size_t)lo]);
                }
                TLS_HIP(ctx, ctx->d_partials.reserve(3 * (size_t)ctx->split_max_items + 3));
                // [split_batch] tiles done | [split_batch] fold ready: all zero between launches (the kernel resets them)
                if (ctx->d_tiles_done.cap < 2 * (size_t)ctx->split_batch) {
                    TLS_HIP(ctx, ctx->d_tiles_done.reserve(2 * (size_t)ctx->split_batch));
                    TLS_HIP(ctx, hipMemsetAsync(ctx->d_tiles_done.ptr, 0, ctx->d_tiles_done.cap * sizeof(unsigned int), ctx->stream));
                }
            }
        }
        const size_t scratch_blocks = std::max<size_t>((size_t)ctx->blocks, ctx->split ? (size_t)ctx->split_batch : 0);
        TLS_HIP(ctx, ctx->d_scratch.reserve(scratch_blocks * regions * (region_doubles + 1) + 16));
    }
    // per-width work units of phase 3 (M is fixed for the plan, so these are period independent)
    // and the layout of one workgroup's live-unit lists: every unit of every width has a slot
    size_t list_cap = 0;
    for (auto& we : widths) {
        const int64_t n_pos = (M - we.width) / we.xth + 1;
        const int64_t r = we.tiled ? tlsdev::kR : 1;
        we.n_pos = (int)n_pos;
        we.n_chunks = (int)((n_pos + r - 1) / r);
        we.list_base = (int)list_cap;
        we.inv_d = 1.0 / (double)we.width;
        list_cap += (size_t)we.n_chunks;
    }
    ctx->list_stride = (list_cap + 63) / 64 * 64;
    // three arrays per workgroup: the live units, (pruning) the bound of each, and the units the bound keeps
    TLS_HIP(ctx, ctx->d_lists.reserve((size_t)std::max(std::max(ctx->blocks, ctx->slim_blocks), (!ctx->resident && ctx->split) ? ctx->split_blocks : 0) * 3 * ctx->list_stride));
    if (ctx->slim_blocks > 0) TLS_HIP(ctx, ctx->d_perm.reserve((size_t)std::max(ctx->blocks, ctx->slim_blocks) * (size_t)n));   // (band resolution stashes the order of a period)
    ctx->prune_min_live = ctx->opt.prune_min_live >= 0 ? (long long)ctx->opt.prune_min_live : 256;
    ctx->p2_shift = 4;  // block length of the coarse prefix sum of e^2: at most kP2MaxBlocks blocks
    while ((((size_t)M + ((size_t)1 << ctx->p2_shift) - 1) >> ctx->p2_shift) > (size_t)tlsdev::kP2MaxBlocks) ++ctx->p2_shift;

    ctx->n = (int)n; ctx->W = (int)W; ctx->M = (int)M; ctx->n_periods = (int)n_periods;
    ctx->n_widths = (int)widths.size();
    ctx->uniform_w = uniform; ctx->w0 = w0; ctx->S0 = S0; ctx->depth_min = params->transit_depth_min;
    ctx->host_widths = widths;
    ctx->band_sigma = -1.0; ctx->d_band_now = nullptr;   // (d_band belongs to the previous width table)
    {
        const double sigma = flux_sigma;
        ctx->flux_sigma = sigma;
        const bool scr_ok = screen_admissible(ctx->resident, uniform, ctx->e_abs_max);
        ctx->prune_kernel = uniform && pruning_pays(ctx->opt, widths, sigma, params->transit_depth_min, ctx->resident, scr_ok);
        ctx->screen_kernel = screen_pays(ctx->opt, widths, sigma, params->transit_depth_min, scr_ok);
    }
    ctx->plan_counters = pc;

    // ONE pinned staging buffer, ONE device allocation, ONE asynchronous copy; nothing is waited for here (the
    // staging buffer is reused only after its event)
    std::vector<tlsdev::RowScreen> screens;
    build_screens(widths, q, screens, ctx->opt.no_screen == 1);
    {
        PlanLayout& L = ctx->layout;
        size_t off = 0;
        auto place = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) / 256 * 256; return at; };
        const size_t nn = (size_t)n, np = (size_t)n_periods, nw = widths.size(), nq = q.size();
        L.t = place(nn * 8); L.y = place(nn * 8); L.w = place(uniform ? 0 : nn * 8);
        L.periods = place(np * 8); L.order = place(np * sizeof(int)); L.rows = place(np * sizeof(tlsdev::PeriodRows));
        L.widths = place(nw * sizeof(tlsdev::WidthEntry)); L.screens = place(nw * sizeof(tlsdev::RowScreen));
        L.q = place(nq * 8); L.q2 = place(nq * 8);   // (uniform weights: the fp32 rows of the screen instead of q^2)
        const bool with_g = !ctx->resident || ctx->slim_blocks > 0;   // the difference taps: dot products on X (HBM slab; four-slot kernel)
        L.g = place(with_g ? nq * 8 : 0);
        const bool with_tiles = !ctx->resident && ctx->split;
        L.tile_prefix = place(with_tiles ? (np + 1) * sizeof(unsigned int) : 0);
        L.total = off;
        int rcs = stage_reserve(ctx, L.total);
        if (rcs) return rcs;
        TLS_HIP(ctx, ctx->d_plan.reserve(L.total));
        unsigned char* h = ctx->h_stage;
        std::memcpy(h + L.t, t, nn * 8);
        std::memcpy(h + L.y, y, nn * 8);
        if (!uniform) std::memcpy(h + L.w, w.data(), nn * 8);
        if (np) {
            std::memcpy(h + L.periods, periods, np * 8);
            std::memcpy(h + L.order, order.data(), np * sizeof(int));
            std::memcpy(h + L.rows, prow.data(), np * sizeof(tlsdev::PeriodRows));
        }
        std::memcpy(h + L.widths, widths.data(), nw * sizeof(tlsdev::WidthEntry));
        std::memcpy(h + L.screens, screens.data(), nw * sizeof(tlsdev::RowScreen));
        std::memcpy(h + L.q, q.data(), nq * 8);
        ctx->q_count = (long long)nq;
        if (!uniform) {
            double* q2 = reinterpret_cast<double*>(h + L.q2);
            for (size_t j = 0; j < nq; ++j) q2[j] = q[j] * q[j];
        } else {
            float* q32 = reinterpret_cast<float*>(h + L.q2);   // [nq] the rows | [nq] the rows one element later
            for (size_t j = 0; j < nq; ++j) { q32[j] = (float)q[j]; q32[nq + j] = j ? (float)q[j - 1] : 0.0f; }
        }
        if (with_g) {
            // Difference taps of every row, same offsets: g_0 = -q_0, g_j = q_{j-1} - q_j, g_L = q_{L-1}.  With e_k =
            // X_{k+1} - X_k (X the running sum of e) a window's dot product is  sum_j q_j e_{i+j} = sum_{j<=L} g_j X_{i+j}
            // (summation by parts): the slab variant's fast mode evaluates it on the X a tile already holds in LDS for
            // the depth predicate, instead of staging the tile's samples a second time (tls_search_body.inc.h, x_dot).
            double* gt = reinterpret_cast<double*>(h + L.g);
            std::memset(gt, 0, nq * 8);
            for (const auto& we : widths) {
                const double* qr = q.data() + we.q_offset;
                double* gr = gt + we.q_offset;
                gr[0] = -qr[0];
                for (int j = 1; j < we.q_len; ++j) gr[j] = qr[j - 1] - qr[j];
                gr[we.q_len] = qr[we.q_len - 1];
            }
        }
        if (with_tiles) {
            std::memcpy(h + L.tile_prefix, ctx->host_tile_prefix.data(), (np + 1) * sizeof(unsigned int));
        }
        unsigned char* d = ctx->d_plan.ptr;
        ctx->d_tile_prefix.ptr = reinterpret_cast<unsigned int*>(d + L.tile_prefix);
        ctx->d_t.ptr = reinterpret_cast<double*>(d + L.t); ctx->d_y.ptr = reinterpret_cast<double*>(d + L.y);
        ctx->d_w.ptr = reinterpret_cast<double*>(d + L.w); ctx->d_periods.ptr = reinterpret_cast<double*>(d + L.periods);
        ctx->d_order.ptr = reinterpret_cast<int*>(d + L.order); ctx->d_rows.ptr = reinterpret_cast<tlsdev::PeriodRows*>(d + L.rows);
        ctx->d_widths.ptr = reinterpret_cast<tlsdev::WidthEntry*>(d + L.widths);
        ctx->d_screens.ptr = reinterpret_cast<tlsdev::RowScreen*>(d + L.screens);
        ctx->d_q.ptr = reinterpret_cast<double*>(d + L.q); ctx->d_q2.ptr = reinterpret_cast<double*>(d + L.q2);
        ctx->d_g.ptr = reinterpret_cast<double*>(d + L.g);
        TLS_HIP(ctx, hipMemcpyAsync(d, h, L.total, hipMemcpyHostToDevice, ctx->stream));
        TLS_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
        ctx->stage_pending = true;
        // results [chi2 | row | depth | counters[4]]
        TLS_HIP(ctx, ctx->d_out.reserve(3 * np + 4));
        ctx->d_chi2.ptr = ctx->d_out.ptr; ctx->d_row.ptr = reinterpret_cast<long long*>(ctx->d_out.ptr + np);
        ctx->d_depth.ptr = ctx->d_out.ptr + 2 * np;
        ctx->d_counters.ptr = reinterpret_cast<unsigned long long*>(ctx->d_out.ptr + 3 * np);
    }
    TLS_HIP(ctx, ctx->d_queue.reserve(1));
    if (!ctx->d_squeue.ptr) {   // zero once per context: the kernel rewinds its queue itself
        TLS_HIP(ctx, ctx->d_squeue.reserve(4));   // [0..1] the search (or fold) kernel's queue, [2..3] the split path's search kernel
        TLS_HIP(ctx, hipMemsetAsync(ctx->d_squeue.ptr, 0, 4 * sizeof(unsigned int), ctx->stream));
    }
    key_store(ctx->key, ctx->opt, t, n, periods, n_periods, tmpl, params);
    ctx->prepared = true;
    return TLS_OK;
}

namespace {
// the flux (and weights) of a prepared plan replaced; kWeightsDiffer when the new dy changes the weight structure
int update_flux_impl(tls_ctx* ctx, const double* y, const double* dy) {
    std::vector<double> w;
    bool uniform; double w0, S0;
    double y_abs_max = 0.0, e_abs_max = 0.0;
    weights_from(y, dy, ctx->n, uniform, w0, w, S0, &y_abs_max, &e_abs_max);
    if (uniform != ctx->uniform_w) return kWeightsDiffer;
    ctx->w0 = w0; ctx->S0 = S0; ctx->y_abs_max = y_abs_max; ctx->e_abs_max = e_abs_max;
    {
        const double sigma = flux_scatter(y, ctx->n);
        ctx->flux_sigma = sigma;
        const bool scr_ok = screen_admissible(ctx->resident, uniform, ctx->e_abs_max);
        ctx->prune_kernel = uniform && pruning_pays(ctx->opt, ctx->host_widths, sigma, ctx->depth_min, ctx->resident, scr_ok);
        ctx->screen_kernel = screen_pays(ctx->opt, ctx->host_widths, sigma, ctx->depth_min, scr_ok);
    }
    const PlanLayout& L = ctx->layout;
    const size_t nn = (size_t)ctx->n;
    int rcs = stage_reserve(ctx, L.total);   // (waits for the previous upload out of the staging buffer)
    if (rcs) return rcs;
    std::memcpy(ctx->h_stage + L.y, y, nn * 8);
    TLS_HIP(ctx, hipMemcpyAsync(ctx->d_y.ptr, ctx->h_stage + L.y, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    if (!uniform) {
        std::memcpy(ctx->h_stage + L.w, w.data(), nn * 8);
        TLS_HIP(ctx, hipMemcpyAsync(ctx->d_w.ptr, ctx->h_stage + L.w, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    TLS_HIP(ctx, hipEventRecord(ctx->ev_stage, ctx->stream));
    ctx->stage_pending = true;
    ctx->executed = false;
    return TLS_OK;
}
}  // namespace

int tls_update_flux(tls_ctx* ctx, const double* y, const double* dy) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_update_flux before tls_prepare");
    if (!y || !dy) return fail(ctx, TLS_E_ARG, "null argument");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = update_flux_impl(ctx, y, dy);
    if (rc == kWeightsDiffer)
        return fail(ctx, TLS_E_STATE, "weight structure (uniform / per-point dy) differs from the prepared search");
    return rc;
}

int tls_execute(tls_ctx* ctx, int count_work) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_execute before tls_prepare");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n_periods == 0) { ctx->executed = true; return TLS_OK; }
    return enqueue(ctx, (count_work & 1) != 0, (count_work & 2) != 0);
}

int tls_t0_fit(tls_ctx* ctx, const double* t, const double* y, int64_t n, double period, const double* signal,
               int64_t dur, const double* epochs, int64_t n_epochs, int64_t roll, double* out_residuals) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!t || !y || !signal || !epochs || !out_residuals) return fail(ctx, TLS_E_ARG, "null argument");
    if (n < 3 || n > 50000000 || dur < 1 || dur > n || n_epochs < 0 || roll < 0 || !(period > 0))
        return fail(ctx, TLS_E_ARG, "tls_t0_fit: argument out of range");
    if (n_epochs == 0) return TLS_OK;
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, ctx->d_queue.reserve(1));
    int rc;
    if ((rc = upload(ctx, ctx->d_ft, t, (size_t)n))) return rc;
    if ((rc = upload(ctx, ctx->d_fy, y, (size_t)n))) return rc;
    if ((rc = upload(ctx, ctx->d_fsig, signal, (size_t)dur))) return rc;
    if ((rc = upload(ctx, ctx->d_fep, epochs, (size_t)n_epochs))) return rc;
    TLS_HIP(ctx, ctx->d_fres.reserve((size_t)n_epochs));
    TLS_HIP(ctx, hipMemsetAsync(ctx->d_queue.ptr, 0, sizeof(unsigned int), ctx->stream));
    double t_lo = t[0], t_hi = t[0];
    for (int64_t i = 1; i < n; ++i) { t_lo = std::min(t_lo, t[i]); t_hi = std::max(t_hi, t[i]); }
    if ((rc = launch_t0_fit(ctx, ctx->d_ft.ptr, ctx->d_fy.ptr, ctx->d_fsig.ptr, ctx->d_fep.ptr, ctx->d_fres.ptr, ctx->d_queue.ptr,
                            n, period, dur, n_epochs, roll, t_lo, t_hi))) return rc;
    TLS_HIP(ctx, hipMemcpyAsync(out_residuals, ctx->d_fres.ptr, (size_t)n_epochs * 8, hipMemcpyDeviceToHost, ctx->stream));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TLS_OK;
}

int tls_pink_noise(tls_ctx* ctx, const double* data, int64_t n, int64_t width, double root_width, double* out) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!data || !out) return fail(ctx, TLS_E_ARG, "null argument");
    if (n < 1 || n > 100000000 || width < 1 || width > n) return fail(ctx, TLS_E_ARG, "pink noise: 1 <= width <= n wanted");
    if (!(root_width > 0.0)) return fail(ctx, TLS_E_ARG, "pink noise: root_width must be width ** 0.5");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n_windows = n - width + 1;
    // data | terms | running sums (n_windows + 1)
    TLS_HIP(ctx, ctx->d_pink.reserve((size_t)n + 2 * (size_t)n_windows + 1));
    double* d_data = ctx->d_pink.ptr; double* d_terms = d_data + n; double* d_sums = d_terms + n_windows;
    TLS_HIP(ctx, hipMemcpyAsync(d_data, data, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(tlsdev::tls_pink_terms, dim3((unsigned)((n_windows + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)d_data, (int)n_windows, (int)width, root_width, d_terms);
    TLS_HIP(ctx, hipGetLastError());
    // (the reference adds the terms one by one from the left: the exact sequential prefix sum of the search, terms >= 0)
    hipLaunchKernelGGL(tlsdev::tls_cumsum_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)d_terms, d_sums, (int)n_windows, 0,
                       static_cast<unsigned long long*>(nullptr));
    TLS_HIP(ctx, hipGetLastError());
    double last = 0.0;
    TLS_HIP(ctx, hipMemcpyAsync(&last, d_sums + n_windows, 8, hipMemcpyDeviceToHost, ctx->stream));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = last / (double)n_windows;
    return TLS_OK;
}

int tls_spectra(tls_ctx* ctx, const double* chi2, int64_t n, int64_t kernel, double* out_SR, double* out_power_raw,
                double* out_power, double* out_sde) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!out_SR || !out_power_raw || !out_power || !out_sde) return fail(ctx, TLS_E_ARG, "null output");
    // ((32 + kernel) doubles of dynamic LDS per median workgroup must stay within the 64 KB a launch gets without asking)
    if (kernel < 1 || kernel > 8000) return fail(ctx, TLS_E_ARG, "median kernel out of range [1, 8000]");
    if (!chi2 && !(ctx->executed && ctx->n_periods > 0)) return fail(ctx, TLS_E_STATE, "tls_spectra without chi2 needs a finished search");
    if (chi2 && (n < 1 || n > 100000000)) return fail(ctx, TLS_E_ARG, "n out of range");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    if (!chi2) n = ctx->n_periods;
    if (kernel % 2 == 0) kernel += 1;                                   // stats.py:115-117
    const size_t nn = (size_t)n;
    TLS_HIP(ctx, ctx->d_spec.reserve(4 * nn + 2));
    double* d_in = ctx->d_chi2.ptr;
    if (chi2) {
        d_in = ctx->d_spec.ptr + 3 * nn + 2;
        TLS_HIP(ctx, hipMemcpyAsync(d_in, chi2, nn * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    tlsdev::SpectraArgs a;
    a.chi2 = d_in; a.SR = ctx->d_spec.ptr; a.power_raw = ctx->d_spec.ptr + nn; a.power = ctx->d_spec.ptr + 2 * nn;
    a.sde = ctx->d_spec.ptr + 3 * nn; a.n = (int)n; a.kernel = (int)kernel; a.detrend = n > 2 * kernel ? 1 : 0;
    a.chi2_stride = 0; a.out_stride = 0; a.sde_stride = 0;
    hipLaunchKernelGGL(tlsdev::tls_spectra_head, dim3(1), dim3(1024), 0, ctx->stream, a);
    if (a.detrend) {
        const int n_med = (int)(n - kernel + 1), threads = 256, per = tlsdev::kMedianWindows;
        const size_t lds = (size_t)(2 * per + kernel) * 8;
        hipLaunchKernelGGL(tlsdev::tls_spectra_median, dim3((unsigned)((n_med + per - 1) / per)), dim3(threads), lds,
                           ctx->stream, a);
        hipLaunchKernelGGL(tlsdev::tls_spectra_tail, dim3(1), dim3(1024), 0, ctx->stream, a);
    }
    TLS_HIP(ctx, hipGetLastError());
    if (out_power_raw == out_SR + nn && out_power == out_power_raw + nn && out_sde == out_power + nn) {
        // the caller's four outputs are one block, like the device's: one copy instead of four
        TLS_HIP(ctx, hipMemcpyAsync(out_SR, a.SR, (3 * nn + 2) * 8, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        TLS_HIP(ctx, hipMemcpyAsync(out_SR, a.SR, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(out_power_raw, a.power_raw, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(out_power, a.power, nn * 8, hipMemcpyDeviceToHost, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(out_sde, a.sde, 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TLS_OK;
}

int tls_debug_folded(tls_ctx* ctx, double* out, int64_t capacity) {
    if (!ctx || !out) return fail(ctx, TLS_E_ARG, "bad argument");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_debug_folded before tls_prepare");
    const int64_t need = ctx->n_periods * ctx->n;
    if (capacity < need) return fail(ctx, TLS_E_ARG, "tls_debug_folded: out holds fewer than n_periods * n doubles");
    if (need == 0) return TLS_OK;
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<double> d_out;
    TLS_HIP(ctx, d_out.reserve((size_t)need));
    int rc = enqueue(ctx, false, false, d_out.ptr);
    if (rc == TLS_OK) {
        ctx->executed = true;
        hipError_t e = hipMemcpyAsync(out, d_out.ptr, (size_t)need * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, TLS_E_HIP, hipGetErrorString(e));
    }
    if (rc != TLS_OK) (void)hipStreamSynchronize(ctx->stream);
    d_out.release();   // (on every path)
    return rc;
}

int tls_debug_prefix(tls_ctx* ctx, double* out, int64_t capacity, int64_t* row_length) {
    if (!ctx || !row_length) return fail(ctx, TLS_E_ARG, "bad argument");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_debug_prefix before tls_prepare");
    *row_length = ctx->M + 1;
    if (!out) return TLS_OK;   // size query
    const int64_t need = ctx->n_periods * (ctx->M + 1);
    if (capacity < need) return fail(ctx, TLS_E_ARG, "tls_debug_prefix: out holds fewer than n_periods * row_length doubles");
    if (need == 0) return TLS_OK;
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<double> d_out;
    TLS_HIP(ctx, d_out.reserve((size_t)need));
    int rc = enqueue(ctx, false, false, nullptr, d_out.ptr);
    if (rc == TLS_OK) {
        ctx->executed = true;
        hipError_t e = hipMemcpyAsync(out, d_out.ptr, (size_t)need * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, TLS_E_HIP, hipGetErrorString(e));
    }
    if (rc != TLS_OK) (void)hipStreamSynchronize(ctx->stream);
    d_out.release();   // (on every path)
    return rc;
}

int tls_debug_period_cycles(tls_ctx* ctx, uint64_t* cycles, int64_t capacity) {
    if (!ctx || !cycles) return fail(ctx, TLS_E_ARG, "bad argument");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_debug_period_cycles before tls_prepare");
    if (capacity < ctx->n_periods) return fail(ctx, TLS_E_ARG, "tls_debug_period_cycles: out holds fewer than n_periods entries");
    if (ctx->n_periods == 0) return TLS_OK;
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<unsigned long long> d_out;
    TLS_HIP(ctx, d_out.reserve((size_t)ctx->n_periods));
    hipError_t e = hipMemsetAsync(d_out.ptr, 0, (size_t)ctx->n_periods * 8, ctx->stream);
    int rc = e == hipSuccess ? enqueue(ctx, false, false, nullptr, nullptr, d_out.ptr) : fail(ctx, TLS_E_HIP, hipGetErrorString(e));
    if (rc == TLS_OK) {
        e = hipMemcpyAsync(cycles, d_out.ptr, (size_t)ctx->n_periods * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, TLS_E_HIP, hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    d_out.release();
    return rc;
}

int tls_debug_cumsum(tls_ctx* ctx, const double* f, int64_t count, double* out, int threads) {
    if (!ctx || !f || !out || count < 0 || count > 100000000) return fail(ctx, TLS_E_ARG, "bad argument");
    if (threads < 64 || threads > 1024 || threads % 64) return fail(ctx, TLS_E_ARG, "threads must be a multiple of 64 in [64, 1024]");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<double> d_f, d_out;
    TLS_HIP(ctx, d_f.reserve((size_t)count));
    TLS_HIP(ctx, d_out.reserve((size_t)count + 1));
    if (count) TLS_HIP(ctx, hipMemcpyAsync(d_f.ptr, f, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
    const int variant = 0;   // (1: the first version of the routine, kept in the kernel for A/B builds)
    // block / fallback counts of this call land in the phase-clock buffer (tls_debug_phase_cycles slots 10, 11)
    TLS_HIP(ctx, ctx->d_phase.reserve(tlsdev::kPhases));
    TLS_HIP(ctx, hipMemsetAsync(ctx->d_phase.ptr, 0, tlsdev::kPhases * sizeof(unsigned long long), ctx->stream));
    { const unsigned long long big = ~0ull; TLS_HIP(ctx, hipMemcpyAsync(ctx->d_phase.ptr + 22, &big, 8, hipMemcpyHostToDevice, ctx->stream)); TLS_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
    hipLaunchKernelGGL(tlsdev::tls_cumsum_kernel, dim3(1), dim3((unsigned)threads), 0, ctx->stream, d_f.ptr, d_out.ptr, (int)count, variant, ctx->d_phase.ptr);
    TLS_HIP(ctx, hipGetLastError());
    TLS_HIP(ctx, hipMemcpyAsync(out, d_out.ptr, ((size_t)count + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    d_f.release(); d_out.release();
    return TLS_OK;
}

int tls_debug_phase_cycles(tls_ctx* ctx, uint64_t* cycles, int n) {
    if (!ctx || !cycles || n < 1) return fail(ctx, TLS_E_ARG, "bad argument");
    if (!ctx->d_phase.ptr) return fail(ctx, TLS_E_STATE, "no execute with the phase clock (count_work & 2)");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long host[tlsdev::kPhases];
    TLS_HIP(ctx, hipMemcpy(host, ctx->d_phase.ptr, sizeof host, hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < tlsdev::kPhases; ++i) cycles[i] = host[i];
    return TLS_OK;
}

namespace {
// test entry: every byte of every CU's LDS set to a pattern (one workgroup with all 160 KB per CU, a few rounds of them)
__global__ void __launch_bounds__(1024) tls_poison_lds_kernel(unsigned int word, unsigned int* sink) {
    extern __shared__ unsigned int lds_words[];
    const unsigned int total = 160u * 1024u / 4u;
    for (unsigned int k = threadIdx.x; k < total; k += blockDim.x) lds_words[k] = word;
    tlsdev::wg_sync();
    if (lds_words[(threadIdx.x * 97u) % total] != word && sink) sink[0] = 1u;   // (keeps the stores alive)
}
}  // namespace

int tls_debug_poison_lds(tls_ctx* ctx, uint32_t word) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tls_poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(tls_poison_lds_kernel, dim3((unsigned)(4 * ctx->n_cu)), dim3(1024), 160 * 1024, ctx->stream, (unsigned int)word,
                       static_cast<unsigned int*>(nullptr));
    TLS_HIP(ctx, hipGetLastError());
    // ... and the per-workgroup scratch in HBM (slabs, live-unit lists, stashed orders, the screen's and the sorts' scratch, the
    // T0 fit's slabs): all ones -- NaNs as doubles -- as fresh device memory may be.  (Not the queues and counters, whose zero
    // state between launches is the kernels' own invariant, nor the plan and the results.)
    auto smear = [&](void* ptr, size_t bytes) -> hipError_t { return ptr && bytes ? hipMemsetAsync(ptr, 0xFF, bytes, ctx->stream) : hipSuccess; };
    TLS_HIP(ctx, smear(ctx->d_scratch.ptr, ctx->d_scratch.cap * sizeof(double)));
    TLS_HIP(ctx, smear(ctx->d_lists.ptr, ctx->d_lists.cap * sizeof(unsigned int)));
    TLS_HIP(ctx, smear(ctx->d_perm.ptr, ctx->d_perm.cap * sizeof(unsigned int)));
    TLS_HIP(ctx, smear(ctx->d_split.ptr, ctx->d_split.cap * sizeof(float)));
    TLS_HIP(ctx, smear(ctx->d_park.ptr, ctx->d_park.cap * sizeof(double)));
    TLS_HIP(ctx, smear(ctx->d_fscratch.ptr, ctx->d_fscratch.cap * sizeof(double)));
    TLS_HIP(ctx, smear(ctx->d_frot.ptr, ctx->d_frot.cap * sizeof(double)));
    TLS_HIP(ctx, smear(ctx->d_frperm.ptr, ctx->d_frperm.cap * sizeof(int)));
    return TLS_OK;
}

int tls_debug_batch_group_ms(const tls_ctx* ctx, double* out, int64_t capacity) {
    if (!ctx) return TLS_E_ARG;
    const int64_t n = (int64_t)ctx->batch_group_ms.size();
    if (out) {
        for (int64_t i = 0; i < std::min(n, capacity); ++i) out[i] = ctx->batch_group_ms[(size_t)i];
        // (capacity for twice the groups: the second half is the part of each group's time spent in its wait for the device)
        const int64_t nw = (int64_t)ctx->batch_group_wait_ms.size();
        for (int64_t i = 0; i < nw && n + i < capacity; ++i) out[n + i] = ctx->batch_group_wait_ms[(size_t)i];
    }
    return (int)std::min<int64_t>(n, 0x7fffffff);
}

int tls_debug_check_counts(tls_ctx* ctx, uint64_t* counts, int n) {
    if (!ctx || !counts || n < 1) return fail(ctx, TLS_E_ARG, "bad argument");
    for (int i = 0; i < n; ++i) counts[i] = 0;
#ifdef TLS_DEBUG_CHECKS
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_check.ptr) {
        unsigned long long host[tlsdev::kChecks];
        TLS_HIP(ctx, hipMemcpy(host, ctx->d_check.ptr, sizeof host, hipMemcpyDeviceToHost));
        for (int i = 0; i < n && i < tlsdev::kChecks; ++i) counts[i] = host[i];
    }
    return 1;   // a checked build
#else
    return TLS_OK;   // not a checked build: all zero
#endif
}

int tls_synchronize(tls_ctx* ctx) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TLS_OK;
}

int tls_execute_timed(tls_ctx* ctx, int reps, double* ms_per_execute) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->prepared) return fail(ctx, TLS_E_STATE, "tls_execute_timed before tls_prepare");
    if (reps < 1 || !ms_per_execute) return fail(ctx, TLS_E_ARG, "reps must be >= 1");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->n_periods == 0) { *ms_per_execute = 0; return TLS_OK; }
    TLS_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r) {
        int rc = enqueue(ctx, false);
        if (rc) return rc;
    }
    TLS_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    TLS_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    TLS_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ms_per_execute = (double)ms / reps;
    return TLS_OK;
}

int tls_fetch(tls_ctx* ctx, double* out_chi2, int64_t* out_row, double* out_depth, tls_counters* counters) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->executed) return fail(ctx, TLS_E_STATE, "tls_fetch before tls_execute");
    if (!out_chi2 || !out_row || !out_depth) return fail(ctx, TLS_E_ARG, "null output");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)ctx->n_periods;
    static_assert(sizeof(long long) == sizeof(int64_t), "int64 layout");
    unsigned long long dev_counts[3] = {0, 0, 0};
    if (np) {
        // [chi2 | row | depth | counters] leave the device as ONE copy into pinned memory
        const size_t words = 3 * np + 4;
        if (ctx->h_out_cap < words) {
            if (ctx->h_out) TLS_HIP(ctx, hipHostFree(ctx->h_out));
            ctx->h_out = nullptr; ctx->h_out_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_out), (words + words / 4) * 8, hipHostMallocDefault));
            ctx->h_out_cap = words + words / 4;
        }
        TLS_HIP(ctx, hipMemcpyAsync(ctx->h_out, ctx->d_out.ptr, words * 8, hipMemcpyDeviceToHost, ctx->stream));
        TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(out_chi2, ctx->h_out, np * 8);
        std::memcpy(out_row, ctx->h_out + np, np * 8);
        std::memcpy(out_depth, ctx->h_out + 2 * np, np * 8);
        std::memcpy(dev_counts, ctx->h_out + 3 * np, sizeof dev_counts);
    } else {
        TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (counters) {
        *counters = ctx->plan_counters;
        counters->evaluated_cells = ctx->counted ? (int64_t)dev_counts[0] : -1;
        counters->inner_steps = ctx->counted ? (int64_t)dev_counts[1] : -1;
        counters->issued_fma = ctx->counted ? (int64_t)dev_counts[2] : -1;
    }
    return TLS_OK;
}

int tls_kernel_timing(tls_ctx* ctx, int reset, double* total_ms, int64_t* launches) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0;
    const size_t n_timed = std::min(ctx->ev_used, ctx->ev_pool.size());
    for (size_t i = 0; i < n_timed; ++i) {
        float ms = 0;
        TLS_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = (int64_t)n_timed;
    if (reset) ctx->ev_used = 0;
    return TLS_OK;
}

const char* tls_last_kernel(const tls_ctx* ctx) { return ctx ? ctx->last_kernel : ""; }

int tls_plan_info(const tls_ctx* ctx, tls_counters* counters, int64_t* lds_bytes, int64_t* n_blocks, int64_t* resident) {
    if (!ctx || !ctx->prepared) return TLS_E_STATE;
    if (counters) { *counters = ctx->plan_counters; counters->evaluated_cells = -1; counters->inner_steps = -1; counters->issued_fma = -1; }
    // (the launch shape of the kernel a plain search of this plan takes: the four-slot kernel where the plan fits it and
    // neither pruning nor the fp32 screen is the host's choice)
    const bool slim = ctx->slim_blocks > 0 && ctx->uniform_w && !ctx->prune_kernel &&
                      !(ctx->screen_kernel && screen_admissible(ctx->resident, ctx->uniform_w, ctx->e_abs_max));
    if (lds_bytes) *lds_bytes = (int64_t)(slim ? ctx->slim_lds : ctx->lds_bytes);
    if (n_blocks) *n_blocks = slim ? ctx->slim_blocks : ctx->blocks;
    if (resident) *resident = ctx->resident ? 1 : 0;
    return TLS_OK;
}

int tls_search(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n, const double* periods,
               int64_t n_periods, const tls_template* tmpl, const tls_params* params, double* out_chi2,
               int64_t* out_row, double* out_depth, tls_counters* counters) {
    int rc = tls_prepare(ctx, t, y, dy, n, periods, n_periods, tmpl, params);
    if (rc) return rc;
    if ((rc = tls_execute(ctx, counters != nullptr))) return rc;
    return tls_fetch(ctx, out_chi2, out_row, out_depth, counters);
}

int tls_search_batch(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n, int64_t n_curves,
                     const double* periods, int64_t n_periods, const tls_template* tmpl, const tls_params* params,
                     double* out_chi2, int64_t* out_row, double* out_depth) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (n_curves < 0) return fail(ctx, TLS_E_ARG, "negative number of light curves");
    if (n_curves == 0) return TLS_OK;
    if (!y || !dy || !out_chi2 || !out_row || !out_depth) return fail(ctx, TLS_E_ARG, "null argument");
    int rc = tls_prepare(ctx, t, y, dy, n, periods, n_periods, tmpl, params);   // the plan, from the first curve
    if (rc) return rc;
    if (n_periods == 0) return TLS_OK;
    // Curves go to the device in groups: ONE launch searches a whole group, and inside the kernel
    // the fold + sort of a period is done once for all curves of the group (it depends on t only).
    // The groups are pipelined over two slots of device and pinned host buffers: while group g is
    // searched, group g+1 is prepared on the host (weights, S0) and uploaded on a second stream, and the
    // results of group g-1 travel back and are copied into the caller's arrays.
    const int64_t group = 32;
    const size_t np = (size_t)n_periods, nn = (size_t)n;
    const bool uni = ctx->uniform_w;
    if (!ctx->copy_stream) TLS_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    const size_t in_doubles = (size_t)group * nn * (uni ? 1 : 2) + 2 * (size_t)group;
    const size_t out_doubles = 3 * (size_t)group * np;
    for (auto& sl : ctx->slot) {
        if (!sl.ev_in) {
            TLS_HIP(ctx, hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
            TLS_HIP(ctx, hipEventCreateWithFlags(&sl.ev_kernel, hipEventDisableTiming));
            TLS_HIP(ctx, hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
        }
        if (sl.h_in_cap < in_doubles) {
            if (sl.h_in) TLS_HIP(ctx, hipHostFree(sl.h_in));
            sl.h_in = nullptr; sl.h_in_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&sl.h_in), in_doubles * 8, hipHostMallocDefault));
            sl.h_in_cap = in_doubles;
        }
        if (sl.h_out_cap < out_doubles) {
            if (sl.h_out) TLS_HIP(ctx, hipHostFree(sl.h_out));
            sl.h_out = nullptr; sl.h_out_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&sl.h_out), out_doubles * 8, hipHostMallocDefault));
            sl.h_out_cap = out_doubles;
        }
        TLS_HIP(ctx, sl.d_y.reserve((size_t)group * nn));
        if (!uni) TLS_HIP(ctx, sl.d_w.reserve((size_t)group * nn));
        TLS_HIP(ctx, sl.d_S0.reserve((size_t)group));
        TLS_HIP(ctx, sl.d_w0.reserve((size_t)group));
        TLS_HIP(ctx, sl.d_chi2.reserve((size_t)group * np));
        TLS_HIP(ctx, sl.d_row.reserve((size_t)group * np));
        TLS_HIP(ctx, sl.d_depth.reserve((size_t)group * np));
    }
    TLS_HIP(ctx, ctx->d_perm.reserve((size_t)std::max(ctx->blocks, ctx->slim_blocks) * nn));
    const int64_t n_groups = (n_curves + group - 1) / group;
    auto drain = [&](int64_t g) -> int {   // results of group g: wait for its download, copy to the caller's arrays
        auto& sl = ctx->slot[g & 1];
        const int64_t c0 = g * group, gc = std::min(group, n_curves - c0);
        TLS_HIP(ctx, hipEventSynchronize(sl.ev_out));
        const size_t cnt = (size_t)gc * np;
        std::memcpy(out_chi2 + c0 * n_periods, sl.h_out, cnt * 8);
        std::memcpy(out_row + c0 * n_periods, sl.h_out + (size_t)group * np, cnt * 8);
        std::memcpy(out_depth + c0 * n_periods, sl.h_out + 2 * (size_t)group * np, cnt * 8);
        return TLS_OK;
    };
    std::vector<double> w;
    // (every HIP failure inside the pipeline leaves through `run`'s return value: the cleanup below then waits for
    // both streams -- asynchronous copies may still target the pinned slots and the caller's arrays -- and clears
    // the launch overrides)
    ctx->batch_group_ms.assign((size_t)n_groups, 0.0);
    ctx->batch_group_wait_ms.clear();
    auto run = [&]() -> int {
    int rc = TLS_OK;
    for (int64_t g = 0; g < n_groups && rc == TLS_OK; ++g) {
        auto& sl = ctx->slot[g & 1];
        const int64_t c0 = g * group, gc = std::min(group, n_curves - c0);
        const auto group_t0 = std::chrono::steady_clock::now();   // (pipelined: a group's time is its host loop pass, waits for older groups included)
        if (g >= 2 && (rc = drain(g - 2))) break;           // the slot's buffers are free again
        // host side of the group: flux into the pinned staging area, weights, S0 (core.py:127)
        double* h_y = sl.h_in;
        double* h_w = sl.h_in + (size_t)group * nn;
        double* h_S0 = sl.h_in + (size_t)group * nn * (uni ? 1 : 2);
        double* h_w0 = h_S0 + group;
        double sigma_sum = 0.0;
        double group_y_max = 0.0, group_e_max = 0.0;
        for (int64_t c = 0; c < gc; ++c) {
            bool uniform; double w0, S0;
            weights_from(y + (c0 + c) * n, dy + (c0 + c) * n, n, uniform, w0, w, S0, &group_y_max, &group_e_max);
            if (uniform != uni) { rc = fail(ctx, TLS_E_ARG, "light curves of a batch must all have uniform or all have per-point dy"); break; }
            h_S0[c] = S0; h_w0[c] = w0;
            std::memcpy(h_y + (size_t)c * nn, y + (c0 + c) * n, nn * 8);
            if (!uniform) std::memcpy(h_w + (size_t)c * nn, w.data(), nn * 8);
            sigma_sum += flux_scatter(y + (c0 + c) * n, n);
        }
        if (rc) break;
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_y.ptr, h_y, (size_t)gc * nn * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        if (!uni) TLS_HIP(ctx, hipMemcpyAsync(sl.d_w.ptr, h_w, (size_t)gc * nn * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_S0.ptr, h_S0, (size_t)gc * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_w0.ptr, h_w0, (size_t)gc * 8, hipMemcpyHostToDevice, ctx->copy_stream));
        TLS_HIP(ctx, hipEventRecord(sl.ev_in, ctx->copy_stream));
        TLS_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.ev_in, 0));
        ctx->S0 = h_S0[0]; ctx->w0 = h_w0[0]; ctx->y_abs_max = group_y_max; ctx->e_abs_max = group_e_max;
        {
            ctx->flux_sigma = sigma_sum / (double)gc;
            const bool scr_ok = screen_admissible(ctx->resident, uni, ctx->e_abs_max);
            ctx->prune_kernel = uni && pruning_pays(ctx->opt, ctx->host_widths, sigma_sum / (double)gc, ctx->depth_min, ctx->resident, scr_ok);
            ctx->screen_kernel = screen_pays(ctx->opt, ctx->host_widths, sigma_sum / (double)gc, ctx->depth_min, scr_ok);
        }
        ctx->batch_curves = (int)gc;
        ctx->over_y = sl.d_y.ptr; ctx->over_w = uni ? nullptr : sl.d_w.ptr; ctx->over_S0 = sl.d_S0.ptr; ctx->over_w0 = sl.d_w0.ptr;
        ctx->over_chi2 = sl.d_chi2.ptr; ctx->over_row = sl.d_row.ptr; ctx->over_depth = sl.d_depth.ptr;
        rc = enqueue(ctx, false);
        ctx->batch_curves = 1;
        ctx->over_y = ctx->over_w = ctx->over_S0 = ctx->over_w0 = nullptr;
        ctx->over_chi2 = nullptr; ctx->over_row = nullptr; ctx->over_depth = nullptr;
        if (rc) break;
        TLS_HIP(ctx, hipEventRecord(sl.ev_kernel, ctx->stream));
        TLS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, sl.ev_kernel, 0));
        TLS_HIP(ctx, hipMemcpyAsync(sl.h_out, sl.d_chi2.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.h_out + (size_t)group * np, sl.d_row.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.h_out + 2 * (size_t)group * np, sl.d_depth.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
        TLS_HIP(ctx, hipEventRecord(sl.ev_out, ctx->copy_stream));
        ctx->batch_group_ms[(size_t)g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - group_t0).count();
    }
    if (rc == TLS_OK)
        for (int64_t g = std::max<int64_t>(0, n_groups - 2); g < n_groups && rc == TLS_OK; ++g) rc = drain(g);
    return rc;
    };
    rc = run();
    if (rc != TLS_OK) {
        (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->copy_stream);
        ctx->batch_curves = 1;
        ctx->over_y = ctx->over_w = ctx->over_S0 = ctx->over_w0 = nullptr;
        ctx->over_chi2 = nullptr; ctx->over_row = nullptr; ctx->over_depth = nullptr;
    }
    // the context keeps the plan, but the search ran on the batch slots: a staged execute
    // must be preceded by tls_update_flux or a new tls_prepare
    ctx->executed = false;
    return rc;
}

static int power_batch_impl(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n, int64_t n_curves,
                            const double* periods, int64_t n_periods, const tls_template* tmpl, const tls_params* params,
                            int64_t median_kernel, tls_power_summary* out_summary, double* out_chi2, int64_t* out_row,
                            double* out_depth, double* out_power, double* out_SR, double* out_power_raw);

int tls_power_batch(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n, int64_t n_curves,
                    const double* periods, int64_t n_periods, const tls_template* tmpl, const tls_params* params,
                    int64_t median_kernel, tls_power_summary* out_summary, double* out_chi2, int64_t* out_row,
                    double* out_depth, double* out_power, double* out_SR, double* out_power_raw) {
    const int rc = power_batch_impl(ctx, t, y, dy, n, n_curves, periods, n_periods, tmpl, params, median_kernel, out_summary,
                                    out_chi2, out_row, out_depth, out_power, out_SR, out_power_raw);
    if (ctx && rc != TLS_OK) {
        // EVERY failure leaves through here: nothing is still copying into or out of the pinned staging buffers or the
        // caller's arrays, and the context does not keep pointing at a batch slot
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        ctx->batch_curves = 1;
        ctx->over_y = ctx->over_w = ctx->over_S0 = ctx->over_w0 = nullptr;
        ctx->over_chi2 = nullptr; ctx->over_row = nullptr; ctx->over_depth = nullptr;
        ctx->executed = false;
    }
    return rc;
}

static int power_batch_impl(tls_ctx* ctx, const double* t, const double* y, const double* dy, int64_t n, int64_t n_curves,
                            const double* periods, int64_t n_periods, const tls_template* tmpl, const tls_params* params,
                            int64_t median_kernel, tls_power_summary* out_summary, double* out_chi2, int64_t* out_row,
                            double* out_depth, double* out_power, double* out_SR, double* out_power_raw) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (n_curves < 0) return fail(ctx, TLS_E_ARG, "negative number of light curves");
    if (n_curves == 0) return TLS_OK;
    if (!t || !y || !dy || !periods || !tmpl || !params || !out_summary) return fail(ctx, TLS_E_ARG, "null argument");
    if (n_periods < 1) return fail(ctx, TLS_E_ARG, "tls_power_batch needs at least one period");
    if (median_kernel < 1 || median_kernel > 8000) return fail(ctx, TLS_E_ARG, "median kernel out of range [1, 8000]");
    if ((out_row == nullptr) != (out_chi2 == nullptr) || (out_depth == nullptr) != (out_chi2 == nullptr))
        return fail(ctx, TLS_E_ARG, "out_chi2, out_row and out_depth go together (all or none)");
    int rc = tls_prepare(ctx, t, y, dy, n, periods, n_periods, tmpl, params);   // the plan, from the first curve
    if (rc) return rc;
    int64_t kernel = median_kernel;
    if (kernel % 2 == 0) kernel += 1;                                   // stats.py:115-117
    const int64_t group = std::min<int64_t>(32, n_curves);              // (one light curve: the drop-in power() call)
    const size_t np = (size_t)n_periods, nn = (size_t)n;
    const bool uni = ctx->uniform_w;
    const int detrend = n_periods > 2 * kernel ? 1 : 0;
    double t_min = t[0], t_max = t[0];
    for (int64_t i = 1; i < n; ++i) { t_min = std::min(t_min, t[i]); t_max = std::max(t_max, t[i]); }
    int64_t max_len = 1;
    for (int64_t r = 0; r < tmpl->n_rows; ++r) max_len = std::max(max_len, tmpl->length[r]);
    // device buffers of one group: flux (weights), per-curve constants, search results, spectra, summaries, T0-fit inputs
    auto& sl = ctx->slot[0];
    TLS_HIP(ctx, sl.d_y.reserve((size_t)group * nn));
    if (!uni) TLS_HIP(ctx, sl.d_w.reserve((size_t)group * nn));
    TLS_HIP(ctx, sl.d_S0.reserve((size_t)group));
    TLS_HIP(ctx, sl.d_w0.reserve((size_t)group));
    TLS_HIP(ctx, sl.d_chi2.reserve((size_t)group * np));
    TLS_HIP(ctx, sl.d_row.reserve((size_t)group * np));
    TLS_HIP(ctx, sl.d_depth.reserve((size_t)group * np));
    TLS_HIP(ctx, ctx->d_perm.reserve((size_t)std::max(ctx->blocks, ctx->slim_blocks) * nn));
    const size_t spec_stride = 3 * np;                                  // SR | power_raw | power of one curve
    TLS_HIP(ctx, ctx->d_spec.reserve((size_t)group * spec_stride + 2 * (size_t)group + 8 * (size_t)group + (size_t)group));
    double* d_sde = ctx->d_spec.ptr + (size_t)group * spec_stride;      // [group][2]
    double* d_pick = d_sde + 2 * (size_t)group;                         // [group][8]
    double* d_T0 = d_pick + 8 * (size_t)group;                          // [group]
    const size_t fit_stride = nn;                                       // epochs / residuals of one curve (<= n each)
    TLS_HIP(ctx, ctx->d_fep.reserve((size_t)group * fit_stride));
    TLS_HIP(ctx, ctx->d_fres.reserve((size_t)group * fit_stride));
    // signals | n_epochs (ints) | fit parameters (T0FitParams, 24 B each)
    TLS_HIP(ctx, ctx->d_fsig.reserve((size_t)group * (size_t)max_len + (size_t)group + 3 * (size_t)group));
    int* d_nep = reinterpret_cast<int*>(ctx->d_fsig.ptr + (size_t)group * (size_t)max_len);
    tlsdev::T0FitParams* d_fit = reinterpret_cast<tlsdev::T0FitParams*>(ctx->d_fsig.ptr + (size_t)group * (size_t)max_len + (size_t)group);
    static_assert(sizeof(tlsdev::T0FitParams) == 24, "three doubles of device scratch per fit");
    // pinned staging: flux in; summaries, T0 and (on request) the per-period arrays out.  TWO sets (the device buffers are
    // one: the stream runs the groups in order): while the device works on group g the host forms group g + 1 in the other
    // set and enqueues it, THEN waits for g -- the device never waits for the host between two groups (round 6)
    const size_t in_doubles = (size_t)group * nn * (uni ? 1 : 2) + 2 * (size_t)group;
    const size_t arrays = (out_chi2 ? 3 : 0) + (out_power ? 1 : 0) + (out_SR ? 1 : 0) + (out_power_raw ? 1 : 0);
    const size_t out_doubles = 11 * (size_t)group + arrays * (size_t)group * np;
    for (auto& hs : ctx->slot) {
        if (!hs.ev_out) {
            TLS_HIP(ctx, hipEventCreateWithFlags(&hs.ev_in, hipEventDisableTiming));
            TLS_HIP(ctx, hipEventCreateWithFlags(&hs.ev_kernel, hipEventDisableTiming));
            TLS_HIP(ctx, hipEventCreateWithFlags(&hs.ev_out, hipEventDisableTiming));
        }
        if (hs.h_in_cap < in_doubles) {
            if (hs.h_in) TLS_HIP(ctx, hipHostFree(hs.h_in));
            hs.h_in = nullptr; hs.h_in_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&hs.h_in), in_doubles * 8, hipHostMallocDefault));
            hs.h_in_cap = in_doubles;
        }
        if (hs.h_out_cap < out_doubles) {
            if (hs.h_out) TLS_HIP(ctx, hipHostFree(hs.h_out));
            hs.h_out = nullptr; hs.h_out_cap = 0;
            TLS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&hs.h_out), out_doubles * 8, hipHostMallocDefault));
            hs.h_out_cap = out_doubles;
        }
    }
    std::vector<double> w;
    rc = TLS_OK;
    const int64_t n_groups = (n_curves + group - 1) / group;
    ctx->batch_group_ms.assign((size_t)n_groups, 0.0);
    ctx->batch_group_wait_ms.assign((size_t)n_groups, 0.0);
    // developer aid (TLS_AMD_STALL_DIAG set): per-period shader cycles of every group's search launch; a group whose wait
    // exceeds a second prints where the cycles went (tools/gpu_stall_probe2.py).  (Not pipelined: a group is waited for at once.)
    const bool diag = std::getenv("TLS_AMD_STALL_DIAG") != nullptr;
    DevBuf<unsigned long long> d_diag;
    if (diag) TLS_HIP(ctx, d_diag.reserve(np));
    struct GroupState { double sigma_sum, y_max, e_max; };
    GroupState gs[2];
    // ---- host side of group g: flux into the pinned staging area, weights, S0 (core.py:127)
    auto prepare_group = [&](int64_t g) -> int {
        auto& hs = ctx->slot[g & 1];
        const int64_t c0 = g * group, gc = std::min(group, n_curves - c0);
        double* h_y = hs.h_in;
        double* h_w = hs.h_in + (size_t)group * nn;
        double* h_S0 = hs.h_in + (size_t)group * nn * (uni ? 1 : 2);
        double* h_w0 = h_S0 + group;
        GroupState& st = gs[g & 1];
        st.sigma_sum = 0.0; st.y_max = 0.0; st.e_max = 0.0;
        for (int64_t c = 0; c < gc; ++c) {
            bool uniform; double w0, S0;
            weights_from(y + (c0 + c) * n, dy + (c0 + c) * n, n, uniform, w0, w, S0, &st.y_max, &st.e_max);
            if (uniform != uni) return fail(ctx, TLS_E_ARG, "light curves of a batch must all have uniform or all have per-point dy");
            h_S0[c] = S0; h_w0[c] = w0;
            std::memcpy(h_y + (size_t)c * nn, y + (c0 + c) * n, nn * 8);
            if (!uniform) std::memcpy(h_w + (size_t)c * nn, w.data(), nn * 8);
            st.sigma_sum += flux_scatter(y + (c0 + c) * n, n);
        }
        return TLS_OK;
    };
    // host layout of a group's results (the same in both sets)
    struct OutLayout { double *sde, *pick, *T0, *chi2, *power, *SR, *praw, *spec3; };
    auto out_layout = [&](int64_t g) -> OutLayout {
        OutLayout o{};
        double* base = ctx->slot[g & 1].h_out;
        o.sde = base; o.pick = o.sde + 2 * (size_t)group; o.T0 = o.pick + 8 * (size_t)group;
        double* h_next = o.T0 + group;                   // chi2 | row | depth | power | SR | power_raw, on request
        if (out_chi2) { o.chi2 = h_next; h_next += 3 * (size_t)group * np; }
        if (out_power && out_SR && out_power_raw) { o.spec3 = h_next; h_next += 3 * (size_t)group * np; }
        else {
            if (out_power) { o.power = h_next; h_next += (size_t)group * np; }
            if (out_SR) { o.SR = h_next; h_next += (size_t)group * np; }
            if (out_power_raw) { o.praw = h_next; h_next += (size_t)group * np; }
        }
        return o;
    };
    // ---- device side of group g, nothing waited for: flux up, search (tls_search_batch's launch: fold + sort shared by the
    // group), spectra, pick, final T0 fit, results down, an event behind them
    auto enqueue_group = [&](int64_t g) -> int {
        auto& hs = ctx->slot[g & 1];
        const int64_t c0 = g * group, gc = std::min(group, n_curves - c0);
        (void)c0;
        const GroupState& st = gs[g & 1];
        double* h_y = hs.h_in;
        double* h_w = hs.h_in + (size_t)group * nn;
        double* h_S0 = hs.h_in + (size_t)group * nn * (uni ? 1 : 2);
        double* h_w0 = h_S0 + group;
        if (diag) TLS_HIP(ctx, hipMemsetAsync(d_diag.ptr, 0, np * 8, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_y.ptr, h_y, (size_t)gc * nn * 8, hipMemcpyHostToDevice, ctx->stream));
        if (!uni) TLS_HIP(ctx, hipMemcpyAsync(sl.d_w.ptr, h_w, (size_t)gc * nn * 8, hipMemcpyHostToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_S0.ptr, h_S0, (size_t)gc * 8, hipMemcpyHostToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(sl.d_w0.ptr, h_w0, (size_t)gc * 8, hipMemcpyHostToDevice, ctx->stream));
        ctx->S0 = h_S0[0]; ctx->w0 = h_w0[0]; ctx->y_abs_max = st.y_max; ctx->e_abs_max = st.e_max;
        {
            ctx->flux_sigma = st.sigma_sum / (double)gc;
            const bool scr_ok = screen_admissible(ctx->resident, uni, ctx->e_abs_max);
            ctx->prune_kernel = uni && pruning_pays(ctx->opt, ctx->host_widths, st.sigma_sum / (double)gc, ctx->depth_min, ctx->resident, scr_ok);
            ctx->screen_kernel = screen_pays(ctx->opt, ctx->host_widths, st.sigma_sum / (double)gc, ctx->depth_min, scr_ok);
        }
        ctx->batch_curves = (int)gc;
        ctx->over_y = sl.d_y.ptr; ctx->over_w = uni ? nullptr : sl.d_w.ptr; ctx->over_S0 = sl.d_S0.ptr; ctx->over_w0 = sl.d_w0.ptr;
        ctx->over_chi2 = sl.d_chi2.ptr; ctx->over_row = sl.d_row.ptr; ctx->over_depth = sl.d_depth.ptr;
        int rc2 = enqueue(ctx, false, diag, nullptr, nullptr, diag ? d_diag.ptr : nullptr);
        ctx->batch_curves = 1;
        ctx->over_y = ctx->over_w = ctx->over_S0 = ctx->over_w0 = nullptr;
        ctx->over_chi2 = nullptr; ctx->over_row = nullptr; ctx->over_depth = nullptr;
        if (rc2) return rc2;
        // ---- spectra of every curve of the group (stats.py:105-132), then what main.py:198-212,269-272 read off them
        tlsdev::SpectraArgs sa;
        sa.chi2 = sl.d_chi2.ptr; sa.SR = ctx->d_spec.ptr; sa.power_raw = ctx->d_spec.ptr + np; sa.power = ctx->d_spec.ptr + 2 * np;
        sa.sde = d_sde; sa.n = (int)n_periods; sa.kernel = (int)kernel; sa.detrend = detrend;
        sa.chi2_stride = (long long)np; sa.out_stride = (long long)spec_stride; sa.sde_stride = 2;
        hipLaunchKernelGGL(tlsdev::tls_spectra_head, dim3(1, (unsigned)gc), dim3(1024), 0, ctx->stream, sa);
        if (detrend) {
            const int n_med = (int)(n_periods - kernel + 1), per = tlsdev::kMedianWindows;
            const size_t lds = (size_t)(2 * per + kernel) * 8;
            hipLaunchKernelGGL(tlsdev::tls_spectra_median, dim3((unsigned)((n_med + per - 1) / per), (unsigned)gc), dim3(256), lds,
                               ctx->stream, sa);
            hipLaunchKernelGGL(tlsdev::tls_spectra_tail, dim3(1, (unsigned)gc), dim3(1024), 0, ctx->stream, sa);
        }
        tlsdev::PickArgs pa;
        pa.chi2 = sl.d_chi2.ptr; pa.row = sl.d_row.ptr; pa.depth = sl.d_depth.ptr; pa.power = ctx->d_spec.ptr + 2 * np;
        pa.periods = ctx->d_periods.ptr; pa.out = d_pick; pa.power_stride = (long long)spec_stride; pa.n = (int)n_periods;
        hipLaunchKernelGGL(tlsdev::tls_power_pick, dim3((unsigned)gc), dim3(1024), 0, ctx->stream, pa);
        // ---- final T0 fit of every curve (stats.py:135-204), WITHOUT a host round trip (round 6): trial epochs, the depth-scaled
        // template and the fit's parameters are formed on the device from the pick (tls_power_prep), all fits of the group run
        // in ONE set of launches (blockIdx.y = light curve), the first minimum is taken on the device
        tlsdev::PrepArgs pr;
        pr.pick = d_pick; pr.widths = ctx->d_widths.ptr; pr.n_widths = ctx->n_widths; pr.q = ctx->d_q.ptr;
        pr.signal = ctx->d_fsig.ptr; pr.signal_stride = (long long)max_len; pr.epochs = ctx->d_fep.ptr; pr.epoch_stride = (long long)fit_stride;
        pr.params = d_fit; pr.n_epochs = d_nep; pr.t_min = t_min; pr.margin = params->T0_fit_margin; pr.n = (int)n;
        hipLaunchKernelGGL(tlsdev::tls_power_prep, dim3((unsigned)gc), dim3(256), 0, ctx->stream, pr);
        TLS_HIP(ctx, hipGetLastError());
        rc2 = launch_t0_fit(ctx, ctx->d_t.ptr, sl.d_y.ptr, ctx->d_fsig.ptr, ctx->d_fep.ptr, ctx->d_fres.ptr, nullptr, n, 1.0, 0, 0, 0,
                            t_min, t_max, d_fit, gc, (int64_t)nn, max_len, (int64_t)fit_stride);
        if (rc2) return rc2;
        tlsdev::FirstMinArgs fa;
        fa.residuals = ctx->d_fres.ptr; fa.epochs = ctx->d_fep.ptr; fa.n_epochs = d_nep;
        fa.T0 = d_T0; fa.stride = (long long)fit_stride;
        hipLaunchKernelGGL(tlsdev::tls_first_min, dim3((unsigned)gc), dim3(1024), 0, ctx->stream, fa);
        TLS_HIP(ctx, hipGetLastError());
        // (sde | pick | T0 lie side by side behind the spectra on the device: ONE copy, the host keeps the layout)
        const OutLayout o = out_layout(g);
        TLS_HIP(ctx, hipMemcpyAsync(o.sde, d_sde, 11 * (size_t)group * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (out_chi2) {
            TLS_HIP(ctx, hipMemcpyAsync(o.chi2, sl.d_chi2.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->stream));
            TLS_HIP(ctx, hipMemcpyAsync(o.chi2 + (size_t)group * np, sl.d_row.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->stream));
            TLS_HIP(ctx, hipMemcpyAsync(o.chi2 + 2 * (size_t)group * np, sl.d_depth.ptr, (size_t)gc * np * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        // (SR | power_raw | power lie side by side per light curve: all three asked for = one contiguous copy, else one strided copy each)
        auto fetch_spec = [&](double* host, size_t which) -> hipError_t {
            return hipMemcpy2DAsync(host, np * 8, ctx->d_spec.ptr + which * np, spec_stride * 8, np * 8, (size_t)gc, hipMemcpyDeviceToHost, ctx->stream);
        };
        if (o.spec3) TLS_HIP(ctx, hipMemcpyAsync(o.spec3, ctx->d_spec.ptr, (size_t)gc * spec_stride * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (o.power) TLS_HIP(ctx, fetch_spec(o.power, 2));
        if (o.SR) TLS_HIP(ctx, fetch_spec(o.SR, 0));
        if (o.praw) TLS_HIP(ctx, fetch_spec(o.praw, 1));
        TLS_HIP(ctx, hipEventRecord(hs.ev_out, ctx->stream));
        return TLS_OK;
    };
    // ---- results of group g: the ONE wait of the group, then the caller's arrays
    auto last_done = std::chrono::steady_clock::now();
    auto consume_group = [&](int64_t g) -> int {
        auto& hs = ctx->slot[g & 1];
        const int64_t c0 = g * group, gc = std::min(group, n_curves - c0);
        const auto wait_t0 = std::chrono::steady_clock::now();
        TLS_HIP(ctx, hipEventSynchronize(hs.ev_out));
        ctx->batch_group_wait_ms[(size_t)g] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wait_t0).count();
        if (diag && ctx->batch_group_wait_ms[(size_t)g] > 1000.0) {
            std::vector<unsigned long long> cyc(np), ph(tlsdev::kPhases);
            TLS_HIP(ctx, hipMemcpy(cyc.data(), d_diag.ptr, np * 8, hipMemcpyDeviceToHost));
            TLS_HIP(ctx, hipMemcpy(ph.data(), ctx->d_phase.ptr, ph.size() * 8, hipMemcpyDeviceToHost));
            std::vector<size_t> idx(np);
            for (size_t i = 0; i < np; ++i) idx[i] = i;
            std::sort(idx.begin(), idx.end(), [&](size_t a_, size_t b_) { return cyc[a_] > cyc[b_]; });
            unsigned long long total = 0;
            for (auto c : cyc) total += c;
            std::fprintf(stderr, "[stall diag] group %lld wait %.1f ms: period cycles total %.4g, median %llu; top:", (long long)g,
                         ctx->batch_group_wait_ms[(size_t)g], (double)total, cyc[idx[np / 2]]);
            for (size_t k = 0; k < std::min<size_t>(8, np); ++k) std::fprintf(stderr, " p%zu(%.6g d)=%.4g", idx[k], periods[idx[k]], (double)cyc[idx[k]]);
            std::fprintf(stderr, "\n[stall diag] phases:");
            for (size_t k = 0; k < ph.size(); ++k) if (ph[k]) std::fprintf(stderr, " %zu:%.4g", k, (double)ph[k]);
            std::fprintf(stderr, "\n");
        }
        const OutLayout o = out_layout(g);
        for (int64_t c = 0; c < gc; ++c) {
            const double* pk = o.pick + 8 * c;
            tls_power_summary& os = out_summary[c0 + c];
            const bool no_fit = pk[6] != 0.0;
            os.chi2_min = pk[0]; os.index_best = (int64_t)pk[1]; os.index_power = (int64_t)pk[2];
            os.best_row = (int64_t)pk[5]; os.no_fit = no_fit ? 1 : 0;
            if (no_fit) {   // main.py:216-267: flat spectra
                os.SDE = 0; os.SDE_raw = 0; os.period = std::nan(""); os.T0 = 0; os.depth = 1;
            } else {
                os.SDE_raw = o.sde[2 * c]; os.SDE = o.sde[2 * c + 1]; os.period = pk[3]; os.depth = pk[4]; os.T0 = o.T0[c];
            }
        }
        if (out_chi2) {
            std::memcpy(out_chi2 + c0 * n_periods, o.chi2, (size_t)gc * np * 8);
            std::memcpy(out_row + c0 * n_periods, o.chi2 + (size_t)group * np, (size_t)gc * np * 8);
            std::memcpy(out_depth + c0 * n_periods, o.chi2 + 2 * (size_t)group * np, (size_t)gc * np * 8);
        }
        if (o.spec3) {
            for (int64_t c = 0; c < gc; ++c) {
                std::memcpy(out_SR + (c0 + c) * n_periods, o.spec3 + (size_t)c * spec_stride, np * 8);
                std::memcpy(out_power_raw + (c0 + c) * n_periods, o.spec3 + (size_t)c * spec_stride + np, np * 8);
                std::memcpy(out_power + (c0 + c) * n_periods, o.spec3 + (size_t)c * spec_stride + 2 * np, np * 8);
            }
        } else {
            if (out_power) std::memcpy(out_power + c0 * n_periods, o.power, (size_t)gc * np * 8);
            if (out_SR) std::memcpy(out_SR + c0 * n_periods, o.SR, (size_t)gc * np * 8);
            if (out_power_raw) std::memcpy(out_power_raw + c0 * n_periods, o.praw, (size_t)gc * np * 8);
        }
        // (pipelined: a group's time is the interval between two groups' results)
        const auto now = std::chrono::steady_clock::now();
        ctx->batch_group_ms[(size_t)g] = std::chrono::duration<double, std::milli>(now - last_done).count();
        last_done = now;
        return TLS_OK;
    };
    rc = prepare_group(0);
    if (rc == TLS_OK) rc = enqueue_group(0);
    for (int64_t g = 0; g < n_groups && rc == TLS_OK; ++g) {
        if (g + 1 < n_groups && !diag) {
            rc = prepare_group(g + 1);
            if (rc == TLS_OK) rc = enqueue_group(g + 1);
            if (rc) break;
        }
        rc = consume_group(g);
        if (rc == TLS_OK && g + 1 < n_groups && diag) {
            rc = prepare_group(g + 1);
            if (rc == TLS_OK) rc = enqueue_group(g + 1);
        }
    }
    ctx->executed = false;   // the search ran on the batch slot, see tls_search_batch
    return rc;
}

int tls_grid_cells(const double* t, int64_t n, const double* periods, int64_t n_periods, const tls_template* tmpl,
                   const tls_params* params, int64_t* cells_per_period) {
    if (!t || !periods || !cells_per_period || n < 3 || n_periods < 0) {
        g_create_error = "tls_grid_cells: invalid argument";
        return TLS_E_ARG;
    }
    std::vector<tlsdev::WidthEntry> widths;
    int rc = build_widths(nullptr, tmpl, params, n, widths, nullptr);
    if (rc) return rc;
    int64_t W = widths.back().width;
    if (W % 2 != 0) W += 1;
    double t_min = t[0], t_max = t[0];
    for (int64_t i = 1; i < n; ++i) { t_min = std::min(t_min, t[i]); t_max = std::max(t_max, t[i]); }
    GridPlan gp;
    if (!plan_periods(widths, params, periods, n_periods, t_max - t_min, n, n + W, nullptr, cells_per_period, &gp)) {
        g_create_error = "tls_grid_cells: periods must be positive and finite";
        return TLS_E_ARG;
    }
    return TLS_OK;
}

int tls_period_costs(const double* t, int64_t n, const double* periods, int64_t n_periods, const tls_template* tmpl,
                     const tls_params* params, double sigma, int64_t* cells_per_period, double* taps_per_period,
                     double* time_per_period, int64_t* workgroups_in_flight, const char* switches) {
    Switches po = process_options();
    if (!switches_parse(po, switches)) {
        g_create_error = "tls_period_costs: malformed switches (expected name=value,... of tls_debug_get_switches)";
        return TLS_E_ARG;
    }
    if (!t || !periods || !cells_per_period || !taps_per_period || n < 3 || n_periods < 0) {
        g_create_error = "tls_period_costs: invalid argument";
        return TLS_E_ARG;
    }
    std::vector<tlsdev::WidthEntry> widths;
    int rc = build_widths(nullptr, tmpl, params, n, widths, nullptr);
    if (rc) return rc;
    int64_t W = widths.back().width;
    if (W % 2 != 0) W += 1;
    const int64_t M = n + W;
    double t_min = t[0], t_max = t[0];
    for (int64_t i = 1; i < n; ++i) { t_min = std::min(t_min, t[i]); t_max = std::max(t_max, t[i]); }
    std::vector<tlsdev::PeriodRows> prow((size_t)n_periods);
    GridPlan gp;
    if (!plan_periods(widths, params, periods, n_periods, t_max - t_min, n, M, prow.data(), cells_per_period, &gp)) {
        g_create_error = "tls_period_costs: periods must be positive and finite";
        return TLS_E_ARG;
    }
    // expected template taps of a row: trial positions x taps x the fraction of windows of white noise whose mean
    // exceeds transit_depth_min (core.py:58): Q(depth_min * sqrt(d) / sigma)
    std::vector<double> prefix(widths.size() + 1, 0.0);
    for (size_t k = 0; k < widths.size(); ++k) {
        const auto& we = widths[k];
        const double n_pos = (double)((M - we.width) / we.xth + 1);
        double frac = 1.0;
        if (sigma > 0) frac = 0.5 * std::erfc(params->transit_depth_min * std::sqrt((double)we.width) / sigma / std::sqrt(2.0));
        prefix[k + 1] = prefix[k] + n_pos * (double)we.q_len * frac;
    }
    for (int64_t p = 0; p < n_periods; ++p)
        taps_per_period[p] = prefix[(size_t)prow[(size_t)p].k_hi] - prefix[(size_t)prow[(size_t)p].k_lo];
    if (time_per_period) {
        // Which kernel variant a search of this light curve runs (as tls_prepare decides it, uniform weights assumed),
        // and that variant's measured cost per period in shader cycles: a0 + aN * n + b * cells + c * taps, fitted to
        // tls_debug_period_cycles on an MI355X (tools/gpu_cost_model.py, profiles/r03_cost_model_fit.json).  Only the
        // ratios matter to the callers (tls_amd/shard.py places block boundaries by the cumulative sum).
        int widest_stride = 1;
        for (const auto& we : widths) if (we.tiled) widest_stride = std::max(widest_stride, we.xth);
        const size_t region_doubles = (size_t)(M + 1 + tlsdev::region_pad_for(widest_stride));
        const size_t hdr = ((size_t)tlsdev::kFixedHeader + 4 * (3 * widths.size() + 2) + 15) / 16 * 16;
        const size_t resident_bytes = hdr + 2 * 8 * region_doubles;
        const bool resident = resident_bytes <= kLdsPerCU && n <= 65535;
        const bool two_per_cu = resident && kLdsPerCU / resident_bytes >= 2;
        const bool prune = pruning_pays(po, widths, sigma, params->transit_depth_min, resident);
        // the four-slot kernel where tls_prepare takes it (a normalised flux admits the fp32 screen: the host's choice between
        // plain and screen is the noise level's)
        const long long slim_need = tlsdev::slim_lds_bytes((int)n, (int)M, tlsdev::region_pad_for(widest_stride), (int)widths.size());
        const bool slim_wanted = po.exact_prefix != 1 && (po.slim == 1 || (po.slim < 0 && po.prune < 0 && po.screen32 < 0));
        const bool slim_base = resident && slim_wanted && po.threads <= 0 && !prune && !screen_pays(po, widths, sigma, params->transit_depth_min, true);
        const long long slim_need_wide = tlsdev::slim_lds_bytes((int)n, (int)M, tlsdev::region_pad_for(widest_stride), (int)widths.size(), tlsdev::kSlimThreadsWide);
        const bool slim_narrow = slim_base && slim_need > 0 && kSlimMinSlots * (size_t)slim_need <= kLdsPerCU;
        const bool slim_wide = slim_base && !slim_narrow && slim_need == 0 && slim_need_wide > 0 && 2 * (size_t)slim_need_wide <= kLdsPerCU;   // (512-thread shape, two to a CU)
        const bool slim = slim_narrow || slim_wide;
        double a0, aN, b, c;
        if (!resident) { a0 = 458384.0; aN = 4.5716; b = 0.4604; c = 0.03275; }        // HBM slab variant (TESS 27 d + Kepler 4 yr)
        else if (slim) { a0 = 59538.0; aN = 0.0; b = 1.553; c = 0.2125; }               // LDS-resident, four 256-thread workgroups per CU (90 d at 50 ppm, round 5)
        else if (prune) { a0 = 116100.0; aN = 0.0; b = 1.906; c = 0.0125; }             // LDS-resident, pruning kernel (90 d at 500 ppm)
        else if (two_per_cu) { a0 = 54603.0; aN = 0.0; b = 0.7823; c = 0.1888; }        // LDS-resident, two 512-thread workgroups per CU (90 d)
        else { a0 = 56564.0; aN = 0.0; b = 0.3189; c = 0.1267; }                        // LDS-resident, one 1024-thread workgroup per CU (100 d)
        for (int64_t p = 0; p < n_periods; ++p)
            time_per_period[p] = a0 + aN * (double)n + b * (double)cells_per_period[p] + c * taps_per_period[p];
        // Series in the HBM slab: the coefficients are exact-mode measurements.  The search runs fast mode (enqueue; whatever
        // the number of periods or ranks: a period's mode depends on the light curve and the period alone): the plain prefix
        // sum saves ~3.5 cycles per point, a period pays an exact prefix pass (kBandHitCost of itself) with the probability that one of its
        // windows hits the undecided band, and a period that expects to hit it starts in exact mode (the same expectation as in
        // enqueue, for a normalised flux).  Without this the block of the longest periods came out a fifth late (PERF_LOG round 4).
        if (!resident && po.fast_slab != 0 && po.exact_prefix != 1 && sigma > 0) {
            const double eps = fast_mode_eps(M, 1.0 + 5.0 * sigma);
            const double band_max = po.band_max >= 0 ? po.band_max : kBandMax;
            std::vector<double> band;
            band_prefix_for(widths, sigma, params->transit_depth_min, eps, band, M);
            for (int64_t p = 0; p < n_periods; ++p) {
                const double lambda = band[(size_t)prow[(size_t)p].k_hi] - band[(size_t)prow[(size_t)p].k_lo];
                if (lambda > band_max) continue;                                  // starts in exact mode
                time_per_period[p] = (time_per_period[p] - 3.5 * (double)n) * (1.0 + kBandHitCost * std::min(1.0, lambda));
            }
        }
        // periods searched side by side on one GPU (one workgroup each): its CUs (256 on an MI355X; the first visible
        // device is asked, a process without one plans for an MI355X), two workgroups per CU when two folded series fit
        // its LDS.  A block of n periods takes ceil(n / this) rounds, not n / this.  (The cycle coefficients above are
        // MI355X measurements; only their ratios matter.)
        if (workgroups_in_flight) *workgroups_in_flight = (slim_narrow ? (int)std::min<size_t>(4, kLdsPerCU / (size_t)slim_need) : slim_wide ? 2 : two_per_cu ? 2 : 1) * visible_compute_units();
    } else if (workgroups_in_flight) {
        *workgroups_in_flight = visible_compute_units();
    }
    return TLS_OK;
}

// ---- RCCL ---------------------------------------------------------------------------
int tls_comm_unique_id(char id_out[128]) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return TLS_E_RCCL; }
    std::memcpy(id_out, &id, 128);
    return TLS_OK;
}

int tls_comm_init(tls_ctx* ctx, int n_ranks, int rank, const char id[128]) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !id) return fail(ctx, TLS_E_ARG, "bad rank layout");
    if (ctx->comm) return fail(ctx, TLS_E_STATE, "communicator already initialised");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, 128);
    TLS_NCCL(ctx, ncclCommInitRank(&ctx->comm, n_ranks, uid, rank));
    ctx->n_ranks = n_ranks; ctx->rank = rank;
    TLS_HIP(ctx, ctx->d_scalar.reserve(2));
    return TLS_OK;
}

int tls_comm_destroy(tls_ctx* ctx) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (ctx->comm) { TLS_NCCL(ctx, ncclCommDestroy(ctx->comm)); ctx->comm = nullptr; }
    ctx->n_ranks = 1; ctx->rank = 0;
    return TLS_OK;
}

int tls_comm_info(tls_ctx* ctx, int* n_ranks, int* rank, int* device) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    int n = 0, r = -1, d = -1;
    if (ctx->comm) {
        TLS_NCCL(ctx, ncclCommCount(ctx->comm, &n));
        TLS_NCCL(ctx, ncclCommUserRank(ctx->comm, &r));
        TLS_NCCL(ctx, ncclCommCuDevice(ctx->comm, &d));
    }
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return TLS_OK;
}

int tls_comm_allgather_device(tls_ctx* ctx, int64_t count_per_rank) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->comm) return fail(ctx, TLS_E_STATE, "tls_comm_init first");
    if (!ctx->executed) return fail(ctx, TLS_E_STATE, "all-gather before tls_execute");
    if (count_per_rank < ctx->n_periods || count_per_rank < 1) return fail(ctx, TLS_E_ARG, "count_per_rank smaller than this rank's shard");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t c = (size_t)count_per_rank, np = (size_t)ctx->n_periods, R = (size_t)ctx->n_ranks;
    // pack [chi2 | row | depth] of this shard, zero padded: 24 B per period
    TLS_HIP(ctx, ctx->d_pack.reserve(3 * c));
    TLS_HIP(ctx, ctx->d_gather.reserve(3 * c * R));
    TLS_HIP(ctx, hipMemsetAsync(ctx->d_pack.ptr, 0, 3 * c * 8, ctx->stream));
    if (np) {
        TLS_HIP(ctx, hipMemcpyAsync(ctx->d_pack.ptr, ctx->d_chi2.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(ctx->d_pack.ptr + c, ctx->d_row.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(ctx->d_pack.ptr + 2 * c, ctx->d_depth.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    TLS_NCCL(ctx, ncclAllGather(ctx->d_pack.ptr, ctx->d_gather.ptr, 3 * c, ncclDouble, ctx->comm, ctx->stream));
    ctx->gathered_count = (int64_t)c;
    return TLS_OK;
}

int tls_comm_fetch_gathered(tls_ctx* ctx, int64_t count_per_rank, double* all_chi2, int64_t* all_row, double* all_depth) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->comm) return fail(ctx, TLS_E_STATE, "tls_comm_init first");
    if (!all_chi2 || !all_row || !all_depth) return fail(ctx, TLS_E_ARG, "null output");
    if (ctx->gathered_count != count_per_rank || count_per_rank < 1)
        return fail(ctx, TLS_E_STATE, "no device-side all-gather of that size to fetch");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t c = (size_t)count_per_rank, R = (size_t)ctx->n_ranks;
    std::vector<double> host(3 * c * R);
    TLS_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_gather.ptr, host.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t r = 0; r < R; ++r) {
        const double* blk = host.data() + r * 3 * c;
        std::memcpy(all_chi2 + r * c, blk, c * 8);
        std::memcpy(all_row + r * c, blk + c, c * 8);  // int64 bit patterns travel as 8-byte words
        std::memcpy(all_depth + r * c, blk + 2 * c, c * 8);
    }
    return TLS_OK;
}

int tls_comm_stage_results(tls_ctx* ctx, int64_t count_per_rank, int64_t slot, int64_t n_slots) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->executed) return fail(ctx, TLS_E_STATE, "staging before tls_execute");
    if (count_per_rank < ctx->n_periods || count_per_rank < 1 || n_slots < 1 || slot < 0 || slot >= n_slots)
        return fail(ctx, TLS_E_ARG, "bad slot layout");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t c = (size_t)count_per_rank, np = (size_t)ctx->n_periods;
    TLS_HIP(ctx, ctx->d_stage.reserve(3 * c * (size_t)n_slots));
    double* dst = ctx->d_stage.ptr + 3 * c * (size_t)slot;   // [chi2 | row | depth] of this slot, 24 B per period
    if (np < c) TLS_HIP(ctx, hipMemsetAsync(dst, 0, 3 * c * 8, ctx->stream));
    if (np) {
        TLS_HIP(ctx, hipMemcpyAsync(dst, ctx->d_chi2.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(dst + c, ctx->d_row.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
        TLS_HIP(ctx, hipMemcpyAsync(dst + 2 * c, ctx->d_depth.ptr, np * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return TLS_OK;
}

int tls_comm_allgather_staged(tls_ctx* ctx, int64_t count_per_rank, int64_t n_slots) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->comm) return fail(ctx, TLS_E_STATE, "tls_comm_init first");
    if (count_per_rank < 1 || n_slots < 1) return fail(ctx, TLS_E_ARG, "bad slot layout");
    const size_t per_rank = 3 * (size_t)count_per_rank * (size_t)n_slots;
    if (ctx->d_stage.cap < per_rank) return fail(ctx, TLS_E_STATE, "nothing staged for that layout");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, ctx->d_gather.reserve(per_rank * (size_t)ctx->n_ranks));
    TLS_NCCL(ctx, ncclAllGather(ctx->d_stage.ptr, ctx->d_gather.ptr, per_rank, ncclDouble, ctx->comm, ctx->stream));
    ctx->gathered_count = -(int64_t)per_rank;   // marks a staged gather (tls_comm_fetch_gathered refuses it)
    return TLS_OK;
}

int tls_comm_fetch_staged(tls_ctx* ctx, int64_t count_per_rank, int64_t n_slots, int64_t slot, double* all_chi2,
                          int64_t* all_row, double* all_depth) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->comm) return fail(ctx, TLS_E_STATE, "tls_comm_init first");
    if (!all_chi2 || !all_row || !all_depth) return fail(ctx, TLS_E_ARG, "null output");
    if (count_per_rank < 1 || n_slots < 1 || slot < 0 || slot >= n_slots) return fail(ctx, TLS_E_ARG, "bad slot layout");
    const size_t c = (size_t)count_per_rank, R = (size_t)ctx->n_ranks, per_rank = 3 * c * (size_t)n_slots;
    if (ctx->gathered_count != -(int64_t)per_rank) return fail(ctx, TLS_E_STATE, "no staged all-gather of that layout to fetch");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<double> host(3 * c);
    for (size_t r = 0; r < R; ++r) {
        TLS_HIP(ctx, hipMemcpyAsync(host.data(), ctx->d_gather.ptr + r * per_rank + 3 * c * (size_t)slot, 3 * c * 8,
                                    hipMemcpyDeviceToHost, ctx->stream));
        TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(all_chi2 + r * c, host.data(), c * 8);
        std::memcpy(all_row + r * c, host.data() + c, c * 8);  // int64 bit patterns travel as 8-byte words
        std::memcpy(all_depth + r * c, host.data() + 2 * c, c * 8);
    }
    return TLS_OK;
}

int tls_comm_allgather_results(tls_ctx* ctx, int64_t count_per_rank, double* all_chi2, int64_t* all_row, double* all_depth) {
    int rc = tls_comm_allgather_device(ctx, count_per_rank);
    if (rc) return rc;
    return tls_comm_fetch_gathered(ctx, count_per_rank, all_chi2, all_row, all_depth);
}

int tls_comm_max(tls_ctx* ctx, double* value_inout) {
    if (!ctx) return fail(nullptr, TLS_E_ARG, "null context");
    if (!ctx->comm) return fail(ctx, TLS_E_STATE, "tls_comm_init first");
    TLS_HIP(ctx, hipSetDevice(ctx->device));
    TLS_HIP(ctx, hipMemcpyAsync(ctx->d_scalar.ptr, value_inout, 8, hipMemcpyHostToDevice, ctx->stream));
    TLS_NCCL(ctx, ncclAllReduce(ctx->d_scalar.ptr, ctx->d_scalar.ptr + 1, 1, ncclDouble, ncclMax, ctx->comm, ctx->stream));
    TLS_HIP(ctx, hipMemcpyAsync(value_inout, ctx->d_scalar.ptr + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
    TLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TLS_OK;
}

int tls_comm_barrier(tls_ctx* ctx) {
    double v = 0;
    return tls_comm_max(ctx, &v);
}

}  // extern "C"
