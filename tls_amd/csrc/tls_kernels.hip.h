// tls_kernels.hip.h -- device code of the transit-least-squares search for gfx950 (CDNA4).
//
// One workgroup searches one trial period end to end (the reference's search_period,
// transitleastsquares/core.py:96-188), fetching periods from a device-side work queue:
//
//   phase 1  fold + stable sort        core.py:119-123   bucket sort on the phase, LDS atomics
//   phase 2  patch + prefix sum        core.py:126-132, helpers.py:70-73 (numpy.cumsum order)
//   phase 3  duration x T0 scan        core.py:162-186 -> core.py:28-76
//   phase 4  argmin reduction          core.py:70-74, 183-188
//
// Data layout.  The phase-folded series lives in LDS for the whole period ("resident"
// variant: 16*(N+W+1) bytes, N = points, W = widest trial transit) or, when it does not fit
// (TESS/Kepler-size N), in a per-workgroup slab of HBM scratch that stays L2/MALL-warm.
//   regA: f[0..M)  folded flux, patched with its first W samples (M = N+W), later e = 1-f
//   regB: C[0..M]  sequential prefix sum of f  (during the sort: bucket counters + indices)
//
// Arithmetic.  Everything is fp64 (chi^2 ~ N with the signal in the 6th digit).  For a trial
// window starting at i with template depth profile q_j = 1 - signal_j and depth scale rs,
// the reference sums  sum_j (f_{i+j} - (1 - q_j rs))^2 w_{i+j}  and adds the out-of-transit
// residual (core.py:67-70).  Expanding the square, the in-window part of sum e^2 w cancels
// against the out-of-transit term and the edge correction, leaving
//       chi2(i) = S0 + rs^2 * A(i) - 2 rs * B(i),
//       A(i) = sum_j q_j^2 w_{i+j},   B(i) = sum_j q_j e_{i+j} w_{i+j},   e = 1 - f,
//       S0 = sum_k (1-y_k)^2 / dy_k^2   (period independent),
// so the hot loop is ONE sliding dot product (two when the weights are not uniform) and no
// per-duration out-of-transit scan is needed.  Differences to the reference's summation
// order are ~1e-13 relative (tolerance: 1e-6).  The depth predicate mean[i] > depth_min
// (core.py:58) is evaluated on the bit-exact reference expression 1 - (C[i+d]-C[i])/d with C
// the sequential cumsum, so the set of evaluated cells is identical.
//
// No MFMA: the contraction is a sliding window with a per-cell scalar, not a GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tlsdev {

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;  // up to 1024 threads per workgroup

// One entry per DISTINCT trial width, ascending (numpy.unique, core.py:113); `row` is the
// first template row with that width (core.py:163-165).
struct WidthEntry {
    int width;        // trial duration d in samples
    int row;          // template row reported for this width
    int q_offset;     // start of q_j = 1 - signal_j in the q array
    int q_len;        // len(signal) (== width in practice)
    int xth;          // T0 stride (core.py:50-55)
    int pad;
    double overshoot; // lc_cache_overview["overshoot"][row]
    double sum_q2;    // sum_j q_j^2 (uniform-weight case: A(i) = w0 * sum_q2)
};

struct SearchArgs {
    const double* t;        // [n]
    const double* y;        // [n]
    const double* w;        // [n] 1/dy^2, or nullptr when all weights equal w0
    const double* periods;  // [n_periods]
    const int* order;       // [n_periods] work order (most expensive first)
    const int* dlo;         // [n_periods] smallest in-range width (samples), core.py:148
    const int* dhi;         // [n_periods] largest in-range width, core.py:149
    const WidthEntry* widths;
    const double* q;        // all q rows back to back
    double* out_chi2;       // [n_periods]
    long long* out_row;     // [n_periods]
    double* out_depth;      // [n_periods]
    unsigned long long* counters;  // [2] evaluated cells, inner steps (nullptr: off)
    unsigned int* queue;    // work-queue head
    double* scratch;        // non-resident: per-workgroup slabs
    long long scratch_stride;  // doubles per slab
    double depth_min;
    double S0;
    double w0;
    int n, W, M;            // points, patch length, n + W
    int n_periods, n_widths, nb;  // nb: sort buckets
};

__device__ __forceinline__ double fold_phase(double t, double period) {
    // core.py:18  time / period - floor(time / period); IEEE division, bit-identical to the CPU
    double x = t / period;
    return x - floor(x);
}

__device__ __forceinline__ int bucket_of(double phase, double nb_d, int nb) {
    int b = (int)(phase * nb_d);  // monotone in phase
    return b < nb - 1 ? b : nb - 1;
}

struct Best {
    double stat;  // rs * (rs*A - 2B), smaller is better; +inf = nothing evaluated
    double td;    // target depth of that cell
    int k;        // index into widths
    int i;        // T0 sample index
};

__device__ __forceinline__ bool better(const Best& a, const Best& b) {
    // strict '<' with first-visited-wins ties: ascending width, then ascending T0
    if (a.stat != b.stat) return a.stat < b.stat;
    if (a.k != b.k) return a.k < b.k;
    return a.i < b.i;
}

__device__ __forceinline__ Best shfl_down_best(const Best& v, int delta) {
    Best o;
    o.stat = __shfl_down(v.stat, delta, kWave);
    o.td = __shfl_down(v.td, delta, kWave);
    o.k = __shfl_down(v.k, delta, kWave);
    o.i = __shfl_down(v.i, delta, kWave);
    return o;
}

// Exclusive prefix sum of cnt[0..nb) in place; `wsum` is LDS scratch of kMaxWaves+1 words.
__device__ inline void block_exclusive_scan(unsigned int* cnt, int nb, unsigned int* wsum) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const int chunk = (nb + nt - 1) / nt;
    const int lo = tid * chunk, hi = min(lo + chunk, nb);
    unsigned int local = 0;
    for (int b = lo; b < hi; ++b) local += cnt[b];
    unsigned int incl = local;  // inclusive scan across the wave
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        unsigned int o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        unsigned int run = 0;
        for (int v = 0; v < nw; ++v) { unsigned int s = wsum[v]; wsum[v] = run; run += s; }
    }
    __syncthreads();
    unsigned int run = wsum[wave] + incl - local;
    for (int b = lo; b < hi; ++b) { unsigned int c = cnt[b]; cnt[b] = run; run += c; }
    __syncthreads();
}

template <bool RESIDENT, bool UNIFORM_W, typename IdxT>
__global__ void __launch_bounds__(1024)
tls_search_kernel(const SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const int n = a.n, W = a.W, M = a.M, nb = a.nb;
    const double nb_d = (double)nb;

    // ---- memory carve-up -----------------------------------------------------------
    // small reduction scratch first (all variants), then the big regions
    unsigned int* wsum = reinterpret_cast<unsigned int*>(smem);            // 32 words
    Best* wbest = reinterpret_cast<Best*>(smem + 128);                      // kMaxWaves * 24 B
    int* s_work = reinterpret_cast<int*>(smem + 128 + kMaxWaves * sizeof(Best));
    constexpr int kHeader = 128 + kMaxWaves * 24 + 16;                      // 528, 16-B multiple
    double *regA, *regB, *regW = nullptr, *regEW = nullptr;
    unsigned int* cnt;
    if constexpr (RESIDENT) {
        regA = reinterpret_cast<double*>(smem + kHeader);
        regB = regA + (M + 1);
        if constexpr (!UNIFORM_W) { regW = regB + (M + 1); }
        cnt = reinterpret_cast<unsigned int*>(regB);
    } else {
        double* slab = a.scratch + (long long)blockIdx.x * a.scratch_stride;
        regA = slab;
        regB = regA + (M + 1);
        if constexpr (!UNIFORM_W) { regW = regB + (M + 1); }
        cnt = reinterpret_cast<unsigned int*>(smem + kHeader);
    }
    (void)regEW;
    // sort scratch inside regB: [cnt (resident only)] idx_tmp[n] perm[n]
    IdxT* idx_tmp = RESIDENT ? reinterpret_cast<IdxT*>(cnt + nb) : reinterpret_cast<IdxT*>(regB);
    IdxT* perm = idx_tmp + n;
    double* ph_orig = regA;  // phase by ORIGINAL index during the sort

    for (;;) {
        // ---- fetch the next period from the queue ----------------------------------
        if (tid == 0) *s_work = (int)atomicAdd(a.queue, 1u);
        __syncthreads();
        const int work = *s_work;
        __syncthreads();
        if (work >= a.n_periods) break;
        const int p = a.order[work];
        const double period = a.periods[p];

        // ---- phase 1: fold + stable sort by phase ----------------------------------
        for (int b = tid; b < nb; b += nt) cnt[b] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            double ph = fold_phase(a.t[i], period);
            ph_orig[i] = ph;
            atomicAdd(&cnt[bucket_of(ph, nb_d, nb)], 1u);
        }
        __syncthreads();
        block_exclusive_scan(cnt, nb, wsum);
        for (int i = tid; i < n; i += nt) {
            int b = bucket_of(ph_orig[i], nb_d, nb);
            unsigned int slot = atomicAdd(&cnt[b], 1u);  // arbitrary order inside a bucket...
            idx_tmp[slot] = (IdxT)i;
        }
        __syncthreads();
        // ...made deterministic here: rank by (phase, original index) inside the bucket.
        // cnt[b] now holds the END of bucket b.
        for (int s = tid; s < n; s += nt) {
            const int i = (int)idx_tmp[s];
            const double ph = ph_orig[i];
            const int b = bucket_of(ph, nb_d, nb);
            const int lo = b ? (int)cnt[b - 1] : 0, hi = (int)cnt[b];
            int rank = 0;
            for (int s2 = lo; s2 < hi; ++s2) {
                const int i2 = (int)idx_tmp[s2];
                const double ph2 = ph_orig[i2];
                rank += (ph2 < ph || (ph2 == ph && i2 < i)) ? 1 : 0;
            }
            perm[lo + rank] = (IdxT)i;
        }
        __syncthreads();
        // gather flux (and weights) in folded order; ph_orig (regA) is dead from here on
        for (int k = tid; k < n; k += nt) {
            const int i = (int)perm[k];
            regA[k] = a.y[i];
            if constexpr (!UNIFORM_W) regW[k] = a.w[i];
        }
        __syncthreads();
        // ---- phase 2: patch (core.py:126-132) and sequential cumsum ----------------
        for (int k = tid; k < W; k += nt) {
            regA[n + k] = regA[k];
            if constexpr (!UNIFORM_W) regW[n + k] = regW[k];
        }
        __syncthreads();
        if (tid == 0) {
            // numpy.cumsum order (helpers.py:72): strictly left to right, so that the depth
            // predicate sees the same bits as the reference.  Batches keep the loads and
            // stores off the dependent-add chain.
            double run = 0.0;
            regB[0] = 0.0;
            int k = 0;
            constexpr int kBatch = 16;
            for (; k + kBatch <= M; k += kBatch) {
                double v[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) v[u] = regA[k + u];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) { run += v[u]; v[u] = run; }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) regB[k + 1 + u] = v[u];
            }
            for (; k < M; ++k) { run += regA[k]; regB[k + 1] = run; }
        }
        __syncthreads();
        // e = 1 - f in place (uniform weights) or e*w (general weights)
        for (int k = tid; k < M; k += nt) {
            double e = 1.0 - regA[k];
            if constexpr (!UNIFORM_W) e *= regW[k];
            regA[k] = e;
        }
        __syncthreads();

        // ---- phase 3: durations x T0 ------------------------------------------------
        Best best;
        best.stat = INFINITY; best.td = 0.0; best.k = 0x7fffffff; best.i = 0x7fffffff;
        const int dlo = a.dlo[p], dhi = a.dhi[p];
        int first_k = -1;
        unsigned long long n_eval = 0, n_steps = 0;
        const double dmin = a.depth_min;
        for (int k = 0; k < a.n_widths; ++k) {
            const WidthEntry we = a.widths[k];
            const int d = we.width;
            if (d < dlo || d > dhi) continue;
            if (first_k < 0) first_k = k;
            const int L = we.q_len, xth = we.xth;
            const double* __restrict__ q = a.q + we.q_offset;
            const double inv_d = 1.0 / (double)d, dd = (double)d;
            const int n_pos = (M - d) / xth + 1;  // i = u*xth <= M-d
            for (int u0 = wave * kWave; u0 < n_pos; u0 += nw * kWave) {
                const int u = u0 + lane;
                const int i = u * xth;
                bool pass = false;
                double dC = 0.0;
                if (u < n_pos) {
                    dC = regB[i + d] - regB[i];
                    const double m_fast = 1.0 - dC * inv_d;  // within 3e-16 of the exact mean
                    if (m_fast > dmin + 1e-15) pass = true;
                    else if (m_fast >= dmin - 1e-15) pass = (1.0 - dC / dd) > dmin;
                }
                if (!__any(pass)) continue;
                if (pass) {
                    const double mean = 1.0 - dC / dd;       // helpers.py:73 + core.py:167
                    const double td = mean * we.overshoot;   // core.py:61
                    const double rs = 2.0 * td;              // 1/(SIGNAL_DEPTH/td), core.py:62-63
                    const double* __restrict__ e = regA + i;
                    double B0 = 0.0, B1 = 0.0, B2 = 0.0, B3 = 0.0;
                    double A0 = 0.0, A1 = 0.0;
                    int j = 0;
                    if constexpr (UNIFORM_W) {
                        for (; j + 4 <= L; j += 4) {
                            B0 = fma(q[j], e[j], B0);
                            B1 = fma(q[j + 1], e[j + 1], B1);
                            B2 = fma(q[j + 2], e[j + 2], B2);
                            B3 = fma(q[j + 3], e[j + 3], B3);
                        }
                        for (; j < L; ++j) B0 = fma(q[j], e[j], B0);
                        const double B = (B0 + B1) + (B2 + B3);
                        const double stat = rs * (rs * we.sum_q2 - 2.0 * B);
                        Best c; c.stat = stat; c.td = td; c.k = k; c.i = i;
                        if (better(c, best)) best = c;
                    } else {
                        const double* __restrict__ wv = regW + i;
                        for (; j + 2 <= L; j += 2) {
                            const double q0 = q[j], q1 = q[j + 1];
                            B0 = fma(q0, e[j], B0);
                            B1 = fma(q1, e[j + 1], B1);
                            A0 = fma(q0 * q0, wv[j], A0);
                            A1 = fma(q1 * q1, wv[j + 1], A1);
                        }
                        for (; j < L; ++j) { B0 = fma(q[j], e[j], B0); A0 = fma(q[j] * q[j], wv[j], A0); }
                        const double stat = rs * (rs * (A0 + A1) - 2.0 * (B0 + B1));
                        Best c; c.stat = stat; c.td = td; c.k = k; c.i = i;
                        if (better(c, best)) best = c;
                    }
                    n_eval += 1; n_steps += (unsigned long long)L;
                }
            }
        }

        // ---- phase 4: argmin over the workgroup --------------------------------------
#pragma unroll
        for (int delta = kWave / 2; delta > 0; delta >>= 1) {
            Best o = shfl_down_best(best, delta);
            if (better(o, best)) best = o;
        }
        if (lane == 0) wbest[wave] = best;
        __syncthreads();
        if (tid == 0) {
            Best g = wbest[0];
            for (int v = 1; v < nw; ++v) if (better(wbest[v], g)) g = wbest[v];
            const double datapoints = (double)n;          // core.py:46 baseline
            double chi2 = INFINITY, depth = 0.0;
            long long row = 0;
            if (first_k >= 0) {
                // uniform weights: A,B were accumulated without the common factor w0
                const double scale = UNIFORM_W ? a.w0 : 1.0;
                const double stat = (g.stat < INFINITY) ? a.S0 + scale * g.stat : INFINITY;
                if (stat < datapoints) {
                    chi2 = stat; row = a.widths[g.k].row; depth = 1.0 - g.td;  // core.py:72-74
                } else {
                    // nothing beat the straight line: first in-range width registers with
                    // chi2 = N and depth 0 (core.py:46-48,183-186)
                    chi2 = datapoints; row = a.widths[first_k].row; depth = 0.0;
                }
            }
            a.out_chi2[p] = chi2;
            a.out_row[p] = row;
            a.out_depth[p] = depth;
        }
        if (a.counters) {
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                n_eval += __shfl_down(n_eval, delta, kWave);
                n_steps += __shfl_down(n_steps, delta, kWave);
            }
            if (lane == 0 && n_eval) {
                atomicAdd(&a.counters[0], n_eval);
                atomicAdd(&a.counters[1], n_steps);
            }
        }
        __syncthreads();
    }
}

}  // namespace tlsdev
