// tls_kernels.hip.h -- device code of the transit-least-squares search for gfx950 (CDNA4).
//
// One workgroup searches one trial period end to end (the reference's search_period,
// transitleastsquares/core.py:96-188), fetching periods from a device-side work queue:
//
//   phase 1  fold + stable sort        core.py:119-123   bucket sort on the phase, LDS atomics
//   phase 2  patch + prefix sum        core.py:126-132, helpers.py:70-73 (numpy.cumsum order)
//   phase 3a depth predicate           core.py:58        -> compacted list of live T0 chunks
//   phase 3b sliding chi^2             core.py:59-74     register-tiled dot products
//   phase 4  argmin reduction          core.py:70-74, 183-188
//
// Data layout.  The phase-folded series lives in LDS for the whole period ("resident"
// variant: 16*(N+W+1+region_pad) bytes, N = points, W = widest trial transit) or, when it does
// not fit (TESS/Kepler-size N), in a per-workgroup slab of HBM scratch that stays L2/MALL-warm
// and is staged through LDS in tiles of window-start positions plus a halo of one window.
//   regA: f[0..M)  folded flux, patched with its first W samples (M = N+W), later e = 1-f
//   regB: C[0..M]  sequential prefix sum of f  (during the sort: bucket counters + indices)
//   regW: w[0..M)  1/dy^2 in folded order (only when the weights are not uniform)
// Each region has region_pad spare entries so that the unrolled dot product may read a few
// samples past a window (they meet the zero padding of the template rows); behind C the spare
// entries are rising sentinels that make every window past the T0 grid fail the predicate.
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
// Mapping to the hardware.  Only ~10 % of the trial cells pass the predicate and they come in
// runs, so phase 3a compacts the live cells of every duration into lists of CHUNKS of kR = 5
// consecutive T0 positions; phase 3b hands 64 chunks of one duration to a wavefront, one chunk
// per lane.  A lane slides the template over its 5 windows at once: every folded sample it
// reads from LDS feeds 5 FMAs (ds_read_b64 at a lane stride of 40 B is bank-conflict free for
// consecutive chunks), and the template value is wave-uniform, fetched with scalar loads into
// SGPRs (the template table is read through the constant address space).  That puts the loop
// on the fp64 FMA pipe instead of the LDS pipe.  Long durations search a strided T0 grid
// (core.py:50-58): strides 2..5 use the same 5-window form with the windows that far apart,
// larger strides a runtime-stride form with one scalar tap stream per window; rows left with a
// handful of live chunks are re-listed and evaluated one window per lane.
//
// Variants (template parameters): RESIDENT / tiled series, UNIFORM_W / per-point weights, WITH_PRUNING (LDS-resident series,
// noisy light curves: an exact branch-and-bound step drops the cells that cannot win before phase 3b, see cell_bound).
// Survey batches (SearchArgs::n_curves > 1): the fold + sort of a period is shared by all light
// curves of the launch; phases 2-4 run per curve.
//
// No MFMA: the contraction is a sliding window with a per-cell scalar, not a GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef TLS_LAUNCH_THREADS
#define TLS_LAUNCH_THREADS 1024   // developer builds: 512 doubles the register budget (launch with TLS_THREADS=512)
#endif
#ifndef TLS_WAVES_PER_EU
#define TLS_WAVES_PER_EU 4
#endif

// Debug build (make -C tls_amd/csrc debug: -DTLS_DEBUG_CHECKS): every hand-computed bound of the kernel --
// LDS carve-up, live-unit list capacity, tile/halo indices of the dot products, sort windows, work-queue
// indices -- is tested on the device; a violation is COUNTED per check (SearchArgs::check, read back by
// tls_debug_check_counts), never trapped, so one run reports every broken bound.
#ifdef TLS_DEBUG_CHECKS
#define TLS_CHECK(a, cond, code) do { if (!(cond) && (a).check) atomicAdd(&(a).check[(code)], 1ull); } while (0)
#else
#define TLS_CHECK(a, cond, code) do { } while (0)
#endif

namespace tlsdev {

constexpr int kChecks = 16;   // slots of SearchArgs::check
enum CheckCode {
    kChkLdsCarve = 0,      // header + regions exceed the dynamic LDS of the launch
    kChkListCap = 1,       // a live-unit list outgrew its row's slots
    kChkDotWindow = 2,     // a dot product reads outside the staged samples
    kChkPredicateRead = 3, // the depth predicate reads C outside [0, M + region_pad]
    kChkSortWindow = 4,    // a sort bin outgrew its LDS window without being caught
    kChkWorkItem = 5,      // period / row index out of range
    kChkSinglesCap = 6,    // re-listed positions outgrew the row's slots
    kChkTileStage = 7,     // a tile was staged beyond the LDS buffer
    kChkSplit = 8,         // fp32 screen: a sample is not the exact sum of its two fp32 halves
};

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;  // up to 1024 threads per workgroup
constexpr int kPhases = 40;    // phase-clock slots 0..31 and work statistics 32..39 (tls_amd/_lib.py names them)
#ifndef TLS_KR
#define TLS_KR 5
#endif
#ifndef TLS_CUMSUM2
#define TLS_CUMSUM2 1   // 1: three-barrier register-resident prefix sum (exact_cumsum); 0: the first version
#endif
#ifndef TLS_PRUNE
#define TLS_PRUNE 1   // exact branch-and-bound pruning of trial cells (uniform weights), see cell_bound()
#endif
constexpr int kR = TLS_KR;          // T0 positions per lane in the sliding dot product (odd: no LDS conflicts)
constexpr int kU = 8;          // template taps per unrolled iteration
#ifndef TLS_SPARSE_ROW
#define TLS_SPARSE_ROW 16
#endif
#ifndef TLS_TAIL_RELIST
#define TLS_TAIL_RELIST 1    // (LDS-resident series) a row's last, mostly empty batch of live chunks runs position by position
#endif
#ifndef TLS_TAIL_MAX
#define TLS_TAIL_MAX 20      // ...when it holds at most this many chunks (and its live positions fit one batch)
#endif
constexpr int kTailMax = TLS_TAIL_MAX;
constexpr int kSparseRow = TLS_SPARSE_ROW;   // rows with at most this many live chunks are re-listed position by position
constexpr int kMaxTiledStride = 5;  // T0 strides up to this have a dot product with compile-time tap offsets
constexpr int kMaxRuntimeStride = 128;  // larger strides (only with a huge T0_fit_margin) go one window per lane
// A row is "tiled" (kR windows per lane share every folded sample) when its stride is small
// against its width: the kR windows then overlap almost completely.
__host__ __device__ constexpr bool row_is_tiled(int width, int xth) {
    return xth <= kMaxTiledStride || (xth <= kMaxRuntimeStride && 8 * xth <= width);
}
// zeros in front of / behind a template row whose windows are `xth` samples apart, and the spare
// entries behind every folded-series region for the widest tiled stride `xs` of the plan
__host__ __device__ constexpr int pad_front(int xth) { return ((kR - 1) * (xth > kMaxTiledStride ? xth : kMaxTiledStride) + 7) / 8 * 8; }
__host__ __device__ constexpr int pad_back(int xth) { return (2 * kU + (kR - 1) * (xth > kMaxTiledStride ? xth : kMaxTiledStride) + 7) / 8 * 8; }
__host__ __device__ constexpr int region_pad_for(int xs) { return (2 * kU + kR * (xs > kMaxTiledStride ? xs : kMaxTiledStride) + 7) / 8 * 8; }
constexpr int kP2MaxBlocks = 236;  // blocks of the coarse prefix sum of e^2 (it lives in the cumsum scratch)
constexpr int kCumsumScratchBytes = 1920;  // >= sizeof(CumsumScratch), 16-B multiple
constexpr int kFixedHeader = 560 + kCumsumScratchBytes;  // wsum[32] | wbest[16] | s_work[12] | cumsum scratch

typedef const __attribute__((address_space(4))) double* const_f64_ptr;  // -> s_load, SGPR operands
struct WidthEntry;
typedef const __attribute__((address_space(4))) WidthEntry* const_width_ptr;
struct PeriodRows;
typedef const __attribute__((address_space(4))) PeriodRows* const_rows_ptr;

// The lanes of the wave for which `pred` holds, as the compare that produced it left them in a scalar register pair
// (HIP's __ballot goes through a 0/1 select and a second compare).
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// v[lane `which`] = value (both wave-uniform).  The lane select travels in M0: before gfx10 one instruction reads one
// scalar register besides it.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (m0 is a reserved register: the compiler never keeps a value of its own in it)
__device__ __forceinline__ void set_lane(int& v, int value, int which) {
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(which) : "m0");
}
#pragma clang diagnostic pop

// kU consecutive doubles from the folded series.  In LDS the reads are issued as eight
// ds_read_b64: hipcc would pair them into ds_read2_b64, which moves half the bytes per LDS
// cycle (MI355X_MICROARCH.md, LDS table: 128 vs 256 B/clk) -- and this loop lives on the LDS.
typedef const __attribute__((address_space(3))) double* lds_f64_ptr;
template <bool IN_LDS>
__device__ __forceinline__ void load_taps(const double* p, double (&x)[kU]) {
    static_assert(kU == 8, "the asm block reads 8 values");
    if constexpr (IN_LDS) {
        const unsigned addr = (unsigned)(uintptr_t)(lds_f64_ptr)p;
        asm volatile(
            "ds_read_b64 %0, %8\n\t"
            "ds_read_b64 %1, %8 offset:8\n\t"
            "ds_read_b64 %2, %8 offset:16\n\t"
            "ds_read_b64 %3, %8 offset:24\n\t"
            "ds_read_b64 %4, %8 offset:32\n\t"
            "ds_read_b64 %5, %8 offset:40\n\t"
            "ds_read_b64 %6, %8 offset:48\n\t"
            "ds_read_b64 %7, %8 offset:56\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
            : "v"(addr) : "memory");
    } else {
#pragma unroll
        for (int u = 0; u < kU; ++u) x[u] = p[u];
    }
}

// Streaming accesses to the per-workgroup HBM slabs (written once, read once or twice by the same CU):
// non-temporal, so that they do not push the light curve (t, y: read at random by every workgroup of
// the XCD) out of the L2.  TLS_NT=0 builds plain accesses for A/B timing.
#ifndef TLS_NT
#define TLS_NT 1
#endif
#ifndef TLS_NT_LOAD
#define TLS_NT_LOAD TLS_NT
#endif
#ifndef TLS_NT_STORE
#define TLS_NT_STORE TLS_NT
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#if TLS_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
#if TLS_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Ordering point for code in which every wavefront works on LDS (and HBM) of its own: the memory
// operations of the wave issued so far are complete before any later one starts.  No other wave
// is waited for.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// The same for LDS traffic only: the wave's LDS operations are complete, its global loads and stores STAY IN
// FLIGHT.  (wave_sync()'s fences compile to s_waitcnt vmcnt(0) lgkmcnt(0): in the slab sort's per-wave bin
// rounds every one of them waited for the prefetch of the next bin and the gather issued a few lines above --
// the whole HBM round trip exposed, six times a round.)
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// A pointer into LDS that reached a real (non-inlined) function as a generic pointer: handing it back
// through its own address space lets the compiler use ds_* instructions again.  Without this every "LDS"
// access of such a function is a FLAT instruction -- through the vector-memory pipe, counted on vmcnt AND
// lgkmcnt, several times slower (found in the ISA of the first non-inlined version of the slab sort).
// A wave-uniform value that reached us in vector registers (arguments of a __noinline__ device function do):
// back into scalar registers, so that addresses formed from it stay scalar base + vector offset.
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(p);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffull));
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T>
__device__ __forceinline__ T* as_lds(T* p) {
    return (T*)(__attribute__((address_space(3))) T*)p;
}
// ... and likewise a pointer into global memory (flat_load/flat_store count on lgkmcnt as well: every LDS wait
// of the function would also wait for its global traffic)
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)p;
}
// Pointer arguments of a __noinline__ function, uniform and in their address space again (the value is made
// scalar INSIDE the address space: an integer round trip of the generic pointer loses it)
template <typename T>
__device__ __forceinline__ T* lds_arg(T* p) {
    typedef __attribute__((address_space(3))) T* lds_t;
    const unsigned int v = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(uintptr_t)(lds_t)p);
    return (T*)(lds_t)(uintptr_t)v;
}
template <typename T>
__device__ __forceinline__ T* global_arg(T* p) {
    typedef __attribute__((address_space(1))) T* glob_t;
    const unsigned long long b = (unsigned long long)(uintptr_t)(glob_t)p;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffull));
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return (T*)(glob_t)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

// The workgroup barrier of every kernel here: __syncthreads() with the wait for the wave's own LDS operations written out.
// hipcc (ROCm 7.2, gfx950) drops the `s_waitcnt lgkmcnt(0)` of __syncthreads()'s release fence in front of some s_barriers
// whose wave still has a ds_write in flight (seen in the ISA of tls_slim_kernel: the store of a row's counts at the end of a
// loop, the barrier behind the loop's exit, no wait between them) -- the waves of a workgroup sit on different SIMDs, each with
// its own queue to the LDS, and a read issued behind the barrier by another wave can overtake that write.  Round 6 met it as
// a stale batch count: one period in ~3e6 spun through 2^26 empty batches, tens of seconds (PERF_LOG "the stalled group").
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Workgroup barrier that orders LDS traffic only: the LDS operations of every wave are complete, global
// loads and stores STAY IN FLIGHT across it.  __syncthreads() also drains the vector-memory counter,
// i.e. every barrier between two LDS steps would expose a full HBM round trip of whatever was
// prefetched or written just before -- on the slab path that was most of the time.  Use it only where
// no thread consumes global data another thread of the workgroup has just written.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// All vector-memory operations of this wave issued so far are complete.  Placed BEFORE a batch of
// stores where only old loads are pending: the compiler's wait-count pass treats a counter with loads
// AND stores pending as out of order and turns the next wait on any load into a full drain, i.e. a
// prefetched value first used behind a batch of stores would wait for those stores.
__device__ __forceinline__ void vmem_wait_all() {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt and lgkmcnt untouched (gfx9 encoding)
}

#ifndef TLS_SLAB_DMA
#define TLS_SLAB_DMA 1
#endif
#ifndef TLS_STRIDED_TICKETS
#define TLS_STRIDED_TICKETS 1   // strided rows of the depth predicate go to the waves through a ticket counter
#endif
#ifndef TLS_FOLD_DEPTH
#define TLS_FOLD_DEPTH 3     // time stamps a thread folds per step
#endif
#ifndef TLS_RANK_BY_INDEX
#define TLS_RANK_BY_INDEX 0  // 1: the ranking pass of the bucket sort walks the points in index order (measured: +-0.5 %)
#endif
#ifndef TLS_GATHER_DEPTH
#define TLS_GATHER_DEPTH 3   // flux values a thread gathers per step (their L2 round trips overlap)
#endif
// Slab -> LDS without registers: `count` doubles (rounded up to a pair) travel by global_load_lds_dwordx4, 1 KiB
// per wave-instruction, all of them in flight together -- a copy through registers keeps 32-64 B per thread
// in flight, which at ~1.5 us of loaded HBM latency is the ~10 B/cycle a workgroup was seen to move.  Element k
// of the destination is src[p < n ? p : p - n], p = p0 + k (the patch mapping of core.py:126; n = INT_MAX for a
// plain copy).  p0 and n even, dst and src 16-byte aligned.  The data is in LDS behind vmem_wait_all() + a
// barrier.  (Inline assembly: see fold_sort_cumsum_tiled for why not the builtin.)
__device__ __forceinline__ void slab_to_lds_async(double* dst_lds, const double* src, int p0, int count, int n, int tid) {
    const int nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const unsigned int dst_addr = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) double*)dst_lds;
    const int pairs = (count + 1) / 2;
    for (int q0 = 0; q0 < pairs; q0 += nt) {
        const int q = q0 + tid;
        const unsigned int dst = (unsigned int)__builtin_amdgcn_readfirstlane((int)(dst_addr + 16u * (unsigned int)(q0 + wave * kWave)));
        if (q < pairs) {
            const int p = p0 + 2 * q;
            const double* g = src + (p < n ? p : p - n);
            unsigned int m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
#if TLS_NT_LOAD
                         " nt"
#endif
                         "\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "v"(g), "s"(dst) : "memory");
        }
    }
}

// dst[k] = src[k] for k in [0, count), all threads of the workgroup, four global reads in flight per
// thread (the compiler does not overlap them itself: the store of one may alias the read of the next)
__device__ __forceinline__ void copy_in_flight4(double* dst, const double* src, int count) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: see fold_and_sort_tiled
    const int nt = blockDim.x;
    for (int k = tid; k < count; k += 4 * nt) {
        const int k1 = k + nt, k2 = k + 2 * nt, k3 = k + 3 * nt;
        const double v0 = stream_load(src + k);     // the source is a slab in HBM
        const double v1 = k1 < count ? stream_load(src + k1) : 0.0;
        const double v2 = k2 < count ? stream_load(src + k2) : 0.0;
        const double v3 = k3 < count ? stream_load(src + k3) : 0.0;
        dst[k] = v0;
        if (k1 < count) dst[k1] = v1;
        if (k2 < count) dst[k2] = v2;
        if (k3 < count) dst[k3] = v3;
    }
}

// Tiled variant: `count` samples of the PATCHED folded series, from position p_lo on, go from the HBM
// slab into an LDS tile as e = 1 - f (or e*w and w).  The slab holds the folded flux f[0..n) (and
// weights) only once: position p in [n, M) is sample p - n again (core.py:126-132), positions from M
// on are zero.  Eight global reads in flight per thread.
template <bool UNIFORM_W>
__device__ __forceinline__ void stage_samples(double* tile_e, double* tile_w, const double* f, const double* w,
                                              int p_lo, int count, int n, int M) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: see fold_and_sort_tiled
    const int nt = blockDim.x;
    constexpr int kInFlight = 8;
    for (int k0 = tid; k0 < count; k0 += kInFlight * nt) {
        double fv[kInFlight], wv[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
            const int k = k0 + j * nt, p = p_lo + k;
            const bool ok = k < count && p < M;
            const int src = p < n ? p : p - n;
            fv[j] = ok ? stream_load(f + src) : 1.0;
            if constexpr (!UNIFORM_W) wv[j] = ok ? stream_load(w + src) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
            const int k = k0 + j * nt;
            if (k < count) {
                double e = 1.0 - fv[j];
                if constexpr (!UNIFORM_W) { e *= wv[j]; tile_w[k] = wv[j]; }
                tile_e[k] = e;
            }
        }
    }
}

// The same in two steps without registers: the raw flux (and weights) travel into the tile by LDS-direct loads
// (stage_samples_async), and once they have landed (vmem_wait_all + barrier) every thread turns its share into
// e = 1 - f (or e*w), zero from position M on (finish_samples).  n even.
template <bool UNIFORM_W>
__device__ __forceinline__ void stage_samples_async(double* tile_e, double* tile_w, const double* f, const double* w,
                                                    int p_lo, int count, int n, int M, int tid) {
    const int have = M - p_lo < count ? (M - p_lo > 0 ? M - p_lo : 0) : count;   // positions below M
    slab_to_lds_async(tile_e, f, p_lo, have, n, tid);
    if constexpr (!UNIFORM_W) slab_to_lds_async(tile_w, w, p_lo, have, n, tid);
}
template <bool UNIFORM_W>
__device__ __forceinline__ void finish_samples(double* tile_e, double* tile_w, int p_lo, int count, int M, int tid) {
    const int nt = blockDim.x;
    for (int k = tid; k < count; k += nt) {
        const bool ok = p_lo + k < M;
        double e = ok ? 1.0 - tile_e[k] : 0.0;
        if constexpr (!UNIFORM_W) { const double ww = ok ? tile_w[k] : 0.0; e *= ww; tile_w[k] = ww; }
        tile_e[k] = e;
    }
}

// developer instrumentation: thread 0 stamps the shader clock at phase boundaries
// Per-phase shader clocks of the search kernel (tls_debug_phase_cycles, slots 0-31): built into the instrumented and the
// checked library (make clocks / make debug), not into the shipped one -- the marks cost 0.4-1.1 % of the kernel even when
// nobody asks for them (the statistics slots 32-39 are kept everywhere).
#ifndef TLS_PHASE_CLOCKS
#define TLS_PHASE_CLOCKS 0
#endif
struct PhaseClock {
    unsigned long long* out;
    long long last;
#if TLS_PHASE_CLOCKS
    __device__ __forceinline__ void start(unsigned long long* o) { out = o; if (out && threadIdx.x == 0) last = clock64(); }
    __device__ __forceinline__ void mark(int phase) {
        if (out && threadIdx.x == 0) {
            const long long now = clock64();
            atomicAdd(&out[phase], (unsigned long long)(now - last));
            last = now;
        }
    }
#else
    __device__ __forceinline__ void start(unsigned long long*) {}
    __device__ __forceinline__ void mark(int) {}
#endif
};

// One entry per DISTINCT trial width, ascending (numpy.unique, core.py:113); `row` is the
// first template row with that width (core.py:163-165).
struct WidthEntry {
    int width;        // trial duration d in samples
    int row;          // template row reported for this width
    int q_offset;     // index of q_0 of this row in the padded q array
    int q_len;        // len(signal) (== width in practice)
    int xth;          // T0 stride (core.py:50-55)
    int n_pos;        // trial T0 positions u = 0..n_pos-1, window start i = u*xth <= M-d
    int n_chunks;     // phase-3 work units of this width: ceil(n_pos/kR) if tiled, else n_pos
    int list_base;    // start of this width's live-unit list inside a workgroup's list slab
    double overshoot; // lc_cache_overview["overshoot"][row]
    double sum_q2;    // sum_j q_j^2 (uniform-weight case: A(i) = w0 * sum_q2)
    double inv_d;     // 1/d
    int tiled;        // row_is_tiled(width, xth)
    int prunable;     // cell_bound() is valid for this row (q_len == width, k_mono >= 0, depth_min >= 0)
    int oversize;     // slab variant: the window is wider than an LDS tile can hold -- predicate and dot
    int pad_;         //   product of this row read the slab directly (very long series, N beyond ~150 k)
    double var_q;     // sum_j (q_j - mean q)^2
    double k_mono;    // sum_j q_j - overshoot * sum_j q_j^2
    double c_proxy;   // 4 * overshoot * k_mono: c_proxy * mean^2 ranks the cells of all rows by promise
    double screen_c;  // fp32 screen: |B32 - B| <= screen_c * max|e| for every window of this row (screen_cells)
};

// In-range widths of one period (core.py:143-156): the contiguous range [k_lo, k_hi) of the
// ascending width table; rows below k_x have the dense T0 grid (stride 1).  Host-computed.
struct PeriodRows {
    int k_lo, k_hi, k_x;
    int pad;   // slab variant: tile length of this period (0: SearchArgs::tile_len)
};

// Piecewise-constant image of a template row for the pruning bound (window_bound): the row's taps
// q_0..q_{L-1} are replaced by kSeg levels (the means of q over [b_k, b_{k+1})), so that the dot product
// with the folded series needs kSeg + 1 look-ups in the prefix sum C instead of L taps:
//     sum_j q~_j e_{i+j} = sq - sum_k g_k C[i + b_k],     e = 1 - f,
// (telescoped: g_0 = -q~_0, g_k = q~_{k-1} - q~_k, g_kSeg = q~_{kSeg-1}; sq = sum_j q~_j), and what the levels
// miss is bounded by Cauchy-Schwarz with r2 = sum_j (q_j - q~_j)^2.  Host-built (build_widths).
constexpr int kSeg = 8;
constexpr int kScreenMinLen = 32;   // shorter rows: the exact dot product costs about as much as the screen
struct RowScreen {
    int b[kSeg + 1];      // segment boundaries, b[0] = 0 < ... < b[kSeg] = q_len
    int valid;            // the screen may be used for this row
    int pad_[2];
    double g[kSeg + 1];   // telescoped level differences
    double sq;            // sum_j q~_j
    double r2;            // sum_j (q_j - q~_j)^2, rounded up
    double pad2_;
};
typedef const __attribute__((address_space(4))) RowScreen* const_screen_ptr;

struct SearchArgs {
    const double* t;        // [n]
    const double* y;        // [n]
    const double* w;        // [n] 1/dy^2, or nullptr when all weights equal w0
    const double* periods;  // [n_periods]
    const int* order;       // [n_periods] work order (most expensive first)
    const PeriodRows* rows; // [n_periods] in-range rows of the width table (core.py:148-156)
    const WidthEntry* widths;
    const double* q;        // template rows q_j = 1 - signal_j, zero padded front and back
    const double* q2;       // q_j^2, same layout (general weights only)
    const double* g;        // difference taps g_j = q_{j-1} - q_j (q_len + 1 per row), same layout (series in the HBM slab)
    const RowScreen* screens;   // [n_widths] piecewise-constant rows of the pruning bound (pruning variant)
    double* out_chi2;       // [n_periods]
    long long* out_row;     // [n_periods]
    double* out_depth;      // [n_periods]
    unsigned long long* counters;      // [3] evaluated cells, inner steps, issued lane-FMAs (nullptr: off)
    unsigned long long* phase_cycles;  // [kPhases] shader cycles per phase (nullptr: off)
    unsigned long long* check;         // [kChecks] violated bounds (debug build; nullptr: off)
    long long lds_bytes;               // dynamic LDS of the launch (debug checks)
    unsigned int* queue;        // [2] next work item, workgroups done (both 0 at launch, rewound by the kernel)
    double* scratch;        // non-resident: per-workgroup slabs of the folded series
    unsigned int* chunk_lists;  // per-workgroup lists of live chunks (phase 3a -> 3b)
    long long scratch_stride;   // doubles per slab
    long long list_stride;      // entries per workgroup
    // survey mode: n_curves light curves on the same time stamps share the fold + sort of a period
    int sort2;                  // tiled variant: use fold_and_sort_tiled (its LDS fits)
    double* debug_folded;                // test entry (tls_debug_folded): [n_periods][n] folded flux of every period, or nullptr
    double* debug_prefix;                // test entry (tls_debug_prefix): [n_periods][M + 1] prefix sum C of every period, or nullptr
    unsigned long long* period_cycles;   // developer entry (tls_debug_period_cycles): [n_periods] shader cycles per period, or nullptr
    int n_curves;               // >= 1; curve c reads y + c*n (w + c*n), writes out_* + c*n_periods
    const double* curve_S0;     // [n_curves] S0 per curve (n_curves > 1; else S0 / w0 below)
    const double* curve_w0;     // [n_curves]
    unsigned int* perm_scratch; // [blocks][n] the sort permutation of the period in flight (n_curves > 1)
    long long list_cap;         // entries of one array: live units | their bounds (float)
    long long prune_min_live;   // prune a period (tile) only when at least this many units are live
    int p2_shift;               // log2 of the block length of the coarse prefix sum of e^2 (pruning bound)
    double depth_min;
    double eps_fast;            // fast mode: half-width of the undecided band around depth_min (depth_pass)
    double slack_unit;          // pruning: rounding allowance of window_bound per sample of a window
    int exact_prefix;           // != 0: every period in exact mode (developer switch TLS_EXACT_PREFIX=1, debug entries)
    int cumsum_round;           // slab variant: elements the prefix sum takes through LDS per round
    int fast_slab;              // != 0: fast mode also for a series in the HBM slab (the host's choice: few undecided windows)
    int x_at_staging;           // != 0 (fast_slab): no prefix-sum pass -- X of a tile is formed when the tile is staged (scan_tile_x)
    const double* band_prefix;  // [n_widths + 1] expected number of windows inside the undecided band, rows < k (fast_slab)
    double band_max;            // a period that expects more of them than this starts in exact mode
    const float* q32;           // fp32 screen: the template rows rounded to fp32, same layout as q (uniform weights)
    long long q32_shifted;      // ... and, this many floats further, the same rows stored one element later
    float* split_lo;            // fp32 screen: [blocks][region] low halves of the folded samples (e = hi + lo exactly)
    void* park_cells;           // fp32 screen: [blocks][kParkCap] parked cells (tlsdev::ParkedCell)
    double e_abs_max;           // fp32 screen: max |1 - flux| of the light curve(s) of this launch
    double S0;
    double w0;
    int n, W, M;            // points, patch length, n + W
    int n_periods, n_widths, nb;  // nb: sort buckets
    int hdr_bytes;          // LDS header: fixed part + per-row tables (16-B multiple)
    int tile_len;           // non-resident: window-start positions per LDS tile (multiple of 320)
    int tile_halo;          // non-resident: samples staged behind a tile (widest window + slack)
    int region_pad;         // spare entries behind every folded-series region (region_pad_for)
    // two-kernel slab path (kRoleFold / kRoleSearch): the periods order[batch_lo .. batch_lo + batch_n) of one batch; the
    // folded series of work item w lives in slab w - batch_lo.  The search kernel's items are (period, position tile)
    // pairs: tile_prefix[w] = tiles of all work items in front of w (in queue order), item g of the batch is tile
    // tile_prefix[batch_lo] + g.  A tile's winner goes to partials[g]; the workgroup that finishes a period's last tile
    // (tiles_done[slot]) compares them and writes the period's result.
    int batch_lo, batch_n;
    const unsigned int* tile_prefix;   // [n_periods + 1]
    double* partials;                  // [items of the largest batch][3]: stat | td | (k, i)
    unsigned int* tiles_done;          // [periods of the largest batch], zero between launches
    unsigned int* fold_ready;          // [periods of the largest batch] 1: the slab is complete; zero between launches
    int split_fast;                    // two roles: != 0 the plan supports fast prefix-sum mode there (uniform weights, X at staging, taps g)
};

__device__ __forceinline__ double fold_phase(double t, double period, double epoch) {
    // core.py:9-18  (time - T0) / period - floor(...); IEEE division, bit-identical to the CPU.
    // The search folds with T0 = 0 (foldfast): t - 0.0 is exact, so one routine serves both.
    double x = (t - epoch) / period;
    return x - floor(x);
}

// 32-bit fixed-point phase: monotone in the phase (the scaling is exact), so it orders two points whenever it
// differs; equal keys are decided by the exact fp64 phase, then the index.
__device__ __forceinline__ unsigned int phase_key(double ph) {
    return ph < 1.0 ? (unsigned int)(ph * 4294967296.0) : 0xffffffffu;
}

__device__ __forceinline__ int bucket_of(double phase, double nb_d, int nb) {
    int b = (int)(phase * nb_d);  // monotone in phase
    return b < nb - 1 ? b : nb - 1;
}

struct Best {
    double stat;  // rs * (rs*A - 2B) as the reference computes it, smaller is better; +inf = nothing evaluated
    double td;    // target depth of that cell
    int k;        // index into widths
    int i;        // T0 sample index
};
// A lane's best cell WHILE a period is searched: what the reference's value is computed from (consider).  The cheap
// estimate of the statistic orders the cells; settle_best turns the winner into a Best (one division per lane and
// period instead of one per promising cell).
struct Lead {
    double stat;  // the estimate of rs * (rs*A - 2B); +inf = nothing evaluated
    double dX;    // X[i+d] - X[i] of the cell
    double B;     // its dot product sum q_j e_j w_j
    double A;     // its sum q_j^2 w_j (per-point weights only; uniform weights: the row's constant)
    int k;        // index into widths
    int i;        // T0 sample index
};
__device__ __forceinline__ Lead no_lead() {
    Lead l;
    l.stat = INFINITY; l.dX = 0.0; l.B = 0.0; l.A = 0.0; l.k = 0x7fffffff; l.i = 0x7fffffff;
    return l;
}

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

// Exclusive prefix sum of cnt[0..nb) in place; `wsum` is LDS scratch of kMaxWaves+1 words.  Two barriers: every
// thread adds up the totals of the waves in front of its own (broadcast reads), the wave scan runs on the DPP crossbar.
__device__ __forceinline__ unsigned int wave_inclusive_sum_u32(unsigned int v);
__device__ __forceinline__ void block_exclusive_scan(unsigned int* cnt, int nb, unsigned int* wsum) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int chunk = (nb + nt - 1) / nt;
    const int lo = tid * chunk, hi = min(lo + chunk, nb);
    unsigned int local = 0;
    for (int b = lo; b < hi; ++b) local += cnt[b];
    const unsigned int incl = wave_inclusive_sum_u32(local);
    if (lane == kWave - 1) wsum[wave] = incl;
    wg_sync();
    unsigned int run = incl - local;
    for (int v = 0; v < wave; ++v) run += wsum[v];
    for (int b = lo; b < hi; ++b) { unsigned int c = cnt[b]; cnt[b] = run; run += c; }
    wg_sync();
}

// ---------------------------------------------------------------------------------------
// Exact parallel evaluation of the SEQUENTIAL fp64 prefix sum  C[k+1] = fl(C[k] + f[k]).
//
// The depth predicate (core.py:58) must see the bits numpy.cumsum produces (helpers.py:72),
// and a left-to-right fp64 sum is a 5000-long dependent chain for one lane.  Inside one
// binade [2^m, 2^(m+1)) every partial sum is an integer multiple S*u of u = 2^(m-52), and
// adding f = a*u + rem (0 <= rem < u) rounds to  S + a + [rem > u/2] + [rem == u/2]*(S+a odd):
// an INTEGER recurrence whose only dependence on the running sum is its parity.  Each element
// is therefore a map  parity -> increment  (two integers), these maps compose associatively,
// and a workgroup scan over them reproduces the sequential rounding exactly.  The scan is
// valid until the running sum leaves the binade (S reaches 2^53); that one element is added
// in plain fp64 and the scan restarts in the new binade (about log2(N) restarts for flux ~ 1).
// Requires f[k] >= 0 and finite (validate.py:32-35 guarantees it for flux).
struct ParityInc {
    long long i0, i1;  // increment of S when S is even / odd before the step(s); saturating
};
constexpr long long kIncSat = 1LL << 54;
constexpr long long kBinadeEnd = 1LL << 53;

__device__ __forceinline__ ParityInc compose(const ParityInc& x, const ParityInc& y) {  // x, then y
    ParityInc z;
    const long long s0 = x.i0 + ((x.i0 & 1) ? y.i1 : y.i0);
    const long long s1 = x.i1 + (((x.i1 + 1) & 1) ? y.i1 : y.i0);
    z.i0 = s0 < kIncSat ? s0 : kIncSat;
    z.i1 = s1 < kIncSat ? s1 : kIncSat;
    return z;
}

// the step of one addend f in the binade with exponent m (u = 2^(m-52))
__device__ __forceinline__ ParityInc addend_step(double f, int m) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(f);
    const int ef = (int)((bits >> 52) & 0x7ff);
    unsigned long long mant = bits & ((1ull << 52) - 1ull);
    int e = -1022;
    if (ef != 0) { mant |= 1ull << 52; e = ef - 1023; }
    const int sh = m - e;
    long long base = 0;
    int tie = 0;
    if (sh <= 0) {
        base = kIncSat;  // the addend alone spans the binade: forces the plain-fp64 restart
    } else if (sh < 64) {
        const unsigned long long below = mant << (64 - sh);  // the bits cut off, left aligned
        base = (long long)(mant >> sh) + (below > (1ull << 63) ? 1 : 0);
        tie = below == (1ull << 63);
    }
    // a tie rounds to even: the increment is a+1 exactly when S + a is odd
    const int a_odd = (int)((mant >> (sh > 0 && sh < 64 ? sh : 0)) & 1ull) & (sh > 0 && sh < 64 ? 1 : 0);
    ParityInc t;
    t.i0 = base + (tie & a_odd);
    t.i1 = base + (tie & (a_odd ^ 1));
    return t;
}

// x, then y, without the saturation: for short local chains (<= kPerMax steps of <= 2^54 each)
__device__ __forceinline__ ParityInc compose_local(const ParityInc& x, const ParityInc& y) {
    ParityInc z;
    z.i0 = x.i0 + ((x.i0 & 1) ? y.i1 : y.i0);
    z.i1 = x.i1 + (((x.i1 + 1) & 1) ? y.i1 : y.i0);
    return z;
}

__device__ __forceinline__ double binade_value(long long S, int m) {
    // S * 2^(m-52) for 0 <= S < 2^53; m == -1022 also covers the subnormals (S < 2^52)
    unsigned long long bits;
    if (S >= (1LL << 52)) bits = ((unsigned long long)(m + 1023) << 52) | ((unsigned long long)S & ((1ull << 52) - 1ull));
    else bits = (unsigned long long)S;
    return __longlong_as_double((long long)bits);
}

// LDS scratch of the prefix-sum routines
constexpr int kCumsumChunk = 8192;  // non-resident variant: elements scanned per LDS round trip
constexpr int kMaxSeg = 32;   // binade changes handled per block by the one-pass variant
constexpr int kPerMax = 16;   // elements per thread and block in the one-pass variant
// The one-pass variant keeps the same parity maps in fp64.  In the binade [2^m, 2^(m+1)) let
// c0 = 2^m (even mantissa) and c1 = c0 + ulp (odd).  The hardware rounds c0 + f and c1 + f on the
// same grid, with the same tie rule, as it rounds S + f for any even / odd S of that binade, so
//     i0 = (c0 + f) - c0,   i1 = (c1 + f) - c1
// ARE the two increments (exact multiples of the ulp) -- four flops instead of forty integer
// instructions; the parity of a partial sum is bit 0 of c0 + increment.  A value >= 2^m marks a
// step that cannot stay inside the binade from any start ("saturated"); sums of steps are exact
// below 2^(m+1) and stay >= 2^m once they got there, so the verification in phase D sees them.
struct StepD {
    double i0, i1;
};
struct BinadeD {              // constants of one binade
    double c0, c1, lim, sat;  // 2^m, 2^m + ulp, 2^m (largest valid increment is below), 2^(m+1)
};
__device__ __forceinline__ BinadeD binade_constants(int m) {   // m in [-1022, 1023]
    BinadeD b;
    const long long bits = (long long)(m + 1023) << 52;
    b.c0 = __longlong_as_double(bits);
    b.c1 = __longlong_as_double(bits + 1);
    b.lim = b.c0;
    b.sat = m < 1023 ? __longlong_as_double((long long)(m + 1024) << 52) : INFINITY;
    return b;
}
__device__ __forceinline__ StepD step_of(double f, const BinadeD& b) {
    StepD t;
    if (!(f < b.lim)) { t.i0 = b.sat; t.i1 = b.sat; return t; }
    t.i0 = (b.c0 + f) - b.c0;
    t.i1 = (b.c1 + f) - b.c1;
    return t;
}
__device__ __forceinline__ int mantissa_bit0(double x) { return (int)(__double_as_longlong(x) & 1LL); }
// x, then y (both steps of the binade b); the result saturates
__device__ __forceinline__ StepD compose(const StepD& x, const StepD& y, const BinadeD& b) {
    const double s0 = x.i0 + (mantissa_bit0(b.c0 + x.i0) ? y.i1 : y.i0);
    const double s1 = x.i1 + (mantissa_bit0(b.c0 + x.i1) ? y.i0 : y.i1);   // odd start: parity flipped
    StepD z;
    z.i0 = s0 < b.lim ? s0 : b.sat;
    z.i1 = s1 < b.lim ? s1 : b.sat;
    return z;
}

struct SegAcc {               // scan element of the one-pass variant
    StepD map;                // composite step since the last binade change (or since the start)
    int cnt;                  // binade changes so far
    int reset;                // 1 if a binade change lies inside
};
struct CumsumScratch {
    ParityInc wave_tot[kMaxWaves];
    SegAcc seg_tot[kMaxWaves];
    StepD tab_T[kMaxSeg];           // composite step of the segment behind binade change j
    double tab_S[kMaxSeg];          // running sum the segment starts from
    double dtot[kMaxWaves];         // phase A: wave totals; afterwards (as int) the binade each wave ends in
    double state_s;                 // running sum at state_k
    double fail_s;
    int tab_c[kMaxSeg + 1];         // element index of binade change j
    int tab_m[kMaxSeg];             // binade (unbiased exponent) behind it
    int cross[kMaxWaves];
    int state_k, n_ok, fail_k, n_seg;
};

__device__ __forceinline__ int unbiased_exponent(double x, long long* mantissa) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int es = (int)((b >> 52) & 0x7ff);
    long long mant = (long long)(b & ((1ull << 52) - 1ull));
    int m = -1022;
    if (es != 0) { mant |= 1LL << 52; m = es - 1023; }
    if (mantissa) *mantissa = mant;
    return m;
}

// value of lane `src` (wave-uniform index) without a trip through the LDS crossbar
__device__ __forceinline__ int lane_value(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ long long lane_value(long long v, int src) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffLL), src);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ double lane_value(double v, int src) {
    return __longlong_as_double(lane_value(__double_as_longlong(v), src));
}

// Robust variant: C[k+1] = fl(C[k] + f[k]) for k in [k_a, k_end), starting from C[k_a] = s_a.
// One workgroup scan per binade of the running sum; all threads of the workgroup call this.
__device__ __forceinline__ void sequential_cumsum_by_binade(const double* f, double* C, int k_a0, int k_end, double s_a,
                                                   CumsumScratch* cs) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (tid == 0) { C[k_a0] = s_a; cs->state_s = s_a; cs->state_k = k_a0; }
    wg_sync();
    constexpr int kMaxPerThread = 8;
    for (;;) {
        const int k_a = cs->state_k;
        const double s_start = cs->state_s;
        if (k_a >= k_end) break;
        long long S_start;
        const int m = unbiased_exponent(s_start, &S_start);
        // how far to scan: flux ~ 1 leaves the binade after ~2^m elements
        long long want = m >= 0 && m < 30 ? (2LL << m) + 64 : (m < 0 ? 64 : (long long)nt * kMaxPerThread);
        if (want > (long long)nt * kMaxPerThread) want = (long long)nt * kMaxPerThread;
        const int k_b = (int)((long long)k_a + want < (long long)k_end ? k_a + want : k_end);
        const int per = (k_b - k_a + nt - 1) / nt;
        const int lo = k_a + tid * per < k_b ? k_a + tid * per : k_b;
        const int hi = lo + per < k_b ? lo + per : k_b;

        ParityInc loc; loc.i0 = 0; loc.i1 = 0;
        for (int k = lo; k < hi; ++k) loc = compose(loc, addend_step(f[k], m));
        ParityInc inc = loc;  // inclusive scan over the lanes, lower lanes first
#pragma unroll
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            ParityInc o;
            o.i0 = __shfl_up(inc.i0, dlt, kWave);
            o.i1 = __shfl_up(inc.i1, dlt, kWave);
            if (lane >= dlt) inc = compose(o, inc);
        }
        if (lane == kWave - 1) cs->wave_tot[wave] = inc;
        wg_sync();
        ParityInc pre; pre.i0 = 0; pre.i1 = 0;
        for (int v = 0; v < wave; ++v) pre = compose(pre, cs->wave_tot[v]);
        ParityInc excl;
        excl.i0 = __shfl_up(inc.i0, 1, kWave);
        excl.i1 = __shfl_up(inc.i1, 1, kWave);
        if (lane == 0) { excl.i0 = 0; excl.i1 = 0; }
        pre = compose(pre, excl);
        long long S = S_start + ((S_start & 1) ? pre.i1 : pre.i0);
        int my_cross = 0x7fffffff;
        if (S < kBinadeEnd) {  // nothing before this thread's slice left the binade
            for (int k = lo; k < hi; ++k) {
                const ParityInc t = addend_step(f[k], m);
                S += (S & 1) ? t.i1 : t.i0;
                if (S >= kBinadeEnd) { my_cross = k; break; }
                C[k + 1] = binade_value(S, m);
            }
        }
#pragma unroll
        for (int dlt = kWave / 2; dlt > 0; dlt >>= 1) {
            const int o = __shfl_down(my_cross, dlt, kWave);
            my_cross = o < my_cross ? o : my_cross;
        }
        if (lane == 0) cs->cross[wave] = my_cross;
        wg_sync();
        if (tid == 0) {
            int k_c = cs->cross[0];
            for (int v = 1; v < nw; ++v) k_c = cs->cross[v] < k_c ? cs->cross[v] : k_c;
            if (k_c < k_b) {           // the sum leaves the binade at element k_c: plain fp64 step
                const double s_new = C[k_c] + f[k_c];
                C[k_c + 1] = s_new;
                cs->state_s = s_new;
                cs->state_k = k_c + 1;
            } else {
                cs->state_s = C[k_b];
                cs->state_k = k_b;
            }
        }
        wg_sync();
    }
}

// x, then y; `b` is the binade y's elements live in (only used when y holds no binade change)
__device__ __forceinline__ SegAcc seg_combine(const SegAcc& x, const SegAcc& y, const BinadeD& b) {
    SegAcc z;
    z.cnt = x.cnt + y.cnt;
    z.reset = x.reset | y.reset;
    z.map = y.reset ? y.map : compose(x.map, y.map, b);
    return z;
}

// C[0] = s_start, C[k+1] = fl(C[k] + f[k]) for k < count; all threads of the workgroup call this.
//
// One-pass variant on top of the binade scan above.  A plain (re-associated) parallel prefix
// sum P first predicts WHERE the running sum changes binade -- P is within ~1e-13 of the
// sequential sum, so exponent(P[k]) is the binade of the sequential sum except within a hair
// of a power of two.  With the binades fixed, one SEGMENTED scan of the parity maps (segments
// restart behind every predicted binade change) covers all binades at once; the few elements
// that change the binade are chained by one lane in plain fp64, which also VERIFIES the
// prediction exactly (start exponent of every segment, integer mantissa staying below 2^53).
// Whatever fails verification is redone from that point by the per-binade routine, so the
// result is always the sequential sum, bit for bit.
__device__ __forceinline__ void exact_sequential_cumsum(const double* f, double* C, int count, CumsumScratch* cs,
                                               unsigned long long* dbg = nullptr, double s_start = 0.0) {
    PhaseClock cpc; cpc.start(dbg);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (tid == 0) C[0] = s_start;
    int k0 = 0;
    double s0 = s_start;
    const int block_len = nt * kPerMax;
    while (k0 < count) {
        const int kb = k0 + block_len < count ? k0 + block_len : count;
        const int per = (kb - k0 + nt - 1) / nt;  // <= kPerMax
        const int lo = k0 + tid * per < kb ? k0 + tid * per : kb;
        const int hi = lo + per < kb ? lo + per : kb;
        // ---- A: re-associated prefix sum P[k] (sum before element k) into C[k0..kb] ----
        if (tid < kMaxSeg) { cs->tab_T[tid].i0 = 0.0; cs->tab_T[tid].i1 = 0.0; }
        double local = 0.0;
        for (int k = lo; k < hi; ++k) local += f[k];
        double incl = local;
#pragma unroll
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            const double o = __shfl_up(incl, dlt, kWave);
            if (lane >= dlt) incl += o;
        }
        if (lane == kWave - 1) cs->dtot[wave] = incl;
        wg_sync();
        {
            double run = s0;
            for (int v = 0; v < wave; ++v) run += cs->dtot[v];
            run += incl - local;
            for (int k = lo; k < hi; ++k) { C[k] = run; run += f[k]; }
            if (hi == kb && lo < hi) C[kb] = run;
        }
        wg_sync();
        cpc.mark(14);
        // ---- B1: predicted binade changes and their running count ----
        unsigned int crossmask = 0;   // bit e: element lo+e changes the binade (or opens the block)
        int m_lo = 0;
        {
            int m_k = unbiased_exponent(C[lo], nullptr);   // lo == kb for the idle threads behind the block
            m_lo = m_k;
            for (int k = lo; k < hi; ++k) {
                const int m_next = unbiased_exponent(C[k + 1], nullptr);
                if (k == k0 || m_next != m_k) crossmask |= 1u << (k - lo);
                m_k = m_next;
            }
        }
        const int n_cross_local = __popc(crossmask);
        int cnt_incl = n_cross_local;
#pragma unroll
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            const int o = __shfl_up(cnt_incl, dlt, kWave);
            if (lane >= dlt) cnt_incl += o;
        }
        if (lane == kWave - 1) cs->cross[wave] = cnt_incl;
        wg_sync();
        int cnt_pre = cnt_incl - n_cross_local;
        int n_seg = 0;
        for (int v = 0; v < nw; ++v) { const int c = cs->cross[v]; n_seg += c; if (v < wave) cnt_pre += c; }
        cpc.mark(15);
        // ---- B2: one walk over the elements: step maps, tables of the binade changes ----
        SegAcc acc; acc.cnt = n_cross_local; acc.reset = n_cross_local > 0;
        StepD head; head.i0 = 0.0; head.i1 = 0.0;   // composite in front of the first change in my slice
        const BinadeD b_lo = binade_constants(m_lo);  // the binade my slice starts in
        int m_hi = m_lo;                              // ... and the one it ends in
        {
            StepD seg; seg.i0 = 0.0; seg.i1 = 0.0;    // composite since the last change (or since lo)
            int j = cnt_pre - 1;
            BinadeD bk = b_lo;
            bool first = true;
            for (int k = lo; k < hi; ++k) {
                if ((crossmask >> (k - lo)) & 1u) {
                    if (first) { head = seg; first = false; }
                    else if (j >= 0 && j < kMaxSeg) cs->tab_T[j] = seg;  // a segment inside my slice
                    ++j;
                    m_hi = unbiased_exponent(C[k + 1], nullptr);
                    bk = binade_constants(m_hi);
                    if (j <= kMaxSeg) cs->tab_c[j] = k;
                    if (j < kMaxSeg) cs->tab_m[j] = m_hi;
                    seg.i0 = 0.0; seg.i1 = 0.0;
                } else {
                    seg = compose(seg, step_of(f[k], bk), bk);
                }
            }
            if (first) head = seg;
            acc.map = seg;
        }
        cpc.mark(16);
        SegAcc inc = acc;
        // a lane's accumulated range holds no binade change <=> all of it lies in the binade its own
        // slice starts in, and so does the tail of whatever is composed in front of it
#pragma unroll
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            SegAcc o;
            o.map.i0 = __shfl_up(inc.map.i0, dlt, kWave);
            o.map.i1 = __shfl_up(inc.map.i1, dlt, kWave);
            o.cnt = __shfl_up(inc.cnt, dlt, kWave);
            o.reset = __shfl_up(inc.reset, dlt, kWave);
            if (lane >= dlt) inc = seg_combine(o, inc, b_lo);
        }
        int* const wave_end_m = reinterpret_cast<int*>(cs->dtot);   // dtot is idle since the end of phase A
        if (lane == kWave - 1) { cs->seg_tot[wave] = inc; wave_end_m[wave] = m_hi; }
        wg_sync();
        // prefix over the wave totals: every wave scans them itself (lane v holds wave v's total), a
        // log-step scan instead of a chain of up to 15 dependent LDS reads and combines
        SegAcc pre; pre.map.i0 = 0.0; pre.map.i1 = 0.0; pre.cnt = 0; pre.reset = 0;
        if (nw <= 8) {   // few waves: the plain chain is shorter than the scan's fixed rounds
            for (int v = 0; v < wave; ++v) pre = seg_combine(pre, cs->seg_tot[v], binade_constants(wave_end_m[v]));
        } else {
            SegAcc w_inc = pre;
            int w_m = 0;
            if (lane < nw) { w_inc = cs->seg_tot[lane]; w_m = wave_end_m[lane]; }
            const BinadeD w_b = binade_constants(w_m);   // the binade wave `lane` ends in
#pragma unroll
            for (int dlt = 1; dlt < kMaxWaves; dlt <<= 1) {
                SegAcc o;
                o.map.i0 = __shfl_up(w_inc.map.i0, dlt, kWave);
                o.map.i1 = __shfl_up(w_inc.map.i1, dlt, kWave);
                o.cnt = __shfl_up(w_inc.cnt, dlt, kWave);
                o.reset = __shfl_up(w_inc.reset, dlt, kWave);
                if (lane >= dlt && lane < nw) w_inc = seg_combine(o, w_inc, w_b);
            }
            // the inclusive prefix of wave - 1 is this wave's exclusive one
            const int src = wave > 0 ? wave - 1 : 0;
            SegAcc got;
            got.map.i0 = lane_value(w_inc.map.i0, src);
            got.map.i1 = lane_value(w_inc.map.i1, src);
            got.cnt = lane_value(w_inc.cnt, src);
            got.reset = lane_value(w_inc.reset, src);
            if (wave > 0) pre = got;
        }
        {
            SegAcc ex;
            ex.map.i0 = __shfl_up(inc.map.i0, 1, kWave);
            ex.map.i1 = __shfl_up(inc.map.i1, 1, kWave);
            ex.cnt = __shfl_up(inc.cnt, 1, kWave);
            ex.reset = __shfl_up(inc.reset, 1, kWave);
            if (lane == 0) { ex.map.i0 = 0.0; ex.map.i1 = 0.0; ex.cnt = 0; ex.reset = 0; }
            pre = seg_combine(pre, ex, b_lo);   // ex ends where my slice starts
        }
        // segments that reach into my slice from the left, and the last one of the block
        if (lo < hi) {
            const int j_in = cnt_pre - 1;
            if (n_cross_local > 0) {
                if (j_in >= 0 && j_in < kMaxSeg) cs->tab_T[j_in] = compose(pre.map, head, b_lo);
                if (hi == kb) { const int j = cnt_pre + n_cross_local - 1; if (j < kMaxSeg) cs->tab_T[j] = acc.map; }
            } else if (hi == kb && j_in >= 0 && j_in < kMaxSeg) {
                cs->tab_T[j_in] = compose(pre.map, head, b_lo);
            }
        }
        wg_sync();
        cpc.mark(17);
        // ---- D: wave 0 chains the binade changes in fp64 and verifies the prediction ----
        // Inside a verified binade the whole segment behind change j adds the step t_j (chosen by the
        // parity of the start mantissa), and staying below 2^(m+1) makes that one exact fp64
        // addition: the chain is two additions per change.
        if (wave == 0) {
            const int n_proc = n_seg < kMaxSeg ? n_seg : kMaxSeg;
            // lane j holds change j: its element, addend, predicted binade and segment step
            int my_c = 0, my_m = 0;
            double my_f = 0.0, my_t0 = 0.0, my_t1 = 0.0, my_lim = 0.0;
            if (lane < n_proc) {
                my_c = cs->tab_c[lane]; my_m = cs->tab_m[lane]; my_f = f[my_c];
                // a saturated step (>= 2^m) pushes any sum of the binade to 2^(m+1) or beyond
                my_t0 = cs->tab_T[lane].i0;
                my_t1 = cs->tab_T[lane].i1;
                my_lim = binade_constants(my_m).sat;
            }
            // The chain itself is two additions and a parity select per change; every lane runs it
            // (uniform values), lane j keeps the three values of step j, and the checks happen
            // afterwards, all steps at once.
            double s_end = s0;
            double keep_prev = 0.0, keep_new = 0.0, keep_end = 0.0;
            for (int j = 0; j < n_proc; ++j) {
                const double s_new = s_end + lane_value(my_f, j);  // the sequential step itself
                const double t = mantissa_bit0(s_new) ? lane_value(my_t1, j) : lane_value(my_t0, j);
                const double s_nxt = s_new + t;
                if (lane == j) { keep_prev = s_end; keep_new = s_new; keep_end = s_nxt; }
                s_end = s_nxt;
            }
            // verification: the predicted binade of every segment start, the segment staying inside it
            bool bad_exp = false, bad_lim = false;
            if (lane < n_proc) {
                const long long sb = __double_as_longlong(keep_new);
                const int es = (int)((sb >> 52) & 0x7ff);
                bad_exp = (es ? es - 1023 : -1022) != my_m;
                bad_lim = !(keep_end < my_lim);
            }
            const unsigned long long bad = ballot64(bad_exp || bad_lim);
            int ok = n_proc, n_written = n_proc, fail_k = kb;
            double fail_s = 0.0;
            if (bad) {
                const int jf = __ffsll((long long)bad) - 1;   // everything behind the first failure is void
                const bool exp_failed = (ballot64(bad_exp) >> jf) & 1ull;
                ok = jf;
                if (exp_failed) { n_written = jf; fail_k = lane_value(my_c, jf); fail_s = lane_value(keep_prev, jf); }
                else { n_written = jf + 1; fail_k = lane_value(my_c, jf) + 1; fail_s = lane_value(keep_new, jf); }
            } else if (n_seg > n_proc) {
                fail_k = cs->tab_c[n_proc]; fail_s = s_end;
            }
            // a change whose own step verified still gets its C value, even if its segment failed
            if (lane < n_written) C[my_c + 1] = keep_new;
            if (lane < ok) cs->tab_S[lane] = keep_new;
            if (lane == 0) { cs->n_ok = ok; cs->fail_k = fail_k; cs->fail_s = fail_s; }
        }
        wg_sync();
        cpc.mark(18);
        // ---- E: exact values of every element inside a verified segment ----
        {
            const int n_ok = cs->n_ok;
            int j = cnt_pre - 1;
            double S = 0.0;
            BinadeD bk = b_lo;
            if (j >= 0 && j < n_ok) {
                const double S0 = cs->tab_S[j];
                S = S0 + (mantissa_bit0(S0) ? pre.map.i1 : pre.map.i0);
                bk = binade_constants(cs->tab_m[j]);
            }
            for (int k = lo; k < hi; ++k) {
                if ((crossmask >> (k - lo)) & 1u) {
                    ++j;
                    if (j < n_ok) { S = cs->tab_S[j]; bk = binade_constants(cs->tab_m[j]); }
                } else if (j < n_ok) {
                    const StepD t = step_of(f[k], bk);
                    S += mantissa_bit0(S) ? t.i1 : t.i0;
                    C[k + 1] = S;
                }
            }
        }
        wg_sync();
        cpc.mark(19);
        const int fail_k = cs->fail_k;
        const double fail_s = cs->fail_s;
        wg_sync();
        if (dbg && tid == 0) { atomicAdd(&dbg[10], 1ull); if (fail_k < kb) atomicAdd(&dbg[11], 1ull); }
        if (fail_k < kb) sequential_cumsum_by_binade(f, C, fail_k, kb, fail_s, cs);
        k0 = kb;
        s0 = C[kb];
        wg_sync();
    }
}

// ---------------------------------------------------------------------------------------
// Wave-level scans on the DPP crossbar (row_shr inside the rows of 16 lanes, row_bcast:15/31 across
// them): a VALU move per 32-bit word and step instead of a ds_bpermute round trip through the LDS.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    // lanes without a source (outside the row, or rows masked off) receive 0
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i32<CTRL, ROW_MASK>((int)(b & 0xffffffffLL));
    const int hi = dpp_i32<CTRL, ROW_MASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// the counter update of the slab sort's partition passes (bin < 0: no point in this lane)
template <bool WANT_TICKET>
__device__ __forceinline__ unsigned int bin_inc(unsigned int* cnt, int bin) {
    // (one atomic per RUN of equal bins in a wave -- time-ordered samples share a coarse bin -- was tried: the
    // ballot/shuffle bookkeeping costs more than the same-address serialisation it removes)
    return bin >= 0 ? atomicAdd(&cnt[bin], 1u) : 0u;
}

// inclusive prefix sum over the 64 lanes (0.0 is what the lanes without a source contribute)
__device__ __forceinline__ double wave_inclusive_sum(double v) {
    v += dpp_f64<kDppRowShr1, 0xF>(v);
    v += dpp_f64<kDppRowShr2, 0xF>(v);
    v += dpp_f64<kDppRowShr4, 0xF>(v);
    v += dpp_f64<kDppRowShr8, 0xF>(v);
    v += dpp_f64<kDppBcast15, 0xA>(v);
    v += dpp_f64<kDppBcast31, 0xC>(v);
    return v;
}

__device__ __forceinline__ unsigned int wave_inclusive_sum_u32(unsigned int v) {
    v += (unsigned int)dpp_i32<kDppRowShr1, 0xF>((int)v);
    v += (unsigned int)dpp_i32<kDppRowShr2, 0xF>((int)v);
    v += (unsigned int)dpp_i32<kDppRowShr4, 0xF>((int)v);
    v += (unsigned int)dpp_i32<kDppRowShr8, 0xF>((int)v);
    v += (unsigned int)dpp_i32<kDppBcast15, 0xA>((int)v);
    v += (unsigned int)dpp_i32<kDppBcast31, 0xC>((int)v);
    return v;
}

// ---------------------------------------------------------------------------------------
// One block of the exact sequential prefix sum, elements held in registers (second formulation).
//
// Inside the binade [2^m, 2^(m+1)) every partial sum is a multiple of u = 2^(m-52).  Adding x rounds
// to S + RN_u(x) where RN_u(x) = (2^m + x) - 2^m does NOT depend on S -- except when x lies exactly
// half way between two multiples of u (a TIE: |x - RN_u(x)| == u/2, about one element per binade for
// noisy flux), where round-to-even looks at the parity of S.  So the elements fall into two classes:
//   ordinary   the step is the constant RN_u(x); sums of such steps inside one binade are exact under
//              ANY association (multiples of u below 2^(m+1)): a plain parallel prefix sum;
//   special    ties and the elements that carry the sum into the next binade (and the first element of
//              the block): their step is whatever the hardware's fp64 addition S + x gives, which is
//              the sequential step by definition.  A handful per block, chained one after the other.
// Which binade a partial sum lies in is PREDICTED from a plain re-associated prefix sum P (phase A); the
// ordinary steps are formed for the predicted binades (phase B), the specials go to a small table, and
// every wave chains the wave totals and the table for itself to obtain the exact sum at its own start
// and behind every special inside its range.  Phase E then runs the ACTUAL recurrence over the thread's
// elements from its start value (constant step; plain addition for ties and binade changes), so a
// thread's values are exact whenever its start value is.  The start values are verified, not trusted:
// thread T+1's start must equal thread T's end bit for bit, and thread 0 starts from s0 -- by induction
// every value is the sequential sum.  Any mismatch (a mispredicted binade, more specials than table
// slots) sends the whole block through sequential_cumsum_by_binade.  Three workgroup barriers.
constexpr int kMaxSpecial = 32;   // special elements handled per block by the fast path
struct SegSum {                   // scan element: sum of the ordinary steps behind the last special of the
    double sum;                   //   scanned range (or of the whole range), and that special's table
    int tag;                      //   slot + 1 (0: no special inside)
};
__device__ __forceinline__ SegSum seg_combine(const SegSum& x, const SegSum& y) {   // x (lower elements), then y
    SegSum z;
    z.tag = y.tag ? y.tag : x.tag;
    z.sum = y.tag ? y.sum : x.sum + y.sum;
    return z;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ SegSum dpp_seg(const SegSum& v) {
    SegSum o;
    o.sum = dpp_f64<CTRL, ROW_MASK>(v.sum);
    o.tag = dpp_i32<CTRL, ROW_MASK>(v.tag);
    return o;
}
__device__ __forceinline__ SegSum wave_inclusive_seg(SegSum v) {
    v = seg_combine(dpp_seg<kDppRowShr1, 0xF>(v), v);
    v = seg_combine(dpp_seg<kDppRowShr2, 0xF>(v), v);
    v = seg_combine(dpp_seg<kDppRowShr4, 0xF>(v), v);
    v = seg_combine(dpp_seg<kDppRowShr8, 0xF>(v), v);
    v = seg_combine(dpp_seg<kDppBcast15, 0xA>(v), v);
    v = seg_combine(dpp_seg<kDppBcast31, 0xC>(v), v);
    return v;
}
struct BinadeU {                  // constants of one binade for the constant-step form
    double c0, half_u, sat;       // 2^m, u/2 = 2^(m-53) (0 in the lowest binades: everything is "special"), 2^(m+1)
};
__device__ __forceinline__ BinadeU binade_u(int m) {   // m in [-1022, 1023]
    BinadeU b;
    b.c0 = __longlong_as_double((long long)(m + 1023) << 52);
    b.half_u = m - 53 >= -1022 ? __longlong_as_double((long long)(m - 53 + 1023) << 52) : 0.0;
    b.sat = m < 1023 ? __longlong_as_double((long long)(m + 1024) << 52) : INFINITY;
    return b;
}

// LDS scratch of exact_cumsum_block (shares its slot with CumsumScratch: the fallback runs after it)
struct Cumsum2Scratch {
    double wsum[kMaxWaves];          // A: plain sums of the waves
    SegSum wseg[kMaxWaves];          // B: ordinary-step total of every wave (behind its last special, if any)
    double tab_H[kMaxSpecial];       // ordinary steps of the owner wave in front of the special (since the
                                     //    wave's start or its previous special)
    double tab_f[kMaxSpecial];       // the special element itself
    unsigned int tab_key[kMaxSpecial];  // thread * 32 + element slot: orders the specials, names the owner wave
    double wstart[kMaxWaves + 1];    // E: exact running sum at the first element of every wave (chain value)
    double wendv[kMaxWaves];         //    ... and behind its last one (recurrence value)
    unsigned int n_special;
    int fail;                        // some check failed: the block is redone by the per-binade routine
};
static_assert(sizeof(Cumsum2Scratch) <= kCumsumScratchBytes, "cumsum scratch does not fit its slot");
static_assert(kMaxSpecial <= kWave && kMaxWaves <= kWave, "the tables are held one entry per lane");
static_assert(128 <= 2 * kWave, "the bin scan of the slab sort holds two bins per lane");

//     C[k0] = s0,  C[k+1] = fl(C[k] + f[k])  for k in [k0, kb),   kb - k0 <= blockDim.x * PER.
// Thread T owns `per` <= PER consecutive elements (slots past its range hold 0.0).
// ALIASED: C[k+1] is stored over f[k] (f == C + 1); the elements are then restored before a fallback.
// Returns C[kb].
template <int PER, bool ALIASED>
__device__ __forceinline__ double exact_cumsum_block_inline(const double* f, double* C, int k0, int kb, double s0,
                                                            Cumsum2Scratch* cs, unsigned long long* dbg) {
    static_assert(PER <= 32, "the special key packs the element slot into 5 bits");
    PhaseClock cpc; cpc.start(dbg);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: nothing derived from it is hoisted out of a caller's loop (and spilled)
    const int nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int per = (kb - k0 + nt - 1) / nt;
    if ((per & 1) == 0 && per < PER) per += 1;      // odd stride: the 8-byte LDS accesses of a wave hit 32 bank pairs
    const int lo = k0 + tid * per < kb ? k0 + tid * per : kb;
    const int mine = lo + per < kb ? per : kb - lo;  // elements of this thread, 0..per
    double x[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) x[e] = e < mine ? f[lo + e] : 0.0;
    if (tid == 0) { cs->fail = 0; cs->n_special = 0u; }
    // ---- A: plain prefix sums ------------------------------------------------------------------
    double local = 0.0;
#pragma unroll
    for (int e = 0; e < PER; ++e) local += x[e];
    const double incl = wave_inclusive_sum(local);
    if (lane == kWave - 1) cs->wsum[wave] = incl;
    lds_barrier();                                                                          // 1
    double wave_base = s0;
    for (int v = 0; v < wave; ++v) wave_base += cs->wsum[v];   // the same additions in every thread of the wave
    // The end of my range IS the start of the next thread's: one definition for both sides.
    const double p_start = wave_base + dpp_f64<kDppWaveShr1, 0xF>(incl);   // lane 0: + 0.0
    double p_next = dpp_f64<kDppWaveShl1, 0xF>(p_start);                   // lane 63: replaced below
    if (lane == kWave - 1) p_next = (wave_base + cs->wsum[wave]) + 0.0;    // = p_start of the next wave's lane 0
    cpc.mark(14);
    // ---- B: ordinary steps for the predicted binades; specials go to the table --------------------------
    SegSum acc; acc.sum = 0.0; acc.tag = 0;
    double head = 0.0;                             // ordinary steps in front of the first special in my range
    int slot_first = -1;                           // table slot of the first special in my range
    bool overflow = false;
    {
        BinadeU bk = binade_u(unbiased_exponent(p_start, nullptr));
        double seg = 0.0, p = p_start;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (e < mine) {
                const double p_after = e == mine - 1 ? p_next : p + x[e];
                const double i0 = (bk.c0 + x[e]) - bk.c0;                   // RN_u(x)
                const bool leaves = !(p_after < bk.sat);
                const bool special = leaves || fabs(x[e] - i0) == bk.half_u || lo + e == k0;
                if (special) {                                                // rare
                    const unsigned int slot = atomicAdd(&cs->n_special, 1u);
                    const bool ok = slot < (unsigned int)kMaxSpecial;
                    overflow |= !ok;
                    if (ok) {
                        cs->tab_key[slot] = (unsigned int)tid * 32u + (unsigned int)e;
                        cs->tab_f[slot] = x[e];
                        if (slot_first >= 0) cs->tab_H[slot] = seg;            // since my previous special
                    }
                    if (slot_first < 0) { head = seg; slot_first = ok ? (int)slot : kMaxSpecial; }
                    acc.tag = ok ? (int)slot + 1 : kMaxSpecial;
                    if (leaves) bk = binade_u(unbiased_exponent(p_after, nullptr));
                    seg = 0.0;
                } else {
                    seg += i0;
                }
                p = p_after;
            }
        }
        if (slot_first < 0) head = seg;
        acc.sum = seg;
    }
    const SegSum inc = wave_inclusive_seg(acc);
    const SegSum exc = dpp_seg<kDppWaveShr1, 0xF>(inc);        // the lanes in front of me, inside my wave
    // in front of my first special: the wave's ordinary steps since its previous special (or its start), then mine
    if (slot_first >= 0 && slot_first < kMaxSpecial) cs->tab_H[slot_first] = exc.sum + head;
    if (lane == kWave - 1) cs->wseg[wave] = inc;
    if (overflow) cs->fail = 1;
    cpc.mark(15);
    lds_barrier();                                                                          // 2
    // ---- chain: every wave for itself, over the earlier waves and the specials up to its own ----------------
    // lane v holds wave v's total, lane j the table entry of slot j and its rank in element order
    const int n_sp = (int)(cs->n_special < (unsigned int)kMaxSpecial ? cs->n_special : (unsigned int)kMaxSpecial);
    double w_sum = 0.0, h_H = 0.0, h_f = 0.0;
    unsigned int h_key = 0xffffffffu;
    if (lane < nw) w_sum = cs->wseg[lane].sum;
    if (lane < n_sp) { h_H = cs->tab_H[lane]; h_f = cs->tab_f[lane]; h_key = cs->tab_key[lane]; }
    int rank = 0;
    for (int i = 0; i < n_sp; ++i) rank += ((unsigned int)lane_value((int)h_key, i) < h_key) ? 1 : 0;
    double S = s0, S_wave = s0;
    double seg_start = 0.0;                        // lane j: exact sum behind special j (those of my wave)
    {
        int r = 0;
        for (int v = 0; v <= wave; ++v) {
            if (v == wave) S_wave = S;
            while (r < n_sp) {
                const unsigned long long sel = ballot64(rank == r && lane < n_sp);
                const int j = __ffsll((long long)sel) - 1;
                const unsigned int key = (unsigned int)lane_value((int)h_key, j);
                if ((int)(key >> 11) != v) break;                              // key / (32 * 64): the owner wave
                S = S + lane_value(h_H, j);                                    // ordinary steps up to the special: exact
                S = S + lane_value(h_f, j);                                    // its own step: the hardware's fp64 addition
                if (v == wave && lane == j) seg_start = S;
                ++r;
            }
            if (v < wave) S = S + lane_value(w_sum, v);
        }
    }
    if (lane == 0) cs->wstart[wave] = S_wave;
    // my start value: the wave's start, or the sum behind the last special in front of me in my wave
    // (the shuffle runs in ALL lanes, outside the select: ds_bpermute returns 0 for a source lane that is
    // masked off, and the lane that holds a special's sum need not itself lie behind one)
    const double seg_base = __shfl(seg_start, exc.tag > 0 && exc.tag <= kMaxSpecial ? exc.tag - 1 : 0, kWave);
    const double S_start = (exc.tag ? seg_base : S_wave) + exc.sum;
    cpc.mark(16);
    // ---- E: the recurrence itself ---------------------------------------------------------------------
    double S_run = S_start;
    {
        BinadeU bk = binade_u(unbiased_exponent(S_run, nullptr));
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            if (e < mine) {
                const double i0 = (bk.c0 + x[e]) - bk.c0;
                double Sn = S_run + i0;
                if (fabs(x[e] - i0) == bk.half_u || !(Sn < bk.sat)) {   // a tie, or the sum leaves the binade:
                    Sn = S_run + x[e];                                     // the plain addition IS the step
                    if (!(Sn < bk.sat)) bk = binade_u(unbiased_exponent(Sn, nullptr));
                }
                S_run = Sn;
                C[lo + e + 1] = Sn;
            }
        }
    }
    if (tid == 0) C[k0] = s0;
    // ---- verification of the start values ---------------------------------------------------------------
    const double next_start = dpp_f64<kDppWaveShl1, 0xF>(S_start);
    bool bad = lane < kWave - 1 && !(S_run == next_start);
    if (tid == 0) bad |= !(S_start == s0);
    if (lane == kWave - 1) cs->wendv[wave] = S_run;
    const unsigned long long bad_lanes = ballot64(bad);
    if (bad_lanes != 0ull && lane == 0) cs->fail = 1;
    cpc.mark(17);
    lds_barrier();                                                                          // 3 (LDS only: C in
    // global memory is the caller's to publish; every thread has written only its own elements)
    bool failed = cs->fail != 0;
    for (int v = 0; v + 1 < nw; ++v) failed |= !(cs->wendv[v] == cs->wstart[v + 1]);
    const double s_end = cs->wendv[nw - 1];
    if (dbg && tid == 0) { atomicAdd(&dbg[10], 1ull); if (failed) atomicAdd(&dbg[11], 1ull); }
#ifdef TLS_CUMSUM_DIAG   // developer build: why a block failed (slots 18..22 of the debug kernel's clock buffer)
    if (dbg) {
        if (overflow) atomicAdd(&dbg[18], 1ull);
        if (lane == 0 && bad_lanes) atomicAdd(&dbg[19], (unsigned long long)__popcll(bad_lanes));
        if (tid == 0) {
            for (int v = 0; v + 1 < nw; ++v) if (!(cs->wendv[v] == cs->wstart[v + 1])) atomicAdd(&dbg[20], 1ull);
            if (!(S_start == s0)) atomicAdd(&dbg[21], 1ull);
            atomicAdd(&dbg[23], (unsigned long long)cs->n_special);
        }
        if (lane == 0 && bad_lanes) atomicMin(&dbg[22], (unsigned long long)(wave * 64 + __ffsll((long long)bad_lanes) - 1));
    }
#endif
    if (failed) {
        wg_sync();                            // everybody has read the scratch: its slot is reused below
        if constexpr (ALIASED) {
#pragma unroll
            for (int e = 0; e < PER; ++e) if (e < mine) const_cast<double*>(f)[lo + e] = x[e];
            wg_sync();
        }
        sequential_cumsum_by_binade(f, C, k0, kb, s0, reinterpret_cast<CumsumScratch*>(cs));
        wg_sync();
        return C[kb];
    }
    return s_end;
}

// The same as a real function: one copy per (PER, ALIASED) shared by all callers and kernels (the block is
// ~4000 instructions; inlined into every phase that needs it, it evicted the rest from the instruction cache)
template <int PER, bool ALIASED, bool IN_LDS>
__device__ __noinline__ double exact_cumsum_block(const double* f, double* C, int k0, int kb, double s0,
                                                  Cumsum2Scratch* cs, unsigned long long* dbg) {
    if constexpr (IN_LDS)
        return exact_cumsum_block_inline<PER, ALIASED>(as_lds(f), as_lds(C), k0, kb, s0, as_lds(cs), dbg);
    else
        return exact_cumsum_block_inline<PER, ALIASED>(f, C, k0, kb, s0, as_lds(cs), dbg);
}

// C[0] = s_start, C[k+1] = fl(C[k] + f[k]) for k < count: blocks of blockDim.x * 16 elements (one for an
// LDS-resident light curve).  All threads of the workgroup call this; C is complete at return.
// IN_LDS: f and C live in LDS (every kernel use; the developer test entry passes global memory).
// INLINE: the block is inlined at the call site (pointers keep their address space and uniformity: the
// LDS-resident kernel, which runs exactly one block per period); otherwise a call to the shared copy.
template <bool ALIASED, bool IN_LDS = true, bool INLINE = false>
__device__ __forceinline__ double exact_cumsum(const double* f, double* C, int count, Cumsum2Scratch* cs,
                                               unsigned long long* dbg = nullptr, double s_start = 0.0) {
    const int nt = blockDim.x;
    if (count <= 0) { if (threadIdx.x == 0) C[0] = s_start; lds_barrier(); return s_start; }
    double s0 = s_start;
    for (int k0 = 0; k0 < count; ) {
        const int left = count - k0;
        int kb;
        if constexpr (INLINE) {
            if (left <= 8 * nt) { kb = count; s0 = exact_cumsum_block_inline<8, ALIASED>(f, C, k0, kb, s0, cs, dbg); }
            else if (left <= 12 * nt) { kb = count; s0 = exact_cumsum_block_inline<12, ALIASED>(f, C, k0, kb, s0, cs, dbg); }
            else { kb = left <= 16 * nt ? count : k0 + 16 * nt; s0 = exact_cumsum_block_inline<16, ALIASED>(f, C, k0, kb, s0, cs, dbg); }
        } else {
            if (left <= 5 * nt) { kb = count; s0 = exact_cumsum_block<5, ALIASED, IN_LDS>(f, C, k0, kb, s0, cs, dbg); }
            else if (left <= 8 * nt) { kb = count; s0 = exact_cumsum_block<8, ALIASED, IN_LDS>(f, C, k0, kb, s0, cs, dbg); }
            else if (left <= 12 * nt) { kb = count; s0 = exact_cumsum_block<12, ALIASED, IN_LDS>(f, C, k0, kb, s0, cs, dbg); }
            else { kb = left <= 16 * nt ? count : k0 + 16 * nt; s0 = exact_cumsum_block<16, ALIASED, IN_LDS>(f, C, k0, kb, s0, cs, dbg); }
        }
        k0 = kb;
        if (k0 < count) lds_barrier();               // the scratch is rewritten by the next block
    }
    return s0;
}

// Fast mode, series in the HBM slab: X of one tile formed in place from the tile's staged flux.  buf[k] holds the folded flux
// of position p_lo + k for k < have; afterwards buf[k] = X[p_lo + k] = carry + sum_{j<k} (1 - f_j) for k <= min(have, cap - 1)
// (a plain parallel scan, any association: depth_pass).  Two LDS-only barriers; wtot: kMaxWaves doubles of scratch.
__device__ __forceinline__ void scan_tile_x(double* buf, int have, int cap, double carry, double* wtot, int tid) {
    const int nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int per = (have + nt - 1) / nt;
    if ((per & 1) == 0) per += 1;    // odd stride: the 8-byte LDS accesses of a wave hit 32 bank pairs
    const int lo = tid * per < have ? tid * per : have;
    const int hi = lo + per < have ? lo + per : have;
    double local = 0.0;
    for (int k = lo; k < hi; ++k) local += 1.0 - buf[k];
    const double incl = wave_inclusive_sum(local);
    if (lane == kWave - 1) wtot[wave] = incl;
    lds_barrier();
    double run = carry;
    for (int v = 0; v < wave; ++v) run += wtot[v];   // the same additions in every thread of the wave
    run += incl - local;
    for (int k = lo; k < hi; ++k) { const double e1 = 1.0 - buf[k]; buf[k] = run; run += e1; }
    if (have < cap) {   // X behind the last staged sample
        if (have == 0) { if (tid == 0) buf[0] = carry; }
        else if (hi == have && lo < hi) buf[have] = run;
    }
    lds_barrier();
}

// Fast mode of the LDS-resident kernel: fe[k] = e_k = 1 - f[k] in place and X[0] = 0, X[k+1] = X[k] + e_k as a plain
// parallel prefix sum (any association: the values are ~1e-4..1e-2, their sums carry ~1e-18 of rounding).  With
// per-point weights the samples become e*w afterwards; X is the sum of the unweighted e.  One LDS-only barrier.
template <bool UNIFORM_W>
__device__ __forceinline__ void prefix_sum_of_e(double* fe, const double* w, double* X, int M, double* wtot) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int nt = blockDim.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int per = (M + nt - 1) / nt;
    if ((per & 1) == 0) per += 1;    // odd stride: the 8-byte LDS accesses of a wave hit 32 bank pairs
    const int lo = tid * per < M ? tid * per : M;
    const int hi = lo + per < M ? lo + per : M;
    double local = 0.0;
    for (int k = lo; k < hi; ++k) { const double e = 1.0 - fe[k]; fe[k] = e; local += e; }
    const double incl = wave_inclusive_sum(local);
    if (lane == kWave - 1) wtot[wave] = incl;
    lds_barrier();
    double run = 0.0;
    for (int v = 0; v < wave; ++v) run += wtot[v];
    run += incl - local;
    for (int k = lo; k < hi; ++k) {
        X[k] = run;
        const double e = fe[k];
        run += e;
        if constexpr (!UNIFORM_W) fe[k] = e * w[k];
    }
    if (hi == M && lo < hi) X[M] = run;
}

// Depth predicate of one trial cell (core.py:58): mean = 1 - (C[i+d]-C[i])/d > depth_min.
// The kernels keep the prefix sum in the form  X[k] = k - C[k]  (the running sum of e = 1 - f): a window's
// X[i+d] - X[i] = d - (C[i+d] - C[i]) is d times its mean depth.  Two ways to fill X:
//   exact mode  C is the sequential fp64 cumsum, bit for bit numpy.cumsum (exact_cumsum), and X[k] = k - C[k]
//               is an EXACT subtraction (both are multiples of ulp(C) and the result is small), as is
//               dd - dX = C[i+d] - C[i]: the reference expression 1 - (C[i+d]-C[i])/d is evaluated on its own bits.
//   fast mode   (LDS-resident series) X is a plain parallel prefix sum of e = 1 - f.  It differs from k - C by
//               the rounding the sequential cumsum has collected, at most half an ulp of the total per step, i.e.
//               |dX/d - mean_reference| <= 2^-53 * C_max + O(1e-16) =: eps/2 for EVERY window.  A window whose
//               mean is farther than eps from transit_depth_min is decided exactly as the reference decides it;
//               one inside the band (about one period in fifty at N = 4320) is not decided at all: the thread
//               raises `undecided`, and the workgroup searches that period again in exact mode.
// So the set of evaluated cells is the reference's in both modes; what fast mode changes is the depth scale of
// a cell by <= eps (~1e-12 absolute on a mean of ~1e-5..1e-2, i.e. <= ~1e-10 relative on chi^2; typically 1e-12).
__device__ __forceinline__ bool depth_pass(double dX, double inv_d, double dd, double dmin, double eps,
                                           bool exact_mode, bool& undecided) {
    const double m_fast = dX * inv_d;
    if (m_fast > dmin + eps) return true;
    if (!(m_fast >= dmin - eps)) return false;
    if (exact_mode) return (1.0 - (dd - dX) / dd) > dmin;   // dd - dX == C[i+d] - C[i], exactly
    undecided = true;
    return false;
}

// Per-row (= per in-range trial duration) bookkeeping of one period, in the LDS header.
struct RowTables {
    unsigned int* live;         // [rows]   live units found (atomic tail of the row's list)
    unsigned int* singles;      // [rows]   > 0: the row was re-listed as that many single positions
    unsigned int* batch_start;  // [rows+1] prefix of 64-unit batches (phase 3b work items)
    unsigned int* next_batch;   // [1]      dynamic batch counter
};

// Branch-and-bound pruning (uniform weights).  For a window with mean depth m (= 1 - dC/d) the
// statistic is  -stat = rs*(2B - rs*A),  rs = 2*m*overshoot,  B = sum_j q_j e_j.  Splitting
// q and e into mean and fluctuation,  B = m*sum(q) + sum (q_j - qbar)(e_j - m),  and Cauchy-Schwarz
// bounds the second term by sqrt(var_q * V),  V = sum (e_j - m)^2 = sum e_j^2 - d*m^2.  Hence
//     -stat <= U = 4*ov*m*(m*k_mono + sqrt(var_q*V)),   k_mono = sum(q) - ov*sum(q^2) >= 0,
// which increases with m and with V.  For a chunk of kR windows m is taken from the deepest
// window, V from the shallowest one, and sum e^2 from a coarse (block-granular, hence
// over-estimating) prefix sum.  A cell whose bound is below a statistic that some evaluated cell
// has already reached cannot win (strict '<', core.py:70-74): skipping it leaves the result
// unchanged.  All roundings go upwards (the float steps are inflated by 1e-6).
__device__ __forceinline__ float cell_bound(double dX_max, double dX_min, double dd, double inv_d, double ov,
                                            double k_mono, double var_q, double e2) {
    const double m_hi = dX_max * inv_d * (1.0 + 1e-9) + 1e-15;
    if (!(dX_max > 0.0)) return -INFINITY;
    const double m_lo = fmax(dX_min * inv_d * (1.0 - 1e-9) - 1e-15, 0.0);
    const double V = fmax(fma(-dd * m_lo, m_lo, e2), 0.0) * (1.0 + 1e-6) + 1e-9 * e2;
    const float f = sqrtf((float)(var_q * V)) * (1.0f + 1e-6f);
    const double U = 4.0 * ov * m_hi * fma(m_hi, k_mono, (double)f);
    return (float)(U * (1.0 + 1e-6));
}

// sum of e^2 over [lo, hi) from the coarse prefix sum, rounded outwards to whole blocks
__device__ __forceinline__ double coarse_e2(const double* P2, int lo, int hi, int shift, int n_blocks) {
    int b_lo = lo >> shift, b_hi = (hi + (1 << shift) - 1) >> shift;
    b_lo = b_lo < n_blocks ? b_lo : n_blocks;
    b_hi = b_hi < n_blocks ? b_hi : n_blocks;
    return P2[b_hi] - P2[b_lo];
}

// Tight bound of ONE window (uniform weights): the template row is replaced by its piecewise-constant image
// (RowScreen), whose dot product B~ = sum_k g_k X[i + b_k] with e comes from kSeg + 1 values of X, and the remainder is
// bounded by Cauchy-Schwarz:  sum_j (q_j - q~_j) e_j = sum_j (q_j - q~_j)(e_j - m)  (the levels are segment
// means, so the differences sum to zero)  <=  sqrt(r2 * V),  V = sum (e_j - m)^2 = sum e_j^2 - d m^2.  Hence
//     -stat = rs (2B - rs A) <= U = rs (2 (B~ + sqrt(r2 V) + slack) - rs A),     rs = 2 m ov > 0.
// The cell_bound above is the one-segment case of this; with eight segments the remainder is ~20x smaller and
// at 50 ppm four out of five template taps of a search are never multiplied.  `slack` covers the rounding of C
// (sequential sum: at most half an ulp of the total per step, over the d steps of a window) and of the
// telescoped sum; sum e^2 comes from the coarse prefix sum (rounded outwards), all other roundings go upwards.
// Windows that fail the depth predicate (core.py:58) -- among them those past the end of the T0 grid, which read
// the sentinels behind C -- return -inf.
__device__ __forceinline__ float window_bound(const double* x, int i, int d, double dd, double inv_d, double ov, double A,
                                              const_screen_ptr s, const double* P2, int shift, int n_blocks,
                                              double dmin, double eps, bool exact_mode, bool& undecided, double slack) {
    const double x0 = x[i], xK = x[i + d];
    double xm[kSeg - 1];
#pragma unroll
    for (int k = 1; k < kSeg; ++k) xm[k - 1] = x[i + s->b[k]];
    const double dX = xK - x0;
    const bool pass = depth_pass(dX, inv_d, dd, dmin, eps, exact_mode, undecided);
    double Bt = s->g[0] * x0;
#pragma unroll
    for (int k = 1; k < kSeg; ++k) Bt = fma(s->g[k], xm[k - 1], Bt);
    Bt = fma(s->g[kSeg], xK, Bt);
    const double m = dX * inv_d;
    const double e2 = coarse_e2(P2, i, i + d, shift, n_blocks);
    const double m_lo = fmax(m, 0.0) * (1.0 - 1e-12);
    const double V = fmax(fma(-dd * m_lo, m_lo, e2), 0.0) * (1.0 + 1e-6) + 1e-9 * e2;
    const float f = sqrtf((float)(s->r2 * V)) * (1.0f + 1e-6f) + 1e-30f;
    const double Bub = Bt + (double)f + slack;
    const double rs = 2.0 * (m * ov);
    const double U = rs * (2.0 * Bub - rs * A);
    float uf = (float)(U + 1e-6 * fabs(U) + 1e-300);
    uf += fabsf(uf) * 1e-6f;
    return pass ? uf : -INFINITY;
}

// The same for a UNIT of n_win windows xth samples apart (the kR windows of a chunk): per window only the
// piecewise-constant dot product B~ and the window sum dX; square root, variance and statistic once, for
//     B_up = max_r B~_r + sqrt(r2 V) + slack,   V from the shallowest passing window and sum e^2 over the whole unit,
//     U = max over rs in [rs(shallowest), rs(deepest)] of rs (2 B_up - rs A)     (a parabola in rs: vertex or end point)
// which is >= window_bound of every window of the unit.  A quarter of the instructions of n_win window_bounds.
__device__ __forceinline__ float unit_bound(const double* x, int b, int n_win, int xth, int d, double dd, double inv_d, double ov,
                                            double A, const_screen_ptr s, const double* P2, int shift, int n_blocks,
                                            double dmin, double eps, bool exact_mode, bool& undecided, double slack) {
    double B_max = -INFINITY, dX_max = -INFINITY, dX_min = INFINITY;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        if (r < n_win) {
            const int i = b + r * xth;
            const double x0 = x[i], xK = x[i + d];
            double xm[kSeg - 1];
#pragma unroll
            for (int k = 1; k < kSeg; ++k) xm[k - 1] = x[i + s->b[k]];
            const double dX = xK - x0;
            double Bt = s->g[0] * x0;
#pragma unroll
            for (int k = 1; k < kSeg; ++k) Bt = fma(s->g[k], xm[k - 1], Bt);
            Bt = fma(s->g[kSeg], xK, Bt);
            if (depth_pass(dX, inv_d, dd, dmin, eps, exact_mode, undecided)) {
                B_max = fmax(B_max, Bt); dX_max = fmax(dX_max, dX); dX_min = fmin(dX_min, dX);
            }
        }
    }
    if (!(dX_max > -INFINITY)) return -INFINITY;   // no window of the unit passes the depth predicate
    // (the depth the evaluation uses, 1 - (d - dX)/d, carries ~1e-16 of absolute rounding, 1e-11 of a 1e-5 depth)
    const double m_hi = dX_max * inv_d * (1.0 + 1e-9) + 1e-15;
    const double m_lo = fmax(dX_min * inv_d * (1.0 - 1e-9) - 1e-15, 0.0);
    const double e2 = coarse_e2(P2, b, b + (n_win - 1) * xth + d, shift, n_blocks);
    const double V = fmax(fma(-dd * m_lo, m_lo, e2), 0.0) * (1.0 + 1e-6) + 1e-9 * e2;
    const float f = sqrtf((float)(s->r2 * V)) * (1.0f + 1e-6f) + 1e-30f;
    const double B_up = B_max + (double)f + slack;
    const double rs_lo = 2.0 * (m_lo * ov), rs_hi = 2.0 * (m_hi * ov);
    const double rs = fmin(fmax(B_up / A, rs_lo), rs_hi);
    const double U = rs * (2.0 * B_up - rs * A);
    float uf = (float)(U + 1e-6 * fabs(U) + 1e-300);
    uf += fabsf(uf) * 1e-6f;
    return uf;
}

// Candidate evaluation shared by all dot-product variants: given the dot products of one
// T0 position, apply the predicate, form the statistic and keep the lane's best.  Positions
// past the end of the T0 grid read the +huge sentinels behind C and fail the predicate.
// Fast prefix-sum mode, series in the HBM slab: a window inside the undecided band is NOTED (row, position, its window
// sum in the plain scan) instead of voiding the attempt.  After the attempt the period's exact prefix sum is formed once
// (the folded flux is still in the slab), every noted window is decided on it by the reference's own expression, and the
// few that pass are evaluated and meet the lanes' leads like any other cell -- with the plain scan's window sum, so that
// every cell of the period is valued on the same X.  The period's result is then that of exact mode's CELL SET on fast
// mode's values: a function of the light curve and the period alone, at the cost of one prefix pass instead of a second
// search (round 4: 0.65 of a period for 2.6 % of the TESS-size and 10 % of the Kepler-size periods).
struct BandEntry { int k, i; double dX; };
constexpr int kBandCap = 256;   // noted windows per period; beyond that the period is searched again in exact mode
struct DepthRule {   // how the depth predicate is decided in this period (depth_pass)
    double dmin, eps;
    double reach;        // relative band in which two cells' estimated statistics do not order them (consider)
    bool exact_mode;
    unsigned int* band_count;   // LDS counter of the noted windows; nullptr: a window inside the band voids the attempt
    BandEntry* band_list;
};
__device__ __forceinline__ void band_window(const DepthRule& rule, int k, int i, double dX, bool& undecided) {
    if (rule.band_count != nullptr) {
        const unsigned int at = atomicAdd(rule.band_count, 1u);
        if (at < (unsigned int)kBandCap) { BandEntry e; e.k = k; e.i = i; e.dX = dX; rule.band_list[at] = e; }
    } else {
        undecided = true;
    }
}
// The statistic of a cell as the reference computes it (core.py:61-69 on helpers.py:73's mean).
__device__ __forceinline__ void exact_cell(double dX, double dd, double overshoot, double A, double B, double& stat, double& td) {
    const double mean = 1.0 - (dd - dX) / dd;   // helpers.py:73 + core.py:167; dd - dX is C[i+d] - C[i] (exact mode: its bits)
    td = mean * overshoot;                        // core.py:61
    const double rs = 2.0 * td;                   // 1/(SIGNAL_DEPTH/td), core.py:62-63
    stat = rs * (rs * A - 2.0 * B);
}

// One trial cell against the lane's best (core.py:58-74).  Cells are ordered by an ESTIMATE of the statistic (the mean
// depth as dX/d instead of the reference's 1 - (d - dX)/d: the two differ by the rounding of a quotient near 1, about
// 1e-16 absolute on the mean, i.e. below rule.reach / 4 relative on the statistic).  An estimate that is lower than the
// best one by more than rule.reach (relative) belongs to a cell whose reference value is lower too; one that is higher by
// more than that cannot win; in between (about one cell in 1e8) both cells are evaluated as the reference does and
// compared with its tie rule.  Straight-line code otherwise: no division and no branch per cell.
// KEEP_DX: the lead carries its dX; otherwise X stays addressable for the whole period (x_all: the series resident in
// LDS) and dX is read again the one or two times it is needed -- two registers less through the whole search.
// FORCE_LIVE: the cell has been decided elsewhere (a noted band window that passed on the exact prefix sum).
template <bool UNIFORM_W, bool KEEP_DX, bool FORCE_LIVE = false>
__device__ __forceinline__ void consider(Lead& best, double x_lo, double x_hi, int i, double inv_d,
                                         double dd, const DepthRule& rule, double overshoot, double A, double B,
                                         int k, unsigned int& n_eval, bool& undecided, const_width_ptr widths_c,
                                         const double* x_all) {
    const double dX = x_hi - x_lo;
    const double m_fast = dX * inv_d;
    bool live = FORCE_LIVE || m_fast > rule.dmin + rule.eps;    // depth_pass, with its rare band out of line
    if (!live && m_fast >= rule.dmin - rule.eps) {
        if (rule.exact_mode) live = (1.0 - (dd - dX) / dd) > rule.dmin;
        else band_window(rule, k, i, dX, undecided);
    }
    n_eval += live ? 1u : 0u;
    const double rs_f = 2.0 * (m_fast * overshoot);
    const double stat_f = rs_f * (rs_f * A - 2.0 * B);
    const double reach = rule.reach * fabs(best.stat);
    // (nothing evaluated yet: inf - inf is NaN and the comparison false, so the first live cell wins)
    bool wins = live && !(stat_f >= best.stat - reach);
    if (live && !wins && stat_f <= best.stat + reach) {
        double s_new, td_new, s_old, td_old;
        exact_cell(dX, dd, overshoot, A, B, s_new, td_new);
        const double A_old = UNIFORM_W ? widths_c[best.k].sum_q2 : best.A;
        const int d_old = widths_c[best.k].width;
        const double dX_old = KEEP_DX ? best.dX : x_all[best.i + d_old] - x_all[best.i];
        exact_cell(dX_old, (double)d_old, widths_c[best.k].overshoot, A_old, best.B, s_old, td_old);
        // strict '<' with first-visited-wins ties: ascending width, then ascending T0
        wins = s_new < s_old || (s_new == s_old && (k < best.k || (k == best.k && i < best.i)));
    }
    if (wins) {
        best.stat = stat_f; best.B = B; best.k = k; best.i = i;
        if constexpr (KEEP_DX) best.dX = dX;
        if constexpr (!UNIFORM_W) best.A = A;
    }
}

// R cells of one row at once (the kR windows of a chunk, or one window): consider()'s decisions in straight-line code
// with selects -- and whatever consider() would settle out of line (a window inside the undecided band of the depth
// predicate, two estimates within reach of each other) only raises `pending`; such a lane then takes all R cells through
// consider() itself.  The result does not depend on the order in which cells meet the lead (every comparison is decided
// by the reference's values whenever the estimates are close), and a cell that meets itself there ties and stays.
// (X is read here, x[i0 + r*step] and x[i0 + r*step + d], and read AGAIN by a pending lane: nothing but the dot products
// stays in registers for the rare case)
// X_LDS: x points into LDS (possibly biased below the tile: only its low 32 bits are used); read through the LDS address
// space -- a pointer that may be an LDS or a global one at run time would be dereferenced as a FLAT address, and a biased
// LDS pointer is not a valid one.
template <bool X_LDS>
__device__ __forceinline__ double x_load(const double* x, int idx) {
    if constexpr (X_LDS) return *((lds_f64_ptr)x + idx);
    else return x[idx];
}
template <bool UNIFORM_W, bool KEEP_DX, int R, bool COUNTING, bool X_LDS = false>
__device__ __forceinline__ void consider_cells(Lead& best, const double* x, int i0, int step, int d,
                                               double inv_d, double dd, const DepthRule& rule, double overshoot,
                                               const double (&A)[R], const double (&B)[R], int k, unsigned int& n_eval,
                                               bool& undecided, const_width_ptr widths_c, const double* x_all) {
    const double hi = rule.dmin + rule.eps, lo = rule.dmin - rule.eps;
    bool pending = false;
    unsigned int n_fast = 0;
    double x_lo[R], x_hi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { x_lo[r] = x_load<X_LDS>(x, i0 + r * step); x_hi[r] = x_load<X_LDS>(x, i0 + r * step + d); }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double dX = x_hi[r] - x_lo[r];
        const double m_fast = dX * inv_d;
        const bool live = m_fast > hi;
        pending |= !live && m_fast >= lo;
        if constexpr (COUNTING) n_fast += live ? 1u : 0u;
        const double rs_f = 2.0 * (m_fast * overshoot);            // (the same expressions as consider)
        const double stat_f = rs_f * (rs_f * A[r] - 2.0 * B[r]);
        const double reach = rule.reach * fabs(best.stat);
        const bool wins = live && !(stat_f >= best.stat - reach);
        pending |= live && !wins && stat_f <= best.stat + reach;
        best.stat = wins ? stat_f : best.stat;
        best.B = wins ? B[r] : best.B;
        best.k = wins ? k : best.k;
        best.i = wins ? i0 + r * step : best.i;
        if constexpr (KEEP_DX) best.dX = wins ? dX : best.dX;
        if constexpr (!UNIFORM_W) best.A = wins ? A[r] : best.A;
    }
    if (pending) {
        unsigned int n_slow = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int at = i0 + r * step;
            asm volatile("" : "+v"(at));   // a second read, not a value kept from the first
            consider<UNIFORM_W, KEEP_DX>(best, x_load<X_LDS>(x, at), x_load<X_LDS>(x, at + d), i0 + r * step, inv_d, dd, rule, overshoot, A[r], B[r], k, n_slow,
                                         undecided, widths_c, x_all);
        }
        n_fast = n_slow;
    }
    n_eval += n_fast;
}

// The lane's best cell in the reference's numbers (before the lanes' candidates are compared with each other).
template <bool UNIFORM_W, bool KEEP_DX>
__device__ __forceinline__ Best settle_best(const Lead& lead, const_width_ptr widths_c, const double* x_all) {
    Best b;
    b.stat = INFINITY; b.td = 0.0; b.k = lead.k; b.i = lead.i;
    if (lead.stat < INFINITY) {
        const double A = UNIFORM_W ? widths_c[lead.k].sum_q2 : lead.A;
        const int d = widths_c[lead.k].width;
        const double dX = KEEP_DX ? lead.dX : x_all[lead.i + d] - x_all[lead.i];
        exact_cell(dX, (double)d, widths_c[lead.k].overshoot, A, lead.B, b.stat, b.td);
    }
    return b;
}

// ---------------------------------------------------------------------------------------
// fp32 SCREEN of the sliding dot products (round 4; LDS-resident kernel, uniform weights, plain variant).
//
// The dot product B(i) = sum_j q_j e_{i+j} is what phase 3b spends its time on, and all but one of the ~1e4 cells of a
// period only have to LOSE.  So the folded samples are kept as two fp32 halves, e = hi + lo EXACTLY (e = 1 - flux is a
// multiple of 2^-53 and the host admits the screen only when max |e| < 2^-5: 48 significant bits at most), the dot
// products of all cells are formed from the high halves in packed fp32 (v_pk_fma_f32: two taps per lane and instruction,
// half the LDS bytes per sample), and a cell is valued in fp64 -- by the very additions the plain kernel performs, so
// the same bits -- only when its fp32 estimate, widened by a rigorous error bound, can still be the period's minimum:
//     |B32 - B| <= screen_c(row) * max|e|      (inputs rounded once, n <= (L + S)/2 + 8 fused accumulations per half)
//     est = rs (rs A - 2 B32),  err = |rs| * 2 screen_c max|e| + |est| * (reach + 1e-13),  v in [est - err, est + err]
// A lane keeps the cell with the smallest upper bound (`ScreenSlot`); a cell whose lower bound exceeds it cannot be the
// minimum and is dropped; the others (near-ties, and the slot cells that survive the workgroup-wide minimum of the
// upper bounds at the end of the period) go through exact_window_dot + consider().  The winner of the plain kernel is
// always among them, consider() is the plain kernel's comparison, so chi2, row and depth are the plain kernel's bits.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* const_f32_ptr;
typedef const __attribute__((address_space(3))) float* lds_f32_ptr;
typedef const __attribute__((address_space(1))) float* glob_f32_ptr;
typedef const __attribute__((address_space(1))) double* glob_f64_ptr;

struct ScreenSlot {      // the lane's cell with the smallest upper bound so far
    double hi, lo;       // est + err, est - err; hi = +inf: empty
    int k, i;            // row (index into widths), T0 sample index
};
__device__ __forceinline__ ScreenSlot empty_slot() {
    ScreenSlot s; s.hi = INFINITY; s.lo = INFINITY; s.k = 0x7fffffff; s.i = 0x7fffffff;
    return s;
}

// four pairs of consecutive fp32 samples; the address is 8-byte aligned (an unaligned ds_read_b64 is served at an
// eighth of the rate on gfx950: measured, tools/micro/pk_dot.hip)
__device__ __forceinline__ void load_pairs32(unsigned addr, f32x2 (&x)[kU / 2]) {
    static_assert(kU == 8, "the asm block reads 4 pairs");
    asm volatile(
        "ds_read_b64 %0, %4\n\t"
        "ds_read_b64 %1, %4 offset:8\n\t"
        "ds_read_b64 %2, %4 offset:16\n\t"
        "ds_read_b64 %3, %4 offset:24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(addr) : "memory");
}

// kR sliding fp32 dot products of one lane, windows XTH samples apart (dot_windows in packed fp32).  Sample-major like
// dot_windows: the pair of samples (t, t+1) feeds window r with the pair of taps (t - r*XTH, t + 1 - r*XTH), which is an
// even-aligned pair of the tap stream for even r*XTH and of the stream shifted by one tap otherwise -- both streams are
// wave-uniform scalar loads (the shifted one from its own copy of the rows), so every v_pk_fma_f32 takes its taps from
// an SGPR pair.
template <int XTH>
__device__ __forceinline__ void dot_windows32(unsigned addr, const_f32_ptr q, const_f32_ptr q_shifted, int L, float (&B)[kR]) {
    constexpr int S = (kR - 1) * XTH;
    constexpr int S0 = (S + 1) & ~1;
    f32x2 acc[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) acc[r] = f32x2{0.0f, 0.0f};
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        const const_f32_ptr qa = q + (t0 - S0);        // qa[m] = q_ext[t0 - S0 + m]
        // the same stream one tap later, from a second copy of the rows stored one element further (q_shifted[j] =
        // q[j - 1]): two aligned scalar loads instead of one load and a dozen s_mov to re-pair its registers
        const const_f32_ptr qb = q_shifted + (t0 - S0);
        float ta[kU + S0], tb[kU + S0 + 2];
#pragma unroll
        for (int m = 0; m < kU + S0; ++m) ta[m] = qa[m];
#pragma unroll
        for (int m = 0; m < kU + S0 + 2; ++m) tb[m] = qb[m];
        f32x2 x[kU / 2];
        load_pairs32(addr + 4u * (unsigned)t0, x);
#pragma unroll
        for (int k = 0; k < kU / 2; ++k)
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int off = r * XTH;
                f32x2 tap;
                if ((off & 1) == 0) { const int m = 2 * k - off + S0; tap = f32x2{ta[m], ta[m + 1]}; }
                else { const int m = 2 * k - off + S0 + 1; tap = f32x2{tb[m], tb[m + 1]}; }
                acc[r] = __builtin_elementwise_fma(tap, x[k], acc[r]);
            }
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) B[r] = acc[r].x + acc[r].y;
}
// one window per lane (wide T0 strides, re-listed positions)
__device__ __forceinline__ float dot_window32(unsigned addr, const_f32_ptr q, int L) {
    f32x2 acc = f32x2{0.0f, 0.0f};
    for (int t0 = 0; t0 < L; t0 += kU) {
        const const_f32_ptr qs = q + t0;
        float ta[kU];
#pragma unroll
        for (int m = 0; m < kU; ++m) ta[m] = qs[m];
        f32x2 x[kU / 2];
        load_pairs32(addr + 4u * (unsigned)t0, x);
#pragma unroll
        for (int k = 0; k < kU / 2; ++k) acc = __builtin_elementwise_fma(f32x2{ta[2 * k], ta[2 * k + 1]}, x[k], acc);
    }
    return acc.x + acc.y;
}

// The dot product of ONE cell as the plain kernel forms it: one fp64 accumulator, taps in order (the zero taps of the
// padded rows add nothing), samples rebuilt from their halves.  One wavefront works on the one cell (wave-uniform
// arguments): lane j rebuilds sample j and fetches tap j of a 64-sample stretch -- one memory round trip per stretch instead
// of one per tap -- into the wavefront's staging buffers in LDS; the chain of additions then runs like the plain kernel's
// one-window loop on broadcast LDS reads, eight taps at a time.  Every period needs this once (its winner) and now and
// then for a near-tie.
__device__ __forceinline__ double exact_window_dot(unsigned hi_addr, glob_f32_ptr lo, glob_f64_ptr q_row, int L, int i, double* stage) {
    // (stage: 2 * kWave doubles -- samples, then taps.  Both are read back as broadcast LDS loads, so the only wait inside
    // the chain is for LDS data requested one step earlier: the next eight taps and samples travel while these eight are added)
    const lds_f32_ptr hi = (lds_f32_ptr)(uintptr_t)hi_addr;
    const int lane = (int)(threadIdx.x & (kWave - 1));
    double* const st_e = stage;
    double* const st_q = stage + kWave;
    double B = 0.0;
#pragma unroll 1
    for (int base = 0; base < L; base += kWave) {
        const int idx = base + lane;
        double ev = 0.0, qv = 0.0;
        if (idx < L) {
            ev = (double)hi[i + idx] + (double)lo[i + idx];
            qv = q_row[idx];
        }
        st_e[lane] = ev;
        st_q[lane] = qv;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the wavefront's own buffers: no barrier)
        const int n = L - base < kWave ? L - base : kWave;
        double x[kU], tq[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) { x[u] = st_e[u]; tq[u] = st_q[u]; }
#pragma unroll 1
        for (int t0 = 0; t0 < n; t0 += kU) {
            double xn[kU], tn[kU];
            const int t1 = t0 + kU < kWave ? t0 + kU : 0;   // (the last step's prefetch is not used)
#pragma unroll
            for (int u = 0; u < kU; ++u) { xn[u] = st_e[t1 + u]; tn[u] = st_q[t1 + u]; }
#pragma unroll
            for (int u = 0; u < kU; ++u) B = fma(tq[u], x[u], B);   // (entries past L are zero taps on zero samples)
#pragma unroll
            for (int u = 0; u < kU; ++u) { x[u] = xn[u]; tq[u] = tn[u]; }
        }
    }
    return B;
}

// The same for ONE lane on its own (lane-divergent arguments, a memory round trip per tap): only when the workgroup's
// list of parked cells is full (a light curve on which thousands of cells tie, e.g. a constant one).
__device__ __forceinline__ double exact_window_dot_lane(unsigned hi_addr, glob_f32_ptr lo, glob_f64_ptr q, int q_offset, int L, int i) {
    const lds_f32_ptr hi = (lds_f32_ptr)(uintptr_t)hi_addr;
    double B = 0.0;
#pragma unroll 1
    for (int j = 0; j < L; ++j) B = fma(q[q_offset + j], (double)hi[i + j] + (double)lo[i + j], B);
    return B;
}

// Cells whose interval overlaps their lane's slot are PARKED in a list of the workgroup (its counter in LDS, the cells in
// the workgroup's global scratch: a few per period on a quiet light curve, hundreds on a noisy one) and valued after the
// last batch, by one wavefront, and only if their lower bound still reaches the workgroup's smallest upper bound by
// then: most of them no longer do.
struct ParkedCell { double lo; int k, i; };
constexpr int kParkCap = 1024;    // cells per workgroup and period; beyond that a lane values its cell on the spot
struct ParkList {                 // lives in the idle prefix-sum scratch (LDS)
    unsigned int n; unsigned int pad_[3];
    double stage[2 * kWave];      // staging buffers of the wavefront that values the parked cells (exact_window_dot)
};
static_assert(sizeof(ParkList) <= kCumsumScratchBytes, "the list head of the parked cells does not fit the prefix-sum scratch");
struct ScreenEnv {          // what the screen needs besides the cells
    unsigned eh_addr;       // LDS address of the high halves
    glob_f32_ptr lo_g;      // low halves (global scratch of the workgroup)
    glob_f64_ptr q_g;       // template rows, fp64
    ParkList* park;
    ParkedCell* cells;      // [kParkCap] the parked cells of this workgroup (global scratch)
    unsigned long long* stat;   // developer statistics (phase-clock buffer) or nullptr
};

// One cell valued as the plain kernel values it and met with a lane's lead (lane-divergent: the overflow path)
__device__ __forceinline__ void screen_value_cell_lane(Lead& lead, int k_c, int i_c, const DepthRule& rule, const_width_ptr widths_c,
                                                       const double* x_all, const ScreenEnv& env) {
    const int d_c = widths_c[k_c].width;
    const double B = exact_window_dot_lane(env.eh_addr, env.lo_g, env.q_g, widths_c[k_c].q_offset, widths_c[k_c].q_len, i_c);
    unsigned int n_dummy = 0;
    bool und_dummy = false;   // (the cell passed the depth predicate in screen_cells, by the same expressions)
    consider<true, false>(lead, x_all[i_c], x_all[i_c + d_c], i_c, widths_c[k_c].inv_d, (double)d_c, rule, widths_c[k_c].overshoot,
                          widths_c[k_c].sum_q2, B, k_c, n_dummy, und_dummy, widths_c, x_all);
}
// ... by the whole wavefront (wave-uniform k_c, i_c), met with the lead of lane `owner`
__device__ __forceinline__ void screen_value_cell(Lead& lead, int owner, int k_c, int i_c, const DepthRule& rule,
                                                  const_width_ptr widths_c, const double* x_all, const ScreenEnv& env) {
    const int d_c = widths_c[k_c].width;
    const double B = exact_window_dot(env.eh_addr, env.lo_g, env.q_g + widths_c[k_c].q_offset, widths_c[k_c].q_len, i_c,
                                      env.park->stage);
    if ((int)(threadIdx.x & (kWave - 1)) == owner) {
        unsigned int n_dummy = 0;
        bool und_dummy = false;
        consider<true, false>(lead, x_all[i_c], x_all[i_c + d_c], i_c, widths_c[k_c].inv_d, (double)d_c, rule,
                              widths_c[k_c].overshoot, widths_c[k_c].sum_q2, B, k_c, n_dummy, und_dummy, widths_c, x_all);
    }
}
// park a cell (lane-divergent); a full list values it on the spot
__device__ __forceinline__ void park_cell(Lead& lead, double lo, int k_c, int i_c, const DepthRule& rule, const_width_ptr widths_c,
                                          const double* x_all, const ScreenEnv& env) {
    const unsigned int at = atomicAdd(&env.park->n, 1u);
    if (at < (unsigned int)kParkCap) {
        env.cells[at].lo = lo; env.cells[at].k = k_c; env.cells[at].i = i_c;
    } else {
        screen_value_cell_lane(lead, k_c, i_c, rule, widths_c, x_all, env);
    }
    if (env.stat) atomicAdd(env.stat, 1ull);
}

// R cells of one row (the kR windows of a chunk, or one window) against the lane's slot: consider_cells for the screen.
// Straight-line code decides which cells can still be the minimum (`qual`); those few are handled out of line.
template <int R>
__device__ __forceinline__ void screen_cells(Lead& lead, ScreenSlot& slot, const double* x, int i0, int step, int d, double inv_d,
                                             double dd, const DepthRule& rule, double overshoot, double A, const float (&B32)[R],
                                             int k, double errB2, bool& undecided, const_width_ptr widths_c, const double* x_all,
                                             const ScreenEnv& env) {
    const double hi_t = rule.dmin + rule.eps, lo_t = rule.dmin - rule.eps;
    const double rel = rule.reach + 1e-13;
    bool pend = false;
    unsigned int qual = 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double dX = x[i0 + r * step + d] - x[i0 + r * step];
        const double m_fast = dX * inv_d;
        const bool live = m_fast > hi_t;
        pend |= !live && m_fast >= lo_t;
        const double rs_f = 2.0 * (m_fast * overshoot);            // (the same expressions as consider)
        const double est = rs_f * (rs_f * A - 2.0 * (double)B32[r]);
        const double err = fabs(rs_f) * errB2 + fabs(est) * rel;
        const double lo = est - err, hi = est + err;
        // below the slot cell for certain: it takes the slot, the old cell (lower bound above this upper bound) is out;
        // intervals that overlap: out of line (about one cell in a few hundred)
        const bool clear = live && hi < slot.lo;
        qual |= (live && !clear && lo <= slot.hi) ? (1u << r) : 0u;
        slot.hi = clear ? hi : slot.hi;
        slot.lo = clear ? lo : slot.lo;
        slot.k = clear ? k : slot.k;
        slot.i = clear ? i0 + r * step : slot.i;
    }
    if (pend) {   // a window inside the undecided band of the depth predicate (depth_pass)
        if (!rule.exact_mode) undecided = true;
        else {
#pragma unroll 1
            for (int r = 0; r < R; ++r) {
                const double* xr = x + (i0 + r * step);
                asm volatile("" : "+v"(xr));
                const double dX = xr[d] - xr[0];
                const double m_fast = dX * inv_d;
                if (!(m_fast > hi_t) && m_fast >= lo_t && (1.0 - (dd - dX) / dd) > rule.dmin) qual |= 1u << r;   // (the slot test follows)
            }
        }
    }
#pragma unroll 1
    while (qual) {
        const int r = __ffs((int)qual) - 1;
        qual &= qual - 1u;
        float b32 = B32[0];
#pragma unroll
        for (int rr = 1; rr < R; ++rr) b32 = r == rr ? B32[rr] : b32;
        const int i = i0 + r * step;
        const double* xr = x + i;
        asm volatile("" : "+v"(xr));   // a second read, not values kept from the first
        const double dX = xr[d] - xr[0];
        const double m_fast = dX * inv_d;
        const double rs_f = 2.0 * (m_fast * overshoot);
        const double est = rs_f * (rs_f * A - 2.0 * (double)b32);
        const double err = fabs(rs_f) * errB2 + fabs(est) * rel;
        const double lo = est - err, hi = est + err;
        if (!(lo <= slot.hi)) continue;   // (the slot may have moved since the straight-line test)
        if (hi < slot.hi) {   // the new smallest upper bound takes the slot; the old slot cell is parked if it still overlaps
            const double lo_o = slot.lo;
            const int k_o = slot.k, i_o = slot.i;
            slot.hi = hi; slot.lo = lo; slot.k = k; slot.i = i;
            if (lo_o <= hi) park_cell(lead, lo_o, k_o, i_o, rule, widths_c, x_all, env);
        } else {
            park_cell(lead, lo, k, i, rule, widths_c, x_all, env);
        }
    }
}

// append the live lanes' units to the row's list (order inside a list is irrelevant)
__device__ __forceinline__ void push_live(bool live, unsigned int unit, unsigned int* live_count,
                                          unsigned int* list, int lane) {
    const unsigned long long mask = ballot64(live);
    if (mask) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(live_count, (unsigned int)__popcll(mask));
        base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
        if (live) list[base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = unit;
    }
}

// kR sliding dot products of one lane: windows r = 0..kR-1 start XTH samples apart (XTH = 1 is
// the dense T0 grid, 2..kMaxTiledStride the strided grids of long durations, core.py:50-58).
// Sample e[t] feeds window r with template tap t - r*XTH; the taps are wave-uniform (SGPRs).
template <bool IN_LDS, int XTH>
__device__ __forceinline__ void dot_windows(const double* e, const_f64_ptr q, int L, double (&B)[kR]) {
    constexpr int S = (kR - 1) * XTH;
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        // the taps are requested BEFORE the folded samples: load_taps ends in a wait for both, so the
        // scalar-cache latency (about 10 % of the tap loads miss it) overlaps the LDS latency
        const const_f64_ptr qs = q + (t0 - S);  // qs[m] = q_ext[t0 - S + m]
        double taps[kU + S];
#pragma unroll
        for (int m = 0; m < kU + S; ++m) taps[m] = qs[m];
        double x[kU];
        load_taps<IN_LDS>(e + t0, x);
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int r = 0; r < kR; ++r) B[r] = fma(taps[u + S - r * XTH], x[u], B[r]);
    }
}
// the same with per-sample weights: B over e*w with taps q, A over w with taps q^2
template <bool IN_LDS, int XTH>
__device__ __forceinline__ void dot_windows_weighted(const double* ew, const double* w, const_f64_ptr q,
                                                     const_f64_ptr q2, int L, double (&B)[kR], double (&A)[kR]) {
    constexpr int S = (kR - 1) * XTH;
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        const const_f64_ptr qs = q + (t0 - S);
        const const_f64_ptr ps = q2 + (t0 - S);
        double taps[kU + S], x[kU];
#pragma unroll
        for (int m = 0; m < kU + S; ++m) taps[m] = qs[m];
        load_taps<IN_LDS>(ew + t0, x);
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int r = 0; r < kR; ++r) B[r] = fma(taps[u + S - r * XTH], x[u], B[r]);
        // the taps of q^2 reuse the scalar registers
#pragma unroll
        for (int m = 0; m < kU + S; ++m) taps[m] = ps[m];
        load_taps<IN_LDS>(w + t0, x);
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int r = 0; r < kR; ++r) A[r] = fma(taps[u + S - r * XTH], x[u], A[r]);
    }
}

// The same for any stride (long durations of long light curves: xth = int(d * T0_fit_margin),
// core.py:50-55): window r reads its own tap stream q[t - r*xth], still wave-uniform scalars.
template <bool IN_LDS>
__device__ __forceinline__ void dot_windows_rt(const double* e, const_f64_ptr q, int L, int xth, double (&B)[kR]) {
    const int S = (kR - 1) * xth;
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        // all kR tap blocks are requested before the folded samples: one wait per iteration
        double taps[kR][kU];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const const_f64_ptr qr = q + (t0 - r * xth);
#pragma unroll
            for (int u = 0; u < kU; ++u) taps[r][u] = qr[u];
        }
        double x[kU];
        load_taps<IN_LDS>(e + t0, x);
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int u = 0; u < kU; ++u) B[r] = fma(taps[r][u], x[u], B[r]);
    }
}
template <bool IN_LDS>
__device__ __forceinline__ void dot_windows_weighted_rt(const double* ew, const double* w, const_f64_ptr q,
                                                        const_f64_ptr q2, int L, int xth, double (&B)[kR], double (&A)[kR]) {
    const int S = (kR - 1) * xth;
    for (int t0 = 0; t0 < L + S; t0 += kU) {
        double taps[kR][kU], x[kU];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const const_f64_ptr qr = q + (t0 - r * xth);
#pragma unroll
            for (int u = 0; u < kU; ++u) taps[r][u] = qr[u];
        }
        load_taps<IN_LDS>(ew + t0, x);
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int u = 0; u < kU; ++u) B[r] = fma(taps[r][u], x[u], B[r]);
        // the kR tap blocks of q^2 reuse the same scalar registers
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const const_f64_ptr pr = q2 + (t0 - r * xth);
#pragma unroll
            for (int u = 0; u < kU; ++u) taps[r][u] = pr[u];
        }
        load_taps<IN_LDS>(w + t0, x);
#pragma unroll
        for (int r = 0; r < kR; ++r)
#pragma unroll
            for (int u = 0; u < kU; ++u) A[r] = fma(taps[r][u], x[u], A[r]);
    }
}

// Fold `t` at (period, epoch) and produce the STABLE ascending order of the phases
// (numpy.argsort(kind="mergesort"), core.py:119-120 / stats.py:178-179): perm[k] = original index
// of the k-th smallest phase.  Bucket sort: histogram of floor(phase*nb) with LDS atomics, scan,
// scatter (arbitrary order inside a bucket), then an exact rank by (phase, index) inside each
// bucket -- O(n) for the near-uniform phases of a folded time series, correct for any input.
// ph_orig: n doubles, cnt: nb words, idx_tmp/perm: n entries each; all workgroup-visible.
// A bucket that many phases fell into -- a trial period commensurate with the cadence folds a regularly sampled series
// onto a few dozen phase values, each holding N / (P / cadence) points -- is not ranked by counting (b^2 comparisons:
// 29 ms for ONE such period of the Kepler-size grid) but sorted by the whole workgroup: a bitonic network on
// (phase, index) in its normalised form (every compare-exchange puts the smaller key at the lower position, so the
// power-of-two padding is implicit: a partner beyond the bucket's end is +infinity and the exchange is skipped).
// b log^2 b / 4 exchanges, a workgroup barrier per step.  Keys either staged in LDS (key_l / idx_l: series in HBM) or
// read through the index (LDS-resident series: everything is in LDS already).  `out[r]` = r-th index of the bucket.
constexpr int kBigBucket = 48;   // buckets beyond this size leave the counting rank
#ifndef TLS_WAVE_SORT_MAX
#define TLS_WAVE_SORT_MAX 2048
#endif
constexpr int kWaveSortMax = TLS_WAVE_SORT_MAX;  // piled-up buckets up to this size are sorted by ONE wave (the waves take buckets in turn)
// WAVE: the calling wave sorts the bucket alone (lanes over the exchanges, wave-level LDS ordering between the steps);
// otherwise all threads of the workgroup call this (workgroup barriers).
template <typename IdxT, bool STAGED, bool WAVE>
__device__ __forceinline__ void sort_big_bucket(const double* ph_orig, IdxT* idx_seg, int m, double* key_l, unsigned int* idx_l,
                                                IdxT* out) {
    const int tid = WAVE ? (int)(threadIdx.x & (kWave - 1)) : (int)threadIdx.x;
    const int nt = WAVE ? kWave : (int)blockDim.x;
    auto step_sync = [&]() { if constexpr (WAVE) wave_sync(); else wg_sync(); };
    if constexpr (STAGED) {
        for (int j = tid; j < m; j += nt) { const unsigned int i = (unsigned int)idx_seg[j]; idx_l[j] = i; key_l[j] = ph_orig[i]; }
    }
    step_sync();
    int lg = 0;
    while ((1 << lg) < m) ++lg;
    const int half_pairs = (1 << lg) >> 1;
    auto exchange = [&](int x, int y) {   // x < y
        if (y >= m) return;
        if constexpr (STAGED) {
            const double kx = key_l[x], ky = key_l[y];
            const unsigned int ix = idx_l[x], iy = idx_l[y];
            if (kx > ky || (kx == ky && ix > iy)) { key_l[x] = ky; key_l[y] = kx; idx_l[x] = iy; idx_l[y] = ix; }
        } else {
            const IdxT ix = idx_seg[x], iy = idx_seg[y];
            const double kx = ph_orig[(int)ix], ky = ph_orig[(int)iy];
            if (kx > ky || (kx == ky && ix > iy)) { idx_seg[x] = iy; idx_seg[y] = ix; }
        }
    };
    for (int lk = 1; lk <= lg; ++lk) {           // merge blocks of k = 2^lk
        const int lh = lk - 1;                   // half = 2^lh
        for (int i = tid; i < half_pairs; i += nt) {
            const int blk = i >> lh, pos = i & ((1 << lh) - 1);
            exchange((blk << lk) + pos, (blk << lk) + ((1 << lk) - 1 - pos));
        }
        step_sync();
        for (int lj = lk - 2; lj >= 0; --lj) {   // j = 2^lj
            for (int i = tid; i < half_pairs; i += nt) {
                const int x = ((i >> lj) << (lj + 1)) + (i & ((1 << lj) - 1));
                exchange(x, x + (1 << lj));
            }
            step_sync();
        }
    }
    for (int j = tid; j < m; j += nt) out[j] = STAGED ? (IdxT)idx_l[j] : idx_seg[j];
    step_sync();
}

template <typename IdxT>
__device__ __forceinline__ void fold_and_sort(const double* t, int n, double period, double epoch,
                                              double* ph_orig, unsigned int* cnt, int nb, IdxT* idx_tmp,
                                              IdxT* perm, unsigned int* wsum, PhaseClock& pc,
                                              unsigned int* big_list = nullptr, int big_cap = 0,
                                              double* stage_key = nullptr, unsigned int* stage_idx = nullptr, int stage_cap = 0) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const double nb_d = (double)nb;
    for (int b = tid; b < nb; b += nt) cnt[b] = 0;
    // [0]: number of big buckets, then (first slot, size) pairs -- the size is recorded by the registering thread from
    // the LDS counters, so that every thread of the workgroup takes the same branches on it below (a size re-derived
    // from idx_tmp in global memory while other waves reorder that segment need not be the same for every reader)
    if (tid == 0 && big_cap > 0) big_list[0] = 0u;
    const int list_slots = (big_cap - 1) / 2;
    wg_sync();
    {
        // kF time stamps per step: their L2 round trips overlap (the compiler keeps a global load behind the LDS atomic
        // of the step before it)
        constexpr int kF = TLS_FOLD_DEPTH;
        for (int i0 = tid; i0 < n; i0 += kF * nt) {
            double tv[kF];
#pragma unroll
            for (int g = 0; g < kF; ++g) tv[g] = t[i0 + g * nt < n ? i0 + g * nt : i0];
#pragma unroll
            for (int g = 0; g < kF; ++g) {
                const int i = i0 + g * nt;
                if (i < n) {
                    const double ph = fold_phase(tv[g], period, epoch);
                    ph_orig[i] = ph;
                    atomicAdd(&cnt[bucket_of(ph, nb_d, nb)], 1u);
                }
            }
        }
    }
    wg_sync();
    pc.mark(0);
    block_exclusive_scan(cnt, nb, wsum);
    pc.mark(1);
    for (int i = tid; i < n; i += nt) {
        int b = bucket_of(ph_orig[i], nb_d, nb);
        unsigned int slot = atomicAdd(&cnt[b], 1u);  // arbitrary order inside a bucket...
        idx_tmp[slot] = (IdxT)i;
    }
    wg_sync();
    pc.mark(2);
    // ...made deterministic here: rank by (phase, original index) inside the bucket.
    // cnt[b] now holds the END of bucket b.
    // (a thread takes the points in index order: phase and bucket of a point come from a sequential read, only the
    // bucket's bounds -- and, where a bucket holds several points, its members -- are dependent reads)
#if TLS_RANK_BY_INDEX
#pragma unroll 2
    for (int i = tid; i < n; i += nt) {
#else
    for (int s = tid; s < n; s += nt) {
        const int i = (int)idx_tmp[s];
#endif
        const double ph = ph_orig[i];
        const int b = bucket_of(ph, nb_d, nb);
        const int lo = b ? (int)cnt[b - 1] : 0, hi = (int)cnt[b];
        if (hi - lo > kBigBucket && big_cap > 0) {
            // piled-up phases: the bucket's first slot registers it for the workgroup sort below; if the list is
            // full (more than big_cap such buckets) the counting rank does it after all
            bool listed = true;
#if TLS_RANK_BY_INDEX
            if ((int)idx_tmp[lo] == i) {
#else
            if (s == lo) {
#endif
                const unsigned int at = atomicAdd(&big_list[0], 1u);
                if (at < (unsigned int)list_slots) { big_list[1 + 2 * at] = (unsigned int)lo; big_list[2 + 2 * at] = (unsigned int)(hi - lo); }
            }
            (void)listed;
            continue;
        }
        int rank = 0;
        if (hi - lo > 1) {   // (most buckets of a folded time series hold one point)
            for (int s2 = lo; s2 < hi; ++s2) {
                const int i2 = (int)idx_tmp[s2];
                const double ph2 = ph_orig[i2];
                rank += (ph2 < ph || (ph2 == ph && i2 < i)) ? 1 : 0;
            }
        }
        perm[lo + rank] = (IdxT)i;
    }
    wg_sync();
    if (big_cap > 0) {
        const int n_big_all = (int)big_list[0];
        const int n_big = n_big_all < list_slots ? n_big_all : list_slots;
        if (stage_cap == 0) {
            // everything is in LDS already (resident series): buckets of moderate size go one to a wave, the waves side
            // by side, sorted in place through the index; larger ones afterwards by the whole workgroup
            const int wave_id = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave)), n_waves = nt / kWave;
            for (int q = wave_id; q < n_big; q += n_waves) {
                const int lo = __builtin_amdgcn_readfirstlane((int)big_list[1 + 2 * q]);
                const int m = __builtin_amdgcn_readfirstlane((int)big_list[2 + 2 * q]);
                if (m <= kWaveSortMax) sort_big_bucket<IdxT, false, true>(ph_orig, idx_tmp + lo, m, nullptr, nullptr, perm + lo);
            }
            wg_sync();
            for (int q = 0; q < n_big; ++q) {
                const int lo = (int)big_list[1 + 2 * q];
                const int m = (int)big_list[2 + 2 * q];
                if (m > kWaveSortMax) sort_big_bucket<IdxT, false, false>(ph_orig, idx_tmp + lo, m, nullptr, nullptr, perm + lo);
            }
        } else {
            // series in HBM: as many buckets as fit the LDS staging area are sorted TOGETHER by the workgroup (one
            // barrier per step of the network for the whole batch instead of one per bucket and step)
            int q0 = 0;
            while (q0 < n_big) {
                int count = 0, used = 0, lg_max = 0;
                // (uniform: every thread walks the same list)
                int lo_b[16], m_b[16], off_b[16];
                while (q0 + count < n_big && count < 16) {
                    const int lo = (int)big_list[1 + 2 * (q0 + count)];
                    const int m = (int)big_list[2 + 2 * (q0 + count)];
                    if (m > stage_cap) break;                     // does not fit at all: alone, below
                    if (used + m > stage_cap) break;
                    lo_b[count] = lo; m_b[count] = m; off_b[count] = used; used += m;
                    int lg = 0; while ((1 << lg) < m) ++lg;
                    lg_max = lg > lg_max ? lg : lg_max;
                    ++count;
                }
                if (count == 0) {   // a bucket larger than the staging area: in place, through global memory
                    const int lo = (int)big_list[1 + 2 * q0];
                    sort_big_bucket<IdxT, false, false>(ph_orig, idx_tmp + lo, (int)big_list[2 + 2 * q0], nullptr, nullptr, perm + lo);
                    ++q0;
                    continue;
                }
                for (int g = 0; g < count; ++g)
                    for (int j = tid; j < m_b[g]; j += nt) {
                        const unsigned int i = (unsigned int)idx_tmp[lo_b[g] + j];
                        stage_idx[off_b[g] + j] = i; stage_key[off_b[g] + j] = ph_orig[i];
                    }
                wg_sync();
                const int half_pairs = (1 << lg_max) >> 1;
                auto exchange = [&](int g, int x, int y) {
                    if (y >= m_b[g]) return;
                    double* key_l = stage_key + off_b[g];
                    unsigned int* idx_l = stage_idx + off_b[g];
                    const double kx = key_l[x], ky = key_l[y];
                    const unsigned int ix = idx_l[x], iy = idx_l[y];
                    if (kx > ky || (kx == ky && ix > iy)) { key_l[x] = ky; key_l[y] = kx; idx_l[x] = iy; idx_l[y] = ix; }
                };
                for (int lk = 1; lk <= lg_max; ++lk) {
                    const int lh = lk - 1;
                    for (int i = tid; i < count * half_pairs; i += nt) {
                        const int g = i >> (lg_max - 1), ii = i & (half_pairs - 1);
                        const int blk = ii >> lh, pos = ii & ((1 << lh) - 1);
                        exchange(g, (blk << lk) + pos, (blk << lk) + ((1 << lk) - 1 - pos));
                    }
                    wg_sync();
                    for (int lj = lk - 2; lj >= 0; --lj) {
                        for (int i = tid; i < count * half_pairs; i += nt) {
                            const int g = i >> (lg_max - 1), ii = i & (half_pairs - 1);
                            const int x = ((ii >> lj) << (lj + 1)) + (ii & ((1 << lj) - 1));
                            exchange(g, x, x + (1 << lj));
                        }
                        wg_sync();
                    }
                }
                for (int g = 0; g < count; ++g)
                    for (int j = tid; j < m_b[g]; j += nt) perm[lo_b[g] + j] = (IdxT)stage_idx[off_b[g] + j];
                wg_sync();
                q0 += count;
            }
        }
        if (n_big_all > n_big) {
            // more piled-up buckets than list slots: those beyond the list are ranked by counting after all
            // (a listed bucket's segment of perm is complete and is not touched: its first slot is in the list)
            for (int s = tid; s < n; s += nt) {
                const int i = (int)idx_tmp[s];
                const double ph = ph_orig[i];
                const int b = bucket_of(ph, nb_d, nb);
                const int lo = b ? (int)cnt[b - 1] : 0, hi = (int)cnt[b];
                if (hi - lo <= kBigBucket) continue;
                bool in_list = false;
                for (int q = 0; q < n_big; ++q) in_list |= (int)big_list[1 + 2 * q] == lo;
                if (in_list) continue;
                int rank = 0;
                for (int s2 = lo; s2 < hi; ++s2) {
                    const int i2 = (int)idx_tmp[s2];
                    const double ph2 = ph_orig[i2];
                    rank += (ph2 < ph || (ph2 == ph && i2 < i)) ? 1 : 0;
                }
                perm[lo + rank] = (IdxT)i;
            }
            wg_sync();
        }
    }
    pc.mark(3);
}

// Pointer parameters typed by address space (a generic pointer parameter makes every access a FLAT instruction,
// counted on vmcnt AND lgkmcnt); the LDS window travels as its 32-bit LDS address.
template <typename T> using global_ptr = __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ T* from_global_arg(global_ptr<T> p) {   // uniform again, then generic for the callee's code
    const unsigned long long b = (unsigned long long)(uintptr_t)p;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffull));
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b >> 32));
    return (T*)(global_ptr<T>)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
template <typename T>
__device__ __forceinline__ unsigned int lds_address(T* p) {
    return (unsigned int)(uintptr_t)(__attribute__((address_space(3))) T*)p;
}
// ---------------------------------------------------------------------------------------
// The same order for a series that lives in HBM (tiled variant): a two-level sort whose HBM accesses
// are all sequential.  fold_and_sort scatters and ranks through global memory with one random
// 4-8 B access per point and step, i.e. a whole 64 B sector each -- most of the HBM traffic of the
// Kepler-size configuration.  Here the points are first partitioned into coarse phase bins (a chunk
// of points is bucketed in LDS and leaves as one contiguous segment per bin), then every wavefront
// sorts one bin entirely inside its private LDS window and writes its piece of the permutation.
//   g_ph[n], g_idx[n]: HBM scratch (phase / original index, grouped by coarse bin); perm[n]: result.
//   lds: at least sort2_lds_bytes(n) bytes.  Returns false (all threads alike) when a coarse bin
//   overflows its LDS window (phases piled up, e.g. a period commensurate with the cadence): the
//   caller then falls back to fold_and_sort.
#ifndef TLS_SORT2_CHUNK
#define TLS_SORT2_CHUNK 10240
#endif
constexpr int kSort2Chunk = TLS_SORT2_CHUNK;    // points bucketed per pass-1 round
constexpr int kSort2BinCap = 384;     // points one wavefront can sort in its LDS window
#ifndef TLS_SORT2_FINE
#define TLS_SORT2_FINE 384
#endif
constexpr int kSort2Fine = TLS_SORT2_FINE;   // fine buckets of a bin (<= kSort2BinCap: the counters share its window slot)
// target points per coarse bin: 280 for long series (round 4, Kepler full grid: 128 233.8 ms, 160 226.4, 200 227.4, 240 221.6,
// 280 219.3, 320 218.5; the cap is 384), 160 below 50 000 points (TESS size: 70 bins for 16 wavefronts fill the last round of
// pass 2 badly -- 2.83 against 2.81 ms, a 307-period block 5 % slower)
#ifndef TLS_SORT2_BIN_MEAN
#define TLS_SORT2_BIN_MEAN 280
#endif
constexpr int kSort2BinMeanLong = TLS_SORT2_BIN_MEAN, kSort2BinMeanShort = 160, kSort2LongFrom = 50000;
constexpr int kSort2MaxBins = 1024;
__host__ __device__ constexpr int sort2_bins(int n) {
    const int mean = n >= kSort2LongFrom ? kSort2BinMeanLong : kSort2BinMeanShort;
    return (n + mean - 1) / mean < kSort2MaxBins ? (n + mean - 1) / mean : kSort2MaxBins;
}
__host__ __device__ constexpr long long sort2_lds_bytes(int n, int threads = 1024) {
    // counters (4 arrays of bins+1 words) + the larger of the pass-1 staging and the pass-2 windows
    const long long counters = 4LL * 4 * (sort2_bins(n) + 1);
    const long long chunk = (long long)(kSort2Chunk / 1024) * threads < kSort2Chunk ? (long long)(kSort2Chunk / 1024) * threads : kSort2Chunk;
    const long long stage = 10LL * chunk;                                            // record (u64) + bin (u16) per point
    const long long windows = (long long)(threads / kWave) * ((8 + 4) * kSort2BinCap + 16);  // record, counter (+ end)
    return (counters + 15) / 16 * 16 + (stage > windows ? stage : windows);
}

//   y_gather (may be null): the folded flux y[perm[k]] is written over g_ph[k] on the way (and
//   w_gather -> w_out likewise), which saves the separate gather pass of a single light curve.
//   g_rec[n]: HBM scratch, the partitioned points as 8-byte records (32-bit phase key << 32 | original index),
//   grouped by coarse bin; f_out[n] (y_gather given): the folded flux.  f_out may BE g_rec: a bin's records are
//   in registers before its flux is stored.
template <bool HAS_W>
__device__ __forceinline__ bool fold_and_sort_tiled(const double* t, int n, double period, double epoch,
                                                    unsigned long long* g_rec, unsigned int* perm, unsigned char* lds,
                                                    PhaseClock& pc, const double* y_gather, const double* w_gather,
                                                    double* f_out, double* w_out_g, unsigned long long* dbg_check) {
    // (Carrying the flux WITH the points -- pass 1 reading y beside t and writing it into the bin segments,
    // pass 2 reading it back with the phases instead of gathering y[index] -- was built and measured: the
    // gather disappears, but the partition pass grows by more (Kepler-size sample 6.5 vs 5.65 ms).)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));   // opaque: indices derived from it are formed here, not hoisted out of the
                                    // period loop and spilled (a spill reload between two loads waits for vmcnt(0))
    const int nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int B = sort2_bins(n);
    const double B_d = (double)B;
    constexpr int kPerStage = kSort2Chunk / 1024;   // points per thread and pass-1 round (as kPer below)
    unsigned int* g_start = reinterpret_cast<unsigned int*>(lds);   // [B+1] first output slot of every bin
    unsigned int* g_cur = g_start + (B + 1);                        // [B+1] pass 1: next free slot of the bin
    unsigned int* l_cnt = g_cur + (B + 1);                          // [B+1] pass 1: points of the bin in this chunk
    unsigned int* l_start = l_cnt + (B + 1);                        // [B+1]
    unsigned char* area = lds + (4 * 4 * (B + 1) + 15) / 16 * 16;

    // ---- counts per coarse bin ------------------------------------------------------------------
    for (int b = tid; b <= B; b += nt) { g_start[b] = 0; l_cnt[b] = 0; }
    wg_sync();
    for (int i0 = 0; i0 < n; i0 += 8 * nt) {   // eight time stamps in flight per thread; whole waves take part
        double tv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) tv[j] = i0 + tid + j * nt < n ? t[i0 + tid + j * nt] : 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            bin_inc<false>(g_start, i0 + tid + j * nt < n ? bucket_of(fold_phase(tv[j], period, epoch), B_d, B) : -1);
    }
    wg_sync();
    // exclusive scan over the bins by wave 0 (B <= 1024: 16 per lane), and the overflow test
    if (wave == 0) {
        const int per = (B + kWave - 1) / kWave;
        const int lo = lane * per < B ? lane * per : B, hi = lo + per < B ? lo + per : B;
        unsigned int local = 0, biggest = 0;
        for (int b = lo; b < hi; ++b) { const unsigned int c = g_start[b]; local += c; biggest = c > biggest ? c : biggest; }
        unsigned int incl = local;
#pragma unroll
        for (int dlt = 1; dlt < kWave; dlt <<= 1) {
            const unsigned int o = __shfl_up(incl, dlt, kWave);
            if (lane >= dlt) incl += o;
        }
#pragma unroll
        for (int delta = kWave / 2; delta > 0; delta >>= 1) {
            const unsigned int o = __shfl_down(biggest, delta, kWave);
            biggest = o > biggest ? o : biggest;
        }
        unsigned int run = incl - local;
        for (int b = lo; b < hi; ++b) { const unsigned int c = g_start[b]; g_start[b] = run; g_cur[b] = run; run += c; }
        if (lane == kWave - 1) g_start[B] = run;
        if (lane == 0) g_cur[B] = biggest;   // parked here for everybody to read
    }
    wg_sync();
    // A bin beyond a wavefront's window (phases piled up: a period commensurate with the cadence folds a regularly sampled
    // series onto P / cadence phase values) is skipped by pass 2 and sorted by the WORKGROUP afterwards, in the staging
    // area of pass 1: exact phase and index of every point side by side, a bitonic network on (phase, index) -- the order
    // of the stable sort --, several bins per run of the network.  Only a bin beyond the staging area itself (a period of
    // fewer than ~9 cadences on a Kepler-size series) still sends the period to the general sort.
    const unsigned int biggest_bin = g_cur[B];
    const int stage_pts = (int)((kPerStage * nt < kSort2Chunk ? kPerStage * nt : kSort2Chunk) * 10LL / 12);   // 12 B a point
    if (biggest_bin > (unsigned int)stage_pts) { wg_sync(); return false; }
    pc.mark(0);

    // ---- pass 1: partition, one chunk of points per round ---------------------------------------
    {
        unsigned long long* st_rec = reinterpret_cast<unsigned long long*>(area);
        constexpr int kPer = kSort2Chunk / 1024;   // points per thread and round (1024-thread workgroups)
        const int chunk = kPer * nt < kSort2Chunk ? kPer * nt : kSort2Chunk;
        unsigned short* st_bin = reinterpret_cast<unsigned short*>(st_rec + chunk);
        for (int c0 = 0; c0 < n; c0 += chunk) {
            const int cn = n - c0 < chunk ? n - c0 : chunk;
            double ph[kPer];
            unsigned int rank[kPer];
            int bin[kPer];
#pragma unroll
            for (int e = 0; e < kPer; ++e) {
                const int k = tid + e * nt;
                ph[e] = k < cn ? t[c0 + k] : 0.0;      // the time stamp for now: all reads of the round in flight together
            }
#pragma unroll
            for (int e = 0; e < kPer; ++e) {
                const int k = tid + e * nt;
                ph[e] = fold_phase(ph[e], period, epoch);
                bin[e] = k < cn ? bucket_of(ph[e], B_d, B) : -1;
                rank[e] = bin_inc<true>(l_cnt, bin[e]);
            }
            lds_barrier();
            pc.mark(28);
            if (wave == 0) {   // chunk-local exclusive scan; reserve the bins' output segments
                const int per = (B + kWave - 1) / kWave;
                const int lo = lane * per < B ? lane * per : B, hi = lo + per < B ? lo + per : B;
                unsigned int local = 0;
                for (int b = lo; b < hi; ++b) local += l_cnt[b];
                unsigned int run = wave_inclusive_sum_u32(local) - local;
                for (int b = lo; b < hi; ++b) { l_start[b] = run; run += l_cnt[b]; }
            }
            lds_barrier();
            pc.mark(29);
#pragma unroll
            for (int e = 0; e < kPer; ++e) {
                if (bin[e] >= 0) {
                    const unsigned int s = l_start[bin[e]] + rank[e];
                    st_rec[s] = ((unsigned long long)phase_key(ph[e]) << 32) | (unsigned long long)(unsigned int)(c0 + tid + e * nt);
                    st_bin[s] = (unsigned short)bin[e];
                }
            }
            lds_barrier();
            pc.mark(30);
            // every bin's points of this chunk leave as one contiguous segment
            for (int sidx = tid; sidx < cn; sidx += nt) {
                const int b = st_bin[sidx];
                g_rec[g_cur[b] + ((unsigned int)sidx - l_start[b])] = st_rec[sidx];
            }
            lds_barrier();
            pc.mark(31);
            for (int b = tid; b < B; b += nt) { g_cur[b] += l_cnt[b]; l_cnt[b] = 0; }
            lds_barrier();
        }
    }
    wg_sync();   // (the barriers inside the rounds order LDS only: the partitioned points are in memory HERE)
    pc.mark(2);

    // ---- pass 2: one coarse bin per wavefront, sorted inside its LDS window -----------------------
    // Every lane keeps its (up to kE) points of the bin in registers and ranks them itself: the points are
    // counted into fine buckets, laid out bucket by bucket (phase and index side by side), and a point's rank
    // is its bucket's start plus the members of its bucket that precede it in (phase, index) order -- the
    // order of numpy's stable sort.  The s-th member of all buckets is read in one batch, so the LDS
    // latency is paid once per step, not per point and step.  Results leave through the window in rank order,
    // i.e. as coalesced stores.  The kE slots of a lane are walked only as far as the bin is filled (160 points
    // on average, the window holds 384), under wave-uniform branches.
    {
        unsigned char* win = area + (size_t)wave * ((8 + 4) * kSort2BinCap + 16);
        unsigned long long* s_rec = reinterpret_cast<unsigned long long*>(win);   // records, bucket by bucket; then the results
        unsigned int* w_cnt = reinterpret_cast<unsigned int*>(s_rec + kSort2BinCap);
        constexpr int kE = kSort2BinCap / kWave;   // entries per lane
        constexpr int kF = kSort2Fine / kWave;     // fine-bucket counters per lane
        // the points of a wave's NEXT bin are requested before the current one is sorted: their HBM
        // latency hides behind the sort
        unsigned long long nx_rec[kE];
        {
            const int b = wave;
            const unsigned int first = b < B ? g_start[b] : 0u;
            const int m = b < B ? (int)(g_start[b + 1] - first) : 0;
#pragma unroll
            for (int e = 0; e < kE; ++e) nx_rec[e] = lane + e * kWave < m ? g_rec[first + lane + e * kWave] : 0ull;
        }
        // fine bucket of a key inside its coarse bin: INTEGER arithmetic -- points with equal keys must land in the
        // same bucket, and a floating-point expression evaluated in several unrolled places need not round alike
        // (contraction is per instance): two of four equal phases on a bucket boundary were ordered by bucket.
        const unsigned int fine_per_key = (unsigned int)kSort2Fine * (unsigned int)B;   // buckets per 2^32 keys
        const unsigned int keys_per_bin = 0xffffffffu / (unsigned int)B;                // (rounded down: any value common to the bin serves)
        for (int b0 = 0; b0 < B; b0 += nw) {
            const int b = b0 + wave;
            const unsigned int first = (unsigned int)__builtin_amdgcn_readfirstlane((int)(b < B ? g_start[b] : 0u));
            const int m_all = __builtin_amdgcn_readfirstlane(b < B ? (int)(g_start[b + 1] - first) : 0);
            const int m = m_all > kSort2BinCap ? 0 : m_all;   // (a piled-up bin is left to the workgroup: below)
            const int e_used = (m + kWave - 1) / kWave;   // slots per lane that hold a point in some lane
            const int bn = b + nw;
            const unsigned int first_n = (unsigned int)__builtin_amdgcn_readfirstlane((int)(bn < B ? g_start[bn] : 0u));
            const int m_n = __builtin_amdgcn_readfirstlane(bn < B ? (int)(g_start[bn + 1] - first_n) : 0);
#ifdef TLS_DEBUG_CHECKS
            if (dbg_check && !(m <= kSort2BinCap) && lane == 0) atomicAdd(&dbg_check[kChkSortWindow], 1ull);
#endif
            unsigned long long rec[kE];
            double yv[kE];
            [[maybe_unused]] double wv[HAS_W ? kE : 1];
            int fb[kE];   // fine bucket of a point inside its coarse bin: monotone in the key
#pragma unroll
            for (int e = 0; e < kE; ++e) rec[e] = nx_rec[e];
#pragma unroll
            for (int e = 0; e < kF; ++e) w_cnt[lane + e * kWave] = 0;
            if (y_gather) {
#pragma unroll
                for (int e = 0; e < kE; ++e) if (e < e_used) yv[e] = y_gather[(unsigned int)rec[e]];       // (index 0 in the unused lanes)
                if constexpr (HAS_W) {
#pragma unroll
                    for (int e = 0; e < kE; ++e) if (e < e_used) wv[e] = w_gather[(unsigned int)rec[e]];
                }
            }
#pragma unroll
            for (int e = 0; e < kE; ++e)
                if (e * kWave < m_n) nx_rec[e] = lane + e * kWave < m_n ? g_rec[first_n + lane + e * kWave] : 0ull;
            const unsigned int key_lo = (unsigned int)(b < B ? b : 0) * keys_per_bin;   // about the smallest key of the bin
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                fb[e] = 0;
                if (e < e_used) {
                    const unsigned int key = (unsigned int)(rec[e] >> 32);
                    const unsigned int f = key > key_lo ? __umulhi(key - key_lo, fine_per_key) : 0u;   // monotone in the key
                    fb[e] = f < (unsigned int)(kSort2Fine - 1) ? (int)f : kSort2Fine - 1;
                }
            }
            wave_lds_sync();
            unsigned int tk[kE];   // ticket of the point inside its fine bucket (arrival order: arbitrary)
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                tk[e] = 0u;
                if (e < e_used) { if (lane + e * kWave < m) tk[e] = atomicAdd(&w_cnt[fb[e]], 1u); }
            }
            wave_lds_sync();
            {   // exclusive scan of the kSort2Fine counters of this window: kF consecutive per lane, the
                // wave's part on the DPP crossbar (no LDS round trips)
                unsigned int c[kF], local = 0;
#pragma unroll
                for (int e = 0; e < kF; ++e) { c[e] = w_cnt[lane * kF + e]; local += c[e]; }
                unsigned int run = wave_inclusive_sum_u32(local) - local;
#pragma unroll
                for (int e = 0; e < kF; ++e) { w_cnt[lane * kF + e] = run; run += c[e]; }
                if (lane == kWave - 1) w_cnt[kSort2Fine] = run;   // the end of the last bucket
            }
            wave_lds_sync();
            int lo[kE], len[kE], rank[kE], longest = 0;
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                lo[e] = 0; len[e] = 0; rank[e] = 0;
                if (e < e_used) {
                    const bool have = lane + e * kWave < m;
                    lo[e] = have ? (int)w_cnt[fb[e]] : 0;
                    len[e] = have ? (int)w_cnt[fb[e] + 1] - lo[e] : 0;
                }
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) s_rec[lo[e] + (int)tk[e]] = rec[e]; }
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < kE; ++e) longest = len[e] > longest ? len[e] : longest;
            constexpr int kHalf = kE / 2;   // the batch of a step in two halves: fewer registers in flight
            for (int sidx = 0; ballot64(sidx < longest) != 0ull; ++sidx) {
#pragma unroll
                for (int h = 0; h < kE; h += kHalf) {
                    if (h < e_used) {
                        unsigned long long rec2[kHalf];
#pragma unroll
                        for (int e = 0; e < kHalf; ++e) rec2[e] = s_rec[sidx < len[h + e] ? lo[h + e] + sidx : 0];
                        bool tie = false;
#pragma unroll
                        for (int e = 0; e < kHalf; ++e) {
                            const bool in = sidx < len[h + e];
                            const bool same_key = (unsigned int)(rec2[e] >> 32) == (unsigned int)(rec[h + e] >> 32);
                            rank[h + e] += (in && !same_key && rec2[e] < rec[h + e]) ? 1 : 0;
                            tie |= in && same_key && rec2[e] != rec[h + e];   // (its own entry aside)
                        }
                        if (ballot64(tie) != 0ull) {
                            // equal 32-bit keys (rare): the exact phases decide, then the index -- a stable sort
#pragma unroll
                            for (int e = 0; e < kHalf; ++e) {
                                const bool in = sidx < len[h + e];
                                const bool same_key = (unsigned int)(rec2[e] >> 32) == (unsigned int)(rec[h + e] >> 32);
                                if (in && same_key && rec2[e] != rec[h + e]) {
                                    const unsigned int i1 = (unsigned int)rec[h + e], i2 = (unsigned int)rec2[e];
                                    const double p1 = fold_phase(t[i1], period, epoch), p2 = fold_phase(t[i2], period, epoch);
                                    rank[h + e] += (p2 < p1 || (p2 == p1 && i2 < i1)) ? 1 : 0;
                                }
                            }
                        }
                    }
                }
            }
            // One wait per round, here: the flux is needed now, and the next bin's points -- requested at the
            // top of this round -- have had the whole sort to arrive.  The stores below then have the whole
            // NEXT round: gfx9 counts loads and stores on one counter, so a wait for a load issued before
            // them (the compiler's vmcnt(0) whenever stores are pending) would expose their round trip.
            vmem_wait_all();
            wave_lds_sync();   // every lane is done with the bucket lists: the window now carries the results
            // in rank order through the window, out as coalesced stores (the bin's records are all in
            // registers: their slab entries may go)
            if (perm) {
                unsigned int* s_out = reinterpret_cast<unsigned int*>(s_rec);
#pragma unroll
                for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) s_out[lo[e] + rank[e]] = (unsigned int)rec[e]; }
                wave_lds_sync();
#pragma unroll
                for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) perm[first + lane + e * kWave] = s_out[lane + e * kWave]; }
                wave_lds_sync();
            }
            if (y_gather) {
                double* s_out = reinterpret_cast<double*>(s_rec);
#pragma unroll
                for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) s_out[lo[e] + rank[e]] = yv[e]; }
                wave_lds_sync();
#pragma unroll
                for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) f_out[first + lane + e * kWave] = s_out[lane + e * kWave]; }
                if constexpr (HAS_W) {
                    wave_lds_sync();
#pragma unroll
                    for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) s_out[lo[e] + rank[e]] = wv[e]; }
                    wave_lds_sync();
#pragma unroll
                    for (int e = 0; e < kE; ++e) if (e < e_used) { if (lane + e * kWave < m) w_out_g[first + lane + e * kWave] = s_out[lane + e * kWave]; }
                }
            }
            wave_lds_sync();
        }
    }
    wg_sync();
    if (biggest_bin > (unsigned int)kSort2BinCap) {
        // ---- the piled-up bins, by the whole workgroup ------------------------------------------------
        unsigned int* big_list = l_cnt;   // [0] count, then the bins (pass 1's chunk counters are idle; l_start follows them: 2 (B + 1) words)
        if (tid == 0) big_list[0] = 0u;
        wg_sync();
        for (int b = tid; b < B; b += nt)
            if (g_start[b + 1] - g_start[b] > (unsigned int)kSort2BinCap) big_list[1 + atomicAdd(&big_list[0], 1u)] = (unsigned int)b;
        wg_sync();
        const int n_big = (int)big_list[0];
        double* stage_key = reinterpret_cast<double*>(area);
        unsigned int* stage_idx = reinterpret_cast<unsigned int*>(stage_key + stage_pts);
        // A run of the network takes bins of (nearly) one size: every bin gets a slot of 2^lg entries, the entries behind
        // its last point are +infinity -- so an exchange needs no table and no bounds test, the slot of entry x is x >> lg.
        // (The first version kept offsets and lengths of up to 16 bins in per-thread arrays: indexed at run time they
        // live in scratch memory, two HBM round trips per exchange -- 2.2 ms for a period of 66.5 cadences.)
        int q0 = 0;
        while (q0 < n_big) {
            int lg = 0;
            {
                const int b = (int)big_list[1 + q0];
                const int m = (int)(g_start[b + 1] - g_start[b]);
                while ((1 << lg) < m) ++lg;
            }
            // as many of the following bins as fit a slot of this size and the staging area
            int count = 0;
            while (q0 + count < n_big && ((count + 1) << lg) <= stage_pts) {
                const int b = (int)big_list[1 + q0 + count];
                if ((int)(g_start[b + 1] - g_start[b]) > (1 << lg)) break;
                ++count;
            }
            if (count == 0) count = 1;   // (a bin whose slot exceeds the area but whose points fit: biggest_bin <= stage_pts; it runs alone, unpadded tail tested below)
            const int slot_len = 1 << lg, total = count << lg;
            const bool padded = total <= stage_pts;
            for (int x = tid; x < (padded ? total : stage_pts); x += nt) {
                const int g = x >> lg, j = x & (slot_len - 1);
                const int b = (int)big_list[1 + q0 + g];
                const unsigned int first = g_start[b], m = g_start[b + 1] - first;
                unsigned int i = 0xffffffffu;
                double key = INFINITY;
                if ((unsigned int)j < m) { i = (unsigned int)g_rec[first + j]; key = fold_phase(t[i], period, epoch); }
                stage_idx[x] = i; stage_key[x] = key;
            }
            wg_sync();   // (also: every record of the run has been read -- f_out may be g_rec)
            const int limit = padded ? total : stage_pts;   // entries that exist
            auto exchange4 = [&](int x0, int x1, int x2, int x3, int d0, int d1, int d2, int d3, int valid) {
                // four independent compare-exchanges (x, x + d): all reads first, then the stores
                const int xs[4] = {x0, x1, x2, x3}, ds[4] = {d0, d1, d2, d3};
                double kx[4], ky[4]; unsigned int ix[4], iy[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool on = e < valid && xs[e] + ds[e] < limit;
                    kx[e] = on ? stage_key[xs[e]] : 0.0; ky[e] = on ? stage_key[xs[e] + ds[e]] : 1.0;
                    ix[e] = on ? stage_idx[xs[e]] : 0u; iy[e] = on ? stage_idx[xs[e] + ds[e]] : 1u;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (kx[e] > ky[e] || (kx[e] == ky[e] && ix[e] > iy[e])) {
                        stage_key[xs[e]] = ky[e]; stage_key[xs[e] + ds[e]] = kx[e];
                        stage_idx[xs[e]] = iy[e]; stage_idx[xs[e] + ds[e]] = ix[e];
                    }
                }
            };
            const int pairs = total >> 1;
            for (int lk = 1; lk <= lg; ++lk) {
                const int lh = lk - 1;
                for (int i0 = tid; i0 < pairs; i0 += 4 * nt) {
                    int xs[4], ds[4], valid = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = i0 + e * nt;
                        const int blk = i >> lh, pos = i & ((1 << lh) - 1);
                        xs[e] = (blk << lk) + pos; ds[e] = (1 << lk) - 1 - 2 * pos;
                        if (i < pairs) valid = e + 1;
                    }
                    exchange4(xs[0], xs[1], xs[2], xs[3], ds[0], ds[1], ds[2], ds[3], valid);
                }
                lds_barrier();
                for (int lj = lk - 2; lj >= 0; --lj) {
                    for (int i0 = tid; i0 < pairs; i0 += 4 * nt) {
                        int xs[4], valid = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = i0 + e * nt;
                            xs[e] = ((i >> lj) << (lj + 1)) + (i & ((1 << lj) - 1));
                            if (i < pairs) valid = e + 1;
                        }
                        exchange4(xs[0], xs[1], xs[2], xs[3], 1 << lj, 1 << lj, 1 << lj, 1 << lj, valid);
                    }
                    lds_barrier();
                }
            }
            for (int x = tid; x < limit; x += nt) {
                const int g = x >> lg, j = x & (slot_len - 1);
                const int b = (int)big_list[1 + q0 + g];
                const unsigned int first = g_start[b], m = g_start[b + 1] - first;
                if ((unsigned int)j < m) {
                    const unsigned int i = stage_idx[x];
                    if (perm) perm[first + j] = i;
                    if (y_gather) f_out[first + j] = y_gather[i];
                    if constexpr (HAS_W) { if (y_gather) w_out_g[first + j] = w_gather[i]; }
                }
            }
            wg_sync();
            q0 += count;
        }
    }
    pc.mark(3);
    return true;
}

// The same as a real function with its own register allocation.  Inlined into the search kernel the sort
// shares the 128 registers with everything the kernel keeps alive across a period; the allocator then spills
// inside the sort's loops, and a spill reload is a scratch load whose s_waitcnt vmcnt(0) also waits for every
// global load in flight -- the reads of a round went out one at a time (twice the time of the partition pass,
// and which loops were hit changed with unrelated edits elsewhere in the kernel).
// The arguments arrive in vector registers: the uniform ones are moved back to scalars, the LDS pointer to
// its address space.  Keeps its own phase clock.
// One block of the exact prefix sum as a real function for the slab rounds (own register allocation: inlined
// into the slab kernel its 16 elements per thread are spilled around every phase).  f and C in LDS, in place
// (C[k+1] over f[k]); addresses as 32-bit LDS addresses.
__device__ __noinline__ double exact_cumsum_round_call(unsigned int buf_addr, int len, double carry, unsigned int cs_addr,
                                                       global_ptr<unsigned long long> dbg) {
    typedef __attribute__((address_space(3))) double* lds_d;
    typedef __attribute__((address_space(3))) Cumsum2Scratch* lds_c;
    double* buf = (double*)(lds_d)(uintptr_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)buf_addr);
    Cumsum2Scratch* cs = (Cumsum2Scratch*)(lds_c)(uintptr_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)cs_addr);
    return exact_cumsum_block_inline<16, true>(buf + 1, buf, 0, uniform_i32(len), uniform_f64(carry), cs, from_global_arg(dbg));
}

template <bool HAS_W>
__device__ __noinline__ bool fold_and_sort_tiled_call(global_ptr<const double> t, int n, double period,
                                                      global_ptr<unsigned long long> g_rec, global_ptr<unsigned int> perm,
                                                      unsigned int lds_addr, global_ptr<unsigned long long> clock_out,
                                                      global_ptr<const double> y_gather, global_ptr<const double> w_gather,
                                                      global_ptr<double> f_out, global_ptr<double> w_out_g,
                                                      global_ptr<unsigned long long> dbg_check) {
    PhaseClock pc; pc.start(from_global_arg(clock_out));
    typedef __attribute__((address_space(3))) unsigned char* lds_t;
    unsigned char* lds = (unsigned char*)(lds_t)(uintptr_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)lds_addr);
    return fold_and_sort_tiled<HAS_W>(from_global_arg(t), uniform_i32(n), uniform_f64(period), 0.0, from_global_arg(g_rec),
                                      from_global_arg(perm), lds, pc, from_global_arg(y_gather), from_global_arg(w_gather),
                                      from_global_arg(f_out), from_global_arg(w_out_g), from_global_arg(dbg_check));
}

// dst[0..count) (slab, HBM) = src[0..count) (LDS): 16-byte stores wherever the destination allows
// the same for a piece of the prefix sum C on its way into the slab, stored as X[k] = k - C[k] (an exact subtraction:
// see depth_pass); src[j] is C[k0 + j]
typedef double f64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void copy_out_stream_x(double* dst, const double* src, int count, int tid, int k0) {
    const int nt = blockDim.x;
    const int head = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1ull);   // 1: dst starts on an odd element
    if (tid == 0 && head && count > 0) stream_store(dst, (double)k0 - src[0]);
    const int pairs = (count - head) / 2;
    f64x2_t* d2 = reinterpret_cast<f64x2_t*>(dst + head);
    const double* s1 = src + head;
    const int kb = k0 + head;
    for (int q = tid; q < pairs; q += nt) {
        f64x2_t v; v.x = (double)(kb + 2 * q) - s1[2 * q]; v.y = (double)(kb + 2 * q + 1) - s1[2 * q + 1];
        stream_store(d2 + q, v);
    }
    if (tid == 0 && head + 2 * pairs < count) stream_store(dst + count - 1, (double)(k0 + count - 1) - src[count - 1]);
}
__device__ __forceinline__ void copy_out_stream(double* dst, const double* src, int count, int tid) {
    const int nt = blockDim.x;
    const int head = (int)((reinterpret_cast<unsigned long long>(dst) >> 3) & 1ull);   // 1: dst starts on an odd element
    if (tid == 0 && head && count > 0) stream_store(dst, src[0]);
    const int pairs = (count - head) / 2;
    f64x2_t* d2 = reinterpret_cast<f64x2_t*>(dst + head);
    const double* s1 = src + head;
    for (int q = tid; q < pairs; q += nt) {
        f64x2_t v; v.x = s1[2 * q]; v.y = s1[2 * q + 1];
        stream_store(d2 + q, v);
    }
    if (tid == 0 && head + 2 * pairs < count) stream_store(dst + count - 1, src[count - 1]);
}

// The exact prefix pass over a period's slab, out of line (registers of its own), for a work item of the two-role kernel's
// search role whose fast attempt noted band windows (2-7 % of the periods): the fold role leaves no X for a fast period, so
// the item forms X = k - numpy.cumsum itself -- f (regA) through LDS in rounds of `round` elements, the patch (core.py:126) an
// index mapping of the copy-in, X to regB, the sentinels behind it.  Needs the full 1024-thread workgroup (one block a round).
__device__ __noinline__ void slab_exact_prefix_call(global_ptr<const double> f_, global_ptr<double> x_, int n_, int M_, int region_pad_,
                                                    int round_, unsigned int buf_addr_, unsigned int cs_addr_,
                                                    global_ptr<unsigned long long> dbg) {
    typedef __attribute__((address_space(3))) double* lds_d;
    const double* regA = from_global_arg(f_);
    double* regB = from_global_arg(x_);
    const int n = uniform_i32(n_), M = uniform_i32(M_), region_pad = uniform_i32(region_pad_), kRound = uniform_i32(round_);
    const unsigned int buf_addr = (unsigned int)__builtin_amdgcn_readfirstlane((int)buf_addr_);
    const unsigned int cs_addr = (unsigned int)__builtin_amdgcn_readfirstlane((int)cs_addr_);
    double* buf = (double*)(lds_d)(uintptr_t)buf_addr;   // C[0..len], f = buf + 1 (16-byte aligned)
    const int tid = threadIdx.x, nt = blockDim.x;
    double carry = 0.0;
    const bool dma = TLS_SLAB_DMA && (n & 1) == 0;   // pairs of samples never straddle the patch boundary
    for (int c0 = 0; c0 < M; c0 += kRound) {
        const int len = M - c0 < kRound ? M - c0 : kRound;
        if (dma) { slab_to_lds_async(buf + 1, regA, c0, len, n, tid); vmem_wait_all(); }
        else {
            for (int k = tid; k < len; k += nt) { const int pp = c0 + k; buf[1 + k] = regA[pp < n ? pp : pp - n]; }
        }
        lds_barrier();
        for (int b0 = 0; b0 < len; b0 += 16 * nt) {   // (16 elements a thread and block: one block when the workgroup has 1024 threads)
            const int blen = len - b0 < 16 * nt ? len - b0 : 16 * nt;
            carry = exact_cumsum_round_call(buf_addr + 8u * (unsigned int)b0, blen, carry, cs_addr, dbg);
        }
        copy_out_stream_x(regB + c0, buf, len + 1, tid, c0);   // the slab keeps X[k] = k - C[k]
        lds_barrier();
    }
    for (int k = tid; k < region_pad; k += nt) regB[M + 1 + k] = -(double)(k + 1) * 1.0e300;
    wg_sync();
}

typedef const __attribute__((address_space(4))) SearchArgs* args_ptr;
// COUNTING: the instantiation that can report evaluated cells, template taps and issued FMAs (tls_execute(ctx, 1)); the
// plain one does not keep the counters at all (-1.4 % on config 2: two VALU instructions per window and a 64-bit
// multiply per batch that nobody reads).
// ROLE (series in the HBM slab, one light curve): kRoleAll = one workgroup takes a period from the fold to the argmin (the
// LDS-resident kernel; survey batches on long series).  kRoleFold = phases 1-2 only, one period per work item, the slab
// of every period of the batch left in HBM; kRoleSearch = phases 3-4 over (period, position tile) work items of those
// slabs.  Two kernels instead of one: each gets its own register allocation (the fold's sort and exact prefix sum no
// longer share 128 registers with the dot products), and a period is searched by as many workgroups as it has tiles --
// a rank of an 8-GPU job that holds 300 periods still fills 256 CUs.  The cells, their values and the comparison are
// the same code, so the results are the same bits.
enum { kRoleAll = 0, kRoleFold = 1, kRoleSearch = 2 };
// One kernel, one workgroup per period from the fold to the argmin (LDS-resident series; survey batches on long series).
// SCREEN: the fp32 screen of the dot products (screen_cells; LDS-resident series, uniform weights, plain variant; the
// host admits it when every sample splits exactly into two fp32 halves).
template <bool RESIDENT, bool UNIFORM_W, typename IdxT, bool WITH_PRUNING = false, bool COUNTING = true, bool SCREEN = false>
__global__ void __launch_bounds__(TLS_LAUNCH_THREADS, TLS_WAVES_PER_EU)
tls_search_kernel(const SearchArgs) {
    constexpr int ROLE = kRoleAll;
#include "tls_search_body.inc.h"
}

// Series in the HBM slab, one light curve: every workgroup first takes periods of the batch through the fold role until
// none is left, then (period, tile) items through the search role; a search item waits for its period's slab
// (SearchArgs::fold_ready).  The two roles follow each other in the kernel, so nothing the fold keeps in registers is
// alive in the search and the other way round -- the register allocation of two kernels in one launch, without a
// second launch's idle tail.  No deadlock: a workgroup asks for a search item only after it has seen the fold queue
// empty, so the fold it may wait for is running on a workgroup that holds a CU and waits for nothing.
// (The roles as called `__noinline__` functions were measured too: no different in time -- and a called function cannot
// reach the kernel-argument segment through the builtin, which is null there.  PERF_LOG.md, round 4.)
template <bool UNI_, bool COUNT_ = true>
__global__ void __launch_bounds__(TLS_LAUNCH_THREADS, TLS_WAVES_PER_EU)
tls_fold_search_kernel(const SearchArgs) {
    {
        constexpr bool RESIDENT = false, UNIFORM_W = UNI_, WITH_PRUNING = false, COUNTING = false, SCREEN = false;
        typedef unsigned int IdxT;
        constexpr int ROLE = kRoleFold;
#include "tls_search_body.inc.h"
    }
    wg_sync();
    {
        constexpr bool RESIDENT = false, UNIFORM_W = UNI_, WITH_PRUNING = false, COUNTING = COUNT_, SCREEN = false;
        typedef unsigned int IdxT;
        constexpr int ROLE = kRoleSearch;
#include "tls_search_body.inc.h"
    }
}

#include "tls_slim_kernel.hip.h"

// ---------------------------------------------------------------------------------------
// Final T0 fit (reference stats.py:135-204): for every trial epoch Tx fold the light curve at
// (period, Tx), stable-sort, roll by `roll` cadences twice and sum the residuals of the
// depth-scaled template over the first `dur` folded samples plus the out-of-transit residuals,
// BOTH weighted by 1/flux^2 of the doubly rolled flux (the reference overwrites dy with the
// rolled flux, stats.py:191 -- kept).  One workgroup per trial epoch; the host takes the first
// minimum (stats.py:199-201).
struct T0FitArgs {
    const double* t;        // [n]
    const double* y;        // [n]
    const double* signal;   // [dur] template scaled to the fitted depth
    const double* epochs;   // [n_epochs] trial T0 values
    double* residuals;      // [n_epochs]
    unsigned int* queue;        // work-queue head (zeroed by the host before the launch)
    double* scratch;        // non-resident slabs: 3*n doubles per workgroup
    long long scratch_stride;
    double period;
    int n, dur, roll, n_epochs, nb;
    // survey batches / the fused power() chain: fit c = blockIdx.y takes period, dur, roll and n_epochs from params[c] (written
    // on the device by tls_power_prep: no host round trip between the search and the fit) and its arrays at y + c * y_stride,
    // signal + c * signal_stride, epochs / residuals + c * epoch_stride.  nullptr: one fit, the scalars above.
    const struct T0FitParams* params;
    long long y_stride, signal_stride, epoch_stride;
    // Rotation path (round 6; tls_t0fit_rot below).  mode 0: every epoch of every fit in this kernel (the original form).
    // mode 1 ("base"): one workgroup per fit sorts the fit's FIRST epoch and leaves, per fit, in `rot` (rot_stride doubles a
    // fit): the flux in that order Fb[n] | the phases phb[n] | the out-of-transit quotients c[n] | state[4] = (sum of c,
    // flag: != 0 the fit must take this kernel for every epoch, smallest gap between neighbouring phases, |q| bound of the
    // base epoch), and the order itself in rot_perm[n].  mode 2: only the fits whose flag is set (mode 0's work for them).
    int mode;
    double* rot;
    int* rot_perm;
    long long rot_stride;
    double t_lo, t_hi;      // min and max of t (the |q| bound of an epoch's fold)
};
struct T0FitParams { double period; int dur, roll, n_epochs, pad_; };
constexpr int kT0RotMinPoints = 16;   // (the rotation path looks at eight neighbours of the wrap)

template <bool RESIDENT, typename IdxT>
__global__ void __launch_bounds__(1024) tls_t0fit_kernel(const T0FitArgs a0) {
    T0FitArgs a = a0;
    if (a0.params != nullptr) {
        const long long c = blockIdx.y;
        const T0FitParams fp = a0.params[c];
        a.period = fp.period; a.dur = fp.dur; a.roll = fp.roll; a.n_epochs = fp.n_epochs;
        a.y = a0.y + c * a0.y_stride; a.signal = a0.signal + c * a0.signal_stride;
        a.epochs = a0.epochs + c * a0.epoch_stride; a.residuals = a0.residuals + c * a0.epoch_stride;
        if ((int)blockIdx.x >= a.n_epochs) return;   // (uniform for the workgroup)
    }
    double* const rot_fit = a.rot ? a.rot + (long long)blockIdx.y * a.rot_stride : nullptr;
    if (a.mode == 2 && rot_fit[3LL * a.n + 1] == 0.0) return;   // (the rotation path has done this fit)
    if (a.mode == 1) a.n_epochs = a.n_epochs > 0 ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int n = a.n;
    unsigned int* wsum = reinterpret_cast<unsigned int*>(smem);       // 32 words
    double* wred = reinterpret_cast<double*>(smem + 128);             // kMaxWaves doubles
    int* s_work = reinterpret_cast<int*>(smem + 256);
    constexpr int kHdr = 272;
    double *regA, *regB;
    unsigned int* cnt;
    if constexpr (RESIDENT) {
        regA = reinterpret_cast<double*>(smem + kHdr);   // phases, then folded flux
        regB = regA + n;                                 // sort scratch
        cnt = reinterpret_cast<unsigned int*>(regB);
    } else {
        regA = a.scratch + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * a.scratch_stride;
        regB = regA + n;
        cnt = reinterpret_cast<unsigned int*>(smem + kHdr);
    }
    IdxT* idx_tmp = RESIDENT ? reinterpret_cast<IdxT*>(cnt + a.nb) : reinterpret_cast<IdxT*>(regB);
    IdxT* perm = idx_tmp + n;
    PhaseClock pc; pc.start(nullptr);
    // The epochs of one fit share period and time stamps: fold(t, P, T0) = frac((t - T0) / P) for another T0 shifts every
    // phase by the same amount, so the stable order of one epoch is -- rounding aside -- a ROTATION of the order of any
    // other.  A workgroup therefore sorts once, for the first epoch it takes (LDS-resident series), and keeps that order
    // (`perm`) and the flux in that order (`regA`); for every further epoch it folds each point and its successor in the
    // kept order and counts the places where (phase, index) DEcreases around the cycle: exactly one such place means the
    // kept order, started behind it, IS numpy's stable order of this epoch (the keys are distinct, so the sorted order is
    // unique) -- bit for bit what a sort of its own would give; anything else (two phases that round the other way
    // round under this epoch, ties that reorder) sends the epoch through the sort, whose order is kept from then on.
    // stats.py:178-201 sorts N points for each of up to N epochs; here an epoch costs two folds per point.
    bool have_base = false;
    // (the epochs of a fit cost the same: a workgroup takes every gridDim.x-th one -- no queue, no device-scope atomic and
    // its round trip per epoch)
    for (int work = blockIdx.x; work < a.n_epochs; work += gridDim.x) {
        if (tid == 0) { s_work[1] = 0; s_work[2] = 0; }
        wg_sync();
        const double epoch = a.epochs[work];
        int start = 0;          // sorted position k holds the kept order's entry (k + start) mod n
        bool rotated = false;
        if constexpr (RESIDENT) {
            if (have_base) {
                int descents = 0, where = 0;
                // (a thread takes a run of consecutive entries of the kept order -- one fold per point and one more for the
                // run's successor, their time stamps requested together --, eight at a time)
                constexpr int kRun = 8;
                const double inv_period = 1.0 / a.period;
                const int per = (n + nt - 1) / nt;
                const int k_lo = tid * per < n ? tid * per : n, k_hi = k_lo + per < n ? k_lo + per : n;
                for (int k0 = k_lo; k0 < k_hi; k0 += kRun) {
                    int idx[kRun + 1];
                    double ph[kRun + 1];
#pragma unroll
                    for (int j = 0; j <= kRun; ++j) {
                        int k = k0 + j < k_hi + 1 ? k0 + j : k_hi;   // (the run's successor included; past it: the successor again)
                        if (k >= n) k -= n;
                        idx[j] = (int)perm[k];
                    }
                    double tv[kRun + 1], margin[kRun + 1];
#pragma unroll
                    for (int j = 0; j <= kRun; ++j) tv[j] = a.t[idx[j]];
                    // The order of two neighbours is read off phases formed with ONE multiplication by 1 / P instead of the
                    // reference's division: q' = (t - T0) * fl(1 / P) is within 4 * 2^-53 |q| of the quotient the exact fold rounds,
                    // so neighbours whose cheap phases differ by more than the margin (and lie off the wrap at 0 / 1 by it) compare
                    // like the exact ones; any other pair (ties, near-ties, the wrap) is folded exactly.  The fp64 divisions of the
                    // folds were most of an epoch: every workgroup of the fit divides at the same time.
#pragma unroll
                    for (int j = 0; j <= kRun; ++j) {
                        const double qv = (tv[j] - epoch) * inv_period;
                        ph[j] = qv - floor(qv);
                        margin[j] = 1.0e-15 * (fabs(qv) + 1.0);
                    }
#pragma unroll
                    for (int j = 0; j < kRun; ++j) {
                        if (k0 + j < k_hi) {
                            const double mg = margin[j] + margin[j + 1];
                            const bool off_wrap = ph[j] > mg && ph[j] < 1.0 - mg && ph[j + 1] > mg && ph[j + 1] < 1.0 - mg;
                            bool down;
                            if (off_wrap && fabs(ph[j + 1] - ph[j]) > mg) {
                                down = ph[j + 1] < ph[j];
                            } else {
                                const double e0 = fold_phase(tv[j], a.period, epoch), e1 = fold_phase(tv[j + 1], a.period, epoch);
                                down = e1 < e0 || (e1 == e0 && idx[j + 1] < idx[j]);
                            }
                            if (down) { ++descents; where = k0 + j + 1 < n ? k0 + j + 1 : 0; }
                        }
                    }
                }
                if (descents) { atomicAdd(&s_work[1], descents); s_work[2] = where; }
                wg_sync();
                rotated = s_work[1] == 1 || n < 2;
                start = n < 2 ? 0 : s_work[2];
                wg_sync();
            }
        }
        if (!rotated) {
            // (a trial epoch does not change how the phases pile up: the period does -- same remedy as in the search)
            fold_and_sort<IdxT>(a.t, n, a.period, epoch, regA, cnt, a.nb, idx_tmp, perm, wsum, pc,
                                reinterpret_cast<unsigned int*>(wred), 2 * kMaxWaves);
            for (int k = tid; k < n; k += nt) regA[k] = a.y[(int)perm[k]];   // phases are dead
            wg_sync();
            have_base = RESIDENT;   // (series in HBM: the sort's scratch is shared with the order; every epoch sorts)
            start = 0;
        }
        if (a.mode == 1) {
            // base of the rotation path: order, flux, phases and out-of-transit quotients of this epoch; the sum of the
            // quotients; the smallest gap between neighbouring phases around the cycle (a gap the folds' rounding could close
            // under another epoch, or a tie, sends the fit through every epoch's own check: mode 2)
            double* Fb = rot_fit; double* phb = Fb + n; double* cq = phb + n; double* state = cq + n;
            int* order = a.rot_perm + (long long)blockIdx.y * n;
            const int r1 = a.roll % n;
            double c_sum = 0.0, gap_min = INFINITY;
            for (int k = tid; k < n; k += nt) {
                const int i = (int)perm[k], i_next = (int)perm[k + 1 < n ? k + 1 : 0];
                const double ph = fold_phase(a.t[i], a.period, epoch), ph_next = fold_phase(a.t[i_next], a.period, epoch);
                int kk = k - r1; if (kk < 0) kk += n;
                const double f1 = regA[k], f2 = regA[kk];
                const double dlt = f1 - 1.0;
                const double c = (dlt * dlt) / (f2 * f2);
                order[k] = i; Fb[k] = f1; phb[k] = ph; cq[k] = c;
                c_sum += c;
                gap_min = fmin(gap_min, k + 1 < n ? ph_next - ph : (ph_next + 1.0) - ph);
            }
#pragma unroll
            for (int dlt = kWave / 2; dlt > 0; dlt >>= 1) {
                c_sum += __shfl_down(c_sum, dlt, kWave);
                gap_min = fmin(gap_min, __shfl_down(gap_min, dlt, kWave));
            }
            if (lane == 0) wred[wave] = c_sum;
            wg_sync();
            double tot = 0.0;
            if (tid == 0) for (int v = 0; v < nw; ++v) tot += wred[v];
            wg_sync();
            if (lane == 0) wred[wave] = gap_min;
            wg_sync();
            if (tid == 0) {
                double gm = INFINITY;
                for (int v = 0; v < nw; ++v) gm = fmin(gm, wred[v]);
                const double q_b = fmax(fabs(a.t_lo - epoch), fabs(a.t_hi - epoch)) / a.period + 1.0;
                state[0] = tot;
                state[1] = (n >= kT0RotMinPoints && gm > 0.0 && tot == tot) ? 0.0 : 1.0;   // (ties, tiny series, NaN flux: the general kernel)
                state[2] = gm; state[3] = q_b;
            }
            return;
        }
        // flux rolled once: F1[k] = F[(k - roll) mod n]; weights: F2[k] = F[(k - 2 roll) mod n]
        const int r1 = a.roll % n, r2 = (2 * a.roll) % n;
        double acc = 0.0;
        for (int k = tid; k < n; k += nt) {
            int k1 = k - r1 + start; if (k1 < 0) k1 += n; if (k1 >= n) k1 -= n;
            int k2 = k - r2 + start; if (k2 < 0) k2 += n; if (k2 >= n) k2 -= n;
            const double f1 = regA[k1], f2 = regA[k2];
            const double model = k < a.dur ? a.signal[k] : 1.0;
            const double dlt = f1 - model;
            acc += (dlt * dlt) / (f2 * f2);
        }
#pragma unroll
        for (int dlt = kWave / 2; dlt > 0; dlt >>= 1) acc += __shfl_down(acc, dlt, kWave);
        if (lane == 0) wred[wave] = acc;
        wg_sync();
        if (tid == 0) {
            double tot = 0.0;
            for (int v = 0; v < nw; ++v) tot += wred[v];
            a.residuals[work] = tot;
        }
        wg_sync();
    }
}

// The epochs of a fit as rotations of ONE sorted order (round 6).  fold(t, P, T0) of another trial epoch shifts every real
// phase by the same amount, so -- rounding aside -- every epoch's stable order is the base epoch's (tls_t0fit_kernel, mode 1)
// started somewhere else.  Rounding: a computed phase is within err = 2^-52 (|q| + 1) of the real one (q = (t - T0) / P: one
// rounded subtraction, one rounded division, an exact x - floor(x)); two neighbours of the base order whose computed phases
// differ by more than 6e-15 (Q + 2) >= 6 err compare the same way under EVERY epoch unless the wrap at 0 / 1 lies between
// them, and at most one point sits within err of the wrap.  So with every gap above that bound an epoch's order has exactly
// one descent, at most one place from where the real phases wrap: the wave looks the wrap up in the base phases
// (lower bound of 1 - shift), folds the eight neighbours exactly, and takes the one descent it finds there; no descent or
// two, or a gap below the epoch's bound: the fit's flag is raised and tls_t0fit_kernel (mode 2) redoes the whole fit with its
// per-epoch check of every pair.  The residual: out of transit the term of a point, (F[j] - 1)^2 / F[j - roll]^2, does not
// depend on the epoch (the doubly rolled weights are positions in the same cyclic order), so
//   residual = sum_j c[j]  +  sum_{k < dur} ( (F[j_k] - signal[k])^2 / F[j_k - roll]^2 - c[j_k] ),   j_k = k - roll + start:
// N + dur terms a fit and dur an epoch instead of N an epoch (and no sort of N points per epoch where the series does not fit
// the LDS: TESS 6.6 -> 0.1 ms, Kepler 216 -> 1 ms a fit).  One wavefront per epoch.
__global__ void __launch_bounds__(256) tls_t0fit_rot(const T0FitArgs a0) {
    T0FitArgs a = a0;
    if (a0.params != nullptr) {
        const long long c = blockIdx.y;
        const T0FitParams fp = a0.params[c];
        a.period = fp.period; a.dur = fp.dur; a.roll = fp.roll; a.n_epochs = fp.n_epochs;
        a.signal = a0.signal + c * a0.signal_stride;
        a.epochs = a0.epochs + c * a0.epoch_stride; a.residuals = a0.residuals + c * a0.epoch_stride;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int e = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave));
    const int n = a.n;
    if (e >= a.n_epochs) return;
    double* const rot_fit = a.rot + (long long)blockIdx.y * a.rot_stride;
    const double* Fb = rot_fit; const double* phb = Fb + n; const double* cq = phb + n; double* state = rot_fit + 3LL * n;
    const int* order = a.rot_perm + (long long)blockIdx.y * n;
    if (state[1] != 0.0) return;
    const double epoch = a.epochs[e], epoch_b = a.epochs[0];
    const double q_e = fmax(fabs(a.t_lo - epoch), fabs(a.t_hi - epoch)) / a.period + 1.0;
    if (!(state[2] > 6.0e-15 * (fmax(q_e, state[3]) + 2.0))) { if (lane == 0) state[1] = 1.0; return; }
    // where the real phases wrap under this epoch: the first base phase >= 1 - shift (64-ary search, the lanes probe)
    const double shift = fold_phase(epoch_b, a.period, epoch);
    const double target = 1.0 - shift;
    int lo = 0, hi = n;   // every base phase in front of lo is < target; the one at hi (if any) is not
    while (lo < hi) {
        const int step = (hi - lo + kWave - 1) / kWave;
        const int idx = lo + lane * step;
        const bool less = idx < hi && phb[idx] < target;
        const int cnt = __popcll(ballot64(less));          // (the base phases ascend: the lanes that see `less` are the first cnt)
        if (cnt == 0) break;
        const int first_not = lo + cnt * step;
        lo = lo + (cnt - 1) * step + 1;
        hi = first_not < hi ? first_not : hi;
    }
    const int m = lo < n ? lo : 0;
    // the eight entries around it, folded exactly as the reference folds them: one descent among the seven pairs
    int pos = m - 4 + (lane & 7); if (pos < 0) pos += n; if (pos >= n) pos -= n;
    const int i_mine = order[pos];
    const double ph = fold_phase(a.t[i_mine], a.period, epoch);
    const double ph_next = __shfl_down(ph, 1, kWave);
    const int i_next = __shfl_down(i_mine, 1, kWave), pos_next = __shfl_down(pos, 1, kWave);
    const bool down = lane < 7 && (ph_next < ph || (ph_next == ph && i_next < i_mine));
    const unsigned long long downs = ballot64(down);
    if (__popcll(downs) != 1) { if (lane == 0) state[1] = 1.0; return; }
    const int start = __builtin_amdgcn_readlane(pos_next, __ffsll((long long)downs) - 1);
    // the template's samples against the flux behind the rolls, minus what these points weigh out of transit
    const int r1 = a.roll % n, dur = a.dur < n ? a.dur : n;
    double acc = 0.0;
    for (int k = lane; k < dur; k += kWave) {
        int j = k - r1 + start; if (j < 0) j += n; if (j >= n) j -= n;
        int jj = j - r1; if (jj < 0) jj += n;
        const double f1 = Fb[j], f2 = Fb[jj];
        const double dlt = f1 - a.signal[k];
        acc += (dlt * dlt) / (f2 * f2) - cq[j];
    }
#pragma unroll
    for (int dlt = kWave / 2; dlt > 0; dlt >>= 1) acc += __shfl_down(acc, dlt, kWave);
    if (lane == 0) a.residuals[e] = state[0] + acc;
}

// ---------------------------------------------------------------------------------------
// SDE spectra from the chi^2 of every period (reference stats.py:105-132 with the running median of
// helpers.py:93-108), on the chi^2 array that is still resident after the search:
//   tls_spectra_head    SR = min(chi2)/chi2, SDE_raw = (1 - mean SR)/std SR, power_raw scaled to SDE_raw
//   tls_spectra_median  power = power_raw - running median (window `kernel`, odd; edges padded)
//   tls_spectra_tail    power -= mean; SDE = max(power)/std(power); power scaled to SDE
// Sums are fp64 tree reductions (numpy sums pairwise: the two agree to ~1e-16 relative).
struct SpectraArgs {
    const double* chi2;
    double* SR;
    double* power_raw;
    double* power;
    double* sde;        // [0] SDE_raw, [1] SDE
    int n, kernel, detrend;   // detrend: n > 2 * kernel (stats.py:118)
    // survey batches (tls_power_batch): light curve c = blockIdx.y reads chi2 + c * chi2_stride and writes
    // SR / power_raw / power + c * out_stride, sde + c * sde_stride (all 0 for one light curve)
    long long chi2_stride, out_stride, sde_stride;
};
__device__ __forceinline__ SpectraArgs spectra_of_curve(const SpectraArgs& a) {
    SpectraArgs b = a;
    const long long c = blockIdx.y;
    b.chi2 = a.chi2 + c * a.chi2_stride;
    b.SR = a.SR + c * a.out_stride; b.power_raw = a.power_raw + c * a.out_stride; b.power = a.power + c * a.out_stride;
    b.sde = a.sde + c * a.sde_stride;
    return b;
}

template <typename Op>
__device__ __forceinline__ double block_reduce(double v, Op op, double* red /* LDS [kMaxWaves + 1] */) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) v = op(v, __shfl_down(v, d, kWave));
    wg_sync();
    if (lane == 0) red[wave] = v;
    wg_sync();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int k = 1; k < nw; ++k) r = op(r, red[k]);
        red[kMaxWaves] = r;
    }
    wg_sync();
    return red[kMaxWaves];
}

__global__ void __launch_bounds__(1024) tls_spectra_head(const SpectraArgs a0) {
    const SpectraArgs a = spectra_of_curve(a0);
    __shared__ double red[kMaxWaves + 1];
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n;
    auto add = [](double x, double y) { return x + y; };
    auto mn = [](double x, double y) { return x < y ? x : y; };
    auto mx = [](double x, double y) { return x > y ? x : y; };
    double v = INFINITY;
    for (int k = tid; k < n; k += nt) v = mn(v, a.chi2[k]);
    const double cmin = block_reduce(v, mn, red);
    v = 0.0;
    for (int k = tid; k < n; k += nt) { const double sr = cmin / a.chi2[k]; a.SR[k] = sr; v += sr; }   // stats.py:106
    const double mean = block_reduce(v, add, red) / (double)n;
    v = 0.0;
    for (int k = tid; k < n; k += nt) { const double d = a.SR[k] - mean; v += d * d; }
    const double sd = sqrt(block_reduce(v, add, red) / (double)n);
    const double sde_raw = (1 - mean) / sd;                                                    // :107
    v = -INFINITY;
    for (int k = tid; k < n; k += nt) v = mx(v, a.SR[k] - mean);
    const double scale = sde_raw / block_reduce(v, mx, red);                                   // :111
    for (int k = tid; k < n; k += nt) {
        const double p = (a.SR[k] - mean) * scale;
        a.power_raw[k] = p;
        if (!a.detrend) a.power[k] = p;                                                        // :128
    }
    if (tid == 0) { a.sde[0] = sde_raw; if (!a.detrend) a.sde[1] = sde_raw; }
}

// kMedianWindows windows per workgroup; the (window, candidate) pairs are spread over the threads: a candidate is
// the median of its (odd) window iff exactly kernel/2 elements precede it in (value, position) order.  (One
// window per thread walked up to kernel^2 elements serially: 0.28 ms for 9679 periods, on 38 CUs.)
constexpr int kMedianWindows = 16;
__global__ void __launch_bounds__(256) tls_spectra_median(const SpectraArgs a0) {
    const SpectraArgs a = spectra_of_curve(a0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n, k = a.kernel;
    const int n_med = n - k + 1, first = blockIdx.x * kMedianWindows;
    const int windows = first + kMedianWindows < n_med ? kMedianWindows : n_med - first;
    const int staged = windows + k - 1;
    double* w = reinterpret_cast<double*>(smem);            // [kMedianWindows + kernel - 1]
    double* med = w + (kMedianWindows + k - 1);               // [kMedianWindows]
    for (int j = tid; j < staged; j += nt) w[j] = a.power_raw[first + j];
    // a window that holds a NaN has no candidate of the right rank (every comparison is false): its median is NaN,
    // as numpy.median's (helpers.py:96)
    for (int i = tid; i < kMedianWindows; i += nt) med[i] = __longlong_as_double(0x7ff8000000000000LL);
    wg_sync();
    for (int pair = tid; pair < windows * k; pair += nt) {
        const int i = pair / k, j = pair - i * k;
        const double* x = w + i;
        const double xj = x[j];
        int rank = 0;
        for (int l = 0; l < k; ++l) rank += (x[l] < xj || (x[l] == xj && l < j)) ? 1 : 0;
        if (rank == k / 2) med[i] = xj;   // exactly one candidate of a window has this rank
    }
    wg_sync();
    // helpers.py:100-108: the medians sit in the middle, the first/last one pads the edges
    const int missing = n - n_med, front = (int)((double)missing * 0.5);
    for (int i = tid; i < windows; i += nt) a.power[front + first + i] = a.power_raw[front + first + i] - med[i];   // stats.py:120
    if (first == 0) for (int j = tid; j < front; j += nt) a.power[j] = a.power_raw[j] - med[0];
    if (first + windows == n_med) for (int j = front + n_med + tid; j < n; j += nt) a.power[j] = a.power_raw[j] - med[windows - 1];
}

__global__ void __launch_bounds__(1024) tls_spectra_tail(const SpectraArgs a0) {
    const SpectraArgs a = spectra_of_curve(a0);
    __shared__ double red[kMaxWaves + 1];
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n;
    auto add = [](double x, double y) { return x + y; };
    auto mx = [](double x, double y) { return x > y ? x : y; };
    double v = 0.0;
    for (int k = tid; k < n; k += nt) v += a.power[k];
    const double mean = block_reduce(v, add, red) / (double)n;                                 // :123
    v = 0.0;
    double top = -INFINITY;
    for (int k = tid; k < n; k += nt) { const double p = a.power[k] - mean; a.power[k] = p; top = mx(top, p); }
    // mean of the shifted values (numpy.std subtracts it again)
    v = 0.0;
    for (int k = tid; k < n; k += nt) v += a.power[k];
    const double mean2 = block_reduce(v, add, red) / (double)n;
    v = 0.0;
    for (int k = tid; k < n; k += nt) { const double d = a.power[k] - mean2; v += d * d; }
    const double sd = sqrt(block_reduce(v, add, red) / (double)n);
    const double pmax = block_reduce(top, mx, red);
    const double sde = pmax / sd;                                                              // :124 (division is monotone)
    const double scale = sde / pmax;                                                           // :126
    for (int k = tid; k < n; k += nt) a.power[k] = a.power[k] * scale;
    if (tid == 0) a.sde[1] = sde;
}

// Survey-mode power(): what main.py:198-212,269-272 read off the spectra of one light curve -- one workgroup per
// light curve (blockIdx.x).  out[c][0..7] = chi2_min, index of the FIRST minimum of chi2 (numpy.argmin), index of the
// FIRST maximum of the detrended power (numpy.argmax), period and depth at the power peak, template row at the chi2
// minimum, max(chi2) == min(chi2) ("no transit was fit", main.py:203), reserved.
struct PickArgs {
    const double* chi2; const long long* row; const double* depth;   // [n_curves][n] (stride n)
    const double* power;                                             // [n_curves] stride power_stride
    const double* periods;                                           // [n]
    double* out;                                                     // [n_curves][8]
    long long power_stride;
    int n;
};
__global__ void __launch_bounds__(1024) tls_power_pick(const PickArgs a) {
    __shared__ double red_v[kMaxWaves];
    __shared__ int red_i[kMaxWaves];
    const int tid = threadIdx.x, nt = blockDim.x, n = a.n;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const long long c = blockIdx.x;
    const double* chi2 = a.chi2 + c * n;
    const double* power = a.power + c * a.power_stride;
    // first minimum of chi2, first maximum of power, maximum of chi2: (value, index) reductions, lower index wins ties
    double vmin = INFINITY, vmax = -INFINITY, cmax = -INFINITY;
    int imin = 0x7fffffff, imax = 0x7fffffff;
    for (int k = tid; k < n; k += nt) {
        const double x = chi2[k], pw = power[k];
        if (x < vmin) { vmin = x; imin = k; }
        if (pw > vmax) { vmax = pw; imax = k; }
        cmax = x > cmax ? x : cmax;
    }
    auto reduce_first = [&](double v, int i, bool want_min) -> int {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            const double ov = __shfl_down(v, d, kWave);
            const int oi = __shfl_down(i, d, kWave);
            const bool take = want_min ? (ov < v || (ov == v && oi < i)) : (ov > v || (ov == v && oi < i));
            if (take) { v = ov; i = oi; }
        }
        wg_sync();
        if (lane == 0) { red_v[wave] = v; red_i[wave] = i; }
        wg_sync();
        if (tid == 0) {
            for (int w = 1; w < nw; ++w) {
                const double ov = red_v[w]; const int oi = red_i[w];
                const bool take = want_min ? (ov < v || (ov == v && oi < i)) : (ov > v || (ov == v && oi < i));
                if (take) { v = ov; i = oi; }
            }
            red_i[0] = i; red_v[0] = v;
        }
        wg_sync();
        return red_i[0];
    };
    const int best = reduce_first(vmin, imin, true);
    const double chi2_min = red_v[0];
    const int peak = reduce_first(vmax, imax, false);
    (void)reduce_first(cmax, 0, false);
    const double chi2_max = red_v[0];
    if (tid == 0) {
        double* o = a.out + c * 8;
        const int b = best < n ? best : 0, pk = peak < n ? peak : 0;   // (all-NaN input: index 0, like numpy)
        o[0] = chi2_min; o[1] = (double)b; o[2] = (double)pk;
        o[3] = a.periods[pk]; o[4] = a.depth[c * n + pk]; o[5] = (double)a.row[c * n + b];
        o[6] = chi2_max == chi2_min ? 1.0 : 0.0; o[7] = 0.0;
    }
}

// The final T0 fit's choice (stats.py:196-201): the trial epoch of the FIRST strict minimum of the residuals, or 0
// when none is below +inf.  One workgroup per light curve; fit c has n_epochs[c] residuals at residuals + c * stride.
struct FirstMinArgs {
    const double* residuals; const double* epochs;   // [n_curves] stride `stride`
    const int* n_epochs;                             // [n_curves]
    double* T0;                                      // [n_curves]
    long long stride;
};
__global__ void __launch_bounds__(1024) tls_first_min(const FirstMinArgs a) {
    __shared__ double red_v[kMaxWaves];
    __shared__ int red_i[kMaxWaves];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nt / kWave;
    const long long c = blockIdx.x;
    const int n = a.n_epochs[c];
    const double* r = a.residuals + c * a.stride;
    double v = INFINITY; int i = 0x7fffffff;
    for (int k = tid; k < n; k += nt) { const double x = r[k]; if (x < v) { v = x; i = k; } }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        const double ov = __shfl_down(v, d, kWave);
        const int oi = __shfl_down(i, d, kWave);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    if (lane == 0) { red_v[wave] = v; red_i[wave] = i; }
    wg_sync();
    if (tid == 0) {
        for (int w = 1; w < nw; ++w) if (red_v[w] < v || (red_v[w] == v && red_i[w] < i)) { v = red_v[w]; i = red_i[w]; }
        a.T0[c] = (v < INFINITY && i < n) ? a.epochs[c * a.stride + i] : 0.0;
    }
}

// What the host did between the search and the final T0 fit of a light curve, on the device (stats.py:135-152 as
// tls_power_batch's host loop wrote it): from the pick of tls_power_pick -- period and depth at the power peak, template row
// at the chi^2 minimum -- the fit's parameters, its trial epochs numpy.linspace(min t, min t + period, points) (arange * step +
// start, two roundings, the end point set exactly) and the template scaled to the fitted depth, 1 - (1 - signal) / (SIGNAL_DEPTH
// / (1 - depth)).  One workgroup per light curve.  No host round trip: the fit is enqueued right behind.
struct PrepArgs {
    const double* pick;             // [n_curves][8] (tls_power_pick)
    const WidthEntry* widths; int n_widths;
    const double* q;                // 1 - template rows, the plan's layout
    double* signal; long long signal_stride;
    double* epochs; long long epoch_stride;
    T0FitParams* params;            // [n_curves]
    int* n_epochs;                  // [n_curves] (tls_first_min reads them)
    double t_min, margin;
    int n;
};
__global__ void __launch_bounds__(256) tls_power_prep(const PrepArgs a) {
    __shared__ int s_k, s_points;
    __shared__ double s_scale, s_step, s_stop;
    const int tid = threadIdx.x, nt = blockDim.x;
    const long long c = blockIdx.x;
    const double* pk = a.pick + 8 * c;
    if (tid == 0) {
        int points = 0, k_row = 0;
        if (pk[6] == 0.0) {   // a transit was fit (main.py:203-216 otherwise: T0 = 0)
            const int best_row = (int)pk[5];
            for (int k = 0; k < a.n_widths; ++k) if (a.widths[k].row == best_row) { k_row = k; break; }   // (reported rows are the first row of their width)
            const int dur = a.widths[k_row].q_len;
            if (a.margin == 0) points = a.n;
            else points = (int)((double)a.n / (a.margin * (double)dur));   // stats.py:150
            if (points > a.n) points = a.n;
            if (points < 0) points = 0;
            const double depth = pk[4], period = pk[3];
            s_scale = 0.5 / (1 - depth);   // SIGNAL_DEPTH / (1 - depth), stats.py:142
            s_stop = a.t_min + period;
            s_step = points > 1 ? (s_stop - a.t_min) / (double)(points - 1) : 0.0;
            T0FitParams fp;
            fp.period = period; fp.dur = dur; fp.roll = (dur / 2 + 1) % a.n; fp.n_epochs = points; fp.pad_ = 0;
            a.params[c] = fp;
        } else {
            T0FitParams fp;
            fp.period = 1.0; fp.dur = 0; fp.roll = 0; fp.n_epochs = 0; fp.pad_ = 0;
            a.params[c] = fp;
        }
        a.n_epochs[c] = points;
        s_k = k_row; s_points = points;
    }
    wg_sync();
    const int points = s_points;
    if (points == 0) return;
    const int dur = a.widths[s_k].q_len;
    const double* qr = a.q + a.widths[s_k].q_offset;
    double* sig = a.signal + c * a.signal_stride;
    for (int j = tid; j < dur; j += nt) sig[j] = 1 - (qr[j] / s_scale);   // (q = 1 - signal as the host formed it)
    double* ep = a.epochs + c * a.epoch_stride;
    if (points == 1) { if (tid == 0) ep[0] = a.t_min; return; }
    for (int i = tid; i < points; i += nt) {
        double prod = (double)i * s_step;
        asm volatile("" : "+v"(prod));   // two roundings, like numpy's arange * step + start: the product must not fuse into the sum
        ep[i] = prod + a.t_min;
    }
    wg_sync();
    if (tid == 0) ep[points - 1] = s_stop;
}

// Pink noise (reference stats.py:72-77): term i = numpy.std(data[i : i + width]) / width ** 0.5, one thread a window.
// numpy.std = sqrt(sum((x - sum(x) / n)^2) / n) with both sums in numpy's pairwise association (loops_utils.h.src,
// pairwise_sum: fewer than 8 elements from the left starting at -0.0; up to 128 as eight interleaved partial sums, combined
// ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), the tail from the left; longer runs halved at a multiple of 8).
template <typename At>
__device__ __forceinline__ double numpy_pairwise_leaf(const At& at, int lo, int n) {   // n <= 128
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res += at(lo + i);
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = at(lo + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += at(lo + i + j);
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += at(lo + i);
    return res;
}
// (numpy recurses; here the frames sit in a fixed array -- a run is at least halved per level, 28 levels cover 2^31 elements --
// so the kernel needs no dynamic stack)
template <typename At>
__device__ double numpy_pairwise_sum(const At& at, int lo0, int n0) {
    constexpr int kDepth = 28;
    int f_lo[kDepth], f_n[kDepth], f_stage[kDepth];
    double f_left[kDepth];
    int sp = 0;
    double ret = 0.0;
    f_lo[0] = lo0; f_n[0] = n0; f_stage[0] = 0;
    while (sp >= 0) {
        const int lo = f_lo[sp], n = f_n[sp];
        int half = n / 2;
        half -= half % 8;
        if (f_stage[sp] == 0) {
            if (n <= 128 || sp + 1 >= kDepth) { ret = numpy_pairwise_leaf(at, lo, n <= 128 ? n : 128); --sp; continue; }
            f_stage[sp] = 1;
            ++sp; f_lo[sp] = lo; f_n[sp] = half; f_stage[sp] = 0;
        } else if (f_stage[sp] == 1) {
            f_left[sp] = ret; f_stage[sp] = 2;
            ++sp; f_lo[sp] = lo + half; f_n[sp] = n - half; f_stage[sp] = 0;
        } else {
            ret = f_left[sp] + ret;
            --sp;
        }
    }
    return ret;
}
__global__ void __launch_bounds__(256) tls_pink_terms(const double* data, int n_windows, int width, double root_width, double* terms) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_windows) return;
    const double* w = data + i;
    const double wd = (double)width;
    const double mean = numpy_pairwise_sum([w](int k) { return w[k]; }, 0, width) / wd;
    // (the square is rounded before it is added, as numpy.multiply leaves it: hipcc would contract it into the sum's FMA)
    const double var = numpy_pairwise_sum([w, mean](int k) { const double d = w[k] - mean; double sq = d * d; asm volatile("" : "+v"(sq)); return sq; }, 0, width) / wd;
    terms[i] = sqrt(var) / root_width;
}

// Developer/test entry: the exact sequential cumsum on an arbitrary non-negative series
// (one workgroup, global memory).  out has count + 1 entries.
__global__ void __launch_bounds__(1024) tls_cumsum_kernel(const double* f, double* out, int count, int variant,
                                                          unsigned long long* dbg) {
    __shared__ __attribute__((aligned(16))) unsigned char scratch[kCumsumScratchBytes];
    if (variant == 0) exact_cumsum<false, false>(f, out, count, reinterpret_cast<Cumsum2Scratch*>(scratch), dbg);
    else exact_sequential_cumsum(f, out, count, reinterpret_cast<CumsumScratch*>(scratch));
}

}  // namespace tlsdev
