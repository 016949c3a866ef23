// tls_slim_kernel.hip.h -- the LDS-resident search with FOUR period slots per CU (included by tls_kernels.hip.h, inside
// namespace tlsdev).
//
// Why.  The LDS-resident kernel (tls_search_kernel<RESIDENT = true>) keeps e = 1 - flux and its running sum X side by side:
// 16 bytes a sample, 81 KB for the 90-day configuration, two 512-thread workgroups per CU.  Round 5 measured what bounds it
// (PERF_LOG.md): a workgroup alone on its CU issues 17 % of the time -- a period is a chain of dependent round trips --, and a
// 42-day series, which fits the LDS four times, ran 11.5 % faster as four 256-thread workgroups per CU than as two 512-thread
// ones: periods in flight hide each other's latency better than waves of one period do.  Four slots need <= 40 KB a period:
//   * phase 3 runs on X ALONE.  The depth predicate reads X anyway; the dot products run on it too -- summation by parts,
//     sum_j q_j e_{i+j} = sum_{j<=L} g_j X_{i+j} with the difference taps g (as the HBM-slab variant's fast mode does);
//   * the sort orders 32-bit phase keys held in registers (ties: the exact phases, then the index -- numpy's stable order),
//     two points to a bucket, its records inside the region X takes afterwards; the permutation waits in registers while
//     the flux is gathered over its LDS home;
//   * the prefix sum runs in place (fast mode: a plain scan; exact mode: exact_cumsum's aliased form, f shifted by one).
// Uniform weights, no pruning, no fp32 screen (the host takes the other kernel for those); survey batches share the sort of a
// period as there.  Cells, predicate, tie rule and reductions are the other kernel's code (consider_cells, settle_best, ...);
// values differ from it by the rounding of the difference taps (chi^2 to ~1e-13).
//
// Reference mapping as tls_search_body.inc.h: core.py:15-18,113-188, helpers.py:70-73.
// Two launch shapes (round 6): 256-thread workgroups, four (three) to a CU, for series up to 5120 points; 512-thread
// workgroups, two to a CU, for series up to 10 240 points whose region fits half the LDS (N + W <= ~9.9 k: 107-200 d at
// 30 min, two TESS sectors at 10 min) -- the classic kernel needs two regions and runs those as ONE 1024-thread workgroup per
// CU.  The shape is a template parameter (THREADS); a sort record holds 13 index bits for the first, 14 for the second.
constexpr int kSlimThreads = 256;     // the smaller shape (host: the look-out for piles, the default)
constexpr int kSlimThreadsWide = 512;
constexpr int kSlimPer = 20;          // samples of the folded order a thread keeps in registers across the gather
__host__ __device__ constexpr int slim_idx_bits(int threads) { return threads <= 256 ? 13 : 14; }
constexpr int kSlimScratchBytes = 1328;   // >= sizeof(Cumsum2Scratch); the per-row tables share its first bytes
#ifndef TLS_SLIM_TAIL_MAX
#define TLS_SLIM_TAIL_MAX 20
#endif
#ifndef TLS_SLIM_TAIL_SINGLES
#define TLS_SLIM_TAIL_SINGLES 64
#endif
constexpr int kSlimTailMax = TLS_SLIM_TAIL_MAX;          // a row's last, mostly empty batch is re-listed window by window up to this many units
constexpr int kSlimTailSingles = TLS_SLIM_TAIL_SINGLES;  // ... if that leaves at most this many windows
#ifndef TLS_SLIM_BIG
#define TLS_SLIM_BIG 8
#endif
constexpr int kSlimBig = TLS_SLIM_BIG;            // a bucket beyond this many points (a commensurate period's pile) is ranked by the workgroup on exact phases
constexpr int kSlimPileMembers = 3;     // members of one pile a thread ranks, at most
constexpr int kSlimStageBytes = 2048;  // at least this much of the region stays free for the phases of such a pile
// (a sort record: sub-bucket key (19 or 18 bits) | original index (13 or 14 bits))
__host__ __device__ constexpr int slim_header_bytes(int threads) { return 128 + (threads / kWave) * 24 + 48 + kSlimScratchBytes; }   // wsum | wbest | s_work | scratch
static_assert(slim_header_bytes(kSlimThreads) % 16 == 0 && slim_header_bytes(kSlimThreadsWide) % 16 == 0, "the region behind the header holds doubles read in pairs");
static_assert(sizeof(Cumsum2Scratch) <= kSlimScratchBytes, "exact_cumsum's scratch does not fit the slim header");
// sort buckets: as many as the region holds beside the records and the order, at most one per two points
// (measured: giving the pile stage 4800 bytes at the price of 3 % fewer buckets cost the ordinary periods 2 %)
__host__ __device__ inline int slim_buckets(int n, int RS) {
    const long long room = (8LL * RS - 6LL * n - kSlimStageBytes) / 4;   // records 4n | bucket counters 4nb | pile stage | order 2n (the region's last bytes)
    const long long want = n / 2 > 16 ? n / 2 : 16;
    return (int)(room < want ? room : want);
}
// what the kernel needs of the LDS for a plan, 0 when the plan does not fit it
__host__ __device__ inline long long slim_lds_bytes(int n, int M, int region_pad, int n_widths, int threads = kSlimThreads) {
    const int RS = M + 1 + region_pad;
    if (n > threads * kSlimPer || n >= (1 << slim_idx_bits(threads)) || n < 64) return 0;
    if (slim_buckets(n, RS) < n / 8 || slim_buckets(n, RS) < 16) return 0;
    if (4LL * (3 * n_widths + 2) > kSlimScratchBytes) return 0;
    return slim_header_bytes(threads) + 8LL * RS;
}

// The piles of a commensurate period: buckets beyond kSlimBig points whose records tie on their key bits (slim_is_pile).
// Every member's exact phase is formed ONCE into a stage, then every member counts the members in front of it -- (phase,
// index), numpy's stable order.  Four, two or one pile at a time (what the largest one leaves of the stage), a group of threads
// each.  Out of line: its registers are its own (called by every thread of the workgroup: it holds barriers).
__device__ __forceinline__ bool slim_is_pile(const unsigned int* recs, int lo, int hi, int stage_cap, int idx_bits) {
    const int size = hi - lo;
    if (size <= kSlimBig || size > stage_cap) return false;
    const unsigned int r0 = recs[lo] >> idx_bits, rm = recs[lo + size / 2] >> idx_bits, r1 = recs[hi - 1] >> idx_bits;
    return r0 == rm || rm == r1;
}
// (the pointers are kept in their address spaces by TYPE: through generic pointers the member loops came out as FLAT loads)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* slim_lds_ptr(T* p) {
    typedef __attribute__((address_space(3))) T* lds_t;
    const unsigned int v = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(uintptr_t)(lds_t)p);
    return (lds_t)(uintptr_t)v;
}
__device__ __noinline__ void slim_rank_piles(const double* t_, double period_, const unsigned int* recs_, const unsigned int* cnt_,
                                             unsigned short* perm_, double* stage_, int stage_cap_, const unsigned short* big_list_, int n_big_, int* flags_, int nb_, unsigned long long* clk_, int idx_bits_) {
    typedef __attribute__((address_space(1))) const double* glob_f64;
    typedef __attribute__((address_space(3))) const unsigned int* lds_u32;
    typedef __attribute__((address_space(3))) unsigned short* lds_u16;
    typedef __attribute__((address_space(3))) const unsigned short* lds_cu16;
    typedef __attribute__((address_space(3))) double* lds_f64;
    const glob_f64 t = (glob_f64)global_arg(t_);
    const double period = uniform_f64(period_);
    const lds_u32 recs = slim_lds_ptr(recs_);
    const lds_u32 cnt = slim_lds_ptr(cnt_);
    const lds_u16 perm = slim_lds_ptr(perm_);
    const lds_f64 stage = slim_lds_ptr(stage_);
    const lds_cu16 big_list = slim_lds_ptr(big_list_);
    __attribute__((address_space(3))) int* const flags = slim_lds_ptr(flags_);
    const int stage_cap = uniform_i32(stage_cap_), n_big = uniform_i32(n_big_), nb = uniform_i32(nb_);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int idx_bits = uniform_i32(idx_bits_);
    const unsigned int kIdx = (1u << idx_bits) - 1u;
    // the largest pile decides how many are ranked side by side, a group of threads and a share of the stage each
    // (every thread looks at its share of the list; the sizes meet in an LDS word the caller has zeroed)
#if TLS_PHASE_CLOCKS
    long long c_last = clock64();
#define SLIM_PILE_MARK(slot) do { if (clk_ && threadIdx.x == 0) { const long long now_ = clock64(); atomicAdd(&clk_[slot], (unsigned long long)(now_ - c_last)); c_last = now_; } } while (0)
#else
#define SLIM_PILE_MARK(slot) do { } while (0)
#endif
    int m_mine = 0;
    for (int gb = tid; gb < n_big; gb += nt) {
        const int b = (int)big_list[gb];
        m_mine = max(m_mine, (int)cnt[b] - (b ? (int)cnt[b - 1] : 0));
    }
    if (m_mine) __hip_atomic_fetch_max(flags, m_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    wg_sync();
    const int m_max = __builtin_amdgcn_readfirstlane(*flags);
    // as many piles side by side as the largest leaves room for: its members fit the group's share of the stage and its
    // threads' registers (a round is a chain of dependent round trips whatever it ranks: few rounds matter more than wide groups)
    int groups = 16;
    while (groups > 1 && (m_max > stage_cap / groups || m_max > kSlimPileMembers * (nt / groups))) groups >>= 1;
    const int per = nt / groups, group = tid / per, first = tid - group * per;
    const lds_f64 mine = stage + group * (stage_cap / groups);
    const __attribute__((address_space(3))) unsigned long long* const keys = (const __attribute__((address_space(3))) unsigned long long*)mine;
    // A phase is >= 0: its bit pattern orders like its value.  Every phase of bucket b is >= b / nb, and from the eighth
    // bucket on the patterns of a bucket span less than 2^50: (pattern - pattern of a double just below b / nb) and the
    // 13-bit index make ONE 64-bit sort key, distinct for every member -- one compare a pair.  The first buckets (phases
    // down to 0: patterns of every exponent) compare pattern and index separately.
    // The keys of the NEXT round's piles are formed (global loads, divisions) while this round's are ranked.
    constexpr int kMine = kSlimPileMembers;   // members of a pile per thread, at most (the caller caps the stage at kMine * nt entries)
    unsigned long long ahead[kMine];
    double ahead_t[kMine];
    unsigned int ahead_rec[kMine];
    // (in two halves around the ranking: a wavefront issues in order, so the time stamps are REQUESTED before it and first
    // touched behind it)
    auto request = [&](int g0, int& lo, int& m, int& b) {
        const int gb = g0 + group;
        b = 0; lo = 0; m = 0;
        if (gb < n_big) {
            b = (int)big_list[gb];
            lo = b ? (int)cnt[b - 1] : 0;
            m = (int)cnt[b] - lo;
        }
#pragma unroll
        for (int a = 0; a < kMine; ++a) {
            const int j = first + a * per;
            ahead_rec[a] = j < m ? recs[lo + j] : 0u;
            ahead_t[a] = t[ahead_rec[a] & kIdx];
        }
    };
    auto form_keys = [&](int b, bool& packed) {
        packed = b >= 8;
        const unsigned long long base = packed ? (unsigned long long)__double_as_longlong((double)b / (double)nb) - 4ull : 0ull;
#pragma unroll
        for (int a = 0; a < kMine; ++a) {
            const unsigned long long pattern = (unsigned long long)__double_as_longlong(fold_phase(ahead_t[a], period, 0.0));
            ahead[a] = packed ? ((pattern - base) << idx_bits) | (ahead_rec[a] & kIdx) : pattern;
        }
    };
    int lo = 0, m = 0, lo_next = 0, m_next = 0, b_next = 0;
    bool packed = false, packed_next = false;
    SLIM_PILE_MARK(14);
    request(0, lo_next, m_next, b_next);
    form_keys(b_next, packed_next);
    SLIM_PILE_MARK(15);
    for (int g0 = 0; g0 < n_big; g0 += groups) {   // (the same trips for every thread: the barriers below are the workgroup's)
        lo = lo_next; m = m_next; packed = packed_next;
#pragma unroll
        for (int a = 0; a < kMine; ++a) {
            const int j = first + a * per;
            if (j < m) ((__attribute__((address_space(3))) unsigned long long*)mine)[j] = ahead[a];
        }
        lds_barrier();
        const bool more = g0 + groups < n_big;
        if (more) request(g0 + groups, lo_next, m_next, b_next);   // (in flight across the ranking below)
        SLIM_PILE_MARK(15);
        constexpr int kB = 16;   // members read per step (the lanes of a group read the same words: broadcasts)
        if (packed) {
            for (int j = first; j < m; j += per) {
                const unsigned long long key = keys[j];
                unsigned int rank = 0;
                for (int u0 = 0; u0 < m; u0 += kB) {
                    unsigned long long k2[kB];
#pragma unroll
                    for (int u = 0; u < kB; ++u) k2[u] = keys[u0 + u < m ? u0 + u : j];   // (past the pile: the member itself, never in front of itself)
#pragma unroll
                    for (int u = 0; u < kB; ++u) rank += (unsigned int)(k2[u] < key);
                }
                perm[lo + rank] = (unsigned short)(key & kIdx);
            }
        } else {
            for (int j = first; j < m; j += per) {
                const unsigned int i = recs[lo + j] & kIdx;
                const unsigned long long ph = keys[j];
                unsigned int rank = 0;
                for (int u0 = 0; u0 < m; u0 += kB) {
                    unsigned long long ph2[kB];
                    unsigned int r2[kB];
#pragma unroll
                    for (int u = 0; u < kB; ++u) { const int uu = u0 + u < m ? u0 + u : m - 1; ph2[u] = keys[uu]; r2[u] = recs[lo + uu] & kIdx; }
#pragma unroll
                    for (int u = 0; u < kB; ++u) {
                        const unsigned int before = (unsigned int)(ph2[u] < ph) | ((unsigned int)(ph2[u] == ph) & (unsigned int)(r2[u] < i));
                        rank += before & (unsigned int)(u0 + u < m);
                    }
                }
                perm[lo + rank] = (unsigned short)i;
            }
        }
        SLIM_PILE_MARK(16);
        if (more) form_keys(b_next, packed_next);
        lds_barrier();   // (the next piles' keys go over these)
        SLIM_PILE_MARK(17);
    }
}

// a value the compiler must not carry across phases in a register (or a spill slot): per-thread indices tid + j * nt are
// cheaper to form again than to keep
__device__ __forceinline__ int slim_fresh(int v) { asm volatile("" : "+v"(v)); return v; }
// two sort records whose 19 key bits tie: the exact phases decide, then the index (out of line: one pair in 1e8)
__device__ __noinline__ int slim_tie_before(const double* t, double period, int i_other, int i_mine) {
    const double ph = fold_phase(t[i_mine], period, 0.0), ph2 = fold_phase(t[i_other], period, 0.0);
    return (ph2 < ph || (ph2 == ph && i_other < i_mine)) ? 1 : 0;
}

// Phase 1 of a period: fold + stable sort by phase (core.py:119-120) -> perm[k] = original index of the k-th folded point, in
// the last 2n bytes of the region.
template <int THREADS>
__device__ __forceinline__ void slim_fold_and_sort(const double* t, int n, double period, int RS, double* X, unsigned int* wsum, int* s_work,
                                                unsigned char* scratch, PhaseClock& pc) {
    constexpr int nt = THREADS;
    constexpr int kSlimIdxBits = slim_idx_bits(THREADS);
    const int tid = threadIdx.x;
    const int nb = slim_buckets(n, RS);
    unsigned int* recs = reinterpret_cast<unsigned int*>(X);                           // [n] sort records, bucket by bucket
    unsigned int* cnt = recs + n;                                                      // [nb]
    unsigned short* perm = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(X) + 8LL * RS) - n;   // [n], the region's last bytes
    // ---- phase 1: fold + stable sort by phase (core.py:119-120), on 32-bit keys -------------------
    // A point's key stays in a register from the fold to its rank.  Bucket = floor(key * nb / 2^32); INSIDE a bucket the
    // low word of key * nb is monotone in the key: its top 19 bits and the index make the point's sort record.  Records
    // are scattered bucket by bucket (any order inside one); a point's place is its bucket's start plus the records of
    // the bucket in front of it -- decided by the records, and by the exact phases (then the index: numpy's stable
    // order) for the pairs whose 19 bits tie: two phases within 2^-30, or the piled-up phases of a commensurate period.
    for (int b = tid; b < nb; b += nt) cnt[b] = 0;
    if (tid == 0) { s_work[5] = 0; s_work[6] = 0; }
    wg_sync();
    unsigned int key[kSlimPer];
    {
        // every time stamp of the thread is requested before the first is used: ONE L2 round trip per period, not one per
        // group of divisions (nothing else is alive in registers at a period's start)
#ifndef TLS_SLIM_FOLD_DEPTH
#define TLS_SLIM_FOLD_DEPTH kSlimPer
#endif
        constexpr int kF = TLS_SLIM_FOLD_DEPTH;
        static_assert(kSlimPer % kF == 0, "the fold takes kF time stamps per step");
#pragma unroll
        for (int j0 = 0; j0 < kSlimPer; j0 += kF) {
            const int base = slim_fresh(tid) + j0 * nt;
            double tv[kF];
#pragma unroll
            for (int j = 0; j < kF; ++j) { const int i = base + j * nt; tv[j] = t[i < n ? i : 0]; }
#pragma unroll
            for (int j = 0; j < kF; ++j) {
                const int i = base + j * nt;
                key[j0 + j] = phase_key(fold_phase(tv[j], period, 0.0));
                if (i < n) atomicAdd(&cnt[__umulhi(key[j0 + j], (unsigned int)nb)], 1u);
            }
        }
    }
    wg_sync();
    pc.mark(0);
    block_exclusive_scan(cnt, nb, wsum);
    pc.mark(1);
#pragma unroll
    for (int j = 0; j < kSlimPer; ++j) {
        const int i = slim_fresh(tid) + j * nt;
        if (i < n) {
            const unsigned int slot = atomicAdd(&cnt[__umulhi(key[j], (unsigned int)nb)], 1u);
            recs[slot] = ((key[j] * (unsigned int)nb) & ~((1u << kSlimIdxBits) - 1u)) | (unsigned int)i;
        }
    }
    wg_sync();
    pc.mark(2);
    // piles: a period commensurate with the cadence folds the series onto a few dozen phase values; ranking a pile record
    // by record would be quadratic in exact-phase comparisons (two divisions each).  A bucket beyond kSlimBig points whose
    // records tie on their key bits (slim_is_pile: every member evaluates the same test) is left out here, listed by the
    // member whose record heads it, and ranked below on exact phases formed once (slim_rank_piles).  A bucket that is merely
    // full -- a NEARLY commensurate period: many points, distinct keys -- is ranked here like any other.
    unsigned short* big_list = reinterpret_cast<unsigned short*>(scratch);
    const int stage_off = (4 * n + 4 * nb + 7) & ~7;
    const int stage_room = (8 * RS - 2 * n - stage_off) / 8;
    const int stage_cap = stage_room < kSlimPileMembers * nt ? stage_room : kSlimPileMembers * nt;   // (a thread keeps that many keys in registers)
    double* stage = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(X) + stage_off);
    {
        // (cnt[b] is now the END of bucket b)
        auto before = [&](unsigned int other, unsigned int mine) -> int {
            if (((other ^ mine) >> kSlimIdxBits) != 0u) return other < mine ? 1 : 0;
            if (other == mine) return 0;
            return slim_tie_before(t, period, (int)(other & ((1u << kSlimIdxBits) - 1u)), (int)(mine & ((1u << kSlimIdxBits) - 1u)));
        };
        constexpr int kG = 5, kWin = 4;   // points ranked together; records of a bucket read up front
        static_assert(kSlimPer % kG == 0, "the rank takes kG points per step");
        static_assert(kSlimPer <= 32, "one bit per point of the thread");
        unsigned int heads = 0;
#pragma unroll
        for (int j0 = 0; j0 < kSlimPer; j0 += kG) {
            int lo[kG], hi[kG];
            unsigned int mine[kG], win[kG][kWin];
            const int base = slim_fresh(tid) + j0 * nt;
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                const int i = base + g * nt;
                const unsigned int b = __umulhi(key[j0 + g], (unsigned int)nb);
                mine[g] = ((key[j0 + g] * (unsigned int)nb) & ~((1u << kSlimIdxBits) - 1u)) | (unsigned int)i;
                lo[g] = b ? (int)cnt[b - 1] : 0;
                hi[g] = i < n ? (int)cnt[b] : lo[g];
                if (hi[g] - lo[g] > kSlimBig) {   // (rare)
                    if (slim_is_pile(recs, lo[g], hi[g], stage_cap, kSlimIdxBits)) {
                        if (recs[lo[g]] == mine[g]) heads |= 1u << (j0 + g);   // the member whose record heads the bucket lists it
                        hi[g] = lo[g];   // ranked below: nothing to do here
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < kG; ++g)
#pragma unroll
                for (int u = 0; u < kWin; ++u) win[g][u] = recs[lo[g] + u < hi[g] ? lo[g] + u : lo[g]];
#pragma unroll
            for (int g = 0; g < kG; ++g) {
                int rank = 0;
#pragma unroll
                for (int u = 0; u < kWin; ++u) if (lo[g] + u < hi[g]) rank += before(win[g][u], mine[g]);
                for (int s2 = lo[g] + kWin; s2 < hi[g]; ++s2) rank += before(recs[s2], mine[g]);
                if (lo[g] < hi[g]) perm[lo[g] + rank] = (unsigned short)(mine[g] & ((1u << kSlimIdxBits) - 1u));
            }
        }
        if (heads != 0u) {   // (rare)
#pragma unroll
            for (int j = 0; j < kSlimPer; ++j) {
                if ((heads >> j) & 1u) {
                    const int at = atomicAdd(&s_work[5], 1);
                    if (at < kSlimScratchBytes / 2) big_list[at] = (unsigned short)__umulhi(key[j], (unsigned int)nb);   // (n / kSlimBig < 664 piles: always)
                }
            }
        }
    }
    wg_sync();
    {
        const int n_big = __builtin_amdgcn_readfirstlane(s_work[5]);
        pc.mark(8);
        if (n_big > 0) slim_rank_piles(t, period, recs, cnt, perm, stage, stage_cap, big_list, n_big, &s_work[6], nb, pc.out, kSlimIdxBits);
    }
}

template <bool COUNTING, int THREADS = kSlimThreads>
__global__ void __launch_bounds__(THREADS, 4)
tls_slim_kernel(const SearchArgs) {
    constexpr bool UNIFORM_W = true;
    constexpr int kSlimWaves = THREADS / kWave;
    args_ptr ap = (args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    constexpr int nt = THREADS, nw = kSlimWaves;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int n = ap->n, W = ap->W, M = ap->M;
    const int region_pad = ap->region_pad;
    const int RS = M + 1 + region_pad;

    // ---- LDS carve-up ----------------------------------------------------------------------------
    unsigned int* wsum = reinterpret_cast<unsigned int*>(smem);                       // 32 words
    Best* wbest = reinterpret_cast<Best*>(smem + 128);                                 // kSlimWaves x 24 B
    int* s_work = reinterpret_cast<int*>(smem + 128 + kSlimWaves * sizeof(Best));     // [12]
    unsigned char* scratch = smem + 128 + kSlimWaves * sizeof(Best) + 48;             // kSlimScratchBytes
    RowTables rt;   // (inside the scratch: written after the prefix sum, which is the scratch's other user)
    rt.live = reinterpret_cast<unsigned int*>(scratch);
    rt.singles = rt.live + ap->n_widths;
    rt.batch_start = rt.singles + ap->n_widths;
    rt.next_batch = rt.batch_start + (ap->n_widths + 1);
    double* X = reinterpret_cast<double*>(smem + slim_header_bytes(THREADS));          // f, then X: RS doubles
    unsigned short* perm = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(X) + 8LL * RS) - n;   // [n], the region's last bytes
    unsigned int* chunk_list = ap->chunk_lists + (long long)blockIdx.x * ap->list_stride;
    if (tid == 0) {
        TLS_CHECK(*ap, slim_lds_bytes(n, M, region_pad, ap->n_widths, THREADS) != 0 && slim_lds_bytes(n, M, region_pad, ap->n_widths, THREADS) <= ap->lds_bytes && 6LL * n + 4LL * slim_buckets(n, RS) <= 8LL * RS, kChkLdsCarve);
    }
    const const_width_ptr widths_c = (const_width_ptr)ap->widths;
    const const_rows_ptr rows_c = (const_rows_ptr)ap->rows;
    const const_f64_ptr g_all = (const_f64_ptr)ap->g;
    const double dmin = ap->depth_min;

    bool retry_exact = false;
    // Fast mode: a window inside the undecided band of the depth predicate is NOTED (band_window) and decided after the
    // attempt on the period's exact prefix sum -- the period loop is entered a second time for the sort, the gather and the
    // exact prefix pass only (`resolve_band`); the lanes' leads and counts of the attempt are kept.  (Cells are VALUED on the
    // plain scan throughout: the dot products by parts want a prefix sum rounded at the size of e, not of k.)
    bool resolve_band = false;
    Lead kept_lead = no_lead();
    unsigned int kept_eval = 0;
    unsigned long long kept_steps = 0, kept_issued = 0;
    BandEntry* const band_list = reinterpret_cast<BandEntry*>(chunk_list + ap->list_cap);   // (the pruning variant's second list: idle here)
    int work = 0;
    for (;;) {
        if (!retry_exact) {
            if (tid == 0) { s_work[0] = (int)atomicAdd(ap->queue, 1u); s_work[1] = 0; s_work[2] = 0; s_work[4] = 0; }
            wg_sync();
            work = __builtin_amdgcn_readfirstlane(s_work[0]);
            wg_sync();
        } else if (tid == 0) {
            s_work[1] = 0; s_work[2] = 0;
        }
        int flag_slot = 1;
        const bool period_exact = retry_exact || ap->exact_prefix != 0;
        retry_exact = false;
        bool curve_exact = false;
        if (work >= ap->n_periods) {
            if (tid == 0) {   // the last workgroup to leave rewinds the queue for the next launch
                __threadfence();
                if (atomicAdd(ap->queue + 1, 1u) == gridDim.x - 1) { atomicExch(ap->queue, 0u); atomicExch(ap->queue + 1, 0u); }
            }
            break;
        }
        const int p = ap->order[work];
        TLS_CHECK(*ap, p >= 0 && p < ap->n_periods, kChkWorkItem);
        const double period = ap->periods[p];
        // Wave priority: the phases that are chains of round trips (sort, gather, prefix sum, predicate, argmin) issue ahead of
        // the other workgroups' dot products, which fill whatever slots are left -- a latency-bound wave that waits for an
        // issue slot behind four FMAs a cycle loses more than the FMA-bound one that lets it pass.  Same box: 1.079 -> 1.046 ms
        // (the other way round: 1.103; the predicate at low priority too: 1.061).
#ifndef TLS_SLIM_PRIO
#define TLS_SLIM_PRIO 1
#endif
#if TLS_SLIM_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        long long t_period = 0;
        if (ap->period_cycles && tid == 0) t_period = clock64();
        PhaseClock pc;
        pc.start(ap->phase_cycles);

        // ---- phase 1: fold + stable sort by phase (core.py:119-120), on 32-bit keys (slim_fold_and_sort) -------------------
        slim_fold_and_sort<THREADS>(ap->t, n, period, RS, X, wsum, s_work, scratch, pc);
        pc.mark(3);
        // survey mode: the permutation outlives the light curves of the batch in global memory
        const unsigned short* perm_g = nullptr;
        if (ap->n_curves > 1 || resolve_band) {   // (resolution evaluates the few windows that pass from the flux in global memory)
            unsigned short* pg = reinterpret_cast<unsigned short*>(ap->perm_scratch + (long long)blockIdx.x * n);
            for (int k = tid; k < n; k += nt) pg[k] = perm[k];
            perm_g = pg;
            wg_sync();
        }
        for (int curve = 0; curve < ap->n_curves; ++curve) {
        const bool exact_mode = period_exact || curve_exact;
        curve_exact = false;
        DepthRule rule;
        rule.dmin = ap->depth_min; rule.eps = exact_mode ? 1e-15 : ap->eps_fast; rule.exact_mode = exact_mode;
        rule.band_count = nullptr; rule.band_list = nullptr;
        if (!exact_mode && ap->list_cap >= 4 * kBandCap) {   // (16 bytes an entry in the idle list region)
            rule.band_count = reinterpret_cast<unsigned int*>(&s_work[4]); rule.band_list = band_list;
        }
        const bool resolving = resolve_band;   // (exact mode: this pass decides the windows the attempt noted, nothing else)
        resolve_band = false;
        rule.reach = (rule.dmin - rule.eps > 4e-15) ? fmin(fmax(1e-9, 4e-15 / (rule.dmin - rule.eps)), 1.0) : 1.0;
        bool undecided = false;
        const double* y_c = ap->y + (long long)curve * n;
        // ---- phase 2: gather (core.py:121-123), patch (core.py:126), prefix sum (helpers.py:72) -- all in the one region ----
        {
            // the folded order into registers first: the flux lands on the order's own LDS home
            int idx[kSlimPer];
            const int tid_g = slim_fresh(tid);
#pragma unroll
            for (int j = 0; j < kSlimPer; ++j) {
                const int k = tid_g + j * nt;
                idx[j] = k < n ? (perm_g ? (int)perm_g[k] : (int)perm[k]) : 0;
            }
            wg_sync();   // (every thread has its part of the order; global reads of perm_g included)
            double* fdst = exact_mode ? X + 1 : X;   // exact mode: C[k+1] goes over f[k] (exact_cumsum's aliased form)
            double v[kSlimPer];
#pragma unroll
            for (int j = 0; j < kSlimPer; ++j) v[j] = y_c[idx[j]];
#pragma unroll
            for (int j = 0; j < kSlimPer; ++j) { const int k = tid_g + j * nt; if (k < n) fdst[k] = v[j]; }
            wg_sync();
            for (int k = tid; k < W; k += nt) fdst[n + k] = fdst[k];   // core.py:126
            wg_sync();
            pc.mark(4);
            if (!exact_mode) {
                // X[k] = sum of e over [0, k), e = 1 - f: a plain scan in place (every thread its own stretch)
                double* wtot = reinterpret_cast<double*>(scratch);
                int per = (M + nt - 1) / nt;
                if ((per & 1) == 0) per += 1;
                const int lo = tid * per < M ? tid * per : M;
                const int hi = lo + per < M ? lo + per : M;
                double local = 0.0;
                for (int k = lo; k < hi; ++k) local += 1.0 - X[k];
                const double incl = wave_inclusive_sum(local);
                if (lane == kWave - 1) wtot[wave] = incl;
                lds_barrier();
                double run = 0.0, total = 0.0;
                for (int u = 0; u < nw; ++u) { const double wv = wtot[u]; if (u < wave) run += wv; total += wv; }
                run += incl - local;
                for (int k = lo; k < hi; ++k) { const double e1 = 1.0 - X[k]; X[k] = run; run += e1; }
                if (tid == nt - 1) X[M] = total;
            } else {
                if (tid == 0) X[0] = 0.0;
                wg_sync();
                (void)exact_cumsum<true>(X + 1, X, M, reinterpret_cast<Cumsum2Scratch*>(scratch), ap->phase_cycles, 0.0);
                wg_sync();
                for (int k = tid; k <= M; k += nt) X[k] = (double)k - X[k];   // X = k - numpy.cumsum: an exact subtraction
            }
            for (int k = tid; k < region_pad; k += nt) X[M + 1 + k] = -(double)(k + 1) * 1.0e300;   // sentinels (see the other kernel)
            wg_sync();
            pc.mark(5);
        }
        const int k_lo = __builtin_amdgcn_readfirstlane(rows_c[p].k_lo);
        const int k_hi = __builtin_amdgcn_readfirstlane(rows_c[p].k_hi);
        const int k_x = __builtin_amdgcn_readfirstlane(rows_c[p].k_x);
        const int n_rows = k_hi - k_lo;
        for (int row = tid; row < n_rows; row += nt) rt.live[row] = 0;
        if (tid == 0) { s_work[3] = 0; if (!resolving) s_work[4] = 0; }
        wg_sync();

        Lead lead = resolving ? kept_lead : no_lead();
        unsigned int n_eval = resolving ? kept_eval : 0u;
        unsigned long long n_steps = resolving ? kept_steps : 0ull, n_issued = resolving ? kept_issued : 0ull;
        const double* c_base = X;
        if (!resolving) {
        // ---- phase 3a: depth predicate over every trial cell -> lists of live units (core.py:58) ----------
        const bool exact_u = __builtin_amdgcn_readfirstlane((int)exact_mode) != 0;
        const double thr_hi = rule.dmin + rule.eps, thr_lo = rule.dmin - rule.eps;
        if (k_x > k_lo) {
            const int units0 = widths_c[k_lo].n_chunks;
            const int n_dense = k_x - k_lo;
            for (int tile = wave; tile * kWave < units0; tile += nw) {
                const int unit = tile * kWave + lane;
                const int u0 = unit * kR;
                const int u0c = u0 < M + 1 ? u0 : M + 1;
                double c_lo[kR];
#pragma unroll
                for (int r = 0; r < kR; ++r) c_lo[r] = c_base[u0c + r];
                int row_lo = 0, row_hi = 0;
                unsigned long long band_mask = 0ull;
                const unsigned long long valid_mask = ballot64(unit < units0);
#ifndef TLS_SLIM_ROW_BATCH
#define TLS_SLIM_ROW_BATCH 2
#endif
                constexpr int kRowBatch = TLS_SLIM_ROW_BATCH;
                for (int k = k_lo; k < k_x; k += kRowBatch) {
                    int dv[kRowBatch];
                    double inv[kRowBatch], dC[kRowBatch];
                    double c_hi[kRowBatch][kR];
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        const int kk = k + j < k_x ? k + j : k_x - 1;
                        dv[j] = widths_c[kk].width;
                        inv[j] = widths_c[kk].inv_d;
                        const int hi0 = min(u0 + dv[j], M + 1);
#pragma unroll
                        for (int r = 0; r < kR; ++r) c_hi[j][r] = c_base[hi0 + r];
                    }
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        double m = c_hi[j][0] - c_lo[0];
#pragma unroll
                        for (int r = 1; r < kR; ++r) m = fmax(m, c_hi[j][r] - c_lo[r]);
                        dC[j] = m;
                    }
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        if (k + j < k_x) {
                            unsigned long long mask;
                            if (exact_u) {
                                bool und_j = false;
                                mask = ballot64(depth_pass(dC[j], inv[j], (double)dv[j], dmin, rule.eps, true, und_j));
                            } else {
                                const double m_fast = dC[j] * inv[j];
                                mask = ballot64(m_fast > thr_hi);
                                const unsigned long long band_j = ballot64(m_fast >= thr_lo) & ~mask & valid_mask;
                                if (band_j != 0ull) {   // (a scalar branch, rarely taken)
                                    if (rule.band_count != nullptr) {
                                        // the unit's deepest window is inside the band: its windows there are noted one by one
                                        if ((band_j >> lane) & 1ull) {
#pragma unroll
                                            for (int r = 0; r < kR; ++r) {
                                                const double dXr = c_hi[j][r] - c_lo[r];
                                                if (dXr * inv[j] >= thr_lo) band_window(rule, k + j, u0 + r, dXr, undecided);
                                            }
                                        }
                                    } else {
                                        band_mask |= band_j;
                                    }
                                }
                            }
                            mask &= valid_mask;
                            if (n_dense <= kWave) {
                                set_lane(row_lo, (int)(unsigned int)mask, k + j - k_lo);
                                set_lane(row_hi, (int)(unsigned int)(mask >> 32), k + j - k_lo);
                            } else {
                                push_live(((mask >> lane) & 1ull) != 0ull, (unsigned int)unit, &rt.live[k + j - k_lo],
                                          chunk_list + widths_c[k + j].list_base, lane);
                            }
                        }
                    }
                }
                undecided |= (band_mask & valid_mask) != 0ull;
                const unsigned long long row_mask = ((unsigned long long)(unsigned int)row_hi << 32) | (unsigned int)row_lo;
                if (n_dense <= kWave) {
                    unsigned int base = 0;
                    const unsigned int mine = (unsigned int)__popcll(row_mask);
                    if (mine) base = atomicAdd(&rt.live[lane], mine);
                    const unsigned long long rows_hit = ballot64(mine != 0u);
                    const unsigned long long below = (1ull << lane) - 1ull;
                    for (unsigned long long left = rows_hit; left; left &= left - 1ull) {
                        const int j = __ffsll((long long)left) - 1;
                        const unsigned long long mask = (unsigned long long)lane_value((long long)row_mask, j);
                        const unsigned int b0 = (unsigned int)lane_value((int)base, j);
                        TLS_CHECK(*ap, b0 + (unsigned int)__popcll(mask) <= (unsigned int)widths_c[k_lo + j].n_chunks, kChkListCap);
                        if ((mask >> lane) & 1ull)
                            chunk_list[widths_c[k_lo + j].list_base + b0 + (unsigned int)__popcll(mask & below)] = (unsigned int)unit;
                    }
                }
            }
        }
        pc.mark(13);
        // strided rows (long durations, core.py:50-58): one row per wave through a ticket counter
        for (;;) {
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&s_work[3], 1);
            const int k = (k_x > k_lo ? k_x : k_lo) + __builtin_amdgcn_readfirstlane(ticket);
            if (k >= k_hi) break;
            const int d = widths_c[k].width, xth = widths_c[k].xth, n_pos = widths_c[k].n_pos;
            const int n_units = widths_c[k].n_chunks;
            const double inv_d = widths_c[k].inv_d;
            unsigned int* list = chunk_list + widths_c[k].list_base;
            unsigned int n_listed = 0;
            const unsigned long long below = (1ull << lane) - 1ull;
            if (widths_c[k].tiled) {
                for (int tile = 0; tile * kWave < n_units; ++tile) {
                    const int unit = tile * kWave + lane;
                    const int uc = unit < n_units ? unit : n_units - 1;
                    const double* c0 = c_base + uc * kR * xth;
                    double c_lo[kR], c_hi[kR];
#pragma unroll
                    for (int r = 0; r < kR; ++r) { c_lo[r] = c0[r * xth]; c_hi[r] = c0[r * xth + d]; }
                    double dC = c_hi[0] - c_lo[0];
#pragma unroll
                    for (int r = 1; r < kR; ++r) dC = fmax(dC, c_hi[r] - c_lo[r]);
                    bool live;
                    if (exact_u) {
                        bool und_u = false;
                        live = depth_pass(dC, inv_d, (double)d, dmin, rule.eps, true, und_u);
                    } else {
                        const double m_fast = dC * inv_d;
                        live = m_fast > thr_hi;
                        if (!live && m_fast >= thr_lo && unit < n_units) {
#pragma unroll
                            for (int r = 0; r < kR; ++r) {
                                const double dXr = c_hi[r] - c_lo[r];
                                if (dXr * inv_d >= thr_lo) band_window(rule, k, (uc * kR + r) * xth, dXr, undecided);
                            }
                        }
                    }
                    live = live && unit < n_units;
                    const unsigned long long mask = ballot64(live);
                    if (live) list[n_listed + (unsigned int)__popcll(mask & below)] = (unsigned int)unit;
                    n_listed += (unsigned int)__popcll(mask);
                }
            } else {
                for (int tile = 0; tile * kWave < n_pos; ++tile) {
                    const int unit = tile * kWave + lane;
                    bool live = false;
                    if (unit < n_pos) {
                        const int i = unit * xth;
                        const double dXw = c_base[i + d] - c_base[i];
                        bool und_w = false;
                        live = depth_pass(dXw, inv_d, (double)d, dmin, rule.eps, exact_mode, und_w);
                        if (und_w) band_window(rule, k, i, dXw, undecided);
                    }
                    const unsigned long long mask = ballot64(live);
                    if (live) list[n_listed + (unsigned int)__popcll(mask & below)] = (unsigned int)unit;
                    n_listed += (unsigned int)__popcll(mask);
                }
            }
            TLS_CHECK(*ap, n_listed <= (unsigned int)widths_c[k].n_chunks, kChkListCap);
            rt.live[k - k_lo] = n_listed;   // (every lane stores the same value: see the other kernel on the ticket loop)
        }
        wg_sync();
        pc.mark(9);
        // sparse rows and mostly empty last batches: re-listed one window per lane (as the other kernel)
        for (int row = wave; row < n_rows; row += nw) {
            const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
            int n_live = __builtin_amdgcn_readfirstlane((int)rt.live[row]);
            const int xth = widths_c[k].xth, n_units = widths_c[k].n_chunks, tiled = widths_c[k].tiled;
            const int d = widths_c[k].width, list_base = widths_c[k].list_base;
            const double inv_d = widths_c[k].inv_d, dd = (double)d;
            unsigned int* list = chunk_list + list_base;
            unsigned int count = 0;
            const bool sparse = n_live <= kSparseRow;
            const int n_tail = sparse ? n_live : (TLS_TAIL_RELIST ? (n_live & (kWave - 1)) : 0);
            const int first = n_live - n_tail;
            if (tiled && n_tail > 0 && n_tail <= kSlimTailMax && n_units >= (kR + 1) * kSparseRow && first + n_tail * kR <= n_units) {
                const unsigned int my_unit = lane < n_tail ? list[first + lane] : 0u;
#pragma unroll 1
                for (int base = 0; base < n_tail * kR; base += kWave) {
                    const int idx = base + lane;
                    bool pass = false;
                    const int u = __shfl((int)my_unit, (idx / kR) & (kWave - 1), kWave) * kR + idx % kR;
                    if (idx < n_tail * kR) {
                        const int i = u * xth;
                        const double dXw = c_base[i + d] - c_base[i];
                        // (fast mode: a window inside the band is listed, not noted -- phase 3b meets it again and notes it
                        // there; a tail that goes back to chunk form would otherwise note its band windows twice)
                        bool und_w = false;
                        pass = exact_mode ? depth_pass(dXw, inv_d, dd, dmin, rule.eps, true, und_w) : dXw * inv_d >= thr_lo;
                    }
                    const unsigned long long mask = ballot64(pass);
                    if (pass) list[first + count + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned int)u;
                    count += (unsigned int)__popcll(mask);
                }
                if (sparse || count <= (unsigned int)kSlimTailSingles) {
                    n_live = first;
                } else {
                    if (lane < n_tail) list[first + lane] = my_unit;
                    count = 0;
                }
            }
            TLS_CHECK(*ap, (unsigned int)n_live + count <= (unsigned int)n_units, kChkSinglesCap);
            if (lane == 0) { rt.live[row] = (unsigned int)n_live; rt.singles[row] = count; }
        }
        pc.mark(25);
        wg_sync();
        if (wave == 0) {   // exclusive scan of the batch counts over the rows
            unsigned int carry = 0;
            for (int r0 = 0; r0 < n_rows; r0 += kWave) {
                const int row = r0 + lane;
                unsigned int mine = 0;
                if (row < n_rows) mine = (rt.live[row] + kWave - 1) / kWave + (rt.singles[row] + kWave - 1) / kWave;
                const unsigned int incl = wave_inclusive_sum_u32(mine);
                if (row < n_rows) rt.batch_start[row] = carry + incl - mine;
                carry += (unsigned int)lane_value((int)incl, kWave - 1);
            }
            if (lane == 0) { rt.batch_start[n_rows] = carry; *rt.next_batch = 0; }
            if (ap->phase_cycles && lane == 0) {
                unsigned int kept = 0, singles = 0;
                for (int row = 0; row < n_rows; ++row) { kept += rt.live[row]; singles += rt.singles[row]; }
                atomicAdd(&ap->phase_cycles[33], (unsigned long long)kept);
                atomicAdd(&ap->phase_cycles[34], (unsigned long long)singles);
                atomicAdd(&ap->phase_cycles[35], (unsigned long long)carry);
            }
        }
        wg_sync();
        pc.mark(6);

        // ---- phase 3b: sliding dot products ON X, 64 live units of one duration per wave (core.py:59-74) ----
        {
#if TLS_SLIM_PRIO
            __builtin_amdgcn_s_setprio(0);   // (the FMA-bound phase yields the issue slots: see the period's start)
#endif
            const unsigned int total_batches = (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[n_rows]);
            int row = n_rows > 0 ? n_rows - 1 : 0;
            for (;;) {
                unsigned int gq = 0;
                if (lane == 0) gq = atomicAdd(rt.next_batch, 1u);
                gq = (unsigned int)__builtin_amdgcn_readfirstlane((int)gq);
                if (gq >= total_batches) break;
                const unsigned int gg = total_batches - 1 - gq;
                while (gg < (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[row])) --row;
                const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
                const int d = widths_c[k].width, L = widths_c[k].q_len, xth = widths_c[k].xth;
                const int q_offset = widths_c[k].q_offset, list_base = widths_c[k].list_base;
                const double overshoot = widths_c[k].overshoot, sum_q2 = widths_c[k].sum_q2;
                const double inv_d = widths_c[k].inv_d, dd = (double)d;
                const int tiled = widths_c[k].tiled;
                const int n_singles = __builtin_amdgcn_readfirstlane((int)rt.singles[row]);
                const int n_live = __builtin_amdgcn_readfirstlane((int)rt.live[row]);
                const unsigned int in_row = gg - (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[row]);
                const unsigned int chunk_batches = ((unsigned int)n_live + kWave - 1) / kWave;
                const bool relisted = in_row >= chunk_batches;
                const unsigned int slot = (relisted ? in_row - chunk_batches : in_row) * kWave + lane;
                const bool have = slot < (unsigned int)(relisted ? n_singles : n_live);
                const int unit = have ? (int)chunk_list[list_base + (relisted ? n_live : 0) + slot] : 0;
                const const_f64_ptr g = g_all + q_offset;
                const int Lx = L + 1;   // difference taps: one more than the row
                const unsigned int evals_before = n_eval;
                if (COUNTING && ap->counters) {
                    const int reach = (tiled && !relisted) ? (kR - 1) * xth : 0;
                    n_issued += (unsigned long long)((Lx + reach + kU - 1) / kU * kU) * (reach ? kR : 1);
                }
                if (tiled && !relisted) {
                    const int b = unit * kR * xth;
                    TLS_CHECK(*ap, !have || (b >= 0 && b + (Lx + (kR - 1) * xth + kU - 1) / kU * kU <= M + 1 + region_pad), kChkDotWindow);
                    const double* e = X + b;
                    double Bv[kR], Av[kR];
#pragma unroll
                    for (int r = 0; r < kR; ++r) { Bv[r] = 0.0; Av[r] = sum_q2; }
                    switch (xth) {
                        case 1: dot_windows<true, 1>(e, g, Lx, Bv); break;
                        case 2: dot_windows<true, 2>(e, g, Lx, Bv); break;
                        case 3: dot_windows<true, 3>(e, g, Lx, Bv); break;
                        case 4: dot_windows<true, 4>(e, g, Lx, Bv); break;
                        case 5: dot_windows<true, 5>(e, g, Lx, Bv); break;
                        default: dot_windows_rt<true>(e, g, Lx, xth, Bv); break;
                    }
                    if (have)
                        consider_cells<UNIFORM_W, false, kR, COUNTING>(lead, c_base, b, xth, d, inv_d, dd, rule, overshoot, Av, Bv, k, n_eval, undecided, widths_c, X);
                } else {
                    const int i = unit * xth;
                    TLS_CHECK(*ap, !have || (i >= 0 && i + (Lx + kU - 1) / kU * kU <= M + 1 + region_pad), kChkDotWindow);
                    const double* e = X + i;
                    double B0 = 0.0;
                    for (int t0 = 0; t0 < Lx; t0 += kU) {
                        const const_f64_ptr gs = g + t0;
                        double taps[kU], x[kU];
#pragma unroll
                        for (int u = 0; u < kU; ++u) taps[u] = gs[u];
                        load_taps<true>(e + t0, x);
#pragma unroll
                        for (int u = 0; u < kU; ++u) B0 = fma(taps[u], x[u], B0);   // one accumulator, taps in order: the tiled form's rounding
                    }
                    if (have) consider<UNIFORM_W, false>(lead, c_base[i], c_base[i + d], i, inv_d, dd, rule, overshoot, sum_q2, B0, k, n_eval, undecided, widths_c, X);
                }
                if constexpr (COUNTING) n_steps += (unsigned long long)(n_eval - evals_before) * (unsigned long long)L;
            }
        }
#if TLS_SLIM_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        pc.mark(7);
        } else {
            // The noted windows, one wavefront each: decided by the reference's expression on X = k - numpy.cumsum; a window
            // that passes is evaluated -- lanes over the template taps, flux from global memory through the stashed order,
            // the patch as an index mapping -- and meets lane 0's lead with the window sum of the plain scan it was noted with.
            const int n_band = __builtin_amdgcn_readfirstlane(s_work[4]);
            for (int e0 = wave; e0 < n_band; e0 += nw) {
                const int k = __builtin_amdgcn_readfirstlane(band_list[e0].k);
                const int i = __builtin_amdgcn_readfirstlane(band_list[e0].i);
                const double dX_noted = band_list[e0].dX;
                const int d = widths_c[k].width, L = widths_c[k].q_len, q_offset = widths_c[k].q_offset;
                const double overshoot = widths_c[k].overshoot, sum_q2 = widths_c[k].sum_q2, inv_d = widths_c[k].inv_d;
                const double dd = (double)d;
                const double dX_exact = X[i + d] - X[i];
                if (!((1.0 - (dd - dX_exact) / dd) > dmin)) continue;   // core.py:58 on the reference's bits
                const double* qg = ap->q + q_offset;
                double Bs = 0.0;
                for (int tt = lane; tt < L; tt += kWave) {
                    const int pp = i + tt, src = pp < n ? pp : pp - n;
                    Bs = fma(qg[tt], 1.0 - y_c[perm_g[src]], Bs);
                }
#pragma unroll
                for (int delta = kWave / 2; delta > 0; delta >>= 1) Bs += __shfl_down(Bs, delta, kWave);
                if (lane == 0) {
                    bool und_none = false;
                    // (valued like every other cell of the period: on the plain scan's window sum)
                    consider<UNIFORM_W, true, true>(lead, 0.0, dX_noted, i, inv_d, dd, rule, overshoot, sum_q2, Bs, k, n_eval, und_none, widths_c, X);
                    if constexpr (COUNTING) n_steps += (unsigned long long)L;
                }
            }
            if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[38], (unsigned long long)n_band);
        }
        // fast mode: a window too close to transit_depth_min to call sends the period (this light curve) through exact mode
        if (undecided) s_work[flag_slot] = 1;
        wg_sync();
        const int any_undecided = __builtin_amdgcn_readfirstlane(s_work[flag_slot]);
        if (tid == 0) s_work[3 - flag_slot] = 0;
        flag_slot = 3 - flag_slot;
        // noted band windows (and nothing that voids the attempt): the exact prefix pass and the resolution above; more of
        // them than the list holds: a second search in exact mode
        if (rule.band_count != nullptr && any_undecided == 0) {
            const int n_band = __builtin_amdgcn_readfirstlane(s_work[4]);
            if (n_band > 0) {
                if (n_band <= kBandCap) {
                    resolve_band = true;
                    kept_lead = lead; kept_eval = n_eval; kept_steps = n_steps; kept_issued = n_issued;
                    // (the lead's window sum as the PLAIN scan has it, taken now: the resolution pass overwrites X with the exact
                    // prefix sum, and every cell of a period is valued on the same X -- ADVICE r05: read again there, the
                    // incumbent and the winner of a resolved period were valued on the other prefix sum, 1e-13 off)
                    kept_lead.dX = lead.stat < INFINITY ? X[lead.i + widths_c[lead.k].width] - X[lead.i] : 0.0;
                }
                if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[37], 1ull);
                if (ap->n_curves > 1) { curve_exact = true; --curve; continue; }
                retry_exact = true;
                break;
            }
        }
        if (any_undecided != 0 && !exact_mode) {
            if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[37], 1ull);
            if (ap->n_curves > 1) { curve_exact = true; --curve; continue; }
            retry_exact = true;
            break;
        }
        pc.mark(21);

        // ---- phase 4: argmin over the workgroup (core.py:70-74, 183-188) ------------------------------
        // (a resolved period: X now holds the exact prefix sum; its leads carry the plain scan's window sums)
        Best best = resolving ? settle_best<UNIFORM_W, true>(lead, widths_c, X) : settle_best<UNIFORM_W, false>(lead, widths_c, X);
#pragma unroll
        for (int delta = kWave / 2; delta > 0; delta >>= 1) {
            Best o = shfl_down_best(best, delta);
            if (better(o, best)) best = o;
        }
        if (lane == 0) wbest[wave] = best;
        wg_sync();
        if (tid == 0) {
            Best gb = wbest[0];
            for (int u = 1; u < nw; ++u) if (better(wbest[u], gb)) gb = wbest[u];
            const double datapoints = (double)n;
            double chi2 = INFINITY, depth = 0.0;
            long long row = 0;
            if (n_rows > 0) {
                const double w0_c = ap->n_curves > 1 ? ap->curve_w0[curve] : ap->w0;
                const double S0_c = ap->n_curves > 1 ? ap->curve_S0[curve] : ap->S0;
                const double stat = (gb.stat < INFINITY) ? S0_c + w0_c * gb.stat : INFINITY;
                if (stat < datapoints) {
                    chi2 = stat; row = ap->widths[gb.k].row; depth = 1.0 - gb.td;
                } else {
                    chi2 = datapoints; row = ap->widths[k_lo].row; depth = 0.0;
                }
            }
            const long long o = (long long)curve * ap->n_periods + p;
            ap->out_chi2[o] = chi2;
            ap->out_row[o] = row;
            ap->out_depth[o] = depth;
        }
        if (COUNTING && ap->counters) {
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                n_eval += __shfl_down(n_eval, delta, kWave);
                n_steps += __shfl_down(n_steps, delta, kWave);
            }
            if (lane == 0 && n_eval) {
                atomicAdd(&ap->counters[0], (unsigned long long)n_eval);
                atomicAdd(&ap->counters[1], n_steps);
            }
            if (lane == 0 && n_issued) atomicAdd(&ap->counters[2], n_issued * kWave);
        }
        wg_sync();
        }  // light curves of the batch
        if (ap->period_cycles && tid == 0) atomicAdd(&ap->period_cycles[p], (unsigned long long)(clock64() - t_period));
    }
}
