// tls_search_body.inc.h -- the body of the search kernels, included textually INSIDE the __global__ functions of
// tls_kernels.hip.h (never on its own).  The including scope defines RESIDENT, UNIFORM_W, IdxT, WITH_PRUNING,
// COUNTING and ROLE (kRoleAll / kRoleFold / kRoleSearch).  Textual inclusion, not a function: as an always-inlined
// device function the same code lost the kernel's work-group-size facts (thread-index ranges) at the point where clang
// emits them, and the slab instantiation -- which sits at the 128-register cliff -- came out with five times the
// spill code and 14 % slower (measured; the "register lottery" of PERF_LOG.md).
    static_assert(ROLE == kRoleAll || !RESIDENT, "the LDS-resident series is searched by one workgroup per period");
    // The arguments are read through a pointer to the kernel-argument segment, where they are used (scalar loads the
    // compiler may repeat), not taken by value: ~100 values loaded at entry compete for 104 scalar registers for the
    // whole kernel, and the losers live in spilled lanes of a vector register and come back one v_readlane -- a VALU
    // instruction -- at a time (measured: config 2 -1.7 %; forcing a re-read per phase or per period: slower again).
    args_ptr ap = (args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & (kWave - 1), nw = nt / kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave-uniform by construction
    const int n = ap->n, W = ap->W, M = ap->M, nb = ap->nb;
    const int region_pad = ap->region_pad;
    // region stride in doubles; even in the slab so that every region starts on a 16-byte boundary
    const int RS = RESIDENT ? M + 1 + region_pad : ((M + 1 + region_pad + 1) & ~1);

    // ---- memory carve-up -----------------------------------------------------------
    unsigned int* wsum = reinterpret_cast<unsigned int*>(smem);            // 32 words
    Best* wbest = reinterpret_cast<Best*>(smem + 128);                      // kMaxWaves * 24 B
    int* s_work = reinterpret_cast<int*>(smem + 128 + kMaxWaves * sizeof(Best));  // [4]
    CumsumScratch* cumsum_scratch = reinterpret_cast<CumsumScratch*>(smem + 560);
    // the pruning variant is a separate instantiation: its extra state costs the plain variant 3-8 %
    // when both live in one kernel, and the host knows from the noise level which one pays
    // (LDS-resident series only: on a series in the HBM slab the bound's look-ups in X are global loads -- forced, it never
    // paid, and round 6 dropped the slab instantiations)
    constexpr bool PRUNE = RESIDENT && UNIFORM_W && WITH_PRUNING && (TLS_PRUNE != 0);
    // the fp32 screen of the dot products (screen_cells): its own instantiation, chosen by the host
    constexpr bool SCR = SCREEN && RESIDENT && UNIFORM_W && !WITH_PRUNING && !COUNTING && ROLE == kRoleAll;
    // coarse prefix sum of e^2 for the pruning bound; the cumsum scratch is dead by the time it is built
    double* P2 = reinterpret_cast<double*>(cumsum_scratch);
    static_assert(kCumsumScratchBytes >= 8 * (kP2MaxBlocks + 1), "coarse prefix sum does not fit the cumsum scratch");
    // counting the evaluated cells (tls_execute(ctx, 1)) means evaluating all of them
    const bool prune_on = PRUNE && ap->counters == nullptr;
    static_assert(sizeof(CumsumScratch) <= kCumsumScratchBytes, "cumsum scratch does not fit its slot");
    RowTables rt;
    rt.live = reinterpret_cast<unsigned int*>(smem + kFixedHeader);
    rt.singles = rt.live + ap->n_widths;
    rt.batch_start = rt.singles + ap->n_widths;
    rt.next_batch = rt.batch_start + (ap->n_widths + 1);
    double *regA, *regB, *regW = nullptr;
    unsigned int* cnt;
    if constexpr (RESIDENT) {
        regA = reinterpret_cast<double*>(smem + ap->hdr_bytes);
        regB = regA + RS;
        if constexpr (!UNIFORM_W) regW = regB + RS;
        cnt = reinterpret_cast<unsigned int*>(regB);
    } else {
        double* slab = ap->scratch + (long long)blockIdx.x * ap->scratch_stride;   // (split roles: the work item's slab, below)
        regA = slab;
        regB = regA + RS;
        if constexpr (!UNIFORM_W) regW = regB + RS;
        cnt = reinterpret_cast<unsigned int*>(smem + ap->hdr_bytes);
    }
    unsigned int* chunk_list = ap->chunk_lists + (long long)blockIdx.x * ap->list_stride;
    // sort scratch inside regB: [cnt (resident only)] idx_tmp[n] perm[n]
    IdxT* idx_tmp = RESIDENT ? reinterpret_cast<IdxT*>(cnt + nb) : reinterpret_cast<IdxT*>(regB);
    IdxT* perm = idx_tmp + n;
    double* ph_orig = regA;  // phase by ORIGINAL index during the sort

    if (tid == 0) {
        [[maybe_unused]] const long long need = RESIDENT ? (long long)ap->hdr_bytes + (UNIFORM_W ? 2 : 3) * 8LL * RS
                                        : (long long)ap->hdr_bytes + (UNIFORM_W ? 1 : 2) * 8LL * (ap->tile_len + ap->tile_halo);
        TLS_CHECK(*ap, need <= ap->lds_bytes, kChkLdsCarve);
        TLS_CHECK(*ap, (long long)kFixedHeader + 4LL * (3 * ap->n_widths + 2) <= ap->hdr_bytes, kChkLdsCarve);
    }
    // the spare entries behind each region are only ever multiplied by zero: make them finite
    if constexpr (ROLE == kRoleAll) {
        for (int k = tid; k < region_pad; k += nt) {
            regA[M + 1 + k] = 0.0;
            if constexpr (!UNIFORM_W) regW[M + 1 + k] = 0.0;
        }
    }

    const const_width_ptr widths_c = (const_width_ptr)ap->widths;  // read-only for the whole launch
    const const_rows_ptr rows_c = (const_rows_ptr)ap->rows;
    const const_f64_ptr q_all = (const_f64_ptr)ap->q;
    const const_f64_ptr q2_all = (const_f64_ptr)ap->q2;
    const const_screen_ptr screens_c = (const_screen_ptr)ap->screens;
    const double dmin = ap->depth_min;

    bool retry_exact = false;   // the period just searched in fast mode left a window undecided: again, in exact mode
    // Series in the HBM slab, one light curve, plain variant: windows inside the undecided band are noted (band_window) and
    // decided after the attempt on the period's exact prefix sum -- the period loop is entered a second time for the prefix
    // pass only (`resolve_band`), the lanes' leads and counts of the attempt are kept
    // (the search role of the two-role kernel likewise: its work item re-enters the loop, forms the period's exact prefix sum
    // itself -- the fold role left none in fast mode -- and decides the windows ITS tile noted)
    constexpr bool BAND = !RESIDENT && ROLE != kRoleFold && !WITH_PRUNING;
    [[maybe_unused]] bool resolve_band = false;
    [[maybe_unused]] Lead kept_lead = no_lead();
    [[maybe_unused]] unsigned int kept_eval = 0;
    [[maybe_unused]] unsigned long long kept_steps = 0, kept_issued = 0;
    [[maybe_unused]] BandEntry* const band_list = reinterpret_cast<BandEntry*>(chunk_list + ap->list_cap);   // (the pruning variant's second list: idle here)
    int work = 0, work_raw = 0;
    for (;;) {
        // ---- fetch the next period from the queue ----------------------------------
        if (!retry_exact) {
            if (tid == 0) { s_work[0] = (int)atomicAdd(ap->queue + (ROLE == kRoleSearch ? 2 : 0), 1u); s_work[1] = 0; s_work[2] = 0; s_work[4] = 0; }
            wg_sync();
            work_raw = __builtin_amdgcn_readfirstlane(s_work[0]);
            wg_sync();
        } else if (tid == 0) {
            s_work[1] = 0; s_work[2] = 0;   // (published by the barriers of the sort, long before any thread may raise them again)
        }
        work = work_raw;   // (the roles turn the ticket into a period below: a second attempt starts from the ticket again)
        [[maybe_unused]] const bool again = retry_exact;   // this entry is a second attempt / a band resolution of the same work item
        int flag_slot = 1;   // the "undecided" flag of the attempt in flight: s_work[1] and s_work[2] take turns
        // exact mode: X = k - numpy.cumsum, bit for bit; fast mode: X = plain prefix sum of 1 - f (depth_pass)
        // (the two-kernel slab path has no second attempt: its fold kernel always leaves X = k - numpy.cumsum)
        // (the mode of a period depends on the series and the period alone, never on its place in the launch: sending the
        // launch's last round straight to exact mode would spare TESS-size grids a late second attempt -- measured 2.90
        // -> 2.81 ms -- but a period's bits would then depend on which other periods the call holds)
        // (decided below, once the roles of the two-role kernel have turned their ticket into a period)
        // (slab variant, one light curve: the folded flux of the fast attempt is still in the slab -- only X was written
        // behind it --, so the second attempt keeps it and starts at the prefix sum)
        [[maybe_unused]] const bool refold = !(retry_exact && !RESIDENT && ROLE == kRoleAll && ap->n_curves == 1 && ap->debug_folded == nullptr);
        bool curve_exact = false;   // batches: this light curve again in exact mode (the permutation is kept: no new sort)
        // split roles: `work` counts the items of this launch -- the periods of the batch (fold), their tiles (search)
        int n_work = ap->n_periods;
        [[maybe_unused]] int item = 0, item_tile = 0;
        if constexpr (ROLE == kRoleFold) n_work = ap->batch_n;
        if constexpr (ROLE == kRoleSearch) n_work = (int)(ap->tile_prefix[ap->batch_lo + ap->batch_n] - ap->tile_prefix[ap->batch_lo]);
        if (work >= n_work) {
            // the last workgroup to leave rewinds the queue for the next launch (no memset between
            // two searches of a prepared plan); queue[1] counts the workgroups that are done
            if (tid == 0) {
                __threadfence();
                unsigned int* const qq = ap->queue + (ROLE == kRoleSearch ? 2 : 0);   // (the search role has a queue of its own)
                if (atomicAdd(qq + 1, 1u) == gridDim.x - 1) { atomicExch(qq, 0u); atomicExch(qq + 1, 0u); }
            }
            break;
        }
        if constexpr (ROLE == kRoleFold) {
            regA = ap->scratch + (long long)work * ap->scratch_stride;   // the slab of this period of the batch
            regB = regA + RS;
            if constexpr (!UNIFORM_W) regW = regB + RS;
            idx_tmp = reinterpret_cast<IdxT*>(regB); perm = idx_tmp + n; ph_orig = regA;
            work += ap->batch_lo;
        }
        if constexpr (ROLE == kRoleSearch) {
            // item -> (period of the batch, tile): the last w with tile_prefix[w] <= G (scalar loads, ~log2(batch) steps)
            item = work;
            const unsigned int G = ap->tile_prefix[ap->batch_lo] + (unsigned int)work;
            int lo_w = ap->batch_lo, hi_w = ap->batch_lo + ap->batch_n;   // tile_prefix[lo_w] <= G < tile_prefix[hi_w]
            while (hi_w - lo_w > 1) {
                const int mid = (lo_w + hi_w) >> 1;
                if (ap->tile_prefix[mid] <= G) lo_w = mid; else hi_w = mid;
            }
            work = __builtin_amdgcn_readfirstlane(lo_w);
            item_tile = (int)(G - ap->tile_prefix[work]);
            regA = ap->scratch + (long long)(work - ap->batch_lo) * ap->scratch_stride;
            regB = regA + RS;
            if constexpr (!UNIFORM_W) regW = regB + RS;
            // the period's fold may still be running on another workgroup (it has been TAKEN: the fold queue was empty
            // when this workgroup left the fold role)
            // (one relaxed poll, one agent-scope acquire -- it invalidates this CU's vector L1 for all of its waves --, barrier)
            if (tid == 0) {
                const unsigned int* ready = ap->fold_ready + (work - ap->batch_lo);
                while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(16);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            wg_sync();
        }
        const int p = ap->order[work];
        TLS_CHECK(*ap, p >= 0 && p < ap->n_periods, kChkWorkItem);
        const double period = ap->periods[p];
        // exact mode: X = k - numpy.cumsum, bit for bit; fast mode: X = plain prefix sum of 1 - f (depth_pass).
        // The mode of a period depends on the series and the period alone, never on its place in the launch, the launch's
        // shape or the kernel (one workgroup per period / two roles): sending a launch's last round straight to exact mode
        // would spare TESS-size grids a late second attempt -- measured 2.90 -> 2.81 ms -- but a period's bits would then
        // depend on which other periods the call holds.  (The two roles run fast mode where the plan supports it --
        // SearchArgs::split_fast: uniform weights, X formed at tile-staging time, dot products on X --, exact mode otherwise.)
        bool period_exact = (!RESIDENT && (ap->fast_slab == 0 || (ROLE != kRoleAll && ap->split_fast == 0))) || retry_exact ||
                            ap->exact_prefix != 0 || ap->debug_prefix != nullptr;
        if constexpr (!RESIDENT) {
            // a period whose windows are EXPECTED to hit the undecided band (long periods of a long series: many wide
            // windows, little noise on their means) starts in exact mode: the fast attempt would mostly be wasted.  The
            // expectation depends on the period's duration window and the light curve alone (host: band_prefix).
            if (!period_exact && ap->band_prefix != nullptr)
                period_exact = ap->band_prefix[rows_c[p].k_hi] - ap->band_prefix[rows_c[p].k_lo] > ap->band_max;
        }
        retry_exact = false;
#ifndef TLS_BODY_PRIO
#define TLS_BODY_PRIO 1
#endif
#if TLS_BODY_PRIO
        // (wave priority as in tls_slim_kernel: the round-trip-bound phases ahead of the partner workgroup's dot products)
        if constexpr (RESIDENT) __builtin_amdgcn_s_setprio(1);
#endif
        long long t_period = 0;
        if (ap->period_cycles && tid == 0) t_period = clock64();
        PhaseClock pc;
        pc.start(ap->phase_cycles);

        // ---- phase 1: fold + stable sort by phase ----------------------------------
        bool sorted = false;
        if constexpr (!RESIDENT && ROLE == kRoleAll) {
            if (!refold) sorted = true;
        }
        if constexpr (!RESIDENT && ROLE != kRoleSearch) {
            // series in HBM: the two-level sort with sequential HBM accesses, unless a phase bin overflows
            if (!sorted && ap->sort2) {
                typedef global_ptr<const double> gcd;
                typedef global_ptr<double> gd;
                typedef global_ptr<unsigned int> gu;
                typedef global_ptr<unsigned long long> gull;
                // (one light curve: the flux is gathered on the way and no permutation is written)
                // (one light curve: the flux is gathered on the way and no permutation is written)
                sorted = fold_and_sort_tiled_call<!UNIFORM_W>((gcd)ap->t, n, period, (gull) reinterpret_cast<unsigned long long*>(regA),
                                                              (gu)(ap->n_curves == 1 ? nullptr : reinterpret_cast<unsigned int*>(perm)),
                                                              lds_address(smem + ap->hdr_bytes), (gull)ap->phase_cycles,
                                                              (gcd)(ap->n_curves == 1 ? ap->y : nullptr), (gcd)(UNIFORM_W ? nullptr : ap->w),
                                                              (gd)regA, (gd)regW, (gull)ap->check);
                pc.start(ap->phase_cycles);   // (the call kept its own clock)
            }
        }
        if (ROLE != kRoleSearch && !sorted) {
            // (piled-up buckets are sorted by the workgroup: their list lives in the idle prefix-sum scratch; the slab
            // variant stages them in the LDS behind its bucket counters, the resident one sorts through the index)
            unsigned int* big_list = reinterpret_cast<unsigned int*>(cumsum_scratch);
            constexpr int kBigCap = kCumsumScratchBytes / 4;
            if constexpr (RESIDENT) {
                fold_and_sort<IdxT>(ap->t, n, period, 0.0, ph_orig, cnt, nb, idx_tmp, perm, wsum, pc, big_list, kBigCap);
            } else {
                double* stage_key = reinterpret_cast<double*>(cnt + ((nb + 1) & ~1));
                const long long room = ap->lds_bytes - ap->hdr_bytes - 4LL * ((nb + 1) & ~1);
                const int stage_cap = room > 0 ? (int)(room / 12) : 0;
                unsigned int* stage_idx = reinterpret_cast<unsigned int*>(stage_key + stage_cap);
                fold_and_sort<IdxT>(ap->t, n, period, 0.0, ph_orig, cnt, nb, idx_tmp, perm, wsum, pc, big_list, kBigCap,
                                    stage_key, stage_idx, stage_cap);
            }
        }
        // survey mode: the permutation depends on (t, period) only, so every light curve of the
        // batch reuses it; it must outlive the prefix sum that overwrites its LDS home
        const IdxT* perm_use = perm;
        if (ROLE == kRoleAll && ap->n_curves > 1) {
            IdxT* perm_g = reinterpret_cast<IdxT*>(ap->perm_scratch + (long long)blockIdx.x * n);
            for (int k = tid; k < n; k += nt) perm_g[k] = perm[k];
            perm_use = perm_g;
            wg_sync();
        }
        for (int curve = 0; curve < ap->n_curves; ++curve) {
        const bool exact_mode = period_exact || curve_exact;
        curve_exact = false;
        DepthRule rule;
        rule.dmin = ap->depth_min; rule.eps = exact_mode ? 1e-15 : ap->eps_fast; rule.exact_mode = exact_mode;
        rule.band_count = nullptr; rule.band_list = nullptr;
        if constexpr (BAND) {
            // (16 bytes an entry in the idle list region: list_cap words hold list_cap / 4 of them)
            if (!exact_mode && ap->list_cap >= 4 * kBandCap) {
                rule.band_count = reinterpret_cast<unsigned int*>(&s_work[4]); rule.band_list = band_list;
            }
        }
        // (the estimate's mean depth is off by ~1e-16 absolute: negligible against transit_depth_min = 1e-5, the whole
        // story for a transit_depth_min near zero -- then every cell takes the exact comparison)
        rule.reach = (rule.dmin - rule.eps > 4e-15) ? fmin(fmax(1e-9, 4e-15 / (rule.dmin - rule.eps)), 1.0) : 1.0;
        bool undecided = false;
        // series in the HBM slab, fast mode: X of a tile is formed from the tile's staged flux (scan_tile_x) instead of by a
        // prefix-sum pass over the whole series -- one pass of slab reads and the pass's barriers less per period
        // (every tile's scan starts at zero: only differences of X are ever read, and a tile's X is then the same function of
        // the tile's samples in the one-workgroup kernel and in a (period, tile) item of the two-role kernel, whatever the flux)
        [[maybe_unused]] const bool x_staging = !RESIDENT && TLS_SLAB_DMA && !exact_mode && ap->x_at_staging != 0 && (n & 1) == 0;
        [[maybe_unused]] int x_got = 0;          // X entries of the tile in flight (scan_tile_x)
        const double* y_c = ap->y + (long long)curve * n;
        if constexpr (ROLE != kRoleSearch) {
        // gather flux (and weights) in folded order; ph_orig (regA) is dead from here on.  kG
        // elements per step: their global reads (L2 latency) are in flight together -- the compiler
        // cannot overlap them itself, the LDS store of one may alias the index read of the next
        const bool gathered = !RESIDENT && sorted && ap->n_curves == 1;   // the sort did it on the way
        constexpr int kG = TLS_GATHER_DEPTH;
        for (int k0 = tid; k0 < (gathered ? 0 : n); k0 += kG * nt) {
            int idx[kG];
            double v[kG];
#pragma unroll
            for (int g = 0; g < kG; ++g) idx[g] = (int)perm_use[k0 + g * nt < n ? k0 + g * nt : k0];
#pragma unroll
            for (int g = 0; g < kG; ++g) v[g] = y_c[idx[g]];
            if constexpr (!UNIFORM_W) {
                const double* w_c = ap->w + (long long)curve * n;
                double u[kG];
#pragma unroll
                for (int g = 0; g < kG; ++g) u[g] = w_c[idx[g]];
#pragma unroll
                for (int g = 0; g < kG; ++g) if (k0 + g * nt < n) regW[k0 + g * nt] = u[g];
            }
#pragma unroll
            for (int g = 0; g < kG; ++g) if (k0 + g * nt < n) regA[k0 + g * nt] = v[g];
        }
        wg_sync();
        if (ap->debug_folded && curve == 0) {   // test entry: the folded flux as the sort left it (core.py:120-123)
            for (int k = tid; k < n; k += nt) ap->debug_folded[(long long)p * n + k] = regA[k];
            wg_sync();
        }
        // ---- phase 2: patch (core.py:126-132) and sequential cumsum ----------------
        if (RESIDENT) {   // (the slab keeps the folded series once; its patch is an index mapping)
            for (int k = tid; k < W; k += nt) {
                regA[n + k] = regA[k];
                if constexpr (!UNIFORM_W) regW[n + k] = regW[k];
            }
            if (tid == 0) {
                regA[M] = 0.0;
                // (entry M of the weights too: the unrolled dot products of the last windows read it against a zero tap, and
                // whatever a previous tenant left in the LDS may be a NaN -- 0 * NaN made A of those cells NaN and lost them
                // the comparison: one period of golden `weights` 0.26 % off on the first launch of a fresh box, round 5;
                // tests: tls_debug_poison_lds)
                if constexpr (!UNIFORM_W) regW[M] = 0.0;
            }
        }
        wg_sync();
        pc.mark(4);

        }   // (search role: the fold kernel has done it)
        // in-range widths of this period: a contiguous range [k_lo, k_hi) of the ascending
        // width table (core.py:148-156); widths below k_x have the dense T0 grid (stride 1).
        // Wave-uniform by construction; say so, or the template taps stop being scalar loads.
        const int k_lo = __builtin_amdgcn_readfirstlane(rows_c[p].k_lo);
        const int k_hi = __builtin_amdgcn_readfirstlane(rows_c[p].k_hi);
        const int k_x = __builtin_amdgcn_readfirstlane(rows_c[p].k_x);
        const int n_rows = k_hi - k_lo;
        TLS_CHECK(*ap, 0 <= k_lo && k_lo <= k_x && k_x <= k_hi && k_hi <= ap->n_widths, kChkWorkItem);
        for (int row = tid; row < n_rows; row += nt) rt.live[row] = 0;  // published by the cumsum's barriers
        if (tid == 0) s_work[3] = 0;   // ticket counter of the strided rows (phase 3a), published the same way
        if constexpr (BAND) { if (tid == 0 && !resolve_band) s_work[4] = 0; }   // the noted band windows of this light curve's attempt
        // (search role: the fold role has done it -- except for the second entry of a work item whose fast attempt noted band
        // windows or overflowed: the fold left no X for a fast period, the item forms the period's exact prefix sum itself.
        // Two items of one period may do so at the same time: they store the same values.)
        if constexpr (ROLE == kRoleSearch) {
            if (again) {   // (out of line, with registers of its own: rare, and the search role's allocation stays what phase 3 needs)
                typedef global_ptr<const double> gcd;
                typedef global_ptr<double> gd;
                slab_exact_prefix_call((gcd)regA, (gd)regB, n, M, region_pad, ap->cumsum_round,
                                       lds_address(reinterpret_cast<double*>(smem + ap->hdr_bytes) + 1), lds_address(cumsum_scratch),
                                       (global_ptr<unsigned long long>)ap->phase_cycles);
            }
        }
        if constexpr (ROLE != kRoleSearch) {
        // numpy.cumsum order (helpers.py:72), bit for bit, evaluated by the whole workgroup -- or, in fast mode,
        // e = 1 - f and its plain prefix sum X in one pass (depth_pass explains why that decides the same cells)
        if constexpr (RESIDENT) {
            if (!exact_mode) {
                prefix_sum_of_e<UNIFORM_W>(regA, regW, regB, M, reinterpret_cast<double*>(cumsum_scratch));
            } else {
#if TLS_CUMSUM2
            exact_cumsum<false, true, true>(regA, regB, M, reinterpret_cast<Cumsum2Scratch*>(cumsum_scratch), ap->phase_cycles);
#else
            exact_sequential_cumsum(regA, regB, M, cumsum_scratch, ap->phase_cycles);
#endif
            }
        } else if (x_staging) {
            // (fast mode with X formed at tile-staging time: no prefix-sum pass, the slab's X region is written tile by tile)
        } else {
            // the series is in the HBM slab: the scan runs through LDS, 16 K elements a round, in place
            // (C[k+1] over f[k]); the patch (core.py:126: the first W samples again) is an index mapping
            // of the copy-in; LDS-only barriers, so the prefix-sum stores of a round stay in flight
            double* buf = reinterpret_cast<double*>(smem + ap->hdr_bytes) + 1;   // C[0..len], f = buf + 1 (16-byte aligned)
            const int kRound = ap->cumsum_round;   // elements per LDS round (the host sizes it to the workgroup's LDS share)
            double carry = 0.0;
            const bool dma = TLS_SLAB_DMA && (n & 1) == 0;   // pairs of samples never straddle the patch boundary
            for (int c0 = 0; c0 < M; c0 += kRound) {
                const int len = M - c0 < kRound ? M - c0 : kRound;
                constexpr int kInFlight = 8;
                if (dma) { slab_to_lds_async(buf + 1, regA, c0, len, n, tid); vmem_wait_all(); }
                for (int k0 = tid; k0 < (dma ? 0 : len); k0 += kInFlight * nt) {
                    double v[kInFlight];
#pragma unroll
                    for (int j = 0; j < kInFlight; ++j) {
                        const int k = k0 + j * nt, pp = c0 + k;
                        v[j] = k < len ? stream_load(regA + (pp < n ? pp : pp - n)) : 0.0;
                    }
#pragma unroll
                    for (int j = 0; j < kInFlight; ++j) if (k0 + j * nt < len) buf[1 + k0 + j * nt] = v[j];
                }
                lds_barrier();
                pc.mark(26);
                if (!exact_mode) {
                    // fast mode (depth_pass): X[k+1] = X[k] + (1 - f[k]) as a plain scan, in place over the staged f
                    double* wtot = reinterpret_cast<double*>(cumsum_scratch);
                    int per = (len + nt - 1) / nt;
                    if ((per & 1) == 0) per += 1;
                    const int lo = tid * per < len ? tid * per : len;
                    const int hi = lo + per < len ? lo + per : len;
                    double local = 0.0;
                    for (int k = lo; k < hi; ++k) local += 1.0 - buf[1 + k];
                    const double incl = wave_inclusive_sum(local);
                    if (lane == kWave - 1) wtot[wave] = incl;
                    lds_barrier();
                    double run = carry;
                    for (int v = 0; v < wave; ++v) run += wtot[v];
                    run += incl - local;
                    for (int k = lo; k < hi; ++k) { const double e1 = 1.0 - buf[1 + k]; run += e1; buf[1 + k] = run; }
                    if (tid == 0) buf[0] = carry;
                    double total = carry;
                    for (int v = 0; v < nw; ++v) total += wtot[v];
                    carry = total;                                    // the same additions in every thread
                    lds_barrier();
                    pc.mark(5);
                    copy_out_stream(regB + c0, buf, len + 1, tid);    // (buf holds X itself)
                } else {
                if (len <= 16 * nt) {
                    carry = exact_cumsum_round_call(lds_address(buf), len, carry, lds_address(cumsum_scratch),
                                                    (global_ptr<unsigned long long>)ap->phase_cycles);
                } else {   // fewer than 1024 threads: several blocks per round
                    carry = exact_cumsum<true>(buf + 1, buf, len, reinterpret_cast<Cumsum2Scratch*>(cumsum_scratch), ap->phase_cycles, carry);
                }
                pc.mark(5);
                copy_out_stream_x(regB + c0, buf, len + 1, tid, c0);   // the slab keeps X[k] = k - C[k]
                }
                lds_barrier();
                pc.mark(27);
            }
        }
        // sentinels behind X: a window that would start past the end of the T0 grid sees an
        // absurdly negative "depth" and fails the depth predicate without any bounds test.  The
        // sentinels FALL (-k * 1e300 at index M + k) so that a window whose both ends lie in the
        // sentinels (possible for widths below kR) still sees a huge negative sum.
        for (int k = tid; k < region_pad; k += nt) regB[M + 1 + k] = -(double)(k + 1) * 1.0e300;
        wg_sync();
        if (ap->debug_prefix && curve == 0) {   // test entry: C as numpy.cumsum gives it (helpers.py:72); exact mode is forced
            if constexpr (RESIDENT) { for (int k = tid; k <= M; k += nt) ap->debug_prefix[(long long)p * (M + 1) + k] = regB[k]; }
            else { for (int k = tid; k <= M; k += nt) ap->debug_prefix[(long long)p * (M + 1) + k] = (double)k - regB[k]; }
            wg_sync();
        }
        pc.mark(5);
        }   // (search role: X is in the period's slab)
        if constexpr (ROLE == kRoleFold) {
            // the slab is complete: every thread's stores are made visible to the other XCDs (agent-scope release), then
            // the period's flag lets its tiles go.  (one light curve: `continue` leaves the curve loop)
            // (MI355X_MICROARCH.md, inter-workgroup visibility: the barrier has drained every wave's stores into the XCD's
            // L2; ONE lane writes the L2's dirty lines back, waits, and only then raises the flag.  Measured alternatives:
            // every thread fencing -- +30 %; write-through (`sc1`) stores of the slab's final content, 8 bytes a lane, so
            // that the write-back finds nothing -- Kepler-size sample 6.33 instead of 5.43 ms, the stores themselves slow down)
            wg_sync();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(ap->fold_ready + (work - ap->batch_lo), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            continue;
        }
        // exact mode, resident: regB holds C -- now X[k] = k - C[k] (an exact subtraction); e = 1 - f in place (uniform
        // weights) or e*w (general weights).  Fast mode has done both; the tiled variant chunk by chunk above.
        if constexpr (RESIDENT) {
            if (exact_mode) {
                for (int k = tid; k <= M; k += nt) regB[k] = (double)k - regB[k];
                for (int k = tid; k < M; k += nt) {
                    double e = 1.0 - regA[k];
                    if constexpr (!UNIFORM_W) e *= regW[k];
                    regA[k] = e;
                }
            }
        }
        wg_sync();
        pc.mark(8);
        // fp32 screen: the samples leave their fp64 form -- high halves twice in LDS (hi[k] and, one sample later, hi1[k] =
        // hi[k + 1]: a lane reads PAIRS of samples with 8-byte-aligned ds_read_b64 whatever the parity of its first
        // sample), low halves in the workgroup's global scratch (read only when a cell is valued in fp64)
        [[maybe_unused]] unsigned eh_addr = 0, eh1_addr = 0;
        [[maybe_unused]] glob_f32_ptr lo_g = nullptr;
        if constexpr (SCR) {
            // in place, in ascending chunks: the floats of a chunk land below every double that is still to be read
            constexpr int kSplitPer = 6;
            float* const hi0 = reinterpret_cast<float*>(regA);
            float* const hi1 = hi0 + ((RS + 1) & ~1);   // 8-byte aligned like hi0; its RS - 1 entries end inside the region
            float* const lo_w = ap->split_lo + (long long)blockIdx.x * RS;
#pragma unroll 1
            for (int c0 = 0; c0 < RS; c0 += kSplitPer * nt) {
                double v[kSplitPer];
#pragma unroll
                // (the spare entries behind the samples hold the previous period's floats by now: zero again)
                for (int j = 0; j < kSplitPer; ++j) { const int k = c0 + tid + j * nt; v[j] = k < M ? regA[k] : 0.0; }
                lds_barrier();   // (LDS only: the low halves in global memory are published by the barrier behind phase 3a)
#pragma unroll
                for (int j = 0; j < kSplitPer; ++j) {
                    const int k = c0 + tid + j * nt;
                    if (k < RS) {
                        const float h = (float)v[j];
                        const float l = (float)(v[j] - (double)h);
                        TLS_CHECK(*ap, (double)h + (double)l == v[j], kChkSplit);
                        hi0[k] = h;
                        lo_w[k] = l;
                    }
                }
                // (the next chunk's doubles lie above this chunk's floats: no barrier between the stores and its loads)
            }
            lds_barrier();
            for (int k = tid; k + 1 < RS; k += nt) hi1[k] = hi0[k + 1];
            eh_addr = lds_address(hi0);
            eh1_addr = lds_address(hi1);
            lo_g = (glob_f32_ptr)lo_w;
            if (tid == 0) reinterpret_cast<ParkList*>(cumsum_scratch)->n = 0u;   // (the prefix-sum scratch is idle in phase 3)
            lds_barrier();
            pc.mark(18);
        }
        [[maybe_unused]] ScreenEnv scr_env;
        if constexpr (SCR) {
            scr_env.eh_addr = eh_addr; scr_env.lo_g = lo_g; scr_env.q_g = (glob_f64_ptr)ap->q;
            scr_env.park = reinterpret_cast<ParkList*>(cumsum_scratch);
            scr_env.cells = reinterpret_cast<ParkedCell*>(ap->park_cells) + (long long)blockIdx.x * kParkCap;
            scr_env.stat = ap->phase_cycles ? ap->phase_cycles + 38 : nullptr;
        }

        Lead lead = no_lead();
        [[maybe_unused]] ScreenSlot scr_slot = empty_slot();
        unsigned int n_eval = 0;          // cells this lane evaluated in this period (32 bits: one add per window)
        unsigned long long n_steps = 0;
        unsigned long long n_issued = 0;   // FMAs per lane of this wave's dot products (wave-uniform)
        bool p2_ready = false;
        // ---- phase 3 runs over TILES of window-start positions [p_lo, p_hi).  Resident variant:
        // one tile, the folded series already sits in LDS.  Otherwise the series is in the HBM slab
        // and each tile (+ halo = widest window) is staged into LDS first; windows are owned by the
        // tile that contains their first sample.
        // (the slab variant's tile length is the period's own: its halo covers the widest in-range window only)
        const int tile_len_p = RESIDENT ? 0 : __builtin_amdgcn_readfirstlane(rows_c[p].pad);
        const int tile_len = RESIDENT ? (1 << 30) : (tile_len_p > 0 ? tile_len_p : ap->tile_len);
        // (search role: the one tile of this work item)
        const int p_first = ROLE == kRoleSearch ? item_tile * tile_len : 0;
        // (second pass over a period whose noted band windows wait for the exact prefix sum: no tile is searched again)
        [[maybe_unused]] bool resolving = false;
        if constexpr (BAND) {
            if (resolve_band) {
                resolving = true; resolve_band = false;
                lead = kept_lead; n_eval = kept_eval; n_steps = kept_steps; n_issued = kept_issued;
            }
        }
        const int p_end = resolving ? p_first : (ROLE == kRoleSearch ? p_first + 1 : M);
        for (int p_lo = p_first; p_lo < p_end; p_lo += tile_len) {
        const int p_hi = p_lo + tile_len;
        const double* e_base = regA;   // e_base[b] = sample b of e (or e*w)
        const double* w_base = regW;
        const double* c_base = regB;   // c_base[i] = C[i] (tiled: the tile's LDS copy while phase 3a runs)
        if constexpr (!RESIDENT) {
            double* tile_e = reinterpret_cast<double*>(smem + ap->hdr_bytes);
            const int staged = ap->tile_len + ap->tile_halo;
            wg_sync();  // the previous tile (or the sort histogram) is no longer read
            pc.mark(20);
            {
                // no room for C beside the samples: the predicate pass gets C in the samples' place
                // (sequential HBM reads instead of the predicate's scattered ones), the samples follow
                // once the live units are listed
                if (x_staging) {
                    // the flux of positions p_lo .. (patch: an index mapping), X in place, X of the tile and its halo to the slab
                    const int have = M - p_lo < staged ? (M - p_lo > 0 ? M - p_lo : 0) : staged;
                    slab_to_lds_async(tile_e, regA, p_lo, have, n, tid);
                    vmem_wait_all();
                    lds_barrier();
                    scan_tile_x(tile_e, have, staged, 0.0, reinterpret_cast<double*>(cumsum_scratch), tid);
                    const int got = have < staged ? have + 1 : staged;          // X entries formed
                    x_got = got;   // (X goes to the slab only if the tile turns out to have live cells: below)
                    for (int k = got + tid; k < staged; k += nt) tile_e[k] = -(double)(p_lo + k - M) * 1.0e300;
                } else {
                    const int avail = M + 1 + region_pad - p_lo;            // entries the slab still holds
                    const int valid = avail < staged ? (avail > 0 ? avail : 0) : staged;
                    if (TLS_SLAB_DMA) {
                        slab_to_lds_async(tile_e, regB, p_lo, valid, 0x7fffffff, tid);   // (p_lo even, regions 16-byte aligned)
                        vmem_wait_all();
                    } else {
                        copy_in_flight4(tile_e, regB + p_lo, valid);
                    }
                    for (int k = valid + tid; k < staged; k += nt) tile_e[k] = -(double)(p_lo + k - M) * 1.0e300;
                }
                c_base = tile_e - p_lo;
            }
            for (int row = tid; row < n_rows; row += nt) rt.live[row] = 0;
            if (tid == 0) s_work[3] = 0;
            wg_sync();
            pc.mark(12);
        }

        // ---- phase 3a: depth predicate over every trial cell -> lists of live units ----
        // dense rows: a lane owns kR consecutive T0 positions and walks all durations with
        // C[u0..u0+kR) held in registers; the chunk is live if its smallest window sum passes
        // (the mean is monotone in the window sum, so min() decides exactly).
        const bool exact_u = __builtin_amdgcn_readfirstlane((int)exact_mode) != 0;
        const double thr_hi = rule.dmin + rule.eps, thr_lo = rule.dmin - rule.eps;
        if (k_x > k_lo) {
            const int units0 = widths_c[k_lo].n_chunks;  // the shortest width has the most positions
            const int unit_lo = p_lo / kR;               // tile bounds are multiples of kR * 64
            const int unit_hi = p_hi / kR < units0 ? p_hi / kR : units0;
            const int n_dense = k_x - k_lo;
            for (int tile = wave; unit_lo + tile * kWave < unit_hi; tile += nw) {
                const int unit = unit_lo + tile * kWave + lane;
                const int u0 = unit * kR;
                const int u0c = u0 < M + 1 ? u0 : M + 1;  // lanes past the row read sentinels and are masked
                double c_lo[kR];
                TLS_CHECK(*ap, u0c >= p_lo && u0c + kR - 1 <= M + region_pad && (RESIDENT || u0c + kR - 1 < p_lo + ap->tile_len + ap->tile_halo), kChkPredicateRead);
#pragma unroll
                for (int r = 0; r < kR; ++r) c_lo[r] = c_base[u0c + r];
                // Lane j collects the live mask of row k_lo + j of this 64-unit tile; the list slots of
                // ALL rows are then reserved with one LDS atomic instruction (one lane per row) instead
                // of one dependent atomic round trip per row.
                int row_lo = 0, row_hi = 0;   // the two halves of the lane's (row's) live mask
                unsigned long long band_mask = 0ull;
                const unsigned long long valid_mask = ballot64(unit < unit_hi);
                // kRowBatch durations per step: all LDS reads of the step are in flight together
#ifndef TLS_ROW_BATCH
#define TLS_ROW_BATCH 2
#endif
                constexpr int kRowBatch = TLS_ROW_BATCH;
                for (int k = k_lo; k < k_x; k += kRowBatch) {
                    int dv[kRowBatch];
                    double inv[kRowBatch], dC[kRowBatch];
                    double c_hi[kRowBatch][kR];
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        const int kk = k + j < k_x ? k + j : k_x - 1;  // the tail repeats the last row
                        dv[j] = widths_c[kk].width;
                        inv[j] = widths_c[kk].inv_d;
                        const int hi0 = min(u0 + dv[j], M + 1);  // past the grid: sentinels
                        TLS_CHECK(*ap, hi0 + kR - 1 <= M + region_pad && (RESIDENT || unit >= unit_hi || hi0 + kR - 1 < p_lo + ap->tile_len + ap->tile_halo), kChkPredicateRead);
#pragma unroll
                        for (int r = 0; r < kR; ++r) c_hi[j][r] = c_base[hi0 + r];
                    }
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        double m = c_hi[j][0] - c_lo[0];
#pragma unroll
                        for (int r = 1; r < kR; ++r) m = fmax(m, c_hi[j][r] - c_lo[r]);
                        dC[j] = m;   // the chunk's largest X[i+d] - X[i]: its deepest window
                    }
#pragma unroll
                    for (int j = 0; j < kRowBatch; ++j) {
                        if (k + j < k_x) {
                            // (a lane past the row's units reads sentinels or foreign cells: it is masked below and
                            // must not raise `undecided`)
                            // the row's live lanes as a wave-uniform mask, straight from the compare
                            unsigned long long mask;
                            if (exact_u) {   // (a scalar branch: the whole workgroup is in one mode)
                                bool und_j = false;
                                mask = ballot64(depth_pass(dC[j], inv[j], (double)dv[j], dmin, rule.eps, true, und_j));
                            } else {         // fast mode: no branch per row; a chunk inside the band is noted for the tile
                                const double m_fast = dC[j] * inv[j];
                                mask = ballot64(m_fast > thr_hi);
                                const unsigned long long band_j = ballot64(m_fast >= thr_lo) & ~mask & valid_mask;   // (two compares, the rest on the scalar unit)
                                if (band_j != 0ull) {   // (a scalar branch, rarely taken)
                                    if (rule.band_count != nullptr) {
                                        // the chunk's deepest window is inside the band: its windows there are noted one by one
                                        // (the others lie below it: decided)
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
                            if (n_dense <= kWave) {   // lane (row) of row_mask := mask
                                if constexpr (RESIDENT) {
                                    set_lane(row_lo, (int)(unsigned int)mask, k + j - k_lo);
                                    set_lane(row_hi, (int)(unsigned int)(mask >> 32), k + j - k_lo);
                                } else if (lane == k + j - k_lo) {   // (the slab kernels keep M0 for their LDS transfers)
                                    row_lo = (int)(unsigned int)mask;
                                    row_hi = (int)(unsigned int)(mask >> 32);
                                }
                            } else {   // more dense rows than lanes (never with the default duration grid)
                                push_live(((mask >> lane) & 1ull) != 0ull, (unsigned int)unit, &rt.live[k + j - k_lo],
                                          chunk_list + widths_c[k + j].list_base, lane);
                            }
                        }
                    }
                }
                undecided |= (band_mask & valid_mask) != 0ull;   // (any lane's sends the whole workgroup to exact mode)
                const unsigned long long row_mask = ((unsigned long long)(unsigned int)row_hi << 32) | (unsigned int)row_lo;
                if (n_dense <= kWave) {
                    unsigned int base = 0;
                    const unsigned int mine = (unsigned int)__popcll(row_mask);
                    if (mine) base = atomicAdd(&rt.live[lane], mine);      // lane j: row k_lo + j
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
        // strided rows (long durations, core.py:50-58): kR strided positions per lane while the
        // stride allows the tiled dot product, else one position per lane
        // (one row per wave: rows are independent, and a row of a few hundred units would leave
        // most waves idle if all of them walked it together)
#if TLS_STRIDED_TICKETS
        // (rows are handed out through a ticket counter, most positions first: a wave that is done with its dense tiles
        // or with a short row takes the next one)
        for (;;) {
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&s_work[3], 1);
            const int k = (k_x > k_lo ? k_x : k_lo) + __builtin_amdgcn_readfirstlane(ticket);
            if (k >= k_hi) break;
#else
        for (int k = (k_x > k_lo ? k_x : k_lo) + wave; k < k_hi; k += nw) {
#endif
            const int d = widths_c[k].width, xth = widths_c[k].xth, n_pos = widths_c[k].n_pos;
            const int n_units = widths_c[k].n_chunks;
            const double inv_d = widths_c[k].inv_d;
            unsigned int* list = chunk_list + widths_c[k].list_base;
            // the row belongs to this wave alone: its list tail is a register, not an LDS atomic
            unsigned int n_listed = 0;
            const unsigned long long below = (1ull << lane) - 1ull;
            // a window wider than an LDS tile: the row is listed once (with the first tile), from the slab
            // (two separate loads below, never a pointer select: c_base is an LDS pointer biased by -p_lo, and a
            // select with a global pointer would turn it into a flat address outside the LDS aperture)
            const bool oversize = !RESIDENT && widths_c[k].oversize != 0;
            if (oversize && p_lo != 0) continue;
            if (widths_c[k].tiled) {
                const int span = kR * xth;  // samples between the first windows of two units
                const int unit_lo = (p_lo + span - 1) / span;
                const int unit_hi = (p_hi + span - 1) / span < n_units ? (p_hi + span - 1) / span : n_units;
                for (int tile = 0; unit_lo + tile * kWave < unit_hi; ++tile) {
                    const int unit = unit_lo + tile * kWave + lane;
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
                        if (!live && m_fast >= thr_lo && unit < unit_hi) {
#pragma unroll
                            for (int r = 0; r < kR; ++r) {
                                const double dXr = c_hi[r] - c_lo[r];
                                if (dXr * inv_d >= thr_lo) band_window(rule, k, (uc * kR + r) * xth, dXr, undecided);
                            }
                        }
                    }
                    live = live && unit < unit_hi;
                    const unsigned long long mask = ballot64(live);
                    if (live) list[n_listed + (unsigned int)__popcll(mask & below)] = (unsigned int)unit;
                    n_listed += (unsigned int)__popcll(mask);
                }
            } else {
                const int unit_lo = oversize ? 0 : (p_lo + xth - 1) / xth;
                const int unit_hi = oversize ? n_pos : ((p_hi + xth - 1) / xth < n_pos ? (p_hi + xth - 1) / xth : n_pos);
                for (int tile = 0; unit_lo + tile * kWave < unit_hi; ++tile) {
                    const int unit = unit_lo + tile * kWave + lane;
                    bool live = false;
                    if (unit < unit_hi) {
                        const int i = unit * xth;
                        double dC;
                        if (oversize) dC = regB[i + d] - regB[i];
                        else dC = c_base[i + d] - c_base[i];
                        bool und_w = false;
                        live = depth_pass(dC, inv_d, (double)d, dmin, rule.eps, exact_mode, und_w);
                        if (und_w) band_window(rule, k, i, dC, undecided);
                    }
                    const unsigned long long mask = ballot64(live);
                    if (live) list[n_listed + (unsigned int)__popcll(mask & below)] = (unsigned int)unit;
                    n_listed += (unsigned int)__popcll(mask);
                }
            }
            TLS_CHECK(*ap, n_listed <= (unsigned int)widths_c[k].n_chunks, kChkListCap);
            // (every lane stores the same value: a store under `lane == 0` here and the ticket fetch under `lane == 0`
            // at the loop head were threaded together by the compiler in the search-role instantiation -- lane 0 left
            // for the next ticket, the other 63 lanes took the row again, for ever)
            if (ROLE != kRoleAll || lane == 0) rt.live[k - k_lo] = n_listed;
        }
        wg_sync();
        pc.mark(9);
        if constexpr (!RESIDENT) {
            // a tile in which no cell passed the depth predicate has nothing to evaluate: its samples are
            // not staged (second staging of the tile) and the dot-product phase is skipped
            unsigned int tile_live = 0;
#pragma unroll 1
            for (int row = 0; row < n_rows; ++row) tile_live += rt.live[row];
            if (__builtin_amdgcn_readfirstlane((int)tile_live) == 0) { wg_sync(); continue; }
        }
        // Uniform weights, fast mode with X formed at staging time: the dot products are evaluated ON X -- summation by
        // parts, sum_j q_j e_{i+j} = sum_{j<=L} g_j X_{i+j} with the difference taps g (host: L.g) --, so the tile keeps its X
        // for phase 3b: no second staging of the samples, no X written to the slab and read back (Kepler size: 4.0 -> 2.8 MB
        // of slab traffic per period).  X is exact here (a sum of multiples of 2^-53 below 1), the taps carry one rounding.
        [[maybe_unused]] const bool x_dot = !RESIDENT && UNIFORM_W && x_staging && ap->g != nullptr;
        [[maybe_unused]] const double* x_tile = nullptr;
        if constexpr (!RESIDENT) {
          if (x_dot) {
            // (the tile's X, biased by -p_lo like every tile pointer.  x_tile is read through the LDS address space only --
            // x_load<true>, load_taps --: c_base is a global pointer on the other path, and one pointer that is either is FLAT)
            x_tile = reinterpret_cast<const double*>(smem + ap->hdr_bytes) - p_lo;
            e_base = x_tile;
            c_base = regB;   // (not read on this path: one value on both, so that the reads on the other stay global loads)
          } else {
            // the folded samples replace C in the tile; the few C values phase 3b needs come from the slab
            double* tile_e = reinterpret_cast<double*>(smem + ap->hdr_bytes);
            const int staged = ap->tile_len + ap->tile_halo;
            double* tile_w = tile_e + staged;
            if (x_staging) {   // ... to which a tile with live cells now sends its X (tile + halo); the LDS copy is then free
                copy_out_stream(regB + p_lo, tile_e, x_got, tid);
                lds_barrier();
            }
            {
                if (TLS_SLAB_DMA && (n & 1) == 0) {
                    stage_samples_async<UNIFORM_W>(tile_e, tile_w, regA, regW, p_lo, staged, n, M, tid);
                    vmem_wait_all();
                    lds_barrier();
                    finish_samples<UNIFORM_W>(tile_e, tile_w, p_lo, staged, M, tid);
                } else {
                    stage_samples<UNIFORM_W>(tile_e, tile_w, regA, regW, p_lo, staged, n, M);
                }
            }
            e_base = tile_e - p_lo;
            w_base = tile_w - p_lo;
            c_base = regB;
            wg_sync();
            pc.mark(12);
          }
        }
        // ---- pruning (exact): drop the units that cannot win before they reach phase 3b ------------
        // Worth its passes only when many cells passed the depth predicate (noisy light curves):
        // decided per period (and tile) from the number of live units.
        double T = -INFINITY;
        bool prune_now = false;
        // rounding allowance of window_bound per sample of a window: the sequential prefix sum is off by at most
        // half an ulp of its total per step
        [[maybe_unused]] const double slack_unit = ap->slack_unit;
        int p2_blocks = 0;
        float* const ulist = reinterpret_cast<float*>(chunk_list + ap->list_cap);   // bound of every live unit
        if (prune_on) {
            unsigned int total_live = 0;
#pragma unroll 1
            for (int row = 0; row < n_rows; ++row) total_live += rt.live[row];
            total_live = (unsigned int)__builtin_amdgcn_readfirstlane((int)total_live);  // uniform: keep it scalar
            prune_now = (long long)total_live >= ap->prune_min_live;
        }
        if (ap->phase_cycles && tid == 0) {   // developer statistics beside the phase clocks
            unsigned int total_live = 0;
            for (int row = 0; row < n_rows; ++row) total_live += rt.live[row];
            atomicAdd(&ap->phase_cycles[32], (unsigned long long)total_live);
            if (prune_now) atomicAdd(&ap->phase_cycles[36], 1ull);
        }
        unsigned int* active_list = chunk_list;   // the lists phase 3b reads (the pruning pass writes a second set)
        int n_groups = 0;
        if (prune_now) {
            // (0) the live units of all rows in groups of 64, numbered row by row (batch_start[row] = first group
            // of the row): the passes below hand GROUPS to the waves, so that a row with 600 live units and a
            // row with 6 cost the workgroup what they cost, not what the slowest wave's rows add up to
            if (wave == 0) {
                unsigned int carry = 0;
                for (int r0 = 0; r0 < n_rows; r0 += kWave) {
                    const int row = r0 + lane;
                    const unsigned int mine = row < n_rows ? (rt.live[row] + kWave - 1) / kWave : 0u;
                    const unsigned int incl = wave_inclusive_sum_u32(mine);
                    if (row < n_rows) rt.batch_start[row] = carry + incl - mine;
                    carry += (unsigned int)lane_value((int)incl, kWave - 1);
                }
                if (lane == 0) rt.batch_start[n_rows] = carry;
            }
            // (1) coarse prefix sum of e^2: P2[b] = sum of e_k^2 over k < b * 2^p2_shift
            if (!p2_ready) {   // once per period, by the first tile that prunes
                p2_ready = true;
                const int sh = ap->p2_shift, G = 1 << sh;
                p2_blocks = (M + G - 1) >> sh;
                if (RESIDENT && G >= 8) {
                    // eight consecutive samples per thread, G/8 neighbouring lanes per block
                    const int per_blk = G >> 3;   // lanes per block: 2, 4 or 8 (G = 16, 32, 64)
#pragma unroll 1
                    for (int c0 = 0; c0 < M; c0 += 8 * nt) {
                        const int k0 = c0 + 8 * tid;
                        double acc = 0.0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const double ev = k0 + j < M ? regA[k0 + j] : 0.0; acc = fma(ev, ev, acc); }
                        for (int dlt = 1; dlt < per_blk; dlt <<= 1) acc += __shfl_xor(acc, dlt, kWave);
                        if ((lane & (per_blk - 1)) == 0 && k0 < M) P2[(k0 >> sh) + 1] = acc;
                    }
                } else if (G <= kWave) {   // short blocks: one thread each
#pragma unroll 1
                    for (int b = tid; b < p2_blocks; b += nt) {
                        const int hi = (b + 1) * G < M ? (b + 1) * G : M;
                        double acc = 0.0;
#pragma unroll 4
                        for (int kk = b * G; kk < hi; ++kk) {
                            const double ev = RESIDENT ? regA[kk] : 1.0 - regA[kk < n ? kk : kk - n];   // the slab keeps f
                            acc = fma(ev, ev, acc);
                        }
                        P2[b + 1] = acc;
                    }
                } else {            // long blocks (series in the HBM slab): one wave each, coalesced
#pragma unroll 1
                    for (int b = wave; b < p2_blocks; b += nw) {
                        const int hi = (b + 1) * G < M ? (b + 1) * G : M;
                        double acc = 0.0;
#pragma unroll 2
                        for (int kk = b * G + lane; kk < hi; kk += kWave) {
                            const double ev = RESIDENT ? regA[kk] : 1.0 - regA[kk < n ? kk : kk - n];
                            acc = fma(ev, ev, acc);
                        }
#pragma unroll
                        for (int delta = kWave / 2; delta > 0; delta >>= 1) acc += __shfl_down(acc, delta, kWave);
                        if (lane == 0) P2[b + 1] = acc;
                    }
                }
                wg_sync();
                if (wave == 0) {    // inclusive scan of at most kP2MaxBlocks block sums
                    const int per = (p2_blocks + kWave - 1) / kWave;
                    const int lo = lane * per < p2_blocks ? lane * per : p2_blocks;
                    const int hi = lo + per < p2_blocks ? lo + per : p2_blocks;
                    double local = 0.0;
#pragma unroll 1
                    for (int b = lo; b < hi; ++b) local += P2[b + 1];
                    const double incl = wave_inclusive_sum(local);
                    double run = incl - local;
#pragma unroll 1
                    for (int b = lo; b < hi; ++b) { run += P2[b + 1]; P2[b + 1] = run; }
                    if (lane == 0) P2[0] = 0.0;
                }
            }
            wg_sync();
            p2_blocks = (M + (1 << ap->p2_shift) - 1) >> ap->p2_shift;
            n_groups = __builtin_amdgcn_readfirstlane((int)rt.batch_start[n_rows]);
            pc.mark(22);
            // (2) the bound of every live unit, group by group; each wave remembers its most promising one
            float cand_u = -INFINITY;
            int cand_k = 0x7fffffff, cand_unit = 0;
            {
                // (the list entry of the NEXT group is requested before this group's bound is formed: the lists live in
                // global memory, and an L2 round trip per group, one after the other, was most of this pass)
                int row = 0, row_next = 0, unit_next = 0;
                if (wave < n_groups) {
                    while (wave >= __builtin_amdgcn_readfirstlane((int)rt.batch_start[row_next + 1])) ++row_next;
                    const int idx0 = (wave - (int)rt.batch_start[row_next]) * kWave + lane;
                    if (idx0 < (int)rt.live[row_next]) unit_next = (int)chunk_list[widths_c[k_lo + row_next].list_base + idx0];
                }
#pragma unroll 1
                for (int g = wave; g < n_groups; g += nw) {
                    row = row_next;
                    const int unit_now = unit_next;
                    if (g + nw < n_groups) {
                        while (g + nw >= __builtin_amdgcn_readfirstlane((int)rt.batch_start[row_next + 1])) ++row_next;
                        const int idx1 = (g + nw - (int)rt.batch_start[row_next]) * kWave + lane;
                        unit_next = idx1 < (int)rt.live[row_next] ? (int)chunk_list[widths_c[k_lo + row_next].list_base + idx1] : 0;
                    }
                    const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
                    const int n_live = __builtin_amdgcn_readfirstlane((int)rt.live[row]);
                    const int xth = widths_c[k].xth, tiled = widths_c[k].tiled, d = widths_c[k].width;
                    const int list_base = widths_c[k].list_base, prunable = widths_c[k].prunable;
                    const double inv_d = widths_c[k].inv_d, dd = (double)d;
                    const double ov = widths_c[k].overshoot, k_mono = widths_c[k].k_mono, var_q = widths_c[k].var_q;
                    const double sum_q2 = widths_c[k].sum_q2;
                    const const_screen_ptr scr = screens_c + k;
                    // (the screen reads kSeg + 1 values of X per window through c_base: LDS when the series is resident or
                    // the tile holds X beside the samples; not worth it from the HBM slab)
                    const bool screened = RESIDENT && prunable && scr->valid != 0;
                    const double slack = slack_unit * (double)(d + 64);
                    const int reach = tiled ? (kR - 1) * xth + d : d;   // samples covered by the windows of a unit
                    const int step = tiled ? kR * xth : xth;            // samples between two units
                    const int idx = (g - (int)rt.batch_start[row]) * kWave + lane;
                    const bool valid = idx < n_live;
                    const int unit = valid ? unit_now : 0;
                    const int b = unit * step;
                    float u = INFINITY;   // rows without a valid bound are always evaluated
                    if (screened) {
                        // tight bound, window by window; the unit keeps its best window's
                        // (an undecided window is listed again by phase 3b or the re-listing, which raise the flag)
                        bool und_b = false;
                        u = unit_bound(c_base, b, tiled ? kR : 1, xth, d, dd, inv_d, ov, sum_q2, scr, P2, ap->p2_shift, p2_blocks,
                                       dmin, rule.eps, exact_mode, und_b, slack);
                        undecided |= und_b && valid;
                    } else if (prunable) {
                        // (every window of a live unit goes through depth_pass here as it does in phase 3b of the plain
                        // kernel: the two variants must send the SAME periods through exact mode to agree bit for bit)
                        bool und_c = false;
                        double dX_max, dX_min;
                        if (!RESIDENT && widths_c[k].oversize) dX_max = regB[b + d] - regB[b];   // (no pointer select, see phase 3a)
                        else dX_max = c_base[b + d] - c_base[b];
                        dX_min = dX_max;
                        (void)depth_pass(dX_max, inv_d, dd, dmin, rule.eps, exact_mode, und_c);
                        if (tiled) {
#pragma unroll
                            for (int r = 1; r < kR; ++r) {
                                const double dX = c_base[b + r * xth + d] - c_base[b + r * xth];
                                (void)depth_pass(dX, inv_d, dd, dmin, rule.eps, exact_mode, und_c);
                                dX_min = fmin(dX_min, dX); dX_max = fmax(dX_max, dX);
                            }
                        }
                        undecided |= und_c && valid;
                        u = cell_bound(dX_max, dX_min, dd, inv_d, ov, k_mono, var_q,
                                       coarse_e2(P2, b, b + reach, ap->p2_shift, p2_blocks));
                    }
                    if (valid) {
                        if (prunable && u > cand_u) { cand_u = u; cand_k = k; cand_unit = unit; }
                        ulist[list_base + idx] = u;
                    }
                }
            }
            for (int row = tid; row < n_rows; row += nt) rt.singles[row] = 0;   // (5a)'s counters; published by (4)'s barrier
            pc.mark(23);
            // (3) each wave evaluates its candidate exactly (lanes over the template taps)
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                const float ou_ = __shfl_down(cand_u, delta, kWave);
                const int ok_ = __shfl_down(cand_k, delta, kWave), on_ = __shfl_down(cand_unit, delta, kWave);
                const bool take = ou_ > cand_u || (ou_ == cand_u && (ok_ < cand_k || (ok_ == cand_k && on_ < cand_unit)));
                if (take) { cand_u = ou_; cand_k = ok_; cand_unit = on_; }
            }
            const int ck = __builtin_amdgcn_readfirstlane(cand_k);
            const int cu = __builtin_amdgcn_readfirstlane(cand_unit);
            // The candidate's statistic only sets the threshold: its taps are summed in another order
            // than phase 3b uses, so it may differ from the reported value in the last bits.  The
            // cell itself survives the threshold and is evaluated again by 3b like any other.
            Lead trial = no_lead();
            if (ck < k_hi) {  // this wave has a candidate
                const int d = widths_c[ck].width, L = widths_c[ck].q_len, xth = widths_c[ck].xth;
                const int tiled = widths_c[ck].tiled, q_offset = widths_c[ck].q_offset;
                const double overshoot = widths_c[ck].overshoot, sum_q2 = widths_c[ck].sum_q2;
                const double inv_d = widths_c[ck].inv_d;
                const int n_win = tiled ? kR : 1;
                const int i0 = tiled ? cu * kR * xth : cu * xth;   // first sample of the first window
                const double* qv = ap->q + q_offset;
                double Bc[kR];
#pragma unroll
                for (int r = 0; r < kR; ++r) Bc[r] = 0.0;
#pragma unroll 2
                for (int tt = lane; tt < L; tt += kWave) {
                    const double qt = qv[tt];
#pragma unroll
                    for (int r = 0; r < kR; ++r)
                        if (r < n_win) Bc[r] = fma(qt, e_base[i0 + r * xth + tt], Bc[r]);
                }
#pragma unroll
                for (int r = 0; r < kR; ++r)
#pragma unroll
                    for (int delta = kWave / 2; delta > 0; delta >>= 1) Bc[r] += __shfl_down(Bc[r], delta, kWave);
                // lane r takes window r: the same bookkeeping as any evaluated cell
                double Bmine = 0.0;
#pragma unroll
                for (int r = 0; r < kR; ++r) { const double v = lane_value(Bc[r], 0); if (lane == r) Bmine = v; }
                if (lane < n_win) {
                    const int i = i0 + lane * xth;
                    unsigned int ignored = 0;
                    consider<UNIFORM_W, !RESIDENT>(trial, c_base[i], c_base[i + d], i, inv_d, (double)d, rule, overshoot, sum_q2, Bmine, ck, ignored, undecided, widths_c, regB);
                }
            }
            // (4) T = the best statistic any evaluated cell has reached (this and earlier tiles)
            // loosened by far more than the summation-order difference (1e-12 relative)
            // (both are ESTIMATES of the statistic, good to rule.reach / 4: consider)
            const double slack = fmin(2.0 * rule.reach, 0.5);
            const double loose = trial.stat < 0.0 ? trial.stat * (1.0 - slack) : trial.stat * (1.0 + slack);
            const double mine = lead.stat < 0.0 ? lead.stat * (1.0 - slack) : lead.stat * (1.0 + slack);
            double mstat = fmin(mine, loose);   // best: cells evaluated by 3b in earlier tiles
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) mstat = fmin(mstat, __shfl_down(mstat, delta, kWave));
            if (lane == 0) wbest[wave].stat = mstat;   // wbest is idle until phase 4
            wg_sync();
            double g = wbest[0].stat;
            for (int v = 1; v < nw; ++v) g = fmin(g, wbest[v].stat);
            T = lane_value(-g, 0);                     // uniform (scalar registers); -inf while nothing has been evaluated
            pc.mark(24);
        }
        if (prune_now) {
            // (5a) keep the units whose bound reaches T: group by group again, compacted into the workgroup's SECOND
            // set of lists (the slots of a row handed out by an LDS atomic per group; the order inside a list is
            // irrelevant).  The first set is still being read by the other waves, so nothing is compacted in place.
            unsigned int* const kept_list = chunk_list + 2 * ap->list_cap;
            const unsigned long long below = (1ull << lane) - 1ull;
            // (all of a wave's entries are requested before the first is used: up to kAhead groups in flight)
            constexpr int kAhead = 4;
#pragma unroll 1
            for (int g0 = wave; g0 < n_groups; g0 += kAhead * nw) {
            unsigned int unit_a[kAhead];
            float u_a[kAhead];
            int row_a[kAhead];
            {
                int row = 0;
#pragma unroll
                for (int j = 0; j < kAhead; ++j) {
                    const int g = g0 + j * nw;
                    unit_a[j] = 0u; u_a[j] = -INFINITY; row_a[j] = 0;
                    if (g < n_groups) {
                        while (g >= __builtin_amdgcn_readfirstlane((int)rt.batch_start[row + 1])) ++row;
                        row_a[j] = row;
                        const int list_base = widths_c[k_lo + row].list_base;
                        const int idx = (g - (int)rt.batch_start[row]) * kWave + lane;
                        if (idx < (int)rt.live[row]) { unit_a[j] = chunk_list[list_base + idx]; u_a[j] = ulist[list_base + idx]; }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                const int g = g0 + j * nw;
                if (g >= n_groups) break;
                const int row = __builtin_amdgcn_readfirstlane(row_a[j]);
                const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
                const int list_base = widths_c[k].list_base;
                const unsigned int unit = unit_a[j];
                const float u = u_a[j];
                const bool sel = (double)u >= T;   // invalid lanes hold -inf
                const unsigned long long mask = ballot64(sel);
                if (mask) {
                    unsigned int base = 0;
                    if (lane == 0) base = atomicAdd(&rt.singles[row], (unsigned int)__popcll(mask));
                    base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                    if (sel) kept_list[list_base + base + (unsigned int)__popcll(mask & below)] = unit;
                }
            }
            }
            wg_sync();
            for (int r2 = tid; r2 < n_rows; r2 += nt) { rt.live[r2] = rt.singles[r2]; rt.singles[r2] = 0; }
            active_list = kept_list;
            wg_sync();
        }
        // Per row: (5b) re-list SPARSE rows.  A handful of
        // live chunks would still occupy a whole 64-lane batch with kR FMAs per tap, so such rows
        // are listed position by position (only positions that pass the predicate and the bound)
        // behind their chunk entries; phase 3b then runs them one window per lane, which costs a
        // fraction of the tiled form when most lanes would idle.
        for (int row = wave; row < n_rows; row += nw) {
            const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
            int n_live = __builtin_amdgcn_readfirstlane((int)rt.live[row]);
            const int xth = widths_c[k].xth, n_units = widths_c[k].n_chunks, tiled = widths_c[k].tiled;
            const int d = widths_c[k].width, list_base = widths_c[k].list_base;
            const double inv_d = widths_c[k].inv_d, dd = (double)d;
            const bool bound_row = prune_now && widths_c[k].prunable;
            const bool screened_row = bound_row && screens_c[k].valid != 0;
            const double ov = widths_c[k].overshoot, k_mono = widths_c[k].k_mono, var_q = widths_c[k].var_q;
            unsigned int* list = active_list + list_base;
            unsigned int count = 0;
            // which units are re-listed: all of a sparse row; of a longer row the LAST batch when it is mostly empty
            // (it would run 64 lanes wide for a handful of units) -- kept only if its positions fit one batch
            const bool sparse = n_live <= kSparseRow;
            const int n_tail = sparse ? n_live : (TLS_TAIL_RELIST && RESIDENT ? (n_live & (kWave - 1)) : 0);
            const int first = n_live - n_tail;
            if (tiled && n_tail > 0 && n_tail <= kTailMax && n_units >= (kR + 1) * kSparseRow && first + n_tail * kR <= n_units) {
                // the units travel in registers (lane j: unit first + j): their list slots take the positions
                const unsigned int my_unit = lane < n_tail ? list[first + lane] : 0u;
#pragma unroll 1
                for (int base = 0; base < n_tail * kR; base += kWave) {
                    const int idx = base + lane;
                    bool pass = false;
                    const int u = __shfl((int)my_unit, (idx / kR) & (kWave - 1), kWave) * kR + idx % kR;   // T0 position index
                    if (idx < n_tail * kR) {
                        const int i = u * xth;
                        double dX;
                        if (x_dot) dX = x_load<true>(x_tile, i + d) - x_load<true>(x_tile, i);
                        else dX = c_base[i + d] - c_base[i];   // past the grid: sentinel
                        bool und_w = false;
                        pass = depth_pass(dX, inv_d, dd, dmin, rule.eps, exact_mode, und_w);
                        if (und_w) band_window(rule, k, i, dX, undecided);
                        if (bound_row && pass) {
                            if (RESIDENT && screened_row)
                                pass = (double)window_bound(c_base, i, d, dd, inv_d, ov, widths_c[k].sum_q2, screens_c + k, P2,
                                                            ap->p2_shift, p2_blocks, dmin, rule.eps, exact_mode, undecided,
                                                            slack_unit * (double)(d + 64)) >= T;
                            else
                                pass = (double)cell_bound(dX, dX, dd, inv_d, ov, k_mono, var_q,
                                                          coarse_e2(P2, i, i + d, ap->p2_shift, p2_blocks)) >= T;
                        }
                    }
                    const unsigned long long mask = ballot64(pass);
                    if (pass) list[first + count + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = (unsigned int)u;
                    count += (unsigned int)__popcll(mask);
                }
                if (sparse || count <= (unsigned int)kWave) {
                    n_live = first;   // (a sparse row whose every window was pruned has nothing left at all)
                } else {          // more positions than one batch: the tail stays in chunk form
                    if (lane < n_tail) list[first + lane] = my_unit;
                    count = 0;
                }
            }
            TLS_CHECK(*ap, (unsigned int)n_live + count <= (unsigned int)n_units, kChkSinglesCap);
            if (lane == 0) { rt.live[row] = (unsigned int)n_live; rt.singles[row] = count; }
        }
        pc.mark(25);
        wg_sync();
        if (wave == 0) {  // exclusive scan of the batch counts over the rows
            unsigned int carry = 0;
            for (int r0 = 0; r0 < n_rows; r0 += kWave) {
                const int row = r0 + lane;
                unsigned int mine = 0;
                if (row < n_rows) {
                    mine = (rt.live[row] + kWave - 1) / kWave + (rt.singles[row] + kWave - 1) / kWave;
                }
                unsigned int incl = mine;
#pragma unroll
                for (int dlt = 1; dlt < kWave; dlt <<= 1) {
                    const unsigned int o = __shfl_up(incl, dlt, kWave);
                    if (lane >= dlt) incl += o;
                }
                if (row < n_rows) rt.batch_start[row] = carry + incl - mine;
                carry += __shfl(incl, kWave - 1, kWave);
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

        // ---- phase 3b: sliding dot products, 64 live units of one duration per wave ----
#if TLS_BODY_PRIO
        if constexpr (RESIDENT) __builtin_amdgcn_s_setprio(0);   // (see the period's start)
#endif
        {
            // batches are numbered from the widest row down (long templates first) and handed out
            // dynamically through an LDS ticket counter
            const unsigned int total_batches = (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[n_rows]);
            int row = n_rows > 0 ? n_rows - 1 : 0;  // batch numbers only decrease within a wave
            for (;;) {
                unsigned int g = 0;
                if (lane == 0) g = atomicAdd(rt.next_batch, 1u);
                g = (unsigned int)__builtin_amdgcn_readfirstlane((int)g);
                if (g >= total_batches) break;
                const unsigned int gg = total_batches - 1 - g;
                while (gg < (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[row])) --row;
                // every per-row constant is fetched with wave-uniform (scalar) loads BEFORE any
                // lane-dependent branch: an address that reaches a load through a divergent join
                // is treated as divergent and the template taps would stop being scalar loads
                const int k = __builtin_amdgcn_readfirstlane(k_lo + row);
                const int d = widths_c[k].width, L = widths_c[k].q_len, xth = widths_c[k].xth;
                const int q_offset = widths_c[k].q_offset, list_base = widths_c[k].list_base;
                const double overshoot = widths_c[k].overshoot, sum_q2 = widths_c[k].sum_q2;
                const double inv_d = widths_c[k].inv_d, dd = (double)d;
                const int tiled = widths_c[k].tiled;
                const int n_singles = __builtin_amdgcn_readfirstlane((int)rt.singles[row]);
                const int n_live = __builtin_amdgcn_readfirstlane((int)rt.live[row]);
                // a row's batches: its chunks (or, strided rows, its positions) first, then its re-listed positions,
                // which sit behind the chunk entries
                const unsigned int in_row = gg - (unsigned int)__builtin_amdgcn_readfirstlane((int)rt.batch_start[row]);
                const unsigned int chunk_batches = ((unsigned int)n_live + kWave - 1) / kWave;
                const bool relisted = in_row >= chunk_batches;
                const unsigned int slot = (relisted ? in_row - chunk_batches : in_row) * kWave + lane;
                const bool have = slot < (unsigned int)(relisted ? n_singles : n_live);
                const int unit = have ? (int)active_list[list_base + (relisted ? n_live : 0) + slot] : 0;
                const const_f64_ptr q = (x_dot ? (const_f64_ptr)ap->g : q_all) + q_offset;
                const int Lx = x_dot ? L + 1 : L;   // taps the dot product runs over (difference taps: one more)
                const unsigned int evals_before = n_eval;
                if (COUNTING && ap->counters) {   // what the loops below issue per lane, padding and idle lanes included
                    const int reach = (tiled && !relisted) ? (kR - 1) * xth : 0;
                    n_issued += (unsigned long long)((Lx + reach + kU - 1) / kU * kU) * (reach ? kR : 1) * (UNIFORM_W ? 1 : 2);
                }
                [[maybe_unused]] const double errB2 = SCR ? widths_c[k].screen_c * ap->e_abs_max + 1e-30 : 0.0;
                if (SCR && tiled && !relisted) {
                    if constexpr (SCR) {
                        // kR windows per lane in packed fp32, then the screen
                        const int b = unit * kR * xth;
                        TLS_CHECK(*ap, !have || (b >= p_lo && b + (L + (kR - 1) * xth + kU - 1) / kU * kU <= M + 1 + region_pad), kChkDotWindow);
                        const unsigned pa = ((b & 1) ? eh1_addr - 4u : eh_addr) + 4u * (unsigned)b;
                        const const_f32_ptr q32 = (const_f32_ptr)ap->q32 + q_offset;
                        const const_f32_ptr q32s = (const_f32_ptr)ap->q32 + (ap->q32_shifted + q_offset);
                        float B32[kR];
                        switch (xth) {
                            case 1: dot_windows32<1>(pa, q32, q32s, L, B32); break;
                            case 2: dot_windows32<2>(pa, q32, q32s, L, B32); break;
                            case 3: dot_windows32<3>(pa, q32, q32s, L, B32); break;
                            case 4: dot_windows32<4>(pa, q32, q32s, L, B32); break;
                            case 5: dot_windows32<5>(pa, q32, q32s, L, B32); break;
                            default:
#pragma unroll
                                for (int r = 0; r < kR; ++r) {
                                    const int br = b + r * xth;
                                    B32[r] = dot_window32(((br & 1) ? eh1_addr - 4u : eh_addr) + 4u * (unsigned)br, q32, L);
                                }
                                break;
                        }
                        if (have)
                            screen_cells<kR>(lead, scr_slot, c_base, b, xth, d, inv_d, dd, rule, overshoot, sum_q2, B32, k, errB2, undecided,
                                             widths_c, regB, scr_env);
                    }
                } else if (tiled && !relisted) {
                    // kR windows per lane, xth samples apart
                    const int u0 = unit * kR;
                    const int b = u0 * xth;
                    // the unrolled loop reads samples b .. b + ceil((L + (kR-1)*xth) / kU) * kU - 1
                    TLS_CHECK(*ap, !have || (b >= p_lo && b + (Lx + (kR - 1) * xth + kU - 1) / kU * kU <= (RESIDENT ? M + 1 + region_pad : p_lo + ap->tile_len + ap->tile_halo)), kChkDotWindow);
                    const double* e = e_base + b;
                    double Bv[kR], Av[kR];
#pragma unroll
                    for (int r = 0; r < kR; ++r) { Bv[r] = 0.0; Av[r] = 0.0; }
                    const int Lr = Lx;
                    if constexpr (UNIFORM_W) {
                        switch (xth) {
                            case 1: dot_windows<true, 1>(e, q, Lr, Bv); break;
                            case 2: dot_windows<true, 2>(e, q, Lr, Bv); break;
                            case 3: dot_windows<true, 3>(e, q, Lr, Bv); break;
                            case 4: dot_windows<true, 4>(e, q, Lr, Bv); break;
                            case 5: dot_windows<true, 5>(e, q, Lr, Bv); break;
                            default: dot_windows_rt<true>(e, q, Lr, xth, Bv); break;
                        }
                    } else {
                        const double* wv = w_base + b;
                        const const_f64_ptr q2 = q2_all + q_offset;
                        switch (xth) {
                            case 1: dot_windows_weighted<true, 1>(e, wv, q, q2, Lr, Bv, Av); break;
                            case 2: dot_windows_weighted<true, 2>(e, wv, q, q2, Lr, Bv, Av); break;
                            case 3: dot_windows_weighted<true, 3>(e, wv, q, q2, Lr, Bv, Av); break;
                            case 4: dot_windows_weighted<true, 4>(e, wv, q, q2, Lr, Bv, Av); break;
                            case 5: dot_windows_weighted<true, 5>(e, wv, q, q2, Lr, Bv, Av); break;
                            default: dot_windows_weighted_rt<true>(e, wv, q, q2, Lr, xth, Bv, Av); break;
                        }
                    }
                    if constexpr (UNIFORM_W) {
#pragma unroll
                        for (int r = 0; r < kR; ++r) Av[r] = sum_q2;
                    }
                    if (have) {
                        if (x_dot) consider_cells<UNIFORM_W, !RESIDENT, kR, COUNTING, true>(lead, x_tile, b, xth, d, inv_d, dd, rule, overshoot, Av, Bv, k, n_eval, undecided, widths_c, regB);
                        else consider_cells<UNIFORM_W, !RESIDENT, kR, COUNTING>(lead, c_base, b, xth, d, inv_d, dd, rule, overshoot, Av, Bv, k, n_eval, undecided, widths_c, regB);
                    }
                } else {
                    // wide T0 strides and re-listed sparse rows: one window per lane
                    if (!RESIDENT && widths_c[k].oversize) {
                        // the window does not fit an LDS tile: the wave takes the batch's windows one after
                        // the other, lanes over the template taps, samples straight from the slab
                        // (e = 1 - f with the patch mapping of stage_samples)
                        const double* qg = ap->q + q_offset;
                        const double* q2g = UNIFORM_W ? nullptr : ap->q2 + q_offset;
                        const unsigned long long have_mask = ballot64(have);
                        double myB = 0.0, myA = sum_q2;
                        for (int s2 = 0; s2 < kWave; ++s2) {
                            if (!((have_mask >> s2) & 1ull)) continue;
                            const int iu = lane_value(unit, s2) * xth;
                            double Bs = 0.0, As = 0.0;
                            for (int tt = lane; tt < L; tt += kWave) {
                                const int pp = iu + tt, src = pp < n ? pp : pp - n;
                                double ev = 1.0 - regA[src];
                                if constexpr (!UNIFORM_W) { const double ww = regW[src]; As = fma(q2g[tt], ww, As); ev *= ww; }
                                Bs = fma(qg[tt], ev, Bs);
                            }
#pragma unroll
                            for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                                Bs += __shfl_down(Bs, delta, kWave);
                                if constexpr (!UNIFORM_W) As += __shfl_down(As, delta, kWave);
                            }
                            const double Bt = lane_value(Bs, 0), At = lane_value(As, 0);
                            if (lane == s2) { myB = Bt; if constexpr (!UNIFORM_W) myA = At; }
                        }
                        if (have) {
                            const int i = unit * xth;
                            consider<UNIFORM_W, !RESIDENT>(lead, regB[i], regB[i + d], i, inv_d, dd, rule, overshoot, myA, myB, k, n_eval, undecided, widths_c, regB);
                        }
                        n_steps += (unsigned long long)(n_eval - evals_before) * (unsigned long long)L;
                        continue;
                    }
                    const int i = unit * xth;
                    TLS_CHECK(*ap, !have || (i >= p_lo && i + (Lx + kU - 1) / kU * kU <= (RESIDENT ? M + 1 + region_pad : p_lo + ap->tile_len + ap->tile_halo)), kChkDotWindow);
                    if constexpr (SCR) {
                        float B1w[1];
                        B1w[0] = dot_window32(((i & 1) ? eh1_addr - 4u : eh_addr) + 4u * (unsigned)i, (const_f32_ptr)ap->q32 + q_offset, L);
                        if (have)
                            screen_cells<1>(lead, scr_slot, c_base, i, 1, d, inv_d, dd, rule, overshoot, sum_q2, B1w, k, errB2, undecided,
                                            widths_c, regB, scr_env);
                        continue;
                    }
                    const double* e = e_base + i;
                    double B0 = 0, B1 = 0, A0 = 0, A1 = 0;
                    if constexpr (UNIFORM_W) {
                        for (int t0 = 0; t0 < Lx; t0 += kU) {
                            const const_f64_ptr qs = q + t0;
                            double taps[kU], x[kU];
#pragma unroll
                            for (int u = 0; u < kU; ++u) taps[u] = qs[u];   // requested before the samples
                            load_taps<true>(e + t0, x);
                            // one accumulator, taps in order: the same rounding as the kR-window form, so a
                            // cell has ONE value whether its row runs tiled or re-listed (pruning moves rows
                            // between the two); the other waves of the SIMD hide the FMA latency
#pragma unroll
                            for (int u = 0; u < kU; ++u) B0 = fma(taps[u], x[u], B0);
                        }
                        A0 = sum_q2;
                    } else {
                        const double* wv = w_base + i;
                        const const_f64_ptr q2 = q2_all + q_offset;
                        for (int t0 = 0; t0 < L; t0 += kU) {
                            double x[kU], z[kU];
                            load_taps<true>(e + t0, x);
                            load_taps<true>(wv + t0, z);
                            const const_f64_ptr qs = q + t0;
                            const const_f64_ptr ps = q2 + t0;
#pragma unroll
                            for (int u = 0; u < kU; ++u) { B0 = fma(qs[u], x[u], B0); A0 = fma(ps[u], z[u], A0); }
                        }
                    }
                    if (have) {
                        double x0, x1;
                        if (x_dot) { x0 = x_load<true>(x_tile, i); x1 = x_load<true>(x_tile, i + d); }
                        else { x0 = c_base[i]; x1 = c_base[i + d]; }
                        consider<UNIFORM_W, !RESIDENT>(lead, x0, x1, i, inv_d, dd, rule, overshoot, A0 + A1, B0 + B1, k, n_eval, undecided, widths_c, regB);
                    }
                }
                if constexpr (COUNTING) n_steps += (unsigned long long)(n_eval - evals_before) * (unsigned long long)L;
            }
        }
#if TLS_BODY_PRIO
        if constexpr (RESIDENT) __builtin_amdgcn_s_setprio(1);
#endif
        pc.mark(7);
        }  // position tiles
        if constexpr (BAND) {
            if (resolving) {
                // The noted windows, one wavefront each: decided by the reference's expression on X = k - numpy.cumsum (the
                // slab's X region, just written by the exact prefix pass); a window that passes is evaluated -- lanes over the
                // template taps, samples from the slab with the patch as an index mapping -- and meets lane 0's lead with
                // the window sum of the plain scan it was noted with (every cell of the period is valued on the same X).
                const int n_band = __builtin_amdgcn_readfirstlane(s_work[4]);
                for (int e0 = wave; e0 < n_band; e0 += nw) {
                    const int k = __builtin_amdgcn_readfirstlane(band_list[e0].k);
                    const int i = __builtin_amdgcn_readfirstlane(band_list[e0].i);
                    const double dX_noted = band_list[e0].dX;
                    const int d = widths_c[k].width, L = widths_c[k].q_len, q_offset = widths_c[k].q_offset;
                    const double overshoot = widths_c[k].overshoot, sum_q2 = widths_c[k].sum_q2, inv_d = widths_c[k].inv_d;
                    const double dd = (double)d;
                    const double dX_exact = regB[i + d] - regB[i];
                    if (!((1.0 - (dd - dX_exact) / dd) > dmin)) continue;   // core.py:58 on the reference's bits (uniform branch)
                    const double* qg = ap->q + q_offset;
                    [[maybe_unused]] const double* q2g = UNIFORM_W ? nullptr : ap->q2 + q_offset;
                    double Bs = 0.0, As = 0.0;
                    for (int tt = lane; tt < L; tt += kWave) {
                        const int pp = i + tt, src = pp < n ? pp : pp - n;
                        double ev = 1.0 - regA[src];
                        if constexpr (!UNIFORM_W) { const double ww = regW[src]; As = fma(q2g[tt], ww, As); ev *= ww; }
                        Bs = fma(qg[tt], ev, Bs);
                    }
#pragma unroll
                    for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                        Bs += __shfl_down(Bs, delta, kWave);
                        if constexpr (!UNIFORM_W) As += __shfl_down(As, delta, kWave);
                    }
                    if (lane == 0) {
                        bool und_none = false;
                        consider<UNIFORM_W, true, true>(lead, 0.0, dX_noted, i, inv_d, dd, rule, overshoot, UNIFORM_W ? sum_q2 : As, Bs, k, n_eval,
                                                        und_none, widths_c, regB);
                        if constexpr (COUNTING) n_steps += (unsigned long long)L;
                    }
                }
                if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[38], (unsigned long long)n_band);
            }
        }
        // fast mode: a window too close to transit_depth_min for the plain prefix sum to decide (depth_pass) sends the
        // whole period through exact mode; nothing of this attempt is written or counted
        // (an LDS flag, not __syncthreads_or: the library routine brings static LDS of its own, and the slab variant's
        // launches already ask for all 160 KB)
        if (undecided) s_work[flag_slot] = 1;
        if constexpr (SCR) {
            // (fp32 screen: every wavefront's smallest upper bound travels with the same barrier; wsum is idle here)
            double u_min = scr_slot.hi;
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) u_min = fmin(u_min, __shfl_down(u_min, delta, kWave));
            if (lane == 0) reinterpret_cast<double*>(wsum)[wave] = u_min;
        }
        wg_sync();
        const int any_undecided = __builtin_amdgcn_readfirstlane(s_work[flag_slot]);
        if (tid == 0) s_work[3 - flag_slot] = 0;   // the next attempt's flag: nobody touches it before several barriers from now
        flag_slot = 3 - flag_slot;
        if constexpr (BAND) {
            // noted band windows (and nothing that voids the attempt): the period goes through the exact prefix pass and the
            // resolution above; more of them than the list holds: a second search in exact mode, as without the list
            if (rule.band_count != nullptr && any_undecided == 0) {
                const int n_band = __builtin_amdgcn_readfirstlane(s_work[4]);
                if (n_band > 0) {
                    if (n_band <= kBandCap) {
                        resolve_band = true;
                        kept_lead = lead; kept_eval = n_eval; kept_steps = n_steps; kept_issued = n_issued;
                    }
                    if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[37], 1ull);
                    if (ap->n_curves > 1) { curve_exact = true; --curve; continue; }   // this curve's prefix pass; the others are not touched
                    retry_exact = true;
                    break;
                }
            }
        }
        if (any_undecided != 0 && !exact_mode) {
            if (ap->phase_cycles && tid == 0) atomicAdd(&ap->phase_cycles[37], 1ull);
            if (ap->n_curves > 1) { curve_exact = true; --curve; continue; }   // this curve again; the others are not touched
            retry_exact = true;   // one light curve: its permutation is gone (the prefix sum took its place) -- sort again
            break;
        }
        pc.mark(21);

        if constexpr (SCR) {
            // the smallest upper bound of the workgroup: a slot cell whose lower bound reaches it may be the period's
            // minimum and is valued now; every other cell of the period has been dropped above it or valued already
            double U = reinterpret_cast<const double*>(wsum)[0];
            for (int v = 1; v < nw; ++v) U = fmin(U, reinterpret_cast<const double*>(wsum)[v]);
            // the slot cells that reach it join the parked ones; then every wavefront takes its share of the list
            if (scr_slot.hi < INFINITY && scr_slot.lo <= U) park_cell(lead, scr_slot.lo, scr_slot.k, scr_slot.i, rule, widths_c, regB, scr_env);
            wg_sync();
            const int n_parked = (int)(scr_env.park->n < (unsigned int)kParkCap ? scr_env.park->n : (unsigned int)kParkCap);
            // (one wavefront values them, one after the other: a period has one or two that still reach U; the others wait
            // at phase 4's barrier.  The list is read 64 cells at a time, one per lane.)
            if (wave == 0) {
#pragma unroll 1
                for (int e0 = 0; e0 < n_parked; e0 += kWave) {
                    ParkedCell pcell;
                    pcell.lo = INFINITY; pcell.k = 0; pcell.i = 0;
                    if (e0 + lane < n_parked) pcell = scr_env.cells[e0 + lane];
                    unsigned long long reach = ballot64(pcell.lo <= U);
#pragma unroll 1
                    while (reach) {
                        const int src = __ffsll((long long)reach) - 1;
                        reach &= reach - 1ull;
                        screen_value_cell(lead, 0, lane_value(pcell.k, src), lane_value(pcell.i, src), rule, widths_c, regB, scr_env);
                        if (ap->phase_cycles && lane == 0) atomicAdd(&ap->phase_cycles[39], 1ull);
                    }
                }
            }
            pc.mark(19);
        }

        // ---- phase 4: argmin over the workgroup --------------------------------------
        Best best = settle_best<UNIFORM_W, !RESIDENT>(lead, widths_c, regB);
#pragma unroll
        for (int delta = kWave / 2; delta > 0; delta >>= 1) {
            Best o = shfl_down_best(best, delta);
            if (better(o, best)) best = o;
        }
        if (lane == 0) wbest[wave] = best;
        wg_sync();
        if (tid == 0) {
            Best g = wbest[0];
            for (int v = 1; v < nw; ++v) if (better(wbest[v], g)) g = wbest[v];
            bool write_out = true;
            [[maybe_unused]] bool reset_ready = false;
            if constexpr (ROLE == kRoleSearch) {
                reset_ready = true;
                // this tile's winner is published; whoever finishes the period's LAST tile compares the winners of all
                // of them (the comparison is the reference's total order on (value, width, T0): any order of arrival
                // gives the same cell).  Agent-scope atomics: the tiles of a period run on different XCDs.
                typedef unsigned long long u64;
                const unsigned int tiles_p = ap->tile_prefix[work + 1] - ap->tile_prefix[work];
                if (tiles_p > 1u) {
                    u64* mine = reinterpret_cast<u64*>(ap->partials) + 3LL * item;
                    __hip_atomic_store(mine + 0, (u64)__double_as_longlong(g.stat), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mine + 1, (u64)__double_as_longlong(g.td), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(mine + 2, ((u64)(unsigned int)g.k << 32) | (u64)(unsigned int)g.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned int* done = ap->tiles_done + (work - ap->batch_lo);
                    const unsigned int before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    write_out = before == tiles_p - 1u;
                    reset_ready = write_out;
                    if (write_out) {
                        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next batch
                        const u64* first = reinterpret_cast<const u64*>(ap->partials) + 3LL * ((long long)item - item_tile);
                        for (unsigned int j = 0; j < tiles_p; ++j) {
                            Best o;
                            o.stat = __longlong_as_double((long long)__hip_atomic_load(first + 3 * j + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            o.td = __longlong_as_double((long long)__hip_atomic_load(first + 3 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            const u64 ki = __hip_atomic_load(first + 3 * j + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            o.k = (int)(unsigned int)(ki >> 32); o.i = (int)(unsigned int)ki;
                            if (better(o, g)) g = o;
                        }
                    }
                }
            }
            if constexpr (ROLE == kRoleSearch) {
                // (every tile of the period is past its wait: the flag goes back to zero for the next batch or launch)
                if (reset_ready) __hip_atomic_store(ap->fold_ready + (work - ap->batch_lo), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (write_out) {
            const double datapoints = (double)n;          // core.py:46 baseline
            double chi2 = INFINITY, depth = 0.0;
            long long row = 0;
            if (n_rows > 0) {
                // uniform weights: A,B were accumulated without the common factor w0
                const double w0_c = ap->n_curves > 1 ? ap->curve_w0[curve] : ap->w0;
                const double S0_c = ap->n_curves > 1 ? ap->curve_S0[curve] : ap->S0;
                const double scale = UNIFORM_W ? w0_c : 1.0;
                const double stat = (g.stat < INFINITY) ? S0_c + scale * g.stat : INFINITY;
                if (stat < datapoints) {
                    chi2 = stat; row = ap->widths[g.k].row; depth = 1.0 - g.td;  // core.py:72-74
                } else {
                    // nothing beat the straight line: first in-range width registers with
                    // chi2 = N and depth 0 (core.py:46-48,183-186)
                    chi2 = datapoints; row = ap->widths[k_lo].row; depth = 0.0;
                }
            }
            const long long o = (long long)curve * ap->n_periods + p;
            ap->out_chi2[o] = chi2;
            ap->out_row[o] = row;
            ap->out_depth[o] = depth;
            }
        }
        if (COUNTING && ap->counters) {
#pragma unroll
            for (int delta = kWave / 2; delta > 0; delta >>= 1) {
                n_eval += __shfl_down(n_eval, delta, kWave);     // (a wave's cells of one period: far below 2^32)
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
        // (a retry re-enters the period loop with the same work item)
    }
