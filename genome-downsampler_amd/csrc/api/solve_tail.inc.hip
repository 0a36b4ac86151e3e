// solve_tail.inc.hip -- part of qmcp_api.hip (one translation unit).
// A solve's tail (waits for the read-back that picks the route, then queues the sweep and the keep mask), collection, context creation, the enqueue / complete pair.
int enqueue_tail(qmcp_hip_ctx* c) {
    SolveRun& run = c->run;
    const Problem& pr = run.pr;
    qmcp_hip_stats& local = run.local;
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot, n_contigs = run.n_contigs, M = run.M;
    const uint64_t n64 = run.n64;
    const uint32_t *d_starts = run.d_starts, *d_ends = run.d_ends;
    uint64_t* const d_mask = run.d_mask;
    const uint64_t* roff = run.roff;
    const uint32_t* lengths = run.lengths;
    if (run.trivial) {
        for (int i = EV_PREP; i <= EV_MARK; ++i) HIP_TRY(hipEventRecord(c->ev[i], c->stream));
        for (int i = 0; i < 8; ++i) c->h_scalars[i] = 0;
        c->pend_stats = local;
        c->pend_whole_contig_chains = 0;
        c->pending = true;
        return QMCP_OK;
    }
    const uint32_t range_shift = run.range_shift;
    const bool may_rank = run.may_rank;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    const uint32_t* hs = c->h_head;
    bool have_gstart = run.have_gstart;
    const bool ranked_counted = run.ranked_counted;
    if (may_rank) {
        HIP_TRY(hipStreamSynchronize(c->stream2));
        if (run.wait_empty) HIP_TRY(hipStreamSynchronize(c->stream));
        if (hs[2] != 0) {
            (void)hipStreamSynchronize(c->stream);  // what was queued stays in bounds; let it drain
            return fail(QMCP_EREAD, "a read has start > end or end >= its contig length");
        }
    }
    const uint32_t max_load = hs[3];
    const uint32_t empty_positions = hs[4];            // (the last solve of this shape's, or this one's: set by the head)
    const uint32_t min_span = hs[0], max_span = hs[1];
    local.min_span = min_span;
    local.max_span = max_span;
    const bool uniform = (min_span == max_span) && max_span <= qmcp::kMaxUniformSpan;
    if (!uniform && max_span > qmcp::kMaxGeneralSpan)
        return fail(QMCP_ERANGE, "mixed-span reads with span %u > %u are not supported by this build",
                    max_span, qmcp::kMaxGeneralSpan);
    local.path = uniform ? QMCP_PATH_UNIFORM : QMCP_PATH_GENERAL;
    if (uniform)
        for (uint32_t k = 0; k < n_contigs; ++k)
            if (roff[k + 1] - roff[k] >= (1ull << 28))
                return fail(QMCP_ERANGE, "contig %u holds %llu reads; the block sweep handles < 2^28 per contig",
                            k, (unsigned long long)(roff[k + 1] - roff[k]));

    // bucketing keys.  gstart (global start position per read) sits in vals[1].
    const uint32_t pos_bits = bit_width(ltot - 1) == 0 ? 1u : bit_width(ltot - 1);
    uint32_t span_bits = 0;
    bool wide = false;
    const uint32_t* d_gstart = (const uint32_t*)c->vals[1].p;
    const uint32_t* d_key32 = d_gstart;  // uniform span: the key is the start position itself
    auto need_gstart = [&]() {
        if (have_gstart) return;
        KernelSpan sp(c, "k_gstart");
        qmcp::launch_gstart(c->stream, d_starts, n, (const uint64_t*)c->roff.p, (const uint64_t*)c->poff.p,
                            n_contigs, (uint32_t*)c->vals[1].p);
        have_gstart = true;
    };
    bool sweep_done = false, ranked = false;
    uint32_t* d_iters = (uint32_t*)((char*)c->scalars.p + 16);
    bool near_done = false;
    uint32_t max_load_now = max_load;
    if (uniform && run.nu_filter != 0 && run.nu_filter != max_span) {
        // the head listed every read as an exception to the last call's span: its stages again, unfiltered
        c->nu_ell = 0;
        if (run.pm) TRY(queue_pm_head(c, c->stream, 0));
        else TRY(queue_rm_head(c, c->stream, 0, true));
        HIP_TRY(hipMemcpyAsync(c->h_nu, (uint32_t*)c->ranges.p + 65540, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        max_load_now = c->h_nu[0];
    }
    if (!uniform) c->mixed_seen = true;
    if (!uniform && max_span <= qmcp::kMaxUniformSpan) {
        HIP_TRY(hipMemsetAsync(c->scalars.p, 0, 64, c->stream));
        TRY(near_uniform_tail(c, min_span, max_span, max_load, d_iters, near_done));
        if (near_done) {
            local.path = QMCP_PATH_NEAR_UNIFORM;
            sweep_done = ranked = true;
        }
    }
    if (!uniform && !near_done) {
        need_gstart();
        span_bits = bit_width(max_span - min_span);
        wide = pos_bits + span_bits > 32;
        c->mixed_seen = true;
        TRY(ensure(c, c->ecnt, ((size_t)ltot + 1) * sizeof(uint32_t)));
        TRY(ensure(c, c->eoff, ((size_t)ltot + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->ecnt.p, 0, ((size_t)ltot + 1) * sizeof(uint32_t), c->stream));
        void* key_dst = wide ? c->keys[0].p : c->vals[0].p;
        {
            KernelSpan sp(c, "k_general_keys");
            qmcp::launch_general_keys(c->stream, wide, d_gstart, d_starts, d_ends, n, span_bits,
                                      max_span, nullptr, key_dst, (uint32_t*)c->ecnt.p, ltot + 1);
        }
        HIP_TRY(hipGetLastError());
        TRY(scan_counts(c, c->ecnt, c->eoff, ltot));
        d_key32 = (const uint32_t*)c->vals[0].p;
    }
    if (!near_done) HIP_TRY(hipEventRecord(c->ev[EV_SCAN], c->stream));
    // Uniform span, large call: neither the sweep nor the keep mask needs a full sort.  One stable
    // partition of {start, index} records by position range, per-range LDS counts, the sweep, and
    // a per-range ordered ranking against S(p) -- see "range-ranked uniform path" in the kernels.
    // The heaviest range's load is read back on the second stream while the partition runs; if
    // one range holds too much (its ranking is one wave's serial walk), the keep mask comes from
    // the radix sort instead (the counts and the sweep done here stay valid).
    bool mixed_whole_contigs = false;
    if (!near_done) HIP_TRY(hipMemsetAsync(c->scalars.p, 0, 64, c->stream));
    if (uniform && may_rank) {
        hipStream_t s1 = c->stream;  // (partition and bucket offsets are already queued)
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[EV_SORT], s1));
        ranked = (uint64_t)max_load_now * kRankBalance <= (uint64_t)n;
        if (c->opt.force_sort_route) ranked = false;  // test hook: force the sort
        // (the pass-major ranking can take its quotas from the event-driven sweep's own output: no expand, no selend[])
        bool expand_left_out = ranked && run.pm && !c->opt.keep_expand;
        TRY(launch_uniform_sweep(c, s1, n, ltot, n_contigs, max_span, M, d_iters, empty_positions, &expand_left_out));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], s1));
        sweep_done = true;
        if (ranked && run.pm) {
            queue_pm_rank(c, s1, expand_left_out ? (const uint32_t*)c->cstart.p : nullptr, (const uint32_t*)c->evlast.p, max_span);
            HIP_TRY(hipGetLastError());
        } else if (ranked) {
            KernelSpan sp(c, "k_rank_mark");
            qmcp::launch_rank_mark(s1, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->vals[0].p,
                                   d_range_start, range_shift, ltot,
                                   (const uint32_t*)c->boff.p, (const uint32_t*)c->selend.p,
                                   (unsigned long long*)d_mask, (unsigned long long*)c->scalars.p,
                                   c->rankamb.p, qmcp::rank_scratch_by_records(range_shift, ltot, n));
            HIP_TRY(hipGetLastError());
        }
    }

    // radix bucketing: stable LSD, 8-bit digits
    const uint32_t key_bits = pos_bits + span_bits;
    const uint32_t passes = (key_bits + 7) / 8;
    local.sort_passes = ranked ? 1u : passes;  // ranked path: one range partition, no sort
    const uint32_t n_tiles = qmcp::sort_tiles(n);
    int kin = 0, vin = 0;  // buffers holding the sorted output at the end
    if (!ranked && uniform) need_gstart();  // the sort-based routes bucket the bare keys
    if (ranked) {
        // keep mask already written by k_rank_mark
    } else if (!wide) {
        // records {key, read index}: keys[0] <-> keys[1]; the first pass reads bare keys
        const void* recs_in = nullptr;
        for (uint32_t p = 0; p < passes; ++p) {
            const bool first = p == 0;
            const int kout = first ? 0 : (kin ^ 1);
            {
                KernelSpan sp(c, "k_radix_hist_rec");
                qmcp::launch_radix_hist_rec(c->stream, first, d_key32, recs_in, n, 8 * p,
                                            (uint32_t*)c->hist.p);
            }
            {
                KernelSpan sp(c, "scan_radix_hist(3 kernels)");
                qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)c->hist.p, 256u * n_tiles,
                                            (uint32_t*)c->hist.p, (uint32_t*)c->spine.p, false);
            }
            {
                KernelSpan sp(c, "k_radix_scatter_rec");
                qmcp::launch_radix_scatter_rec(c->stream, first, d_key32, recs_in, n, 8 * p,
                                               (const uint32_t*)c->hist.p, c->keys[kout].p);
            }
            HIP_TRY(hipGetLastError());
            kin = kout;
            recs_in = c->keys[kin].p;
        }
    } else {
        // 64-bit composite keys (huge genome x wide span range): split key / payload arrays
        const uint32_t* vals_in = nullptr;
        for (uint32_t p = 0; p < passes; ++p) {
            const int kout = kin ^ 1, vout = (vals_in == nullptr) ? 0 : (vin ^ 1);
            {
                KernelSpan sp(c, "k_radix_hist");
                qmcp::launch_radix_hist(c->stream, true, c->keys[kin].p, n, 8 * p, (uint32_t*)c->hist.p);
            }
            {
                KernelSpan sp(c, "scan_radix_hist(3 kernels)");
                qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)c->hist.p, 256u * n_tiles,
                                            (uint32_t*)c->hist.p, (uint32_t*)c->spine.p, false);
            }
            {
                KernelSpan sp(c, "k_radix_scatter");
                qmcp::launch_radix_scatter(c->stream, true, c->keys[kin].p, vals_in, n, 8 * p,
                                           (const uint32_t*)c->hist.p, c->keys[kout].p,
                                           (uint32_t*)c->vals[vout].p);
            }
            HIP_TRY(hipGetLastError());
            kin = kout;
            vin = vout;
            vals_in = (const uint32_t*)c->vals[vin].p;
        }
    }
    // bucket offsets straight from the sorted keys (no atomics)
    if (!sweep_done) {
    HIP_TRY(hipMemsetAsync(c->boff.p, 0xFF, ((size_t)ltot + 1) * sizeof(uint32_t), c->stream));
    {
        KernelSpan sp(c, "k_bucket_heads");
        qmcp::launch_bucket_heads(c->stream, wide, c->keys[kin].p, (const uint32_t*)c->vals[vin].p, n,
                                  span_bits, ltot, (uint32_t*)c->boff.p);
    }
    {
        KernelSpan sp(c, "reverse_min_scan(3 kernels)");
        qmcp::launch_reverse_min_scan(c->stream, (uint32_t*)c->boff.p, ltot + 1, (uint32_t*)c->spine.p);
    }
    }
    HIP_TRY(hipGetLastError());
    if (!sweep_done) HIP_TRY(hipEventRecord(c->ev[EV_SORT], c->stream));

    // selection sweep
    if (sweep_done) {
        // done above, from the early counts
    } else if (uniform) {
        TRY(launch_uniform_sweep(c, c->stream, n, ltot, n_contigs, max_span, M, d_iters, empty_positions));
    } else {
        uint32_t ring = 64;
        while (ring <= max_span) ring <<= 1;
        // shallow or gapped data: stretches between cut points, one wave each (depth judged with the
        // longest span: an upper bound)
        const uint32_t* seg = nullptr;
        uint32_t n_seg_max = 0;
        const double depth = (double)n * (double)max_span / ((double)ltot * (double)(M ? M : 1));
        // (a mixed-span walk is one light workgroup per stretch and slow per position: five times the windows
        //  the one-span sweeps get, whose seven-wave workgroups fill the chip at three per compute unit)
        // (many times M and yet sparse -- a small M -- is shallow in standard deviations: launch_uniform_sweep)
        const bool sparse_deep = depth >= kGenDepth && (double)n / (double)ltot < std::log((double)max_span / 0.693) &&
                                 spec_sigma_depth(depth, M) < kGenDepth;
        const double depth_gate = sparse_deep ? spec_sigma_depth(depth, M) : depth;
        const uint32_t windows = sweep_cut_windows(c, ltot, max_span, n_contigs, depth_gate < kGenDepth, qmcp::kMaxSweepWindows);
        if (windows != 0) {
            KernelSpan sp(c, "k_find_cuts");
            seg = qmcp::launch_sweep_segments(c->stream, (const uint32_t*)c->boff.p, (const uint32_t*)c->eoff.p,
                                              (const uint64_t*)c->poff.p, n_contigs, ltot, max_span, M, windows,
                                              (uint32_t*)c->segs.p);
            n_seg_max = n_contigs + windows;
            // stats.sweep_stretches: the table's count (the uniform kernels count themselves)
            HIP_TRY(hipMemcpyAsync(d_iters + 2, seg, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        }
        mixed_whole_contigs = seg == nullptr;
        if (max_span <= qmcp::kMaxCachedSpan) {
            ring = 64;
            while (ring < max_span + 64) ring <<= 1;  // 64 buckets enter per chunk
            // run lengths of equal (start, end) groups: heads + reverse min-scan -> next_head[]
            TRY(ensure(c, c->next_head, ((size_t)n + 2) * sizeof(uint32_t)));
            TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(n + 1) + 1) * sizeof(uint32_t) + 16));
            {
                KernelSpan sp(c, "k_group_heads");
                qmcp::launch_group_heads(c->stream, wide, c->keys[kin].p, n, (uint32_t*)c->next_head.p);
            }
            {
                KernelSpan sp(c, "reverse_min_scan(3 kernels)");
                qmcp::launch_reverse_min_scan(c->stream, (uint32_t*)c->next_head.p, n + 1,
                                              (uint32_t*)c->spine.p);
            }
            // spans up to 448: the window of live buckets fits the wave's registers (8 per lane)
            const bool in_regs = max_span + 64 <= 512 && !c->opt.mixed_sweep_in_lds;
            // speculative stretch boundaries, as for one span (launch_uniform_sweep): the state is the
            // selected reads still alive, i.e. the kept counts of the last max_span start positions, which
            // k_spec_verify compares (selend = bucket start + kept count); the run-in is counted in
            // windows of max_span positions
            // (the first tier starts lower than for one span: a walk is slow per position, so short stretches
            //  matter more, and the second tier is there)
            uint32_t burn_blocks = c->opt.speculation_run_in ? spec_first_run_in(c, depth) : spec_first_run_in(c, depth) * 3u / 5u;
            bool hopeless = c->spec_hopeless_n == n64 && c->spec_hopeless_ltot == pr.ltot && c->spec_hopeless_M == M;
            if (!hopeless && spec_wanted(c, depth_gate, spec_depth_in_sigma(depth_gate, M)) && in_regs && seg != nullptr && n >= (1u << 20)) {
                // One dominant read length (what is left for this route once the shorter reads have their own: a few
                // LONGER ones) forgets its state as slowly as one-length data, and the walk's boundaries then disagree
                // nearly everywhere (lab/mixed_spec_check.py: 430 against 185 ms at 7.5 x M); a broad mix of lengths
                // forgets fast and gains (lab/mixed_spec_broad.py: 117 against 271 ms at 5 x M).  A sample of the spans
                // tells the two apart before anything is queued.
                uint32_t* d_share = (uint32_t*)c->stats.p + 6;
                qmcp::launch_span_mode_share(c->stream, d_starts, d_ends, n, d_share);
                uint32_t share[3] = {0, 0, 0};
                HIP_TRY(hipMemcpyAsync(share, d_share, sizeof(share), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                if (share[0] != 0 && (uint64_t)share[1] * 10u >= (uint64_t)share[0] * 9u) {
                    // Second half of round 4: that finding was about how DEEP the data is in standard deviations, not about
                    // the one length.  The run-in table (spec_burn_blocks) was measured at M = 50; what makes a sweep forget
                    // is how often the coverage comes near M, i.e. z = (mean coverage - M) / sqrt(mean coverage) =
                    // sqrt(M) (d - 1) / sqrt(d) for Poisson starts -- the lab's reads at 2.1 x M with M = 350 are as deep as
                    // M = 50 at 6 x M, and their boundaries disagreed at the run-in of 2.1 x M.  So: the depth at which
                    // M = 50 has the same z; below 3.1 of it the walk speculates with that depth's whole run-in (not three
                    // fifths: a boundary that disagrees costs its exact stretch again, and on such data exact stretches are
                    // long), deeper it does not.  One GPU's real share of configs[4] (117.7 M positions in its longest
                    // contig, 2 x M, M = 50) with 1 % clipped reads, this route: 14.5 s as one chain per contig, 52 ms in
                    // 1 925 stretches, no boundary disagreeing (lab/cfg5_share_mixed_spec.py) -- which is what a whole-genome
                    // BAM with reads LONGER than the dominant length (deletions) gets, since those leave the near-uniform route.
                    // (the call's depth is counted with the LONGEST span; nine tenths of the reads have this one)
                    const double depth_mode = share[2] >= 1 && share[2] < 511 && share[2] < max_span
                                                  ? depth * (double)share[2] / (double)max_span : depth;
                    const double depth_eff = sparse_deep ? spec_sigma_depth(depth_mode, M)     // (a small M: shallower than its depth)
                                                         : spec_depth_in_sigma(depth_mode, M);  // (uniform_sweep.inc.hip; >= its argument)
                    const double d_run = depth_eff;
                    hopeless = !(depth_eff < 3.1);
                    if (hopeless && depth_eff < kSpecDepth) {
                        // deeper than that (run-ins of 1 536 blocks and more) only where the longest contig holds a dozen
                        // run-ins: a 10^6-position contig would become two stretches, a chromosome becomes hundreds
                        uint32_t longest = 0;
                        for (uint32_t k = 0; k < n_contigs; ++k) longest = lengths[k] > longest ? lengths[k] : longest;
                        hopeless = (uint64_t)longest < 12ull * spec_first_run_in(c, d_run) * max_span;
                    }
                    if (!hopeless) burn_blocks = spec_first_run_in(c, sparse_deep ? depth : d_run) * ((sparse_deep && !c->opt.speculation_run_in) ? 3u : 1u);  // (launch_uniform_sweep)
                }
            }
            if (c->opt.speculation != 0 || c->opt.speculation_run_in != 0) hopeless = false;
            const bool speculate = !hopeless && spec_wanted(c, depth_gate, spec_depth_in_sigma(depth_gate, M)) && in_regs && seg != nullptr && burn_blocks >= 2 &&
                                   (uint64_t)ltot >= 4ull * burn_blocks * max_span;
            if (speculate) {
                TRY(ensure(c, c->specsnap, qmcp::spec_snap_bytes(n_seg_max)));
                const void* sorted = c->keys[kin].p;
                // (a walk is one light workgroup: many short stretches beat few long ones -- two run-ins apart)
                TRY(speculative_sweep(
                    c, c->stream, n_contigs, ltot, windows, max_span, 64, burn_blocks, 2, seg, "k_sweep_general_reg",
                    [&](const uint32_t* table, uint32_t* out_odd, const uint32_t* redo_in) {
                        return qmcp::launch_sweep_general_reg(c->stream, wide, (const uint32_t*)c->boff.p, (const uint32_t*)c->eoff.p,
                                                              sorted, (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                              n_contigs, span_bits, max_span, M, (uint32_t*)c->selend.p, table,
                                                              n_seg_max, out_odd, redo_in, (uint32_t*)c->specsnap.p);
                    },
                    [&](const uint32_t* table, uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
                        qmcp::launch_spec_verify_merge_mixed(c->stream, table, n_seg_max, max_span, (uint32_t*)c->selend.p,
                                                             (const uint32_t*)c->cstart.p, (const uint32_t*)c->specsnap.p,
                                                             mismatches, redo_in, redo_out);
                    }));
                // stats.sweep_stretches: the first tier's table
                HIP_TRY(hipMemcpyAsync(d_iters + 2, (uint32_t*)c->segs.p + windows + (1 + 5 * (size_t)n_seg_max), sizeof(uint32_t),
                                       hipMemcpyDeviceToDevice, c->stream));
            }
            if (!speculate) {  // (else: swept above)
                KernelSpan sp(c, in_regs ? "k_sweep_general_reg" : "k_sweep_general_cached");
                if (!in_regs ||
                    !qmcp::launch_sweep_general_reg(c->stream, wide, (const uint32_t*)c->boff.p,
                                                    (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                                    (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                    n_contigs, span_bits, max_span, M, (uint32_t*)c->selend.p, seg,
                                                    n_seg_max))
                    qmcp::launch_sweep_general_cached(c->stream, wide, (const uint32_t*)c->boff.p,
                                                      (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                                      (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                      n_contigs, span_bits, max_span, M,
                                                      (uint32_t*)c->selend.p, ring, seg, n_seg_max);
            }
        } else {
            uint32_t* g_rings = nullptr;
            if (max_span > qmcp::kMaxLdsRingSpan) {
                // long reads: the two rings of a workgroup no longer fit LDS
                const size_t n_wg = seg ? n_seg_max : n_contigs;
                TRY(ensure(c, c->rings, n_wg * 2 * (size_t)ring * sizeof(uint32_t)));
                g_rings = (uint32_t*)c->rings.p;
            }
            KernelSpan sp(c, "k_sweep_general");
            qmcp::launch_sweep_general(c->stream, wide, (const uint32_t*)c->boff.p,
                                       (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                       (const uint64_t*)c->poff.p, n_contigs, span_bits, max_span, M,
                                       (uint32_t*)c->selend.p, ring, seg, n_seg_max, g_rings);
        }
    }
    HIP_TRY(hipGetLastError());
    if (!sweep_done) HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], c->stream));

    // keep mask
    if (!ranked) {
        KernelSpan sp(c, "k_mark");
        qmcp::launch_mark(c->stream, wide, c->keys[kin].p, (const uint32_t*)c->vals[vin].p, ltot,
                          (const uint32_t*)c->boff.p, (const uint32_t*)c->selend.p, d_mask,
                          (unsigned long long*)c->scalars.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_MARK], c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_scalars, c->scalars.p, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));
    c->pend_spiky = ranked_counted;
    if (ranked_counted) {
        HIP_TRY(hipMemcpyAsync(c->h_scalars + 7, (uint32_t*)c->stats.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        c->spiky_n = n64;
        c->spiky_ltot = pr.ltot;
    }
    c->pend_stats = local;
    c->pend_M = M;
    c->pend_whole_contig_chains = 0;
    if (mixed_whole_contigs)  // one wave per non-empty contig
        for (uint32_t k = 0; k < n_contigs; ++k) c->pend_whole_contig_chains += lengths[k] != 0 ? 1u : 0u;
    c->pending = true;
    return QMCP_OK;
}

int solve_enqueue(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends, const uint64_t* roff,
                  const uint32_t* lengths, uint32_t n_contigs, uint64_t n64, uint32_t M, uint64_t* d_mask);
int collect_one(qmcp_hip_ctx* c, qmcp_hip_stats* st) {
    if (!c->pending) return fail(QMCP_EINVAL, "no solve is pending on this context");
    c->pending = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->nu_deferred) {
        // the near-uniform route queued its rounds and the ranking unseen (near_uniform_tail): did they settle?
        c->nu_deferred = false;
        if (c->h_nu[2] != 0 || c->h_nu[1] != 0) {
            // no: the call again, the blocking way (the head clears the mask; the contig tables are the context's copies)
            c->nu_need_rounds = 0;
            const SolveRun r = c->run;
            const size_t count = (size_t)r.n_contigs + 1;
            std::vector<uint64_t> roff(c->h_tables, c->h_tables + count);
            std::vector<uint32_t> lengths(r.n_contigs);
            for (uint32_t k = 0; k < r.n_contigs; ++k) lengths[k] = (uint32_t)(c->h_tables[count + k + 1] - c->h_tables[count + k]);
            TRY(solve_enqueue(c, r.d_starts, r.d_ends, roff.data(), lengths.data(), r.n_contigs, r.n64, r.M, r.d_mask));
            return collect_one(c, st);
        }
        c->pend_stats.near_uniform_rounds = c->h_nu[6] + 1;
        c->pend_stats.near_uniform_selected = c->h_nu[3];
    }
    collect_spans(c);
#ifdef QMCP_EV_STAMP
    if (c->scalars.p) {
        uint32_t raw[16];
        HIP_TRY(hipMemcpy(raw, c->scalars.p, sizeof(raw), hipMemcpyDeviceToHost));
        const uint32_t* it = raw + 4;
        fprintf(stderr, "[ev stamp] changed %u of %u blocks, stretches %u | x16 cycles: total %u general %u wait-for-ring %u slow-pieces %u (%u pieces) "
                        "failing rounds %u (%u rounds), blocks through the two scans %u\n",
                it[0], it[1], it[2], it[4], it[5], it[6], it[7], it[8], it[9], it[10], it[11]);
    }
#endif
    qmcp_hip_stats local = c->pend_stats;
    const unsigned long long* host_scalars = c->h_scalars;
    if (c->pend_spiky) {
        c->spiky_empty = (uint32_t)(c->h_scalars[7] & 0xFFFFFFFFull);
        c->spiky_known = true;
    }
    local.n_kept = host_scalars[0];
    c->last_iters = (uint32_t)(host_scalars[2] & 0xFFFFFFFFu);
    c->last_blocks = (uint32_t)(host_scalars[2] >> 32);
    local.sweep_blocks_changed = c->last_iters;
    local.sweep_blocks = c->last_blocks;
    local.sweep_stretches = (uint32_t)(host_scalars[3] & 0xFFFFFFFFu) + c->pend_whole_contig_chains;
    local.spec_mismatches = (uint32_t)(host_scalars[4] & 0xFFFFFFFFu);
    local.spec_boundaries = (uint32_t)(host_scalars[4] >> 32);
    local.spec_retry_mismatches = local.spec_mismatches ? (uint32_t)(host_scalars[5] & 0xFFFFFFFFu) : 0u;
    if (local.path == QMCP_PATH_GENERAL && local.spec_boundaries >= 4 && 2u * local.spec_mismatches > local.spec_boundaries &&
        c->opt.speculation_run_in == 0) {
        // (three sweeps -- both tiers and the exact one -- where one would have done: 430 against 185 ms on cfg4's reads
        //  with 1 % clipped at 7.5 x M, lab/mixed_spec_check.py)
        c->spec_hopeless_n = local.n_reads; c->spec_hopeless_ltot = local.total_length; c->spec_hopeless_M = c->pend_M;
    }
    local.ms_prepare = elapsed(c->ev[EV_BEGIN], c->ev[EV_PREP]);
    local.ms_scan = elapsed(c->ev[EV_PREP], c->ev[EV_SCAN]);
    local.ms_sort = elapsed(c->ev[EV_SCAN], c->ev[EV_SORT]);
    local.ms_sweep = elapsed(c->ev[EV_SORT], c->ev[EV_SWEEP]);
    local.ms_mark = elapsed(c->ev[EV_SWEEP], c->ev[EV_MARK]);
    local.ms_total = elapsed(c->ev[EV_BEGIN], c->ev[EV_MARK]);
    local.arena_grown_mid_solve = c->grew_mid_solve;
    if (st) *st = local;
    return QMCP_OK;
}

int solve_complete(qmcp_hip_ctx* c, qmcp_hip_stats* st) { return collect_one(c, st); }

// The environment's debug overrides of a new context's options -- the only place the library reads the environment.
void options_from_env(qmcp_hip_options& o) {
    auto num = [](const char* name) -> uint32_t { const char* e = std::getenv(name); return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 0u; };
    auto flag = [](const char* name) -> int32_t { return std::getenv(name) != nullptr ? 1 : 0; };
    auto tri = [](const char* name) -> int32_t { const char* e = std::getenv(name); return !e ? 0 : e[0] == '1' ? 1 : e[0] == '0' ? -1 : 0; };
    o.pass_major = tri("QMCP_HIP_PM");
    if (const char* e = std::getenv("QMCP_HIP_SWEEP"))
        o.sweep = std::strcmp(e, "fast") == 0 ? QMCP_SWEEP_FAST : std::strcmp(e, "gen") == 0 ? QMCP_SWEEP_GENERAL
                  : std::strcmp(e, "ev") == 0 ? QMCP_SWEEP_EVENTS : QMCP_SWEEP_AUTO;
    o.cut_points = tri("QMCP_HIP_CUTS");
    o.speculation = tri("QMCP_HIP_SPEC");
    o.speculation_run_in = num("QMCP_HIP_SPEC_BURN");
    o.near_uniform = tri("QMCP_HIP_NEAR") < 0 ? -1 : 0;
    o.near_uniform_rounds = num("QMCP_HIP_NEAR_ROUNDS");
    if (const char* e = std::getenv("QMCP_HIP_NEAR_MIN_DEPTH")) o.near_uniform_min_depth = (float)std::strtod(e, nullptr);
    o.near_uniform_debug = flag("QMCP_HIP_NEAR_DEBUG");
    o.force_sort_route = flag("QMCP_HIP_NO_RANK");
    o.keep_expand = flag("QMCP_HIP_EXPAND");
    o.mixed_sweep_in_lds = flag("QMCP_HIP_GENERAL_LDS");
    o.rank_min_reads = num("QMCP_HIP_RANK_MIN");
    o.host_threads = num("QMCP_HIP_HOST_THREADS");
    o.copy_streams = num("QMCP_HIP_COPY_STREAMS");
    o.host_both_columns = flag("QMCP_HIP_HOST_BOTH_COLUMNS");
}

int create_ctx(int device, qmcp_hip_ctx** out_ctx) {
    if (!out_ctx) return fail(QMCP_EINVAL, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(QMCP_ENODEVICE, "no HIP device available (quasi-mcp-hip has no CPU fallback)");
    }
    if (device < 0 || device >= n) return fail(QMCP_ENODEVICE, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    qmcp_hip_ctx* c = new (std::nothrow) qmcp_hip_ctx();
    if (!c) return fail(QMCP_ENOMEM, "host allocation failed");
    c->device = device;
    qmcp_hip_default_options(&c->opt);
    options_from_env(c->opt);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < EV_COUNT; ++i) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;  // numerically lower == higher priority
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_head, 8 * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        qmcp_hip_destroy(c);
        return fail(QMCP_EHIP, "context setup: %s", hipGetErrorString(e));
    }
    *out_ctx = c;
    return QMCP_OK;
}


// Everything of a solve up to and including its last launch; nothing here waits for the device
// except the small read-back that picks the route (span statistics, heaviest range), and that
// wait leaves the device free to work on whatever else is queued.  solve_complete collects.
int solve_enqueue(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                  const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                  uint32_t M, uint64_t* d_mask) {
    if (c->pending) return fail(QMCP_EINVAL, "a solve is already pending on this context (call qmcp_hip_solve_end)");
    TRY(enqueue_head(c, d_starts, d_ends, roff, lengths, n_contigs, n64, M, d_mask));
    return enqueue_tail(c);
}

int solve_on_device(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                    const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                    uint32_t M, uint64_t* d_mask, qmcp_hip_stats* st) {
    TRY(solve_enqueue(c, d_starts, d_ends, roff, lengths, n_contigs, n64, M, d_mask));
    return solve_complete(c, st);
}

int use_device(qmcp_hip_ctx* c, bool may_be_pending = false) {
    if (!c) return fail(QMCP_EINVAL, "null context");
    if (c->pending && !may_be_pending)
        return fail(QMCP_EINVAL, "a solve is pending on this context (call qmcp_hip_solve_end first)");
    HIP_TRY(hipSetDevice(c->device));
    return QMCP_OK;
}

// Order the solver stream after the caller's producer stream (NULL = the default stream:
// the solver stream is non-blocking, so even that needs an explicit edge).
int order_after(qmcp_hip_ctx* c, void* user_stream) {
    if ((hipStream_t)user_stream != c->stream) {
        HIP_TRY(hipEventRecord(c->ev_in, (hipStream_t)user_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    }
    return QMCP_OK;
}
