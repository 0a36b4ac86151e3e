// solve_head.inc.hip -- part of qmcp_api.hip (one translation unit).
// A solve's head: everything that depends only on the reads' start positions -- arena sizing, the producers of the range-ranked route (pass-major or range-major form), bucket offsets -- and the pass-major ranking.
int queue_rm_head(qmcp_hip_ctx* c, hipStream_t s1, uint32_t filter, bool clear_mask);
int queue_pm_head(qmcp_hip_ctx* c, hipStream_t st, uint32_t filter);

// The pass-major form of the range-ranked route (kernels/pass_major.inc.hip) pads every (range, pass) slice to whole
// groups of 64 slots: it pays where slices are long -- a pass's 8 192 reads over the ranges its contig spans --, and
// where they would be short (narrow ranges: small genomes) the padding is most of a group and the range-major form is
// kept.  Hard limits: one partition level, slots addressable with 32-bit byte offsets.
bool pm_route_ok(const qmcp_hip_ctx* c, const uint64_t* roff, const Problem& pr, uint32_t shift) {
    if (c->opt.pass_major < 0) return false;  // (A/B: the range-major form)
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot;
    if ((uint64_t)qmcp::pm_slots(n, ltot, shift) >= (1ull << 31)) return false;
    if (c->opt.pass_major > 0) return true;   // (tests: the form on small inputs)
    // expected wave-slots against the records' own 1 / 64: a contig's pass deals its reads to the ranges the contig spans
    double slots = 0.0;
    for (uint32_t k = 0; k < pr.n_contigs; ++k) {
        const uint64_t reads = roff[k + 1] - roff[k];
        if (reads == 0 || pr.poff[k + 1] == pr.poff[k]) continue;
        const double ranges = (double)(((pr.poff[k + 1] - 1) >> shift) - (pr.poff[k] >> shift) + 1);
        const double passes = (double)reads / (double)qmcp::pm_pass() < 1.0 ? 1.0 : (double)reads / (double)qmcp::pm_pass();
        const double slice = (double)reads / (passes * ranges);
        slots += passes * ranges * std::ceil(slice / 64.0);
    }
    return slots <= 1.3 * ((double)n / 64.0);
}
// ---------------------------------------------------------------------------------------------------
// One solve = enqueue_head (everything that depends only on the reads' start positions: prepare, the
// range partition and the bucket offsets; nothing in it waits for the device on large calls) +
// enqueue_tail (waits for the 16-byte read-back that picks the route, then queues the sweep and the
// keep mask) + solve_complete (collects).  SolveRun is what the two enqueue halves share.
int enqueue_head(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                 const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                 uint32_t M, uint64_t* d_mask) {
    if (c->pending) return fail(QMCP_EINVAL, "a solve is already pending on this context (call qmcp_hip_solve_end)");
    c->pend_spiky = false;
    if (!c->h_scalars) HIP_TRY(hipHostMalloc((void**)&c->h_scalars, 16 * sizeof(unsigned long long), hipHostMallocDefault));
    SolveRun& run = c->run;
    run = SolveRun();
    run.d_starts = d_starts; run.d_ends = d_ends; run.roff = roff; run.lengths = lengths;
    run.n_contigs = n_contigs; run.n64 = n64; run.M = M; run.d_mask = d_mask;
    Problem& pr = run.pr;
    TRY(check_problem(roff, lengths, n_contigs, n64, pr));
    const uint32_t n = (uint32_t)pr.n;
    const uint32_t ltot = (uint32_t)pr.ltot;
    const size_t mask_words = (size_t)((n64 + 63) / 64);
    qmcp_hip_stats& local = run.local;
    std::memset(&local, 0, sizeof(local));
    local.n_reads = n64;
    local.n_contigs = n_contigs;
    local.total_length = pr.ltot;
    if (n == 0 || ltot == 0) {
        if (mask_words) HIP_TRY(hipMemsetAsync(d_mask, 0, mask_words * sizeof(uint64_t), c->stream));
        if (n != 0) return fail(QMCP_EREAD, "reads given for zero-length contigs");
        run.trivial = true;
        HIP_TRY(hipEventRecord(c->ev[EV_BEGIN], c->stream));
        return QMCP_OK;
    }
    // size the whole arena before anything is enqueued (growing a buffer frees it, and
    // hipFree would stall on the work in flight)
    c->sized = false;
    {
        const uint32_t tiles_seg = qmcp::seg_tile_bound(n);  // second partition level: tiles aligned to super-ranges
        const uint32_t spine_a = qmcp::scan_spine_entries(256u * tiles_seg);
        const uint32_t spine_b = qmcp::scan_spine_entries(ltot + 1) + 1;
        TRY(ensure(c, c->spine, (size_t)(spine_a > spine_b ? spine_a : spine_b) * sizeof(uint32_t) + 16));
        TRY(ensure(c, c->hist, (size_t)256 * tiles_seg * sizeof(uint32_t)));
        // (the pass-major form's two 16-bit record streams live in keys[0] and keys[1]: padded slices, ~6 B per read)
        const bool may_pm = n >= rank_min_reads(c) && qmcp::range_path_supported(ltot) && !qmcp::range_path_two_level(ltot) &&
                            pm_route_ok(c, roff, pr, qmcp::range_shift_for(ltot));
        const size_t pm_bytes = may_pm ? qmcp::pm_slots(n, ltot, qmcp::range_shift_for(ltot)) * sizeof(uint16_t) : 0;
        TRY(ensure(c, c->keys[0], std::max((size_t)n * sizeof(uint64_t), pm_bytes)));
        TRY(ensure(c, c->keys[1], std::max((size_t)n * sizeof(uint64_t), pm_bytes)));
        if (may_pm) {
            const size_t groups = pm_bytes / (64 * sizeof(uint16_t));
            TRY(ensure(c, c->pm_desc, groups * sizeof(uint32_t)));
            TRY(ensure(c, c->pm_work, 1024 * sizeof(uint32_t)));
        }
        TRY(ensure(c, c->vals[0], (size_t)n * sizeof(uint32_t)));
        TRY(ensure(c, c->vals[1], (size_t)n * sizeof(uint32_t)));
        TRY(ensure(c, c->spine2, (size_t)(spine_a > spine_b ? spine_a : spine_b) * sizeof(uint32_t) + 16));
        TRY(ensure(c, c->hist2, ((size_t)256 * qmcp::part_pass_pitch(n) + 4) * sizeof(uint32_t)));  // (+ the scan's total)
        TRY(ensure(c, c->cstart, ((size_t)ltot + 8) * sizeof(uint32_t)));  // also the event sweep's changed-block S
        TRY(ensure(c, c->boff, ((size_t)ltot + 1) * sizeof(uint32_t)));
        TRY(ensure(c, c->selend, ((size_t)ltot + 8) * sizeof(uint32_t)));  // + spare words for idle lanes
        TRY(ensure(c, c->scalars, 64));
        TRY(ensure(c, c->segs, qmcp::sweep_segment_words(n_contigs < 256 ? n_contigs : 0, qmcp::kMaxSweepWindows) * sizeof(uint32_t)));
        TRY(ensure(c, c->specflags, 2 * 4096 * sizeof(uint32_t)));           // speculative sweeps: marks per exact stretch, two tiers
        TRY(ensure(c, c->specsnap, qmcp::spec_snap_bytes(4096)));            // ... and the mixed-span walk's states at boundaries
        TRY(ensure(c, c->ranges, (65537 + 7 + 771 + 5) * sizeof(uint32_t)));  // range starts, heaviest load, level-2 tables
        if (n >= rank_min_reads(c) && qmcp::range_path_supported(ltot))
            TRY(ensure(c, c->rankamb, qmcp::rank_scratch_bytes(qmcp::range_shift_for(ltot), ltot, n)));
        TRY(ensure(c, c->stats, 12 * sizeof(uint32_t)));  // (words 6..8: the mixed-span route's sample of the spans)
        // (the near-uniform route's buffers: a context that has met mixed spans may look at the route on any call)
        if (c->nu_ell != 0 || c->mixed_seen) TRY(ensure_near_uniform(c, n, ltot, n_contigs));
        // The mixed-span route's own arrays.  Which route a call takes is known only after its first kernel,
        // so a context that has taken the mixed route once sizes them for every later call up front: growing
        // them after the partition has been queued would stall on it (ensure() waits for the streams).
        if (c->mixed_seen || !(n >= rank_min_reads(c) && qmcp::range_path_supported(ltot))) {
            TRY(ensure(c, c->ecnt, ((size_t)ltot + 1) * sizeof(uint32_t)));
            TRY(ensure(c, c->eoff, ((size_t)ltot + 1) * sizeof(uint32_t)));
            TRY(ensure(c, c->next_head, ((size_t)n + 2) * sizeof(uint32_t)));
            TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(n + 1) + 1) * sizeof(uint32_t) + 16));
        }
    }
    c->grew_mid_solve = 0;
    HIP_TRY(hipEventRecord(c->ev[EV_BEGIN], c->stream));
    TRY(upload_tables(c, roff, pr));
    c->sized = true;

    uint32_t* const hs = c->h_head;  // pinned: [0..2] span min / max / error flag, [3] heaviest range, [4] empty positions
    // the range partition's per-tile histogram is produced by the same pass when the range-ranked
    // path can be taken (uniformity is only known afterwards; the table is cheap)
    run.range_shift = qmcp::range_shift_for(ltot);
    run.may_rank = n >= rank_min_reads(c) && qmcp::range_path_supported(ltot);
    const uint32_t range_shift = run.range_shift;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    uint32_t* d_seg_tables = d_range_start + 65544;  // super_start, tile_base, pass_base (257 each)
    run.two_level = qmcp::range_path_two_level(ltot);
    const bool two_level = run.two_level;
    hs[3] = 0;
    hs[4] = 0xFFFFFFFFu;  // unknown unless the range-ranked route counted them
    run.have_gstart = true;
    if (!run.may_rank) {
        HIP_TRY(hipMemsetAsync(d_mask, 0, mask_words * sizeof(uint64_t), c->stream));
        uint32_t hs3[3];
        TRY(run_prepare(c, d_starts, d_ends, pr, nullptr, true, false, false, range_shift, nullptr, hs3));
        hs[0] = hs3[0]; hs[1] = hs3[1]; hs[2] = hs3[2];
        HIP_TRY(hipEventRecord(c->ev[EV_PREP], c->stream));
    } else {
        // Large call that can take the range-ranked route if its spans turn out uniform.  The host
        // needs the span statistics before it can pick the sweep, but the device need not idle for
        // that round trip: the partition and the bucket offsets depend only on the start positions,
        // so they are queued behind k_prepare at once and the statistics (and the heaviest range's
        // load) are fetched on the side stream meanwhile.  k_prepare does not write the global
        // start positions on this route -- the partition rebuilds them from the starts.
        run.have_gstart = false;
        hipStream_t s1 = c->stream;
        static const uint32_t init[8] = {0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        HIP_TRY(hipMemcpyAsync(c->stats.p, init, sizeof(init), hipMemcpyHostToDevice, s1));
        run.pm = !two_level && pm_route_ok(c, roff, pr, range_shift);
        run.nu_filter = c->nu_ell;
        hs[5] = hs[6] = 0;
        if (run.pm) {
            // One pass over the reads: validate, statistics, mask clear, and every pass of 8 192 reads sorted by range
            // (4 B per read out, two [range][pass] tables); a scan of the padded count table gives the padded flat
            // coordinates, one more small kernel the wave-slot descriptors the per-range kernels follow.  No range-major
            // copy, no second read of the starts.
            TRY(queue_pm_head(c, s1, run.nu_filter));
        } else {
            // the range-major form: k_prepare, scan, partition (one or two levels), bucket offsets
            TRY(queue_rm_head(c, s1, run.nu_filter, true));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_head, s1));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        HIP_TRY(hipMemcpyAsync(hs, c->stats.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        HIP_TRY(hipMemcpyAsync(hs + 3, d_max_load, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        if (run.nu_filter) HIP_TRY(hipMemcpyAsync(hs + 5, (uint32_t*)c->stats.p + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        // How spiky the starts are only decides WHICH exact sweep kernel runs, so the count of the previous
        // call of this shape is good enough (and saves waiting for k_range_offsets); a first call waits.
        if (c->spiky_known && c->spiky_n == n64 && c->spiky_ltot == pr.ltot) {
            hs[4] = c->spiky_empty;
        } else {
            HIP_TRY(hipMemcpyAsync(hs + 4, (uint32_t*)c->stats.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s1));
            run.wait_empty = true;
        }
        run.ranked_counted = true;
    }
    run.head_done = true;
    return QMCP_OK;
}

// The range-major head's stages on `st`: k_prepare (span statistics, partition table; regular reads: span == filter, or
// every read when filter == 0), scan, the partition (one level, or two for genomes beyond 8.39 M positions), bucket
// offsets.  Used by enqueue_head and, for a call whose head ran with the wrong idea of the spans, again by the tail.
int queue_rm_head(qmcp_hip_ctx* c, hipStream_t s1, uint32_t filter, bool clear_mask) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot, n_contigs = run.n_contigs;
    const uint32_t range_shift = run.range_shift;
    const bool two_level = run.two_level;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    uint32_t* d_seg_tables = d_range_start + 65544;  // super_start, tile_base, pass_base (257 each)
    // (a re-run of the head -- after a span change, or with the filter switched on -- must not add to what the first
    //  run counted: empty positions, exceptions, list flag, overflow entries)
    static const uint32_t zeros[4] = {0u, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync((uint32_t*)c->stats.p + 3, zeros, sizeof(zeros), hipMemcpyHostToDevice, s1));
    uint32_t* exc = filter ? (uint32_t*)c->nu_exc.p : nullptr;
    const uint32_t cap = nu_cap_for(n);
    uint32_t* exc_cnt = filter ? qmcp::nu_exc_counts(exc, cap) : nullptr;
    if (filter) HIP_TRY(hipMemsetAsync(exc_cnt, 0, ((size_t)cap / 128 + 4) * sizeof(uint32_t), s1));
    {
        KernelSpan sp(c, "k_prepare");
        qmcp::launch_prepare(s1, run.d_starts, run.d_ends, n, (const uint64_t*)c->roff.p,
                             (const uint64_t*)c->poff.p, n_contigs, nullptr, nullptr, nullptr,
                             (uint32_t*)c->stats.p, two_level ? range_shift + 8 : range_shift,
                             (uint32_t*)c->hist2.p, nullptr, nullptr,
                             clear_mask ? (unsigned long long*)run.d_mask : nullptr,  // also clears the keep mask
                             filter, exc, cap, exc_cnt);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_PREP], s1));
    {
        KernelSpan sp(c, "scan_radix_hist(3 kernels)");
        qmcp::launch_exclusive_scan(s1, (const uint32_t*)c->hist2.p, 256u * qmcp::part_pass_pitch(n),
                                    (uint32_t*)c->hist2.p, (uint32_t*)c->spine2.p, true);
    }
    if (!two_level) {
        KernelSpan sp(c, "k_range_partition");
        qmcp::launch_range_partition(s1, nullptr, run.d_starts, (const uint64_t*)c->roff.p,
                                     (const uint64_t*)c->poff.p, n_contigs, n, range_shift,
                                     (const uint32_t*)c->hist2.p, (uint16_t*)c->keys[0].p,
                                     (uint32_t*)c->vals[0].p, d_range_start, d_max_load, run.d_ends, filter);
    } else {
        // more than 256 ranges (genomes beyond 8.39 M positions): first into <= 256 super-ranges as
        // {global start, index} records, then every super-range into its final ranges
        {
            KernelSpan sp(c, "k_range_partition(level 1)");
            qmcp::launch_partition_level1(s1, run.d_starts, (const uint64_t*)c->roff.p,
                                          (const uint64_t*)c->poff.p, n_contigs, n, range_shift + 8,
                                          (const uint32_t*)c->hist2.p, c->keys[1].p, d_seg_tables, d_max_load,
                                          run.d_ends, filter);
        }
        KernelSpan sp(c, "partition level 2 (tables, hist, scan, scatter)");
        qmcp::launch_partition_level2(s1, c->keys[1].p, n, range_shift, d_seg_tables, (uint32_t*)c->hist.p,
                                      (uint32_t*)c->spine.p, (uint16_t*)c->keys[0].p,
                                      (uint32_t*)c->vals[0].p, d_range_start, d_max_load);
    }
    HIP_TRY(hipEventRecord(c->ev_fork, s1));  // statistics and heaviest load are final here
    {
        // per-range LDS histogram scanned in place: bucket offsets without a genome-wide scan; it also
        // counts the positions that start no read (stats word 3: the host picks the sweep kernel by it)
        KernelSpan sp(c, "k_range_offsets");
        qmcp::launch_range_offsets(s1, (const uint16_t*)c->keys[0].p, d_range_start, range_shift, ltot,
                                   (uint32_t*)c->boff.p, (uint32_t*)c->stats.p + 3);
    }
    HIP_TRY(hipGetLastError());
    run.nu_filter = filter;
    return QMCP_OK;
}

// The pass-major head's stages once more on `st` -- producer (regular reads: span == filter, or every read when
// filter == 0), scan, range table, bucket offsets -- for a call whose head ran with the wrong idea of the spans.
int queue_pm_head(qmcp_hip_ctx* c, hipStream_t st, uint32_t filter) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot, n_contigs = run.n_contigs;
    uint32_t* d_stats = (uint32_t*)c->stats.p;
    static const uint32_t zeros[4] = {0u, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(d_stats + 3, zeros, sizeof(zeros), hipMemcpyHostToDevice, st));  // empty positions, exceptions, list flag, overflow entries
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    {
        KernelSpan sp(c, "k_pm_prepare_sort", st);
        qmcp::launch_pm_prepare_sort(st, run.d_starts, run.d_ends, n, (const uint64_t*)c->roff.p, (const uint64_t*)c->poff.p,
                                     n_contigs, run.range_shift, ltot, (uint16_t*)c->keys[0].p, (uint16_t*)c->keys[1].p,
                                     (uint32_t*)c->hist2.p, (uint32_t*)c->hist.p,
                                     (uint32_t*)c->pm_work.p, d_stats, (unsigned long long*)run.d_mask, filter,
                                     filter ? (uint32_t*)c->nu_exc.p : nullptr, nu_cap_for(n),
                                     filter ? qmcp::nu_exc_counts((uint32_t*)c->nu_exc.p, nu_cap_for(n)) : nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_PREP], st));
    {
        KernelSpan sp(c, "scan_radix_hist(3 kernels)", st);
        qmcp::launch_exclusive_scan(st, (const uint32_t*)c->hist2.p, 256u * qmcp::pm_pitch(n), (uint32_t*)c->hist2.p,
                                    (uint32_t*)c->spine2.p, true);
    }
    {
        KernelSpan sp(c, "k_pm_descr + k_pm_range_table", st);
        qmcp::launch_pm_descr(st, (const uint32_t*)c->hist2.p, (const uint32_t*)c->hist.p, n, ltot, run.range_shift,
                              (uint32_t*)c->pm_desc.p, (uint32_t*)c->pm_work.p, d_range_start, d_max_load);
    }
    HIP_TRY(hipEventRecord(c->ev_fork, st));  // statistics and heaviest load are final here
    {
        KernelSpan sp(c, "k_pm_offsets", st);
        qmcp::launch_pm_offsets(st, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->pm_desc.p, (const uint32_t*)c->hist2.p, n,
                                d_range_start, run.range_shift, ltot, (uint32_t*)c->boff.p, d_stats + 3);
    }
    HIP_TRY(hipGetLastError());
    run.nu_filter = filter;
    return QMCP_OK;
}

// The ranking of the pass-major form on `st`: the ordered walk, then the settling of the quota-crossing groups it listed.
void queue_pm_rank(qmcp_hip_ctx* c, hipStream_t st, const uint32_t* ev_sev, const uint32_t* ev_lastns, uint32_t ell) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot;
    const bool by_records = qmcp::rank_scratch_by_records(run.range_shift, ltot, n);
    const uint16_t* keys16 = (const uint16_t*)c->keys[0].p;
    const uint16_t* idx16 = (const uint16_t*)c->keys[1].p;
    const uint32_t* desc = (const uint32_t*)c->pm_desc.p;
    const uint32_t* Tp = (const uint32_t*)c->hist2.p;
    const uint32_t* range_start = (const uint32_t*)c->ranges.p;
    uint32_t* amb_count = (uint32_t*)c->pm_work.p + 512;
    unsigned long long* kept_total = (unsigned long long*)c->scalars.p;
    {
        KernelSpan sp(c, "k_pm_walk", st);
        qmcp::launch_pm_walk(st, keys16, idx16, desc, Tp, n, range_start, run.range_shift, ltot, (const uint32_t*)c->boff.p,
                             (const uint32_t*)c->selend.p, (unsigned long long*)run.d_mask, kept_total, c->rankamb.p, by_records,
                             amb_count, ev_sev, ev_lastns, (const uint64_t*)c->poff.p, run.n_contigs, ell);
    }
    KernelSpan sp(c, "k_pm_settle", st);
    qmcp::launch_pm_settle(st, keys16, idx16, desc, Tp, n, range_start, run.range_shift, ltot, c->rankamb.p, by_records,
                           amb_count, (unsigned long long*)run.d_mask, kept_total);
}
