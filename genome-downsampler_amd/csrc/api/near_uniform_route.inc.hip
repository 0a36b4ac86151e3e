// near_uniform_route.inc.hip -- part of qmcp_api.hip (one translation unit).
// The near-uniform route's half of the tail (kernels/near_uniform.inc.hip).  Called when the call's spans differ.
// done = true: the keep mask is written (sweep, ranking and the selected exceptions); false: the caller takes the
// mixed-span route (nothing has touched the mask).  The head's producer may already have filtered on c->nu_ell
// (run.nu_filter); otherwise the reads are counted first and, if the longest span is the dominant one, the head's
// stages are queued again with the filter on.
constexpr uint32_t kNuRunInsApart = 2;
constexpr double kNuStretchDepth = 3.1;
int near_uniform_tail(qmcp_hip_ctx* c, uint32_t min_span, uint32_t max_span, uint32_t max_load, uint32_t* d_iters, bool& done) {
    done = false;
    SolveRun& run = c->run;
    const Problem& pr = run.pr;
    qmcp_hip_stats& local = run.local;
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot, n_contigs = run.n_contigs, M = run.M;
    local.near_uniform_giveup = QMCP_NU_GIVEUP_NOT_TRIED;
    if (c->opt.near_uniform < 0) return QMCP_OK;
    const uint32_t ell = max_span;
    const double depth = (double)n * (double)ell / ((double)ltot * (double)(M ? M : 1));
    const bool dbg = c->opt.near_uniform_debug == 1;
    if (dbg) fprintf(stderr, "[near] pm %d may_rank %d ell %u ev %d depth %.2f min_span %u filter %u\n", (int)run.pm,
                     (int)run.may_rank, ell, (int)qmcp::sweep_uniform_ev_supported(ell, M), depth, min_span, run.nu_filter);
    double min_depth = kNuMinDepth;
    if (c->opt.near_uniform_min_depth > 0.f) min_depth = c->opt.near_uniform_min_depth;  // (lab)
    // (the sigma depth where it is larger: M = 400 at 1.2 x M has as few cut points as M = 50 at 1.67 x M -- and the mixed-span
    //  walk it was left to took 233 ms for 11.9 M reads on 3.7 M positions, 200 x the one-length solve: lab/cliff_hunt.py)
    if (!run.may_rank || spec_sigma_depth(depth, M) < min_depth || min_span == 0) return QMCP_OK;
    // Which sweep the rounds run.  Deeper than 11 x M: one chain per contig in the event-driven form -- what the one-span
    // route runs there too -- restarted from its checkpoints.  Shallower (round 4): the block-scan pipeline in STRETCHES,
    // as the one-span route does -- real cut points (coverage of ALL reads <= M: every read over them is kept in every
    // round, whatever has been selected) and speculative boundaries checked on the device (uniform_sweep.inc.hip) -- with
    // the need moved by nadj; every round sweeps everything (hundreds of short chains side by side: a whole chain per
    // contig was 7 ms a sweep for cfg4's 10^6 positions at 1.5 x M, and long shallow contigs did not take the route).
    // Between 3.1 and 11 x M contigs of up to 2 M positions keep the chain: the speculative run-ins there are 1 536 - 2 304
    // blocks, as long as such a contig, and the chain changes fewer blocks the deeper the data (lab/near_uniform_depths.py,
    // cfg4's reads with 1 % clipped, chain / stretches ms: 6.3 x M 6.0 / 13.1; 4.7 x M 5.5 / 13.2; 3.75 x M 6.4 / 11.5;
    // with 40 % of the reads: 5 x M 12.4 / 15.3; 3 x M 15.7 / 11.2; 2.1 x M gives up / 14.2; 1.5 x M 41.9 / 8.6).
    uint32_t longest = 0;
    for (uint32_t k = 0; k < n_contigs; ++k) longest = run.lengths[k] > longest ? run.lengths[k] : longest;
    // (the event-driven form's own limits: scratch for short spans, M in a packed field)
    const bool ev_ok = ell >= ev_min_span() && qmcp::sweep_uniform_ev_supported(ell, M);
    // (many times M and yet sparse -- a small M: more than half of the blocks hold a position without a read, the chain
    //  would run its general step on nearly every block -- is shallow in standard deviations: stretches, as in
    //  launch_uniform_sweep.  66 M reads on one contig of 82.6 M positions at 12 x M with M = 10, 1 % clipped: 3.2 s in chains.)
    const bool sparse = (double)n / (double)ltot < std::log((double)ell / 0.693);
    const double depth_gate = (depth >= kGenDepth && sparse && spec_sigma_depth(depth, M) < kGenDepth) ? spec_sigma_depth(depth, M) : depth;
    // (sparse data -- more than half of the blocks hold a position without a read -- makes the chain run its general step on
    //  nearly every block: 551 k reads on 1 M positions at 4 x M with M = 20, 1 % clipped: 36.7 ms in chains, lab/cliff_hunt.py)
    bool stretches = depth_gate < kGenDepth && qmcp::sweep_uniform_mw_supported(ell) && (depth_gate < kNuStretchDepth || longest > 2000000u || !ev_ok || sparse);
    // (deep data whose M does not fit a packed field of the event-driven form -- M = 200 at reads of 250 --: the block-scan
    //  pipeline, one chain per contig, every round a whole sweep; contigs of up to 2 M positions -- 99.7 M reads on 24
    //  contigs at 12 x M were 112 ms on the mixed-span walk against 2.7 with one length, lab/cliff_hunt.py)
    if (depth_gate >= kGenDepth && !ev_ok && qmcp::sweep_uniform_mw_supported(ell) && longest <= 2000000u) stretches = true;
    if (c->opt.sweep == QMCP_SWEEP_EVENTS) stretches = false;
    if (c->opt.sweep == QMCP_SWEEP_GENERAL) stretches = qmcp::sweep_uniform_mw_supported(ell);
    if (!stretches) {
        if (!ev_ok) return QMCP_OK;
        if (depth_gate < kGenDepth && longest > 2000000u) return QMCP_OK;  // (spans the pipeline does not take: a whole chain per round)
    }
    if (c->nu_failed_n == run.n64 && c->nu_failed_ltot == pr.ltot && c->nu_failed_ell == ell && c->nu_failed_M == M) {
        local.near_uniform_giveup = QMCP_NU_GIVEUP_REMEMBERED;
        c->nu_ell = 0;
        return QMCP_OK;
    }
    hipStream_t st = c->stream;
    const uint32_t cap = nu_cap_for(n);
    uint32_t n_exc = 0;
    uint32_t* d_stats = (uint32_t*)c->stats.p;
    if (run.nu_filter == ell) {
        n_exc = c->h_head[5];  // (read back beside the statistics)
        if (c->h_head[6] != 0) { c->nu_ell = 0; local.near_uniform_giveup = QMCP_NU_GIVEUP_TOO_MANY; return QMCP_OK; }  // a pass held more exceptions than it can stage
    } else {
        // how many reads have the longest span?  (one pass over the spans; the host waits for the count)
        TRY(ensure_near_uniform(c, n, ltot, n_contigs));
        HIP_TRY(hipMemsetAsync(d_stats + 7, 0, sizeof(uint32_t), st));
        qmcp::launch_nu_count_span(st, run.d_starts, run.d_ends, n, ell, d_stats + 7);
        HIP_TRY(hipMemcpyAsync(c->h_nu, d_stats + 7, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_exc = n - c->h_nu[0];
        if (dbg) fprintf(stderr, "[near] reads of span %u: %u of %u, list holds %u\n", ell, c->h_nu[0], n, cap);
        if (n_exc > n / 10u) {
            // (fewer than nine tenths of the reads have the LONGEST span: either many exceptions, or -- nearly all reads
            //  shorter than a few -- the dominant span is not the longest: reads lengthened by a deletion)
            c->nu_ell = 0;
            local.near_uniform_giveup = n_exc > n - n / 10u ? QMCP_NU_GIVEUP_LONGER_READS : QMCP_NU_GIVEUP_TOO_MANY;
            return QMCP_OK;
        }
        // the head again, regular reads only (exceptions listed): producer, scan, range table, bucket offsets
        c->nu_ell = ell;
        if (run.pm) TRY(queue_pm_head(c, st, ell));
        else TRY(queue_rm_head(c, st, ell, true));
        uint32_t* d_max_load = (uint32_t*)c->ranges.p + 65540;
        HIP_TRY(hipMemcpyAsync(c->h_nu, d_max_load, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(c->h_nu + 1, d_stats + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        max_load = c->h_nu[0];
        if (c->h_nu[1] != n_exc || c->h_nu[2] != 0) { c->nu_ell = 0; local.near_uniform_giveup = QMCP_NU_GIVEUP_TOO_MANY; return QMCP_OK; }  // (a pass held more than it can stage)
    }
    local.near_uniform_exceptions = n_exc;
    if (n_exc == 0 || n_exc > n / 10u || (uint64_t)max_load * kRankBalance > (uint64_t)n) {
        if (n_exc > n / 10u) c->nu_ell = 0;
        local.near_uniform_giveup = n_exc > n / 10u ? QMCP_NU_GIVEUP_TOO_MANY : n_exc == 0 ? QMCP_NU_GIVEUP_NOT_TRIED : QMCP_NU_GIVEUP_HEAVY_RANGE;
        return QMCP_OK;
    }
    // scratch of the event-driven sweep (launch_uniform_sweep)
    if (!stretches) {
        TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, ell, n_contigs + 768)));
        TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, ell, n_contigs + 768)));
    }
    const uint32_t* boff = (const uint32_t*)c->boff.p;
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    uint32_t* selend = (uint32_t*)c->selend.p;
    uint32_t* exc = (uint32_t*)c->nu_exc.p;
    int32_t* nadj = (int32_t*)c->nu_nadj.p;
    uint32_t* state = (uint32_t*)c->nu_state.p;
    unsigned long long* viol_key = (unsigned long long*)((char*)c->nu_state.p + 64);
    uint32_t* viol_idx = (uint32_t*)(viol_key + n_contigs);
    uint32_t* sweep_from[2] = {viol_idx + n_contigs, viol_idx + 2 * (size_t)n_contigs};  // this round's, the next round's
    if (!stretches) TRY(ensure(c, c->nu_ckpt, qmcp::sweep_ev_ckpt_bytes(ltot, ell, n_contigs + 768)));
    HIP_TRY(hipMemsetAsync(sweep_from[0], 0, (size_t)n_contigs * sizeof(uint32_t), st));
    HIP_TRY(hipEventRecord(c->ev[EV_SCAN], st));
    HIP_TRY(hipEventRecord(c->ev[EV_SORT], st));
    {
        KernelSpan sp(c, "near-uniform setup (exception coverage, need, pre-selection)");
        qmcp::launch_nu_setup(st, exc, cap, n_exc, d_stats + 6, boff, ltot, ell, M, (uint32_t*)c->nu_ce.p, (uint32_t*)c->spine.p,
                              nadj, state);
    }
    // stretches: the exact table once (the cut points do not move between rounds), the speculative ones per sweep
    const uint32_t* seg = nullptr;
    uint32_t n_seg_max = 0, windows = 0;
    bool speculate = false;
    const uint32_t burn_blocks = spec_first_run_in(c, spec_depth_in_sigma(depth, M)) * ((depth_gate != depth && !c->opt.speculation_run_in) ? 3u : 1u);  // (launch_uniform_sweep)
    if (stretches) {
        windows = sweep_cut_windows(c, ltot, ell, n_contigs, true);
        if (windows != 0) {
            KernelSpan sp(c, "k_find_cuts", st);
            seg = qmcp::launch_sweep_segments(st, boff, nullptr, poff, n_contigs, ltot, ell, M, windows, (uint32_t*)c->segs.p,
                                              (const uint32_t*)c->nu_ce.p);
            n_seg_max = n_contigs + windows;
        }
        const double sig = spec_depth_in_sigma(depth, M);
        speculate = spec_wanted(c, depth_gate != depth ? depth_gate : (sig > depth ? sig * (kSpecDepth / 9.0) : depth), spec_depth_in_sigma(depth_gate, M)) && windows != 0 && burn_blocks >= 2 && (uint64_t)ltot >= 8ull * burn_blocks * ell;
        // (stretches two run-ins long instead of the one-span route's four: the route sweeps several times, and a sweep is
        //  as long as its longest stretch -- two 10^7-position contigs at 1.5 x M: 0.43 -> 0.24 ms a sweep; one run-in long: 0.37)
    }
    // later rounds sweep only the exact stretches a selection of the round before touched (k_nu_select_apply marks them)
    uint32_t* marks[2] = {(uint32_t*)((char*)c->nu_sus.p + qmcp::nu_suspect_bytes(nu_suspects_for(n))), nullptr};
    const uint32_t mw = nu_marks_words(n_contigs);
    marks[1] = marks[0] + mw;
    // ... and, where the sweeps speculate, only the first tier's stretches that read or write something a selection changed
    // (k_nu_select_apply; a boundary is compared when the stretch on either side of it was swept)
    uint32_t* fine[2] = {marks[1] + mw, marks[1] + 2 * (size_t)mw};
    uint32_t* dirty[2] = {fine[1] + mw, nullptr};  // (cells a replay reads from that changed: kernels/near_uniform.inc.hip NuBins)
    dirty[1] = dirty[0] + qmcp::nu_cells_bytes() / sizeof(uint32_t);
    if (stretches) HIP_TRY(hipMemsetAsync(dirty[0], 0, 2 * qmcp::nu_cells_bytes(), st));
    // (the first tier's table: launch_sweep_segments_speculative(..., tier 1) builds it at this place on every sweep, from
    //  the same cut points -- the same table every round)
    const uint32_t* seg_fine = speculate ? (const uint32_t*)c->segs.p + windows + (1 + 5 * ((size_t)n_contigs + windows)) : nullptr;
    if (c->opt.near_uniform_debug == 2) seg_fine = nullptr;  // (lab: every round sweeps every stretch its exact marks cover)
    // Rounds are queued two at a time and the host looks at the state words after each pair: a round whose contigs are
    // all settled is eight launches that return at once (the chain sweeps nothing, the verification skips every
    // exception: ~0.1 ms), about what one more host round trip costs; measured at cfg4 with 1 % clipped reads (7 rounds),
    // batches of 1 / 2 / 2 + 4 + 4: 4.62 / 4.5 / 4.60 ms.
    uint32_t rounds = 0;
    bool settled = false;
    const uint32_t budget = nu_round_budget(c, run.lengths, n_contigs);
    // A shape that settled within a few rounds last time: queue that many (a round whose contigs are all settled sweeps
    // and verifies nothing) and the ranking behind them WITHOUT waiting in between; collect_one looks at the state words.
    const bool defer = !dbg && c->opt.near_uniform_rounds == 0 && c->nu_need_rounds != 0 && c->nu_need_rounds <= 8 &&
                       c->nu_need_n == run.n64 && c->nu_need_ltot == pr.ltot && c->nu_need_ell == ell && c->nu_need_M == M &&
                       run.nu_filter == ell;
    while (rounds < budget && !settled) {
        const uint32_t batch = 2u;
        for (uint32_t r = 0; r < batch; ++r) {
            ++rounds;
            if (stretches) {
                if (speculate) {
                    TRY(speculative_sweep(
                        c, st, n_contigs, ltot, windows, ell, ell, burn_blocks, kNuRunInsApart, seg, "k_sweep_uniform_gen",
                        [&](const uint32_t* table, uint32_t* run_in_out, const uint32_t* redo_in) {
                            return qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, ell, M, ltot, selend, d_iters, table,
                                                                  n_seg_max, run_in_out, redo_in, nadj,
                                                                  (table == seg_fine && rounds > 1) ? fine[0] : nullptr);
                        },
                        [&](const uint32_t* table, uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
                            qmcp::launch_spec_verify(st, table, n_seg_max, ell, selend, (const uint32_t*)c->cstart.p, mismatches,
                                                     redo_in, redo_out, (table == seg_fine && rounds > 1) ? fine[0] : nullptr);
                        },
                        rounds == 1 ? nullptr : marks[0]));
                } else {
                    KernelSpan sp(c, "k_sweep_uniform_gen", st);
                    if (!qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, ell, M, ltot, selend, d_iters, seg, n_seg_max,
                                                        nullptr, (seg != nullptr && rounds > 1) ? marks[0] : nullptr, nadj))
                        return fail(QMCP_ERANGE, "near-uniform route: span %u not supported", ell);
                }
            } else {
            {
                KernelSpan sp(c, "k_sweep_pack", st);
                qmcp::launch_sweep_ev_pack(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (uint32_t*)c->evpk.p, nadj, sweep_from[0]);
            }
            {
                KernelSpan sp(c, "k_sweep_uniform_ev", st);
                qmcp::launch_sweep_ev_chain(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (const uint32_t*)c->evpk.p,
                                            (uint32_t*)c->cstart.p, (uint32_t*)c->evlast.p, d_iters, nadj, (uint32_t*)c->nu_ckpt.p,
                                            sweep_from[0]);
            }
            {
                KernelSpan sp(c, "k_sweep_expand", st);
                qmcp::launch_sweep_ev_expand(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (const uint32_t*)c->cstart.p,
                                             (const uint32_t*)c->evlast.p, selend, sweep_from[0]);
            }
            }
            {
                KernelSpan sp(c, "near-uniform round (verify, replay, select, apply)");
                qmcp::launch_nu_round(st, exc, cap, n_exc, d_stats + 6, rounds == 1, boff, selend, nadj, (const uint32_t*)c->nu_ce.p, poff, n_contigs, ell, M,
                                      (uint2*)c->nu_sus.p, nu_suspects_for(n), state, viol_key, viol_idx, sweep_from[0], sweep_from[1],
                                      ltot, (uint32_t*)c->spine.p, stretches ? seg : nullptr, n_seg_max, marks[1],
                                      stretches ? (uint32_t*)c->nu_prev.p : nullptr, dirty[0], dirty[1], seg_fine, fine[1]);
                std::swap(sweep_from[0], sweep_from[1]);
                std::swap(marks[0], marks[1]);
                std::swap(fine[0], fine[1]);
                std::swap(dirty[0], dirty[1]);
            }
        }
        HIP_TRY(hipGetLastError());
        if (defer && rounds < c->nu_need_rounds) continue;
        HIP_TRY(hipMemcpyAsync(c->h_nu, state, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        if (defer) break;  // (looked at when the solve is collected)
        HIP_TRY(hipStreamSynchronize(st));
        if (dbg) {
            uint32_t more[8];
            (void)hipMemcpy(more, state + 8, sizeof(more), hipMemcpyDeviceToHost);
            fprintf(stderr, "[near] open question: s %u u1 %u S(u1-1) %u C(u1-1) %u b %u c0 %u\n", more[0], more[1], more[2], more[3], more[4], more[5]);
            fprintf(stderr, "[near] after %u rounds: last round selected %u, flags %u (read %u), selected in all %u, suspects %u, rounds that selected %u; next sweeps from block",
                    rounds, c->h_nu[1], c->h_nu[2], c->h_nu[5], c->h_nu[3], c->h_nu[4], c->h_nu[6]);
            std::vector<uint32_t> from(n_contigs);
            (void)hipMemcpy(from.data(), sweep_from[0], (size_t)n_contigs * sizeof(uint32_t), hipMemcpyDeviceToHost);
            for (uint32_t k = 0; k < n_contigs && k < 16; ++k) fprintf(stderr, " %d", (int)from[k]);
            fprintf(stderr, "\n");
        }
        if (c->h_nu[2] != 0) break;          // a run the replay does not model, or too many suspects
        settled = c->h_nu[1] == 0;           // the last round wanted no exception: the sweep's counts are the greedy's
    }
    c->nu_deferred = defer;
    if (defer) {
        settled = true;  // (provisionally: collect_one checks, and solves the call again if it is not)
    } else if (settled) {
        c->nu_need_n = run.n64; c->nu_need_ltot = pr.ltot; c->nu_need_ell = ell; c->nu_need_M = M;
        c->nu_need_rounds = rounds;              // (queued: an even number)
        rounds = c->h_nu[6] + 1;                 // (the rounds that did something, and the one that found nothing left)
    }
    local.near_uniform_rounds = rounds;
    local.near_uniform_selected = defer ? 0u : c->h_nu[3];
    if (!settled) {
        c->nu_need_rounds = 0;
        // (the head must not filter on this span again, and the next call of this shape must not burn the budget again)
        local.near_uniform_giveup = c->h_nu[2] != 0 ? QMCP_NU_GIVEUP_UNMODELLED : QMCP_NU_GIVEUP_BUDGET;
        c->nu_ell = 0;
        c->nu_failed_n = run.n64; c->nu_failed_ltot = pr.ltot; c->nu_failed_ell = ell; c->nu_failed_M = M;
        return QMCP_OK;
    }
    local.near_uniform_giveup = QMCP_NU_GIVEUP_NONE;
    HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], st));
    if (run.pm) {
        queue_pm_rank(c, st, nullptr, nullptr, 0);
    } else {
        KernelSpan sp(c, "k_rank_mark");
        qmcp::launch_rank_mark(st, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->vals[0].p, (const uint32_t*)c->ranges.p,
                               run.range_shift, ltot, boff, selend, (unsigned long long*)run.d_mask,
                               (unsigned long long*)c->scalars.p, c->rankamb.p,
                               qmcp::rank_scratch_by_records(run.range_shift, ltot, n));
    }
    {
        KernelSpan sp(c, "k_nu_mark_selected");
        qmcp::launch_nu_mark_selected(st, exc, cap, n_exc, d_stats + 6, (unsigned long long*)run.d_mask,
                                      (unsigned long long*)c->scalars.p);
    }
    HIP_TRY(hipGetLastError());
    done = true;
    return QMCP_OK;
}
