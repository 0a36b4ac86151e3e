// uniform_sweep.inc.hip -- part of qmcp_api.hip (one translation unit).
// Which one-length sweep a call takes (block-scan pipelines, event-driven form, stretches at cut points, speculative boundaries in tiers) and its launches.
// The uniform-span sweep: seven waves per contig where the span allows it (fast form with checked
// fallback on deep data, every block in the general form on shallow data -- both exact, the
// choice is about speed only), else the single-wave kernel.  QMCP_HIP_SWEEP=fast|gen overrides.
// Cut-point segmentation of the uniform sweeps (QMCP_HIP_CUTS=0|1 overrides): looked for where mean
// coverage is a small multiple of M -- deep data has no cut points, and the look costs two launches.
uint32_t sweep_cut_windows(const qmcp_hip_ctx* c, uint32_t ltot, uint32_t span, uint32_t n_contigs, bool shallow,
                           uint32_t max_windows = qmcp::kSweepWindowsOneSpan) {
    bool on = shallow;
    if (c->opt.cut_points != 0) on = c->opt.cut_points > 0;
    return on ? qmcp::sweep_segment_windows(ltot, span, n_contigs, max_windows) : 0u;
}

// Speculative stretch boundaries: below this mean coverage (in units of M), with a run-in (in blocks)
// that grows with the depth.  lab/spec_burn_study.py, cfg5's shape at 1/32 scale, boundaries that
// disagreed at a run-in of 128 / 256 / 512 / 1024 blocks: depth 2.0: 2 of 364 / 0 / 0 / 0; 2.5: 67 of 364 /
// 1 of 240 / 0 / 0; 3.0: 157 / 28 / 0 of 118 / 0; 4.0: 273 / 86 / 6 of 118 / 0 of 56 -- about twice
// the run-in per half unit of depth.  Two tiers: the first with the run-in of this table, and -- only
// if some boundary disagreed -- a second with three times that (or, where the genome is too short for it,
// none: the exact table); the exact sweep runs only if the second tier disagrees somewhere too.  Every
// tier's launches are queued at once and gated by device words, so nothing waits for the host.
// Round 3 (lab/spec_depth_gap.py, one contig of 20 M positions at 100 x coverage, profiles/r03_spec_depth_gap.log):
// between 4.1 and 11 x M -- where round 2 swept whole contigs as one chain each -- the sweep forgets its start too,
// within about a thousand blocks: boundaries that disagreed at a run-in of 256 / 512 / 1024 / 2048 blocks: depth 4.2:
// 55 of 127 / 2 of 63 / 0 of 31 / 0; 5.9: 85 / 11 / 0 / 0; 8.3: 100 / 20 / 0 / 0; 10: 108 / 23 / 1 of 31 / 0 of 15 --
// sweep 30.9 -> 1.3 ms.  So every depth the general-form sweep takes (below kGenDepth) is speculated on; at cfg4's
// depth (18.75, and at 37.5) every boundary still disagrees at 2 048 blocks (lab/spec_deep_probe.py): the event-driven
// chain stays whole there.
constexpr double kSpecDepth = kGenDepth, kSpecMinDepth = 1.3;
uint32_t spec_burn_blocks(double depth) {
    return depth < 2.1 ? 320u : depth < 2.6 ? 640u : depth < 3.1 ? 1152u : depth < 4.1 ? 2304u : 1536u;
}
double spec_depth_in_sigma(double depth, uint32_t M);
bool spec_wanted(const qmcp_hip_ctx* c, double depth, double depth_lo = -1.0 /* what the lower bound looks at: the sigma
                 depth where it is larger -- "nearly every window has a real cut point" below 1.3 x M holds for M = 50, not
                 for M = 400, whose 1.2 x M is as far above M in standard deviations as M = 50 at 1.67 x M */) {
    if (depth_lo < 0.0) depth_lo = depth;
    bool on = depth < kSpecDepth && depth_lo > kSpecMinDepth;  // (shallower: nearly every window has a real cut point)
    if (c->opt.speculation != 0) on = c->opt.speculation > 0;  // (never / at any depth)
    return on;
}
uint32_t spec_first_run_in(const qmcp_hip_ctx* c, double depth) {
    return c->opt.speculation_run_in ? c->opt.speculation_run_in : spec_burn_blocks(depth);
}
// The table above was measured at M = 50.  What makes a sweep forget its start is how often the coverage comes near M --
// how many standard deviations above M it sits: z = (mean coverage - M) / sqrt(mean coverage) = sqrt(M) (d - 1) / sqrt(d)
// for Poisson starts at depth d x M.  This is the depth at which M = 50 has the same z; the run-in is looked up at the
// larger of the two (a smaller M keeps the table's value: measured over-provisioned, not under).  Second half of round 4,
// lab/spec_run_in_vs_M.py, one contig of 60 M positions, one read length, boundaries that disagreed / sweep ms at the
// table's run-in and at the corrected one: M = 100 at 2 x M: 12 of 255 / 1.98 -> (1 152 blocks) none; M = 100 at 3 x M:
// 16 of 85 / 5.75 -> 4 of 63 / 7.5; **M = 200 at 2 x M: 161 of 255, the second tier failing too: the exact sweep, 94 ms ->
// (2 304 blocks) ~10 ms**.
double spec_sigma_depth(double depth, uint32_t M) {   // (may be smaller than the depth: a small M)
    if (!(depth > 1.0)) return depth;
    const double y = std::sqrt((double)M / 50.0) * (depth - 1.0) / std::sqrt(depth);
    const double x = 0.5 * (y + std::sqrt(y * y + 4.0));
    return x * x;
}
double spec_depth_in_sigma(double depth, uint32_t M) {
    const double d_eff = spec_sigma_depth(depth, M);
    return d_eff > depth ? d_eff : depth;
}

// device words of a speculative sweep, behind the solve's other scalars
struct SpecWords {
    uint32_t* mismatches1;  // tier 1: boundaries that disagreed
    uint32_t* n_spec1;      //         speculative boundaries
    uint32_t* mismatches2;  // tier 2
    uint32_t* n_spec2;
};
SpecWords spec_words(qmcp_hip_ctx* c) {
    uint32_t* w = (uint32_t*)((char*)c->scalars.p + 32);
    return SpecWords{w, w + 1, w + 2, w + 3};
}

// The tiers of a speculative sweep.  `unit`: positions per block of run-in (the span; the largest span of a
// mix), `round_to`: the run-in is made a multiple of this many positions.  sweep(table, second output or null,
// marks to obey or null) launches the sweep kernel; check(table, mismatch counter, marks to obey or
// null, marks to set) the comparison and the merge behind it.  A disagreement marks the exact stretch it
// lies in; tier 2 (three times the run-in) sweeps only marked parts, the exact sweep only what tier 2 marked.
template <class Sweep, class Check>
int speculative_sweep(qmcp_hip_ctx* c, hipStream_t st, uint32_t n_contigs, uint32_t ltot, uint32_t windows,
                      uint32_t unit, uint32_t round_to, uint32_t burn_blocks, uint32_t run_ins_apart,
                      const uint32_t* seg_exact, const char* sweep_name, Sweep sweep, Check check,
                      const uint32_t* only_marked = nullptr /* per exact stretch: the parts of the genome to sweep at all
                                                               (the near-uniform route's later rounds); null: everything */) {
    const SpecWords w = spec_words(c);
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    const uint32_t n_cand = n_contigs + windows;
    TRY(ensure(c, c->specflags, 2 * (size_t)n_cand * sizeof(uint32_t)));
    uint32_t* redo1 = (uint32_t*)c->specflags.p;
    uint32_t* redo2 = redo1 + n_cand;
    HIP_TRY(hipMemsetAsync(redo1, 0, 2 * (size_t)n_cand * sizeof(uint32_t), st));
    auto positions = [&](uint64_t blocks) { return (uint32_t)((blocks * unit + round_to - 1) / round_to * round_to); };
    const uint32_t burn1 = positions(burn_blocks);
    uint32_t burn2 = positions(3ull * burn_blocks);
    if ((uint64_t)ltot < 2ull * run_ins_apart * burn2) burn2 = 0;  // too short a genome: tier 2 is the exact table
    const uint32_t *seg1, *seg2;
    {
        KernelSpan sp(c, "k_find_cuts", st);
        seg1 = qmcp::launch_sweep_segments_speculative(st, poff, n_contigs, ltot, windows, burn1, (uint32_t*)c->segs.p,
                                                       w.n_spec1, run_ins_apart, 1);
        seg2 = qmcp::launch_sweep_segments_speculative(st, poff, n_contigs, ltot, windows, burn2, (uint32_t*)c->segs.p,
                                                       w.n_spec2, run_ins_apart, 2);
    }
    // the second output: one span -- every stretch's run-in; a mix of spans -- the odd stretches' whole output
    uint32_t* second_out = (uint32_t*)c->cstart.p;
    {
        KernelSpan sp(c, sweep_name, st);
        if (!sweep(seg1, second_out, only_marked)) return fail(QMCP_ERANGE, "speculative sweep: span not supported");
    }
    {
        KernelSpan sp(c, "k_spec_verify + k_spec_merge", st);
        check(seg1, w.mismatches1, only_marked, redo1);
    }
    {
        KernelSpan sp(c, "second tier, where the first disagreed", st);
        (void)sweep(seg2, second_out, redo1);
        check(seg2, w.mismatches2, redo1, redo2);
    }
    KernelSpan sp(c, "exact sweep, where the second tier disagreed", st);
    (void)sweep(seg_exact, nullptr, redo2);
    HIP_TRY(hipGetLastError());
    return QMCP_OK;
}

int launch_uniform_sweep(qmcp_hip_ctx* c, hipStream_t st, uint32_t n, uint32_t ltot, uint32_t n_contigs,
                         uint32_t span, uint32_t M, uint32_t* d_iters, uint32_t empty_positions,
                         bool* expand_left_out = nullptr /* in: the caller can read the event sweep's own output;
                                                            out: the event sweep ran whole contigs and selend[] was not written */) {
    const bool may_leave_expand = expand_left_out != nullptr && *expand_left_out;
    if (expand_left_out) *expand_left_out = false;
    // mean coverage in units of M: the fast form needs the binding jumps to come from the previous
    // block, which holds while coverage is many times M
    const double depth = (double)n * (double)span / ((double)ltot * (double)(M ? M : 1));
    bool gen = depth < kGenDepth;
    // Many times M and yet SPARSE (a small M: depth 12 x M with M = 10 is 0.8 reads a position): more than half of the blocks
    // hold a position without a read, so the event-driven form is out (below) and the fast form's check fails nearly
    // everywhere (~1 130 cycles a block measured); and in standard deviations such data is shallow -- it forgets as M = 50
    // at 3.7 x M does.  The general pipeline in speculative stretches, then, as below 11 x M (second half of round 4,
    // lab/cliff_hunt.py: one contig of 82.6 M positions, 66 M reads of one length, M = 10: 259 ms as one chain).
    double depth_gate = depth;   // what decides whether boundaries are speculated on
    {
        const double structural0 = (double)n_contigs * (double)(span - 1);
        const double holes0 = (double)empty_positions > structural0 ? (double)empty_positions - structural0 : 0.0;
        const bool sparse = empty_positions != 0xFFFFFFFFu && holes0 * (double)span > 0.693 * (double)ltot;
        const double ds = spec_sigma_depth(depth, M);
        if (!gen && sparse && ds < kGenDepth) { gen = true; depth_gate = ds; }
    }
    if (c->opt.sweep == QMCP_SWEEP_GENERAL) gen = true;
    if (c->opt.sweep == QMCP_SWEEP_FAST) gen = false;
    const uint32_t* boff = (const uint32_t*)c->boff.p;
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    uint32_t* selend = (uint32_t*)c->selend.p;
    // shallow or gapped data: split the contigs at cut points so that more than n_contigs chains run
    const uint32_t* seg = nullptr;
    uint32_t n_seg_max = 0;
    const uint32_t windows = sweep_cut_windows(c, ltot, span, n_contigs, gen);
    // Data a few times deeper than M: hardly any cut points, but the sweep forgets its start within tens
    // of blocks (kernels/sweep_segments.inc.hip), so windows without a cut get a speculative boundary with a
    // run-in (every few windows, so that stretches stay several run-ins long); the stretches' outputs are compared where they
    // meet, and if any pair disagrees the exact sweep runs after all (its launch is there either way and
    // returns at once when all agreed).
    // (sparse and many times M: three times the table's run-in -- lab/sparse_deep_run_ins.py, boundaries that disagreed, first /
    //  second tier, and sweep ms at 1 536 and at 4 608 blocks: M = 10 at 20 x M: 17 of 21 / 0, 7.2 -> 0 of 7, 5.4; M = 10 at
    //  12 x M: 7 of 42 / 0, 7.3 -> 0 of 14, 5.5; M = 20 at 15 x M: 31 of 31 / 7 -- the exact sweep --, 53 -> 7 of 10 / 0, 21)
    const uint32_t burn_blocks = spec_first_run_in(c, spec_depth_in_sigma(depth, M)) * ((depth_gate != depth && !c->opt.speculation_run_in) ? 3u : 1u);
    // (the upper bound in standard deviations too where that is the larger -- lab/spec_run_in_large_M.py, one contig of 20 M
    //  positions: M = 400 at 3 x M and M = 200 at 4 x M, sigma depths 12.6 and 11.5, do not forget within the contig at any
    //  run-in up to 9 216 blocks: 42 ms with the tiers against 35 as one chain; M = 400 at 2 x M, 5.4: 10.5 against 34)
    // (a sigma depth beyond the raw one counts a little more: 10.9 -- M = 200 at 4 x M -- behaves as 11.5 and 12.6 do)
    const double sig = spec_depth_in_sigma(depth, M);
    const double depth_hi = depth_gate != depth ? depth_gate : (sig > depth ? sig * (kSpecDepth / 9.0) : depth);
    const bool speculate = spec_wanted(c, depth_hi, spec_depth_in_sigma(depth_gate, M)) && gen && windows != 0 && qmcp::sweep_uniform_mw_supported(span) &&
                           burn_blocks >= 2 && (uint64_t)ltot >= 8ull * burn_blocks * span;
    if (windows != 0) {
        KernelSpan sp(c, "k_find_cuts", st);
        seg = qmcp::launch_sweep_segments(st, boff, nullptr, poff, n_contigs, ltot, span, M, windows, (uint32_t*)c->segs.p);
        n_seg_max = n_contigs + windows;
    }
    // deep data: the event-driven form (a block is only TESTED unless its counts fall below the kept
    // profile); spans below ev_min_span() would need more scratch than the arena holds for it
    // ... and only where few blocks have a start position that holds no read: such a block nearly always
    // changes the kept profile, and a changed block costs the event-driven chain ~6 x the block-scan
    // pipeline's chain step (amplicon panels, whose reads start in a few windows: cfg3 took 0.16 ms against
    // 0.05).  With a fraction z of empty positions about 1 - (1 - z)^span of the blocks have one: more than
    // half of them from z = ln 2 / span on.  (Unknown on the small-call route: block scan, as in round 1.)
    // (no read can start in the last span - 1 positions of a contig: those are not holes in the data)
    const double structural = (double)n_contigs * (double)(span - 1);
    const double holes = (double)empty_positions > structural ? (double)empty_positions - structural : 0.0;
    const bool spiky = empty_positions == 0xFFFFFFFFu || holes * (double)span > 0.693 * (double)ltot;
    bool ev = !gen && !spiky && span >= ev_min_span();
    if (c->opt.sweep == QMCP_SWEEP_EVENTS) ev = span >= ev_min_span();
    if (c->opt.sweep == QMCP_SWEEP_FAST || c->opt.sweep == QMCP_SWEEP_GENERAL) ev = false;
    if (ev && qmcp::sweep_uniform_ev_supported(span, M)) {
        // scratch of the event-driven form: 256 bytes per block, so it depends on the span, which is only
        // known here -- grown on the first deep call of a size (ensure() waits for the streams then), kept after
        {
            const uint32_t wg_max = n_contigs + 768;
            TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, span, wg_max)));
            TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, span, wg_max)));
        }
        uint32_t* pk = (uint32_t*)c->evpk.p;
        uint32_t* sev = (uint32_t*)c->cstart.p;
        uint32_t* lastns = (uint32_t*)c->evlast.p;
        {
            KernelSpan sp(c, "k_sweep_pack", st);
            qmcp::launch_sweep_ev_pack(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, pk);
        }
        {
            KernelSpan sp(c, "k_sweep_uniform_ev", st);
            qmcp::launch_sweep_ev_chain(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, pk, sev, lastns, d_iters);
        }
        if (may_leave_expand && seg == nullptr) {
            *expand_left_out = true;  // (the ranking reads sev / lastns itself)
            return QMCP_OK;
        }
        KernelSpan sp(c, "k_sweep_expand", st);
        qmcp::launch_sweep_ev_expand(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, sev, lastns, selend);
        return QMCP_OK;
    }
    if (speculate && seg != nullptr) {
        return speculative_sweep(
            c, st, n_contigs, ltot, windows, span, span, burn_blocks, 4, seg, "k_sweep_uniform_gen",
            [&](const uint32_t* table, uint32_t* run_in_out, const uint32_t* redo_in) {
                return qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, table, n_seg_max,
                                                      run_in_out, redo_in);
            },
            [&](const uint32_t* table, uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
                qmcp::launch_spec_verify(st, table, n_seg_max, span, selend, (const uint32_t*)c->cstart.p, mismatches,
                                         redo_in, redo_out);
            });
    }
    if (qmcp::sweep_uniform_mw_supported(span)) {
        KernelSpan sp(c, gen ? "k_sweep_uniform_gen" : "k_sweep_uniform_mw", st);
        const bool ok = gen ? qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max)
                            : qmcp::launch_sweep_uniform_mw(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max);
        if (ok) return QMCP_OK;
    }
    KernelSpan sp(c, "k_sweep_uniform", st);
    if (!qmcp::launch_sweep_uniform(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max))
        return fail(QMCP_ERANGE, "uniform span %u not supported", span);
    return QMCP_OK;
}
