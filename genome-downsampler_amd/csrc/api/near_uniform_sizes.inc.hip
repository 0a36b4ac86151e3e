// near_uniform_sizes.inc.hip -- part of qmcp_api.hip (one translation unit).
// The near-uniform route (kernels/near_uniform.inc.hip): sizes, round budget, buffers (a solve's head sizes them).
// Near-uniform route: sizes.  Exceptions beyond a tenth of the reads are not worth the route (every one the sweep
// wants costs a sweep of its own); the list holds an eighth of every wave's reads.
// The suspects of a round (exceptions whose own bucket is used up in the sweep's final counts): 64 Ki entries, or a
// sixty-fourth of the reads on large calls -- one GPU's real share of configs[4] (129.7 M reads at 2 x M, where half of
// all reads are kept) lists 125 k of its 1.3 M clipped reads in a round, and the fixed 64 Ki sent it to the mixed-span
// walk: 14.5 s for a 117.7 M-position contig.
uint32_t nu_suspects_for(uint32_t n) { return n / 64u > (1u << 16) ? n / 64u : (1u << 16); }
// Rounds the route may take before it gives way to the mixed-span walk: that walk is one serial chain per contig at
// ~0.07 us per position (79.8 ms for cfg4's 10^6-position contigs), a round is ~0.1 ms + what it sweeps again (at most
// a contig: 0.5 ms per 10^6 positions); the route may spend up to about half of what the walk would take.
constexpr uint32_t kNuMinRounds = 8, kNuMaxRoundsCap = 160;
uint32_t nu_round_budget(const qmcp_hip_ctx* c, const uint32_t* lengths, uint32_t n_contigs) {
    if (c->opt.near_uniform_rounds) return c->opt.near_uniform_rounds;
    uint32_t longest = 0;
    for (uint32_t k = 0; k < n_contigs; ++k) longest = lengths[k] > longest ? lengths[k] : longest;
    const double walk_ms = 0.07e-3 * (double)longest;
    const double round_ms = 0.1 + 0.5e-6 * (double)longest;
    const double r = 0.5 * walk_ms / round_ms;
    return r < kNuMinRounds ? kNuMinRounds : r > kNuMaxRoundsCap ? kNuMaxRoundsCap : (uint32_t)r;
}
// mean coverage / M below which the route is not tried: the shallower the data, the more exceptions are wanted and the
// longer the runs of used-up buckets (cfg4's reads with 1 % clipped, lab/near_uniform_depths.py, near-uniform /
// mixed-span ms: 12.5 x M 2.9 / 105; 6.3 x M 6.0 / 401; 4.7 x M 5.5 / 503; 3.75 x M 6.4 / 659; with 40 % of the reads:
// 5 x M 12.4 / 579; 3 x M 15.7 / 710; 2.1 x M: gives up after four sweeps, 648 / 627 -- runs of used-up buckets with
// neither an anchor nor a cut point --; 1.5 x M 41.9 / 600, from cut points).  Below 1.3 x M nearly every window of the
// mixed-span sweep has a real cut point and that sweep is quick.
// Second half of round 4: the gate is the SIGMA depth (spec_sigma_depth: how far above M the coverage sits, as the depth at
// which M = 50 sits as far), and the crossover with the mixed-span walk -- which real cut points make quick where the
// coverage comes near M often -- was measured (lab/near_uniform_long_shallow.py, two contigs of 40 M positions, 1 % clipped,
// near-uniform / mixed-span ms by sigma depth): 1.24 (M = 20 at 1.4 x M) 18.9 / 7.3; 1.30 (30, 1.4) 23.4 / 10.0; 1.35 (20,
// 1.6) 14.1 / 7.9; 1.37 (10, 2.0) 8.9 / 5.6; 1.40 (50, 1.4) 24.1 / 17.3; 1.44 (30, 1.6) 15.9 / 13.7 | 1.56 (20, 2.0) 10.4 /
// 16.1; 1.60 (50, 1.6) 16.7 / 33.9; 1.67 (10, 3.0) 10.0 / 22.3; 1.72 (30, 2.0) 12.5 / 22.7 -- the route is tried from 1.5.
constexpr double kNuMinDepth = 1.5;
// marks per stretch of a table: contig starts + at most kSweepWindowsOneSpan windows (a fixed 4 096 until round 4: a call of
// more than 3 328 contigs would have cleared past its array)
uint32_t nu_marks_words(uint32_t n_contigs) { return n_contigs + qmcp::kSweepWindowsOneSpan + 256u; }
uint32_t nu_cap_for(uint32_t n) {  // 128 slots per wave and pass (or tile): an eighth of the reads, on either producer
    const uint32_t a = qmcp::pm_exc_slots(n), b = qmcp::prepare_exc_slots(n);
    return a > b ? a : b;
}
int ensure_near_uniform(qmcp_hip_ctx* c, uint32_t n, uint32_t ltot, uint32_t n_contigs) {
    TRY(ensure(c, c->nu_exc, qmcp::nu_exc_bytes(nu_cap_for(n))));
    TRY(ensure(c, c->nu_nadj, ((size_t)ltot + 2) * sizeof(int32_t)));
    TRY(ensure(c, c->nu_ce, ((size_t)ltot + 4) * sizeof(uint32_t)));
    TRY(ensure(c, c->nu_state, 64 + (size_t)n_contigs * 20 + 16));
    TRY(ensure(c, c->nu_sus, qmcp::nu_suspect_bytes(nu_suspects_for(n)) + 4 * (size_t)nu_marks_words(n_contigs) * sizeof(uint32_t) + 2 * qmcp::nu_cells_bytes()));  // + marks per exact stretch, per speculative stretch, and dirty cells: two rounds' each
    TRY(ensure(c, c->nu_prev, ((size_t)ltot + 8) * sizeof(uint32_t)));  // sweeps in stretches: the round before's kept counts
    if (c->nu_ell != 0) {
        // (the route's sweep scratch depends on the span: known from the last call that took the route, so a second call
        //  of the shape grows nothing after its first launch)
        TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, c->nu_ell, n_contigs + 768)));
        TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, c->nu_ell, n_contigs + 768)));
        TRY(ensure(c, c->nu_ckpt, qmcp::sweep_ev_ckpt_bytes(ltot, c->nu_ell, n_contigs + 768)));
    }
    TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(ltot + 2) + 1) * sizeof(uint32_t) + 16));
    if (!c->h_nu) HIP_TRY(hipHostMalloc((void**)&c->h_nu, 8 * sizeof(uint32_t), hipHostMallocDefault));
    return QMCP_OK;
}
