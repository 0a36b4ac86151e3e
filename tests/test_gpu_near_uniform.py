"""The near-uniform route (csrc/kernels/near_uniform.inc.hip): deep calls whose reads have one dominant span and a
small share of shorter ones (soft clips, insertions -- BamApi takes a read's span from its CIGAR,
libs/bam-api/src/read.cpp:11-13) keep the one-span machinery: the shorter reads are listed as exceptions, the sweep
runs over the regular reads, and every exception the greedy would take is found from the sweep's counts and selected,
one event per contig and round.  The keep mask must be the oracle's bit for bit whichever route a call ends on;
tests/test_near_uniform_model.py checks the scheme itself on the CPU."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(rng, n, L, span, fraction, max_clip, longer=0):
    s = rng.integers(0, L - span + 1, size=n).astype(np.int64)
    e = s + span - 1
    pick = rng.random(n) < fraction
    clip = rng.integers(1, max_clip + 1, size=n)
    front = rng.random(n) < 0.5
    s = np.where(pick & front, s + clip, s)
    e = np.where(pick & ~front, e - clip, e)
    if longer:
        j = rng.choice(n, size=longer, replace=False)
        ok = e[j] + 2 < L
        e[j[ok]] += 2
    return s.astype(np.uint32), e.astype(np.uint32)


def _contigs(rng, lengths, counts, span, fraction, max_clip, longer=0):
    ss, ee = [], []
    for L, k in zip(lengths, counts):
        a, b = _reads(rng, int(k), int(L), span, fraction, max_clip, longer)
        ss.append(a); ee.append(b)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    return np.concatenate(ss), np.concatenate(ee), offs


CASES = [
    # lengths, reads per contig, span, M, fraction, max clip
    ([60_000], [500_000], 150, 100, 0.01, 50),
    ([40_000, 25_000, 70_001, 12_345], [330_000, 210_000, 580_000, 100_000], 150, 100, 0.02, 50),
    ([50_000], [400_000], 100, 60, 0.01, 99),           # clips down to one base
    ([80_000, 80_000], [420_000, 900_000], 151, 64, 0.03, 20),   # one contig twice as deep as the other
    ([30_000, 30_000, 30_000], [260_000, 0, 260_000], 150, 100, 0.005, 30),   # a contig without reads
]


@pytest.mark.parametrize("lengths,counts,span,M,fraction,max_clip", CASES)
def test_near_uniform_equals_the_oracle(pkg, oracle, solver, lengths, counts, span, M, fraction, max_clip):
    rng = np.random.default_rng(sum(counts) % 9973 + span)
    lengths = np.array(lengths, np.uint32)
    s, e, offs = _contigs(rng, lengths, counts, span, fraction, max_clip)
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = solver.last_stats
    want = oracle.solve(s, e, lengths, M, offs)
    assert np.array_equal(got, want), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM, st.as_dict()
    assert st.near_uniform_exceptions == int(((e - s + 1) != span).sum())
    assert st.near_uniform_rounds >= 1 and st.n_kept == int(sum(bin(int(w)).count("1") for w in want))
    # the same call again: the head filters on the remembered span at once -- same mask
    again = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(again, want) and solver.last_stats.path == pkg.PATH_NEAR_UNIFORM
    # ... and nothing of the arena grows after the second call's first launch (the route's sweep scratch is sized with the
    # remembered span, in the head)
    assert solver.last_stats.arena_grown_mid_solve == 0 and solver.last_stats.near_uniform_giveup == 0


def test_route_switches_between_calls(pkg, oracle, solver):
    """near-uniform, then one span that differs from the remembered one (the filtered head listed every read as an
    exception: its stages run again), then the remembered span with no exception at all, then a mix without a dominant
    span -- every mask the oracle's"""
    rng = np.random.default_rng(5)
    L, n = 50_000, 400_000
    lengths = np.array([L], np.uint32)
    s, e, _ = _contigs(rng, [L], [n], 150, 0.01, 40)
    assert np.array_equal(solver.solve(s, e, lengths, 100), oracle.solve(s, e, lengths, 100))
    assert solver.last_stats.path == pkg.PATH_NEAR_UNIFORM
    s2, e2, _ = _contigs(rng, [L], [n], 120, 0.0, 1)
    assert np.array_equal(solver.solve(s2, e2, lengths, 100), oracle.solve(s2, e2, lengths, 100))
    assert solver.last_stats.path == pkg.PATH_UNIFORM
    assert np.array_equal(solver.solve(s, e, lengths, 100), oracle.solve(s, e, lengths, 100))
    assert solver.last_stats.path == pkg.PATH_NEAR_UNIFORM
    s3, e3, _ = _contigs(rng, [L], [n], 150, 0.0, 1)
    assert np.array_equal(solver.solve(s3, e3, lengths, 100), oracle.solve(s3, e3, lengths, 100))
    assert solver.last_stats.path == pkg.PATH_UNIFORM
    span = rng.integers(100, 151, size=n)
    s4 = rng.integers(0, L - 150, size=n).astype(np.uint32)
    e4 = (s4 + span - 1).astype(np.uint32)
    assert np.array_equal(solver.solve(s4, e4, lengths, 100), oracle.solve(s4, e4, lengths, 100))
    assert solver.last_stats.path == pkg.PATH_GENERAL


def test_gives_way_to_the_mixed_route(pkg, oracle, solver):
    """what the route does not model goes the mixed-span way, same mask: reads LONGER than the dominant span, more
    exceptions than a sixteenth of the reads, data too shallow for it"""
    rng = np.random.default_rng(11)
    L = 40_000
    lengths = np.array([L], np.uint32)
    # (give-up reasons: include/qmcp_hip.h QMCP_NU_GIVEUP_*: 2 longer reads, 3 too many exceptions, 1 not tried)
    for kwargs, M, n, why, L in ((dict(fraction=0.01, max_clip=30, longer=5), 100, 330_000, 2, 40_000),
                                 (dict(fraction=0.2, max_clip=30), 100, 330_000, 3, 40_000),
                                 # 1.25 x M at M = 10: nearly every window has a real cut point (M = 400 at 1.27 x M, this
                                 # case until round 4, is as far above M in standard deviations as M = 50 at 1.8 x M and
                                 # takes the route now)
                                 (dict(fraction=0.01, max_clip=30), 10, 250_000, 1, 3_000_000)):
        lengths = np.array([L], np.uint32)
        s, e, _ = _contigs(rng, [L], [n], 150, **kwargs)
        got = solver.solve(s, e, lengths, M)
        assert np.array_equal(got, oracle.solve(s, e, lengths, M)), (kwargs, M, solver.last_stats.as_dict())
        assert solver.last_stats.path == pkg.PATH_GENERAL, (kwargs, M, solver.last_stats.as_dict())
        assert solver.last_stats.near_uniform_giveup == why, (kwargs, solver.last_stats.as_dict())


def test_many_wanted_exceptions(pkg, oracle, solver):
    """5 % exceptions at a moderate depth: dozens are wanted, one round each per contig -- or the route gives up
    at its budget; either way the oracle's mask"""
    rng = np.random.default_rng(17)
    lengths = np.array([30_000, 30_000], np.uint32)
    s, e, offs = _contigs(rng, lengths, [200_000, 200_000], 150, 0.05, 50)
    got = solver.solve(s, e, lengths, 100, contig_read_offsets=offs)
    st = solver.last_stats
    assert np.array_equal(got, oracle.solve(s, e, lengths, 100, offs)), st.as_dict()
    assert st.path in (pkg.PATH_NEAR_UNIFORM, pkg.PATH_GENERAL)


def test_shallow_data_starts_from_cut_points(pkg, oracle, solver):
    """1.5 - 2 x M: runs of used-up buckets longer than a span; the replays start from cut points (cov_all <= M) where
    there is no anchor -- or the route gives way; the oracle's mask either way"""
    rng = np.random.default_rng(29)
    lengths = np.array([60_000, 45_000], np.uint32)
    for M in (250, 330):
        s, e, offs = _contigs(rng, lengths, [300_000, 230_000], 150, 0.02, 40)
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        st = solver.last_stats
        assert np.array_equal(got, oracle.solve(s, e, lengths, M, offs)), (M, st.as_dict())
        assert st.path in (pkg.PATH_NEAR_UNIFORM, pkg.PATH_GENERAL)


def test_env_switch_off(pkg, oracle, solver):
    rng = np.random.default_rng(23)
    lengths = np.array([60_000], np.uint32)
    s, e, _ = _contigs(rng, [60_000], [500_000], 150, 0.01, 50)
    solver.set_options(near_uniform=-1)
    try:
        got = solver.solve(s, e, lengths, 100)
    finally:
        solver.set_options()
    assert solver.last_stats.path == pkg.PATH_GENERAL
    assert np.array_equal(got, oracle.solve(s, e, lengths, 100))


def test_range_major_form_and_long_genomes(pkg, oracle, solver):
    """the same route behind the range-major producers (k_prepare lists the exceptions, the partition leaves them out):
    forced with QMCP_HIP_PM=0 on a one-level genome, and taken by itself on a genome beyond 8.39 M positions (two
    partition levels) -- the oracle's mask, on the near-uniform route"""
    rng = np.random.default_rng(41)
    lengths = np.array([50_000, 61_000], np.uint32)
    s, e, offs = _contigs(rng, lengths, [400_000, 500_000], 150, 0.02, 40)
    solver.set_options(pass_major=-1)
    try:
        got = solver.solve(s, e, lengths, 100, contig_read_offsets=offs)
        st = solver.last_stats
    finally:
        solver.set_options()
    assert np.array_equal(got, oracle.solve(s, e, lengths, 100, offs)), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_exceptions == int(((e - s + 1) != 150).sum()), st.as_dict()
    # two levels: 9.2 M positions in three contigs, 12 x M deep at M = 12
    lengths = np.array([4_000_000, 3_000_000, 2_200_000], np.uint32)
    counts = [int(12 * 12 * int(L) / 150) for L in lengths]
    s, e, offs = _contigs(rng, lengths, counts, 150, 0.01, 40)
    got = solver.solve(s, e, lengths, 12, contig_read_offsets=offs)
    st = solver.last_stats
    want = np.concatenate([np.unpackbits(oracle.solve(s[int(offs[c]):int(offs[c + 1])], e[int(offs[c]):int(offs[c + 1])], int(lengths[c]), 12).view(np.uint8), bitorder="little")[:counts[c]] for c in range(3)])
    assert np.array_equal(np.unpackbits(got.view(np.uint8), bitorder="little")[:want.size], want), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM, st.as_dict()


def test_contigs_shorter_than_the_dominant_span(pkg, oracle, solver):
    """a contig shorter than the dominant span holds exceptions only -- no regular read fits -- deeper than M: the
    replays there start from the contig's first position with every bucket empty, and the selection among the
    exceptions is the greedy's own (largest end, largest start, smallest index)"""
    rng = np.random.default_rng(61)
    lengths = np.array([50_000, 90, 40_000, 149], np.uint32)
    big = [420_000, 330_000]
    parts_s, parts_e, counts = [], [], []
    for L, k in ((50_000, big[0]), (90, 400), (40_000, big[1]), (149, 700)):
        if L >= 1000:
            a, b = _reads(rng, k, L, 150, 0.01, 40)
        else:
            span = rng.integers(1, L + 1, size=k)
            a = (rng.random(k) * (L - span + 1)).astype(np.int64)
            b = a + span - 1
            a, b = a.astype(np.uint32), b.astype(np.uint32)
        parts_s.append(a); parts_e.append(b); counts.append(k)
    s, e = np.concatenate(parts_s), np.concatenate(parts_e)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    for M in (100, 30):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        st = solver.last_stats
        assert np.array_equal(got, oracle.solve(s, e, lengths, M, offs)), (M, st.as_dict())
        assert st.path in (pkg.PATH_NEAR_UNIFORM, pkg.PATH_GENERAL)


def test_clipped_reads_bunched_in_read_order(pkg, oracle, solver):
    """a coordinate-sorted file: around a breakpoint nearly every read is clipped, and those reads are NEIGHBOURS in read
    order -- far more than the 128 list slots of the wave that meets them; the rest goes to the list's overflow region
    and the route carries on (it used to give way)"""
    rng = np.random.default_rng(73)
    L, n = 60_000, 500_000
    s = np.sort(rng.integers(0, L - 150 + 1, size=n)).astype(np.int64)      # sorted by position, as in a sorted BAM
    e = s + 149
    hot = (s > 30_000) & (s < 30_150)                                        # ~1 250 consecutive reads
    clip = rng.integers(1, 60, size=n)
    front = rng.random(n) < 0.5
    pick = hot & (rng.random(n) < 0.9)
    pick |= rng.random(n) < 0.002
    s = np.where(pick & front, s + clip, s)
    e = np.where(pick & ~front, e - clip, e)
    s, e = s.astype(np.uint32), e.astype(np.uint32)
    lengths = np.array([L], np.uint32)
    got = solver.solve(s, e, lengths, 100)
    st = solver.last_stats
    assert np.array_equal(got, oracle.solve(s, e, lengths, 100)), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_exceptions == int(pick.sum()), st.as_dict()


def test_two_contexts_in_flight(pkg, oracle):
    """the two-phase entry (begin / end) on two contexts, as a pipelined caller uses it: the route's look-ups happen
    inside begin(); masks as from the blocking entry"""
    import torch
    rng = np.random.default_rng(83)
    lengths = np.array([50_000, 40_000], np.uint32)
    s, e, offs = _contigs(rng, lengths, [420_000, 330_000], 150, 0.01, 40)
    want = oracle.solve(s, e, lengths, 100, offs)
    d_s = torch.from_numpy(s.view(np.int32)).cuda()
    d_e = torch.from_numpy(e.view(np.int32)).cuda()
    masks = [torch.zeros(pkg.mask_words(s.size), dtype=torch.int64, device="cuda") for _ in range(2)]
    with pkg.Solver(0) as a, pkg.Solver(0) as b:
        sv = [a, b]
        for k in range(2):
            sv[k].solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, lengths, 100, masks[k].data_ptr(), contig_read_offsets=offs)
        for step in range(4):
            k = step & 1
            st = sv[k].solve_end()
            assert st.path == pkg.PATH_NEAR_UNIFORM, st.as_dict()
            assert np.array_equal(masks[k].cpu().numpy().view(np.uint64), want)
            masks[k].zero_()
            sv[k].solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, lengths, 100, masks[k].data_ptr(), contig_read_offsets=offs)
        for k in range(2):
            sv[k].solve_end()
            assert np.array_equal(masks[k].cpu().numpy().view(np.uint64), want)


@pytest.mark.parametrize("L,depth,M,fraction", [(400_000, 1.5, 100, 0.01), (800_000, 2.4, 40, 0.02), (3_000_000, 4.0, 20, 0.01),
                                                (1_500_000, 8.0, 12, 0.01)])
def test_shallow_contigs_are_swept_in_stretches(pkg, oracle, solver, L, depth, M, fraction):
    """below 11 x M the route's sweeps run in stretches (cut points where the coverage of ALL reads is <= M, speculative
    boundaries checked on the device) with the need moved by the selected exceptions: two contigs long enough for
    speculative boundaries at their depth -- the oracle's mask, on the near-uniform route, whole contigs never swept as
    one chain (stats.sweep_stretches counts them)"""
    rng = np.random.default_rng(int(L * depth) % 7919)
    lengths = np.array([L, L // 2 + 12_345], np.uint32)
    counts = [int(depth * M * int(x) / 150) for x in lengths]
    s, e, offs = _contigs(rng, lengths, counts, 150, fraction, 50)
    # (between 3.1 and 11 x M contigs of up to 2 M positions keep the chain unless the block-scan sweep is asked for)
    with solver.options(sweep=pkg.SWEEP_GENERAL if (depth >= 3.1 and L <= 2_000_000) else pkg.SWEEP_AUTO):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = solver.last_stats
    want = oracle.solve(s, e, lengths, M, offs)
    assert np.array_equal(got, want), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_giveup == 0, st.as_dict()
    assert st.sweep_stretches > 2 * st.near_uniform_rounds, st.as_dict()


def test_thousands_of_contigs_in_stretches(pkg, oracle, solver):
    """5 000 small contigs at a shallow depth with clipped reads: the per-stretch marks of the rounds are sized by the
    contig count (a fixed 4 096 words until round 4: a call of more than 3 328 contigs cleared past its array)"""
    rng = np.random.default_rng(5000)
    n_contigs = 5000
    lengths = rng.integers(1_500, 3_000, size=n_contigs).astype(np.uint32)
    counts = [int(2.0 * 30 * int(x) / 150) for x in lengths]
    s, e, offs = _contigs(rng, lengths, counts, 150, 0.01, 50)
    got = solver.solve(s, e, lengths, 30, contig_read_offsets=offs)
    st = solver.last_stats
    want = oracle.solve(s, e, lengths, 30, offs)
    assert np.array_equal(got, want), st.as_dict()
    assert st.n_contigs == n_contigs


def test_later_rounds_sweep_only_the_stretches_a_selection_reaches(pkg, oracle, solver):
    """two contigs of 6 M positions at 2 x M with 1 % clipped reads: the rounds after the first sweep only the
    speculative stretches whose sweep reads or writes something a selection changed (a boundary is compared where the
    stretch on either side of it was swept) -- fewer stretches in all than rounds x stretches of a whole sweep, and the
    oracle's mask"""
    rng = np.random.default_rng(606)
    lengths = np.array([6_000_000, 6_000_000], np.uint32)
    counts = [int(2.0 * 20 * int(x) / 150) for x in lengths]
    s, e, offs = _contigs(rng, lengths, counts, 150, 0.01, 50)
    got = solver.solve(s, e, lengths, 20, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_giveup == 0, st.as_dict()
    with solver.options(near_uniform_debug=2):   # (lab switch: every round sweeps all its exact marks cover)
        again = solver.solve(s, e, lengths, 20, contig_read_offsets=offs)
        st_all = solver.last_stats
    assert np.array_equal(got, again)
    assert st.near_uniform_rounds == st_all.near_uniform_rounds and st.near_uniform_rounds >= 3, st.as_dict()
    assert st.sweep_stretches < st_all.sweep_stretches, (st.as_dict(), st_all.as_dict())
    want = oracle.solve(s, e, lengths, 20, offs)
    assert np.array_equal(got, want), st.as_dict()


@pytest.mark.parametrize("run_in", [2, 4, 16])
def test_partial_rounds_with_boundaries_that_disagree(pkg, oracle, solver, run_in):
    """the rounds' partial sweeps with a run-in far too short: boundaries disagree in every round, also where a stretch
    that was swept again meets one that was left alone -- the later tiers take over (k_spec_verify marks the exact
    stretch), and the mask is still the oracle's"""
    rng = np.random.default_rng(1000 + run_in)
    lengths = np.array([3_000_000, 2_000_000], np.uint32)
    counts = [int(2.2 * 20 * int(x) / 150) for x in lengths]   # (a sigma depth of 1.66: the route is tried from 1.5)
    s, e, offs = _contigs(rng, lengths, counts, 150, 0.01, 50)
    with solver.options(speculation=1, speculation_run_in=run_in):
        got = solver.solve(s, e, lengths, 20, contig_read_offsets=offs)
        st = solver.last_stats
    want = oracle.solve(s, e, lengths, 20, offs)
    assert np.array_equal(got, want), st.as_dict()
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_rounds >= 2, st.as_dict()
    # (stats.spec_mismatches is the LAST sweep's count -- a partial one: lab/stress_near_uniform.py with QMCP_HIP_SPEC_BURN
    #  = 2, 8, 32 is where the disagreeing rounds are counted: profiles/r04_stress_near_uniform_shallow_burn*.log)


def test_rounds_queued_unseen_and_a_call_that_needs_more(pkg, oracle):
    """a shape that settled within a few rounds is remembered: the next call of it queues those rounds and the ranking
    without waiting for the device in between (qmcp_hip_solve_device_begin returns at once), and the state words are
    looked at when the solve is collected.  Same shape, other reads that need MORE rounds: the collection notices and
    solves the call again the blocking way -- every mask the oracle's"""
    rng = np.random.default_rng(2024)
    lengths = np.array([400_000], np.uint32)
    n, M = 1_200_000, 40          # 11.25 x M: the event-driven chain
    sa, ea, offs = _contigs(rng, lengths, [n], 150, 0.001, 10)
    sb, eb, _ = _contigs(rng, lengths, [n], 150, 0.05, 60)
    wa = oracle.solve(sa, ea, lengths, M, offs)
    wb = oracle.solve(sb, eb, lengths, M, offs)
    with pkg.Solver(0) as sv:
        rounds = []
        for s, e, w in ((sa, ea, wa), (sa, ea, wa), (sa, ea, wa), (sb, eb, wb), (sb, eb, wb), (sb, eb, wb), (sa, ea, wa)):
            got = sv.solve(s, e, lengths, M, contig_read_offsets=offs)
            st = sv.last_stats
            assert np.array_equal(got, w), (rounds, st.as_dict())
            assert st.path == pkg.PATH_NEAR_UNIFORM, st.as_dict()
            rounds.append(int(st.near_uniform_rounds))
        assert rounds[0] == rounds[1] == rounds[2] == rounds[6] and rounds[3] == rounds[4] == rounds[5], rounds
