"""Speculative stretch boundaries (csrc/kernels/sweep_segments.inc.hip): on data a few times deeper than M
there are hardly any cut points, but two sweeps of the same positions from different states agree from
some point on.  Windows without a cut therefore start a stretch with a run-in from the state of a cut
point; where it meets the stretch before it the two outputs are compared, and one disagreement anywhere
re-runs the exact sweep.  Whatever happens the kept set must be the oracle's, bit for bit."""

import numpy as np
import pytest

from forcing import forced

pytestmark = pytest.mark.gpu


def _uniform_contigs(rng, lengths, depth_in_m, M, span):
    """reads of one span, uniformly placed, mean coverage depth_in_m * M on every contig"""
    ss, ee, offs = [], [], [0]
    for L in lengths:
        n = int(depth_in_m * M * L / span)
        s = rng.integers(0, L - span + 1, size=n, dtype=np.uint32)
        ss.append(s)
        ee.append(s + np.uint32(span - 1))
        offs.append(offs[-1] + n)
    return np.concatenate(ss), np.concatenate(ee), np.asarray(offs, np.uint64), np.asarray(lengths, np.uint32)


@pytest.mark.parametrize("span,M,depth", [(150, 50, 2.0), (150, 30, 1.6), (100, 40, 2.4), (250, 60, 2.0), (40, 20, 2.0)])
def test_speculative_boundaries_hold_and_change_nothing(pkg, oracle, solver, span, M, depth):
    rng = np.random.default_rng(span + M)
    s, e, offs, lengths = _uniform_contigs(rng, [2_400_000, 700_000, 1_300_000], depth, M, span)
    with forced(solver, QMCP_HIP_SPEC=None, QMCP_HIP_SPEC_BURN=None):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.path == pkg.PATH_UNIFORM and st.spec_mismatches == 0, st.as_dict()
    if depth >= 2.0:   # (shallower: nearly every window has a real cut point)
        assert st.spec_boundaries >= 2, st.as_dict()
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want)
    with forced(solver, QMCP_HIP_SPEC="0"):
        plain = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        assert solver.last_stats.spec_boundaries == 0
    assert np.array_equal(plain, want)


def test_a_run_in_that_is_too_short_is_noticed_and_the_exact_sweep_takes_over(pkg, oracle, solver):
    rng = np.random.default_rng(8)
    s, e, offs, lengths = _uniform_contigs(rng, [3_000_000], 2.0, 50, 150)
    want = oracle.solve(s, e, lengths, 50, contig_read_offsets=offs)
    with forced(solver, QMCP_HIP_SPEC_BURN="4"):
        got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.spec_boundaries > 0 and st.spec_mismatches > 0, st.as_dict()
    assert np.array_equal(got, want)
    assert st.spec_retry_mismatches > 0          # three times four blocks is not enough either: the exact sweep ran
    with forced(solver, QMCP_HIP_SPEC_BURN="2"):
        assert np.array_equal(solver.solve(s, e, lengths, 50, contig_read_offsets=offs), want)
    # 64 blocks: a good part of the boundaries disagree; the second tier (192 blocks) settles it
    with forced(solver, QMCP_HIP_SPEC_BURN="64"):
        got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.spec_mismatches > 0 and st.spec_retry_mismatches == 0, st.as_dict()
    assert np.array_equal(got, want)


def test_speculation_beside_real_cut_points_and_gaps(pkg, oracle, solver):
    """stretches of 2 x M separated by shallow runs and holes: exact and speculative boundaries in one table"""
    rng = np.random.default_rng(99)
    L, M, span = 4_000_000, 40, 150
    parts = []
    for lo, hi, depth in ((0, 900_000, 2.0), (900_000, 1_000_000, 0.5), (1_000_000, 2_600_000, 2.2),
                          (2_700_000, 3_999_000, 1.9)):
        n = int(depth * M * (hi - lo) / span)
        parts.append(rng.integers(lo, hi - span + 1, size=n, dtype=np.uint32))
    s = np.concatenate(parts)
    s = s[rng.permutation(s.size)]
    e = s + np.uint32(span - 1)
    with forced(solver, QMCP_HIP_SPEC="1"):
        got = solver.solve(s, e, L, M)
        st = solver.last_stats
    assert st.spec_boundaries > 0
    assert np.array_equal(got, oracle.solve(s, e, L, M))


def test_deeper_data_forced_to_speculate_falls_back_exactly(pkg, oracle, solver):
    """at 6 x M the sweep does not forget its start within the run-in: every boundary disagrees"""
    rng = np.random.default_rng(5)
    s, e, offs, lengths = _uniform_contigs(rng, [1_500_000], 6.0, 20, 150)
    with forced(solver, QMCP_HIP_SPEC="1", QMCP_HIP_SPEC_BURN="64"):
        got = solver.solve(s, e, lengths, 20, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.spec_boundaries > 0 and st.spec_mismatches > 0
    assert np.array_equal(got, oracle.solve(s, e, lengths, 20, contig_read_offsets=offs))


def _mixed_contigs(rng, lengths, depth_in_m, M, lo, hi):
    ss, ee, offs = [], [], [0]
    for L in lengths:
        n = int(depth_in_m * M * L / ((lo + hi) / 2))
        span = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
        s = (rng.random(n) * (L - span + 1)).astype(np.uint32)
        ss.append(s)
        ee.append(s + span - 1)
        offs.append(offs[-1] + n)
    return np.concatenate(ss), np.concatenate(ee), np.asarray(offs, np.uint64), np.asarray(lengths, np.uint32)


@pytest.mark.parametrize("lo,hi,M,depth", [(100, 150, 40, 2.0), (140, 151, 30, 1.8), (200, 250, 50, 1.9), (30, 60, 25, 2.0)])
def test_mixed_spans_speculate_too(pkg, oracle, solver, lo, hi, M, depth):
    """the register-resident event sweep: the state compared at a boundary is how many reads every bucket
    that is still alive has given so far"""
    rng = np.random.default_rng(lo * 1000 + hi)
    s, e, offs, lengths = _mixed_contigs(rng, [2_600_000, 900_000], depth, M, lo, hi)
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and st.spec_boundaries >= 2 and st.spec_mismatches == 0, st.as_dict()
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want)
    with forced(solver, QMCP_HIP_SPEC="0"):
        assert np.array_equal(solver.solve(s, e, lengths, M, contig_read_offsets=offs), want)
        assert solver.last_stats.spec_boundaries == 0


def test_mixed_spans_with_a_run_in_that_is_too_short(pkg, oracle, solver):
    rng = np.random.default_rng(17)
    s, e, offs, lengths = _mixed_contigs(rng, [2_000_000], 2.0, 40, 100, 150)
    want = oracle.solve(s, e, lengths, 40, contig_read_offsets=offs)
    for burn in ("3", "16"):
        with forced(solver, QMCP_HIP_SPEC_BURN=burn):
            got = solver.solve(s, e, lengths, 40, contig_read_offsets=offs)
            st = solver.last_stats
        assert st.spec_boundaries > 0
        assert np.array_equal(got, want), (burn, st.as_dict())



def test_a_deep_island_in_a_shallow_genome_is_swept_again_alone(pkg, oracle, solver):
    """depth 2.2 x M, but one long region is 12 x M deep: the boundaries inside it disagree (the sweep
    does not forget its start there), which marks that part -- between the cut points around it -- for the
    later tiers, and only that part; the rest keeps its speculative stretches"""
    rng = np.random.default_rng(123)
    L, M, span = 6_000_000, 30, 150
    parts = [rng.integers(0, L - span + 1, size=int(2.2 * M * L / span), dtype=np.uint32)]            # 2.2 x M everywhere
    lo, hi = 2_000_000, 3_200_000
    parts.append(rng.integers(lo, hi - span + 1, size=int(10.0 * M * (hi - lo) / span), dtype=np.uint32))  # + 10 x M there
    s = np.concatenate(parts)
    s = s[rng.permutation(s.size)]
    e = s + np.uint32(span - 1)
    with forced(solver, QMCP_HIP_SPEC="1", QMCP_HIP_SPEC_BURN="160"):
        got = solver.solve(s, e, L, M)
        st = solver.last_stats
    # boundaries outside the island hold, those inside disagree in both speculative tiers
    assert st.spec_boundaries > 20 and 0 < st.spec_mismatches < st.spec_boundaries and st.spec_retry_mismatches > 0, st.as_dict()
    assert np.array_equal(got, oracle.solve(s, e, L, M))


def test_two_speculative_solves_in_flight_on_two_contexts(pkg, oracle):
    """the two-phase entry (qmcp_hip_solve_device_begin / _end) with every tier's launches queued and gated on
    the device: two contexts, two different problems, collected in the other order"""
    import torch
    rng = np.random.default_rng(4)
    probs = []
    for M in (50, 35):
        s, e, offs, lengths = _uniform_contigs(rng, [1_900_000, 800_000], 2.0, M, 150)
        probs.append((s, e, offs, lengths, M))
    with pkg.Solver(0) as a, pkg.Solver(0) as b:
        dev, masks = [], []
        for sv, (s, e, offs, lengths, M) in zip((a, b), probs):
            d_s = torch.from_numpy(s.view(np.int32)).cuda()
            d_e = torch.from_numpy(e.view(np.int32)).cuda()
            d_m = torch.zeros(pkg.mask_words(s.size), dtype=torch.int64, device="cuda")
            dev.append((d_s, d_e))
            masks.append(d_m)
            sv.solve_device_begin(d_s.data_ptr(), d_e.data_ptr(), s.size, lengths, M, d_m.data_ptr(), contig_read_offsets=offs)
        st_b, st_a = b.solve_end(), a.solve_end()
        for st, d_m, (s, e, offs, lengths, M) in zip((st_a, st_b), masks, probs):
            assert st.spec_boundaries >= 2 and st.spec_mismatches == 0, st.as_dict()
            assert np.array_equal(d_m.cpu().numpy().view(np.uint64), oracle.solve(s, e, lengths, M, contig_read_offsets=offs))


@pytest.mark.parametrize("depth,M", [(1.0, 30), (2.0, 20)])
def test_mixed_spans_with_more_than_a_thousand_stretches(pkg, oracle, solver, depth, M):
    """the mixed-span route's table holds up to 3 840 windows (a thread of the table kernel then looks after
    several candidates): shallow data full of real cut points, and 2 x M with speculative boundaries between them"""
    rng = np.random.default_rng(int(depth * 10) + M)
    s, e, offs, lengths = _mixed_contigs(rng, [11_000_000, 5_000_000], depth, M, 100, 150)
    # (at 2 x M the run-in is forced short, so that boundaries are close enough together for a table this
    #  large on a genome this small; those that then disagree are settled by the later tiers)
    with forced(solver, QMCP_HIP_SPEC_BURN="20" if depth >= 2.0 else None):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and st.sweep_stretches > 1024, st.as_dict()
    if depth >= 2.0:
        assert st.spec_boundaries > 500, st.as_dict()
    assert np.array_equal(got, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))


@pytest.mark.parametrize("M,depth", [(12, 5.0), (8, 7.5), (6, 10.0)])
def test_depths_between_4_and_11_times_m_are_swept_as_stretches_too(pkg, oracle, solver, M, depth):
    """round 2 swept such data as one chain per contig (it had measured the forgetting only up to 4 x M); measured in
    round 3 (lab/spec_depth_gap.py) the sweep forgets its start within about a thousand blocks up to 10 x M, so
    boundaries are speculated on at every depth the general-form sweep takes; whatever they do, the kept set is the
    oracle's"""
    rng = np.random.default_rng(1000 + M)
    s, e, offs, lengths = _uniform_contigs(rng, [3_200_000, 900_000], depth, M, 150)
    with forced(solver, QMCP_HIP_SPEC=None, QMCP_HIP_SPEC_BURN=None):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        st = solver.last_stats
    assert st.path == pkg.PATH_UNIFORM and st.spec_boundaries >= 1, st.as_dict()
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want)
    with forced(solver, QMCP_HIP_SPEC="0"):
        plain = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        assert solver.last_stats.spec_boundaries == 0
    assert np.array_equal(plain, want)


def test_one_dominant_read_length_does_not_speculate_a_broad_mix_does(pkg, oracle, solver):
    """the mixed-span route samples the spans before it queues anything: one dominant read length with a few LONGER reads,
    5 x M deep at M = 60 (as many standard deviations above M as M = 50 at 5.7 x M: a run-in of 1 536 blocks) on a contig
    that does not hold a dozen such run-ins, forgets its state too slowly for stretches to pay -- it does not speculate;
    a broad mix of lengths at the same depth does, and its boundaries hold; the oracle's mask either way"""
    rng = np.random.default_rng(5)
    L, M, span = 2_000_000, 60, 150
    n = int(5 * M * L / span)
    lengths = np.array([L], np.uint32)
    s = rng.integers(0, L - span - 8, size=n, dtype=np.uint32)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    j = rng.choice(n, size=n // 200, replace=False)
    e[j] += rng.integers(1, 6, size=j.size).astype(np.uint32)          # deletions: longer spans
    got = solver.solve(s, e, lengths, M)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and np.array_equal(got, oracle.solve(s, e, lengths, M)), st.as_dict()
    assert st.spec_boundaries == 0, st.as_dict()
    sp = rng.integers(100, 151, size=n)
    s2 = (rng.random(n) * (L - sp + 1)).astype(np.uint32)
    e2 = (s2 + sp - 1).astype(np.uint32)
    got = solver.solve(s2, e2, lengths, M)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and np.array_equal(got, oracle.solve(s2, e2, lengths, M)), st.as_dict()
    assert st.spec_boundaries >= 2 and st.spec_mismatches == 0, st.as_dict()


def test_a_shallow_genome_with_longer_reads_speculates_on_the_mixed_span_route(pkg, oracle, solver):
    """one dominant read length plus reads LONGER than it (deletions) leaves the near-uniform route; on a long shallow
    contig (2 x M, M = 50: cfg5's depth -- five standard deviations above M, a handful of real cut points) the mixed-span walk must not be
    one chain per contig: its speculative boundaries are used (the rule is how deep the data is in standard deviations,
    not that one length dominates) and hold; the oracle's mask"""
    rng = np.random.default_rng(77)
    L, M, span = 12_000_000, 50, 150
    n = int(2.0 * M * L / span)
    lengths = np.array([L], np.uint32)
    s = rng.integers(0, L - span - 24, size=n, dtype=np.uint32)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    j = rng.choice(n, size=n // 200, replace=False)
    e[j] += rng.integers(1, 21, size=j.size).astype(np.uint32)          # deletions: longer spans
    k = rng.choice(n, size=n // 100, replace=False)
    s[k] += rng.integers(1, 40, size=k.size).astype(np.uint32)          # and soft clips: shorter ones
    got = solver.solve(s, e, lengths, M)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and st.near_uniform_giveup == 2, st.as_dict()   # (longer reads)
    assert np.array_equal(got, oracle.solve(s, e, lengths, M)), st.as_dict()
    assert st.spec_boundaries >= 10 and st.spec_mismatches == 0 and st.sweep_stretches >= 12, st.as_dict()
