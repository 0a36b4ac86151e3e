"""The range-ranked uniform path (calls of >= 4 Mi reads): one stable partition by position range,
per-range LDS counts, sweep, per-range ordered ranking against S(p).  The kept set must be the
S(p) lowest read indices of every start bucket -- bit-identical to the oracle -- whatever order the
LDS arbitrates colliding lanes in, and identical to what the radix-sort route produces."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_BIG = (1 << 22) + 12_345  # just above the threshold of the ranked path, not a tile multiple


def _uniform_reads(rng, n, L, span):
    s = rng.integers(0, L - span + 1, size=n, dtype=np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _solve_both_routes(solver, s, e, lengths, M, offs=None):
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    passes = solver.last_stats.sort_passes
    with solver.options(force_sort_route=1):
        sorted_route = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        sorted_passes = solver.last_stats.sort_passes
    return got, passes, sorted_route, sorted_passes


@pytest.mark.parametrize("M", [1, 3, 40])
def test_small_genome_pileups_collide_inside_a_step(pkg, oracle, solver, M):
    """4.2 M reads on 20 k positions: ~210 reads per start, 128-position ranges, so most lanes of a
    64-record step share positions and quotas run out in the middle of a step (slow path)"""
    rng = np.random.default_rng(100 + M)
    L, span = 20_000, 100
    s, e = _uniform_reads(rng, N_BIG, L, span)
    got, passes, sorted_route, sorted_passes = _solve_both_routes(solver, s, e, L, M)
    assert solver.last_stats.path == pkg.PATH_UNIFORM
    assert passes == 1 and sorted_passes == 2
    assert np.array_equal(got, sorted_route)
    assert np.array_equal(got, oracle.solve(s, e, L, M))


def test_last_range_partial_and_reads_at_the_last_position(pkg, oracle, solver):
    rng = np.random.default_rng(7)
    L, span = 3 * 32768 + 77, 150   # shift 9: 193 ranges, the last one 77 positions wide
    s, e = _uniform_reads(rng, N_BIG, L, span)
    s[:5000] = L - span              # pile on the very last start position
    e[:5000] = L - 1
    s[5000:9000] = 0
    e[5000:9000] = span - 1
    got, passes, sorted_route, _ = _solve_both_routes(solver, s, e, L, 25)
    assert passes == 1
    assert np.array_equal(got, sorted_route)
    assert np.array_equal(got, oracle.solve(s, e, L, 25))


def test_multi_contig_ranges_straddle_contig_borders(pkg, oracle, solver):
    rng = np.random.default_rng(11)
    lengths = np.array([70_001, 33_333, 250_000, 1_000], np.uint32)
    counts = np.array([1_500_000, 700_000, 1_994_304 + 12_345, 40_000])
    span = 120
    ss, ee = zip(*[_uniform_reads(rng, int(c), int(L), span) for c, L in zip(counts, lengths)])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    got, passes, sorted_route, _ = _solve_both_routes(solver, s, e, lengths, 60, offs)
    assert passes == 1 and solver.last_stats.n_contigs == 4
    assert np.array_equal(got, sorted_route)
    assert np.array_equal(got, oracle.solve(s, e, lengths, 60, contig_read_offsets=offs))


def test_one_heavy_range_falls_back_to_the_sort(pkg, oracle, solver):
    """a third of the reads start inside one 8 Ki-position range: its ranking would be one wave's
    walk over 1.4 M records, so the keep mask comes from the radix sort (3 passes), same answer"""
    rng = np.random.default_rng(13)
    L, span = 2_000_000, 150
    s, e = _uniform_reads(rng, N_BIG, L, span)
    hot = rng.integers(1_000_000, 1_004_000, size=N_BIG // 3, dtype=np.uint32)
    s[: hot.size] = hot
    e[: hot.size] = hot + np.uint32(span - 1)
    got = solver.solve(s, e, L, 30)
    assert solver.last_stats.sort_passes == 3
    assert np.array_equal(got, oracle.solve(s, e, L, 30))


def test_kept_count_and_validity_match(pkg, oracle, solver):
    rng = np.random.default_rng(17)
    L, span, M = 500_000, 151, 100
    s, e = _uniform_reads(rng, N_BIG, L, span)
    got = solver.solve(s, e, L, M)
    assert solver.last_stats.sort_passes == 1
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")[: s.size]
    assert int(bits.sum()) == solver.last_stats.n_kept
    in_cov = solver.coverage(s, e, L)
    out_cov = solver.coverage(s, e, L, keep_mask=got)
    assert oracle.is_out_cover_valid(in_cov, out_cov, M)
    assert np.array_equal(got, oracle.solve(s, e, L, M))


def test_speculation_is_abandoned_cleanly(pkg, oracle, solver):
    """The partition and bucket offsets of a large call are queued before the host has seen the span
    statistics.  An invalid read must still give QMCP_EREAD (with everything queued staying in
    bounds), mixed spans must still take the general route, and the context must stay usable."""
    rng = np.random.default_rng(23)
    L, span = 20_000, 100
    s, e = _uniform_reads(rng, N_BIG, L, span)
    bad_e = e.copy()
    bad_e[123_457] = L + 5_000_000          # end beyond the contig
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve(s, bad_e, L, 10)
    assert err.value.code == -2  # QMCP_EREAD
    bad_s = s.copy()
    bad_s[N_BIG - 3] = 4_000_000_000        # start far beyond everything (garbage key in the partition)
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve(bad_s, e, L, 10)
    assert err.value.code == -2  # QMCP_EREAD
    # mixed spans: a few reads one base longer
    e2 = e.copy()
    longer = rng.choice(N_BIG, size=1000, replace=False)
    e2[longer] = np.minimum(e2[longer] + 1, L - 1)
    got = solver.solve(s, e2, L, 7)
    assert solver.last_stats.path == pkg.PATH_GENERAL
    assert np.array_equal(got, oracle.solve(s, e2, L, 7))
    # and the uniform call right after is served by the ranked route again
    got = solver.solve(s, e, L, 7)
    assert solver.last_stats.sort_passes == 1
    assert np.array_equal(got, oracle.solve(s, e, L, 7))


@pytest.mark.parametrize("span", [1, 2, 63, 64, 65, 128, 129, 150, 192, 193, 256, 257, 300, 512])
def test_ranked_route_for_every_span_class(pkg, oracle, solver, span):
    """mid-size calls (>= 128 Ki reads) take the ranked route too: every register-count class of
    the seven-wave sweep (spans up to 256) and the single-wave sweep beyond, deep and shallow"""
    rng = np.random.default_rng(1000 + span)
    n = 140_000 + span
    for L, M in ((max(4 * span, 3_000), 30), (60_000 + span, 3)):
        s, e = _uniform_reads(rng, n, L, span)
        got = solver.solve(s, e, L, M)
        assert solver.last_stats.path == pkg.PATH_UNIFORM and solver.last_stats.sort_passes == 1
        assert np.array_equal(got, oracle.solve(s, e, L, M)), (span, L, M)


def test_ranked_route_many_small_contigs(pkg, oracle, solver):
    rng = np.random.default_rng(77)
    lengths = rng.integers(400, 9_000, size=40).astype(np.uint32)
    counts = rng.integers(1_000, 9_000, size=40)
    span = 90
    ss, ee = zip(*[_uniform_reads(rng, int(c), int(L), span) for c, L in zip(counts, lengths)])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    got = solver.solve(s, e, lengths, 12, contig_read_offsets=offs)
    assert solver.last_stats.sort_passes == 1 and solver.last_stats.n_contigs == 40
    assert np.array_equal(got, oracle.solve(s, e, lengths, 12, contig_read_offsets=offs))


def test_two_level_partition_beyond_256_ranges(pkg, oracle, solver):
    """genomes beyond 8.39 M positions need more than 256 ranges of 32 Ki: the reads go first into
    super-ranges, then every super-range into its ranges; contigs straddle both kinds of border"""
    rng = np.random.default_rng(31)
    lengths = np.array([9_000_001, 5_000_000, 8_388_608 + 5, 123], np.uint32)   # Ltot ~ 22.4 M
    counts = np.array([900_000, 480_000, 850_000, 0])                           # the tiny contig has no reads
    span = 150
    ss, ee = zip(*[_uniform_reads(rng, int(c), int(L), span) for c, L in zip(counts[:3], lengths[:3])])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    got, passes, sorted_route, _ = _solve_both_routes(solver, s, e, lengths, 6, offs)
    assert passes == 1 and solver.last_stats.n_contigs == 4
    assert np.array_equal(got, sorted_route)
    assert np.array_equal(got, oracle.solve(s, e, lengths, 6, contig_read_offsets=offs))


def test_two_level_partition_piled_up_super_range(pkg, oracle, solver):
    """one super-range holds almost everything, most super-ranges are empty"""
    rng = np.random.default_rng(37)
    L, span, n = 40_000_000, 101, 600_000
    s = rng.integers(20_000_000, 20_400_000, size=n, dtype=np.uint32)
    s[:2000] = rng.integers(0, L - span, size=2000, dtype=np.uint32)     # a few reads everywhere else
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    got = solver.solve(s, e, L, 9)
    assert np.array_equal(got, oracle.solve(s, e, L, 9))
