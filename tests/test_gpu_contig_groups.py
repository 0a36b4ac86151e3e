"""Contig groups: a deep multi-contig call is dealt to child contexts (one group of contigs each, on a
stream of its own) so that one group's selection chain runs beside the next groups' bandwidth-bound
stages.  Contigs are independent solves (the reference holds one contig per BamApi,
libs/bam-api/src/bam_api.cpp:422), so the keep mask must be bit-identical to the unsplit solve and to the
oracle -- whatever the group borders are (a contig need not start at a multiple of 64 reads: neighbouring
groups share a mask word), whichever route each group takes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _uniform_reads(rng, n, L, span):
    s = rng.integers(0, L - span + 1, size=n, dtype=np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _with_groups(groups, fn):
    old = os.environ.get("QMCP_HIP_GROUPS")
    os.environ["QMCP_HIP_GROUPS"] = str(groups)
    try:
        return fn()
    finally:
        if old is None:
            del os.environ["QMCP_HIP_GROUPS"]
        else:
            os.environ["QMCP_HIP_GROUPS"] = old


def _contigs(rng, counts, lengths, span):
    ss, ee = zip(*[_uniform_reads(rng, int(c), int(L), span) for c, L in zip(counts, lengths)])
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    return np.concatenate(ss), np.concatenate(ee), offs


@pytest.mark.parametrize("groups", [2, 3, 5])
def test_groups_share_mask_words_at_unaligned_contig_borders(pkg, oracle, solver, groups):
    """five deep contigs whose read counts are odd numbers: every group border falls inside a mask word"""
    rng = np.random.default_rng(40 + groups)
    lengths = np.array([40_000, 25_001, 33_333, 60_000, 9_000], np.uint32)
    counts = np.array([400_001, 250_013, 333_331, 600_007, 170_003])
    s, e, offs = _contigs(rng, counts, lengths, 150)
    whole = _with_groups(0, lambda: solver.solve(s, e, lengths, 30, contig_read_offsets=offs))
    assert solver.last_stats.contig_groups == 1
    got = _with_groups(groups, lambda: solver.solve(s, e, lengths, 30, contig_read_offsets=offs))
    st = solver.last_stats
    assert st.contig_groups == groups and st.n_contigs == 5 and st.n_reads == s.size
    assert st.path == pkg.PATH_UNIFORM and st.min_span == 150 and st.max_span == 150
    assert np.array_equal(got, whole)
    want = oracle.solve(s, e, lengths, 30, contig_read_offsets=offs)
    assert np.array_equal(got, want)
    assert st.n_kept == int(np.unpackbits(want.view(np.uint8)).sum())


def test_groups_that_disagree_on_the_read_length_are_solved_again_whole(pkg, oracle):
    """The first group's read length decides what is queued for the others without a host wait of their own;
    here the third contig has mixed read lengths (and the fourth is empty, the fifth too small for the ranked
    route): the check at collection notices, the call is solved again whole -- same mask as the oracle's --
    and the context stops splitting."""
    rng = np.random.default_rng(5)
    lengths = np.array([50_000, 30_000, 20_000, 1_000, 8_000], np.uint32)
    counts = np.array([300_001, 200_000, 150_000, 0, 20_001])
    ss, ee = [], []
    for c, (cnt, L) in enumerate(zip(counts, lengths)):
        if c == 2:   # mixed spans
            span = rng.integers(100, 151, size=int(cnt))
            st_ = (rng.random(int(cnt)) * (int(L) - span + 1)).astype(np.int64)
            ss.append(st_.astype(np.uint32)); ee.append((st_ + span - 1).astype(np.uint32))
        else:
            a, b = _uniform_reads(rng, int(cnt), int(L), 120)
            ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    want = oracle.solve(s, e, lengths, 40, contig_read_offsets=offs)
    with pkg.Solver(0) as sv:
        # warm the groups' contexts once (a group's first call of a shape waits for its own statistics)
        for _ in range(2):
            got = _with_groups(5, lambda: sv.solve(s, e, lengths, 40, contig_read_offsets=offs))
            st = sv.last_stats
            assert st.path == pkg.PATH_GENERAL and st.min_span == 100 and st.max_span == 150
            assert np.array_equal(got, want)
        assert st.contig_groups == 1   # solved again whole, and not split any more


def test_small_and_empty_groups_ride_along(pkg, oracle, solver):
    """an empty contig and one too small for the ranked route as groups of their own, all of one read length"""
    rng = np.random.default_rng(6)
    lengths = np.array([50_000, 30_000, 1_000, 8_000, 20_000], np.uint32)
    counts = np.array([300_001, 200_000, 0, 20_001, 150_000])
    s, e, offs = _contigs(rng, counts, lengths, 120)
    want = oracle.solve(s, e, lengths, 40, contig_read_offsets=offs)
    for _ in range(2):
        got = _with_groups(5, lambda: solver.solve(s, e, lengths, 40, contig_read_offsets=offs))
        assert solver.last_stats.contig_groups == 5
        assert np.array_equal(got, want)


def test_an_invalid_read_in_a_later_group_fails_the_call_and_the_context_lives_on(pkg, oracle, solver):
    rng = np.random.default_rng(9)
    lengths = np.array([30_000, 30_000, 30_000], np.uint32)
    counts = np.array([200_000, 200_000, 200_000])
    s, e, offs = _contigs(rng, counts, lengths, 100)
    bad_s, bad_e = s.copy(), e.copy()
    bad_s[450_000] = 40_000          # start beyond its contig (group 3 of 3)
    bad_e[450_000] = 40_099
    for _ in range(2):   # (the second time the later groups are queued without a host wait: caught at collection)
        with pytest.raises(pkg.QmcpError):
            _with_groups(3, lambda: solver.solve(bad_s, bad_e, lengths, 20, contig_read_offsets=offs))
    got = _with_groups(3, lambda: solver.solve(s, e, lengths, 20, contig_read_offsets=offs))
    assert solver.last_stats.contig_groups == 3
    assert np.array_equal(got, oracle.solve(s, e, lengths, 20, contig_read_offsets=offs))


def test_auto_split_of_a_deep_multi_contig_call_and_the_arena_is_sized_once(pkg, oracle):
    """17 M reads on four contigs at 25 x M: QMCP_HIP_GROUPS=auto deals it to two groups; a second call of
    the same shape grows no buffer after its first launch"""
    rng = np.random.default_rng(3)
    lengths = np.array([300_000, 200_000, 250_000, 250_000], np.uint32)
    counts = (lengths.astype(np.int64) * 17).astype(np.int64) + np.array([1, 3, 5, 7])
    s, e, offs = _contigs(rng, counts, lengths, 150)
    with pkg.Solver(0) as sv:
        got = _with_groups("auto", lambda: sv.solve(s, e, lengths, 100, contig_read_offsets=offs))
        assert sv.last_stats.contig_groups == 2
        again = _with_groups("auto", lambda: sv.solve(s, e, lengths, 100, contig_read_offsets=offs))
        assert sv.last_stats.contig_groups == 2 and sv.last_stats.arena_grown_mid_solve == 0
        assert np.array_equal(got, again)
        whole = sv.solve(s, e, lengths, 100, contig_read_offsets=offs)   # (not asked: solved whole)
        assert sv.last_stats.contig_groups == 1 and sv.last_stats.arena_grown_mid_solve == 0
        assert np.array_equal(got, whole)
    for c in range(4):   # oracle contig by contig (its memory is per-base)
        lo, hi = int(offs[c]), int(offs[c + 1])
        want = oracle.solve(s[lo:hi], e[lo:hi], int(lengths[c]), 100)
        bits = np.unpackbits(got.view(np.uint8), bitorder="little")[lo:hi]
        assert np.array_equal(bits, np.unpackbits(want.view(np.uint8), bitorder="little")[:hi - lo])


def test_the_mixed_route_sizes_its_arrays_up_front_after_the_first_mixed_call(pkg, oracle):
    """the first mixed-span call of a context grows the end-count arrays after the partition was queued (it
    did not know); every later call -- also a larger one -- has them sized before its first launch"""
    rng = np.random.default_rng(21)
    with pkg.Solver(0) as sv:
        for n, L in ((300_000, 40_000), (600_000, 90_000)):
            span = rng.integers(100, 151, size=n)
            st_ = (rng.random(n) * (L - span + 1)).astype(np.int64)
            s, e = st_.astype(np.uint32), (st_ + span - 1).astype(np.uint32)
            got = sv.solve(s, e, L, 30)
            assert sv.last_stats.path == pkg.PATH_GENERAL
            if n == 600_000:
                assert sv.last_stats.arena_grown_mid_solve == 0
            assert np.array_equal(got, oracle.solve(s, e, L, 30))
