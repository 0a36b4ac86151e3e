"""Cut points: a position with coverage <= M forces every read covering it to be kept
(SURVEY.md section 7), so the uniform-span sweep may restart behind it.  The solver splits contigs
there into stretches swept side by side (shallow or gapped data, where a contig's serial chain is the
whole cost).  The kept set must not change by a bit: segmented == unsegmented == oracle."""

import numpy as np
import pytest

from forcing import forced

pytestmark = pytest.mark.gpu


def _reads(rng, n, L, span, lo=0, hi=None):
    hi = L if hi is None else hi
    s = rng.integers(lo, hi - span + 1, size=n, dtype=np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _check(pkg, oracle, solver, s, e, lengths, M, offs=None, sweep=None, expect_split=True):
    with forced(solver, QMCP_HIP_CUTS="0", QMCP_HIP_SWEEP=sweep):
        whole = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        assert solver.last_stats.path == pkg.PATH_UNIFORM
        chains_whole = solver.last_stats.sweep_stretches
    with forced(solver, QMCP_HIP_CUTS="1", QMCP_HIP_SWEEP=sweep):
        split = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        chains_split = solver.last_stats.sweep_stretches
    assert np.array_equal(split, whole)
    assert np.array_equal(split, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))
    if expect_split:
        assert chains_split > chains_whole, (chains_split, chains_whole)
    return chains_whole, chains_split


@pytest.mark.parametrize("span", [40, 100, 150, 250, 300])
@pytest.mark.parametrize("sweep", ["gen", "fast"])
def test_shallow_coverage_around_the_cap(pkg, oracle, solver, span, sweep):
    """mean coverage ~ M: positions above and below the cap alternate every few bases"""
    rng = np.random.default_rng(span)
    L, M = 400_000, 12
    n = int(L * M / span)
    s, e = _reads(rng, n, L, span)
    whole, split = _check(pkg, oracle, solver, s, e, L, M, sweep=sweep)
    assert whole == 1 and split > 8


@pytest.mark.parametrize("M", [1, 5, 60])
def test_deep_islands_between_empty_gaps(pkg, oracle, solver, M):
    """exome-like: deep pile-ups separated by stretches nothing covers"""
    rng = np.random.default_rng(M)
    L, span = 600_000, 120
    parts = []
    for k in range(40):
        lo = k * 15_000 + int(rng.integers(0, 3000))
        parts.append(_reads(rng, 6000, L, span, lo, lo + 2_000 + int(rng.integers(0, 4000))))
    s = np.concatenate([p[0] for p in parts])
    e = np.concatenate([p[1] for p in parts])
    perm = rng.permutation(s.size)
    _check(pkg, oracle, solver, s[perm], e[perm], L, M)


def test_cut_right_behind_a_pileup_and_at_window_borders(pkg, oracle, solver):
    """coverage returns to the cap exactly where windows begin and one base after a pile-up ends"""
    rng = np.random.default_rng(3)
    L, span, M = 256 * 64 * 50, 50, 4          # 256 windows of 64 blocks
    win = 64 * 50
    s_list = []
    for w in range(0, 250, 3):                   # pile-ups ending on the last base of a window
        s_list.append(np.full(30, (w + 1) * win - span, dtype=np.uint32))
        s_list.append(np.full(2, (w + 1) * win, dtype=np.uint32))     # thin start of the next one
    s_list.append(rng.integers(0, L - span + 1, size=20_000, dtype=np.uint32))
    s = np.concatenate(s_list)
    s = s[rng.permutation(s.size)]
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    _check(pkg, oracle, solver, s, e, L, M)


def test_contigs_with_empty_ones_between(pkg, oracle, solver):
    rng = np.random.default_rng(11)
    span, M = 90, 8
    lengths = np.array([150_000, 0, 90, 260_000, 0, 0, 89, 120_000], dtype=np.uint32)
    parts, offs = [], [0]
    for L in lengths:
        n = 0 if L < span else (1 if L == span else int(L * M * 1.2 / span))
        parts.append(_reads(rng, n, int(L), span) if n else (np.zeros(0, np.uint32), np.zeros(0, np.uint32)))
        offs.append(offs[-1] + n)
    s = np.concatenate([p[0] for p in parts])
    e = np.concatenate([p[1] for p in parts])
    whole, split = _check(pkg, oracle, solver, s, e, lengths, M, offs=np.array(offs, dtype=np.uint64))
    assert whole == 5  # stretches = non-empty contigs when nothing is split


def test_deep_data_has_only_the_contig_ends(pkg, oracle, solver):
    """forced on where the look finds next to nothing: cuts only in the thin ends of the contig"""
    rng = np.random.default_rng(5)
    L, span, M = 300_000, 150, 20
    s, e = _reads(rng, 2_000_000, L, span)
    _check(pkg, oracle, solver, s, e, L, M, expect_split=False)


def test_default_is_on_for_shallow_data_and_off_for_deep(pkg, oracle, solver):
    rng = np.random.default_rng(8)
    L, span = 400_000, 100
    with forced(solver, QMCP_HIP_CUTS=None, QMCP_HIP_SWEEP=None):
        s, e = _reads(rng, 60_000, L, span)              # coverage 15 = 1.5 M
        got = solver.solve(s, e, L, 10)
        assert solver.last_stats.sweep_stretches > 1
        assert np.array_equal(got, oracle.solve(s, e, L, 10))
        s, e = _reads(rng, 4_000_000, L, span)           # coverage 1 000 = 25 M at M = 40, 10 reads a position: one chain
        solver.solve(s, e, L, 40)
        assert solver.last_stats.sweep_stretches == 1
        # many times M and yet SPARSE (M = 10: 2.5 reads a position at 25 x M) is shallow in standard deviations: the
        # general pipeline with cut points, as below 11 x M (round 4) -- and the oracle's mask
        s, e = _reads(rng, 1_000_000, L, span)
        got = solver.solve(s, e, L, 10)
        assert np.array_equal(got, oracle.solve(s, e, L, 10))


# ---------------------------------------------------------------- mixed spans: the event sweeps


def _mixed_reads(rng, n, L, lo_span, hi_span, lo=0, hi=None):
    hi = L if hi is None else hi
    span = rng.integers(lo_span, hi_span + 1, size=n).astype(np.int64)
    s = (lo + rng.random(n) * np.maximum(hi - lo - span + 1, 1)).astype(np.int64)
    s = np.minimum(s, L - span)
    return s.astype(np.uint32), (s + span - 1).astype(np.uint32)


def _check_mixed(pkg, oracle, solver, s, e, lengths, M, offs=None, lds=False, expect_split=True):
    with forced(solver, QMCP_HIP_CUTS="0", QMCP_HIP_GENERAL_LDS="1" if lds else None):
        whole = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        assert solver.last_stats.path == pkg.PATH_GENERAL
        chains_whole = solver.last_stats.sweep_stretches
    with forced(solver, QMCP_HIP_CUTS="1", QMCP_HIP_GENERAL_LDS="1" if lds else None):
        split = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        chains_split = solver.last_stats.sweep_stretches
    assert np.array_equal(split, whole)
    assert np.array_equal(split, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))
    if expect_split:
        assert chains_split > chains_whole, (chains_split, chains_whole)
    return chains_whole, chains_split


@pytest.mark.parametrize("spans,lds", [((80, 150), False), ((20, 60), False), ((100, 400), False),
                                       ((80, 150), True), ((300, 1500), False), ((5000, 9000), False)])
def test_mixed_spans_shallow_coverage(pkg, oracle, solver, spans, lds):
    """register-resident, LDS-cached and plain event sweeps, each started behind cut points with the
    coverage (and expiry) of the reads that reach over from before"""
    rng = np.random.default_rng(spans[0])
    lo, hi = spans
    L, M = 64 * hi * 12, 9
    n = int(L * M * 1.1 / ((lo + hi) / 2))
    s, e = _mixed_reads(rng, n, L, lo, hi)
    whole, split = _check_mixed(pkg, oracle, solver, s, e, L, M, lds=lds)
    assert whole == 1 and split > 4


def test_mixed_spans_islands_and_contigs(pkg, oracle, solver):
    rng = np.random.default_rng(21)
    M = 6
    lengths = np.array([200_000, 0, 140, 350_000], dtype=np.uint32)
    parts, offs = [], [0]
    for L in lengths:
        L = int(L)
        if L == 0:
            ps = []
        elif L < 1000:
            ps = [_mixed_reads(rng, 30, L, 60, 140)]
        else:
            ps = [_mixed_reads(rng, 900, L, 60, 140, lo, lo + 2500) for lo in range(0, L - 3000, 9000)]
        cnt = sum(p[0].size for p in ps)
        parts += ps
        offs.append(offs[-1] + cnt)
    s = np.concatenate([p[0] for p in parts])
    e = np.concatenate([p[1] for p in parts])
    # shuffle inside each contig
    for a, b in zip(offs[:-1], offs[1:]):
        perm = rng.permutation(b - a) + a
        s[a:b], e[a:b] = s[perm], e[perm]
    whole, split = _check_mixed(pkg, oracle, solver, s, e, lengths, M, offs=np.array(offs, dtype=np.uint64))
    assert whole == 3


def test_mixed_spans_long_reads_reach_over_several_cuts(pkg, oracle, solver):
    """a few reads far longer than the rest keep covering across many cut points"""
    rng = np.random.default_rng(4)
    L, M = 64 * 400 * 20, 3
    s1, e1 = _mixed_reads(rng, 40_000, L, 30, 60)
    s2, e2 = _mixed_reads(rng, 300, L, 380, 400)
    s, e = np.concatenate([s1, s2]), np.concatenate([e1, e2])
    perm = rng.permutation(s.size)
    _check_mixed(pkg, oracle, solver, s[perm], e[perm], L, M)


# ---------------------------------------------------------------- odd shapes


@pytest.mark.parametrize("span", [1, 2, 64, 65])
def test_many_contigs_tiny_spans_and_m_one(pkg, oracle, solver, span):
    """200 contigs (stretch table: 200 contig starts + window cuts), spans of one and two bases, M = 1"""
    rng = np.random.default_rng(1000 + span)
    lengths = rng.integers(max(span, 50), 6000, size=200).astype(np.uint32)
    lengths[rng.integers(0, 200, size=20)] = 0
    parts, offs = [], [0]
    for L in lengths:
        L = int(L)
        n = 0 if L < span or rng.random() < 0.2 else int(rng.integers(1, max(2, 3 * L // max(span, 8))))
        s = rng.integers(0, L - span + 1, size=n, dtype=np.uint32) if n else np.zeros(0, np.uint32)
        parts.append(s)
        offs.append(offs[-1] + n)
    s = np.concatenate(parts)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    offs = np.array(offs, dtype=np.uint64)
    for M in (1, 3):
        _check(pkg, oracle, solver, s, e, lengths, M, offs=offs, expect_split=False)
    assert solver.last_stats.sweep_stretches >= int((lengths > 0).sum())


def test_more_contigs_than_the_stretch_table_takes(pkg, oracle, solver):
    """300 contigs: no split is attempted (one workgroup per contig already fills the chip)"""
    rng = np.random.default_rng(9)
    span, M = 40, 2
    lengths = np.full(300, 5000, dtype=np.uint32)
    counts = rng.integers(0, 400, size=300)
    s = np.concatenate([rng.integers(0, 5000 - span + 1, size=int(c), dtype=np.uint32) for c in counts])
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    whole, split = _check(pkg, oracle, solver, s, e, lengths, M, offs=offs, expect_split=False)
    assert whole == split == 300


def test_single_read_and_cap_above_every_coverage(pkg, oracle, solver):
    """M above the deepest coverage: every position is a cut point and every read is kept"""
    rng = np.random.default_rng(2)
    L, span = 500_000, 100
    s, e = _reads(rng, 40_000, L, span)
    with forced(solver, QMCP_HIP_CUTS="1"):
        got = solver.solve(s, e, L, 1_000_000)
        assert solver.last_stats.n_kept == s.size and solver.last_stats.sweep_stretches > 50
        assert np.array_equal(got, oracle.solve(s, e, L, 1_000_000))
        one = solver.solve(s[:1], e[:1], L, 5)
        assert solver.last_stats.n_kept == 1
        assert np.array_equal(one, oracle.solve(s[:1], e[:1], L, 5))
