"""Cut points: a position with coverage <= M forces every read covering it to be kept
(SURVEY.md section 7), so the uniform-span sweep may restart behind it.  The solver splits contigs
there into stretches swept side by side (shallow or gapped data, where a contig's serial chain is the
whole cost).  The kept set must not change by a bit: segmented == unsegmented == oracle."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@contextmanager
def _env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    try:
        for k, v in kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _reads(rng, n, L, span, lo=0, hi=None):
    hi = L if hi is None else hi
    s = rng.integers(lo, hi - span + 1, size=n, dtype=np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _check(pkg, oracle, solver, s, e, lengths, M, offs=None, sweep=None, expect_split=True):
    with _env(QMCP_HIP_CUTS="0", QMCP_HIP_SWEEP=sweep):
        whole = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        assert solver.last_stats.path == pkg.PATH_UNIFORM
        chains_whole = solver.last_stats.sweep_stretches
    with _env(QMCP_HIP_CUTS="1", QMCP_HIP_SWEEP=sweep):
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
    with _env(QMCP_HIP_CUTS=None, QMCP_HIP_SWEEP=None):
        s, e = _reads(rng, 60_000, L, span)              # coverage 15 = 1.5 M
        got = solver.solve(s, e, L, 10)
        assert solver.last_stats.sweep_stretches > 1
        assert np.array_equal(got, oracle.solve(s, e, L, 10))
        s, e = _reads(rng, 1_000_000, L, span)           # coverage 250 = 25 M
        solver.solve(s, e, L, 10)
        assert solver.last_stats.sweep_stretches == 1
