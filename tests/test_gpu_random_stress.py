"""Seeded random uniform-span cases through every route the host can pick (ranked one- and two-level,
sort-based; fast and general sweep pipelines; single-wave sweep), each bit-identical to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng):
    span = int(rng.choice([1, 7, 64, 100, 150, 151, 200, 256, 300]))
    n_contigs = int(rng.integers(1, 6))
    big = rng.random() < 0.15                      # beyond 8.39 M positions: two partition levels
    lengths, counts = [], []
    for _ in range(n_contigs):
        L = int(rng.integers(span, 40_000)) if not big else int(rng.integers(3_000_000, 6_000_000))
        lengths.append(max(L, span))
        counts.append(int(rng.integers(0, 120_000)))
    if rng.random() < 0.3:
        counts[int(rng.integers(0, n_contigs))] = 0   # a contig without reads
    if sum(counts) == 0:
        counts[0] = 1000
    ss, ee = [], []
    for L, c in zip(lengths, counts):
        hi = L - span + 1
        if rng.random() < 0.4 and hi > 50:            # hot spots: a few start positions hold most reads
            spots = rng.integers(0, hi, size=int(rng.integers(1, 6)))
            s = np.where(rng.random(c) < 0.7, rng.choice(spots, size=c), rng.integers(0, hi, size=c))
        else:
            s = rng.integers(0, hi, size=c)
        s = s.astype(np.uint32)
        ss.append(s)
        ee.append((s + np.uint32(span - 1)).astype(np.uint32))
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    M = int(rng.choice([1, 2, 5, 17, 50, 100, 300, 5000]))
    return s, e, np.array(lengths, np.uint32), offs, M


@pytest.mark.parametrize("seed", range(40))
def test_random_uniform_span_case(pkg, oracle, solver, seed):
    rng = np.random.default_rng(90_000 + seed)
    s, e, lengths, offs, M = _case(rng)
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want), (seed, s.size, lengths.tolist(), M, solver.last_stats.sort_passes)
    assert solver.last_stats.n_kept == int(np.unpackbits(want.view(np.uint8)).sum())


def _shallow_case(rng):
    """coverage within a few times the cap, patchy: the data cut points split"""
    span = int(rng.choice([30, 64, 100, 150, 151, 256, 300]))
    n_contigs = int(rng.integers(1, 5))
    M = int(rng.choice([1, 2, 5, 17, 50]))
    lengths, counts, ss = [], [], []
    for _ in range(n_contigs):
        L = int(rng.integers(20_000, 400_000))
        depth = float(rng.choice([0.3, 0.8, 1.0, 1.5, 3.0]))
        c = min(int(L * M * depth / span), 400_000)
        hi = L - span + 1
        s = rng.integers(0, hi, size=c)
        if rng.random() < 0.5:                        # patches: starts folded into part of every period
            period = int(rng.integers(2 * span, 20 * span))
            fill = float(rng.uniform(0.2, 0.9))
            s = np.minimum((s // period) * period + ((s % period) * fill).astype(np.int64), hi - 1)
        lengths.append(L)
        counts.append(c)
        ss.append(s.astype(np.uint32))
    s = np.concatenate(ss)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    return s, e, np.array(lengths, np.uint32), offs, M


@pytest.mark.parametrize("seed", range(24))
def test_random_shallow_case_split_at_cut_points(pkg, oracle, solver, seed, monkeypatch):
    rng = np.random.default_rng(70_000 + seed)
    s, e, lengths, offs, M = _shallow_case(rng)
    # (seed % 3 == 0: the checked fast form with its fallbacks, segmented)
    with solver.options(cut_points=1, sweep="fast" if seed % 3 == 0 else None):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want), (seed, s.size, lengths.tolist(), M, solver.last_stats.sweep_stretches)


@pytest.mark.parametrize("seed", range(16))
def test_random_shallow_mixed_span_case_split_at_cut_points(pkg, oracle, solver, seed, monkeypatch):
    """the same shallow cases with every read's span drawn at random: the event sweeps, segmented"""
    rng = np.random.default_rng(50_000 + seed)
    s, e, lengths, offs, M = _shallow_case(rng)
    span = int(e[0] - s[0]) + 1
    cut = rng.integers(0, max(span // 2, 1), size=s.size).astype(np.uint32)   # shorten: ends stay inside the contig
    e = (e - cut).astype(np.uint32)
    with solver.options(cut_points=1, mixed_sweep_in_lds=1 if seed % 4 == 0 else 0):
        got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert solver.last_stats.path == pkg.PATH_GENERAL
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    assert np.array_equal(got, want), (seed, s.size, lengths.tolist(), M, solver.last_stats.sweep_stretches)


@pytest.mark.parametrize("seed", range(10))
def test_random_wide_mixed_spans(pkg, oracle, solver, seed, monkeypatch):
    """spans from tens to thousands of bases (the LDS-cached event sweep, beyond 4 032 the plain one), deep
    and shallow, with and without the cut-point split"""
    rng = np.random.default_rng(30_000 + seed)
    hi = int(rng.choice([500, 1200, 2500, 4032, 5000]))
    lo = int(rng.integers(1, hi // 2))
    n_contigs = int(rng.integers(1, 4))
    lengths, counts, ss, ee = [], [], [], []
    for _ in range(n_contigs):
        L = int(rng.integers(hi + 10, 60_000))
        c = int(rng.integers(0, 30_000))
        span = rng.integers(lo, hi + 1, size=c)
        st = (rng.random(c) * (L - span + 1)).astype(np.int64)
        lengths.append(L); counts.append(c)
        ss.append(st.astype(np.uint32)); ee.append((st + span - 1).astype(np.uint32))
    if sum(counts) == 0:
        return
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    M = int(rng.choice([1, 3, 20, 150]))
    with solver.options(cut_points=1 if seed % 2 else -1):
        got = solver.solve(s, e, np.array(lengths, np.uint32), M, contig_read_offsets=offs)
    want = oracle.solve(s, e, np.array(lengths, np.uint32), M, contig_read_offsets=offs)
    assert np.array_equal(got, want), (seed, lo, hi, lengths, counts, M)
