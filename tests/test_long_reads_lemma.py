"""Reads LONGER than the dominant span (a deletion lengthens a read's reference span: libs/bam-api/src/read.cpp:11-13) -- the
lemma the next step of the near-uniform route rests on (lab/long_reads_lemma.py): with the long reads
the canonical greedy keeps counted from their START, the one-span greedy over the regular reads alone reproduces the oracle's
selection of the regular reads.  CPU only: the oracle and a plain Python sweep."""
import numpy as np
import pytest


def _sweep(c, need, ell):
    L = c.size
    cur = np.zeros(L, np.int64)
    for t in range(L):
        lo = max(0, t - ell + 1)
        d = int(need[t]) - int(cur[lo:t].sum())
        u = t
        while d > 0 and u >= lo:
            k = min(d, int(c[u] - cur[u]))
            cur[u] += k
            d -= k
            u -= 1
    return cur


@pytest.mark.parametrize("seed", range(12))
def test_kept_long_reads_counted_from_their_start_reproduce_the_regular_selection(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    L = int(rng.integers(500, 1500)); ell = int(rng.choice([20, 50, 80])); M = int(rng.choice([4, 8, 15]))
    depth = float(rng.choice([1.5, 3, 6, 12])); n = int(depth * M * L / ell)
    s = rng.integers(0, L - ell - 12, size=n).astype(np.int64)
    e = s + ell - 1
    lg = rng.random(n) < float(rng.choice([0.005, 0.02, 0.1]))
    e = np.where(lg, e + rng.integers(1, 11, size=n), e)
    mask = oracle.solve(s.astype(np.uint32), e.astype(np.uint32), np.array([L], np.uint32), M)
    kept = np.unpackbits(mask.view(np.uint8), bitorder="little")[:n].astype(bool)
    cov = np.zeros(L + 1, np.int64); np.add.at(cov, s, 1); np.add.at(cov, e + 1, -1)
    need = np.minimum(np.cumsum(cov)[:L], M)
    a = np.zeros(L + 1, np.int64); np.add.at(a, s[lg & kept], 1); np.add.at(a, e[lg & kept] + 1, -1)
    c = np.bincount(s[~lg], minlength=L).astype(np.int64)
    want = np.bincount(s[~lg & kept], minlength=L)
    assert np.array_equal(_sweep(c, need - np.cumsum(a)[:L], ell), want)
