"""Properties of the oracle's canonical selection (the part the reference leaves to OR-Tools)."""
import itertools

import numpy as np
import pytest

from conftest import random_reads


def py_rule(starts, ends, L, M):
    """the canonical rule written the slow, obvious way (oracle/qmcp_oracle.c header)"""
    n = len(starts)
    cov = np.zeros(L, dtype=np.int64)
    for s, e in zip(starts, ends):
        cov[s:e + 1] += 1
    sel = np.zeros(n, dtype=bool)
    for p in range(L):
        cur = sum(1 for i in range(n) if sel[i] and starts[i] <= p <= ends[i])
        k = min(int(cov[p]), M) - cur
        if k > 0:
            cand = [i for i in range(n) if not sel[i] and starts[i] <= p <= ends[i]]
            cand.sort(key=lambda i: (-int(ends[i]), -int(starts[i]), i))
            for i in cand[:k]:
                sel[i] = True
    return sel


def valid(starts, ends, L, M, sel):
    cov = np.zeros(L, dtype=np.int64)
    out = np.zeros(L, dtype=np.int64)
    for i, (s, e) in enumerate(zip(starts, ends)):
        cov[s:e + 1] += 1
        if sel[i]:
            out[s:e + 1] += 1
    return bool(np.all(np.minimum(cov, M) <= out))


def brute_min(starts, ends, L, M):
    n = len(starts)
    for k in range(n + 1):
        for comb in itertools.combinations(range(n), k):
            sel = np.zeros(n, dtype=bool)
            sel[list(comb)] = True
            if valid(starts, ends, L, M, sel):
                return k
    raise AssertionError


def to_bool(pkg, mask, n):
    b = np.zeros(n, dtype=bool)
    b[pkg.mask_to_indices(mask, n).astype(np.int64)] = True
    return b


@pytest.mark.parametrize("seed", range(40))
def test_oracle_matches_plain_python_rule(oracle, pkg, seed):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 60))
    n = int(rng.integers(0, 80))
    s, e = random_reads(rng, n, L, 1, int(rng.integers(1, 20)))
    M = int(rng.integers(0, 6))
    got = to_bool(pkg, oracle.solve(s, e, L, M), n)
    assert np.array_equal(got, py_rule(s.tolist(), e.tolist(), L, M))


@pytest.mark.parametrize("seed", range(300))
def test_canonical_answer_has_minimum_cardinality(oracle, pkg, seed):
    """the rule returns a minimum-size valid subset (what the reference's mcp-cpu optimises,
    mcp_cpu_cost_scaling_solver.cpp:45-48): exhaustive check on tiny instances"""
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(1, 10))
    n = int(rng.integers(0, 11))
    s, e = random_reads(rng, n, L, 1, L)
    M = int(rng.integers(1, 5))
    sel = to_bool(pkg, oracle.solve(s, e, L, M), n)
    assert valid(s, e, L, M, sel)
    assert int(sel.sum()) == brute_min(s.tolist(), e.tolist(), L, M)


@pytest.mark.parametrize("seed", range(30))
def test_kept_set_is_a_maximum_flow_of_the_reference_network(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    L = int(rng.integers(5, 300))
    n = int(rng.integers(1, 2000))
    s, e = random_reads(rng, n, L, 1, int(rng.integers(1, 60)))
    M = int(rng.integers(1, 30))
    mask = oracle.solve(s, e, L, M)
    ok, value = oracle.check_flow(s, e, L, M, mask)
    assert ok
    # independent Dinic on the reference's exact arc list: same flow value
    assert value == oracle.maxflow_value(s, e, L, M) == oracle.graph(s, e, L, M).total_supply
    # and the reference's own test assertion
    assert oracle.is_out_cover_valid(oracle.cover(s, e, L), oracle.cover(s, e, L, keep_mask=mask), M)


def test_flow_certificate_rejects_invalid_sets(oracle, pkg):
    s, e = [0, 0, 0, 2], [3, 3, 3, 5]
    good = oracle.solve(s, e, 6, 2)
    assert oracle.check_flow(s, e, 6, 2, good)[0]
    bad = pkg.indices_to_mask([0], 4)  # covers position 0..3 once, need is 2, position 4..5 uncovered
    assert not oracle.check_flow(s, e, 6, 2, bad)[0]
    everything = pkg.indices_to_mask([0, 1, 2, 3], 4)  # "keep all" is always a maximum flow
    assert oracle.check_flow(s, e, 6, 2, everything)[0]


def test_edge_cases(oracle, pkg):
    z = np.zeros(0, np.uint32)
    assert oracle.solve(z, z, 10, 3).size == 0
    # M = 0 -> b is all zero -> no supply -> nothing kept
    assert pkg.mask_to_indices(oracle.solve([1, 2], [5, 6], 10, 0), 2).size == 0
    # M above the coverage -> every read is forced
    assert pkg.mask_to_indices(oracle.solve([1, 2, 2], [5, 6, 2], 10, 99), 3).tolist() == [0, 1, 2]
    # tie-break: equal ends -> larger start first; equal (start, end) -> smaller index first
    assert pkg.mask_to_indices(oracle.solve([0, 1, 1], [4, 4, 4], 5, 1), 3).tolist() == [0]
    assert pkg.mask_to_indices(oracle.solve([1, 1, 0], [4, 4, 0], 5, 1), 3).tolist() == [0, 2]
    assert pkg.mask_to_indices(oracle.solve([1, 2, 2, 1], [4, 4, 4, 4], 5, 2), 4).tolist() == [0, 3]
    # read 0 covers 0..1; at p = 2 reads 1 (start 1) and 2, 3 (start 2) all end at 4 -> start 2, index 2
    assert pkg.mask_to_indices(oracle.solve([0, 1, 2, 2], [1, 4, 4, 4], 5, 1), 4).tolist() == [0, 2]
    with pytest.raises(ValueError):
        oracle.solve([3], [2], 10, 1)
    with pytest.raises(ValueError):
        oracle.solve([3], [10], 10, 1)


def test_multi_contig_is_independent_solves(oracle):
    rng = np.random.default_rng(9)
    lens = [200, 50, 1000]
    parts = [random_reads(rng, c, L, 5, 40) for c, L in zip([500, 0, 3000], lens)]
    s = np.concatenate([p[0] for p in parts])
    e = np.concatenate([p[1] for p in parts])
    offs = np.array([0, 500, 500, 3500], np.uint64)
    whole = oracle.solve(s, e, np.array(lens, np.uint32), 7, contig_read_offsets=offs)
    bits = np.unpackbits(whole.view(np.uint8), bitorder="little")[:3500]
    for c, (lo, hi) in enumerate([(0, 500), (500, 500), (500, 3500)]):
        one = oracle.solve(s[lo:hi], e[lo:hi], lens[c], 7)
        assert np.array_equal(np.unpackbits(one.view(np.uint8), bitorder="little")[:hi - lo], bits[lo:hi])


def test_find_pairs_and_amplicon_filter_restatements(oracle, pkg):
    m = pkg.indices_to_mask([0, 5, 63, 64, 130], 131)
    got = pkg.mask_to_indices(oracle.find_pairs(m, 131), 131).tolist()
    # index 130 has no mate in a 131-read set (the reference would index out of range there,
    # bam_api.cpp:258; inputs always hold whole pairs): it stays as it was, no bit past n_reads
    assert got == [0, 1, 4, 5, 62, 63, 64, 65, 130]
    s = np.array([10, 100, 10, 300, 50, 60], np.uint32)
    e = np.array([59, 149, 59, 349, 99, 109], np.uint32)
    keep = oracle.amplicon_filter(s, e, [0, 40], [200, 120])
    assert pkg.mask_to_indices(keep, 3).tolist() == [0, 2]  # pair 1 straddles both amplicons
    keep = oracle.amplicon_filter(s, e, [0, 40], [200, 120], seq_lengths=[50] * 6, min_length=60)
    assert pkg.mask_to_indices(keep, 3).size == 0
    keep = oracle.amplicon_filter(s, e, [0], [400], qualities=[30, 30, 30, 29, 60, 60], min_mapq=30)
    assert pkg.mask_to_indices(keep, 3).tolist() == [0, 2]
