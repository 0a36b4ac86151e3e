"""Parity of the HIP solver against the CPU oracle, through the C ABI (bit-exact kept sets)."""
import numpy as np
import pytest

from conftest import random_reads

pytestmark = pytest.mark.gpu

SMALL_STARTS = [0, 6, 2, 6, 1, 7, 3, 9, 0, 7, 4, 9, 1, 6, 0, 4]
SMALL_ENDS = [2, 9, 4, 8, 3, 10, 6, 10, 4, 9, 6, 10, 4, 8, 2, 6]


def _check(pkg, oracle, solver, s, e, lengths, M, offs=None, expect_path=None):
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = solver.last_stats
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        gi, wi = pkg.mask_to_indices(got, len(s)), pkg.mask_to_indices(want, len(s))
        raise AssertionError(f"kept set differs: hip {gi.size} reads vs oracle {wi.size}; "
                             f"first diff near {np.setxor1d(gi, wi)[:8]}")
    assert st.n_kept == int(np.unpackbits(got.view(np.uint8)).sum())
    if expect_path is not None and len(s):
        assert st.path == expect_path
    return got


def test_small_fixture(pkg, oracle, solver):
    # src/tests/coverage_tester.cpp:72-93,109-118 (L = 11, M = 4); spans 2..5 -> mixed path
    m = _check(pkg, oracle, solver, SMALL_STARTS, SMALL_ENDS, 11, 4, expect_path=pkg.PATH_GENERAL)
    incov = oracle.cover(SMALL_STARTS, SMALL_ENDS, 11)
    outcov = oracle.cover(SMALL_STARTS, SMALL_ENDS, 11, keep_mask=m)
    assert oracle.is_out_cover_valid(incov, outcov, 4)


@pytest.mark.parametrize("M", [0, 1, 2, 3, 4, 5, 7, 100])
def test_small_fixture_all_m(pkg, oracle, solver, M):
    _check(pkg, oracle, solver, SMALL_STARTS, SMALL_ENDS, 11, M)


@pytest.mark.parametrize("seed", range(8))
def test_random_uniform_span_small(pkg, oracle, solver, seed):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 400))
    ell = int(rng.integers(1, min(L, 200) + 1))
    n = int(rng.integers(0, 3000))
    s, e = random_reads(rng, n, L, ell, ell)
    M = int(rng.integers(0, 12))
    _check(pkg, oracle, solver, s, e, L, M, expect_path=pkg.PATH_UNIFORM)


@pytest.mark.parametrize("seed", range(8))
def test_random_mixed_span_small(pkg, oracle, solver, seed):
    rng = np.random.default_rng(100 + seed)
    L = int(rng.integers(2, 500))
    n = int(rng.integers(1, 4000))
    s, e = random_reads(rng, n, L, 1, int(rng.integers(2, 120)))
    M = int(rng.integers(0, 15))
    _check(pkg, oracle, solver, s, e, L, M)


@pytest.mark.parametrize("ell", [1, 2, 63, 64, 65, 128, 129, 150, 191, 192, 193, 256, 300, 511, 512])
def test_uniform_span_block_edges(pkg, oracle, solver, ell):
    rng = np.random.default_rng(ell)
    L = 5 * ell + int(rng.integers(0, ell + 1))
    s, e = random_reads(rng, 20000, L, ell, ell)
    _check(pkg, oracle, solver, s, e, L, 9, expect_path=pkg.PATH_UNIFORM)


def test_uniform_span_above_block_limit_takes_general_path(pkg, oracle, solver):
    rng = np.random.default_rng(7)
    s, e = random_reads(rng, 5000, 4000, 700, 700)
    _check(pkg, oracle, solver, s, e, 4000, 6, expect_path=pkg.PATH_GENERAL)


def test_sparse_coverage_many_cut_points(pkg, oracle, solver):
    rng = np.random.default_rng(11)
    s, e = random_reads(rng, 3000, 200000, 100, 100)
    _check(pkg, oracle, solver, s, e, 200000, 3)
    s, e = random_reads(rng, 3000, 200000, 20, 180)
    _check(pkg, oracle, solver, s, e, 200000, 3)


def test_all_reads_same_start(pkg, oracle, solver):
    s = np.full(5000, 17, np.uint32)
    _check(pkg, oracle, solver, s, s + 49, 100, 7)
    rng = np.random.default_rng(3)
    e = (s + rng.integers(0, 60, size=s.size)).astype(np.uint32)
    _check(pkg, oracle, solver, s, e, 100, 7)


def test_empty_and_single(pkg, oracle, solver):
    z = np.zeros(0, np.uint32)
    got = solver.solve(z, z, 100, 5)
    assert got.size == 0 and solver.last_stats.n_kept == 0
    _check(pkg, oracle, solver, [3], [3], 10, 1)
    _check(pkg, oracle, solver, [0], [9], 10, 5)


def test_multi_contig_matches_per_contig(pkg, oracle, solver):
    rng = np.random.default_rng(5)
    lengths = np.array([300, 1, 4000, 150, 977], np.uint32)
    counts = [2000, 0, 9000, 40, 3000]
    ss, ee = [], []
    for L, c in zip(lengths, counts):
        a, b = random_reads(rng, c, int(L), 1 if L < 150 else 150, 1 if L < 150 else 150) if L != 4000 \
            else random_reads(rng, c, int(L), 30, 200)
        ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    _check(pkg, oracle, solver, s, e, lengths, 8, offs=offs)
    # all-uniform multi-contig -> block sweep, one wave per contig
    ss, ee = [], []
    for L, c in zip([3000, 4500, 150, 20000], [20000, 100, 7, 60000]):
        a, b = random_reads(rng, c, L, 150, 150)
        ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.array([0, 20000, 20100, 20107, 80107], np.uint64)
    _check(pkg, oracle, solver, s, e, np.array([3000, 4500, 150, 20000], np.uint32), 25, offs=offs,
           expect_path=pkg.PATH_UNIFORM)


def test_invalid_reads_are_rejected(pkg, solver):
    with pytest.raises(pkg.QmcpError) as ei:
        solver.solve([5], [4], 10, 1)
    assert ei.value.code == -2
    with pytest.raises(pkg.QmcpError):
        solver.solve([5], [10], 10, 1)
    # the context stays usable after an error
    assert solver.solve([5], [9], 10, 1)[0] == 1


def test_cfg1_and_determinism(pkg, oracle, solver):
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 5000, 3000)
    a = _check(pkg, oracle, solver, s, e, 3000, 100, expect_path=pkg.PATH_UNIFORM)
    b = solver.solve(s, e, 3000, 100)
    assert np.array_equal(a, b)
    ok, val = oracle.check_flow(s, e, 3000, 100, a)
    assert ok and val == 100


@pytest.mark.parametrize("max_span", [150, 1000, 4032, 4033, 6000])
def test_mixed_span_both_sweep_kernels(pkg, oracle, solver, max_span):
    """spans up to 4032 use the LDS-cached mixed-span sweep, wider ones the plain one"""
    rng = np.random.default_rng(max_span)
    L = 40_000
    s, e = random_reads(rng, 60_000, L, max(1, max_span // 3), max_span)
    s[0], e[0] = 100, 100 + max_span - 1  # make sure the maximum span occurs
    _check(pkg, oracle, solver, s, e, L, 12, expect_path=pkg.PATH_GENERAL)


def test_mixed_span_deep_amplicon_like(pkg, oracle, solver):
    """deep coverage, a dominant span plus a tail of shorter ones (clipped reads)"""
    rng = np.random.default_rng(77)
    L, n = 30_000, 400_000
    span = np.where(rng.random(n) < 0.85, 150, rng.integers(60, 150, size=n))
    st = (rng.random(n) * (L - span + 1)).astype(np.int64)
    s, e = st.astype(np.uint32), (st + span - 1).astype(np.uint32)
    _check(pkg, oracle, solver, s, e, L, 200, expect_path=pkg.PATH_GENERAL)
    _check(pkg, oracle, solver, s, e, L, 3, expect_path=pkg.PATH_GENERAL)


def test_many_small_contigs(pkg, oracle, solver):
    """300 contigs (more than the 64 the prepare kernel caches in LDS), some empty, some tiny"""
    rng = np.random.default_rng(2024)
    lens, counts, ss, ee = [], [], [], []
    for c in range(300):
        L = int(rng.integers(150, 3000))
        n = 0 if c % 7 == 0 else int(rng.integers(1, 4000))
        a, b = random_reads(rng, n, L, 150, 150)
        lens.append(L); counts.append(n); ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    _check(pkg, oracle, solver, s, e, np.array(lens, np.uint32), 10, offs=offs, expect_path=pkg.PATH_UNIFORM)
    # same layout with mixed spans
    ss, ee = [], []
    for L, n in zip(lens, counts):
        a, b = random_reads(rng, n, L, 30, 150)
        ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    _check(pkg, oracle, solver, s, e, np.array(lens, np.uint32), 10, offs=offs, expect_path=pkg.PATH_GENERAL)


@pytest.mark.parametrize("max_span,contigs", [(16_384, 1), (40_000, 1), (131_000, 2)])
def test_long_reads_rings_in_global_memory(pkg, oracle, solver, max_span, contigs):
    """spans beyond 16 383 (long reads, up to whole-genome length): the plain event sweep keeps its two
    rings in global memory instead of LDS"""
    rng = np.random.default_rng(max_span)
    L = max_span + 9_000
    ss, ee, offs = [], [], [0]
    for c in range(contigs):
        n = 3_000
        span = np.where(rng.random(n) < 0.3, rng.integers(max_span // 2, max_span + 1, size=n), rng.integers(200, 3_000, size=n))
        st = (rng.random(n) * (L - span + 1)).astype(np.int64)
        if c == 0:
            st[0], span[0] = 5, max_span  # the longest span occurs
        ss.append(st.astype(np.uint32)); ee.append((st + span - 1).astype(np.uint32))
        offs.append(offs[-1] + n)
    s, e = np.concatenate(ss), np.concatenate(ee)
    lengths = np.full(contigs, L, dtype=np.uint32)
    offs = np.array(offs, dtype=np.uint64)
    for M in (2, 25):
        _check(pkg, oracle, solver, s, e, lengths, M, offs=offs, expect_path=pkg.PATH_GENERAL)
