"""The two forms of the range-ranked route -- range-major (k_prepare, scan, k_range_partition: the reads are read
twice and written as 6 B per read) and pass-major (k_pm_prepare_sort: one pass, every 8 192 reads sorted by range in
place, 4 B per read; csrc/kernels/pass_major.inc.hip) -- must give the same keep mask, bit for bit, and the
oracle's, on ragged inputs: last pass partial, ranges that straddle contig borders, contigs without reads, a genome
of a few hundred positions, one range that holds everything, an odd number of reads.  QMCP_HIP_PM=0 / 1 picks the
form (default: pass-major where every range's row of passes fits the kernels' share of LDS).  The layout's index
arithmetic has a host-side model of its own: tests/test_pass_major_model.py."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _uniform_reads(rng, n, L, span):
    s = rng.integers(0, L - span + 1, size=n, dtype=np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _with_pm(solver, v, fn):
    with solver.options(pass_major=1 if v == "1" else -1):
        return fn()


CASES = [
    # lengths, read counts, span, M
    ([100_000], [3 * 8192 * 8 + 77], 150, 40),                               # one contig, last pass partial, odd n
    ([70_001, 33_333, 250_000, 1_000], [300_000, 200_011, 500_000, 40_005], 120, 60),  # ranges straddle contig borders
    ([5_000, 50_000, 5_000], [0, 400_001, 0], 100, 25),                      # contigs without reads
    ([300], [200_000], 30, 50),                                               # tiny genome
    ([30_000], [400_000], 150, 100),                                          # ONE range holds everything -> sort route
    ([(1 << 21) - 5], [600_000], 150, 3),                                     # 256 ranges, sparse: many empty slices
]


@pytest.mark.parametrize("lengths,counts,span,M", CASES)
def test_both_forms_equal_the_oracle(pkg, oracle, solver, lengths, counts, span, M):
    rng = np.random.default_rng(sum(counts) % 9973)
    lengths = np.array(lengths, np.uint32)
    ss, ee = [], []
    for L, k in zip(lengths, counts):
        a, b = _uniform_reads(rng, int(k), int(L), min(span, int(L)))
        ss.append(a); ee.append(b)
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    solve = lambda: solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    pm = _with_pm(solver, "1", solve)
    st_pm = solver.last_stats
    rm = _with_pm(solver, "0", solve)
    st_rm = solver.last_stats
    assert np.array_equal(pm, rm)
    assert st_pm.n_kept == st_rm.n_kept and st_pm.sort_passes == st_rm.sort_passes
    assert np.array_equal(pm, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))


def test_pass_major_form_runs_where_it_should(pkg, solver):
    """kernel names tell which form ran: a one-level genome takes the pass-major producer unless told otherwise"""
    rng = np.random.default_rng(1)
    s, e = _uniform_reads(rng, 500_000, 200_000, 150)
    with pkg.Solver(0) as sv:
        sv.set_profiling(True)
        _with_pm(sv, "1", lambda: sv.solve(s, e, 200_000, 30))
        assert "k_pm_prepare_sort" in sv.kernel_times() and "k_range_partition" not in sv.kernel_times()
        sv.set_profiling(True)
        _with_pm(sv, "0", lambda: sv.solve(s, e, 200_000, 30))
        assert "k_range_partition" in sv.kernel_times() and "k_pm_prepare_sort" not in sv.kernel_times()


def test_invalid_read_fails_the_call_in_the_pass_major_form(pkg, oracle, solver):
    """the producer is queued before the host knows: a start beyond its contig is counted at the contig's last
    position, everything stays in bounds, the call reports the read and the context goes on working"""
    rng = np.random.default_rng(4)
    s, e = _uniform_reads(rng, 300_000, 100_000, 150)
    bad_s, bad_e = s.copy(), e.copy()
    bad_s[123_456] = 4_000_000_000
    bad_e[123_456] = 4_000_000_100
    with pytest.raises(pkg.QmcpError):
        _with_pm(solver, "1", lambda: solver.solve(bad_s, bad_e, 100_000, 20))
    got = _with_pm(solver, "1", lambda: solver.solve(s, e, 100_000, 20))
    assert np.array_equal(got, oracle.solve(s, e, 100_000, 20))
