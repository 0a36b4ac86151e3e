"""Parity cases for the BASELINE.json configs beyond the bench workload."""
import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def _valid(oracle, solver, s, e, lengths, offs, M, mask):
    in_cov = solver.coverage(s, e, lengths, contig_read_offsets=offs)
    out_cov = solver.coverage(s, e, lengths, contig_read_offsets=offs, keep_mask=mask)
    return oracle.is_out_cover_valid(in_cov, out_cov, M)


def test_cfg3_amplicon_filter_then_solve(pkg, oracle, solver):
    """configs[2] at 1/10 scale (3 M reads): amplicon FILTER pre-pass (bam_api.cpp:311-319) on
    the device, survivors solved at M = 200; every stage bit-identical to the oracle"""
    n_pairs = 1_500_000
    s, e, a0, a1, straddle = workloads.amplicon_reads(n_pairs)
    keep_pairs = solver.amplicon_filter(s, e, a0, a1)
    assert np.array_equal(keep_pairs, oracle.amplicon_filter(s, e, a0, a1))
    kept = np.zeros(n_pairs, bool)
    kept[pkg.mask_to_indices(keep_pairs, n_pairs).astype(np.int64)] = True
    assert np.array_equal(kept, ~straddle)  # exactly the straddling pairs are filtered out
    sel = np.repeat(kept, 2)
    fs, fe = s[sel], e[sel]
    got = solver.solve(fs, fe, 29_903, 200)
    assert solver.last_stats.path == pkg.PATH_UNIFORM
    assert np.array_equal(got, oracle.solve(fs, fe, 29_903, 200))
    ok, _ = oracle.check_flow(fs, fe, 29_903, 200, got)
    assert ok
    # completing mates afterwards (src/app.cpp:141) keeps whole pairs only
    paired = solver.complete_pairs(got, fs.size)
    assert np.array_equal(paired, oracle.find_pairs(got, fs.size))
    bits = np.unpackbits(paired.view(np.uint8), bitorder="little")[:fs.size]
    assert np.array_equal(bits[0::2], bits[1::2])


def test_cfg3_filters_min_length_and_mapq(pkg, oracle, solver):
    s, e, a0, a1, _ = workloads.amplicon_reads(50_000, seed=3)
    rng = np.random.default_rng(1)
    lens = rng.integers(60, 151, size=s.size).astype(np.uint32)
    mapq = rng.integers(0, 61, size=s.size).astype(np.uint32)
    # reference defaults: -l 90, -q 30 (src/app.hpp:22-25)
    got = solver.amplicon_filter(s, e, a0, a1, seq_lengths=lens, qualities=mapq, min_length=90, min_mapq=30)
    assert np.array_equal(got, oracle.amplicon_filter(s, e, a0, a1, seq_lengths=lens, qualities=mapq,
                                                      min_length=90, min_mapq=30))


def test_cfg5_shape_sparse_multi_contig(pkg, oracle, solver):
    """configs[4] shape at 1/100 scale: 24 contigs ~ GRCh38 proportions, 10 M reads over 15 Mb
    (mean coverage 100x), M = 50 -- the sparse regime (most positions start no read)"""
    s, e, offs, lengths = workloads.wgs_contigs(15_000_000, 5_000_000)
    got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
    assert solver.last_stats.n_contigs == 24 and solver.last_stats.path == pkg.PATH_UNIFORM
    assert np.array_equal(got, oracle.solve(s, e, lengths, 50, contig_read_offsets=offs))
    assert _valid(oracle, solver, s, e, lengths, offs, 50, got)


def test_cfg4_full_size_properties(pkg, oracle, solver):
    """configs[3] at full size (100 M reads, 8 contigs, M = 100): size-independent properties --
    validity everywhere (device coverage probes), determinism, contig independence, and exact
    oracle parity on every contig"""
    pairs, L, M = 6_250_000, 1_000_000, 100
    ss, ee = zip(*[pkg.reads_gen(0, pairs, L, seed=12345 + c) for c in range(8)])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.arange(9, dtype=np.uint64) * np.uint64(2 * pairs)
    lengths = np.full(8, L, np.uint32)
    got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    n_kept = solver.last_stats.n_kept
    assert np.array_equal(got, solver.solve(s, e, lengths, M, contig_read_offsets=offs))  # determinism
    assert _valid(oracle, solver, s, e, lengths, offs, M, got)
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")
    n_c = 2 * pairs
    for c in (0, 5):  # contigs are independent solves
        one = np.unpackbits(solver.solve(ss[c], ee[c], L, M).view(np.uint8), bitorder="little")[:n_c]
        assert np.array_equal(one, bits[c * n_c:(c + 1) * n_c])
    # exact oracle parity on ALL eight contigs (the oracle takes ~14 s for the 10^8 reads)
    assert np.array_equal(got, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))
    # minimum cardinality on deep uniform data: M reads per read length of genome, per contig
    assert n_kept == int(bits.sum()) and abs(n_kept - 8 * M * L / 150) < 8 * 2 * M


def _oracle_filter_solve(oracle, pkg, s, e, L, M, a0, a1, lens, mapq, min_len, min_q, pairs):
    n = s.size
    if a0 is None:
        keep_pairs = oracle.amplicon_filter(s, e, [0], [0xFFFFFFFF], seq_lengths=lens, qualities=mapq,
                                            min_length=min_len, min_mapq=min_q)
    else:
        keep_pairs = oracle.amplicon_filter(s, e, a0, a1, seq_lengths=lens, qualities=mapq,
                                            min_length=min_len, min_mapq=min_q)
    kp = np.zeros(n // 2, bool)
    kp[pkg.mask_to_indices(keep_pairs, n // 2).astype(np.int64)] = True
    sel = np.repeat(kp, 2)
    orig = np.flatnonzero(sel)
    m = oracle.solve(s[sel], e[sel], L, M)
    if pairs:
        m = oracle.find_pairs(m, orig.size)
    kept_c = pkg.mask_to_indices(m, orig.size).astype(np.int64)
    return pkg.indices_to_mask(orig[kept_c], n), int((~kp).sum())


@pytest.mark.parametrize("pairs", [False, True])
def test_filter_solve_pipeline_matches_composed_oracle(pkg, oracle, solver, pairs):
    """App::execute around the solver, device-resident: FILTER -> compaction -> solve -> find_pairs,
    result over the ORIGINAL read indices (src/app.cpp:113-142)"""
    s, e, a0, a1, straddle = workloads.amplicon_reads(400_000, seed=21)
    rng = np.random.default_rng(8)
    lens = rng.integers(60, 151, size=s.size).astype(np.uint32)
    mapq = rng.integers(0, 61, size=s.size).astype(np.uint32)
    got, dropped = solver.filter_solve(s, e, 29_903, 200, amp_starts=a0, amp_ends=a1, seq_lengths=lens,
                                       qualities=mapq, min_length=90, min_mapq=30, complete_pairs=pairs)
    want, want_dropped = _oracle_filter_solve(oracle, pkg, s, e, 29_903, 200, a0, a1, lens, mapq, 90, 30, pairs)
    assert dropped == want_dropped and dropped > int(straddle.sum())
    assert np.array_equal(got, want)
    # no amplicons given: AmpliconBehaviour::IGNORE, only the length / MAPQ filters act
    got, dropped = solver.filter_solve(s, e, 29_903, 50, seq_lengths=lens, qualities=mapq, min_length=90,
                                       min_mapq=30, complete_pairs=pairs)
    want, want_dropped = _oracle_filter_solve(oracle, pkg, s, e, 29_903, 50, None, None, lens, mapq, 90, 30, pairs)
    assert dropped == want_dropped and np.array_equal(got, want)
    # nothing filtered at all == the plain solve
    got, dropped = solver.filter_solve(s, e, 29_903, 50, complete_pairs=False)
    assert dropped == 0 and np.array_equal(got, solver.solve(s, e, 29_903, 50))
    # everything filtered
    got, dropped = solver.filter_solve(s, e, 29_903, 50, seq_lengths=lens, min_length=1000)
    assert dropped == s.size // 2 and not got.any()
    with pytest.raises(pkg.QmcpError):
        solver.filter_solve(s[:-1], e[:-1], 29_903, 50)


def test_cfg3_shape_with_a_clipped_tail_takes_the_mixed_span_route(pkg, oracle, solver):
    """what a real amplicon BAM looks like: 85 % of the reads at the modal length, the rest soft-clipped
    shorter (read.cpp:11-13: the span is the CIGAR's reference length).  One different length sends the call
    to the mixed-span event sweep; same contract: bit-identical to the oracle (tie-break: end desc, start
    desc, index asc -- DESIGN.md section 2)"""
    n_pairs = 1_500_000
    s, e, a0, a1, _ = workloads.amplicon_reads(n_pairs, seed=77, straddle_fraction=0.0)
    rng = np.random.default_rng(6)
    clipped = rng.random(s.size) < 0.15
    cut = rng.integers(1, 51, size=s.size)
    front = rng.random(s.size) < 0.5
    s2 = np.where(clipped & front, s + cut, s).astype(np.uint32)
    e2 = np.where(clipped & ~front, e - cut, e).astype(np.uint32)
    got = solver.solve(s2, e2, 29_903, 200)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and st.min_span == 100 and st.max_span == 150
    assert np.array_equal(got, oracle.solve(s2, e2, 29_903, 200))
    ok, _ = oracle.check_flow(s2, e2, 29_903, 200, got)
    assert ok
