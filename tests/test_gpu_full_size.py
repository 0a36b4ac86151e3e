"""Full-size runs of the BASELINE.json configs that do not fit the bench line, and the multi-rank
composition emulated on one GPU.  Each compares the HIP result with the oracle (composed where the
path is a pipeline) -- the comparisons the reference never makes (its tests pin only the validity
inequality, src/tests/coverage_tester.cpp:101-107)."""
import importlib
import os

import numpy as np
import pytest

import workloads

pytestmark = pytest.mark.gpu


def test_cfg3_full_size_through_bed_and_tsv_files(pkg, oracle, solver, tmp_path):
    """configs[2] at full size: 30 M amplicon reads on 29 903 bases; primers written to BED + TSV files,
    parsed back by the host mirror of AmpliconSet construction (bam_api.cpp:53-95,113-180), FILTER ->
    compaction -> solve at M = 200 -> find_pairs on the device (src/app.cpp:113-142) == composed oracle"""
    n_pairs = 15_000_000
    s, e, a0, a1, straddle = workloads.amplicon_reads(n_pairs)
    bed, tsv = tmp_path / "primers.bed", tmp_path / "pairs.tsv"
    with open(bed, "w") as fb, open(tsv, "w") as ft:
        for k, (lo, hi) in enumerate(zip(a0.tolist(), a1.tolist())):
            # primers are the 25 bp at each end; the amplicon is [left.start, right.end] with the BED
            # numbers taken as they stand (bam_api.cpp:64-73)
            fb.write(f"MN908947.3\t{lo}\t{lo + 24}\tamp{k}_LEFT\n")
            fb.write(f"MN908947.3\t{hi - 24}\t{hi}\tamp{k}_RIGHT\n")
            ft.write(f"amp{k}_LEFT\tamp{k}_RIGHT\n")
    fa0, fa1 = pkg.amplicons_from_files(str(bed), str(tsv))
    assert np.array_equal(fa0, a0) and np.array_equal(fa1, a1)
    got, dropped = solver.filter_solve(s, e, 29_903, 200, amp_starts=fa0, amp_ends=fa1, complete_pairs=True)
    # composed oracle with the amplicons as parsed from the files
    keep_pairs = oracle.amplicon_filter(s, e, fa0, fa1)
    kp = np.zeros(n_pairs, bool)
    kp[pkg.mask_to_indices(keep_pairs, n_pairs).astype(np.int64)] = True
    assert dropped == int((~kp).sum())
    sel = np.repeat(kp, 2)
    orig = np.flatnonzero(sel)
    m = oracle.find_pairs(oracle.solve(s[sel], e[sel], 29_903, 200), orig.size)
    want = pkg.indices_to_mask(orig[pkg.mask_to_indices(m, orig.size).astype(np.int64)], s.size)
    assert np.array_equal(got, want)
    assert solver.last_stats.path == pkg.PATH_UNIFORM and solver.last_stats.n_reads == orig.size


def test_cfg5_one_gpus_share_contig_by_contig(pkg, oracle, solver):
    """one GPU's share of configs[4]: 24 contigs ~ GRCh38 proportions at 1/8 of the genome-scale
    configuration (187.5 M positions, 125 M reads, M = 50): two partition levels, general-form sweep in
    speculative stretches"""
    s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9 / 8), int(0.5e9 / 8))
    got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_UNIFORM and st.sort_passes == 1 and st.n_contigs == 24
    # coverage 2 x M, a few dozen cut points in 187.5 M positions: the sweep runs as several hundred stretches
    # from speculative boundaries (csrc/kernels/sweep_segments.inc.hip), all of which must have held
    assert st.spec_boundaries > 500 and st.spec_mismatches == 0 and st.sweep_stretches > 500, st.as_dict()
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")
    for c in range(lengths.size):   # contig by contig: bounded memory on the oracle side
        a, b = int(offs[c]), int(offs[c + 1])
        want = oracle.solve(s[a:b], e[a:b], int(lengths[c]), 50)
        wbits = np.unpackbits(want.view(np.uint8), bitorder="little")[:b - a]
        assert np.array_equal(bits[a:b], wbits), f"contig {c}"


def test_cfg5_real_share_of_the_heaviest_rank_at_full_length(pkg, oracle, solver):
    """configs[4] at FULL size dealt to 8 ranks by sharding.assign_contigs (24 contigs, 1.5e9 positions, 1e9
    reads): the rank with the most reads owns three contigs at their full length -- 117.7 M, 52.0 M and 24.8 M
    positions, 129.7 M reads -- unlike the 1/8-scale genome above, whose longest contig has 15 M positions.
    Limits of the C ABI (include/qmcp_hip.h: 2^31 - 2 positions per call, 2^28 reads per contig on the block
    sweep) hold with room; kept set == oracle contig by contig."""
    share, owned = workloads.cfg5_heaviest_share(8)
    assert sorted(sum(owned, [])) == list(range(24))
    s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9), int(0.5e9), only=share)
    assert int(lengths.sum()) < (1 << 31) - 2 and int(np.diff(offs.astype(np.int64)).max()) < (1 << 28)
    assert int(lengths.max()) > 100_000_000 and s.size > 120_000_000
    got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_UNIFORM and st.sort_passes == 1 and st.n_contigs == len(share)
    assert st.spec_boundaries > 500 and st.spec_mismatches == 0 and st.sweep_stretches > 500, st.as_dict()
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")
    for c in range(lengths.size):
        a, b = int(offs[c]), int(offs[c + 1])
        want = oracle.solve(s[a:b], e[a:b], int(lengths[c]), 50)
        wbits = np.unpackbits(want.view(np.uint8), bitorder="little")[:b - a]
        assert np.array_equal(bits[a:b], wbits), f"contig {share[c]}"


def test_cfg5_real_share_with_clipped_reads(pkg, oracle, solver):
    """the same share (117.7 M, 52.0 M and 24.8 M positions, 129.7 M reads, M = 50) with 1 % of its reads clipped by
    1 ... 50 bases: long shallow contigs with a tail of shorter spans take the near-uniform route with its sweeps in
    stretches (a round lists ~140 k suspects there -- with the fixed list of 64 Ki the call
    went to the mixed-span walk, one chain per contig: 14.5 s); kept set == oracle contig by contig."""
    share, _ = workloads.cfg5_heaviest_share(8)
    s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9), int(0.5e9), only=share)
    s, e = workloads.clipped_mix(s, e, 0.01)
    got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_giveup == 0, st.as_dict()
    assert st.near_uniform_exceptions > 1_000_000 and st.near_uniform_selected > 10_000 and st.sweep_stretches > 500
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")
    for c in range(lengths.size):
        a, b = int(offs[c]), int(offs[c + 1])
        want = oracle.solve(s[a:b], e[a:b], int(lengths[c]), 50)
        wbits = np.unpackbits(want.view(np.uint8), bitorder="little")[:b - a]
        assert np.array_equal(bits[a:b], wbits), f"contig {share[c]}"


def test_cfg5_real_share_with_longer_reads(pkg, oracle, solver):
    """the same share with 1 % clipped AND 0.5 % of the reads lengthened by 1 ... 20 bases (deletions): reads longer than
    the dominant length leave the near-uniform route; the mixed-span walk then must not be one chain per contig (14.5 s for
    the 117.7 M-position contig): it speculates -- the rule is how deep the data is in standard deviations, not that one
    length dominates -- and its boundaries hold; kept set == oracle contig by contig"""
    share, _ = workloads.cfg5_heaviest_share(8)
    s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9), int(0.5e9), only=share)
    s, e = workloads.clipped_mix(s, e, 0.01)
    e = workloads.lengthened_mix(s, e, offs, lengths, 0.005)
    got = solver.solve(s, e, lengths, 50, contig_read_offsets=offs)
    st = solver.last_stats
    assert st.path == pkg.PATH_GENERAL and st.near_uniform_giveup == 2, st.as_dict()
    assert st.spec_boundaries > 300 and st.sweep_stretches > 300 and st.ms_total < 1000.0, st.as_dict()
    bits = np.unpackbits(got.view(np.uint8), bitorder="little")
    for c in range(lengths.size):
        a, b = int(offs[c]), int(offs[c + 1])
        want = oracle.solve(s[a:b], e[a:b], int(lengths[c]), 50)
        wbits = np.unpackbits(want.view(np.uint8), bitorder="little")[:b - a]
        assert np.array_equal(bits[a:b], wbits), f"contig {share[c]}"


@pytest.mark.parametrize("world", [2, 3, 8])
def test_ranks_emulated_on_one_gpu(pkg, oracle, solver, world):
    """the N > 1 composition with the HIP solver: assign_contigs -> local_problem -> HIP solve per
    rank -> merge_masks == the whole HIP solve == the oracle (bench.py --mode sharded runs this over RCCL)"""
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    rng = np.random.default_rng(world)
    lengths = np.array([300_000, 40_000, 1_000_000, 150, 90_000, 500_000, 20_000, 250_000, 700_000, 60_000, 333_333],
                       np.uint32)
    counts = [900_000, 60_000, 1_500_000, 0, 300_000, 100_000, 200_000, 600_000, 1_200_000, 5_000, 400_000]
    ss = [rng.integers(0, int(L) - 150 + 1, size=c).astype(np.uint32) for L, c in zip(lengths, counts)]
    s = np.concatenate(ss)
    e = (s + np.uint32(149)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    M = 30
    whole = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
    owned = sh.assign_contigs(counts, world, contig_lengths=lengths)
    assert sorted(c for o in owned for c in o) == list(range(len(counts)))
    locals_ = []
    for r in range(world):
        ls, le, loffs, llens = sh.local_problem(s, e, offs, lengths, owned[r])
        locals_.append(solver.solve(ls, le, llens, M, contig_read_offsets=loffs) if len(llens) else
                       np.zeros(0, np.uint64))
    merged = sh.merge_masks(locals_, owned, offs, s.size)
    assert np.array_equal(merged, whole)
    assert np.array_equal(whole, oracle.solve(s, e, lengths, M, contig_read_offsets=offs))
    # the assignment balances the modelled cost (reads + the longest chain of a rank)
    costs = [sh.rank_cost(counts, lengths, o) for o in owned]
    assert max(costs) <= 2.0 * (sum(sh.NS_PER_READ * c for c in counts) / world + sh.NS_PER_POSITION * int(lengths.max()))


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_multi_device_entry_with_contexts_on_one_device(pkg, oracle, solver, n_ctx):
    """qmcp_hip_multi_solve_host with devices = {0, 0[, 0]}: one context and host thread per entry, contigs
    dealt by cost, masks merged at bit positions that are not multiples of 64 == the one-device solve"""
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    rng = np.random.default_rng(70 + n_ctx)
    lengths = np.array([120_000, 7_000, 400_000, 300, 90_001, 55_555, 1_000_000], np.uint32)
    counts = [200_001, 9_999, 700_003, 0, 150_007, 33_333, 1_250_001]      # odd counts: unaligned boundaries
    ss = [rng.integers(0, int(L) - 150 + 1, size=c).astype(np.uint32) for L, c in zip(lengths, counts)]
    s = np.concatenate(ss)
    e = (s + np.uint32(149)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    with pkg.MultiSolver([0] * n_ctx) as ms:
        got = ms.solve(s, e, lengths, 25, contig_read_offsets=offs)
        where = ms.last_assignment
        kept = sum(int(st.n_kept) for st in ms.last_stats)
        again = ms.solve(s, e, lengths, 25, contig_read_offsets=offs)
    one = solver.solve(s, e, lengths, 25, contig_read_offsets=offs)
    assert np.array_equal(got, one) and np.array_equal(again, one)
    assert np.array_equal(one, oracle.solve(s, e, lengths, 25, contig_read_offsets=offs))
    assert kept == int(np.unpackbits(one.view(np.uint8)).sum())
    # the C side deals the contigs exactly as sharding.assign_contigs does
    owned = sh.assign_contigs(counts, n_ctx, contig_lengths=lengths, read_length=150, max_coverage=25)
    assert [where[c] for c in range(len(counts))] == [next(r for r, o in enumerate(owned) if c in o) for c in range(len(counts))]
