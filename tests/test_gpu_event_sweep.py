"""The event-driven uniform sweep (kernels/sweep_uniform_events.inc.hip: pack -> chain -> expand), forced
with QMCP_HIP_SWEEP=ev on every kind of input -- deep (where the host picks it by itself), shallow,
gapped, split at cut points, every lane layout E = 1..4 -- must equal the oracle bit for bit, and must
equal what the block-scan kernels (QMCP_HIP_SWEEP=fast / gen) produce.  The selection it replaces:
SimpleMaxFlow::Solve at libs/qmcp-solver/src/quasi_mcp_cpu_max_flow_solver.cpp:19-20."""

import numpy as np
import pytest

from forcing import forced

pytestmark = pytest.mark.gpu


def _reads(rng, n, L, span, hot=False):
    hi = L - span + 1
    if hot and hi > 50:
        spots = rng.integers(0, hi, size=4)
        s = np.where(rng.random(n) < 0.6, rng.choice(spots, size=n), rng.integers(0, hi, size=n))
    else:
        s = rng.integers(0, hi, size=n)
    s = s.astype(np.uint32)
    return s, (s + np.uint32(span - 1)).astype(np.uint32)


def _ran_event_sweep(solver):
    return any(name.startswith("k_sweep_uniform_ev") for name in solver.kernel_times())


def _check(solver, oracle, s, e, lengths, M, offs=None, expect_ev=True, **extra_env):
    want = oracle.solve(s, e, lengths, M, contig_read_offsets=offs)
    solver.set_profiling(1)
    try:
        with forced(solver, QMCP_HIP_SWEEP="ev", **extra_env):
            got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
        ran = _ran_event_sweep(solver)
    finally:
        solver.set_profiling(0)
    assert np.array_equal(got, want)
    assert ran == expect_ev
    return got


@pytest.mark.parametrize("span", [32, 33, 63, 64, 65, 100, 128, 129, 150, 192, 193, 200, 256])
def test_every_lane_layout_deep_and_multi_contig(pkg, oracle, solver, span):
    rng = np.random.default_rng(span)
    lengths = np.array([40_000 + span, 3 * span + 1, 25_000, span], np.uint32)
    counts = [300_000, 2_000, 150_000, 500]
    ss, ee = zip(*[_reads(rng, c, int(L), span) for c, L in zip(counts, lengths)])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    M = 10 if span > 192 else 40  # M must stay below the 6-bit counts of the four-slot layout
    _check(solver, oracle, s, e, lengths, M, offs)
    assert solver.last_stats.path == pkg.PATH_UNIFORM


@pytest.mark.parametrize("M", [1, 7, 100, 510])
def test_cfg2_shape_with_hot_spots_and_piled_ends(pkg, oracle, solver, M):
    rng = np.random.default_rng(1000 + M)
    L, span = 30_000, 150
    s, e = _reads(rng, 600_000, L, span, hot=True)
    s[:3000] = L - span
    e[:3000] = L - 1
    s[3000:5000] = 0
    e[3000:5000] = span - 1
    _check(solver, oracle, s, e, L, M)


def test_cap_beyond_the_fields_takes_the_scan_kernels(pkg, oracle, solver):
    rng = np.random.default_rng(5)
    s, e = _reads(rng, 400_000, 20_000, 150)
    _check(solver, oracle, s, e, 20_000, 511, expect_ev=False)   # three slots per lane: 9-bit counts
    s, e = _reads(rng, 400_000, 20_000, 200)
    _check(solver, oracle, s, e, 20_000, 63, expect_ev=False)    # four slots per lane: 6-bit counts
    _check(solver, oracle, s, e, 20_000, 62)


@pytest.mark.parametrize("depth", [0.3, 1.0, 2.0, 6.0])
def test_shallow_and_gapped_data_every_block_flagged(pkg, oracle, solver, depth):
    """coverage around the cap: hardly any block is 'deep', pushed-back amounts cross block borders all
    the time (the hand-back loop), islands separated by empty gaps"""
    rng = np.random.default_rng(int(depth * 10))
    span, M = 100, 12
    lengths = np.array([120_000, 90_001], np.uint32)
    ss, counts = [], []
    for L in lengths:
        n = int(L * M * depth / span)
        s = rng.integers(0, int(L) - span + 1, size=n)
        s = s[(s // 5000) % 4 != 2]                   # empty gaps
        ss.append(s.astype(np.uint32))
        counts.append(s.size)
    s = np.concatenate(ss)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    for cuts in ("0", "1"):
        _check(solver, oracle, s, e, lengths, M, offs, QMCP_HIP_CUTS=cuts)
    if depth <= 1.0:
        assert solver.last_stats.sweep_stretches > 2   # the last run split at cut points


def test_equals_the_block_scan_kernels_on_a_ranked_call(pkg, oracle, solver):
    """4.2 M reads (range-ranked route) at 21 x M coverage: the event sweep is the host's own choice"""
    rng = np.random.default_rng(77)
    lengths = np.array([180_000, 120_000], np.uint32)
    counts = [2_600_000, 1_600_000 + 4_321]
    ss, ee = zip(*[_reads(rng, c, int(L), 150) for c, L in zip(counts, lengths)])
    s, e = np.concatenate(ss), np.concatenate(ee)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    solver.set_profiling(1)
    try:
        own = solver.solve(s, e, lengths, 100, contig_read_offsets=offs)
        assert _ran_event_sweep(solver)
        assert solver.last_stats.sort_passes == 1
    finally:
        solver.set_profiling(0)
    with forced(solver, QMCP_HIP_SWEEP="fast"):
        fast = solver.solve(s, e, lengths, 100, contig_read_offsets=offs)
    with forced(solver, QMCP_HIP_SWEEP="gen"):
        gen = solver.solve(s, e, lengths, 100, contig_read_offsets=offs)
    assert np.array_equal(own, fast) and np.array_equal(own, gen)
    assert np.array_equal(own, oracle.solve(s, e, lengths, 100, contig_read_offsets=offs))


@pytest.mark.parametrize("seed", range(30))
def test_random_cases_forced_event_sweep(pkg, oracle, solver, seed):
    rng = np.random.default_rng(424_200 + seed)
    span = int(rng.choice([32, 50, 64, 100, 150, 151, 200, 256]))
    n_contigs = int(rng.integers(1, 6))
    M = int(rng.choice([1, 2, 5, 17, 40])) if span > 192 else int(rng.choice([1, 2, 5, 17, 50, 100, 300]))
    lengths, counts, ss = [], [], []
    for _ in range(n_contigs):
        L = int(rng.integers(span, 60_000))
        depth = float(rng.choice([0.5, 1.0, 2.0, 8.0, 30.0]))
        c = min(int(L * M * depth / span) + 1, 300_000)
        if rng.random() < 0.15:
            c = 0
        s, _ = _reads(rng, c, L, span, hot=rng.random() < 0.3)
        lengths.append(L)
        counts.append(c)
        ss.append(s)
    if sum(counts) == 0:
        return
    s = np.concatenate(ss)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    _check(solver, oracle, s, e, np.array(lengths, np.uint32), M, offs,
           QMCP_HIP_CUTS=str(int(rng.integers(0, 2))))


@pytest.mark.parametrize("span", [32, 65, 97, 150, 193, 256])
def test_tiny_genomes_of_one_to_eight_blocks(pkg, oracle, solver, span):
    """genomes of a few blocks: the first block, a cut last block and nothing between (a hipcc fold of
    `x >= ell ? min(x - ell, ltot) : 0` once read the wrong bucket offset here)"""
    for f in (1.0, 1.01, 1.5, 1.95, 2.0, 2.05, 2.5, 3.99, 4.0, 4.01, 8.2):
        L = max(span, int(f * span))
        for M in (1, 11):
            rng = np.random.default_rng(int(f * 100) + span + M)
            s, e = _reads(rng, 3000, L, span)
            _check(solver, oracle, s, e, L, M)


def test_host_picks_the_sweep_by_how_spiky_the_starts_are(pkg, oracle, solver):
    """deep data whose reads start in a few windows (amplicon panels) leaves most blocks with an empty start
    position: the event-driven chain would pay for every one of them, so the host -- told the number of empty
    positions by k_range_offsets (of this call if its shape is new to the context, else of the previous call of
    the shape) -- takes the block-scan pipeline there and the event-driven form on data that starts reads
    everywhere; same answers either way"""
    import workloads
    s, e, a0, a1, _ = workloads.amplicon_reads(400_000, seed=5, straddle_fraction=0.0)   # 800 k reads: ranked route
    solver.set_profiling(1)
    try:
        got = solver.solve(s, e, 29_903, 200)
        spiky_kernels = set(solver.kernel_times())
        rng = np.random.default_rng(2)
        s2, e2 = _reads(rng, 800_002, 29_903, 150)   # (another shape: the count is taken afresh)
        solver.set_profiling(1)
        got2 = solver.solve(s2, e2, 29_903, 200)
        flat_kernels = set(solver.kernel_times())
    finally:
        solver.set_profiling(0)
    assert "k_sweep_uniform_mw" in spiky_kernels and "k_sweep_uniform_ev" not in spiky_kernels
    assert "k_sweep_uniform_ev" in flat_kernels and "k_sweep_uniform_mw" not in flat_kernels
    assert np.array_equal(got, oracle.solve(s, e, 29_903, 200))
    assert np.array_equal(got2, oracle.solve(s2, e2, 29_903, 200))
