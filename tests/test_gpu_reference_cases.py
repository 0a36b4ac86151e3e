"""The reference's own test-suite (src/tests/coverage_tester.cpp:28-43,109-175) restated:
five inputs, the validity invariant min(in_cov, M) <= out_cov, plus what the reference never
pins -- bit-identity with the oracle's canonical maximum flow."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # name, reads-gen kind, pairs, L, M   (coverage_tester.cpp:120-175)
    ("random_uniform_dist_test", 0, 1_000_000, 30_000, 1000),
    ("random_low_coverage_on_both_sides_test", 1, 1_000_000, 30_000, 8000),
    ("random_with_hole_test", 2, 1_000_000, 30_000, 8000),
    ("random_zero_coverage_on_both_sides_test", 3, 1_000_000, 30_000, 8000),
]


@pytest.mark.parametrize("name,kind,pairs,L,M", CASES, ids=[c[0] for c in CASES])
def test_reference_case(pkg, oracle, solver, name, kind, pairs, L, M):
    s, e = pkg.reads_gen(kind, pairs, L)
    got = solver.solve(s, e, L, M)
    st = solver.last_stats
    assert st.path == pkg.PATH_UNIFORM and st.min_span == 150
    # the reference's assertion, computed by the device coverage probes ...
    in_cov = solver.coverage(s, e, L)
    out_cov = solver.coverage(s, e, L, keep_mask=got)
    assert oracle.is_out_cover_valid(in_cov, out_cov, M)
    # ... which themselves must equal BamApi::find_input_cover / find_filtered_cover
    assert np.array_equal(in_cov, oracle.cover(s, e, L))
    assert np.array_equal(out_cov, oracle.cover(s, e, L, keep_mask=got))
    # the device's b and d == create_b_function / create_demand_function (per-base loop, in-place
    # differences with the i < n bound: quasi_mcp_cpu_max_flow_solver.cpp:58-87)
    b_dev, d_dev = solver.demand(s, e, L, M)
    assert np.array_equal(b_dev, oracle.b_function(s, e, L, M))
    assert np.array_equal(d_dev, oracle.demand_function(s, e, L, M))
    # bit-identical kept set and a valid maximum flow of the reference network
    want = oracle.solve(s, e, L, M)
    assert np.array_equal(got, want)
    ok, value = oracle.check_flow(s, e, L, M, got)
    assert ok and value == oracle.graph(s, e, L, M).total_supply


def test_small_fixture_b_and_d_on_the_device(pkg, oracle, solver):
    """the reference's 16-read example (src/tests/coverage_tester.cpp:72-93): b and d from the device
    equal the golden vectors captured from the reference's own code (tests/golden)"""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
    fx = gold["small_fixture"]
    s = np.array(fx["starts"], np.uint32)
    e = np.array(fx["ends"], np.uint32)
    b_dev, d_dev = solver.demand(s, e, fx["ref_genome_length"], fx["M"])
    assert b_dev.tolist() == fx["b"] and d_dev.tolist() == fx["d"]
    for M in (0, 1, 3, 100):
        b_dev, d_dev = solver.demand(s, e, fx["ref_genome_length"], M)
        assert np.array_equal(b_dev, oracle.b_function(s, e, fx["ref_genome_length"], M))
        assert np.array_equal(d_dev, oracle.demand_function(s, e, fx["ref_genome_length"], M))


def test_cfg2_one_million_reads(pkg, oracle, solver):
    # BASELINE.json configs[1]: 1 M-read single-contig, M = 100
    s, e = pkg.reads_gen(0, 500_000, 30_000)
    got = solver.solve(s, e, 30_000, 100)
    assert np.array_equal(got, oracle.solve(s, e, 30_000, 100))
    assert solver.last_stats.n_kept == int(np.unpackbits(got.view(np.uint8)).sum())
    again = solver.solve(s, e, 30_000, 100)
    assert np.array_equal(got, again)


def test_solver_instance_reuse_across_cases(pkg, oracle, solver):
    """one instance, many solves of different shapes (coverage_tester.cpp:30-34)"""
    rng = np.random.default_rng(0)
    for L, n, M in [(30_000, 200_000, 50), (500, 100, 3), (100_000, 1_000_000, 200), (11, 16, 4)]:
        s = rng.integers(0, max(L - 10, 1), size=n).astype(np.uint32)
        e = np.minimum(s + rng.integers(0, 10, size=n), L - 1).astype(np.uint32)
        assert np.array_equal(solver.solve(s, e, L, M), oracle.solve(s, e, L, M))


def test_cfg4_shard_overlapped_sweep_path(pkg, oracle, solver):
    """one contig of BASELINE.json configs[3] (12.5 M reads, L = 1e6, M = 100): large enough for the
    two-stream path (early counts + sweep beside the radix passes); bit-identical to the oracle"""
    s, e = pkg.reads_gen(0, 6_250_000, 1_000_000)
    got = solver.solve(s, e, 1_000_000, 100)
    assert solver.last_stats.path == pkg.PATH_UNIFORM
    assert np.array_equal(got, oracle.solve(s, e, 1_000_000, 100))
    # two contigs, different depths, same path
    s2, e2 = pkg.reads_gen(1, 2_500_000, 400_000, seed=7)
    S, E = np.concatenate([s, s2]), np.concatenate([e, e2])
    offs = np.array([0, s.size, s.size + s2.size], np.uint64)
    lens = np.array([1_000_000, 400_000], np.uint32)
    got = solver.solve(S, E, lens, 60, contig_read_offsets=offs)
    assert np.array_equal(got, oracle.solve(S, E, lens, 60, contig_read_offsets=offs))


def test_profiling_records_kernels_only_when_on(pkg, oracle):
    """kernel brackets (qmcp_hip_set_profiling) cost nothing when off and change no result"""
    s, e = pkg.reads_gen(0, 1_000_000, 200_000, seed=99)
    with pkg.Solver(0) as sv:
        got = sv.solve(s, e, 200_000, 30)
        assert sv.kernel_times() == {}  # profiling off: nothing recorded
        sv.set_profiling(True)
        again = sv.solve(s, e, 200_000, 30)
        assert "k_pm_prepare_sort" in sv.kernel_times() or "k_prepare" in sv.kernel_times()
    assert np.array_equal(got, again)
    assert np.array_equal(got, oracle.solve(s, e, 200_000, 30))
