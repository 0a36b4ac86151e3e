"""The reference's `test -a <solver>` flow through the C++ plugin surface
(SolverManager -> qmcp::Solver::solve(M, BamApi&) -> find_filtered_cover -> validity)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_coverage_tester_restatement_all_five_cases():
    exe = os.path.join(ROOT, "tests", "cpp", "coverage_harness")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "harness"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("PASSED") == 10 and "FAILED" not in out.stdout
    assert "differ" not in out.stdout


def test_cpp_adapter_matches_c_abi(pkg, oracle, solver):
    s, e = pkg.reads_gen(pkg.KIND_LOW_BOTH_SIDES, 20000, 5000)
    assert pkg.solver_names() == ["quasi-mcp-hip"]
    ids = pkg.host_solve("quasi-mcp-hip", s, e, 5000, 40)
    want = pkg.mask_to_indices(oracle.solve(s, e, 5000, 40), s.size)
    assert np.array_equal(ids, want)
    assert np.all(np.diff(ids.astype(np.int64)) > 0)  # ascending ReadIndex, like obtain_sequence
    # src/app.cpp:141: find_pairs after the solve -- host BamApi::find_pairs vs the mask kernel
    paired = np.sort(pkg.host_solve("quasi-mcp-hip", s, e, 5000, 40, with_pairs=True))
    m = solver.complete_pairs(solver.solve(s, e, 5000, 40), s.size)
    assert np.array_equal(paired, pkg.mask_to_indices(m, s.size))
    assert np.array_equal(m, oracle.find_pairs(oracle.solve(s, e, 5000, 40), s.size))
    with pytest.raises(KeyError):
        pkg.host_solve("quasi-mcp-cpu", s, e, 5000, 40)


def test_plugin_boundary_from_size_t_columns(pkg, oracle):
    """BamApi(SOAPairedReads) -> SolverManager -> QuasiMcpHipSolver::solve: the 64-bit columns are narrowed
    inside the library (threads + pinned staging, qmcp_hip_solve_host64); same Solution as the oracle"""
    import numpy as np
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 1_600_000, 300_000, seed=5)   # 3.2 M reads: 13 chunks of 256 Ki, several threads
    kept, times = pkg.plugin_solve_timed("quasi-mcp-hip", s, e, 300_000, 100)
    want = pkg.mask_to_indices(oracle.solve(s, e, 300_000, 100), s.size)
    assert np.array_equal(kept, want)
    assert times["chunks"] == 13 and 1 <= times["host_threads"] <= 13 and times["solve_call_ms"] >= times["library_ms"] > 0
    kept1, _ = pkg.plugin_solve_timed("quasi-mcp-hip", s[:10], e[:10], 300_000, 3)   # a partial chunk
    assert np.array_equal(kept1, pkg.mask_to_indices(oracle.solve(s[:10], e[:10], 300_000, 3), 10))


def test_plugin_boundary_sends_one_column_when_every_span_is_the_same(pkg, oracle):
    """qmcp_hip_solve_host64: reads of one span cross the link as starts only and the device rebuilds the
    ends; mixed spans seen in the first reads send both columns; a call that looks uniform at first and is
    not sends its ends in a second pass.  Same Solution as the oracle each time."""
    L, M = 200_000, 60
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 400_000, L, seed=9)          # 800 k reads, 4 chunks
    kept, t = pkg.plugin_solve_timed("quasi-mcp-hip", s, e, L, M)
    assert t["columns_sent"] == 1
    assert np.array_equal(kept, pkg.mask_to_indices(oracle.solve(s, e, L, M), s.size))
    # mixed from the first reads on
    e2 = e.copy()
    e2[::7] -= 11
    kept, t = pkg.plugin_solve_timed("quasi-mcp-hip", s, e2, L, M)
    assert t["columns_sent"] == 2
    assert np.array_equal(kept, pkg.mask_to_indices(oracle.solve(s, e2, L, M), s.size))
    # uniform through the probe and three chunks, one shorter read in the last chunk
    e3 = e.copy()
    e3[-5] -= 3
    kept, t = pkg.plugin_solve_timed("quasi-mcp-hip", s, e3, L, M)
    assert t["columns_sent"] == 2
    assert np.array_equal(kept, pkg.mask_to_indices(oracle.solve(s, e3, L, M), s.size))


def test_size_t_entry_reports_what_the_device_entry_reports(pkg, oracle, solver):
    """qmcp_hip_solve_host64 through ctypes: the one-column shortcut does not hide an invalid read, a
    coordinate beyond 32 bits is refused, empty input is fine"""
    L, M = 50_000, 20
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 30_000, L, seed=3)
    s64, e64 = s.astype(np.uint64), e.astype(np.uint64)
    got = solver.solve64(s64, e64, L, M)
    assert solver.last_breakdown.columns_sent == 1
    assert np.array_equal(got, oracle.solve(s, e, L, M))
    bad_s, bad_e = s64.copy(), e64.copy()
    bad_s[0], bad_e[0] = e64[0], s64[0]                       # start > end in the very first read
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve64(bad_s, bad_e, L, M)
    assert err.value.code == pkg.QMCP_EREAD
    bad_e = e64.copy()
    bad_e[12345] = L                                          # end == L, span differs there
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve64(s64, bad_e, L, M)
    assert err.value.code == pkg.QMCP_EREAD
    big = e64.copy()
    big[777] += 1 << 32
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve64(s64, big, L, M)
    assert err.value.code == pkg.QMCP_ERANGE
    assert solver.solve64(np.zeros(0, np.uint64), np.zeros(0, np.uint64), L, M).size == 0
    # and the context is still good
    assert np.array_equal(solver.solve64(s64, e64, L, M), got)


def test_host_entry_sends_one_column_when_every_span_is_the_same(pkg, oracle, solver):
    """qmcp_hip_solve_host (uint32 columns), calls of 2^20 reads or more: host threads check the spans while
    the starts are copied and the ends stay behind when they are all the same"""
    L, M = 150_000, 50
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, 600_000, L, seed=21)         # 1.2 M reads
    want = oracle.solve(s, e, L, M)
    assert np.array_equal(solver.solve(s, e, L, M), want) and solver.last_stats.columns_sent == 1
    e2 = e.copy()
    e2[-1] -= 5                                                          # the very last read is shorter
    assert np.array_equal(solver.solve(s, e2, L, M), oracle.solve(s, e2, L, M))
    # (one shorter read on deep data: the near-uniform route, or the mixed-span one -- not the one-span route)
    assert solver.last_stats.columns_sent == 2 and solver.last_stats.path in (pkg.PATH_GENERAL, pkg.PATH_NEAR_UNIFORM)
    k = 1 << 19                                                          # below the threshold: both columns
    assert np.array_equal(solver.solve(s[:k], e[:k], L, M), oracle.solve(s[:k], e[:k], L, M))
    assert solver.last_stats.columns_sent == 2
    bad = e.copy()
    bad[900_000] = L                                                     # invalid read far from the probe
    with pytest.raises(pkg.QmcpError) as err:
        solver.solve(s, bad, L, M)
    assert err.value.code == pkg.QMCP_EREAD
