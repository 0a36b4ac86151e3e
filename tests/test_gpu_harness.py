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
