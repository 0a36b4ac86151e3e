"""qmcp_hip_options: a context's routes and kernels are forced through ONE call of the ABI (the counterpart of the
reference solver's setters, libs/qmcp-solver/include/qmcp-solver/quasi_mcp_cuda_max_flow_solver.hpp:30-31); the
environment variables of the same names are a debug override read once, when the context is created."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_defaults_set_get_and_restore(pkg):
    with pkg.Solver(0) as sv:
        o = sv.get_options()
        assert o.struct_size == C.sizeof(pkg.Options) and o.sweep == 0 and o.pass_major == 0 and o.near_uniform == 0
        sv.set_options(sweep="ev", cut_points=1, speculation_run_in=64, near_uniform=-1, host_threads=3)
        o = sv.get_options()
        assert (o.sweep, o.cut_points, o.speculation_run_in, o.near_uniform, o.host_threads) == (pkg.SWEEP_EVENTS, 1, 64, -1, 3)
        with sv.options(sweep="gen"):
            assert sv.get_options().sweep == pkg.SWEEP_GENERAL and sv.get_options().cut_points == 1
        assert sv.get_options().sweep == pkg.SWEEP_EVENTS
        sv.set_options()
        assert sv.get_options().sweep == 0 and sv.get_options().host_threads == 0


def test_bad_options_are_refused(pkg):
    with pkg.Solver(0) as sv:
        o = sv.get_options()
        o.struct_size = C.sizeof(pkg.Options) + 8
        assert pkg._hip.qmcp_hip_set_options(sv._ctx, C.byref(o)) == -1      # QMCP_EINVAL
        o.struct_size = C.sizeof(pkg.Options)
        o.sweep = 9
        assert pkg._hip.qmcp_hip_set_options(sv._ctx, C.byref(o)) == -1
        with pytest.raises(AttributeError):
            sv.set_options(no_such_field=1)


def test_an_older_callers_shorter_struct(pkg):
    """struct_size smaller than this build's: the fields it has are taken, the rest stay `the library chooses`"""
    with pkg.Solver(0) as sv:
        sv.set_options(host_both_columns=1)
        o = pkg.Options()
        o.struct_size = pkg.Options.cut_points.offset + 4      # ... up to and including cut_points
        o.sweep, o.cut_points, o.host_both_columns = pkg.SWEEP_FAST, -1, 1
        assert pkg._hip.qmcp_hip_set_options(sv._ctx, C.byref(o)) == 0
        got = sv.get_options()
        assert (got.sweep, got.cut_points, got.host_both_columns) == (pkg.SWEEP_FAST, -1, 0)


def test_environment_is_read_when_the_context_is_created_and_never_again(pkg, oracle, monkeypatch):
    monkeypatch.setenv("QMCP_HIP_SWEEP", "gen")
    monkeypatch.setenv("QMCP_HIP_NEAR", "0")
    with pkg.Solver(0) as sv:
        monkeypatch.delenv("QMCP_HIP_SWEEP")
        monkeypatch.setenv("QMCP_HIP_CUTS", "1")          # after the fact: changes nothing
        o = sv.get_options()
        assert o.sweep == pkg.SWEEP_GENERAL and o.near_uniform == -1 and o.cut_points == 0
        rng = np.random.default_rng(1)
        s = rng.integers(0, 30_000 - 150, size=200_000).astype(np.uint32)
        e = (s + 149).astype(np.uint32)
        sv.set_profiling(1)
        got = sv.solve(s, e, 30_000, 20)
        names = set(sv.kernel_times())
        sv.set_profiling(0)
        assert np.array_equal(got, oracle.solve(s, e, 30_000, 20))
        assert any(n.startswith("k_sweep_uniform_gen") for n in names) and not any("uniform_ev" in n for n in names)
