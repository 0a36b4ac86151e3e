"""Routes and kernels forced through the ABI's options call (qmcp_hip_set_options), written with the names the library's
debug environment variables have -- the tests' vocabulary since round 1:
    with forced(solver, QMCP_HIP_CUTS="1", QMCP_HIP_SWEEP="ev"): solver.solve(...)
None leaves a choice to the library.  (The environment is read once, when a context is created; a test that changed it
afterwards would change nothing.)"""
from contextlib import contextmanager


def _tri(v):
    return 0 if v is None else (1 if str(v) == "1" else -1)


def fields(**env_style):
    out = {}
    for k, v in env_style.items():
        if k == "QMCP_HIP_PM": out["pass_major"] = _tri(v)
        elif k == "QMCP_HIP_SWEEP": out["sweep"] = v
        elif k == "QMCP_HIP_CUTS": out["cut_points"] = _tri(v)
        elif k == "QMCP_HIP_SPEC": out["speculation"] = _tri(v)
        elif k == "QMCP_HIP_SPEC_BURN": out["speculation_run_in"] = 0 if v is None else int(v)
        elif k == "QMCP_HIP_NEAR": out["near_uniform"] = 0 if v is None else (-1 if str(v) == "0" else 0)
        elif k == "QMCP_HIP_NEAR_MIN_DEPTH": out["near_uniform_min_depth"] = 0.0 if v is None else float(v)
        elif k == "QMCP_HIP_NO_RANK": out["force_sort_route"] = 0 if v is None else 1
        elif k == "QMCP_HIP_GENERAL_LDS": out["mixed_sweep_in_lds"] = 0 if v is None else 1
        elif k == "QMCP_HIP_EXPAND": out["keep_expand"] = 0 if v is None else 1
        else: raise KeyError(k)
    return out


@contextmanager
def forced(solver, **env_style):
    with solver.options(**fields(**env_style)):
        yield
