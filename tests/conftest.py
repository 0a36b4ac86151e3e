import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [
        os.path.join(ROOT, "genome-downsampler_amd", "lib", "libqmcp_hip.so"),
        os.path.join(ROOT, "genome-downsampler_amd", "lib", "libqmcp_host.so"),
        os.path.join(ROOT, "oracle", "libqmcp_oracle.so"),
    ]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-C", ROOT, "lib", "oracle"], check=True, stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("genome-downsampler_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def solver(pkg):
    """one context reused across tests, like one reference solver instance across cases"""
    if pkg.device_count() < 1:
        pytest.fail("gpu test collected but no HIP device is visible (no CPU fallback exists)")
    s = pkg.Solver(0)
    yield s
    s.close()


def random_reads(rng, n, L, min_span, max_span):
    """variable-span reads on [0, L)"""
    max_span = min(max_span, L)
    min_span = min(min_span, max_span)
    span = rng.integers(min_span, max_span + 1, size=n, dtype=np.int64)
    start = (rng.random(n) * (L - span + 1)).astype(np.int64)
    return start.astype(np.uint32), (start + span - 1).astype(np.uint32)
