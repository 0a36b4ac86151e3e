"""bench.py --gpus N without a launcher starts its own ranks (CPU side of the check: no GPU here, so both child ranks
must come up, find no device and say so -- which shows that the ranks were started and that the exit code is relayed;
the working run is tests/test_gpu_bench_contract.py::test_plain_gpus_2_starts_its_own_ranks)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU-side test covers the working run")
def test_gpus_2_without_world_size_starts_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "cfg2",
                          "--dist-backend", "gloo", "--single-device"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "needs torch.distributed.run" not in out.stderr
    assert out.stderr.count("bench.py needs an MI355X") >= 2, out.stderr[-2000:]
