"""bench.py's contract with the driver, on a small workload: ONE JSON line on stdout with the metric, the
timing fields, `roofline` (dominant kernel, live launch average, whole-solve fraction, traffic source) and --
at N = 1 -- `cpu_baseline` with the parity spot check; and a two-rank gloo rehearsal of the N > 1 plumbing
(both ranks on this one GPU), gather at rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _one_json_line(stdout):
    """the bench line: exactly one line of stdout is JSON (gloo announces its ranks' connections on stdout too)"""
    lines = [ln for ln in stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_carries_what_the_driver_reads():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    b = _one_json_line(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    assert b["n_gpus"] == 1 and b["steps"] == 6 and b["warmup"] == 2 and b["unit"] == "Mreads/s" and b["vs_baseline"] is None
    assert b["value"] > 0 and abs(b["value"] - 1.0 / b["ms_per_step"] * 1e-3 * 1_000_000 * 1e-0) / b["value"] < 0.02   # 10^6 reads per step
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_launch_ms",
                "kernel_bound", "whole_solve"):
        assert key in r, key
    assert r["peak"] == 8000.0 and 0 < r["frac"] < 1 and 0 < r["whole_solve"]["frac"] < 1
    c = b["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and b["parity_vs_oracle_on_sample"] is True
    assert "host_entry" in b and "plugin_entry" in b
    # SURVEY 8(d)'s own definition beside the pipelined rate: N / time of one solve alone
    assert b["value_single_solve"] > 0 and b["single_solve_ms"] > 0
    assert abs(b["value_single_solve"] - 1.0 / b["single_solve_ms"] * 1e3) / b["value_single_solve"] < 0.02


def test_two_ranks_over_gloo_on_one_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "cfg2", "--dist-backend", "gloo",
                          "--single-device"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    b = _one_json_line(out.stdout)
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and "gather of the keep masks at rank 0" in b["config"]["workload"]
    assert b["value"] > 0 and "cpu_baseline" not in b
    # what the collective library saw, answerable from the line alone
    rr = b["config"]["rccl_ranks"]
    assert rr["backend"] == "gloo" and rr["world_size"] == 2 and sorted(r["rank"] for r in rr["ranks"]) == [0, 1]
    assert b["config"]["exchange"] == "gather"


def test_two_ranks_without_the_exchange():
    """--exchange none: the solves alone, so that a scaling run can tell them from the gather"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29573", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "4", "--warmup", "2", "--workload", "cfg2", "--dist-backend", "gloo",
                          "--single-device", "--exchange", "none"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    b = _one_json_line(out.stdout)
    assert b["n_gpus"] == 2 and b["config"]["exchange"] == "none" and b["value"] > 0


def test_plain_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it and no WORLD_SIZE in the environment (what the driver's
    plain command line is): the script starts torch.distributed.run as a child itself and relays rank 0's line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--workload", "cfg2", "--dist-backend", "gloo", "--single-device"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    b = _one_json_line(out.stdout)
    assert b["n_gpus"] == 2 and b["config"]["rccl_ranks"]["world_size"] == 2 and b["value"] > 0


def test_one_rank_through_rccl():
    """the RCCL calls of the N > 1 path -- init_process_group("nccl", device_id), the asynchronous gather of the keep
    masks, the barrier, the all-reduce of the elapsed time -- executed on a one-rank communicator (one GPU here)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for exchange in ("gather", "all_gather"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "4", "--warmup", "2",
                              "--workload", "cfg2", "--no-extras", "--exchange", exchange],
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        b = _one_json_line(out.stdout)
        rr = b["config"]["rccl_ranks"]
        assert rr["backend"] == "nccl" and rr["world_size"] == 1 and b["config"]["exchange"] == exchange and b["value"] > 0
