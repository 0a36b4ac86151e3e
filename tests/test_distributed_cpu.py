"""N > 1 composition on CPU: two gloo ranks shard the contigs, each solves its share, the keep
bitmasks are all-gathered and merged into global ReadIndex order.  The per-rank solve is the
oracle here (no GPU in this container; on the GPU box bench.py runs the same composition with
the HIP solver and the nccl == RCCL backend)."""
import importlib
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    from conftest import random_reads
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)  # same problem on every rank
        lens = [3000, 500, 8000, 150, 2000]
        counts = [4000, 300, 9000, 0, 2500]
        parts = [random_reads(rng, c, L, 150, 150) if i != 2 else random_reads(rng, c, L, 40, 200)
                 for i, (c, L) in enumerate(zip(counts, lens))]
        s = np.concatenate([p[0] for p in parts])
        e = np.concatenate([p[1] for p in parts])
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        M = 9
        owned = sh.assign_contigs(counts, world)
        ls, le, loffs, llens = sh.local_problem(s, e, offs, lens, owned[rank])
        local = oracle_py.solve(ls, le, llens, M, contig_read_offsets=loffs) if len(llens) else \
            np.zeros(0, np.uint64)
        max_words = max((sum(counts[c] for c in o) + 63) // 64 for o in owned)
        gathered = sh.gather_masks(local, max(max_words, 1), dist)
        merged = sh.merge_masks(gathered, owned, offs, s.size)
        whole = oracle_py.solve(s, e, np.array(lens, np.uint32), M, contig_read_offsets=offs)
        assert np.array_equal(merged, whole), f"rank {rank}: merged mask differs"
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_contig_sharding_and_mask_gather(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
