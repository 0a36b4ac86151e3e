"""Ad-hoc: many more random cases than tests/test_gpu_random_stress.py runs by default.
   python lab/stress_more.py <first seed> <last seed>   -- every seed runs a deep case (all routes), a shallow
   uniform case split at cut points and the same with mixed spans"""
import importlib, os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
from test_gpu_random_stress import _case, _shallow_case
pkg = importlib.import_module("genome-downsampler_amd")
sol = pkg.Solver(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
routes, stretches = {}, 0


def check(tag, seed, s, e, lengths, offs, M):
    global bad, stretches
    got = sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle_py.solve(s, e, lengths, M, contig_read_offsets=offs)
    key = (tag, sol.last_stats.path, sol.last_stats.sort_passes)
    routes[key] = routes.get(key, 0) + 1
    stretches += sol.last_stats.sweep_stretches
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", tag, "seed", seed, s.size, lengths.tolist(), M, flush=True)


for seed in range(lo, hi):
    sol.set_options()
    check("deep", seed, *_case(np.random.default_rng(seed)))
    sol.set_options(cut_points=1)
    s, e, lengths, offs, M = _shallow_case(np.random.default_rng(1_000_000 + seed))
    check("shallow", seed, s, e, lengths, offs, M)
    rng = np.random.default_rng(2_000_000 + seed)
    span = int(e[0] - s[0]) + 1
    e2 = (e - rng.integers(0, max(span // 2, 1), size=s.size).astype(np.uint32)).astype(np.uint32)
    check("shallow mixed", seed, s, e2, lengths, offs, M)
    if (seed - lo) % 50 == 49:
        print("...", seed - lo + 1, "seeds, mismatches", bad, flush=True)
print("seeds", hi - lo, "mismatches", bad, "stretches swept", stretches, "routes (kind, path, sort passes):", routes)
