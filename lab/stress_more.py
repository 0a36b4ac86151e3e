"""Ad-hoc: many more random cases than tests/test_gpu_random_stress.py runs by default."""
import importlib, os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
from test_gpu_random_stress import _case
pkg = importlib.import_module("genome-downsampler_amd")
sol = pkg.Solver(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
routes = {}
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    s, e, lengths, offs, M = _case(rng)
    got = sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle_py.solve(s, e, lengths, M, contig_read_offsets=offs)
    routes[sol.last_stats.sort_passes] = routes.get(sol.last_stats.sort_passes, 0) + 1
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH seed", seed, s.size, lengths.tolist(), M)
print("cases", hi - lo, "mismatches", bad, "sort_passes histogram", routes)
