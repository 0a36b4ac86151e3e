"""tiny problems through the event-driven sweep (found a hipcc select fold: boff index below ell)"""
import sys, importlib, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/oracle"); sys.path.insert(0, R + "/tests")
pkg = importlib.import_module('genome-downsampler_amd')
import oracle_py
sv = pkg.Solver(0)
os.environ["QMCP_HIP_SWEEP"] = "ev"
bad = 0
def run(L, ell, n, M, seed=0):
    global bad
    rng = np.random.default_rng(seed)
    s = rng.integers(0, L - ell + 1, size=n).astype(np.uint32); e = (s + ell - 1).astype(np.uint32)
    want = oracle_py.solve(s, e, L, M); got = sv.solve(s, e, L, M)
    ok = np.array_equal(got, want); bad += 0 if ok else 1
    if not ok: print("L %5d ell %3d n %6d M %3d  MISMATCH kept %d want %d" % (L, ell, n, M, sv.last_stats.n_kept, int(np.unpackbits(want.view(np.uint8)).sum())), flush=True)
for ell in (32, 33, 64, 65, 96, 97, 98, 128, 129, 150, 192, 193, 256):
    for f in (1.0, 1.01, 1.5, 1.95, 2.0, 2.05, 2.5, 3.0, 3.99, 4.0, 4.01, 5.5, 8.2):
        for M in (1, 11):
            run(max(ell, int(f * ell)), ell, 3000, M, seed=int(f * 100) + ell)
print("done, mismatches:", bad)
