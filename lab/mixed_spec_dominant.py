"""lab: the mixed-span route's speculative boundaries on data with ONE dominant read length (near-uniform route off):
a single clipped read, 1 % clipped, and a broad mix of lengths, at several run-ins -- boundaries that disagree.
   python lab/mixed_spec_dominant.py [L = 4_000_000] [M = 20] [depth = 2.0]"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20
depth = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
rng = np.random.default_rng(11)
n = int(depth * M * L / 150)
s0 = rng.integers(0, L - 150, size=n).astype(np.uint32)
e0 = (s0 + 149).astype(np.uint32)
def clipped(frac):
    s, e = s0.copy(), e0.copy()
    pick = np.flatnonzero(rng.random(n) < frac) if frac > 0 else np.array([n // 2])
    clip = rng.integers(1, 51, size=pick.size).astype(np.uint32)
    s[pick[::2]] += clip[::2]; e[pick[1::2]] -= clip[1::2]
    return s, e
def broad():
    return s0.copy(), (s0 + rng.integers(99, 150, size=n).astype(np.uint32)).astype(np.uint32)
lengths = np.array([L], np.uint32)
ref = {}
with pkg.Solver(0) as sv:
    for name, (s, e) in (("one clipped read", clipped(0)), ("1 % clipped", clipped(0.01)), ("spans 100..150", broad())):
        for run_in in (0, 64, 256, 1024):
            with sv.options(near_uniform=-1, speculation=1, speculation_run_in=run_in):
                m = sv.solve(s, e, lengths, M)
                d = sv.last_stats.as_dict()
            key = name
            same = True
            if key in ref: same = bool(np.array_equal(ref[key], m))
            else: ref[key] = m.copy()
            print(f"{name:18s} run-in {run_in:5d}: path {d['path']} {d['ms_total']:8.2f} ms stretches {d['sweep_stretches']:5d} boundaries {d['spec_boundaries']:4d} "
                  f"disagreeing {d['spec_mismatches']:4d} / second tier {d['spec_retry_mismatches']:4d}  same mask {same}", flush=True)
