import os, sys, importlib
import numpy as np
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module('genome-downsampler_amd')
L=60_000_000
rng=np.random.default_rng(4)
for M,depth in ((100,2.0),(100,3.0),(200,2.0),(20,3.0)):
    n=int(depth*M*L/150)
    s=rng.integers(0,L-150,size=n).astype(np.uint32); e=(s+149).astype(np.uint32)
    lengths=np.array([L],np.uint32)
    with pkg.Solver(0) as sv:
        ref=None
        for run_in in (0,640,1536):
            with sv.options(speculation_run_in=run_in):
                m=sv.solve(s,e,lengths,M); m=sv.solve(s,e,lengths,M)
                d=sv.last_stats.as_dict()
            same = True if ref is None else bool(np.array_equal(ref,m))
            if ref is None: ref=m.copy()
            print(f"one length, M {M} depth {depth}: run-in {run_in:5d}: {d['ms_total']:8.2f} ms sweep {d['ms_sweep']:8.2f} stretches {d['sweep_stretches']:5d} boundaries {d['spec_boundaries']:4d} disagreeing {d['spec_mismatches']:3d} / {d['spec_retry_mismatches']:3d} same {same}",flush=True)
