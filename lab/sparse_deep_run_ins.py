import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo')
pkg=importlib.import_module('genome-downsampler_amd')
for (L,M,depth) in ((20_500_000,10,20.0),(40_000_000,10,12.0),(30_000_000,20,15.0)):
    rng=np.random.default_rng(2); n=int(depth*M*L/150)
    s=rng.integers(0,L-174,size=n).astype(np.uint32); e=(s+149).astype(np.uint32); lengths=np.array([L],np.uint32)
    with pkg.Solver(0) as sv:
        for run_in in (0,768,1536,2304,4608):
            with sv.options(speculation_run_in=run_in):
                sv.solve(s,e,lengths,M); m=sv.solve(s,e,lengths,M); d=sv.last_stats.as_dict()
            print(f"L {L} M {M} depth {depth}: run-in {run_in:5d}: {d['ms_total']:8.2f} ms sweep {d['ms_sweep']:8.2f} stretches {d['sweep_stretches']:4d} boundaries {d['spec_boundaries']:4d} disagreeing {d['spec_mismatches']:3d} / {d['spec_retry_mismatches']:3d}",flush=True)
