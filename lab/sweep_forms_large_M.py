import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo')
pkg=importlib.import_module('genome-downsampler_amd')
L=20_000_000
rng=np.random.default_rng(4)
for M,depth in ((400,3.0),(200,4.0),(100,8.0),(400,2.0)):
    n=int(depth*M*L/150)
    s=rng.integers(0,L-150,size=n).astype(np.uint32); e=(s+149).astype(np.uint32); lengths=np.array([L],np.uint32)
    with pkg.Solver(0) as sv:
        for name,opt in (("library",{}),("events",dict(sweep=pkg.SWEEP_EVENTS)),("fast",dict(sweep=pkg.SWEEP_FAST)),("general, no speculation",dict(sweep=pkg.SWEEP_GENERAL,speculation=-1))):
            with sv.options(**opt):
                sv.solve(s,e,lengths,M); m=sv.solve(s,e,lengths,M); d=sv.last_stats.as_dict()
            print(f"M {M} depth {depth}: {name:24s}: {d['ms_total']:8.2f} ms sweep {d['ms_sweep']:8.2f} stretches {d['sweep_stretches']:4d} changed {d['sweep_blocks_changed']} of {d['sweep_blocks']}",flush=True)
