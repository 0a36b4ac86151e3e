"""lab: one read length, large M (deep in standard deviations while a few times M), one contig: boundaries that disagree at the
table's run-in and at longer ones.   python lab/spec_run_in_large_M.py"""
import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo')
pkg=importlib.import_module('genome-downsampler_amd')
L=20_000_000
rng=np.random.default_rng(4)
for M,depth in ((400,2.0),(400,3.0),(200,4.0)):
    n=int(depth*M*L/150)
    s=rng.integers(0,L-150,size=n).astype(np.uint32); e=(s+149).astype(np.uint32); lengths=np.array([L],np.uint32)
    with pkg.Solver(0) as sv:
        for run_in in (0,2304,4608,9216):
            with sv.options(speculation_run_in=run_in):
                sv.solve(s,e,lengths,M); m=sv.solve(s,e,lengths,M); d=sv.last_stats.as_dict()
            print(f"one length, M {M} depth {depth}: run-in {run_in:5d}: {d['ms_total']:8.2f} ms sweep {d['ms_sweep']:8.2f} stretches {d['sweep_stretches']:4d} boundaries {d['spec_boundaries']:4d} disagreeing {d['spec_mismatches']:3d} / {d['spec_retry_mismatches']:3d}",flush=True)
        with sv.options(speculation=-1):
            sv.solve(s,e,lengths,M); d=sv.last_stats.as_dict()
        print(f"one length, M {M} depth {depth}: no speculation: {d['ms_total']:8.2f} ms sweep {d['ms_sweep']:8.2f} stretches {d['sweep_stretches']}",flush=True)
