"""lab: deep data whose M does not fit a packed field of the event-driven sweep (M = 200, reads of 250), 24 contigs, 1 % clipped
   python lab/deep_large_M.py"""
import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
pkg=importlib.import_module('genome-downsampler_amd'); syn=importlib.import_module('genome-downsampler_amd.synthetic'); import oracle_py
rng=np.random.default_rng(22); ell=250; M=200; depth=12.0
lengths=rng.integers(150_000,950_000,size=24).astype(np.uint32)
counts=[int(depth*M*int(L)/ell) for L in lengths]
s0=np.concatenate([rng.integers(0,int(L)-ell-8,size=k).astype(np.uint32) for L,k in zip(lengths,counts)]); e0=(s0+np.uint32(ell-1)).astype(np.uint32)
offs=np.concatenate([[0],np.cumsum(counts)]).astype(np.uint64)
s1,e1=syn.clipped_mix(s0,e0,0.01)
with pkg.Solver(0) as sv:
    for name,(s,e) in (("one length",(s0,e0)),("1 % clipped",(s1,e1))):
        for rep in range(3):
            m=sv.solve(s,e,lengths,M,contig_read_offsets=offs); d=sv.last_stats.as_dict()
        ok=bool(np.array_equal(m,oracle_py.solve(s,e,lengths,M,offs)))
        print(f"{name:12s}: {s.size} reads {d['ms_total']:8.2f} ms path {d['path']} giveup {d['near_uniform_giveup']} rounds {d['near_uniform_rounds']} selected {d['near_uniform_selected']} == oracle {ok}",flush=True)
