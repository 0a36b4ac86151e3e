import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
pkg=importlib.import_module('genome-downsampler_amd'); syn=importlib.import_module('genome-downsampler_amd.synthetic'); import oracle_py
rng=np.random.default_rng(3)
lengths=np.array([2_399_326,1_329_763],np.uint32); M=400; depth=1.2
counts=[int(depth*M*int(L)/150) for L in lengths]
ss=[rng.integers(0,int(L)-174,size=k).astype(np.uint32) for L,k in zip(lengths,counts)]
s0=np.concatenate(ss); e0=(s0+149).astype(np.uint32); offs=np.concatenate([[0],np.cumsum(counts)]).astype(np.uint64)
s1,e1=syn.clipped_mix(s0,e0,0.01); e2=syn.lengthened_mix(s1,e1,offs,lengths,0.005)
with pkg.Solver(0) as sv:
    for name,(s,e) in (("one length",(s0,e0)),("1 % clipped",(s1,e1)),("clipped + longer",(s1,e2))):
        for rep in range(2):
            m=sv.solve(s,e,lengths,M,contig_read_offsets=offs); d=sv.last_stats.as_dict()
        ok=bool(np.array_equal(m,oracle_py.solve(s,e,lengths,M,offs)))
        print(f"{name:17s}: {d['ms_total']:8.2f} ms path {d['path']} giveup {d['near_uniform_giveup']} stretches {d['sweep_stretches']} boundaries {d['spec_boundaries']} disagreeing {d['spec_mismatches']} / {d['spec_retry_mismatches']} rounds {d['near_uniform_rounds']} == oracle {ok}",flush=True)
