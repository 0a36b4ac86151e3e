import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
pkg=importlib.import_module('genome-downsampler_amd'); syn=importlib.import_module('genome-downsampler_amd.synthetic'); import oracle_py
for (L,M,depth,ell) in ((1_033_528,20,4.0,150),(1_250_000,10,6.0,100),(13_890_064,10,6.0,150)):
    rng=np.random.default_rng(5); n=int(depth*M*L/ell)
    s0=rng.integers(0,L-ell-8,size=n).astype(np.uint32); e0=(s0+np.uint32(ell-1)).astype(np.uint32); lengths=np.array([L],np.uint32)
    s1,e1=syn.clipped_mix(s0,e0,0.01,max_clip=min(50,ell//2))
    with pkg.Solver(0) as sv:
        for name,(s,e) in (("one length",(s0,e0)),("1 % clipped",(s1,e1))):
            for rep in range(3):
                m=sv.solve(s,e,lengths,M); d=sv.last_stats.as_dict()
            ok=bool(np.array_equal(m,oracle_py.solve(s,e,lengths,M)))
            print(f"L {L} M {M} depth {depth} ell {ell} {name:12s}: {s.size} reads {d['ms_total']:8.2f} ms path {d['path']} giveup {d['near_uniform_giveup']} rounds {d['near_uniform_rounds']} stretches {d['sweep_stretches']} == oracle {ok}",flush=True)
