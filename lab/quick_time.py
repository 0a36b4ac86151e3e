import sys, importlib, time, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'oracle'))
pkg=importlib.import_module('genome-downsampler_amd')
sv=pkg.Solver(0)
def run(name, s,e,lengths,M,offs=None,reps=3):
    for r in range(reps):
        t=time.time(); m=sv.solve(s,e,lengths,M,contig_read_offsets=offs); dt=time.time()-t
    st=sv.last_stats.as_dict()
    print(name, "n=%d kept=%d path=%d passes=%d iters=%d"%(st['n_reads'],st['n_kept'],st['path'],st['sort_passes'],st['reserved0']),
          " ".join("%s=%.3f"%(k,v) for k,v in st.items() if k.startswith('ms_')), "wall=%.1fms"%(dt*1e3), flush=True)
    return m
s,e=pkg.reads_gen(0,500000,30000); run("cfg2",s,e,30000,100)
s,e=pkg.reads_gen(0,1000000,30000); run("ref_uniform_M1000",s,e,30000,1000)
t=time.time(); s,e=pkg.reads_gen(0,6250000,1000000); print("gen %.1fs"%(time.time()-t))
run("cfg4_1contig",s,e,1000000,100)
ss=[s];ee=[e]
for c in range(1,8):
    a,b=pkg.reads_gen(0,6250000,1000000,seed=12345+c); ss.append(a); ee.append(b)
S=np.concatenate(ss);E=np.concatenate(ee)
offs=np.arange(9,dtype=np.uint64)*12500000
m=run("cfg4_8contig",S,E,np.full(8,1000000,np.uint32),100,offs)
# variable-length general path timing
rng=np.random.default_rng(0)
n=200000; L=30000
span=rng.integers(100,151,size=n); st_=(rng.random(n)*(L-span+1)).astype(np.int64)
run("general_200k",st_.astype(np.uint32),(st_+span-1).astype(np.uint32),L,100,reps=1)
