import sys, os, importlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg=importlib.import_module('genome-downsampler_amd')
sv=pkg.Solver(0)
rng=np.random.default_rng(0)
n=200000; L=30000
span=rng.integers(100,151,size=n); st_=(rng.random(n)*(L-span+1)).astype(np.int64)
for _ in range(3):
    sv.solve(st_.astype(np.uint32),(st_+span-1).astype(np.uint32),L,100)
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in sv.last_stats.as_dict().items()})
