import sys, importlib, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
rng = np.random.default_rng(3)
s5, e5 = pkg.reads_gen(pkg.KIND_UNIFORM, 125_000, 30_000, seed=11)
s5 = s5.astype(np.int64); e5 = e5.astype(np.int64)
pick = rng.random(s5.size) < 0.01
clip = rng.integers(1, 40, size=s5.size)
front = rng.random(s5.size) < 0.5
s = np.where(pick & front, s5 + clip, s5).astype(np.uint32); e = np.where(pick & ~front, e5 - clip, e5).astype(np.uint32)
with pkg.Solver(0) as sv:
    sv.solve(s, e, 30_000, 100)
    print(sv.last_stats.as_dict())
for i in [int(a) for a in sys.argv[1:]]:
    print("read", i, "start", s[i], "end", e[i])
