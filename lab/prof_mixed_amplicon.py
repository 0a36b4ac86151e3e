"""cfg3's shape with a realistic read-length mix (85 % modal length, 15 % soft-clipped by 1..50 bases) at
full size (30 M reads on 29 903 bases, M = 200): what the mixed-span route costs next to the uniform one."""
import sys, importlib, time, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import workloads
pkg = importlib.import_module('genome-downsampler_amd')
s, e, a0, a1, _ = workloads.amplicon_reads(15_000_000, seed=77, straddle_fraction=0.0)
rng = np.random.default_rng(6)
clipped = rng.random(s.size) < 0.15
cut = rng.integers(1, 51, size=s.size); front = rng.random(s.size) < 0.5
s2 = np.where(clipped & front, s + cut, s).astype(np.uint32); e2 = np.where(clipped & ~front, e - cut, e).astype(np.uint32)
sv = pkg.Solver(0)
for name, (a, b) in (("uniform (all 150)", (s, e)), ("85 % of 150 + clipped tail", (s2, e2))):
    sv.solve(a, b, 29_903, 200)
    sv.set_profiling(1)
    sv.solve(a, b, 29_903, 200)
    st = sv.last_stats
    print("%-28s device %.3f ms  (prepare %.3f scan %.3f sort %.3f sweep %.3f mark %.3f)  kept %d  path %d" % (
        name, st.ms_total, st.ms_prepare, st.ms_scan, st.ms_sort, st.ms_sweep, st.ms_mark, st.n_kept, st.path))
    for k, (n, ms) in sorted(sv.kernel_times().items(), key=lambda kv: -kv[1][1])[:6]:
        print("      %-40s %.3f ms" % (k, ms / n))
    sv.set_profiling(0)
