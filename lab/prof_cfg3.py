"""cfg3 at full size: 15 M amplicon pairs (30 M reads) on a 29 903-base genome, BED/TSV-style FILTER,
M = 200: the fused host entry (filter -> compaction -> solve -> mate completion) and its kernels."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import workloads
pkg = importlib.import_module("genome-downsampler_amd")
pairs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 15_000_000
s, e, a0, a1, straddle = workloads.amplicon_reads(pairs)
sol = pkg.Solver(0)
sol.filter_solve(s, e, 29_903, 200, amp_starts=a0, amp_ends=a1, complete_pairs=True)
sol.set_profiling(True)
t0 = time.perf_counter()
for _ in range(3):
    mask = sol.filter_solve(s, e, 29_903, 200, amp_starts=a0, amp_ends=a1, complete_pairs=True)
wall = (time.perf_counter() - t0) / 3
st = sol.last_stats
print(f"reads = {s.size}, straddling pairs filtered = {int(straddle.sum())}, host entry wall = {wall * 1e3:.2f} ms "
      f"({s.size / wall / 1e6:.0f} Mreads/s incl. PCIe), solve device ms = {st.ms_total:.3f} "
      f"(h2d {st.ms_h2d:.2f}, d2h {st.ms_d2h:.2f}), reads solved = {st.n_reads}, kept = {st.n_kept}, sort passes = {st.sort_passes}")
for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {name:44s} {ms / launches:.3f} ms avg x {launches // 3} per call")
