"""cfg5 shape at reduced scale (24 contigs ~ GRCh38 proportions, mean coverage 100x, M = 50)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import workloads
pkg = importlib.import_module("genome-downsampler_amd")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 64
total_len = int(1.5e9 * scale); pairs = int(0.5e9 * scale)
s, e, offs, lengths = workloads.wgs_contigs(total_len, pairs)
sol = pkg.Solver(0)
sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
sol.set_profiling(True)
t0 = time.perf_counter()
for _ in range(3):
    sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
st = sol.last_stats
print(f"N = {s.size}, Ltot = {int(lengths.sum())}, contigs = {lengths.size}, device ms = {st.ms_total:.2f}, "
      f"sweep ms = {st.ms_sweep:.2f}, kept = {st.n_kept}, sort passes = {st.sort_passes}, stretches = {st.sweep_stretches} (speculative {st.spec_boundaries}, mismatching {st.spec_mismatches})")
for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {name:36s} {ms / launches:.3f} ms avg")
