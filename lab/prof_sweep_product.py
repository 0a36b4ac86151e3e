"""Product-side view of the sweep on cfg4-like data: kernel time and general-form block count."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("genome-downsampler_amd")
contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pairs, L, M = 6_250_000, 1_000_000, 100
ss, ee = zip(*[pkg.reads_gen(0, pairs, L, seed=12345 + c) for c in range(contigs)])
s, e = np.concatenate(ss), np.concatenate(ee)
offs = np.arange(contigs + 1, dtype=np.uint64) * np.uint64(2 * pairs)
lengths = np.full(contigs, L, np.uint32)
sol = pkg.Solver(0)
sol.solve(s, e, lengths, M, contig_read_offsets=offs)
sol.set_profiling(True)
for _ in range(5):
    sol.solve(s, e, lengths, M, contig_read_offsets=offs)
st = sol.last_stats
print("general-form blocks per solve:", st.reserved0, "of", contigs * ((L + 149) // 150))
for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1]):
    print(f"{name:36s} {launches:4d} launches  {ms / launches:.4f} ms avg")
