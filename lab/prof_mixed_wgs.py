"""cfg5's shape (24 contigs ~ GRCh38 proportions, mean coverage ~ 2 x M) with a mix of read lengths
(100 ... 150): the mixed-span route with and without speculative stretch boundaries.
   python lab/prof_mixed_wgs.py [scale = 1/64]"""
import importlib, os, sys, time
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import workloads
pkg = importlib.import_module("genome-downsampler_amd")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 64
s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9 * scale), int(0.5e9 * scale))
rng = np.random.default_rng(3)
e = (s + rng.integers(100, 151, size=s.size).astype(np.uint32) - 1).astype(np.uint32)
sol = pkg.Solver(0)
masks = {}
for spec in ("1", "0"):
    os.environ["QMCP_HIP_SPEC"] = spec
    sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
    sol.set_profiling(True)
    masks[spec] = sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
    st = sol.last_stats
    print(f"speculative boundaries {'on ' if spec == '1' else 'off'}: N = {s.size}, Ltot = {int(lengths.sum())}, device ms = {st.ms_total:.2f}, "
          f"sweep ms = {st.ms_sweep:.2f}, kept = {st.n_kept}, path {st.path}, stretches = {st.sweep_stretches} "
          f"(speculative {st.spec_boundaries}, mismatching {st.spec_mismatches})", flush=True)
    for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:9]:
        print(f"      {name:50s} {ms / launches:.3f} ms")
    sol.set_profiling(False)
print("identical:", bool(np.array_equal(masks["1"], masks["0"])))
