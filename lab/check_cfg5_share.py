"""One GPU's share of cfg5 at full size: 24 contigs ~ GRCh38 proportions scaled to 1/8 of the
genome-scale configuration (187.5 M positions, 125 M reads, M = 50) -- the two-level partition, the
general-form sweep split at whatever cut points there are, the ranking -- against the oracle.
   python lab/check_cfg5_share.py [scale = 0.125] [oracle = 1]"""
import importlib, os, sys, time
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import workloads
pkg = importlib.import_module("genome-downsampler_amd")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.125
with_oracle = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
t0 = time.time()
s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9 * scale), int(0.5e9 * scale))
print(f"generated {s.size} reads on {int(lengths.sum())} positions in {time.time() - t0:.0f} s", flush=True)
sol = pkg.Solver(0)
got = sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
sol.set_profiling(True)
got = sol.solve(s, e, lengths, 50, contig_read_offsets=offs)
st = sol.last_stats
print(f"device ms = {st.ms_total:.1f} (sweep {st.ms_sweep:.1f}), kept = {st.n_kept}, sort passes = {st.sort_passes}, "
      f"stretches = {st.sweep_stretches} (speculative {st.spec_boundaries}, mismatching {st.spec_mismatches}), {s.size / st.ms_total / 1e3:.0f} Mreads/s", flush=True)
for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:44s} {ms / launches:.3f} ms", flush=True)
if with_oracle:
    import oracle_py
    t0 = time.time()
    # contig by contig: progress lines, bounded memory
    ok = True
    for c in range(lengths.size):
        a, b = int(offs[c]), int(offs[c + 1])
        want = oracle_py.solve(s[a:b], e[a:b], int(lengths[c]), 50)
        idx = pkg.mask_to_indices(want, b - a).astype(np.int64) + a
        bits = np.unpackbits(got.view(np.uint8), bitorder="little")[a:b]
        same = int(bits.sum()) == idx.size and bool(bits[idx - a].all())
        ok &= same
        print(f"  contig {c}: {b - a} reads, kept {idx.size}, identical {same} ({time.time() - t0:.0f} s)", flush=True)
    print("identical to the oracle:", ok)
