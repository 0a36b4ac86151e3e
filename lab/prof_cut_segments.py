"""Cut-point segmentation on shallow / gapped data: the same call with QMCP_HIP_CUTS=0 and =1.
   python lab/prof_cut_segments.py [positions] [M] [depth in units of M] [gap fraction] [shortest span]
   (spans are 150, or drawn from [shortest, 150] when given: the mixed-span event sweeps)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("genome-downsampler_amd")
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 30
depth = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
gaps = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
lo_span = int(sys.argv[5]) if len(sys.argv) > 5 else 150
rng = np.random.default_rng(1)
n = int(L * M * depth * (1 - gaps) / ((150 + lo_span) / 2))
span = rng.integers(lo_span, 151, size=n).astype(np.uint32) if lo_span < 150 else np.uint32(150)
s = rng.integers(0, L - 150 + 1, size=n, dtype=np.uint32)
if gaps > 0:  # islands: fold the starts into the first (1 - gaps) of every 50 000-base period
    period = 50_000
    s = ((s // period) * period + ((s % period) * (1 - gaps)).astype(np.uint32)).astype(np.uint32)
    s = np.minimum(s, L - 150).astype(np.uint32)
e = (s + span - np.uint32(1)).astype(np.uint32)
sol = pkg.Solver(0)
masks = {}
for cuts in ("0", "1"):
    os.environ["QMCP_HIP_CUTS"] = cuts
    sol.solve(s, e, L, M)
    sol.set_profiling(True)
    for _ in range(3):
        masks[cuts] = sol.solve(s, e, L, M)
    st = sol.last_stats
    kt = sol.kernel_times()
    sweep = {k: v[1] / v[0] for k, v in kt.items() if k.startswith("k_sweep") or k == "k_find_cuts"}
    print(f"cuts={cuts}: N = {n}, L = {L}, M = {M}, device ms = {st.ms_total:.3f}, sweep ms = {st.ms_sweep:.3f}, "
          f"stretches = {st.sweep_stretches}, kept = {st.n_kept}, kernels = "
          + ", ".join(f"{k} {v:.3f}" for k, v in sweep.items()))
    sol.set_profiling(False)
print("identical:", bool(np.array_equal(masks["0"], masks["1"])))
