"""Coverage that varies smoothly along the genome (between ~1 x and ~5 x M, mean ~3 x M): speculative stretch
boundaries hold in the shallow parts and disagree in the deep ones; the marks send only those parts to the
later tiers.  Timing with and without speculation, and the oracle.
   python lab/prof_varying_depth.py [positions = 2e7] [M = 40]"""
import importlib, os, sys, time
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
pkg = importlib.import_module("genome-downsampler_amd")
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 40
span = 150
rng = np.random.default_rng(11)
# density ~ 3 + 2 sin: rejection sampling of start positions
n_target = int(3.0 * M * L / span)
cand = rng.integers(0, L - span + 1, size=int(n_target * 1.8), dtype=np.int64)
dens = (3.0 + 2.0 * np.sin(cand * (2 * np.pi / 2_500_000))) / 5.0
s = cand[rng.random(cand.size) < dens][:n_target].astype(np.uint32)
e = s + np.uint32(span - 1)
sol = pkg.Solver(0)
res = {}
for spec in ("1", "0"):
    os.environ["QMCP_HIP_SPEC"] = spec
    sol.solve(s, e, L, M)
    sol.set_profiling(True)
    res[spec] = sol.solve(s, e, L, M)
    st = sol.last_stats
    print(f"speculation {'on ' if spec == '1' else 'off'}: N = {s.size}, L = {L}, M = {M}, device ms = {st.ms_total:.2f}, sweep ms = {st.ms_sweep:.2f}, "
          f"stretches = {st.sweep_stretches}, speculative {st.spec_boundaries}, disagreeing {st.spec_mismatches}, "
          f"in the second tier {st.spec_retry_mismatches}", flush=True)
    for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:5]:
        print(f"      {name:50s} {ms / launches:.3f} ms")
    sol.set_profiling(False)
t0 = time.time()
want = oracle_py.solve(s, e, L, M)
print("identical to each other:", bool(np.array_equal(res["1"], res["0"])), " to the oracle:", bool(np.array_equal(res["1"], want)),
      f"(oracle {time.time() - t0:.1f} s)")
