import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("genome-downsampler_amd")
rng = np.random.default_rng(5)
for n, L, lo, hi, M in ((1_000_000, 30_000, 300, 900, 100), (1_000_000, 30_000, 100, 600, 100), (200_000, 30_000, 400, 1200, 50), (100_000, 30_000, 1000, 4000, 50)):
    span = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
    s = (rng.random(n) * (L - span + 1)).astype(np.uint32)
    e = s + span - 1
    sol = pkg.Solver(0)
    sol.solve(s, e, L, M)
    sol.set_profiling(True)
    for _ in range(3):
        sol.solve(s, e, L, M)
    st = sol.last_stats
    kt = sol.kernel_times()
    sw = {k: v[1] / v[0] for k, v in kt.items() if k.startswith("k_sweep")}
    print(f"n={n} L={L} spans {lo}..{hi} M={M}: device {st.ms_total:.2f} ms, kept {st.n_kept}, {sw}")
