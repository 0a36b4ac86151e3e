"""Mixed-span route timing: cfg2-sized input with spans drawn from a range."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("genome-downsampler_amd")
rng = np.random.default_rng(5)
for n, L, lo, hi, M in ((1_000_000, 30_000, 100, 150, 100), (1_000_000, 30_000, 149, 150, 100),
                        (10_000_000, 1_000_000, 100, 150, 100), (200_000, 30_000, 50, 250, 50)):
    span = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
    s = (rng.random(n) * (L - span + 1)).astype(np.uint32)
    e = s + span - 1
    sol = pkg.Solver(0)
    sol.solve(s, e, L, M)
    sol.set_profiling(True)
    for _ in range(3):
        sol.solve(s, e, L, M)
    st = sol.last_stats
    print(f"n={n} L={L} spans {lo}..{hi} M={M}: path {st.path}, device {st.ms_total:.2f} ms (sweep {st.ms_sweep:.2f}, sort {st.ms_sort:.2f}), kept {st.n_kept}")
    for name, (launches, ms) in sorted(sol.kernel_times().items(), key=lambda kv: -kv[1][1])[:4]:
        print(f"     {name:34s} {ms / launches:.3f} ms")
