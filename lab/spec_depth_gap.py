"""Depths between 4.1 x M and 11 x M (where the solver neither speculates nor takes the event-driven sweep): how long a
run-in would a speculative stretch boundary need there?  One contig of 20 M positions at 100 x coverage (13.3 M reads
of 150 bases), M = 20, 17, 14, 12 (depth 5.0, 5.9, 7.1, 8.3 x M), speculation forced at any depth with run-ins of
2 048 ... 16 384 blocks: boundaries that disagreed, sweep time, and the plain sweep (no speculation) beside it.
   python lab/spec_depth_gap.py"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
if len(sys.argv) > 1:   # child: one (M, burn) point
    pkg = importlib.import_module("genome-downsampler_amd")
    M = int(sys.argv[1])
    L = 20_000_000
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, int(L * 100 / 150 / 2), L, 150, seed=4242)
    sol = pkg.Solver(0)
    sol.solve(s, e, L, M)
    sol.solve(s, e, L, M)
    st = sol.last_stats
    print(f"M={M} depth={100 / M:.2f} run-in={os.environ.get('QMCP_HIP_SPEC_BURN', '-')}: stretches {st.sweep_stretches}, "
          f"speculative {st.spec_boundaries}, disagreeing {st.spec_mismatches} (second tier {st.spec_retry_mismatches}), "
          f"sweep {st.ms_sweep:.2f} ms, solve {st.ms_total:.2f} ms, kept {st.n_kept}", flush=True)
    sys.exit(0)
burns = [int(b) for b in os.environ.get("BURNS", "2048,4096,8192,16384").split(",")]
for M in [int(m) for m in os.environ.get("MS", "20,17,14,12").split(",")]:
    subprocess.run([sys.executable, __file__, str(M)], env=dict(os.environ, QMCP_HIP_SPEC="0"), check=False)
    for burn in burns:
        env = dict(os.environ, QMCP_HIP_SPEC="1", QMCP_HIP_SPEC_BURN=str(burn))
        subprocess.run([sys.executable, __file__, str(M)], env=env, check=False)
