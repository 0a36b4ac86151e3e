"""How long a run-in do speculative stretch boundaries need?  cfg5's shape (24 contigs ~ GRCh38 proportions,
mean coverage 100x) at a given scale, swept at several M (depth = 100 / M in units of M) with the run-in
forced to several lengths: boundaries that disagreed with the stretch before them.
   python lab/spec_burn_study.py [scale = 1/32]"""
import importlib, os, subprocess, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
if len(sys.argv) > 2:   # child: one (M, burn) point
    import workloads
    pkg = importlib.import_module("genome-downsampler_amd")
    scale, M = float(sys.argv[1]), int(sys.argv[2])
    s, e, offs, lengths = workloads.wgs_contigs(int(1.5e9 * scale), int(0.5e9 * scale))
    sol = pkg.Solver(0)
    sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = sol.last_stats
    print(f"M={M} depth={100 / M:.2f} burn={os.environ.get('QMCP_HIP_SPEC_BURN')}: speculative {st.spec_boundaries}, "
          f"mismatching {st.spec_mismatches}, sweep {st.ms_sweep:.2f} ms", flush=True)
    sys.exit(0)
scale = sys.argv[1] if len(sys.argv) > 1 else str(1 / 32)
for M in (67, 50, 40, 33, 25):
    for burn in (64, 128, 256, 512, 1024, 2048):
        env = dict(os.environ, QMCP_HIP_SPEC="1", QMCP_HIP_SPEC_BURN=str(burn))
        subprocess.run([sys.executable, __file__, scale, str(M)], env=env, check=False)
