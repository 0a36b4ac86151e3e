"""Does the sweep forget its start on DEEP data too (cfg4's depth, 18.75 x M)?  One contig of 4 M positions at 1875 x
coverage (50 M reads of 150 bases, M = 100), the general block-scan sweep forced, speculation forced with run-ins of
256 ... 2 048 blocks: boundaries that disagreed.   python lab/spec_deep_probe.py"""
import importlib, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
if len(sys.argv) > 1:
    pkg = importlib.import_module("genome-downsampler_amd")
    L, M = 4_000_000, int(sys.argv[1])
    s, e = pkg.reads_gen(pkg.KIND_UNIFORM, int(L * 1875 / 150 / 2), L, 150, seed=777)
    sol = pkg.Solver(0)
    sol.solve(s, e, L, M)
    st = sol.last_stats
    print(f"M={M} depth={1875 / M:.2f} run-in={os.environ.get('QMCP_HIP_SPEC_BURN', '-')} sweep={os.environ.get('QMCP_HIP_SWEEP')}: "
          f"stretches {st.sweep_stretches}, speculative {st.spec_boundaries}, disagreeing {st.spec_mismatches} "
          f"(second tier {st.spec_retry_mismatches}), sweep {st.ms_sweep:.2f} ms, kept {st.n_kept}", flush=True)
    sys.exit(0)
for M in (100, 50):
    subprocess.run([sys.executable, __file__, str(M)], env=dict(os.environ, QMCP_HIP_SWEEP="ev"), check=False)
    for burn in (256, 512, 1024, 2048):
        env = dict(os.environ, QMCP_HIP_SPEC="1", QMCP_HIP_SPEC_BURN=str(burn), QMCP_HIP_SWEEP="gen", QMCP_HIP_CUTS="1")
        subprocess.run([sys.executable, __file__, str(M)], env=env, check=False)
