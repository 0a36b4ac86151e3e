"""lab: one long contig with one dominant read length, 1 % clipped and 0.5 % LENGTHENED reads (the mixed-span route), at a
depth that is deep in standard deviations (large M): the library's choice of speculation, against the oracle (ORACLE=1)
and (CHAIN=1) against the walk as one chain.
   python lab/mixed_spec_long_contig.py [L = 60_000_000] [M = 100] [depth = 3.0]"""
import os, sys, importlib, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
L = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100
depth = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
rng = np.random.default_rng(21)
n = int(depth * M * L / 150)
s = rng.integers(0, L - 150 - 24, size=n).astype(np.uint32); e = (s + 149).astype(np.uint32)
s, e = syn.clipped_mix(s, e, 0.01)
lengths = np.array([L], np.uint32); offs = np.array([0, n], np.uint64)
e = syn.lengthened_mix(s, e, offs, lengths, 0.005)
with pkg.Solver(0) as sv:
    for rep in range(2):
        m = sv.solve(s, e, lengths, M)
        d = sv.last_stats.as_dict()
        print(f"{n} reads on {L} positions, M {M}, {depth} x M: {d['ms_total']:.1f} ms (sweep {d['ms_sweep']:.1f}) path {d['path']} stretches {d['sweep_stretches']} "
              f"boundaries {d['spec_boundaries']} disagreeing {d['spec_mismatches']} / {d['spec_retry_mismatches']}", flush=True)
    if os.environ.get("CHAIN"):
        with sv.options(speculation=-1):
            m2 = sv.solve(s, e, lengths, M)
            d = sv.last_stats.as_dict()
        print(f"   as one chain per stretch between real cut points: {d['ms_total']:.1f} ms, stretches {d['sweep_stretches']}, same mask {bool(np.array_equal(m, m2))}", flush=True)
if os.environ.get("ORACLE"):
    import oracle_py
    t0 = time.time()
    print("== oracle", bool(np.array_equal(m, oracle_py.solve(s, e, lengths, M))), f"({time.time() - t0:.0f} s)")
