"""lab: one GPU's real share of configs[4] with 1 % clipped reads on the MIXED-SPAN route (near-uniform route off), its
speculative boundaries forced on (the route does not speculate by itself where one length holds nine tenths of a sample).
   python lab/cfg5_share_mixed_spec.py [run-in blocks ...]"""
import os, sys, importlib, hashlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
share, owned = syn.cfg5_heaviest_share(8)
S0, E0, offs, lengths = syn.wgs_contigs(int(1.5e9), int(0.5e9), only=share)
S, E = syn.clipped_mix(S0, E0, 0.01)
del S0, E0
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
with pkg.Solver(0) as sv:
    st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 50, dM.data_ptr(), contig_read_offsets=offs)
    ref = hashlib.sha1(dM.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"near-uniform route: {st.ms_total:.1f} ms path {st.path} mask {ref}", flush=True)
if os.environ.get("LONGER"):
    # the same share with 0.5 % of the reads LENGTHENED by 1 ... 20 bases as well (deletions): the near-uniform route gives
    # up, the library's own choice on the mixed-span route
    rng = np.random.default_rng(9)
    j = rng.choice(S.size, size=S.size // 200, replace=False)
    contig_end = np.repeat(np.cumsum(lengths.astype(np.int64)) - np.cumsum(lengths.astype(np.int64)) + lengths.astype(np.int64), np.diff(offs.astype(np.int64)))
    E2 = E.copy(); E2[j] = np.minimum(E[j].astype(np.int64) + rng.integers(1, 21, size=j.size), contig_end[j] - 1).astype(np.uint32)
    dE.copy_(torch.from_numpy(E2.view(np.int32)))
    with pkg.Solver(0) as sv:
        for rep in range(2):
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 50, dM.data_ptr(), contig_read_offsets=offs)
            d = st.as_dict()
            print(f"with longer reads, the library's choice: {d['ms_total']:.1f} ms (sweep {d['ms_sweep']:.1f}) path {d['path']} giveup {d['near_uniform_giveup']} "
                  f"stretches {d['sweep_stretches']} boundaries {d['spec_boundaries']} disagreeing {d['spec_mismatches']} / {d['spec_retry_mismatches']}", flush=True)
        mine = dM.cpu().numpy().copy()
    if os.environ.get("ORACLE"):
        sys.path.insert(0, os.path.join(R, "oracle")); import oracle_py, time
        bits = np.unpackbits(mine.view(np.uint8), bitorder="little"); t0 = time.time()
        for c in range(lengths.size):
            a, b = int(offs[c]), int(offs[c + 1])
            want = np.unpackbits(oracle_py.solve(S[a:b], E2[a:b], int(lengths[c]), 50).view(np.uint8), bitorder="little")[:b - a]
            print(f"contig {share[c]}: == oracle {bool(np.array_equal(bits[a:b], want))} ({time.time() - t0:.0f} s)", flush=True)
    sys.exit(0)
for run_in in [int(x) for x in sys.argv[1:]] or [0, 256]:
    with pkg.Solver(0) as sv:
        sv.set_options(near_uniform=-1, speculation=1, speculation_run_in=run_in)
        for rep in range(2):
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 50, dM.data_ptr(), contig_read_offsets=offs)
            d = st.as_dict()
            h = hashlib.sha1(dM.cpu().numpy().tobytes()).hexdigest()[:12]
            print(f"mixed-span route, speculation forced, run-in {run_in}: {d['ms_total']:.1f} ms (sweep {d['ms_sweep']:.1f}) stretches {d['sweep_stretches']} "
                  f"boundaries {d['spec_boundaries']} disagreeing {d['spec_mismatches']} / {d['spec_retry_mismatches']} same mask {h == ref}", flush=True)
