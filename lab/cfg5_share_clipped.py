"""lab: one GPU's REAL share of configs[4] (the heaviest rank's three full-length contigs, 129.7 M reads, M = 50) with
1 % of the reads clipped by 1 ... 50 bases -- VERDICT round 3, item 6: the near-uniform route with its sweeps in
stretches, against the one-length solve of the same reads and (MIXED=1) the mixed-span walk; ORACLE=1 compares the
kept set with the oracle contig by contig.
   python lab/cfg5_share_clipped.py [fraction = 0.01]"""
import os, sys, importlib, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
M = 50
share, owned = syn.cfg5_heaviest_share(8)
S0, E0, offs, lengths = syn.wgs_contigs(int(1.5e9), int(0.5e9), only=share)
S, E = syn.clipped_mix(S0, E0, frac)
print(f"contigs {share}: {lengths.tolist()} positions, {S.size} reads, {(E - S != 149).sum()} clipped", flush=True)
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")

def solve(sv, s, e, reps=3):
    dS = torch.from_numpy(s.view(np.int32)).cuda(); dE = torch.from_numpy(e.view(np.int32)).cuda()
    best = 1e9
    for r in range(reps):
        st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), s.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        best = min(best, st.ms_total)
        print(f"   rep {r}: {st.ms_total:.2f} ms path {st.path} giveup {st.near_uniform_giveup} rounds {st.near_uniform_rounds} "
              f"selected {st.near_uniform_selected} stretches {st.sweep_stretches}", flush=True)
    return best, st.as_dict(), dM.cpu().numpy().copy()

with pkg.Solver(0) as sv:
    one = solve(sv, S0, E0)
print(f"one length: {one[0]:.2f} ms path {one[1]['path']} stretches {one[1]['sweep_stretches']}", flush=True)
with pkg.Solver(0) as sv:
    if os.environ.get("DEBUG"): sv.set_options(near_uniform_debug=1)
    near = solve(sv, S, E, reps=int(os.environ.get("REPS", "3")))
    d = near[1]
    print(f"{frac:.1%} clipped: {near[0]:.2f} ms = {near[0] / one[0]:.2f} x (path {d['path']}, giveup {d['near_uniform_giveup']}, "
          f"{d['near_uniform_exceptions']} exceptions, {d['near_uniform_selected']} kept, {d['near_uniform_rounds']} sweeps, "
          f"{d['sweep_stretches']} stretches, grown mid-solve {d.get('arena_grown_mid_solve')})", flush=True)
    if os.environ.get("KERNELS"):
        dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
        sv.set_profiling(1)
        for _ in range(3):
            sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        for name, (k, ms) in sv.kernel_times().items():
            print("   %-60s %5d x %8.4f ms = %8.3f per solve" % (name, k // 3, ms / k, ms / 3))
        sv.set_profiling(0)
        del dS, dE
ok = True
if os.environ.get("MIXED"):
    with pkg.Solver(0) as sv:
        sv.set_options(near_uniform=-1)
        mixed = solve(sv, S, E, reps=1)
    same = bool(np.array_equal(near[2], mixed[2]))
    ok &= same
    print(f"mixed-span route: {mixed[0]:.2f} ms, same mask {same}", flush=True)
if os.environ.get("ORACLE"):
    sys.path.insert(0, os.path.join(R, "oracle")); import oracle_py as ora
    bits = np.unpackbits(near[2].view(np.uint8), bitorder="little")
    t0 = time.time()
    for c in range(lengths.size):
        a, b = int(offs[c]), int(offs[c + 1])
        want = ora.solve(S[a:b], E[a:b], int(lengths[c]), M)
        wbits = np.unpackbits(want.view(np.uint8), bitorder="little")[:b - a]
        same = bool(np.array_equal(bits[a:b], wbits))
        ok &= same
        print(f"contig {share[c]}: == oracle {same} ({time.time() - t0:.0f} s)", flush=True)
sys.exit(0 if ok else 1)
