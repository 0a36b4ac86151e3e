"""lab: long SHALLOW contigs with a tail of clipped reads: two contigs of L positions at
depth x M (M = 100, reads of 150), 1 % of the reads clipped -- the near-uniform route (its sweeps in stretches) against
the mixed-span route and the one-length solve of the same reads unclipped.
   python lab/near_uniform_long_shallow.py [L = 10_000_000] [depth = 1.5] [fraction = 0.01]"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
depth = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
M = int(sys.argv[4]) if len(sys.argv) > 4 else 100
pairs = int(depth * M * L / 150 / 2)
ss, ee = [], []
for c in range(2):
    a, b = pkg.reads_gen(0, pairs, L, seed=4242 + c); ss.append(a); ee.append(b)
S0, E0 = np.concatenate(ss), np.concatenate(ee)
S, E = syn.clipped_mix(S0, E0, frac)
offs = np.arange(3, dtype=np.uint64) * np.uint64(2 * pairs)
lengths = np.full(2, L, np.uint32)
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")

def solve(sv, s, e, reps=3):
    dS = torch.from_numpy(s.view(np.int32)).cuda(); dE = torch.from_numpy(e.view(np.int32)).cuda()
    best = 1e9
    for _ in range(reps):
        st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), s.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        best = min(best, st.ms_total)
    return best, st.as_dict(), dM.cpu().numpy().copy()

with pkg.Solver(0) as sv:
    one = solve(sv, S0, E0)
with pkg.Solver(0) as sv:
    near = solve(sv, S, E)
    if os.environ.get("KERNELS"):
        dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
        sv.set_profiling(1)
        for _ in range(3):
            sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        for name, (k, ms) in sv.kernel_times().items():
            print("   %-70s %4d x %8.4f ms = %8.3f per solve" % (name, k // 3, ms / k, ms / 3))
        sv.set_profiling(0)
        del dS, dE
with pkg.Solver(0) as sv:
    sv.set_options(near_uniform=-1)
    mixed = solve(sv, S, E, reps=1)
d = near[1]
print(f"{S.size} reads on 2 x {L} positions, {depth} x M, {frac:.1%} clipped: one-length {one[0]:.2f} ms (path {one[1]['path']}) | "
      f"near-uniform {near[0]:.2f} ms = {near[0] / one[0]:.1f} x (path {d['path']}, giveup {d['near_uniform_giveup']}, "
      f"{d['near_uniform_exceptions']} exceptions, {d['near_uniform_selected']} kept, {d['near_uniform_rounds']} sweeps) | "
      f"mixed-span {mixed[0]:.2f} ms | same mask {bool(np.array_equal(near[2], mixed[2]))}")
sys.exit(0 if np.array_equal(near[2], mixed[2]) else 1)
