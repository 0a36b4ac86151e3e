"""lab: cfg3's shape (30 M amplicon reads on 29 903 bases, M = 200) with 15 % of the reads clipped: per-kernel device times"""
import sys, importlib, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
a, b, _, _, _ = syn.amplicon_reads(15_000_000)
S, E = syn.clipped_mix(a, b, 0.15)
sv = pkg.Solver(0)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
L = np.array([29_903], np.uint32)
for _ in range(2):
    sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, L, 200, dM.data_ptr())
sv.set_profiling(1)
st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, L, 200, dM.data_ptr())
d = st.as_dict()
print("ms_total %.3f path %d passes %d stretches %d spec %d/%d giveup %d exceptions %d selected %d rounds %d" % (d["ms_total"], d["path"], d["sort_passes"], d["sweep_stretches"], d["spec_boundaries"], d["spec_mismatches"], d["near_uniform_giveup"], d["near_uniform_exceptions"], d["near_uniform_selected"], d["near_uniform_rounds"]))
for name, (n, ms) in sorted(sv.kernel_times().items(), key=lambda kv: -kv[1][1]):
    print("   %-45s %3d x %9.4f ms" % (name, n, ms / n))
