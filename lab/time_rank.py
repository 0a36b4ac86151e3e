"""cfg4 (8 contigs x 12.5 M reads, M = 100), one solve alone: per-kernel device times (all kernels), 10 solves.
usage: python lab/time_rank.py"""
import sys, importlib, time, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
import torch
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S = np.concatenate(ss); E = np.concatenate(ee)
offs = np.arange(9, dtype=np.uint64) * 12_500_000
lengths = np.full(8, 1_000_000, np.uint32)
sv = pkg.Solver(0)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
sv.set_profiling(1)
for _ in range(10):
    sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
kt = sv.kernel_times(); sv.set_profiling(0)
st = sv.last_stats.as_dict()
m = dM.cpu().numpy()
import hashlib
print("ms_total %.3f  kept %d  mask sha1 %s" % (st['ms_total'], st['n_kept'], hashlib.sha1(m.tobytes()).hexdigest()[:12]))
for name, (n, ms) in kt.items():
    print("   %-45s %8.4f ms" % (name, ms / n))
