"""lab: k_pm_prepare_sort's time on cfg4 with 1 % clipped reads and the head filtering on 150 (QMCP_HIP_NEAR_ELL forces
the filter from the first call) -- timing only, the variants' results are not valid"""
import sys, importlib, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S, E = syn.clipped_mix(np.concatenate(ss), np.concatenate(ee), 0.01)
offs = np.arange(9, dtype=np.uint64) * 12_500_000
lengths = np.full(8, 1_000_000, np.uint32)
sv = pkg.Solver(0)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
for rep in range(3):
    if rep == 2: sv.set_profiling(1)
    try:
        sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
    except Exception as ex:
        print("solve failed (expected for a timing-only variant):", str(ex)[:80])
t = sv.kernel_times()
print({k: round(v[1] / v[0], 4) for k, v in t.items() if "prepare" in k}, "path", sv.last_stats.path)
