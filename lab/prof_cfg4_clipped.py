"""cfg4 with 1 % of the reads soft-clipped (the mixed-span route on deep data): per-kernel device times.
   python lab/prof_cfg4_clipped.py [fraction = 0.01]"""
import sys, importlib, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S, E = syn.clipped_mix(np.concatenate(ss), np.concatenate(ee), frac)
offs = np.arange(9, dtype=np.uint64) * 12_500_000
lengths = np.full(8, 1_000_000, np.uint32)
sv = pkg.Solver(0)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
for _ in range(2):
    sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
sv.set_profiling(1)
sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
st = sv.last_stats.as_dict()
print("clipped fraction", frac, "ms_total %.2f" % st["ms_total"], "path", st["path"], "spans", st["min_span"], st["max_span"], "stretches", st["sweep_stretches"])
for name, (n, ms) in sorted(sv.kernel_times().items(), key=lambda kv: -kv[1][1]):
    print("   %-45s %3d x %9.4f ms" % (name, n, ms / n))
print({k: st[k] for k in ("near_uniform_exceptions", "near_uniform_rounds", "near_uniform_selected", "ms_prepare", "ms_scan", "ms_sort", "ms_sweep", "ms_mark")})
# the same reads, none clipped, through a head that still filters on the remembered span: what the filter itself costs
S0 = np.concatenate(ss); E0 = np.concatenate(ee)
dS.copy_(torch.from_numpy(S0.view(np.int32))); dE.copy_(torch.from_numpy(E0.view(np.int32)))
sv.set_profiling(0)
sv.solve_device(dS.data_ptr(), dE.data_ptr(), S0.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
sv.set_profiling(1)
sv.solve_device(dS.data_ptr(), dE.data_ptr(), S0.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
print("one length, filtering head: ms_total %.3f path %d" % (sv.last_stats.ms_total, sv.last_stats.path))
for name, (n, ms) in sorted(sv.kernel_times().items(), key=lambda kv: -kv[1][1])[:4]:
    print("   %-45s %3d x %9.4f ms" % (name, n, ms / n))
