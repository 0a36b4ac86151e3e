"""cfg4, one solve alone, dealt to 0 / 2 / 3 / 4 / 8 contig groups (QMCP_HIP_GROUPS): device and wall time without
per-kernel brackets.  usage: python lab/groups_probe.py [group counts ...]"""
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
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
ref = None
for g in (sys.argv[1:] or ["0", "2", "3", "4", "8"]):
    if g == "auto": os.environ.pop("QMCP_HIP_GROUPS", None)
    else: os.environ["QMCP_HIP_GROUPS"] = g

    sv = pkg.Solver(0)
    for _ in range(4):
        sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
    dev, wall = [], []
    for _ in range(10):
        torch.cuda.synchronize(); t = time.perf_counter()
        sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
        wall.append((time.perf_counter() - t) * 1e3); dev.append(sv.last_stats.ms_total)
    m = dM.cpu().numpy()
    if ref is None: ref = m
    st = sv.last_stats
    print("groups %s (ran as %d): device ms min %.3f median %.3f | wall ms min %.3f median %.3f | kept %d same %s" % (
        g, st.contig_groups, min(dev), sorted(dev)[5], min(wall), sorted(wall)[5], st.n_kept, np.array_equal(m, ref)), flush=True)
    sv.close()
