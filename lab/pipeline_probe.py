"""cfg4 with two solves in flight (two contexts, qmcp_hip_solve_device_begin / _end): does the
selection sweep of one call hide behind the bandwidth-bound stages of the next?"""
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
words = (S.size + 63) // 64
for depth in (1, 2, 3):
    svs = [pkg.Solver(0) for _ in range(depth)]
    masks = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(depth)]
    def run(K):
        for s in range(K):
            i = s % depth
            if s >= depth: svs[i].solve_end()
            svs[i].solve_device_begin(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, masks[i].data_ptr(), contig_read_offsets=offs)
        for s in range(K, K + depth):
            if s - depth >= 0 and s - depth < K: svs[s % depth].solve_end()
    run(6)
    torch.cuda.synchronize(); t = time.time(); K = 40; run(K); torch.cuda.synchronize(); dt = (time.time() - t) / K
    same = all(torch.equal(masks[0], m) for m in masks)
    print("in flight %d: %.3f ms per solve, %.0f Mreads/s, device ms_total of the last %.3f, masks equal %s" % (depth, dt * 1e3, S.size / dt / 1e6, svs[0].last_stats.ms_total, same))
    for sv in svs: sv.close()
