"""lab: a two-level genome (2 contigs x 8 M positions, 12.5 x M deep at M = 100, 1 % of the reads clipped): the
near-uniform route behind the range-major producers against the mixed-span route"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
L, pairs = 8_000_000, 33_333_333
ss, ee = [], []
for c in range(2):
    a, b = pkg.reads_gen(0, pairs, L, seed=777 + c); ss.append(a); ee.append(b)
S, E = syn.clipped_mix(np.concatenate(ss), np.concatenate(ee), 0.01)
offs = np.arange(3, dtype=np.uint64) * np.uint64(2 * pairs)
lengths = np.full(2, L, np.uint32)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
res = {}
for near in ("1", "0"):
    os.environ["QMCP_HIP_NEAR"] = near
    with pkg.Solver(0) as sv:
        best = 1e9
        for _ in range(2):
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
            best = min(best, st.ms_total)
        res[near] = (best, st.as_dict(), dM.cpu().numpy().copy())
d = res["1"][1]
print(f"{S.size} reads on 2 x {L} positions: near-uniform {res['1'][0]:.2f} ms (path {d['path']}, {d['near_uniform_exceptions']} exceptions, {d['near_uniform_selected']} kept, {d['near_uniform_rounds']} sweeps) | mixed-span {res['0'][0]:.2f} ms | same mask {bool(np.array_equal(res['1'][2], res['0'][2]))}")
