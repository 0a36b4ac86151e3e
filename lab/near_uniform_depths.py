"""lab: cfg4's reads with 1 % clipped at larger M (shallower in units of M): near-uniform route (depth floor lifted by
QMCP_HIP_NEAR_MIN_DEPTH=0) against the mixed-span route"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S, E = syn.clipped_mix(np.concatenate(ss), np.concatenate(ee), 0.01)
KEEP = float(os.environ.get("KEEP", "1"))   # every contig's first KEEP share of its reads (shallower data at one M)
per = int(12_500_000 * KEEP)
if KEEP < 1:
    S = np.concatenate([S[c * 12_500_000:c * 12_500_000 + per] for c in range(8)])
    E = np.concatenate([E[c * 12_500_000:c * 12_500_000 + per] for c in range(8)])
offs = np.arange(9, dtype=np.uint64) * np.uint64(per)
lengths = np.full(8, 1_000_000, np.uint32)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
for M in [int(x) for x in sys.argv[1:]] or [200, 300, 400]:
    res = {}
    for near in ("1", "0"):
        os.environ["QMCP_HIP_NEAR"] = near
        os.environ["QMCP_HIP_NEAR_MIN_DEPTH"] = "0"
        with pkg.Solver(0) as sv:
            best = 1e9
            for _ in range(2):
                st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
                best = min(best, st.ms_total)
            res[near] = (best, st.as_dict(), dM.cpu().numpy().copy())
    d = res["1"][1]
    print(f"M {M} (depth {1875 * KEEP / M:.2f} x M): near {res['1'][0]:8.2f} ms path {d['path']} kept-exc {d['near_uniform_selected']} sweeps {d['near_uniform_rounds']} | mixed {res['0'][0]:8.2f} ms stretches {res['0'][1]['sweep_stretches']} | same mask {bool(np.array_equal(res['1'][2], res['0'][2]))}", flush=True)
