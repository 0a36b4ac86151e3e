"""lab: cfg4 with 0.2 ... 6 % of the reads clipped: route, rounds, device time (near-uniform route on and off)"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S0, E0 = np.concatenate(ss), np.concatenate(ee)
offs = np.arange(9, dtype=np.uint64) * 12_500_000
lengths = np.full(8, 1_000_000, np.uint32)
dS = torch.empty(S0.size, dtype=torch.int32, device="cuda"); dE = torch.empty_like(dS)
dM = torch.zeros((S0.size + 63) // 64, dtype=torch.int64, device="cuda")
masks = {}
for frac in [float(x) for x in sys.argv[1:]] or [0.002, 0.01, 0.02, 0.03, 0.05]:
    S, E = syn.clipped_mix(S0, E0, frac)
    dS.copy_(torch.from_numpy(S.view(np.int32))); dE.copy_(torch.from_numpy(E.view(np.int32)))
    with pkg.Solver(0) as sv:
        best = 1e9
        for _ in range(3):
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
            best = min(best, st.ms_total)
        near = dM.cpu().numpy().copy()
        d = st.as_dict()
        print(f"clipped {frac:5.3f}: {best:8.2f} ms path {d['path']} exceptions {d['near_uniform_exceptions']} kept {d['near_uniform_selected']} sweeps {d['near_uniform_rounds']}", flush=True)
        if frac in (0.03,):
            os.environ["QMCP_HIP_NEAR"] = "0"
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dM.data_ptr(), contig_read_offsets=offs)
            del os.environ["QMCP_HIP_NEAR"]
            print(f"               mixed-span route {st.ms_total:8.2f} ms, same mask: {bool(np.array_equal(near, dM.cpu().numpy()))}", flush=True)
