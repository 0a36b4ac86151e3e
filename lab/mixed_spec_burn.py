"""lab: mixed-span route (near-uniform off) with the run-in of its speculative boundaries forced (QMCP_HIP_SPEC_BURN)"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
KEEP = float(os.environ.get("KEEP", "0.4")); per = int(12_500_000 * KEEP)
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a[:per]); ee.append(b[:per])
S, E = syn.clipped_mix(np.concatenate(ss), np.concatenate(ee), 0.01)
offs = np.arange(9, dtype=np.uint64) * np.uint64(per)
lengths = np.full(8, 1_000_000, np.uint32)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
os.environ["QMCP_HIP_NEAR"] = "0"
for arg in sys.argv[1:]:
    M, burn = [int(x) for x in arg.split(":")]
    os.environ["QMCP_HIP_SPEC_BURN"] = str(burn)
    with pkg.Solver(0) as sv:
        st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
        d = st.as_dict()
    print(f"M {M} (depth {1875 * KEEP / M:.2f} x M) run-in {burn} blocks: {d['ms_total']:.1f} ms, stretches {d['sweep_stretches']}, boundaries {d['spec_boundaries']}, mismatches {d['spec_mismatches']} / {d['spec_retry_mismatches']}", flush=True)
