"""lab: the mixed-span route (near-uniform off) on cfg4's reads with 1 % clipped, KEEP share of the reads, at several M:
speculative stretch boundaries on (default) and off (QMCP_HIP_SPEC=0)"""
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
for M in [int(x) for x in sys.argv[1:]] or [150, 250]:
    out = []
    for spec in (None, "0"):
        if spec is None: os.environ.pop("QMCP_HIP_SPEC", None)
        else: os.environ["QMCP_HIP_SPEC"] = spec
        with pkg.Solver(0) as sv:
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
            d = st.as_dict()
            out.append(f"spec {'default' if spec is None else 'off'}: {d['ms_total']:.1f} ms, stretches {d['sweep_stretches']}, boundaries {d['spec_boundaries']}, mismatches {d['spec_mismatches']} / {d['spec_retry_mismatches']}")
    print(f"M {M} (depth {1875 * KEEP / M:.2f} x M): " + " | ".join(out), flush=True)
