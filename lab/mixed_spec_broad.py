"""lab: the mixed-span route on a BROAD mix of spans (100...150, uniform) at several depths: speculative boundaries on / off"""
import os, sys, importlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
import torch
rng = np.random.default_rng(3)
L, C, M = 1_000_000, 8, 100
for depth in [float(x) for x in sys.argv[1:]] or [3.0, 6.0, 9.0]:
    per = int(depth * M * L / 125)
    span = rng.integers(100, 151, size=per * C)
    s = (rng.random(per * C) * (L - span + 1)).astype(np.int64)
    e = s + span - 1
    S, E = s.astype(np.uint32), e.astype(np.uint32)
    offs = np.arange(C + 1, dtype=np.uint64) * np.uint64(per)
    lengths = np.full(C, L, np.uint32)
    dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
    dM = torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda")
    out = []
    for spec in (None, "0"):
        if spec is None: os.environ.pop("QMCP_HIP_SPEC", None)
        else: os.environ["QMCP_HIP_SPEC"] = spec
        with pkg.Solver(0) as sv:
            st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
            d = st.as_dict()
            out.append(f"spec {'default' if spec is None else 'off'}: {d['ms_total']:.1f} ms, stretches {d['sweep_stretches']}, boundaries {d['spec_boundaries']}, mismatches {d['spec_mismatches']} / {d['spec_retry_mismatches']}")
    print(f"depth {depth} x M, {S.size} reads: " + " | ".join(out), flush=True)
    del dS, dE, dM
