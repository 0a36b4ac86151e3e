"""lab: cfg4 solved again and again (one context, then two in flight): every mask's digest must be the first one's --
the check that showed the in-flight-register copy of round 4's first walk (cfg4 differed from run to run).
   python lab/soak_cfg4.py [solves = 60]"""
import sys, importlib, os, hashlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
import torch
n_solves = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ss, ee = [], []
for c in range(8):
    a, b = pkg.reads_gen(0, 6_250_000, 1_000_000, seed=12345 + c); ss.append(a); ee.append(b)
S = np.concatenate(ss); E = np.concatenate(ee)
offs = np.arange(9, dtype=np.uint64) * 12_500_000
lengths = np.full(8, 1_000_000, np.uint32)
dS = torch.from_numpy(S.view(np.int32)).cuda(); dE = torch.from_numpy(E.view(np.int32)).cuda()
svs = [pkg.Solver(0), pkg.Solver(0)]
dMs = [torch.zeros((S.size + 63) // 64, dtype=torch.int64, device="cuda") for _ in range(2)]
digests = set()
for i in range(n_solves):
    svs[0].solve_device(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dMs[0].data_ptr(), contig_read_offsets=offs)
    digests.add(hashlib.sha1(dMs[0].cpu().numpy().tobytes()).hexdigest())
print("one at a time:", n_solves, "solves,", len(digests), "distinct masks", flush=True)
for i in range(n_solves):
    k = i % 2
    if i >= 2:
        svs[k].solve_end()
        digests.add(hashlib.sha1(dMs[k].cpu().numpy().tobytes()).hexdigest())
    svs[k].solve_device_begin(dS.data_ptr(), dE.data_ptr(), S.size, lengths, 100, dMs[k].data_ptr(), contig_read_offsets=offs)
for k in range(2):
    svs[k].solve_end()
    digests.add(hashlib.sha1(dMs[k].cpu().numpy().tobytes()).hexdigest())
print("two in flight:", n_solves, "more solves,", len(digests), "distinct masks in all:", sorted(digests)[0][:12])
sys.exit(0 if len(digests) == 1 else 1)
