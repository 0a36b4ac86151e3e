"""lab: where are the cliffs?  Random shapes (one contig or several, positions 10^6 ... 10^8, M 10 ... 400, depth 1.2 ... 30 x M)
in three variants -- one read length, 1 % clipped, 1 % clipped + 0.5 % lengthened -- through the library's own choices; prints
every case and, at the end, the worst by nanoseconds per read.  (No oracle: parity is the tests' and the stress harnesses'.)
   python lab/cliff_hunt.py [cases = 30] [seed = 0]"""
import os, sys, importlib, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
pkg = importlib.import_module('genome-downsampler_amd')
syn = importlib.import_module('genome-downsampler_amd.synthetic')
import torch
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rows = []; ratios = []
with pkg.Solver(0) as sv:
    for case in range(cases):
        n_contigs = int(rng.choice([1, 1, 2, 8, 24]))
        ltot = int(10 ** rng.uniform(6.0, 8.0))
        M = int(rng.choice([10, 20, 50, 100, 200, 400]))
        depth = float(rng.choice([1.2, 1.6, 2.0, 2.5, 3.0, 4.0, 6.0, 9.0, 12.0, 20.0, 30.0]))
        ell = int(rng.choice([76, 100, 150, 150, 250]))
        frac = float(rng.choice([0.002, 0.01, 0.05]))
        n = int(depth * M * ltot / ell)
        if n > 140_000_000 or n < 300_000: continue
        w = rng.dirichlet(np.ones(n_contigs) * 3.0)
        lengths = np.maximum((w * ltot).astype(np.int64), 2000).astype(np.uint32)
        counts = np.maximum((n * lengths.astype(np.float64) / lengths.sum()).astype(np.int64), 1)
        ss = [rng.integers(0, int(L) - ell - 24, size=int(k)).astype(np.uint32) for L, k in zip(lengths, counts)]
        s0 = np.concatenate(ss); e0 = (s0 + np.uint32(ell - 1)).astype(np.uint32)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        s1, e1 = syn.clipped_mix(s0, e0, frac, max_clip=min(50, ell // 2))
        e2 = syn.lengthened_mix(s1, e1, offs, lengths, 0.005)
        dM = torch.zeros((s0.size + 63) // 64, dtype=torch.int64, device="cuda")
        base_ms = None
        for name, (s, e) in (("one length", (s0, e0)), ("clipped", (s1, e1)), ("clipped + longer", (s1, e2))):
            dS = torch.from_numpy(s.view(np.int32)).cuda(); dE = torch.from_numpy(e.view(np.int32)).cuda()
            best = None
            for rep in range(2):
                st = sv.solve_device(dS.data_ptr(), dE.data_ptr(), s.size, lengths, M, dM.data_ptr(), contig_read_offsets=offs)
                if best is None or st.ms_total < best.ms_total: best = st
            d = best.as_dict()
            if base_ms is None: base_ms = d["ms_total"]
            ratios.append((d["ms_total"] / base_ms, name, ell, frac, n_contigs, int(lengths.max()), M, depth, s.size, d["ms_total"], base_ms, d["path"], d["near_uniform_giveup"]))
            row = (d["ms_total"] * 1e6 / s.size, name, n_contigs, int(lengths.max()), ltot, M, depth, s.size, d["ms_total"], d["path"], d["near_uniform_giveup"],
                   d["sweep_stretches"], d["spec_boundaries"], d["spec_mismatches"], d["spec_retry_mismatches"])
            rows.append(row)
            print("%7.2f ns/read  %-16s contigs %2d longest %9d of %9d  M %3d  depth %4.1f  reads %9d: %9.2f ms path %d giveup %d stretches %5d boundaries %4d disagreeing %3d / %3d" % row, flush=True)
            del dS, dE
print("--- worst by ns per read")
for row in sorted(rows, reverse=True)[:12]:
    print("%7.2f ns/read  %-16s contigs %2d longest %9d of %9d  M %3d  depth %4.1f  reads %9d: %9.2f ms path %d giveup %d stretches %5d boundaries %4d disagreeing %3d / %3d" % row)
print("--- worst by ratio to the same reads with one length")
for r in sorted(ratios, reverse=True)[:14]:
    print("%7.1f x  %-16s ell %3d clipped %.3f contigs %2d longest %9d M %3d depth %4.1f reads %9d: %9.2f ms against %8.2f  path %d giveup %d" % r)
