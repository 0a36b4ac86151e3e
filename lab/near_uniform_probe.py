"""near-uniform route: what a call did (route, exceptions, rounds, selected) and how long it took, against the
mixed-span route on the same reads.  usage: near_uniform_probe.py [cfg4 fraction | small]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import importlib
pkg = importlib.import_module("genome-downsampler_amd")
from test_gpu_near_uniform import _contigs, CASES

def show(tag, st):
    d = st.as_dict()
    print(tag, {k: d[k] for k in ("path", "near_uniform_exceptions", "near_uniform_rounds", "near_uniform_selected",
                                  "n_kept", "ms_total", "ms_sweep", "ms_mark", "arena_grown_mid_solve")}, flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else "small"
with pkg.Solver(0) as solver:
    if mode == "small":
        import oracle_py
        for lengths, counts, span, M, fraction, max_clip in CASES + [([30_000, 30_000], [200_000, 200_000], 150, 100, 0.05, 50)]:
            rng = np.random.default_rng(sum(counts) % 9973 + span)
            lengths = np.array(lengths, np.uint32)
            s, e, offs = _contigs(rng, lengths, counts, span, fraction, max_clip)
            for rep in range(2):
                got = solver.solve(s, e, lengths, M, contig_read_offsets=offs)
                show(f"{list(lengths)} f={fraction} rep{rep}", solver.last_stats)
            print("   equal to oracle:", bool(np.array_equal(got, oracle_py.solve(s, e, lengths, M, offs))), flush=True)
    else:
        import torch
        synthetic = importlib.import_module("genome-downsampler_amd.synthetic")
        frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
        n_contigs, pairs, L = 8, 6_250_000, 1_000_000
        ss, ee = [], []
        for c in range(n_contigs):
            a, b = pkg.reads_gen(pkg.KIND_UNIFORM, pairs, L, 150, seed=12345 + c)
            ss.append(a); ee.append(b)
        s, e = synthetic.clipped_mix(np.concatenate(ss), np.concatenate(ee), frac)
        offs = (np.arange(n_contigs + 1, dtype=np.uint64) * np.uint64(2 * pairs))
        lengths = np.full(n_contigs, L, np.uint32)
        ds = torch.from_numpy(s.view(np.int32)).cuda(); de = torch.from_numpy(e.view(np.int32)).cuda()
        n = s.size
        masks = {}
        for env in ("1", "0"):
            os.environ["QMCP_HIP_NEAR"] = env
            mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
            for rep in range(3):
                t0 = time.perf_counter()
                solver.solve_device(ds.data_ptr(), de.data_ptr(), n, lengths, 100, mask.data_ptr(), contig_read_offsets=offs)
                torch.cuda.synchronize()
                show(f"cfg4 clipped {frac} NEAR={env} rep{rep} wall {1e3 * (time.perf_counter() - t0):.2f} ms", solver.last_stats)
            masks[env] = mask.cpu().numpy()
        print("near-uniform mask == mixed-span mask:", bool(np.array_equal(masks["1"], masks["0"])), flush=True)
