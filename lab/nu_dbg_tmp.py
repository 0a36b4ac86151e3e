import sys, importlib, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
pkg = importlib.import_module('genome-downsampler_amd')
import test_gpu_near_uniform as T
L, depth, M, fraction = 400_000, 1.5, 100, 0.01
rng = np.random.default_rng(int(L * depth) % 7919)
lengths = np.array([L, L // 2 + 12_345], np.uint32)
counts = [int(depth * M * int(x) / 150) for x in lengths]
s, e, offs = T._contigs(rng, lengths, counts, 150, fraction, 50)
with pkg.Solver(0) as sv:
    sv.set_options(near_uniform_debug=1)
    sv.solve(s, e, lengths, M, contig_read_offsets=offs)
    print(sv.last_stats.as_dict())
