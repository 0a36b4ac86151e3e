"""Probe: does solving two halves of cfg4 on two contexts, the second one delayed, beat one solve?"""
import importlib, os, sys, threading, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("genome-downsampler_amd")
pairs, L, M = 6_250_000, 1_000_000, 100
ss, ee = zip(*[pkg.reads_gen(0, pairs, L, seed=12345 + c) for c in range(8)])
dev = torch.device("cuda:0")
def half(lo, hi):
    s = torch.from_numpy(np.concatenate(ss[lo:hi]).view(np.int32)).to(dev)
    e = torch.from_numpy(np.concatenate(ee[lo:hi]).view(np.int32)).to(dev)
    n = s.numel()
    offs = np.arange(hi - lo + 1, dtype=np.uint64) * np.uint64(2 * pairs)
    lengths = np.full(hi - lo, L, np.uint32)
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    return s, e, n, offs, lengths, mask
full = half(0, 8); A = half(0, 4); B = half(4, 8)
sol_full, solA, solB = pkg.Solver(0), pkg.Solver(0), pkg.Solver(0)
def run(sol, h):
    s, e, n, offs, lengths, mask = h
    sol.solve_device(s.data_ptr(), e.data_ptr(), n, lengths, M, mask.data_ptr(), contig_read_offsets=offs, stream=0)
for _ in range(3):
    run(sol_full, full); run(solA, A); run(solB, B)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run(sol_full, full)
torch.cuda.synchronize()
print("one solve of 8 contigs: %.3f ms" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10):
    run(solA, A); run(solB, B)
torch.cuda.synchronize()
print("two half solves back to back: %.3f ms" % ((time.perf_counter() - t0) * 100))
for delay_us in (0, 300, 500, 700, 900):
    def worker(sol, h, d):
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < d:
            pass
        run(sol, h)
    t0 = time.perf_counter()
    for _ in range(10):
        ta = threading.Thread(target=worker, args=(solA, A, 0))
        tb = threading.Thread(target=worker, args=(solB, B, delay_us))
        ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    print("two halves on two threads, second delayed %4d us: %.3f ms per pair" % (delay_us, (time.perf_counter() - t0) * 100))
