"""lab: random one-length inputs through the PASS-MAJOR form of the range-ranked route (QMCP_HIP_PM=1 forces it where
the host would keep the range-major form: small genomes, short slices) against the oracle -- every sweep form the host
picks, ragged contigs (empty, tiny, straddled ranges), hot spots, quotas that run out inside chunks; every third case
with a few shorter reads (the near-uniform route's filtered producer on this form).
   python lab/stress_pm.py <first seed> <last seed>"""
import importlib, os, sys, collections
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
pkg = importlib.import_module("genome-downsampler_amd")
sol = pkg.Solver(0)
sol.set_options(pass_major=1)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
routes = collections.Counter()
for seed in range(lo, hi):
    rng = np.random.default_rng(4_000_000 + seed)
    span = int(rng.choice([30, 64, 100, 150, 151, 200, 256, 300]))
    n_contigs = int(rng.integers(1, 7))
    M = int(rng.choice([1, 2, 5, 17, 50, 100, 300]))
    kind = rng.choice(["deep", "mid", "shallow", "long"])
    lengths, counts, ss = [], [], []
    for _ in range(n_contigs):
        L = int(rng.integers(span, 60_000)) if kind != "long" else int(rng.integers(200_000, 1_500_000))
        depth = {"deep": rng.uniform(8, 40), "mid": rng.uniform(2, 8), "shallow": rng.uniform(0.3, 2), "long": rng.uniform(0.5, 6)}[kind]
        c = int(min(500_000, max(1, L * M * depth / span)))
        lengths.append(L); counts.append(c)
    if rng.random() < 0.3:
        counts[int(rng.integers(0, n_contigs))] = 0
    tot = sum(counts)
    if tot < 140_000:                                   # the ranked route wants >= 128 Ki reads
        k = int(np.argmax(counts)); counts[k] += 140_000 - tot
    for L, c in zip(lengths, counts):
        top = L - span + 1
        if rng.random() < 0.35 and top > 50:            # hot spots: a few start positions hold most reads
            spots = rng.integers(0, top, size=int(rng.integers(1, 6)))
            s = np.where(rng.random(c) < 0.7, rng.choice(spots, size=c), rng.integers(0, top, size=c))
        elif rng.random() < 0.2:                        # sorted by position: a pass's reads in one or two ranges
            s = np.sort(rng.integers(0, top, size=c))
        else:
            s = rng.integers(0, top, size=c)
        ss.append(s.astype(np.uint32))
    s = np.concatenate(ss)
    e = (s + np.uint32(span - 1)).astype(np.uint32)
    if seed % 3 == 2 and span > 40:                    # a few shorter reads
        clip = rng.random(s.size) < float(rng.choice([0.002, 0.01, 0.03]))
        e = np.where(clip, e - rng.integers(1, span // 2, size=s.size).astype(np.uint32), e).astype(np.uint32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    lengths = np.array(lengths, np.uint32)
    got = sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle_py.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = sol.last_stats
    routes[(str(kind), int(st.path), int(st.sort_passes))] += 1
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH seed", seed, s.size, lengths.tolist(), M, span, flush=True)
    if (seed - lo) % 50 == 49:
        print("...", seed - lo + 1, "seeds, mismatches", bad, flush=True)
print("seeds", hi - lo, "mismatches", bad, "routes (kind, path, sort passes):", dict(routes))
