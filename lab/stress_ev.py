"""Many random cases through the event-driven uniform sweep (forced with QMCP_HIP_SWEEP=ev), aimed at its rare
paths: profiles that change at nearly every block, amounts handed back across block and 64-block group borders,
blocks that are not deep in the middle of a contig, counts at the packed fields' maximum, every lane layout,
stretches behind cut points, contigs of a handful of blocks.   python lab/stress_ev.py <first seed> <last seed>"""
import importlib, os, sys
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
pkg = importlib.import_module("genome-downsampler_amd")
sol = pkg.Solver(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
sol.set_options(sweep="ev")
bad = ran = 0
kinds = {}


def case(rng):
    span = int(rng.choice([32, 40, 64, 65, 100, 128, 129, 150, 192, 193, 230, 256]))
    e_lanes = (span + 63) // 64
    m_max = {1: 5000, 2: 5000, 3: 510, 4: 62}[e_lanes]
    M = int(min(m_max, rng.choice([1, 2, 3, 7, 20, 61, 62, 100, 300, 510])))
    n_contigs = int(rng.integers(1, 5))
    lengths, counts, ss = [], [], []
    kind = str(rng.choice(["deep", "around_M", "sparse", "spiky", "ramps", "tiny"]))
    for _ in range(n_contigs):
        L = int(rng.integers(span, 8 * span)) if kind == "tiny" else int(rng.integers(span, 70_000))
        hi_ = L - span + 1
        if kind == "deep":
            c = int(min(400_000, L * M * rng.uniform(8, 40) / span))
            s = rng.integers(0, hi_, size=c)
        elif kind == "around_M":
            c = int(L * M * rng.uniform(0.5, 3.0) / span)
            s = rng.integers(0, hi_, size=c)
        elif kind == "sparse":
            c = int(L * rng.uniform(0.02, 0.7))
            s = rng.integers(0, hi_, size=c)
        elif kind == "spiky":   # reads start in a few windows only (amplicon-like): long hand-backs
            c = int(min(300_000, L * M * rng.uniform(2, 20) / span))
            w = rng.integers(0, hi_, size=int(rng.integers(1, 12)))
            s = np.clip(rng.choice(w, size=c) + rng.integers(0, 26, size=c), 0, hi_ - 1)
        elif kind == "ramps":   # deep islands with gaps and slopes: blocks that are not deep mid-contig
            c = int(min(300_000, L * M * rng.uniform(4, 15) / span))
            s = rng.integers(0, hi_, size=c)
            s = s[((s // 3000) % 3 != 1) | (rng.random(c) < 0.03)]
        else:
            c = int(rng.integers(0, 20_000))
            s = rng.integers(0, hi_, size=c)
        if rng.random() < 0.1:
            s = s[:0]
        lengths.append(L); counts.append(s.size); ss.append(s.astype(np.uint32))
    s = np.concatenate(ss)
    return kind, s, (s + np.uint32(span - 1)).astype(np.uint32), np.array(lengths, np.uint32), \
        np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64), M


for seed in range(lo, hi):
    rng = np.random.default_rng(77_000_000 + seed)
    kind, s, e, lengths, offs, M = case(rng)
    if s.size == 0:
        continue
    sol.set_options(sweep="ev", cut_points=1 if int(rng.integers(0, 2)) else -1)
    got = sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    want = oracle_py.solve(s, e, lengths, M, contig_read_offsets=offs)
    ran += 1
    kinds[kind] = kinds.get(kind, 0) + 1
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH seed", seed, kind, "span", int(e[0] - s[0]) + 1, "M", M, lengths.tolist(), s.size, flush=True)
    if (seed - lo) % 200 == 199:
        print("...", seed - lo + 1, "seeds, mismatches", bad, flush=True)
print("cases", ran, "mismatches", bad, kinds)
