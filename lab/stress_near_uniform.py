"""lab: random instances through the near-uniform route against the oracle.
   python lab/stress_near_uniform.py <cases> [seed0] [shallow]
shallow: depths 1.4 - 8 x M on contigs long enough for the sweeps' speculative boundaries (small spans and M keep the
oracle quick) -- the route's sweeps in stretches (round 4)"""
import os, sys, importlib, collections
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
pkg = importlib.import_module("genome-downsampler_amd")
import oracle_py

cases = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shallow = len(sys.argv) > 3 and sys.argv[3] == "shallow"
stretches = 0
paths = collections.Counter(); rounds = collections.Counter(); wrong = 0; selected = 0
with pkg.Solver(0) as sv:
    for case in range(cases):
        rng = np.random.default_rng(seed0 + case)
        n_contigs = int(rng.integers(1, 6))
        ell = int(rng.choice([40, 64, 100, 150, 151, 200, 250]))
        M = int(rng.choice([20, 50, 100, 200]))
        depth = float(rng.choice([6.5, 9, 12.5, 19, 30]))
        lengths = rng.integers(4 * ell, 60_000, size=n_contigs)
        if shallow:
            n_contigs = int(rng.integers(1, 4))
            ell = int(rng.choice([40, 64, 100]))
            M = int(rng.choice([6, 10, 16]))
            depth = float(rng.choice([1.4, 1.7, 2.2, 3.0, 4.5, 8.0]))
            lengths = rng.integers(4 * ell, int(rng.choice([60_000, 400_000, 900_000])), size=n_contigs)
        counts = np.maximum((depth * M * lengths / ell).astype(np.int64), 1)
        scale = max(1.0, 140_000 / counts.sum())          # the ranked route wants >= 128 Ki reads
        if counts.sum() > 1_500_000: scale = 1_500_000 / counts.sum()
        counts = np.maximum((counts * scale).astype(np.int64), 1)
        if rng.random() < 0.15: counts[rng.integers(0, n_contigs)] = 0
        frac = float(rng.choice([0.001, 0.005, 0.01, 0.02, 0.04]))
        max_clip = int(rng.integers(1, ell - 1))
        style = rng.choice(["spread", "cluster", "twins", "ends"])
        ss, ee = [], []
        for L, k in zip(lengths, counts):
            L = int(L); k = int(k)
            s = rng.integers(0, L - ell + 1, size=k).astype(np.int64); e = s + ell - 1
            if style == "cluster":      # every clipped read inside one window
                lo = int(rng.integers(0, max(1, L - 3 * ell))); pick = (s >= lo) & (s < lo + 3 * ell) & (rng.random(k) < min(1.0, frac * L / (3 * ell)))
            elif style == "ends":       # clipped reads at the contig's two ends
                pick = ((s < 2 * ell) | (s > L - 3 * ell)) & (rng.random(k) < min(1.0, 20 * frac))
            else:
                pick = rng.random(k) < frac
            clip = rng.integers(1, max_clip + 1, size=k); front = rng.random(k) < 0.5
            if style == "twins" and pick.any():   # groups of identical clipped reads (ties settled by read index)
                j = np.flatnonzero(pick); s[j] = s[j[0]]; e[j] = e[j[0]]; clip[j] = clip[j[0]]; front[j] = front[j[0]]
            s = np.where(pick & front, s + clip, s); e = np.where(pick & ~front, e - clip, e)
            ss.append(s.astype(np.uint32)); ee.append(e.astype(np.uint32))
        s = np.concatenate(ss); e = np.concatenate(ee)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        L32 = lengths.astype(np.uint32)
        got = sv.solve(s, e, L32, M, contig_read_offsets=offs)
        st = sv.last_stats
        want = oracle_py.solve(s, e, L32, M, offs)
        paths[int(st.path)] += 1
        if st.path == pkg.PATH_NEAR_UNIFORM:
            rounds[int(st.near_uniform_rounds)] += 1; selected += int(st.near_uniform_selected)
            stretches += int(st.sweep_stretches)
            bits = np.unpackbits(want.view(np.uint8), bitorder="little")[:s.size].astype(bool)
            exc_kept = int((bits & ((e - s + 1) != ell)).sum())
            if exc_kept != int(st.near_uniform_selected):
                print("STAT case", seed0 + case, "exceptions kept", exc_kept, "stats say", int(st.near_uniform_selected), flush=True)
        if not np.array_equal(got, want):
            wrong += 1
            print("MISMATCH case", seed0 + case, dict(n_contigs=n_contigs, ell=ell, M=M, depth=depth, frac=frac, style=str(style)), st.as_dict(), flush=True)
        if case % 25 == 24:
            print(f"{case + 1} cases, {wrong} wrong, paths {dict(paths)}, rounds {dict(sorted(rounds.items()))}, selected exceptions {selected}", flush=True)
print(f"done: {cases} cases, {wrong} wrong, paths {dict(paths)}, rounds {dict(sorted(rounds.items()))}, selected exceptions {selected}, stretches swept {stretches}")
