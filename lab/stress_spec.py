"""Speculative stretch boundaries under stress: random one-span instances at depths 1.3 ... 3.5 x M with the
run-in forced SHORT (8 ... 128 blocks), so that many boundaries are accepted only just.  Whatever the
verification decides, the kept set must equal the oracle's; the interesting class is "boundaries accepted
(no mismatch) and the run-in far shorter than the product would use".
   python lab/stress_spec.py [instances = 300] [seed = 1] [big]
   (big: genomes of 8 ... 20 M positions, so that the mixed-span route's tables hold more than a thousand stretches)"""
import importlib, os, sys, time
import numpy as np
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "oracle"))
import oracle_py
pkg = importlib.import_module("genome-downsampler_amd")
n_inst = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
sol = pkg.Solver(0)
sol.set_options(speculation=1)
accepted = rejected = none = bad = second = 0
acc_boundaries = 0
t0 = time.time()
for it in range(n_inst):
    span = int(rng.choice([40, 64, 100, 150, 151, 200, 250]))
    M = int(rng.integers(8, 60))
    depth = float(rng.uniform(1.3, 3.5))
    burn = int(rng.choice([8, 16, 32, 64, 128]))
    n_contigs = int(rng.integers(1, 4))
    if big:
        lengths = rng.integers(8_000_000, 20_000_000, size=n_contigs).astype(np.uint32) // np.uint32(n_contigs)
        M = int(rng.integers(8, 25))
    else:
        lengths = rng.integers(8 * burn * span + 1000, max(8 * burn * span + 2000, 1_200_000), size=n_contigs).astype(np.uint32)
    ss, offs = [], [0]
    for L in lengths:
        n = int(depth * M * int(L) / span)
        ss.append(rng.integers(0, int(L) - span + 1, size=n, dtype=np.uint32))
        offs.append(offs[-1] + n)
    s = np.concatenate(ss)
    mixed = rng.random() < 0.4
    if mixed:   # a mix of lengths up to `span`: the register-resident event sweep (the run-in counts windows of `span`)
        lo = int(rng.integers(max(span // 2, 20), span))
        e = s + rng.integers(lo, span + 1, size=s.size).astype(np.uint32) - 1
        e = np.minimum(e, np.repeat(lengths, np.diff(np.asarray(offs))).astype(np.uint32) - 1)
    else:
        e = s + np.uint32(span - 1)
    offs = np.asarray(offs, np.uint64)
    sol.set_options(speculation=1, speculation_run_in=int(burn))
    got = sol.solve(s, e, lengths, M, contig_read_offsets=offs)
    st = sol.last_stats
    want = oracle_py.solve(s, e, lengths, M, contig_read_offsets=offs)
    same = np.array_equal(got, want)
    if not same:
        bad += 1
        print(f"MISMATCH with the oracle: mixed {mixed} span {span} M {M} depth {depth:.2f} burn {burn} lengths {lengths.tolist()} "
              f"spec {st.spec_boundaries} disagreeing {st.spec_mismatches}", flush=True)
    if st.spec_boundaries == 0: none += 1
    elif st.spec_mismatches == 0: accepted += 1; acc_boundaries += st.spec_boundaries
    elif st.spec_retry_mismatches == 0: second += 1
    else: rejected += 1
    if (it + 1) % (5 if big else 50) == 0:
        print(f"{it + 1} instances, {time.time() - t0:.0f} s: all boundaries accepted {accepted} (boundaries {acc_boundaries}), "
              f"settled by the second tier {second}, exact sweep needed {rejected}, none speculative {none}, wrong kept sets {bad}", flush=True)
print(f"done: {n_inst} instances, first tier accepted {accepted} ({acc_boundaries} boundaries), second tier {second}, exact sweep {rejected}, none {none}, WRONG {bad}")
sys.exit(1 if bad else 0)
