#!/usr/bin/env python3
"""Writes tests/golden/regression_digests.json: FNV-1a-64 digests (oracle/qmcp_oracle.c mixing
rule) of this repository's own reads-gen streams, arc lists and oracle kept sets.  They are
regression anchors for THIS code base (generator restatement + oracle), produced by running it,
not reference-derived vectors -- those are in reference_vectors.json."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
pkg = importlib.import_module("genome-downsampler_amd")
import oracle_py as O  # noqa: E402

CASES = [("random_uniform_dist_test", 0, 1_000_000, 30_000, 1000),
         ("random_low_coverage_on_both_sides_test", 1, 1_000_000, 30_000, 8000),
         ("random_with_hole_test", 2, 1_000_000, 30_000, 8000),
         ("random_zero_coverage_on_both_sides_test", 3, 1_000_000, 30_000, 8000),
         ("cfg1", 0, 5000, 3000, 100), ("cfg2", 0, 500_000, 30_000, 100)]
out = {}
for name, kind, pairs, L, M in CASES:
    s, e, q = pkg.reads_gen(kind, pairs, L, with_qualities=True)
    g = O.graph(s, e, L, M)
    mask = O.solve(s, e, L, M)
    out[name] = {"reads_fnv": f"{O.reads_fnv(s, e, q):016x}", "arc_fnv": f"{g.arc_fnv:016x}",
                 "kept_fnv": f"{O.mask_fnv(mask, s.size):016x}",
                 "n_kept": int(pkg.mask_to_indices(mask, s.size).size)}
    print(name, out[name])
json.dump(out, open(os.path.join(os.path.dirname(__file__), "regression_digests.json"), "w"), indent=1)
