"""Pins the oracle's deterministic stages and the reads-gen restatement to vectors captured
from the reference's own code (tests/golden/reference_vectors.json)."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.json")))
DIGESTS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "regression_digests.json")))
INT64_MAX = (1 << 63) - 1


def test_small_fixture_cov_b_d_arcs(oracle):
    f = GOLD["small_fixture"]
    s, e, L, M = f["starts"], f["ends"], f["ref_genome_length"], f["M"]
    assert oracle.cover(s, e, L).tolist() == f["cov"]
    assert oracle.b_function(s, e, L, M).tolist() == f["b"]
    assert oracle.demand_function(s, e, L, M).tolist() == f["d"]
    info, arcs = oracle.graph(s, e, L, M, want_arcs=True)
    assert info.n_arcs == f["n_arcs"] == len(arcs)
    # arc id == ReadIndex for the first N arcs, cap 1 (quasi_mcp_cpu_max_flow_solver.cpp:34-36)
    assert arcs[:16].tolist() == [[a, b, 1] for a, b in f["read_arcs"]]
    # back arcs i+1 -> i, i = 0..10, cap INT64_MAX (:39-41)
    assert arcs[16:27].tolist() == [[i + 1, i, INT64_MAX] for i in range(11)]
    assert arcs[27:].tolist() == f["terminal_arcs"]
    assert info.total_supply == 5 == info.total_demand


def test_small_fixture_reference_answer_is_a_maximum_flow(oracle, pkg):
    """the survey's substitute max-flow answer and the oracle's canonical answer are both
    maximum flows of the same network: same value, both valid, canonical one never larger"""
    f = GOLD["small_fixture"]
    s, e, L, M = f["starts"], f["ends"], f["ref_genome_length"], f["M"]
    other = pkg.indices_to_mask(f["illustrative_valid_answer"]["kept"], 16)
    ok, val = oracle.check_flow(s, e, L, M, other)
    assert ok and val == f["illustrative_valid_answer"]["flow"]
    mine = oracle.solve(s, e, L, M)
    ok, val2 = oracle.check_flow(s, e, L, M, mine)
    assert ok and val2 == val == oracle.maxflow_value(s, e, L, M)
    assert pkg.mask_to_indices(mine, 16).size <= len(f["illustrative_valid_answer"]["kept"])


@pytest.mark.parametrize("case", GOLD["generator_and_graph"], ids=lambda c: c["name"])
def test_generator_and_graph_counts(oracle, pkg, case):
    s, e, q = pkg.reads_gen(case["kind"], case["pairs"], case["L"], with_qualities=True)
    for i, (rs, re, rq) in enumerate(case.get("first_reads", [])):
        assert (int(s[i]), int(e[i]), int(q[i])) == (rs, re, rq)
    want_arcs = "terminal_arcs" in case or "source_caps_nodes_0_to_14" in case
    res = oracle.graph(s, e, case["L"], case["M"], want_arcs=want_arcs)
    info = res[0] if want_arcs else res
    assert info.n_arcs == case["n_arcs"]
    assert info.n_terminal_arcs == case["n_terminal_arcs"]
    assert info.total_supply == case["supply"] == info.total_demand
    if want_arcs:
        term = res[1][s.size + case["L"]:]
        if "terminal_arcs" in case:
            assert term.tolist() == case["terminal_arcs"]
        if "source_caps_nodes_0_to_14" in case:
            src = term[term[:, 0] == case["L"] + 1]
            assert src[:, 1].tolist() == list(range(15))
            assert src[:, 2].tolist() == case["source_caps_nodes_0_to_14"]
            snk = term[term[:, 1] == case["L"] + 2]
            assert len(snk) == case["n_sink_arcs"]
            assert [int(snk[0, 0]), int(snk[-1, 0])] == case["sink_nodes_range"]
    # regression anchors of this repository's own code
    dg = DIGESTS[case["name"]]
    assert f"{oracle.reads_fnv(s, e, q):016x}" == dg["reads_fnv"]
    assert f"{info.arc_fnv:016x}" == dg["arc_fnv"]


def test_generator_aos_and_soa_paths_agree(pkg):
    for kind in range(4):
        a = pkg.reads_gen(kind, 3000, 2000, with_qualities=True)
        b = pkg.reads_gen(kind, 3000, 2000, with_qualities=True, aos=True)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        s, e, _ = a
        assert np.all(e - s == 149) and e.max() < 2000
        # mates never overlap and the first mate starts first (reads_gen.cpp:35-44,71-77)
        assert np.all(s[1::2] >= s[0::2] + 150)


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_oracle_kept_set_regression(oracle, pkg, name):
    case = next(c for c in GOLD["generator_and_graph"] if c["name"] == name)
    s, e = pkg.reads_gen(case["kind"], case["pairs"], case["L"])
    mask = oracle.solve(s, e, case["L"], case["M"])
    assert f"{oracle.mask_fnv(mask, s.size):016x}" == DIGESTS[name]["kept_fnv"]
    ok, val = oracle.check_flow(s, e, case["L"], case["M"], mask)
    assert ok and val == case["supply"]
    # on deep uniform data the canonical answer keeps exactly M * L / len reads
    assert DIGESTS[name]["n_kept"] == case["M"] * case["L"] // 150
