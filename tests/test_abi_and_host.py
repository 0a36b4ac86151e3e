"""The C-ABI library loads and exports every symbol include/qmcp_hip.h declares; the host
mirror keeps the reference's plugin surface; without a GPU the product fails loudly."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "qmcp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qmcp_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    declared = header_functions()
    assert len(declared) >= 12
    assert sorted(pkg.ABI_SYMBOLS) == declared
    assert sorted(pkg.exported_symbols()) == declared
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg.HIP_LIB_PATH], capture_output=True, text=True)
    exported = set(re.findall(r" T (qmcp_hip_\w+)", nm.stdout))
    assert exported == set(declared)
    assert pkg.abi_version() == 5


def test_signatures_are_plain_c(pkg):
    """no C++ / torch types at the boundary: the header compiles as C"""
    src = '#include "qmcp_hip.h"\nint main(void){ return qmcp_hip_abi_version() == QMCP_HIP_ABI_VERSION ? 0 : 1; }\n'
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I",
                          os.path.join(ROOT, "include"), "-x", "c", "-"], input=src, text=True,
                         capture_output=True)
    assert out.returncode == 0, out.stderr


def test_product_does_not_link_the_oracle(pkg):
    for lib in (pkg.HIP_LIB_PATH, pkg.HOST_LIB_PATH):
        ldd = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "oracle" not in ldd
        nm = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
        assert "qmcp_oracle" not in nm
    # source scan: comments may CITE the oracle; nothing may include, import, link or load it
    import re
    forbidden = re.compile(r'#\s*include\s*[<"][^>"]*oracle|\bimport\s+oracle|\bfrom\s+oracle|oracle_py|'
                           r'libqmcp_oracle|qmcp_oracle_\w+\s*\(|-lqmcp_oracle|dlopen\([^)]*oracle|CDLL\([^)]*oracle',
                           re.IGNORECASE)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "genome-downsampler_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                hit = forbidden.search(text)
                assert hit is None, f"{os.path.join(dirpath, f)} refers to the oracle: {hit.group(0)!r}"
    # bench.py may use the oracle only in its cpu_baseline leg (and the parity spot check beside it)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    timed = bench[bench.index("def main():"):bench.index("if world == 1 and not args.no_cpu_baseline")]
    assert "oracle" not in timed.replace("parity_vs_oracle_on_sample", "")


def test_solver_registry_mirrors_reference_surface(pkg):
    assert pkg.solver_names() == ["quasi-mcp-hip"]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="box has a GPU")
def test_no_gpu_fails_loudly_never_falls_back(pkg):
    assert pkg.device_count() == 0
    with pytest.raises(pkg.QmcpError) as ei:
        pkg.Solver(0)
    assert ei.value.code == -4 and "no CPU fallback" in str(ei.value)


def test_mask_helpers_roundtrip(pkg):
    idx = np.array([0, 1, 63, 64, 65, 127, 1000], np.uint64)
    m = pkg.indices_to_mask(idx, 1001)
    assert m.size == 16 and np.array_equal(pkg.mask_to_indices(m, 1001), idx)


def test_contig_sharding_helpers():
    import importlib
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    owned = sh.assign_contigs([10, 50, 20, 20, 5], 2)
    assert sorted(sum(owned, [])) == [0, 1, 2, 3, 4]
    loads = [sum([10, 50, 20, 20, 5][c] for c in o) for o in owned]
    assert abs(loads[0] - loads[1]) <= 10
    assert sh.assign_contigs([7] * 8, 8) == [[c] for c in range(8)]


def test_amplicon_set_from_bed_and_tsv(pkg, tmp_path):
    """BamApi::set_amplicon_filter restated (bam_api.cpp:53-95,101-186)"""
    bed = tmp_path / "primers.bed"
    bed.write_text(
        "MN908947.3\t30\t54\tnCoV_1_LEFT\tpool1\t+\n"
        "MN908947.3\t385\t410\tnCoV_1_RIGHT\tpool1\t-\n"
        "MN908947.3\t320\t342\tnCoV_2_LEFT\tpool2\t+\n"
        "MN908947.3\t704\t726\tnCoV_2_RIGHT\tpool2\t-\n"
        "MN908947.3\t999\t1020\tnCoV_2_RIGHT\tpool2\t-\n"   # duplicate name: first one is kept
        "MN908947.3\tabc\t10\tbroken\n"                        # unparsable: skipped
        "MN908947.3\t5\t9\t\n")                                 # empty name: skipped
    tsv = tmp_path / "pairs.tsv"
    tsv.write_text("nCoV_1_LEFT\tnCoV_1_RIGHT\n"
                   "nCoV_2_RIGHT\tnCoV_2_LEFT\n"   # listed right-first: ordered by start
                   "only_one\t\n")                  # invalid line: skipped
    a0, a1 = pkg.amplicons_from_files(bed, tsv)
    assert a0.tolist() == [30, 320] and a1.tolist() == [410, 726]
    # without a TSV: consecutive primers in name-sorted order
    a0, a1 = pkg.amplicons_from_files(bed)
    assert a0.tolist() == [30, 320] and a1.tolist() == [410, 726]
    # the start-ordering swap acts on the map entries (visible to later pairs), unknown names are (0, 0)
    tsv.write_text("nCoV_2_RIGHT\tnCoV_1_LEFT\n"    # swaps the two entries: 2_RIGHT := (30,54), 1_LEFT := (704,726)
                   "nCoV_1_LEFT\tnCoV_1_RIGHT\n"    # now (704,726) vs (385,410) -> swapped again -> [385, 726]
                   "ghost\tnCoV_2_LEFT\n")          # (0,0) and (320,342) -> [0, 342]
    a0, a1 = pkg.amplicons_from_files(bed, tsv)
    assert a0.tolist() == [30, 385, 0] and a1.tolist() == [726, 726, 342]
    with pytest.raises(OSError):
        pkg.amplicons_from_files(tmp_path / "missing.bed")


@pytest.mark.parametrize("layout", [0, 1])
def test_host_bamapi_mirror_matches_oracle(pkg, layout):
    """BamApi::find_input_cover / find_filtered_cover / find_pairs of the host mirror
    (bam_api.cpp:239-301) against the oracle's restatements"""
    import oracle_py
    rng = np.random.default_rng(5 + layout)
    n, L = 2000, 700
    s = rng.integers(0, L - 60, size=n).astype(np.uint32)
    e = (s + rng.integers(0, 60, size=n)).astype(np.uint32)
    ids = np.unique(rng.integers(0, n, size=300)).astype(np.uint64)
    cin, cout, paired = pkg.bamapi_probe(s, e, L, ids, layout=layout)
    mask = pkg.indices_to_mask(ids, n)
    assert np.array_equal(cin, oracle_py.cover(s, e, L))
    assert np.array_equal(cout, oracle_py.cover(s, e, L, keep_mask=mask))
    want = pkg.mask_to_indices(oracle_py.find_pairs(mask, n), n)
    assert np.array_equal(np.sort(paired), want)
    # reference order: each id followed by its mate, first occurrence only (bam_api.cpp:251-263)
    assert paired[0] == ids[0] and paired[1] == (ids[0] ^ 1)


def test_assignment_knows_which_contigs_sweep_as_stretches():
    """cfg5's shape on 8 ranks: at 2 x M every contig's sweep is cut into stretches, so a rank's cost is its
    reads (and positions), not its longest contig -- the shares come out even in reads; the same contigs at
    20 x M are chains, and the longest contig decides"""
    import importlib
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    mb = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]
    lengths = [int(1.5e9 * m / sum(mb)) for m in mb]
    reads = [int(1.0e9 * m / sum(mb)) for m in mb]                      # 100 x coverage at 150 bases
    shallow = sh.assign_contigs(reads, 8, contig_lengths=lengths, read_length=150, max_coverage=50)
    share = [sum(reads[c] for c in o) for o in shallow]
    assert max(share) < 1.08 * (sum(reads) / 8), share
    blind = sh.assign_contigs(reads, 8, contig_lengths=lengths)        # no depth: every contig counted as a chain
    assert max(sum(reads[c] for c in o) for o in blind) > max(share)
    deep = sh.assign_contigs(reads, 8, contig_lengths=lengths, read_length=150, max_coverage=5)
    assert deep == blind
    assert sh.rank_cost(reads, lengths, [0], 150, 50) < 0.1 * sh.rank_cost(reads, lengths, [0])


def test_a_share_is_priced_by_the_predicate_the_solver_uses():
    """one deep contig beside shallow ones: the solver judges the WHOLE share's depth (launch_uniform_sweep in
    csrc/qmcp_api.hip), so a rank that owns both sweeps whole contigs and pays for its longest; priced contig by
    contig the shallow ones would have counted as stretches"""
    import importlib
    sh = importlib.import_module("genome-downsampler_amd.sharding")
    L, rl, M = 20_000_000, 150, 50
    shallow_reads = int(2.0 * M * L / rl)            # 2 x M
    deep_reads = int(30.0 * M * L / rl)              # 30 x M
    assert sh.share_sweeps_as_stretches(shallow_reads, L, 1, rl, M)
    assert not sh.share_sweeps_as_stretches(deep_reads, L, 1, rl, M)
    # together: (2 + 30) / 2 = 16 x M over the share -> chains
    assert not sh.share_sweeps_as_stretches(shallow_reads + deep_reads, 2 * L, 2, rl, M)
    mixed = sh.rank_cost([shallow_reads, deep_reads], [L, L], [0, 1], rl, M)
    assert mixed == sh.NS_PER_READ * (shallow_reads + deep_reads) + sh.NS_PER_POSITION * L
    # too short for a run-in: no speculation even at 2 x M (8 run-ins of 320 blocks)
    assert not sh.share_sweeps_as_stretches(1000, 300_000, 1, rl, M) or 1000 * rl / (300_000 * M) <= sh.CUT_DEPTH
    assert not sh.share_sweeps_as_stretches(int(2.0 * M * 300_000 / rl), 300_000, 1, rl, M)
    # 256 contigs or more: the stretch tables are not built
    assert not sh.share_sweeps_as_stretches(shallow_reads, L, 256, rl, M)
    # with two ranks the deep contig goes alone, the two shallow ones together
    owned = sh.assign_contigs([shallow_reads, deep_reads, shallow_reads], 2, contig_lengths=[L, L, L],
                              read_length=rl, max_coverage=M)
    assert owned == [[1], [0, 2]]


def test_options_struct_matches_the_header(pkg, tmp_path):
    """qmcp_hip_options as the Python wrapper lays it out == as a C compiler lays out the header's; the defaults call
    fills in the size and nothing else"""
    import ctypes as C
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "qmcp_hip.h"\nint main(void){ printf("%zu %zu %zu\\n", sizeof(qmcp_hip_options), '
           'offsetof(qmcp_hip_options, near_uniform_min_depth), offsetof(qmcp_hip_options, host_both_columns)); return 0; }\n')
    exe = tmp_path / "opt_size"
    out = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", str(exe)],
                         input=src, text=True, capture_output=True)
    assert out.returncode == 0, out.stderr
    size, off_depth, off_last = map(int, subprocess.run([str(exe)], capture_output=True, text=True).stdout.split())
    assert size == C.sizeof(pkg.Options)
    assert off_depth == pkg.Options.near_uniform_min_depth.offset and off_last == pkg.Options.host_both_columns.offset
    o = pkg.Options()
    pkg._hip.qmcp_hip_default_options(C.byref(o))
    assert o.struct_size == size and all(getattr(o, name) == 0 for name, _ in pkg.Options._fields_[1:])
