"""BAM ingest / emit of the host mirror: read_bam's qname-map pairing and
filters (libs/bam-api/src/bam_api.cpp:359-507), end = pos + CIGAR reference length - 1 (read.cpp:11-13),
write_bam's second pass (bam_api.cpp:534-656).  The reference holds no BAM fixture and HTSlib is absent here,
so the in-repo reader / writer (zlib) are checked against an independent Python reading of the same files
(tests/bam_py.py) and against a Python restatement of the pairing rules: parity unpinned, self-consistent."""
import importlib

import numpy as np
import pytest

import bam_py


def _synthetic(tmp_path, n_pairs=4000, L=20_000, seed=3, triples=True):
    pkg = importlib.import_module("genome-downsampler_amd")
    rng = np.random.default_rng(seed)
    n = 2 * n_pairs
    names = np.repeat(np.arange(n_pairs), 2)
    first = np.tile([True, False], n_pairs)
    flags = np.where(first, 0x41, 0x81).astype(np.uint16)          # paired + first / last in pair
    match = rng.integers(60, 151, size=n)
    clip = np.where(rng.random(n) < 0.15, rng.integers(1, 30, size=n), 0)
    dele = np.where(rng.random(n) < 0.10, rng.integers(1, 12, size=n), 0)
    match2 = np.where(dele > 0, rng.integers(5, 40, size=n), 0)
    pos = rng.integers(0, L - 250, size=n)
    mapq = rng.integers(0, 61, size=n)
    order = rng.permutation(n)                                      # mates are NOT adjacent in the file
    # quirks: some second mates come first in the file, a few names occur once or three times
    cols = [a[order] for a in (names, flags, pos, mapq, clip, match, dele, match2)]
    extra = 40
    # (a name that occurs three times imports its first record twice; write_bam, given an id twice, stops
    # matching after it -- bam_api.cpp:605-616 -- so the file-to-file test leaves such names out)
    again = rng.integers(0, n_pairs, size=extra // 2) if triples else n_pairs + extra + np.arange(extra // 2)
    cols[0] = np.concatenate([cols[0], again, n_pairs + np.arange(extra // 2)])
    for k in range(1, 8):
        cols[k] = np.concatenate([cols[k], cols[k][:extra]])
    path = tmp_path / "in.bam"
    pkg.write_synthetic_bam(path, L, *cols)
    return pkg, path, L


def test_reader_matches_an_independent_parse_and_the_pairing_rules(tmp_path):
    pkg, path, L = _synthetic(tmp_path)
    header, recs, ref_lengths = bam_py.parse(path)
    assert ref_lengths == [L] and len(recs) == 8040
    got = pkg.read_bam(path)
    want, filtered = bam_py.pair_like_the_reference(recs)
    assert got["ref_genome_length"] == L
    assert got["bam_ids"].tolist() == [r["bam_id"] for r in want]
    assert got["starts"].tolist() == [r["start"] for r in want]
    assert got["ends"].tolist() == [r["end"] for r in want]           # pos + M/D/N/=/X lengths - 1
    assert got["qualities"].tolist() == [r["q"] for r in want]
    assert got["seq_lengths"].tolist() == [r["l"] for r in want]
    assert got["is_first"].tolist() == [r["first"] for r in want]
    assert got["filtered_out"].tolist() == filtered and len(filtered) > 0
    # bam_api.cpp:481 asserts imported + filtered == records; a name that occurs three times pairs its first
    # record twice (the map entry is never erased), so the identity holds on distinct ids only
    assert len({r["bam_id"] for r in want}) + len(filtered) == len(recs)
    # the reference's defaults -l 90 -q 30 (src/app.hpp:22-25)
    got = pkg.read_bam(path, min_length=90, min_mapq=30)
    want, filtered = bam_py.pair_like_the_reference(recs, min_len=90, min_mapq=30)
    assert got["bam_ids"].tolist() == [r["bam_id"] for r in want] and got["filtered_out"].tolist() == filtered


def test_amplicon_filter_and_grade_while_ingesting(tmp_path):
    pkg, path, L = _synthetic(tmp_path, n_pairs=1500, seed=9)
    bed = tmp_path / "p.bed"
    bed.write_text("".join(f"ref1\t{a}\t{a + 20}\tA{k}_LEFT\nref1\t{a + 900}\t{a + 920}\tA{k}_RIGHT\n"
                           for k, a in enumerate(range(0, L - 1000, 700))))
    tsv = tmp_path / "p.tsv"
    tsv.write_text("".join(f"A{k}_LEFT\tA{k}_RIGHT\n" for k, _ in enumerate(range(0, L - 1000, 700))))
    a0, a1 = pkg.amplicons_from_files(bed, tsv)
    _, recs, _ = bam_py.parse(path)
    inside = lambda r1, r2: any(lo <= r1["start"] and r1["end"] <= hi and lo <= r2["start"] and r2["end"] <= hi
                                for lo, hi in zip(a0.tolist(), a1.tolist()))
    got = pkg.read_bam(path, bed=bed, tsv=tsv, amplicon_mode=1)       # FILTER (bam_api.cpp:311-319)
    want, filtered = bam_py.pair_like_the_reference(recs, inside=inside)
    assert got["bam_ids"].tolist() == [r["bam_id"] for r in want] and got["filtered_out"].tolist() == filtered
    assert 0 < len(want) < len(recs) - 100
    # GRADE (bam_api.cpp:334-357): quality - min + (max - min if the pair sits in one amplicon)
    got = pkg.read_bam(path, bed=bed, tsv=tsv, amplicon_mode=2)
    want, _ = bam_py.pair_like_the_reference(recs)
    qs = [r["q"] for r in want]
    lo, hi = min(qs), max(qs)
    grades = [q - lo + ((hi - lo) if inside(want[i - i % 2], want[i - i % 2 + 1]) else 0) for i, q in enumerate(qs)]
    assert got["qualities"].tolist() == grades


def test_bgzf_layer_of_the_writer_reads_back_with_pythons_gzip(tmp_path):
    pkg, path, L = _synthetic(tmp_path, n_pairs=2500, seed=4)
    header, recs, _ = bam_py.parse(path)
    assert header.startswith(b"BAM\x01") and b"@SQ\tSN:ref1" in header
    # round trip of the BGZF layer: the file the in-repo writer produced inflates (by Python's gzip) to
    # records that re-parse to the inputs -- and ends with the 28-byte BGZF end-of-file block
    raw = open(path, "rb").read()
    assert raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert sum(len(r["raw"]) for r in recs) + len(header) == len(__import__("gzip").decompress(raw))


@pytest.mark.gpu
def test_file_to_file_downsampling(tmp_path):
    """App::execute's flow on files (src/app.cpp:113-151): BamApi(in.bam) -> quasi-mcp-hip -> find_pairs ->
    write_paired_reads(out.bam); the output holds exactly the records of the oracle's kept pairs, in file
    order, and the header, byte for byte"""
    import oracle_py
    pkg, path, L = _synthetic(tmp_path, n_pairs=6000, L=5_000, seed=12, triples=False)
    header, recs, _ = bam_py.parse(path)
    reads = pkg.read_bam(path)
    M = 40
    out, filt = tmp_path / "out.bam", tmp_path / "filtered.bam"
    written = pkg.downsample_bam("quasi-mcp-hip", path, out, M, filtered_path=filt)
    mask = oracle_py.find_pairs(oracle_py.solve(reads["starts"], reads["ends"], L, M), reads["starts"].size)
    kept_ids = np.sort(reads["bam_ids"][pkg.mask_to_indices(mask, reads["starts"].size).astype(np.int64)])
    oh, orecs, _ = bam_py.parse(out)
    assert written == kept_ids.size == len(orecs) and oh == header
    assert [r["raw"] for r in orecs] == [recs[i]["raw"] for i in kept_ids.tolist()]
    _, frecs, _ = bam_py.parse(filt)
    assert [r["raw"] for r in frecs] == [recs[i]["raw"] for i in reads["filtered_out"].tolist()]


# ---------------------------------------------------------------- files from ANOTHER writer, and corrupt ones
import json
import os
import struct

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reader_on_a_file_written_by_another_writer():
    """tests/golden/tiny_other_writer.bam comes from the pure-Python writer (make_tiny_bam.py): two references,
    unmapped / secondary / supplementary records, every CIGAR operation, aux tags, records across BGZF block
    borders.  read_bam treats flags as the reference does -- it pairs by name only (bam_api.cpp:428-461)."""
    pkg = importlib.import_module("genome-downsampler_amd")
    path = os.path.join(GOLDEN, "tiny_other_writer.bam")
    want = json.load(open(os.path.join(GOLDEN, "tiny_other_writer.expected.json")))
    ok, n, msg = pkg.check_bam(path)
    assert ok and n == len(want["bam_ids"]), msg
    got = pkg.read_bam(path)
    assert got["ref_genome_length"] == want["ref_lengths"][0]          # bam_api.cpp:422: target_len[0]
    assert got["bam_ids"].tolist() == want["bam_ids"]
    assert got["starts"].tolist() == [v % (1 << 32) for v in want["starts"]]   # (the binding narrows Index to 32 bits)
    assert got["ends"].tolist() == [v % (1 << 32) for v in want["ends"]]
    assert got["qualities"].tolist() == want["qualities"]
    assert got["seq_lengths"].tolist() == want["seq_lengths"]
    assert got["is_first"].tolist() == want["is_first"]
    assert got["filtered_out"].tolist() == want["filtered_out"]
    # and the fixture is what the committed script makes of the in-test parse today
    header, recs, ref_lengths = bam_py.parse(path)
    reads, filtered = bam_py.pair_like_the_reference(recs)
    assert [r["bam_id"] for r in reads] == want["bam_ids"] and filtered == want["filtered_out"]


def test_reader_on_a_header_over_several_members_without_an_eof_member():
    """tests/golden/tiny_multi_member_header.bam (make_tiny_bam.py): the same records behind a 1.2 KB header text cut
    into 200-byte BGZF members -- text and reference list span seven of them --, an empty member in the middle of the
    stream, and NO end-of-file member (a truncated-but-whole file; HTSlib warns and reads it): the same import"""
    pkg = importlib.import_module("genome-downsampler_amd")
    path = os.path.join(GOLDEN, "tiny_multi_member_header.bam")
    meta = json.load(open(os.path.join(GOLDEN, "tiny_multi_member_header.expected.json")))
    want = json.load(open(os.path.join(GOLDEN, meta["same_import_as"])))
    mem = list(bam_py.members(path))
    assert len(mem) == meta["members"] and mem[-1][2] != 0 and any(u == 0 for _, _, u in mem[:-1])
    assert open(path, "rb").read()[-28:] != bam_py.BGZF_EOF
    header, recs, ref_lengths = bam_py.parse(path)
    assert len(bam_py.header_text(header)) == meta["header_text_bytes"] > 5 * 200
    ok, n, msg = pkg.check_bam(path)
    assert ok and n == len(want["bam_ids"]), msg
    got = pkg.read_bam(path)
    assert got["ref_genome_length"] == want["ref_lengths"][0]
    assert got["bam_ids"].tolist() == want["bam_ids"]
    assert got["starts"].tolist() == [v % (1 << 32) for v in want["starts"]]
    assert got["ends"].tolist() == [v % (1 << 32) for v in want["ends"]]
    assert got["qualities"].tolist() == want["qualities"] and got["is_first"].tolist() == want["is_first"]
    assert got["filtered_out"].tolist() == want["filtered_out"]


def test_python_written_file_with_many_records_equals_the_pairing_rules(tmp_path):
    """2 000 pairs in random order with soft clips, deletions, skips and unmapped mates, blocks cut every 5 000
    bytes: C++ reader == independent parse + restated pairing, with and without the -l / -q filters"""
    pkg = importlib.import_module("genome-downsampler_amd")
    rng = np.random.default_rng(17)
    recs = []
    for q in range(2000):
        for first in (True, False):
            unmapped = rng.random() < 0.03
            flag = (0x41 if first else 0x81) | (0x4 if unmapped else 0) | (0x100 if rng.random() < 0.02 else 0)
            if unmapped:
                recs.append(bam_py.pack_record(f"r{q}", flag, -1, 0, [], 100, ref_id=-1))
                continue
            m1, dl, m2 = int(rng.integers(30, 120)), int(rng.integers(0, 9)), int(rng.integers(1, 40))
            cigar = ([(int(rng.integers(1, 20)), "S")] if rng.random() < 0.2 else []) + [(m1, "M")]
            if dl:
                cigar += [(dl, "D" if rng.random() < 0.7 else "N"), (m2, "M")]
            l_seq = sum(n for n, op in cigar if op in "MIS=X")
            recs.append(bam_py.pack_record(f"r{q}", flag, int(rng.integers(0, 29_000)), int(rng.integers(0, 61)),
                                           cigar, l_seq))
    order = rng.permutation(len(recs))
    path = tmp_path / "py.bam"
    bam_py.write_bam(path, [("ref", 30_000)], [recs[i] for i in order], block_payload=5_000)
    header, parsed, _ = bam_py.parse(path)
    for kw, (min_len, min_mapq) in (({}, (0, 0)), (dict(min_length=90, min_mapq=30), (90, 30))):
        got = pkg.read_bam(path, **kw)
        want, filtered = bam_py.pair_like_the_reference(parsed, min_len, min_mapq)
        assert got["bam_ids"].tolist() == [r["bam_id"] for r in want]
        assert got["starts"].tolist() == [r["start"] % (1 << 32) for r in want]
        assert got["ends"].tolist() == [r["end"] % (1 << 32) for r in want]
        assert got["filtered_out"].tolist() == filtered


def _corrupt(tmp_path, name, mutate):
    """a valid two-record file whose uncompressed bytes are changed by mutate(bytearray) before compression"""
    import gzip
    good = tmp_path / "good.bam"
    recs = [bam_py.pack_record("a", 0x41, 10, 30, [(50, "M")], 50), bam_py.pack_record("a", 0x81, 90, 30, [(50, "M")], 50)]
    bam_py.write_bam(good, [("ref", 1000)], recs)
    data = bytearray(gzip.decompress(open(good, "rb").read()))
    data = mutate(data)
    path = tmp_path / name
    with open(path, "wb") as f:
        for o in range(0, len(data), 40_000):
            f.write(bam_py._bgzf_block(bytes(data[o:o + 40_000])))
        f.write(bam_py.BGZF_EOF)
    return path


def test_corrupt_length_fields_are_rejected_not_wrapped(tmp_path):
    """block_size 0xFFFFFFFE (4 + block_size wrapped to 2 in 32-bit arithmetic: a heap overflow), absurd l_text and
    l_name, a record cut short: each is an error message, none an allocation of its claimed size"""
    pkg = importlib.import_module("genome-downsampler_amd")
    ok, n, _ = pkg.check_bam(_corrupt(tmp_path, "fine.bam", lambda d: d))
    assert ok and n == 2

    def header_end(d):
        l_text, = struct.unpack_from("<I", d, 4)
        o = 8 + l_text + 4
        l_name, = struct.unpack_from("<I", d, o)
        return o + 4 + l_name + 4

    def huge_block(d):
        struct.pack_into("<I", d, header_end(d), 0xFFFFFFFE)
        return d

    def huge_text(d):
        struct.pack_into("<I", d, 4, 0xFFFFFFF0)
        return d

    def huge_name(d):
        l_text, = struct.unpack_from("<I", d, 4)
        struct.pack_into("<I", d, 8 + l_text + 4, 0xFFFFFFFC)
        return d

    def cut_short(d):
        return d[:len(d) - 20]

    for name, mut, word in (("block.bam", huge_block, "record length"), ("text.bam", huge_text, "header text"),
                            ("name.bam", huge_name, "reference name"), ("cut.bam", cut_short, "truncated")):
        ok, n, msg = pkg.check_bam(_corrupt(tmp_path, name, mut))
        assert not ok and word in msg, (name, msg)


def test_inflating_on_several_threads_reads_the_same_file(tmp_path, monkeypatch):
    """BGZF members inflate on their own: read in batches of ~8 MiB of compressed bytes and inflated on up to 8
    threads (QMCP_BAM_THREADS), as the reference hands the file to HTSlib's thread pool (bam_api.cpp:386-397).  A
    file of several batches read with one thread and with eight: the same reads, and the independent parse's."""
    pkg = importlib.import_module("genome-downsampler_amd")
    rng = np.random.default_rng(23)
    n_pairs = 200_000
    n = 2 * n_pairs
    names = np.repeat(np.arange(n_pairs), 2)[rng.permutation(n)]
    flags = np.where(rng.random(n) < 0.5, 0x41, 0x81).astype(np.uint16)
    pos = rng.integers(0, 2_000_000, size=n)
    mapq = rng.integers(0, 61, size=n)
    clip = rng.integers(0, 30, size=n)
    match = rng.integers(50, 151, size=n)
    dele = np.where(rng.random(n) < 0.2, rng.integers(1, 9, size=n), 0)
    match2 = np.where(dele > 0, rng.integers(5, 40, size=n), 0)
    path = tmp_path / "big.bam"
    pkg.write_synthetic_bam(path, 2_100_000, names, flags, pos, mapq, clip, match, dele, match2)
    assert path.stat().st_size > 9 * (1 << 20)      # more than one batch of compressed bytes
    monkeypatch.setenv("QMCP_BAM_THREADS", "1")
    one = pkg.read_bam(path)
    monkeypatch.setenv("QMCP_BAM_THREADS", "8")
    many = pkg.read_bam(path)
    for k in ("bam_ids", "starts", "ends", "qualities", "seq_lengths", "is_first", "filtered_out"):
        assert np.array_equal(one[k], many[k]), k
    header, recs, _ = bam_py.parse(path)
    want, filtered = bam_py.pair_like_the_reference(recs)
    assert many["bam_ids"].tolist() == [r["bam_id"] for r in want] and many["filtered_out"].tolist() == filtered


@pytest.mark.gpu
def test_file_with_soft_clipped_reads_takes_the_near_uniform_route(tmp_path):
    """a BAM as aligners write them -- one read length, a few per cent of the records soft-clipped (their reference
    span is shorter: read.cpp:11-13) -- downsampled file to file: the output holds the oracle's kept pairs, and the
    solve behind it ran on the near-uniform route (kernels/near_uniform.inc.hip), not the mixed-span walk"""
    import oracle_py
    pkg = importlib.import_module("genome-downsampler_amd")
    rng = np.random.default_rng(31)
    n_pairs, L, M = 100_000, 20_000, 100
    n = 2 * n_pairs
    names = np.repeat(np.arange(n_pairs), 2)
    flags = np.where(np.tile([True, False], n_pairs), 0x41, 0x81).astype(np.uint16)
    clip = np.where(rng.random(n) < 0.02, rng.integers(1, 40, size=n), 0)
    match = 150 - clip                                   # 150-base reads: <clip>S<150 - clip>M
    zeros = np.zeros(n, np.int64)
    pos = rng.integers(0, L - 150, size=n)
    mapq = rng.integers(20, 61, size=n)
    order = rng.permutation(n)
    cols = [a[order] for a in (names, flags, pos, mapq, clip, match, zeros, zeros)]
    path = tmp_path / "clipped.bam"
    pkg.write_synthetic_bam(path, L, *cols)
    header, recs, _ = bam_py.parse(path)
    reads = pkg.read_bam(path)
    assert reads["starts"].size == n and int((reads["ends"] - reads["starts"] + 1 != 150).sum()) == int((clip > 0).sum())
    out = tmp_path / "out.bam"
    written = pkg.downsample_bam("quasi-mcp-hip", path, out, M)
    keep = oracle_py.solve(reads["starts"], reads["ends"], L, M)
    mask = oracle_py.find_pairs(keep, n)
    kept_ids = np.sort(reads["bam_ids"][pkg.mask_to_indices(mask, n).astype(np.int64)])
    oh, orecs, _ = bam_py.parse(out)
    assert written == kept_ids.size == len(orecs) and oh == header
    assert [r["raw"] for r in orecs] == [recs[i]["raw"] for i in kept_ids.tolist()]
    with pkg.Solver(0) as sv:                            # the same columns through the C ABI: which route, which stats
        got = sv.solve(reads["starts"], reads["ends"], L, M)
        st = sv.last_stats
    assert np.array_equal(got, keep)
    assert st.path == pkg.PATH_NEAR_UNIFORM and st.near_uniform_exceptions == int((clip > 0).sum()), st.as_dict()


# ---------------------------------------------------------------- SAM text output, deflate on several threads
def test_sam_text_when_the_extension_is_not_bam(tmp_path):
    """BamApi::write_bam opens its output with "wb" for ".bam" and with "w" -- SAM text -- for anything else
    (libs/bam-api/src/bam_api.cpp:564).  The committed two-reference BAM of another writer (unmapped mates, every CIGAR
    operation, hard clips, aux tags of several types) copied to .sam: header text, then one line per chosen record, equal
    to an independent rendering (tests/bam_py.py: to_sam) of an independent parse"""
    pkg = importlib.import_module("genome-downsampler_amd")
    src = os.path.join(GOLDEN, "tiny_other_writer.bam")
    header, recs, _ = bam_py.parse(src)
    refs = bam_py.parse_references(header)
    ids = [0, 1, 3, 5, 7, 11, 12, 13]
    out = tmp_path / "picked.sam"
    assert pkg.copy_records(src, out, ids[::-1]) == len(ids)          # (the ids are sorted by the writer, as the reference does)
    text = open(out).read()
    want_header = bam_py.header_text(header)
    assert text.startswith(want_header) and "@SQ\tSN:chrT\tLN:5000" in want_header
    lines = text[len(want_header):].splitlines()
    assert lines == [bam_py.to_sam(recs[i]["raw"], refs) for i in ids]
    assert lines[0].split("\t")[11:] == ["NM:i:2", "RG:Z:grp1"] and lines[4].split("\t")[2:6] == ["*", "0", "0", "*"]
    # ... and the same records as BAM: byte-identical records, header copied
    outb = tmp_path / "picked.bam"
    assert pkg.copy_records(src, outb, ids) == len(ids)
    oh, orecs, _ = bam_py.parse(outb)
    assert oh == header and [r["raw"] for r in orecs] == [recs[i]["raw"] for i in ids]


def test_sam_lines_with_every_aux_type(tmp_path):
    pkg = importlib.import_module("genome-downsampler_amd")
    aux = (b"XAAq" + b"Xcc\xfe" + b"XCC\xfe" + b"Xss" + struct.pack("<h", -300) + b"XSS" + struct.pack("<H", 65000) +
           b"Xii" + struct.pack("<i", -70000) + b"XII" + struct.pack("<I", 4_000_000_000) + b"Xff" + struct.pack("<f", 1.5) +
           b"XZZtext with spaces\0" + b"XHH1AE301\0" + b"XBBs" + struct.pack("<I3h", 3, -1, 2, 300) +
           b"XGBf" + struct.pack("<I2f", 2, 0.25, -8.0) + b"XEBC" + struct.pack("<I", 0))
    recs = [bam_py.pack_record("r1", 0x63, 10, 60, [(4, "S"), (20, "M"), (2, "I"), (24, "M")], 50, next_ref=0, next_pos=200, tlen=240, aux=aux),
            bam_py.pack_record("r1", 0x93, 200, 60, [(50, "=")], 50, next_ref=0, next_pos=10, tlen=-240)]
    path = tmp_path / "aux.bam"
    bam_py.write_bam(path, [("chrA", 1000)], recs)
    out = tmp_path / "aux.txt"                                        # any extension but .bam
    assert pkg.copy_records(path, out, [0, 1]) == 2
    header, parsed, _ = bam_py.parse(path)
    lines = open(out).read()[len(bam_py.header_text(header)):].splitlines()
    assert lines == [bam_py.to_sam(r["raw"], [("chrA", 1000)]) for r in parsed]
    assert "Xc:i:-2" in lines[0] and "XC:i:254" in lines[0] and "XI:i:4000000000" in lines[0] and "XB:B:s,-1,2,300" in lines[0]
    assert lines[0].split("\t")[6:9] == ["=", "201", "240"] and lines[0].split("\t")[9] == "A" * 50


def test_deflating_on_several_threads_writes_the_same_file(tmp_path, monkeypatch):
    """BGZF blocks deflate on their own: batches of 64 blocks on up to 8 threads (QMCP_BAM_THREADS), written in order --
    the reference hands its output file to HTSlib's thread pool (bam_api.cpp:569-586).  A file of several batches written
    with one thread and with eight: the same bytes, and Python's gzip reads back the chosen records"""
    pkg = importlib.import_module("genome-downsampler_amd")
    rng = np.random.default_rng(5)
    n_pairs = 120_000
    n = 2 * n_pairs
    names = np.repeat(np.arange(n_pairs), 2)[rng.permutation(n)]
    flags = np.where(rng.random(n) < 0.5, 0x41, 0x81).astype(np.uint16)
    src = tmp_path / "src.bam"
    pkg.write_synthetic_bam(src, 1_000_000, names, flags, rng.integers(0, 900_000, size=n), rng.integers(0, 61, size=n),
                            np.zeros(n, np.uint32), rng.integers(50, 151, size=n), np.zeros(n, np.uint32), np.zeros(n, np.uint32))
    ids = np.flatnonzero(rng.random(n) < 0.8)
    monkeypatch.setenv("QMCP_BAM_THREADS", "1")
    assert pkg.copy_records(src, tmp_path / "one.bam", ids) == ids.size
    monkeypatch.setenv("QMCP_BAM_THREADS", "8")
    assert pkg.copy_records(src, tmp_path / "eight.bam", ids) == ids.size
    one, eight = open(tmp_path / "one.bam", "rb").read(), open(tmp_path / "eight.bam", "rb").read()
    assert one == eight
    _, recs, _ = bam_py.parse(src)
    _, got, _ = bam_py.parse(tmp_path / "eight.bam")
    assert [r["raw"] for r in got] == [recs[i]["raw"] for i in ids.tolist()]
    assert sum(len(r["raw"]) for r in got) > 3 * 64 * 0xFF00      # several batches of 64 blocks


@pytest.mark.gpu
def test_file_to_file_downsampling_to_sam_text(tmp_path):
    """App::execute's flow on files with `-o out.sam`: the kept pairs' records as SAM lines, in file order"""
    import oracle_py
    pkg, path, L = _synthetic(tmp_path, n_pairs=5000, L=4_000, seed=21, triples=False)
    header, recs, _ = bam_py.parse(path)
    reads = pkg.read_bam(path)
    M = 30
    out = tmp_path / "out.sam"
    written = pkg.downsample_bam("quasi-mcp-hip", path, out, M)
    mask = oracle_py.find_pairs(oracle_py.solve(reads["starts"], reads["ends"], L, M), reads["starts"].size)
    kept_ids = np.sort(reads["bam_ids"][pkg.mask_to_indices(mask, reads["starts"].size).astype(np.int64)])
    text = open(out).read()
    want_header = bam_py.header_text(header)
    assert text.startswith(want_header)
    refs = bam_py.parse_references(header)
    assert written == kept_ids.size
    assert text[len(want_header):].splitlines() == [bam_py.to_sam(recs[i]["raw"], refs) for i in kept_ids.tolist()]
