"""Writes tests/golden/tiny_other_writer.bam: a small two-reference BAM produced by the pure-Python writer of
tests/bam_py.py (zlib BGZF blocks, struct-packed records) -- a different writer than the C++ one under test --
with the record kinds other tools emit: unmapped mates (flag 0x4, pos -1, no CIGAR), secondary (0x100) and
supplementary (0x800) alignments, every CIGAR operation, hard clips, aux tags, a read name that occurs once and one
that occurs three times, records that straddle BGZF block borders.
    python tests/golden/make_tiny_bam.py
The expected import (tiny_other_writer.expected.json) is what tests/bam_py.py's restatement of read_bam's pairing
rules (libs/bam-api/src/bam_api.cpp:420-476) makes of an independent parse of the file."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bam_py  # noqa: E402


def records():
    P = bam_py.pack_record
    aux = b"NMC\x02" + b"RGZgrp1\0"
    return [
        P("pairA", 0x63, 100, 60, [(150, "M")], 150, next_ref=0, next_pos=300, tlen=350, aux=aux),
        P("pairB", 0x93, 520, 42, [(5, "S"), (70, "M"), (3, "I"), (40, "M"), (2, "D"), (32, "M")], 150),
        P("lonely", 0x41, 900, 20, [(100, "M")], 100),                      # its mate never comes
        P("pairA", 0x93, 300, 59, [(20, "M"), (500, "N"), (130, "=")], 150),  # second mate: the pair is appended
        P("split", 0x841, 2000, 30, [(60, "H"), (90, "M")], 90),            # supplementary, first in pair
        P("pairB", 0x63, 400, 41, [(148, "M"), (2, "X")], 150),             # FIRST mate met second: swapped
        P("half", 0x49, 1500, 37, [(75, "M"), (1, "P"), (75, "M")], 150),   # mapped, mate unmapped
        P("half", 0x85, -1, 0, [], 150, ref_id=-1),                         # unmapped mate: pos -1, no CIGAR
        P("split", 0x81, 2600, 30, [(150, "M")], 150),
        P("tri", 0x41, 10, 11, [(50, "M")], 50),
        P("tri", 0x81, 70, 12, [(50, "M")], 50),
        P("tri", 0x181, 4000, 0, [(50, "M")], 50),                          # secondary: third record of the name
        P("other_ref", 0x41, 5, 50, [(30, "M")], 30, ref_id=1),
        P("other_ref", 0x81, 45, 50, [(30, "M")], 30, ref_id=1),
    ]


def main():
    path = os.path.join(HERE, "tiny_other_writer.bam")
    bam_py.write_bam(path, [("chrT", 5000), ("chrU", 800)], records(), block_payload=300)
    header, recs, ref_lengths = bam_py.parse(path)
    reads, filtered = bam_py.pair_like_the_reference(recs)
    expected = {
        "ref_lengths": ref_lengths, "records": len(recs),
        "bam_ids": [r["bam_id"] for r in reads],
        # (Index is size_t: the unmapped mate's pos -1 and end -2 wrap, as in the reference's Read::Read)
        "starts": [r["start"] % (1 << 64) for r in reads], "ends": [r["end"] % (1 << 64) for r in reads],
        "qualities": [r["q"] for r in reads], "seq_lengths": [r["l"] for r in reads],
        "is_first": [r["first"] for r in reads], "filtered_out": filtered,
    }
    with open(os.path.join(HERE, "tiny_other_writer.expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print(path, os.path.getsize(path), "bytes;", len(recs), "records ->", len(reads), "reads imported")
    # A second file of the same records the way other tools leave them: a header text of ~1.2 KB (@PG / @CO lines) cut
    # into BGZF members of 200 bytes -- the header and the reference list span seven members, an empty member sits in
    # the middle of the stream -- and NO end-of-file member at the end.
    path2 = os.path.join(HERE, "tiny_multi_member_header.bam")
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join(
        f"@PG\tID:step{k}\tPN:tool{k}\tVN:0.{k}\tCL:tool{k} --in a.bam --out b.bam --threads {k + 1}\n" for k in range(12)
    ) + "@CO\t" + "a comment line that is there to be long " * 6 + "\n"
    bam_py.write_bam(path2, [("chrT", 5000), ("chrU", 800)], records(), text=text, block_payload=200, eof=False,
                     empty_member_after=9)
    header2, recs2, ref_lengths2 = bam_py.parse(path2)
    reads2, filtered2 = bam_py.pair_like_the_reference(recs2)
    assert [r["bam_id"] for r in reads2] == expected["bam_ids"] and ref_lengths2 == ref_lengths
    with open(os.path.join(HERE, "tiny_multi_member_header.expected.json"), "w") as f:
        json.dump({"same_import_as": "tiny_other_writer.expected.json", "header_text_bytes": len(bam_py.header_text(header2)),
                   "members": sum(1 for _ in bam_py.members(path2)), "ends_with_eof_member": False}, f, indent=1)
    print(path2, os.path.getsize(path2), "bytes; header text", len(text), "bytes")


if __name__ == "__main__":
    main()
