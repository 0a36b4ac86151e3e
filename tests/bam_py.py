"""A second, independent reading of BAM files for the tests of the in-repo BAM reader / writer: Python's gzip
module inflates the BGZF members (it accepts concatenated gzip members and skips their extra fields), struct
unpacks the records -- nothing shared with genome-downsampler_amd/host/src/bam_io.cpp."""
import gzip
import struct

REF_CONSUMING = {0, 2, 3, 7, 8}  # M D N = X


def parse(path):
    """-> (header bytes up to the first record, list of records), a record = dict(raw, qname, flag, pos, mapq,
    l_seq, rlen)"""
    data = gzip.decompress(open(path, "rb").read())
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<I", data, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<I", data, o)
    o += 4
    ref_lengths = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<I", data, o)
        o += 4 + l_name
        ref_lengths.append(struct.unpack_from("<I", data, o)[0])
        o += 4
    header = data[:o]
    recs = []
    while o < len(data):
        block_size, = struct.unpack_from("<I", data, o)
        raw = data[o:o + 4 + block_size]
        ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq = struct.unpack_from("<iiBBHHHI", data, o + 4)
        qname = data[o + 36:o + 36 + l_name - 1].decode()
        cig = struct.unpack_from(f"<{n_cigar}I", data, o + 36 + l_name)
        rlen = sum(v >> 4 for v in cig if (v & 0xF) in REF_CONSUMING)
        recs.append(dict(raw=raw, qname=qname, flag=flag, pos=pos, mapq=mapq, l_seq=l_seq, rlen=rlen))
        o += 4 + block_size
    return header, recs, ref_lengths


def pair_like_the_reference(recs, min_len=0, min_mapq=0, inside=None):
    """read_bam's pairing (libs/bam-api/src/bam_api.cpp:420-476), restated: the pair is appended when its
    second mate is met; the qname map keeps the FIRST record of a name (and, after an accepted pair, whichever
    mate the swap left there); filters act per pair.  -> (list of reads as tuples, filtered-out ids)"""
    seen, out, accepted = {}, [], [False] * len(recs)
    for i, r in enumerate(recs):
        cur = dict(bam_id=i, start=r["pos"], end=r["pos"] + r["rlen"] - 1, q=r["mapq"], l=r["l_seq"],
                   first=bool(r["flag"] & 0x40))
        if r["qname"] in seen:
            r1, r2 = seen[r["qname"]], cur
            drop = not (r1["q"] >= min_mapq and r2["q"] >= min_mapq) or not (r1["l"] >= min_len and r2["l"] >= min_len)
            if inside is not None:
                drop = drop or not inside(r1, r2)
            if drop:
                continue
            if r2["first"]:
                r1, r2 = r2, r1
                seen[r["qname"]] = r1      # std::swap through the map reference
            out += [r1, r2]
            accepted[r1["bam_id"]] = accepted[r2["bam_id"]] = True
        else:
            seen[r["qname"]] = cur
    return out, [i for i, a in enumerate(accepted) if not a]


# ---------------------------------------------------------------- an independent WRITER (tests only)
# BGZF blocks by Python's zlib (raw deflate + the gzip member header with the "BC" extra field and the CRC32 /
# ISIZE trailer, SAM specification section 4.1), records packed with struct -- nothing shared with the C++
# writer in genome-downsampler_amd/host/src/bam_io.cpp, so the C++ reader is not only checked against its
# sibling.
import zlib

CIGAR_OPS = "MIDNSHP=X"
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(payload, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = co.compress(payload) + co.flush()
    bsize = 12 + 6 + len(body) + 8 - 1     # total block size - 1
    assert bsize < 65536
    head = struct.pack("<4BIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


def pack_record(qname, flag, pos, mapq, cigar, l_seq, ref_id=0, next_ref=-1, next_pos=-1, tlen=0, aux=b""):
    """one alignment record (block_size included); cigar = [(length, op char), ...]; sequence all 'A', no
    qualities (0xFF)"""
    name = qname.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for n, op in cigar)
    seq = bytes([0x11]) * ((l_seq + 1) // 2)
    qual = b"\xff" * l_seq
    body = struct.pack("<iiBBHHHIiii", ref_id, pos, len(name), mapq, 4680, len(cigar), flag, l_seq, next_ref,
                       next_pos, tlen) + name + cig + seq + qual + aux
    return struct.pack("<I", len(body)) + body


def write_bam(path, references, records, text="@HD\tVN:1.6\tSO:unsorted\n", block_payload=40_000):
    """references = [(name, length), ...]; records = packed records (pack_record).  Records straddle BGZF
    block borders (blocks are cut every `block_payload` bytes), as in files written by other tools."""
    text = text + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in references)
    data = b"BAM\x01" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(references))
    for n, l in references:
        data += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", l)
    data += b"".join(records)
    with open(path, "wb") as f:
        for o in range(0, len(data), block_payload):
            f.write(_bgzf_block(data[o:o + block_payload]))
        f.write(BGZF_EOF)
