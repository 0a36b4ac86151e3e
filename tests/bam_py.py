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


def members(path):
    """the BGZF members of a file: (offset, total size, uncompressed size) of each, from the BC subfield (BSIZE)"""
    data = open(path, "rb").read()
    o = 0
    while o < len(data):
        assert data[o:o + 4] == b"\x1f\x8b\x08\x04" and data[o + 12:o + 14] == b"BC"
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        yield o, bsize, struct.unpack_from("<I", data, o + bsize - 4)[0]
        o += bsize


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


def write_bam(path, references, records, text="@HD\tVN:1.6\tSO:unsorted\n", block_payload=40_000, eof=True,
              empty_member_after=None):
    """references = [(name, length), ...]; records = packed records (pack_record).  Records straddle BGZF
    block borders (blocks are cut every `block_payload` bytes), as in files written by other tools.  eof=False leaves
    the 28-byte end-of-file member out (a truncated-but-whole file: HTSlib warns and reads it); empty_member_after=k
    puts an empty member (legal anywhere in a BGZF stream) behind the k-th block."""
    text = text + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in references)
    data = b"BAM\x01" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(references))
    for n, l in references:
        data += struct.pack("<I", len(n) + 1) + n.encode() + b"\0" + struct.pack("<I", l)
    data += b"".join(records)
    with open(path, "wb") as f:
        for k, o in enumerate(range(0, len(data), block_payload)):
            f.write(_bgzf_block(data[o:o + block_payload]))
            if empty_member_after is not None and k == empty_member_after:
                f.write(BGZF_EOF)   # (the end-of-file marker IS an empty member)
        if eof:
            f.write(BGZF_EOF)


# ---------------------------------------------------------------- an independent SAM rendering (tests only)
def parse_references(header):
    """[(name, length)] from the header bytes `parse` returns"""
    l_text, = struct.unpack_from("<I", header, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<I", header, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<I", header, o)
        name = header[o + 4:o + 4 + l_name - 1].decode()
        refs.append((name, struct.unpack_from("<I", header, o + 4 + l_name)[0]))
        o += 8 + l_name
    return refs


def header_text(header):
    l_text, = struct.unpack_from("<I", header, 4)
    return header[8:8 + l_text].rstrip(b"\0").decode()


def to_sam(raw, refs):
    """one record (block_size included) as a SAM line without the newline (SAM specification sections 1.4, 4.2)"""
    ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen = struct.unpack_from("<iiBBHHHIiii", raw, 4)
    o = 36
    qname = raw[o:o + l_name - 1].decode() or "*"
    o += l_name
    cig = struct.unpack_from(f"<{n_cigar}I", raw, o)
    o += 4 * n_cigar
    cigar = "".join(f"{v >> 4}{CIGAR_OPS[v & 0xF]}" for v in cig) or "*"
    seq = "".join("=ACMGRSVTWYHKDBN"[(raw[o + i // 2] >> (0 if i % 2 else 4)) & 0xF] for i in range(l_seq)) or "*"
    o += (l_seq + 1) // 2
    q = raw[o:o + l_seq]
    qual = "*" if l_seq == 0 or q[0] == 0xFF else "".join(chr(b + 33) for b in q)
    o += l_seq
    name_of = lambda i: refs[i][0] if 0 <= i < len(refs) else "*"
    rnext = "*" if next_ref < 0 else ("=" if next_ref == ref_id else name_of(next_ref))
    fields = [qname, str(flag), name_of(ref_id), str(pos + 1), str(mapq), cigar, rnext, str(next_pos + 1), str(tlen), seq, qual]
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while o < len(raw):
        tag, typ = raw[o:o + 2].decode(), chr(raw[o + 2])
        o += 3
        if typ == "A":
            fields.append(f"{tag}:A:{chr(raw[o])}"); o += 1
        elif typ in "ZH":
            e = raw.index(b"\0", o)
            fields.append(f"{tag}:{typ}:{raw[o:e].decode()}"); o = e + 1
        elif typ == "B":
            sub = chr(raw[o]); count, = struct.unpack_from("<I", raw, o + 1); o += 5
            w = struct.calcsize(fmt[sub])
            vals = [struct.unpack_from(fmt[sub], raw, o + k * w)[0] for k in range(count)]
            o += count * w
            fields.append(f"{tag}:B:{sub}" + "".join("," + (f"{v:g}" if sub == "f" else str(v)) for v in vals))
        else:
            v, = struct.unpack_from(fmt[typ], raw, o); o += struct.calcsize(fmt[typ])
            fields.append(f"{tag}:f:{v:g}" if typ == "f" else f"{tag}:i:{v}")
    return "\t".join(fields)
