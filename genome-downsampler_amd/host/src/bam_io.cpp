// BAM ingest / emit on zlib alone (see include/bam-api/bam_io.hpp for the reference lines mirrored).
#include "bam-api/bam_io.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <map>
#include <thread>

namespace bam_api {
namespace {

constexpr std::size_t kBgzfMaxBlock = 0x10000;   // a BGZF block inflates to at most 64 KiB
constexpr std::size_t kBgzfFill = 0xFF00;        // payload per written block (htslib's choice)

std::uint16_t le16(const unsigned char* p) { return (std::uint16_t)(p[0] | (p[1] << 8)); }
std::uint32_t le32(const unsigned char* p) {
    return (std::uint32_t)p[0] | ((std::uint32_t)p[1] << 8) | ((std::uint32_t)p[2] << 16) | ((std::uint32_t)p[3] << 24);
}
void put16(std::vector<unsigned char>& v, std::uint16_t x) { v.push_back((unsigned char)(x & 0xFF)); v.push_back((unsigned char)(x >> 8)); }
void put32(std::vector<unsigned char>& v, std::uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((unsigned char)((x >> (8 * i)) & 0xFF)); }

bool set_err(std::string* err, const std::string& msg) {
    if (err) *err = msg;
    return false;
}

// threads for BGZF blocks (inflate on ingest, deflate on emit): QMCP_BAM_THREADS, default up to 8 of the host's cores
// (the reference hands both files one HTSlib thread pool: bam_api.cpp:386-397, 569-586)
unsigned bgzf_threads() {
    unsigned t = 8;
    if (const char* e = std::getenv("QMCP_BAM_THREADS")) t = (unsigned)std::strtoul(e, nullptr, 10);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw != 0 && t > hw) t = hw;
    return t < 1 ? 1 : t;
}

// Sequential reader of the uncompressed stream of a BGZF file.  The file is a sequence of gzip members whose extra
// subfield "BC" gives the member's size (SAM specification 4.1); every member inflates on its own, so the calling
// thread reads a batch of members from the file (about 8 MiB of compressed bytes) and several threads inflate it,
// one member per task, CRC and length checked -- what HTSlib's thread pool does for the reference
// (bam_api.cpp:386-397: hts_tpool_init, hts_set_opt(HTS_OPT_THREAD_POOL)).  QMCP_BAM_THREADS sets the number of
// inflating threads (default: up to 8 of the host's cores; 1: the calling thread alone).
class BgzfReader {
   public:
    ~BgzfReader() { if (f_) std::fclose(f_); }
    bool open(const std::filesystem::path& p) {
        f_ = std::fopen(p.c_str(), "rb");
        threads_ = bgzf_threads();
        return f_ != nullptr;
    }
    // reads exactly n bytes; false at a clean end of file before the first byte (eof() tells) or on error
    bool read(void* dst, std::size_t n) {
        unsigned char* out = static_cast<unsigned char*>(dst);
        while (n != 0) {
            if (pos_ == block_.size() && !next_batch()) return false;
            const std::size_t take = std::min(n, block_.size() - pos_);
            std::memcpy(out, block_.data() + pos_, take);
            pos_ += take; out += take; n -= take;
        }
        return true;
    }
    bool eof() const { return eof_; }
    const std::string& error() const { return error_; }

   private:
    struct Member { std::size_t in_off, in_len, out_off, out_len; std::uint32_t crc; };

    bool next_batch() {
        constexpr std::size_t kBatchBytes = std::size_t(8) << 20;  // compressed bytes per batch
        block_.clear();
        pos_ = 0;
        if (!error_.empty() || eof_) return false;
        comp_.clear();
        members_.clear();
        std::size_t out_total = 0;
        while (comp_.size() < kBatchBytes) {
            unsigned char hdr[12];
            const std::size_t got = std::fread(hdr, 1, sizeof(hdr), f_);
            if (got == 0) { at_end_ = true; break; }
            if (got != sizeof(hdr) || hdr[0] != 0x1F || hdr[1] != 0x8B || hdr[2] != 8 || !(hdr[3] & 4)) {
                error_ = "not a BGZF block (gzip header without an extra field)";
                return false;
            }
            const std::size_t xlen = le16(hdr + 10);
            extra_.resize(xlen);
            if (std::fread(extra_.data(), 1, xlen, f_) != xlen) { error_ = "truncated BGZF header"; return false; }
            std::size_t bsize = 0;
            for (std::size_t o = 0; o + 4 <= extra_.size();) {
                const std::size_t slen = le16(extra_.data() + o + 2);
                if (extra_[o] == 'B' && extra_[o + 1] == 'C' && slen == 2 && o + 6 <= extra_.size()) bsize = (std::size_t)le16(extra_.data() + o + 4) + 1u;
                o += 4u + slen;
            }
            if (bsize < 12u + xlen + 8u) { error_ = "BGZF block without a BC size field"; return false; }
            const std::size_t payload = bsize - 12u - xlen - 8u;
            const std::size_t off = comp_.size();
            comp_.resize(off + payload + 8);
            if (std::fread(comp_.data() + off, 1, payload + 8, f_) != payload + 8) { error_ = "truncated BGZF block"; return false; }
            const std::uint32_t crc = le32(comp_.data() + off + payload);
            const std::size_t isize = le32(comp_.data() + off + payload + 4);
            if (isize > kBgzfMaxBlock) { error_ = "BGZF block larger than 64 KiB"; return false; }
            if (isize != 0) {  // (empty blocks -- the end-of-file marker is one -- are skipped)
                members_.push_back(Member{off, payload, out_total, isize, crc});
                out_total += isize;
            }
        }
        if (members_.empty()) {
            if (at_end_) eof_ = true;
            if (!at_end_) return next_batch();  // (a batch of empty members only: go on)
            return false;
        }
        block_.resize(out_total);
        std::atomic<std::size_t> next{0};
        std::atomic<int> failed{0};
        auto work = [&]() {
            for (std::size_t i = next.fetch_add(1); i < members_.size() && failed.load() == 0; i = next.fetch_add(1)) {
                const Member& m = members_[i];
                z_stream zs;
                std::memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) { failed.store(1); return; }
                zs.next_in = comp_.data() + m.in_off; zs.avail_in = (uInt)m.in_len;
                zs.next_out = block_.data() + m.out_off; zs.avail_out = (uInt)m.out_len;
                const int rc = inflate(&zs, Z_FINISH);
                const bool sized = rc == Z_STREAM_END && zs.avail_out == 0;
                inflateEnd(&zs);
                if (!sized) { failed.store(2); return; }
                if (crc32(crc32(0L, Z_NULL, 0), block_.data() + m.out_off, (uInt)m.out_len) != m.crc) { failed.store(3); return; }
            }
        };
        const unsigned t = (unsigned)std::min<std::size_t>(threads_, members_.size());
        if (t <= 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            pool.reserve(t - 1);
            for (unsigned k = 1; k < t; ++k) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
        }
        if (failed.load() != 0) {
            error_ = failed.load() == 1 ? "inflateInit2 failed"
                   : failed.load() == 2 ? "BGZF block does not inflate to its stated size" : "BGZF block checksum mismatch";
            block_.clear();
            return false;
        }
        return true;
    }
    std::FILE* f_ = nullptr;
    unsigned threads_ = 1;
    std::vector<unsigned char> comp_, block_, extra_;
    std::vector<Member> members_;
    std::size_t pos_ = 0;
    bool eof_ = false, at_end_ = false;
    std::string error_;
};

// Writer of a BGZF file: the payload is cut into blocks of kBgzfFill bytes, every block deflates on its own, so a batch
// of blocks (4 MiB of payload) is deflated by several threads, one block per task, and written out in order -- what
// HTSlib's thread pool does for the reference's output file (bam_api.cpp:569-586, hts_set_thread_pool(outfile)).
class BgzfWriter {
   public:
    ~BgzfWriter() { if (f_) std::fclose(f_); }
    bool open(const std::filesystem::path& p) {
        f_ = std::fopen(p.c_str(), "wb");
        threads_ = bgzf_threads();
        return f_ != nullptr;
    }
    bool write(const void* src, std::size_t n) {
        const unsigned char* in = static_cast<const unsigned char*>(src);
        buf_.insert(buf_.end(), in, in + n);
        while (buf_.size() >= kBatchBlocks * kBgzfFill)
            if (!flush_blocks(kBatchBlocks)) return false;
        return true;
    }
    bool close() {
        if (!f_) return false;
        bool ok = flush_blocks((buf_.size() + kBgzfFill - 1) / kBgzfFill);
        static const unsigned char eof_block[28] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0,
                                                    0x1B, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // the empty block that ends a BGZF file
        ok = ok && std::fwrite(eof_block, 1, sizeof(eof_block), f_) == sizeof(eof_block);
        ok = (std::fclose(f_) == 0) && ok;
        f_ = nullptr;
        return ok;
    }

   private:
    static constexpr std::size_t kBatchBlocks = 64;
    // one block: gzip member header with the BC field, raw deflate, CRC32 and ISIZE
    static bool deflate_block(const unsigned char* in, std::size_t n, std::vector<unsigned char>& out) {
        std::vector<unsigned char> comp(compressBound((uLong)n) + 64);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = const_cast<unsigned char*>(in); zs.avail_in = (uInt)n;
        zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
        const int rc = deflate(&zs, Z_FINISH);
        const std::size_t clen = comp.size() - zs.avail_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END || 18 + clen + 8 > kBgzfMaxBlock) return false;
        out.clear();
        const unsigned char head[] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0};
        out.insert(out.end(), head, head + sizeof(head));
        put16(out, (std::uint16_t)(18 + clen + 8 - 1));  // BSIZE = total block size - 1
        out.insert(out.end(), comp.begin(), comp.begin() + (std::ptrdiff_t)clen);
        put32(out, (std::uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n));
        put32(out, (std::uint32_t)n);
        return true;
    }
    // deflates and writes the first `blocks` blocks of the buffer (the last may be short)
    bool flush_blocks(std::size_t blocks) {
        if (blocks == 0) return true;
        std::vector<std::vector<unsigned char>> out(blocks);
        std::atomic<std::size_t> next{0};
        std::atomic<int> failed{0};
        const std::size_t total = std::min(buf_.size(), blocks * kBgzfFill);
        auto work = [&]() {
            for (std::size_t i = next.fetch_add(1); i < blocks && failed.load() == 0; i = next.fetch_add(1)) {
                const std::size_t off = i * kBgzfFill, n = std::min(kBgzfFill, total - off);
                if (!deflate_block(buf_.data() + off, n, out[i])) failed.store(1);
            }
        };
        const unsigned t = (unsigned)std::min<std::size_t>(threads_, blocks);
        if (t <= 1) {
            work();
        } else {
            std::vector<std::thread> pool;
            pool.reserve(t - 1);
            for (unsigned k = 1; k < t; ++k) pool.emplace_back(work);
            work();
            for (auto& th : pool) th.join();
        }
        if (failed.load() != 0) return false;
        for (const auto& b : out)
            if (std::fwrite(b.data(), 1, b.size(), f_) != b.size()) return false;
        buf_.erase(buf_.begin(), buf_.begin() + (std::ptrdiff_t)total);
        return true;
    }
    std::FILE* f_ = nullptr;
    unsigned threads_ = 1;
    std::vector<unsigned char> buf_;
};

// caps on the sizes a file may claim (a corrupt or hostile length field must not wrap an allocation)
constexpr std::size_t kMaxHeaderText = std::size_t(1) << 30;   // l_text
constexpr std::size_t kMaxRefName = std::size_t(1) << 16;      // l_name
constexpr std::size_t kMaxRecordBytes = std::size_t(1) << 28;  // block_size
constexpr std::size_t kMaxReferences = std::size_t(1) << 24;   // n_ref

// the BAM header (magic, text, reference names and lengths), kept verbatim for the copy in write_bam
struct BamHeader {
    std::vector<unsigned char> raw;
    std::uint32_t n_ref = 0;
    std::uint32_t first_ref_length = 0;
    std::string text;                                               // the SAM header text (l_text bytes)
    std::vector<std::pair<std::string, std::uint32_t>> references;  // name, length
};

bool read_header(BgzfReader& in, BamHeader& h, std::string* err) {
    unsigned char b[8];
    if (!in.read(b, 8) || std::memcmp(b, "BAM\1", 4) != 0) return set_err(err, in.error().empty() ? "not a BAM file (magic)" : in.error());
    h.raw.assign(b, b + 8);
    // sizes come from the file: all arithmetic on them in size_t, every one capped before anything is allocated
    // (the reference lets HTSlib reject such files: sam_hdr_read, bam_api.cpp:366-372)
    const std::size_t l_text = le32(b + 4);
    if (l_text > kMaxHeaderText) return set_err(err, "BAM header text length out of range");
    std::vector<unsigned char> text(l_text);
    if (l_text && !in.read(text.data(), l_text)) return set_err(err, "truncated BAM header text");
    h.raw.insert(h.raw.end(), text.begin(), text.end());
    h.text.assign(text.begin(), text.end());
    while (!h.text.empty() && h.text.back() == '\0') h.text.pop_back();  // (some writers pad the text with NULs)
    if (!in.read(b, 4)) return set_err(err, "truncated BAM header (n_ref)");
    h.raw.insert(h.raw.end(), b, b + 4);
    h.n_ref = le32(b);
    if (h.n_ref > kMaxReferences) return set_err(err, "BAM reference count out of range");
    for (std::uint32_t r = 0; r < h.n_ref; ++r) {
        if (!in.read(b, 4)) return set_err(err, "truncated BAM reference list");
        h.raw.insert(h.raw.end(), b, b + 4);
        const std::size_t l_name = le32(b);
        if (l_name == 0 || l_name > kMaxRefName) return set_err(err, "BAM reference name length out of range");
        std::vector<unsigned char> name(l_name + 4);
        if (!in.read(name.data(), name.size())) return set_err(err, "truncated BAM reference entry");
        h.raw.insert(h.raw.end(), name.begin(), name.end());
        if (r == 0) h.first_ref_length = le32(name.data() + l_name);
        h.references.emplace_back(std::string(reinterpret_cast<const char*>(name.data()), l_name - 1), le32(name.data() + l_name));
    }
    return true;
}

// ---- SAM text (the reference writes it when the output's extension is not ".bam": bam_api.cpp:564, sam_open(..., "w"))
// One alignment record as a SAM line (SAM specification 1.4 and 4.2): the eleven mandatory fields, then the optional
// ones as TAG:TYPE:VALUE.  false if the record's fields run past its end.
bool record_to_sam(const std::vector<unsigned char>& rec, const BamHeader& h, std::string& line) {
    const unsigned char* p = rec.data() + 4;
    const std::size_t size = rec.size() - 4;
    const std::int32_t ref_id = (std::int32_t)le32(p), pos = (std::int32_t)le32(p + 4);
    const std::uint32_t l_read_name = p[8], mapq = p[9], n_cigar = le16(p + 12), flag = le16(p + 14), l_seq = le32(p + 16);
    const std::int32_t next_ref = (std::int32_t)le32(p + 20), next_pos = (std::int32_t)le32(p + 24), tlen = (std::int32_t)le32(p + 28);
    const std::size_t fixed_end = 32u + (std::size_t)l_read_name + 4u * (std::size_t)n_cigar + ((std::size_t)l_seq + 1) / 2 + (std::size_t)l_seq;
    if (l_read_name == 0 || fixed_end > size) return false;
    auto ref_name = [&](std::int32_t id) -> std::string {
        return id >= 0 && (std::size_t)id < h.references.size() ? h.references[(std::size_t)id].first : std::string("*");
    };
    line.clear();
    line.append(reinterpret_cast<const char*>(p + 32), l_read_name - 1);
    if (line.empty()) line = "*";
    line += '\t'; line += std::to_string(flag);
    line += '\t'; line += ref_name(ref_id);
    line += '\t'; line += std::to_string((std::int64_t)pos + 1);
    line += '\t'; line += std::to_string(mapq);
    line += '\t';
    const unsigned char* cg = p + 32 + l_read_name;
    if (n_cigar == 0) line += '*';
    for (std::uint32_t k = 0; k < n_cigar; ++k) {
        const std::uint32_t v = le32(cg + 4 * k);
        line += std::to_string(v >> 4);
        line += (v & 0xF) < 9 ? "MIDNSHP=X"[v & 0xF] : '?';
    }
    line += '\t';
    line += next_ref < 0 ? std::string("*") : (next_ref == ref_id ? std::string("=") : ref_name(next_ref));
    line += '\t'; line += std::to_string((std::int64_t)next_pos + 1);
    line += '\t'; line += std::to_string(tlen);
    line += '\t';
    const unsigned char* sq = cg + 4 * (std::size_t)n_cigar;
    if (l_seq == 0) line += '*';
    for (std::uint32_t i = 0; i < l_seq; ++i) line += "=ACMGRSVTWYHKDBN"[(sq[i >> 1] >> ((~i & 1u) << 2)) & 0xF];
    line += '\t';
    const unsigned char* ql = sq + ((std::size_t)l_seq + 1) / 2;
    if (l_seq == 0 || ql[0] == 0xFF) line += '*';
    else for (std::uint32_t i = 0; i < l_seq; ++i) line += (char)(ql[i] + 33);
    // optional fields
    const unsigned char* a = p + fixed_end;
    const unsigned char* const end = p + size;
    auto number = [&](char type, const unsigned char* v, std::string& out) -> std::size_t {  // appends the value, returns its size (0: unknown type)
        switch (type) {
            case 'c': out += std::to_string((int)(std::int8_t)v[0]); return 1;
            case 'C': out += std::to_string((unsigned)v[0]); return 1;
            case 's': out += std::to_string((int)(std::int16_t)le16(v)); return 2;
            case 'S': out += std::to_string((unsigned)le16(v)); return 2;
            case 'i': out += std::to_string((std::int32_t)le32(v)); return 4;
            case 'I': out += std::to_string(le32(v)); return 4;
            case 'f': { float f; const std::uint32_t u = le32(v); std::memcpy(&f, &u, 4); char buf[32]; std::snprintf(buf, sizeof(buf), "%g", (double)f); out += buf; return 4; }
            default: return 0;
        }
    };
    auto width = [](char type) -> std::size_t { return type == 'c' || type == 'C' ? 1 : type == 's' || type == 'S' ? 2 : type == 'i' || type == 'I' || type == 'f' ? 4 : 0; };
    while (a + 3 <= end) {
        line += '\t';
        line += (char)a[0]; line += (char)a[1]; line += ':';
        const char type = (char)a[2];
        a += 3;
        if (type == 'A') {
            if (a + 1 > end) return false;
            line += "A:"; line += (char)a[0]; a += 1;
        } else if (type == 'Z' || type == 'H') {
            line += type; line += ':';
            while (a < end && *a != 0) line += (char)*a++;
            if (a >= end) return false;
            ++a;
        } else if (type == 'B') {
            if (a + 5 > end) return false;
            const char sub = (char)a[0];
            const std::size_t count = le32(a + 1), w = width(sub);
            a += 5;
            if (w == 0 || count > (std::size_t)(end - a) / w) return false;
            line += "B:"; line += sub;
            for (std::size_t k = 0; k < count; ++k) { line += ','; number(sub, a, line); a += w; }
        } else {
            const std::size_t w = width(type);
            if (w == 0 || a + w > end) return false;
            line += (type == 'f' ? "f:" : "i:");   // (SAM text has one integer type)
            number(type, a, line);
            a += w;
        }
    }
    if (a != end) return false;
    line += '\n';
    return true;
}

// the header as SAM text: the BAM's text, and -- if it names no reference -- @SQ lines from the binary reference list
// (what HTSlib's sam_hdr_write makes of such a header)
std::string header_to_sam(const BamHeader& h) {
    std::string t = h.text;
    if (!t.empty() && t.back() != '\n') t += '\n';
    if (t.find("@SQ\t") == std::string::npos)
        for (const auto& r : h.references) t += "@SQ\tSN:" + r.first + "\tLN:" + std::to_string(r.second) + "\n";
    return t;
}

// one alignment record: block_size then the block; false at the end of the file (err stays empty) or on error
bool read_record(BgzfReader& in, std::vector<unsigned char>& rec, std::string* err) {
    unsigned char b[4];
    if (!in.read(b, 4)) {
        if (!in.eof()) set_err(err, in.error().empty() ? "truncated BAM record" : in.error());
        return false;
    }
    const std::size_t block_size = le32(b);
    if (block_size < 32) return set_err(err, "BAM record shorter than its fixed fields");
    if (block_size > kMaxRecordBytes) return set_err(err, "BAM record length out of range");
    rec.resize(4 + block_size);
    std::memcpy(rec.data(), b, 4);
    if (!in.read(rec.data() + 4, block_size)) return set_err(err, in.error().empty() ? "truncated BAM record" : in.error());
    return true;
}

// Read::Read(BAMReadId, bam1_t*), read.cpp:5-14: rlen = bases of the reference the CIGAR consumes (M D N = X)
bool record_to_read(const std::vector<unsigned char>& rec, BAMReadId id, Read& out, std::string& qname) {
    const unsigned char* p = rec.data() + 4;
    const std::int32_t pos = (std::int32_t)le32(p + 4);
    const std::uint32_t l_read_name = p[8];
    const std::uint32_t mapq = p[9];
    const std::uint32_t n_cigar = le16(p + 12);
    const std::uint16_t flag = le16(p + 14);
    const std::uint32_t l_seq = le32(p + 16);
    if (32u + l_read_name + 4u * n_cigar > rec.size() - 4) return false;
    qname.assign(reinterpret_cast<const char*>(p + 32), l_read_name ? l_read_name - 1 : 0);
    std::uint64_t rlen = 0;
    const unsigned char* cg = p + 32 + l_read_name;
    for (std::uint32_t k = 0; k < n_cigar; ++k) {
        const std::uint32_t v = le32(cg + 4 * k), op = v & 0xF, len = v >> 4;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += len;
    }
    out = Read(id, static_cast<Index>(pos), static_cast<Index>((std::int64_t)pos + (std::int64_t)rlen - 1), mapq,
               l_seq, (flag & 0x40) != 0);
    return true;
}

}  // namespace

bool read_bam(const std::filesystem::path& path, const BamFilters& filters, PairedReads& out,
              std::vector<BAMReadId>& filtered_out, BamIngestStats* stats, std::string* err) {
    BgzfReader in;
    if (!in.open(path)) return set_err(err, "could not open " + path.string());
    BamHeader h;
    if (!read_header(in, h, err)) return false;
    if (h.n_ref == 0) return set_err(err, "BAM without a reference sequence");
    out.ref_genome_length = h.first_ref_length;  // bam_api.cpp:422: single contig, target_len[0]

    BamIngestStats st;
    std::vector<bool> is_accepted, in_single_amplicon;
    std::map<std::string, Read> read_map;
    std::vector<unsigned char> rec;
    std::string qname, rec_err;
    BAMReadId id = 0;
    const bool filter_amplicons = filters.amplicon_behaviour == AmpliconBehaviour::FILTER && filters.amplicons != nullptr;
    const bool grade = filters.amplicon_behaviour == AmpliconBehaviour::GRADE && filters.amplicons != nullptr;
    while (read_record(in, rec, &rec_err)) {
        Read current(id, 0, 0, 0, 0, false);
        if (!record_to_read(rec, id, current, qname)) return set_err(err, "BAM record with fields past its end");
        is_accepted.push_back(false);
        auto it = read_map.find(qname);
        if (it != read_map.end()) {
            // bam_api.cpp:434-461: the pair is appended when its second mate is met; r1 IS the map entry
            Read& r1 = it->second;
            Read& r2 = current;
            bool drop = !(r1.quality >= filters.min_mapq && r2.quality >= filters.min_mapq) ||
                        !(r1.seq_length >= filters.min_seq_length && r2.seq_length >= filters.min_seq_length);
            if (filter_amplicons) drop = drop || !filters.amplicons->member_includes_both(r1, r2);
            if (drop) { ++id; continue; }
            if (grade) {
                for (const Read* r : {&r1, &r2}) {
                    st.min_imported_mapq = std::min(st.min_imported_mapq, (std::uint32_t)r->quality);
                    st.max_imported_mapq = std::max(st.max_imported_mapq, (std::uint32_t)r->quality);
                }
                const bool single = filters.amplicons->member_includes_both(r1, r2);
                in_single_amplicon.push_back(single);
                in_single_amplicon.push_back(single);
            }
            if (r2.is_first_read) std::swap(r1, r2);   // (swaps the map entry too, as the reference does)
            out.push_back(r1);
            out.push_back(r2);
            is_accepted[r1.bam_id] = true;
            is_accepted[r2.bam_id] = true;
        } else {
            read_map.insert({qname, current});
        }
        ++id;
    }
    if (!rec_err.empty()) return set_err(err, rec_err);
    filtered_out.clear();
    for (BAMReadId b = 0; b < is_accepted.size(); ++b)
        if (!is_accepted[b]) filtered_out.push_back(b);
    if (grade && st.max_imported_mapq > 0 && st.min_imported_mapq < UINT32_MAX) {
        // apply_amplicon_inclusion_grading, bam_api.cpp:334-349
        for (ReadIndex i = 0; i < out.get_reads_count(); ++i) {
            ReadQuality q = out.get_quality(i);
            q -= st.min_imported_mapq;
            if (in_single_amplicon[i]) q += st.max_imported_mapq - st.min_imported_mapq;
            out.set_quality(i, q);
        }
    }
    st.records = id;
    st.imported = out.get_reads_count();
    if (stats) *stats = st;
    return true;
}

std::uint32_t write_bam(const std::filesystem::path& input, const std::filesystem::path& output,
                        std::vector<BAMReadId>& bam_ids, std::string* err) {
    BgzfReader in;
    if (!in.open(input)) { set_err(err, "could not open " + input.string()); return UINT32_MAX; }
    BamHeader h;
    if (!read_header(in, h, err)) return UINT32_MAX;
    // bam_api.cpp:564: `output_filepath.extension() == ".bam" ? "wb" : "w"` -- any other extension is SAM text
    const bool as_bam = output.extension() == ".bam";
    BgzfWriter out;
    std::FILE* sam = nullptr;
    std::string sam_buf;
    if (as_bam) {
        if (!out.open(output)) { set_err(err, "could not open " + output.string()); return UINT32_MAX; }
        if (!out.write(h.raw.data(), h.raw.size())) { set_err(err, "write failed"); return UINT32_MAX; }
    } else {
        sam = std::fopen(output.c_str(), "wb");
        if (!sam) { set_err(err, "could not open " + output.string()); return UINT32_MAX; }
        sam_buf = header_to_sam(h);
    }
    auto fail_sam = [&](const std::string& msg) { if (sam) std::fclose(sam); set_err(err, msg); return UINT32_MAX; };
    std::sort(bam_ids.begin(), bam_ids.end());   // bam_api.cpp:603
    auto next = bam_ids.begin();
    std::vector<unsigned char> rec;
    std::string rec_err, line;
    BAMReadId id = 0;
    std::uint32_t written = 0;
    while (next != bam_ids.end() && read_record(in, rec, &rec_err)) {
        if (id == *next) {
            if (as_bam) {
                if (!out.write(rec.data(), rec.size())) { set_err(err, "write failed"); return UINT32_MAX; }
            } else {
                if (!record_to_sam(rec, h, line)) return fail_sam("BAM record with fields past its end");
                sam_buf += line;
                if (sam_buf.size() >= (std::size_t(1) << 20)) {
                    if (std::fwrite(sam_buf.data(), 1, sam_buf.size(), sam) != sam_buf.size()) return fail_sam("write failed");
                    sam_buf.clear();
                }
            }
            ++written;
            ++next;
        }
        ++id;
    }
    if (!rec_err.empty()) { if (sam) std::fclose(sam); set_err(err, rec_err); return UINT32_MAX; }
    if (as_bam) {
        if (!out.close()) { set_err(err, "closing " + output.string() + " failed"); return UINT32_MAX; }
    } else {
        const bool ok = std::fwrite(sam_buf.data(), 1, sam_buf.size(), sam) == sam_buf.size();
        if (std::fclose(sam) != 0 || !ok) { set_err(err, "closing " + output.string() + " failed"); return UINT32_MAX; }
    }
    return written;
}

bool write_synthetic_bam(const std::filesystem::path& path, const std::string& ref_name,
                         std::uint32_t ref_length, const std::vector<BamRecordSpec>& records,
                         std::string* err) {
    BgzfWriter out;
    if (!out.open(path)) return set_err(err, "could not open " + path.string());
    std::vector<unsigned char> b;
    const std::string text = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:" + ref_name + "\tLN:" + std::to_string(ref_length) + "\n";
    b.insert(b.end(), {'B', 'A', 'M', 1});
    put32(b, (std::uint32_t)text.size());
    b.insert(b.end(), text.begin(), text.end());
    put32(b, 1);
    put32(b, (std::uint32_t)ref_name.size() + 1);
    b.insert(b.end(), ref_name.begin(), ref_name.end());
    b.push_back(0);
    put32(b, ref_length);
    if (!out.write(b.data(), b.size())) return set_err(err, "write failed");
    const std::string ops = "MIDNSHP=X";
    for (const BamRecordSpec& r : records) {
        std::vector<unsigned char> blk;
        std::uint64_t rlen = 0;
        for (const auto& c : r.cigar)
            if (c.second == 'M' || c.second == 'D' || c.second == 'N' || c.second == '=' || c.second == 'X') rlen += c.first;
        // reg2bin of [pos, end): the bin field is not used on this path, but a reader may check it
        std::uint32_t beg = (std::uint32_t)r.pos, end = (std::uint32_t)(r.pos + (rlen ? rlen : 1)) - 1, bin = 0;
        if (beg >> 14 == end >> 14) bin = ((1u << 15) - 1) / 7 + (beg >> 14);
        else if (beg >> 17 == end >> 17) bin = ((1u << 12) - 1) / 7 + (beg >> 17);
        else if (beg >> 20 == end >> 20) bin = ((1u << 9) - 1) / 7 + (beg >> 20);
        else if (beg >> 23 == end >> 23) bin = ((1u << 6) - 1) / 7 + (beg >> 23);
        else if (beg >> 26 == end >> 26) bin = ((1u << 3) - 1) / 7 + (beg >> 26);
        put32(blk, 0);                                  // refID
        put32(blk, (std::uint32_t)r.pos);
        blk.push_back((unsigned char)(r.qname.size() + 1));
        blk.push_back(r.mapq);
        put16(blk, (std::uint16_t)bin);
        put16(blk, (std::uint16_t)r.cigar.size());
        put16(blk, r.flag);
        put32(blk, r.l_seq);
        put32(blk, 0xFFFFFFFFu);                        // next_refID = -1
        put32(blk, 0xFFFFFFFFu);                        // next_pos = -1
        put32(blk, 0);                                  // tlen
        blk.insert(blk.end(), r.qname.begin(), r.qname.end());
        blk.push_back(0);
        for (const auto& c : r.cigar) {
            const std::size_t op = ops.find(c.second);
            if (op == std::string::npos) return set_err(err, std::string("unknown CIGAR operation ") + c.second);
            put32(blk, (c.first << 4) | (std::uint32_t)op);
        }
        blk.insert(blk.end(), (r.l_seq + 1) / 2, 0x11);  // sequence: 'A' nibbles
        blk.insert(blk.end(), r.l_seq, 0xFF);            // qualities: absent
        std::vector<unsigned char> sz;
        put32(sz, (std::uint32_t)blk.size());
        if (!out.write(sz.data(), 4) || !out.write(blk.data(), blk.size())) return set_err(err, "write failed");
    }
    if (!out.close()) return set_err(err, "closing " + path.string() + " failed");
    return true;
}

}  // namespace bam_api
