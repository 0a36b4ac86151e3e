// Read containers at the solver boundary.  Mirrors the reference's
//   PairedReads     libs/bam-api/include/bam-api/paired_reads.hpp:10-21
//   AOSPairedReads  libs/bam-api/include/bam-api/aos_paired_reads.hpp:16-27
//   SOAPairedReads  libs/bam-api/include/bam-api/soa_paired_reads.hpp:18-59
// Mates sit at indices (2q, 2q+1); reads are NOT sorted by position.
#ifndef QMCP_AMD_BAM_API_PAIRED_READS_HPP
#define QMCP_AMD_BAM_API_PAIRED_READS_HPP

#include <vector>

#include "bam-api/read.hpp"

namespace bam_api {

struct PairedReads {
    Index ref_genome_length = 0;

    virtual ~PairedReads() = default;
    virtual void push_back(const Read& read) = 0;
    virtual Read get_read_by_index(ReadIndex index) const = 0;
    virtual ReadQuality get_quality(ReadIndex index) const = 0;
    virtual void set_quality(ReadIndex index, ReadQuality quality) = 0;
    virtual ReadIndex get_reads_count() const = 0;
    virtual void reserve(std::size_t size) = 0;
};

struct SOAPairedReads;

struct AOSPairedReads : PairedReads {
    std::vector<Read> reads;

    void push_back(const Read& read) override { reads.push_back(read); }
    Read get_read_by_index(ReadIndex index) const override { return reads[index]; }
    ReadQuality get_quality(ReadIndex index) const override { return reads[index].quality; }
    void set_quality(ReadIndex index, ReadQuality q) override { reads[index].quality = q; }
    ReadIndex get_reads_count() const override { return reads.size(); }
    void reserve(std::size_t size) override { reads.reserve(size); }
    void clear() { reads.clear(); }
    AOSPairedReads& from(const SOAPairedReads& soa);
};

struct SOAPairedReads : PairedReads {
    std::vector<BAMReadId> ids;
    std::vector<Index> start_inds;
    std::vector<Index> end_inds;
    std::vector<ReadQuality> qualities;
    std::vector<std::uint32_t> seq_lengths;
    std::vector<bool> is_first_reads;

    void push_back(const Read& r) override {
        ids.push_back(r.bam_id);
        start_inds.push_back(r.start_ind);
        end_inds.push_back(r.end_ind);
        qualities.push_back(r.quality);
        seq_lengths.push_back(r.seq_length);
        is_first_reads.push_back(r.is_first_read);
    }
    Read get_read_by_index(ReadIndex i) const override {
        return Read(ids[i], start_inds[i], end_inds[i], qualities[i], seq_lengths[i],
                    is_first_reads[i]);
    }
    ReadQuality get_quality(ReadIndex i) const override { return qualities[i]; }
    void set_quality(ReadIndex i, ReadQuality q) override { qualities[i] = q; }
    ReadIndex get_reads_count() const override { return ids.size(); }
    void reserve(std::size_t n) override {
        ids.reserve(n); start_inds.reserve(n); end_inds.reserve(n);
        qualities.reserve(n); seq_lengths.reserve(n); is_first_reads.reserve(n);
    }
    void clear() {
        ids.clear(); start_inds.clear(); end_inds.clear();
        qualities.clear(); seq_lengths.clear(); is_first_reads.clear();
    }
    SOAPairedReads& from(const AOSPairedReads& aos);
};

inline AOSPairedReads& AOSPairedReads::from(const SOAPairedReads& soa) {
    ref_genome_length = soa.ref_genome_length;
    reads.clear();
    reads.reserve(soa.get_reads_count());
    for (ReadIndex i = 0; i < soa.get_reads_count(); ++i) reads.push_back(soa.get_read_by_index(i));
    return *this;
}
inline SOAPairedReads& SOAPairedReads::from(const AOSPairedReads& aos) {
    ref_genome_length = aos.ref_genome_length;
    clear();
    reserve(aos.reads.size());
    for (const Read& r : aos.reads) push_back(r);
    return *this;
}

}  // namespace bam_api
#endif
