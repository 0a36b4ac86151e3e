// In-memory BamApi (see include/bam-api/bam_api.hpp for the reference lines mirrored).
#include "bam-api/bam_api.hpp"

namespace bam_api {

BamApi::BamApi(const AOSPairedReads& paired_reads)
    : aos_paired_reads_(paired_reads), is_aos_loaded_(true) {}

BamApi::BamApi(const SOAPairedReads& paired_reads)
    : soa_paired_reads_(paired_reads), is_soa_loaded_(true) {}

// The reference converts lazily between layouts on first request of the other one
// (bam_api.cpp:189-233); same here.
const AOSPairedReads& BamApi::get_paired_reads_aos() {
    if (!is_aos_loaded_) {
        aos_paired_reads_.from(soa_paired_reads_);
        is_aos_loaded_ = true;
    }
    return aos_paired_reads_;
}

const SOAPairedReads& BamApi::get_paired_reads_soa() {
    if (!is_soa_loaded_) {
        soa_paired_reads_.from(aos_paired_reads_);
        is_soa_loaded_ = true;
    }
    return soa_paired_reads_;
}

const PairedReads& BamApi::get_paired_reads() const {
    if (is_soa_loaded_) return soa_paired_reads_;
    return aos_paired_reads_;
}

// bam_api.cpp:239-273: output keeps first-seen order: id, then its mate, de-duplicated.
std::vector<ReadIndex> BamApi::find_pairs(const std::vector<ReadIndex>& ids) const {
    const PairedReads& reads = get_paired_reads();
    const ReadIndex n = reads.get_reads_count();
    std::vector<ReadIndex> out;
    out.reserve(n);
    std::vector<bool> seen(n, false);
    for (ReadIndex id : ids) {
        if (!seen[id]) { seen[id] = true; out.push_back(id); }
        const ReadIndex mate = reads.get_read_by_index(id).is_first_read ? id + 1 : id - 1;
        if (!seen[mate]) { seen[mate] = true; out.push_back(mate); }
    }
    return out;
}

std::vector<std::uint32_t> BamApi::find_input_cover() {
    const PairedReads& reads = get_paired_reads();
    std::vector<std::uint32_t> cover(reads.ref_genome_length, 0);
    for (ReadIndex i = 0; i < reads.get_reads_count(); ++i) {
        const Read r = reads.get_read_by_index(i);
        for (Index p = r.start_ind; p <= r.end_ind; ++p) ++cover[p];
    }
    return cover;
}

std::vector<std::uint32_t> BamApi::find_filtered_cover(const std::vector<ReadIndex>& ids) {
    const PairedReads& reads = get_paired_reads();
    std::vector<std::uint32_t> cover(reads.ref_genome_length, 0);
    for (ReadIndex id : ids) {
        const Read r = reads.get_read_by_index(id);
        for (Index p = r.start_ind; p <= r.end_ind; ++p) ++cover[p];
    }
    return cover;
}

}  // namespace bam_api
