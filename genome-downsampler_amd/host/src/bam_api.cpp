// In-memory BamApi (see include/bam-api/bam_api.hpp for the reference lines mirrored).
#include "bam-api/bam_api.hpp"

#include <cstdio>
#include <cstdlib>

#include "bam-api/bam_io.hpp"

namespace bam_api {

// bam_api.cpp:32-43: filters from the config; amplicons only when a BED file is given
BamApi::BamApi(const std::filesystem::path& input_filepath, const BamApiConfig& config)
    : input_filepath_(input_filepath), min_seq_length_(config.min_seq_length), min_mapq_(config.min_mapq) {
    if (!config.bed_filepath.empty()) {
        if (!amplicon_set_from_files(config.bed_filepath, config.tsv_filepath, amplicon_set_)) {
            std::fprintf(stderr, "[ERROR] could not open %s\n", config.bed_filepath.c_str());
            std::exit(EXIT_FAILURE);  // the reference exits the process on an unreadable input
        }
        amplicon_behaviour_ = config.amplicon_behaviour;
    }
}

void BamApi::read_bam_into(PairedReads& reads) {
    BamFilters f;
    f.min_seq_length = min_seq_length_;
    f.min_mapq = min_mapq_;
    f.amplicon_behaviour = amplicon_behaviour_;
    f.amplicons = &amplicon_set_;
    std::string err;
    if (!read_bam(input_filepath_, f, reads, filtered_out_reads_, nullptr, &err)) {
        std::fprintf(stderr, "[ERROR] %s\n", err.c_str());
        std::exit(EXIT_FAILURE);
    }
}

std::uint32_t BamApi::write_paired_reads(const std::filesystem::path& output_filepath,
                                         std::vector<ReadIndex>& active_ids) const {
    const PairedReads& reads = get_paired_reads();
    std::vector<BAMReadId> bam_ids;
    bam_ids.reserve(active_ids.size());
    for (ReadIndex id : active_ids) bam_ids.push_back(reads.get_read_by_index(id).bam_id);
    std::string err;
    const std::uint32_t n = write_bam(input_filepath_, output_filepath, bam_ids, &err);
    if (n == UINT32_MAX) { std::fprintf(stderr, "[ERROR] %s\n", err.c_str()); std::exit(EXIT_FAILURE); }
    return n;
}

std::uint32_t BamApi::write_bam_api_filtered_out_reads(const std::filesystem::path& output_filepath) {
    std::string err;
    const std::uint32_t n = write_bam(input_filepath_, output_filepath, filtered_out_reads_, &err);
    if (n == UINT32_MAX) { std::fprintf(stderr, "[ERROR] %s\n", err.c_str()); std::exit(EXIT_FAILURE); }
    return n;
}

BamApi::BamApi(const AOSPairedReads& paired_reads)
    : aos_paired_reads_(paired_reads), is_aos_loaded_(true) {}

BamApi::BamApi(const SOAPairedReads& paired_reads)
    : soa_paired_reads_(paired_reads), is_soa_loaded_(true) {}

// The reference converts lazily between layouts on first request of the other one
// (bam_api.cpp:189-233); same here.
const AOSPairedReads& BamApi::get_paired_reads_aos() {
    if (!is_aos_loaded_) {
        if (is_soa_loaded_ || input_filepath_.empty()) aos_paired_reads_.from(soa_paired_reads_);
        else read_bam_into(aos_paired_reads_);
        is_aos_loaded_ = true;
    }
    return aos_paired_reads_;
}

const SOAPairedReads& BamApi::get_paired_reads_soa() {
    if (!is_soa_loaded_) {
        if (is_aos_loaded_ || input_filepath_.empty()) soa_paired_reads_.from(aos_paired_reads_);
        else read_bam_into(soa_paired_reads_);
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
