// In-memory BamApi: the part of the reference's BamApi the solver path and its tests use.
// Mirrors libs/bam-api/include/bam-api/bam_api.hpp:21-88 for
//   BamApi(const AOSPairedReads&) / BamApi(const SOAPairedReads&)   (bam_api.cpp:44-51)
//   get_paired_reads_aos / get_paired_reads_soa / get_paired_reads  (bam_api.cpp:189-233,303-309)
//   find_pairs                                                     (bam_api.cpp:239-273)
//   find_input_cover / find_filtered_cover                         (bam_api.cpp:275-301)
//   BamApi(path, BamApiConfig), read on first use, get_filtered_out_reads,
//   write_paired_reads, write_bam_api_filtered_out_reads            (bam_api.cpp:32-43,189-233,520-532)
// BAM files are read and written by bam-api/bam_io.hpp on zlib alone (HTSlib is not in this image).
#ifndef QMCP_AMD_BAM_API_BAM_API_HPP
#define QMCP_AMD_BAM_API_BAM_API_HPP

#include <cstdint>
#include <filesystem>
#include <vector>

#include "bam-api/amplicon_set.hpp"
#include "bam-api/paired_reads.hpp"

namespace bam_api {

// bam_api_config.hpp:19-26
struct BamApiConfig {
    std::filesystem::path bed_filepath;
    std::filesystem::path tsv_filepath;
    std::uint32_t hts_thread_count = 1;  // kept for source compatibility; ingest here is single-threaded
    std::uint32_t min_seq_length = 0;
    std::uint32_t min_mapq = 0;
    AmpliconBehaviour amplicon_behaviour = AmpliconBehaviour::IGNORE;
};

class BamApi {
   public:
    // reading and filtering happen on the first get_paired_reads_* call (bam_api.cpp:189-233)
    BamApi(const std::filesystem::path& input_filepath, const BamApiConfig& config);
    explicit BamApi(const AOSPairedReads& paired_reads);
    explicit BamApi(const SOAPairedReads& paired_reads);

    const AOSPairedReads& get_paired_reads_aos();
    const SOAPairedReads& get_paired_reads_soa();
    const PairedReads& get_paired_reads() const;
    void set_amplicon_behaviour(AmpliconBehaviour b) { amplicon_behaviour_ = b; }
    const std::vector<BAMReadId>& get_filtered_out_reads() const { return filtered_out_reads_; }
    // number of records written; the output is always BAM
    std::uint32_t write_paired_reads(const std::filesystem::path& output_filepath,
                                     std::vector<ReadIndex>& active_ids) const;
    std::uint32_t write_bam_api_filtered_out_reads(const std::filesystem::path& output_filepath);

    std::vector<ReadIndex> find_pairs(const std::vector<ReadIndex>& ids) const;

    // testing purposes (same role as in the reference)
    std::vector<std::uint32_t> find_input_cover();
    std::vector<std::uint32_t> find_filtered_cover(const std::vector<ReadIndex>& active_ids);

   private:
    SOAPairedReads soa_paired_reads_;
    bool is_soa_loaded_ = false;
    AOSPairedReads aos_paired_reads_;
    bool is_aos_loaded_ = false;
    AmpliconSet amplicon_set_;
    AmpliconBehaviour amplicon_behaviour_ = AmpliconBehaviour::IGNORE;
    std::vector<BAMReadId> filtered_out_reads_;
    std::filesystem::path input_filepath_;
    std::uint32_t min_seq_length_ = 0, min_mapq_ = 0;
    void read_bam_into(PairedReads& reads);
};

}  // namespace bam_api
#endif
