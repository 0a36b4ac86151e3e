// In-memory BamApi: the part of the reference's BamApi the solver path and its tests use.
// Mirrors libs/bam-api/include/bam-api/bam_api.hpp:21-88 for
//   BamApi(const AOSPairedReads&) / BamApi(const SOAPairedReads&)   (bam_api.cpp:44-51)
//   get_paired_reads_aos / get_paired_reads_soa / get_paired_reads  (bam_api.cpp:189-233,303-309)
//   find_pairs                                                     (bam_api.cpp:239-273)
//   find_input_cover / find_filtered_cover                         (bam_api.cpp:275-301)
// The file-backed constructor, BED/TSV parsing and BAM writing need HTSlib and are out of
// scope.
#ifndef QMCP_AMD_BAM_API_BAM_API_HPP
#define QMCP_AMD_BAM_API_BAM_API_HPP

#include <cstdint>
#include <vector>

#include "bam-api/paired_reads.hpp"

namespace bam_api {

class BamApi {
   public:
    explicit BamApi(const AOSPairedReads& paired_reads);
    explicit BamApi(const SOAPairedReads& paired_reads);

    const AOSPairedReads& get_paired_reads_aos();
    const SOAPairedReads& get_paired_reads_soa();
    const PairedReads& get_paired_reads() const;

    std::vector<ReadIndex> find_pairs(const std::vector<ReadIndex>& ids) const;

    // testing purposes (same role as in the reference)
    std::vector<std::uint32_t> find_input_cover();
    std::vector<std::uint32_t> find_filtered_cover(const std::vector<ReadIndex>& active_ids);

   private:
    SOAPairedReads soa_paired_reads_;
    bool is_soa_loaded_ = false;
    AOSPairedReads aos_paired_reads_;
    bool is_aos_loaded_ = false;
};

}  // namespace bam_api
#endif
