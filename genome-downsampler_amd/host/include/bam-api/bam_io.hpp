// BAM ingest / emit without HTSlib: BGZF is a sequence of gzip members
// (zlib is enough) and the BAM records the path needs are fixed-layout.  Mirrors
//   Read::Read(BAMReadId, bam1_t*)      libs/bam-api/src/read.cpp:5-14   (end = pos + reference length of the CIGAR - 1)
//   BamApi::read_bam                    libs/bam-api/src/bam_api.cpp:359-507 (qname map pairing, pair appended when its
//                                       second mate is met, filters per pair, filtered_out_reads_)
//   BamApi::write_bam                   bam_api.cpp:534-656 (second pass over the input, records whose running id is in
//                                       the sorted id list are copied; header copied)
// Parity note: the reference holds no BAM fixture and HTSlib is not in this image, so this module is checked
// by round trips on BAMs its own writer produced (tests/test_bam_io.py): parity unpinned.
#ifndef QMCP_AMD_BAM_API_BAM_IO_HPP
#define QMCP_AMD_BAM_API_BAM_IO_HPP

#include <cstdint>
#include <filesystem>
#include <string>
#include <vector>

#include "bam-api/amplicon_set.hpp"
#include "bam-api/paired_reads.hpp"

namespace bam_api {

struct BamFilters {
    std::uint32_t min_seq_length = 0;  // on l_seq of both mates   (bam_api.cpp:321-323)
    std::uint32_t min_mapq = 0;        // on MAPQ of both mates    (bam_api.cpp:325-327)
    AmpliconBehaviour amplicon_behaviour = AmpliconBehaviour::IGNORE;
    const AmpliconSet* amplicons = nullptr;
};

struct BamIngestStats {
    std::uint64_t records = 0;        // alignments in the file (BAMReadId runs over these)
    std::uint64_t imported = 0;       // reads appended to the container (whole pairs)
    std::uint32_t min_imported_mapq = UINT32_MAX, max_imported_mapq = 0;  // GRADE only (bam_api.cpp:351-357)
};

// Appends accepted pairs to `out` (mate with FREAD1 first), sets out.ref_genome_length to the first
// reference's length, lists every record id that was not imported in `filtered_out` (ascending).
// false + *err on a malformed or unreadable file (the reference exits the process there).
bool read_bam(const std::filesystem::path& path, const BamFilters& filters, PairedReads& out,
              std::vector<BAMReadId>& filtered_out, BamIngestStats* stats, std::string* err);

// Copies the header and the records whose running id is in `bam_ids` (sorted in place, as the reference does)
// to a new file: BAM if the output's extension is ".bam", SAM text otherwise (bam_api.cpp:564).  BGZF blocks are
// deflated on several threads (QMCP_BAM_THREADS; the reference: hts_set_thread_pool(outfile), bam_api.cpp:569-586).
// Returns the number of records written, or UINT32_MAX + *err on failure.
std::uint32_t write_bam(const std::filesystem::path& input, const std::filesystem::path& output,
                        std::vector<BAMReadId>& bam_ids, std::string* err);

// A record for the synthetic-BAM writer used by the tests (the reference's tests build reads in memory
// and never write one): single reference, CIGAR given as (length, op) with op in "MIDNSHP=X".
struct BamRecordSpec {
    std::string qname;
    std::uint16_t flag = 0;
    std::int32_t pos = 0;
    std::uint8_t mapq = 0;
    std::vector<std::pair<std::uint32_t, char>> cigar;
    std::uint32_t l_seq = 0;
};
bool write_synthetic_bam(const std::filesystem::path& path, const std::string& ref_name,
                         std::uint32_t ref_length, const std::vector<BamRecordSpec>& records,
                         std::string* err);

}  // namespace bam_api
#endif
