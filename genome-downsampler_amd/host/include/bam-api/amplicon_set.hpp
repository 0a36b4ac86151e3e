// Amplicon intervals for the FILTER pre-pass.  Mirrors
//   Amplicon / Amplicon::includes            libs/bam-api/include/bam-api/amplicon.hpp:8-16, src/amplicon.cpp:5-7
//   AmpliconSet::member_includes_both        libs/bam-api/src/amplicon_set.cpp:5-9
//   BamApi::set_amplicon_filter              libs/bam-api/src/bam_api.cpp:53-95
//   BamApi::process_bed_file / process_tsv_file   bam_api.cpp:101-186
// The device predicate is qmcp_hip_amplicon_filter_host (include/qmcp_hip.h); this header is
// the host-side construction of the interval list it consumes.
#ifndef QMCP_AMD_BAM_API_AMPLICON_SET_HPP
#define QMCP_AMD_BAM_API_AMPLICON_SET_HPP

#include <filesystem>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "bam-api/read.hpp"

namespace bam_api {

enum class AmpliconBehaviour { IGNORE, FILTER, GRADE };  // bam_api_config.hpp:10-17

struct Amplicon {
    Index start;
    Index end;  // compared inclusively, although BED ends are exclusive: kept as the reference does
    Amplicon(Index s, Index e) : start(s), end(e) {}
    bool includes(const Read& read) const { return start <= read.start_ind && read.end_ind <= end; }
};

struct AmpliconSet {
    std::vector<Amplicon> amplicons;
    bool member_includes_both(const Read& r1, const Read& r2) const {
        for (const Amplicon& a : amplicons)
            if (a.includes(r1) && a.includes(r2)) return true;
        return false;
    }
};

using PrimerMap = std::map<std::string, std::pair<Index, Index>>;

// BED: chrom \t start \t end \t name; the first line carrying a name wins; lines whose
// coordinates do not parse or with an empty field are skipped.  Returns false if the file
// cannot be opened (the reference exits the process there).
bool read_primer_bed(const std::filesystem::path& path, PrimerMap& out);
// TSV: left \t right primer names.
bool read_primer_pairs_tsv(const std::filesystem::path& path,
                           std::vector<std::pair<std::string, std::string>>& out);

// set_amplicon_filter: with a TSV every listed pair gives [left.start, right.end] after
// ordering the two primers by start (the reorder is applied to the map entries themselves,
// as in the reference, so it is visible to later pairs that reuse a primer; unknown names
// behave as (0, 0) primers); without a TSV consecutive primers in name order are paired.
AmpliconSet build_amplicon_set(PrimerMap primers,
                               const std::vector<std::pair<std::string, std::string>>* pairs);
bool amplicon_set_from_files(const std::filesystem::path& bed, const std::filesystem::path& tsv,
                             AmpliconSet& out);

}  // namespace bam_api
#endif
