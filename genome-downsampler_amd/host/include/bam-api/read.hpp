// Host-side mirror of the reference's read record for the quasi-MCP solver path.
// Mirrors libs/bam-api/include/bam-api/read.hpp:11-25 (same type names, field names and
// meaning; end_ind is INCLUSIVE, libs/bam-api/src/read.cpp:13).  The HTSlib constructor
// (read.hpp:27) is out of scope: BAM ingest is host IO the solver never sees.
#ifndef QMCP_AMD_BAM_API_READ_HPP
#define QMCP_AMD_BAM_API_READ_HPP

#include <cstddef>
#include <cstdint>

namespace bam_api {

using BAMReadId = std::size_t;   // line of the read in the BAM file
using ReadIndex = std::size_t;   // index of the read in the in-memory arrays
using Index = std::size_t;       // position on the reference genome
using ReadQuality = std::uint32_t;

struct Read {
    BAMReadId bam_id = 0;
    Index start_ind = 0;
    Index end_ind = 0;  // inclusive
    ReadQuality quality = 0;
    std::uint32_t seq_length = 0;
    bool is_first_read = false;

    Read() = default;
    Read(BAMReadId id, Index start, Index end, ReadQuality q, std::uint32_t len, bool first)
        : bam_id(id), start_ind(start), end_ind(end), quality(q), seq_length(len),
          is_first_read(first) {}
};

}  // namespace bam_api
#endif
