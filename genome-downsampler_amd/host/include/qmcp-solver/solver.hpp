// Solver plugin interface.  Mirrors libs/qmcp-solver/include/qmcp-solver/solver.hpp:13-20:
// one virtual call in, one heap vector of kept ReadIndex out.
#ifndef QMCP_AMD_SOLVER_HPP
#define QMCP_AMD_SOLVER_HPP

#include <cstdint>
#include <memory>
#include <vector>

#include "bam-api/bam_api.hpp"
#include "bam-api/read.hpp"

namespace qmcp {

using Solution = std::vector<bam_api::ReadIndex>;

class Solver {
   public:
    virtual ~Solver() = default;
    virtual std::unique_ptr<Solution> solve(std::uint32_t max_coverage,
                                            bam_api::BamApi& bam_api) = 0;
    virtual bool uses_quality_of_reads() = 0;
};

}  // namespace qmcp
#endif
