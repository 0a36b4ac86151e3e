// "quasi-mcp-hip": the MI355X solver behind the reference's plugin surface.
// Takes the place QuasiMcpCudaMaxFlowSolver has in the reference
// (libs/qmcp-solver/include/qmcp-solver/quasi_mcp_cuda_max_flow_solver.hpp:17-36):
// same base class, same two overrides, uses_quality_of_reads() == false so the app selects
// amplicon FILTER (src/app.cpp:121-127).  All device work goes through the C ABI in
// include/qmcp_hip.h; this class only hands over the SoA columns and expands the keep mask.
#ifndef QMCP_AMD_QUASI_MCP_HIP_SOLVER_HPP
#define QMCP_AMD_QUASI_MCP_HIP_SOLVER_HPP

#include <cstdint>
#include <memory>
#include <vector>

#include "qmcp-solver/solver.hpp"
#include "qmcp_hip.h"

namespace qmcp {

class QuasiMcpHipSolver : public Solver {
   public:
    QuasiMcpHipSolver() = default;  // trivial: solvers are built eagerly (src/app.hpp:35)
    ~QuasiMcpHipSolver() override;
    QuasiMcpHipSolver(const QuasiMcpHipSolver&) = delete;
    QuasiMcpHipSolver& operator=(const QuasiMcpHipSolver&) = delete;

    std::unique_ptr<Solution> solve(std::uint32_t required_cover,
                                    bam_api::BamApi& bam_api) override;
    bool uses_quality_of_reads() override { return false; }

    void set_device(int device);  // before the first solve; default 0
    // devices for multi-contig callers of the C ABI's qmcp_hip_multi_* entry points; the reference's BamApi
    // holds ONE contig (libs/bam-api/src/bam_api.cpp:422), which is one independent problem, so solve()
    // below runs it on the first device of the list
    void set_devices(const std::vector<int>& devices) { if (!devices.empty()) { devices_ = devices; device_ = devices[0]; } }
    const std::vector<int>& devices() const { return devices_; }
    // complete mate pairs on the device before returning (what src/app.cpp:141 does on the
    // host with BamApi::find_pairs); off by default, like the reference solvers
    void set_complete_pairs(bool on) { complete_pairs_ = on; }
    const qmcp_hip_stats& last_stats() const { return stats_; }
    // host wall-clock of the last solve(): the library's parts, the mask -> Solution expansion, the whole call
    const qmcp_hip_host_breakdown& last_breakdown() const { return breakdown_; }
    float last_expand_ms() const { return ms_expand_; }
    float last_solve_call_ms() const { return ms_solve_call_; }

   private:
    qmcp_hip_ctx* ctx_ = nullptr;  // created on first solve, reused across solves
    int device_ = 0;
    std::vector<int> devices_{0};
    bool complete_pairs_ = false;
    qmcp_hip_stats stats_{};
    qmcp_hip_host_breakdown breakdown_{};
    float ms_expand_ = 0.f, ms_solve_call_ = 0.f;
};

}  // namespace qmcp
#endif
