#include "qmcp-solver/quasi_mcp_hip_solver.hpp"

#include <cstdio>
#include <exception>
#include <limits>
#include <vector>

namespace qmcp {
namespace {

// The reference's GPU solver has no error channel: any device failure ends in
// std::terminate() (libs/qmcp-solver/include/qmcp-solver/cuda_helpers.cuh:13-22).
// The C ABI returns codes; the adapter restores the reference behaviour.
[[noreturn]] void die(const char* what, int rc) {
    std::fprintf(stderr, "[ERROR] quasi-mcp-hip: %s failed (%d): %s\n", what, rc,
                 qmcp_hip_last_error());
    std::terminate();
}

}  // namespace

QuasiMcpHipSolver::~QuasiMcpHipSolver() {
    if (ctx_ != nullptr) qmcp_hip_destroy(ctx_);
}

void QuasiMcpHipSolver::set_device(int device) { device_ = device; }

std::unique_ptr<Solution> QuasiMcpHipSolver::solve(std::uint32_t required_cover,
                                                   bam_api::BamApi& bam_api) {
    // The reference solvers deep-copy the whole container (quasi_mcp_cpu_max_flow_solver.cpp:13,
    // quasi_mcp_cuda_max_flow_solver.cu:321); only two columns are needed, narrowed to the
    // uint32 coordinates the CUDA solver already uses (quasi_mcp_cuda_max_flow_solver.hpp:19).
    const bam_api::SOAPairedReads& reads = bam_api.get_paired_reads_soa();
    const std::size_t n = reads.start_inds.size();
    constexpr std::size_t kMax = std::numeric_limits<std::uint32_t>::max();
    if (reads.ref_genome_length > kMax) die("narrowing ref_genome_length", QMCP_ERANGE);

    std::vector<std::uint32_t> starts(n), ends(n);
    for (std::size_t i = 0; i < n; ++i) {
        if (reads.start_inds[i] > kMax || reads.end_inds[i] > kMax) die("narrowing reads", QMCP_ERANGE);
        starts[i] = static_cast<std::uint32_t>(reads.start_inds[i]);
        ends[i] = static_cast<std::uint32_t>(reads.end_inds[i]);
    }

    if (ctx_ == nullptr) {
        const int rc = qmcp_hip_create(device_, &ctx_);
        if (rc != QMCP_OK) die("qmcp_hip_create", rc);
    }

    const std::uint64_t offsets[2] = {0, n};
    const std::uint32_t length = static_cast<std::uint32_t>(reads.ref_genome_length);
    std::vector<std::uint64_t> mask((n + 63) / 64, 0);
    int rc = qmcp_hip_solve_host(ctx_, starts.data(), ends.data(), n, offsets, &length, 1,
                                 required_cover, mask.data(), &stats_);
    if (rc != QMCP_OK) die("qmcp_hip_solve_host", rc);
    if (complete_pairs_) {
        rc = qmcp_hip_complete_pairs_host(ctx_, mask.data(), n);
        if (rc != QMCP_OK) die("qmcp_hip_complete_pairs_host", rc);
    }

    // ascending ReadIndex, as obtain_sequence produces (quasi_mcp_cpu_max_flow_solver.cpp:93-97)
    auto kept = std::make_unique<Solution>();
    kept->reserve(stats_.n_kept);
    for (std::size_t w = 0; w < mask.size(); ++w) {
        std::uint64_t bits = mask[w];
        while (bits != 0) {
            const int b = __builtin_ctzll(bits);
            kept->push_back(w * 64 + static_cast<std::size_t>(b));
            bits &= bits - 1;
        }
    }
    return kept;
}

}  // namespace qmcp
