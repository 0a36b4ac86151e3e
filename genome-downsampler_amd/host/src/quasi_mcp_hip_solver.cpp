#include "qmcp-solver/quasi_mcp_hip_solver.hpp"

#include <chrono>
#include <cstdio>
#include <exception>
#include <algorithm>
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
    const auto t0 = std::chrono::steady_clock::now();
    const std::size_t n = reads.start_inds.size();
    constexpr std::size_t kMax = std::numeric_limits<std::uint32_t>::max();
    if (reads.ref_genome_length > kMax) die("narrowing ref_genome_length", QMCP_ERANGE);
    static_assert(sizeof(bam_api::Index) == sizeof(std::uint64_t), "Index is size_t on an LP64 host (read.hpp:11)");

    if (ctx_ == nullptr) {
        const int rc = qmcp_hip_create(device_, &ctx_);
        if (rc != QMCP_OK) die("qmcp_hip_create", rc);
    }

    // the 64-bit columns go to the library as they are: it narrows them chunk by chunk on several
    // threads straight into pinned staging, each chunk's copy to the device issued as it is ready
    const std::uint64_t offsets[2] = {0, n};
    const std::uint32_t length = static_cast<std::uint32_t>(reads.ref_genome_length);
    // complete_pairs needs the mask on the host side of the ABI; otherwise it stays on the device and only
    // the Solution comes back
    std::vector<std::uint64_t> mask;
    if (complete_pairs_) mask.assign((n + 63) / 64, 0);
    int rc = qmcp_hip_solve_host64(ctx_, reinterpret_cast<const std::uint64_t*>(reads.start_inds.data()),
                                   reinterpret_cast<const std::uint64_t*>(reads.end_inds.data()), n, offsets,
                                   &length, 1, required_cover, complete_pairs_ ? mask.data() : nullptr, &stats_,
                                   &breakdown_);
    if (rc != QMCP_OK) die("qmcp_hip_solve_host64", rc);
    if (complete_pairs_) {
        rc = qmcp_hip_complete_pairs_host(ctx_, mask.data(), n);   // (leaves the completed mask in the context)
        if (rc != QMCP_OK) die("qmcp_hip_complete_pairs_host", rc);
    }
    const auto t1 = std::chrono::steady_clock::now();

    // ascending ReadIndex, as obtain_sequence produces (quasi_mcp_cpu_max_flow_solver.cpp:93-97): expanded from
    // the keep mask on the device (popcounts, scan, scatter) and copied out -- 5 % of the reads at cfg4's depth
    auto kept = std::make_unique<Solution>();
    static_assert(sizeof(bam_api::ReadIndex) == sizeof(std::uint64_t), "ReadIndex is size_t on an LP64 host");
    const std::uint64_t upper = complete_pairs_ ? std::min<std::uint64_t>(n, 2 * stats_.n_kept) : stats_.n_kept;
    kept->resize(upper);
    std::uint64_t n_out = 0;
    rc = qmcp_hip_kept_indices_host(ctx_, n, reinterpret_cast<std::uint64_t*>(kept->data()), upper, &n_out);
    if (rc != QMCP_OK) die("qmcp_hip_kept_indices_host", rc);
    kept->resize(n_out);
    const auto t2 = std::chrono::steady_clock::now();
    ms_expand_ = std::chrono::duration<float, std::milli>(t2 - t1).count();
    ms_solve_call_ = std::chrono::duration<float, std::milli>(t2 - t0).count();
    return kept;
}

}  // namespace qmcp
