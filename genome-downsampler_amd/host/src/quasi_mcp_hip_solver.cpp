#include "qmcp-solver/quasi_mcp_hip_solver.hpp"

#include <chrono>
#include <cstdio>
#include <exception>
#include <limits>
#include <thread>
#include <utility>
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
    std::vector<std::uint64_t> mask((n + 63) / 64, 0);
    int rc = qmcp_hip_solve_host64(ctx_, reinterpret_cast<const std::uint64_t*>(reads.start_inds.data()),
                                   reinterpret_cast<const std::uint64_t*>(reads.end_inds.data()), n, offsets,
                                   &length, 1, required_cover, mask.data(), &stats_, &breakdown_);
    if (rc != QMCP_OK) die("qmcp_hip_solve_host64", rc);
    if (complete_pairs_) {
        rc = qmcp_hip_complete_pairs_host(ctx_, mask.data(), n);
        if (rc != QMCP_OK) die("qmcp_hip_complete_pairs_host", rc);
    }
    const auto t1 = std::chrono::steady_clock::now();

    // ascending ReadIndex, as obtain_sequence produces (quasi_mcp_cpu_max_flow_solver.cpp:93-97): the mask
    // is cut into word ranges, each counted and then expanded by its own thread into its slice
    auto kept = std::make_unique<Solution>();
    const std::size_t n_words = mask.size();
    const unsigned parts = n_words >= (1u << 16) ? 4u : 1u;
    std::vector<std::size_t> first(parts + 1, 0);
    auto range = [&](unsigned p) { return std::pair<std::size_t, std::size_t>(n_words * p / parts, n_words * (p + 1) / parts); };
    auto run = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (unsigned p = 1; p < parts; ++p) pool.emplace_back(body, p);
        body(0u);
        for (auto& th : pool) th.join();
    };
    run([&](unsigned p) {
        std::size_t cnt = 0;
        for (std::size_t w = range(p).first; w < range(p).second; ++w) cnt += (std::size_t)__builtin_popcountll(mask[w]);
        first[p + 1] = cnt;
    });
    for (unsigned p = 0; p < parts; ++p) first[p + 1] += first[p];
    kept->resize(first[parts]);
    run([&](unsigned p) {
        std::size_t* out = kept->data() + first[p];
        for (std::size_t w = range(p).first; w < range(p).second; ++w) {
            std::uint64_t bits = mask[w];
            while (bits != 0) {
                *out++ = w * 64 + static_cast<std::size_t>(__builtin_ctzll(bits));
                bits &= bits - 1;
            }
        }
    });
    const auto t2 = std::chrono::steady_clock::now();
    ms_expand_ = std::chrono::duration<float, std::milli>(t2 - t1).count();
    ms_solve_call_ = std::chrono::duration<float, std::milli>(t2 - t0).count();
    return kept;
}

}  // namespace qmcp
