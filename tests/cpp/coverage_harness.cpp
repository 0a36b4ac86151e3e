// CoverageTester restatement: drives solvers through the C++ plugin surface exactly as the
// reference's `test` subcommand does (src/test_command.cpp:31-70,
// src/tests/coverage_tester.cpp:28-43,109-175): five inputs, one assertion --
// min(in_cover, M) <= out_cover -- for every registered solver.  Unlike the reference (whose
// asserts are compiled out in Release builds) a violation here is a non-zero exit code.
// Additionally registers the CPU oracle under the same qmcp::Solver interface as
// "oracle-quasi-mcp" (test-only; the product's SolverManager never links the oracle) and
// requires the two kept sets to be identical.
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "bam-api/bam_api.hpp"
#include "qmcp_oracle.h"
#include "reads_gen.hpp"
#include "solver_manager.hpp"

namespace {

class OracleSolver : public qmcp::Solver {
   public:
    std::unique_ptr<qmcp::Solution> solve(std::uint32_t m, bam_api::BamApi& api) override {
        const bam_api::SOAPairedReads& r = api.get_paired_reads_soa();
        const std::size_t n = r.start_inds.size();
        std::vector<std::uint32_t> s(n), e(n);
        for (std::size_t i = 0; i < n; ++i) {
            s[i] = static_cast<std::uint32_t>(r.start_inds[i]);
            e[i] = static_cast<std::uint32_t>(r.end_inds[i]);
        }
        std::vector<std::uint64_t> mask((n + 63) / 64 + 1, 0);
        const std::uint64_t offs[2] = {0, n};
        const std::uint32_t len = static_cast<std::uint32_t>(r.ref_genome_length);
        if (qmcp_oracle_solve(s.data(), e.data(), n, offs, &len, 1, m, mask.data()) != 0) std::abort();
        auto out = std::make_unique<qmcp::Solution>();
        for (std::size_t i = 0; i < n; ++i)
            if ((mask[i >> 6] >> (i & 63)) & 1ull) out->push_back(i);
        return out;
    }
    bool uses_quality_of_reads() override { return false; }
};

bam_api::AOSPairedReads small_example() {
    const unsigned se[16][2] = {{0, 2}, {6, 9}, {2, 4}, {6, 8}, {1, 3}, {7, 10}, {3, 6}, {9, 10},
                                {0, 4}, {7, 9}, {4, 6}, {9, 10}, {1, 4}, {6, 8}, {0, 2}, {4, 6}};
    bam_api::AOSPairedReads r;
    r.ref_genome_length = 11;
    for (unsigned i = 0; i < 16; ++i)
        r.push_back(bam_api::Read(i, se[i][0], se[i][1], 0, se[i][1] - se[i][0] + 1, i % 2 == 0));
    return r;
}

bool is_out_cover_valid(const std::vector<std::uint32_t>& in, const std::vector<std::uint32_t>& out,
                        std::uint32_t m) {
    for (std::size_t p = 0; p < in.size(); ++p)
        if (!(std::min(in[p], m) <= out[p])) return false;
    return true;
}

struct Case {
    const char* name;
    std::function<bam_api::AOSPairedReads()> make;
    std::uint32_t m;
};

}  // namespace

int main(int argc, char** argv) {
    SolverManager manager;
    manager.add("oracle-quasi-mcp", std::make_unique<OracleSolver>());
    const bool small_only = argc > 1 && std::string(argv[1]) == "--small";

    auto weighted = [](const std::function<double(double)>& f) {
        return [f]() {
            std::mt19937 mt(12345);
            return reads_gen::rand_reads(mt, 1'000'000, 30'000, 150, f);
        };
    };
    std::vector<Case> cases = {{"small_example_test", small_example, 4}};
    if (!small_only) {
        cases.push_back({"random_uniform_dist_test", []() {
                             std::mt19937 mt(12345);
                             return reads_gen::rand_reads_uniform(mt, 1'000'000, 30'000, 150);
                         }, 1000});
        cases.push_back({"random_low_coverage_on_both_sides_test",
                         weighted([](double x) { return x - x * x; }), 8000});
        cases.push_back({"random_with_hole_test", weighted([](double x) {
                             if (x > 0.3684 && x < 0.6316)
                                 return 1000.0 * (x * x - x + 0.25) * (x * x - x + 0.25) + 0.2;
                             return 0.5;
                         }), 8000});
        cases.push_back({"random_zero_coverage_on_both_sides_test",
                         weighted([](double x) { return -10.0 * (x - 0.5) * (x - 0.5) + 1.0; }), 8000});
    }

    int failures = 0;
    for (const Case& c : cases) {
        const bam_api::AOSPairedReads input = c.make();
        std::vector<std::vector<bam_api::ReadIndex>> kept;
        for (const std::string& name : manager.get_names()) {
            bam_api::BamApi api(input);
            const auto in_cover = api.find_input_cover();
            auto ids = manager.get(name).solve(c.m, api);
            const auto out_cover = api.find_filtered_cover(*ids);
            const bool ok = is_out_cover_valid(in_cover, out_cover, c.m);
            std::printf("%-44s %-18s kept %8zu of %8zu  %s\n", c.name, name.c_str(), ids->size(),
                        input.reads.size(), ok ? "PASSED" : "FAILED");
            if (!ok) ++failures;
            kept.push_back(*ids);
        }
        for (std::size_t i = 1; i < kept.size(); ++i)
            if (kept[i] != kept[0]) {
                std::printf("%-44s kept sets of %s and %s differ\n", c.name,
                            manager.get_names()[0].c_str(), manager.get_names()[i].c_str());
                ++failures;
            }
    }
    return failures == 0 ? 0 : 1;
}
