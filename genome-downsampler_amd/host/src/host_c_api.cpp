// C entry points of libqmcp_host.so for ctypes callers (tests, bench.py): the reads-gen
// restatement and the C++ solver adapter driven exactly as the reference drives a solver
// (SolverManager::get(name).solve(M, BamApi)).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <random>
#include <string>

#include "bam-api/amplicon_set.hpp"
#include "bam-api/bam_api.hpp"
#include "reads_gen.hpp"
#include "solver_manager.hpp"

namespace {

// Weight functions of the reference's CoverageTester (src/tests/coverage_tester.cpp:157-175).
double low_both_sides(double x) { return x - x * x; }
double with_hole(double x) {
    if (x > 0.3684 && x < 0.6316) return 1000.0 * (x * x - x + 0.25) * (x * x - x + 0.25) + 0.2;
    return 0.5;
}
double zero_both_sides(double x) { return -10.0 * (x - 0.5) * (x - 0.5) + 1.0; }

SolverManager& manager() {
    static SolverManager m;
    return m;
}

}  // namespace

extern "C" {

// kind: 0 uniform, 1 x-x^2, 2 hole, 3 zero-on-both-sides.  Writes 2*pairs reads.
int qmcp_host_reads_gen(std::uint32_t seed, int kind, std::uint64_t pairs,
                        std::uint32_t genome_length, std::uint32_t read_length,
                        std::uint32_t* starts, std::uint32_t* ends, std::uint32_t* qualities) {
    if (genome_length < 2ull * read_length || read_length == 0) return -1;
    std::mt19937 gen(seed);
    switch (kind) {
        case 0:
            reads_gen::rand_reads_uniform_soa(gen, pairs, genome_length, read_length, starts, ends,
                                              qualities);
            return 0;
        case 1:
            reads_gen::rand_reads_soa(gen, pairs, genome_length, read_length, low_both_sides,
                                      starts, ends, qualities);
            return 0;
        case 2:
            reads_gen::rand_reads_soa(gen, pairs, genome_length, read_length, with_hole, starts,
                                      ends, qualities);
            return 0;
        case 3:
            reads_gen::rand_reads_soa(gen, pairs, genome_length, read_length, zero_both_sides,
                                      starts, ends, qualities);
            return 0;
        default:
            return -1;
    }
}

// AoS path of the generator (what the reference's tests call), copied out column-wise;
// lets a test check that the lean SoA variant and the AoS variant agree.
int qmcp_host_reads_gen_aos(std::uint32_t seed, int kind, std::uint64_t pairs,
                            std::uint32_t genome_length, std::uint32_t read_length,
                            std::uint32_t* starts, std::uint32_t* ends, std::uint32_t* qualities) {
    std::mt19937 gen(seed);
    bam_api::AOSPairedReads r;
    if (kind == 0) r = reads_gen::rand_reads_uniform(gen, pairs, genome_length, read_length);
    else if (kind == 1) r = reads_gen::rand_reads(gen, pairs, genome_length, read_length, low_both_sides);
    else if (kind == 2) r = reads_gen::rand_reads(gen, pairs, genome_length, read_length, with_hole);
    else if (kind == 3) r = reads_gen::rand_reads(gen, pairs, genome_length, read_length, zero_both_sides);
    else return -1;
    for (std::size_t i = 0; i < r.reads.size(); ++i) {
        starts[i] = static_cast<std::uint32_t>(r.reads[i].start_ind);
        ends[i] = static_cast<std::uint32_t>(r.reads[i].end_ind);
        if (qualities) qualities[i] = r.reads[i].quality;
        if (r.reads[i].bam_id != i || r.reads[i].is_first_read != (i % 2 == 0)) return -2;
    }
    return 0;
}

// BED (+ optional TSV, may be NULL or "") -> amplicon intervals, as BamApi::set_amplicon_filter
// builds them.  Returns the number of amplicons, -1 if a file cannot be opened, -2 if cap is
// too small.
int qmcp_host_amplicons_from_files(const char* bed_path, const char* tsv_path,
                                   std::uint32_t* starts, std::uint32_t* ends, std::size_t cap) {
    bam_api::AmpliconSet set;
    if (!bam_api::amplicon_set_from_files(bed_path, tsv_path ? tsv_path : "", set)) return -1;
    if (set.amplicons.size() > cap) return -2;
    for (std::size_t i = 0; i < set.amplicons.size(); ++i) {
        starts[i] = static_cast<std::uint32_t>(set.amplicons[i].start);
        ends[i] = static_cast<std::uint32_t>(set.amplicons[i].end);
    }
    return static_cast<int>(set.amplicons.size());
}

// In-memory BamApi helpers as the reference's tests use them (src/tests/coverage_tester.cpp:
// find_input_cover / find_filtered_cover; src/app.cpp:141: find_pairs).  `layout` 0 builds the
// BamApi from an AoS container, 1 from an SoA one (both constructors exist in the reference).
// cover_in / cover_out have ref_genome_length entries; paired_out capacity n; returns the
// number of paired ids.
std::int64_t qmcp_host_bamapi_probe(const std::uint32_t* starts, const std::uint32_t* ends,
                                    std::uint64_t n, std::uint32_t ref_genome_length, int layout,
                                    const std::uint64_t* ids, std::uint64_t n_ids,
                                    std::uint32_t* cover_in, std::uint32_t* cover_out,
                                    std::uint64_t* paired_out) {
    bam_api::AOSPairedReads aos;
    aos.ref_genome_length = ref_genome_length;
    for (std::uint64_t i = 0; i < n; ++i)
        aos.push_back(bam_api::Read(i, starts[i], ends[i], 0, ends[i] - starts[i] + 1, i % 2 == 0));
    bam_api::SOAPairedReads soa;
    soa.from(aos);
    bam_api::BamApi api = layout == 0 ? bam_api::BamApi(aos) : bam_api::BamApi(soa);
    // exercise the lazy layout conversion both ways
    if (api.get_paired_reads_aos().reads.size() != n || api.get_paired_reads_soa().ids.size() != n)
        return -1;
    const std::vector<bam_api::ReadIndex> idv(ids, ids + n_ids);
    const auto in = api.find_input_cover();
    const auto out = api.find_filtered_cover(idv);
    for (std::size_t p = 0; p < in.size(); ++p) { cover_in[p] = in[p]; cover_out[p] = out[p]; }
    const auto paired = api.find_pairs(idv);
    for (std::size_t i = 0; i < paired.size(); ++i) paired_out[i] = paired[i];
    return static_cast<std::int64_t>(paired.size());
}

// Names registered in the SolverManager, '\n'-separated, into buf.
int qmcp_host_solver_names(char* buf, std::size_t cap) {
    std::string all;
    for (const std::string& n : manager().get_names()) { all += n; all += '\n'; }
    if (all.size() + 1 > cap) return -1;
    std::memcpy(buf, all.c_str(), all.size() + 1);
    return static_cast<int>(manager().get_names().size());
}

// Runs SolverManager::get(name).solve(M, BamApi(aos)) the way src/app.cpp:134-135 and
// src/tests/coverage_tester.cpp:109-118 do; returns the number of kept reads, fills
// kept_out (capacity n) with ascending ReadIndex; with_pairs != 0 additionally applies
// BamApi::find_pairs (src/app.cpp:141).  Negative on unknown solver.
std::int64_t qmcp_host_solve(const char* solver_name, const std::uint32_t* starts,
                             const std::uint32_t* ends, std::uint64_t n,
                             std::uint32_t ref_genome_length, std::uint32_t max_coverage,
                             int with_pairs, std::uint64_t* kept_out) {
    if (!manager().contains(solver_name)) return -1;
    bam_api::AOSPairedReads aos;
    aos.ref_genome_length = ref_genome_length;
    aos.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i)
        aos.push_back(bam_api::Read(i, starts[i], ends[i], 0, ends[i] - starts[i] + 1, i % 2 == 0));
    bam_api::BamApi api(aos);
    qmcp::Solver& solver = manager().get(solver_name);
    auto solution = solver.solve(max_coverage, api);
    std::vector<bam_api::ReadIndex> ids = with_pairs ? api.find_pairs(*solution) : *solution;
    for (std::size_t i = 0; i < ids.size(); ++i) kept_out[i] = ids[i];
    return static_cast<std::int64_t>(ids.size());
}

// The span the reference times as "solve took" (src/app.cpp:132-139) at the plugin boundary: a BamApi
// that already holds the reads as SOAPairedReads (size_t columns) -> solver.solve(M, api) -> Solution.
// Building the BamApi is outside the span, as parsing the BAM is in the reference.  `times` receives
// {wall of solve(), library total, narrow + H2D, device solve, D2H, mask expansion, threads, chunks};
// returns the number of kept reads (kept_out may be NULL), negative on an unknown solver.
std::int64_t qmcp_host_plugin_solve_timed(const char* solver_name, const std::uint32_t* starts,
                                          const std::uint32_t* ends, std::uint64_t n,
                                          std::uint32_t ref_genome_length, std::uint32_t max_coverage,
                                          std::uint64_t* kept_out, float* times) {
    if (!manager().contains(solver_name)) return -1;
    bam_api::SOAPairedReads soa;
    soa.ref_genome_length = ref_genome_length;
    soa.reserve(n);
    for (std::uint64_t i = 0; i < n; ++i)
        soa.push_back(bam_api::Read(i, starts[i], ends[i], 0, ends[i] - starts[i] + 1, i % 2 == 0));
    bam_api::BamApi api(soa);
    qmcp::Solver& solver = manager().get(solver_name);
    const auto t0 = std::chrono::steady_clock::now();
    auto solution = solver.solve(max_coverage, api);
    const float wall = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (times) {
        for (int i = 0; i < 8; ++i) times[i] = 0.f;
        times[0] = wall;
        if (auto* hip = dynamic_cast<qmcp::QuasiMcpHipSolver*>(&solver)) {
            const qmcp_hip_host_breakdown& b = hip->last_breakdown();
            times[1] = b.ms_total; times[2] = b.ms_narrow_h2d; times[3] = b.ms_solve; times[4] = b.ms_d2h;
            times[5] = hip->last_expand_ms(); times[6] = (float)b.host_threads; times[7] = (float)b.chunks;
        }
    }
    if (kept_out) for (std::size_t i = 0; i < solution->size(); ++i) kept_out[i] = (*solution)[i];
    return static_cast<std::int64_t>(solution->size());
}

}  // extern "C"
