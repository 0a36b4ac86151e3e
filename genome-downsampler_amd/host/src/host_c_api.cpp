// C entry points of libqmcp_host.so for ctypes callers (tests, bench.py): the reads-gen
// restatement and the C++ solver adapter driven exactly as the reference drives a solver
// (SolverManager::get(name).solve(M, BamApi)).
#include <chrono>
#include <new>
#include <cstdint>
#include <cstring>
#include <functional>
#include <random>
#include <string>

#include "bam-api/amplicon_set.hpp"
#include "bam-api/bam_api.hpp"
#include "bam-api/bam_io.hpp"
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

// ---- BAM ingest / emit (bam-api/bam_io.hpp) for ctypes callers
// Writes a synthetic single-reference BAM: record i = {qname "q<names[i]>", flags[i], pos[i], mapq[i], CIGAR
// "<clip_front[i]>S<match[i]>M<del[i]>D<match2[i]>M" with zero-length parts left out, l_seq = sum of S and M}.
int qmcp_host_write_synthetic_bam(const char* path, std::uint32_t ref_length, std::uint64_t n,
                                  const std::uint32_t* names, const std::uint16_t* flags,
                                  const std::uint32_t* pos, const std::uint8_t* mapq,
                                  const std::uint32_t* clip_front, const std::uint32_t* match,
                                  const std::uint32_t* del, const std::uint32_t* match2) {
    std::vector<bam_api::BamRecordSpec> recs(n);
    for (std::uint64_t i = 0; i < n; ++i) {
        bam_api::BamRecordSpec& r = recs[i];
        r.qname = "q" + std::to_string(names[i]);
        r.flag = flags[i];
        r.pos = (std::int32_t)pos[i];
        r.mapq = mapq[i];
        if (clip_front[i]) r.cigar.push_back({clip_front[i], 'S'});
        if (match[i]) r.cigar.push_back({match[i], 'M'});
        if (del[i]) r.cigar.push_back({del[i], 'D'});
        if (match2[i]) r.cigar.push_back({match2[i], 'M'});
        r.l_seq = clip_front[i] + match[i] + match2[i];
    }
    std::string err;
    return bam_api::write_synthetic_bam(path, "ref1", ref_length, recs, &err) ? 0 : -1;
}

// BamApi(path, config) -> get_paired_reads_soa(): columns out (capacity cap reads); returns the number of
// imported reads, *n_filtered_out = size of get_filtered_out_reads() (ids into filtered_out, capacity cap_f),
// *ref_len = ref_genome_length.  amplicon_mode: 0 IGNORE, 1 FILTER, 2 GRADE (bed/tsv may be NULL or "").
std::int64_t qmcp_host_read_bam(const char* path, const char* bed, const char* tsv, int amplicon_mode,
                                std::uint32_t min_len, std::uint32_t min_mapq, std::uint64_t cap,
                                std::uint64_t* bam_ids, std::uint32_t* starts, std::uint32_t* ends,
                                std::uint32_t* qualities, std::uint32_t* seq_lengths, std::uint8_t* is_first,
                                std::uint64_t cap_f, std::uint64_t* filtered_out, std::uint64_t* n_filtered_out,
                                std::uint32_t* ref_len) {
    bam_api::BamApiConfig cfg;
    if (bed && bed[0]) cfg.bed_filepath = bed;
    if (tsv && tsv[0]) cfg.tsv_filepath = tsv;
    cfg.min_seq_length = min_len;
    cfg.min_mapq = min_mapq;
    cfg.amplicon_behaviour = amplicon_mode == 1 ? bam_api::AmpliconBehaviour::FILTER
                           : amplicon_mode == 2 ? bam_api::AmpliconBehaviour::GRADE : bam_api::AmpliconBehaviour::IGNORE;
    try {
        bam_api::BamApi api(path, cfg);
        const bam_api::SOAPairedReads& r = api.get_paired_reads_soa();
        const std::uint64_t n = r.ids.size();
        if (n > cap || api.get_filtered_out_reads().size() > cap_f) return -2;
        for (std::uint64_t i = 0; i < n; ++i) {
            bam_ids[i] = r.ids[i]; starts[i] = (std::uint32_t)r.start_inds[i]; ends[i] = (std::uint32_t)r.end_inds[i];
            qualities[i] = r.qualities[i]; seq_lengths[i] = r.seq_lengths[i]; is_first[i] = r.is_first_reads[i] ? 1 : 0;
        }
        *n_filtered_out = api.get_filtered_out_reads().size();
        for (std::size_t i = 0; i < api.get_filtered_out_reads().size(); ++i) filtered_out[i] = api.get_filtered_out_reads()[i];
        *ref_len = (std::uint32_t)r.ref_genome_length;
        return (std::int64_t)n;
    } catch (const std::bad_alloc&) {
        return -3;  // (nothing is thrown through the C boundary)
    }
}

// read_bam's own verdict on a file, without BamApi's exit-on-error (the reference exits the process on an
// unreadable input; a caller that wants to look first -- and the tests of corrupt files -- use this): 0 and the
// number of imported reads in *n_reads, or -1 and the reader's message in err (capacity cap).  Allocation
// failures on absurd sizes are reported the same way, never thrown through the C boundary.
int qmcp_host_check_bam(const char* path, std::uint64_t* n_reads, char* err, std::size_t cap) {
    std::string msg;
    int rc = -1;
    try {
        bam_api::SOAPairedReads reads;
        std::vector<bam_api::BAMReadId> filtered;
        bam_api::BamFilters f;
        if (bam_api::read_bam(path, f, reads, filtered, nullptr, &msg)) {
            if (n_reads) *n_reads = reads.ids.size();
            rc = 0;
        }
    } catch (const std::bad_alloc&) {
        msg = "out of memory while reading the BAM file";
    } catch (const std::exception& e) {
        msg = e.what();
    }
    if (err && cap) {
        std::strncpy(err, msg.c_str(), cap - 1);
        err[cap - 1] = 0;
    }
    return rc;
}

// BamApi::write_bam alone (bam_api.cpp:534-656): the header and the records whose running id is in `ids` (n of them;
// sorted here, as the reference sorts them) copied to `out_path` -- BAM if it ends in ".bam", SAM text otherwise.
// Returns the number of records written, -1 on failure (message in err, capacity cap).
std::int64_t qmcp_host_copy_records(const char* in_path, const char* out_path, const std::uint64_t* ids, std::uint64_t n,
                                    char* err, std::size_t cap) {
    std::string msg;
    std::int64_t rc = -1;
    try {
        std::vector<bam_api::BAMReadId> v(ids, ids + n);
        const std::uint32_t written = bam_api::write_bam(in_path, out_path, v, &msg);
        if (written != UINT32_MAX) rc = (std::int64_t)written;
    } catch (const std::exception& e) {
        msg = e.what();
    }
    if (err && cap) {
        std::strncpy(err, msg.c_str(), cap - 1);
        err[cap - 1] = 0;
    }
    return rc;
}

// The file-to-file flow of App::execute (src/app.cpp:113-151) for one solver: BamApi(path) -> solve ->
// find_pairs -> write_paired_reads(out) (+ write_bam_api_filtered_out_reads(filtered) if given).
// Returns the number of records written to `out_path`, negative on an unknown solver.
std::int64_t qmcp_host_downsample_bam(const char* solver_name, const char* in_path, const char* out_path,
                                      const char* filtered_path, std::uint32_t max_coverage,
                                      std::uint32_t min_len, std::uint32_t min_mapq) {
    if (!manager().contains(solver_name)) return -1;
    bam_api::BamApiConfig cfg;
    cfg.min_seq_length = min_len;
    cfg.min_mapq = min_mapq;
    try {
        bam_api::BamApi api(in_path, cfg);
        auto solution = manager().get(solver_name).solve(max_coverage, api);
        std::vector<bam_api::ReadIndex> paired = api.find_pairs(*solution);
        const std::uint32_t written = api.write_paired_reads(out_path, paired);
        if (filtered_path && filtered_path[0]) api.write_bam_api_filtered_out_reads(filtered_path);
        return written;
    } catch (const std::bad_alloc&) {
        return -3;
    }
}

// The span the reference times as "solve took" (src/app.cpp:132-139) at the plugin boundary: a BamApi
// that already holds the reads as SOAPairedReads (size_t columns) -> solver.solve(M, api) -> Solution.
// Building the BamApi is outside the span, as parsing the BAM is in the reference.  `times` receives
// {wall of solve(), library total, narrow + H2D, device solve, D2H, mask expansion, threads, chunks,
// columns sent};
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
        for (int i = 0; i < 9; ++i) times[i] = 0.f;
        times[0] = wall;
        if (auto* hip = dynamic_cast<qmcp::QuasiMcpHipSolver*>(&solver)) {
            const qmcp_hip_host_breakdown& b = hip->last_breakdown();
            times[1] = b.ms_total; times[2] = b.ms_narrow_h2d; times[3] = b.ms_solve; times[4] = b.ms_d2h;
            times[5] = hip->last_expand_ms(); times[6] = (float)b.host_threads; times[7] = (float)b.chunks;
            times[8] = (float)b.columns_sent;
        }
    }
    if (kept_out) for (std::size_t i = 0; i < solution->size(); ++i) kept_out[i] = (*solution)[i];
    return static_cast<std::int64_t>(solution->size());
}

}  // extern "C"
