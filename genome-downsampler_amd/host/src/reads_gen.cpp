// reads-gen restatement (reference: libs/reads-gen/src/reads_gen.cpp:5-86).
//
// Draw order per pair, which the golden hashes pin: position of mate 1, position of mate 2,
// quality of mate 1, quality of mate 2 (reads_gen.cpp:31-32,45-48 and :66-67,79-82; the
// quality draws are evaluated as constructor arguments of the two push_back calls in
// source order).  Pair shaping:
//   uniform  (:56-84)  first ~ U[0, L-2len], second ~ U[0, L-len]; order them; if they
//                      overlap, second = first + len.
//   weighted (:5-52)   both ~ discrete(dist_func(i/(S-1))), S = L-len+1, negatives clamped
//                      to 0 and weights normalised by their sum; order them; if both lie in
//                      the last 2*len bases they are pinned to (L-2len, L-len); else if they
//                      overlap, second = first + len.
#include "reads_gen.hpp"

#include <utility>
#include <vector>

namespace reads_gen {
namespace {

struct PairSink {
    std::uint32_t* starts;
    std::uint32_t* ends;
    std::uint32_t* quals;
    void put(std::uint64_t i, std::uint64_t s, std::uint32_t len, std::uint32_t q) const {
        starts[i] = static_cast<std::uint32_t>(s);
        ends[i] = static_cast<std::uint32_t>(s + len - 1);
        if (quals) quals[i] = q;
    }
};

template <class Emit>
void gen_uniform(std::mt19937& gen, std::uint64_t pairs, std::uint64_t L, std::uint32_t len,
                 std::int32_t max_q, Emit&& emit) {
    std::uniform_int_distribution<> pos_first(0, static_cast<std::int32_t>(L - 2 * len));
    std::uniform_int_distribution<> pos_second(0, static_cast<std::int32_t>(L - len));
    std::uniform_int_distribution<> qual(0, max_q);
    for (std::uint64_t q = 0; q < pairs; ++q) {
        std::uint64_t a = pos_first(gen);
        std::uint64_t b = pos_second(gen);
        if (a > b) std::swap(a, b);
        if (a + len > b) b = a + len;
        const std::uint32_t qa = qual(gen);
        const std::uint32_t qb = qual(gen);
        emit(2 * q, a, qa, true);
        emit(2 * q + 1, b, qb, false);
    }
}

template <class Emit>
void gen_weighted(std::mt19937& gen, std::uint64_t pairs, std::uint64_t L, std::uint32_t len,
                  const std::function<double(double)>& f, std::int32_t max_q, Emit&& emit) {
    const std::uint32_t n_starts = static_cast<std::uint32_t>(L - len + 1);
    std::vector<double> w(n_starts, 0.0);
    double total = 0;
    for (std::uint32_t i = 0; i < n_starts; ++i) {
        w[i] = f(static_cast<double>(i) / static_cast<double>(n_starts - 1));
        if (w[i] < 0.0) w[i] = 0.0;
        total += w[i];
    }
    for (std::uint32_t i = 0; i < n_starts; ++i) w[i] /= total;
    std::discrete_distribution<> pos(w.begin(), w.end());
    std::uniform_int_distribution<> qual(0, max_q);
    const std::uint64_t tail = L - 2ull * len;
    for (std::uint64_t q = 0; q < pairs; ++q) {
        std::uint64_t a = pos(gen);
        std::uint64_t b = pos(gen);
        if (a > b) std::swap(a, b);
        if (a > tail && b > tail) {
            a = tail;
            b = L - len;
        } else if (a + len > b) {
            b = a + len;
        }
        const std::uint32_t qa = qual(gen);
        const std::uint32_t qb = qual(gen);
        emit(2 * q, a, qa, true);
        emit(2 * q + 1, b, qb, false);
    }
}

}  // namespace

bam_api::AOSPairedReads rand_reads(std::mt19937& generator, bam_api::ReadIndex pairs_count,
                                   bam_api::Index genome_length, std::uint32_t read_length,
                                   const std::function<double(double)>& dist_func,
                                   std::int32_t max_quality) {
    bam_api::AOSPairedReads out;
    out.ref_genome_length = genome_length;
    out.reserve(2 * pairs_count);
    gen_weighted(generator, pairs_count, genome_length, read_length, dist_func, max_quality,
                 [&](std::uint64_t id, std::uint64_t s, std::uint32_t q, bool first) {
                     out.push_back(bam_api::Read(id, s, s + read_length - 1, q, read_length, first));
                 });
    return out;
}

bam_api::AOSPairedReads rand_reads_uniform(std::mt19937& generator,
                                           bam_api::ReadIndex pairs_count,
                                           bam_api::Index genome_length,
                                           std::uint32_t read_length, std::int32_t max_quality) {
    bam_api::AOSPairedReads out;
    out.ref_genome_length = genome_length;
    out.reserve(2 * pairs_count);
    gen_uniform(generator, pairs_count, genome_length, read_length, max_quality,
                [&](std::uint64_t id, std::uint64_t s, std::uint32_t q, bool first) {
                    out.push_back(bam_api::Read(id, s, s + read_length - 1, q, read_length, first));
                });
    return out;
}

void rand_reads_uniform_soa(std::mt19937& generator, std::uint64_t pairs_count,
                            std::uint32_t genome_length, std::uint32_t read_length,
                            std::uint32_t* starts, std::uint32_t* ends, std::uint32_t* qualities,
                            std::int32_t max_quality) {
    const PairSink sink{starts, ends, qualities};
    gen_uniform(generator, pairs_count, genome_length, read_length, max_quality,
                [&](std::uint64_t id, std::uint64_t s, std::uint32_t q, bool) {
                    sink.put(id, s, read_length, q);
                });
}

void rand_reads_soa(std::mt19937& generator, std::uint64_t pairs_count,
                    std::uint32_t genome_length, std::uint32_t read_length,
                    const std::function<double(double)>& dist_func, std::uint32_t* starts,
                    std::uint32_t* ends, std::uint32_t* qualities, std::int32_t max_quality) {
    const PairSink sink{starts, ends, qualities};
    gen_weighted(generator, pairs_count, genome_length, read_length, dist_func, max_quality,
                 [&](std::uint64_t id, std::uint64_t s, std::uint32_t q, bool) {
                     sink.put(id, s, read_length, q);
                 });
}

}  // namespace reads_gen
