// Synthetic paired-read generator: the measurement input of the solver path.
// Mirrors libs/reads-gen/include/reads_gen.hpp:10-25 (same namespace, names, argument
// order and defaults).  Bit-exactness with the reference's streams relies on using the same
// library objects it uses: std::mt19937 with libstdc++'s uniform_int_distribution<> and
// discrete_distribution<> (libs/reads-gen/src/reads_gen.cpp:26-27,56-62); pinned by the
// FNV hashes in tests/golden/reads_gen_hashes.json.
#ifndef QMCP_AMD_READS_GEN_HPP
#define QMCP_AMD_READS_GEN_HPP

#include <cstdint>
#include <functional>
#include <random>

#include "bam-api/paired_reads.hpp"

namespace reads_gen {

constexpr std::int32_t kMaxGenQuality = 100;

bam_api::AOSPairedReads rand_reads(std::mt19937& generator, bam_api::ReadIndex pairs_count,
                                   bam_api::Index genome_length, std::uint32_t read_length,
                                   const std::function<double(double)>& dist_func,
                                   std::int32_t max_quality = kMaxGenQuality);

bam_api::AOSPairedReads rand_reads_uniform(std::mt19937& generator,
                                           bam_api::ReadIndex pairs_count,
                                           bam_api::Index genome_length,
                                           std::uint32_t read_length,
                                           std::int32_t max_quality = kMaxGenQuality);

// Lean variants used by the bench and the ctypes bridge: same random stream and the same
// values, written straight into uint32 SoA columns (an AoS of 10^8 reads would be 4 GB).
// `qualities` may be null (the draws still happen so the stream stays aligned).
void rand_reads_uniform_soa(std::mt19937& generator, std::uint64_t pairs_count,
                            std::uint32_t genome_length, std::uint32_t read_length,
                            std::uint32_t* starts, std::uint32_t* ends, std::uint32_t* qualities,
                            std::int32_t max_quality = kMaxGenQuality);
void rand_reads_soa(std::mt19937& generator, std::uint64_t pairs_count,
                    std::uint32_t genome_length, std::uint32_t read_length,
                    const std::function<double(double)>& dist_func, std::uint32_t* starts,
                    std::uint32_t* ends, std::uint32_t* qualities,
                    std::int32_t max_quality = kMaxGenQuality);

}  // namespace reads_gen
#endif
