/*
 * qmcp_oracle.h -- CPU oracle for the quasi-MCP solver path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (include/qmcp_hip.h, genome-downsampler_amd/) never links,
 * imports or executes anything under oracle/.
 *
 * See qmcp_oracle.c for the parity statement and the reference file:line each function follows.
 */
#ifndef QMCP_ORACLE_H
#define QMCP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct qmcp_oracle_graph_info {
    uint64_t n_arcs;          /* N + n + k                                      */
    uint64_t n_terminal_arcs; /* k                                              */
    int64_t total_supply;     /* sum of s->i capacities                         */
    int64_t total_demand;     /* sum of i->t capacities (== total_supply)       */
    uint64_t arc_fnv;         /* FNV-1a-64 over (tail, head, cap) in emission order */
} qmcp_oracle_graph_info;

uint64_t qmcp_oracle_fnv_init(void);
uint64_t qmcp_oracle_fnv_mix(uint64_t h, uint64_t v);
uint64_t qmcp_oracle_reads_fnv(const uint32_t* starts, const uint32_t* ends,
                               const uint32_t* qualities, uint64_t n);
uint64_t qmcp_oracle_mask_fnv(const uint64_t* mask, uint64_t n_reads);

int qmcp_oracle_b_function(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                           uint32_t ref_len, uint32_t M, int32_t* b /* ref_len+1 */);
void qmcp_oracle_demand_function(int32_t* b_inout /* ref_len+1 */, uint32_t ref_len);
int qmcp_oracle_graph(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                      uint32_t ref_len, uint32_t M, qmcp_oracle_graph_info* info,
                      int64_t* arcs_out /* NULL or 3*(n+ref_len+ref_len+1) */);

int qmcp_oracle_select(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                       uint32_t ref_len, uint32_t M, uint64_t* keep_mask, uint64_t mask_bit_base);

int qmcp_oracle_solve(const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                      const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                      uint32_t n_contigs, uint32_t max_coverage, uint64_t* keep_mask_out);

int qmcp_oracle_cover(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                      uint32_t ref_len, const uint64_t* keep_mask /* NULL = all */,
                      uint64_t mask_bit_base, uint32_t* cov_out /* ref_len */);
int qmcp_oracle_is_out_cover_valid(const uint32_t* in_cover, const uint32_t* out_cover,
                                   uint32_t ref_len, uint32_t M);

int qmcp_oracle_check_flow(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                           uint32_t ref_len, uint32_t M, const uint64_t* keep_mask,
                           uint64_t mask_bit_base, int64_t* flow_value_out);
int64_t qmcp_oracle_maxflow_value(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                                  uint32_t ref_len, uint32_t M);

void qmcp_oracle_find_pairs(uint64_t* keep_mask, uint64_t n_reads);
void qmcp_oracle_amplicon_filter(const uint32_t* starts, const uint32_t* ends,
                                 const uint32_t* seq_lengths, const uint32_t* qualities,
                                 uint64_t n_reads, const uint32_t* amp_starts,
                                 const uint32_t* amp_ends, uint32_t n_amplicons,
                                 uint32_t min_length, uint32_t min_mapq, uint64_t* pair_keep_out);
#ifdef __cplusplus
}
#endif
#endif
