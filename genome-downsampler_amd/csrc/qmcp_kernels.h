// Launchers of the gfx950 kernels in qmcp_kernels.hip (internal to the library; the public
// surface is include/qmcp_hip.h).
#ifndef QMCP_KERNELS_H
#define QMCP_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qmcp {

static constexpr uint32_t kMaxLdsRingSpan = 16383;   // two LDS rings of 16384 u32 = 128 KiB
static constexpr uint32_t kMaxGeneralSpan = (1u << 24) - 1;  // beyond kMaxLdsRingSpan the rings live in global memory
static constexpr uint32_t kMaxUniformSpan = 512;    // 8 positions per lane in the block sweep
static constexpr uint32_t kMaxCachedSpan = 4032;    // LDS-cached mixed-span sweep: 8 words x 4096 slots,
                                                    // ring >= max_span + 64 (a chunk enters 64 buckets at once)

void launch_prepare(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                    const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                    const uint64_t* keep_mask, uint32_t* gstart, uint32_t* cstart,
                    uint32_t* stats, uint32_t part_shift,
                    uint32_t* part_hist /* digit-major [256][tiles] of (gstart >> part_shift), or null */,
                    uint32_t* digit0_hist /* same shape, low byte of gstart; needs part_hist; or null */,
                    uint32_t* global_digit_hist /* [4][256] whole-call digit counts of gstart, or null */,
                    unsigned long long* zero_mask /* keep mask to clear (ceil(n/64) words), or null */,
                    uint32_t ell_reg = 0, uint32_t* exc = nullptr, uint32_t exc_cap = 0, uint32_t* exc_cnt = nullptr
                    /* near-uniform route on the range-major form: see k_prepare */);
uint32_t prepare_exc_slots(uint32_t n);  // slots of the exception list when k_prepare fills it (groups of 128 per wave and tile)
void launch_general_keys(hipStream_t st, bool wide, const uint32_t* gstart, const uint32_t* starts,
                         const uint32_t* ends, uint32_t n, uint32_t span_bits, uint32_t max_span,
                         const uint64_t* keep_mask, void* keys, uint32_t* ecnt,
                         uint32_t ecnt_len /* entries of ecnt (positions + 1), 0: unknown */);
uint32_t scan_spine_entries(uint32_t n);
void launch_exclusive_scan(hipStream_t st, const uint32_t* in, uint32_t n, uint32_t* out,
                           uint32_t* spine, bool write_total);
uint32_t sort_tiles(uint32_t n);
void launch_radix_hist(hipStream_t st, bool wide, const void* keys_in, uint32_t n, uint32_t shift,
                       uint32_t* hist);
void launch_radix_scatter(hipStream_t st, bool wide, const void* keys_in, const uint32_t* vals_in,
                          uint32_t n, uint32_t shift, const uint32_t* offs, void* keys_out,
                          uint32_t* vals_out);
// Cut points (coverage <= M) split a contig's sweep exactly.  sweep_segment_windows: how many
// windows to look for cuts in (0: not worth it); launch_sweep_segments fills `seg_words`
// (sweep_segment_words() uint32) and returns the stretch table the sweep launchers take as `seg`
// together with n_seg_max = n_contigs + n_windows workgroups (null: one workgroup per contig).
// row pitch, in partition passes (pairs of 4096-read tiles), of the range partition's [digit][pass]
// table (k_prepare writes it, one plain scan over 256 * pitch entries turns it into offsets, the
// partition reads it)
uint32_t part_pass_pitch(uint32_t n);
// (at most max_windows of them; the table kernels take up to 3840: kMaxSweepWindows)
constexpr uint32_t kSweepWindowsOneSpan = 768, kMaxSweepWindows = 3840;
uint32_t sweep_segment_windows(uint32_t ltot, uint32_t ell, uint32_t n_contigs, uint32_t max_windows = kSweepWindowsOneSpan);
size_t sweep_segment_words(uint32_t n_contigs, uint32_t n_windows);
// eoff: prefix counts of read ends for mixed spans (coverage = starts - ends); null for one span ell
const uint32_t* launch_sweep_segments(hipStream_t st, const uint32_t* boff, const uint32_t* eoff,
                                      const uint64_t* d_poff, uint32_t n_contigs, uint32_t ltot, uint32_t ell,
                                      uint32_t M, uint32_t n_windows, uint32_t* seg_words,
                                      const uint32_t* other_cov = nullptr /* reads outside boff covering position q - 1:
                                          other_cov[q] (the near-uniform route's exceptions) */);
// seven-wave pipelined forms (spans <= 256); false if the span needs the single-wave kernel
bool sweep_uniform_mw_supported(uint32_t ell);
// the same pipeline with every block in the general form (sparse data: the fast form rarely holds)
// selend_run_in (or null): speculative tables -- a stretch's run-in is stored there, what it owns in selend;
// redo_in (or null): a later tier -- only stretches whose exact stretch is marked there do anything
bool launch_sweep_uniform_gen(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                              uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                              uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg,
                              uint32_t n_seg_max, uint32_t* selend_run_in = nullptr,
                              const uint32_t* redo_in = nullptr,
                              const int32_t* nadj = nullptr /* near-uniform route: need(p) += nadj[p], capped at the
                                                               swept reads' coverage (ltot + 1 entries) */,
                              const uint32_t* own_marks = nullptr /* per stretch of `seg`: 0 = leave its output alone */);
// Speculative boundaries (kernels/sweep_segments.inc.hip): further tables of the same windows (tier 1, 2
// behind the exact one in seg_words), with a boundary `burn` positions of run-in wide wherever a window has no
// cut (call launch_sweep_segments first; *n_speculative receives how many; burn == 0: the exact table again),
// and the check + merge of the two outputs behind a sweep (a disagreement marks its exact stretch in
// redo_out; a later tier passes the previous tier's marks as redo_in).
const uint32_t* launch_sweep_segments_speculative(hipStream_t st, const uint64_t* d_poff, uint32_t n_contigs,
                                                  uint32_t ltot, uint32_t n_windows, uint32_t burn,
                                                  uint32_t* seg_words, uint32_t* n_speculative,
                                                  uint32_t run_ins_apart, uint32_t tier);
void launch_spec_verify(hipStream_t st, const uint32_t* seg, uint32_t n_cand, uint32_t ell,
                        const uint32_t* owned, const uint32_t* run_in, uint32_t* mismatches,
                        const uint32_t* redo_in, uint32_t* redo_out,
                        const uint32_t* own_marks = nullptr /* per stretch of `seg`: which were swept this time */);
bool launch_sweep_uniform_mw(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                             uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                             uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg,
                             uint32_t n_seg_max);
// event-driven form for deep data (kernels/sweep_uniform_events.inc.hip): pack, chain, expand.
// pk / lastns are scratch of sweep_ev_pack_bytes / sweep_ev_last_bytes; sev has ltot + 8 words.
bool sweep_uniform_ev_supported(uint32_t ell, uint32_t M);
size_t sweep_ev_pack_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg);
size_t sweep_ev_last_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg);
bool launch_sweep_ev_pack(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                          uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                          uint32_t* pk, const int32_t* nadj = nullptr, const uint32_t* from = nullptr);
bool launch_sweep_ev_chain(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                           uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                           const uint32_t* pk, uint32_t* sev, uint32_t* lastns, uint32_t* iter_stats,
                           const int32_t* nadj = nullptr, uint32_t* ckpt = nullptr /* sweep_ev_ckpt_bytes */,
                           const uint32_t* restart = nullptr /* per stretch: first block to sweep (multiple of 64) */);
size_t sweep_ev_ckpt_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg);
bool launch_sweep_ev_expand(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                            uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                            const uint32_t* sev, const uint32_t* lastns, uint32_t* selend, const uint32_t* from = nullptr);
bool launch_sweep_uniform(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                          uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                          uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg,
                          uint32_t n_seg_max);
void launch_sweep_general(hipStream_t st, bool wide, const uint32_t* boff, const uint32_t* eoff,
                          const void* skeys, const uint64_t* d_poff, uint32_t n_contigs,
                          uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* selend,
                          uint32_t ring_size, const uint32_t* seg, uint32_t n_seg_max,
                          uint32_t* g_rings /* 2 * ring_size words per workgroup when the rings do not fit LDS, else null */);
void launch_group_heads(hipStream_t st, bool wide, const void* sorted, uint32_t n,
                        uint32_t* next_head /* n + 1 entries; reverse-min-scan it afterwards */);
void launch_sweep_general_cached(hipStream_t st, bool wide, const uint32_t* boff,
                                 const uint32_t* eoff, const void* sorted, const uint32_t* next_head,
                                 const uint64_t* d_poff, uint32_t n_contigs, uint32_t span_bits,
                                 uint32_t max_span, uint32_t M, uint32_t* selend, uint32_t ring,
                                 const uint32_t* seg, uint32_t n_seg_max);
// `sorted` is a Rec{key,val} array (wide == false) or u64 keys with `svals` beside them
bool launch_sweep_general_reg(hipStream_t st, bool wide, const uint32_t* boff, const uint32_t* eoff,
                              const void* sorted, const uint32_t* next_head, const uint64_t* d_poff,
                              uint32_t n_contigs, uint32_t span_bits, uint32_t max_span, uint32_t M,
                              uint32_t* selend, const uint32_t* seg, uint32_t n_seg_max,
                              uint32_t* selend_odd = nullptr, const uint32_t* redo_in = nullptr,
                              uint32_t* snap = nullptr /* speculative tables: spec_snap_bytes(n_seg_max) */);
size_t spec_snap_bytes(uint32_t n_cand);
void launch_spec_verify_merge_mixed(hipStream_t st, const uint32_t* seg, uint32_t n_cand, uint32_t max_span,
                                    uint32_t* out_even, const uint32_t* out_odd, const uint32_t* snap,
                                    uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out);
void launch_mark(hipStream_t st, bool wide, const void* sorted, const uint32_t* svals, uint32_t ltot,
                 const uint32_t* boff, const uint32_t* selend, uint64_t* mask,
                 unsigned long long* n_kept);
void launch_bucket_heads(hipStream_t st, bool wide, const void* sorted, const uint32_t* svals,
                         uint32_t n, uint32_t span_bits, uint32_t ltot, uint32_t* boff);
void launch_reverse_min_scan(hipStream_t st, uint32_t* data, uint32_t n, uint32_t* spine);
void launch_radix_hist_rec(hipStream_t st, bool first, const uint32_t* keys, const void* recs,
                           uint32_t n, uint32_t shift, uint32_t* hist);
void launch_radix_scatter_rec(hipStream_t st, bool first, const uint32_t* keys, const void* recs_in,
                              uint32_t n, uint32_t shift, const uint32_t* offs, void* recs_out);
void launch_coverage(hipStream_t st, const uint32_t* boff, const uint32_t* eoff, uint32_t ltot,
                     uint32_t* cov);
void launch_b_and_demand(hipStream_t st, const uint32_t* cov, uint32_t n, uint32_t M, int32_t* b, int32_t* d);
void launch_complete_pairs(hipStream_t st, uint64_t* mask, uint32_t n_words, uint64_t n_reads);
void launch_word_popcounts(hipStream_t st, const uint64_t* words, uint32_t n_words, uint32_t* counts);
void launch_mask_to_indices(hipStream_t st, const uint64_t* mask, uint32_t n_words, const uint32_t* word_base,
                            unsigned long long* out);
void launch_compact_pairs(hipStream_t st, const uint32_t* starts, const uint32_t* ends,
                          const uint64_t* pair_keep, const uint32_t* word_base, uint64_t n_pairs,
                          uint32_t* starts_c, uint32_t* ends_c, uint32_t* orig_pair);
void launch_expand_mask(hipStream_t st, const uint64_t* mask_c, const uint32_t* orig_pair,
                        uint32_t n_reads_c, uint64_t* mask);
void launch_amplicon_filter(hipStream_t st, const uint32_t* starts, const uint32_t* ends,
                            const uint32_t* seq_lengths, const uint32_t* qualities,
                            uint64_t n_pairs, const uint32_t* amp_starts, const uint32_t* amp_ends,
                            uint32_t n_amp, uint32_t min_length, uint32_t min_mapq,
                            uint64_t* pair_keep);

// range-ranked uniform path: one stable partition of {start, index} records by position range,
// then per-range LDS histograms (counts) and per-range ordered ranking against S(p) (keep mask)
uint32_t range_shift_for(uint32_t ltot);
bool range_path_supported(uint32_t ltot);
bool range_path_two_level(uint32_t ltot);  // more than 256 ranges: two partition levels
uint32_t seg_tile_bound(uint32_t n);
void launch_partition_level1(hipStream_t st, const uint32_t* starts, const uint64_t* d_roff,
                             const uint64_t* d_poff, uint32_t n_contigs, uint32_t n, uint32_t shift_hi,
                             const uint32_t* offs, void* recs_out, uint32_t* super_start,
                             uint32_t* max_super_load, const uint32_t* ends = nullptr, uint32_t ell_reg = 0);
void launch_partition_level2(hipStream_t st, const void* recs_in, uint32_t n, uint32_t shift,
                             uint32_t* tables /* 771 words */, uint32_t* hist, uint32_t* spine,
                             uint16_t* keys16_out, uint32_t* idx_out, uint32_t* range_start /* 65537 */,
                             uint32_t* max_load);
void launch_range_partition(hipStream_t st, const uint32_t* gstart_or_null, const uint32_t* starts,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                            uint32_t n, uint32_t shift, const uint32_t* offs, uint16_t* keys16_out,
                            uint32_t* idx_out, uint32_t* range_start /* [257] */, uint32_t* max_load,
                            const uint32_t* ends = nullptr, uint32_t ell_reg = 0 /* near-uniform route: leave out other spans */);
void launch_gstart(hipStream_t st, const uint32_t* starts, uint32_t n, const uint64_t* d_roff,
                   const uint64_t* d_poff, uint32_t n_contigs, uint32_t* gstart);
void launch_fill_ends(hipStream_t st, const uint32_t* starts, uint32_t n, uint32_t span_minus_1, uint32_t* ends);
void launch_range_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* range_start,
                          uint32_t shift, uint32_t ltot, uint32_t* boff,
                          uint32_t* empty_positions /* zeroed counter: positions that start no read; or null */);
size_t rank_scratch_bytes(uint32_t shift, uint32_t ltot, uint32_t n);  // list slots: per position or per read
bool rank_scratch_by_records(uint32_t shift, uint32_t ltot, uint32_t n);
void launch_rank_mark(hipStream_t st, const uint16_t* keys16, const uint32_t* idx,
                      const uint32_t* range_start, uint32_t shift, uint32_t ltot, const uint32_t* boff,
                      const uint32_t* selend, unsigned long long* mask, unsigned long long* kept_total,
                      void* scratch, bool scratch_by_records);

// range-ranked route, pass-major layout (kernels/pass_major.inc.hip): one-level genomes
uint32_t pm_pitch(uint32_t n);     // row pitch of the [range][pass] tables
uint32_t pm_pass();                // reads per pass
uint32_t pm_stride(uint32_t ltot, uint32_t shift);              // slots between the beginnings of two passes
size_t pm_slots(uint32_t n, uint32_t ltot, uint32_t shift);     // slots of the two 16-bit record streams
uint32_t pm_work_words();          // words of k_pm_descr's working buffer
void launch_pm_prepare_sort(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs, uint32_t shift, uint32_t ltot,
                            uint16_t* keys16, uint16_t* idx16, uint32_t* cnt_tab, uint32_t* lst_tab,
                            uint32_t* work /* pm_work_words() */,
                            uint32_t* stats, unsigned long long* zero_mask, uint32_t ell_reg = 0, uint32_t* exc = nullptr,
                            uint32_t exc_cap = 0, uint32_t* exc_cnt = nullptr);
uint32_t pm_exc_slots(uint32_t n);  // slots of the near-uniform route's exception list (groups of 64, one per wave and pass)
// wave-slot descriptors (pm_slots / 64 words), the ranges' true flat starts (257 words), heaviest load
void launch_pm_descr(hipStream_t st, const uint32_t* Tp, const uint32_t* lstw, uint32_t n, uint32_t ltot, uint32_t shift,
                     uint32_t* desc, uint32_t* work, uint32_t* range_start, uint32_t* max_load);
void launch_pm_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* desc, const uint32_t* Tp, uint32_t n,
                       const uint32_t* range_start, uint32_t shift, uint32_t ltot, uint32_t* boff, uint32_t* empty_positions);
// the ranking: ordered walk (kept reads marked 64 at a time), then the settling of the quota-crossing groups it listed --
// two launches, in this order on one stream.  amb_count: 256 words
void launch_pm_walk(hipStream_t st, const uint16_t* keys16, const uint16_t* idx16, const uint32_t* desc, const uint32_t* Tp,
                    uint32_t n, const uint32_t* range_start, uint32_t shift, uint32_t ltot, const uint32_t* boff,
                    const uint32_t* selend, unsigned long long* mask, unsigned long long* kept_total, void* scratch,
                    bool scratch_by_records, uint32_t* amb_count,
                    // quotas straight from the event-driven sweep's output (whole contigs, no stretch table), instead
                    // of selend[] - boff[]: the changed blocks' kept counts, the last changed block per block
                    const uint32_t* ev_sev = nullptr, const uint32_t* ev_lastns = nullptr,
                    const uint64_t* d_poff = nullptr, uint32_t n_contigs = 0, uint32_t ell = 0);
void launch_pm_settle(hipStream_t st, const uint16_t* keys16, const uint16_t* idx16, const uint32_t* desc, const uint32_t* Tp,
                      uint32_t n, const uint32_t* range_start, uint32_t shift, uint32_t ltot, const void* scratch,
                      bool scratch_by_records, const uint32_t* amb_count, unsigned long long* mask,
                      unsigned long long* kept_total);


// near-uniform route (kernels/near_uniform.inc.hip): one dominant span, a few shorter reads as listed exceptions
void launch_span_mode_share(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n, uint32_t* out /* 3 words: sampled, in the fullest bin, its span */);
size_t nu_exc_bytes(uint32_t cap);
uint32_t* nu_exc_counts(uint32_t* exc, uint32_t cap);  // the groups' counts inside the list's buffer  // the exception list: start, end, index (k_pm_prepare_sort), selection time, event key, group counts
void launch_nu_count_span(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n, uint32_t span, uint32_t* out);
void launch_nu_setup(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over /* the producer's stats[6] */,
                     const uint32_t* boff, uint32_t ltot, uint32_t ell, uint32_t M, uint32_t* ce /* ltot + 3 words */,
                     uint32_t* spine, int32_t* nadj /* ltot + 1 */, uint32_t* state /* 8 words */);
void launch_nu_round(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over, bool first_round,
                     const uint32_t* boff, const uint32_t* selend, int32_t* nadj, const uint32_t* ce /* launch_nu_setup's */,
                     const uint64_t* d_poff, uint32_t n_contigs,
                     uint32_t ell, uint32_t M, uint2* suspects, uint32_t suspects_cap, uint32_t* state,
                     unsigned long long* viol_key, uint32_t* viol_idx,
                     const uint32_t* swept_from /* per contig: first block the round's sweep covered (0xFFFFFFFF: none) */,
                     uint32_t* sweep_from_next /* per contig, out: where the next round's sweep starts (0xFFFFFFFF: settled) */,
                     uint32_t ltot, uint32_t* spine /* scan_spine_entries(2^17 + 2) words */,
                     const uint32_t* seg_exact /* sweeps in stretches: launch_sweep_segments' table; or null */, uint32_t n_cand,
                     uint32_t* marks_next /* n_cand words, out: the exact stretches the next round's sweep must cover */,
                     uint32_t* selend_prev /* sweeps in stretches: ltot words, the round before's selend (kept here); or null:
                                              every listed exception is replayed */,
                     uint32_t* dirty /* nu_cells_bytes(): cells that changed since the round before (the last round's
                                        dirty_next) */,
                     uint32_t* dirty_next,
                     const uint32_t* seg_fine = nullptr /* sweeps in speculative stretches: the first tier's table */,
                     uint32_t* fine_next = nullptr /* n_cand words, out: the stretches of that table the next round's sweep
                                                      must cover */);
size_t nu_cells_bytes();
size_t nu_suspect_bytes(uint32_t suspects_cap);  // `suspects`: the list and, behind it, its bins by start position
void launch_nu_mark_selected(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over,
                             unsigned long long* mask, unsigned long long* kept_total);

}  // namespace qmcp
#endif
