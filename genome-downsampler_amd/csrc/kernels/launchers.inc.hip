// launchers.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ host-side launchers
static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap = 256 * 8) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

static inline uint32_t tiles_per_block_for(uint32_t n_tiles);
// row pitch, in partition passes (pairs of tiles), of the range partition's [digit][pass] count /
// offset table
uint32_t part_pass_pitch(uint32_t n) { return (((sort_tiles(n) + 1u) >> 1) + 3u) & ~3u; }

void launch_prepare(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                    const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                    const uint64_t* keep_mask, uint32_t* gstart, uint32_t* cstart,
                    uint32_t* stats, uint32_t part_shift, uint32_t* part_hist,
                    uint32_t* digit0_hist, uint32_t* global_digit_hist, unsigned long long* zero_mask,
                    uint32_t ell_reg, uint32_t* exc, uint32_t exc_cap, uint32_t* exc_cnt) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    if (exc == nullptr) ell_reg = 0;
    uint32_t g = tiles_per_block_for(n_tiles);
    // the partition table has one entry per pass (a pair of tiles) and rows padded to part_pass_pitch(n):
    // a workgroup takes whole passes, and the workgroups cover the padding too (empty tiles: zero counts)
    if (part_hist) g = (g + 1u) & ~1u;
    const uint32_t covered = part_hist ? 2u * part_pass_pitch(n) : n_tiles;
    hipLaunchKernelGGL(k_prepare, dim3((covered + g - 1) / g), dim3(256), 0, st, starts, ends, n,
                       d_roff, d_poff, n_contigs, keep_mask, gstart, cstart, stats, n_tiles, g,
                       part_shift, part_hist, part_pass_pitch(n), digit0_hist, global_digit_hist, zero_mask,
                       ell_reg, exc, exc_cap, exc_cnt);
    if (ell_reg != 0u)  // stats[4] = exceptions listed
        hipLaunchKernelGGL(k_nu_count_groups, dim3(32), dim3(256), 0, st, exc_cnt, n_tiles * 4u, stats);
}
uint32_t prepare_exc_slots(uint32_t n) { return sort_tiles(n) * 4u * 128u + kNuOverflow; }  // the list's slots on this form: 128 per wave and tile, and the overflow region

void launch_general_keys(hipStream_t st, bool wide, const uint32_t* gstart, const uint32_t* starts,
                         const uint32_t* ends, uint32_t n, uint32_t span_bits, uint32_t max_span,
                         const uint64_t* keep_mask, void* keys, uint32_t* ecnt, uint32_t ecnt_len) {
    // small genomes (the end counts fit a workgroup's LDS) with many reads per position: count in LDS
    constexpr uint32_t kLdsBins = 36u * 1024;
    if (ecnt != nullptr && ecnt_len != 0 && ecnt_len <= kLdsBins && n >= 16u * ecnt_len) {
        const size_t lds = (size_t)ecnt_len * sizeof(uint32_t);
        const uint32_t grid = 256;
        if (wide) {
            (void)hipFuncSetAttribute((const void*)k_general_keys_lds<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_general_keys_lds<uint64_t>, dim3(grid), dim3(1024), lds, st, gstart, starts, ends, n,
                               span_bits, max_span, keep_mask, (uint64_t*)keys, ecnt, ecnt_len);
        } else {
            (void)hipFuncSetAttribute((const void*)k_general_keys_lds<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k_general_keys_lds<uint32_t>, dim3(grid), dim3(1024), lds, st, gstart, starts, ends, n,
                               span_bits, max_span, keep_mask, (uint32_t*)keys, ecnt, ecnt_len);
        }
        return;
    }
    if (wide)
        hipLaunchKernelGGL(k_general_keys<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, gstart,
                           starts, ends, n, span_bits, max_span, keep_mask, (uint64_t*)keys, ecnt);
    else
        hipLaunchKernelGGL(k_general_keys<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, st, gstart,
                           starts, ends, n, span_bits, max_span, keep_mask, (uint32_t*)keys, ecnt);
}

uint32_t scan_spine_entries(uint32_t n) { return (n + kScanTile - 1) / kScanTile + 1; }

void launch_exclusive_scan(hipStream_t st, const uint32_t* in, uint32_t n, uint32_t* out,
                           uint32_t* spine, bool write_total) {
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(n_tiles), dim3(kScanThreads), 0, st, in, n, spine);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(kScanThreads), 0, st, spine, n_tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(kScanThreads), 0, st, in, n, spine, out,
                       write_total ? 1 : 0);
}

uint32_t sort_tiles(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }

void launch_radix_hist(hipStream_t st, bool wide, const void* keys_in, uint32_t n, uint32_t shift,
                       uint32_t* hist) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    if (wide)
        hipLaunchKernelGGL(k_radix_hist<uint64_t>, dim3(n_tiles), dim3(kSortThreads), 0, st,
                           (const uint64_t*)keys_in, n, shift, n_tiles, hist);
    else
        hipLaunchKernelGGL(k_radix_hist<uint32_t>, dim3(n_tiles), dim3(kSortThreads), 0, st,
                           (const uint32_t*)keys_in, n, shift, n_tiles, hist);
}

void launch_radix_scatter(hipStream_t st, bool wide, const void* keys_in, const uint32_t* vals_in,
                          uint32_t n, uint32_t shift, const uint32_t* offs, void* keys_out,
                          uint32_t* vals_out) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    if (wide)
        hipLaunchKernelGGL(k_radix_scatter<uint64_t>, dim3(n_tiles), dim3(kSortThreads), 0, st,
                           (const uint64_t*)keys_in, vals_in, n, shift, n_tiles, offs,
                           (uint64_t*)keys_out, vals_out);
    else
        hipLaunchKernelGGL(k_radix_scatter<uint32_t>, dim3(n_tiles), dim3(kSortThreads), 0, st,
                           (const uint32_t*)keys_in, vals_in, n, shift, n_tiles, offs,
                           (uint32_t*)keys_out, vals_out);
}

// Cut-point segmentation: window count for a genome of ltot positions (0: not worth it).  Windows
// hold at least 64 blocks, so a stretch is long enough to amortise a pipeline start.
uint32_t sweep_segment_windows(uint32_t ltot, uint32_t ell, uint32_t n_contigs, uint32_t max_windows) {
    if (n_contigs >= 256 || ell == 0) return 0;
    const uint64_t w = (uint64_t)ltot / (64ull * ell);
    const uint32_t most = (uint32_t)kSegMaxCandidates - 256;
    const uint32_t cap = max_windows < most ? max_windows : most;
    return (uint32_t)(w < 2 ? 0 : (w > cap ? cap : w));
}
size_t sweep_segment_words(uint32_t n_contigs, uint32_t n_windows) {
    // the windows' cuts, then three tables (the exact one, and two with speculative boundaries): count,
    // {start, end, contig end}, the owned-from position and the exact table's stretch per stretch
    return (size_t)n_windows + 3 * (1 + 5 * ((size_t)n_contigs + n_windows));
}
// fills seg_words: [windows' cuts | count, stretches]; returns the table the sweep launchers take
const uint32_t* launch_sweep_segments(hipStream_t st, const uint32_t* boff, const uint32_t* eoff,
                                      const uint64_t* d_poff, uint32_t n_contigs, uint32_t ltot, uint32_t ell,
                                      uint32_t M, uint32_t n_windows, uint32_t* seg_words, const uint32_t* other_cov) {
    uint32_t* cut = seg_words;
    uint32_t* seg = seg_words + n_windows;
    const uint32_t win = (ltot + n_windows - 1) / n_windows;
    hipLaunchKernelGGL(k_find_cuts, dim3(n_windows), dim3(256), 0, st, boff, eoff, d_poff, n_contigs, ltot, ell, M, win, cut, other_cov);
    hipLaunchKernelGGL(k_build_segments, dim3(1), dim3(kSegThreads), 0, st, cut, n_windows, d_poff, n_contigs,
                       ltot, win, 0u, 1u, seg, (uint32_t*)nullptr);
    return seg;
}
// the same windows' further tables (tier 1 or 2): speculative boundaries where a window has no cut (behind
// the exact table in seg_words; launch_sweep_segments comes first); burn == 0 gives the exact table again
const uint32_t* launch_sweep_segments_speculative(hipStream_t st, const uint64_t* d_poff, uint32_t n_contigs,
                                                  uint32_t ltot, uint32_t n_windows, uint32_t burn,
                                                  uint32_t* seg_words, uint32_t* n_speculative, uint32_t run_ins_apart,
                                                  uint32_t tier) {
    // candidates this many run-ins apart at least (>= 2)
    const uint32_t win0 = (ltot + n_windows - 1) / n_windows;
    const uint32_t stride = (uint32_t)(((uint64_t)run_ins_apart * burn + win0 - 1) / win0);
    const uint32_t* cut = seg_words;
    uint32_t* seg = seg_words + n_windows + (size_t)tier * (1 + 5 * ((size_t)n_contigs + n_windows));
    const uint32_t win = (ltot + n_windows - 1) / n_windows;
    hipLaunchKernelGGL(k_build_segments, dim3(1), dim3(kSegThreads), 0, st, cut, n_windows, d_poff, n_contigs,
                       ltot, win, burn, stride < 1 ? 1u : stride, seg, n_speculative);
    return seg;
}
size_t spec_snap_bytes(uint32_t n_cand) { return (size_t)n_cand * kSpecSnapWords * sizeof(uint32_t); }
void launch_spec_verify_merge_mixed(hipStream_t st, const uint32_t* seg, uint32_t n_cand, uint32_t max_span,
                                    uint32_t* out_even, const uint32_t* out_odd, const uint32_t* snap,
                                    uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
    hipLaunchKernelGGL(k_spec_verify_mixed, dim3(n_cand), dim3(256), 0, st, seg, n_cand, max_span, out_even, out_odd, snap,
                       kSpecSnapWords, mismatches, redo_in, redo_out);
    hipLaunchKernelGGL(k_spec_merge_mixed, dim3(n_cand, 32), dim3(256), 0, st, seg, n_cand, max_span, out_even, out_odd,
                       redo_in);
}
void launch_spec_verify(hipStream_t st, const uint32_t* seg, uint32_t n_cand, uint32_t ell,
                        const uint32_t* owned, const uint32_t* run_in, uint32_t* mismatches,
                        const uint32_t* redo_in, uint32_t* redo_out, const uint32_t* own_marks) {
    hipLaunchKernelGGL(k_spec_verify, dim3(n_cand), dim3(256), 0, st, seg, n_cand, ell, owned, run_in, mismatches,
                       redo_in, redo_out, own_marks);
}

bool sweep_uniform_mw_supported(uint32_t ell) { return ell >= 1 && (ell + 63) / 64 <= 4; }

bool launch_sweep_uniform_mw(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                             uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                             uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg, uint32_t n_seg_max) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP_MW(EE)                                                                              \
    {                                                                                                   \
        const size_t lds = MwLayout<EE>::kBytes;                                                        \
        (void)hipFuncSetAttribute((const void*)k_sweep_uniform_mw<EE>,                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL(k_sweep_uniform_mw<EE>, dim3(n_wg), dim3(448), lds, st, boff, d_poff,       \
                           ell, M, ltot, selend, iter_stats, seg);                                           \
    }
    switch (e) {
        case 1: QMCP_SWEEP_MW(1); break;
        case 2: QMCP_SWEEP_MW(2); break;
        case 3: QMCP_SWEEP_MW(3); break;
        case 4: QMCP_SWEEP_MW(4); break;
        default: return false;  // wider spans: single-wave kernel
    }
#undef QMCP_SWEEP_MW
    return true;
}

bool launch_sweep_uniform_gen(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                              uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                              uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg, uint32_t n_seg_max,
                              uint32_t* selend_run_in, const uint32_t* redo_in, const int32_t* nadj,
                              const uint32_t* own_marks) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP_GEN_K(EE, ADJ)                                                                      \
    {                                                                                                   \
        const size_t lds = MgLayout<EE>::kBytes;                                                        \
        (void)hipFuncSetAttribute((const void*)k_sweep_uniform_gen<EE, ADJ>,                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((k_sweep_uniform_gen<EE, ADJ>), dim3(n_wg), dim3(448), lds, st, boff, d_poff, \
                           ell, M, ltot, selend, iter_stats, seg, selend_run_in, redo_in, n_seg_max, nadj, own_marks);  \
    }
#define QMCP_SWEEP_GEN(EE)                                                                             \
    {                                                                                                   \
        if (nadj != nullptr) QMCP_SWEEP_GEN_K(EE, true) else QMCP_SWEEP_GEN_K(EE, false)               \
    }
    switch (e) {
        case 1: QMCP_SWEEP_GEN(1); break;
        case 2: QMCP_SWEEP_GEN(2); break;
        case 3: QMCP_SWEEP_GEN(3); break;
        case 4: QMCP_SWEEP_GEN(4); break;
        default: return false;
    }
#undef QMCP_SWEEP_GEN
#undef QMCP_SWEEP_GEN_K
    return true;
}

// event-driven form (deep data): field widths per E in EvPack; M must leave room in a field (see the kernel)
bool sweep_uniform_ev_supported(uint32_t ell, uint32_t M) {
    if (!sweep_uniform_mw_supported(ell)) return false;
    const uint32_t e = (ell + 63) / 64;
    const uint64_t sat = (1ull << (30 / e - 1)) - 1;  // EvPack<E>::kSat
    return (uint64_t)M + 1 <= sat;  // no kept count can reach a saturated field's value
}
uint32_t sweep_ev_pieces(uint32_t ltot, uint32_t ell, uint32_t n_wg) { return ltot / (4u * ell) + n_wg + 1; }
size_t sweep_ev_pack_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg) { return (size_t)sweep_ev_pieces(ltot, ell, n_wg) * 1024; }
size_t sweep_ev_ckpt_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg) {  // the chain's state at every 64th block
    return ((size_t)sweep_ev_pieces(ltot, ell, n_wg) / 16 + 2 * (size_t)n_wg + 4) * 512 * sizeof(uint32_t);
}
size_t sweep_ev_last_bytes(uint32_t ltot, uint32_t ell, uint32_t n_wg) {
    return ((size_t)sweep_ev_pieces(ltot, ell, n_wg) * 4 + 64) * sizeof(uint32_t);
}
// the three launches of the event-driven form, separately (the host brackets each with events)
#define QMCP_EV_DISPATCH(e, CALL)              \
    switch (e) {                                \
        case 1: { CALL(1) } break;              \
        case 2: { CALL(2) } break;              \
        case 3: { CALL(3) } break;              \
        case 4: { CALL(4) } break;              \
        default: return false;                  \
    }
bool launch_sweep_ev_pack(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                          uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                          uint32_t* pk, const int32_t* nadj, const uint32_t* from) {
    if (!sweep_uniform_ev_supported(ell, M)) return false;
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t pieces = sweep_ev_pieces(ltot, ell, n_wg);
#define QMCP_CALL(EE)                                                                                      \
    hipLaunchKernelGGL(k_sweep_pack<EE>, dim3((pieces + 3) / 4), dim3(256), 0, st, boff, d_poff, n_wg, ell, \
                       M, ltot, seg, pieces, pk, nadj, from);
    QMCP_EV_DISPATCH((ell + 63) / 64, QMCP_CALL)
#undef QMCP_CALL
    return true;
}
bool launch_sweep_ev_chain(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                           uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                           const uint32_t* pk, uint32_t* sev, uint32_t* lastns, uint32_t* iter_stats,
                           const int32_t* nadj, uint32_t* ckpt, const uint32_t* restart) {
    if (!sweep_uniform_ev_supported(ell, M)) return false;
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const size_t lds = (size_t)kEvSlots * 1024;
#define QMCP_CALL(EE)                                                                                       \
    (void)hipFuncSetAttribute((const void*)k_sweep_uniform_ev<EE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                    \
    hipLaunchKernelGGL(k_sweep_uniform_ev<EE>, dim3(n_wg), dim3(128), lds, st, boff, d_poff, n_contigs, ell,  \
                       M, ltot, pk, sev, lastns, iter_stats, seg, nadj, ckpt, restart);
    QMCP_EV_DISPATCH((ell + 63) / 64, QMCP_CALL)
#undef QMCP_CALL
    return true;
}
bool launch_sweep_ev_expand(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff, uint32_t n_contigs,
                            uint32_t ell, uint32_t M, uint32_t ltot, const uint32_t* seg, uint32_t n_seg_max,
                            const uint32_t* sev, const uint32_t* lastns, uint32_t* selend, const uint32_t* from) {
    if (!sweep_uniform_ev_supported(ell, M)) return false;
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t pieces = sweep_ev_pieces(ltot, ell, n_wg);
#define QMCP_CALL(EE)                                                                                    \
    hipLaunchKernelGGL(k_sweep_expand<EE>, dim3(pieces), dim3(256), 0, st, boff, d_poff, n_wg, ell, ltot, \
                       seg, pieces, sev, lastns, selend, from);
    QMCP_EV_DISPATCH((ell + 63) / 64, QMCP_CALL)
#undef QMCP_CALL
    return true;
}
#undef QMCP_EV_DISPATCH

bool launch_sweep_uniform(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                          uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                          uint32_t* selend, uint32_t* iter_stats, const uint32_t* seg, uint32_t n_seg_max) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP(EE)                                                                          \
    hipLaunchKernelGGL(k_sweep_uniform<EE>, dim3(n_wg), dim3(64), 0, st, boff, d_poff, ell, \
                       M, ltot, selend, iter_stats, seg)
    switch (e) {
        case 1: QMCP_SWEEP(1); break;
        case 2: QMCP_SWEEP(2); break;
        case 3: QMCP_SWEEP(3); break;
        case 4: QMCP_SWEEP(4); break;
        case 5: QMCP_SWEEP(5); break;
        case 6: QMCP_SWEEP(6); break;
        case 7: QMCP_SWEEP(7); break;
        case 8: QMCP_SWEEP(8); break;
        default: return false;
    }
#undef QMCP_SWEEP
    return true;
}

void launch_sweep_general(hipStream_t st, bool wide, const uint32_t* boff, const uint32_t* eoff,
                          const void* sorted, const uint64_t* d_poff, uint32_t n_contigs,
                          uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* selend,
                          uint32_t ring_size, const uint32_t* seg, uint32_t n_seg_max, uint32_t* g_rings) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    // rings in LDS while they fit (two of ring_size words); beyond that in g_rings (2 * ring_size words
    // per workgroup)
    const size_t lds = g_rings ? 0 : 2 * (size_t)ring_size * sizeof(uint32_t);
#define QMCP_SWEEP_GENERAL(SORTED, ARG, GRING)                                                          \
    {                                                                                                   \
        (void)hipFuncSetAttribute((const void*)k_sweep_general<SORTED, GRING>,                          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL((k_sweep_general<SORTED, GRING>), dim3(n_wg), dim3(64), lds, st, boff, eoff, \
                           ARG, d_poff, span_bits, max_span, M, selend, ring_size, seg, g_rings);       \
    }
    if (wide) {
        if (g_rings) QMCP_SWEEP_GENERAL(SortedK64, SortedK64{(const uint64_t*)sorted}, true)
        else QMCP_SWEEP_GENERAL(SortedK64, SortedK64{(const uint64_t*)sorted}, false)
    } else {
        if (g_rings) QMCP_SWEEP_GENERAL(SortedRec, SortedRec{(const Rec*)sorted}, true)
        else QMCP_SWEEP_GENERAL(SortedRec, SortedRec{(const Rec*)sorted}, false)
    }
#undef QMCP_SWEEP_GENERAL
}

void launch_group_heads(hipStream_t st, bool wide, const void* sorted, uint32_t n,
                        uint32_t* next_head) {
    if (wide)
        hipLaunchKernelGGL(k_group_heads<SortedK64>, dim3(grid_for((uint64_t)n + 1, 256)), dim3(256), 0, st,
                           SortedK64{(const uint64_t*)sorted}, n, next_head);
    else
        hipLaunchKernelGGL(k_group_heads<SortedRec>, dim3(grid_for((uint64_t)n + 1, 256)), dim3(256), 0, st,
                           SortedRec{(const Rec*)sorted}, n, next_head);
}

void launch_sweep_general_cached(hipStream_t st, bool wide, const uint32_t* boff,
                                 const uint32_t* eoff, const void* sorted, const uint32_t* next_head,
                                 const uint64_t* d_poff, uint32_t n_contigs, uint32_t span_bits,
                                 uint32_t max_span, uint32_t M, uint32_t* selend, uint32_t ring,
                                 const uint32_t* seg, uint32_t n_seg_max) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const size_t lds = (size_t)GenSlots::kWords * ring * sizeof(uint32_t);
    if (wide) {
        (void)hipFuncSetAttribute((const void*)k_sweep_general_cached<SortedK64>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general_cached<SortedK64>, dim3(n_wg), dim3(64), lds, st, boff,
                           eoff, SortedK64{(const uint64_t*)sorted}, next_head, d_poff, span_bits,
                           max_span, M, selend, ring, seg);
    } else {
        (void)hipFuncSetAttribute((const void*)k_sweep_general_cached<SortedRec>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general_cached<SortedRec>, dim3(n_wg), dim3(64), lds, st, boff,
                           eoff, SortedRec{(const Rec*)sorted}, next_head, d_poff, span_bits, max_span,
                           M, selend, ring, seg);
    }
}

// register-resident event sweep: buckets per lane B = ceil((max_span + 64) / 64), up to 8
bool launch_sweep_general_reg(hipStream_t st, bool wide, const uint32_t* boff, const uint32_t* eoff,
                              const void* sorted, const uint32_t* next_head, const uint64_t* d_poff,
                              uint32_t n_contigs, uint32_t span_bits, uint32_t max_span, uint32_t M,
                              uint32_t* selend, const uint32_t* seg, uint32_t n_seg_max,
                              uint32_t* selend_odd, const uint32_t* redo_in, uint32_t* snap) {
    const uint32_t n_wg = seg ? n_seg_max : n_contigs;
    const uint32_t b = (max_span + 64 + 63) / 64;
#ifdef QMCP_GEN_STAMP
#define QMCP_GEN_STAMP_ARG , (unsigned long long*)nullptr
#else
#define QMCP_GEN_STAMP_ARG
#endif
#define QMCP_GEN_REG_K(BB, KK)                                                                        \
    if (wide)                                                                                          \
        hipLaunchKernelGGL((k_sweep_general_reg<SortedK64, BB, KK>), dim3(n_wg), dim3(64 * (1 + KK)), 0, st, boff, \
                           eoff, SortedK64{(const uint64_t*)sorted}, next_head, d_poff, span_bits,    \
                           max_span, M, selend, seg, selend_odd, redo_in, n_seg_max, snap QMCP_GEN_STAMP_ARG); \
    else                                                                                               \
        hipLaunchKernelGGL((k_sweep_general_reg<SortedRec, BB, KK>), dim3(n_wg), dim3(64 * (1 + KK)), 0, st, boff, \
                           eoff, SortedRec{(const Rec*)sorted}, next_head, d_poff, span_bits, max_span, \
                           M, selend, seg, selend_odd, redo_in, n_seg_max, snap QMCP_GEN_STAMP_ARG);
    // loader waves per walker: few workgroups (contigs) -> many loaders, so the walker never waits for an
    // entering chunk's three trips to memory; many workgroups (stretches) fill the chip by themselves
    // and extra waves only get in the walkers' way
    int loaders = n_wg <= 64 ? 4 : 1;  // (769 stretches, cfg3-like mix: 1 loader 2.52 ms, 8 loaders 4.92 ms; one contig: 1.19 vs 1.09 ms)
#define QMCP_GEN_REG(BB)                                                                              \
    if (loaders >= 8) { QMCP_GEN_REG_K(BB, 8) }                                                        \
    else if (loaders >= 4) { QMCP_GEN_REG_K(BB, 4) }                                                   \
    else if (loaders >= 2) { QMCP_GEN_REG_K(BB, 2) }                                                   \
    else { QMCP_GEN_REG_K(BB, 1) }
    if (b <= 2) { QMCP_GEN_REG(2) }
    else if (b == 3) { QMCP_GEN_REG(3) }
    else if (b == 4) { QMCP_GEN_REG(4) }
    else if (b <= 6) { QMCP_GEN_REG(6) }
    else if (b <= 8) { QMCP_GEN_REG(8) }
    else return false;
#undef QMCP_GEN_REG
#undef QMCP_GEN_REG_K
#undef QMCP_GEN_STAMP_ARG
    return true;
}

void launch_mark(hipStream_t st, bool wide, const void* sorted, const uint32_t* svals, uint32_t ltot,
                 const uint32_t* boff, const uint32_t* selend, uint64_t* mask,
                 unsigned long long* n_kept) {
    if (wide)
        hipLaunchKernelGGL(k_mark<KeysSplit64>, dim3(grid_for(ltot, 256)), dim3(256), 0, st,
                           KeysSplit64{(const uint64_t*)sorted, svals}, ltot, boff, selend,
                           (uint32_t*)mask, n_kept);
    else
        hipLaunchKernelGGL(k_mark<KeysRec>, dim3(grid_for(ltot, 256)), dim3(256), 0, st,
                           KeysRec{(const Rec*)sorted}, ltot, boff, selend, (uint32_t*)mask, n_kept);
}

void launch_bucket_heads(hipStream_t st, bool wide, const void* sorted, const uint32_t* svals,
                         uint32_t n, uint32_t span_bits, uint32_t ltot, uint32_t* boff) {
    if (wide)
        hipLaunchKernelGGL(k_bucket_heads<KeysSplit64>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           KeysSplit64{(const uint64_t*)sorted, svals}, n, span_bits, ltot, boff);
    else
        hipLaunchKernelGGL(k_bucket_heads<KeysRec>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           KeysRec{(const Rec*)sorted}, n, span_bits, ltot, boff);
}

void launch_reverse_min_scan(hipStream_t st, uint32_t* data, uint32_t n, uint32_t* spine) {
    const uint32_t n_tiles = (n + kScanTile - 1) / kScanTile;
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_rmin_tile_mins, dim3(n_tiles), dim3(kScanThreads), 0, st, data, n, spine);
    hipLaunchKernelGGL(k_rmin_spine, dim3(1), dim3(kScanThreads), 0, st, spine, n_tiles);
    hipLaunchKernelGGL(k_rmin_tiles, dim3(n_tiles), dim3(kScanThreads), 0, st, data, n, spine);
}

static inline uint32_t tiles_per_block_for(uint32_t n_tiles) {
    // keep >= ~2048 workgroups in flight; up to 8 consecutive tiles per workgroup
    uint32_t g = n_tiles / 2048;
    return g < 1 ? 1 : (g > 8 ? 8 : g);
}

void launch_radix_hist_rec(hipStream_t st, bool first, const uint32_t* keys, const void* recs,
                           uint32_t n, uint32_t shift, uint32_t* hist) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const uint32_t g = tiles_per_block_for(n_tiles);
    const uint32_t grid = (n_tiles + g - 1) / g;
    if (first)
        hipLaunchKernelGGL(k_radix_hist_rec<true>, dim3(grid), dim3(kSortThreads), 0, st, keys,
                           (const Rec*)recs, n, shift, n_tiles, g, hist);
    else
        hipLaunchKernelGGL(k_radix_hist_rec<false>, dim3(grid), dim3(kSortThreads), 0, st, keys,
                           (const Rec*)recs, n, shift, n_tiles, g, hist);
}

void launch_radix_scatter_rec(hipStream_t st, bool first, const uint32_t* keys, const void* recs_in,
                              uint32_t n, uint32_t shift, const uint32_t* offs, void* recs_out) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const uint32_t g = tiles_per_block_for(n_tiles);
    const uint32_t grid = (n_tiles + g - 1) / g;
    if (first)
        hipLaunchKernelGGL((k_radix_scatter_rec<true, false>), dim3(grid), dim3(kSortThreads), 0, st,
                           keys, (const Rec*)recs_in, n, shift, n_tiles, g, offs, recs_out);
    else
        hipLaunchKernelGGL((k_radix_scatter_rec<false, false>), dim3(grid), dim3(kSortThreads), 0, st,
                           keys, (const Rec*)recs_in, n, shift, n_tiles, g, offs, recs_out);
}


// range-ranked uniform path: geometry, partition table, counts, rank + mark
static constexpr uint32_t kTwoLevelRangeShift = 15;
uint32_t range_shift_for(uint32_t ltot) {
    // smallest shift whose ranges (positions 0..ltot inclusive) fit the 256 digits of one pass; beyond
    // 256 ranges of 32 Ki positions a second partition level supplies eight more digit bits
    uint32_t shift = 0;
    while (shift < kMaxRangeShift && (ltot >> shift) >= 256u) ++shift;
    if ((ltot >> kMaxRangeShift) >= 256u) {
        // two levels: any shift with <= 65 536 ranges (and so <= 256 super-ranges) will do
        uint32_t lowest = 0;
        while ((ltot >> lowest) >= 65536u) ++lowest;
        const uint32_t want = kTwoLevelRangeShift;
        shift = want < lowest ? lowest : (want > kMaxRangeShift ? kMaxRangeShift : want);
    }
    return shift;
}
bool range_path_two_level(uint32_t ltot) { return (ltot >> kMaxRangeShift) >= 256u; }
bool range_path_supported(uint32_t ltot) { return (ltot >> kMaxRangeShift) < 65536u; }  // always, for 32-bit positions < 2^31

template <int MODE, bool OUT_REC>
static void launch_partition_t(hipStream_t st, dim3 grid, const uint32_t* keys, const Rec* recs_in,
                               SegTables seg, const uint64_t* d_roff, const uint64_t* d_poff,
                               uint32_t n_contigs, uint32_t n, uint32_t shift, uint32_t n_tiles,
                               const uint32_t* offs, uint16_t* k16, uint32_t* idx, Rec* out_rec,
                               uint32_t* range_start, uint32_t* max_load, const uint32_t* ends = nullptr,
                               uint32_t ell_reg = 0) {
    (void)hipFuncSetAttribute((const void*)k_range_partition<MODE, OUT_REC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartLds);
    hipLaunchKernelGGL((k_range_partition<MODE, OUT_REC>), grid, dim3(kPartThreads), kPartLds, st, keys,
                       recs_in, seg, d_roff, d_poff, n_contigs, n, shift, n_tiles, offs, k16, idx, out_rec,
                       range_start, max_load, ends, ends != nullptr ? ell_reg : 0u);
}

void launch_range_partition(hipStream_t st, const uint32_t* gstart_or_null, const uint32_t* starts,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                            uint32_t n, uint32_t shift, const uint32_t* offs, uint16_t* keys16_out,
                            uint32_t* idx_out, uint32_t* range_start, uint32_t* max_load, const uint32_t* ends,
                            uint32_t ell_reg) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const dim3 grid((n_tiles + kPartTiles - 1) / kPartTiles);
    const SegTables none{nullptr, nullptr, nullptr};
    if (gstart_or_null)
        launch_partition_t<0, false>(st, grid, gstart_or_null, nullptr, none, d_roff, d_poff, n_contigs, n, shift,
                                     part_pass_pitch(n), offs, keys16_out, idx_out, nullptr, range_start, max_load);
    else
        launch_partition_t<1, false>(st, grid, starts, nullptr, none, d_roff, d_poff, n_contigs, n, shift,
                                     part_pass_pitch(n), offs, keys16_out, idx_out, nullptr, range_start, max_load, ends, ell_reg);
}

// Two-level route.  Level 1: stable partition of the reads into <= 256 super-ranges of 2^(shift+8)
// positions, as {global start, index} records; its first workgroup publishes super_start[257].
void launch_partition_level1(hipStream_t st, const uint32_t* starts, const uint64_t* d_roff,
                             const uint64_t* d_poff, uint32_t n_contigs, uint32_t n, uint32_t shift_hi,
                             const uint32_t* offs, void* recs_out, uint32_t* super_start,
                             uint32_t* max_super_load, const uint32_t* ends, uint32_t ell_reg) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const dim3 grid((n_tiles + kPartTiles - 1) / kPartTiles);
    const SegTables none{nullptr, nullptr, nullptr};
    launch_partition_t<1, true>(st, grid, starts, nullptr, none, d_roff, d_poff, n_contigs, n, shift_hi,
                                part_pass_pitch(n), offs, nullptr, nullptr, (Rec*)recs_out, super_start,
                                max_super_load, ends, ell_reg);
}
// Level 2: every super-range is partitioned on its own into its (<= 256) final ranges.
// tables: [0,257) super_start  [257,514) tile_base  [514,771) pass_base (written here)
uint32_t seg_tile_bound(uint32_t n) { return sort_tiles(n) + 256; }  // upper bound of the tile count
void launch_partition_level2(hipStream_t st, const void* recs_in, uint32_t n, uint32_t shift,
                             uint32_t* tables, uint32_t* hist, uint32_t* spine, uint16_t* keys16_out,
                             uint32_t* idx_out, uint32_t* range_start, uint32_t* max_load) {
    const SegTables seg{tables, tables + 257, tables + 514};
    const uint32_t t_bound = seg_tile_bound(n);
    hipLaunchKernelGGL(k_seg_tables, dim3(1), dim3(256), 0, st, tables, tables + 257, tables + 514, max_load);
    (void)hipMemsetAsync(hist, 0, (size_t)256 * t_bound * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_seg_hist, dim3(t_bound), dim3(kSortThreads), 0, st, (const Rec*)recs_in, seg, shift, hist);
    launch_exclusive_scan(st, hist, 256u * t_bound, hist, spine, false);
    launch_partition_t<2, false>(st, dim3((t_bound + kPartTiles - 1) / kPartTiles + 256), nullptr,
                                 (const Rec*)recs_in, seg, nullptr, nullptr, 0, n, shift, 0, hist, keys16_out,
                                 idx_out, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_seg_range_table, dim3(256), dim3(256), 0, st, hist, seg, n, range_start, max_load);
}

// global start position per read (what k_prepare writes when asked to): for the routes that need
// the bare keys after a call that did not ask for them
__global__ __launch_bounds__(256) void k_gstart(const uint32_t* __restrict__ starts, uint32_t n,
                                               const uint64_t* __restrict__ contig_read_off,
                                               const uint64_t* __restrict__ contig_pos_off,
                                               uint32_t n_contigs, uint32_t* __restrict__ gstart) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t lo = 0, hi = n_contigs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (contig_read_off[mid] <= i) lo = mid; else hi = mid;
        }
        gstart[i] = (uint32_t)contig_pos_off[lo] + starts[i];
    }
}
void launch_gstart(hipStream_t st, const uint32_t* starts, uint32_t n, const uint64_t* d_roff,
                   const uint64_t* d_poff, uint32_t n_contigs, uint32_t* gstart) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_gstart, dim3(grid_for(n, 256)), dim3(256), 0, st, starts, n, d_roff, d_poff,
                       n_contigs, gstart);
}
// ends of reads that all have one span (the 64-bit host entry then sends the starts only)
__global__ __launch_bounds__(256) void k_fill_ends(const uint32_t* __restrict__ starts, uint32_t n,
                                                  uint32_t span_minus_1, uint32_t* __restrict__ ends) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) ends[i] = starts[i] + span_minus_1;
}
void launch_fill_ends(hipStream_t st, const uint32_t* starts, uint32_t n, uint32_t span_minus_1, uint32_t* ends) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fill_ends, dim3(grid_for(n, 256 * 4)), dim3(256), 0, st, starts, n, span_minus_1, ends);
}
void launch_range_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* range_start,
                          uint32_t shift, uint32_t ltot, uint32_t* boff, uint32_t* empty_positions) {
    const uint32_t n_ranges = (ltot >> shift) + 1;  // covers positions 0..ltot
    const size_t lds = (((size_t)1 << shift) + ((size_t)1 << shift) / 32 + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_range_offsets, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(k_range_offsets, dim3(n_ranges), dim3(1024), lds, st, keys16, range_start, shift,
                       ltot, boff, empty_positions);
}
bool rank_scratch_by_records(uint32_t shift, uint32_t ltot, uint32_t n) {
    return (size_t)n < (size_t)((ltot >> shift) + 1) * ((size_t)1 << shift);
}
size_t rank_scratch_bytes(uint32_t shift, uint32_t ltot, uint32_t n) {
    const size_t by_pos = (size_t)((ltot >> shift) + 1) * ((size_t)1 << shift);
    return (by_pos < (size_t)n ? by_pos : (size_t)n) * sizeof(uint2) + 64;
}
void launch_rank_mark(hipStream_t st, const uint16_t* keys16, const uint32_t* idx,
                      const uint32_t* range_start, uint32_t shift, uint32_t ltot, const uint32_t* boff,
                      const uint32_t* selend, unsigned long long* mask, unsigned long long* kept_total,
                      void* scratch, bool scratch_by_records) {
    const uint32_t n_ranges = (ltot >> shift) + 1;
    const size_t lds = (((size_t)1 << shift) + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_rank_mark, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(k_rank_mark, dim3(n_ranges), dim3(1024), lds, st, keys16, idx, range_start,
                       shift, ltot, boff, selend, mask, kept_total, (uint2*)scratch, scratch_by_records ? 1 : 0);
}

void launch_coverage(hipStream_t st, const uint32_t* boff, const uint32_t* eoff, uint32_t ltot,
                     uint32_t* cov) {
    hipLaunchKernelGGL(k_coverage, dim3(grid_for(ltot, 256)), dim3(256), 0, st, boff, eoff, ltot, cov);
}

void launch_b_and_demand(hipStream_t st, const uint32_t* cov, uint32_t n, uint32_t M, int32_t* b, int32_t* d) {
    const uint32_t grid = (n + 1 + 255) / 256;
    hipLaunchKernelGGL(k_b_and_demand, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, st, cov, n, M, b, d);
}
void launch_complete_pairs(hipStream_t st, uint64_t* mask, uint32_t n_words, uint64_t n_reads) {
    hipLaunchKernelGGL(k_complete_pairs, dim3(grid_for(n_words, 256)), dim3(256), 0, st, mask,
                       n_words, n_reads);
}

void launch_word_popcounts(hipStream_t st, const uint64_t* words, uint32_t n_words, uint32_t* counts) {
    hipLaunchKernelGGL(k_word_popcounts, dim3(grid_for(n_words, 256)), dim3(256), 0, st, words, n_words,
                       counts);
}
void launch_mask_to_indices(hipStream_t st, const uint64_t* mask, uint32_t n_words, const uint32_t* word_base,
                            unsigned long long* out) {
    hipLaunchKernelGGL(k_mask_to_indices, dim3(grid_for(n_words, 256)), dim3(256), 0, st, mask, n_words, word_base, out);
}
void launch_compact_pairs(hipStream_t st, const uint32_t* starts, const uint32_t* ends,
                          const uint64_t* pair_keep, const uint32_t* word_base, uint64_t n_pairs,
                          uint32_t* starts_c, uint32_t* ends_c, uint32_t* orig_pair) {
    hipLaunchKernelGGL(k_compact_pairs, dim3(grid_for(n_pairs, 256)), dim3(256), 0, st, starts, ends,
                       pair_keep, word_base, n_pairs, starts_c, ends_c, orig_pair);
}
void launch_expand_mask(hipStream_t st, const uint64_t* mask_c, const uint32_t* orig_pair,
                        uint32_t n_reads_c, uint64_t* mask) {
    hipLaunchKernelGGL(k_expand_mask, dim3(grid_for(n_reads_c, 256)), dim3(256), 0, st, mask_c,
                       orig_pair, n_reads_c, (uint32_t*)mask);
}

void launch_amplicon_filter(hipStream_t st, const uint32_t* starts, const uint32_t* ends,
                            const uint32_t* seq_lengths, const uint32_t* qualities,
                            uint64_t n_pairs, const uint32_t* amp_starts, const uint32_t* amp_ends,
                            uint32_t n_amp, uint32_t min_length, uint32_t min_mapq,
                            uint64_t* pair_keep) {
    const uint32_t n_cached = n_amp < 4096u ? n_amp : 4096u;
    const uint64_t n_words = (n_pairs + 63) / 64;
    hipLaunchKernelGGL(k_amplicon_filter, dim3(grid_for(n_words * 64, 256)), dim3(256),
                       2 * n_cached * sizeof(uint32_t), st, starts, ends, seq_lengths, qualities,
                       n_pairs, amp_starts, amp_ends, n_amp, min_length, min_mapq, pair_keep);
}
