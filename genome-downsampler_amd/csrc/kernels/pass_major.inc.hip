// pass_major.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ range-ranked route, pass-major layout
// The range-ranked route (ranked_route.inc.hip) needs the reads grouped by position range, in read-index order
// inside a range.  Its first form reads the reads twice before it can write them -- k_prepare (histogram of every
// pass of 8 192 reads), a scan of the histograms, k_range_partition (re-reads the starts, writes 6 B per read to
// where the scan says) -- because a range-major array needs every pass's counts before the first record can be
// placed: 2.0 GB of the 3.2 GB a cfg4 solve moved.  Here the grouped array is never materialised:
//   k_pm_prepare_sort   ONE pass over the reads (8 B per read in): validate, span statistics, clear the keep mask,
//                       sort every pass of 8 192 reads by range IN PLACE -- slot P * 8192 + j of two 16-bit streams
//                       holds the pass's j-th record in (range, read index) order: position inside the range and
//                       read index inside the pass, 4 B per read out -- and two small tables laid out [range][pass]:
//                       how many records of the range the pass holds, and where they begin inside the pass.
//   (scan)              exclusive scan over the count table: T[d][P] = where pass P's slice of range d WOULD begin in
//                       a range-major array -- the flat coordinate the consumers walk.
//   k_pm_offsets        per range: LDS histogram of its slices' positions -> bucket offsets (k_range_offsets' job).
//   k_pm_rank_mark      per range: k_rank_mark's ordered walk over the range's records in flat order; a wave finds
//                       the slices under its 64 flat positions with a cursor over the range's row of T, kept in LDS.
// A range's records in read-index order are its slices in pass order, so nothing about the selection changes:
// the kept set is bit for bit the first form's.  tests/pass_major_model.py restates the layout and both mappings on
// the host (every index asserted in bounds) and tests/test_pass_major_model.py runs it on ragged inputs.
// One-level genomes only (<= 256 ranges); longer ones keep the two-level partition.
static constexpr int kPmPass = 8192;             // reads per pass
#ifndef QMCP_PM_PAD
#define QMCP_PM_PAD 0  // (measured: 0, 64 and 2048 slots of pad give the same times -- no channel aliasing to avoid)
#endif
// slots between the beginnings of two passes: a pass and a pad, so that the slices of one range in successive passes
// are not a power of two apart (a range's kernels have dozens of them in flight)
static constexpr uint32_t kPmStride = kPmPass + QMCP_PM_PAD;
static constexpr int kPmThreads = 512;           // 8 waves: wave w owns records [1024 w, 1024 (w + 1)) of the pass
static constexpr int kPmWaves = kPmThreads / 64;
static constexpr int kPmPassesPerWg = 4;         // a workgroup's passes leave their table entries as 16-byte runs
static constexpr uint32_t kPmMaxRow = 3072;      // passes of a range's row the consumers hold in LDS (2 x 12 KiB): 25 M reads over the contigs a range overlaps
static constexpr size_t kPmSortLds = ((size_t)kPmPass + kPmWaves * 256 + 256 + 16 + 2 * kPmPassesPerWg * 256 + kPmWaves * 3 * 128) * sizeof(uint32_t);
static constexpr uint32_t kPmExcPerWave = 128;  // list slots per wave and pass: an eighth of the wave's 1 024 reads

#ifndef QMCP_PM_MIN_WAVES
#define QMCP_PM_MIN_WAVES 4  // waves per SIMD the register allocation aims at (6 -- three workgroups per CU -- spills: 0.42 against 0.29 ms)
#endif
__global__ __launch_bounds__(kPmThreads, QMCP_PM_MIN_WAVES) void k_pm_prepare_sort(
    const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends, uint32_t n,
    const uint64_t* __restrict__ contig_read_off, const uint64_t* __restrict__ contig_pos_off, uint32_t n_contigs,
    uint32_t shift, uint16_t* __restrict__ keys16, uint16_t* __restrict__ idx16,
    uint32_t* __restrict__ cnt_tab, uint32_t* __restrict__ lst_tab, uint32_t pitch /* multiple of 4 */,
    uint32_t* __restrict__ stats, unsigned long long* __restrict__ zero_mask,
    // near-uniform route (kernels/near_uniform.inc.hip): reads whose span is not ell_reg are left out of the sorted
    // passes and listed instead -- {global start, global end, read index}, three arrays of exc_cap words.  Every wave
    // of every pass owns kPmExcPerWave slots of the list (pass P, wave w: from (8 P + w) * 128) and says how many it
    // filled in exc_cnt[8 P + w]: no counter is shared (24 k same-address atomics, one per wave, took longer than
    // the whole kernel: 0.32 -> 0.69 ms); k_nu_count_groups adds the groups up into stats[4] afterwards.  stats[5]
    // is set if a wave met more than its slots hold (the list is then incomplete).  ell_reg == 0: every read is regular.
    uint32_t ell_reg, uint32_t* __restrict__ exc, uint32_t exc_cap, uint32_t* __restrict__ exc_cnt) {
    extern __shared__ uint32_t s_pm[];
    uint32_t* s_stage = s_pm;                           // [8192] a pass's records, sorted: key | index in pass << 16
    uint32_t* s_cnt = s_stage + kPmPass;                // [8][256] per-wave digit counts, then offsets
    uint32_t* s_gbase = s_cnt + kPmWaves * 256;         // [256] where a digit's records begin inside the pass
    uint32_t* s_wave = s_gbase + 256;                   // [16]
    uint32_t* s_tabc = s_wave + 16;                     // [4][256] the workgroup's table entries
    uint32_t* s_tabl = s_tabc + kPmPassesPerWg * 256;   // [4][256]
    uint32_t* s_exc = s_tabl + kPmPassesPerWg * 256;    // [8][128][3] every wave's exceptions of the pass, until the pass is written out
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t mn = 0xFFFFFFFFu, mx = 0, bad = 0;
    auto contig_of = [&](uint32_t i) {
        uint32_t lo = 0, hi = n_contigs;  // last c with roff[c] <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (contig_read_off[mid] <= i) lo = mid; else hi = mid;
        }
        return lo;
    };
    for (int g = 0; g < kPmPassesPerWg; ++g) {
        const uint32_t P = blockIdx.x * kPmPassesPerWg + g;
        const uint64_t base64 = (uint64_t)P * kPmPass;
        if (base64 >= n) {  // (uniform) a pass beyond the reads: zero table entries, the scan runs over the padding too
            if (threadIdx.x < 256) { s_tabc[g * 256 + threadIdx.x] = 0; s_tabl[g * 256 + threadIdx.x] = 0; }
            if (ell_reg != 0u && lane == 0 && P < pitch) exc_cnt[P * kPmWaves + w] = 0;
            continue;
        }
        const uint32_t base = (uint32_t)base64;
        const uint32_t count = min((uint32_t)kPmPass, n - base);
        const uint32_t wbase = base + w * (kSortItems * 64);
        // all of the pass's loads first (32 in flight per thread)
        Rec rec[kSortItems];  // key: start, then global start; val: end
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t i = min(wbase + k * 64 + lane, n - 1);  // clamped: every lane loads
            rec[k].key = starts[i];
            rec[k].val = ends[i];
        }
        for (int i = threadIdx.x; i < kPmWaves * 256; i += kPmThreads) s_cnt[i] = 0;
        // the pass's 128 words of the keep mask are cleared here (saves a memset launch)
        if (zero_mask && threadIdx.x < 128 && P * 128u + threadIdx.x < (n + 63u) / 64u) zero_mask[P * 128u + threadIdx.x] = 0ull;
        const uint32_t c_first = n_contigs > 1 ? contig_of(base) : 0u;
        const uint32_t c_last = n_contigs > 1 ? contig_of(base + count - 1) : 0u;
        uint32_t match_bits;
        {
            // the pass's digits lie in the range its contigs span (k_range_partition, MODE 1)
            const uint32_t d_lo = ((uint32_t)contig_pos_off[c_first] >> shift) & 255u;
            const uint32_t d_hi = ((uint32_t)contig_pos_off[c_last + 1] >> shift) & 255u;
            match_bits = d_hi >= d_lo ? 32u - (uint32_t)__builtin_clz((d_hi - d_lo) | 1u) : 8u;
            if (d_hi == d_lo) match_bits = 0;
        }
        uint32_t skip = 0;  // bit k: the thread's k-th read is an exception (near-uniform route)
        uint32_t filled = 0;  // (uniform) exceptions of the pass this wave has met
        // An exception is put aside the moment it is recognised (its end is then dead: kept until later, the sixteen
        // ends cost sixteen registers and the kernel spilled), in the wave's own corner of LDS, and goes to the list
        // when the pass is written out -- a store to memory here would have to be drained at the next barrier.
        // (an LDS-qualified pointer: through a generic one the three stores become flat_ stores, which wait for the
        //  pass's thirty-two loads in flight -- 1 us per wave-round that holds an exception, 0.32 -> 0.69 ms)
        typedef __attribute__((address_space(3))) uint32_t LdsWord;
        LdsWord* const mine = (LdsWord*)(s_exc + w * (3 * kPmExcPerWave));
        auto put_aside = [&, mine](int k, bool isx, uint32_t gs, uint32_t ge, uint32_t i) {
            const uint64_t m = __ballot(isx);
            if (m != 0ull) {  // (uniform)
                const uint32_t slot = filled + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (isx) {
                    skip |= 1u << k;
                    if (slot < kPmExcPerWave) { mine[3u * slot] = gs; mine[3u * slot + 1u] = ge; mine[3u * slot + 2u] = i; }
                    else nu_overflow_put(exc, exc_cap, stats, gs, ge, i);  // (more than the wave's slots hold: the overflow region)
                }
                filled += (uint32_t)__popcll(m);
            }
        };
        // validate, span range, global start (a start beyond its contig -- the call fails -- is taken as the
        // contig's last position, so that its digit lies inside the pass's digit interval)
        if (c_first == c_last) {
            const uint32_t p0 = (uint32_t)contig_pos_off[c_first];
            const uint32_t len = (uint32_t)contig_pos_off[c_first + 1] - p0;
            const uint32_t last = len ? len - 1u : 0u;
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) {
                const uint32_t i = wbase + k * 64 + lane;
                const uint32_t s = rec[k].key, e = rec[k].val;
                bool isx = false;
                if (i < n) {
                    bad |= (s > e || e >= len) ? 1u : 0u;
                    const uint32_t span = e - s + 1;
                    mn = min(mn, span);
                    mx = max(mx, span);
                    isx = span != ell_reg;
                }
                rec[k].key = p0 + min(s, last);
                if (ell_reg != 0u) put_aside(k, isx, rec[k].key, p0 + min(e, last), i);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) {
                const uint32_t i = wbase + k * 64 + lane;
                bool isx = false;
                uint32_t ge = 0;
                if (i < n) {
                    const uint32_t cc = contig_of(i);
                    const uint32_t p0 = (uint32_t)contig_pos_off[cc];
                    const uint32_t len = (uint32_t)contig_pos_off[cc + 1] - p0;
                    const uint32_t s = rec[k].key, e = rec[k].val;
                    bad |= (s > e || e >= len) ? 1u : 0u;
                    const uint32_t span = e - s + 1;
                    mn = min(mn, span);
                    mx = max(mx, span);
                    rec[k].key = p0 + min(s, len ? len - 1u : 0u);
                    isx = span != ell_reg;
                    ge = p0 + min(e, len ? len - 1u : 0u);
                }
                if (ell_reg != 0u) put_aside(k, isx, rec[k].key, ge, i);
            }
        }
        __syncthreads();  // (counters cleared)
        uint32_t rank[kSortItems];
        {
            uint32_t* const s_cnt_w = s_cnt + w * 256;
            const uint32_t bound = base + count;
            switch (match_bits) {  // uniform
                case 0: part_rank_rounds<0>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 1: part_rank_rounds<1>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 2: part_rank_rounds<2>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 3: part_rank_rounds<3>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 4: part_rank_rounds<4>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 5: part_rank_rounds<5>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 6: part_rank_rounds<6>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                case 7: part_rank_rounds<7>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
                default: part_rank_rounds<8>(rec, rank, wbase, bound, shift, lane, s_cnt_w, skip); break;
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            // digit d = threadIdx.x: its total over the waves, then (below) where it begins inside the pass
            const uint32_t d = threadIdx.x;
            uint32_t tot = 0;
#pragma unroll
            for (int x = 0; x < kPmWaves; ++x) tot += s_cnt[x * 256 + d];
            const uint32_t inc = wave_incl_scan_add(tot);
            if (lane == 63) s_wave[w] = inc;
            s_gbase[d] = inc - tot;  // exclusive inside the wave; completed after the barrier
            s_tabc[g * 256 + d] = tot;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            const uint32_t d = threadIdx.x;
            uint32_t wave_base = 0;
            for (int x = 0; x < w; ++x) wave_base += s_wave[x];
            const uint32_t tile_off = s_gbase[d] + wave_base;
            uint32_t run = tile_off;
#pragma unroll
            for (int x = 0; x < kPmWaves; ++x) { const uint32_t cx = s_cnt[x * 256 + d]; s_cnt[x * 256 + d] = run; run += cx; }
            s_tabl[g * 256 + d] = tile_off;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t i = wbase + k * 64 + lane;
            if (i < n && !((skip >> k) & 1u)) {
                const uint32_t d = (rec[k].key >> shift) & 255u;
                const uint32_t local = (uint32_t)(w * (kSortItems * 64) + k * 64 + lane);  // < 8192
                s_stage[s_cnt[w * 256 + d] + rank[k]] = (rec[k].key & ((1u << shift) - 1u)) | (local << 16);
            }
        }
        __syncthreads();
        // out, two records per thread and store: a dword of two positions, a dword of two indices (base and j are even)
#pragma unroll
        for (int k = 0; k < kSortItems / 2; ++k) {
            const uint32_t j = 2u * (uint32_t)(k * kPmThreads + threadIdx.x);
            if (j < count) {
                const uint32_t v0 = s_stage[j];
                const uint32_t v1 = j + 1 < count ? s_stage[j + 1] : 0u;
                *reinterpret_cast<uint32_t*>(keys16 + (size_t)P * kPmStride + j) = (v0 & 0xFFFFu) | (v1 << 16);
                *reinterpret_cast<uint32_t*>(idx16 + (size_t)P * kPmStride + j) = (v0 >> 16) | (v1 & 0xFFFF0000u);
            }
        }
#ifndef QMCP_LAB_NO_EXC_OUT
        if (ell_reg != 0u) {
            const size_t slot0 = ((size_t)P * kPmWaves + w) * kPmExcPerWave;
            for (uint32_t r = (uint32_t)lane; r < min(filled, kPmExcPerWave); r += 64u) {
                exc[slot0 + r] = mine[3u * r];
                exc[exc_cap + slot0 + r] = mine[3u * r + 1u];
                exc[2 * (size_t)exc_cap + slot0 + r] = mine[3u * r + 2u];
            }
            if (lane == 0) {
                exc_cnt[P * kPmWaves + w] = min(filled, kPmExcPerWave);
            }
        }
#endif
        __syncthreads();  // (the next pass clears the counters and re-fills the stage)
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        // [digit][pass] tables: the workgroup's four passes are one aligned 16-byte run of the digit's row
        const uint32_t P0 = blockIdx.x * kPmPassesPerWg;
        if (P0 < pitch) {
            uint4 a, b;
            a.x = s_tabc[threadIdx.x]; a.y = s_tabc[256 + threadIdx.x]; a.z = s_tabc[512 + threadIdx.x]; a.w = s_tabc[768 + threadIdx.x];
            b.x = s_tabl[threadIdx.x]; b.y = s_tabl[256 + threadIdx.x]; b.z = s_tabl[512 + threadIdx.x]; b.w = s_tabl[768 + threadIdx.x];
            *reinterpret_cast<uint4*>(cnt_tab + (size_t)threadIdx.x * pitch + P0) = a;
            *reinterpret_cast<uint4*>(lst_tab + (size_t)threadIdx.x * pitch + P0) = b;
        }
    }
    // statistics: block reduction, at most one atomic per statistic per workgroup (k_prepare)
    __shared__ uint32_t s_red[3][kPmWaves];
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    bad = wave_max_u32(bad);
    if (lane == 0) { s_red[0][w] = mn; s_red[1][w] = mx; s_red[2][w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int x = 1; x < kPmWaves; ++x) { mn = min(mn, s_red[0][x]); mx = max(mx, s_red[1][x]); bad |= s_red[2][x]; }
        if (mn < __hip_atomic_load(&stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&stats[0], mn);
        if (mx > __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&stats[1], mx);
        if (bad) atomicOr(&stats[2], 1u);
    }
}

// where every range begins in flat coordinates (257 entries) and the heaviest range's load, from the scanned table
__global__ __launch_bounds__(256) void k_pm_range_table(const uint32_t* __restrict__ T, uint32_t pitch, uint32_t n,
                                                        uint32_t* __restrict__ range_start, uint32_t* __restrict__ max_load) {
    __shared__ uint32_t s_red[4];
    const uint32_t d = threadIdx.x;
    const uint32_t lo = T[(size_t)d * pitch];
    const uint32_t hi = T[(size_t)(d + 1) * pitch];  // (d == 255: the scan's total -- the records listed, which the
    range_start[d] = lo;                             //  near-uniform route makes fewer than the call's reads)
    if (d == 255) range_start[256] = hi;
    (void)n;
    const uint32_t m = wave_max_u32(hi - lo);
    if ((d & 63) == 0) s_red[d >> 6] = m;
    __syncthreads();
    if (d == 0) max_load[0] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}

// k_range_offsets for the pass-major layout: the range's positions come as slices, one per pass, in any order.
__global__ __launch_bounds__(1024) void k_pm_offsets(const uint16_t* __restrict__ keys16, const uint32_t* __restrict__ T,
                                                     const uint32_t* __restrict__ lst_tab, uint32_t pitch,
                                                     const uint32_t* __restrict__ rows /* [2][256]: p_lo, p_hi of every range (the host's pm_relevant_passes) */,
                                                     uint32_t shift, uint32_t ltot, uint32_t* __restrict__ boff,
                                                     uint32_t* __restrict__ empty_positions) {
    extern __shared__ uint32_t s_cnt32[];  // [(1 << shift) padded] counters, then [kPmMaxRow + 1] + [kPmMaxRow] row copies
#define PADDED(i) ((i) + ((i) >> 5))
    __shared__ uint32_t s_wsum[16], s_esum[16];
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    uint32_t* const s_T = s_cnt32 + width + (width >> 5) + 1;
    uint32_t* const s_L = s_T + kPmMaxRow + 1;
    const uint32_t lo = T[(size_t)range * pitch];
    const uint32_t p_lo = rows[range], p_hi = rows[256 + range];
    const uint32_t n_rel = min(p_hi - p_lo, kPmMaxRow);  // (the host takes this route only where no row is longer)
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) s_cnt32[PADDED(i)] = 0;
    for (uint32_t i = threadIdx.x; i <= n_rel; i += blockDim.x) s_T[i] = T[(size_t)range * pitch + p_lo + i];
    for (uint32_t i = threadIdx.x; i < n_rel; i += blockDim.x) s_L[i] = lst_tab[(size_t)range * pitch + p_lo + i];
    __syncthreads();
    auto count = [&](uint32_t li) {
        if (li < width) atomicAdd(&s_cnt32[PADDED(li)], 1u);
    };
    // wave w takes the slices w, w + 16, ...; four slices' loads in flight (records four at a time, 8-byte loads from
    // the 8-byte-aligned address at or below the slice: what lies outside the slice is the neighbouring ranges')
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // Two batches of four slices alternate: the next batch's loads are under way while this one's are counted
    // (single batches of eight left the wave idle for a trip to memory twelve to twenty-four times: 0.12 ms).
    constexpr int U = 4;
    struct Batch { uint32_t first[U], n_q[U], skip[U], end[U]; uint2 q[U][2]; };
    auto fetch = [&](Batch& b, uint32_t k0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = k0 + u * nw;
            const uint32_t cnt = k < n_rel ? s_T[k + 1] - s_T[k] : 0u;
            const uint32_t f = k < n_rel ? (p_lo + k) * kPmStride + s_L[k] : 0u;
            const uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);  // (uniform: scalar branches below)
            const uint32_t fu = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
            b.first[u] = fu & ~3u;                       // aligned record index the quads start at
            b.skip[u] = fu & 3u;                         // elements of the first quad that belong to the slice before
            b.end[u] = b.skip[u] + cu;                   // one past the slice's last element, counted from first[u]
            b.n_q[u] = cu ? (b.end[u] + 3u) >> 2 : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint2* __restrict__ quads = reinterpret_cast<const uint2*>(keys16 + b.first[u]);
            b.q[u][0] = lane < b.n_q[u] ? quads[lane] : make_uint2(0u, 0u);
            if (b.n_q[u] > 64u) b.q[u][1] = lane + 64u < b.n_q[u] ? quads[lane + 64u] : make_uint2(0u, 0u);
        }
    };
    auto consume = [&](const Batch& b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            auto count_quad = [&](uint2 v, uint32_t j) {
                const uint32_t e0 = 4u * j;  // element index of the quad's first, from first[u]
                if (e0 >= b.skip[u] && e0 < b.end[u]) count(v.x & 0xFFFFu);
                if (e0 + 1 >= b.skip[u] && e0 + 1 < b.end[u]) count(v.x >> 16);
                if (e0 + 2 >= b.skip[u] && e0 + 2 < b.end[u]) count(v.y & 0xFFFFu);
                if (e0 + 3 >= b.skip[u] && e0 + 3 < b.end[u]) count(v.y >> 16);
            };
            if (lane < b.n_q[u]) count_quad(b.q[u][0], lane);
            if (b.n_q[u] > 64u) {
                if (lane + 64u < b.n_q[u]) count_quad(b.q[u][1], lane + 64u);
                // slices longer than 128 quads (a pass whose reads fall into few ranges): the rest, plainly
                for (uint32_t j = lane + 128u; j < b.n_q[u]; j += 64u)
                    count_quad(reinterpret_cast<const uint2*>(keys16 + b.first[u])[j], j);
            }
        }
    };
    {
        Batch A, B;
        const uint32_t step = U * nw;
        uint32_t k0 = w;
        if (k0 < n_rel) fetch(A, k0);
        for (; k0 < n_rel; k0 += 2 * step) {
            if (k0 + step < n_rel) fetch(B, k0 + step);
            consume(A);
            if (k0 + 2 * step < n_rel) fetch(A, k0 + 2 * step);
            if (k0 + step < n_rel) consume(B);
        }
    }
    __syncthreads();
    // counts -> bucket offsets, in place (k_range_offsets)
    const uint32_t per = width >= 1024u ? width >> 10 : 1u;  // positions per thread
    const uint32_t firstp = threadIdx.x * per;
    uint32_t sum = 0, empties = 0;
    if (firstp < width)
        for (uint32_t qq = 0; qq < per; ++qq) {
            const uint32_t cq = s_cnt32[PADDED(firstp + qq)];
            sum += cq;
            empties += (cq == 0 && pos0 + firstp + qq < ltot) ? 1u : 0u;
        }
    const uint32_t inc = wave_incl_scan_add(sum);
    if (lane == 63) s_wsum[w] = inc;
    if (empty_positions != nullptr) {
        empties = wave_sum_u32(empties);
        if (lane == 0) s_esum[w] = empties;
    }
    __syncthreads();
    if (empty_positions != nullptr && threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t x = 0; x < nw; ++x) total += s_esum[x];
        if (total != 0) atomicAdd(empty_positions, total);
    }
    uint32_t run = lo + inc - sum;
    for (uint32_t x = 0; x < w; ++x) run += s_wsum[x];
    if (firstp < width)
        for (uint32_t qq = 0; qq < per; ++qq) {
            const uint32_t cq = s_cnt32[PADDED(firstp + qq)];
            s_cnt32[PADDED(firstp + qq)] = run;
            run += cq;
        }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x)
        if (pos0 + i <= ltot) boff[pos0 + i] = s_cnt32[PADDED(i)];
#undef PADDED
}

// A wave's 64 consecutive flat positions [x0, x0 + 64) of a range -> the slots of their records.  `cur` is the wave's
// cursor: the last pass (relative to p_lo) whose slice begins at or before the wave's previous first position; it
// only moves forward.  tests/pass_major_model.py: wave_cursor_walk.  Returns the lane's slot; its pass is slot >> 13.
__device__ __forceinline__ uint32_t pm_slot_of(const uint32_t* __restrict__ s_T, const uint32_t* __restrict__ s_L,
                                               uint32_t n_rel, uint32_t p_lo, uint32_t& cur, uint32_t x0, uint32_t x,
                                               uint32_t lane) {
    uint32_t cand;
    for (;;) {
        const uint32_t i = cur + 1u + lane;
        cand = i <= n_rel ? s_T[i] : 0xFFFFFFFFu;
        const uint32_t nb = (uint32_t)__popcll(__ballot(cand <= x0));  // slices that begin at or before x0
        cur += nb;
        if (nb < 64u) {
            if (nb != 0u) {  // (re-read relative to the cursor's new place)
                const uint32_t i2 = cur + 1u + lane;
                cand = i2 <= n_rel ? s_T[i2] : 0xFFFFFFFFu;
            }
            break;
        }
    }
    const uint32_t n_in = (uint32_t)__popcll(__ballot(cand <= x0 + 63u));  // borders inside the wave's positions
    uint32_t s = cur;
    if (n_in < 64u) {
        for (uint32_t t = 0; t < n_in; ++t) {  // uniform
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)t);
            s += x >= b ? 1u : 0u;
        }
    } else {
        // more than 63 borders under 64 positions (runs of empty slices): every lane searches the row
        uint32_t lo = cur, hi = n_rel;  // last k with s_T[k] <= x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_T[mid] <= x) lo = mid; else hi = mid;
        }
        s = lo;
    }
    s = min(s, n_rel - 1u);
    return (p_lo + s) * kPmStride + s_L[s] + (x - s_T[s]);
}

// Where the quotas come from when the event-driven sweep ran whole contigs (no stretch table): the kept count of
// position p is sev[p - (k - lastns[k]) * ell], k its block -- what k_sweep_expand would write into selend[] for the
// ranking to read back (30 us and 100 MB a solve; here the ranking reads the sweep's own output).
struct EvQuota { const uint32_t* sev; const uint32_t* lastns; const uint64_t* poff; uint32_t n_contigs, ell; };

// k_rank_mark for the pass-major layout (the walk, the quota protocol and the settling of quota-crossing groups
// are k_rank_mark's, word for word; what differs is where a record is found).
__global__ __launch_bounds__(1024) void k_pm_rank_mark(const uint16_t* __restrict__ keys16,
                                                       const uint16_t* __restrict__ idx16,
                                                       const uint32_t* __restrict__ T,
                                                       const uint32_t* __restrict__ lst_tab, uint32_t pitch,
                                                       const uint32_t* __restrict__ rows /* [2][256]: p_lo, p_hi of every range */,
                                                       uint32_t shift, uint32_t ltot, uint32_t n,
                                                       const uint32_t* __restrict__ boff,
                                                       const uint32_t* __restrict__ selend,
                                                       unsigned long long* __restrict__ mask,
                                                       unsigned long long* __restrict__ kept_total,
                                                       uint2* __restrict__ amb_lists, int lists_by_records,
                                                       uint32_t* __restrict__ chunk_cursor /* [n / 1024 + 256]: wave 0's cursor per chunk */,
                                                       EvQuota evq) {
    extern __shared__ int32_t s_q[];  // [(1 << shift) + 1] quotas; then the range's rows of T [kPmMaxRow + 1] and lst [kPmMaxRow]
    __shared__ uint32_t s_namb;
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    const uint32_t live = pos0 < ltot ? min(width, ltot - pos0) : 0u;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, nw = nthreads >> 6;
    const uint32_t lane = tid & 63u, w = tid >> 6;
    uint32_t* const s_T = reinterpret_cast<uint32_t*>(s_q) + width + 1;
    uint32_t* const s_L = s_T + kPmMaxRow + 1;
    const uint32_t lo = T[(size_t)range * pitch];
    const uint32_t hi = T[(size_t)(range + 1) * pitch];  // (range 255: the scan's total)
    if (lo >= hi) return;  // uniform: a range without reads needs no quotas either
    const uint32_t p_lo = rows[range], p_hi = rows[256 + range];
    const uint32_t n_rel = min(p_hi - p_lo, kPmMaxRow);
    uint2* const amb = amb_lists + (lists_by_records ? (size_t)lo : (size_t)range * width);
    uint32_t* const ccur = chunk_cursor + (lo >> 10) + range;  // (ranges' chunk counts add up to at most n / 1024 + one each)
    for (uint32_t i = tid; i <= n_rel; i += nthreads) s_T[i] = T[(size_t)range * pitch + p_lo + i];
    for (uint32_t i = tid; i < n_rel; i += nthreads) s_L[i] = lst_tab[(size_t)range * pitch + p_lo + i];
    if (evq.sev != nullptr) {
        // straight from the event-driven sweep's output (two dependent loads per position, eight positions in flight)
        for (uint32_t i0 = tid; i0 < live; i0 += 8 * nthreads) {
            uint32_t src[8], back[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t p = pos0 + min(i0 + u * nthreads, live - 1);  // clamped: every load is issued
                uint32_t c_lo = 0, c_hi = evq.n_contigs;  // last contig with poff[c] <= p
                while (c_hi - c_lo > 1) {
                    const uint32_t mid = (c_lo + c_hi) >> 1;
                    if ((uint32_t)evq.poff[mid] <= p) c_lo = mid; else c_hi = mid;
                }
                const uint32_t base = (uint32_t)evq.poff[c_lo];
                const uint32_t k = (p - base) / evq.ell;
                src[u] = p;
                back[u] = k - evq.lastns[(size_t)(base / (4u * evq.ell) + c_lo) * 4u + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) src[u] = evq.sev[src[u] - back[u] * evq.ell];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u * nthreads < live) s_q[i0 + u * nthreads] = (int32_t)src[u];
        }
    } else
    for (uint32_t i0 = tid; i0 < live; i0 += 8 * nthreads) {
        uint32_t a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t i = min(i0 + u * nthreads, live - 1);  // clamped: every load is issued
            a[u] = selend[pos0 + i];
            b[u] = boff[pos0 + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * nthreads < live) s_q[i0 + u * nthreads] = (int32_t)(a[u] - b[u]);
    }
    if (tid == 0) s_namb = 0;
    __syncthreads();
    const uint32_t chunk_recs = nthreads;
    const uint32_t n_chunks = (hi - lo + chunk_recs - 1) / chunk_recs;
    uint32_t kept = 0;
    uint32_t cur = 0;  // the wave's cursor over the range's row

    struct Recs { uint32_t key, val, slot; };
    // The wave's cursor and the row entries at it (uniform): slice `cur` begins at flat position t_cur, at l_cur inside
    // its pass.  A fetch reads the next 64 row entries (one LDS read per table), counts the slices that begin at or
    // before the wave's first position (the cursor moves on by that many) and those that begin inside its 64
    // positions; every lane then picks its slice's entries out of the candidate registers by lane number -- no
    // dependent LDS gathers.  More than 63 borders under one read (runs of empty slices): pm_slot_of, the plain way.
    uint32_t t_cur = s_T[0], l_cur = s_L[0];
    auto cand_read = [&](uint32_t& cand_t, uint32_t& cand_l) {
        const uint32_t i = cur + 1u + lane;
        cand_t = i <= n_rel ? s_T[i] : 0xFFFFFFFFu;
        cand_l = i < n_rel ? s_L[i] : 0u;
    };
    auto slot_from = [&](uint32_t cand_t, uint32_t cand_l, uint32_t c) -> uint32_t {
        const uint32_t x0 = min(lo + c * chunk_recs + 64u * w, hi - 1);
        const uint32_t x = min(lo + c * chunk_recs + tid, hi - 1);
        const uint32_t nb = (uint32_t)__popcll(__ballot(cand_t <= x0));
        const uint32_t tot = (uint32_t)__popcll(__ballot(cand_t <= x0 + 63u));
        uint32_t slot;
        if (tot < 64u) {
            if (nb != 0u) {
                t_cur = (uint32_t)__builtin_amdgcn_readlane((int)cand_t, (int)(nb - 1u));
                l_cur = (uint32_t)__builtin_amdgcn_readlane((int)cand_l, (int)(nb - 1u));
                cur += nb;
            }
            uint32_t ts = t_cur, ls = l_cur, sl = cur;
            for (uint32_t t = nb; t < tot; ++t) {  // uniform: the borders inside the wave's positions
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)cand_t, (int)t);
                const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)cand_l, (int)t);
                const bool in = x >= b;
                ts = in ? b : ts;
                ls = in ? l : ls;
                sl += in ? 1u : 0u;
            }
            sl = min(sl, n_rel - 1u);
            slot = (p_lo + cur) * kPmStride + __umul24(sl - cur, kPmStride) + ls + (x - ts);  // (uniform base + a small per-lane part)
        } else {
            slot = pm_slot_of(s_T, s_L, n_rel, p_lo, cur, x0, x, lane);
            t_cur = s_T[cur];
            l_cur = s_L[min(cur, n_rel - 1u)];
        }
        // (wave 0's cursor at the chunk's first position, for the settling pass below; read after the walk)
        if (w == 0 && lane == 0 && c < n_chunks) ccur[c] = cur;
#ifdef QMCP_LAB_PM_TRIVIAL_SLOT  // (lab: the look-up's arithmetic kept, its result replaced by the flat position -- wrong masks)
        slot = (slot & 1u) + (x & ~1u);
#endif
        return slot;
    };
    auto issue = [&](Recs& dst) {
        asm volatile("global_load_ushort %0, %2, %3\n\tglobal_load_ushort %1, %2, %4"
                     : "=&v"(dst.key), "=&v"(dst.val)
                     : "v"(dst.slot * 2u), "s"(keys16), "s"(idx16)
                     : "memory");
    };
    // One slot of the walk: chunk c is consumed from `r` while chunk c + kRankDepth - 1 is looked up and asked for
    // into `f`.  The look-up's LDS reads are issued before the quota draw and used after the first barrier, its
    // arithmetic runs between the barriers: the walk is bound by the latency of its two LDS round trips and two
    // barriers per chunk, and the look-up hides in them (as a block in front of the draw it cost 0.14 ms at cfg4).
    auto step = [&](Recs& r, Recs& f, uint32_t c) {
        uint32_t cand_t, cand_l;
#ifdef QMCP_LAB_PM_NO_LOOKUP  // (lab: no look-up at all, records read at their flat positions -- wrong masks)
        cand_t = 0xFFFFFFFFu; cand_l = 0;
#else
        cand_read(cand_t, cand_l);
#endif
        // the chunk's records have landed once at most the loads of the kRankDepth - 2 chunks asked for after it are
        // outstanding (this slot's own request comes below)
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.key), "+v"(r.val) : "n"(2 * (kRankDepth - 2)) : "memory");
        const bool valid = lo + c * chunk_recs + tid < hi;
        const uint32_t li = valid ? r.key : width;
        const int32_t old = atomicSub(&s_q[li], 1);
        // (the look-up's arithmetic sits between the draw and its barrier: it runs while the LDS atomic is under way)
#ifdef QMCP_LAB_PM_NO_LOOKUP
        f.slot = min(lo + (c + (uint32_t)(kRankDepth - 1)) * chunk_recs + tid, hi - 1);
#else
        f.slot = slot_from(cand_t, cand_l, c + (uint32_t)(kRankDepth - 1));
#endif
        __syncthreads();
        const int32_t aft = s_q[li];
        issue(f);
        const bool keep = valid && old > 0 && aft >= 0;
        if (valid && old == 1 && aft < 0) {
            const uint32_t k = atomicAdd(&s_namb, 1u);
            amb[k] = make_uint2((c << 15) | li, (uint32_t)(-aft));  // c < 2^17, li < 2^15
        }
        __syncthreads();  // every q_after is read before the next chunk draws
        if (keep) {
            const uint32_t v = (kPmStride == (uint32_t)kPmPass ? (r.slot & ~(uint32_t)(kPmPass - 1)) : (r.slot / kPmStride) * (uint32_t)kPmPass) + r.val;  // pass * 8192 + index in pass
            atomicOr(&mask[v >> 6], 1ull << (v & 63u));
        }
        kept += (uint32_t)__popcll(__ballot(keep));
    };
    Recs R[kRankDepth];
#pragma unroll
    for (int k = 0; k < kRankDepth - 1; ++k) {
        uint32_t cand_t, cand_l;
        cand_read(cand_t, cand_l);
        R[k].slot = slot_from(cand_t, cand_l, (uint32_t)k);
        issue(R[k]);
    }
    for (uint32_t c = 0; c < n_chunks; c += kRankDepth) {
#pragma unroll
        for (int k = 0; k < kRankDepth; ++k) step(R[k], R[(k + kRankDepth - 1) % kRankDepth], c + (uint32_t)k);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the loads asked for beyond the last chunk; ccur is written)
    // settle the listed (chunk, position) groups: one wave per entry, walking the chunk backwards
    __threadfence_block();
    __syncthreads();
    const uint32_t namb = s_namb;
    const uint64_t gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);  // lanes above this one
    for (uint32_t k = w; k < namb; k += nw) {
        const uint2 ent = amb[k];
        const uint32_t c = ent.x >> 15, p = ent.x & 0x7FFFu;
        uint32_t skip = ent.y;  // matches still to be passed over, from the chunk's end
        const uint32_t first = lo + c * chunk_recs;
        const uint32_t last = min(first + chunk_recs, hi);
        constexpr int kSteps = 16;  // blockDim.x == 1024: 64-record steps per chunk
        uint32_t key[kSteps], slot[kSteps];
        uint32_t scur = __hip_atomic_load(&ccur[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (uniform; own workgroup's store)
        scur = (uint32_t)__builtin_amdgcn_readfirstlane((int)scur);
        {
            // the chunk's 1024 positions cross few slices: ONE read of the 64 row entries behind the chunk's cursor serves
            // all sixteen steps from registers (more than 63 borders inside a chunk: step by step through pm_slot_of)
            const uint32_t ci = scur + 1u + lane;
            const uint32_t cand_t = ci <= n_rel ? s_T[ci] : 0xFFFFFFFFu;
            const uint32_t cand_l = ci < n_rel ? s_L[ci] : 0u;
            const uint32_t t0 = s_T[scur], l0 = s_L[min(scur, n_rel - 1u)];
            const bool few = (uint32_t)__popcll(__ballot(cand_t <= min(first + chunk_recs - 1u, hi - 1))) < 64u;
#pragma unroll
            for (int t = 0; t < kSteps; ++t) {
                const uint32_t x0 = min(first + t * 64u, hi - 1);
                const uint32_t x = min(first + t * 64u + lane, hi - 1);
                if (few) {
                    const uint32_t nb = (uint32_t)__popcll(__ballot(cand_t <= x0));
                    const uint32_t tot = (uint32_t)__popcll(__ballot(cand_t <= x0 + 63u));
                    uint32_t ts = nb ? (uint32_t)__builtin_amdgcn_readlane((int)cand_t, (int)(nb - 1u)) : t0;
                    uint32_t ls = nb ? (uint32_t)__builtin_amdgcn_readlane((int)cand_l, (int)(nb - 1u)) : l0;
                    uint32_t sl = scur + nb;
                    for (uint32_t u = nb; u < tot; ++u) {  // uniform: the borders inside the step's positions
                        const uint32_t bt = (uint32_t)__builtin_amdgcn_readlane((int)cand_t, (int)u);
                        const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)cand_l, (int)u);
                        const bool in = x >= bt;
                        ts = in ? bt : ts;
                        ls = in ? bl : ls;
                        sl += in ? 1u : 0u;
                    }
                    sl = min(sl, n_rel - 1u);
                    slot[t] = (p_lo + sl) * kPmStride + ls + (x - ts);
                } else {
                    slot[t] = pm_slot_of(s_T, s_L, n_rel, p_lo, scur, x0, x, lane);
                }
                key[t] = keys16[slot[t]];
            }
        }
#pragma unroll
        for (int t = kSteps - 1; t >= 0; --t) {
            const uint32_t j = first + t * 64u + lane;
            const bool member = j < last && key[t] == p;
            const uint64_t m = __ballot(member);
            if (m == 0) continue;
            const uint32_t above = (uint32_t)__popcll(m & gt_mask);  // matches after this one in the step
            if (member && above >= skip) {
                const uint32_t v = (kPmStride == (uint32_t)kPmPass ? (slot[t] & ~(uint32_t)(kPmPass - 1)) : (slot[t] / kPmStride) * (uint32_t)kPmPass) + idx16[slot[t]];
                atomicOr(&mask[v >> 6], 1ull << (v & 63u));
            }
            const uint32_t in_step = (uint32_t)__popcll(m);
            kept += in_step > skip ? in_step - skip : 0u;
            skip = skip > in_step ? skip - in_step : 0u;
        }
    }
    __syncthreads();
    if (tid == 0) s_namb = 0;
    __syncthreads();
    if (lane == 0 && kept) atomicAdd(&s_namb, kept);
    __syncthreads();
    if (tid == 0 && s_namb) atomicAdd(kept_total, (unsigned long long)s_namb);
}

// ---- launchers
uint32_t pm_pitch(uint32_t n) { return part_pass_pitch(n); }  // passes of the call, rounded up to a multiple of 4
uint32_t pm_max_row() { return kPmMaxRow; }
uint32_t pm_pass() { return (uint32_t)kPmPass; }
uint32_t pm_exc_slots(uint32_t n) { return pm_pitch(n) * kPmWaves * kPmExcPerWave + kNuOverflow; }  // the exception list's slots: 128 per wave and pass, and the overflow region
void launch_pm_prepare_sort(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs, uint32_t shift,
                            uint16_t* keys16, uint16_t* idx16, uint32_t* cnt_tab, uint32_t* lst_tab,
                            uint32_t* stats, unsigned long long* zero_mask, uint32_t ell_reg, uint32_t* exc,
                            uint32_t exc_cap, uint32_t* exc_cnt) {
    const uint32_t pitch = pm_pitch(n);
    if (pitch == 0) return;
    (void)hipFuncSetAttribute((const void*)k_pm_prepare_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPmSortLds);
    hipLaunchKernelGGL(k_pm_prepare_sort, dim3(pitch / kPmPassesPerWg), dim3(kPmThreads), kPmSortLds, st, starts, ends, n,
                       d_roff, d_poff, n_contigs, shift, keys16, idx16, cnt_tab, lst_tab, pitch, stats, zero_mask,
                       exc != nullptr ? ell_reg : 0u, exc, exc_cap, exc_cnt);
    if (exc != nullptr && ell_reg != 0u)
        hipLaunchKernelGGL(k_nu_count_groups, dim3(32), dim3(256), 0, st, exc_cnt, pitch * kPmWaves, stats);
}
void launch_pm_range_table(hipStream_t st, const uint32_t* T, uint32_t n, uint32_t* range_start, uint32_t* max_load) {
    hipLaunchKernelGGL(k_pm_range_table, dim3(1), dim3(256), 0, st, T, pm_pitch(n), n, range_start, max_load);
}
void launch_pm_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* T, const uint32_t* lst_tab, uint32_t n,
                       const uint32_t* rows, uint32_t shift, uint32_t ltot, uint32_t* boff, uint32_t* empty_positions) {
    const uint32_t n_ranges = (ltot >> shift) + 1;  // covers positions 0..ltot
    const size_t width = (size_t)1 << shift;
    const size_t lds = (width + width / 32 + 1 + 2 * (size_t)kPmMaxRow + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_pm_offsets, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_pm_offsets, dim3(n_ranges), dim3(1024), lds, st, keys16, T, lst_tab, pm_pitch(n), rows, shift, ltot,
                       boff, empty_positions);
}
void launch_pm_rank_mark(hipStream_t st, const uint16_t* keys16, const uint16_t* idx16, const uint32_t* T,
                         const uint32_t* lst_tab, uint32_t n, const uint32_t* rows, uint32_t shift, uint32_t ltot,
                         const uint32_t* boff, const uint32_t* selend,
                         unsigned long long* mask, unsigned long long* kept_total, void* scratch, bool scratch_by_records,
                         uint32_t* chunk_cursor, const uint32_t* ev_sev, const uint32_t* ev_lastns,
                         const uint64_t* d_poff, uint32_t n_contigs, uint32_t ell) {
    const uint32_t n_ranges = (ltot >> shift) + 1;
    const EvQuota evq{ev_sev, ev_lastns, d_poff, n_contigs, ell};
    const size_t lds = (((size_t)1 << shift) + 1 + 2 * (size_t)kPmMaxRow + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_pm_rank_mark, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_pm_rank_mark, dim3(n_ranges), dim3(1024), lds, st, keys16, idx16, T, lst_tab, pm_pitch(n), rows,
                       shift, ltot, n, boff, selend, mask, kept_total, (uint2*)scratch,
                       scratch_by_records ? 1 : 0, chunk_cursor, evq);
}
