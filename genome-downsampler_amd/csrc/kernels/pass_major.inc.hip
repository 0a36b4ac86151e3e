// pass_major.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ range-ranked route, pass-major layout
// The range-ranked route (ranked_route.inc.hip) needs the reads grouped by position range, in read-index order
// inside a range.  Its first form reads the reads twice before it can write them -- k_prepare (histogram of every
// pass of 8 192 reads), a scan of the histograms, k_range_partition (re-reads the starts, writes 6 B per read to
// where the scan says) -- because a range-major array needs every pass's counts before the first record can be
// placed.  Here the grouped array is never materialised (host model with every index asserted in bounds:
// tests/pass_major_model.py, tests/test_pass_major_model.py):
//   k_pm_prepare_sort   ONE pass over the reads (8 B per read in): validate, span statistics, clear the keep mask,
//                       sort every pass of 8 192 reads by range; the pass's records of range d -- its SLICE (d, P) --
//                       leave as two 16-bit streams (position inside the range, read index inside the pass: 4 B per
//                       read) at a slot that is a multiple of 64: pass P owns the slots [P stride, (P + 1) stride),
//                       stride = 8192 + 64 n_ranges, the slices follow each other padded to whole groups of 64 slots.
//                       Two tables laid out [range][pass]: the slice's record count rounded up to a multiple of 64, and
//                       (first slot - P stride) / 64 | true count << 16.
//   (scan)              exclusive scan over the padded count table: Tp[d][P] = the PADDED FLAT position of the slice --
//                       range d's records in read-index order are its slices in pass order, each padded to whole
//                       groups, so a group of 64 padded flat positions (a WAVE-SLOT) lies in exactly one slice.
//   k_pm_descr          one thread per table entry: one descriptor word per wave-slot -- pass << 15 | slot group inside the
//                       pass << 6 | (records in the group - 1) --; the rows' true record counts (k_pm_range_table scans
//                       them: the ranges' true flat starts -- the bucket offsets' bases -- and the heaviest load).
//   k_pm_offsets        per range: LDS histogram of its wave-slots' positions -> bucket offsets (k_range_offsets' job).
//   k_pm_walk           per range: k_rank_mark's ordered walk, a chunk of sixteen wave-slots per step, each wave's
//                       records found through ONE descriptor word (round 3 searched the table's row with a cursor in
//                       LDS: 57 vector instructions per wave and chunk against the range-major walk's 21).  Kept
//                       records' read indices collect in a ring in LDS, per wave, and are marked 64 at a time.
//   k_pm_settle         the (chunk, position) groups whose quota ran out inside a chunk, one wave per group, chip-wide.
// A range's records in read-index order are its slices in pass order, so nothing about the selection changes: the
// kept set is bit for bit the first form's.  One-level genomes only (<= 256 ranges); longer ones keep the two-level
// partition, and so do calls whose slices would be short (narrow ranges: the padding would be most of a group).
static constexpr int kPmPass = 8192;             // reads per pass
// slots between the beginnings of two passes: the pass's records and up to 63 slots of padding per range
__host__ __device__ inline uint32_t pm_stride_of(uint32_t n_ranges) { return (uint32_t)kPmPass + 64u * n_ranges; }
static constexpr int kPmThreads = 512;           // 8 waves: wave w owns records [1024 w, 1024 (w + 1)) of the pass
static constexpr int kPmWaves = kPmThreads / 64;
static constexpr int kPmPassesPerWg = 4;         // a workgroup's passes leave their table entries as 16-byte runs
static constexpr size_t kPmSortLds = ((size_t)kPmPass + kPmWaves * 256 + 2 * 256 + 16 + 2 * kPmPassesPerWg * 256 + kPmWaves * 3 * 128) * sizeof(uint32_t);
static constexpr uint32_t kPmExcPerWave = 128;  // list slots per wave and pass: an eighth of the wave's 1 024 reads

#ifndef QMCP_PM_MIN_WAVES
#define QMCP_PM_MIN_WAVES 4  // waves per SIMD the register allocation aims at (6 -- three workgroups per CU -- spills: 0.42 against 0.29 ms)
#endif
__global__ __launch_bounds__(kPmThreads, QMCP_PM_MIN_WAVES) void k_pm_prepare_sort(
    const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends, uint32_t n,
    const uint64_t* __restrict__ contig_read_off, const uint64_t* __restrict__ contig_pos_off, uint32_t n_contigs,
    uint32_t shift, uint32_t stride /* pm_stride_of(ranges of the genome) */,
    uint16_t* __restrict__ keys16, uint16_t* __restrict__ idx16,
    uint32_t* __restrict__ cnt_tab, uint32_t* __restrict__ lst_tab, uint32_t pitch /* multiple of 4 */,
    uint32_t* __restrict__ work /* 260 words k_pm_descr adds into: cleared here */,
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
    uint32_t* s_gbase = s_cnt + kPmWaves * 256;         // [256] where a digit's records begin inside the sorted pass
    uint32_t* s_pbase = s_gbase + 256;                  // [256] ... and where its slice begins inside the pass's slots (padded)
    uint32_t* s_wave = s_pbase + 256;                   // [16]
    uint32_t* s_tabc = s_wave + 16;                     // [4][256] the workgroup's table entries
    uint32_t* s_tabl = s_tabc + kPmPassesPerWg * 256;   // [4][256]
    uint32_t* s_exc = s_tabl + kPmPassesPerWg * 256;    // [8][128][3] every wave's exceptions of the pass, until the pass is written out
    if (blockIdx.x == 0 && threadIdx.x < 260) work[threadIdx.x] = 0;  // (k_pm_descr's row sums and ticket)
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
            // digit d = threadIdx.x: its total over the waves, then (below) where it begins inside the sorted pass and
            // where its slice -- padded to whole groups of 64 slots -- begins inside the pass's slots
            const uint32_t d = threadIdx.x;
            uint32_t tot = 0;
#pragma unroll
            for (int x = 0; x < kPmWaves; ++x) tot += s_cnt[x * 256 + d];
            const uint32_t ptot = (tot + 63u) & ~63u;
            const uint32_t inc = wave_incl_scan_add(tot);
            const uint32_t pinc = wave_incl_scan_add(ptot);
            if (lane == 63) { s_wave[w] = inc; s_wave[8 + w] = pinc; }
            s_gbase[d] = inc - tot;  // exclusive inside the wave; completed after the barrier
            s_pbase[d] = pinc - ptot;
            s_tabc[g * 256 + d] = ptot;
            s_tabl[g * 256 + d] = tot << 16;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            const uint32_t d = threadIdx.x;
            uint32_t wave_base = 0, pwave_base = 0;
            for (int x = 0; x < w; ++x) { wave_base += s_wave[x]; pwave_base += s_wave[8 + x]; }
            const uint32_t tile_off = s_gbase[d] + wave_base;
            const uint32_t ptile_off = s_pbase[d] + pwave_base;
            uint32_t run = tile_off;
#pragma unroll
            for (int x = 0; x < kPmWaves; ++x) { const uint32_t cx = s_cnt[x * 256 + d]; s_cnt[x * 256 + d] = run; run += cx; }
            s_gbase[d] = tile_off;
            s_pbase[d] = ptile_off;
            s_tabl[g * 256 + d] |= ptile_off >> 6;   // (< 2^16: a pass has at most 8192 / 64 + 256 slot groups)
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
        // out, slice by slice: wave w takes the digits w, w + 8, ..., two at a time -- one per half-wave --, four records
        // per lane and store (8 bytes of positions, 8 of indices: slices begin at multiples of 64 slots, so every store is
        // aligned and a half-wave's store is one 256-byte run)
        {
            const uint32_t my_d = (uint32_t)w + 8u * ((uint32_t)lane & 31u);
            const uint32_t my_tot = lane < 32 ? s_tabl[g * 256 + my_d] >> 16 : 0u;
            const uint32_t my_gb = s_gbase[my_d], my_pb = s_pbase[my_d];
            const bool upper = lane >= 32;
            const uint32_t hl = (uint32_t)lane & 31u;
            uint64_t todo = __ballot(my_tot != 0u);
            while (todo != 0ull) {  // uniform
                const int l0 = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                int l1 = l0;
                bool two = false;
                if (todo != 0ull) { l1 = __builtin_ctzll(todo); todo &= todo - 1ull; two = true; }
                const uint32_t cnt0 = (uint32_t)__builtin_amdgcn_readlane((int)my_tot, l0);
                const uint32_t cnt1 = two ? (uint32_t)__builtin_amdgcn_readlane((int)my_tot, l1) : 0u;
                const uint32_t cnt = upper ? cnt1 : cnt0;
                const uint32_t gb = upper ? (uint32_t)__builtin_amdgcn_readlane((int)my_gb, l1) : (uint32_t)__builtin_amdgcn_readlane((int)my_gb, l0);
                const uint32_t pb = upper ? (uint32_t)__builtin_amdgcn_readlane((int)my_pb, l1) : (uint32_t)__builtin_amdgcn_readlane((int)my_pb, l0);
                const size_t out0 = (size_t)P * stride + pb;
                const uint32_t most = max(cnt0, cnt1);
                for (uint32_t j0 = 0; j0 < most; j0 += 128u) {  // uniform
                    const uint32_t j = j0 + 4u * hl;
                    if (j < cnt) {
                        const uint32_t v0 = s_stage[gb + j];
                        const uint32_t v1 = j + 1 < cnt ? s_stage[gb + j + 1] : 0u;
                        const uint32_t v2 = j + 2 < cnt ? s_stage[gb + j + 2] : 0u;
                        const uint32_t v3 = j + 3 < cnt ? s_stage[gb + j + 3] : 0u;
                        *reinterpret_cast<uint2*>(keys16 + out0 + j) = make_uint2((v0 & 0xFFFFu) | (v1 << 16), (v2 & 0xFFFFu) | (v3 << 16));
                        *reinterpret_cast<uint2*>(idx16 + out0 + j) = make_uint2((v0 >> 16) | (v1 & 0xFFFF0000u), (v2 >> 16) | (v3 & 0xFFFF0000u));
                    }
                }
            }
        }
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

// k_pm_descr's working words (cleared by k_pm_prepare_sort): [0..255] the rows' true record counts
static constexpr uint32_t kPmWorkWords = 260;
static constexpr uint32_t kPmDescrY = 8;  // workgroups per range

// A wave-slot's descriptor: pass << 15 | slot group inside the pass << 6 | (records - 1).  (2^17 passes: 2^30 reads; 512
// slot groups: a pass holds at most 8192 / 64 + 256.)
struct PmSlot { uint32_t slot0, nv, pass; };
// (kVector: the descriptor sits in a vector register -- the 24-bit multiply is the full-rate one there; a uniform
//  descriptor goes through the scalar unit, which has the plain multiply only)
template <bool kVector>
__device__ __forceinline__ PmSlot pm_unpack(uint32_t dsc, bool has, uint32_t stride) {
    PmSlot s;
    s.pass = dsc >> 15;
    s.slot0 = has ? (kVector ? __umul24(s.pass, stride) : s.pass * stride) + (((dsc >> 6) & 511u) << 6) : 0u;
    s.nv = has ? (dsc & 63u) + 1u : 0u;
    return s;
}

// A thread per (range, pass) table entry: the slice's wave-slot descriptors; per range the true record count (one atomic
// per workgroup; kPmDescrY workgroups a range).  (A first form had a workgroup per 256 entries, every one ending in an
// atomic on one ticket word: 12 288 same-address atomics, 0.49 ms; one workgroup per range with a ticket: 0.044 ms, a
// serial loop of dependent loads per thread.)
__global__ __launch_bounds__(256) void k_pm_descr(const uint32_t* __restrict__ Tp, const uint32_t* __restrict__ lstw,
                                                  uint32_t pitch, uint32_t n_groups /* total padded flat / 64 bound */,
                                                  uint32_t* __restrict__ desc, uint32_t* __restrict__ work) {
    __shared__ uint32_t s_red[4];
    const uint32_t d = blockIdx.x;
    const uint32_t per = (pitch + gridDim.y - 1u) / gridDim.y;
    const uint32_t P_end = min(pitch, (blockIdx.y + 1u) * per);
    uint32_t mine = 0;
    for (uint32_t P0 = blockIdx.y * per + threadIdx.x; P0 < P_end; P0 += 4u * 256u) {
        uint32_t w[4], t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t P = min(P0 + (uint32_t)u * 256u, P_end - 1u);  // clamped: every load is issued
            w[u] = lstw[(size_t)d * pitch + P];
            t[u] = Tp[(size_t)d * pitch + P];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t P = P0 + (uint32_t)u * 256u;
            const uint32_t cnt = P < P_end ? w[u] >> 16 : 0u;
            if (cnt == 0u) continue;
            mine += cnt;
            const uint32_t g = t[u] >> 6;
            const uint32_t n_ws = (cnt + 63u) >> 6;
            for (uint32_t j = 0; j < n_ws; ++j)
                if (g + j < n_groups)  // (always: the bound is the buffer's size)
                    desc[g + j] = (P << 15) | (((w[u] & 0xFFFFu) + j) << 6) | (min(64u, cnt - 64u * j) - 1u);
        }
    }
    const uint32_t sum = wave_sum_u32(mine);
    if ((threadIdx.x & 63u) == 0u) s_red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (tot != 0u) atomicAdd(&work[d], tot);
    }
}

// the ranges' true flat starts (257 entries) and the heaviest range's load, from the rows' true counts
__global__ __launch_bounds__(256) void k_pm_range_table(const uint32_t* __restrict__ work, uint32_t* __restrict__ range_start,
                                                        uint32_t* __restrict__ max_load) {
    __shared__ uint32_t s_red[4];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t c = work[threadIdx.x];
    const uint32_t inc = wave_incl_scan_add(c);
    const uint32_t mx = wave_max_u32(c);
    if (lane == 63u) s_red[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t x = 0; x < wv; ++x) base += s_red[x];
    range_start[threadIdx.x] = base + inc - c;
    if (threadIdx.x == 255u) range_start[256] = base + inc;
    __syncthreads();
    if (lane == 0u) s_red[wv] = mx;
    __syncthreads();
    if (threadIdx.x == 0) max_load[0] = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
}

// k_range_offsets for the pass-major layout: the range's positions come as wave-slots, in any order.  A wave takes four
// wave-slots at a time -- sixteen lanes each, four records (8 bytes) per lane: one 512-byte request -- and keeps kU
// such requests in flight.
__global__ __launch_bounds__(1024) void k_pm_offsets(const uint16_t* __restrict__ keys16, const uint32_t* __restrict__ desc,
                                                     const uint32_t* __restrict__ Tp, uint32_t pitch,
                                                     const uint32_t* __restrict__ range_start /* true flat starts */,
                                                     uint32_t shift, uint32_t stride, uint32_t ltot, uint32_t* __restrict__ boff,
                                                     uint32_t* __restrict__ empty_positions) {
    extern __shared__ uint32_t s_cnt32[];  // [(1 << shift) padded] counters
#define PADDED(i) ((i) + ((i) >> 5))
    __shared__ uint32_t s_wsum[16], s_esum[16];
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    const uint32_t lo_p = Tp[(size_t)range * pitch], hi_p = Tp[(size_t)(range + 1) * pitch];
    const uint32_t g0 = lo_p >> 6, n_ws = (hi_p - lo_p) >> 6;
    const uint32_t lo = range_start[range];
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) s_cnt32[PADDED(i)] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, nw = blockDim.x >> 6;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t sub = lane >> 4, l16 = lane & 15u;
    constexpr int kU = 4;
    const uint32_t n_quads = (n_ws + 3u) >> 2;
    for (uint32_t q0 = w; q0 < n_quads; q0 += kU * nw) {
        uint2 v[kU];
        uint32_t nv[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t q = q0 + (uint32_t)u * nw;
            // the lane's wave-slot of the quad: its descriptor (four scalar words, the lane's one picked out)
            uint32_t dsc = 0;
            bool has = false;
            if (q < n_quads) {  // uniform
                const uint32_t ws = 4u * q;
                const uint32_t d0 = desc[g0 + ws];
                const uint32_t d1 = ws + 1 < n_ws ? desc[g0 + ws + 1] : 0u;
                const uint32_t d2 = ws + 2 < n_ws ? desc[g0 + ws + 2] : 0u;
                const uint32_t d3 = ws + 3 < n_ws ? desc[g0 + ws + 3] : 0u;
                dsc = sub == 0 ? d0 : sub == 1 ? d1 : sub == 2 ? d2 : d3;
                has = ws + sub < n_ws;
            }
            const PmSlot at = pm_unpack<true>(dsc, has, stride);
            nv[u] = at.nv;
            v[u] = *reinterpret_cast<const uint2*>(keys16 + at.slot0 + 4u * l16);   // (inside the slice's padded group: always readable)
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t e = 4u * l16;
            if (e < nv[u]) atomicAdd(&s_cnt32[PADDED(v[u].x & 0xFFFFu)], 1u);
            if (e + 1 < nv[u]) atomicAdd(&s_cnt32[PADDED(v[u].x >> 16)], 1u);
            if (e + 2 < nv[u]) atomicAdd(&s_cnt32[PADDED(v[u].y & 0xFFFFu)], 1u);
            if (e + 3 < nv[u]) atomicAdd(&s_cnt32[PADDED(v[u].y >> 16)], 1u);
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
    const uint32_t wv = threadIdx.x >> 6;
    if (lane == 63) s_wsum[wv] = inc;
    if (empty_positions != nullptr) {
        empties = wave_sum_u32(empties);
        if (lane == 0) s_esum[wv] = empties;
    }
    __syncthreads();
    if (empty_positions != nullptr && threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t x = 0; x < nw; ++x) total += s_esum[x];
        if (total != 0) atomicAdd(empty_positions, total);
    }
    uint32_t run = lo + inc - sum;
    for (uint32_t x = 0; x < wv; ++x) run += s_wsum[x];
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

// Where the quotas come from when the event-driven sweep ran whole contigs (no stretch table): the kept count of
// position p is sev[p - (k - lastns[k]) * ell], k its block -- what k_sweep_expand would write into selend[] for the
// ranking to read back (30 us and 100 MB a solve; here the ranking reads the sweep's own output).
struct EvQuota { const uint32_t* sev; const uint32_t* lastns; const uint64_t* poff; uint32_t n_contigs, ell; };

// k_rank_mark's ordered walk for the pass-major layout: one workgroup (16 waves) per range, quota array q[p] = S(p) in
// LDS, the range's wave-slots in order, sixteen (a CHUNK: one per wave) per step: `old = q[p]--`, barrier, `aft = q[p]`,
// barrier; kept iff old > 0 -- except where the quota runs out inside the chunk (old > 0 but aft < 0: the draws of one
// chunk come in no particular order): those groups are listed (chunk, position, -aft) by the record that drew
// old == 1 and settled by k_pm_settle.  A wave's records are the 64 slots its descriptor names -- one word, asked for
// fifteen chunks ahead -- and their positions and read indices are asked for seven chunks ahead, issued and waited for by
// hand.  Kept records: the wave collects their read indices in a ring of 128 words in LDS and marks them 64 at a time --
// one atomic instruction every ~20 chunks instead of one per chunk: atomics and stores share the loads' counter, and a
// counted wait also waits for everything of theirs that is still on its way.
__global__ __launch_bounds__(1024) void k_pm_walk(const uint16_t* __restrict__ keys16, const uint16_t* __restrict__ idx16,
                                                  const uint32_t* __restrict__ desc,
                                                  const uint32_t* __restrict__ Tp, uint32_t pitch,
                                                  const uint32_t* __restrict__ range_start /* true flat starts */,
                                                  uint32_t shift, uint32_t stride, uint32_t ltot,
                                                  const uint32_t* __restrict__ boff, const uint32_t* __restrict__ selend,
                                                  EvQuota evq,
                                                  unsigned long long* __restrict__ mask,
                                                  uint2* __restrict__ amb_lists, int lists_by_records,
                                                  uint32_t* __restrict__ amb_count /* [256] */,
                                                  unsigned long long* __restrict__ kept_total) {
    extern __shared__ int32_t s_q[];  // [(1 << shift) + 1] quotas
    __shared__ uint32_t s_namb, s_kept;
    __shared__ uint32_t s_ring[16][128];  // per wave: kept records' read indices on their way to the mask
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    const uint32_t live = pos0 < ltot ? min(width, ltot - pos0) : 0u;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
    const uint32_t lane = tid & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t lo_p = Tp[(size_t)range * pitch];
    const uint32_t hi_p = Tp[(size_t)(range + 1) * pitch];  // (range 255: the scan's total)
    if (tid == 0) amb_count[range] = 0;
    if (lo_p >= hi_p) return;  // uniform: a range without reads needs no quotas either
    const uint32_t g0 = lo_p >> 6, n_ws = (hi_p - lo_p) >> 6;
    const uint32_t n_chunks = (n_ws + 15u) >> 4;
    const uint32_t lo_true = range_start[range];
    uint2* const amb = amb_lists + (lists_by_records ? (size_t)lo_true : (size_t)range * width);
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
    if (tid == 0) { s_namb = 0; s_kept = 0; }
    __syncthreads();

    // The loads in flight -- positions and descriptors of the chunks ahead -- land in registers the COMPILER NEVER SEES:
    // v96..v103 (positions of the chunk consumed at slot k of the unrolled loop), v104..v111 (descriptors), v112..v119 (read
    // indices inside the pass).  They are named in the assembly, issued and waited for by hand, and every
    // assembly statement of the kernel lists all of them as clobbered, so the compiler keeps nothing of its own there.
    // The first form of this kernel held them in compiler-allocated registers, as k_rank_mark does: the compiler then
    // kept a loop-carried descriptor in another register than the one its load writes and COPIED it at the loop's end --
    // a copy of a register whose load is still in flight, i.e. of what was there before: cfg4 came out different from
    // run to run, one run ended in a memory access fault (a stale descriptor's slots); small inputs never showed it.
    // tools/isa_hazards.py checks the assembly for such reads and that v96..v119 occur in hand-written assembly only
    // (tests/test_isa_hazards.py).
#define QMCP_PM_RING "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119"
#define QMCP_PM_KEYREG(k) "v" QMCP_PM_STR(QMCP_PM_CAT(QMCP_PM_KEY_, k))
#define QMCP_PM_DSCREG(k) "v" QMCP_PM_STR(QMCP_PM_CAT(QMCP_PM_DSC_, k))
#define QMCP_PM_IDXREG(k) "v" QMCP_PM_STR(QMCP_PM_CAT(QMCP_PM_IDX_, k))
#define QMCP_PM_STR(x) QMCP_PM_STR2(x)
#define QMCP_PM_STR2(x) #x
#define QMCP_PM_CAT(a, b) QMCP_PM_CAT2(a, b)
#define QMCP_PM_CAT2(a, b) a##b
#define QMCP_PM_KEY_0 96
#define QMCP_PM_KEY_1 97
#define QMCP_PM_KEY_2 98
#define QMCP_PM_KEY_3 99
#define QMCP_PM_KEY_4 100
#define QMCP_PM_KEY_5 101
#define QMCP_PM_KEY_6 102
#define QMCP_PM_KEY_7 103
#define QMCP_PM_DSC_0 104
#define QMCP_PM_DSC_1 105
#define QMCP_PM_DSC_2 106
#define QMCP_PM_DSC_3 107
#define QMCP_PM_DSC_4 108
#define QMCP_PM_DSC_5 109
#define QMCP_PM_DSC_6 110
#define QMCP_PM_DSC_7 111
#define QMCP_PM_IDX_0 112
#define QMCP_PM_IDX_1 113
#define QMCP_PM_IDX_2 114
#define QMCP_PM_IDX_3 115
#define QMCP_PM_IDX_4 116
#define QMCP_PM_IDX_5 117
#define QMCP_PM_IDX_6 118
#define QMCP_PM_IDX_7 119
    static_assert(kRankDepth == 8, "the ring registers are named for eight slots");
    struct Slot { uint32_t slot0, nv, read0; bool has; };  // of a chunk whose records are in flight or being consumed (uniform; the compiler's)
    auto desc_offset = [&](uint32_t c) -> uint32_t {  // byte offset of the wave's descriptor of chunk c (the range's last beyond it: never used)
        return (g0 + min(16u * c + w, n_ws - 1u)) * 4u;
    };
    auto slot_of = [&](uint32_t dsc, uint32_t c) -> Slot {
        const bool has = 16u * c + w < n_ws;  // uniform
        // (on the vector unit, although every lane holds the same word: through v_readfirstlane and the scalar unit --
        //  fewer vector instructions -- the walk took 0.285 instead of 0.255 ms at cfg4: the scalar chain's dependent
        //  multi-cycle instructions sit in front of the requests for the chunk seven steps ahead)
        const PmSlot at = pm_unpack<true>(dsc, has, stride);
        Slot s;
        s.slot0 = at.slot0;
        s.nv = at.nv;
        s.read0 = at.pass * (uint32_t)kPmPass;  // the pass's first read
        s.has = has;
        return s;
    };
    uint32_t kept = 0;  // (set from `cur` after the walk)
    // kept read indices collected / marked so far (uniform)
    uint32_t cur = 0, flushed = 0;
    uint32_t* const ring = s_ring[w];
    // One slot of the walk (K: its place in the unrolled loop, KF = (K + 7) % 8): chunk c is consumed from ring slot K;
    // chunk c + 7's positions are asked for into slot KF through slot K's descriptor (asked for eight slots ago), and
    // slot K's descriptor is asked for again, for chunk c + 15.
    // Loads complete in issue order.  Per slot three are issued: positions, read indices, a descriptor.  The read indices
    // of chunk c were asked for seven slots ago and 19 loads have been issued since; its positions and the descriptor read
    // with them are older.  (Younger stores and atomics only make the wait longer.)
#define QMCP_PM_STEP(K, KF)                                                                                                   \
    {                                                                                                                         \
        const uint32_t c = c0 + (uint32_t)(K);                                                                                \
        uint32_t key, dsc, idx;                                                                                               \
        asm volatile("s_waitcnt vmcnt(19)\n\tv_mov_b32 %0, " QMCP_PM_KEYREG(K) "\n\tv_mov_b32 %1, " QMCP_PM_DSCREG(K)          \
                     "\n\tv_mov_b32 %2, " QMCP_PM_IDXREG(K)                                                                   \
                     : "=v"(key), "=v"(dsc), "=v"(idx) : : QMCP_PM_RING, "memory");                                                      \
        const Slot at = S[K];                                                                                                 \
        const bool valid = lane < at.nv;                                                                                      \
        int32_t old = 0;                                                                                                      \
        if (valid) old = atomicSub(&s_q[key], 1); /* (the 16-bit load zero-extends) */                                        \
        S[KF] = slot_of(dsc, c + 7u);                                                                                         \
        asm volatile("global_load_ushort " QMCP_PM_KEYREG(KF) ", %0, %1\n\tglobal_load_ushort " QMCP_PM_IDXREG(KF) ", %0, %2"   \
                     "\n\tglobal_load_dword " QMCP_PM_DSCREG(K) ", %3, %4"                                                    \
                     : : "v"((S[KF].slot0 + lane) * 2u), "s"(keys16), "s"(idx16), "v"(desc_offset(c + 15u)), "s"(desc)        \
                     : QMCP_PM_RING, "memory");                                                                               \
        __syncthreads();                                                                                                      \
        int32_t aft = 0;                                                                                                      \
        if (valid) aft = s_q[key];                                                                                            \
        const bool keep = valid && old > 0 && aft >= 0;                                                                       \
        if (valid && old == 1 && aft < 0) {                                                                                   \
            const uint32_t k = atomicAdd(&s_namb, 1u);                                                                        \
            amb[k] = make_uint2((c << 15) | key, (uint32_t)(-aft)); /* c < 2^17, key < 2^15 */                                \
        }                                                                                                                     \
        __syncthreads(); /* every q_after is read before the next chunk draws */                                              \
        if (at.has) { /* uniform: the wave has a wave-slot in this chunk */                                                   \
            const uint64_t kb = __ballot(keep);                                                                               \
            const uint32_t cw = (uint32_t)__popcll(kb);                                                                       \
            if (keep) ring[(cur + __builtin_amdgcn_mbcnt_hi((uint32_t)(kb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kb, 0u))) & 127u] = at.read0 + idx; \
            cur += cw;                                                                                                        \
            if (cur - flushed >= 64u) { /* uniform */                                                                         \
                const uint32_t v = ring[(flushed + lane) & 127u];                                                             \
                atomicOr(&mask[v >> 6], 1ull << (v & 63u));                                                                   \
                flushed += 64u;                                                                                               \
            }                                                                                                                 \
        }                                                                                                                     \
    }
    Slot S[kRankDepth];
    {
        // chunks 0..6: descriptors the plain way (nothing is in flight yet), their positions into the ring; descriptors
        // of chunks 7..14 into the ring; then everything is waited for once -- the loop's counted wait assumes its own
        // issue order (positions, descriptor, positions, ...), which these requests do not have
        uint32_t d0[kRankDepth - 1];
#pragma unroll
        for (int k = 0; k < kRankDepth - 1; ++k) d0[k] = desc[desc_offset((uint32_t)k) / 4u];
#pragma unroll
        for (int k = 0; k < kRankDepth - 1; ++k) S[k] = slot_of(d0[k], (uint32_t)k);
        S[kRankDepth - 1] = Slot{0u, 0u, 0u, false};
#define QMCP_PM_ASK_KEYS(K) asm volatile("global_load_ushort " QMCP_PM_KEYREG(K) ", %0, %1\n\tglobal_load_ushort " QMCP_PM_IDXREG(K) ", %0, %2" : : "v"((S[K].slot0 + lane) * 2u), "s"(keys16), "s"(idx16) : QMCP_PM_RING, "memory");
#define QMCP_PM_ASK_DSC(K) asm volatile("global_load_dword " QMCP_PM_DSCREG(K) ", %0, %1" : : "v"(desc_offset(7u + (uint32_t)(K))), "s"(desc) : QMCP_PM_RING, "memory");
        QMCP_PM_ASK_KEYS(0) QMCP_PM_ASK_KEYS(1) QMCP_PM_ASK_KEYS(2) QMCP_PM_ASK_KEYS(3) QMCP_PM_ASK_KEYS(4) QMCP_PM_ASK_KEYS(5) QMCP_PM_ASK_KEYS(6)
        QMCP_PM_ASK_DSC(0) QMCP_PM_ASK_DSC(1) QMCP_PM_ASK_DSC(2) QMCP_PM_ASK_DSC(3) QMCP_PM_ASK_DSC(4) QMCP_PM_ASK_DSC(5) QMCP_PM_ASK_DSC(6) QMCP_PM_ASK_DSC(7)
        asm volatile("s_waitcnt vmcnt(0)" : : : QMCP_PM_RING, "memory");
#undef QMCP_PM_ASK_KEYS
#undef QMCP_PM_ASK_DSC
    }
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += kRankDepth) {
        QMCP_PM_STEP(0, 7) QMCP_PM_STEP(1, 0) QMCP_PM_STEP(2, 1) QMCP_PM_STEP(3, 2)
        QMCP_PM_STEP(4, 3) QMCP_PM_STEP(5, 4) QMCP_PM_STEP(6, 5) QMCP_PM_STEP(7, 6)
    }
#undef QMCP_PM_STEP
    asm volatile("s_waitcnt vmcnt(0)" : : : QMCP_PM_RING, "memory");  // (the loads asked for beyond the last chunk)
    if (lane < cur - flushed) {
        const uint32_t v = ring[(flushed + lane) & 127u];
        atomicOr(&mask[v >> 6], 1ull << (v & 63u));
    }
    kept = cur;
    if (lane == 0 && kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    if (tid == 0) {
        amb_count[range] = s_namb;
        if (s_kept) atomicAdd(kept_total, (unsigned long long)s_kept);
    }
}

// The listed (chunk, position) groups of every range: the position's `skip` LAST records of that chunk are the ones the
// quota did not reach, so a wave walks the chunk's sixteen wave-slots backwards, passes over that many matches and
// keeps the rest.  grid (ranges, kPmSettleY), sixteen waves per
// workgroup: a range's groups (a few dozen at cfg4: 20 000 in all) are dealt to 16 kPmSettleY waves, and a wave takes two
// groups at a time (all their loads asked for before either is looked at: a group is four dependent trips to memory and
// little else).  Few, large workgroups: every workgroup ends in one atomic on the kept count, and same-address atomics
// queue up behind one another (a first form with 32 small workgroups per range: 5 000 of them, 0.07 ms for 0.006 of work).
static constexpr uint32_t kPmSettleY = 2;
static constexpr uint32_t kPmSettleWaves = 16;
__global__ __launch_bounds__(1024) void k_pm_settle(const uint16_t* __restrict__ keys16, const uint16_t* __restrict__ idx16,
                                                   const uint32_t* __restrict__ desc, const uint32_t* __restrict__ Tp,
                                                   uint32_t pitch, const uint32_t* __restrict__ range_start, uint32_t shift,
                                                   uint32_t stride, const uint2* __restrict__ amb_lists, int lists_by_records,
                                                   const uint32_t* __restrict__ amb_count,
                                                   unsigned long long* __restrict__ mask,
                                                   unsigned long long* __restrict__ kept_total) {
    __shared__ uint32_t s_kept;
    const uint32_t range = blockIdx.x, width = 1u << shift;
    const uint32_t namb = amb_count[range];
    if (blockIdx.y * kPmSettleWaves >= namb) return;  // uniform
    if (threadIdx.x == 0) s_kept = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lo_p = Tp[(size_t)range * pitch], hi_p = Tp[(size_t)(range + 1) * pitch];
    const uint32_t g0 = lo_p >> 6, n_ws = (hi_p - lo_p) >> 6;
    const uint2* const amb = amb_lists + (lists_by_records ? (size_t)range_start[range] : (size_t)range * width);
    const uint64_t gt_mask = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);  // lanes above this one
    uint32_t kept = 0;
    constexpr int kSteps = 16, kE = 2;
    const uint32_t stride_k = kPmSettleWaves * gridDim.y;
    for (uint32_t k0 = blockIdx.y * kPmSettleWaves + w; k0 < namb; k0 += kE * stride_k) {
        uint32_t key[kE][kSteps], dscv[kE] /* lane t < 16: the descriptor of the chunk's wave-slot t */, p[kE], skip[kE], c_of[kE];
#pragma unroll
        for (int x = 0; x < kE; ++x) {
            const uint32_t k = k0 + (uint32_t)x * stride_k;
            const uint2 ent = k < namb ? amb[k] : make_uint2(0u, 0u);
            c_of[x] = ent.x >> 15;
            p[x] = k < namb ? ent.x & 0x7FFFu : 0xFFFFFFFFu;  // (no 16-bit position equals it)
            skip[x] = ent.y;  // matches still to be passed over, from the chunk's end
            const uint32_t ws = 16u * c_of[x] + (lane & 15u);
            dscv[x] = ws < n_ws ? desc[g0 + ws] : 0u;
        }
#pragma unroll
        for (int x = 0; x < kE; ++x) {
#pragma unroll
            for (int t = 0; t < kSteps; ++t) {
                const uint32_t dsc = (uint32_t)__builtin_amdgcn_readlane((int)dscv[x], t);
                key[x][t] = keys16[pm_unpack<false>(dsc, 16u * c_of[x] + (uint32_t)t < n_ws, stride).slot0 + lane];
            }
        }
#pragma unroll
        for (int x = 0; x < kE; ++x) {
#pragma unroll
            for (int t = kSteps - 1; t >= 0; --t) {
                const uint32_t dsc = (uint32_t)__builtin_amdgcn_readlane((int)dscv[x], t);
                const PmSlot at = pm_unpack<false>(dsc, 16u * c_of[x] + (uint32_t)t < n_ws, stride);
                const bool member = lane < at.nv && key[x][t] == p[x];
                const uint64_t m = __ballot(member);
                if (m == 0) continue;
                const uint32_t above = (uint32_t)__popcll(m & gt_mask);  // matches after this one in the step
                if (member && above >= skip[x]) {
                    const uint32_t v = at.pass * (uint32_t)kPmPass + idx16[at.slot0 + lane];
                    atomicOr(&mask[v >> 6], 1ull << (v & 63u));
                }
                const uint32_t in_step = (uint32_t)__popcll(m);
                kept += in_step > skip[x] ? in_step - skip[x] : 0u;
                skip[x] = skip[x] > in_step ? skip[x] - in_step : 0u;
            }
        }
    }
    if (lane == 0 && kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    if (threadIdx.x == 0 && s_kept) atomicAdd(kept_total, (unsigned long long)s_kept);
}

// ---- launchers
uint32_t pm_pitch(uint32_t n) { return part_pass_pitch(n); }  // passes of the call, rounded up to a multiple of 4
uint32_t pm_pass() { return (uint32_t)kPmPass; }
uint32_t pm_stride(uint32_t ltot, uint32_t shift) { return pm_stride_of((ltot >> shift) + 1u); }
size_t pm_slots(uint32_t n, uint32_t ltot, uint32_t shift) { return (size_t)pm_pitch(n) * pm_stride(ltot, shift); }  // slots of keys16 / idx16
uint32_t pm_work_words() { return kPmWorkWords; }
uint32_t pm_exc_slots(uint32_t n) { return pm_pitch(n) * kPmWaves * kPmExcPerWave + kNuOverflow; }  // the exception list's slots: 128 per wave and pass, and the overflow region
void launch_pm_prepare_sort(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs, uint32_t shift, uint32_t ltot,
                            uint16_t* keys16, uint16_t* idx16, uint32_t* cnt_tab, uint32_t* lst_tab,
                            uint32_t* work, uint32_t* stats, unsigned long long* zero_mask, uint32_t ell_reg, uint32_t* exc,
                            uint32_t exc_cap, uint32_t* exc_cnt) {
    const uint32_t pitch = pm_pitch(n);
    if (pitch == 0) return;
    (void)hipFuncSetAttribute((const void*)k_pm_prepare_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPmSortLds);
    hipLaunchKernelGGL(k_pm_prepare_sort, dim3(pitch / kPmPassesPerWg), dim3(kPmThreads), kPmSortLds, st, starts, ends, n,
                       d_roff, d_poff, n_contigs, shift, pm_stride(ltot, shift), keys16, idx16, cnt_tab, lst_tab, pitch,
                       work, stats, zero_mask, exc != nullptr ? ell_reg : 0u, exc, exc_cap, exc_cnt);
    if (exc != nullptr && ell_reg != 0u)
        hipLaunchKernelGGL(k_nu_count_groups, dim3(32), dim3(256), 0, st, exc_cnt, pitch * kPmWaves, stats);
}
void launch_pm_descr(hipStream_t st, const uint32_t* Tp, const uint32_t* lstw, uint32_t n, uint32_t ltot, uint32_t shift,
                     uint32_t* desc, uint32_t* work, uint32_t* range_start, uint32_t* max_load) {
    const uint32_t pitch = pm_pitch(n), n_ranges = (ltot >> shift) + 1u;
    const uint32_t s64 = pm_stride(ltot, shift) / 64u;
    hipLaunchKernelGGL(k_pm_descr, dim3(n_ranges, kPmDescrY), dim3(256), 0, st, Tp, lstw, pitch, pitch * s64, desc, work);
    hipLaunchKernelGGL(k_pm_range_table, dim3(1), dim3(256), 0, st, work, range_start, max_load);
}
void launch_pm_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* desc, const uint32_t* Tp, uint32_t n,
                       const uint32_t* range_start, uint32_t shift, uint32_t ltot, uint32_t* boff, uint32_t* empty_positions) {
    const uint32_t n_ranges = (ltot >> shift) + 1;  // covers positions 0..ltot
    const size_t width = (size_t)1 << shift;
    const size_t lds = (width + width / 32 + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_pm_offsets, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_pm_offsets, dim3(n_ranges), dim3(1024), lds, st, keys16, desc, Tp, pm_pitch(n), range_start, shift,
                       pm_stride(ltot, shift), ltot, boff, empty_positions);
}
// The ranking: walk, then the settling of the groups it listed -- two launches, in this order.
void launch_pm_walk(hipStream_t st, const uint16_t* keys16, const uint16_t* idx16, const uint32_t* desc, const uint32_t* Tp,
                    uint32_t n, const uint32_t* range_start, uint32_t shift, uint32_t ltot, const uint32_t* boff,
                    const uint32_t* selend, unsigned long long* mask, unsigned long long* kept_total, void* scratch,
                    bool scratch_by_records, uint32_t* amb_count, const uint32_t* ev_sev, const uint32_t* ev_lastns,
                    const uint64_t* d_poff, uint32_t n_contigs, uint32_t ell) {
    const uint32_t n_ranges = (ltot >> shift) + 1;
    const EvQuota evq{ev_sev, ev_lastns, d_poff, n_contigs, ell};
    const size_t lds = (((size_t)1 << shift) + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_pm_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_pm_walk, dim3(n_ranges), dim3(1024), lds, st, keys16, idx16, desc, Tp, pm_pitch(n), range_start, shift,
                       pm_stride(ltot, shift), ltot, boff, selend, evq, mask, (uint2*)scratch, scratch_by_records ? 1 : 0,
                       amb_count, kept_total);
}
void launch_pm_settle(hipStream_t st, const uint16_t* keys16, const uint16_t* idx16, const uint32_t* desc, const uint32_t* Tp,
                      uint32_t n, const uint32_t* range_start, uint32_t shift, uint32_t ltot, const void* scratch,
                      bool scratch_by_records, const uint32_t* amb_count, unsigned long long* mask,
                      unsigned long long* kept_total) {
    const uint32_t n_ranges = (ltot >> shift) + 1;
    hipLaunchKernelGGL(k_pm_settle, dim3(n_ranges, kPmSettleY), dim3(64 * kPmSettleWaves), 0, st, keys16, idx16, desc, Tp, pm_pitch(n),
                       range_start, shift, pm_stride(ltot, shift), (const uint2*)scratch, scratch_by_records ? 1 : 0, amb_count,
                       mask, kept_total);
}
