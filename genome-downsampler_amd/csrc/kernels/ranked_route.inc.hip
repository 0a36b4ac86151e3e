// ranked_route.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ range-ranked uniform path
// Uniform-span selection needs two things per start position p: the number of reads starting
// there (for the sweep) and, afterwards, the S(p) lowest read indices of the bucket.  Neither
// needs a full sort.  ONE stable partition of {global start, read index} records by position
// range (digit = gstart >> shift, <= 256 ranges of <= 32 Ki positions, the first pass of the
// record radix with a different shift) groups the reads of a range together IN INDEX ORDER;
// per range an LDS array then gives
//   k_range_offsets: reads per position (LDS histogram) scanned into bucket offsets, and
//   k_rank_mark   : the kept reads, by walking the range's records in order against a
//                   per-position quota that starts at S(p).
// 8 + 8 B/read for the partition instead of three radix passes (3 x 24 B/read) plus k_mark.
static constexpr uint32_t kMaxRangeShift = 15;  // 32 Ki positions x 4 B = 128 KiB of LDS

// The range partition itself: the stable scatter of k_radix_scatter_rec<true, false>, but a
// workgroup stages kPartTiles consecutive 4096-read tiles before writing, so a range leaves as
// one run of kPartTiles x ~16 records instead of separate unaligned 128-byte runs (those made the
// scatter move 1.7 x its bytes).  Two tiles (8 waves, 74 KiB of LDS) keep two workgroups on a CU,
// whose phases overlap; four tiles make longer runs but leave the CU's memory pipes idle while
// its single workgroup ranks.  The per-4096-tile histogram of k_prepare and its scan are used as
// they are: for a fixed range the tiles' runs are adjacent, so a pass starts at its first tile's
// offset.
#ifndef QMCP_PART_TILES
#define QMCP_PART_TILES 2  // measured on cfg4: 1 tile 0.64 ms, 2 tiles 0.34 ms, 4 tiles 0.46 ms
#endif
static constexpr int kPartTiles = QMCP_PART_TILES;             // 4096-read tiles per workgroup pass
static_assert(kPartTiles == 2, "k_prepare writes its partition table per pair of tiles");
static constexpr int kPartRecs = kPartTiles * kSortTile;      // 16384
static constexpr int kPartThreads = 256 * kPartTiles;
static constexpr int kPartWaves = kPartThreads / 64;
static constexpr size_t kPartLds = (size_t)kPartRecs * sizeof(Rec) + kPartWaves * 256 * sizeof(uint32_t) +
                                   256 * sizeof(uint32_t) + 64;

// MODE 0: keys = global start positions (k_prepare wrote them); MODE 1: keys = contig-relative
// starts, the global start is built here from the contig's offset (so k_prepare need not write it:
// 4 B/read less traffic); MODE 2: second level of a two-level partition -- the input are the
// {global start, index} records of the first level, already grouped into <= 256 super-ranges, and
// every super-range is partitioned on its own (tiles aligned to its first record; the offset table
// is laid out [super-range][range digit][tile of the super-range], so one plain exclusive scan
// over it yields absolute destinations).
// OUT_REC: emit {global start, index} records (first level) instead of the two final streams.
struct SegTables {                   // device tables of the two-level route (257 entries each)
    const uint32_t* super_start;     // first record of every super-range in first-level order
    const uint32_t* tile_base;       // 4096-record tiles of all lower super-ranges
    const uint32_t* pass_base;       // partition passes (kPartTiles tiles) of all lower super-ranges
};

// In-wave stable ranking of one pass's records by digit: BITS ballots per round of 64 records tell the
// lanes of one digit apart (the digits of a pass differ in their low BITS bits only, see the kernel).  A
// lane's peers are a 64-bit mask per lane, kept as two words: per bit one sign-extended bit field (all ones
// or zero), one compare (the ballot) and an XNOR + AND per word; ranks inside the group and the group's
// first lane come from the mask-count instructions.  BITS is a template argument so that no bit costs a
// scalar branch (the run-time loop over eight bits took 72 instructions per record; this takes 12 + 6 BITS,
// and the partition is bound by instruction issue, not by its bytes).
template <int BITS>
__device__ __forceinline__ void part_rank_rounds(const Rec (&rec)[kSortItems], uint32_t (&rank)[kSortItems],
                                                 uint32_t wbase, uint32_t bound, uint32_t shift, int lane,
                                                 uint32_t* __restrict__ s_cnt_w /* the wave's 256 counters */,
                                                 uint32_t skip = 0 /* bit k: leave the lane's k-th record out */) {
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < bound && !((skip >> k) & 1u);
        const uint32_t d = (rec[k].key >> shift) & 255u;
        const uint64_t v = __ballot(valid);
        uint32_t plo = (uint32_t)v, phi = (uint32_t)(v >> 32);
        if (!valid) { plo = ~plo; phi = ~phi; }
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const uint32_t fill = (uint32_t)((int32_t)(d << (31 - b)) >> 31);  // bit b of d, in every bit
            const uint64_t m = __ballot(fill != 0u);
            plo &= ~(fill ^ (uint32_t)m);
            phi &= ~(fill ^ (uint32_t)(m >> 32));
        }
        const uint32_t in_group = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));  // peers below this lane
        // every lane of the group reads the wave's running count of its digit (a broadcast read), then
        // the group's first lane moves it on: a wave's LDS operations execute in order, so the read is
        // the value before this round and the next round's read sees the write
        const uint32_t old = s_cnt_w[d];
        if (valid && in_group == 0u) s_cnt_w[d] = old + (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
        rank[k] = old + in_group;
    }
}

template <int MODE, bool OUT_REC>
__global__ __launch_bounds__(kPartThreads) void k_range_partition(
    const uint32_t* __restrict__ keys, const Rec* __restrict__ recs_in, SegTables seg,
    const uint64_t* __restrict__ contig_read_off, const uint64_t* __restrict__ contig_pos_off,
    uint32_t n_contigs, uint32_t n, uint32_t shift, uint32_t n_tiles /* row pitch of offs (MODE 0/1: in passes) */,
    const uint32_t* __restrict__ offs,
    uint16_t* __restrict__ out_key, uint32_t* __restrict__ out_idx, Rec* __restrict__ out_rec,
    uint32_t* __restrict__ range_start, uint32_t* __restrict__ max_load,
    // near-uniform route (MODE 1): reads whose span is not ell_reg are left out, as k_prepare left them out of the
    // table (which then holds the listed records' total behind its last row: the scan's total)
    const uint32_t* __restrict__ ends, uint32_t ell_reg) {
    extern __shared__ uint32_t s_part[];
    Rec* s_rec = reinterpret_cast<Rec*>(s_part);                       // [kPartRecs]
    uint32_t* s_cnt = s_part + 2 * kPartRecs;                          // [kPartWaves][256]
    uint32_t* s_gbase = s_cnt + kPartWaves * 256;                      // [256]
    uint32_t* s_wave = s_gbase + 256;                                  // [4] (+ pad to 16)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // this pass: records [base, base + count), none at or beyond `bound`; its row of the offset
    // table: entry of digit d = offs[off0 + d * off_stride]
    uint32_t base, bound, off0, off_stride;
    if (MODE == 2) {
        uint32_t lo = 0, hi = 256;  // last super-range whose first pass is <= blockIdx.x
        if (blockIdx.x >= seg.pass_base[256]) return;  // the grid is an upper bound
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (seg.pass_base[mid] <= blockIdx.x) lo = mid; else hi = mid;
        }
        const uint32_t pass = blockIdx.x - seg.pass_base[lo];
        const uint32_t t_h = seg.tile_base[lo + 1] - seg.tile_base[lo];
        base = seg.super_start[lo] + pass * kPartRecs;
        bound = seg.super_start[lo + 1];
        off0 = seg.tile_base[lo] * 256u + pass * kPartTiles;
        off_stride = t_h;
    } else {
        base = blockIdx.x * kPartRecs;
        bound = n;
        off0 = blockIdx.x;  // k_prepare's table has one entry per pass
        off_stride = n_tiles;
    }
    const uint32_t count = min((uint32_t)kPartRecs, bound - base);
    // where this pass's run of every range begins: a lone word per range from the offset table, asked
    // for now so that its trip to memory is over when the counts are in
    const uint32_t run_begin = threadIdx.x < 256 ? offs[off0 + threadIdx.x * off_stride] : 0u;
    // (loads first: they travel while the counters are cleared)
    // wave w owns records [w * 1024, (w + 1) * 1024) of the pass, in 16 rounds of 64: order inside a
    // range = (wave, round, lane) = input order
    const uint32_t wbase = base + w * (kSortItems * 64);
    Rec rec[kSortItems];
    uint32_t rank[kSortItems];
    uint32_t skip = 0;  // bit k: the thread's k-th read is an exception (left out)
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (MODE == 2) {
            rec[k] = i < bound ? recs_in[i] : Rec{0u, 0u};
        } else {
            rec[k].key = i < bound ? keys[i] : 0u;
            rec[k].val = i;
            if (MODE == 1 && ell_reg != 0u && i < bound && ends[i] - rec[k].key + 1u != ell_reg) skip |= 1u << k;
        }
    }
    for (int i = threadIdx.x; i < kPartWaves * 256; i += kPartThreads) s_cnt[i] = 0;
    if (MODE != 2 && blockIdx.x == 0) {
        // the first workgroup also publishes where every range's records begin (257 entries) and the
        // heaviest range's load, for the per-range kernels and the host's balance test
        uint32_t load = 0;
        if (threadIdx.x < 256) {
            const uint32_t d = threadIdx.x;
            // (behind the last row: the scan's total -- the records listed, fewer than n on the near-uniform route)
            const uint32_t total = ell_reg != 0u ? offs[256u * n_tiles] : n;
            const uint32_t r_lo = offs[d * n_tiles];
            const uint32_t r_hi = d + 1 < 256 ? offs[(d + 1) * n_tiles] : total;
            range_start[d] = r_lo;
            if (d == 255) range_start[256] = total;
            load = r_hi - r_lo;
        }
        load = wave_max_u32(load);
        if (lane == 0) s_gbase[w] = load;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = 0;
            for (int x = 0; x < kPartWaves; ++x) m = max(m, s_gbase[x]);
            max_load[0] = m;
        }
    }
    __syncthreads();
    uint32_t match_bits = 8;  // digit bits that can differ inside this pass
    if (MODE == 1) {
        // `keys` holds contig-relative starts: add the contig's position offset.  Almost every pass
        // lies inside one contig (reads are grouped by contig); otherwise search per read.
        auto contig_of = [&](uint32_t i) {
            uint32_t lo = 0, hi = n_contigs;  // last c with roff[c] <= i
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (contig_read_off[mid] <= i) lo = mid; else hi = mid;
            }
            return lo;
        };
        const uint32_t c_first = contig_of(base), c_last = contig_of(base + count - 1);
        // the pass's digits lie in the range the contigs it touches span: their low bit_width(hi - lo)
        // bits tell them apart, so the in-wave match below needs that many ballots, not eight
        {
            const uint32_t d_lo = ((uint32_t)contig_pos_off[c_first] >> shift) & 255u;
            // (up to and including the first position behind the last contig: where a read of a zero-length
            //  contig is counted, see below)
            const uint32_t d_hi = ((uint32_t)contig_pos_off[c_last + 1] >> shift) & 255u;
            match_bits = d_hi >= d_lo ? 32u - (uint32_t)__builtin_clz((d_hi - d_lo) | 1u) : 8u;
            if (d_hi == d_lo) match_bits = 0;
        }
        // A start at or beyond its contig's length (an invalid read: the call will fail, but this pass is
        // queued before the host knows) is taken as the contig's last position, here and in k_prepare's
        // histogram alike: its digit then lies in the pass's digit interval, which the BITS-bit match and the
        // scanned table's run lengths both rely on.
        if (c_first == c_last) {
            const uint32_t p0 = (uint32_t)contig_pos_off[c_first];
            const uint32_t len = (uint32_t)contig_pos_off[c_first + 1] - p0;
            const uint32_t last = len ? len - 1u : 0u;
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) rec[k].key = p0 + min(rec[k].key, last);
        } else {
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) {
                const uint32_t i = wbase + k * 64 + lane;
                if (i < bound) {
                    const uint32_t cc = contig_of(i);
                    const uint32_t p0 = (uint32_t)contig_pos_off[cc];
                    const uint32_t len = (uint32_t)contig_pos_off[cc + 1] - p0;
                    rec[k].key = p0 + min(rec[k].key, len ? len - 1u : 0u);
                }
            }
        }
    }
    {
        uint32_t* const s_cnt_w = s_cnt + w * 256;
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
        // range d = threadIdx.x: where each wave's records of the range go inside the pass, and the
        // global base of the range's run
        const uint32_t d = threadIdx.x;
        uint32_t c[kPartWaves], tot = 0;
#pragma unroll
        for (int x = 0; x < kPartWaves; ++x) { c[x] = s_cnt[x * 256 + d]; tot += c[x]; }
        // exclusive scan of the 256 range totals over four waves
        const uint32_t inc = wave_incl_scan_add(tot);
        if (lane == 63) s_wave[w] = inc;
        s_gbase[d] = inc - tot;  // exclusive inside the wave; completed after the barrier
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const uint32_t d = threadIdx.x;
        uint32_t wave_base = 0;
        for (int x = 0; x < w; ++x) wave_base += s_wave[x];
        const uint32_t tile_off = s_gbase[d] + wave_base;
        uint32_t run = tile_off;
#pragma unroll
        for (int x = 0; x < kPartWaves; ++x) { const uint32_t cx = s_cnt[x * 256 + d]; s_cnt[x * 256 + d] = run; run += cx; }
        s_gbase[d] = run_begin - tile_off;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < bound && !((skip >> k) & 1u)) {
            const uint32_t d = (rec[k].key >> shift) & 255u;
            s_rec[s_cnt[w * 256 + d] + rank[k]] = rec[k];
        }
    }
    // records staged: the pass's, less the exceptions (the four waves' digit totals, from the scan above)
    const uint32_t staged = (MODE == 1 && ell_reg != 0u) ? s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3] : count;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t j = k * kPartThreads + threadIdx.x;
        if (j < staged) {
            const Rec r = s_rec[j];
            const uint32_t dst = s_gbase[(r.key >> shift) & 255u] + j;
            if (OUT_REC) {
                out_rec[dst] = r;
            } else {
                // two streams: the position inside the range (< 2^15: 16 bits) and the read index; the
                // per-range kernels stream 2 + 0 and 2 + 4 bytes per read instead of 8 and 8
                out_key[dst] = (uint16_t)(r.key & ((1u << shift) - 1u));
                out_idx[dst] = r.val;
            }
        }
    }
}

// ---- two-level route (more than 256 ranges: genomes beyond 8.39 M positions) ----
// tile and pass tables of the super-ranges, from where the first level put them
__global__ __launch_bounds__(256) void k_seg_tables(const uint32_t* __restrict__ super_start,
                                                    uint32_t* __restrict__ tile_base,
                                                    uint32_t* __restrict__ pass_base,
                                                    uint32_t* __restrict__ max_load) {
    __shared__ uint32_t s_wave[4];
    const uint32_t h = threadIdx.x;
    const uint32_t n_h = super_start[h + 1] - super_start[h];
    const uint32_t t_h = (n_h + kSortTile - 1) / kSortTile;
    const uint32_t p_h = (t_h + kPartTiles - 1) / kPartTiles;
    uint32_t tot;
    const uint32_t tb = block_excl_scan_256(t_h, s_wave, tot);
    tile_base[h] = tb;
    if (h == 255) tile_base[256] = tot;
    const uint32_t pb = block_excl_scan_256(p_h, s_wave, tot);
    pass_base[h] = pb;
    if (h == 255) pass_base[256] = tot;
    if (h == 0) max_load[0] = 0;
}

// per-tile histogram of the second-level digit, tiles aligned to the super-ranges
__global__ __launch_bounds__(kSortThreads) void k_seg_hist(const Rec* __restrict__ recs, SegTables seg,
                                                           uint32_t shift, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    if (blockIdx.x >= seg.tile_base[256]) return;  // the grid is an upper bound
    uint32_t lo = 0, hi = 256;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg.tile_base[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const uint32_t t = blockIdx.x - seg.tile_base[lo];
    const uint32_t t_h = seg.tile_base[lo + 1] - seg.tile_base[lo];
    const uint32_t base = seg.super_start[lo] + t * kSortTile, bound = seg.super_start[lo + 1];
    s_h[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = base + k * kSortThreads + threadIdx.x;
        if (i < bound) atomicAdd(&s_h[(recs[i].key >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[seg.tile_base[lo] * 256u + threadIdx.x * t_h + t] = s_h[threadIdx.x];
}

// where every final range begins (65 536 + 1 entries), and the heaviest range's load
__global__ __launch_bounds__(256) void k_seg_range_table(const uint32_t* __restrict__ scanned, SegTables seg,
                                                         uint32_t n, uint32_t* __restrict__ range_start,
                                                         uint32_t* __restrict__ max_load) {
    __shared__ uint32_t s_red[4];
    const uint32_t h = blockIdx.x, d = threadIdx.x;
    const uint32_t t_h = seg.tile_base[h + 1] - seg.tile_base[h];
    const uint32_t row = seg.tile_base[h] * 256u;
    const uint32_t lo = t_h ? scanned[row + d * t_h] : seg.super_start[h];
    const uint32_t hi = d + 1 < 256 ? (t_h ? scanned[row + (d + 1) * t_h] : seg.super_start[h])
                                    : seg.super_start[h + 1];
    range_start[h * 256u + d] = lo;
    if (h == 255 && d == 255) range_start[65536] = seg.super_start[256];  // (the records listed: n, or fewer on the near-uniform route)
    (void)n;
    const uint32_t m = wave_max_u32(hi - lo);
    if ((d & 63) == 0) s_red[d >> 6] = m;
    __syncthreads();
    if (d == 0) atomicMax(max_load, max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
}

__global__ __launch_bounds__(1024) void k_range_offsets(const uint16_t* __restrict__ keys16,
                                                        const uint32_t* __restrict__ range_start,
                                                        uint32_t shift, uint32_t ltot,
                                                        uint32_t* __restrict__ boff,
                                                        uint32_t* __restrict__ empty_positions /* += positions of the
                                                            range that start no read (one add per workgroup); or null */) {
    // [1 << shift] counters, one pad word after every 32: a thread's 32 consecutive positions then
    // sit in 32 different banks during the scan
    extern __shared__ uint32_t s_cnt32[];
#define PADDED(i) ((i) + ((i) >> 5))
    __shared__ uint32_t s_wsum[16], s_esum[16];
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    const uint32_t lo = range_start[range], hi = range_start[range + 1];  // (asked for before the clearing)
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) s_cnt32[PADDED(i)] = 0;
    __syncthreads();
    // The range's 16-bit records are read four at a time (8-byte loads from the first 8-byte-aligned
    // record on), eight loads in flight per thread: 64 KiB in flight per workgroup -- what it takes
    // to keep a CU's share of HBM busy at ~2 us latency (2-byte loads, 16 KiB in flight: 2.9 TB/s).
    auto count = [&](uint32_t li) {
        if (li < width) atomicAdd(&s_cnt32[PADDED(li)], 1u);
    };
    const uint32_t head = min((4u - (lo & 3u)) & 3u, hi - lo);  // records before the aligned part
    if (threadIdx.x < head) count(keys16[lo + threadIdx.x]);
    const uint32_t a0 = lo + head;
    const uint32_t n_quads = (hi - a0) >> 2;
    const uint2* __restrict__ quads = reinterpret_cast<const uint2*>(keys16 + a0);
    constexpr int U = 8;
    uint32_t j = threadIdx.x;
    for (; j + (U - 1) * 1024u < n_quads; j += U * 1024u) {
        uint2 k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = quads[j + u * 1024u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            count(k[u].x & 0xFFFFu); count(k[u].x >> 16);
            count(k[u].y & 0xFFFFu); count(k[u].y >> 16);
        }
    }
    for (; j < n_quads; j += 1024u) {
        const uint2 k = quads[j];
        count(k.x & 0xFFFFu); count(k.x >> 16);
        count(k.y & 0xFFFFu); count(k.y >> 16);
    }
    const uint32_t tail0 = a0 + 4u * n_quads;  // at most three records behind the last whole quad
    if (tail0 + threadIdx.x < hi) count(keys16[tail0 + threadIdx.x]);
    __syncthreads();
    // counts -> bucket offsets, in place: exclusive scan over the range's positions, started at
    // the number of records in all lower ranges (= the offset of the range's first position), so
    // the array needs no separate scan pass over the whole genome
    const uint32_t per = width >= 1024u ? width >> 10 : 1u;  // positions per thread
    const uint32_t first = threadIdx.x * per;
    uint32_t sum = 0, empties = 0;
    if (first < width)
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t cq = s_cnt32[PADDED(first + q)];
            sum += cq;
            empties += (cq == 0 && pos0 + first + q < ltot) ? 1u : 0u;
        }
    const uint32_t inc = wave_incl_scan_add(sum);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 63) s_wsum[w] = inc;
    if (empty_positions != nullptr) {
        // how spiky the starts are: the host picks the sweep kernel by it (the event-driven form pays
        // dearly for blocks with an empty start position).  One global add per workgroup: a long genome has
        // tens of thousands of ranges, and sixteen same-address atomics from each queue up behind one another.
        empties = wave_sum_u32(empties);
        if (lane == 0) s_esum[w] = empties;
    }
    __syncthreads();
    if (empty_positions != nullptr && threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t x = 0; x < (blockDim.x >> 6); ++x) total += s_esum[x];
        if (total != 0) atomicAdd(empty_positions, total);
    }
    uint32_t run = lo + inc - sum;
    for (uint32_t x = 0; x < w; ++x) run += s_wsum[x];
    if (first < width)
        for (uint32_t q = 0; q < per; ++q) {
            const uint32_t cq = s_cnt32[PADDED(first + q)];
            s_cnt32[PADDED(first + q)] = run;
            run += cq;
        }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x)
        if (pos0 + i <= ltot) boff[pos0 + i] = s_cnt32[PADDED(i)];
#undef PADDED
}

// One workgroup (16 waves) per range.  The quota array starts at q[p] = S(p) = selend - boff.
// The range's records are walked in order (they are in read-index order), a chunk of
// kRankU x blockDim.x records at a time: every thread draws old = q[p]-- for its records, a
// barrier, then reads q_after = q[p], a barrier.  Chunks are ordered by the barriers, so a read is
// kept iff old > 0 -- except that threads of ONE chunk that hit the same position draw their
// `old` values in an unspecified order, which matters only where the quota runs out inside the
// chunk: exactly the threads with old > 0 that see q_after < 0 afterwards.  Those are not decided
// on the spot.  The one of them that drew old == 1 appends (chunk, position, q_after) to a list,
// and after the walk each wave takes list entries and settles them alone: the position's
// -q_after LAST records of that chunk are the ones the quota did not reach, so the wave walks the
// chunk's records backwards, skips that many matches and keeps the rest.  Every position runs out
// at most once, so the list needs at most one entry per position.  The kept set is exactly the
// S(p) lowest indices of every bucket, independent of LDS arbitration order.
#ifndef QMCP_RANK_U
#define QMCP_RANK_U 1
#endif
static constexpr int kRankU = QMCP_RANK_U;  // records per thread and chunk
#ifndef QMCP_RANK_DEPTH
#define QMCP_RANK_DEPTH 8
#endif
static constexpr int kRankDepth = QMCP_RANK_DEPTH;  // register sets of records in flight per thread

__global__ __launch_bounds__(1024) void k_rank_mark(const uint16_t* __restrict__ keys16,
                                                    const uint32_t* __restrict__ idx,
                                                    const uint32_t* __restrict__ range_start,
                                                    uint32_t shift, uint32_t ltot,
                                                    const uint32_t* __restrict__ boff,
                                                    const uint32_t* __restrict__ selend,
                                                    unsigned long long* __restrict__ mask,
                                                    unsigned long long* __restrict__ kept_total,
                                                    uint2* __restrict__ amb_lists, int lists_by_records) {
    extern __shared__ int32_t s_q[];  // [(1 << shift) + 1]; the last entry absorbs idle threads
    __shared__ uint32_t s_namb;
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    const uint32_t live = pos0 < ltot ? min(width, ltot - pos0) : 0u;
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, nw = nthreads >> 6;
    const uint32_t lane = tid & 63u, w = tid >> 6;
    // list slots: one per position of the range, or (when the call has fewer reads than positions)
    // one per record of the range -- a listed position has at least one record
    const uint32_t lo = range_start[range], hi = range_start[range + 1];
    if (lo >= hi) return;  // uniform: a range without reads needs no quotas either
    uint2* const amb = amb_lists + (lists_by_records ? (size_t)lo : (size_t)range * width);
    // quotas: eight positions' two loads in flight per thread (a range of 32 Ki positions is 32 trips to
    // memory per thread if taken one by one -- on sparse data, thousands of ranges of few reads, that
    // was most of this kernel's time)
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
    const uint32_t chunk_recs = kRankU * nthreads;
    const uint32_t n_chunks = (hi - lo + chunk_recs - 1) / chunk_recs;
    uint32_t kept = 0;

    // thread tid owns record tid of each of the chunk's kRankU sub-chunks
    struct Recs { uint32_t key[kRankU], val[kRankU]; };
    // The record loads are issued and waited for BY HAND (inline assembly, which the compiler's wait-count
    // pass does not look into).  Left to the compiler, the loop's other vector-memory operations -- the mask
    // atomics and the list stores, of another event type than loads -- make that pass give up counting: it
    // waited for all but the two youngest loads once per eight chunks (s_waitcnt vmcnt(2)), i.e. for records
    // asked for one chunk earlier, a full trip to memory on the chain of every eighth chunk (rocprofv3: 0.25 ms
    // at cfg4 where the bytes take 0.12).  A counted wait is safe whatever else is in flight: loads complete in
    // issue order, so while a record's load is outstanding so are the 2 (kRankDepth - 1) kRankU loads issued
    // after it, and "at most that many outstanding" cannot hold.  Younger atomics and stores only make the wait
    // longer.  Addresses are the stream's base (scalar) + a 32-bit byte offset: record j < 2^30.
    auto fetch = [&](Recs& dst, uint32_t c) {
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            const uint32_t j = min(lo + c * chunk_recs + u * nthreads + tid, hi - 1);
            asm volatile("global_load_ushort %0, %2, %3\n\tglobal_load_dword %1, %4, %5"
                         : "=&v"(dst.key[u]), "=&v"(dst.val[u])
                         : "v"(j * 2u), "s"(keys16), "v"(j * 4u), "s"(idx)
                         : "memory");
        }
    };
    auto chunk = [&](Recs& r, uint32_t c) {
        bool valid[kRankU];
        uint32_t li[kRankU];
        int32_t old[kRankU];
        // the chunk's records have landed once at most the loads of the kRankDepth - 1 chunks asked for
        // after it are outstanding (the operands tie the wait to the registers: nothing reads them before)
#pragma unroll
        for (int u = 0; u < kRankU; ++u)
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.key[u]), "+v"(r.val[u]) : "n"(2 * (kRankDepth - 1) * kRankU) : "memory");
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            valid[u] = lo + c * chunk_recs + u * nthreads + tid < hi;
            li[u] = valid[u] ? r.key[u] : width;  // (the 16-bit load zero-extends)
            old[u] = atomicSub(&s_q[li[u]], 1);
        }
        __syncthreads();
        bool keep[kRankU];
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            const int32_t aft = s_q[li[u]];
            // old > 0 and quota not exhausted by the end of the chunk: kept whatever the order was
            keep[u] = valid[u] && old[u] > 0 && aft >= 0;
            if (valid[u] && old[u] == 1 && aft < 0) {
                const uint32_t k = atomicAdd(&s_namb, 1u);
                amb[k] = make_uint2((c << 15) | li[u], (uint32_t)(-aft));  // c < 2^17, li < 2^15
            }
        }
        __syncthreads();  // every q_after is read before the next chunk draws
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            if (keep[u]) { const uint32_t v = r.val[u]; atomicOr(&mask[v >> 6], 1ull << (v & 63u)); }
            kept += (uint32_t)__popcll(__ballot(keep[u]));
        }
    };
    // Records are prefetched kRankDepth - 1 chunks ahead into kRankDepth register sets that rotate by NAME (the
    // loop is unrolled kRankDepth times, every index below is a constant): no register copies -- a copy of a
    // register whose load is still in flight would read what was there before.  A thread loads only 6 bytes
    // per chunk, so this depth is what keeps enough bytes in flight per CU (3 chunks ahead: 2.2 TB/s over the
    // chip).  Chunks past the end run with every thread idle (dummy quota slot).
    Recs R[kRankDepth];
#pragma unroll
    for (int k = 0; k < kRankDepth - 1; ++k) fetch(R[k], (uint32_t)k);
    for (uint32_t c = 0; c < n_chunks; c += kRankDepth) {
#pragma unroll
        for (int k = 0; k < kRankDepth; ++k) {
            fetch(R[(k + kRankDepth - 1) % kRankDepth], c + (uint32_t)k + (uint32_t)(kRankDepth - 1));
            chunk(R[k], c + (uint32_t)k);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the loads asked for beyond the last chunk)
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
        // all of the chunk's positions first (kSteps loads in flight), then the backward walk
        constexpr int kSteps = kRankU * 16;  // blockDim.x == 1024: 64-record steps per chunk
        uint32_t key[kSteps];
#pragma unroll
        for (int t = 0; t < kSteps; ++t) key[t] = keys16[min(first + t * 64u + lane, hi - 1)];
#pragma unroll
        for (int t = kSteps - 1; t >= 0; --t) {
            const uint32_t j = first + t * 64u + lane;
            const bool member = j < last && key[t] == p;
            const uint64_t m = __ballot(member);
            if (m == 0) continue;
            const uint32_t above = (uint32_t)__popcll(m & gt_mask);  // matches after this one in the step
            if (member && above >= skip) {
                const uint32_t v = idx[j];
                atomicOr(&mask[v >> 6], 1ull << (v & 63u));
            }
            const uint32_t in_step = (uint32_t)__popcll(m);
            kept += in_step > skip ? in_step - skip : 0u;
            skip = skip > in_step ? skip - in_step : 0u;
        }
    }
    // one global add per workgroup (thousands of ranges on a long genome would otherwise queue sixteen
    // same-address atomics each)
    __syncthreads();
    if (tid == 0) s_namb = 0;
    __syncthreads();
    if (lane == 0 && kept) atomicAdd(&s_namb, kept);
    __syncthreads();
    if (tid == 0 && s_namb) atomicAdd(kept_total, (unsigned long long)s_namb);
}
