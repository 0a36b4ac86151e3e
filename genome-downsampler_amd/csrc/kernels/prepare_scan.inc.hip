// prepare_scan.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ prepare
// One pass over the reads: validate (start <= end < contig length), reduce min/max span,
// write the global start position of every read (the bucketing key of the uniform path) and
// count reads per start position.  Reference counterpart: the read loop of
// create_b_function (quasi_mcp_cpu_max_flow_solver.cpp:61-67) -- here O(1) per read.
//
// stats[0] = min span, stats[1] = max span, stats[2] = error flag

// Near-uniform route: the exception list's OVERFLOW region -- its last kNuOverflow slots.  A wave that meets more
// exceptions than its own 128 slots hold (a coordinate-sorted file's reads around a breakpoint, say: nearly all of
// them clipped) appends the rest here, one atomic apiece on stats[6]; only if this region fills too is the list
// incomplete (stats[5]).
static constexpr uint32_t kNuOverflow = 1u << 20;
__device__ __forceinline__ void nu_overflow_put(uint32_t* __restrict__ exc, uint32_t exc_cap, uint32_t* __restrict__ stats,
                                                uint32_t gs, uint32_t ge, uint32_t i) {
    const uint32_t k = atomicAdd(&stats[6], 1u);
    if (k < kNuOverflow) {
        const size_t at = (size_t)exc_cap - kNuOverflow + k;
        exc[at] = gs;
        exc[exc_cap + at] = ge;
        exc[2 * (size_t)exc_cap + at] = i;
    } else {
        atomicOr(&stats[5], 1u);
    }
}

__global__ __launch_bounds__(256) void k_prepare(const uint32_t* __restrict__ starts,
                                                 const uint32_t* __restrict__ ends, uint32_t n,
                                                 const uint64_t* __restrict__ contig_read_off,
                                                 const uint64_t* __restrict__ contig_pos_off,
                                                 uint32_t n_contigs,
                                                 const uint64_t* __restrict__ keep_mask,
                                                 uint32_t* __restrict__ gstart_out,
                                                 uint32_t* __restrict__ cstart,
                                                 uint32_t* __restrict__ stats,
                                                 uint32_t n_tiles, uint32_t tiles_per_block,
                                                 uint32_t part_shift,
                                                 uint32_t* __restrict__ part_hist, uint32_t part_pitch,
                                                 uint32_t* __restrict__ digit0_hist,
                                                 uint32_t* __restrict__ global_digit_hist,
                                                 unsigned long long* __restrict__ zero_mask,
                                                 // near-uniform route (kernels/near_uniform.inc.hip) on the range-major
                                                 // form: reads whose span is not ell_reg are left out of the partition
                                                 // histogram (the partition skips them too) and listed -- every wave
                                                 // of every tile owns 128 slots of the list (tile t, wave w: from
                                                 // (4 t + w) * 128) and writes how many it filled; stats[5] is set if a
                                                 // wave met more.  ell_reg == 0: every read is regular.
                                                 uint32_t ell_reg, uint32_t* __restrict__ exc, uint32_t exc_cap,
                                                 uint32_t* __restrict__ exc_cnt) {
    __shared__ uint32_t s_gh[4][256];  // whole-call digit histograms of the start key (all 4 bytes)
    if (global_digit_hist) {
        for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&s_gh[0][0])[i] = 0;
    }
    __shared__ uint64_t s_roff[65];
    __shared__ uint64_t s_poff[65];
    // partition histograms of the workgroup's consecutive tiles, one per partition PASS (a pair of
    // tiles: the partition kernel stages two and only needs the pair's counts), written together at the
    // end: the table is laid out [digit][pass] (row pitch part_pitch, a multiple of 4), so a thread's
    // four pass counts are one aligned 16-byte run of its digit's row instead of lone words in rows'
    // worth of evicted lines (that cost 8x the table's size in HBM writes)
    __shared__ uint32_t s_ph[4][256];
    __shared__ uint32_t s_h0[256];
    __shared__ uint32_t s_fill[4];  // near-uniform route, tiles off the lean path: exceptions a wave has listed
    if (part_hist)
        for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&s_ph[0][0])[i] = 0;
    const uint32_t nc = min(n_contigs, 64u);
    for (uint32_t i = threadIdx.x; i <= nc; i += blockDim.x) {
        s_roff[i] = contig_read_off[i];
        s_poff[i] = contig_pos_off[i];
    }
    __syncthreads();
    uint32_t mn = 0xFFFFFFFFu, mx = 0, bad = 0;
    // Work is laid out in the radix tiles (4096 reads) so that per-tile histograms fall out of
    // the same pass: the range partition's (digit = global start >> part_shift) and, optionally,
    // the first LSD radix pass's (digit = low byte of the global start).
    const uint32_t t0 = blockIdx.x * tiles_per_block;
    for (uint32_t g = 0; g < tiles_per_block && t0 + g < n_tiles; ++g) {
        const uint32_t tile = t0 + g;
        uint32_t* s_h = s_ph[g >> 1];  // the pass's own row (the launcher keeps tiles_per_block even): no barrier between tiles on its account
        if (part_hist && digit0_hist) {
            s_h0[threadIdx.x] = 0;
            __syncthreads();
        }
        // the tile's 64 words of the output keep mask are cleared here (saves a memset launch)
        if (zero_mask && threadIdx.x < 64 && tile * 64u + threadIdx.x < (n + 63u) / 64u)
            zero_mask[tile * 64u + threadIdx.x] = 0ull;
        // all of the tile's loads first (32 in flight per thread), then the arithmetic
        const uint32_t tbase = tile * 4096u;
        const uint32_t tcount = min(4096u, n - tbase);
        uint32_t sv[16], ev[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t j = k * 256u + threadIdx.x;
            const uint32_t i = tbase + min(j, tcount - 1);  // clamped: every lane loads
            sv[k] = starts[i];
            ev[k] = ends[i];
        }
        // reads are grouped by contig, so almost every tile lies inside one contig
        auto contig_of = [&](uint32_t i) {
            uint32_t lo = 0, hi = n_contigs;  // last c with roff[c] <= i
            if (n_contigs <= 64) {
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_roff[mid] <= i) lo = mid; else hi = mid;
                }
            } else {
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (contig_read_off[mid] <= i) lo = mid; else hi = mid;
                }
            }
            return lo;
        };
        const uint32_t c_first = n_contigs > 1 ? contig_of(tbase) : 0u;
        const uint32_t c_last = n_contigs > 1 ? contig_of(tbase + tcount - 1) : 0u;
        // the usual tile lies inside one contig: its offset and length once per tile (uniform)
        const bool one_contig = c_first == c_last;
        uint32_t tile_p0 = 0, tile_len = 0;
        {
            uint64_t p0, p1;
            if (n_contigs <= 64) { p0 = s_poff[c_first]; p1 = s_poff[c_first + 1]; }
            else { p0 = contig_pos_off[c_first]; p1 = contig_pos_off[c_first + 1]; }
            tile_p0 = (uint32_t)p0;
            tile_len = (uint32_t)(p1 - p0);
        }
        const uint32_t tile_last = tile_len ? tile_len - 1u : 0u;
        uint32_t filled = 0;  // (uniform over the wave) exceptions of this tile the wave has listed
        if (ell_reg != 0u && (threadIdx.x & 63u) == 0u) s_fill[threadIdx.x >> 6] = 0;
        const size_t slot0 = ((size_t)tile * 4u + (threadIdx.x >> 6)) * 128u;
        auto list_exception = [&](bool isx, uint32_t gs, uint32_t ge, uint32_t i) {
            const uint64_t m = __ballot(isx);
            if (m != 0ull) {  // (uniform)
                const uint32_t slot = filled + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (isx && slot < 128u) {
                    exc[slot0 + slot] = gs;
                    exc[exc_cap + slot0 + slot] = ge;
                    exc[2 * (size_t)exc_cap + slot0 + slot] = i;
                } else if (isx) {
                    nu_overflow_put(exc, exc_cap, stats, gs, ge, i);
                }
                filled += (uint32_t)__popcll(m);
            }
        };
        // the ranked route's whole tiles inside one contig (all but a handful): validate, span range and
        // the partition digit, nothing else and no branch per read (an invalid read makes the call fail;
        // what its span adds to the statistics is then never looked at)
        const bool lean = part_hist != nullptr && gstart_out == nullptr && digit0_hist == nullptr &&
                          global_digit_hist == nullptr && cstart == nullptr && one_contig && tcount == 4096u;
        if (lean) {
            // A contig that spans at most four digits (long genomes: the first partition level's digits are
            // 8.4 M positions wide) sends all 64 lanes of a wave to one to four counters, and same-address LDS
            // atomics take their turns: there the wave counts each digit with a ballot and adds once.
            const bool few_digits = ((tile_p0 + tile_last) >> part_shift) - (tile_p0 >> part_shift) < 4u;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t s = sv[k], e = ev[k];
                bad |= (s > e || e >= tile_len) ? 1u : 0u;
                const uint32_t span = e - s + 1;
                mn = min(mn, span);
                mx = max(mx, span);
                // (a start beyond the contig -- the call will fail -- is counted at the contig's last position,
                //  as the partition queued behind this kernel will place it: k_range_partition)
                const uint32_t d = ((tile_p0 + min(s, tile_last)) >> part_shift) & 255u;
                const bool isx = ell_reg != 0u && span != ell_reg;
                if (ell_reg != 0u) list_exception(isx, tile_p0 + min(s, tile_last), tile_p0 + min(e, tile_last), tbase + k * 256u + threadIdx.x);
                if (few_digits) {
                    uint64_t rest = ~__ballot(isx);  // (every lane holds a read: the tile is whole; exceptions are not counted)
                    for (int round = 0; round < 4 && rest != 0; ++round) {
                        const int first = __ffsll((long long)rest) - 1;
                        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
                        const uint64_t same = __ballot(d == d0) & rest;
                        if ((int)(threadIdx.x & 63u) == first) atomicAdd(&s_h[d0], (uint32_t)__popcll(same));
                        rest &= ~same;
                    }
                    // (an invalid start can fall outside the contig's digits: whatever is left goes one by one)
                    if ((rest >> (threadIdx.x & 63u)) & 1ull) atomicAdd(&s_h[d], 1u);
                } else if (!isx) {
                    atomicAdd(&s_h[d], 1u);
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t j = k * 256u + threadIdx.x;
            if (j >= tcount) break;
            const uint32_t i = tbase + j;
            const uint32_t s = sv[k], e = ev[k];
            uint32_t len_c = tile_len, pos0 = tile_p0;
            if (!one_contig) {
                const uint32_t c = contig_of(i);
                uint64_t p0, p1;
                if (n_contigs <= 64) { p0 = s_poff[c]; p1 = s_poff[c + 1]; }
                else { p0 = contig_pos_off[c]; p1 = contig_pos_off[c + 1]; }
                len_c = (uint32_t)(p1 - p0);
                pos0 = (uint32_t)p0;
            }
            const uint32_t gs = pos0 + s;
            if (s > e || e >= len_c) {
                // the call will fail, but kernels queued behind this one before the host knows must
                // stay in bounds: the read is counted under the digit the partition will compute
                bad = 1;
                if (gstart_out) gstart_out[i] = gs;
                if (part_hist) atomicAdd(&s_h[((pos0 + min(s, len_c ? len_c - 1u : 0u)) >> part_shift) & 255u], 1u);
                continue;
            }
            const uint32_t span = e - s + 1;
            mn = min(mn, span);
            mx = max(mx, span);
            if (gstart_out) gstart_out[i] = gs;
            const bool isx = ell_reg != 0u && span != ell_reg;
            if (isx) {
                // (tiles off the lean path are a handful: a slot per exception from the wave's LDS counter)
                const uint32_t slot = atomicAdd(&s_fill[threadIdx.x >> 6], 1u);
                if (slot < 128u) {
                    exc[slot0 + slot] = gs;
                    exc[exc_cap + slot0 + slot] = pos0 + e;
                    exc[2 * (size_t)exc_cap + slot0 + slot] = i;
                } else {
                    nu_overflow_put(exc, exc_cap, stats, gs, pos0 + e, i);
                }
            }
            if (part_hist && !isx) {
                atomicAdd(&s_h[(gs >> part_shift) & 255u], 1u);
                if (digit0_hist) atomicAdd(&s_h0[gs & 255u], 1u);
            }
            if (global_digit_hist) {
                // digit 0 is taken from s_h0 below when it exists; otherwise count it here too
                if (!(part_hist && digit0_hist)) atomicAdd(&s_gh[0][gs & 255u], 1u);
                atomicAdd(&s_gh[1][(gs >> 8) & 255u], 1u);
                atomicAdd(&s_gh[2][(gs >> 16) & 255u], 1u);
                atomicAdd(&s_gh[3][(gs >> 24) & 255u], 1u);
            }
            if (cstart) {
                bool on = true;
                if (keep_mask) on = (keep_mask[i >> 6] >> (i & 63)) & 1ull;
                if (on) atomicAdd(&cstart[gs], 1u);
            }
        }
        if (ell_reg != 0u) {
            if (!lean) filled = s_fill[threadIdx.x >> 6];
            if ((threadIdx.x & 63u) == 0u) {
                exc_cnt[tile * 4u + (threadIdx.x >> 6)] = min(filled, 128u);
            }
        }
        if (part_hist && digit0_hist) {
            __syncthreads();
            digit0_hist[threadIdx.x * n_tiles + tile] = s_h0[threadIdx.x];
            if (global_digit_hist) s_gh[0][threadIdx.x] += s_h0[threadIdx.x];
            __syncthreads();
        }
    }
    if (part_hist) {
        __syncthreads();
        const uint32_t pass0 = t0 >> 1, n_pass = tiles_per_block >> 1;
        uint32_t* row = part_hist + (size_t)threadIdx.x * part_pitch + pass0;
        if (tiles_per_block == 8) {  // pass0 and the pitch are multiples of 4: one aligned 16-byte store
            uint4 a;
            a.x = s_ph[0][threadIdx.x]; a.y = s_ph[1][threadIdx.x]; a.z = s_ph[2][threadIdx.x]; a.w = s_ph[3][threadIdx.x];
            reinterpret_cast<uint4*>(row)[0] = a;
        } else {
            for (uint32_t h = 0; h < n_pass && pass0 + h < part_pitch; ++h) row[h] = s_ph[h][threadIdx.x];
        }
    }
    if (global_digit_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) {
            const uint32_t v = (&s_gh[0][0])[i];
            if (v) atomicAdd(&global_digit_hist[i], v);
        }
    }
    // block reduction, then at most one atomic per statistic per workgroup -- and none when the
    // global value already dominates (same-address atomics serialise; thousands of them cost
    // more than streaming the reads)
    __shared__ uint32_t s_red[3][4];
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    bad = wave_max_u32(bad);
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = mn;
        s_red[1][threadIdx.x >> 6] = mx;
        s_red[2][threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(s_red[0][0], s_red[0][1]), min(s_red[0][2], s_red[0][3]));
        mx = max(max(s_red[1][0], s_red[1][1]), max(s_red[1][2], s_red[1][3]));
        bad = s_red[2][0] | s_red[2][1] | s_red[2][2] | s_red[2][3];
        if (mn < __hip_atomic_load(&stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(&stats[0], mn);
        if (mx > __hip_atomic_load(&stats[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&stats[1], mx);
        if (bad) atomicOr(&stats[2], 1u);
    }
}

// stats[4] = exceptions listed (the sum of the groups' counts)
__global__ __launch_bounds__(256) void k_nu_count_groups(const uint32_t* __restrict__ exc_cnt, uint32_t n_groups,
                                                             uint32_t* __restrict__ stats) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_groups; i += gridDim.x * blockDim.x) acc += exc_cnt[i];
    acc = wave_sum_u32(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0u) atomicAdd(&stats[4], acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&stats[4], min(stats[6], kNuOverflow));  // (the overflow region's)
}

// Mixed-span path: per-position end counts and the composite bucketing key
// (gstart << span_bits) | (max_span - span): ascending key == (start asc, end desc).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_general_keys(const uint32_t* __restrict__ gstart,
                                                      const uint32_t* __restrict__ starts,
                                                      const uint32_t* __restrict__ ends,
                                                      uint32_t n, uint32_t span_bits,
                                                      uint32_t max_span,
                                                      const uint64_t* __restrict__ keep_mask,
                                                      KeyT* __restrict__ keys,
                                                      uint32_t* __restrict__ ecnt) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t span = ends[i] - starts[i] + 1;
        const uint32_t gs = gstart[i];
        if (keys) keys[i] = ((KeyT)gs << span_bits) | (KeyT)(max_span - span);
        if (ecnt) {
            bool on = true;
            if (keep_mask) on = (keep_mask[i >> 6] >> (i & 63)) & 1ull;
            if (on) atomicAdd(&ecnt[gs + span - 1], 1u);
        }
    }
}
// The same for genomes whose end counts fit one workgroup's LDS (amplicon panels: 30 k positions, 10^3 reads
// ending at each): every workgroup counts into its own LDS table and adds its non-zero bins to memory once --
// 30 M same-address global atomics on 30 k counters took 2.4 ms at cfg3's size, this takes 0.1.
template <typename KeyT>
__global__ __launch_bounds__(1024) void k_general_keys_lds(const uint32_t* __restrict__ gstart,
                                                           const uint32_t* __restrict__ starts,
                                                           const uint32_t* __restrict__ ends,
                                                           uint32_t n, uint32_t span_bits, uint32_t max_span,
                                                           const uint64_t* __restrict__ keep_mask,
                                                           KeyT* __restrict__ keys, uint32_t* __restrict__ ecnt,
                                                           uint32_t ecnt_len) {
    extern __shared__ uint32_t s_ecnt[];
    for (uint32_t b = threadIdx.x; b < ecnt_len; b += blockDim.x) s_ecnt[b] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t span = ends[i] - starts[i] + 1;
        const uint32_t gs = gstart[i];
        if (keys) keys[i] = ((KeyT)gs << span_bits) | (KeyT)(max_span - span);
        bool on = true;
        if (keep_mask) on = (keep_mask[i >> 6] >> (i & 63)) & 1ull;
        // (validated reads end inside the genome; the clamp keeps a stray one inside the table)
        if (on) atomicAdd(&s_ecnt[min(gs + span - 1, ecnt_len - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < ecnt_len; b += blockDim.x) {
        const uint32_t v = s_ecnt[b];
        if (v != 0) atomicAdd(&ecnt[b], v);
    }
}
template __global__ void k_general_keys_lds<uint32_t>(const uint32_t*, const uint32_t*, const uint32_t*, uint32_t,
                                                      uint32_t, uint32_t, const uint64_t*, uint32_t*, uint32_t*, uint32_t);
template __global__ void k_general_keys_lds<uint64_t>(const uint32_t*, const uint32_t*, const uint32_t*, uint32_t,
                                                      uint32_t, uint32_t, const uint64_t*, uint64_t*, uint32_t*, uint32_t);
template __global__ void k_general_keys<uint32_t>(const uint32_t*, const uint32_t*, const uint32_t*,
                                                  uint32_t, uint32_t, uint32_t, const uint64_t*,
                                                  uint32_t*, uint32_t*);
template __global__ void k_general_keys<uint64_t>(const uint32_t*, const uint32_t*, const uint32_t*,
                                                  uint32_t, uint32_t, uint32_t, const uint64_t*,
                                                  uint64_t*, uint32_t*);

// ------------------------------------------------------------------ exclusive scan (u32)
// Three launches: tile sums -> spine scan (one workgroup) -> tile scan with carried offset.
// out may alias in.  out has n+1 entries when write_total is set (out[n] = grand total).
static constexpr int kScanThreads = 256;
static constexpr int kScanItems = 16;
static constexpr int kScanTile = kScanThreads * kScanItems;  // 4096

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* s_wave,
                                                        uint32_t& block_total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan_add(v);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t x = s_wave[k];
        if (k < w) base += x;
        tot += x;
    }
    block_total = tot;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const uint32_t* __restrict__ in,
                                                                  uint32_t n,
                                                                  uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k * kScanThreads + threadIdx.x;
        if (i < n) acc += in[i];
    }
    acc = wave_sum_u32(acc);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// single workgroup, exclusive in place; spine[n_tiles] = total
__global__ __launch_bounds__(kScanThreads) void k_scan_spine(uint32_t* __restrict__ spine,
                                                              uint32_t n_tiles) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_tiles; base += kScanThreads) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_tiles ? spine[i] : 0;
        uint32_t tot;
        uint32_t ex = block_excl_scan_256(v, s_wave, tot);
        if (i < n_tiles) spine[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) spine[n_tiles] = carry;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tiles(const uint32_t* __restrict__ in,
                                                              uint32_t n,
                                                              const uint32_t* __restrict__ spine,
                                                              uint32_t* __restrict__ out,
                                                              int write_total) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t sum = 0;
    // a thread owns 16 consecutive entries (64 bytes): whole tiles of 16-byte-aligned tables move as
    // four 16-byte accesses per thread instead of sixteen lone words 64 bytes apart across the wave
    const bool vec = (uint64_t)blockIdx.x * kScanTile + kScanTile <= n &&
                     (((uintptr_t)in | (uintptr_t)out) & 15u) == 0;  // uniform over the workgroup
    if (vec) {
        const uint4* in4 = reinterpret_cast<const uint4*>(in + base);
#pragma unroll
        for (int k = 0; k < kScanItems / 4; ++k) {
            const uint4 x = in4[k];
            v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
        }
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) sum += v[k];
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            uint32_t i = base + k;
            v[k] = i < n ? in[i] : 0;
            sum += v[k];
        }
    }
    uint32_t tot;
    uint32_t run = spine[blockIdx.x] + block_excl_scan_256(sum, s_wave, tot);
    if (vec) {
        uint4* out4 = reinterpret_cast<uint4*>(out + base);
#pragma unroll
        for (int k = 0; k < kScanItems / 4; ++k) {
            uint4 y;
            y.x = run; run += v[4 * k];
            y.y = run; run += v[4 * k + 1];
            y.z = run; run += v[4 * k + 2];
            y.w = run; run += v[4 * k + 3];
            out4[k] = y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            uint32_t i = base + k;
            if (i < n) out[i] = run;
            run += v[k];
        }
    }
    if (write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = spine[gridDim.x];
}
