// qmcp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the quasi-MCP solver path.
//
// Everything here is integer scatter / scan / selection: HBM- or latency-bound, no MFMA.
// wave = 64 lanes throughout.  The path these kernels replace is the reference's
//   coverage build        quasi_mcp_cpu_max_flow_solver.cpp:58-73 (O(N*len) per-base loop)
//   max-flow + readout    quasi_mcp_cpu_max_flow_solver.cpp:19-20,89-100
//   (CUDA equivalent      quasi_mcp_cuda_max_flow_solver.cu:12-79,319-435)
// with the canonical selection rule stated in oracle/qmcp_oracle.c.
//
// Coordinates: contigs are concatenated into one global axis; gpos = pos_offset[c] + pos,
// Ltot = sum of contig lengths.  Reads never cross a contig, so per-position prefix counts
// taken over the global axis cancel exactly at contig borders.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmcp_kernels.h"

namespace qmcp {

static constexpr int kWave = 64;
// "no constraint".  Real counts stay < 2^28 per contig (checked by the host).  In the map
// algebra below b, u and v never accumulate (a composite's b is <= its first element's b, its
// u and v are <= its last element's), so the largest intermediate is u + b <= 2 kInf + 2^29
// < 2^32: nothing wraps, no saturation is needed, and anything >= kInf just means "infinite"
// (every finite value is < 2^29 < kInf).
static constexpr uint32_t kInf = 0x40000000u;

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, kWave));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, kWave));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, kWave);
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t w = (uint64_t)__shfl_xor((long long)v, o, kWave);
        v = v > w ? v : w;
    }
    return v;
}

// DPP-based wave64 inclusive scans (row_shr within 16-lane rows, then row_bcast:15 / :31).
// `id` is the identity the shifted-in lanes see.
#define QMCP_DPP(old, src, ctrl, rmask) \
    (uint32_t) __builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rmask), 0xF, false)

__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v) {
    v += QMCP_DPP(0u, v, 0x111, 0xF);
    v += QMCP_DPP(0u, v, 0x112, 0xF);
    v += QMCP_DPP(0u, v, 0x114, 0xF);
    v += QMCP_DPP(0u, v, 0x118, 0xF);
    v += QMCP_DPP(0u, v, 0x142, 0xA);
    v += QMCP_DPP(0u, v, 0x143, 0xC);
    return v;
}
// ------------------------------------------------------------------ prepare
// One pass over the reads: validate (start <= end < contig length), reduce min/max span,
// write the global start position of every read (the bucketing key of the uniform path) and
// count reads per start position.  Reference counterpart: the read loop of
// create_b_function (quasi_mcp_cpu_max_flow_solver.cpp:61-67) -- here O(1) per read.
//
// stats[0] = min span, stats[1] = max span, stats[2] = error flag
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
                                                 uint32_t* __restrict__ part_hist,
                                                 uint32_t* __restrict__ digit0_hist,
                                                 uint32_t* __restrict__ global_digit_hist,
                                                 unsigned long long* __restrict__ zero_mask) {
    __shared__ uint32_t s_gh[4][256];  // whole-call digit histograms of the start key (all 4 bytes)
    if (global_digit_hist) {
        for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) (&s_gh[0][0])[i] = 0;
    }
    __shared__ uint64_t s_roff[65];
    __shared__ uint64_t s_poff[65];
    __shared__ uint32_t s_h[256];
    __shared__ uint32_t s_h0[256];
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
        if (part_hist) {
            s_h[threadIdx.x] = 0;
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
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t j = k * 256u + threadIdx.x;
            if (j >= tcount) break;
            const uint32_t i = tbase + j;
            const uint32_t s = sv[k], e = ev[k];
            const uint32_t c = c_first == c_last ? c_first : contig_of(i);
            uint64_t p0, p1;
            if (n_contigs <= 64) { p0 = s_poff[c]; p1 = s_poff[c + 1]; }
            else { p0 = contig_pos_off[c]; p1 = contig_pos_off[c + 1]; }
            const uint32_t len_c = (uint32_t)(p1 - p0);
            const uint32_t gs = (uint32_t)p0 + s;
            if (s > e || e >= len_c) {
                // the call will fail, but kernels queued behind this one before the host knows must
                // stay in bounds: the read is counted under the digit the partition will compute
                bad = 1;
                if (gstart_out) gstart_out[i] = gs;
                if (part_hist) atomicAdd(&s_h[(gs >> part_shift) & 255u], 1u);
                continue;
            }
            const uint32_t span = e - s + 1;
            mn = min(mn, span);
            mx = max(mx, span);
            if (gstart_out) gstart_out[i] = gs;
            if (part_hist) {
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
        if (part_hist) {
            __syncthreads();
            part_hist[threadIdx.x * n_tiles + tile] = s_h[threadIdx.x];
            if (digit0_hist) {
                digit0_hist[threadIdx.x * n_tiles + tile] = s_h0[threadIdx.x];
                if (global_digit_hist) s_gh[0][threadIdx.x] += s_h0[threadIdx.x];
            }
            __syncthreads();
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
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k;
        v[k] = i < n ? in[i] : 0;
        sum += v[k];
    }
    uint32_t tot;
    uint32_t run = spine[blockIdx.x] + block_excl_scan_256(sum, s_wave, tot);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
    if (write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = spine[gridDim.x];
}

// ------------------------------------------------------------------ LSD radix sort, 8-bit digits
// Stable, keys + u32 payload.  Tile = 256 threads x 16 keys; wave w of a tile owns a
// contiguous 1024-key slice and walks it 64 keys at a time, so tile order == memory order.
static constexpr int kSortThreads = 256;
static constexpr int kSortItems = 16;
static constexpr int kSortTile = kSortThreads * kSortItems;  // 4096

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const KeyT* __restrict__ keys,
                                                              uint32_t n, uint32_t shift,
                                                              uint32_t n_tiles,
                                                              uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        uint32_t i = base + k * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + blockIdx.x] = s_h[threadIdx.x];  // digit-major
}

template <typename KeyT>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter(const KeyT* __restrict__ keys_in,
                                                                 const uint32_t* __restrict__ vals_in,
                                                                 uint32_t n, uint32_t shift,
                                                                 uint32_t n_tiles,
                                                                 const uint32_t* __restrict__ offs,
                                                                 KeyT* __restrict__ keys_out,
                                                                 uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_cnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();

    const uint32_t wbase = blockIdx.x * kSortTile + w * (kSortItems * 64);
    KeyT key[kSortItems];
    uint32_t rank[kSortItems];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : (KeyT)0;
        const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
        // lanes holding the same digit (invalid lanes form their own group)
        uint64_t peers = __ballot(valid);
        if (!valid) peers = ~peers;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t in_group = __popcll(peers & lt_mask);
        const int leader = __ffsll((long long)peers) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) {
            old = s_cnt[w][d];
            s_cnt[w][d] = old + __popcll(peers);
        }
        old = (uint32_t)__shfl((int)old, leader, kWave);
        rank[k] = old + in_group;
    }
    __syncthreads();
    {
        // digit = threadIdx.x: turn per-wave counts into absolute output bases
        const uint32_t d = threadIdx.x;
        uint32_t run = offs[d * n_tiles + blockIdx.x];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const uint32_t c = s_cnt[ww][d];
            s_cnt[ww][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[k] >> shift) & 255u;
            const uint32_t dst = s_cnt[w][d] + rank[k];
            keys_out[dst] = key[k];
            vals_out[dst] = vals_in ? vals_in[i] : i;
        }
    }
}

// ------------------------------------------------------------------ record radix (32-bit keys)
// Same stable LSD pass on {key, read index} records (one 8-byte store per element), with the
// tile reordered through LDS so that consecutive lanes store to consecutive addresses: a
// tile's elements of one digit leave as one contiguous run.  The first pass reads bare keys
// (the payload is the element's own index).
struct Rec { uint32_t key, val; };

template <bool FIRST>
__device__ __forceinline__ uint32_t rec_key(const uint32_t* __restrict__ keys,
                                            const Rec* __restrict__ recs, uint32_t i) {
    if (FIRST) return keys[i];
    return recs[i].key;
}

// A workgroup handles `tiles_per_block` consecutive tiles so that its accesses to the
// digit-major table (stride n_tiles between digits) touch runs of consecutive entries.
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist_rec(const uint32_t* __restrict__ keys,
                                                                  const Rec* __restrict__ recs,
                                                                  uint32_t n, uint32_t shift,
                                                                  uint32_t n_tiles,
                                                                  uint32_t tiles_per_block,
                                                                  uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    const uint32_t t0 = blockIdx.x * tiles_per_block;
    for (uint32_t g = 0; g < tiles_per_block && t0 + g < n_tiles; ++g) {
        s_h[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t base = (t0 + g) * kSortTile;
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            uint32_t i = base + k * kSortThreads + threadIdx.x;
            if (i < n) atomicAdd(&s_h[(rec_key<FIRST>(keys, recs, i) >> shift) & 255u], 1u);
        }
        __syncthreads();
        hist[threadIdx.x * n_tiles + t0 + g] = s_h[threadIdx.x];  // digit-major
        __syncthreads();
    }
}

// OUT_KEYS: emit bare keys (u32) instead of records -- used by the counting partition.
template <bool FIRST, bool OUT_KEYS>
__global__ __launch_bounds__(kSortThreads) void k_radix_scatter_rec(
    const uint32_t* __restrict__ keys, const Rec* __restrict__ recs_in, uint32_t n, uint32_t shift,
    uint32_t n_tiles, uint32_t tiles_per_block, const uint32_t* __restrict__ offs,
    void* __restrict__ out) {
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_wave[4];
    __shared__ Rec s_rec[kSortTile];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t t0 = blockIdx.x * tiles_per_block;
    for (uint32_t g = 0; g < tiles_per_block && t0 + g < n_tiles; ++g) {
        const uint32_t tile = t0 + g;
        for (int i = threadIdx.x; i < 4 * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
        __syncthreads();

        const uint32_t tile_base = tile * kSortTile;
        const uint32_t tile_count = min((uint32_t)kSortTile, n - tile_base);
        const uint32_t wbase = tile_base + w * (kSortItems * 64);
        Rec rec[kSortItems];
        uint32_t rank[kSortItems];
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t i = wbase + k * 64 + lane;
            const bool valid = i < n;
            if (FIRST) { rec[k].key = valid ? keys[i] : 0u; rec[k].val = i; }
            else { rec[k] = valid ? recs_in[i] : Rec{0u, 0u}; }
            const uint32_t d = (rec[k].key >> shift) & 255u;
            uint64_t peers = __ballot(valid);
            if (!valid) peers = ~peers;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t in_group = __popcll(peers & lt_mask);
            const int leader = __ffsll((long long)peers) - 1;
            uint32_t old = 0;
            if (valid && lane == leader) {
                old = s_cnt[w][d];
                s_cnt[w][d] = old + __popcll(peers);
            }
            old = (uint32_t)__shfl((int)old, leader, kWave);
            rank[k] = old + in_group;
        }
        __syncthreads();
        {
            // digit = threadIdx.x: position of (wave, digit) inside the tile's digit-sorted
            // order, and the global base of the digit's run
            const uint32_t d = threadIdx.x;
            const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
            uint32_t tot;
            const uint32_t tile_off = block_excl_scan_256(c0 + c1 + c2 + c3, s_wave, tot);
            s_cnt[0][d] = tile_off;
            s_cnt[1][d] = tile_off + c0;
            s_cnt[2][d] = tile_off + c0 + c1;
            s_cnt[3][d] = tile_off + c0 + c1 + c2;
            s_gbase[d] = offs[d * n_tiles + tile] - tile_off;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t i = wbase + k * 64 + lane;
            if (i < n) {
                const uint32_t d = (rec[k].key >> shift) & 255u;
                s_rec[s_cnt[w][d] + rank[k]] = rec[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSortItems; ++k) {
            const uint32_t j = k * kSortThreads + threadIdx.x;
            if (j < tile_count) {
                const Rec r = s_rec[j];
                const uint32_t d = (r.key >> shift) & 255u;
                if (OUT_KEYS) ((uint32_t*)out)[s_gbase[d] + j] = r.key;
                else ((Rec*)out)[s_gbase[d] + j] = r;
            }
        }
        __syncthreads();
    }
}

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

template <int MODE, bool OUT_REC>
__global__ __launch_bounds__(kPartThreads) void k_range_partition(
    const uint32_t* __restrict__ keys, const Rec* __restrict__ recs_in, SegTables seg,
    const uint64_t* __restrict__ contig_read_off, const uint64_t* __restrict__ contig_pos_off,
    uint32_t n_contigs, uint32_t n, uint32_t shift, uint32_t n_tiles, const uint32_t* __restrict__ offs,
    uint16_t* __restrict__ out_key, uint32_t* __restrict__ out_idx, Rec* __restrict__ out_rec,
    uint32_t* __restrict__ range_start, uint32_t* __restrict__ max_load) {
    extern __shared__ uint32_t s_part[];
    Rec* s_rec = reinterpret_cast<Rec*>(s_part);                       // [kPartRecs]
    uint32_t* s_cnt = s_part + 2 * kPartRecs;                          // [kPartWaves][256]
    uint32_t* s_gbase = s_cnt + kPartWaves * 256;                      // [256]
    uint32_t* s_wave = s_gbase + 256;                                  // [4] (+ pad to 16)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
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
        off0 = blockIdx.x * kPartTiles;
        off_stride = n_tiles;
    }
    const uint32_t count = min((uint32_t)kPartRecs, bound - base);
    for (int i = threadIdx.x; i < kPartWaves * 256; i += kPartThreads) s_cnt[i] = 0;
    if (MODE != 2 && blockIdx.x == 0) {
        // the first workgroup also publishes where every range's records begin (257 entries) and the
        // heaviest range's load, for the per-range kernels and the host's balance test
        uint32_t load = 0;
        if (threadIdx.x < 256) {
            const uint32_t d = threadIdx.x;
            const uint32_t r_lo = offs[d * n_tiles];
            const uint32_t r_hi = d + 1 < 256 ? offs[(d + 1) * n_tiles] : n;
            range_start[d] = r_lo;
            if (d == 255) range_start[256] = n;
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
    // wave w owns records [w * 1024, (w + 1) * 1024) of the pass, in 16 rounds of 64: order inside a
    // range = (wave, round, lane) = input order
    const uint32_t wbase = base + w * (kSortItems * 64);
    Rec rec[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (MODE == 2) {
            rec[k] = i < bound ? recs_in[i] : Rec{0u, 0u};
        } else {
            rec[k].key = i < bound ? keys[i] : 0u;
            rec[k].val = i;
        }
    }
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
        if (c_first == c_last) {
            const uint32_t p0 = (uint32_t)contig_pos_off[c_first];
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) rec[k].key += p0;
        } else {
#pragma unroll
            for (int k = 0; k < kSortItems; ++k) {
                const uint32_t i = wbase + k * 64 + lane;
                if (i < bound) rec[k].key += (uint32_t)contig_pos_off[contig_of(i)];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < bound;
        const uint32_t d = (rec[k].key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
        if (!valid) peers = ~peers;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t in_group = __popcll(peers & lt_mask);
        const int leader = __ffsll((long long)peers) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) {
            old = s_cnt[w * 256 + d];
            s_cnt[w * 256 + d] = old + __popcll(peers);
        }
        old = (uint32_t)__shfl((int)old, leader, kWave);
        rank[k] = old + in_group;
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
        s_gbase[d] = offs[off0 + d * off_stride] - tile_off;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < bound) {
            const uint32_t d = (rec[k].key >> shift) & 255u;
            s_rec[s_cnt[w * 256 + d] + rank[k]] = rec[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t j = k * kPartThreads + threadIdx.x;
        if (j < count) {
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
    if (h == 255 && d == 255) range_start[65536] = n;
    const uint32_t m = wave_max_u32(hi - lo);
    if ((d & 63) == 0) s_red[d >> 6] = m;
    __syncthreads();
    if (d == 0) atomicMax(max_load, max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
}

__global__ __launch_bounds__(1024) void k_range_offsets(const uint16_t* __restrict__ keys16,
                                                        const uint32_t* __restrict__ range_start,
                                                        uint32_t shift, uint32_t ltot,
                                                        uint32_t* __restrict__ boff) {
    // [1 << shift] counters, one pad word after every 32: a thread's 32 consecutive positions then
    // sit in 32 different banks during the scan
    extern __shared__ uint32_t s_cnt32[];
#define PADDED(i) ((i) + ((i) >> 5))
    __shared__ uint32_t s_wsum[16];
    const uint32_t range = blockIdx.x, width = 1u << shift, pos0 = range << shift;
    for (uint32_t i = threadIdx.x; i < width; i += blockDim.x) s_cnt32[PADDED(i)] = 0;
    __syncthreads();
    const uint32_t lo = range_start[range], hi = range_start[range + 1];
    // eight loads in flight per thread: the range's records stream in at L2 speed instead of one
    // round trip per iteration
    constexpr int U = 8;
    uint32_t j = lo + threadIdx.x;
    for (; j + (U - 1) * 1024u < hi; j += U * 1024u) {
        uint32_t k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = keys16[j + u * 1024u];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k[u] < width) atomicAdd(&s_cnt32[PADDED(k[u])], 1u);
    }
    for (; j < hi; j += 1024u) {
        const uint32_t li = keys16[j];
        if (li < width) atomicAdd(&s_cnt32[PADDED(li)], 1u);
    }
    __syncthreads();
    // counts -> bucket offsets, in place: exclusive scan over the range's positions, started at
    // the number of records in all lower ranges (= the offset of the range's first position), so
    // the array needs no separate scan pass over the whole genome
    const uint32_t per = width >= 1024u ? width >> 10 : 1u;  // positions per thread
    const uint32_t first = threadIdx.x * per;
    uint32_t sum = 0;
    if (first < width)
        for (uint32_t q = 0; q < per; ++q) sum += s_cnt32[PADDED(first + q)];
    const uint32_t inc = wave_incl_scan_add(sum);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
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
static constexpr int kRankU = 1;

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
    uint2* const amb = amb_lists + (lists_by_records ? (size_t)range_start[range] : (size_t)range * width);
    for (uint32_t i = tid; i < live; i += nthreads)
        s_q[i] = (int32_t)(selend[pos0 + i] - boff[pos0 + i]);
    if (tid == 0) s_namb = 0;
    const uint32_t lo = range_start[range], hi = range_start[range + 1];
    if (lo >= hi) return;  // uniform
    __syncthreads();
    const uint32_t chunk_recs = kRankU * nthreads;
    const uint32_t n_chunks = (hi - lo + chunk_recs - 1) / chunk_recs;
    uint32_t kept = 0;

    // thread tid owns record tid of each of the chunk's kRankU sub-chunks
    struct Recs { uint32_t key[kRankU], val[kRankU]; };
    auto fetch = [&](Recs& dst, uint32_t c) {
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            const uint32_t j = min(lo + c * chunk_recs + u * nthreads + tid, hi - 1);
            dst.key[u] = keys16[j];  // position inside the range
            dst.val[u] = idx[j];
        }
    };
    auto chunk = [&](const Recs& r, uint32_t c) {
        bool valid[kRankU];
        uint32_t li[kRankU];
        int32_t old[kRankU];
#pragma unroll
        for (int u = 0; u < kRankU; ++u) {
            valid[u] = lo + c * chunk_recs + u * nthreads + tid < hi;
            li[u] = valid[u] ? r.key[u] : width;
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
            if (keep[u]) atomicOr(&mask[r.val[u] >> 6], 1ull << (r.val[u] & 63u));
            kept += (uint32_t)__popcll(__ballot(keep[u]));
        }
    };
    // Records are prefetched seven chunks ahead into eight register sets that rotate by NAME (the
    // loop is unrolled eight times): no register copies, so the wait for a chunk's records is a
    // counted s_waitcnt that leaves the younger loads (and the fire-and-forget mask atomics) in
    // flight.  A thread loads only 6 bytes per chunk, so this depth is what keeps enough bytes in
    // flight per CU (3 chunks ahead: 2.2 TB/s over the chip).  Chunks past the end run with every
    // thread idle (dummy quota slot).
    Recs R0, R1, R2, R3, R4, R5, R6, R7;
    fetch(R0, 0); fetch(R1, 1); fetch(R2, 2); fetch(R3, 3); fetch(R4, 4); fetch(R5, 5); fetch(R6, 6);
    for (uint32_t c = 0; c < n_chunks; c += 8) {
        fetch(R7, c + 7);  chunk(R0, c);
        fetch(R0, c + 8);  chunk(R1, c + 1);
        fetch(R1, c + 9);  chunk(R2, c + 2);
        fetch(R2, c + 10); chunk(R3, c + 3);
        fetch(R3, c + 11); chunk(R4, c + 4);
        fetch(R4, c + 12); chunk(R5, c + 5);
        fetch(R5, c + 13); chunk(R6, c + 6);
        fetch(R6, c + 14); chunk(R7, c + 7);
    }
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
    if (lane == 0 && kept) atomicAdd(kept_total, (unsigned long long)kept);
}

// ------------------------------------------------------------------ chained radix pass
// One kernel per digit instead of histogram + scan + scatter: a tile publishes its digit counts
// and obtains its base by looking back over its predecessors' published values (decoupled
// look-back).  Inter-workgroup protocol (cdna_hip_programming.md, Guideline 16, recipe R2): every
// (tile, digit) status is ONE aligned 8-byte granule {epoch << 2 | state, value}, written by one
// relaxed agent-scope atomic store and polled with relaxed agent-scope atomic loads -- the data
// is the flag, no fence, no separate flag word.  state 1 = this tile's own count ("aggregate"),
// 2 = inclusive prefix over all tiles up to this one.  The epoch (unique per pass for the life of
// the context) makes stale granules of earlier passes read as "not published", so the table is
// zeroed only when it is (re)allocated.
// Forward progress: tile numbers are drawn from an atomic ticket in the order workgroups start,
// so every predecessor of a running tile is itself running or finished; predecessors never wait
// on successors.  Every spin is bounded; on expiry the tile raises `timeout_flag` and the host
// redoes the bucketing with the three-kernel passes.
// `digit_base` = exclusive scan of this pass's whole-call digit histogram (from k_prepare).
static constexpr uint32_t kSpinLimit = 1u << 22;

__global__ __launch_bounds__(256) void k_digit_bases(const uint32_t* __restrict__ hist4,
                                                     uint32_t* __restrict__ base4) {
    __shared__ uint32_t s_wave[4];
    for (int p = 0; p < 4; ++p) {
        uint32_t tot;
        const uint32_t v = hist4[p * 256 + threadIdx.x];
        base4[p * 256 + threadIdx.x] = block_excl_scan_256(v, s_wave, tot);
    }
}

template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void k_radix_onesweep(
    const uint32_t* __restrict__ keys, const Rec* __restrict__ recs_in, uint32_t n, uint32_t shift,
    uint32_t n_tiles, const uint32_t* __restrict__ digit_base, unsigned long long* __restrict__ status,
    uint32_t epoch, uint32_t* __restrict__ ticket, uint32_t* __restrict__ timeout_flag,
    Rec* __restrict__ recs_out) {
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_tile;
    __shared__ Rec s_rec[kSortTile];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = threadIdx.x; i < 4 * 256; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= n_tiles) return;  // uniform

    const uint32_t tile_base = tile * kSortTile;
    const uint32_t tile_count = min((uint32_t)kSortTile, n - tile_base);
    const uint32_t wbase = tile_base + w * (kSortItems * 64);
    Rec rec[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < n;
        if (FIRST) { rec[k].key = valid ? keys[i] : 0u; rec[k].val = i; }
        else { rec[k] = valid ? recs_in[i] : Rec{0u, 0u}; }
        const uint32_t d = (rec[k].key >> shift) & 255u;
        uint64_t peers = __ballot(valid);
        if (!valid) peers = ~peers;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t in_group = __popcll(peers & lt_mask);
        const int leader = __ffsll((long long)peers) - 1;
        uint32_t old = 0;
        if (valid && lane == leader) {
            old = s_cnt[w][d];
            s_cnt[w][d] = old + __popcll(peers);
        }
        old = (uint32_t)__shfl((int)old, leader, kWave);
        rank[k] = old + in_group;
    }
    __syncthreads();
    {
        const uint32_t d = threadIdx.x;  // one thread per digit from here to the next barrier
        const uint32_t c0 = s_cnt[0][d], c1 = s_cnt[1][d], c2 = s_cnt[2][d], c3 = s_cnt[3][d];
        const uint32_t mine = c0 + c1 + c2 + c3;
        unsigned long long* row = status + (size_t)tile * 256;
        const unsigned long long tag_agg = ((unsigned long long)((epoch << 2) | 1u)) << 32;
        const unsigned long long tag_pre = ((unsigned long long)((epoch << 2) | 2u)) << 32;
        if (tile > 0)
            __hip_atomic_store(&row[d], tag_agg | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // look back: sum aggregates until a tile with a published inclusive prefix is met
        uint32_t excl = 0;
        uint32_t spins = 0;
        bool failed = false;
        for (uint32_t t = tile; t > 0 && !failed;) {
            const unsigned long long g = __hip_atomic_load(&status[(size_t)(t - 1) * 256 + d],
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t hi = (uint32_t)(g >> 32);
            if ((hi >> 2) != epoch || (hi & 3u) == 0u) {  // not published yet in this pass
                if (++spins > kSpinLimit) failed = true;
                else __builtin_amdgcn_s_sleep(1);
                continue;
            }
            excl += (uint32_t)g;
            if ((hi & 3u) == 2u) break;
            --t;
        }
        if (failed) atomicOr(timeout_flag, 1u);
        __hip_atomic_store(&row[d], tag_pre | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t tot;
        const uint32_t tile_off = block_excl_scan_256(mine, s_wave, tot);
        s_cnt[0][d] = tile_off;
        s_cnt[1][d] = tile_off + c0;
        s_cnt[2][d] = tile_off + c0 + c1;
        s_cnt[3][d] = tile_off + c0 + c1 + c2;
        s_gbase[d] = digit_base[d] + excl - tile_off;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < n) {
            const uint32_t d = (rec[k].key >> shift) & 255u;
            s_rec[s_cnt[w][d] + rank[k]] = rec[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortItems; ++k) {
        const uint32_t j = k * kSortThreads + threadIdx.x;
        if (j < tile_count) {
            const Rec r = s_rec[j];
            const uint32_t d = (r.key >> shift) & 255u;
            // a timed-out look-back leaves a wrong base: keep the store inside the buffer
            const uint32_t dst = s_gbase[d] + j;
            if (dst < n) recs_out[dst] = r;
        }
    }
}

// ------------------------------------------------------------------ bucket offsets from sorted keys
// boff[q] = first sorted entry whose start position is >= q.  Run heads write their own
// slot (boff pre-filled with 0xFFFFFFFF, boff[ltot] = n); a reverse inclusive min-scan then
// fills the positions nobody starts at.  No atomics, any gap structure.
struct KeysRec { const Rec* r; __device__ uint32_t pos(uint32_t j, uint32_t sb) const { return r[j].key >> sb; }
                 __device__ uint32_t idx(uint32_t j) const { return r[j].val; } };
struct KeysSplit64 { const uint64_t* k; const uint32_t* v;
                     __device__ uint32_t pos(uint32_t j, uint32_t sb) const { return (uint32_t)(k[j] >> sb); }
                     __device__ uint32_t idx(uint32_t j) const { return v[j]; } };

template <typename Keys>
__global__ __launch_bounds__(256) void k_bucket_heads(Keys keys, uint32_t n, uint32_t span_bits,
                                                      uint32_t ltot, uint32_t* __restrict__ boff) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t q = keys.pos(j, span_bits);
        if (j == 0 || keys.pos(j - 1, span_bits) != q) boff[q] = j;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) boff[ltot] = n;
}

// reverse inclusive min-scan, in place: data[i] = min(data[i .. n-1]).  Implemented as a
// forward scan over mirrored indices m -> n-1-m.
__global__ __launch_bounds__(kScanThreads) void k_rmin_tile_mins(const uint32_t* __restrict__ data,
                                                                  uint32_t n,
                                                                  uint32_t* __restrict__ tile_mins) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile;
    uint32_t acc = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        uint32_t m = base + k * kScanThreads + threadIdx.x;
        if (m < n) acc = min(acc, data[n - 1 - m]);
    }
    acc = wave_min_u32(acc);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) tile_mins[blockIdx.x] = min(min(s_wave[0], s_wave[1]), min(s_wave[2], s_wave[3]));
}

__device__ __forceinline__ uint32_t wave_incl_scan_min_full(uint32_t v) {
    const uint32_t id = 0xFFFFFFFFu;
    v = min(v, QMCP_DPP(id, v, 0x111, 0xF));
    v = min(v, QMCP_DPP(id, v, 0x112, 0xF));
    v = min(v, QMCP_DPP(id, v, 0x114, 0xF));
    v = min(v, QMCP_DPP(id, v, 0x118, 0xF));
    v = min(v, QMCP_DPP(id, v, 0x142, 0xA));
    v = min(v, QMCP_DPP(id, v, 0x143, 0xC));
    return v;
}
// exclusive min-scan across the 256 threads of a block (identity 0xFFFFFFFF)
__device__ __forceinline__ uint32_t block_excl_minscan_256(uint32_t v, uint32_t* s_wave,
                                                           uint32_t& block_min) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan_min_full(v);
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint32_t before = 0xFFFFFFFFu, all = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = s_wave[k];
        if (k < w) before = min(before, x);
        all = min(all, x);
    }
    block_min = all;
    __syncthreads();
    uint32_t prev = QMCP_DPP(0xFFFFFFFFu, inc, 0x138, 0xF);  // wave_shr:1
    return min(before, prev);
}

__global__ __launch_bounds__(kScanThreads) void k_rmin_spine(uint32_t* __restrict__ spine,
                                                              uint32_t n_tiles) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < n_tiles; base += kScanThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? spine[i] : 0xFFFFFFFFu;
        uint32_t all;
        const uint32_t ex = block_excl_minscan_256(v, s_wave, all);
        if (i < n_tiles) spine[i] = min(carry, ex);
        carry = min(carry, all);
    }
}

__global__ __launch_bounds__(kScanThreads) void k_rmin_tiles(uint32_t* __restrict__ data, uint32_t n,
                                                              const uint32_t* __restrict__ spine) {
    __shared__ uint32_t s_wave[4];
    const uint32_t base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t m = base + k;
        v[k] = m < n ? data[n - 1 - m] : 0xFFFFFFFFu;
        mn = min(mn, v[k]);
    }
    uint32_t all;
    uint32_t run = min(spine[blockIdx.x], block_excl_minscan_256(mn, s_wave, all));
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const uint32_t m = base + k;
        run = min(run, v[k]);
        if (m < n) data[n - 1 - m] = run;
    }
}

// ------------------------------------------------------------------ uniform-span sweep
// All reads of the call have span `ell`.  With W(p) = #dropped reads with start <= p the
// canonical greedy (oracle/qmcp_oracle.c) is the pointwise-maximal W under
//     0 <= W(p) - W(p-1) <= c(p)          c(p)  = reads starting at p
//     W(p) - W(p-ell) <= ex(p)            ex(p) = cov(p) - min(cov(p), M)
// i.e. single-source shortest paths on a line graph with edges p-1 -> p (c(p)),
// p-ell -> p (ex(p)) and p -> p-1 (0).  Distances obey
//     d(p) = min( d(p-1) + c(p),  min_{j in [p-ell, p-1]} ( d(j) + ex(j+ell) ) )
// Positions are processed in blocks of `ell`.  The window minimum splits into
//     A(p) = suffix minimum over the previous block of h(j) = d(j) + ex(j+ell)
//            (held in registers: same lane, same slot as p), and
//     m(p) = running minimum of h over the current block before p.
// Carrying (d, m) turns one position into the min-plus map
//     d' = min(d + c, m, A)          m' = min(m, d' + ex) = min(d + c + ex, m, A + ex)
// and maps of the form  d' = min(d + a, m, u),  m' = min(d + b, m, v)  (b >= a, v >= u) are
// closed under composition:
//     a = min(a1 + a2, b1)   b = min(a1 + b2, b1)
//     u = min(u1 + a2, v1, u2)   v = min(u1 + b2, v1, v2)
// so a block is ONE wave-wide inclusive scan of 4-tuples (DPP row shifts + row broadcasts),
// a lane-local replay, and a suffix-min for the next block -- no iteration, no LDS.
// Selected count at p: S(p) = c(p) - (d(p) - d(p-1)); the kept reads of a start position are
// its S(p) lowest read indices (all ends are equal, so the rule's tie-break is the index).
//
// One wave per contig; local index i = lane * E + r, valid while i < ell.  Loads of block
// b+1 are issued before block b is computed (the chain is latency-bound).
struct Map4 { uint32_t a, b, u, v; };
__device__ __forceinline__ Map4 map_identity() { return Map4{0u, kInf, kInf, kInf}; }
__device__ __forceinline__ Map4 map_compose(const Map4& f, const Map4& g) {  // f first, then g
    Map4 r;
    r.a = min(f.a + g.a, f.b);
    r.b = min(f.a + g.b, f.b);
    r.u = min(min(f.u + g.a, f.v), g.u);
    r.v = min(min(f.u + g.b, f.v), g.v);
    return r;
}
__device__ __forceinline__ Map4 wave_incl_scan_map(Map4 x) {
#define QMCP_STEP(ctrl, rmask)                                  \
    {                                                           \
        Map4 p;                                                 \
        p.a = QMCP_DPP(0u, x.a, ctrl, rmask);                   \
        p.b = QMCP_DPP(kInf, x.b, ctrl, rmask);                 \
        p.u = QMCP_DPP(kInf, x.u, ctrl, rmask);                 \
        p.v = QMCP_DPP(kInf, x.v, ctrl, rmask);                 \
        x = map_compose(p, x);                                  \
    }
    QMCP_STEP(0x111, 0xF)
    QMCP_STEP(0x112, 0xF)
    QMCP_STEP(0x114, 0xF)
    QMCP_STEP(0x118, 0xF)
    QMCP_STEP(0x142, 0xA)
    QMCP_STEP(0x143, 0xC)
#undef QMCP_STEP
    return x;
}
// min over lanes strictly above this lane (>= kInf for lane 63).  Row totals are read
// with v_readlane and merged with per-lane masks (all-ones = "row does not count"), so there
// is no divergent control flow.
__device__ __forceinline__ uint32_t wave_excl_suffix_min(uint32_t t) {
    uint32_t s = t;
    s = min(s, QMCP_DPP(kInf, s, 0x101, 0xF));  // row_shl:1
    s = min(s, QMCP_DPP(kInf, s, 0x102, 0xF));
    s = min(s, QMCP_DPP(kInf, s, 0x104, 0xF));
    s = min(s, QMCP_DPP(kInf, s, 0x108, 0xF));
    const uint32_t r1 = __builtin_amdgcn_readlane(s, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane(s, 32);
    const uint32_t r3 = __builtin_amdgcn_readlane(s, 48);
    const uint32_t row = (threadIdx.x & 63) >> 4;
    const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;  // rows that lie above this lane's row
    const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
    const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
    s = min(min(s, r1 | off1), min(r2 | off2, r3 | off3));  // inclusive suffix min
    return QMCP_DPP(kInf, s, 0x130, 0xF);                   // wave_shl:1 -> exclusive
}

template <int E>
struct SweepLoads { uint32_t x0[E], x1[E], x2[E]; };

// Unconditional loads with clamped addresses (boff has ltot + 1 entries, base + L is always
// in range): every lane issues the same number of loads, so the compiler can keep the next
// block's loads in flight behind a counted s_waitcnt instead of draining them.
template <int E>
__device__ __forceinline__ void sweep_load(const uint32_t* __restrict__ cb /* boff + base */,
                                           uint32_t a, uint32_t ell, uint32_t L, uint32_t lane,
                                           SweepLoads<E>& o) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t p = a + lane * E + r;
        o.x0[r] = cb[min(p, L)];
        o.x1[r] = cb[min(p + 1, L)];
        o.x2[r] = cb[min(p + ell + 1, L)];
    }
}

// DPP reads whose `old` operand is the operator's TRUE identity (INT32_MAX for signed min,
// 0xFFFFFFFF for unsigned min, 0 for add): LLVM's DPP combiner then folds the move into the
// operation (v_min_i32_dpp / v_min_u32_dpp / v_add_u32_dpp), one instruction per scan step.
#define QMCP_DPP_IMIN(v, ctrl, rmask) \
    __builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)(v), (ctrl), (rmask), 0xF, false)
#define QMCP_DPP_UMIN(v, ctrl, rmask) \
    (uint32_t) __builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), (ctrl), (rmask), 0xF, false)

// One block of `ell` positions starting at contig position a.  State carried between blocks:
// h (the previous block's h(j) = d(j) + ex(j + ell), slot-aligned) and d_last.
//
// Fast form.  Without intra-block jumps the block recurrence is
//     d'(i) = min( d'(i-1) + c(i), A(i) ),   A(i) = min_{j >= i} h(j)   (previous block)
// Unrolling it and using that the inclusive count prefix C(i) is non-decreasing gives
//     d'(i) = min( d_last + C(i),  C(i) + min_{j <= i} (h(j) - C(j)),  min_{j > i} h(j) )
// i.e. two INDEPENDENT wave scans over the previous block's h (a prefix-min and a suffix-min,
// one value each) plus a prefix sum of counts that does not depend on the chain at all.
// Then the block's own running minimum m(i) = min_{j<i} h'(j) is compared with d'(i): if it
// never undercuts, d' also satisfies the full recurrence (with intra-block jumps) position by
// position and is exact.  The function returns whether some lane saw an undercut; the caller
// then redoes the group with sweep_block_full.  On deep data the binding jumps come from the
// previous block, so that is rare; either way the result is the same distances.
//
// A lone wave pays two wait states between dependent DPP operations, so the scans are written
// pairwise interleaved (prefix-min with suffix-min; the verification scan with the NEXT block's
// count prefix, which is why the per-block terms are prepared one block ahead).
template <int E>
struct BlockTerms { uint32_t cnt[E], exj[E]; };

template <int E>
struct BlockPrep {       // everything about a block that does not depend on the chain
    uint32_t x0[E];      // bucket offset of each slot (for the store)
    uint32_t cnt[E];     // reads starting at the slot
    uint32_t exj[E];     // ex at the landing position of the slot's jump (kInf: none)
    uint32_t C[E];       // inclusive prefix of cnt over the block
    uint32_t before;     // sum of cnt over all lower lanes
};

template <int E>
__device__ __forceinline__ void block_terms(const SweepLoads<E>& cur, uint32_t a, uint32_t ell,
                                            uint32_t L, uint32_t M, uint32_t lane, BlockTerms<E>& t) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        const bool valid = i < ell && p < L;
        const uint32_t cov = cur.x2[r] - cur.x1[r];
        t.cnt[r] = valid ? cur.x1[r] - cur.x0[r] : 0u;
        t.exj[r] = (valid && p + ell < L) ? (cov > M ? cov - M : 0u) : kInf;
    }
}

// chain-independent part of a block, up to (not including) the wave scan of the lane sums
template <int E>
__device__ __forceinline__ uint32_t prep_local(const SweepLoads<E>& ld, uint32_t a, uint32_t ell,
                                               uint32_t L, uint32_t M, uint32_t lane,
                                               BlockPrep<E>& pr) {
    uint32_t lsum = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        const bool valid = i < ell && p < L;
        const uint32_t cov = ld.x2[r] - ld.x1[r];
        pr.x0[r] = ld.x0[r];
        pr.cnt[r] = valid ? ld.x1[r] - ld.x0[r] : 0u;
        pr.exj[r] = (valid && p + ell < L) ? (cov > M ? cov - M : 0u) : kInf;
        lsum += pr.cnt[r];
        pr.C[r] = lsum;
    }
    return lsum;
}
template <int E>
__device__ __forceinline__ void prep_finish(BlockPrep<E>& pr, uint32_t incl_lane_sums) {
    pr.before = QMCP_DPP(0u, incl_lane_sums, 0x138, 0xF);  // wave_shr:1, lane 0 gets 0
#pragma unroll
    for (int r = 0; r < E; ++r) pr.C[r] += pr.before;
}
template <int E>
__device__ __forceinline__ void prep_block(const SweepLoads<E>& ld, uint32_t a, uint32_t ell,
                                           uint32_t L, uint32_t M, uint32_t lane, BlockPrep<E>& pr) {
    prep_finish<E>(pr, wave_incl_scan_add(prep_local<E>(ld, a, ell, L, M, lane, pr)));
}

template <int E>
__device__ __forceinline__ void block_emit(const uint32_t (&x0)[E], const uint32_t (&cnt)[E],
                                           const uint32_t (&dn)[E], const uint32_t (&hn)[E],
                                           uint32_t d_in, uint32_t a, uint32_t trash, uint32_t ell,
                                           uint32_t L, uint32_t lane, uint32_t last_lane,
                                           uint32_t last_r, uint32_t (&h)[E], uint32_t& d_last,
                                           uint32_t* __restrict__ csel) {
    uint32_t prev = d_in, pick = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        // unconditional store: slots outside the contig write the spare entry selend[ltot]
        csel[(i < ell && p < L) ? p : trash] = x0[r] + (cnt[r] - (dn[r] - prev));
        prev = dn[r];
        if ((uint32_t)r == last_r) pick = dn[r];
        h[r] = hn[r];
    }
    d_last = __builtin_amdgcn_readlane(pick, last_lane);
}

// `pr` describes the block being solved; `nxt_ld` / `a_next` the block after it, whose terms
// are prepared here (into `nx`) in the shadow of this block's verification scan.
template <int E>
__device__ __forceinline__ bool sweep_block_fast(const BlockPrep<E>& pr, uint32_t a,
                                                 const SweepLoads<E>& nxt_ld, uint32_t a_next,
                                                 BlockPrep<E>& nx, uint32_t trash, uint32_t ell,
                                                 uint32_t L, uint32_t M, uint32_t lane,
                                                 uint32_t last_lane, uint32_t last_r,
                                                 uint32_t (&h)[E], uint32_t& d_last,
                                                 uint32_t* __restrict__ csel) {
    uint32_t dn[E], hn[E];
    // prefix-min of (h - C) and exclusive suffix-min of h over the previous block.
    // h < 2^31 and (h - C) >= -2^28, so the signed arithmetic cannot overflow.
    int32_t lp[E];
    int32_t pm = 0x7FFFFFFF;
    uint32_t sx[E];
    uint32_t sm = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        pm = min(pm, (int32_t)h[r] - (int32_t)pr.C[r]);
        lp[r] = pm;
    }
#pragma unroll
    for (int r = E - 1; r >= 0; --r) {
        sx[r] = sm;  // min over slots r' > r of this lane
        sm = min(sm, h[r]);
    }
    const uint32_t srun = sm;
    // interleaved: inclusive prefix-min of pm (row_shr ...) and inclusive suffix-min of sm (row_shl ...)
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x111, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x101, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x112, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x102, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x114, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x104, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x118, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x108, 0xF));
    const uint32_t r1 = __builtin_amdgcn_readlane(sm, 16);
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x142, 0xA));
    const uint32_t r2 = __builtin_amdgcn_readlane(sm, 32);
    const uint32_t r3 = __builtin_amdgcn_readlane(sm, 48);
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x143, 0xC));
    {
        const uint32_t row = lane >> 4;
        const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;  // rows that lie above this lane's row
        const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
        const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
        sm = min(min(sm, r1 | off1), min(r2 | off2, r3 | off3));  // inclusive suffix min
    }
    const int32_t pp = __builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)pm, 0x138, 0xF, 0xF, false);
    const uint32_t after = QMCP_DPP(0xFFFFFFFFu, sm, 0x130, 0xF);  // lanes above (all-ones: none)

    // d entering the lane = d' at the last slot of the lane below (C = before, prefix-min = pp,
    // suffix = everything from this lane's first slot on); for lane 0 it is d_last.
    uint32_t d_in = min(min(d_last + pr.before, (uint32_t)(pp + (int32_t)pr.before)), min(srun, after));
    d_in = lane == 0 ? d_last : d_in;
    uint32_t vm = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t viaP = (uint32_t)((int32_t)pr.C[r] + min(pp, lp[r]));
        dn[r] = min(min(d_last + pr.C[r], viaP), min(sx[r], after));
        hn[r] = dn[r] + pr.exj[r];
        vm = min(vm, hn[r]);
    }
    // interleaved: verification scan (inclusive prefix-min of the lanes' min h') and the next
    // block's count prefix (inclusive prefix-sum of its lane sums)
    uint32_t cs = prep_local<E>(nxt_ld, a_next, ell, L, M, lane, nx);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x111, 0xF));  cs += QMCP_DPP(0u, cs, 0x111, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x112, 0xF));  cs += QMCP_DPP(0u, cs, 0x112, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x114, 0xF));  cs += QMCP_DPP(0u, cs, 0x114, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x118, 0xF));  cs += QMCP_DPP(0u, cs, 0x118, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x142, 0xA));  cs += QMCP_DPP(0u, cs, 0x142, 0xA);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x143, 0xC));  cs += QMCP_DPP(0u, cs, 0x143, 0xC);
    prep_finish<E>(nx, cs);
    // m entering this lane = min of h' over all lower lanes
    uint32_t run = QMCP_DPP(0xFFFFFFFFu, vm, 0x138, 0xF);
    bool undercut = false;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        undercut |= run < dn[r];
        run = min(run, hn[r]);
    }
    block_emit<E>(pr.x0, pr.cnt, dn, hn, d_in, a, trash, ell, L, lane, last_lane, last_r, h, d_last, csel);
    return __any(undercut);
}

// General form of the same block: maps carrying (d, m) -- see Map4 above.
template <int E>
__device__ __forceinline__ void sweep_block_full(const SweepLoads<E>& cur, uint32_t a,
                                                 uint32_t trash, uint32_t ell, uint32_t L,
                                                 uint32_t M, uint32_t lane, uint32_t last_lane,
                                                 uint32_t last_r, uint32_t (&h)[E],
                                                 uint32_t& d_last, uint32_t* __restrict__ csel) {
    BlockTerms<E> t;
    block_terms<E>(cur, a, ell, L, M, lane, t);
    uint32_t sufA[E], dn[E], hn[E];
    {
        uint32_t srun = kInf;
#pragma unroll
        for (int r = E - 1; r >= 0; --r) { srun = min(srun, h[r]); sufA[r] = srun; }
        const uint32_t after = min(wave_excl_suffix_min(srun), kInf);
#pragma unroll
        for (int r = 0; r < E; ++r) sufA[r] = min(sufA[r], after);  // min_{j >= i} h(j)
    }
    Map4 acc = map_identity();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        Map4 e;
        e.a = t.cnt[r];
        e.b = t.cnt[r] + t.exj[r];
        e.u = sufA[r];
        e.v = sufA[r] + t.exj[r];
        acc = map_compose(acc, e);
    }
    Map4 inc = wave_incl_scan_map(acc);
    Map4 pre;  // composition of all lower lanes (identity for lane 0)
    pre.a = QMCP_DPP(0u, inc.a, 0x138, 0xF);
    pre.b = QMCP_DPP(kInf, inc.b, 0x138, 0xF);
    pre.u = QMCP_DPP(kInf, inc.u, 0x138, 0xF);
    pre.v = QMCP_DPP(kInf, inc.v, 0x138, 0xF);
    // state entering this lane: (d, m) = pre applied to (d_last, +inf)
    const uint32_t d_in = min(d_last + pre.a, pre.u);
    uint32_t dd = d_in;
    uint32_t m = min(d_last + pre.b, pre.v);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        dd = min(min(dd + t.cnt[r], m), sufA[r]);
        dn[r] = dd;
        hn[r] = dd + t.exj[r];
        m = min(m, hn[r]);
    }
    block_emit<E>(cur.x0, t.cnt, dn, hn, d_in, a, trash, ell, L, lane, last_lane, last_r, h, d_last, csel);
}

// blocks [b_begin, b_end) in the general form, loads of block b+1 in flight under block b
template <int E>
__device__ __forceinline__ void sweep_full_run(const uint32_t* __restrict__ cb, uint32_t b_begin,
                                               uint32_t b_end, uint32_t trash, uint32_t ell,
                                               uint32_t L, uint32_t M, uint32_t lane,
                                               uint32_t last_lane, uint32_t last_r,
                                               uint32_t (&h)[E], uint32_t& d_last,
                                               uint32_t* __restrict__ csel) {
    SweepLoads<E> T0, T1;
    sweep_load<E>(cb, b_begin * ell, ell, L, lane, T0);
    for (uint32_t b = b_begin; b < b_end; b += 2) {
        sweep_load<E>(cb, (b + 1) * ell, ell, L, lane, T1);
        sweep_block_full<E>(T0, b * ell, trash, ell, L, M, lane, last_lane, last_r, h, d_last, csel);
        sweep_load<E>(cb, (b + 2) * ell, ell, L, lane, T0);
        if (b + 1 < b_end)
            sweep_block_full<E>(T1, (b + 1) * ell, trash, ell, L, M, lane, last_lane, last_r, h, d_last, csel);
    }
}

template <int E>
__global__ __launch_bounds__(64) void k_sweep_uniform(const uint32_t* __restrict__ boff,
                                                      const uint64_t* __restrict__ contig_pos_off,
                                                      uint32_t ell, uint32_t M, uint32_t ltot,
                                                      uint32_t* __restrict__ selend,
                                                      uint32_t* __restrict__ iter_stats) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    if (L == 0) return;
    const uint32_t n_blocks = (L + ell - 1) / ell;
    // this wave is a serial dependency chain that may share its SIMD with streaming kernels:
    // win the issue arbitration
    __builtin_amdgcn_s_setprio(3);

    uint32_t h[E];  // previous block's h(j) = d(j) + ex(j + ell), aligned with this block's slots
    // virtual block -1: d == 0 and the jump from j = i - ell lands on p = i
    {
        const uint32_t b0 = boff[base];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t i = lane * E + r;
            const uint32_t cov = boff[base + min(i + 1, L)] - b0;
            h[r] = (i < ell && i < L) ? (cov > M ? cov - M : 0u) : kInf;
        }
    }
    uint32_t d_last = 0;
    uint32_t n_full = 0;  // blocks that needed the 4-component form
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;

    // Four register sets in rotation: the loads of block b+3 are issued before block b is
    // computed, so three blocks of work (~1.5 us) cover the load latency even when the radix
    // passes on the other stream keep HBM busy.  No register copies between iterations; loads
    // past the contig end clamp to a valid address.
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    const uint32_t trash = ltot - base;  // csel[trash] == selend[ltot], the spare entry
    // Steady state: groups of four blocks in the fast form, four register sets in rotation
    // (loads of block b+3 are issued before block b is computed), no branch around any load
    // or store so the compiler keeps counted waits.  A group in which some block reports an
    // undercut leaves the loop, is redone from the saved state in the general form, and the
    // pipeline restarts behind it.
    const uint32_t n_groups = n_blocks / 4;
    uint32_t g = 0;
    uint32_t penalty = 0;  // groups to run in the general form after a failed fast attempt
    while (g < n_groups) {
        if (penalty > 0) {
            // sparse / low-coverage stretch: the fast form keeps failing here, do not try it
            const uint32_t run = min(penalty, n_groups - g);
            sweep_full_run<E>(cb, g * 4, (g + run) * 4, trash, ell, L, M, lane, last_lane, last_r, h,
                              d_last, csel);
            n_full += run * 4;
            g += run;
            if (g >= n_groups) break;
        }
        // pipeline start: loads for the group's first four blocks, terms of its first block
        SweepLoads<E> S0, S1, S2, S3;
        BlockPrep<E> PA, PB;
        sweep_load<E>(cb, g * 4 * ell, ell, L, lane, S0);
        sweep_load<E>(cb, (g * 4 + 1) * ell, ell, L, lane, S1);
        sweep_load<E>(cb, (g * 4 + 2) * ell, ell, L, lane, S2);
        sweep_load<E>(cb, (g * 4 + 3) * ell, ell, L, lane, S3);
        prep_block<E>(S0, g * 4 * ell, ell, L, M, lane, PA);
        uint32_t h_save[E];
        uint32_t d_save = d_last;
        bool bad = false;
        uint32_t good = 0;
        for (; g < n_groups; ++g) {
            const uint32_t a = g * 4 * ell;
#pragma unroll
            for (int r = 0; r < E; ++r) h_save[r] = h[r];
            d_save = d_last;
            // block k solves with terms prepared during block k-1 and prepares block k+1 from
            // loads issued three blocks earlier; S_k is re-loaded for block k+4 once consumed
#define QMCP_FAST(PR, pos, LD_NEXT, PR_NEXT)                                                        \
    sweep_block_fast<E>(PR, pos, LD_NEXT, (pos) + ell, PR_NEXT, trash, ell, L, M, lane, last_lane,  \
                        last_r, h, d_last, csel)
            sweep_load<E>(cb, a + 4 * ell, ell, L, lane, S0);
            bad = QMCP_FAST(PA, a, S1, PB);
            sweep_load<E>(cb, a + 5 * ell, ell, L, lane, S1);
            bad |= QMCP_FAST(PB, a + ell, S2, PA);
            sweep_load<E>(cb, a + 6 * ell, ell, L, lane, S2);
            bad |= QMCP_FAST(PA, a + 2 * ell, S3, PB);
            sweep_load<E>(cb, a + 7 * ell, ell, L, lane, S3);
            bad |= QMCP_FAST(PB, a + 3 * ell, S0, PA);
#undef QMCP_FAST
            if (bad) break;
            ++good;
        }
        if (bad) {
            // redo group g in the general form (its stores overwrite the speculative ones);
            // back off from the fast form: 1, 3, 7, ... 63 groups, reset by a fast success
#pragma unroll
            for (int r = 0; r < E; ++r) h[r] = h_save[r];
            d_last = d_save;
            sweep_full_run<E>(cb, g * 4, g * 4 + 4, trash, ell, L, M, lane, last_lane, last_r, h, d_last, csel);
            n_full += 4;
            ++g;
            penalty = good > 0 ? 1u : min(2 * penalty + 1, 63u);
        } else {
            penalty = 0;
        }
    }
    // tail: at most three blocks, general form
    if (n_groups * 4 < n_blocks)
        sweep_full_run<E>(cb, n_groups * 4, n_blocks, trash, ell, L, M, lane, last_lane, last_r, h, d_last, csel);
    if (iter_stats && lane == 0) {
        atomicAdd(&iter_stats[0], n_full);
        atomicAdd(&iter_stats[1], n_blocks);
    }
}

// ------------------------------------------------------------------ uniform sweep, three waves
// Same algorithm as k_sweep_uniform, with the work of a group of four blocks split over three
// waves of one workgroup (one workgroup per contig) that advance in lockstep, one
// __syncthreads() per group, all hand-offs through LDS (no spinning, uniform control flow):
//   wave 0  PREP    loads bucket offsets and prepares the chain-independent terms of group t
//   wave 1  CHAIN   solves group t-1 in the fast form: only the two min-scans and the combine
//                   remain on the serial path
//   wave 2  CHECK   verifies group t-2 (running-minimum test) and stores its results
// If CHECK finds an undercut in group f, every wave sees the flag after the barrier; the chain
// wave restores its state from the start of group f, redoes f (and, while the fast form keeps
// failing, a growing run of following groups) alone in the general form, and the pipeline
// restarts behind it.  Nothing of a failed group is stored by CHECK (it verifies all four blocks
// before storing), and the chain wave is at most one group ahead of it, so no speculative value
// ever reaches memory.
// LDS: three group slots x four blocks x (6E + 2) words x 64 lanes, word-major (conflict-free).
// A "row" is one block's bucket offsets in slot layout: X[r] = cb[min(a + lane*E + r, L)].
// Each lane's E slots are adjacent in memory, so a row is ONE vector load per lane (dwordx2/x3/
// x4) instead of E scalar ones; rows that poke past the end of the table fall back to clamped
// scalar loads.  The three views a block needs are then built in registers:
//   x0 = X(b),  x1 = X(b) shifted by one slot,  x2 = X(b+1) shifted by one slot,
// where "shifted" takes slot r+1 of the same lane, the next lane's slot 0 (DPP wave_shl:1) for
// the lane's last slot, and the following row's very first entry for the block's last slot.
template <int E> struct RowVec;
template <> struct RowVec<1> { typedef uint32_t type; };
template <> struct RowVec<2> { typedef uint32_t type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct RowVec<3> { typedef uint32_t type __attribute__((ext_vector_type(3), aligned(4))); };
template <> struct RowVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };

// Split in two so that the select fix-up (which needs the data) can sit a whole stage after the
// load was issued: row_issue starts the load, row_finish turns it into slot values.
template <int E>
struct RowRaw { typename RowVec<E>::type v; uint32_t sh; };

template <int E>
__device__ __forceinline__ void row_issue(const uint32_t* __restrict__ cb, uint32_t a, uint32_t L,
                                          uint32_t lane, RowRaw<E>& raw) {
    // branch-free (a load inside a divergent branch makes the compiler drain vmcnt at the join):
    // the vector is read at a base clamped so that it ends at cb[L] at the latest
    const uint32_t p = a + lane * E;
    const uint32_t pe = min(p, L + 1 - E);  // L >= ell > E always
    raw.sh = p - pe;
    raw.v = *reinterpret_cast<const typename RowVec<E>::type*>(cb + pe);
}
template <int E>
__device__ __forceinline__ void row_finish(const RowRaw<E>& raw, uint32_t (&X)[E]) {
    // lanes whose base moved pick their entries with selects: X[r] = v[min(sh + r, E-1)]
    if constexpr (E == 1) {
        X[0] = raw.v;
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            uint32_t x = raw.v[E - 1];
#pragma unroll
            for (int q = E - 2; q >= r; --q) x = (raw.sh + r <= (uint32_t)q) ? raw.v[q] : x;
            X[r] = x;
        }
    }
}
template <int E>
__device__ __forceinline__ void row_load(const uint32_t* __restrict__ cb, uint32_t a, uint32_t L,
                                         uint32_t lane, uint32_t (&X)[E]) {
    RowRaw<E> raw;
    row_issue<E>(cb, a, L, lane, raw);
    row_finish<E>(raw, X);
}

template <int E>
__device__ __forceinline__ void rows_to_loads(const uint32_t (&Xa)[E], const uint32_t (&Xb)[E],
                                              const uint32_t (&Xc)[E], uint32_t lane,
                                              uint32_t last_lane, uint32_t last_r, SweepLoads<E>& o) {
    const uint32_t nb0 = QMCP_DPP(0u, Xa[0], 0x130, 0xF);  // next lane's first slot (wave_shl:1)
    const uint32_t nb1 = QMCP_DPP(0u, Xb[0], 0x130, 0xF);
    const uint32_t tail1 = __builtin_amdgcn_readlane(Xb[0], 0);  // cb[a + ell]
    const uint32_t tail2 = __builtin_amdgcn_readlane(Xc[0], 0);  // cb[a + 2 ell]
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const bool is_last = lane == last_lane && (uint32_t)r == last_r;
        o.x0[r] = Xa[r];
        o.x1[r] = is_last ? tail1 : (r + 1 < E ? Xa[r + 1 < E ? r + 1 : r] : nb0);
        o.x2[r] = is_last ? tail2 : (r + 1 < E ? Xb[r + 1 < E ? r + 1 : r] : nb1);
    }
}

template <int E>
struct MwLayout {
    static constexpr int kG = 8;            // blocks per group (one pipeline stage)
    static constexpr int kC = 0;            // [E]  inclusive count prefix       PREP -> CHAIN
    static constexpr int kEx = E;           // [E]  ex at the jump landing       PREP -> CHAIN, CHECK
    static constexpr int kX0 = 2 * E;       // [E]  bucket offsets               PREP -> CHECK
    static constexpr int kCnt = 3 * E;      // [E]  counts                       PREP -> CHECK
    static constexpr int kDn = 4 * E;       // [E]  distances                    CHAIN -> CHECK
    static constexpr int kH0 = 5 * E;       // [E]  block 0 only: h entering the group (rollback state)
    static constexpr int kDin = 6 * E;      // [1]  block 0 only: d entering the group   CHAIN -> CHECK
    static constexpr int kWords = 6 * E + 1;
    static constexpr int kSlots = 3;
    static constexpr size_t kBytes = (size_t)kSlots * kG * kWords * 64 * sizeof(uint32_t) + 64;
};

// Pipelined form of the sweep: one workgroup of seven waves per contig.
//   waves 0,1,2,4  PREP   two blocks of the group each: bucket-offset rows -> counts, prefix, ex
//   wave  3        CHAIN  the serial part: two min-scans + combine per block (alone on its SIMD:
//                         waves are placed round-robin on the four SIMDs)
//   waves 5,6      CHECK  four blocks of the group each: undercut check, selected counts, stores
// A lone wave issues an instruction every 5-8 cycles (lab/issue_lab.hip), so everything that does
// not depend on the chain is kept off the chain wave.  Blocks go in groups of kG; stage t has PREP
// on group g0+t, CHAIN on g0+t-1, CHECK on g0+t-2, one barrier per stage, all hand-offs through
// three LDS slots.  Results reach memory only after the verdict: every block before the first
// failed one of a group is exact and is stored.  On a failed check every wave but CHAIN leaves the
// pipeline right after the barrier; CHAIN, which reads the flag without waiting for it, notices
// at the end of the stage it has already started, rebuilds the state entering the failed block
// from LDS (it publishes h and d at every group entry; inside a group the state is the previous
// block's distances + ex), redoes the rest of that group in the general form, and the pipeline
// restarts behind it -- warm: what PREP made for the next two groups is still in LDS.  Only when
// the first group of a run fails again does CHAIN back off to runs of 1, 3, 7 ... 63 groups in the
// general form (sparse data, where the fast form rarely holds).
template <int E>
__global__ __launch_bounds__(448) void k_sweep_uniform_mw(const uint32_t* __restrict__ boff,
                                                          const uint64_t* __restrict__ contig_pos_off,
                                                          uint32_t ell, uint32_t M, uint32_t ltot,
                                                          uint32_t* __restrict__ selend,
                                                          uint32_t* __restrict__ iter_stats) {
    using Ly = MwLayout<E>;
    constexpr int kG = Ly::kG;
    extern __shared__ uint32_t s_mw[];
    uint32_t* s_flag = s_mw + (size_t)Ly::kSlots * kG * Ly::kWords * 64;
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform in the compiler's eyes too: role branches are scalar branches, so the PREP
    // waves can run their own copy of the stage loop
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t role = wv == 3 ? 1u : (wv >= 5 ? 2u : 0u);  // 0 PREP, 1 CHAIN, 2 CHECK
    const uint32_t pblk = 2 * (wv == 4 ? 3u : wv);            // PREP: first of its two blocks
    const uint32_t cblk = 4 * (wv - 5);                       // CHECK: first of its four blocks
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    if (L == 0) return;
    const uint32_t n_blocks = (L + ell - 1) / ell;
    const uint32_t n_groups = n_blocks / kG;
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    const uint32_t trash = ltot - base;  // E spare words behind the table absorb idle lanes' stores
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;
    // the waves form one serial pipeline: each must win issue arbitration against the
    // streaming kernels that may share their SIMDs
    __builtin_amdgcn_s_setprio(3);
    // s_flag[0]: first block of the checked group whose check failed (kNoFail: none)
    constexpr uint32_t kNoFail = 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_flag[0] = kNoFail;

    // chain state (meaningful in the CHAIN wave only)
    uint32_t h[E];
    {
        const uint32_t b0 = cb[0];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t i = lane * E + r;
            const uint32_t cov = cb[min(i + 1, L)] - b0;
            h[r] = (i < ell && i < L) ? (cov > M ? cov - M : 0u) : kInf;
        }
    }
    uint32_t d_last = 0;
    uint32_t n_full = 0;
    uint32_t penalty = 0;
    uint32_t g0 = 0;
    bool warm = false;  // the PREP data of groups g0 and g0+1 survive from the failed run
#ifdef QMCP_MW_STAMP
    unsigned long long stamp_work = 0, stamp_wait = 0, stamp_fail = 0, stamp_fails = 0, stamp_post = 0, stamp_iters = 0;
    unsigned long long stamp_prev = 0;
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();

    // MW_AT: word w of block k of a slot; `slot` is a per-stage base pointer, so k and w fold into
    // the instruction's immediate offset
#define MW_SLOT0(idx) (s_mw + (size_t)(idx) * kG * Ly::kWords * 64)
#define MW_SLOT(idx) (MW_SLOT0(idx) + lane)
#define MW_AT(slot, k, w) (slot)[((k) * Ly::kWords + (w)) * 64]

    while (g0 < n_groups) {
        if (penalty > 0) {
            warm = false;
            const uint32_t run = min(penalty, n_groups - g0);
            if (role == 1) {
                sweep_full_run<E>(cb, g0 * kG, (g0 + run) * kG, trash, ell, L, M, lane, last_lane, last_r,
                                  h, d_last, csel);
                n_full += run * kG;
            }
            g0 += run;
            if (g0 >= n_groups) break;
        }
        const uint32_t n_left = n_groups - g0;
        uint32_t failed = 0xFFFFFFFFu;  // group whose check failed
        uint32_t sel[4][E];   // CHECK: results of the group checked in this stage, stored after the verdict
        uint32_t flag_seen = kNoFail;  // CHAIN: the flag as read after the previous stage's barrier
        // A warm run starts one stage in: the chain can take group g0 at once, and PREP resumes
        // with group g0+2.
        const uint32_t prep_from = warm ? 2u : 0u;
        const uint32_t t_begin = warm ? 1u : 0u;
#ifdef QMCP_MW_STAMP
#define MW_STAGE_BEGIN()                                                  \
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();       \
    if (stamp_prev != 0) stamp_post += stamp0 - stamp_prev;               \
    stamp_iters += 1;
#define MW_STAGE_BARRIER()                                                \
    const unsigned long long stamp1 = __builtin_amdgcn_s_memtime();       \
    __syncthreads();                                                      \
    const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();       \
    stamp_work += stamp1 - stamp0;                                        \
    stamp_wait += stamp2 - stamp1;                                        \
    stamp_prev = stamp2;
#else
#define MW_STAGE_BEGIN()
#define MW_STAGE_BARRIER() __syncthreads();
#endif
        if (role == 0) {
            // PREP waves run their own copy of the stage loop (same barriers, same exits), unrolled
            // kD times: a wave needs four rows for its two blocks (blocks kG*g+pblk .. +3), and loads
            // them kD stages ahead into kD register sets that take turns in a FIXED order in the
            // instruction stream -- so the compiler waits for exactly the oldest set (a counted
            // s_waitcnt) and a row has kD whole stages to land.  Memory latency beside a
            // bandwidth-bound kernel is several microseconds; one stage is about one.
            constexpr uint32_t kD = 3;
            RowRaw<E> R0[4], R1[4], R2[4];
            auto issue_rows = [&](RowRaw<E> (&buf)[4], uint32_t g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) row_issue<E>(cb, (g * kG + pblk + k) * ell, L, lane, buf[k]);
            };
            issue_rows(R0, g0 + t_begin);
            issue_rows(R1, g0 + t_begin + 1);
            issue_rows(R2, g0 + t_begin + 2);
            // one stage with register set `buf` (which holds the rows of group g0+t); false: run over
            auto pstage = [&](RowRaw<E> (&buf)[4], uint32_t t) -> bool {
                MW_STAGE_BEGIN()
                const uint32_t g = g0 + t;
                if (t >= prep_from && t < n_left) {
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots) + pblk * Ly::kWords * 64;
                    uint32_t Wr[4][E];
#pragma unroll
                    for (int k = 0; k < 4; ++k) row_finish<E>(buf[k], Wr[k]);
                    __builtin_amdgcn_sched_barrier(0);  // rows consumed before the set is reloaded
                    issue_rows(buf, g + kD);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        SweepLoads<E> ldk;
                        rows_to_loads<E>(Wr[kk], Wr[kk + 1], Wr[kk + 2], lane, last_lane, last_r, ldk);
                        BlockPrep<E> pr;
                        prep_block<E>(ldk, (g * kG + pblk + kk) * ell, ell, L, M, lane, pr);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            MW_AT(slot, kk, Ly::kC + r) = pr.C[r];
                            MW_AT(slot, kk, Ly::kEx + r) = pr.exj[r];
                            MW_AT(slot, kk, Ly::kX0 + r) = pr.x0[r];
                            MW_AT(slot, kk, Ly::kCnt + r) = pr.cnt[r];
                        }
                    }
                } else if (t < n_left) {
                    issue_rows(buf, g + kD);  // a warm run's first stage: the group exists already, the set moves on
                }
                MW_STAGE_BARRIER()
                const uint32_t bad = t >= 2 ? s_flag[0] : kNoFail;
                if (bad != kNoFail) { failed = g0 + t - 2; return false; }
                return true;
            };
            for (uint32_t t = t_begin;; t += kD) {
                if (t >= n_left + 2 || !pstage(R0, t)) break;
                if (t + 1 >= n_left + 2 || !pstage(R1, t + 1)) break;
                if (t + 2 >= n_left + 2 || !pstage(R2, t + 2)) break;
            }
        } else
        for (uint32_t t = t_begin; t < n_left + 2; ++t) {
            MW_STAGE_BEGIN()
            if (role == 1) {
                if (t >= 1 && t <= n_left) {
                    const uint32_t g = g0 + t - 1;
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MW_SLOT0(g % Ly::kSlots);
                    // state entering the group: d for the CHECK waves (inside a group they read the
                    // previous block's last distance themselves), h and d for a rollback
                    MW_AT(slot, 0, Ly::kDin) = d_last;
#pragma unroll
                    for (int r = 0; r < E; ++r) MW_AT(slot, 0, Ly::kH0 + r) = h[r];
                    // terms of block k+1 are read from LDS before block k's scans start, so their
                    // latency hides under the scans (the compiler will not hoist LDS reads above the
                    // previous block's LDS writes by itself)
                    uint32_t Cn[E], exn[E];
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        Cn[r] = MW_AT(slot, 0, Ly::kC + r);
                        exn[r] = MW_AT(slot, 0, Ly::kEx + r);
                    }
#pragma unroll
                    for (int k = 0; k < kG; ++k) {
                        uint32_t C[E], exj[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) { C[r] = Cn[r]; exj[r] = exn[r]; }
                        if (k + 1 < kG) {
#pragma unroll
                            for (int r = 0; r < E; ++r) {
                                Cn[r] = MW_AT(slot, k + 1, Ly::kC + r);
                                exn[r] = MW_AT(slot, k + 1, Ly::kEx + r);
                            }
                        }
                        // the two chain scans, interleaved (see sweep_block_fast)
                        int32_t lp[E];
                        int32_t pm = 0x7FFFFFFF;
                        uint32_t sx[E];
                        uint32_t sm = 0xFFFFFFFFu;
#pragma unroll
                        for (int r = 0; r < E; ++r) { pm = min(pm, (int32_t)h[r] - (int32_t)C[r]); lp[r] = pm; }
#pragma unroll
                        for (int r = E - 1; r >= 0; --r) { sx[r] = sm; sm = min(sm, h[r]); }
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x111, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x101, 0xF));
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x112, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x102, 0xF));
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x114, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x104, 0xF));
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x118, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x108, 0xF));
                        const uint32_t r1 = __builtin_amdgcn_readlane(sm, 16);
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x142, 0xA));
                        const uint32_t r2 = __builtin_amdgcn_readlane(sm, 32);
                        const uint32_t r3 = __builtin_amdgcn_readlane(sm, 48);
                        pm = min(pm, QMCP_DPP_IMIN(pm, 0x143, 0xC));
                        {
                            const uint32_t row = lane >> 4;
                            const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
                            sm = min(min(sm, r1 | off1), min(r2 | off2, r3 | off3));
                        }
                        // min(d_last + C, C + min(pp, lp)) = C + min(d_last, pp, lp): d_last joins the prefix.
                        // For the lane's last slot min(pp, lp) is the inclusive scan value itself.
                        const int32_t dl = (int32_t)d_last;
                        const int32_t pp = min(__builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)pm, 0x138, 0xF, 0xF, false), dl);
                        const int32_t pin = min(pm, dl);
                        const uint32_t after = QMCP_DPP(0xFFFFFFFFu, sm, 0x130, 0xF);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            const int32_t pre = r == E - 1 ? pin : min(pp, lp[r]);
                            const uint32_t viaP = (uint32_t)((int32_t)C[r] + pre);
                            const uint32_t dnr = r == E - 1 ? min(viaP, after) : min(viaP, min(sx[r], after));
                            MW_AT(slot, k, Ly::kDn + r) = dnr;
                            h[r] = dnr + exj[r];
                        }
                        // the block's last distance, read back as a broadcast: cheaper for a lone wave than
                        // selecting the slot and v_readlane, and its latency hides under the next scans
                        d_last = slot0[((k * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                    }
                }
                // the flag raised for group g0+t-3 (read after the previous barrier) is looked at only
                // now, so the chain never waits for it
                if (flag_seen != kNoFail) { failed = g0 + t - 3; break; }
            } else {
                if (t >= 2) {
                    const uint32_t g = g0 + t - 2;
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MW_SLOT0(g % Ly::kSlots);
                    uint32_t first_bad = kNoFail;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const uint32_t k = cblk + kk;
                        bool undercut = false;
                        uint32_t dn[E], hn[E];
                        uint32_t vm = 0xFFFFFFFFu;
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            dn[r] = MW_AT(slot, k, Ly::kDn + r);
                            hn[r] = dn[r] + MW_AT(slot, k, Ly::kEx + r);
                            vm = min(vm, hn[r]);
                        }
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x111, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x112, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x114, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x118, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x142, 0xA));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x143, 0xC));
                        uint32_t run = QMCP_DPP(0xFFFFFFFFu, vm, 0x138, 0xF);
                        // d entering the lane: the lane below's last distance; lane 0: d entering the block
                        const uint32_t d_blk = k == 0 ? MW_AT(slot, 0, Ly::kDin)
                                                      : slot0[(((k - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                        uint32_t prev = QMCP_DPP(0u, dn[E - 1], 0x138, 0xF);
                        prev = lane == 0 ? d_blk : prev;
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            undercut |= run < dn[r];
                            run = min(run, hn[r]);
                            sel[kk][r] = MW_AT(slot, k, Ly::kX0 + r) + (MW_AT(slot, k, Ly::kCnt + r) - (dn[r] - prev));
                            prev = dn[r];
                        }
                        if (__any(undercut)) first_bad = min(first_bad, k);
                    }
                    if (first_bad != kNoFail && lane == 0) atomicMin(&s_flag[0], first_bad);
                }
            }
            MW_STAGE_BARRIER()  // (diagnostic builds: stamps around it, summed in registers)
            if (role == 1) {
                flag_seen = t >= 2 ? s_flag[0] : kNoFail;  // not waited for here
            } else {
                const uint32_t bad = t >= 2 ? s_flag[0] : kNoFail;
                // only now do results reach memory: every block before the first failed one is exact
                // (its check passed and so did those of all blocks before it)
                if (role == 2 && t >= 2) {
                    const uint32_t g = g0 + t - 2;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (cblk + kk < bad) {
                            // one contiguous vector store per lane (its E slots are adjacent positions);
                            // lanes without E valid slots write the spare words behind the table
                            const uint32_t blk_first = (g * kG + cblk + kk) * ell;
                            const uint32_t p0 = blk_first + lane * E;
                            const bool full = lane * E + E <= ell && p0 + E <= L;
                            if constexpr (E == 1) {
                                csel[full ? p0 : trash] = sel[kk][0];
                            } else {
                                typedef typename RowVec<E>::type V;
                                V v;
#pragma unroll
                                for (int r = 0; r < E; ++r) v[r] = sel[kk][r];
                                *reinterpret_cast<V*>(csel + (full ? p0 : trash)) = v;
                                // a lane with only some valid slots exists only if E does not divide the
                                // span or the block is cut by the contig's end (uniform test)
                                if (ell % E != 0 || blk_first + ell > L) {
#pragma unroll
                                    for (int r = 0; r < E; ++r) {
                                        const uint32_t i = lane * E + r;
                                        if (!full && i < ell && blk_first + i < L) csel[blk_first + i] = sel[kk][r];
                                    }
                                }
                            }
                        }
                    }
                }
                if (bad != kNoFail) { failed = g0 + t - 2; break; }
            }
        }
#ifdef QMCP_MW_STAMP
        const unsigned long long stamp_f0 = __builtin_amdgcn_s_memtime();
        stamp_prev = 0;
#endif
        // CHAIN leaves the loop one stage late (or at its end): the last flag it read is still unseen
        if (role == 1 && failed == 0xFFFFFFFFu && flag_seen != kNoFail) failed = g0 + n_left - 1;
        // every wave must agree on whether the pipeline failed: the flag itself says so
        __syncthreads();
        const uint32_t bad_blk = s_flag[0];
        if (bad_blk == kNoFail) { g0 = n_groups; break; }
        __syncthreads();  // everyone has read the flag
        if (threadIdx.x == 0) s_flag[0] = kNoFail;
        if (role == 1) {
            // state on entering the failed block: published at the group's entry, or rebuilt from the
            // (exact) distances of the block before it
            uint32_t* const slot = MW_SLOT(failed % Ly::kSlots);
            if (bad_blk == 0) {
#pragma unroll
                for (int r = 0; r < E; ++r) h[r] = MW_AT(slot, 0, Ly::kH0 + r);
                d_last = MW_AT(slot, 0, Ly::kDin);
            } else {
                uint32_t* const blk = slot + (bad_blk - 1) * Ly::kWords * 64;
#pragma unroll
                for (int r = 0; r < E; ++r) h[r] = MW_AT(blk, 0, Ly::kDn + r) + MW_AT(blk, 0, Ly::kEx + r);
                d_last = MW_SLOT0(failed % Ly::kSlots)[(((bad_blk - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
            }
            sweep_full_run<E>(cb, failed * kG + bad_blk, failed * kG + kG, trash, ell, L, M, lane, last_lane,
                              last_r, h, d_last, csel);
            n_full += kG - bad_blk;
        }
        // `failed` is known to every wave: CHAIN derived the same group one stage later.
        // An isolated failure (the usual case on deep data) costs only the rest of the failed group:
        // what PREP made for the two groups after it is still in LDS, so the next run starts warm.
        // A failure of the very first group of a run means the fast form keeps failing here: back off.
        penalty = failed > g0 ? 0u : min(2 * penalty + 1, 63u);
        warm = penalty == 0;
        g0 = failed + 1;
        __syncthreads();
#ifdef QMCP_MW_STAMP
        stamp_fail += __builtin_amdgcn_s_memtime() - stamp_f0;
        stamp_fails += 1;
#endif
    }
#undef MW_AT
#undef MW_SLOT
#undef MW_SLOT0
#undef MW_STAGE_BEGIN
#undef MW_STAGE_BARRIER
#ifdef QMCP_MW_STAMP
    if (lane == 0 && iter_stats) {
        atomicAdd(&iter_stats[4 + 2 * wv], (uint32_t)(stamp_work >> 4));
        atomicAdd(&iter_stats[5 + 2 * wv], (uint32_t)(stamp_wait >> 4));
        if (wv == 3) {
            atomicAdd(&iter_stats[20], (uint32_t)(stamp_fail >> 4));
            atomicAdd(&iter_stats[21], (uint32_t)stamp_fails);
            atomicAdd(&iter_stats[22], (uint32_t)((__builtin_amdgcn_s_memtime() - stamp_begin) >> 4));
            atomicAdd(&iter_stats[23], (uint32_t)(stamp_post >> 4));
            atomicAdd(&iter_stats[24], (uint32_t)stamp_iters);
        }
    }
#endif
    if (role == 1) {
        if (n_groups * kG < n_blocks)
            sweep_full_run<E>(cb, n_groups * kG, n_blocks, trash, ell, L, M, lane, last_lane, last_r, h, d_last,
                              csel);
        if (iter_stats && lane == 0) {
            atomicAdd(&iter_stats[0], n_full);
            atomicAdd(&iter_stats[1], n_blocks);
        }
    }
}

// ------------------------------------------------------------------ pipelined general form
// The same seven-wave pipeline for data where the fast form rarely holds (mean coverage within a few
// multiples of M): every block is solved in the GENERAL form, which needs no check and no rollback.
// A block is an inclusive scan of the maps (a, b, u, v) above; the (a, b) half depends only on the
// counts and ex -- not on the chain -- so the PREP waves run that half of the scan ahead of time and
// hand the chain wave, for each of the six scan steps, the (a, b) the current lane holds before the
// step (the "right operand" of the composition).  The chain wave is left with the (u, v) half:
//     u' = min(uL + a_s, vL, u),   v' = min(uL + b_s, vL, v)          (uL, vL: DPP reads)
// six instructions per step, plus the suffix-min of the previous block's h that feeds u and v.
// About 90 instructions per block against 50 for the fast form -- and against ~200 for the
// single-wave general block the fast kernel falls back to.
template <int E>
struct MgLayout {
    static constexpr int kG = E <= 3 ? 8 : 4; // blocks per group (LDS: 3 slots x kG x kWords x 256 B)
    static constexpr int kA = 0;              // [6] a before each scan step          PREP -> CHAIN
    static constexpr int kB = 6;              // [6] b before each scan step          PREP -> CHAIN
    static constexpr int kPa = 12;            // [1] a of all lower lanes (0: none)   PREP -> CHAIN
    static constexpr int kPb = 13;            // [1] b of all lower lanes (inf: none) PREP -> CHAIN
    static constexpr int kCnt = 14;           // [E] counts                           PREP -> CHAIN, CHECK
    static constexpr int kEx = 14 + E;        // [E] ex at the jump landing           PREP -> CHAIN
    static constexpr int kX0 = 14 + 2 * E;    // [E] bucket offsets                   PREP -> CHECK
    static constexpr int kDn = 14 + 3 * E;    // [E] distances                        CHAIN -> CHECK
    static constexpr int kWords = 14 + 4 * E;
    static constexpr int kSlots = 3;
    // + per slot one row of 64 words: d entering the group (CHAIN -> CHECK)
    static constexpr size_t kBytes = ((size_t)kSlots * kG * kWords + kSlots) * 64 * sizeof(uint32_t);
};

template <int E>
__global__ __launch_bounds__(448) void k_sweep_uniform_gen(const uint32_t* __restrict__ boff,
                                                           const uint64_t* __restrict__ contig_pos_off,
                                                           uint32_t ell, uint32_t M, uint32_t ltot,
                                                           uint32_t* __restrict__ selend,
                                                           uint32_t* __restrict__ iter_stats) {
    using Ly = MgLayout<E>;
    constexpr int kG = Ly::kG;
    extern __shared__ uint32_t s_mw[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t role = wv == 3 ? 1u : (wv >= 5 ? 2u : 0u);  // 0 PREP, 1 CHAIN, 2 CHECK
    constexpr int kPB = kG / 4, kCB = kG / 2;                  // blocks per PREP wave / per CHECK wave
    const uint32_t pblk = kPB * (wv == 4 ? 3u : wv);           // PREP: first of its blocks of the group
    const uint32_t cblk = kCB * (wv - 5);                      // CHECK: first of its blocks
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    if (L == 0) return;
    const uint32_t n_blocks = (L + ell - 1) / ell;
    const uint32_t n_groups = n_blocks / kG;
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    const uint32_t trash = ltot - base;
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;
    __builtin_amdgcn_s_setprio(3);

    uint32_t h[E];
    {
        const uint32_t b0 = cb[0];
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t i = lane * E + r;
            const uint32_t cov = cb[min(i + 1, L)] - b0;
            h[r] = (i < ell && i < L) ? (cov > M ? cov - M : 0u) : kInf;
        }
    }
    uint32_t d_last = 0;

#define MG_SLOT0(idx) (s_mw + (size_t)(idx) * kG * Ly::kWords * 64)
#define MG_SLOT(idx) (MG_SLOT0(idx) + lane)
#define MG_AT(slot, k, w) (slot)[((k) * Ly::kWords + (w)) * 64]
#define MG_DIN(idx) (s_mw + ((size_t)Ly::kSlots * kG * Ly::kWords + (idx)) * 64 + lane)[0]

    if (role == 0) {
        // PREP: one block of every group; rows three stages ahead in three register sets (see
        // k_sweep_uniform_mw); its own unrolled copy of the stage loop
        constexpr uint32_t kD = 3;
        constexpr int kRows = kPB + 2;  // rows its blocks need
        RowRaw<E> R0[kRows], R1[kRows], R2[kRows];
        auto issue_rows = [&](RowRaw<E> (&buf)[kRows], uint32_t g) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) row_issue<E>(cb, (g * kG + pblk + k) * ell, L, lane, buf[k]);
        };
        issue_rows(R0, 0);
        issue_rows(R1, 1);
        issue_rows(R2, 2);
        auto pstage = [&](RowRaw<E> (&buf)[kRows], uint32_t t) {
            if (t < n_groups) {
                uint32_t Wr[kRows][E];
#pragma unroll
                for (int k = 0; k < kRows; ++k) row_finish<E>(buf[k], Wr[k]);
                __builtin_amdgcn_sched_barrier(0);
                issue_rows(buf, t + kD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < kPB; ++kk) {
                    uint32_t* const slot = MG_SLOT(t % Ly::kSlots) + (pblk + kk) * Ly::kWords * 64;
                    SweepLoads<E> ld;
                    rows_to_loads<E>(Wr[kk], Wr[kk + 1], Wr[kk + 2], lane, last_lane, last_r, ld);
                    BlockTerms<E> bt;
                    block_terms<E>(ld, (t * kG + pblk + kk) * ell, ell, L, M, lane, bt);
                    // the lane's own (a, b): composition of its E single-position maps
                    uint32_t a = bt.cnt[0], b = bt.cnt[0] + bt.exj[0];
#pragma unroll
                    for (int r = 1; r < E; ++r) {
                        const uint32_t a2 = bt.cnt[r], b2 = bt.cnt[r] + bt.exj[r];
                        const uint32_t na = min(a + a2, b), nb = min(a + b2, b);
                        a = na; b = nb;
                    }
                    // the (a, b) half of the wave scan, recording what each lane holds before every step
#define MG_AB_STEP(s, ctrl, rmask)                                                         \
                    {                                                                      \
                        MG_AT(slot, 0, Ly::kA + (s)) = a;                                  \
                        MG_AT(slot, 0, Ly::kB + (s)) = b;                                  \
                        const uint32_t aL = QMCP_DPP(0u, a, ctrl, rmask);                  \
                        const uint32_t bL = QMCP_DPP(kInf, b, ctrl, rmask);                \
                        const uint32_t na = min(aL + a, bL), nb = min(aL + b, bL);         \
                        a = na; b = nb;                                                    \
                    }
                    MG_AB_STEP(0, 0x111, 0xF)
                    MG_AB_STEP(1, 0x112, 0xF)
                    MG_AB_STEP(2, 0x114, 0xF)
                    MG_AB_STEP(3, 0x118, 0xF)
                    MG_AB_STEP(4, 0x142, 0xA)
                    MG_AB_STEP(5, 0x143, 0xC)
#undef MG_AB_STEP
                    MG_AT(slot, 0, Ly::kPa) = QMCP_DPP(0u, a, 0x138, 0xF);     // all lower lanes (lane 0: identity)
                    MG_AT(slot, 0, Ly::kPb) = QMCP_DPP(kInf, b, 0x138, 0xF);
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        MG_AT(slot, 0, Ly::kCnt + r) = bt.cnt[r];
                        MG_AT(slot, 0, Ly::kEx + r) = bt.exj[r];
                        MG_AT(slot, 0, Ly::kX0 + r) = ld.x0[r];
                    }
                }
            }
            __syncthreads();
        };
        for (uint32_t t = 0;; t += kD) {
            if (t >= n_groups + 2) break;
            pstage(R0, t);
            if (t + 1 >= n_groups + 2) break;
            pstage(R1, t + 1);
            if (t + 2 >= n_groups + 2) break;
            pstage(R2, t + 2);
        }
    } else {
        for (uint32_t t = 0; t < n_groups + 2; ++t) {
            if (role == 1) {
                if (t >= 1 && t <= n_groups) {
                    const uint32_t g = t - 1;
                    uint32_t* const slot = MG_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MG_SLOT0(g % Ly::kSlots);
                    MG_DIN(g % Ly::kSlots) = d_last;
#pragma unroll
                    for (int k = 0; k < kG; ++k) {
                        uint32_t cnt[E], exj[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            cnt[r] = MG_AT(slot, k, Ly::kCnt + r);
                            exj[r] = MG_AT(slot, k, Ly::kEx + r);
                        }
                        uint32_t as[6], bs[6];
#pragma unroll
                        for (int q = 0; q < 6; ++q) {
                            as[q] = MG_AT(slot, k, Ly::kA + q);
                            bs[q] = MG_AT(slot, k, Ly::kB + q);
                        }
                        const uint32_t pa = MG_AT(slot, k, Ly::kPa), pb = MG_AT(slot, k, Ly::kPb);
                        // A(i) = min over j >= i of the previous block's h: in-lane suffix + wave suffix
                        uint32_t A[E];
                        uint32_t sm = kInf;
#pragma unroll
                        for (int r = E - 1; r >= 0; --r) { sm = min(sm, h[r]); A[r] = sm; }
                        {
                            uint32_t ws = sm;
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x101, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x102, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x104, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x108, 0xF));
                            const uint32_t r1 = __builtin_amdgcn_readlane(ws, 16);
                            const uint32_t r2 = __builtin_amdgcn_readlane(ws, 32);
                            const uint32_t r3 = __builtin_amdgcn_readlane(ws, 48);
                            const uint32_t row = lane >> 4;
                            const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
                            ws = min(min(ws, r1 | off1), min(r2 | off2, r3 | off3));
                            const uint32_t after = min(QMCP_DPP(0xFFFFFFFFu, ws, 0x130, 0xF), kInf);
#pragma unroll
                            for (int r = 0; r < E; ++r) A[r] = min(A[r], after);
                        }
                        // the lane's own (u, v)
                        uint32_t u = A[0], v = A[0] + exj[0];
#pragma unroll
                        for (int r = 1; r < E; ++r) {
                            const uint32_t nu = min(min(u + cnt[r], v), A[r]);
                            const uint32_t nv = min(min(u + cnt[r] + exj[r], v), A[r] + exj[r]);
                            u = nu; v = nv;
                        }
                        // the (u, v) half of the wave scan
#define MG_UV_STEP(s, ctrl, rmask)                                                     \
                        {                                                              \
                            const uint32_t uL = QMCP_DPP(kInf, u, ctrl, rmask);        \
                            const uint32_t vL = QMCP_DPP(kInf, v, ctrl, rmask);        \
                            const uint32_t nu = min(min(uL + as[s], vL), u);           \
                            const uint32_t nv = min(min(uL + bs[s], vL), v);           \
                            u = nu; v = nv;                                            \
                        }
                        MG_UV_STEP(0, 0x111, 0xF)
                        MG_UV_STEP(1, 0x112, 0xF)
                        MG_UV_STEP(2, 0x114, 0xF)
                        MG_UV_STEP(3, 0x118, 0xF)
                        MG_UV_STEP(4, 0x142, 0xA)
                        MG_UV_STEP(5, 0x143, 0xC)
#undef MG_UV_STEP
                        const uint32_t pu = QMCP_DPP(kInf, u, 0x138, 0xF);
                        const uint32_t pv = QMCP_DPP(kInf, v, 0x138, 0xF);
                        // state entering this lane: the map of all lower lanes applied to (d_last, +inf)
                        uint32_t dd = min(d_last + pa, pu);
                        uint32_t m = min(d_last + pb, pv);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            dd = min(min(dd + cnt[r], m), A[r]);
                            MG_AT(slot, k, Ly::kDn + r) = dd;
                            h[r] = dd + exj[r];
                            m = min(m, h[r]);
                        }
                        d_last = slot0[((k * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                    }
                }
            } else {
                if (t >= 2) {
                    const uint32_t g = t - 2;
                    uint32_t* const slot = MG_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MG_SLOT0(g % Ly::kSlots);
#pragma unroll
                    for (int kk = 0; kk < kCB; ++kk) {
                        const uint32_t k = cblk + kk;
                        uint32_t dn[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) dn[r] = MG_AT(slot, k, Ly::kDn + r);
                        const uint32_t d_blk = k == 0 ? MG_DIN(g % Ly::kSlots)
                                                      : slot0[(((k - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                        uint32_t prev = QMCP_DPP(0u, dn[E - 1], 0x138, 0xF);
                        prev = lane == 0 ? d_blk : prev;
                        uint32_t sel[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            sel[r] = MG_AT(slot, k, Ly::kX0 + r) + (MG_AT(slot, k, Ly::kCnt + r) - (dn[r] - prev));
                            prev = dn[r];
                        }
                        const uint32_t blk_first = (g * kG + k) * ell;
                        const uint32_t p0 = blk_first + lane * E;
                        const bool full = lane * E + E <= ell && p0 + E <= L;
                        if constexpr (E == 1) {
                            csel[full ? p0 : trash] = sel[0];
                        } else {
                            typedef typename RowVec<E>::type V;
                            V vv;
#pragma unroll
                            for (int r = 0; r < E; ++r) vv[r] = sel[r];
                            *reinterpret_cast<V*>(csel + (full ? p0 : trash)) = vv;
                            if (ell % E != 0 || blk_first + ell > L) {
#pragma unroll
                                for (int r = 0; r < E; ++r) {
                                    const uint32_t i = lane * E + r;
                                    if (!full && i < ell && blk_first + i < L) csel[blk_first + i] = sel[r];
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#undef MG_AT
#undef MG_DIN
#undef MG_SLOT
#undef MG_SLOT0
    if (role == 1) {
        if (n_groups * kG < n_blocks)
            sweep_full_run<E>(cb, n_groups * kG, n_blocks, trash, ell, L, M, lane, last_lane, last_r, h, d_last,
                              csel);
        if (iter_stats && lane == 0) {
            atomicAdd(&iter_stats[0], n_blocks);  // every block is in the general form here
            atomicAdd(&iter_stats[1], n_blocks);
        }
    }
}

// ------------------------------------------------------------------ general (mixed-span) sweep
// Event-driven form of the canonical rule for arbitrary spans.  Reads are bucketed by start
// and ordered (end desc, index asc) inside a bucket, so the selected reads of a bucket are
// always a prefix; the pool of candidates at position p is the set of bucket heads of the
// last max_span start positions, compared by (end desc, start desc).
// One wave per contig.  Two rings of `ring_size` (power of two > max_span) entries live in
// LDS: the prefix pointer of every bucket still inside the window, and the number of
// selected reads by end position (what stops covering when the sweep passes that end).
// A bucket's pointer is flushed to selend (as an absolute offset) when its slot is recycled.
struct SortedRec { const Rec* r; __device__ uint64_t key(uint32_t j) const { return r[j].key; } };
struct SortedK64 { const uint64_t* k; __device__ uint64_t key(uint32_t j) const { return k[j]; } };

template <typename Sorted>
__global__ __launch_bounds__(64) void k_sweep_general(const uint32_t* __restrict__ boff,
                                                      const uint32_t* __restrict__ eoff,
                                                      Sorted skeys,
                                                      const uint64_t* __restrict__ contig_pos_off,
                                                      uint32_t span_bits, uint32_t max_span,
                                                      uint32_t M, uint32_t* __restrict__ selend,
                                                      uint32_t ring_size) {
    extern __shared__ uint32_t s_ring[];
    uint32_t* s_ptr = s_ring;              // [ring_size] bucket prefix pointers
    uint32_t* s_exp = s_ring + ring_size;  // [ring_size] selected reads by end position
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    const uint32_t rmask = ring_size - 1;
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = lane; i < 2 * ring_size; i += 64) s_ring[i] = 0;
    __syncthreads();
    uint32_t cur = 0;
    for (uint32_t p = 0; p < L; ++p) {
        const uint32_t gp = base + p;
        if (lane == 0) {
            if (p >= ring_size) {
                const uint32_t q = p - ring_size;  // long dead: ring_size > max_span
                selend[base + q] = boff[base + q] + s_ptr[p & rmask];
            }
            s_ptr[p & rmask] = 0;
        }
        __syncthreads();
        const uint32_t cov = boff[gp + 1] - eoff[gp];
        const uint32_t need = min(cov, M);
        uint32_t k = need > cur ? need - cur : 0u;
        while (k > 0) {
            // best head among buckets q in (p - max_span, p]
            uint64_t best = 0;
            for (uint32_t t = lane; t < max_span && t <= p; t += 64) {
                const uint32_t q = p - t;
                const uint32_t gq = base + q;
                const uint32_t b0 = boff[gq], b1 = boff[gq + 1];
                const uint32_t ptr = s_ptr[q & rmask];
                if (b0 + ptr < b1) {
                    const uint64_t key = skeys.key(b0 + ptr);
                    const uint32_t span = max_span - (uint32_t)(key & code_mask);
                    const uint32_t end = q + span - 1;
                    if (end >= p) {
                        const uint64_t pri = ((uint64_t)(end + 1) << 32) | (uint64_t)(q + 1);
                        best = pri > best ? pri : best;
                    }
                }
            }
            best = wave_max_u64(best);
            // need <= cov guarantees a candidate; guard anyway so the loop always ends
            if (best == 0) break;
            const uint32_t bend = (uint32_t)(best >> 32) - 1;
            const uint32_t bq = (uint32_t)(best & 0xFFFFFFFFu) - 1;
            const uint32_t gq = base + bq;
            const uint32_t b0 = boff[gq], b1 = boff[gq + 1];
            const uint32_t ptr = s_ptr[bq & rmask];
            // length of the run of equal-end reads at the head of the winning bucket (<= 64)
            bool same = false;
            const uint32_t j = b0 + ptr + lane;
            if (j < b1) {
                const uint64_t key = skeys.key(j);
                const uint32_t span = max_span - (uint32_t)(key & code_mask);
                same = (bq + span - 1) == bend;
            }
            const uint64_t ball = __ballot(same);
            const uint32_t run = (~ball == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~ball) - 1);
            const uint32_t take = min(k, run);
            __syncthreads();
            if (lane == 0) {
                s_ptr[bq & rmask] = ptr + take;
                s_exp[bend & rmask] += take;
            }
            __syncthreads();
            cur += take;
            k -= take;
        }
        // reads ending at p stop covering p+1
        const uint32_t ex = s_exp[p & rmask];
        cur -= ex;
        __syncthreads();
        if (lane == 0) s_exp[p & rmask] = 0;
    }
    __syncthreads();
    // flush the buckets still in the ring
    const uint32_t first = L > ring_size ? L - ring_size : 0u;
    for (uint32_t q = first + lane; q < L; q += 64) selend[base + q] = boff[base + q] + s_ptr[q & rmask];
}
// ------------------------------------------------------------------ mixed-span sweep, LDS-cached
// Same rule as k_sweep_general, organised so that the serial loop touches LDS only:
//   * a preprocessing pass marks run heads of equal composite keys (a "group": reads with the
//     same start and end) and a reverse min-scan turns them into next_head[], so the length of
//     the run starting at j is next_head[j + 1] - j;
//   * positions are taken 64 at a time: bucket bounds, coverage and the first TWO groups of
//     every entering bucket are loaded with wave-wide (not serially dependent) loads into an LDS
//     ring of `ring` slots (power of two >= max_span + 64, so that a slot is only recycled once
//     its previous bucket is dead even for the last position of a chunk), and the previous
//     occupants of those slots flush their selected counts to selend;
//   * a selection event is a wave-wide maximum over the cached bucket heads, key
//     (end - p + 1) << 16 | (0xFFFF - (p - q)): largest end, then largest start; it takes
//     min(deficit, run) reads from the winning group.  Only when a bucket has used up both cached
//     groups is its next group fetched from memory.
// One wave per contig; spans up to kMaxCachedSpan.
struct GenSlots {  // layout of the LDS ring, in 32-bit words per slot
    // G0 / G1: (end + 1, run) of the bucket's head group and of the cached second group,
    // 8 bytes each so one ds_read_b64 fetches both fields
    enum { kG0 = 0, kG1 = 2, kNextJ = 4, kB1 = 5, kTaken = 6, kExp = 7, kWords = 8 };
};

template <typename Sorted>
__global__ __launch_bounds__(256) void k_group_heads(Sorted skeys, uint32_t n,
                                                     uint32_t* __restrict__ next_head) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += stride) {
        uint32_t v = 0xFFFFFFFFu;
        if (j == n) v = n;
        else if (j == 0 || skeys.key(j - 1) != skeys.key(j)) v = j;
        next_head[j] = v;
    }
}

template <typename Sorted>
__global__ __launch_bounds__(64) void k_sweep_general_cached(
    const uint32_t* __restrict__ boff, const uint32_t* __restrict__ eoff, Sorted skeys,
    const uint32_t* __restrict__ next_head, const uint64_t* __restrict__ contig_pos_off,
    uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* __restrict__ selend, uint32_t ring) {
    extern __shared__ uint32_t s_gen[];
    uint2* s_g0 = reinterpret_cast<uint2*>(s_gen + GenSlots::kG0 * ring);
    uint2* s_g1 = reinterpret_cast<uint2*>(s_gen + GenSlots::kG1 * ring);
    uint32_t* s_nextj = s_gen + GenSlots::kNextJ * ring;
    uint32_t* s_b1 = s_gen + GenSlots::kB1 * ring;
    uint32_t* s_taken = s_gen + GenSlots::kTaken * ring;
    uint32_t* s_exp = s_gen + GenSlots::kExp * ring;
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    const uint32_t rmask = ring - 1;
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = lane; i < GenSlots::kWords * ring; i += 64) s_gen[i] = 0;
    __syncthreads();
    const uint32_t* __restrict__ cb = boff + base;
    const uint32_t* __restrict__ ce = eoff + base;
    uint32_t* __restrict__ csel = selend + base;
    uint32_t cur = 0;
    // One wave per workgroup: its LDS operations execute in program order, so lane-0 updates
    // are visible to every lane's next read without barriers.
    for (uint32_t p0 = 0; p0 < L; p0 += 64) {
        // ---- enter the chunk's 64 buckets (lane = position p0 + lane)
        const uint32_t q = p0 + lane;
        const uint32_t slot = q & rmask;
        uint32_t need = 0;
        uint32_t exp_c = 0;  // selected reads ending at position p0 + lane
        if (q < L) {
            if (q >= ring) csel[q - ring] = cb[q - ring] + s_taken[slot];  // recycled slot
            exp_c = s_exp[slot];
            s_exp[slot] = 0;
            const uint32_t b0 = cb[q], b1 = cb[q + 1];
            need = min(b1 - ce[q], M);  // cov(q) = boff[q + 1] - eoff[q]
            uint2 g0 = make_uint2(0, 0), g1 = make_uint2(0, 0);
            uint32_t nj = b1;
            if (b0 < b1) {
                const uint64_t k0 = skeys.key(b0);
                g0.y = min(next_head[b0 + 1], b1) - b0;
                g0.x = q + (max_span - (uint32_t)(k0 & code_mask));  // end + 1
                const uint32_t j1 = b0 + g0.y;
                nj = j1;
                if (j1 < b1) {
                    const uint64_t k1 = skeys.key(j1);
                    g1.y = min(next_head[j1 + 1], b1) - j1;
                    g1.x = q + (max_span - (uint32_t)(k1 & code_mask));
                    nj = j1 + g1.y;
                }
            }
            s_g0[slot] = g0;
            s_g1[slot] = g1;
            s_nextj[slot] = nj;
            s_b1[slot] = b1;
            s_taken[slot] = 0;
        }
        // ---- walk the chunk's positions; per-position need / expiry come from lane registers
        const uint32_t chunk = min(64u, L - p0);
        for (uint32_t j = 0; j < chunk; ++j) {
            const uint32_t p = p0 + j;
            const uint32_t need_p = __builtin_amdgcn_readlane(need, j);
            uint32_t k = need_p > cur ? need_p - cur : 0u;
            while (k > 0) {
                // best live head among buckets q' in (p - max_span, p]:
                // key = (end + 1 - p) << 16 | (0xFFFF - (p - q')): largest end, then largest start
                uint32_t best = 0, my_run = 0;
                for (uint32_t t = lane; t < max_span && t <= p; t += 64) {
                    const uint2 g = s_g0[(p - t) & rmask];
                    if (g.x > p) {
                        const uint32_t key = ((g.x - p) << 16) | (0xFFFFu - t);
                        if (key > best) { best = key; my_run = g.y; }
                    }
                }
                uint32_t top = best;
                top = max(top, QMCP_DPP(0u, top, 0x111, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x112, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x114, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x118, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x142, 0xA));
                top = max(top, QMCP_DPP(0u, top, 0x143, 0xC));
                top = __builtin_amdgcn_readlane(top, 63);
                if (top == 0) break;  // cannot happen (need <= cov); keeps the loop finite
                const uint32_t src = (uint32_t)__ffsll((long long)__ballot(best == top)) - 1;
                const uint32_t run = __builtin_amdgcn_readlane(my_run, src);
                const uint32_t bq = p - (0xFFFFu - (top & 0xFFFFu));
                const uint32_t bslot = bq & rmask;
                const uint32_t bend = p + (top >> 16) - 1;  // end of the winning group
                const uint32_t take = min(k, run);
                // expiry bookkeeping: inside the chunk in the lane register, beyond it in the ring
                if (bend < p0 + 64) {
                    exp_c += (lane == bend - p0) ? take : 0u;
                } else if (lane == 0) {
                    atomicAdd(&s_exp[bend & rmask], take);
                }
                if (lane == 0) {
                    atomicAdd(&s_taken[bslot], take);
                    if (take < run) {
                        s_g0[bslot].y = run - take;
                    } else {
                        const uint2 g1 = s_g1[bslot];
                        if (g1.y != 0) {
                            // group used up: promote the cached second group (refilled lazily)
                            s_g0[bslot] = g1;
                            s_g1[bslot].y = 0;
                        } else {
                            // both cached groups used: fetch the bucket's next group, if any
                            const uint32_t nj = s_nextj[bslot];
                            const uint32_t b1 = s_b1[bslot];
                            uint2 g0 = make_uint2(0, 0);
                            if (nj < b1) {
                                const uint64_t kk = skeys.key(nj);
                                g0.y = min(next_head[nj + 1], b1) - nj;
                                g0.x = bq + (max_span - (uint32_t)(kk & code_mask));
                                s_nextj[bslot] = nj + g0.y;
                            }
                            s_g0[bslot] = g0;
                        }
                    }
                }
                cur += take;
                k -= take;
            }
            // reads ending at p stop covering p + 1
            cur -= __builtin_amdgcn_readlane(exp_c, j);
        }
    }
    // flush the buckets still in the ring
    const uint32_t first = L > ring ? L - ring : 0u;
    for (uint32_t qq = first + lane; qq < L; qq += 64) csel[qq] = cb[qq] + s_taken[qq & rmask];
}

// ------------------------------------------------------------------ keep-mask emission
// sorted entry j (bucket = its start position) is kept iff j < selend[bucket].
// obtain_sequence counterpart (quasi_mcp_cpu_max_flow_solver.cpp:89-100).
// One thread per start position walks that bucket's selected prefix [boff[q], selend[q]) --
// at most M entries, usually 0..2 -- and sets the kept reads' bits: the sorted records of the
// other ~95 % of the reads are never touched.
// Register-resident form of the cached event sweep, for max_span + 64 <= 64 * B: the window of
// live buckets is at most 64 * B positions wide, so every lane OWNS B of them (bucket q belongs to
// lane q % 64, slot (q / 64) % B) and keeps their head group, cached second group, read pointers
// and selected count in registers.  A selection event is then: every lane's best over its own B
// slots (register compares), a fused-DPP wave maximum, and a register update in the winning lane
// -- no LDS round trip on the serial path (the LDS version pays three or four per event).  Only
// the expiry counts of reads that end beyond the current 64-position chunk go through an LDS
// ring (fire-and-forget adds, read back one chunk later).  The chunk loop is unrolled B times so
// that the slot a chunk's buckets enter is a compile-time index.
template <typename Sorted, int B>
__global__ __launch_bounds__(64) void k_sweep_general_reg(
    const uint32_t* __restrict__ boff, const uint32_t* __restrict__ eoff, Sorted skeys,
    const uint32_t* __restrict__ next_head, const uint64_t* __restrict__ contig_pos_off,
    uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* __restrict__ selend
#ifdef QMCP_GEN_STAMP
    , unsigned long long* __restrict__ stamps  // lab builds: [0] entry cycles [1] events [2] event cycles [3] fetches [4] fetch cycles [5] walk cycles
#endif
    ) {
    constexpr uint32_t kRing = 64 * B;  // >= max_span + 64
    __shared__ uint32_t s_exp[kRing];
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    const uint32_t base = (uint32_t)contig_pos_off[c_id];
    const uint32_t L = (uint32_t)(contig_pos_off[c_id + 1] - contig_pos_off[c_id]);
    if (L == 0) return;
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = lane; i < kRing; i += 64) s_exp[i] = 0;
    __syncthreads();
    const uint32_t* __restrict__ cb = boff + base;
    const uint32_t* __restrict__ ce = eoff + base;
    uint32_t* __restrict__ csel = selend + base;
    // per owned bucket: head group (end + 1, run), cached second group, next unread group, bucket
    // end, reads selected so far
    uint32_t g0x[B], g0y[B], g1x[B], g1y[B], nextj[B], bend1[B], taken[B];
#pragma unroll
    for (int b = 0; b < B; ++b) { g0x[b] = g0y[b] = g1x[b] = g1y[b] = nextj[b] = bend1[b] = taken[b] = 0; }
    uint32_t cur = 0;
    const uint32_t n_chunks = (L + 63) / 64;

    auto load_group = [&](uint32_t j, uint32_t b1, uint32_t q, uint32_t& gx, uint32_t& gy) {
        // group starting at sorted index j of the bucket of position q ending at b1 (gy = 0: none)
        gx = 0; gy = 0;
        if (j < b1) {
            const uint64_t kk = skeys.key(j);
            gy = min(next_head[j + 1], b1) - j;
            gx = q + (max_span - (uint32_t)(kk & code_mask));  // end + 1
        }
    };

    for (uint32_t c0 = 0; c0 < n_chunks; c0 += B) {
#pragma unroll
        for (int e = 0; e < B; ++e) {
            const uint32_t c = c0 + e;
            if (c >= n_chunks) break;
            const uint32_t p0 = c * 64;
#ifdef QMCP_GEN_STAMP
            const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
            // ---- the chunk's 64 buckets enter slot e (lane = position p0 + lane)
            const uint32_t q = p0 + lane;
            uint32_t need = 0, exp_c = 0;
            if (c >= (uint32_t)B && q - kRing < L) csel[q - kRing] = cb[q - kRing] + taken[e];  // recycled slot
            g0x[e] = g0y[e] = g1x[e] = g1y[e] = 0; nextj[e] = bend1[e] = taken[e] = 0;
            if (q < L) {
                exp_c = s_exp[q % kRing];
                s_exp[q % kRing] = 0;
                const uint32_t b0 = cb[q], b1 = cb[q + 1];
                need = min(b1 - ce[q], M);  // cov(q) = boff[q + 1] - eoff[q]
                bend1[e] = b1;
                load_group(b0, b1, q, g0x[e], g0y[e]);
                load_group(b0 + g0y[e], b1, q, g1x[e], g1y[e]);
                nextj[e] = b0 + g0y[e] + g1y[e];
            }
#ifdef QMCP_GEN_STAMP
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long st_e1 = __builtin_amdgcn_s_memtime();
            unsigned long long st_ev = 0, st_nev = 0, st_fe = 0, st_nfe = 0;
#endif
            // ---- this lane's best head over the buckets it owns, kept up to date incrementally.
            // key = (end + 1 - pbase) << 16 | (q' - pbase), pbase = p0 - 64 (B - 1): largest end first,
            // then largest start; valid for this chunk.  Its head group is live at p iff end >= p; if the
            // lane's best is dead so is everything else it owns (smaller ends).
            const uint32_t pbase = p0 - 64u * (uint32_t)(B - 1);  // (wraps for the first chunks: consistently)
            uint32_t lbest = 0, lrun = 0, lslot = 0, key_e = 0;
            auto lane_best = [&](bool with_entering) {
                lbest = 0; lrun = 0; lslot = 0;
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const uint32_t back = 64u * (uint32_t)((e - b + B) % B);   // chunks ago, in positions
                    const uint32_t qrel = 64u * (uint32_t)(B - 1) - back + lane; // q' - pbase
                    const uint32_t key = g0y[b] != 0 ? (((g0x[b] - pbase) << 16) | qrel) : 0u;
                    if (b == e) key_e = key;
                    const bool started = b != e || with_entering;
                    if (started && key > lbest) { lbest = key; lrun = g0y[b]; lslot = b; }
                }
            };
            lane_best(false);
            // ---- walk the chunk's positions
            const uint32_t chunk = min(64u, L - p0);
            for (uint32_t j = 0; j < chunk; ++j) {
                const uint32_t p = p0 + j;
                // the bucket of position p starts now
                if (lane == j && key_e > lbest) { lbest = key_e; lrun = g0y[e]; lslot = e; }
                const uint32_t need_p = __builtin_amdgcn_readlane(need, j);
                uint32_t k = need_p > cur ? need_p - cur : 0u;
                while (k > 0) {
#ifdef QMCP_GEN_STAMP
                    const unsigned long long st_v0 = __builtin_amdgcn_s_memtime();
#endif
                    const uint32_t best = (lbest >> 16) > p - pbase ? lbest : 0u;  // live: end + 1 > p
                    uint32_t top = best;
                    top = max(top, QMCP_DPP(0u, top, 0x111, 0xF));
                    top = max(top, QMCP_DPP(0u, top, 0x112, 0xF));
                    top = max(top, QMCP_DPP(0u, top, 0x114, 0xF));
                    top = max(top, QMCP_DPP(0u, top, 0x118, 0xF));
                    top = max(top, QMCP_DPP(0u, top, 0x142, 0xA));
                    top = max(top, QMCP_DPP(0u, top, 0x143, 0xC));
                    top = __builtin_amdgcn_readlane(top, 63);
                    if (top == 0) break;  // cannot happen (need <= cov); keeps the loop finite
                    const uint32_t src = (uint32_t)__ffsll((long long)__ballot(best == top)) - 1;
                    const uint32_t run = __builtin_amdgcn_readlane(lrun, src);
                    const uint32_t bend = pbase + (top >> 16) - 1;  // end of the winning group
                    const uint32_t take = min(k, run);
                    // expiry bookkeeping: inside the chunk in the lane register, beyond it in the ring
                    if (bend < p0 + 64) {
                        exp_c += (lane == bend - p0) ? take : 0u;
                    } else if (lane == 0) {
                        atomicAdd(&s_exp[bend % kRing], take);
                    }
                    // the winning lane updates its own bucket in registers; the slot is made uniform so
                    // that only that slot's code runs
                    const uint32_t wslot = __builtin_amdgcn_readlane(lslot, src);
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        if (wslot == (uint32_t)b) {
                            if (lane == src) {
                                taken[b] += take;
                                if (take < run) {
                                    g0y[b] = run - take;
                                } else if (g1y[b] != 0) {
                                    g0x[b] = g1x[b]; g0y[b] = g1y[b]; g1y[b] = 0;   // promote the cached group
                                } else {
                                    // both cached groups used: fetch the bucket's next group, if any
                                    const uint32_t back = 64u * (uint32_t)((e - b + B) % B);
#ifdef QMCP_GEN_STAMP
                                    const unsigned long long st_f0 = __builtin_amdgcn_s_memtime();
#endif
                                    load_group(nextj[b], bend1[b], p0 + lane - back, g0x[b], g0y[b]);
                                    nextj[b] += g0y[b];
#ifdef QMCP_GEN_STAMP
                                    __builtin_amdgcn_s_waitcnt(0);
                                    st_fe += __builtin_amdgcn_s_memtime() - st_f0;
                                    st_nfe += 1;
#endif
                                }
                            }
                        }
                    }
                    if (lane == src) lane_best(lane <= j);  // its bucket changed: the lane's best again
                    cur += take;
                    k -= take;
#ifdef QMCP_GEN_STAMP
                    st_ev += __builtin_amdgcn_s_memtime() - st_v0;
                    st_nev += 1;
#endif
                }
                // reads ending at p stop covering p + 1
                cur -= __builtin_amdgcn_readlane(exp_c, j);
            }
#ifdef QMCP_GEN_STAMP
            {
                const unsigned long long st_w = __builtin_amdgcn_s_memtime() - st_e1;
                // fetch counters live in the winning lanes: reduce over the wave
                unsigned long long fe = 0, nfe = 0;
                for (int l = 0; l < 64; ++l) {
                    fe += __shfl((unsigned long long)st_fe, l, 64);
                    nfe += __shfl((unsigned long long)st_nfe, l, 64);
                }
                if (lane == 0 && stamps) {
                    atomicAdd(&stamps[0], st_e1 - st_e0);
                    atomicAdd(&stamps[1], st_nev);
                    atomicAdd(&stamps[2], st_ev);
                    atomicAdd(&stamps[3], nfe);
                    atomicAdd(&stamps[4], fe);
                    atomicAdd(&stamps[5], st_w);
                }
            }
#endif
        }
    }
    // flush the buckets still owned
    const uint32_t last_c = n_chunks - 1;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        // the most recent chunk that filled slot b
        if (last_c >= (uint32_t)b) {
            const uint32_t cc = last_c - ((last_c - (uint32_t)b) % (uint32_t)B);
            const uint32_t qq = cc * 64 + lane;
            if (qq < L) csel[qq] = cb[qq] + taken[b];
        }
    }
}

template <typename Keys>
__global__ __launch_bounds__(256) void k_mark(Keys keys, uint32_t ltot,
                                              const uint32_t* __restrict__ boff,
                                              const uint32_t* __restrict__ selend,
                                              uint32_t* __restrict__ mask32,
                                              unsigned long long* __restrict__ n_kept) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < ltot; q += stride) {
        const uint32_t j0 = boff[q], j1 = selend[q];
        for (uint32_t j = j0; j < j1; ++j) {
            const uint32_t idx = keys.idx(j);
            atomicOr(&mask32[idx >> 5], 1u << (idx & 31));
        }
        mine += j1 - j0;
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_kept, (unsigned long long)mine);
}

// cov[p] = #reads started at or before p - #reads ended before p
// (what BamApi::find_input_cover builds with per-base increments, bam_api.cpp:275-286)
__global__ __launch_bounds__(256) void k_coverage(const uint32_t* __restrict__ boff,
                                                  const uint32_t* __restrict__ eoff,
                                                  uint32_t ltot, uint32_t* __restrict__ cov) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < ltot; p += stride)
        cov[p] = boff[p + 1] - eoff[p];
}

// ------------------------------------------------------------------ "next" rows
// BamApi::find_pairs on the bitmask (bam_api.cpp:239-273): mates are (2q, 2q+1).
__global__ __launch_bounds__(256) void k_complete_pairs(uint64_t* __restrict__ mask,
                                                        uint32_t n_words, uint64_t n_reads) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint64_t even = 0x5555555555555555ull;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint64_t m = mask[w];
        m |= ((m & even) << 1) | ((m >> 1) & even);
        // an unpaired trailing read (odd n_reads) has no mate: never set bits past n_reads
        const uint64_t first = (uint64_t)w * 64;
        if (first + 64 > n_reads) {
            const uint32_t live = (uint32_t)(n_reads - first);
            m &= live >= 64 ? ~0ull : ((1ull << live) - 1ull);
        }
        mask[w] = m;
    }
}

// Amplicon FILTER predicate per pair (bam_api.cpp:311-327, amplicon.cpp:5-7,
// amplicon_set.cpp:5-9); one wave emits one 64-pair word with a ballot.
__global__ __launch_bounds__(256) void k_amplicon_filter(const uint32_t* __restrict__ starts,
                                                         const uint32_t* __restrict__ ends,
                                                         const uint32_t* __restrict__ seq_lengths,
                                                         const uint32_t* __restrict__ qualities,
                                                         uint64_t n_pairs,
                                                         const uint32_t* __restrict__ amp_starts,
                                                         const uint32_t* __restrict__ amp_ends,
                                                         uint32_t n_amp, uint32_t min_length,
                                                         uint32_t min_mapq,
                                                         uint64_t* __restrict__ pair_keep) {
    extern __shared__ uint32_t s_amp[];  // [2 * n_cached]
    const uint32_t n_cached = min(n_amp, 4096u);
    for (uint32_t i = threadIdx.x; i < n_cached; i += blockDim.x) {
        s_amp[i] = amp_starts[i];
        s_amp[n_cached + i] = amp_ends[i];
    }
    __syncthreads();
    const uint64_t n_words = (n_pairs + 63) / 64;
    const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t w = wave_global; w < n_words; w += n_waves) {
        const uint64_t q = w * 64 + lane;
        bool ok = false;
        if (q < n_pairs) {
            const uint64_t i = 2 * q, j = i + 1;
            const uint32_t s1 = starts[i], e1 = ends[i], s2 = starts[j], e2 = ends[j];
            bool pass = true;
            if (qualities) pass = pass && qualities[i] >= min_mapq && qualities[j] >= min_mapq;
            if (seq_lengths) pass = pass && seq_lengths[i] >= min_length && seq_lengths[j] >= min_length;
            bool in_one = false;
            for (uint32_t a = 0; a < n_cached && !in_one; ++a) {
                const uint32_t as = s_amp[a], ae = s_amp[n_cached + a];
                in_one = as <= s1 && e1 <= ae && as <= s2 && e2 <= ae;
            }
            for (uint32_t a = n_cached; a < n_amp && !in_one; ++a) {
                const uint32_t as = amp_starts[a], ae = amp_ends[a];
                in_one = as <= s1 && e1 <= ae && as <= s2 && e2 <= ae;
            }
            ok = pass && in_one;
        }
        const uint64_t word = __ballot(ok);
        if (lane == 0) pair_keep[w] = word;
    }
}

// ------------------------------------------------------------------ filter -> solve pipeline glue
// Stream compaction of the pairs that survive the FILTER (pairs stay adjacent: survivor q'
// becomes reads 2q', 2q'+1) and the map back to original read indices -- the device-resident
// equivalent of what BamApi does while ingesting (bam_api.cpp:434-461: only accepted pairs are
// appended) and of the id bookkeeping around the solver in App::execute (src/app.cpp:134-142).
__global__ __launch_bounds__(256) void k_word_popcounts(const uint64_t* __restrict__ words,
                                                        uint32_t n_words,
                                                        uint32_t* __restrict__ counts) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride)
        counts[w] = __popcll(words[w]);
}

__global__ __launch_bounds__(256) void k_compact_pairs(const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ ends,
                                                       const uint64_t* __restrict__ pair_keep,
                                                       const uint32_t* __restrict__ word_base,
                                                       uint64_t n_pairs,
                                                       uint32_t* __restrict__ starts_c,
                                                       uint32_t* __restrict__ ends_c,
                                                       uint32_t* __restrict__ orig_pair) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pairs; q += stride) {
        const uint64_t word = pair_keep[q >> 6];
        const uint32_t bit = (uint32_t)(q & 63);
        if ((word >> bit) & 1ull) {
            const uint32_t dst = word_base[q >> 6] + (uint32_t)__popcll(word & ((1ull << bit) - 1ull));
            starts_c[2 * dst] = starts[2 * q];
            starts_c[2 * dst + 1] = starts[2 * q + 1];
            ends_c[2 * dst] = ends[2 * q];
            ends_c[2 * dst + 1] = ends[2 * q + 1];
            orig_pair[dst] = (uint32_t)q;
        }
    }
}

__global__ __launch_bounds__(256) void k_expand_mask(const uint64_t* __restrict__ mask_c,
                                                     const uint32_t* __restrict__ orig_pair,
                                                     uint32_t n_reads_c,
                                                     uint32_t* __restrict__ mask32) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reads_c; i += stride) {
        if ((mask_c[i >> 6] >> (i & 63)) & 1ull) {
            const uint32_t orig = 2u * orig_pair[i >> 1] + (i & 1u);
            atomicOr(&mask32[orig >> 5], 1u << (orig & 31));
        }
    }
}

// ------------------------------------------------------------------ host-side launchers
static inline uint32_t grid_for(uint64_t n, uint32_t block, uint32_t cap = 256 * 8) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

static inline uint32_t tiles_per_block_for(uint32_t n_tiles);

void launch_prepare(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n,
                    const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                    const uint64_t* keep_mask, uint32_t* gstart, uint32_t* cstart,
                    uint32_t* stats, uint32_t part_shift, uint32_t* part_hist,
                    uint32_t* digit0_hist, uint32_t* global_digit_hist, unsigned long long* zero_mask) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const uint32_t g = tiles_per_block_for(n_tiles);
    hipLaunchKernelGGL(k_prepare, dim3((n_tiles + g - 1) / g), dim3(256), 0, st, starts, ends, n,
                       d_roff, d_poff, n_contigs, keep_mask, gstart, cstart, stats, n_tiles, g,
                       part_shift, part_hist, digit0_hist, global_digit_hist, zero_mask);
}

void launch_general_keys(hipStream_t st, bool wide, const uint32_t* gstart, const uint32_t* starts,
                         const uint32_t* ends, uint32_t n, uint32_t span_bits, uint32_t max_span,
                         const uint64_t* keep_mask, void* keys, uint32_t* ecnt) {
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

bool sweep_uniform_mw_supported(uint32_t ell) { return ell >= 1 && (ell + 63) / 64 <= 4; }

bool launch_sweep_uniform_mw(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                             uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                             uint32_t* selend, uint32_t* iter_stats) {
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP_MW(EE)                                                                              \
    {                                                                                                   \
        const size_t lds = MwLayout<EE>::kBytes;                                                        \
        (void)hipFuncSetAttribute((const void*)k_sweep_uniform_mw<EE>,                                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL(k_sweep_uniform_mw<EE>, dim3(n_contigs), dim3(448), lds, st, boff, d_poff,   \
                           ell, M, ltot, selend, iter_stats);                                           \
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
                              uint32_t* selend, uint32_t* iter_stats) {
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP_GEN(EE)                                                                             \
    {                                                                                                   \
        const size_t lds = MgLayout<EE>::kBytes;                                                        \
        (void)hipFuncSetAttribute((const void*)k_sweep_uniform_gen<EE>,                                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
        hipLaunchKernelGGL(k_sweep_uniform_gen<EE>, dim3(n_contigs), dim3(448), lds, st, boff, d_poff,  \
                           ell, M, ltot, selend, iter_stats);                                           \
    }
    switch (e) {
        case 1: QMCP_SWEEP_GEN(1); break;
        case 2: QMCP_SWEEP_GEN(2); break;
        case 3: QMCP_SWEEP_GEN(3); break;
        case 4: QMCP_SWEEP_GEN(4); break;
        default: return false;
    }
#undef QMCP_SWEEP_GEN
    return true;
}

bool launch_sweep_uniform(hipStream_t st, const uint32_t* boff, const uint64_t* d_poff,
                          uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                          uint32_t* selend, uint32_t* iter_stats) {
    const uint32_t e = (ell + 63) / 64;
#define QMCP_SWEEP(EE)                                                                          \
    hipLaunchKernelGGL(k_sweep_uniform<EE>, dim3(n_contigs), dim3(64), 0, st, boff, d_poff, ell, \
                       M, ltot, selend, iter_stats)
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
                          uint32_t ring_size) {
    const size_t lds = 2 * (size_t)ring_size * sizeof(uint32_t);
    if (wide) {
        (void)hipFuncSetAttribute((const void*)k_sweep_general<SortedK64>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general<SortedK64>, dim3(n_contigs), dim3(64), lds, st, boff, eoff,
                           SortedK64{(const uint64_t*)sorted}, d_poff, span_bits, max_span, M, selend,
                           ring_size);
    } else {
        (void)hipFuncSetAttribute((const void*)k_sweep_general<SortedRec>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general<SortedRec>, dim3(n_contigs), dim3(64), lds, st, boff, eoff,
                           SortedRec{(const Rec*)sorted}, d_poff, span_bits, max_span, M, selend,
                           ring_size);
    }
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
                                 uint32_t max_span, uint32_t M, uint32_t* selend, uint32_t ring) {
    const size_t lds = (size_t)GenSlots::kWords * ring * sizeof(uint32_t);
    if (wide) {
        (void)hipFuncSetAttribute((const void*)k_sweep_general_cached<SortedK64>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general_cached<SortedK64>, dim3(n_contigs), dim3(64), lds, st, boff,
                           eoff, SortedK64{(const uint64_t*)sorted}, next_head, d_poff, span_bits,
                           max_span, M, selend, ring);
    } else {
        (void)hipFuncSetAttribute((const void*)k_sweep_general_cached<SortedRec>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_sweep_general_cached<SortedRec>, dim3(n_contigs), dim3(64), lds, st, boff,
                           eoff, SortedRec{(const Rec*)sorted}, next_head, d_poff, span_bits, max_span,
                           M, selend, ring);
    }
}

// register-resident event sweep: buckets per lane B = ceil((max_span + 64) / 64), up to 8
bool launch_sweep_general_reg(hipStream_t st, bool wide, const uint32_t* boff, const uint32_t* eoff,
                              const void* sorted, const uint32_t* next_head, const uint64_t* d_poff,
                              uint32_t n_contigs, uint32_t span_bits, uint32_t max_span, uint32_t M,
                              uint32_t* selend) {
    const uint32_t b = (max_span + 64 + 63) / 64;
#ifdef QMCP_GEN_STAMP
#define QMCP_GEN_STAMP_ARG , (unsigned long long*)nullptr
#else
#define QMCP_GEN_STAMP_ARG
#endif
#define QMCP_GEN_REG(BB)                                                                              \
    if (wide)                                                                                          \
        hipLaunchKernelGGL((k_sweep_general_reg<SortedK64, BB>), dim3(n_contigs), dim3(64), 0, st, boff,  \
                           eoff, SortedK64{(const uint64_t*)sorted}, next_head, d_poff, span_bits,    \
                           max_span, M, selend QMCP_GEN_STAMP_ARG);                                    \
    else                                                                                               \
        hipLaunchKernelGGL((k_sweep_general_reg<SortedRec, BB>), dim3(n_contigs), dim3(64), 0, st, boff,  \
                           eoff, SortedRec{(const Rec*)sorted}, next_head, d_poff, span_bits, max_span, \
                           M, selend QMCP_GEN_STAMP_ARG);
    if (b <= 2) { QMCP_GEN_REG(2) }
    else if (b == 3) { QMCP_GEN_REG(3) }
    else if (b == 4) { QMCP_GEN_REG(4) }
    else if (b <= 6) { QMCP_GEN_REG(6) }
    else if (b <= 8) { QMCP_GEN_REG(8) }
    else return false;
#undef QMCP_GEN_REG
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

void launch_digit_bases(hipStream_t st, const uint32_t* hist4, uint32_t* base4) {
    hipLaunchKernelGGL(k_digit_bases, dim3(1), dim3(256), 0, st, hist4, base4);
}

void launch_radix_onesweep(hipStream_t st, bool first, const uint32_t* keys, const void* recs_in,
                           uint32_t n, uint32_t shift, const uint32_t* digit_base,
                           unsigned long long* status, uint32_t epoch, uint32_t* ticket,
                           uint32_t* timeout_flag, void* recs_out) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    if (first)
        hipLaunchKernelGGL(k_radix_onesweep<true>, dim3(n_tiles), dim3(kSortThreads), 0, st, keys,
                           (const Rec*)recs_in, n, shift, n_tiles, digit_base, status, epoch, ticket,
                           timeout_flag, (Rec*)recs_out);
    else
        hipLaunchKernelGGL(k_radix_onesweep<false>, dim3(n_tiles), dim3(kSortThreads), 0, st, keys,
                           (const Rec*)recs_in, n, shift, n_tiles, digit_base, status, epoch, ticket,
                           timeout_flag, (Rec*)recs_out);
}

// range-ranked uniform path: geometry, partition table, counts, rank + mark
uint32_t range_shift_for(uint32_t ltot) {
    // smallest shift whose ranges (positions 0..ltot inclusive) fit the 256 digits of one pass; beyond
    // 256 ranges of 32 Ki positions a second partition level supplies eight more digit bits
    uint32_t shift = 0;
    while (shift < kMaxRangeShift && (ltot >> shift) >= 256u) ++shift;
    return shift;
}
bool range_path_two_level(uint32_t ltot) { return (ltot >> kMaxRangeShift) >= 256u; }
bool range_path_supported(uint32_t ltot) { return (ltot >> kMaxRangeShift) < 65536u; }  // always, for 32-bit positions < 2^31

template <int MODE, bool OUT_REC>
static void launch_partition_t(hipStream_t st, dim3 grid, const uint32_t* keys, const Rec* recs_in,
                               SegTables seg, const uint64_t* d_roff, const uint64_t* d_poff,
                               uint32_t n_contigs, uint32_t n, uint32_t shift, uint32_t n_tiles,
                               const uint32_t* offs, uint16_t* k16, uint32_t* idx, Rec* out_rec,
                               uint32_t* range_start, uint32_t* max_load) {
    (void)hipFuncSetAttribute((const void*)k_range_partition<MODE, OUT_REC>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartLds);
    hipLaunchKernelGGL((k_range_partition<MODE, OUT_REC>), grid, dim3(kPartThreads), kPartLds, st, keys,
                       recs_in, seg, d_roff, d_poff, n_contigs, n, shift, n_tiles, offs, k16, idx, out_rec,
                       range_start, max_load);
}

void launch_range_partition(hipStream_t st, const uint32_t* gstart_or_null, const uint32_t* starts,
                            const uint64_t* d_roff, const uint64_t* d_poff, uint32_t n_contigs,
                            uint32_t n, uint32_t shift, const uint32_t* offs, uint16_t* keys16_out,
                            uint32_t* idx_out, uint32_t* range_start, uint32_t* max_load) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const dim3 grid((n_tiles + kPartTiles - 1) / kPartTiles);
    const SegTables none{nullptr, nullptr, nullptr};
    if (gstart_or_null)
        launch_partition_t<0, false>(st, grid, gstart_or_null, nullptr, none, d_roff, d_poff, n_contigs, n, shift,
                                     n_tiles, offs, keys16_out, idx_out, nullptr, range_start, max_load);
    else
        launch_partition_t<1, false>(st, grid, starts, nullptr, none, d_roff, d_poff, n_contigs, n, shift,
                                     n_tiles, offs, keys16_out, idx_out, nullptr, range_start, max_load);
}

// Two-level route.  Level 1: stable partition of the reads into <= 256 super-ranges of 2^(shift+8)
// positions, as {global start, index} records; its first workgroup publishes super_start[257].
void launch_partition_level1(hipStream_t st, const uint32_t* starts, const uint64_t* d_roff,
                             const uint64_t* d_poff, uint32_t n_contigs, uint32_t n, uint32_t shift_hi,
                             const uint32_t* offs, void* recs_out, uint32_t* super_start,
                             uint32_t* max_super_load) {
    const uint32_t n_tiles = sort_tiles(n);
    if (n_tiles == 0) return;
    const dim3 grid((n_tiles + kPartTiles - 1) / kPartTiles);
    const SegTables none{nullptr, nullptr, nullptr};
    launch_partition_t<1, true>(st, grid, starts, nullptr, none, d_roff, d_poff, n_contigs, n, shift_hi, n_tiles,
                                offs, nullptr, nullptr, (Rec*)recs_out, super_start, max_super_load);
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
void launch_range_offsets(hipStream_t st, const uint16_t* keys16, const uint32_t* range_start,
                          uint32_t shift, uint32_t ltot, uint32_t* boff) {
    const uint32_t n_ranges = (ltot >> shift) + 1;  // covers positions 0..ltot
    const size_t lds = (((size_t)1 << shift) + ((size_t)1 << shift) / 32 + 1) * sizeof(uint32_t);
    (void)hipFuncSetAttribute((const void*)k_range_offsets, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL(k_range_offsets, dim3(n_ranges), dim3(1024), lds, st, keys16, range_start, shift,
                       ltot, boff);
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

void launch_complete_pairs(hipStream_t st, uint64_t* mask, uint32_t n_words, uint64_t n_reads) {
    hipLaunchKernelGGL(k_complete_pairs, dim3(grid_for(n_words, 256)), dim3(256), 0, st, mask,
                       n_words, n_reads);
}

void launch_word_popcounts(hipStream_t st, const uint64_t* words, uint32_t n_words, uint32_t* counts) {
    hipLaunchKernelGGL(k_word_popcounts, dim3(grid_for(n_words, 256)), dim3(256), 0, st, words, n_words,
                       counts);
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

}  // namespace qmcp
