// radix_chained.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
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
        // broadcast read of the wave's running count, then the group's first lane moves it on (a wave's
        // LDS operations execute in order)
        const uint32_t old = s_cnt[w][d];
        if (valid && lane == leader) s_cnt[w][d] = old + __popcll(peers);
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
