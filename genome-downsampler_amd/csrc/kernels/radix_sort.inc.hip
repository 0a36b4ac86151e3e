// radix_sort.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
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
        // broadcast read of the wave's running count, then the group's first lane moves it on (a wave's
        // LDS operations execute in order)
        const uint32_t old = s_cnt[w][d];
        if (valid && lane == leader) s_cnt[w][d] = old + __popcll(peers);
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
            // broadcast read of the wave's running count, then the group's first lane moves it on (a wave's
            // LDS operations execute in order)
            const uint32_t old = s_cnt[w][d];
            if (valid && lane == leader) s_cnt[w][d] = old + __popcll(peers);
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
