// bucket_offsets.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
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
