// wave_primitives.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
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
