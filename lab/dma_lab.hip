// Lab: what does ONE wave pay per 1-KiB piece it streams through an LDS ring by LDS-DMA
// (global_load_lds_dwordx4), as a function of the number of pieces kept in flight and of what the
// wave does with a piece (nothing / one ds_read_b128 + a few VALU)?  And the same stream taken
// through registers (global_load_dwordx4, 8 in flight).  Eight workgroups of one wave, like the
// event-driven sweep's chain kernel (kernels/sweep_uniform_events.inc.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int DEPTH, bool READ>
__global__ __launch_bounds__(64) void k_dma(const uint4* __restrict__ src, uint32_t n_pieces, uint32_t* __restrict__ out) {
    extern __shared__ uint4 ring[];
    const uint32_t lane = threadIdx.x;
    const uint4* my = src + (size_t)blockIdx.x * n_pieces * 64 + lane;
    const uint32_t ring0 = (uint32_t)(uintptr_t)ring;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t i = 0; i < DEPTH; ++i) glds16(my + (size_t)i * 64, ring0 + (i % 64) * 1024u);
    for (uint32_t q = 0; q < n_pieces; ++q) {
        const uint32_t idx = q + DEPTH < n_pieces ? q + DEPTH : n_pieces - 1;
        glds16(my + (size_t)idx * 64, ring0 + ((q + DEPTH) % 64) * 1024u);
        if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if (DEPTH == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (READ) {
            const uint4 w = ring[(q % 64) * 64 + lane];
            acc += (w.x - acc) & (w.y ^ w.z) & w.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[2 * blockIdx.x] = (uint32_t)(t1 - t0); out[2 * blockIdx.x + 1] = acc; }
}

// the same stream through registers: eight 16-byte loads in flight per lane
__global__ __launch_bounds__(64) void k_reg(const uint4* __restrict__ src, uint32_t n_pieces, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const uint4* my = src + (size_t)blockIdx.x * n_pieces * 64 + lane;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint4 r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = my[(size_t)i * 64];
    for (uint32_t q = 0; q + 8 <= n_pieces; q += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 w = r[i];
            const uint32_t idx = q + 8 + i < n_pieces ? q + 8 + i : n_pieces - 1;
            r[i] = my[(size_t)idx * 64];
            acc += (w.x - acc) & (w.y ^ w.z) & w.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[2 * blockIdx.x] = (uint32_t)(t1 - t0); out[2 * blockIdx.x + 1] = acc; }
}

int main() {
    const uint32_t n_pieces = 1667, wgs = 8;
    const size_t bytes = (size_t)wgs * n_pieces * 1024;
    uint4* d = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 64 * 4));
    CK(hipMemset(d, 1, bytes));
    uint32_t* flush = nullptr; const size_t fb = 512ull << 20;
    CK(hipMalloc(&flush, fb));
    auto report = [&](const char* name) {
        uint32_t h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 8; ++i) s += h[2 * i];
        printf("%-34s %8.1f cycles per piece (mean of 8 waves)\n", name, s / 8 / n_pieces);
    };
#define RUN(D, R, cold)                                                                       \
    for (int rep = 0; rep < 2; ++rep) {                                                        \
        if (cold) CK(hipMemset(flush, rep, fb));                                               \
        hipLaunchKernelGGL((k_dma<D, R>), dim3(wgs), dim3(64), 64 * 1024, 0, d, n_pieces, out); \
        CK(hipDeviceSynchronize());                                                            \
    }                                                                                          \
    report("dma depth " #D " read " #R " cold " #cold);
    RUN(4, false, 0) RUN(16, false, 0) RUN(32, false, 0) RUN(48, false, 0)
    RUN(4, true, 0) RUN(16, true, 0) RUN(32, true, 0) RUN(48, true, 0)
    RUN(16, true, 1) RUN(48, true, 1)
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_reg, dim3(wgs), dim3(64), 0, 0, d, n_pieces, out); CK(hipDeviceSynchronize()); }
    report("registers, 8 loads in flight");
    CK(hipMemset(flush, 3, fb));
    hipLaunchKernelGGL(k_reg, dim3(wgs), dim3(64), 0, 0, d, n_pieces, out); CK(hipDeviceSynchronize());
    report("registers, 8 in flight, cold");
    return 0;
}
