// Where do the waves of a workgroup land?  768 workgroups of 7 waves (the general-form sweep's shape, with
// its LDS footprint): every wave records its XCC, SE, CU, SIMD from the hardware id registers.
//   hipcc --offload-arch=gfx950 -O2 lab/place_lab.hip -o lab/place_lab && lab/place_lab [workgroups] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(448) void k_where(uint32_t* out, int spin) {
    extern __shared__ uint32_t s_pad[];
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));    // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 7 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 7 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    // stay resident for a while so that all workgroups are placed side by side
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) s_pad[0] = 1;
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 768;
    const int lds = argc > 2 ? atoi(argv[2]) : 40000;
    uint32_t* d;
    hipMalloc(&d, (size_t)wgs * 7 * 2 * 4);
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_where, dim3(wgs), dim3(448), lds, 0, d, 2000000);
    std::vector<uint32_t> h((size_t)wgs * 14);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // per (xcc, se, cu): which workgroups, and the SIMD of each one's wave 3
    std::map<uint32_t, std::vector<int>> by_cu;
    int simd_of_wave[7][4] = {};
    for (int b = 0; b < wgs; ++b) {
        for (int w = 0; w < 7; ++w) simd_of_wave[w][(h[(b * 7 + w) * 2] >> 4) & 3]++;
        const uint32_t hw = h[(b * 7 + 3) * 2], xcc = h[(b * 7 + 3) * 2 + 1] & 15;
        const uint32_t cu = (hw >> 8) & 15, se = (hw >> 13) & 7, sh = (hw >> 12) & 1;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b * 4 + ((hw >> 4) & 3));
    }
    printf("compute units used: %zu\n", by_cu.size());
    for (int w = 0; w < 7; ++w) printf("wave %d on SIMD 0..3: %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    int shown = 0, clash = 0;
    for (auto& kv : by_cu) {
        int cnt[4] = {};
        for (int v : kv.second) cnt[v & 3]++;
        for (int i = 0; i < 4; ++i) clash += cnt[i] > 1 ? cnt[i] - 1 : 0;
        if (shown++ < 12) {
            printf("xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int v : kv.second) printf("  wg %4d (wave 3 on SIMD %d)", v >> 2, v & 3);
            printf("\n");
        }
    }
    printf("wave-3 pairs sharing a SIMD: %d\n", clash);
    return 0;
}
