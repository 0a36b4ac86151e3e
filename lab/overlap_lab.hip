// Lab (not part of the product): how much does the sweep slow down beside a bandwidth-bound kernel,
// with and without disjoint CU masks on the two streams?
#define QMCP_MW_STAMP 1
#include "../genome-downsampler_amd/csrc/qmcp_kernels.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace qmcp;

__global__ void k_hog(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i]; v.x += 1; out[i] = v;
    }
}

int main() {
    const uint32_t L = 1000000, ell = 150, M = 100; const int contigs = 8;
    std::mt19937 g(1); std::poisson_distribution<int> pd(12.5);
    const uint64_t Lt = (uint64_t)L * contigs;
    std::vector<uint32_t> boff(Lt + 1, 0);
    for (uint64_t p = 0; p < Lt; ++p) boff[p + 1] = boff[p] + ((p % L) + ell <= L ? pd(g) : 0);
    std::vector<uint64_t> poff(contigs + 1);
    for (int c = 0; c <= contigs; ++c) poff[c] = (uint64_t)c * L;
    uint32_t *d_boff, *d_sel, *d_it; uint64_t* d_poff;
    hipMalloc(&d_boff, (Lt + 1) * 4); hipMalloc(&d_sel, (Lt + 8) * 4); hipMalloc(&d_it, 128); hipMalloc(&d_poff, (contigs + 1) * 8);
    hipMemcpy(d_boff, boff.data(), (Lt + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_poff, poff.data(), (contigs + 1) * 8, hipMemcpyHostToDevice);
    hipMemset(d_it, 0, 128);
    const size_t hog_n = (1ull << 30) / 16;
    uint4 *h_in, *h_out; hipMalloc(&h_in, hog_n * 16); hipMalloc(&h_out, hog_n * 16);
    hipMemset(h_in, 1, hog_n * 16);

    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    printf("CUs: %d\n", ncu);
    // masks: sweep gets every 16th CU (16 CUs), the hog the rest
    std::vector<uint32_t> m_sw((ncu + 31) / 32, 0), m_bw((ncu + 31) / 32, 0);
    for (int i = 0; i < ncu; ++i) { if (i % 16 == 0) m_sw[i / 32] |= 1u << (i % 32); else m_bw[i / 32] |= 1u << (i % 32); }
    hipStream_t s_plain1, s_plain2, s_sw, s_bw;
    hipStreamCreateWithFlags(&s_plain1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s_plain2, hipStreamNonBlocking);
    hipError_t e1 = hipExtStreamCreateWithCUMask(&s_sw, (uint32_t)m_sw.size(), m_sw.data());
    hipError_t e2 = hipExtStreamCreateWithCUMask(&s_bw, (uint32_t)m_bw.size(), m_bw.data());
    printf("cu-mask streams: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    hipEvent_t a, b, ha, hb; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&ha); hipEventCreate(&hb);
    auto run = [&](const char* name, hipStream_t ss, hipStream_t sb, bool with_hog) {
        float best = 1e9, hog_ms = 0;
        for (int it = 0; it < 4; ++it) {
            hipDeviceSynchronize();
            if (with_hog) { hipEventRecord(ha, sb); for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(k_hog, dim3(4096), dim3(256), 0, sb, h_in, h_out, hog_n); hipEventRecord(hb, sb); }
            hipEventRecord(a, ss);
            launch_sweep_uniform_mw(ss, d_boff, d_poff, contigs, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr, 0);
            hipEventRecord(b, ss);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            if (with_hog) hipEventElapsedTime(&hog_ms, ha, hb);
        }
        printf("%-44s sweep %.3f ms   hog(6 x 2 GiB moved) %.3f ms\n", name, best, hog_ms);
        uint32_t st[32]; hipMemcpy(st, d_it, 128, hipMemcpyDeviceToHost); hipMemset(d_it, 0, 128);
        const double stages = 833.0 * 4 * contigs;
        const char* names[7] = {"prep0", "prep1", "prep2", "chain", "prep3", "checkA", "checkB"};
        printf("    per stage (work/wait):");
        for (int wv = 0; wv < 7; ++wv) printf(" %s %.0f/%.0f", names[wv], 16.0 * st[4 + 2 * wv] / stages, 16.0 * st[5 + 2 * wv] / stages);
        printf("\n");
    };
    run("sweep alone", s_plain1, s_plain2, false);
    run("sweep + hog, plain streams", s_plain1, s_plain2, true);
    if (e1 == hipSuccess && e2 == hipSuccess) {
        run("sweep alone on masked stream", s_sw, s_bw, false);
        run("sweep + hog, disjoint CU masks", s_sw, s_bw, true);
        run("hog alone on masked stream (no sweep)", s_plain1, s_bw, true);
    }
    return 0;
}
