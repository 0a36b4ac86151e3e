// Timing lab for the uniform sweep (not part of the product): runs the production kernel on
// synthetic bucket offsets.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<csrc>
#define QMCP_MW_STAMP 1
#include "../genome-downsampler_amd/csrc/qmcp_kernels.hip"
#include <cstdio>
#include <random>
#include <vector>
using namespace qmcp;

static void run_case(const char* name, double mean, uint32_t L, uint32_t ell, uint32_t M, int contigs) {
    std::mt19937 g(1);
    std::poisson_distribution<int> pd(mean);
    const uint64_t Lt = (uint64_t)L * contigs;
    std::vector<uint32_t> boff(Lt + 1, 0);
    for (uint64_t p = 0; p < Lt; ++p) boff[p + 1] = boff[p] + ((p % L) + ell <= L ? pd(g) : 0);
    std::vector<uint64_t> poff(contigs + 1);
    for (int c = 0; c <= contigs; ++c) poff[c] = (uint64_t)c * L;
    uint32_t *d_boff, *d_sel, *d_it; uint64_t* d_poff;
    hipMalloc(&d_boff, (Lt + 1) * 4); hipMalloc(&d_sel, (Lt + 8) * 4); hipMalloc(&d_it, 128);
    hipMalloc(&d_poff, (contigs + 1) * 8);
    hipMemcpy(d_boff, boff.data(), (Lt + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_poff, poff.data(), (contigs + 1) * 8, hipMemcpyHostToDevice);
    hipMemset(d_it, 0, 128);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); float best = 1e9;
    const int reps = 5;
    for (int it = 0; it < reps; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_sweep_uniform<3>), dim3(contigs), dim3(64), 0, 0, d_boff, d_poff, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr);
        hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    uint32_t it2[2]; hipMemcpy(it2, d_it, 8, hipMemcpyDeviceToHost);
    const uint32_t nb = (L + ell - 1) / ell;
    printf("%-34s %.3f ms  %.1f ns/block ~%.0f cyc/block (general-form blocks %u of %u)\n", name, best,
           best * 1e6 / nb, best * 1e6 / nb * 2.4, it2[0] / reps, it2[1] / reps);
    // seven-wave fast pipeline on the same data: must produce identical selend
    {
        std::vector<uint32_t> ref(Lt + 1), got(Lt + 1);
        hipMemcpy(ref.data(), d_sel, (Lt + 1) * 4, hipMemcpyDeviceToHost);
        hipMemset(d_sel, 0xEE, (Lt + 1) * 4);
        hipMemset(d_it, 0, 128);
        float best2 = 1e9;
        bool ok = true;
        static char* d_flush = nullptr;
        if (!d_flush) hipMalloc(&d_flush, 768u << 20);
        float cold = 1e9;
        uint32_t stc[32];
        for (int it = 0; it < reps; ++it) {
            // cold run: evict the table from L2 and the memory-side cache first (what the product sees:
            // the table was just written by other CUs)
            hipMemsetAsync(d_flush, it, 768u << 20, 0);
            hipEventRecord(a);
            launch_sweep_uniform_mw(0, d_boff, d_poff, contigs, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr, 0);
            hipEventRecord(b); hipEventSynchronize(b); float msc; hipEventElapsedTime(&msc, a, b);
            if (msc < cold) cold = msc;
        }
        hipMemcpy(stc, d_it, 128, hipMemcpyDeviceToHost);
        hipMemset(d_it, 0, 128);
        for (int it = 0; it < reps; ++it) {
            hipEventRecord(a);
            ok = launch_sweep_uniform_mw(0, d_boff, d_poff, contigs, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr, 0);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best2) best2 = ms;
        }
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(got.data(), d_sel, (Lt + 1) * 4, hipMemcpyDeviceToHost);
        hipMemcpy(it2, d_it, 8, hipMemcpyDeviceToHost);
        const double stages = (double)(nb / 8) * reps * contigs;
        uint32_t st[32]; hipMemcpy(st, d_it, 128, hipMemcpyDeviceToHost);
        printf("   per stage, cycles (work/wait):");
        const char* names[7] = {"prep0", "prep1", "prep2", "chain", "prep3", "checkA", "checkB"};
        for (int wv = 0; wv < 7; ++wv)
            printf(" %s %.0f/%.0f", names[wv], 16.0 * st[4 + 2 * wv] / stages, 16.0 * st[5 + 2 * wv] / stages);
        printf("\n   same, cold runs:              ");
        for (int wv = 0; wv < 7; ++wv)
            printf(" %s %.0f/%.0f", names[wv], 16.0 * stc[4 + 2 * wv] / stages, 16.0 * stc[5 + 2 * wv] / stages);
        printf("\n   chain wave, cycles per launch per contig: total %.0f  failure handling %.0f in %.1f failures; between stages %.0f; %.0f stage iterations\n", 16.0 * st[22] / (reps * 1.0 * contigs), 16.0 * st[20] / (reps * 1.0 * contigs), st[21] / (reps * 1.0 * contigs), 16.0 * st[23] / (reps * 1.0 * contigs), st[24] / (reps * 1.0 * contigs));
        size_t diff = 0, first = 0;
        for (uint64_t i = 0; i < Lt; ++i) if (ref[i] != got[i]) { if (!diff) first = i; ++diff; }
        printf("   cold (caches flushed): %.3f ms\n", cold);
        printf("   seven-wave: %s %.3f ms  %.1f ns/block ~%.0f cyc/block (general-form %u) mismatches %zu (first at %zu) %s\n",
               ok ? "" : "(unsupported span)", best2, best2 * 1e6 / nb, best2 * 1e6 / nb * 2.4, it2[0] / reps, diff, first,
               hipGetErrorString(e));
    }
    // the pipelined general form on the same data: identical selend required
    {
        std::vector<uint32_t> ref(Lt + 1), got(Lt + 1);
        hipLaunchKernelGGL((k_sweep_uniform<3>), dim3(contigs), dim3(64), 0, 0, d_boff, d_poff, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), d_sel, (Lt + 1) * 4, hipMemcpyDeviceToHost);
        hipMemset(d_sel, 0xEE, (Lt + 1) * 4);
        float best3 = 1e9;
        for (int it = 0; it < reps; ++it) {
            hipEventRecord(a);
            launch_sweep_uniform_gen(0, d_boff, d_poff, contigs, ell, M, (uint32_t)Lt, d_sel, d_it, nullptr, 0);
            hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best3) best3 = ms;
        }
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(got.data(), d_sel, (Lt + 1) * 4, hipMemcpyDeviceToHost);
        size_t diff = 0, first = 0;
        for (uint64_t i = 0; i < Lt; ++i) if (ref[i] != got[i]) { if (!diff) first = i; ++diff; }
        printf("   general pipeline: %.3f ms  %.1f ns/block ~%.0f cyc/block  mismatches %zu (first at %zu) %s\n",
               best3, best3 * 1e6 / nb, best3 * 1e6 / nb * 2.4, diff, first, hipGetErrorString(e));
    }
    hipFree(d_boff); hipFree(d_sel); hipFree(d_it); hipFree(d_poff);
}

int main() {
    run_case("deep 12.5/pos, 1 contig", 12.5, 1000000, 150, 100, 1);
    run_case("deep 12.5/pos, 8 contigs", 12.5, 1000000, 150, 100, 8);
    run_case("sparse 0.67/pos M=50, 1 contig", 0.67, 1000000, 150, 50, 1);
    run_case("shallow 0.3/pos M=50 (cut points)", 0.3, 1000000, 150, 50, 1);
    run_case("mixed: deep with sparse holes", 12.5, 100003, 150, 100, 3);
    run_case("tiny contigs", 5.0, 1000, 150, 20, 5);
    // depth sweep for the fast / general choice: mean coverage = mean * 150, in units of M = 50
    run_case("depth 1.5 x M", 0.5, 400000, 150, 50, 1);
    run_case("depth 3 x M", 1.0, 400000, 150, 50, 1);
    run_case("depth 4.5 x M", 1.5, 400000, 150, 50, 1);
    run_case("depth 6 x M", 2.0, 400000, 150, 50, 1);
    run_case("depth 9 x M", 3.0, 400000, 150, 50, 1);
    run_case("depth 10.5 x M", 3.5, 400000, 150, 50, 1);
    run_case("depth 12 x M", 4.0, 400000, 150, 50, 1);
    run_case("depth 13.5 x M", 4.5, 400000, 150, 50, 1);
    run_case("depth 15 x M", 5.0, 400000, 150, 50, 1);
    return 0;
}
