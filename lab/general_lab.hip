// Lab (not part of the product): where does the register-resident mixed-span sweep spend its time?
// Builds sorted composite keys for synthetic reads on the host and runs the kernel with cycle stamps.
#define QMCP_GEN_STAMP 1
#include "../genome-downsampler_amd/csrc/qmcp_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
using namespace qmcp;

int main() {
    const uint32_t L = 30000, n = 1000000, lo = 100, hi = 150, M = 100;
    std::mt19937 g(5);
    std::vector<uint64_t> keys(n);
    const uint32_t span_bits = 6, max_span = hi;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t span = lo + g() % (hi - lo + 1);
        const uint32_t s = g() % (L - span + 1);
        keys[i] = ((uint64_t)s << span_bits) | (max_span - span);
    }
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> boff(L + 1, 0), ecnt(L + 2, 0), eoff(L + 1, 0), next_head(n + 2, n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t s = (uint32_t)(keys[i] >> span_bits), span = max_span - (uint32_t)(keys[i] & 63);
        boff[s + 1]++; ecnt[s + span]++;  // ends at s + span - 1: stops covering at s + span
    }
    for (uint32_t p = 0; p < L; ++p) boff[p + 1] += boff[p];
    for (uint32_t p = 0; p < L; ++p) eoff[p + 1] = eoff[p] + ecnt[p + 1];  // reads with end < p + 1 ... as k_general_keys scans them
    // eoff[p] = reads with end < p
    std::vector<uint32_t> e2(L + 1, 0);
    { std::vector<uint32_t> c(L + 1, 0); for (uint32_t i = 0; i < n; ++i) { const uint32_t s = (uint32_t)(keys[i] >> span_bits), span = max_span - (uint32_t)(keys[i] & 63); c[s + span - 1]++; }
      uint32_t run = 0; for (uint32_t p = 0; p <= L; ++p) { e2[p] = run; if (p < L) run += c[p]; } }
    for (uint32_t i = n; i-- > 0;) next_head[i] = (i + 1 < n && keys[i + 1] == keys[i]) ? next_head[i + 1] : i + 1;
    // next_head[j] = first index >= j that starts a new (start, end) group ... as k_group_heads + reverse min-scan give it
    std::vector<uint32_t> nh(n + 2, n);
    for (uint32_t i = n; i-- > 1;) nh[i] = (keys[i] != keys[i - 1]) ? i : nh[i + 1];
    nh[0] = 0;
    std::vector<SortedRec> dummy;
    uint64_t* d_keys; uint32_t *d_boff, *d_eoff, *d_nh, *d_sel; uint64_t* d_poff; unsigned long long* d_st;
    hipMalloc(&d_keys, n * 8); hipMalloc(&d_boff, (L + 1) * 4); hipMalloc(&d_eoff, (L + 1) * 4);
    hipMalloc(&d_nh, (n + 2) * 4); hipMalloc(&d_sel, (L + 8) * 4); hipMalloc(&d_poff, 16); hipMalloc(&d_st, 64);
    const uint64_t poff[2] = {0, L};
    hipMemcpy(d_keys, keys.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_boff, boff.data(), (L + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_eoff, e2.data(), (L + 1) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_nh, nh.data(), (n + 2) * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_poff, poff, 16, hipMemcpyHostToDevice);
    hipMemset(d_st, 0, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) {
        hipMemset(d_st, 0, 64);
        hipEventRecord(a);
        hipLaunchKernelGGL((k_sweep_general_reg<SortedK64, 4, 1>), dim3(1), dim3(128), 0, 0, d_boff, d_eoff,
                           SortedK64{d_keys}, d_nh, d_poff, span_bits, max_span, M, d_sel, nullptr, nullptr, nullptr, 0u, nullptr, d_st);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long st[8]; hipMemcpy(st, d_st, 64, hipMemcpyDeviceToHost);
        printf("%.3f ms: entry %.0f cycles/chunk, events %llu (%.0f cycles each), third-group fetches %llu (%.0f cycles each), positions loop rest %.0f cycles/position\n",
               ms, (double)st[0] / ((L + 63) / 64), st[1], st[1] ? (double)st[2] / st[1] : 0.0, st[3], st[3] ? (double)st[4] / st[3] : 0.0,
               ((double)st[5] - (double)st[2]) / L);
    }
    return 0;
}
