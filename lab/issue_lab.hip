// Micro-benchmarks (not part of the product): what one lone wave64 pays per instruction on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, unsigned long long* ticks, int iters) {
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11;
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 64 dependent v_min_u32 (VOP2)
            REP64(asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (MODE == 1) {  // 2 independent chains, 32 each
            REP16(asm volatile("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));)
            REP16(asm volatile("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2" : "+v"(a), "+v"(c) : "v"(b));)
        } else if (MODE == 2) {  // 64 dependent DPP mins (row_shr:1), compiler-free
            REP64(asm volatile("s_nop 1\n v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
        } else if (MODE == 3) {  // two interleaved dependent DPP chains, 32 each
            REP16(asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c));)
            REP16(asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c));)
        } else if (MODE == 4) {  // 64 v_readlane + dependent valu use
            REP64(asm volatile("v_readlane_b32 s20, %0, 5\n s_nop 3\n v_min_u32 %0, s20, %0" : "+v"(a) : : "s20");)
        } else if (MODE == 5) {  // 64 dependent VOP3 min3
            REP64(asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(d));)
        } else if (MODE == 6) {  // LDS round trip: write + read dependent
            REP16(asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(threadIdx.x * 4));)
        } else if (MODE == 7) {  // 4 independent chains
            REP16(asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(b) : "v"(threadIdx.x));)
        } else if (MODE == 8) {  // s_barrier alone (single wave)
            REP16(asm volatile("s_barrier");)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a + b + c + d;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int per_iter) {
    uint32_t* out; unsigned long long* t;
    hipMalloc(&out, 256); hipMalloc(&t, 8);
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, t, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, t, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long ticks; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-44s %.2f ticks/op  %.3f ns/op  (tick = %.3f ns)\n", name, ticks / n, ms * 1e6 / n, ms * 1e6 / ticks);
}

int main() {
    run<0>("dependent v_min_u32 (VOP2)", 64);
    run<1>("2 independent v_min chains", 64);
    run<7>("4 independent v_min chains", 64);
    run<5>("dependent v_min3_u32 (VOP3)", 64);
    run<2>("dependent v_min_u32_dpp (+s_nop 1)", 64);
    run<3>("2 interleaved dependent dpp chains", 64);
    run<4>("readlane -> s_nop 3 -> valu", 64);
    run<6>("lds write+read round trip", 16);
    run<8>("s_barrier (lone wave)", 16);
    return 0;
}
