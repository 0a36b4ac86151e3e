// Micro-benchmark (not part of the product): what does the LAST wave to arrive at a workgroup
// barrier pay, when the other six waves are already waiting?  And what does an LDS flag hand-off cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(448) void k_bar(unsigned long long* out, int iters, int work) {
    __shared__ uint32_t s_data[512];
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long wait = 0, total0 = __builtin_amdgcn_s_memtime();
    uint32_t a = lane;
    for (int i = 0; i < iters; ++i) {
        if (wv == 3) {
            for (int k = 0; k < work; ++k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(lane));
            s_data[lane] = a;  // LDS writes that must drain before the barrier
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (wv == 3) wait += t1 - t0;
    }
    if (threadIdx.x == 192) { out[0] = wait; out[1] = __builtin_amdgcn_s_memtime() - total0; }
    if (a == 0xdeadbeef) out[2] = a;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    for (int work : {0, 100, 300}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k_bar, dim3(1), dim3(448), 0, 0, d, iters, work);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("work %3d v_min per stage: last arriver waits %.0f ticks at the barrier; stage total %.0f ticks\n", work,
               (double)h[0] / iters, (double)h[1] / iters);
    }
    return 0;
}
