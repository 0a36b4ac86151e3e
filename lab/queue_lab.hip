// lab: how many HIP streams of one process run kernels side by side?  N streams, one long one-workgroup kernel
// each (a wave spinning on s_memrealtime for ~1 ms): wall time of the batch / 1 ms = serialisation factor.
// Run with and without GPU_MAX_HW_QUEUES=<n> in the environment.
//   hipcc --offload-arch=gfx950 -O3 lab/queue_lab.hip -o lab/queue_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long ticks, unsigned* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) ++n;  // 100 MHz counter
    if (threadIdx.x == 0) out[blockIdx.x] = n;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096);
    for (int n_streams : {1, 2, 3, 4, 5, 6, 8, 12, 16}) {
        std::vector<hipStream_t> st(n_streams);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (auto& s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 100000ull, d);
            hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%2d streams: %.2f ms\n", n_streams, ms);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
