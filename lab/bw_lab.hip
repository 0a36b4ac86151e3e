// Lab: what does this box give a plain streaming READ (no writes) and a plain copy?  Calibrates the
// "achievable" line next to the 8 TB/s peak for the read-dominated kernels (k_prepare reads 8 B/read).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int W>  // W = 1: dword per lane, 4: dwordx4 per lane
__global__ __launch_bounds__(256) void k_read(const uint32_t* __restrict__ a, size_t n, uint32_t* __restrict__ sink, int unroll_dummy) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (W == 1) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 7 * stride < n; i += 8 * stride) {
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = a[i + k * stride];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
    } else {
        const uint4* a4 = reinterpret_cast<const uint4*>(a);
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n4; i += 4 * stride) {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = a4[i + k * stride];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// tile-ordered read like k_prepare: a workgroup walks 8 consecutive 16 KiB tiles of two arrays
__global__ __launch_bounds__(256) void k_read_tiles(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    const size_t t0 = (size_t)blockIdx.x * 8;
    for (int g = 0; g < 8; ++g) {
        const size_t base = (t0 + g) * 4096;
        if (base + 4096 > n) break;
        uint32_t v[16], w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = a[base + k * 256 + threadIdx.x]; w[k] = b[base + k * 256 + threadIdx.x]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k] ^ w[k];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ o, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) o[i] = a[i];
}

int main() {
    const size_t n = (size_t)200 << 20;  // 200 Mi words = 800 MiB per array
    uint32_t *a, *b, *sink;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 2, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, double bytes, const char* name) {
        float best = 1e9f;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    for (int grid : {2048, 4096, 8192, 16384}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read dword, grid %d", grid);
        time([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, a, n, sink, 0); }, n * 4.0, nm);
        snprintf(nm, sizeof nm, "read dwordx4, grid %d", grid);
        time([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, n, sink, 0); }, n * 4.0, nm);
    }
    time([&] { hipLaunchKernelGGL(k_read_tiles, dim3((unsigned)(n / 4096 / 8)), dim3(256), 0, 0, a, b, n, sink); }, n * 8.0,
         "two arrays, 8 tiles per workgroup (k_prepare)");
    for (int grid : {4096, 16384})
        time([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n / 4); }, n * 8.0,
             grid == 4096 ? "copy dwordx4, grid 4096 (read + write bytes)" : "copy dwordx4, grid 16384 (read + write bytes)");
    return 0;
}
