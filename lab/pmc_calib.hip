// lab: kernels of KNOWN byte counts, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths the solver's kernels use (MI355X_MICROARCH.md gives the x2 FETCH_SIZE correction for 16 B per lane only:
// "other access widths are uncalibrated").  Each kernel moves exactly kBytes (512 MiB) of a 1 GiB buffer, coalesced:
//   calib_read_W    W = 2, 4, 8, 16 bytes per lane and load instruction
//   calib_write_W   W = 2, 4, 16
//   calib_rw_2_4    reads 2 B + 4 B streams the way k_rank_mark does (u16 keys + u32 indices of the same records)
//   calib_atomic_or64   4 Mi 64-bit atomic ORs on random words of a 64 MiB region (k_rank_mark's mask writes)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes); profiles/summarize_pmc.py reads
// the CSVs and writes bytes-per-counter-unit factors.
//   hipcc --offload-arch=gfx950 -O3 lab/pmc_calib.hip -o lab/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
constexpr size_t kBytes = 512ull << 20;
template <typename T> __global__ void calib_read(const T* __restrict__ p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        acc += b[0] + b[sizeof(T) - 1];
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
template <typename T> __global__ void calib_write(T* __restrict__ p, size_t n) {
    T v;
    unsigned char* b = reinterpret_cast<unsigned char*>(&v);
    for (unsigned k = 0; k < sizeof(T); ++k) b[k] = (unsigned char)(threadIdx.x + k);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void calib_rw_2_4(const uint16_t* __restrict__ k, const uint32_t* __restrict__ v, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += k[i] + v[i];
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ void calib_atomic_or64(unsigned long long* __restrict__ m, size_t words, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        atomicOr(&m[h % words], 1ull << (h & 63));
    }
}
template __global__ void calib_read<uint16_t>(const uint16_t*, size_t, uint32_t*);
template __global__ void calib_read<uint32_t>(const uint32_t*, size_t, uint32_t*);
template __global__ void calib_read<uint2>(const uint2*, size_t, uint32_t*);
template __global__ void calib_read<uint4>(const uint4*, size_t, uint32_t*);
template __global__ void calib_write<uint16_t>(uint16_t*, size_t);
template __global__ void calib_write<uint32_t>(uint32_t*, size_t);
template __global__ void calib_write<uint4>(uint4*, size_t);
int main() {
    void* buf; uint32_t* out;
    if (hipMalloc(&buf, 2 * kBytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, 2 * kBytes);
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read<uint16_t>, grid, block, 0, 0, (const uint16_t*)buf, kBytes / 2, out);
        hipLaunchKernelGGL(calib_read<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, kBytes / 4, out);
        hipLaunchKernelGGL(calib_read<uint2>, grid, block, 0, 0, (const uint2*)buf, kBytes / 8, out);
        hipLaunchKernelGGL(calib_read<uint4>, grid, block, 0, 0, (const uint4*)buf, kBytes / 16, out);
        hipLaunchKernelGGL(calib_write<uint16_t>, grid, block, 0, 0, (uint16_t*)buf, kBytes / 2);
        hipLaunchKernelGGL(calib_write<uint32_t>, grid, block, 0, 0, (uint32_t*)buf, kBytes / 4);
        hipLaunchKernelGGL(calib_write<uint4>, grid, block, 0, 0, (uint4*)buf, kBytes / 16);
        // 2 B + 4 B of the same records: kBytes in all = n * 6
        hipLaunchKernelGGL(calib_rw_2_4, grid, block, 0, 0, (const uint16_t*)buf, (const uint32_t*)((char*)buf + kBytes), kBytes / 6, out);
        hipLaunchKernelGGL(calib_atomic_or64, grid, block, 0, 0, (unsigned long long*)buf, (64ull << 20) / 8, (size_t)4 << 20);
        (void)hipDeviceSynchronize();
    }
    printf("pmc_calib: every calib_read_* / calib_write_* kernel moved %zu bytes; calib_rw_2_4 %zu; calib_atomic_or64 %zu atomics\n",
           kBytes, kBytes / 6 * 6, (size_t)4 << 20);
    return 0;
}
