// Is the write rate of a buffer larger than the Infinity Cache a property of the ALLOCATION?
// Allocates several buffers side by side and times the same fills on each, twice:
//   * config 5's shape: 65 536 rows of 20 480 B, a workgroup of 4 waves owns 32 rows, wave w writes chunks
//     [w*320, (w+1)*320) of each (the split-row stream of step_kernel), XCD-contiguous blocks, sc1 stores;
//   * level 6's shape at 262 144 rows of 1 920 B, 16 rows per wave;
//   * hipMemsetAsync of the same bytes.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n) {
    const uint32_t x = b & 7u, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
__global__ void __launch_bounds__(256) fill_split(uint4* __restrict__ out, uint4 v) {  // grid = rows / 32
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 32; k++) {
        uint4* p = out + ((size_t)blk * 32 + k) * 1280 + wiw * 320;
        for (uint32_t c = lane; c < 320; c += 64) st(p + c, w);
    }
}
__global__ void __launch_bounds__(256) fill_rows16(uint4* __restrict__ out, uint4 v) {  // grid = rows / 64
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * 16 + k) * 120;
        for (uint32_t c = lane; c < 120; c += 64) st(p + c, w);
    }
}
int main(int argc, char** argv) {
    const int n_buf = argc > 1 ? atoi(argv[1]) : 8;
    const size_t big = (size_t)65536 * 20480, small = (size_t)262144 * 1920;
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<uint4*> bufs;
    for (int i = 0; i < n_buf; i++) {
        uint4* p = nullptr;
        if (hipMalloc(&p, big + (i % 3) * (1 << 20)) != hipSuccess) break;  // (sizes differ a little: no two allocations alike)
        bufs.push_back(p);
    }
    uint4 v = {1, 2, 3, 4};
    auto timeit = [&](auto&& launch) {
        for (int i = 0; i < 3; i++) launch();
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < 20; i++) launch();
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 20 * 1e3;
    };
    // clocks up
    for (int i = 0; i < 400; i++) hipLaunchKernelGGL(fill_rows16, dim3(262144 / 64), dim3(256), 0, s, bufs[0], v);
    (void)hipStreamSynchronize(s);
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < bufs.size(); i++) {
            uint4* b = bufs[i];
            const double a = timeit([&] { hipLaunchKernelGGL(fill_split, dim3(65536 / 32), dim3(256), 0, s, b, v); });
            const double c = timeit([&] { hipLaunchKernelGGL(fill_rows16, dim3(262144 / 64), dim3(256), 0, s, b, v); });
            const double m = timeit([&] { (void)hipMemsetAsync(b, 1, big, s); });
            const double m2 = timeit([&] { (void)hipMemsetAsync(b, 1, small, s); });
            printf("pass %d buffer %zu at %p: config-5 shape %7.1f us (%4.0f GB/s) | level-6 x 262144 shape %6.1f us (%4.0f GB/s) | memset 1.34 GB %6.1f us (%4.0f GB/s), 503 MB %5.1f us (%4.0f GB/s)\n",
                   pass, i, (void*)b, a, big / a / 1e3, c, small / c / 1e3, m, big / m / 1e3, m2, small / m2 / 1e3);
            fflush(stdout);
        }
    return 0;
}
