// What write pattern gets the most out of the HBM (buffers larger than the Infinity Cache)?  hipMemsetAsync reaches
// 6.6 TB/s where wave-owned row blocks reach 5.5-5.8.  Variants of "who writes which 1 920-byte row when".  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }

// grid-stride fill, 16 B per thread per iteration (what a memset kernel does)
__global__ void __launch_bounds__(256) fill_gridstride(uint4* __restrict__ out, size_t n16, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) st(out + i, w);
}
// persistent waves, rows interleaved over the resident waves: wave w writes rows w, w + n_waves, ... (a global front)
__global__ void __launch_bounds__(256) fill_rows_front(uint4* __restrict__ out, uint32_t n_rows, uint32_t chunks, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t r = wave; r < n_rows; r += n_waves) {
        uint4* p = out + (size_t)r * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) st(p + c, w);
    }
}
// persistent waves, each owning a contiguous block of rows (n_rows / n_waves rows)
__global__ void __launch_bounds__(256) fill_rows_owned(uint4* __restrict__ out, uint32_t n_rows, uint32_t chunks, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t per = n_rows / n_waves;
    for (uint32_t k = 0; k < per; k++) {
        uint4* p = out + ((size_t)wave * per + k) * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) st(p + c, w);
    }
}
// one launch of n_rows / 16 waves, wave owns 16 rows (the step kernel's shape), XCD-contiguous blocks
__global__ void __launch_bounds__(256) fill_rows_step(uint4* __restrict__ out, uint32_t chunks, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * 16 + k) * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) st(p + c, w);
    }
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    const uint32_t n_rows = 262144, chunks = 120;
    const size_t bytes = (size_t)n_rows * chunks * 16;
    uint4* buf; if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
    auto bench = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 5; i++) launch();
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < 30; i++) launch();
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-70s %8.2f us  %6.0f GB/s\n", name, ms / 30 * 1e3, bytes / (ms / 30 * 1e-3) / 1e9); fflush(stdout);
    };
    char name[128];
    for (int rep = 0; rep < 2; rep++) {
        bench("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, bytes, s); });
        for (uint32_t wgs : {256u, 512u, 1024u, 2048u, 4096u, 16384u}) {
            snprintf(name, sizeof name, "grid-stride 16 B/thread, %u workgroups", wgs);
            bench(name, [&] { hipLaunchKernelGGL(fill_gridstride, dim3(wgs), dim3(256), 0, s, buf, bytes / 16, v); });
        }
        for (uint32_t wgs : {256u, 512u, 1024u, 2048u}) {
            snprintf(name, sizeof name, "rows interleaved over %u persistent waves (global front)", wgs * 4);
            bench(name, [&] { hipLaunchKernelGGL(fill_rows_front, dim3(wgs), dim3(256), 0, s, buf, n_rows, chunks, v); });
            snprintf(name, sizeof name, "rows owned in blocks by %u persistent waves", wgs * 4);
            bench(name, [&] { hipLaunchKernelGGL(fill_rows_owned, dim3(wgs), dim3(256), 0, s, buf, n_rows, chunks, v); });
        }
        bench("step-kernel shape: 16384 waves x 16 rows, XCD-contiguous", [&] { hipLaunchKernelGGL(fill_rows_step, dim3(n_rows / 64), dim3(256), 0, s, buf, chunks, v); });
    }
    return 0;
}
