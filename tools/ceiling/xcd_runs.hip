// How should workgroups be assigned to blocks of rows?  Workgroup b runs on XCD b % 8.  RUN = r: XCD x serves runs of r
// consecutive blocks, the runs of the eight XCDs interleaved round-robin (r = 1: dispatch order; r = n/8: each XCD one
// contiguous eighth).  Several separately allocated buffers: the result depends on where a buffer lands.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) fill(uint4* __restrict__ out, uint32_t rows_per_wave, uint32_t chunks, uint32_t run, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x, x = b & 7u, i = b >> 3;       // i-th workgroup of XCD x
    uint32_t blk = (i / run) * 8u * run + x * run + i % run;
    if (blk >= gridDim.x) blk = b;                                // (ragged tail: identity; grids here are multiples of 8 * run)
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t wave = blk * 4 + wiw;
    for (uint32_t k = 0; k < rows_per_wave; k++) {
        uint4* p = out + ((size_t)wave * rows_per_wave + k) * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + c), "v"(w));
    }
}
int main() {
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    struct Shape { const char* name; uint32_t envs, chunks, rows_per_wave; } shapes[] = {
        {"cfg5 65536 x 20480 B (8 rows per wave)", 65536, 1280, 8}, {"level6 262144 x 1920 B (16 rows per wave)", 262144, 120, 16}};
    for (auto& s : shapes) {
        const size_t bytes = (size_t)s.envs * s.chunks * 16;
        uint4* bufs[3];
        for (auto& b : bufs) if (hipMalloc(&b, bytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
        const uint32_t grid = s.envs / (4 * s.rows_per_wave);
        for (uint32_t run : {1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u, grid / 8}) {
            printf("%s, run %5u:", s.name, run);
            for (int rep = 0; rep < 2; rep++)
                for (auto b : bufs) {
                    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, st, b, s.rows_per_wave, s.chunks, run, v);
                    (void)hipStreamSynchronize(st);
                    (void)hipEventRecord(e0, st);
                    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, st, b, s.rows_per_wave, s.chunks, run, v);
                    (void)hipEventRecord(e1, st);
                    (void)hipStreamSynchronize(st);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf("  %7.1f us (%4.0f GB/s)", ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
                }
            printf("\n"); fflush(stdout);
        }
        for (auto b : bufs) (void)hipFree(b);
    }
    return 0;
}
