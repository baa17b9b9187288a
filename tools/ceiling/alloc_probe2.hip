// Round 4, second step: physically CONTIGUOUS buffers are reproducibly slow for the step kernel's row stream (alloc_probe: 5.9-6.0 TB/s
// on every hipDeviceMallocContiguous buffer, 6.5-6.6 on most hipMalloc ones, hipMemsetAsync 6.6 / 6.8) -- so the rate is a
// function of the physical address pattern of the concurrent write streams, and a contiguous buffer is a reproducible test bed.
// Which orderings of the SAME bytes recover the rate there?  All variants write 262 144 rows of 1 920 B with sc1 dwordx4 stores
// unless said otherwise.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n) {
    const uint32_t x = b & 7u, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
// V0: the step kernel's stream.  skew: XCD x walks its share rotated by x * skew blocks (fronts out of phase)
__global__ void __launch_bounds__(256) rows_wave_owned(uint4* __restrict__ out, uint4 v, uint32_t skew, int plain) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x & 7u, q = gridDim.x >> 3;
    uint32_t in = (blockIdx.x >> 3) + x * skew;
    in %= q;
    const uint32_t blk = x * q + in;  // (gridDim.x is a multiple of 8 here)
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * 16 + k) * 120;
        if (plain) { for (uint32_t c = lane; c < 120; c += 64) p[c] = v; }
        else { for (uint32_t c = lane; c < 120; c += 64) st(p + c, w); }
    }
}
// V1: the waves of a workgroup interleaved row by row (wave w writes rows w, w + 4, ...): the workgroup's 120 KB is one front
__global__ void __launch_bounds__(256) rows_wave_interleaved(uint4* __restrict__ out, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < 16; k++) {
        uint4* p = out + ((size_t)blk * 64 + k * 4 + wiw) * 120;
        for (uint32_t c = lane; c < 120; c += 64) st(p + c, w);
    }
}
// V2: the workgroup writes its 120 KB as 30 consecutive 4-KiB blocks, all 256 threads on one block at a time
__global__ void __launch_bounds__(256) wg_sequential(uint4* __restrict__ out, uint4 v) {
    const uint32_t blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    uint4* p = out + (size_t)blk * 7680;
    for (uint32_t c = threadIdx.x; c < 7680; c += 256) st(p + c, w);
}
// V3: E rows per wavefront instead of 16 (grid scales), wave-owned
__global__ void __launch_bounds__(256) rows_wave_owned_e(uint4* __restrict__ out, uint4 v, uint32_t E) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < E; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * E + k) * 120;
        for (uint32_t c = lane; c < 120; c += 64) st(p + c, w);
    }
}
// V4: persistent: gridDim.x workgroups (a multiple of 8), each walks its XCD's share in steps of (workgroups per XCD), wave-owned rows
__global__ void __launch_bounds__(256) rows_persistent(uint4* __restrict__ out, uint4 v, uint32_t n_blocks) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x & 7u, per = gridDim.x >> 3, q = n_blocks >> 3;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t in = blockIdx.x >> 3; in < q; in += per) {
        const uint32_t blk = x * q + in;
        for (uint32_t k = 0; k < 16; k++) {
            uint4* p = out + ((size_t)(blk * 4 + wiw) * 16 + k) * 120;
            for (uint32_t c = lane; c < 120; c += 64) st(p + c, w);
        }
    }
}

static hipStream_t s;
static hipEvent_t e0, e1;
static const size_t ROWS = 262144, BYTES = ROWS * 1920;  // 480 MiB
static double timeit(const std::function<void()>& launch, int reps = 20) {
    for (int i = 0; i < 3; i++) launch();
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return BYTES / (ms / reps * 1e-3) / 1e9;  // GB/s
}

int main(int argc, char** argv) {
    const int n_cont = argc > 1 ? atoi(argv[1]) : 3, n_malloc = argc > 2 ? atoi(argv[2]) : 6;
    (void)hipStreamCreate(&s);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct B { const char* kind; uint4* p; };
    std::vector<B> bufs;
    for (int i = 0; i < n_malloc; i++) {
        uint4* p = nullptr;
        if (hipMalloc(&p, BYTES + (size_t)(i % 3) * (1 << 20)) == hipSuccess) bufs.push_back({"malloc", p});
    }
    for (int i = 0; i < n_cont; i++) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, BYTES + (size_t)(i % 3) * (1 << 20), hipDeviceMallocContiguous) == hipSuccess) bufs.push_back({"contiguous", (uint4*)p});
        else (void)hipGetLastError();
    }
    uint4 v = {1, 2, 3, 4};
    for (int i = 0; i < 400; i++) hipLaunchKernelGGL(rows_wave_owned, dim3(4096), dim3(256), 0, s, bufs[0].p, v, 0u, 0);
    (void)hipStreamSynchronize(s);
    printf("%-12s %3s | %9s %9s %9s %9s | %9s %9s | %9s %9s %9s | %9s %9s %9s | %9s %9s\n", "buffer", "#", "owned", "skew1", "skew37", "skew q/8+1", "interleav", "wg-seq",
           "E=4", "E=8", "E=32", "pers1024", "pers2048", "pers512", "plain", "memset");
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < bufs.size(); i++) {
            uint4* b = bufs[i].p;
            auto owned = [&](uint32_t skew, int plain) { return timeit([&] { hipLaunchKernelGGL(rows_wave_owned, dim3(4096), dim3(256), 0, s, b, v, skew, plain); }); };
            const double a0 = owned(0, 0), a1 = owned(1, 0), a2 = owned(37, 0), a3 = owned(512 / 8 + 1, 0);
            const double b0 = timeit([&] { hipLaunchKernelGGL(rows_wave_interleaved, dim3(4096), dim3(256), 0, s, b, v); });
            const double b1 = timeit([&] { hipLaunchKernelGGL(wg_sequential, dim3(4096), dim3(256), 0, s, b, v); });
            auto e = [&](uint32_t E) { return timeit([&] { hipLaunchKernelGGL(rows_wave_owned_e, dim3(ROWS / (4 * E)), dim3(256), 0, s, b, v, E); }); };
            const double c0 = e(4), c1 = e(8), c2 = e(32);
            auto pers = [&](uint32_t g) { return timeit([&] { hipLaunchKernelGGL(rows_persistent, dim3(g), dim3(256), 0, s, b, v, 4096u); }); };
            const double d0 = pers(1024), d1 = pers(2048), d2 = pers(512);
            const double p0 = owned(0, 1);
            const double m = timeit([&] { (void)hipMemsetAsync(b, 1, BYTES, s); });
            printf("%-12s %3zu | %9.0f %9.0f %9.0f %9.0f | %9.0f %9.0f | %9.0f %9.0f %9.0f | %9.0f %9.0f %9.0f | %9.0f %9.0f\n", bufs[i].kind, i, a0, a1, a2, a3, b0, b1, c0, c1, c2,
                   d0, d1, d2, p0, m);
            fflush(stdout);
        }
    return 0;
}
