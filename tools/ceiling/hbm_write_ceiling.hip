// Write ceiling of the HBM itself (buffers larger than the 256 MB Infinity Cache) for the shapes the step kernel can
// choose from: store policy x which wave writes which row x row pitch.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_write_ceiling hbm_write_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// POLICY 0 plain, 1 sc1, 2 nt, 3 sc1 nt, 4 sc0 sc1
template <int POLICY>
__device__ __forceinline__ void st16(uint4* p, const u32x4& w) {
    if constexpr (POLICY == 0) *p = make_uint4(w.x, w.y, w.z, w.w);
    else if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w));
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(w));
    else if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(w));
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(w));
}

// ORDER 0: wave w owns rows [w*epw, (w+1)*epw) (the step kernel's shape)
// ORDER 1: the 4 waves of a workgroup interleave: workgroup owns 4*epw consecutive rows, wave i takes rows i, i+4, ...
// ORDER 2: rows interleaved over ALL waves of the grid: wave w takes rows w, w + n_waves, ...
// ORDER 3: like 0, but workgroup b -> block ((b % 8) * (grid / 8) + b / 8): the workgroups of one XCD write one eighth of the buffer
template <int POLICY, int ORDER>
__global__ void __launch_bounds__(256) fill_rows(uint4* __restrict__ out, uint32_t epw, uint32_t chunks, uint32_t pitch, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    uint32_t blk = blockIdx.x;
    if (ORDER == 3) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t wave = blk * 4 + wiw, n_waves = gridDim.x * 4;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < epw; k++) {
        size_t row;
        if (ORDER == 1) row = (size_t)blk * 4 * epw + (size_t)k * 4 + wiw;
        else if (ORDER == 2) row = (size_t)k * n_waves + wave;
        else row = (size_t)wave * epw + k;
        uint4* p = out + row * pitch;
        for (uint32_t c = lane; c < chunks; c += 64) st16<POLICY>(p + c, w);
    }
}

// ALIGNED: the wave's epw rows are one contiguous region of epw * pitch chunks (a whole number of KiB when epw * pitch
// is a multiple of 64): written as full, 1-KiB aligned wave stores that straddle row boundaries.
// XCD = 1: workgroup b -> block (b % 8) * (grid / 8) + b / 8.
template <int POLICY, int XCD>
__global__ void __launch_bounds__(256) fill_region(uint4* __restrict__ out, uint32_t epw, uint32_t pitch, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    uint32_t blk = blockIdx.x;
    if (XCD) blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t wave = blk * 4 + wiw;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    uint4* p = out + (size_t)wave * epw * pitch;
    const uint32_t total = epw * pitch;
    for (uint32_t c = lane; c < total; c += 64) st16<POLICY>(p + c, w);
}

int main(int argc, char** argv) {
    const uint32_t envs_list[] = {65536u, 131072u, 262144u, 524288u};
    uint4* big;
    if (hipMalloc(&big, (size_t)524288 * 2048 + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40;
    uint4 v = {1, 2, 3, 4};
    auto bench = [&](const char* name, size_t bytes, auto&& launch) {
        for (int i = 0; i < 5; i++) launch();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; i++) launch();
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-86s %8.2f us  %6.0f GB/s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9);
        fflush(stdout);
    };
    char name[160];
    static const char* pol[] = {"plain", "sc1", "nt", "sc1 nt", "sc0 sc1"};
    static const char* ord[] = {"wave-contiguous", "interleaved in WG", "interleaved over grid", "XCD-contiguous"};
#define RUN(P, O, envs, chunks, pitch)                                                                                          \
    do {                                                                                                                        \
        snprintf(name, sizeof name, "%u envs, %u B written of pitch %u B, %-7s %s", envs, chunks * 16, pitch * 16, pol[P], ord[O]); \
        bench(name, (size_t)envs * chunks * 16, [&] { hipLaunchKernelGGL((fill_rows<P, O>), dim3(envs / 64), dim3(256), 0, st, big, 16u, chunks, pitch, v); }); \
    } while (0)
    for (uint32_t envs : envs_list) {
        // packed rows (117 chunks at pitch 117), padded rows with the padding unwritten (117 of 120) and written (120 of 120)
        RUN(0, 0, envs, 117u, 117u); RUN(1, 0, envs, 117u, 117u);
        RUN(0, 0, envs, 117u, 120u); RUN(1, 0, envs, 117u, 120u);
        RUN(0, 0, envs, 120u, 120u); RUN(1, 0, envs, 120u, 120u); RUN(2, 0, envs, 120u, 120u); RUN(3, 0, envs, 120u, 120u); RUN(4, 0, envs, 120u, 120u);
        RUN(0, 1, envs, 120u, 120u); RUN(1, 1, envs, 120u, 120u);
        RUN(0, 2, envs, 120u, 120u); RUN(1, 2, envs, 120u, 120u);
        RUN(0, 3, envs, 120u, 120u); RUN(1, 3, envs, 120u, 120u);
        RUN(0, 0, envs, 128u, 128u); RUN(1, 0, envs, 128u, 128u);
#define RUNR(P, X, envs, pitch)                                                                                                 \
    do {                                                                                                                        \
        snprintf(name, sizeof name, "%u envs, pitch %u B, %-7s aligned 1-KiB stores over the wave's 16 rows%s", envs, pitch * 16, pol[P], X ? ", XCD-contiguous" : ""); \
        bench(name, (size_t)envs * pitch * 16, [&] { hipLaunchKernelGGL((fill_region<P, X>), dim3(envs / 64), dim3(256), 0, st, big, 16u, pitch, v); }); \
    } while (0)
        RUNR(0, 0, envs, 120u); RUNR(1, 0, envs, 120u); RUNR(0, 1, envs, 120u); RUNR(1, 1, envs, 120u);
        snprintf(name, sizeof name, "%u envs x 1920 B: hipMemsetAsync", envs);
        bench(name, (size_t)envs * 1920, [&] { hipMemsetAsync(big, 1, (size_t)envs * 1920, st); });
    }
    return 0;
}
