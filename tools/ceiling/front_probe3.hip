// Round 3: does it matter WHICH XCD writes which 4-KiB block?  256 persistent workgroups (one per CU, as a memset-class fill has);
// 4-KiB block j belongs to XCD ((j / R) + s) % 8 (R = run length in blocks, s = shift), the 32 workgroups of an XCD take its blocks
// round robin.  R = 1, s = 0 is the plain grid-stride fill.  Also: the same with `waves` wavefronts per workgroup and with several
// workgroups per CU, to separate "few wavefronts" from "which XCD".  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }

// gridDim.x = 8 * wgs_per_xcd; block of 4 KiB = 256 pieces of 16 B, written by `blockDim.x` threads in 256 / blockDim.x passes
__global__ void __launch_bounds__(1024) fill_xcd_runs(uint4* __restrict__ out, uint32_t n_blocks, uint32_t R, uint32_t s, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t x = blockIdx.x & 7u, q = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const uint32_t col = (x + 8u - (s & 7u)) & 7u;      // the residue (j / R) % 8 this XCD owns
    const uint32_t owned = n_blocks / 8;               // blocks per XCD
    const uint32_t bpp = blockDim.x / 256u;            // 4-KiB blocks per pass when the workgroup is wider than 256 threads
    const uint32_t sub = threadIdx.x >> 8, t = threadIdx.x & 255u;
    for (uint32_t k = q * bpp + sub; k < owned; k += per_xcd * bpp) {
        const uint32_t j = ((k / R) * 8u + col) * R + k % R;
        st(out + (size_t)j * 256u + t, w);
    }
}

// the same, but a workgroup takes B consecutive blocks of its XCD's share at a time (B = 30: the 120 KB of rows a step-kernel workgroup owns)
__global__ void __launch_bounds__(256) fill_xcd_owned(uint4* __restrict__ out, uint32_t n_blocks, uint32_t B, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t x = blockIdx.x & 7u, q = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const uint32_t owned = n_blocks / 8;
    for (uint32_t k0 = q * B; k0 < owned; k0 += per_xcd * B)
        for (uint32_t k = k0; k < k0 + B && k < owned; k++) st(out + ((size_t)x * owned + k) * 256u + threadIdx.x, w);
}
// ... and with each of the workgroup's 4 wavefronts walking its own quarter of those B blocks, 1 KiB at a time (what row-owning wavefronts do)
__global__ void __launch_bounds__(256) fill_xcd_owned_waves(uint4* __restrict__ out, uint32_t n_blocks, uint32_t B, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t x = blockIdx.x & 7u, q = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const uint32_t owned = n_blocks / 8, lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    for (uint32_t k0 = q * B; k0 < owned; k0 += per_xcd * B) {
        uint4* p = out + ((size_t)x * owned + k0) * 256u + (size_t)wiw * B * 64u;  // B blocks = B * 256 pieces; a quarter = B * 64
        for (uint32_t c = lane; c < B * 64u; c += 64) st(p + c, w);
    }
}

// front_probe2's kernel (one group of 16 rows of 1 920 B per wavefront; an LDS preamble and a barrier in front), in two copies: with and
// without the > 64 KiB dynamic-LDS opt-in on the function
extern __shared__ uint32_t dyn_lds[];
template <int COPY>
__global__ void __launch_bounds__(256) fill_groups(uint4* __restrict__ out, uint32_t chunks, uint32_t groups_per_wave, int preamble, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (preamble) {
        for (uint32_t i = threadIdx.x; i < 1024; i += 256) dyn_lds[i] = i * 40503u + v.x;
        __syncthreads();
    }
    u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t per_group = 16 * chunks;
    for (uint32_t k = 0; k < groups_per_wave; k++) {
        uint4* p = out + ((size_t)(blk * 4 + wiw) * groups_per_wave + k) * per_group;
        for (uint32_t c = lane; c < per_group; c += 64) st(p + c, w);
    }
}

int main(int argc, char** argv) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 480) << 20;   // multiple of 8 * 4 KiB * every R below
    const uint32_t n_blocks = (uint32_t)(bytes / 4096);
    uint4* buf; if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
    auto bench = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 5; i++) launch();
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < 30; i++) launch();
        (void)hipEventRecord(e1, s);
        (void)hipStreamSynchronize(s);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (hipGetLastError() != hipSuccess) { printf("%-86s launch failed\n", name); return; }
        printf("%-86s %8.2f us  %6.0f GB/s\n", name, ms / 30 * 1e3, bytes / (ms / 30 * 1e-3) / 1e9); fflush(stdout);
    };
    char name[160];
    printf("%zu MB, %u blocks of 4 KiB\n", bytes >> 20, n_blocks);
    if (argc > 2) {  // many buffers side by side: does the rate of a pattern depend on the allocation?
        const int n_bufs = atoi(argv[2]);
        for (int pass = 0; pass < 2; pass++)
            for (int b = 0; b < n_bufs; b++) {
                static uint4* bufs[64];
                if (!pass && hipMalloc(&bufs[b], bytes + (1 << 20)) != hipSuccess) { printf("buffer %d: no memory\n", b); return 0; }
                uint4* const B = bufs[b];
                printf("-- pass %d buffer %d at %p\n", pass, b, (void*)B);
                bench("  hipMemsetAsync", [&] { (void)hipMemsetAsync(B, 1, bytes, s); });
                bench("  step-kernel shape: 4096 workgroups own 30 blocks each, every wavefront its quarter", [&] { hipLaunchKernelGGL(fill_xcd_owned_waves, dim3(n_blocks / 30u), dim3(256), 0, s, B, n_blocks, 30u, v); });
                bench("  2048 workgroups, XCD-contiguous shares, one block at a time (1 MB front per XCD)", [&] { hipLaunchKernelGGL(fill_xcd_owned, dim3(2048), dim3(256), 0, s, B, n_blocks, 1u, v); });
                bench("  2048 workgroups, XCD-contiguous shares, 8 blocks at a time", [&] { hipLaunchKernelGGL(fill_xcd_owned, dim3(2048), dim3(256), 0, s, B, n_blocks, 8u, v); });
                bench("  256 workgroups grid-stride", [&] { hipLaunchKernelGGL(fill_xcd_runs, dim3(256), dim3(256), 0, s, B, n_blocks, 1u, 0u, v); });
            }
        return 0;
    }
    for (int rep = 0; rep < 2; rep++) {
        bench("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, bytes, s); });
        for (uint32_t R : {1u, 16u, 256u})
            for (uint32_t sh : {0u}) {
                snprintf(name, sizeof name, "256 workgroups x 256 threads, runs of %u blocks per XCD, shift %u", R, sh);
                bench(name, [&] { hipLaunchKernelGGL(fill_xcd_runs, dim3(256), dim3(256), 0, s, buf, n_blocks, R, sh, v); });
            }
        for (uint32_t wgs : {512u, 1024u, 2048u})
            for (uint32_t R : {1u, 16u, 1024u}) {
                snprintf(name, sizeof name, "%u workgroups x 256 threads, runs of %u blocks per XCD", wgs, R);
                bench(name, [&] { hipLaunchKernelGGL(fill_xcd_runs, dim3(wgs), dim3(256), 0, s, buf, n_blocks, R, 0u, v); });
            }
        for (uint32_t wgs : {256u, 2048u, n_blocks / 30u / 8u * 8u})
            for (uint32_t B : {1u, 2u, 8u, 30u}) {
                if (wgs > 2048u && B != 30u) continue;
                snprintf(name, sizeof name, "%u workgroups, XCD-contiguous shares, a workgroup takes %u blocks at a time", wgs, B);
                bench(name, [&] { hipLaunchKernelGGL(fill_xcd_owned, dim3(wgs), dim3(256), 0, s, buf, n_blocks, B, v); });
                if (B >= 8u) {
                    snprintf(name, sizeof name, "%u workgroups, ... %u blocks at a time, each wavefront walking its own quarter", wgs, B);
                    bench(name, [&] { hipLaunchKernelGGL(fill_xcd_owned_waves, dim3(wgs), dim3(256), 0, s, buf, n_blocks, B, v); });
                }
            }
        if (rep == 0) (void)hipFuncSetAttribute((const void*)fill_groups<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int pre : {0, 1}) {
            snprintf(name, sizeof name, "front_probe2's kernel, no LDS opt-in, preamble %d, 4 KB of LDS", pre);
            bench(name, [&] { hipLaunchKernelGGL(fill_groups<0>, dim3(n_blocks / 30u), dim3(256), 4096, s, buf, 120u, 1u, pre, v); });
            snprintf(name, sizeof name, "front_probe2's kernel, 160 KB LDS opt-in, preamble %d, 4 KB of LDS", pre);
            bench(name, [&] { hipLaunchKernelGGL(fill_groups<1>, dim3(n_blocks / 30u), dim3(256), 4096, s, buf, 120u, 1u, pre, v); });
            snprintf(name, sizeof name, "front_probe2's kernel, no LDS opt-in, preamble %d, no LDS", pre);
            if (!pre) bench(name, [&] { hipLaunchKernelGGL(fill_groups<0>, dim3(n_blocks / 30u), dim3(256), 0, s, buf, 120u, 1u, pre, v); });
        }
        for (uint32_t threads : {512u, 1024u})
            for (uint32_t R : {1u, 16u, 1024u}) {
                snprintf(name, sizeof name, "256 workgroups x %u threads, runs of %u blocks per XCD", threads, R);
                bench(name, [&] { hipLaunchKernelGGL(fill_xcd_runs, dim3(256), dim3(threads), 0, s, buf, n_blocks, R, 0u, v); });
            }
    }
    return 0;
}
