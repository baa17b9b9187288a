// Round 3: does a NARROWER write front (fewer resident wavefronts, each owning a longer run of rows) reach the memset-class rate on the
// boxes where 16 384 row-owning wavefronts do not?  Every kernel writes the step kernel's shape (262 144 rows of 1 920 B, a wavefront
// owns groups of 16 rows, XCD-contiguous blocks); what varies is how many wavefronts are resident (occupancy capped by an LDS
// allocation, or a persistent grid looping over its groups) and whether a dependent LDS + ALU chain (the state machine's stand-in)
// runs in front of each group's stores.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }

extern __shared__ uint32_t dyn_lds[];

__device__ __forceinline__ uint32_t chain(uint32_t x, int steps, uint32_t* lds) {
    for (int i = 0; i < steps; i++) x = lds[(x * 2654435761u >> 20) & 1023u] + (x ^ (x >> 7));
    return x;
}

// groups_per_wave consecutive groups of 16 rows per wavefront; the grid's workgroups are laid out XCD-contiguously
__global__ void __launch_bounds__(256) fill_groups(uint4* __restrict__ out, uint32_t chunks, uint32_t groups_per_wave, int chain_steps, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) dyn_lds[i] = i * 40503u + v.x;
    __syncthreads();
    u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t per_group = 16 * chunks;  // 16-byte pieces of a group: 1 920 for level 6
    for (uint32_t k = 0; k < groups_per_wave; k++) {
        if (chain_steps) w.x = chain(w.x + k + lane, chain_steps, dyn_lds);
        uint4* p = out + ((size_t)(blk * 4 + wiw) * groups_per_wave + k) * per_group;
        for (uint32_t c = lane; c < per_group; c += 64) st(p + c, w);
    }
}
// groups interleaved over the persistent wavefronts of an XCD's share: wave w of the XCD takes groups w, w + W, ... of that share
__global__ void __launch_bounds__(256) fill_groups_interleaved(uint4* __restrict__ out, uint32_t chunks, uint32_t groups_per_wave, int chain_steps, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t xcd = blockIdx.x & 7u, in_xcd = blockIdx.x >> 3, waves_per_xcd = (gridDim.x >> 3) * 4;
    for (uint32_t i = threadIdx.x; i < 1024; i += 256) dyn_lds[i] = i * 40503u + v.x;
    __syncthreads();
    u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t per_group = 16 * chunks;
    for (uint32_t k = 0; k < groups_per_wave; k++) {
        if (chain_steps) w.x = chain(w.x + k + lane, chain_steps, dyn_lds);
        const size_t group = (size_t)xcd * waves_per_xcd * groups_per_wave + (size_t)k * waves_per_xcd + in_xcd * 4 + wiw;
        uint4* p = out + group * per_group;
        for (uint32_t c = lane; c < per_group; c += 64) st(p + c, w);
    }
}
__global__ void __launch_bounds__(256) fill_gridstride(uint4* __restrict__ out, size_t n16, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) st(out + i, w);
}

int main(int argc, char** argv) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    const uint32_t n_rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144, chunks = argc > 2 ? (uint32_t)atoi(argv[2]) : 120;
    const size_t bytes = (size_t)n_rows * chunks * 16;
    const uint32_t n_groups = n_rows / 16;
    uint4* buf; if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
    (void)hipFuncSetAttribute((const void*)fill_groups, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)fill_groups_interleaved, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    printf("%u rows x %u B = %.0f MB\n", n_rows, chunks * 16, bytes / 1e6);
    for (int rep = 0; rep < 2; rep++) {
        bench("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, bytes, s); });
        bench("grid-stride 16 B/thread, 256 workgroups", [&] { hipLaunchKernelGGL(fill_gridstride, dim3(256), dim3(256), 0, s, buf, bytes / 16, v); });
        for (int steps : {0, 100}) {
            // one group per wavefront (the step kernel's launch), resident wavefronts capped by the LDS a workgroup asks for
            for (uint32_t lds_kb : {4u, 20u, 40u, 80u, 160u}) {
                snprintf(name, sizeof name, "one group per wave, %u KB LDS per workgroup (<= %u workgroups per CU), chain %d", lds_kb, 160 / lds_kb > 8 ? 8 : 160 / lds_kb, steps);
                bench(name, [&] { hipLaunchKernelGGL(fill_groups, dim3(n_groups / 4), dim3(256), lds_kb * 1024, s, buf, chunks, 1u, steps, v); });
            }
            // persistent grids: W wavefronts, each with n_groups / W groups
            for (uint32_t wgs : {256u, 512u, 1024u, 2048u}) {
                const uint32_t gpw = n_groups / (wgs * 4);
                snprintf(name, sizeof name, "persistent, %u waves x %u consecutive groups, chain %d", wgs * 4, gpw, steps);
                bench(name, [&] { hipLaunchKernelGGL(fill_groups, dim3(wgs), dim3(256), 4096, s, buf, chunks, gpw, steps, v); });
                snprintf(name, sizeof name, "persistent, %u waves x %u groups interleaved within the XCD's share, chain %d", wgs * 4, gpw, steps);
                bench(name, [&] { hipLaunchKernelGGL(fill_groups_interleaved, dim3(wgs), dim3(256), 4096, s, buf, chunks, gpw, steps, v); });
            }
        }
    }
    return 0;
}
