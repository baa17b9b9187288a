// Round 3: would a PERSISTENT step kernel (a workgroup pays its prologue -- dependent global round trips for the tables, a barrier --
// once and then loops over several groups of environments) beat one group per wavefront in a many-round launch?  A stand-in: per
// workgroup a chain of `rt` dependent global loads + an LDS fill + a barrier (the prologue), per group of 16 rows a dependent
// LDS + ALU chain (the state machine) and then the rows' stores; alternating walk as in the product.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }
extern __shared__ uint32_t lds[];

__global__ void __launch_bounds__(256) fill(uint4* __restrict__ out, const uint32_t* __restrict__ tab, uint32_t per_wave, uint32_t groups, int rt, int chain,
                                            int reverse, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (reverse) blk = gridDim.x - 1u - blk;
    // prologue: rt dependent round trips, then 16 KB into LDS, barrier
    uint32_t x = v.x & 1023u;
    for (int i = 0; i < rt; i++) x = __builtin_nontemporal_load(tab + ((x + lane) & 4095u)) & 1023u;
    for (uint32_t i = threadIdx.x; i < 4096; i += 256) lds[i] = tab[(i + x) & 4095u];
    __syncthreads();
    u32x4 w = {v.x, v.y, v.z, v.w + x};
    for (uint32_t g = 0; g < groups; g++) {
        uint32_t y = w.x + g + lane;
        for (int i = 0; i < chain; i++) y = lds[(y * 2654435761u >> 20) & 4095u] + (y ^ (y >> 7));
        w.x = y;
        const uint32_t grp = reverse ? groups - 1u - g : g;
        uint4* p = out + (((size_t)blk * 4u + wiw) * groups + grp) * per_wave;
        for (uint32_t c = lane; c < per_wave; c += 64) st(p + c, w);
    }
}

int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    const uint32_t per_wave = 1920, n_groups = 16384;  // 16 384 groups of 30 KB = 503 MB
    const size_t bytes = (size_t)n_groups * per_wave * 16;
    uint4* buf; uint32_t* tab;
    if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess || hipMalloc(&tab, 4096 * 4) != hipSuccess) return 1;
    (void)hipMemset(tab, 0, 4096 * 4);
    for (int rep = 0; rep < 2; rep++)
        for (int rt : {0, 3})
            for (int chain : {0, 60, 100})
                for (uint32_t groups : {1u, 2u, 4u, 8u}) {
                    const uint32_t wgs = n_groups / 4 / groups;
                    int launch = 0;
                    auto go = [&] { hipLaunchKernelGGL(fill, dim3(wgs), dim3(256), 16384, s, buf, tab, per_wave, groups, rt, chain, (launch++) & 1, v); };
                    for (int i = 0; i < 6; i++) go();
                    (void)hipStreamSynchronize(s);
                    (void)hipEventRecord(e0, s);
                    for (int i = 0; i < 30; i++) go();
                    (void)hipEventRecord(e1, s);
                    (void)hipStreamSynchronize(s);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    printf("prologue round trips %d, chain %3d, %u group(s) per wavefront (%4u workgroups): %7.2f us  %5.0f GB/s\n", rt, chain, groups, wgs,
                           ms / 30 * 1e3, bytes / (ms / 30 * 1e-3) / 1e9);
                    fflush(stdout);
                }
    return 0;
}
