// Round 3: a buffer of 1-2x the Infinity Cache (256 MB) rewritten launch after launch in the SAME order never hits the cache (what the
// previous launch left there is its tail, and this launch starts at the head); rewritten in ALTERNATING order (ascending, descending,
// ...) each launch starts with the lines the previous one finished with.  Does the memory system reward that?  The step kernel's
// stream shape (workgroups own 120 KB, XCD-contiguous, a wavefront walks its 30 KB).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }

// gridDim.x workgroups of 4 wavefronts; workgroup `blk` owns pieces [blk * per_wg, (blk + 1) * per_wg); reverse: the grid walks the buffer from its end
__global__ void __launch_bounds__(256) fill_owned(uint4* __restrict__ out, uint32_t per_wave, int reverse, uint4 v) {
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (reverse) blk = gridDim.x - 1u - blk;
    uint4* p = out + ((size_t)blk * 4u + (reverse ? 3u - wiw : wiw)) * per_wave;
    if (!reverse) for (uint32_t c = lane; c < per_wave; c += 64) st(p + c, w);
    else for (uint32_t c = per_wave - 64u + lane; (int32_t)c >= 0; c -= 64) st(p + c, w);
}

int main(int argc, char** argv) {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    uint4 v = {1, 2, 3, 4};
    const uint32_t per_wave = 1920;  // 16-byte pieces: 30 KB
    for (uint32_t mb : {240u, 360u, 480u, 720u, 960u, 1320u, 2040u}) {
        const size_t bytes = (size_t)mb << 20;
        const uint32_t wgs = (uint32_t)(bytes / (4u * per_wave * 16u)) / 8u * 8u;
        const size_t used = (size_t)wgs * 4u * per_wave * 16u;
        uint4* buf; if (hipMalloc(&buf, used + (1 << 20)) != hipSuccess) return 1;
        for (int mode = 0; mode < 3; mode++) {  // 0: always ascending, 1: alternating, 2: always descending
            int launch = 0;
            auto go = [&] { const int rev = mode == 2 ? 1 : (mode == 1 ? (launch & 1) : 0); launch++;
                            hipLaunchKernelGGL(fill_owned, dim3(wgs), dim3(256), 0, s, buf, per_wave, rev, v); };
            for (int i = 0; i < 6; i++) go();
            (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < 30; i++) go();
            (void)hipEventRecord(e1, s);
            (void)hipStreamSynchronize(s);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MB  %-18s %8.2f us  %6.0f GB/s\n", used >> 20, mode == 0 ? "ascending" : mode == 1 ? "alternating" : "descending", ms / 30 * 1e3,
                   used / (ms / 30 * 1e-3) / 1e9);
            fflush(stdout);
        }
        (void)hipFree(buf);
    }
    return 0;
}
