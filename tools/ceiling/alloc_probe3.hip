// Round 4, third step.  alloc_probe2 on slow (physically contiguous-like) buffers: rows per wavefront E = 4 / 8 / 16 / 32 give
// 6.05 / 5.95 / 5.73 / 5.5 TB/s -- monotone in the STRIDE between the addresses that the resident wavefronts write at the same
// time (E x 1 920 B = 7.5 / 15 / 30 / 60 KiB: in 256-byte units 30 / 60 / 120 / 240, sharing a factor 2 / 4 / 8 / 16 with any
// power-of-two channel interleave).  Hypothesis: the row stream aliases on the memory channels; buffers made of scattered pages
// break the regularity by accident.  Test: keep every wavefront's rows, but start each wavefront at a different row of its own
// (rotation), so that the concurrently written addresses lose their common stride.  Also E = 1, 2 (odd strides) as a control,
// and config 5's split-row stream (4 wavefronts x 5 KB slices of 32 rows of 20 480 B per workgroup; workgroups 640 KiB apart).
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t n) {
    const uint32_t x = b & 7u, q = n >> 3, r = n & 7u;
    return x * q + (x < r ? x : r) + (b >> 3);
}
__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
// E rows of 1 920 B per wavefront; rot: 0 none, 1 start row = wave index mod E, 2 start row = hash(wave index) mod E,
// 3 = 1 and the two store instructions of a row swapped for odd waves
__global__ void __launch_bounds__(256) rows_rot(uint4* __restrict__ out, uint4 v, uint32_t E, int rot) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const uint32_t wave = blk * 4 + wiw;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t r0 = rot == 0 ? 0u : (rot == 2 ? mixu(wave) % E : wave % E);
    for (uint32_t k = 0; k < E; k++) {
        uint32_t row = k + r0;
        row = row >= E ? row - E : row;
        uint4* p = out + ((size_t)wave * E + row) * 120;
        if (rot == 3 && (wave & 1u)) {
            if (lane < 56) st(p + 64 + lane, w);
            st(p + lane, w);
        } else {
            st(p + lane, w);
            if (lane < 56) st(p + 64 + lane, w);
        }
    }
}
// config 5: 65 536 rows of 20 480 B (1 280 chunks); a workgroup owns 32 rows, wave w the chunks [320 w, 320 (w + 1)) of each.
// rot: 0 none, 1 start row = blk mod 32, 2 start row = hash(blk) mod 32, 3 = 2 and the wave's slice walked from a rotated chunk
__global__ void __launch_bounds__(256) split_rot(uint4* __restrict__ out, uint4 v, int rot) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    const uint32_t r0 = rot == 0 ? 0u : (rot == 1 ? blk & 31u : mixu(blk) & 31u);
    const uint32_t c0 = rot == 3 ? (mixu(blk * 4 + wiw) % 5u) * 64u : 0u;
    for (uint32_t k = 0; k < 32; k++) {
        const uint32_t row = (k + r0) & 31u;
        uint4* p = out + ((size_t)blk * 32 + row) * 1280 + wiw * 320;
        for (uint32_t i = 0; i < 5; i++) {
            uint32_t c = c0 + i * 64u;
            c = c >= 320u ? c - 320u : c;
            st(p + c + lane, w);
        }
    }
}

static hipStream_t s;
static hipEvent_t e0, e1;
static double timeit(size_t bytes, const std::function<void()>& launch, int reps = 20) {
    for (int i = 0; i < 3; i++) launch();
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return bytes / (ms / reps * 1e-3) / 1e9;  // GB/s
}

int main(int argc, char** argv) {
    const int n_cont = argc > 1 ? atoi(argv[1]) : 2, n_malloc = argc > 2 ? atoi(argv[2]) : 6;
    const size_t ROWS = 262144, SMALL = ROWS * 1920, BIG = (size_t)65536 * 20480;  // 480 MiB, 1.25 GiB
    (void)hipStreamCreate(&s);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct B { const char* kind; uint4* p; };
    std::vector<B> bufs;
    for (int i = 0; i < n_malloc; i++) {
        uint4* p = nullptr;
        if (hipMalloc(&p, BIG + (size_t)(i % 3) * (1 << 20)) == hipSuccess) bufs.push_back({"malloc", p});
    }
    for (int i = 0; i < n_cont; i++) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, BIG + (size_t)(i % 3) * (1 << 20), hipDeviceMallocContiguous) == hipSuccess) bufs.push_back({"contiguous", (uint4*)p});
        else (void)hipGetLastError();
    }
    uint4 v = {1, 2, 3, 4};
    for (int i = 0; i < 400; i++) hipLaunchKernelGGL(rows_rot, dim3(4096), dim3(256), 0, s, bufs[0].p, v, 16u, 0);
    (void)hipStreamSynchronize(s);
    printf("level-6 shape (480 MiB): E16 none/mod/hash/mod+swap | E4 none/mod | E2 | E1 | memset  ||  config-5 shape (1.25 GiB): none/blk/hash/hash+chunk | memset\n");
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < bufs.size(); i++) {
            uint4* b = bufs[i].p;
            auto rows = [&](uint32_t E, int rot) {
                return timeit(SMALL, [&] { hipLaunchKernelGGL(rows_rot, dim3(ROWS / (4 * E)), dim3(256), 0, s, b, v, E, rot); });
            };
            const double a0 = rows(16, 0), a1 = rows(16, 1), a2 = rows(16, 2), a3 = rows(16, 3), c0 = rows(4, 0), c1 = rows(4, 1), d0 = rows(2, 0), d1 = rows(1, 0);
            const double m0 = timeit(SMALL, [&] { (void)hipMemsetAsync(b, 1, SMALL, s); });
            auto split = [&](int rot) { return timeit(BIG, [&] { hipLaunchKernelGGL(split_rot, dim3(65536 / 32), dim3(256), 0, s, b, v, rot); }, 10); };
            const double s0 = split(0), s1 = split(1), s2 = split(2), s3 = split(3);
            const double m1 = timeit(BIG, [&] { (void)hipMemsetAsync(b, 1, BIG, s); }, 10);
            printf("%-10s %2zu | %5.0f %5.0f %5.0f %5.0f | %5.0f %5.0f | %5.0f | %5.0f | %5.0f || %5.0f %5.0f %5.0f %5.0f | %5.0f\n", bufs[i].kind, i, a0, a1, a2, a3, c0, c1, d0, d1, m0,
                   s0, s1, s2, s3, m1);
            fflush(stdout);
        }
    return 0;
}
