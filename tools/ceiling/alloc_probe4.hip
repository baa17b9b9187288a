// Round 4, fourth step.  alloc_probe3: past the Infinity Cache the row stream goes faster the closer together the resident wavefronts
// write (E = 16 / 4 / 1 rows per wavefront: 5.7 / 6.05 / 6.45 TB/s on slow buffers).  The state machine wants 16 environments per
// wavefront -- but WHICH rows a wavefront streams afterwards is free once the hand-over records sit in LDS: the 4 wavefronts of a
// workgroup can sweep the workgroup's 64 rows together (wave w: rows w, w + 4, ...), i.e. 4 adjacent rows in flight per workgroup
// instead of 4 rows 30 KB apart.  This probe times that pattern against the plain one, on ordinary and on contiguous buffers.
// Patterns (262 144 rows of 1 920 B, 4 096 workgroups of 4 wavefronts, XCD-contiguous blocks):
//   0 plain: wave owns 16 consecutive rows      1 plain + rotation (start row = wave mod 16)
//   2 interleaved: wave w of the workgroup writes rows 4k + w     3 interleaved + rotation (start k = block mod 16)
//   4 interleaved by pairs: rows 8k + 2w, 8k + 2w + 1             5 quarter rows: every row written by all 4 waves (480 B each, 2 rows per store)
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
__global__ void __launch_bounds__(256) rows(uint4* __restrict__ out, uint4 v, int pat) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, blk = xcd_block(blockIdx.x, gridDim.x);
    const u32x4 w = {v.x, v.y, v.z, v.w};
    uint4* base = out + (size_t)blk * 64 * 120;
    if (pat == 5) {
        // 30 chunks per quarter row: lanes 0-29 one row, 32-61 the next
        const uint32_t half = lane >> 5, l = lane & 31u;
        for (uint32_t k = 0; k < 32; k++) {
            uint4* p = base + (size_t)(2 * k + half) * 120 + wiw * 30;
            if (l < 30) st(p + l, w);
        }
        return;
    }
    for (uint32_t k = 0; k < 16; k++) {
        uint32_t row;
        if (pat == 0) row = wiw * 16 + k;
        else if (pat == 1) row = wiw * 16 + ((k + blk * 4 + wiw) & 15u);
        else if (pat == 2) row = k * 4 + wiw;
        else if (pat == 3) row = ((k + blk) & 15u) * 4 + wiw;
        else row = (k >> 1) * 8 + wiw * 2 + (k & 1u);
        uint4* p = base + (size_t)row * 120;
        st(p + lane, w);
        if (lane < 56) st(p + 64 + lane, w);
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
    return bytes / (ms / reps * 1e-3) / 1e9;
}
int main(int argc, char** argv) {
    const int n_cont = argc > 1 ? atoi(argv[1]) : 2, n_malloc = argc > 2 ? atoi(argv[2]) : 6;
    const size_t ROWS = 262144, SMALL = ROWS * 1920;
    (void)hipStreamCreate(&s);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct B { const char* kind; uint4* p; };
    std::vector<B> bufs;
    for (int i = 0; i < n_malloc; i++) {
        uint4* p = nullptr;
        if (hipMalloc(&p, SMALL + (size_t)(i % 3) * (1 << 20)) == hipSuccess) bufs.push_back({"malloc", p});
    }
    for (int i = 0; i < n_cont; i++) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, SMALL + (size_t)(i % 3) * (1 << 20), hipDeviceMallocContiguous) == hipSuccess) bufs.push_back({"contiguous", (uint4*)p});
        else (void)hipGetLastError();
    }
    uint4 v = {1, 2, 3, 4};
    for (int i = 0; i < 400; i++) hipLaunchKernelGGL(rows, dim3(4096), dim3(256), 0, s, bufs[0].p, v, 0);
    (void)hipStreamSynchronize(s);
    printf("GB/s: plain | plain+rot | interleaved | interleaved+rot | pairs | quarter rows | memset\n");
    for (int pass = 0; pass < 2; pass++)
        for (size_t i = 0; i < bufs.size(); i++) {
            uint4* b = bufs[i].p;
            double r[6];
            for (int pat = 0; pat < 6; pat++) r[pat] = timeit(SMALL, [&] { hipLaunchKernelGGL(rows, dim3(ROWS / 64), dim3(256), 0, s, b, v, pat); });
            const double m0 = timeit(SMALL, [&] { (void)hipMemsetAsync(b, 1, SMALL, s); });
            printf("%-10s %2zu | %5.0f | %5.0f | %5.0f | %5.0f | %5.0f | %5.0f | %5.0f\n", bufs[i].kind, i, r[0], r[1], r[2], r[3], r[4], r[5], m0);
            fflush(stdout);
        }
    return 0;
}
