// Does it pay to store the STATIC lines of the observation rows before the state machine runs?
// The step kernel's launch = (ramp) + (state machine: a dependent ALU chain, every wave at the same time) + (stream).
// Lines of a row that no agent / beam / gem can touch could go out before the chain.  This probe has the step kernel's
// shape (65 536 rows of 1 920 B, 16 rows per wave, 4 waves per workgroup, XCD-contiguous blocks, sc1 stores) with a
// dependent ALU chain of `n_alu` v_mad in front of the stream, and `head` of the 15 lines of each row stored BEFORE
// the chain.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(uint4* p, const u32x4& w) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w)); }

__global__ void __launch_bounds__(256, 4) rows_with_chain(uint4* __restrict__ out, const uint32_t* __restrict__ seedp, uint32_t n_alu, uint32_t head, uint32_t lds_words) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x, nb = gridDim.x;
    const uint32_t x = b & 7u, q = nb >> 3, r = nb & 7u;
    const uint32_t blk = x * q + (x < r ? x : r) + (b >> 3);
    uint32_t v = seedp[lane & 15u];  // one global round trip, like the table loads
    v += __builtin_amdgcn_readfirstlane(v);  // (waited for HERE: a vmcnt wait after the head stores would wait for their acknowledgements)
    for (uint32_t i = threadIdx.x; i < lds_words; i += 256) lds[i] = v + i;
    __syncthreads();
    const u32x4 w0 = {v, 2, 3, 4};
    uint4* base = out + (size_t)(blk * 4 + wiw) * 16 * 120;
    // head: lines [0, head) of the wave's 16 rows (8 lanes per 128-byte line)
    const uint32_t hl = lane >> 3, hc = lane & 7u;
    for (uint32_t k = 0; k < 16; k++)
        for (uint32_t l = hl; l < head; l += 8) st(base + k * 120 + l * 8 + hc, w0);
    // the chain
    uint32_t acc = v;
    for (uint32_t i = 0; i < n_alu; i++) acc = acc * 1664525u + lds[(acc >> 7) % lds_words];
    const u32x4 w = {acc, 2, 3, 4};
    // the rest of the rows
    for (uint32_t k = 0; k < 16; k++)
        for (uint32_t l = head + hl; l < 15; l += 8) st(base + k * 120 + l * 8 + hc, w);
}

int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t n_rows = 65536, chunks = 120;
    const size_t bytes = (size_t)n_rows * chunks * 16;
    uint4* buf; if (hipMalloc(&buf, bytes + (1 << 20)) != hipSuccess) return 1;
    uint32_t* seed; (void)hipMalloc(&seed, 64); (void)hipMemset(seed, 1, 64);
    const uint32_t alus[] = {0, 20, 40, 60, 100, 150};
    const uint32_t heads[] = {0, 2, 4, 6, 8};
    for (uint32_t n_alu : alus)
        for (uint32_t head : heads) {
            auto launch = [&]() { hipLaunchKernelGGL(rows_with_chain, dim3(n_rows / 64), dim3(256), 8192, s, buf, seed, n_alu, head, 2048u); };
            // a second or so of launches first: the clocks of a fresh box ramp up
            for (int i = 0; i < 3000; i++) launch();
            (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < 300; i++) launch();
            (void)hipEventRecord(e1, s);
            (void)hipStreamSynchronize(s);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("chain of %4u dependent LDS+ALU steps, %u of 15 lines stored before it: %7.2f us  %6.0f GB/s\n", n_alu, head, ms / 300 * 1e3,
                   bytes / (ms / 300 * 1e-3) / 1e9);
            fflush(stdout);
        }
    return 0;
}
