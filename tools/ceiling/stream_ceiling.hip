// Write-stream ceiling probe for MI355X: how fast can N bytes be written by W wavefronts, each writing a contiguous
// chunk with 16 B/lane stores (the shape of the observation stream of step_kernel)?  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) fill_chunks(uint4* __restrict__ out, uint32_t rows_per_wave, uint32_t delay_iters, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // optional ALU delay before streaming (emulates the state machine in front of the stream)
    uint32_t x = v.x;
    for (uint32_t i = 0; i < delay_iters; i++) x = x * 1664525u + 1013904223u;
    v.y ^= (x & 1u);
    uint4* p = out + (size_t)wave * rows_per_wave * 64 + lane;
    for (uint32_t r = 0; r < rows_per_wave; r++) p[(size_t)r * 64] = v;
}

// persistent: fewer waves, each loops over blocks (block = rows_per_block rows) with stride n_waves
__global__ void __launch_bounds__(256) fill_strided(uint4* __restrict__ out, uint32_t n_blocks, uint32_t rows_per_block, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t b = wave; b < n_blocks; b += n_waves) {
        uint4* p = out + (size_t)b * rows_per_block * 64 + lane;
        for (uint32_t r = 0; r < rows_per_block; r++) p[(size_t)r * 64] = v;
    }
}


// observation-shaped stream: each wave writes `epw` environments of `chunks` 16-byte chunks each; environment index
// of the wave's k-th env = w * epw + k (contiguous per wave) or k * n_waves + w (interleaved across waves)
__global__ void __launch_bounds__(256) fill_envs(uint4* __restrict__ out, uint32_t epw, uint32_t chunks, uint32_t interleaved, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t k = 0; k < epw; k++) {
        const size_t env = interleaved ? (size_t)k * n_waves + wave : (size_t)wave * epw + k;
        uint4* p = out + env * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) p[c] = v;
    }
}
// same, the 4 waves of a workgroup interleaved with each other only (workgroup owns 4 * epw consecutive envs)
__global__ void __launch_bounds__(256) fill_envs_wg(uint4* __restrict__ out, uint32_t epw, uint32_t chunks, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    for (uint32_t k = 0; k < epw; k++) {
        const size_t env = (size_t)blockIdx.x * 4 * epw + k * 4 + wiw;
        uint4* p = out + env * chunks;
        for (uint32_t c = lane; c < chunks; c += 64) p[c] = v;
    }
}

int main(int argc, char** argv) {
    const size_t total = (size_t)65536 * 1872;  // bytes, level-6 observation batch
    const size_t rows = total / 1024;            // 1-KiB rows
    uint4* buf;
    hipMalloc(&buf, total + (1 << 20));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200;
    auto bench = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 20; i++) launch();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; i++) launch();
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %7.2f us  %6.0f GB/s\n", name, ms / iters * 1e3, total / (ms / iters * 1e-3) / 1e9);
        fflush(stdout);
    };
    uint4 v = {1, 2, 3, 4};
    char name[128];
    for (uint32_t waves : {1024u, 2048u, 4096u, 8192u, 16384u, 32768u}) {
        uint32_t rpw = (uint32_t)(rows / waves);
        snprintf(name, sizeof name, "chunks: %u waves x %u KiB (256-thread WGs)", waves, rpw);
        bench(name, [&] { hipLaunchKernelGGL(fill_chunks, dim3(waves / 4), dim3(256), 0, st, buf, rpw, 0u, v); });
    }
    for (uint32_t delay : {500u, 1000u, 2000u}) {
        uint32_t waves = 4096, rpw = (uint32_t)(rows / waves);
        snprintf(name, sizeof name, "chunks: 4096 waves, %u dependent mads before streaming", delay);
        bench(name, [&] { hipLaunchKernelGGL(fill_chunks, dim3(waves / 4), dim3(256), 0, st, buf, rpw, delay, v); });
    }
    for (uint32_t waves : {1024u, 2048u, 4096u}) {
        for (uint32_t rpb : {2u, 8u, 29u}) {
            uint32_t nb = (uint32_t)(rows / rpb);
            snprintf(name, sizeof name, "strided: %u waves, blocks of %u KiB", waves, rpb);
            bench(name, [&] { hipLaunchKernelGGL(fill_strided, dim3(waves / 4), dim3(256), 0, st, buf, nb, rpb, v); });
        }
    }
    for (uint32_t il : {0u, 1u}) {
        for (uint32_t epw : {16u, 8u, 4u}) {
            snprintf(name, sizeof name, "envs: %u waves x %u envs of 1872 B, %s", 65536 / epw, epw, il ? "interleaved" : "contiguous");
            bench(name, [&] { hipLaunchKernelGGL(fill_envs, dim3(65536 / epw / 4), dim3(256), 0, st, buf, epw, 117u, il, v); });
        }
    }
    bench("envs: 4096 waves x 16 envs, interleaved within the workgroup", [&] { hipLaunchKernelGGL(fill_envs_wg, dim3(1024), dim3(256), 0, st, buf, 16u, 117u, v); });
    bench("hipMemsetAsync", [&] { hipMemsetAsync(buf, 1, total, st); });
    return 0;
}
