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

// observation-shaped stream with the store policy of stream_store (obs_stream.hpp): WT = `sc1` write-through.
// `pitch` = 16-byte chunks between the rows of consecutive envs (117 = packed 1 872 B rows, 120 = rows padded to a
// whole number of 128-byte lines).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool WT>
__global__ void __launch_bounds__(256) fill_envs_policy(uint4* __restrict__ out, uint32_t epw, uint32_t chunks, uint32_t pitch, uint4 v) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32x4 w = {v.x, v.y, v.z, v.w};
    for (uint32_t k = 0; k < epw; k++) {
        uint4* p = out + ((size_t)wave * epw + k) * pitch;
        for (uint32_t c = lane; c < chunks; c += 64) {
            if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + c), "v"(w));
            else p[c] = v;
        }
    }
}

// the step kernel's phase-2 shape: per env the wave patches a private LDS copy of the row, reads it back (2 x 16 B per
// lane) and stores it; `lds_bytes` of dynamic LDS per workgroup bound the occupancy like the real kernel's tables do;
// `reads` = 4-byte global loads per lane in front (the state arrays).  unroll2: two envs' LDS reads before the stores.
template <int UNROLL>
__global__ void __launch_bounds__(256) fill_envs_lds(uint4* __restrict__ out, const uint32_t* __restrict__ state, uint32_t epw, uint32_t chunks,
                                                     uint32_t reads, uint4 v) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint4* row = reinterpret_cast<uint4*>(lds + wiw * 4096u);
    uint32_t acc = 0;
    for (uint32_t r = 0; r < reads; r++) acc += state[((size_t)r * gridDim.x * 4 + wave) * 64 + lane];
    row[lane] = v; row[lane + 64u] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t k = 0; k < epw; k += UNROLL) {
        uint4 a[UNROLL], b[UNROLL];
        for (int u = 0; u < UNROLL; u++) {
            reinterpret_cast<uint8_t*>(row)[(lane * 29u + k + u + acc) % 1872u] = (uint8_t)(k + u);  // the patch
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            a[u] = row[lane]; b[u] = row[lane + 64u < chunks ? lane + 64u : 0u];
        }
        for (int u = 0; u < UNROLL; u++) {
            uint4* p = out + ((size_t)wave * epw + k + u) * chunks;
            p[lane] = a[u];
            if (lane + 64u < chunks) p[lane + 64u] = b[u];
        }
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
    // store policy x row pitch x batch size (GB/s below are per 65 536 x 1 872 B: scale by the env count)
    uint4* big;
    hipMalloc(&big, (size_t)524288 * 1920 + (1 << 20));
    for (uint32_t envs : {65536u, 131072u, 262144u, 524288u})
        for (uint32_t pitch : {117u, 120u})
            for (int wt = 0; wt < 2; wt++) {
                snprintf(name, sizeof name, "policy: %u envs, pitch %u B, %s (x%u bytes)", envs, pitch * 16, wt ? "sc1" : "plain", envs / 65536);
                bench(name, [&] {
                    if (wt) hipLaunchKernelGGL(fill_envs_policy<true>, dim3(envs / 64), dim3(256), 0, st, big, 16u, 117u, pitch, v);
                    else hipLaunchKernelGGL(fill_envs_policy<false>, dim3(envs / 64), dim3(256), 0, st, big, 16u, 117u, pitch, v);
                });
            }
    uint32_t* state;
    hipMalloc(&state, (size_t)16 * 32768 * 64 * 4);
    hipMemset(state, 0, (size_t)16 * 32768 * 64 * 4);
    for (uint32_t envs : {65536u, 524288u})
        for (uint32_t lds_bytes : {16384u, 21504u, 40960u})
            for (uint32_t reads : {0u, 10u})
                for (int unroll = 1; unroll <= 2; unroll++) {
                    snprintf(name, sizeof name, "lds-shaped: %u envs, %u B LDS/WG, %u state loads, unroll %d (x%u bytes)", envs, lds_bytes, reads, unroll, envs / 65536);
                    bench(name, [&] {
                        if (unroll == 1) hipLaunchKernelGGL(fill_envs_lds<1>, dim3(envs / 64), dim3(256), lds_bytes, st, big, state, 16u, 117u, reads, v);
                        else hipLaunchKernelGGL(fill_envs_lds<2>, dim3(envs / 64), dim3(256), lds_bytes, st, big, state, 16u, 117u, reads, v);
                    });
                }
    return 0;
}
