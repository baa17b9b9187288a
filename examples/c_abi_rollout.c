/* c_abi_rollout.c -- the drop-in boundary from a compiled host: plain C over include/lle_hip.h and the HIP runtime,
 * no Python and no torch anywhere.  What a Rust / C / Go maintainer of the reference would bind:
 *
 *   map = lle_map_level(6)  ->  batch of n worlds on device 0  ->  steps with on-device action sampling and auto-reset
 *   ->  state vector / reward / done / available actions of every env in one further launch (lle_batch_env_outputs)
 *   ->  a few values copied back and printed, counters summed (lle_batch_stats).
 *
 * Build (done by __graft_entry__.build()):
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_rollout.c -o examples/c_abi_rollout \
 *       -Llle_amd -llle_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/lle_amd -Wl,-rpath,/opt/rocm/lib
 * Run: examples/c_abi_rollout [n_envs] [steps]      (needs an MI355X; exits non-zero with the library's message otherwise)
 */
#define _POSIX_C_SOURCE 199309L /* clock_gettime */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>

#include "lle_hip.h"

#define CHECK_LLE(call)                                                                      \
    do {                                                                                     \
        if ((call) != 0) {                                                                   \
            fprintf(stderr, "%s failed: %s\n", #call, lle_last_error());                     \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define CHECK_HIP(call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
    const int steps = argc > 2 ? atoi(argv[2]) : 100;
    int candidates = argc > 3 ? atoi(argv[3]) : 1; /* > 1: place the arena (INTEGRATION.md 5b "Where the arena lands") */
    if (candidates < 1) candidates = 1;
    if (candidates > 8) candidates = 8;
    int parse_error = 0;
    lle_map* map = lle_map_level(6, &parse_error);
    if (!map) { fprintf(stderr, "lle_map_level: parse error %d: %s\n", parse_error, lle_last_error()); return 1; }
    lle_map_info info;
    CHECK_LLE(lle_map_get_info(map, &info));
    const int A = info.n_agents, G = info.n_gems;
    printf("level 6: %dx%d, %d agents, %d gems, %d sources, observation %d B per env\n", info.height, info.width, A, G,
           info.n_sources, info.obs_bytes);

    hipStream_t stream;
    CHECK_HIP(hipSetDevice(0));
    CHECK_HIP(hipStreamCreate(&stream));
    /* the caller may own the arena (here: hipMalloc of exactly what the library asks for) */
    const int64_t arena_bytes = lle_batch_arena_bytes(map, n);
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    /* Past the 256 MB Infinity Cache the write rate depends on where the allocation landed: k candidate arenas side by side, the
     * step kernel's store pattern timed on each (lle_batch_probe_row_fill), the fastest kept, the others freed. */
    void* arenas[8] = {NULL};
    lle_batch* batches[8] = {NULL};
    int best = 0;
    float best_us = 0.f;
    for (int c = 0; c < candidates; c++) {
        CHECK_HIP(hipMalloc(&arenas[c], (size_t)arena_bytes));
        batches[c] = lle_batch_create(map, n, 0, arenas[c], arena_bytes, stream);
        if (!batches[c]) { fprintf(stderr, "lle_batch_create: %s\n", lle_last_error()); return 1; }
    }
    for (int c = 0; c < candidates && candidates > 1; c++) {
        for (int k = 0; k < 3; k++) CHECK_LLE(lle_batch_probe_row_fill(batches[c], 0, stream));
        CHECK_HIP(hipEventRecord(e0, stream));
        for (int k = 0; k < 10; k++) CHECK_LLE(lle_batch_probe_row_fill(batches[c], 0, stream));
        CHECK_HIP(hipEventRecord(e1, stream));
        CHECK_HIP(hipEventSynchronize(e1));
        float fill_ms = 0.f;
        CHECK_HIP(hipEventElapsedTime(&fill_ms, e0, e1));
        printf("arena %d: row fill %.2f us\n", c, fill_ms * 100.f);
        if (c == 0 || fill_ms * 100.f < best_us) { best = c; best_us = fill_ms * 100.f; }
    }
    for (int c = 0; c < candidates; c++)
        if (c != best) { lle_batch_free(batches[c]); CHECK_HIP(hipFree(arenas[c])); }
    lle_batch* b = batches[best];
    void* arena = arenas[best];
    if (candidates > 1) CHECK_LLE(lle_batch_reset(b, NULL, stream)); /* the probe overwrote the rows */

    const int warmup = 20;  /* (the first launch loads the code object) */
    for (int t = 0; t < warmup; t++)
        CHECK_LLE(lle_batch_step(b, NULL, LLE_STEP_SAMPLE_ACTIONS | LLE_STEP_AUTO_RESET, 1234, (uint64_t)t, 0, stream));
    int64_t zero[8];
    CHECK_LLE(lle_batch_stats(b, zero, 1, stream)); /* reset the counters */
    CHECK_HIP(hipEventRecord(e0, stream));
    struct timespec h0, h1;
    clock_gettime(CLOCK_MONOTONIC, &h0);
    for (int t = warmup; t < warmup + steps; t++)
        CHECK_LLE(lle_batch_step(b, NULL, LLE_STEP_SAMPLE_ACTIONS | LLE_STEP_AUTO_RESET, 1234, (uint64_t)t, 0, stream));
    clock_gettime(CLOCK_MONOTONIC, &h1); /* the calls are issued, nothing has been waited for: the HOST's cost of a step */
    CHECK_HIP(hipEventRecord(e1, stream));
    CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    printf("%d steps of %lld envs: %.2f us per step, %.2f G agent-steps/s; host: %.2f us per lle_batch_step call to issue\n", steps, (long long)n,
           ms * 1e3 / steps, (double)n * A * steps / (ms * 1e-3) / 1e9,
           ((double)(h1.tv_sec - h0.tv_sec) * 1e6 + (double)(h1.tv_nsec - h0.tv_nsec) * 1e-3) / steps);

    /* everything LLE.step returns besides the observation, one launch */
    float *state = NULL, *reward = NULL;
    uint8_t *done = NULL, *avail = NULL;
    CHECK_HIP(hipMalloc((void**)&state, (size_t)n * (3 * A + G) * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&reward, (size_t)n * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&done, (size_t)n));
    CHECK_HIP(hipMalloc((void**)&avail, (size_t)n * A * 5));
    lle_env_outputs out = {0};
    out.state = state; out.reward = reward; out.done = done; out.available = avail;
    out.reward_kind = 0; out.walkable_lasers = 1;
    CHECK_LLE(lle_batch_env_outputs(b, &out, stream));

    /* env 0: its state vector, reward, done, and the first bytes of its layered observation (straight from the arena) */
    float st[64];
    float r0;
    uint8_t d0, av[80];
    int8_t obs_head[16];
    lle_buffer_desc obs_desc;
    CHECK_LLE(lle_batch_get_buffer(b, LLE_BUF_OBS, &obs_desc));
    CHECK_HIP(hipMemcpyAsync(st, state, (size_t)(3 * A + G) * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(&r0, reward, sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(&d0, done, 1, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(av, avail, (size_t)A * 5, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(obs_head, obs_desc.ptr, sizeof obs_head, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    printf("env 0 after %d steps: state [", steps);
    for (int k = 0; k < 3 * A + G; k++) printf("%s%g", k ? " " : "", st[k]);
    printf("] reward %g done %d available(agent 0: N S E W STAY) %d %d %d %d %d\n", r0, d0, av[0], av[1], av[2], av[3], av[4]);
    printf("obs buffer: %lld envs x %lld B (row pitch %lld)\n", (long long)obs_desc.shape[0], (long long)info.obs_bytes,
           (long long)obs_desc.stride[0]);

    int64_t stats[8];
    CHECK_LLE(lle_batch_stats(b, stats, 0, stream));
    printf("env_steps %lld agent_steps %lld gems %lld exits %lld deaths %lld invalid %lld auto_resets %lld reward_sum %lld\n",
           (long long)stats[0], (long long)stats[1], (long long)stats[2], (long long)stats[3], (long long)stats[4],
           (long long)stats[5], (long long)stats[6], (long long)stats[7]);
    if (stats[0] != (int64_t)n * steps || stats[5] != 0) { fprintf(stderr, "unexpected counters\n"); return 1; }

    lle_batch_free(b);
    lle_map_free(map);
    hipFree(state); hipFree(reward); hipFree(done); hipFree(avail); hipFree(arena);
    hipStreamDestroy(stream);
    printf("ok\n");
    return 0;
}
