/* c_abi_multi_gpu.c -- the "thin host that owns device buffers" of BASELINE.json's north_star, in plain C: ONE process, one
 * batch handle + one stream + one RCCL communicator per visible GPU, env ranges sharded over the GPUs with no data-path
 * exchange, and the one collective of the path at the end -- the all-reduce of the eight rollout counters
 * (lle_batch_stats_allreduce_group; SURVEY.md section 8(e), BASELINE configs[3]: World.level(6) x 524 288 over 8 GPUs).
 *
 *   examples/c_abi_multi_gpu [envs_per_gpu] [steps] [n_gpus]        (n_gpus: default = all visible; 1 works)
 *
 * The host never makes a device current for the library: every lle_batch_* call runs on its batch's device and restores
 * the caller's (include/lle_hip.h "Threading"); to show it, the loop below leaves the LAST device current throughout.
 * Rank r owns the envs [r*n, (r+1)*n) and samples with env_offset = r*n, so N GPUs reproduce one batch of N*n envs bit for
 * bit (checked against GPU 0 stepping rank 1's range when two devices are present).
 *
 * Build (done by __graft_entry__.build()): as examples/c_abi_rollout.c.
 */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lle_hip.h"

#define MAX_GPUS 16
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

/* 64-bit checksum of the 64-env window that starts at global env id `offset`, stepped 8 times on `device`: positions, alive / arrived /
 * occupant bits, beam words and the int8 layered observation, copied back and hashed on the host (FNV-1a over the bytes, in that order). */
static int window_hash(const lle_map* map, int device, hipStream_t stream, int64_t offset, uint64_t* out) {
    enum { ENVS = 64, STEPS = 8 };
    lle_batch* w = lle_batch_create(map, ENVS, device, NULL, 0, stream);
    if (!w) { fprintf(stderr, "window batch: %s\n", lle_last_error()); return 1; }
    for (int t = 0; t < STEPS; t++)
        CHECK_LLE(lle_batch_step(w, NULL, LLE_STEP_SAMPLE_ACTIONS | LLE_STEP_AUTO_RESET, 1234, (uint64_t)t, offset, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    uint64_t h = 1469598103934665603ull;
    const int which[4] = {LLE_BUF_POS, LLE_BUF_BITS, LLE_BUF_BEAMS, LLE_BUF_OBS};
    for (int k = 0; k < 4; k++) {
        lle_buffer_desc d;
        CHECK_LLE(lle_batch_get_buffer(w, which[k], &d));
        const size_t bytes = (size_t)ENVS * (size_t)d.stride[0] * (size_t)d.elem_bytes;
        uint8_t* host = (uint8_t*)malloc(bytes);
        CHECK_HIP(hipMemcpy(host, d.ptr, bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < bytes; i++) { h ^= host[i]; h *= 1099511628211ull; }
        free(host);
    }
    lle_batch_free(w);
    *out = h;
    return 0;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    int visible = 0;
    CHECK_HIP(hipGetDeviceCount(&visible));
    int n_gpus = argc > 3 ? atoi(argv[3]) : visible;
    if (n_gpus < 1 || n_gpus > visible || n_gpus > MAX_GPUS) {
        fprintf(stderr, "asked for %d GPUs, %d visible: refusing to run on fewer than asked for\n", n_gpus, visible);
        return 2;
    }
    const uint32_t flags = LLE_STEP_SAMPLE_ACTIONS | LLE_STEP_AUTO_RESET;
    const uint64_t seed = 1234;
    int parse_error = 0;
    lle_map* map = lle_map_level(6, &parse_error);
    if (!map) { fprintf(stderr, "lle_map_level: %s\n", lle_last_error()); return 1; }
    lle_map_info info;
    CHECK_LLE(lle_map_get_info(map, &info));

    lle_batch* batch[MAX_GPUS];
    lle_comm* comm[MAX_GPUS];
    hipStream_t stream[MAX_GPUS];
    void* streams[MAX_GPUS];
    hipEvent_t e0[MAX_GPUS], e1[MAX_GPUS];
    for (int g = 0; g < n_gpus; g++) {
        CHECK_HIP(hipSetDevice(g));  /* streams and events belong to a device; the library calls below do not need this */
        CHECK_HIP(hipStreamCreate(&stream[g]));
        CHECK_HIP(hipEventCreate(&e0[g]));
        CHECK_HIP(hipEventCreate(&e1[g]));
        streams[g] = stream[g];
    }
    /* from here on the LAST device stays current: a wrong current device for every handle but one */
    for (int g = 0; g < n_gpus; g++) {
        batch[g] = lle_batch_create(map, n, g, NULL, 0, stream[g]);
        if (!batch[g]) { fprintf(stderr, "lle_batch_create on GPU %d: %s\n", g, lle_last_error()); return 1; }
    }
    CHECK_LLE(lle_comm_create_all(comm, n_gpus, NULL));
    int current = -1;
    CHECK_HIP(hipGetDevice(&current));
    if (current != n_gpus - 1) { fprintf(stderr, "the library changed the current device (%d)\n", current); return 1; }

    /* Before anything is timed: shard invariance over the run's own collective (bench.py's `shard_check`).  GPU g
     * hashes the window at the head of ITS shard (env_offset = g * n) into slot g of an int64[n_gpus] array on its device, one RCCL
     * all-reduce (sum; the other slots are zero) hands every rank every hash, and GPU 0 recomputes all windows with the matching offsets. */
    {
        int64_t* slots[MAX_GPUS];
        for (int g = 0; g < n_gpus; g++) {
            uint64_t h = 0;
            if (window_hash(map, g, stream[g], (int64_t)g * n, &h)) return 1;
            int64_t mine[MAX_GPUS] = {0};
            mine[g] = (int64_t)(h >> 1);  /* (63 bits: a sum of one value and zeros cannot overflow) */
            CHECK_HIP(hipSetDevice(g));
            CHECK_HIP(hipMalloc((void**)&slots[g], sizeof mine));
            CHECK_HIP(hipMemcpy(slots[g], mine, sizeof mine, hipMemcpyHostToDevice));
        }
        CHECK_HIP(hipSetDevice(n_gpus - 1));
        CHECK_LLE(lle_comm_allreduce_i64_group(comm, slots, streams, n_gpus, n_gpus, LLE_COMM_SUM));
        int64_t got[MAX_GPUS];
        CHECK_HIP(hipStreamSynchronize(stream[0]));
        CHECK_HIP(hipMemcpy(got, slots[0], sizeof(int64_t) * (size_t)n_gpus, hipMemcpyDeviceToHost));
        for (int g = 0; g < n_gpus; g++) {
            uint64_t h = 0;
            if (window_hash(map, 0, stream[0], (int64_t)g * n, &h)) return 1;
            if ((int64_t)(h >> 1) != got[g]) { fprintf(stderr, "shard check FAILED: GPU %d's window differs from GPU 0's replay of it\n", g); return 1; }
        }
        for (int g = 0; g < n_gpus; g++) { CHECK_HIP(hipStreamSynchronize(stream[g])); CHECK_HIP(hipFree(slots[g])); }
        printf("shard check: ok (%d window(s) of 64 envs x 8 steps hashed on their GPUs, all-reduced over RCCL, replayed on GPU 0)\n", n_gpus);
    }

    const int warmup = 50;
    for (int t = 0; t < warmup; t++)
        for (int g = 0; g < n_gpus; g++) CHECK_LLE(lle_batch_step(batch[g], NULL, flags, seed, (uint64_t)t, (int64_t)g * n, stream[g]));
    int64_t total[8];
    CHECK_LLE(lle_batch_stats_allreduce_group(batch, comm, streams, n_gpus, total, 1)); /* also: every GPU idle, counters zero */

    const double t0 = now_s();
    for (int g = 0; g < n_gpus; g++) CHECK_HIP(hipEventRecord(e0[g], stream[g]));
    for (int t = warmup; t < warmup + steps; t++)
        for (int g = 0; g < n_gpus; g++) CHECK_LLE(lle_batch_step(batch[g], NULL, flags, seed, (uint64_t)t, (int64_t)g * n, stream[g]));
    for (int g = 0; g < n_gpus; g++) CHECK_HIP(hipEventRecord(e1[g], stream[g]));
    /* the end-of-batch reduction: the only collective of the path (64 bytes over RCCL / xGMI) */
    CHECK_LLE(lle_batch_stats_allreduce_group(batch, comm, streams, n_gpus, total, 0));
    const double wall = now_s() - t0;

    float slowest = 0.f;
    for (int g = 0; g < n_gpus; g++) {
        float ms = 0.f;
        CHECK_HIP(hipEventElapsedTime(&ms, e0[g], e1[g]));
        printf("gpu %d: %.2f us per step by its own events (%.2f G agent-steps/s)\n", g, ms * 1e3 / steps,
               (double)n * info.n_agents * steps / (ms * 1e-3) / 1e9);
        if (ms > slowest) slowest = ms;
    }
    printf("%d GPU(s) x %lld envs x %d steps: %.2f G agent-steps/s by the slowest GPU's events, %.2f by the host's wall clock "
           "(reduction included)\n", n_gpus, (long long)n, steps, (double)n_gpus * n * info.n_agents * steps / (slowest * 1e-3) / 1e9,
           (double)n_gpus * n * info.n_agents * steps / wall / 1e9);
    printf("env_steps %lld agent_steps %lld gems %lld exits %lld deaths %lld invalid %lld auto_resets %lld reward_sum %lld\n",
           (long long)total[0], (long long)total[1], (long long)total[2], (long long)total[3], (long long)total[4],
           (long long)total[5], (long long)total[6], (long long)total[7]);
    if (total[0] != (int64_t)n_gpus * n * steps || total[5] != 0) { fprintf(stderr, "unexpected counters\n"); return 1; }

    /* shard invariance: a second handle on GPU 0 steps the LAST rank's env range; same positions after the same steps */
    if (n_gpus > 1) {
        lle_batch* twin = lle_batch_create(map, n, 0, NULL, 0, stream[0]);
        if (!twin) { fprintf(stderr, "twin: %s\n", lle_last_error()); return 1; }
        for (int t = 0; t < warmup + steps; t++)
            CHECK_LLE(lle_batch_step(twin, NULL, flags, seed, (uint64_t)t, (int64_t)(n_gpus - 1) * n, stream[0]));
        lle_buffer_desc da, db;
        CHECK_LLE(lle_batch_get_buffer(twin, LLE_BUF_POS, &da));
        CHECK_LLE(lle_batch_get_buffer(batch[n_gpus - 1], LLE_BUF_POS, &db));
        uint8_t* ha = (uint8_t*)malloc((size_t)da.bytes);
        uint8_t* hb = (uint8_t*)malloc((size_t)db.bytes);
        CHECK_HIP(hipStreamSynchronize(stream[0]));
        CHECK_HIP(hipMemcpy(ha, da.ptr, (size_t)da.bytes, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(hb, db.ptr, (size_t)db.bytes, hipMemcpyDeviceToHost));
        if (da.bytes != db.bytes || memcmp(ha, hb, (size_t)n * (size_t)da.stride[0]) != 0) { fprintf(stderr, "shards differ\n"); return 1; }
        printf("shard invariance: GPU 0 replaying rank %d's range ends in the same positions\n", n_gpus - 1);
        free(ha); free(hb);
        lle_batch_free(twin);
    }
    CHECK_HIP(hipGetDevice(&current));
    if (current != n_gpus - 1) { fprintf(stderr, "the library changed the current device (%d)\n", current); return 1; }

    for (int g = 0; g < n_gpus; g++) { lle_comm_free(comm[g]); lle_batch_free(batch[g]); }
    lle_map_free(map);
    printf("ok\n");
    return 0;
}
