// kernel_common.hpp -- pieces shared by world_kernel (kernels.hip) and step_kernel (step_kernel.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "obs_stream.hpp"
#include "step_logic.hpp"
#include "tables.h"

namespace lle {

enum Mode : int { MODE_STEP = 0, MODE_RESET = 1, MODE_SET_STATE = 2, MODE_OBSERVE = 3, MODE_SOURCES = 4, MODE_ENV_SOURCES = 5 };

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// per-wave partial counters, written last so that their read-modify-write latency is off the observation's path;
// the slot of a wave is private, so no atomics
struct StepCounts { uint32_t steps, gems, exits, died, invalid, resets, bonus; };
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The wavefront's row of the counters: lane k < 8 holds counter k.  `preloaded`: the old values were read at the top of
// the kernel (stats_preload) -- a load HERE is answered only after every observation store the wavefront has in flight
// (the vmcnt counter is in order), which keeps the wavefront alive for one more memory round trip after its last
// store.  (Atomic adds without return need no load either, but 8 atomics per wavefront measured 0.6 us slower per launch.)
__device__ __forceinline__ int64_t stats_preload(const int64_t* __restrict__ stats, uint32_t wave_id, uint32_t lane) {
    return lane < 8 ? stats[(int64_t)wave_id * 8 + lane] : 0;
}
__device__ __forceinline__ void flush_stats(int64_t* __restrict__ stats, uint32_t wave_id, const StepCounts& c, int A, uint32_t lane,
                                            bool preloaded = false, int64_t old = 0) {
    const int64_t steps = wave_sum_u32(c.steps), gems = wave_sum_u32(c.gems), exits = wave_sum_u32(c.exits);
    const int64_t died = wave_sum_u32(c.died), invalid = wave_sum_u32(c.invalid), resets = wave_sum_u32(c.resets);
    const int64_t bonus = wave_sum_u32(c.bonus);
    const int64_t v = lane == 0 ? steps : lane == 1 ? steps * A : lane == 2 ? gems : lane == 3 ? exits : lane == 4 ? died
                    : lane == 5 ? invalid : lane == 6 ? resets : gems + exits - died + bonus;
    if (preloaded) {
        if (lane < 8) stats[(int64_t)wave_id * 8 + lane] = old + v;
    } else if (lane == 0) {
        int64_t* out = stats + (int64_t)wave_id * 8;
        out[0] += steps; out[1] += steps * A; out[2] += gems; out[3] += exits; out[4] += died;
        out[5] += invalid; out[6] += resets; out[7] += gems + exits - died + bonus;
    }
}

// ---- per-env record I/O.  The per-agent buffers (pos, avail, actions, events) are laid out with a stride of AM
// agents per env (AM = the kernel instantiation's bound, >= the map's A), so a record is a whole number of dwords
// whatever A is, and moves as dwords (the compiler merges neighbours into dwordx2/x4).
template <int AM>
__device__ __forceinline__ void store_u16_record(uint16_t* __restrict__ base, int64_t env, const uint32_t (&v)[AM]) {
    uint32_t* __restrict__ w = reinterpret_cast<uint32_t*>(base) + env * (AM / 2);
#pragma unroll
    for (int k = 0; k < AM / 2; k++) w[k] = (v[2 * k] & 0xFFFFu) | (v[2 * k + 1] << 16);
}
template <int AM>
__device__ __forceinline__ void load_u8_record(const uint8_t* __restrict__ base, int64_t env, uint32_t (&out)[AM]) {
    const uint32_t* __restrict__ w = reinterpret_cast<const uint32_t*>(base) + env * (AM / 4);
#pragma unroll
    for (int k = 0; k < AM / 4; k++) {
        const uint32_t v = w[k];
#pragma unroll
        for (int q = 0; q < 4; q++) out[4 * k + q] = (v >> (8 * q)) & 0xFFu;
    }
}
template <int AM>
__device__ __forceinline__ void store_u8_record(uint8_t* __restrict__ base, int64_t env, const uint32_t (&v)[AM]) {
    uint32_t* __restrict__ w = reinterpret_cast<uint32_t*>(base) + env * (AM / 4);
#pragma unroll
    for (int k = 0; k < AM / 4; k++)
        w[k] = (v[4 * k] & 0xFFu) | ((v[4 * k + 1] & 0xFFu) << 8) | ((v[4 * k + 2] & 0xFFu) << 16) | (v[4 * k + 3] << 24);
}

}  // namespace lle
