// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched World.
//
// One workgroup = one 64-lane wavefront.  Phase 1 (lane = environment): load the env's packed state, run the
// state machine of step_logic.hpp in registers, store the new state / events / availability masks.
// Phase 2 (wave = one environment at a time): the wave owns a private LDS copy of the map's static observation;
// for each of its environments it patches the few dynamic bytes (laser on/off bits, gems, agents) into that copy
// and streams it to HBM with one 16-byte store per lane, i.e. 1 KiB fully coalesced per wave instruction.
// The observation is >= 95 % of all bytes moved, so phase 2 is what the HBM roofline measures; phase 1 is
// integer work hidden behind it by running several waves per SIMD.
//
// No MFMA: there is no contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "step_logic.hpp"
#include "tables.h"

namespace lle {

enum Mode : int { MODE_STEP = 0, MODE_RESET = 1, MODE_SET_STATE = 2, MODE_OBSERVE = 3, MODE_SOURCES = 4 };

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

template <int AM, int LM, int MODE>
__global__ void __launch_bounds__(64) world_kernel(BatchPtrs P, LaunchArgs K) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(P.tables);
    const uint32_t lane = threadIdx.x;
    const int A = (int)hdr->A, L = (int)hdr->L;
    const uint32_t epw = K.envs_per_wave;
    const int64_t env0 = (int64_t)blockIdx.x * epw;
    const int64_t env = env0 + lane;
    const bool active = lane < epw && env < P.n_envs;

    // ---- static tables -> LDS (the wave's private copy; section offsets are those of the blob)
    const uint32_t tab_bytes = hdr->lds_table_bytes, tab_off = hdr->off_cell_lay;
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(P.tables + tab_off);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        for (uint32_t i = lane; i < tab_bytes / 16; i += 64) dst[i] = src[i];
    }
    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds);
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (hdr->off_cell_meta - tab_off));
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + (hdr->off_dyn - tab_off));
    int8_t* tmpl = reinterpret_cast<int8_t*>(lds + (hdr->off_template - tab_off));
    uint32_t* scratch = reinterpret_cast<uint32_t*>(lds + tab_bytes);
    const uint32_t scr_stride = (uint32_t)(L + A + 1) | 1u;  // odd: lanes spread over banks
    __syncthreads();

    MapView mv;
    mv.cell_lay = cell_lay; mv.cell_meta = cell_meta; mv.hdr = hdr;
    mv.W = (int)hdr->W; mv.A = A; mv.L = L; mv.G = (int)hdr->G;
    mv.enabled = hdr->enabled_mask; mv.max_layers = hdr->max_layers;
    const uint32_t amask = (1u << A) - 1u;
    uint64_t stat1 = 0, stat2 = 0;  // packed per-env counters, summed over the wave below

    if (active) {
        Env<AM, LM> s;
#pragma unroll
        for (int a = 0; a < AM; a++) s.pos[a] = (a < A) ? (uint32_t)P.pos[env * A + a] : 0xFFFF0000u + (uint32_t)a;
        {
            const uint64_t bits = P.bits[env];
            s.alive = (uint32_t)bits & 0xFFFFu; s.arrived = (uint32_t)(bits >> 16) & 0xFFFFu; s.occ = (uint32_t)(bits >> 32) & 0xFFFFu;
        }
        s.gems = P.gems[env];
#pragma unroll
        for (int b = 0; b < LM; b++) s.beams[b] = (b < L) ? P.beams[env * L + b] : 0u;

        bool store_state = true, store_avail = false, touched = true;
        uint32_t avail[AM];
        Events<AM> ev;
        ev.clear();
        uint32_t err = 0, was_reset = 0;

        if (MODE == MODE_STEP) {
#pragma unroll
            for (int a = 0; a < AM; a++) avail[a] = (a < A) ? (uint32_t)P.avail[env * A + a] : 0u;
            if ((K.flags & STEP_AUTO_RESET) && (s.alive != amask || s.arrived == amask)) {
                reset_env<AM, LM>(s, mv);
                compute_avail<AM, LM>(s, mv, avail);
                was_reset = 1;
            }
            uint32_t act[AM];
            if (K.flags & STEP_SAMPLE_ACTIONS) {
                const uint64_t he = action_hash_env(K.seed, (uint64_t)(K.env_offset + env), K.t);
#pragma unroll
                for (int a = 0; a < AM; a++) act[a] = (a < A) ? sample_action(avail[a], action_hash_agent(he, (uint64_t)a)) : 4u;
#pragma unroll
                for (int a = 0; a < AM; a++)
                    if (a < A) P.actions[env * A + a] = (uint8_t)act[a];
            } else {
                const uint8_t* __restrict__ src = K.actions_in ? K.actions_in : P.actions;
#pragma unroll
                for (int a = 0; a < AM; a++) act[a] = (a < A) ? (uint32_t)src[env * A + a] : 4u;
                if (K.actions_in) {
#pragma unroll
                    for (int a = 0; a < AM; a++)
                        if (a < A) P.actions[env * A + a] = (uint8_t)act[a];
                }
            }
            // availability check: lowest offending agent (world.rs:444-453), before any mutation
#pragma unroll
            for (int a = AM - 1; a >= 0; a--) {
                if (a < A) {
                    // `avail` is the cached list of the reference (world.rs:444-453).  It can only disagree with the
                    // static walk mask after a failed set_state left it stale (world.rs:588-594 returns before
                    // recomputing it); the reference would then index out of the grid and panic, we refuse the action.
                    const uint32_t walk = ((mv.cell_meta[cell_of(s.pos[a], mv.W)] >> 8) & 15u) | 16u;
                    if (act[a] > 4u || !((avail[a] >> act[a]) & 1u) || !((walk >> act[a]) & 1u)) err = (uint32_t)a + 1u;
                }
            }
            if (err == 0) {
                step_env<AM, LM>(s, act, mv, ev);
                compute_avail<AM, LM>(s, mv, avail);
                store_avail = true;
            } else {
                store_state = was_reset != 0;
                store_avail = was_reset != 0;
            }
        } else if (MODE == MODE_RESET) {
            if (!K.env_mask || K.env_mask[env]) {
                reset_env<AM, LM>(s, mv);
                compute_avail<AM, LM>(s, mv, avail);
                store_avail = true;
            } else {
                store_state = false;
                touched = false;
            }
        } else if (MODE == MODE_SET_STATE) {
            uint32_t rp[AM];
#pragma unroll
            for (int a = 0; a < AM; a++) rp[a] = (a < A) ? (uint32_t)P.req_pos[env * A + a] : 0xFFFF0000u + (uint32_t)a;
            bool dirty = false;
            err = set_state_env<AM, LM>(s, rp, P.req_gems[env], (uint32_t)P.req_alive[env], mv, ev, dirty);
            if (err != 0) ev.clear();
            if (dirty) { compute_avail<AM, LM>(s, mv, avail); store_avail = true; }
        } else if (MODE == MODE_SOURCES) {
            // LaserBeam::disable -> all off; LaserBeam::enable -> all on (laser.rs:69-77)
#pragma unroll
            for (int b = 0; b < LM; b++) {
                if (b < L) {
                    const bool was = (K.old_enabled >> b) & 1u, now = (mv.enabled >> b) & 1u;
                    if (was && !now) s.beams[b] = 0u;
                    if (!was && now) s.beams[b] = hdr->beam_full[b];
                }
            }
        } else {
            store_state = false;
        }

        if (store_state) {
#pragma unroll
            for (int a = 0; a < AM; a++)
                if (a < A) P.pos[env * A + a] = (uint16_t)s.pos[a];
            P.bits[env] = (uint64_t)s.alive | ((uint64_t)s.arrived << 16) | ((uint64_t)s.occ << 32);
            P.gems[env] = s.gems;
#pragma unroll
            for (int b = 0; b < LM; b++)
                if (b < L) P.beams[env * L + b] = s.beams[b];
        }
        if (store_avail) {
#pragma unroll
            for (int a = 0; a < AM; a++)
                if (a < A) P.avail[env * A + a] = (uint8_t)avail[a];
        }
        if ((MODE == MODE_STEP || MODE == MODE_RESET || MODE == MODE_SET_STATE) && touched) {
            P.err[env] = (uint8_t)err;
            P.evcount[env] = (uint8_t)(ev.n | (was_reset << 7));
#pragma unroll
            for (int k = 0; k < 2 * AM; k++)
                if (k < 2 * A) P.events[env * 2 * A + k] = (uint8_t)(ev.w[k >> 3] >> ((k & 7) * 8));
            P.done[env] = (s.alive != amask || s.arrived == amask) ? 1 : 0;
        }

        // hand the dynamic state to phase 2
        uint32_t* sc = scratch + lane * scr_stride;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) sc[b] = s.beams[b];
        sc[L] = s.gems;
#pragma unroll
        for (int a = 0; a < AM; a++)
            if (a < A) sc[L + 1 + a] = (uint32_t)a * hdr->HW + cell_of(s.pos[a], mv.W);

        if (MODE == MODE_STEP) {
            uint32_t n_gem = 0, n_exit = 0, n_died = 0;
#pragma unroll
            for (int k = 0; k < 2 * AM; k++) {
                if ((uint32_t)k < ev.n) {
                    const uint32_t ty = ((uint32_t)(ev.w[k >> 3] >> ((k & 7) * 8)) >> 4) & 3u;
                    n_gem += ty == EV_GEM; n_exit += ty == EV_EXIT; n_died += ty == EV_DIED;
                }
            }
            const uint32_t bonus = (err == 0 && s.arrived == amask) ? 1u : 0u;
            stat1 = (uint64_t)n_gem | ((uint64_t)n_exit << 12) | ((uint64_t)n_died << 24) |
                    ((uint64_t)(err != 0) << 36) | ((uint64_t)was_reset << 48);
            stat2 = 1ull | ((uint64_t)bonus << 12);
        }
    }
    __syncthreads();

    if (MODE == MODE_STEP) {
        // per-wave partial counters; the slot of this wave is private, so no atomics
        const uint64_t p1 = wave_sum_u64(stat1);
        const uint64_t p2 = wave_sum_u64(stat2);
        if (lane == 0) {
            int64_t* out = P.stats + (int64_t)blockIdx.x * 8;
            const int64_t gems = p1 & 0xFFF, exits = (p1 >> 12) & 0xFFF, died = (p1 >> 24) & 0xFFF;
            const int64_t invalid = (p1 >> 36) & 0xFFF, resets = (p1 >> 48) & 0xFFF;
            const int64_t steps = p2 & 0xFFF, bonus = (p2 >> 12) & 0xFFF;
            out[0] += steps; out[1] += steps * A; out[2] += gems; out[3] += exits; out[4] += died;
            out[5] += invalid; out[6] += resets; out[7] += gems + exits - died + bonus;
        }
    }

    // ---- phase 2: layered observation, one environment of the wave at a time
    if (MODE == MODE_STEP && (K.flags & STEP_NO_OBS)) return;
    if (!hdr->obs_supported) return;
    const uint32_t D = hdr->D, n_chunks = hdr->n_chunks;
    const uint64_t obs_stride = hdr->obs_stride;
    const int64_t n_here = (P.n_envs - env0) < (int64_t)epw ? (P.n_envs - env0) : (int64_t)epw;
    for (int64_t k = 0; k < n_here; k++) {
        const uint32_t* sc = scratch + (uint32_t)k * scr_stride;
        // (a) bytes that depend on beams / gems
        for (uint32_t d = lane; d < D; d += 64) {
            const uint64_t e = dyn[d];
            const uint32_t idx = (uint32_t)e & 0xFFFFFu;
            int32_t v = (int8_t)(uint8_t)(e >> 20);
            const uint32_t n_refs = (uint32_t)(e >> 28) & 3u;
            const uint32_t r0 = (uint32_t)(e >> 30) & 0x3FFu, r1 = (uint32_t)(e >> 40) & 0x3FFu;
            const uint32_t gem = (uint32_t)(e >> 50) & 63u;
            if (n_refs >= 1 && ((sc[r0 & 31u] >> (r0 >> 5)) & 1u)) v = 1;
            if (n_refs >= 2 && ((sc[r1 & 31u] >> (r1 >> 5)) & 1u)) v = 1;
            if (gem != NO_GEM && !((sc[L] >> gem) & 1u)) v = 1;
            tmpl[idx] = (int8_t)v;
        }
        // (b) agents (dead ones included, observations.py:264-265)
        uint32_t agent_idx = 0;
        if ((int)lane < A) {
            agent_idx = sc[L + 1 + lane];
            tmpl[agent_idx] = 1;
        }
        __syncthreads();
        // (c) stream the patched copy: 16 B per lane, 1 KiB contiguous per wave instruction
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(P.obs + (uint64_t)(env0 + k) * obs_stride);
        const uint4* srcv = reinterpret_cast<const uint4*>(tmpl);
        for (uint32_t c = lane; c < n_chunks; c += 64) dst[c] = srcv[c];
        __syncthreads();
        // (d) agents off again (their layers are all-zero in the static copy)
        if ((int)lane < A) tmpl[agent_idx] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ launchers
template <int AM, int LM>
static hipError_t launch_mode(int mode, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_blocks, uint32_t lds_bytes,
                              hipStream_t stream) {
    dim3 grid(n_blocks), block(64);
    switch (mode) {
        case MODE_STEP: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_STEP>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_RESET: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_RESET>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_SET_STATE: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_SET_STATE>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_OBSERVE: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_OBSERVE>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_SOURCES: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_SOURCES>), grid, block, lds_bytes, stream, P, K); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

int kernel_variant(int A, int L) {
    if (A <= 4 && L <= 4) return 0;
    if (A <= 8 && L <= 8) return 1;
    if (A <= 16 && L <= 16) return 2;
    return 3;
}

const char* kernel_variant_name(int variant) {
    static const char* names[4] = {"world_kernel<4,4>", "world_kernel<8,8>", "world_kernel<16,16>", "world_kernel<16,32>"};
    return names[variant & 3];
}

uint32_t kernel_lds_bytes(const MapHeader& h) {
    const uint32_t scr_stride = (h.L + h.A + 1) | 1u;
    return h.lds_table_bytes + 64 * scr_stride * 4 + 64;
}

hipError_t launch_world_kernel(int mode, const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K, hipStream_t stream) {
    const uint32_t epw = K.envs_per_wave;
    const uint32_t n_blocks = (uint32_t)((P.n_envs + epw - 1) / epw);
    const uint32_t lds = kernel_lds_bytes(h);
    switch (kernel_variant((int)h.A, (int)h.L)) {
        case 0: return launch_mode<4, 4>(mode, P, K, n_blocks, lds, stream);
        case 1: return launch_mode<8, 8>(mode, P, K, n_blocks, lds, stream);
        case 2: return launch_mode<16, 16>(mode, P, K, n_blocks, lds, stream);
        default: return launch_mode<16, 32>(mode, P, K, n_blocks, lds, stream);
    }
}

}  // namespace lle
