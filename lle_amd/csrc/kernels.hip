// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched World.
//
// One workgroup = up to four 64-lane wavefronts sharing the map tables in LDS.  Phase 1 (lane = environment): load the env's packed state, run the
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

// The workgroup IS one wavefront, and a wave's LDS operations execute in issue order, so lanes of the wave may hand
// data to each other through LDS without `s_barrier` and without the `s_waitcnt vmcnt(0)` that `__syncthreads()`
// emits (which would stall every environment of phase 2 on the completion of the previous environment's global
// stores).  wave_sync() only pins the compiler's ordering.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t lo = __shfl_xor((uint32_t)v, o, 64);
        uint32_t hi = __shfl_xor((uint32_t)(v >> 32), o, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
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

template <int AM, int LM, int MODE>
__global__ void __launch_bounds__(256) world_kernel(BatchPtrs P, LaunchArgs K) {
    // The uniform map constants are read from the head of the table blob (device memory, scalar loads).  Passing the
    // 400-byte header by value in the kernel-argument segment measured ~2.4 us SLOWER per launch: the kernarg
    // segment is fetched with a much longer latency than device memory.
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(P.tables);
    // A workgroup is 1, 2 or 4 wavefronts that share ONE copy of the cell / dyn tables in LDS (a quarter of the L2
    // traffic and of the copy latency of a per-wave copy); everything else is private to a wavefront.
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t wave_id = blockIdx.x * waves_per_wg + wave_in_wg;
    const int A = (int)hdr->A, L = (int)hdr->L;
    const uint32_t epw = K.envs_per_wave;
    const int64_t env0 = K.env_base + (int64_t)wave_id * epw;
    const int64_t env = env0 + lane;
    const bool active = lane < epw && env < K.env_limit;
    const bool write_obs = hdr->obs_supported && !(K.flags & STEP_NO_OBS);
    const int64_t n_here = (K.env_limit - env0) < (int64_t)epw ? (K.env_limit - env0) : (int64_t)epw;

#define LLE_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (K.stamps && lane == 0) K.stamps[(uint64_t)wave_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    LLE_STAMP(0);
    // ---- static tables -> LDS, once per workgroup (section offsets are those of the blob).  The section is a whole
    // number of 1 KiB rows; every thread requests all of its rows (up to four) before the first LDS write.
    const uint32_t tab_bytes = hdr->lds_table_bytes, tab_off = hdr->off_cell_lay;
    {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(P.tables + tab_off) + lane;
        uint4* dst = reinterpret_cast<uint4*>(lds) + lane;
        const uint32_t rows = tab_bytes / 1024;
        for (uint32_t r0 = wave_in_wg; r0 < rows; r0 += 4 * waves_per_wg) {
            uint4 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (r0 + q * waves_per_wg < rows) v[q] = src[(r0 + q * waves_per_wg) * 64];
            __builtin_amdgcn_sched_barrier(0);  // keep the loads together, ahead of the LDS writes
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (r0 + q * waves_per_wg < rows) dst[(r0 + q * waves_per_wg) * 64] = v[q];
        }
    }
    LLE_STAMP(7);
    __syncthreads();  // the only workgroup barrier: nothing is in flight yet but the loads above
    // ---- the env's packed state and (for auto-reset) the reset-state record, requested raw and together.  (Issued
    // after the table copy on purpose: at the very start of a launch the memory system is still draining the
    // previous launch's stores and cold loads issued then complete later than loads issued after the L2-resident
    // table copy.)
    Env<AM, LM> s;
    uint32_t avail[AM];
    uint32_t raw_pos[AM / 2], raw_avail[AM / 4];
    uint64_t raw_bits = 0;
    if (active) {
        const uint32_t* __restrict__ wp = reinterpret_cast<const uint32_t*>(P.pos) + env * (AM / 2);
#pragma unroll
        for (int k = 0; k < AM / 2; k++) raw_pos[k] = wp[k];
        if (MODE == MODE_STEP) {
            const uint32_t* __restrict__ wa = reinterpret_cast<const uint32_t*>(P.avail) + env * (AM / 4);
#pragma unroll
            for (int k = 0; k < AM / 4; k++) raw_avail[k] = wa[k];
        }
        raw_bits = P.bits[env];
        s.gems = P.gems[env];
#pragma unroll
        for (int b = 0; b < LM; b++) s.beams[b] = (b < L) ? P.beams[env * L + b] : 0u;
    }
    InitRecord init;
    if (MODE == MODE_STEP) init = *P.init;
    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds);
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (hdr->off_cell_meta - tab_off));
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + (hdr->off_dyn - tab_off));
    // private to this wavefront: a patchable copy of the static observation and the phase-1 -> phase-2 hand-over
    const uint32_t scr_stride = (uint32_t)(L + A + 2) | 1u;  // odd: lanes spread over banks
    const uint32_t priv_bytes = hdr->obs_stride + 64u * scr_stride * 4u;
    int8_t* tmpl = reinterpret_cast<int8_t*>(lds + tab_bytes + wave_in_wg * priv_bytes);
    uint32_t* scratch = reinterpret_cast<uint32_t*>(tmpl + hdr->obs_stride);
    const uint64_t obs_stride = hdr->obs_stride;
    {
        const uint4* pristine = reinterpret_cast<const uint4*>(lds + (hdr->off_template - tab_off));
        uint4* mine = reinterpret_cast<uint4*>(tmpl);
        for (uint32_t c = lane; c < hdr->n_chunks; c += 64) mine[c] = pristine[c];
    }
    wave_sync();
    LLE_STAMP(1);
    if (active) {
#pragma unroll
        for (int a = 0; a < AM; a++) {
            const uint32_t p16 = (raw_pos[a >> 1] >> (16 * (a & 1))) & 0xFFFFu;
            s.pos[a] = (a < A) ? p16 : 0xFFFF0000u + (uint32_t)a;  // unused slots: distinct off-grid sentinels
            if (MODE == MODE_STEP) avail[a] = (raw_avail[a >> 2] >> (8 * (a & 3))) & 0xFFu;
        }
        s.alive = (uint32_t)raw_bits & 0xFFFFu; s.arrived = (uint32_t)(raw_bits >> 16) & 0xFFFFu; s.occ = (uint32_t)(raw_bits >> 32) & 0xFFFFu;
    }

    MapView mv;
    mv.cell_lay = cell_lay; mv.cell_meta = cell_meta; mv.hdr = hdr;
    mv.W = (int)hdr->W; mv.A = A; mv.L = L; mv.G = (int)hdr->G;
    mv.enabled = hdr->enabled_mask; mv.max_layers = hdr->max_layers;
    const uint32_t amask = (1u << A) - 1u;
    uint64_t stat1 = 0, stat2 = 0;  // packed per-env counters, summed over the wave below

    LLE_STAMP(2);
    if (active) {
        bool store_state = true, store_avail = false, touched = true;
        Events<AM> ev;
        ev.clear();
        uint32_t err = 0, was_reset = 0;

        if (MODE == MODE_STEP) {
            if (K.flags & STEP_AUTO_RESET) {
                // a finished env restarts from the reset state: identical for every env of the map, computed once on
                // the device into P.init (uniform scalar loads + selects instead of re-running World::reset per lane)
                const bool over = s.alive != amask || s.arrived == amask;
                const InitRecord* in0 = &init;
                const uint64_t ib = in0->bits;
#pragma unroll
                for (int a = 0; a < AM; a++) {
                    if (a < A) {
                        s.pos[a] = over ? (uint32_t)in0->pos[a] : s.pos[a];
                        avail[a] = over ? (uint32_t)in0->avail[a] : avail[a];
                    }
                }
                s.alive = over ? ((uint32_t)ib & 0xFFFFu) : s.alive;
                s.arrived = over ? ((uint32_t)(ib >> 16) & 0xFFFFu) : s.arrived;
                s.occ = over ? ((uint32_t)(ib >> 32) & 0xFFFFu) : s.occ;
                s.gems = over ? in0->gems : s.gems;
#pragma unroll
                for (int b = 0; b < LM; b++)
                    if (b < L) s.beams[b] = over ? in0->beams[b] : s.beams[b];
                was_reset = over ? 1u : 0u;
            }
            uint32_t act[AM];
            if (K.flags & STEP_SAMPLE_ACTIONS) {
                const uint64_t he = action_hash_env(K.seed, (uint64_t)(K.env_offset + env), K.t);
                uint64_t hg = 0;
#pragma unroll
                for (int a = 0; a < AM; a++) {
                    if ((a & 3) == 0 && a < A) hg = action_hash_group(he, (uint64_t)(a >> 2));
                    act[a] = (a < A) ? sample_action(avail[a], action_field(hg, (uint32_t)a)) : 4u;
                }
                store_u8_record<AM>(P.actions, env, act);
            } else if (K.actions_in) {
                // caller's buffer: contiguous [n][A] bytes
                if (A == AM) {
                    load_u8_record<AM>(K.actions_in, env, act);
                } else {
#pragma unroll
                    for (int a = 0; a < AM; a++) act[a] = (a < A) ? (uint32_t)K.actions_in[env * A + a] : 4u;
                }
                store_u8_record<AM>(P.actions, env, act);
            } else {
                load_u8_record<AM>(P.actions, env, act);
            }
            Cells<AM> cur;
            load_cells<AM>(mv, s.pos, cur);
            // availability check: lowest offending agent (world.rs:444-453), before any mutation
#pragma unroll
            for (int a = AM - 1; a >= 0; a--) {
                if (a < A) {
                    // `avail` is the cached list of the reference (world.rs:444-453).  It can only disagree with the
                    // static walk mask after a failed set_state left it stale (world.rs:588-594 returns before
                    // recomputing it); the reference would then index out of the grid and panic, we refuse the action.
                    const uint32_t walk = ((cur.meta[a] >> 8) & 15u) | 16u;
                    const bool bad = act[a] > 4u || !(((avail[a] & walk) >> (act[a] & 7u)) & 1u);
                    err = bad ? (uint32_t)a + 1u : err;
                }
            }
            if (err == 0) {
                Cells<AM> fin;
                step_env<AM, LM>(s, act, mv, ev, cur, fin);
                compute_avail<AM, LM>(s, mv, fin, avail);
                store_avail = true;
            } else {
                store_state = was_reset != 0;
                store_avail = was_reset != 0;
            }
        } else if (MODE == MODE_RESET) {
            if (!K.env_mask || K.env_mask[env]) {
                Cells<AM> at;
                reset_env<AM, LM>(s, mv, at);
                compute_avail<AM, LM>(s, mv, at, avail);
                store_avail = true;
            } else {
                store_state = false;
                touched = false;
            }
        } else if (MODE == MODE_SET_STATE) {
            uint32_t rp[AM];
            {
                const uint32_t* __restrict__ wr = reinterpret_cast<const uint32_t*>(P.req_pos) + env * (AM / 2);
#pragma unroll
                for (int a = 0; a < AM; a++) rp[a] = (a < A) ? ((wr[a >> 1] >> (16 * (a & 1))) & 0xFFFFu) : 0xFFFF0000u + (uint32_t)a;
            }
            bool dirty = false;
            Cells<AM> at;
            err = set_state_env<AM, LM>(s, rp, P.req_gems[env], (uint32_t)P.req_alive[env], mv, ev, dirty, at);
            if (err != 0) ev.clear();
            if (dirty) { load_cells<AM>(mv, s.pos, at); compute_avail<AM, LM>(s, mv, at, avail); store_avail = true; }
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

        LLE_STAMP(3);
        if (store_state) {
            store_u16_record<AM>(P.pos, env, s.pos);
            P.bits[env] = (uint64_t)s.alive | ((uint64_t)s.arrived << 16) | ((uint64_t)s.occ << 32);
            P.gems[env] = s.gems;
#pragma unroll
            for (int b = 0; b < LM; b++)
                if (b < L) P.beams[env * L + b] = s.beams[b];
        }
        if (store_avail) store_u8_record<AM>(P.avail, env, avail);
        if ((MODE == MODE_STEP || MODE == MODE_RESET || MODE == MODE_SET_STATE) && touched) {
            P.err[env] = (uint8_t)err;
            P.evcount[env] = (uint8_t)(ev.n | (was_reset << 7));
            {
                uint32_t* __restrict__ w = reinterpret_cast<uint32_t*>(P.events) + env * (AM / 2);
#pragma unroll
                for (int k = 0; k < AM / 2; k++) w[k] = (uint32_t)(ev.w[k >> 1] >> ((k & 1) * 32));
            }
            P.done[env] = (s.alive != amask || s.arrived == amask) ? 1 : 0;
        }

        // hand the dynamic state to phase 2: [0 | beam masks | ~gem bits | byte index of each agent in the observation]
        uint32_t* sc = scratch + lane * scr_stride;
        sc[0] = 0u;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) sc[1 + b] = s.beams[b];
        sc[L + 1] = ~s.gems;
#pragma unroll
        for (int a = 0; a < AM; a++)
            if (a < A) sc[L + 2 + a] = (uint32_t)a * hdr->HW + cell_of(s.pos[a], mv.W);

        if (MODE == MODE_STEP) {
            // event bytes are type << 4 | agent: DIED sets bit 5, GEM bit 4, EXIT neither
            uint32_t n_died = 0, n_gem = 0;
#pragma unroll
            for (int k = 0; k < Events<AM>::NW; k++) {
                n_died += (uint32_t)__popcll(ev.w[k] & 0x2020202020202020ull);
                n_gem += (uint32_t)__popcll(ev.w[k] & 0x1010101010101010ull);
            }
            const uint32_t n_exit = ev.n - n_died - n_gem;
            const uint32_t bonus = (err == 0 && s.arrived == amask) ? 1u : 0u;
            stat1 = (uint64_t)n_gem | ((uint64_t)n_exit << 12) | ((uint64_t)n_died << 24) |
                    ((uint64_t)(err != 0) << 36) | ((uint64_t)was_reset << 48);
            stat2 = 1ull | ((uint64_t)bonus << 12);
        }
    }
    wave_sync();

    LLE_STAMP(4);
    // ---- phase 2: layered observation, one environment of the wave at a time
    if (write_obs && n_here > 0) {
    const uint32_t D = hdr->D, n_chunks = hdr->n_chunks;
    // Each lane serves the same dyn entry for every environment: decode it once.
    // A laser / gem reference becomes (dword of the hand-over record, bit); an absent one points at the record's
    // zero word, so the per-environment evaluation is branch-free.
    const bool has_d0 = lane < D;
    const uint64_t e0 = has_d0 ? dyn[lane] : 0ull;
    const uint32_t d0_idx = (uint32_t)e0 & 0xFFFFFu;
    const int32_t d0_base = (int8_t)(uint8_t)(e0 >> 20);
    const uint32_t d0_refs = (uint32_t)(e0 >> 28) & 3u, d0_gem = (uint32_t)(e0 >> 50) & 63u;
    const uint32_t d0_r0 = (uint32_t)(e0 >> 30) & 0x3FFu, d0_r1 = (uint32_t)(e0 >> 40) & 0x3FFu;
    const uint32_t d0_w0 = d0_refs >= 1 ? 1u + (d0_r0 & 31u) : 0u, d0_s0 = d0_refs >= 1 ? d0_r0 >> 5 : 0u;
    const uint32_t d0_w1 = d0_refs >= 2 ? 1u + (d0_r1 & 31u) : 0u, d0_s1 = d0_refs >= 2 ? d0_r1 >> 5 : 0u;
    const uint32_t d0_wg = d0_gem != NO_GEM ? (uint32_t)L + 1u : 0u, d0_sg = d0_gem != NO_GEM ? (d0_gem & 31u) : 0u;
    const bool is_agent_lane = (int)lane < A;
    const uint4* srcv = reinterpret_cast<const uint4*>(tmpl);

    for (int64_t k = 0; k < n_here; k++) {
        const uint32_t* sc = scratch + (uint32_t)k * scr_stride;
        // (a) bytes that depend on beams / gems
        {
            const uint32_t lit = ((sc[d0_w0] >> d0_s0) | (sc[d0_w1] >> d0_s1) | (sc[d0_wg] >> d0_sg)) & 1u;
            if (has_d0) tmpl[d0_idx] = (int8_t)(lit ? 1 : d0_base);
        }
        for (uint32_t d = lane + 64u; d < D; d += 64) {  // maps with more than 64 dynamic bytes
            const uint64_t e = dyn[d];
            const uint32_t refs = (uint32_t)(e >> 28) & 3u, gem = (uint32_t)(e >> 50) & 63u;
            const uint32_t r0 = (uint32_t)(e >> 30) & 0x3FFu, r1 = (uint32_t)(e >> 40) & 0x3FFu;
            const uint32_t w0 = refs >= 1 ? 1u + (r0 & 31u) : 0u, w1 = refs >= 2 ? 1u + (r1 & 31u) : 0u;
            const uint32_t wg = gem != NO_GEM ? (uint32_t)L + 1u : 0u;
            const uint32_t lit = ((sc[w0] >> (refs >= 1 ? r0 >> 5 : 0u)) | (sc[w1] >> (refs >= 2 ? r1 >> 5 : 0u)) |
                                  (sc[wg] >> (gem != NO_GEM ? (gem & 31u) : 0u))) & 1u;
            tmpl[(uint32_t)e & 0xFFFFFu] = (int8_t)(lit ? 1 : (int32_t)(int8_t)(uint8_t)(e >> 20));
        }
        // (b) agents (dead ones included, observations.py:264-265)
        const uint32_t agent_idx = is_agent_lane ? sc[L + 2 + lane] : 0u;
        if (is_agent_lane) tmpl[agent_idx] = 1;
        wave_sync();
        // (c) stream the patched copy as one contiguous row: 16 B per lane, 1 KiB per wave instruction
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(P.obs + (uint64_t)(env0 + k) * obs_stride);
        {
            const uint32_t c0 = lane, c1 = lane + 64u;
            const uint4 v0 = srcv[c0 < n_chunks ? c0 : 0u], v1 = srcv[c1 < n_chunks ? c1 : 0u];
            // plain stores: `nt` measured 40 % slower and `sc0 sc1` (write-through) no faster at 65 536 envs and
            // 50 % slower at 262 144
            if (c0 < n_chunks) dst[c0] = v0;
            if (c1 < n_chunks) dst[c1] = v1;
        }
        for (uint32_t c = lane + 128u; c < n_chunks; c += 64) dst[c] = srcv[c];  // rows longer than 2 KiB
        wave_sync();
        // (d) agents off again (their layers are all-zero in the static copy); LDS is in order, so this lands after
        // the reads above and before the next environment's patches
        if (is_agent_lane) tmpl[agent_idx] = 0;
    }
    }
    LLE_STAMP(5);
    if (MODE == MODE_STEP) {
        // per-wave partial counters, last so that their read-modify-write latency is off the observation's path;
        // the slot of this wave is private, so no atomics
        const uint64_t p1 = wave_sum_u64(stat1);
        const uint64_t p2 = wave_sum_u64(stat2);
        if (lane == 0) {
            int64_t* out = P.stats + (int64_t)wave_id * 8;
            const int64_t gems = p1 & 0xFFF, exits = (p1 >> 12) & 0xFFF, died = (p1 >> 24) & 0xFFF;
            const int64_t invalid = (p1 >> 36) & 0xFFF, resets = (p1 >> 48) & 0xFFF;
            const int64_t steps = p2 & 0xFFF, bonus = (p2 >> 12) & 0xFFF;
            out[0] += steps; out[1] += steps * A; out[2] += gems; out[3] += exits; out[4] += died;
            out[5] += invalid; out[6] += resets; out[7] += gems + exits - died + bonus;
        }
    }


    if (K.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LLE_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
template <int AM, int LM>
static hipError_t launch_mode(int mode, const BatchPtrs& P, const LaunchArgs& K, const MapHeader& H, uint32_t n_waves,
                              uint32_t waves_per_wg, uint32_t lds_bytes, hipStream_t stream) {
    dim3 grid((n_waves + waves_per_wg - 1) / waves_per_wg), block(64 * waves_per_wg);
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

int agent_stride(int A, int L) {
    static const int strides[4] = {4, 8, 16, 16};
    return strides[kernel_variant(A, L)];
}

const char* kernel_variant_name(int variant) {
    static const char* names[4] = {"world_kernel<4,4>", "world_kernel<8,8>", "world_kernel<16,16>", "world_kernel<16,32>"};
    return names[variant & 3];
}

uint32_t kernel_lds_bytes(const MapHeader& h, uint32_t waves_per_wg) {
    const uint32_t scr_stride = (h.L + h.A + 2) | 1u;
    return h.lds_table_bytes + waves_per_wg * (h.obs_stride + 64 * scr_stride * 4) + 64;
}

// wavefronts per workgroup: as many (4, 2, 1) as keep the workgroup's LDS under 64 KiB
uint32_t kernel_waves_per_wg(const MapHeader& h) {
    for (uint32_t w = 4; w > 1; w >>= 1)
        if (kernel_lds_bytes(h, w) <= 64 * 1024) return w;
    return 1;
}

hipError_t launch_world_kernel(int mode, const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K, hipStream_t stream) {
    const uint32_t epw = K.envs_per_wave;
    const uint32_t n_waves = (uint32_t)((K.env_limit - K.env_base + epw - 1) / epw);
    const uint32_t wpw = kernel_waves_per_wg(h);
    const uint32_t lds = kernel_lds_bytes(h, wpw);
    switch (kernel_variant((int)h.A, (int)h.L)) {
        case 0: return launch_mode<4, 4>(mode, P, K, h, n_waves, wpw, lds, stream);
        case 1: return launch_mode<8, 8>(mode, P, K, h, n_waves, wpw, lds, stream);
        case 2: return launch_mode<16, 16>(mode, P, K, h, n_waves, wpw, lds, stream);
        default: return launch_mode<16, 32>(mode, P, K, h, n_waves, wpw, lds, stream);
    }
}

}  // namespace lle
