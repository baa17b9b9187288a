// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched World.
//
//   step_kernel<G, LM, MODE, ML1> World.step, the hot path: one LANE PER AGENT (G lanes per environment, 64/G environments
//                                per wavefront), state machine on bitmasks with DPP / ds_swizzle group reductions.
//   world_kernel<AM, LM, MODE>   one lane per environment, the state machine of step_logic.hpp: reset, set_state,
//                                observe, source updates (and step, as a diagnostic).
//
// Both: a workgroup = up to four 64-lane wavefronts sharing ONE copy of the map tables in LDS.  Phase 1: load the
// packed state, run the state machine in registers, store state / events / availability masks.  Phase 2 (wave = one
// environment at a time): the wave owns a private LDS copy of the map's static observation; for each of its
// environments it patches the few dynamic bytes (laser on/off bits, gems, agents) into that copy and streams it to HBM
// with one 16-byte store per lane, i.e. 1 KiB fully coalesced per wave instruction (obs_stream.hpp).
// The observation is >= 95 % of all bytes moved, so phase 2 is what the HBM roofline measures.
//
// No MFMA: there is no contraction anywhere on this path.
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
__device__ __forceinline__ void flush_stats(int64_t* __restrict__ stats, uint32_t wave_id, const StepCounts& c, int A, uint32_t lane) {
    const int64_t steps = wave_sum_u32(c.steps), gems = wave_sum_u32(c.gems), exits = wave_sum_u32(c.exits);
    const int64_t died = wave_sum_u32(c.died), invalid = wave_sum_u32(c.invalid), resets = wave_sum_u32(c.resets);
    const int64_t bonus = wave_sum_u32(c.bonus);
    if (lane == 0) {
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

template <int AM, int LM, int MODE>
__global__ void __launch_bounds__(256) world_kernel(BatchPtrs P, LaunchArgs K) {
    // The uniform map constants are read from the head of the table blob (device memory, scalar loads).  Passing the
    // 400-byte header by value in the kernel-argument segment measured ~2.4 us SLOWER per launch: the kernarg
    // segment is fetched with a much longer latency than device memory.
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t map_idx = map_index_of(K, K.env_base + (int64_t)(blockIdx.x * (blockDim.x >> 6)) * K.envs_per_wave);
    const uint8_t* __restrict__ tables = P.tables + (uint64_t)map_idx * K.table_stride;  // this workgroup's map
    const InitRecord* __restrict__ initp = P.init + map_idx;
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
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
    const uint32_t tab_bytes = hdr->lds_table_bytes, tab_off = hdr->off_cell_lay;
    copy_tables_to_lds(tables + tab_off, lds, tab_bytes, lane, wave_in_wg, waves_per_wg);
    // per-environment sources (lle_batch_set_sources): the bare static observation + element list follow in LDS
    const bool pes = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;
    const uint32_t ext_bytes = pes ? hdr->ext_bytes : 0u;
    if (pes) copy_tables_to_lds(tables + hdr->off_bare, lds + tab_bytes, ext_bytes, lane, wave_in_wg, waves_per_wg);
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
    if (MODE == MODE_STEP) init = *initp;
    const int CW = src_stride_of(L) / 4;  // colour words per env
    uint32_t env_enabled = hdr->enabled_mask, colw[LM / 4];
#pragma unroll
    for (int q = 0; q < LM / 4; q++) colw[q] = 0;
    if (pes && active && !(K.flags & LAUNCH_ARRAYS_INVALID)) {
        env_enabled = P.src_enabled[env];
#pragma unroll
        for (int q = 0; q < LM / 4; q++)
            if (q < CW) colw[q] = reinterpret_cast<const uint32_t*>(P.src_colour)[env * CW + q];
        if (MODE == MODE_STEP && (K.flags & STEP_AUTO_RESET)) {  // the env's own reset state
#pragma unroll
            for (int a = 0; a < AM; a++)
                if (a < A) { init.pos[a] = P.init_pos[env * AM + a]; init.avail[a] = P.init_avail[env * AM + a]; }
            init.bits = P.init_bits[env];
            init.gems = P.init_gems[env];
#pragma unroll
            for (int b = 0; b < LM; b++)
                if (b < L) init.beams[b] = P.init_beams[env * L + b];
        }
    }
    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds);
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (hdr->off_cell_meta - tab_off));
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + (hdr->off_dyn - tab_off));
    // private to this wavefront: a patchable copy of the static observation and the phase-1 -> phase-2 hand-over
    const uint32_t scr_stride = (uint32_t)(L + A + 2 + (pes ? CW : 0)) | 1u;  // odd: lanes spread over banks
    const uint32_t priv_bytes = hdr->obs_stride + 64u * scr_stride * 4u;
    int8_t* tmpl = reinterpret_cast<int8_t*>(lds + tab_bytes + ext_bytes + wave_in_wg * priv_bytes);
    const int8_t* bare = reinterpret_cast<const int8_t*>(lds + tab_bytes);
    const uint32_t* elems = reinterpret_cast<const uint32_t*>(lds + tab_bytes + (hdr->off_elems - hdr->off_bare));
    uint32_t* scratch = reinterpret_cast<uint32_t*>(tmpl + hdr->obs_stride);
    {
        const uint4* pristine = pes ? reinterpret_cast<const uint4*>(bare) : reinterpret_cast<const uint4*>(lds + (hdr->off_template - tab_off));
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
    mv.enabled = env_enabled; mv.max_layers = hdr->max_layers;
    mv.per_env = pes;
#pragma unroll
    for (int q = 0; q < MAX_SOURCES / 4; q++) mv.colw[q] = q < LM / 4 ? colw[q < LM / 4 ? q : 0] : 0u;
    const uint32_t amask = (1u << A) - 1u;
    StepCounts cnt = {0, 0, 0, 0, 0, 0, 0};  // per-env counters, summed over the wave at the end

    LLE_STAMP(2);
    if (active) {
        bool store_state = true, store_avail = false, touched = true, reset_first = false;
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
                const uint64_t key = action_step_key(K.seed, K.t);
                uint32_t hp = 0;
#pragma unroll
                for (int a = 0; a < AM; a++) {
                    if ((a & 1) == 0 && a < A) hp = action_hash_pair(key, (uint64_t)(K.env_offset + env), (uint32_t)(a >> 1));
                    act[a] = (a < A) ? sample_action(avail[a], action_field(hp, (uint32_t)a)) : 4u;
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
        } else if (MODE == MODE_ENV_SOURCES) {
            // LaserSource.set_colour / enable / disable for this env (pylaser_source.rs:55-75,107-142; laser.rs:69-86)
            // and the env's reset state with the new sources (copied by the auto-reset path of the step kernel)
            const bool fill = (K.flags & LAUNCH_FILL_DEFAULTS) != 0;
            if (!K.env_mask || K.env_mask[env]) {
                if (K.flags & LAUNCH_RESET_FIRST) {
                    // LLE.reset with randomize_lasers (python/lle/env/env.py:189-203): world.reset() under the sources the
                    // env has, THEN the new colours on the live world (beams blocked at reset stay as they are)
                    Cells<AM> at0;
                    reset_env<AM, LM>(s, mv, at0);
                    compute_avail<AM, LM>(s, mv, at0, avail);
                    store_avail = true;
                    reset_first = true;
                }
                uint32_t new_en = fill ? hdr->enabled_mask : (K.enabled_in ? K.enabled_in[env] : mv.enabled);
                new_en &= L >= 32 ? 0xFFFFFFFFu : ((1u << L) - 1u);
                uint32_t ncol[LM / 4];
#pragma unroll
                for (int q = 0; q < LM / 4; q++) ncol[q] = colw[q];
                bool bad = false;
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    if (b < L && (fill || K.colours_in)) {
                        const uint32_t c = fill ? (uint32_t)hdr->beam_colour[b] : (uint32_t)K.colours_in[env * L + b];
                        bad |= !fill && c >= (uint32_t)A;  // "Agent ID is greater than the number of agents"
                        ncol[b >> 2] = (ncol[b >> 2] & ~(0xFFu << ((b & 3) * 8))) | ((c & 0xFFu) << ((b & 3) * 8));
                    }
                }
                err = bad ? ENV_INVALID_COLOUR : 0u;
                if (!bad) {
#pragma unroll
                    for (int b = 0; b < LM; b++) {
                        if (b < L) {
                            const bool was = (mv.enabled >> b) & 1u, now = (new_en >> b) & 1u;
                            if (was && !now) s.beams[b] = 0u;
                            if (!was && now) s.beams[b] = hdr->beam_full[b];
                        }
                    }
                    mv.enabled = new_en;
#pragma unroll
                    for (int q = 0; q < LM / 4; q++) { mv.colw[q] = ncol[q]; colw[q] = ncol[q]; }
                    P.src_enabled[env] = new_en;
#pragma unroll
                    for (int q = 0; q < LM / 4; q++)
                        if (q < CW) reinterpret_cast<uint32_t*>(P.src_colour)[env * CW + q] = ncol[q];
                    Env<AM, LM> r = s;
                    uint32_t ravail[AM];
                    Cells<AM> at;
                    reset_env<AM, LM>(r, mv, at);
                    compute_avail<AM, LM>(r, mv, at, ravail);
                    store_u16_record<AM>(P.init_pos, env, r.pos);
                    store_u8_record<AM>(P.init_avail, env, ravail);
                    P.init_bits[env] = (uint64_t)r.alive | ((uint64_t)r.arrived << 16) | ((uint64_t)r.occ << 32);
                    P.init_gems[env] = r.gems;
#pragma unroll
                    for (int b = 0; b < LM; b++)
                        if (b < L) P.init_beams[env * L + b] = r.beams[b];
                } else {
                    store_state = reset_first;
                }
                P.err[env] = (uint8_t)err;
            } else {
                store_state = false;
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
        if ((MODE == MODE_STEP || MODE == MODE_RESET || MODE == MODE_SET_STATE || (MODE == MODE_ENV_SOURCES && reset_first)) && touched) {
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

        if (pes) {
#pragma unroll
            for (int q = 0; q < LM / 4; q++)
                if (q < CW) sc[L + 2 + A + q] = colw[q];
        }

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
            P.reward[env] = n_gem | (n_exit << 8) | (n_died << 16) | (bonus << 24);
            cnt = StepCounts{1u, n_gem, n_exit, n_died, err != 0 ? 1u : 0u, was_reset, bonus};
        }
    }
    wave_sync();

    LLE_STAMP(4);
    // ---- phase 2: layered observation, one environment of the wave at a time
    if (write_obs && n_here > 0) {
        const bool wt = (K.flags & LAUNCH_WRITE_THROUGH) != 0;
        if (pes) {
            if (wt) write_observations_env<true>(A, L, hdr->HW, hdr->n_elems, hdr->n_chunks, hdr->obs_stride, elems, bare, tmpl, scratch,
                                                 scr_stride, P.obs, env0, n_here, lane);
            else write_observations_env<false>(A, L, hdr->HW, hdr->n_elems, hdr->n_chunks, hdr->obs_stride, elems, bare, tmpl, scratch,
                                               scr_stride, P.obs, env0, n_here, lane);
        } else {
            if (wt) write_observations<true>(A, L, hdr->D, hdr->n_chunks, hdr->obs_stride, dyn, tmpl, scratch, scr_stride, P.obs, env0, n_here, lane);
            else write_observations<false>(A, L, hdr->D, hdr->n_chunks, hdr->obs_stride, dyn, tmpl, scratch, scr_stride, P.obs, env0, n_here, lane);
        }
    }
    LLE_STAMP(5);
    if (MODE == MODE_STEP) flush_stats(P.stats, wave_id, cnt, A, lane);

    if (K.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LLE_STAMP(6);
    }
}

// ================================================================================================
// step_kernel<G, LM>: World.step() with one LANE PER AGENT (G = lanes per environment = power of two >= A).
//
// world_kernel<.., MODE_STEP> above runs one environment per lane: simple, but phase 1 is then a ~4k-instruction
// dependency chain of ONE wave that no amount of occupancy shortens, and every wave runs it at the same time
// (launch cost = chain + observation stream).  Here the agents of an environment sit in G neighbouring lanes, so the
// per-agent loops of move_agents / compute_available_actions / the sampler become lane-parallel and the chain is
// ~G times shorter; a 64-lane wave carries 64/G environments and four times as many waves share each SIMD.
//
// What makes the split legal (same results as the sequential reference, src/core/world.rs:477-505):
//   * leaves of one pass commute (each only turns bits ON, and a leave skipped because an earlier one already lit its
//     bit would have been a no-op): beam |= OR over the group of every lane's suffix;
//   * pre-enters commute (each only clears a suffix): beam &= AND over the group of every lane's prefix;
//   * enter reads the beams (final after leave+pre-enter) and touches only the agent's own flags, its own cell's
//     gem and the occupant slot of its own cell (agents never share a cell), so the enters of one pass are
//     independent; their events are ordered by agent id = lane order (prefix count inside the group);
//   * the three loops stay in the reference's order, and passes repeat while any agent of the environment died.
// Cross-lane traffic is DPP quad permutes (G <= 4) / ds_swizzle (G = 8, 16); nothing goes through memory.

template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    static_assert(J >= 1 && J < 16, "group offsets only");
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    if (J == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, true);  // quad_perm [3,2,1,0]
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (J << 10));                     // lane ^ J within 32 lanes
}
template <int G>
__device__ __forceinline__ uint32_t grp_or(uint32_t v) {
    if (G > 1) v |= lane_xor<1>(v);
    if (G > 2) v |= lane_xor<2>(v);
    if (G > 4) v |= lane_xor<4>(v);
    if (G > 8) v |= lane_xor<8>(v);
    return v;
}
template <int G>
__device__ __forceinline__ uint64_t grp_or64(uint64_t v) {
    return (uint64_t)grp_or<G>((uint32_t)v) | ((uint64_t)grp_or<G>((uint32_t)(v >> 32)) << 32);
}

// value of `v` in the group lane whose agent id is (a ^ J), for every J in 1..G-1, fed to f(other_agent_offset J, value)
template <int G, int J = 1, typename F>
__device__ __forceinline__ void for_each_other(uint32_t v, F&& f) {
    if constexpr (J < G) {
        f(J, lane_xor<J>(v));
        for_each_other<G, J + 1>(v, f);
    }
}

// GEN: the general instantiation -- environments with their own source colours / enabled flags (lle_batch_set_sources)
// and/or batches of several maps (lle_batch_create_multi).  A separate instantiation, so that the default path (one
// map, sources of the map) is compiled exactly as before: every `PES` / `tables` / `initp` below folds to a constant.
// ML1: no cell of the map carries more than one laser layer (every level of the reference; no crossing beams): the
// per-layer loops run exactly once and unroll (no variable 64-bit shifts of the layer word).
// MODE 0: one step in place, one map, the map's sources (the default).  MODE 1: + fused rollout (n_steps, trajectory
// rings) and timeline stamps.  MODE 2 (general): + several maps.  MODE 3: + per-env sources (a mode of its own, not a
// run-time flag of MODE 2, so that neither keeps the other's reset state and colour words in registers: both are short
// of them).
// LX >= 0: the exact number of sources, known at compile time (instantiated for the default path of maps with at
// most four sources: the per-beam loops lose their guards and the unused beam registers disappear; 0.4 us on level 6).
template <int G, int LM, int MODE, bool ML1, int LX = -1>
__global__ void __launch_bounds__(256, 4) step_kernel(BatchPtrs P, LaunchArgs K) {
    constexpr bool GEN = MODE >= 2, ROLL = MODE >= 1;
    constexpr bool PES = MODE == 3;  // (the launcher picks MODE 3 exactly when LAUNCH_PER_ENV_SOURCES is set)
    // The default instantiation is one step in place and nothing else: the fused rollout (n_steps, trajectory rings)
    // and the timeline stamps run on the general one, so that their arguments do not occupy scalar registers here.
    uint64_t* const stamps = ROLL ? K.stamps : nullptr;
#undef LLE_STAMP
#define LLE_STAMP(i)                                                                              \
    do {                                                                                          \
        if (ROLL && stamps && lane == 0) stamps[(uint64_t)wave_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    // environments per wavefront: at most 64 / G; fewer (lanes left idle) when the batch is small, so that there
    // are enough wavefronts to spread phase 2 over the chip
    const uint32_t EPW = K.envs_per_wave < (uint32_t)(64 / G) ? K.envs_per_wave : (uint32_t)(64 / G);
    constexpr int NW = (2 * G + 7) / 8;             // 64-bit words of the event list (2 events per agent at most)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t wave_id = blockIdx.x * waves_per_wg + wave_in_wg;
    const uint8_t* __restrict__ tables = P.tables;
    const InitRecord* __restrict__ initp = P.init;
    if (GEN) {  // this workgroup's map (its envs never straddle two maps: the launcher sizes workgroups accordingly)
        const uint32_t EPW0 = K.envs_per_wave < (uint32_t)(64 / G) ? K.envs_per_wave : (uint32_t)(64 / G);
        const uint32_t map_idx = map_index_of(K, K.env_base + (int64_t)(blockIdx.x * waves_per_wg) * EPW0);
        tables += (uint64_t)map_idx * K.table_stride;
        initp += map_idx;
    }
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    const int A = (int)hdr->A, L = LX >= 0 ? LX : (int)hdr->L, W = (int)hdr->W;
    const uint32_t a = lane & (G - 1), grp = lane / G;  // agent id, environment slot in the wave
    const int64_t As = agent_stride_of(A, L);           // env pitch of the per-agent buffers
    const int64_t env0 = K.env_base + (int64_t)wave_id * EPW;
    const int64_t env = env0 + grp;
    const bool env_ok = grp < EPW && env < K.env_limit;
    const bool me = env_ok && (int)a < A;           // this lane carries a real agent
    const bool write_obs = hdr->obs_supported && !(K.flags & STEP_NO_OBS);
    // Header fields that are needed late (after the first stores) are read HERE, as scalar loads next to the kernel
    // arguments: read where they are used they become vector loads from global memory with a full wait each, three of
    // them in a row between the state machine and the first observation store.
    const uint32_t h_HW = hdr->HW, h_obs_stride = hdr->obs_stride, h_n_chunks = hdr->n_chunks, h_D = hdr->D;
    uint32_t h_beam_full[LM];
#pragma unroll
    for (int b = 0; b < LM; b++) h_beam_full[b] = (b < L) ? hdr->beam_full[b] : 0u;
    const uint32_t h_enabled = hdr->enabled_mask;
    uint32_t h_init_beams[LM];  // the reset state's beams (shared record; the per-env one is read where it is used)
#pragma unroll
    for (int b = 0; b < LM; b++) h_init_beams[b] = (b < L) ? initp->beams[b] : 0u;
    const int64_t n_here = (K.env_limit - env0) < (int64_t)EPW ? (K.env_limit - env0) : (int64_t)EPW;
    const uint32_t bit = 1u << a, amask = (1u << A) - 1u;
    LLE_STAMP(0);

    const uint32_t tab_bytes = hdr->lds_table_bytes, tab_off = hdr->off_cell_lay;
    copy_tables_to_lds(tables + tab_off, lds, tab_bytes, lane, wave_in_wg, waves_per_wg);
    const uint32_t ext_bytes = PES ? hdr->ext_bytes : 0u;
    if (PES) copy_tables_to_lds(tables + hdr->off_bare, lds + tab_bytes, ext_bytes, lane, wave_in_wg, waves_per_wg);
    LLE_STAMP(7);
    __syncthreads();  // the only workgroup barrier

    // ---- packed state: own position / availability, and the env-wide words replicated in the group's lanes
    uint32_t pos = 0xFFFF0000u + a, avail = 0, beams[LM];
    uint64_t raw_bits = 0;
    uint32_t gems = 0;
#pragma unroll
    for (int b = 0; b < LM; b++) beams[b] = 0;
    // Per-lane addresses of this env's / agent's records, computed once and kept in vector registers for the stores at
    // the end: the scalar base pointers are then dead during the state machine (scalar registers are the scarce
    // resource of this kernel, vector registers are not).
    const int64_t env_c = env_ok ? env : 0;
    uint64_t* const p_bits = P.bits + env_c;
    uint32_t* const p_gems = P.gems + env_c;
    uint32_t* const p_beams = P.beams + env_c * L;
    uint16_t* const p_pos = P.pos + env_c * As + a;
    uint8_t* const p_avail = P.avail + env_c * As + a;
    uint8_t* const p_err = P.err + env_c;
    uint8_t* const p_evcount = P.evcount + env_c;
    uint8_t* const p_events = P.events + env_c * 2 * As;
    uint8_t* const p_done = P.done + env_c;
    // The general mode is short of VECTOR registers instead (it spilled eight to scratch, and a scratch reload inside
    // the state machine is a memory round trip): there the addresses of the late stores are rebuilt from the scalar
    // bases where they are used, and the per-lane copies above die after the loads.
#define LLE_LATE(field, offset) (GEN ? P.field + (offset) : p_##field)
    if (env_ok) {
        raw_bits = *p_bits;
        gems = *p_gems;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) beams[b] = p_beams[b];
    }
    uint32_t init_pos_a = 0xFFFF0000u + a, init_avail_a = 0;  // this agent's reset position / availability
    if (me) {
        pos = (uint32_t)*p_pos;
        avail = (uint32_t)*p_avail;
        init_pos_a = PES ? (uint32_t)P.init_pos[env * As + a] : (uint32_t)initp->pos[a];
        init_avail_a = PES ? (uint32_t)P.init_avail[env * As + a] : (uint32_t)initp->avail[a];
    }
    uint64_t init_bits = initp->bits;
    uint32_t init_gems = initp->gems;
    // per-environment sources: colours (4 per word), enabled mask, and the env's own reset state
    constexpr int CWM = LM / 4;
    const int CW = src_stride_of(L) / 4;
    uint32_t colw[CWM], env_enabled = h_enabled;
#pragma unroll
    for (int q = 0; q < CWM; q++) colw[q] = 0;
    if (PES && env_ok) {
        env_enabled = P.src_enabled[env];
#pragma unroll
        for (int q = 0; q < CWM; q++)
            if (q < CW) colw[q] = reinterpret_cast<const uint32_t*>(P.src_colour)[env * CW + q];
        if (K.flags & STEP_AUTO_RESET) {
            init_bits = P.init_bits[env];
            init_gems = P.init_gems[env];
        }
    }

    const uint64_t* cell_lay = reinterpret_cast<const uint64_t*>(lds);
    const uint32_t* cell_meta = reinterpret_cast<const uint32_t*>(lds + (hdr->off_cell_meta - tab_off));
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(lds + (hdr->off_dyn - tab_off));
    const uint32_t scr_stride = (uint32_t)(L + A + 2 + (PES ? CW : 0)) | 1u;
    const uint32_t priv_bytes = h_obs_stride + 64u * scr_stride * 4u;
    int8_t* tmpl = reinterpret_cast<int8_t*>(lds + tab_bytes + ext_bytes + wave_in_wg * priv_bytes);
    uint32_t* scratch = reinterpret_cast<uint32_t*>(tmpl + h_obs_stride);
    const int8_t* bare = reinterpret_cast<const int8_t*>(lds + tab_bytes);
    const uint32_t* elems = reinterpret_cast<const uint32_t*>(lds + tab_bytes + (hdr->off_elems - hdr->off_bare));
    {
        const uint4* pristine = PES ? reinterpret_cast<const uint4*>(bare) : reinterpret_cast<const uint4*>(lds + (hdr->off_template - tab_off));
        uint4* mine = reinterpret_cast<uint4*>(tmpl);
        for (uint32_t c = lane; c < h_n_chunks; c += 64) mine[c] = pristine[c];
    }
    wave_sync();
    LLE_STAMP(1);

    uint32_t alive = (uint32_t)raw_bits & 0xFFFFu, arrived = (uint32_t)(raw_bits >> 16) & 0xFFFFu, occ = (uint32_t)(raw_bits >> 32) & 0xFFFFu;
    const uint32_t enabled = PES ? env_enabled : h_enabled, max_layers = ML1 ? 1u : hdr->max_layers;
    LLE_STAMP(2);

    // ---- n_steps consecutive steps of the wave's environments; the state stays in registers in between.
    // (n_steps = 1 is World.step; more is a fused rollout with on-device action sampling, lle_batch_rollout.)
    StepCounts cnt = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t n_steps = ROLL ? (K.n_steps ? K.n_steps : 1u) : 1u;
    for (uint32_t it = 0; it < n_steps; it++) {
    const uint64_t t_now = K.t + it;
    // where this step's observation / actions / reward counts go: in place, or slot (ring_pos + step) % ring_slots of
    // the trajectory rings (the launcher passes ring_pos already reduced modulo ring_slots; the slot advances by
    // increment, so the single-step path carries no 64-bit division)
    uint8_t* __restrict__ actions_out = P.actions;
    uint32_t* __restrict__ reward_out = P.reward;
    int8_t* __restrict__ obs_out = P.obs;
    if (ROLL && K.ring_slots) {
        uint32_t slot = (uint32_t)K.ring_pos + it;
        while (slot >= K.ring_slots) slot -= K.ring_slots;
        actions_out = K.ring_actions + (int64_t)slot * K.ring_env_count * As;
        reward_out = K.ring_reward + (int64_t)slot * K.ring_env_count;
        obs_out = K.ring_obs + (int64_t)slot * K.ring_env_count * (int64_t)h_obs_stride;
    }

    // ---- auto-reset: a finished env restarts from the reset state (identical for every env, see InitRecord)
    uint32_t was_reset = 0;
    if (K.flags & STEP_AUTO_RESET) {
        const bool over = env_ok && (alive != amask || arrived == amask);
        pos = (over && me) ? init_pos_a : pos;
        avail = (over && me) ? init_avail_a : avail;
        alive = over ? ((uint32_t)init_bits & 0xFFFFu) : alive;
        arrived = over ? ((uint32_t)(init_bits >> 16) & 0xFFFFu) : arrived;
        occ = over ? ((uint32_t)(init_bits >> 32) & 0xFFFFu) : occ;
        gems = over ? init_gems : gems;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) beams[b] = over ? (PES ? P.init_beams[env_ok ? env * L + b : 0] : h_init_beams[b]) : beams[b];
        was_reset = over ? 1u : 0u;
    }

    // ---- joint action: sampled on the device, or given
    uint32_t act = 4u;
    if (K.flags & STEP_SAMPLE_ACTIONS) {
        const uint32_t hp = action_hash_pair(action_step_key(K.seed, t_now), (uint64_t)(K.env_offset + env), a >> 1);
        act = sample_action(avail, action_field(hp, a));
        if (me) actions_out[env * As + a] = (uint8_t)act;
    } else if (K.actions_in) {
        if (me) {
            act = (uint32_t)K.actions_in[env * A + a];  // caller's buffer: contiguous [n][A]
            P.actions[env * As + a] = (uint8_t)act;
        }
    } else if (me) {
        act = (uint32_t)P.actions[env * As + a];
    }

    // ---- availability check (world.rs:444-453): lowest offending agent, before any mutation.  The cached list can
    // only disagree with the static walk mask after a failed set_state left it stale; such an action is refused.
    const uint32_t cur_cell = me ? cell_of(pos, W) : 0u;
    uint64_t lay_cur = cell_lay[cur_cell];
    if (PES) lay_cur = recolour_lay<CWM>(lay_cur, colw);
    const uint32_t meta_cur = cell_meta[cur_cell];
    const uint32_t walk_cur = ((meta_cur >> 8) & 15u) | 16u;
    const bool bad = me && (act > 4u || !(((avail & walk_cur) >> (act & 7u)) & 1u));
    const uint32_t badmask = grp_or<G>(bad ? bit : 0u);
    const uint32_t err = badmask ? (uint32_t)__ffs((int)badmask) : 0u;

    uint64_t evw[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) evw[k] = 0;
    uint32_t n_ev = 0;
    uint32_t meta_step = 0;   // cell meta of the agent's new cell, for the availability mask computed in post_step()
    bool stepped = false;

    if (env_ok && err == 0) {
        // target cell (src/action.rs:18-26 on the packed i | j << 8 form); lanes without an agent keep a unique sentinel
        uint32_t np = me ? apply_action(pos, act) : pos;
        // solve_vertex_conflicts (world.rs:365-378): every agent whose target is shared goes back to its cell
        bool again = true;
        while (__any(again)) {
            bool dup = false;
            for_each_other<G>(np, [&](int, uint32_t other) { dup |= (other == np); });
            np = dup ? pos : np;
            again = grp_or<G>(dup ? 1u : 0u) != 0;
        }
        const uint32_t new_cell = me ? cell_of(np, W) : 0u;
        uint64_t lay_new = cell_lay[new_cell];
        if (PES) lay_new = recolour_lay<CWM>(lay_new, colw);
        const uint32_t meta_new = cell_meta[new_cell];
        const uint32_t kind = meta_new & 7u;
        const uint32_t gbit = 1u << ((meta_new >> 3) & 31u);

        // move_agents passes (world.rs:464-472)
        bool go = true;
        bool first_pass = true;
        uint64_t lay_from = lay_cur;  // pass 1 leaves the old cells, later passes the new ones
        while (__any(go)) {
            // leave (laser.rs:199-202,157-162): what the alive agents of the env re-light, per beam
            const uint32_t alive0 = alive;
            const bool me_alive = go && me && (alive0 & bit);
            uint32_t lit[LM], any_lit = 0;
#pragma unroll
            for (int b = 0; b < LM; b++) {
                lit[b] = 0;
                if (b < L) {
                    uint32_t light = 0;
                    for (uint32_t k = 0; k < max_layers; k++) {
                        const uint32_t eo = (uint32_t)(lay_from >> (16 * k)) & 0xFFFFu;
                        const bool lo = me_alive && (eo & LAY_VALID) && ((eo >> 1) & 31u) == (uint32_t)b &&
                                        !((beams[b] >> ((eo >> 6) & 31u)) & 1u);
                        light |= lo ? (0xFFFFFFFFu << ((eo >> 6) & 31u)) : 0u;
                    }
                    lit[b] = grp_or<G>(((enabled >> b) & 1u) ? light : 0u);
                    any_lit |= lit[b];
                }
            }
            // A pass after the first one leaves and re-enters the SAME cells.  If no alive agent re-lights anything, the
            // beams cannot change (the owners' cuts are repeated as they are), so every enter repeats its outcome: alive
            // agents stay alive, occupants / arrivals / gems are already recorded, the dead stay blocked or buried.
            // The pass is then a no-op and `while agent_died` ends (world.rs:468-472).
            if (!first_pass && any_lit == 0u) go = false;
            if (go) {
                occ &= ~alive0;  // Tile::leave: slot.take() for every alive agent
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    if (b < L) {
                        uint32_t keep = 0xFFFFFFFFu;
                        for (uint32_t k = 0; k < max_layers; k++) {
                            const uint32_t en = (uint32_t)(lay_new >> (16 * k)) & 0xFFFFu;   // pre_enter (laser.rs:173-182)
                            const bool pe = me_alive && (en & LAY_VALID) && ((en >> 1) & 31u) == (uint32_t)b && (en >> 11) == a;
                            keep &= pe ? ((1u << ((en >> 6) & 31u)) - 1u) : 0xFFFFFFFFu;
                        }
                        const uint32_t cut = grp_or<G>(((enabled >> b) & 1u) ? ~keep : 0u);
                        beams[b] = (beams[b] | lit[b]) & ~cut;
                    }
                }
                // enter (tile.rs:29-50, laser.rs:184-197)
                bool blocked = false;
                for (uint32_t k = 0; k < max_layers; k++) {
                    const uint32_t en = (uint32_t)(lay_new >> (16 * k)) & 0xFFFFu;
                    const uint32_t m = beam_get<LM>(beams, (en >> 1) & 31u);
                    blocked |= (en & LAY_VALID) && ((m >> ((en >> 6) & 31u)) & 1u) && ((en >> 11) != a);
                }
                const bool is_alive = (alive & bit) != 0;
                const bool inner = me && !blocked;
                const bool ev_exit = inner && kind == K_EXIT && !(arrived & bit);
                const bool ev_gem = inner && kind == K_GEM && !(gems & gbit);
                const bool died = me && is_alive && (blocked || kind == K_VOID);
                const bool has_ev = died || ev_exit || ev_gem;
                const uint32_t p1 = grp_or<G>((died ? bit : 0u) | (ev_exit ? bit << 16 : 0u));
                const uint32_t p2 = grp_or<G>((inner ? bit : 0u) | (has_ev ? bit << 16 : 0u));
                gems |= grp_or<G>(ev_gem ? gbit : 0u);
                alive &= ~(p1 & 0xFFFFu);
                arrived |= p1 >> 16;
                occ |= p2 & 0xFFFFu;
                const uint32_t evmask = p2 >> 16;  // agents with an event this pass: ordered by agent id
                const uint32_t slot = n_ev + (uint32_t)__popc(evmask & (bit - 1u));
                const uint64_t byte = has_ev ? (uint64_t)(((died ? EV_DIED : (ev_gem ? EV_GEM : EV_EXIT)) << 4) | a) : 0ull;
#pragma unroll
                for (int k = 0; k < NW; k++) evw[k] |= (NW == 1 || (slot >> 3) == (uint32_t)k) ? (byte << ((slot & 7u) * 8u)) : 0ull;
                n_ev += (uint32_t)__popc(evmask);
                go = (p1 & 0xFFFFu) != 0;  // while agent_died
            }
            lay_from = lay_new;
            first_pass = false;
        }
        pos = np;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) beams[b] &= h_beam_full[b];
        meta_step = meta_new;
        stepped = true;
    }
    LLE_STAMP(3);

    // ---- everything of the step that the observation does not need: availability masks (compute_available_actions,
    // world.rs:343-363), error code, ordered event list, done flag, reward counts, counters.  The observation needs
    // positions, beams and gems only, so a wavefront of the OLDER half of the grid (the one the SIMD serves first, i.e.
    // the one whose first store ends the idle time of the memory system) does this after its stream, a younger one --
    // which waits for memory anyway and would otherwise add it to the end of the launch -- before.
    auto post_step = [&]() {
    if (stepped) {
        const bool can_move = me && (alive & bit) && !(arrived & bit);
        uint32_t blocked_dirs = 0;
        for_each_other<G>(pos, [&](int j, uint32_t other) {
            const int d = (int)other - (int)pos;
            uint32_t hit = (d == -1) ? 1u : 0u;
            hit |= (d == 1) ? 2u : 0u;
            hit |= (d == 256) ? 4u : 0u;
            hit |= (d == -256) ? 8u : 0u;
            blocked_dirs |= ((occ >> (a ^ (uint32_t)j)) & 1u) ? hit : 0u;
        });
        avail = 16u | (can_move ? (((meta_step >> 8) & 15u) & ~blocked_dirs) : 0u);
    }
#pragma unroll
    for (int k = 0; k < NW; k++) evw[k] = grp_or64<G>(evw[k]);
    if (env_ok && a == 0) {
        *LLE_LATE(err, env_c) = (uint8_t)err;
        *LLE_LATE(evcount, env_c) = (uint8_t)(n_ev | (was_reset << 7));
        {
            uint8_t* row = LLE_LATE(events, env_c * 2 * As);  // 2*As bytes per env; this kernel fills the first 2*G
            if (G >= 2) {
                uint32_t* __restrict__ w = reinterpret_cast<uint32_t*>(row);
#pragma unroll
                for (int k = 0; k < G / 2; k++) w[k] = (uint32_t)(evw[k >> 1] >> ((k & 1) * 32));
            } else {
                *reinterpret_cast<uint16_t*>(row) = (uint16_t)evw[0];
            }
        }
        *LLE_LATE(done, env_c) = (alive != amask || arrived == amask) ? 1 : 0;
        uint32_t n_died = 0, n_gem = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            n_died += (uint32_t)__popcll(evw[k] & 0x2020202020202020ull);
            n_gem += (uint32_t)__popcll(evw[k] & 0x1010101010101010ull);
        }
        const uint32_t n_exit = n_ev - n_died - n_gem;
        const uint32_t bonus = (err == 0 && arrived == amask) ? 1u : 0u;
        reward_out[env] = n_gem | (n_exit << 8) | (n_died << 16) | (bonus << 24);
        cnt.steps += 1u; cnt.gems += n_gem; cnt.exits += n_exit; cnt.died += n_died;
        cnt.invalid += err != 0 ? 1u : 0u; cnt.resets += was_reset; cnt.bonus += bonus;
    }
    };  // post_step

    if (env_ok && a == 0) {
        // hand-over record of this env for phase 2: [0 | beam masks | ~gem bits | ...
        uint32_t* sc = scratch + grp * scr_stride;
        sc[0] = 0u;
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) sc[1 + b] = beams[b];
        sc[L + 1] = ~gems;
        if (PES) {
#pragma unroll
            for (int q = 0; q < CWM; q++)
                if (q < CW) sc[L + 2 + A + q] = colw[q];
        }
    }
    if (me) scratch[grp * scr_stride + L + 2 + a] = a * h_HW + cell_of(pos, W);  // ... | byte index of each agent]
    wave_sync();
    LLE_STAMP(4);
    // (deferring it in the single-step launches of MODE 1 / 2 as well measured 1.3-1.5 us SLOWER there: those
    // instantiations already spill, and the deferral lengthens the live ranges)
    const bool post_first = MODE != 0 || blockIdx.x * 4u >= gridDim.x * 3u;
    if (post_first) post_step();

    if (write_obs && n_here > 0) {
        const bool wt = (K.flags & LAUNCH_WRITE_THROUGH) != 0;  // see stream_store (obs_stream.hpp)
        if (PES) {
            if (wt) write_observations_env<true>(A, L, h_HW, hdr->n_elems, h_n_chunks, h_obs_stride, elems, bare, tmpl, scratch, scr_stride,
                                                 obs_out, env0, n_here, lane);
            else write_observations_env<false>(A, L, h_HW, hdr->n_elems, h_n_chunks, h_obs_stride, elems, bare, tmpl, scratch, scr_stride,
                                               obs_out, env0, n_here, lane);
        } else {
            if (wt) write_observations<true>(A, L, h_D, h_n_chunks, h_obs_stride, dyn, tmpl, scratch, scr_stride, obs_out, env0, n_here, lane);
            else write_observations<false>(A, L, h_D, h_n_chunks, h_obs_stride, dyn, tmpl, scratch, scr_stride, obs_out, env0, n_here, lane);
        }
    }
    wave_sync();
    if (!post_first) post_step();
    }  // steps
    LLE_STAMP(5);

    // ---- final state.  Written unconditionally: an env whose action was refused kept its registers unchanged
    // (world.rs:436-453: errors precede any mutation), so this rewrites the same bytes.
    if (me) {
        *LLE_LATE(pos, env_c * As + a) = (uint16_t)pos;
        *LLE_LATE(avail, env_c * As + a) = (uint8_t)avail;
    }
    if (env_ok && a == 0) {
        *LLE_LATE(bits, env_c) = (uint64_t)alive | ((uint64_t)arrived << 16) | ((uint64_t)occ << 32);
        *LLE_LATE(gems, env_c) = gems;
        uint32_t* const beams_out = LLE_LATE(beams, env_c * L);
#pragma unroll
        for (int b = 0; b < LM; b++)
            if (b < L) beams_out[b] = beams[b];
#undef LLE_LATE
    }
    flush_stats(P.stats, wave_id, cnt, A, lane);
    if (ROLL && stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LLE_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
template <int AM, int LM>
static hipError_t launch_mode(int mode, const BatchPtrs& P, const LaunchArgs& K, const MapHeader& H, uint32_t n_waves,
                              uint32_t waves_per_wg, uint32_t lds_bytes, hipStream_t stream) {
    dim3 grid((n_waves + waves_per_wg - 1) / waves_per_wg), block(64 * waves_per_wg);
    if (lds_bytes > 64 * 1024) {
        static uint32_t granted[6] = {0, 0, 0, 0, 0, 0};
        if (mode >= 0 && mode < 6 && lds_bytes > granted[mode]) {
            const void* fn = mode == MODE_STEP ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_STEP>)
                           : mode == MODE_RESET ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_RESET>)
                           : mode == MODE_SET_STATE ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_SET_STATE>)
                           : mode == MODE_OBSERVE ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_OBSERVE>)
                           : mode == MODE_SOURCES ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_SOURCES>)
                                                  : reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_ENV_SOURCES>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
            granted[mode] = lds_bytes;
        }
    }
    switch (mode) {
        case MODE_STEP: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_STEP>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_RESET: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_RESET>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_SET_STATE: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_SET_STATE>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_OBSERVE: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_OBSERVE>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_SOURCES: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_SOURCES>), grid, block, lds_bytes, stream, P, K); break;
        case MODE_ENV_SOURCES: hipLaunchKernelGGL((world_kernel<AM, LM, MODE_ENV_SOURCES>), grid, block, lds_bytes, stream, P, K); break;
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

int agent_stride(int A, int L) { return agent_stride_of(A, L); }

const char* kernel_variant_name(int variant) {
    static const char* names[4] = {"world_kernel<4,4>", "world_kernel<8,8>", "world_kernel<16,16>", "world_kernel<16,32>"};
    return names[variant & 3];
}

// `pes`: per-environment sources (the second table section in LDS, colour words in the hand-over records)
uint32_t kernel_lds_bytes(const MapHeader& h, uint32_t waves_per_wg, bool pes) {
    const uint32_t scr_stride = (h.L + h.A + 2 + (pes ? (uint32_t)src_stride_of((int)h.L) / 4u : 0u)) | 1u;
    return h.lds_table_bytes + (pes ? h.ext_bytes : 0u) + waves_per_wg * (h.obs_stride + 64 * scr_stride * 4) + 64;
}

// wavefronts per workgroup: four when that fits a CU's LDS twice over (two workgroups per CU), else as many (4, 2, 1)
// as fit the 160 KiB at all
constexpr uint32_t LDS_PER_CU = 160 * 1024;
uint32_t kernel_waves_per_wg(const MapHeader& h, bool pes) {
    for (uint32_t w = 4; w > 1; w >>= 1)
        if (kernel_lds_bytes(h, w, pes) <= LDS_PER_CU) return w;
    return 1;
}

// Store policy of a launch that writes `bytes` of observation rows (WRITE_THROUGH_MAX_BYTES, tables.h).
// LLE_WRITE_THROUGH=0 / 1 forces it (tuning aid).
bool write_through_pays(uint64_t bytes) {
    const char* o = getenv("LLE_WRITE_THROUGH");  // read per launch: the parity tests run both policies in one process
    if (o && (o[0] == '0' || o[0] == '1') && !o[1]) return o[0] == '1';
    return bytes <= WRITE_THROUGH_MAX_BYTES;
}

hipError_t launch_world_kernel(int mode, const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K_in, hipStream_t stream) {
    LaunchArgs K = K_in;
    if (write_through_pays((uint64_t)(K.env_limit - K.env_base) * h.obs_stride)) K.flags |= LAUNCH_WRITE_THROUGH;
    const bool pes = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;
    uint32_t wpw = kernel_waves_per_wg(h, pes);
    if (K.envs_per_map && !K.map_override) {  // a workgroup's environments must belong to one map
        while (K.envs_per_wave > 1 && K.envs_per_map % (int64_t)K.envs_per_wave != 0) K.envs_per_wave >>= 1;
        while (wpw > 1 && K.envs_per_map % (int64_t)(wpw * K.envs_per_wave) != 0) wpw >>= 1;
    }
    const uint32_t epw = K.envs_per_wave;
    const uint32_t n_waves = (uint32_t)((K.env_limit - K.env_base + epw - 1) / epw);
    const uint32_t lds = kernel_lds_bytes(h, wpw, pes);
    switch (kernel_variant((int)h.A, (int)h.L)) {
        case 0: return launch_mode<4, 4>(mode, P, K, h, n_waves, wpw, lds, stream);
        case 1: return launch_mode<8, 8>(mode, P, K, h, n_waves, wpw, lds, stream);
        case 2: return launch_mode<16, 16>(mode, P, K, h, n_waves, wpw, lds, stream);
        default: return launch_mode<16, 32>(mode, P, K, h, n_waves, wpw, lds, stream);
    }
}

// ---- step_kernel<G, LM> dispatch: G = lanes per environment (power of two >= A), LM = beam registers (>= L)
int step_group(int A) { return A <= 1 ? 1 : (A <= 2 ? 2 : (A <= 4 ? 4 : (A <= 8 ? 8 : 16))); }
int step_lm(int L) { return L <= 4 ? 4 : (L <= 8 ? 8 : (L <= 16 ? 16 : 32)); }

template <int G, int LM, int MODE, bool ML1, int LX = -1>
static hipError_t launch_step_glp(const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    dim3 grid((n_waves + wpw - 1) / wpw), block(64 * wpw);
    if (lds > 64 * 1024) {  // gfx950 has 160 KiB of LDS per CU; more than 64 KiB per workgroup is opt-in
        static uint32_t granted = 0;
        if (lds > granted) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel<G, LM, MODE, ML1, LX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            granted = lds;
        }
    }
    hipLaunchKernelGGL((step_kernel<G, LM, MODE, ML1, LX>), grid, block, lds, stream, P, K);
    return hipGetLastError();
}
template <int G, int LM>
static hipError_t launch_step_gl(const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    const bool ml1 = (K.flags & LAUNCH_SINGLE_LAYER) != 0;
    if (K.flags & LAUNCH_GENERAL) {
        const bool pes_mode = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;  // MODE 3: + per-env sources
        if constexpr (LM == 4) {
            if (ml1) {
#define LLE_STEP_LX_GEN(X)                                                                                 \
    case X:                                                                                                \
        return pes_mode ? launch_step_glp<G, 4, 3, true, X>(P, K, n_waves, wpw, lds, stream)               \
                        : launch_step_glp<G, 4, 2, true, X>(P, K, n_waves, wpw, lds, stream);
                switch (K.n_sources) { LLE_STEP_LX_GEN(0) LLE_STEP_LX_GEN(1) LLE_STEP_LX_GEN(2) LLE_STEP_LX_GEN(3) LLE_STEP_LX_GEN(4) default: break; }
#undef LLE_STEP_LX_GEN
            }
        }
        if (pes_mode)
            return ml1 ? launch_step_glp<G, LM, 3, true>(P, K, n_waves, wpw, lds, stream)
                       : launch_step_glp<G, LM, 3, false>(P, K, n_waves, wpw, lds, stream);
        return ml1 ? launch_step_glp<G, LM, 2, true>(P, K, n_waves, wpw, lds, stream)
                   : launch_step_glp<G, LM, 2, false>(P, K, n_waves, wpw, lds, stream);
    }
    const bool roll = (K.flags & LAUNCH_ROLLOUT) != 0;
    if constexpr (LM == 4) {  // maps with at most four sources and no crossing beams: exact source count at compile time
        if (ml1) {
#define LLE_STEP_LX(X)                                                                                    \
    case X:                                                                                               \
        return roll ? launch_step_glp<G, LM, 1, true, X>(P, K, n_waves, wpw, lds, stream)                 \
                    : launch_step_glp<G, LM, 0, true, X>(P, K, n_waves, wpw, lds, stream);
            switch (K.n_sources) {
                LLE_STEP_LX(0) LLE_STEP_LX(1) LLE_STEP_LX(2) LLE_STEP_LX(3) LLE_STEP_LX(4)
                default: break;
            }
#undef LLE_STEP_LX
        }
    }
    if (roll)
        return ml1 ? launch_step_glp<G, LM, 1, true>(P, K, n_waves, wpw, lds, stream) : launch_step_glp<G, LM, 1, false>(P, K, n_waves, wpw, lds, stream);
    return ml1 ? launch_step_glp<G, LM, 0, true>(P, K, n_waves, wpw, lds, stream) : launch_step_glp<G, LM, 0, false>(P, K, n_waves, wpw, lds, stream);
}
template <int G>
static hipError_t launch_step_g(int lm, const BatchPtrs& P, const LaunchArgs& K, uint32_t n_waves, uint32_t wpw, uint32_t lds, hipStream_t stream) {
    switch (lm) {
        case 4: return launch_step_gl<G, 4>(P, K, n_waves, wpw, lds, stream);
        case 8: return launch_step_gl<G, 8>(P, K, n_waves, wpw, lds, stream);
        case 16: return launch_step_gl<G, 16>(P, K, n_waves, wpw, lds, stream);
        default: return launch_step_gl<G, 32>(P, K, n_waves, wpw, lds, stream);
    }
}

// environments per wavefront of the step kernel for a batch of n: as many as fit (64 / G) once that still leaves
// ~4096 wavefronts (16 per CU), fewer (down to 4) for small batches
uint32_t step_envs_per_wave(int64_t n, int A) {
    uint32_t e = 64u / (uint32_t)step_group(A);
    if (const char* o = getenv("LLE_STEP_EPW")) {  // tuning override
        const uint32_t v = (uint32_t)atoi(o);
        if (v >= 1 && v <= e && !(v & (v - 1))) return v;
    }
    while (e > 4 && n / e < 4096) e >>= 1;  // measured on level 1: 4 beats 1-2 even at n = 4096
    return e;
}

hipError_t launch_step_kernel(const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K_in, hipStream_t stream) {
    LaunchArgs K = K_in;
    const int G = step_group((int)h.A), lm = step_lm((int)h.L);
    const bool pes = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;
    K.n_sources = h.L;
    {   // a ring keeps the rows of min(n_steps, ring_slots) steps; without one every step overwrites the same rows
        const uint64_t slots = K.ring_slots ? (K.n_steps < K.ring_slots ? (K.n_steps ? K.n_steps : 1u) : K.ring_slots) : 1u;
        if (write_through_pays((uint64_t)(K.env_limit - K.env_base) * h.obs_stride * slots)) K.flags |= LAUNCH_WRITE_THROUGH;
    }
    if (pes || K.envs_per_map) K.flags |= LAUNCH_GENERAL;
    else if (K.n_steps > 1 || K.ring_slots || K.stamps) K.flags |= LAUNCH_ROLLOUT;
    if (h.max_layers <= 1) K.flags |= LAUNCH_SINGLE_LAYER;  // several maps: `h` carries the maximum over the maps
    uint32_t wpw = kernel_waves_per_wg(h, pes);
    if (K.envs_per_map) {  // a workgroup's environments must belong to one map
        const uint32_t cap = 64u / (uint32_t)G;
        while (K.envs_per_wave > 1 && K.envs_per_map % (int64_t)(K.envs_per_wave < cap ? K.envs_per_wave : cap) != 0) K.envs_per_wave >>= 1;
        const uint32_t e = K.envs_per_wave < cap ? K.envs_per_wave : cap;
        while (wpw > 1 && K.envs_per_map % (int64_t)(wpw * e) != 0) wpw >>= 1;
    }
    const uint32_t epw = K.envs_per_wave;
    const uint32_t n_waves = (uint32_t)((K.env_limit - K.env_base + epw - 1) / epw);
    const uint32_t lds = kernel_lds_bytes(h, wpw, pes);
    switch (G) {
        case 1: return launch_step_g<1>(lm, P, K, n_waves, wpw, lds, stream);
        case 2: return launch_step_g<2>(lm, P, K, n_waves, wpw, lds, stream);
        case 4: return launch_step_g<4>(lm, P, K, n_waves, wpw, lds, stream);
        case 8: return launch_step_g<8>(lm, P, K, n_waves, wpw, lds, stream);
        default: return launch_step_g<16>(lm, P, K, n_waves, wpw, lds, stream);
    }
}

}  // namespace lle
