// kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the batched World.
//
//   step_kernel<G, LM, MODE, ML1, LX>  World.step, the hot path (step_kernel.hpp; one translation unit per MODE,
//                                step_mode0-5.hip): one LANE PER AGENT (G lanes per environment, 64/G environments per
//                                wavefront), state machine on bitmasks with DPP / ds_swizzle group reductions.
//   world_kernel<AM, LM, MODE>   (this file) one lane per environment, the state machine of step_logic.hpp: reset,
//                                set_state, observe, source updates (and step, as a diagnostic).
// This file also holds the host-side launch logic of both.
//
// Both: a workgroup = up to four 64-lane wavefronts sharing ONE copy of the map tables in LDS.  Phase 1: load the
// packed state, run the state machine in registers, store state / events / availability masks.  Phase 2 (wave = one
// environment at a time): the wave owns a private LDS copy of the map's static observation; for each of its
// environments it patches the few dynamic bytes (laser on/off bits, gems, agents) into that copy and streams it to HBM
// with one 16-byte store per lane, i.e. 1 KiB fully coalesced per wave instruction (obs_stream.hpp).
// The observation is >= 95 % of all bytes moved, so phase 2 is what the HBM roofline measures.
//
// No MFMA: there is no contraction anywhere on this path.
#include "kernel_common.hpp"
#include "partial_stream.hpp"

#include <string.h>

#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <vector>

namespace lle {

template <int AM, int LM, int MODE>
__global__ void __launch_bounds__(256) world_kernel(BatchPtrs P, LaunchArgs K) {
    // The uniform map constants are read from the head of the table blob (device memory, scalar loads).  Passing the
    // 400-byte header by value in the kernel-argument segment measured ~2.4 us SLOWER per launch: the kernarg
    // segment is fetched with a much longer latency than device memory.
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t blk = xcd_block_dir(blockIdx.x, gridDim.x, K.flags);  // the block of environments this workgroup serves (obs_stream.hpp)
    const uint32_t map_idx = map_index_of(K, K.env_base + (int64_t)(blk * (blockDim.x >> 6)) * K.envs_per_wave);
    const uint8_t* __restrict__ tables = P.tables + (uint64_t)map_idx * K.table_stride;  // this workgroup's map
    const InitRecord* __restrict__ initp = P.init + map_idx;
    const MapHeader* __restrict__ hdr = reinterpret_cast<const MapHeader*>(tables);
    // A workgroup is 1, 2 or 4 wavefronts that share ONE copy of the cell / dyn tables in LDS (a quarter of the L2
    // traffic and of the copy latency of a per-wave copy); everything else is private to a wavefront.
    const uint32_t lane = threadIdx.x & 63u, wave_in_wg = threadIdx.x >> 6, waves_per_wg = blockDim.x >> 6;
    const uint32_t wave_id = blk * waves_per_wg + wave_in_wg;
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
    // per-environment sources (lle_batch_set_sources): the bare static observation + element list follow in LDS
    const bool pes = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;
    const uint32_t ext_bytes = pes ? hdr->ext_bytes : 0u;
    copy_tables2_to_lds(tables + tab_off, tab_bytes, tables + hdr->off_bare, ext_bytes, lds, lane, wave_in_wg, waves_per_wg);
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
    uint32_t ghost = (uint32_t)(raw_bits >> GHOST_SHIFT);  // dead by set_state without an event (tables.h)

    MapView mv;
    mv.cell_lay = cell_lay; mv.cell_meta = cell_meta; mv.hdr = hdr;
    mv.W = (int)hdr->W; mv.A = A; mv.L = L; mv.G = (int)hdr->G;
    mv.enabled = env_enabled; mv.max_layers = hdr->max_layers;
    mv.per_env = pes;
    mv.chain = hdr->chain_mask;
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
                const bool over = (s.alive | ghost) != amask || s.arrived == amask;
                ghost = over ? 0u : ghost;
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
                    const uint32_t walk = meta_walk(cur.meta[a]) | 16u;
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
                ghost = 0u;
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
            {   // who is dead now without having died in this call (world.rs:571-579: agent.die() for `alive = false` emits nothing)
                uint32_t died = 0;
#pragma unroll
                for (int q = 0; q < 2 * AM; q++) {
                    const uint32_t byte = (uint32_t)(ev.w[q >> 3] >> ((q & 7) * 8)) & 0xFFu;
                    died |= ((uint32_t)q < ev.n && (byte >> 4) == EV_DIED) ? (1u << (byte & 15u)) : 0u;
                }
                if (err == 0) ghost = ~s.alive & ~died & amask;  // (a refused request leaves the bookkeeping as it was)
            }
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
                    ghost = 0u;
                }
                // (the caller speaks of SOURCES -- colours u8 [n][n_sources], bit s of the enabled mask = source s; the kernels of beam
                // WORDS, tables.h: every word of a source takes its colour and its flag.  word_source is the identity, and
                // n_sources == L, unless a beam of the map is longer than 32 cells)
                const int n_src = (int)hdr->n_sources;
                uint32_t new_en = mv.enabled;
                if (fill) new_en = hdr->enabled_mask;
                else if (K.enabled_in) {
                    const uint32_t in = K.enabled_in[env];
                    new_en = 0u;
#pragma unroll
                    for (int b = 0; b < LM; b++)
                        if (b < L) new_en |= ((in >> hdr->word_source[b]) & 1u) << b;
                }
                new_en &= hdr->word_mask;
                uint32_t ncol[LM / 4];
#pragma unroll
                for (int q = 0; q < LM / 4; q++) ncol[q] = colw[q];
                bool bad = false, crosses = false;
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    if (b < L && ((hdr->word_mask >> b) & 1u) && (fill || K.colours_in)) {  // (not the padding words of a chained map)
                        const uint32_t c = fill ? (uint32_t)hdr->beam_colour[b] : (uint32_t)K.colours_in[env * n_src + hdr->word_source[b]];
                        bad |= !fill && c >= (uint32_t)A;  // "Agent ID is greater than the number of agents"
                        // "... would cross the start position of agent ..." (pylaser_source.rs:121-139; MapHeader.colour_ok)
                        crosses |= !fill && c < (uint32_t)A && !(((uint32_t)hdr->colour_ok[b] >> c) & 1u);
                        ncol[b >> 2] = (ncol[b >> 2] & ~(0xFFu << ((b & 3) * 8))) | ((c & 0xFFu) << ((b & 3) * 8));
                    }
                }
                err = bad ? ENV_INVALID_COLOUR : (crosses ? ENV_COLOUR_CROSSES_START : 0u);
                bad |= crosses;
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
                P.done[env] = ((s.alive | ghost) != amask || s.arrived == amask) ? 1 : 0;  // (a function of the state: lle_batch_restore relies on it)
            } else {
                store_state = false;
            }
        } else {
            store_state = false;
            // (MODE_OBSERVE: `done` is a function of the state; lle_batch_restore rebuilds it here together with the observation)
            if (MODE == MODE_OBSERVE) P.done[env] = ((s.alive | ghost) != amask || s.arrived == amask) ? 1 : 0;
        }

        LLE_STAMP(3);
        if (store_state) {
            store_u16_record<AM>(P.pos, env, s.pos);
            P.bits[env] = (uint64_t)s.alive | ((uint64_t)s.arrived << 16) | ((uint64_t)s.occ << 32) | ((uint64_t)ghost << GHOST_SHIFT);
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
            P.done[env] = ((s.alive | ghost) != amask || s.arrived == amask) ? 1 : 0;
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
        const uint32_t et = (K.flags & LAUNCH_OBS_ELEM_MASK) >> LAUNCH_OBS_ELEM_SHIFT;  // (tables.h ObsElem: the rows' element type)
        const uint64_t row_pitch = (uint64_t)hdr->obs_stride << obs_elem_shift(et);
        dispatch_stream<true>(K.flags, [&](auto wt_, auto wide_) {
            constexpr bool WT = decltype(wt_)::value, WIDE = decltype(wide_)::value;
            if (pes) write_observations_env<WT, false, false, WIDE>(A, L, hdr->HW, hdr->n_elems, hdr->n_chunks, row_pitch, elems, bare, tmpl, scratch, scr_stride, P.obs, env0,
                                                                    n_here, lane, nullptr, 0xFFFFFFFFu, 0u, 0u, 0u, nullptr, 0u, 0u, 0u, et);
            else write_observations<WT, false, false, WIDE>(A, L, hdr->D, hdr->n_chunks, row_pitch, dyn, tmpl, scratch, scr_stride, P.obs, env0, n_here, lane, 0u, 0u, 0u,
                                                            nullptr, 0u, et);
        });
    }
    LLE_STAMP(5);
    if (MODE == MODE_STEP) flush_stats(P.stats, wave_id, cnt, A, lane);

    if (K.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LLE_STAMP(6);
    }
}

// ------------------------------------------------------------------------------------------------ launchers
template <int AM, int LM>
static hipError_t launch_mode(int mode, const BatchPtrs& P, const LaunchArgs& K, const MapHeader& H, uint32_t n_waves,
                              uint32_t waves_per_wg, uint32_t lds_bytes, hipStream_t stream) {
    if (mode >= 0 && mode < 6) {  // (kernels.h debug registry)
        static std::atomic<uint32_t> noted{0};
        const uint32_t bit = 1u << (mode + ((K.flags & LAUNCH_DRY_RUN) ? 8 : 0));
        if (!(noted.load(std::memory_order_relaxed) & bit)) {
            noted.fetch_or(bit, std::memory_order_relaxed);
            debug_note(debug_key(DBG_WORLD, AM, LM, mode, false, -1), (K.flags & LAUNCH_DRY_RUN) != 0);
        }
    }
    if (K.flags & LAUNCH_DRY_RUN) return hipSuccess;
    dim3 grid((n_waves + waves_per_wg - 1) / waves_per_wg), block(64 * waves_per_wg);
    if (lds_bytes > 64 * 1024) {
        static LdsGrant granted[6];  // per mode, per device (kernels.h)
        if (mode >= 0 && mode < 6) {
            const void* fn = mode == MODE_STEP ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_STEP>)
                           : mode == MODE_RESET ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_RESET>)
                           : mode == MODE_SET_STATE ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_SET_STATE>)
                           : mode == MODE_OBSERVE ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_OBSERVE>)
                           : mode == MODE_SOURCES ? reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_SOURCES>)
                                                  : reinterpret_cast<const void*>(&world_kernel<AM, LM, MODE_ENV_SOURCES>);
            hipError_t e = granted[mode].ensure(fn, lds_bytes);
            if (e != hipSuccess) return e;
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

// ---- debug registry (kernels.h): the instantiations launched by this process / reachable through the dispatch
static std::mutex g_dbg_mutex;
static std::set<uint32_t> g_dbg_launched, g_dbg_reachable;
void debug_note(uint32_t key, bool reachable_walk) {
    std::lock_guard<std::mutex> lock(g_dbg_mutex);
    (reachable_walk ? g_dbg_reachable : g_dbg_launched).insert(key);
}
void debug_reset_launched() {
    // (the per-instantiation `noted` flags stay set: a reset registry only sees instantiations launched for the FIRST time afterwards --
    // the coverage test runs in a process of its own and never resets; the call exists for symmetry in long-lived tools)
    std::lock_guard<std::mutex> lock(g_dbg_mutex);
    g_dbg_launched.clear();
}
static const char* const OBS_KERNEL_NAMES[OBSK_COUNT] = {
    "view_observe_kernel", "partial_observe_kernel", "partial_project_kernel", "partial_lanes_kernel", "state_observe_kernel", "avail_kernel",
    "env_outputs_kernel", "row_fill_probe_kernel<true>", "row_fill_probe_kernel<false>", "cast_rows_kernel", "stats_sum_kernel"};
static std::string debug_name(uint32_t key) {
    const int kind = (int)(key >> 24), g = (int)(key >> 16) & 0xFF, lm = (int)(key >> 8) & 0xFF, mode = (int)(key >> 4) & 0xF, ml1 = (int)(key >> 3) & 1,
              lx = (int)(key & 7u) - 1;
    char buf[96];
    if (kind == DBG_STEP) snprintf(buf, sizeof buf, "step_kernel<%d,%d,%d,%s,%d>", g, lm, mode, ml1 ? "true" : "false", lx);
    else if (kind == DBG_WORLD) snprintf(buf, sizeof buf, "world_kernel<%d,%d,%d>", g, lm, mode);
    else snprintf(buf, sizeof buf, "%s", mode * 16 + g < OBSK_COUNT ? OBS_KERNEL_NAMES[mode * 16 + g] : "?");
    return buf;
}
// Every (agents, beam words, crossing or not, MODE) the launcher can be asked for, walked through the SAME dispatch code a launch takes
// (LAUNCH_DRY_RUN: launch_step_glp / launch_mode note the instantiation and return): the list cannot drift from the dispatch.
static void debug_walk_reachable() {
    BatchPtrs P{};
    for (int A = 1; A <= MAX_AGENTS; A++) {
        for (int L = 0; L <= MAX_SOURCES; L++) {
            const int G = step_group(A), lm = step_lm(L);
            for (int ml1 = 0; ml1 < 2; ml1++) {
                if (!ml1 && L < 2) continue;  // (a cell with two laser layers needs two sources)
                LaunchArgs K{};
                K.flags = LAUNCH_DRY_RUN | (ml1 ? LAUNCH_SINGLE_LAYER : 0u);
                K.n_sources = (uint32_t)L;
                (void)launch_step_mode0(G, lm, P, K, 1, 1, 0, nullptr);
                (void)launch_step_mode1(G, lm, P, K, 1, 1, 0, nullptr);
                (void)launch_step_mode2(G, lm, P, K, 1, 1, 0, nullptr);
                (void)launch_step_mode3(G, lm, P, K, 1, 1, 0, nullptr);
                (void)launch_step_mode4(G, lm, P, K, 1, 1, 0, nullptr);
                (void)launch_step_mode5(G, lm, P, K, 1, 1, 0, nullptr);
                if (lm <= 8) {  // (launch_step_kernel: row heads and the partial writer serve maps with at most 8 beam words)
                    (void)launch_step_mode6(G, lm, P, K, 1, 1, 0, nullptr);
                    (void)launch_step_mode7(G, lm, P, K, 1, 1, 0, nullptr);
                    (void)launch_step_mode8(G, lm, P, K, 1, 1, 0, nullptr);
                    (void)launch_step_mode9(G, lm, P, K, 1, 1, 0, nullptr);
                }
            }
            MapHeader h{};
            h.A = (uint32_t)A; h.L = (uint32_t)L; h.obs_stride = 16; h.lds_table_bytes = 1024;
            for (int mode = 0; mode < 6; mode++) {
                LaunchArgs K{};
                K.flags = LAUNCH_DRY_RUN;
                K.envs_per_wave = 4; K.env_limit = 4;
                (void)launch_world_kernel(mode, h, P, K, nullptr);
            }
        }
    }
    for (int k = 0; k < OBSK_COUNT; k++) debug_note(debug_key(DBG_OBSERVER, k & 15, 0, k >> 4, false, -1), true);
}
size_t debug_list(bool reachable, char* buf, size_t cap) {
    if (reachable) {
        bool empty;
        { std::lock_guard<std::mutex> lock(g_dbg_mutex); empty = g_dbg_reachable.empty(); }
        if (empty) debug_walk_reachable();
    }
    std::vector<std::string> names;
    {
        std::lock_guard<std::mutex> lock(g_dbg_mutex);
        for (uint32_t key : (reachable ? g_dbg_reachable : g_dbg_launched)) names.push_back(debug_name(key));
    }
    std::sort(names.begin(), names.end());
    std::string all;
    for (const std::string& n : names) { all += n; all += '\n'; }
    if (buf && cap) {
        const size_t n = std::min(cap - 1, all.size());
        memcpy(buf, all.data(), n);
        buf[n] = 0;
    }
    return all.size() + 1;
}

// ---- the environment overrides, read once (kernels.h Tuning)
// The snapshot the launch path reads is published through an atomic pointer: lle_tuning_refresh() may run while other threads launch
// (distinct handles are independent, lle_hip.h).  A refresh builds a NEW snapshot and swaps the pointer; the old one is never freed (a
// launch may still be reading it, and refreshes are a handful per process: tests and tuning tools).
static std::atomic<const Tuning*> g_tuning{nullptr};
static int env_bool(const char* name) {
    const char* o = getenv(name);
    return (o && (o[0] == '0' || o[0] == '1') && !o[1]) ? o[0] - '0' : -1;
}
static int env_uint(const char* name) {
    const char* o = getenv(name);
    const int v = o ? atoi(o) : 0;
    return v > 0 ? v : 0;
}
void tuning_refresh() {
    Tuning* fresh = new Tuning();
    Tuning& t = *fresh;
    t.step_wpw = env_uint("LLE_STEP_WPW");
    t.step_split = env_bool("LLE_STEP_SPLIT");
    t.write_through = env_bool("LLE_WRITE_THROUGH");
    t.row_heads = env_bool("LLE_ROW_HEADS");
    t.step_epw = env_uint("LLE_STEP_EPW");
    t.pingpong = env_bool("LLE_PINGPONG");
    t.partial_project = env_bool("LLE_PARTIAL_PROJECT");
    if (const char* w = getenv("LLE_PARTIAL_KERNEL")) t.partial_kernel = !strcmp(w, "lanes") ? 1 : (!strcmp(w, "window") ? 2 : (!strcmp(w, "project") ? 3 : 4));
    t.partial_e = env_uint("LLE_PARTIAL_E");
    t.partial_batches = env_uint("LLE_PARTIAL_BATCHES");
    if (const char* o = getenv("LLE_PARTIAL_WT")) t.partial_wt = o[0] == '1' ? 1 : 0;
    t.partial_epw = env_uint("LLE_PARTIAL_EPW");
    t.row_rotate = env_bool("LLE_ROW_ROTATE");
    t.head_group = env_uint("LLE_HEAD_GROUP");
    t.post_first = env_bool("LLE_POST_FIRST");
    t.packed_tables = env_bool("LLE_PACKED_TABLES");
    g_tuning.store(fresh, std::memory_order_release);
}
const Tuning& tuning() {
    const Tuning* t = g_tuning.load(std::memory_order_acquire);
    if (!t) {
        tuning_refresh();
        t = g_tuning.load(std::memory_order_acquire);
    }
    return *t;
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

// step_kernel keeps the beam masks of these maps in the env's LDS record (step_kernel.hpp BM): more than 4 sources
static bool beams_in_lds(const MapHeader& h) { return h.L > 4; }

// `pes`: per-environment sources (the second table section in LDS, colour words in the hand-over records)
uint32_t kernel_lds_bytes(const MapHeader& h, uint32_t waves_per_wg, bool pes) {
    const uint32_t scr_stride = (h.L + h.A + 2 + (pes ? (uint32_t)src_stride_of((int)h.L) / 4u : 0u)) | 1u;
    // (+ 256 B: step_kernel's beam tables of maps with more than 8 sources, step_kernel.hpp BM)
    return h.lds_table_bytes + (pes ? h.ext_bytes : 0u) + (beams_in_lds(h) ? 256u : 0u) + waves_per_wg * (h.obs_stride + 64 * scr_stride * 4 + (pes ? PES_WAVE_EXTRA_BYTES : 0u)) + 64;
}

// wavefronts per workgroup: as many (4, 2, 1) as fit the 160 KiB of a CU.  More wavefronts per table copy beat more
// workgroups per CU: config 5 (33 KB of tables, 20 KB rows) 4 -> 274 us per step, 2 -> 350, 1 -> 377 (LLE_STEP_WPW)
constexpr uint32_t LDS_PER_CU = 160 * 1024;
uint32_t kernel_waves_per_wg(const MapHeader& h, bool pes) {
    if (const uint32_t v = (uint32_t)tuning().step_wpw) {  // LLE_STEP_WPW: tuning override
        if ((v == 1 || v == 2 || v == 4) && kernel_lds_bytes(h, v, pes) <= LDS_PER_CU) return v;
    }
    for (uint32_t w = 4; w > 1; w >>= 1)
        if (kernel_lds_bytes(h, w, pes) <= LDS_PER_CU) return w;
    return 1;
}

// ---- split rows (step_kernel, LAUNCH_SPLIT_ROWS): when whole-row private copies would leave ONE workgroup per CU (config
// 5: 33 KB of tables + 4 x 20.5 KB = 135 KB), every wavefront keeps one slice of the row instead and streams that slice
// of every environment of the workgroup: tables without the pristine row (13 KB) + the row once (20 KB) + records
// = 36 KB, four workgroups per CU.  LLE_STEP_SPLIT=0 / 1 forces it (tuning aid; 1 only where the kernels carry it).
bool step_can_split_rows(const MapHeader& h, bool pes) { return !pes && step_group((int)h.A) >= 8; }
bool step_splits_rows(const MapHeader& h, bool pes, const StepTune& tune) {
    if (!step_can_split_rows(h, pes)) return false;
    if (tuning().step_split >= 0) return tuning().step_split == 1;
    if (tune.split >= 0) return tune.split == 1;
    return kernel_lds_bytes(h, 4, false) > LDS_PER_CU / 2;
}
uint32_t split_lds_bytes(const MapHeader& h, uint32_t wpw, uint32_t epw) {
    const uint32_t scr_stride = (h.L + h.A + 2) | 1u, cpw = (h.n_chunks + wpw - 1) / wpw;
    return h.lds_split_table_bytes + (beams_in_lds(h) ? 256u : 0u) + wpw * cpw * 16u + wpw * epw * scr_stride * 4u + 64u;
}

// Store policy of a launch that writes `bytes` of observation rows (WRITE_THROUGH_MAX_BYTES, tables.h).
// LLE_WRITE_THROUGH=0 / 1 forces it (tuning aid).
// Rows whose pitch is a whole number of 128-byte lines never share a line, so written through each line leaves L2 once
// and whole: `sc1` then wins at every size (level 6 at 1 920 B: 262 144 envs 99.7 us vs 107.7 plain; 65 536 envs 21.2 vs
// 23.3).  Packed rows (1 872 B) share lines between neighbours; past the Infinity Cache a shared line written through
// goes to HBM twice as partial writes (262 144 envs: 163 us vs 98-105 plain), so there the policy follows the size.
bool write_through_pays(uint64_t bytes, uint32_t row_pitch, int chosen) {
    if (tuning().write_through >= 0) return tuning().write_through == 1;
    if (chosen >= 0) return chosen == 1;  // (a batch's own choice: lle_batch_autotune)
    return row_pitch % 128u == 0 || bytes <= WRITE_THROUGH_MAX_BYTES;
}

// Row heads ahead of the state machine (step_kernel.hpp HEAD; tables.h head_lo / head_n): pays when the launch is about
// one or two rounds of workgroups -- every wavefront then runs its state machine at the same time with the memory system
// idle.  Measured on level 6 with a 3-line head: 32 768 envs (2 048 wavefronts) 13.3 -> 12.4 us, 65 536 21.4 -> 19.8,
// 131 072 36.7 -> 35.1; a small launch (16 384 envs: 9.4 -> 9.7) or many rounds (262 144: 98.9 -> 100.1) lose a little.
// LLE_ROW_HEADS=0 / 1 forces it; a batch's own measurement (lle_batch_autotune: StepTune.heads) comes next; these are the defaults.
static bool row_heads_pay(uint32_t n_waves, bool general, int chosen) {
    if (tuning().row_heads >= 0) return tuning().row_heads == 1;
    if (chosen >= 0) return chosen == 1;
    // The general instantiations (several maps / fused LLE.step outputs: MODE 7 against 4; per-env sources: MODE 8 against 5) win
    // with the heads at every size from 2 048 wavefronts up -- they are also the ones with every load up front and, for MODE 8,
    // without the spills of MODE 5.  Level 6, us per step without / with (round 3, one box): per-env sources 8 192 envs 9.7 / 8.8,
    // 65 536 24.3 / 21.8, 262 144 102.2 / 87.5, 524 288 202.5 / 190.4; fused outputs 8 192 9.2 / 8.9, 65 536 23.2 / 21.1,
    // 262 144 87.5 / 80.6; five other maps: profiles/r03_heads_rule.md (below 2 048 wavefronts it is a wash: -0.5 ... +0.6 us).
    if (general) return n_waves >= 2048u;
    // The default instantiation (MODE 6 against 0): 16 384 envs 9.1 / 9.5, 32 768 13.0 / 12.3, 65 536 21.2 / 19.5, 131 072 37.3 / 35.2,
    // 196 608 59.9 / 58.8, 262 144 74.3 / 74.4, 524 288 170.0 / 173.4.
    return n_waves >= 2048u && n_waves <= 12288u;
}

hipError_t launch_world_kernel(int mode, const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K_in, hipStream_t stream) {
    LaunchArgs K = K_in;
    const uint32_t pitch = h.obs_stride << obs_elem_shift((K.flags & LAUNCH_OBS_ELEM_MASK) >> LAUNCH_OBS_ELEM_SHIFT);  // row pitch in bytes
    if (write_through_pays((uint64_t)(K.env_limit - K.env_base) * pitch, pitch)) K.flags |= LAUNCH_WRITE_THROUGH;
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

// environments per wavefront of the step kernel for a batch of n: as many as fit (64 / G) once that still leaves
// ~4096 wavefronts (16 per CU), fewer (down to 4) for small batches
uint32_t step_envs_per_wave(int64_t n, int A, const StepTune& tune, int64_t split_block) {
    uint32_t e = 64u / (uint32_t)step_group(A);
    // LLE_STEP_EPW, then the batch's own choice.  LLE_BUF_STATS holds max(MIN_STAT_SLOTS, n / MIN_ENVS_PER_WAVE) slots, one per wavefront
    // (capi.cpp make_layout), and a wavefront indexes it by its id: fewer than MIN_ENVS_PER_WAVE environments per wavefront only while
    // the batch is small enough for its wavefronts to fit the slots that always exist.
    for (const uint32_t v : {(uint32_t)tuning().step_epw, (uint32_t)tune.epw})
        if (v >= 1 && v <= e && !(v & (v - 1)) && (v >= MIN_ENVS_PER_WAVE || (n + v - 1) / v <= (int64_t)MIN_STAT_SLOTS)) return v;
    while (e > 4 && n / e < 4096) e >>= 1;
    // ... and below four for batches that would otherwise leave most of the chip without a wavefront: a wavefront streams its rows one
    // after the other (0.45 us each on level 1), so at 4 096 envs two per wavefront beat four (5.67 -> 5.41 us), at 1 024 one beats four
    // (5.72 -> 4.73 us); one per wavefront at 4 096 loses again (6.27: four times the table copies).  tools/small_batch_epw.py, round 5.
    while (e > 1 && n / e < 2048) e >>= 1;
    // Small blocks of a multi-map batch whose rows are split (`split_block` = environments per map): a workgroup serves one map, and what it
    // pays per map -- the table copy, the one shared row in LDS -- buys the stores of as many wavefronts as the block fills.  Fewer environments
    // per wavefront until the block fills four: config 5's shape, 4 096 maps x 16: 8 per wavefront (two wavefronts per workgroup) 246 us, 4 per
    // wavefront (four) 238; 8 192 maps x 8: 314 -> 282 (one map: 211; tools/multimap_epw.py, round 5).  Whole-row launches (small maps) lose
    // with it: their wavefronts each keep a row of their own, and fewer environments per wavefront only idle the state machine's lanes.
    if (split_block > 0)
        while (e > MIN_ENVS_PER_WAVE && split_block % (int64_t)(4u * e) != 0) e >>= 1;
    return e;
}

// Row rotation (obs_stream.hpp row_rotation).  Measured on a box whose buffers are all of the slow kind (tools/rotate_ab.py, level 6,
// us per step without / with; `fill` = the row-fill probe): every row to DRAM (one-directional walk) 262 144 envs 95.9 / 90.9 (fill
// 87.4 / 83.9), 524 288 envs 184.2 / 179.5; with the alternating walk 262 144 envs 73.9 / 74.4, 524 288 170.7 / 167.1; inside the
// Infinity Cache it costs 0.1-0.2 us (65 536 envs 19.8 / 19.9, fill 15.5 / 15.6), and packed rows (level 1: 944-byte pitch) lose with
// it (108.8 / 113.2).  Hence: rows of whole 128-byte lines, launches that write more than the cache holds.  LLE_ROW_ROTATE=0 / 1
// forces it; lle_batch_autotune measures it on the batch's own arena.
bool rotate_rows_pays(const StepTune& tune, uint64_t row_bytes_per_launch, uint32_t row_pitch) {
    if (tuning().row_rotate >= 0) return tuning().row_rotate == 1;
    if (tune.rotate >= 0) return tune.rotate == 1;
    return row_pitch % 128u == 0 && row_bytes_per_launch > WRITE_THROUGH_MAX_BYTES;
}

// ---- step_kernel MODE 9: LDS of a workgroup and the writer's batch size (partial_stream.hpp; the rule of observers.hip's lane kernel:
// the largest E whose block of rows stays within 16 KiB per wavefront, halved above 9 KiB while the observer's lanes stay busy)
// (`sets`: the launch carries the window sets of k -- tables.h -- instead of building the map's non-empty bitmap)
static uint32_t partial_step_lds(const MapHeader& h, uint32_t wpw, uint32_t E, int k, uint32_t epw = 64u, bool sets = false) {
    const uint32_t scr_stride = (h.L + h.A + 2) | 1u, pitch = ((h.A * (2 * h.A + 3)) * (uint32_t)(k * k) + 15u) & ~15u;
    uint32_t tab = ((h.off_dyn - h.off_cell_lay) + 1023u) & ~1023u;
    if (tab > h.lds_table_bytes) tab = h.lds_table_bytes;
    const uint32_t window = sets ? win_sets_bytes(h.HW) : partial_bitmap_bytes(h.H, h.W);
    return tab + (h.L > 4 ? 256u : 0u) + window + 32u + wpw * (E * pitch + 16u + ((epw * scr_stride * 4u + 15u) & ~15u)) + 64u;
}
uint32_t step_partial_batch(const MapHeader& h, int k, bool pes) {
    if (pes || step_lm((int)h.L) > 8 || k < 1 || k > 15 || !(k & 1)) return 0u;
    const uint32_t a_pad = (uint32_t)step_group((int)h.A), pitch = ((h.A * (2 * h.A + 3)) * (uint32_t)(k * k) + 15u) & ~15u;
    const uint32_t s_min = k <= 8 ? 1u : ((uint32_t)k + 3u) / 4u;
    uint32_t E = 64u / a_pad;
    while (E > 1 && 64u / (E * a_pad) < s_min) E >>= 1;
    if (64u / (E * a_pad) < s_min) return 0u;  // (16 agents and a window above 8: an observer's rows do not fit its lanes)
    while (E > 1 && E * pitch > 16384u) E >>= 1;
    if (E > 1 && E * pitch > 9216u) {
        const uint32_t S2 = 64u / ((E / 2u) * a_pad), rl = ((uint32_t)k + S2 - 1u) / S2;
        if (S2 <= (uint32_t)k && 4u * (uint32_t)k >= 3u * S2 * rl) E >>= 1;
    }
    if (const uint32_t v = (uint32_t)tuning().partial_e)  // LLE_PARTIAL_E: tuning override (a power of two whose lanes still cover a window's rows)
        if (!(v & (v - 1)) && v * a_pad <= 64u && 64u / (v * a_pad) >= s_min) E = v;
    return partial_step_lds(h, 1, E, k, 64u, win_sets_serve(k)) <= LDS_PER_CU ? E : 0u;
}

bool step_has_row_heads(const MapHeader& h, bool pes) {
    return step_lm((int)h.L) <= 8 && (pes ? h.pes_head_n : h.head_n) != 0;
}

#ifdef LLE_STAMP_SINGLE
static bool stamps_roll(const LaunchArgs&) { return false; }  // (diagnostic build: a stamped single step stays a single step)
#else
static bool stamps_roll(const LaunchArgs& K) { return K.stamps != nullptr; }
#endif
static bool roll_requested(const LaunchArgs& K) { return K.n_steps > 1 || K.ring_slots || stamps_roll(K); }

hipError_t launch_step_kernel(const MapHeader& h, const BatchPtrs& P, const LaunchArgs& K_in, hipStream_t stream, const StepTune& tune) {
    LaunchArgs K = K_in;
    const int G = step_group((int)h.A), lm = step_lm((int)h.L);
    const bool pes = (K.flags & LAUNCH_PER_ENV_SOURCES) != 0;
    K.n_sources = h.L;
    const uint32_t et = (K.flags & LAUNCH_OBS_ELEM_MASK) >> LAUNCH_OBS_ELEM_SHIFT;  // (tables.h ObsElem)
    const uint32_t pitch = h.obs_stride << obs_elem_shift(et);                      // row pitch in bytes
    {   // a ring keeps the rows of min(n_steps, ring_slots) steps; without one every step overwrites the same rows
        const uint64_t slots = K.ring_slots ? (K.n_steps < K.ring_slots ? (K.n_steps ? K.n_steps : 1u) : K.ring_slots) : 1u;
        if (write_through_pays((uint64_t)(K.env_limit - K.env_base) * pitch * slots, pitch, tune.write_through)) K.flags |= LAUNCH_WRITE_THROUGH;
        if (rotate_rows_pays(tune, (uint64_t)(K.env_limit - K.env_base) * pitch * slots, pitch)) K.flags |= LAUNCH_ROTATE_ROWS;
    }
    if (pes || K.envs_per_map || K.env_out) K.flags |= LAUNCH_GENERAL;  // (fused LLE.step outputs: MODE 4 / 5 carry the epilogue)
    if (K.env_out && roll_requested(K)) return hipErrorInvalidValue;  // single steps only
    if (tuning().post_first >= 0) K.flags |= tuning().post_first ? LAUNCH_POST_FIRST : LAUNCH_POST_LAST;
    if (roll_requested(K)) K.flags |= LAUNCH_ROLLOUT;
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
    uint32_t lds = kernel_lds_bytes(h, wpw, pes);
    if (step_splits_rows(h, pes, tune)) {
        const uint32_t cap = 64u / (uint32_t)G, e = epw < cap ? epw : cap;
        wpw = 4;
        if (K.envs_per_map)
            while (wpw > 1 && K.envs_per_map % (int64_t)(wpw * e) != 0) wpw >>= 1;
        lds = split_lds_bytes(h, wpw, e);
        K.flags |= LAUNCH_SPLIT_ROWS;
        // A map that serves at most four workgroups is read from memory by each of them, cold, ahead of its first store: those launches take the packed
        // image of the table section (tables.h off_packed; the common header carries the offset only when every map has one).
        // LLE_PACKED_TABLES=0 / 1 forces it.
        // Four-wavefront workgroups only: with two (8 environments per map) the expansion's share per thread doubles and eats the gain
        // (config 5's shape, us per step verbatim / packed on one arena: 1 024 x 64 250.9 / 245.9, 4 096 x 16 259.1 / 251.2, 8 192 x 8 271.6 / 275.5).
        if (K.envs_per_map && h.off_packed && (tuning().packed_tables >= 0 ? tuning().packed_tables == 1 : (wpw == 4 && (uint64_t)K.envs_per_map <= 4ull * wpw * e)))
            K.flags |= LAUNCH_PACKED_TABLES;
    }
    if (K.partial_k) {  // the partial observation written by this launch (MODE 9): single steps with the fused outputs, the map's sources
        if (pes || roll_requested(K) || !K.env_out || !K.partial_E || lm > 8) return hipErrorInvalidValue;
        const uint32_t cap = 64u / (uint32_t)G, e = epw < cap ? epw : cap;  // environments (= records) per wavefront
        // the window sets (tables.h; a third fewer vector instructions in the writer) where they do not cost the launch a workgroup per CU:
        // a launch of this kernel is ONE round of workgroups, and 24 B per cell of sets against a bitmap of a few hundred bytes can be the
        // difference between four per CU and three (level 6, 7 x 7: 39.2 -> 42.1 KB; profiles/r05_partial.md).  LLE_PARTIAL_SETS=1 forces them.
        bool sets = K.win_sets != nullptr && win_sets_serve((int)K.partial_k);
        if (sets && !getenv("LLE_PARTIAL_SETS") &&
            LDS_PER_CU / partial_step_lds(h, 4, K.partial_E, (int)K.partial_k, e, true) < LDS_PER_CU / partial_step_lds(h, 4, K.partial_E, (int)K.partial_k, e, false))
            sets = false;
        if (!sets) K.win_sets = nullptr;
        wpw = 4;
        while (wpw > 1 && partial_step_lds(h, wpw, K.partial_E, (int)K.partial_k, e, sets) > LDS_PER_CU / 2) wpw >>= 1;  // (two workgroups per CU at least)
        if (K.envs_per_map)
            while (wpw > 1 && K.envs_per_map % (int64_t)(wpw * e) != 0) wpw >>= 1;
        K.flags &= ~LAUNCH_SPLIT_ROWS;
        return launch_step_mode9(G, lm, P, K, n_waves, wpw, partial_step_lds(h, wpw, K.partial_E, (int)K.partial_k, e, sets), stream);
    }
    // MODE of the instantiation (step_kernel.hpp): per-env sources 3 / 5, several maps 2 / 4, one map 1 / 0 -- the
    // first of each pair with the rollout loop, rings and stamps, the second for single-step launches
    const bool roll = (K.flags & LAUNCH_ROLLOUT) != 0;
    if (pes) {
        if (roll) return launch_step_mode3(G, lm, P, K, n_waves, wpw, lds, stream);
        // single steps: the colour-independent head lines ahead of the state machine (MODE 8) under the same conditions as below
        const bool incr_pes = (K.flags & STEP_INCREMENTAL_OBS) != 0 && h.n_pes_dyn_chunks < h.n_chunks;  // (static lines not written: no head to send ahead)
        // (widened rows -- et != int8 -- go out whole behind the state machine: a head in front of it is a third to a ninth of its int8 size in time)
        const bool heads_pes = !incr_pes && et == OBS_I8 && lm <= 8 && h.pes_head_n != 0 && !(K.flags & (LAUNCH_SPLIT_ROWS | STEP_NO_OBS)) && row_heads_pay(n_waves, true, tune.heads);
        return heads_pes ? launch_step_mode8(G, lm, P, K, n_waves, wpw, lds, stream) : launch_step_mode5(G, lm, P, K, n_waves, wpw, lds, stream);
    }
    // single-step launches with the map's sources: the rows' head lines go out ahead of the state machine (MODE 6 / 7) when
    // the map has a head, the rows are not split and the launch is of the size where it pays
    // (STEP_INCREMENTAL_OBS: the static lines are not written at all, so there is no head to send ahead)
    const bool incr = (K.flags & STEP_INCREMENTAL_OBS) != 0 && !roll && h.n_dyn_chunks < h.n_chunks;
    const bool heads = !incr && !roll && et == OBS_I8 && lm <= 8 && h.head_n != 0 && !(K.flags & (LAUNCH_SPLIT_ROWS | STEP_NO_OBS)) &&
                       row_heads_pay(n_waves, (K.flags & LAUNCH_GENERAL) != 0, tune.heads);
    if (K.flags & LAUNCH_GENERAL) {
        if (roll) return launch_step_mode2(G, lm, P, K, n_waves, wpw, lds, stream);
        return heads ? launch_step_mode7(G, lm, P, K, n_waves, wpw, lds, stream) : launch_step_mode4(G, lm, P, K, n_waves, wpw, lds, stream);
    }
    if (roll) return launch_step_mode1(G, lm, P, K, n_waves, wpw, lds, stream);
    if (heads) {
        const int hg = tuning().head_group ? tuning().head_group : (tune.head_group ? (int)tune.head_group : 1);
        if (hg == 2) K.flags |= LAUNCH_HEAD_GROUP2;
        if (hg == 4) K.flags |= LAUNCH_HEAD_GROUP4;
        return launch_step_mode6(G, lm, P, K, n_waves, wpw, lds, stream);
    }
    return launch_step_mode0(G, lm, P, K, n_waves, wpw, lds, stream);
}

}  // namespace lle
