// hostsim.cpp -- TEST-ONLY host build of the device state machine (lle_amd/csrc/step_logic.hpp) and of the
// table-driven observation patching, so that the logic and the map tables can be diffed against the oracle on a
// machine without a GPU.  It is compiled by tests/hostsim/__init__.py with g++, lives under tests/, and is never
// loaded by the lle_amd package: the product has no CPU execution path.
//
// It mirrors phase 1 / phase 2 of world_kernel (kernels.hip) one environment at a time, on buffers with exactly the
// device layout (include/lle_hip.h LLE_BUF_*).
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/lle_hip.h"
#include "../../lle_amd/csrc/map_compile.hpp"
#include "../../lle_amd/csrc/observers_logic.hpp"
#include "../../lle_amd/csrc/step_lanes.hpp"
#include "../../lle_amd/csrc/step_logic.hpp"

using namespace lle;

struct hs_batch {
    Map map;
    int64_t n;
    std::vector<uint16_t> pos, req_pos, req_alive;
    std::vector<uint64_t> bits;
    std::vector<uint32_t> gems, beams, req_gems;
    std::vector<uint8_t> avail, actions, err, evcount, events, done;
    std::vector<int8_t> obs;
    int64_t stats[8];
    // per-environment sources (hs_set_sources): colour bytes [n][32], enabled masks [n]
    bool per_env = false;
    std::vector<uint8_t> src_colour;
    std::vector<uint32_t> src_enabled;
    // which restatement steps: 0 = one lane per ENV (step_logic.hpp step_env, the world_kernel path), 1 = one lane per
    // AGENT (step_lanes.hpp, the step_kernel path: the product's hot path), 2 = the same without the no-op-pass shortcut
    int engine = 0;
    int64_t lane_passes = 0;  // (diagnostic of the lane engines: move_agents passes executed / skipped by the shortcut)
};

// ---- World.step through step_lanes.hpp: the G lanes of ONE environment, exactly what a group of lanes of step_kernel runs.
// Replicated words must come back identical in every lane (uniform<> throws LaneDivergence otherwise).
template <int AM, int LM, int G, bool ML1, bool PES, bool SHORTCUT>
static uint32_t lanes_step_g(Env<AM, LM>& s, const uint32_t (&act)[AM], uint32_t (&avail)[AM], const MapView& mv, Events<AM>& ev, int64_t* passes) {
    constexpr int NWG = (2 * G + 7) / 8, CWM = LM / 4;
    // like step_kernel: more than 4 sources -> the beam masks live in the env's record
    // (here: `bm`), not in lane registers
    constexpr bool BM = LM >= 8;
    constexpr int LR = BM ? 1 : LM;
    static_assert(G <= AM || AM == 16, "the group is the power of two above the agent count");
    const int A = mv.A, L = mv.L;
    LV<G, uint32_t> a, pos, av, alive, arrived, occ, gems, beams[LR], err, n_ev, meta_step, actv, enabled, colw[CWM];
    LV<G, bool> me, env_ok(true), stepped;
    LV<G, uint64_t> evw[NWG];
    uint32_t beam_full[LM], beam_full_r[LR], bm[LM];
    for (int b = 0; b < LM; b++) { beam_full[b] = b < L ? mv.hdr->beam_full[b] : 0u; bm[b] = s.beams[b]; }
    for (int b = 0; b < LR; b++) beam_full_r[b] = beam_full[b];
    for (int i = 0; i < G; i++) {
        a.v[i] = (uint32_t)i; me.v[i] = i < A;
        pos.v[i] = i < A ? s.pos[i] : 0xFFFF0000u + (uint32_t)i;
        av.v[i] = i < A ? avail[i] : 0u;
        actv.v[i] = i < A ? act[i] : 4u;
        alive.v[i] = s.alive; arrived.v[i] = s.arrived; occ.v[i] = s.occ; gems.v[i] = s.gems; enabled.v[i] = mv.enabled;
        for (int b = 0; b < LR; b++) beams[b].v[i] = s.beams[b];
        for (int q = 0; q < CWM; q++) colw[q].v[i] = mv.colw[q];
    }
    // (bm is sized L on the device; here LM words, so that an out-of-range beam index would still be caught by ASan via `bm + L` below)
    std::vector<uint32_t> bm_exact(bm, bm + (L > 0 ? L : 1)), full_exact(beam_full, beam_full + (L > 0 ? L : 1));
    step_lanes<G, LR, ML1, PES, CWM, SHORTCUT, BM>(mv.cell_lay, mv.cell_meta, A, L, mv.W, mv.max_layers, beam_full_r, a, me, env_ok, enabled, colw,
                                                   actv, pos, av, alive, arrived, occ, gems, beams, err, evw, n_ev, meta_step, stepped, passes,
                                                   BM ? bm_exact.data() : nullptr, BM ? full_exact.data() : nullptr, BM ? mv.chain : 0u);
    const uint32_t e = uniform<G>(err);
    if (!uniform<G>(stepped)) return e;
    avail_lanes<G>(a, me, pos, occ, alive, arrived, meta_step, av);
    for (int i = 0; i < AM; i++) {
        if (i < A) { s.pos[i] = pos.v[i]; avail[i] = av.v[i]; }
    }
    s.alive = uniform<G>(alive); s.arrived = uniform<G>(arrived); s.occ = uniform<G>(occ); s.gems = uniform<G>(gems);
    if (BM) { for (int b = 0; b < L; b++) s.beams[b] = bm_exact[b]; }
    else { for (int b = 0; b < LR; b++) s.beams[b] = uniform<G>(beams[b]); }
    ev.clear();
    ev.n = uniform<G>(n_ev);
    for (int k = 0; k < NWG && k < Events<AM>::NW; k++) ev.w[k] = grp_or64<G>(evw[k]);
    return 0;
}
template <int AM, int LM, int G>
static uint32_t lanes_step(int engine, Env<AM, LM>& s, const uint32_t (&act)[AM], uint32_t (&avail)[AM], const MapView& mv, Events<AM>& ev, int64_t* passes) {
    const bool ml1 = mv.max_layers <= 1, pes = mv.per_env, sc = engine == 1;
#define HS_LANES(ML1, PES) (sc ? lanes_step_g<AM, LM, G, ML1, PES, true>(s, act, avail, mv, ev, passes) : lanes_step_g<AM, LM, G, ML1, PES, false>(s, act, avail, mv, ev, passes))
    if (ml1) return pes ? HS_LANES(true, true) : HS_LANES(true, false);
    return pes ? HS_LANES(false, true) : HS_LANES(false, false);
#undef HS_LANES
}
template <int AM, int LM>
static uint32_t lanes_step_any(int engine, Env<AM, LM>& s, const uint32_t (&act)[AM], uint32_t (&avail)[AM], const MapView& mv, Events<AM>& ev, int64_t* passes) {
    const int A = mv.A;  // group size of the step kernel (kernels.hip step_group)
    if (A <= 1) return lanes_step<AM, LM, 1>(engine, s, act, avail, mv, ev, passes);
    if (A <= 2) return lanes_step<AM, LM, 2>(engine, s, act, avail, mv, ev, passes);
    if (A <= 4) return lanes_step<AM, LM, 4>(engine, s, act, avail, mv, ev, passes);
    if constexpr (AM >= 8) { if (A <= 8) return lanes_step<AM, LM, 8>(engine, s, act, avail, mv, ev, passes); }
    if constexpr (AM >= 16) return lanes_step<AM, LM, 16>(engine, s, act, avail, mv, ev, passes);
    return 0xFFu;
}

enum { M_STEP = 0, M_RESET = 1, M_SET_STATE = 2, M_OBSERVE = 3, M_SOURCES = 4 };

template <int AM, int LM>
static void run(hs_batch* b, int mode, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset,
                const uint8_t* actions_in, const uint8_t* env_mask, uint32_t old_enabled) {
    const MapHeader* hdr = &b->map.header;
    const uint8_t* blob = b->map.blob.data();
    const int A = (int)hdr->A, L = (int)hdr->L;
    MapView mv;
    mv.cell_lay = reinterpret_cast<const uint64_t*>(blob + hdr->off_cell_lay);
    mv.cell_meta = reinterpret_cast<const uint32_t*>(blob + hdr->off_cell_meta);
    mv.hdr = hdr; mv.W = (int)hdr->W; mv.A = A; mv.L = L; mv.G = (int)hdr->G;
    mv.enabled = hdr->enabled_mask; mv.max_layers = hdr->max_layers;
    mv.per_env = false;
    mv.chain = hdr->chain_mask;
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(blob + hdr->off_dyn);
    std::vector<int8_t> tmpl(blob + hdr->off_template, blob + hdr->off_template + hdr->obs_stride);
    const uint32_t amask = (1u << A) - 1u;

    const int8_t* bare = reinterpret_cast<const int8_t*>(blob + hdr->off_bare);
    const uint32_t* elems = reinterpret_cast<const uint32_t*>(blob + hdr->off_elems);
    for (int64_t env = 0; env < b->n; env++) {
        if (b->per_env) {  // this env's colours / enabled flags replace the map's (kernels: LAUNCH_PER_ENV_SOURCES)
            mv.per_env = true;
            mv.enabled = b->src_enabled[env];
            for (int q = 0; q < MAX_SOURCES / 4; q++) {
                const uint8_t* c = &b->src_colour[env * 32 + 4 * q];
                mv.colw[q] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
            }
        }
        Env<AM, LM> s;
        for (int a = 0; a < AM; a++) s.pos[a] = (a < A) ? (uint32_t)b->pos[env * A + a] : 0xFFFF0000u + (uint32_t)a;
        const uint64_t bits = b->bits[env];
        s.alive = (uint32_t)bits & 0xFFFFu; s.arrived = (uint32_t)(bits >> 16) & 0xFFFFu; s.occ = (uint32_t)(bits >> 32) & 0xFFFFu;
        uint32_t ghost = (uint32_t)(bits >> GHOST_SHIFT);  // dead by set_state without an event (tables.h)
        s.gems = b->gems[env];
        for (int k = 0; k < LM; k++) s.beams[k] = (k < L) ? b->beams[env * L + k] : 0u;
        bool store_state = true, store_avail = false, touched = true;
        uint32_t avail[AM];
        Events<AM> ev;
        ev.clear();
        uint32_t err = 0, was_reset = 0;
        if (mode == M_STEP) {
            for (int a = 0; a < AM; a++) avail[a] = (a < A) ? (uint32_t)b->avail[env * A + a] : 0u;
            if ((flags & STEP_AUTO_RESET) && ((s.alive | ghost) != amask || s.arrived == amask)) {
                Cells<AM> at;
                reset_env<AM, LM>(s, mv, at);
                compute_avail<AM, LM>(s, mv, at, avail);
                was_reset = 1;
                ghost = 0;
            }
            Cells<AM> cur;
            load_cells<AM>(mv, s.pos, cur);
            uint32_t act[AM];
            if (flags & STEP_SAMPLE_ACTIONS) {
                const uint64_t key = action_step_key(seed, t);
                uint32_t hp = 0;
                for (int a = 0; a < AM; a++) {
                    if ((a & 1) == 0 && a < A) hp = action_hash_pair(key, (uint64_t)(env_offset + env), (uint32_t)(a >> 1));
                    act[a] = (a < A) ? sample_action(avail[a], action_field(hp, (uint32_t)a)) : 4u;
                }
            } else {
                const uint8_t* src = actions_in ? actions_in : b->actions.data();
                for (int a = 0; a < AM; a++) act[a] = (a < A) ? (uint32_t)src[env * A + a] : 4u;
            }
            for (int a = 0; a < A; a++) b->actions[env * A + a] = (uint8_t)act[a];
            for (int a = AM - 1; a >= 0; a--) {
                if (a < A) {
                    // `avail` is the cached list of the reference (world.rs:444-453).  It can only disagree with the
                    // static walk mask after a failed set_state left it stale (world.rs:588-594 returns before
                    // recomputing it); the reference would then index out of the grid and panic, we refuse the action.
                    const uint32_t walk = meta_walk(cur.meta[a]) | 16u;
                    if (act[a] > 4u || !((avail[a] >> act[a]) & 1u) || !((walk >> act[a]) & 1u)) err = (uint32_t)a + 1u;
                }
            }
            if (b->engine != 0) {
                // the lane-per-agent restatement does its own availability check; it must agree with the one above
                const uint32_t lane_err = lanes_step_any<AM, LM>(b->engine, s, act, avail, mv, ev, &b->lane_passes);
                if (lane_err != err) throw LaneDivergence();
            }
            if (err == 0) {
                if (b->engine == 0) {
                    Cells<AM> fin;
                    step_env<AM, LM>(s, act, mv, ev, cur, fin);
                    compute_avail<AM, LM>(s, mv, fin, avail);
                }
                store_avail = true;
            } else {
                store_state = was_reset != 0;
                store_avail = was_reset != 0;
            }
        } else if (mode == M_RESET) {
            if (!env_mask || env_mask[env]) {
                Cells<AM> at;
                reset_env<AM, LM>(s, mv, at);
                compute_avail<AM, LM>(s, mv, at, avail);
                store_avail = true;
                ghost = 0;
            } else {
                store_state = false;
                touched = false;
            }
        } else if (mode == M_SET_STATE) {
            uint32_t rp[AM];
            for (int a = 0; a < AM; a++) rp[a] = (a < A) ? (uint32_t)b->req_pos[env * A + a] : 0xFFFF0000u + (uint32_t)a;
            bool dirty = false;
            Cells<AM> at;
            err = set_state_env<AM, LM>(s, rp, b->req_gems[env], (uint32_t)b->req_alive[env], mv, ev, dirty, at);
            if (err != 0) ev.clear();
            {
                uint32_t died = 0;
                for (uint32_t q = 0; q < ev.n; q++) {
                    const uint32_t byte = (uint32_t)(ev.w[q >> 3] >> ((q & 7) * 8)) & 0xFFu;
                    if ((byte >> 4) == EV_DIED) died |= 1u << (byte & 15u);
                }
                if (err == 0) ghost = ~s.alive & ~died & amask;  // (a refused request leaves the bookkeeping as it was)
            }
            if (dirty) { load_cells<AM>(mv, s.pos, at); compute_avail<AM, LM>(s, mv, at, avail); store_avail = true; }
        } else if (mode == M_SOURCES) {
            for (int k = 0; k < LM; k++) {
                if (k < L) {
                    const bool was = (old_enabled >> k) & 1u, now = (mv.enabled >> k) & 1u;
                    if (was && !now) s.beams[k] = 0u;
                    if (!was && now) s.beams[k] = hdr->beam_full[k];
                }
            }
        } else {
            store_state = false;
        }
        if (store_state) {
            for (int a = 0; a < A; a++) b->pos[env * A + a] = (uint16_t)s.pos[a];
            b->bits[env] = (uint64_t)s.alive | ((uint64_t)s.arrived << 16) | ((uint64_t)s.occ << 32) | ((uint64_t)ghost << GHOST_SHIFT);
            b->gems[env] = s.gems;
            for (int k = 0; k < L; k++) b->beams[env * L + k] = s.beams[k];
        }
        if (store_avail) for (int a = 0; a < A; a++) b->avail[env * A + a] = (uint8_t)avail[a];
        if ((mode == M_STEP || mode == M_RESET || mode == M_SET_STATE) && touched) {
            b->err[env] = (uint8_t)err;
            b->evcount[env] = (uint8_t)(ev.n | (was_reset << 7));
            for (int k = 0; k < 2 * A; k++) b->events[env * 2 * A + k] = (uint8_t)(ev.w[k >> 3] >> ((k & 7) * 8));
            b->done[env] = ((s.alive | ghost) != amask || s.arrived == amask) ? 1 : 0;
        }
        if (mode == M_STEP) {
            int64_t n_gem = 0, n_exit = 0, n_died = 0;
            for (uint32_t k = 0; k < ev.n; k++) {
                const uint32_t ty = ((uint32_t)(ev.w[k >> 3] >> ((k & 7) * 8)) >> 4) & 3u;
                n_gem += ty == EV_GEM; n_exit += ty == EV_EXIT; n_died += ty == EV_DIED;
            }
            const int64_t bonus = (err == 0 && s.arrived == amask) ? 1 : 0;
            b->stats[0] += 1; b->stats[1] += A; b->stats[2] += n_gem; b->stats[3] += n_exit; b->stats[4] += n_died;
            b->stats[5] += err != 0; b->stats[6] += was_reset; b->stats[7] += n_gem + n_exit - n_died + bonus;
        }
        // phase 2
        if ((mode == M_STEP && (flags & STEP_NO_OBS)) || !hdr->obs_supported) continue;
        if (b->per_env) {  // same element evaluation as write_observations_env (obs_stream.hpp)
            std::vector<int8_t> row(bare, bare + hdr->obs_stride);
            for (uint32_t d = 0; d < hdr->n_elems; d++) {
                const uint32_t e = elems[d], cell = elem_cell(e), i5 = elem_index(e), off = elem_bit(e), type = elem_type(e);
                const uint32_t colour = b->src_colour[env * 32 + i5];
                if (type == ELEM_SOURCE) row[((uint32_t)A + colour) * hdr->HW + cell] = -1;
                else if (type == ELEM_TILE) { if ((s.beams[i5] >> off) & 1u) row[((uint32_t)A + colour) * hdr->HW + cell] = 1; }
                else if (!((s.gems >> i5) & 1u)) row[(uint32_t)(2 * A + 2) * hdr->HW + cell] = 1;
            }
            for (int a = 0; a < A; a++) row[(uint32_t)a * hdr->HW + cell_of(s.pos[a], mv.W)] = 1;
            std::memcpy(b->obs.data() + (size_t)env * hdr->obs_stride, row.data(), hdr->obs_stride);
            continue;
        }
        for (uint32_t d = 0; d < hdr->D; d++) {
            const uint64_t e = dyn[d];
            const uint32_t idx = dyn_index(e);
            int32_t v = dyn_base(e);
            const uint32_t n_refs = dyn_refs(e);
            const uint32_t r0 = dyn_ref0(e), r1 = dyn_ref1(e);
            const uint32_t gem = dyn_gem(e);
            if (n_refs >= 1 && ((s.beams[ref_word(r0)] >> ref_bit(r0)) & 1u)) v = 1;
            if (n_refs >= 2 && ((s.beams[ref_word(r1)] >> ref_bit(r1)) & 1u)) v = 1;
            if (gem != NO_GEM && !((s.gems >> gem) & 1u)) v = 1;
            tmpl[idx] = (int8_t)v;
        }
        for (int a = 0; a < A; a++) tmpl[(uint32_t)a * hdr->HW + cell_of(s.pos[a], mv.W)] = 1;
        std::memcpy(b->obs.data() + (size_t)env * hdr->obs_stride, tmpl.data(), hdr->obs_stride);
        for (int a = 0; a < A; a++) tmpl[(uint32_t)a * hdr->HW + cell_of(s.pos[a], mv.W)] = 0;
    }
}

static void dispatch(hs_batch* b, int mode, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset,
                     const uint8_t* actions_in, const uint8_t* env_mask, uint32_t old_enabled) {
    const int A = (int)b->map.header.A, L = (int)b->map.header.L;
    if (A <= 4 && L <= 4) run<4, 4>(b, mode, flags, seed, t, env_offset, actions_in, env_mask, old_enabled);
    else if (A <= 8 && L <= 8) run<8, 8>(b, mode, flags, seed, t, env_offset, actions_in, env_mask, old_enabled);
    else if (A <= 16 && L <= 16) run<16, 16>(b, mode, flags, seed, t, env_offset, actions_in, env_mask, old_enabled);
    else run<16, 32>(b, mode, flags, seed, t, env_offset, actions_in, env_mask, old_enabled);
}

extern "C" {
hs_batch* hs_create(const char* text, int64_t n, int* parse_error) {
    hs_batch* b = new hs_batch();
    int rc = parse_map(text, std::strlen(text), b->map);
    if (parse_error) *parse_error = rc;
    if (rc != 0) { delete b; return nullptr; }
    const MapHeader& h = b->map.header;
    b->n = n;
    const size_t A = h.A, L = h.L ? h.L : 1;
    b->pos.assign(n * A, 0); b->req_pos.assign(n * A, 0); b->req_alive.assign(n, 0);
    b->bits.assign(n, 0); b->gems.assign(n, 0); b->beams.assign(n * L, 0); b->req_gems.assign(n, 0);
    b->avail.assign(n * A, 0); b->actions.assign(n * A, 0); b->err.assign(n, 0); b->evcount.assign(n, 0);
    b->events.assign(n * 2 * A, 0); b->done.assign(n, 0); b->obs.assign((size_t)n * h.obs_stride, 0);
    std::memset(b->stats, 0, sizeof b->stats);
    dispatch(b, M_RESET, 0, 0, 0, 0, nullptr, nullptr, 0);
    return b;
}
void hs_free(hs_batch* b) { delete b; }
void hs_set_engine(hs_batch* b, int engine) { b->engine = engine; }
int64_t hs_lane_passes(hs_batch* b) { return b->lane_passes; }
void hs_reset(hs_batch* b, const uint8_t* mask) { dispatch(b, M_RESET, 0, 0, 0, 0, nullptr, mask, 0); }
void hs_step(hs_batch* b, const uint8_t* actions, uint32_t flags, uint64_t seed, uint64_t t, int64_t env_offset) {
    try {
        dispatch(b, M_STEP, flags, seed, t, env_offset, actions, nullptr, 0);
    } catch (const LaneDivergence&) {  // lanes of one group disagree on a replicated word / on the error code: poison the batch
        std::fill(b->err.begin(), b->err.end(), (uint8_t)0xEE);
    }
}
void hs_set_state(hs_batch* b) { dispatch(b, M_SET_STATE, 0, 0, 0, 0, nullptr, nullptr, 0); }
void hs_observe(hs_batch* b) { dispatch(b, M_OBSERVE, 0, 0, 0, 0, nullptr, nullptr, 0); }
void hs_set_source(hs_batch* b, int laser_id, int enabled, int agent_id) {
    const uint32_t old = b->map.header.enabled_mask;
    if (enabled >= 0) b->map.sources[laser_id].enabled = enabled != 0;
    if (agent_id >= 0) b->map.sources[laser_id].agent_id = agent_id;
    b->map.compile();
    dispatch(b, M_SOURCES, 0, 0, 0, 0, nullptr, nullptr, old);
}
// World.exit_pos = [...] (lle_map_set_exits + lle_batch_update_map): new tables, same dynamic state, observation rewritten
int hs_set_exits(hs_batch* b, const int32_t* exits_ij, int n_exits) {
    std::vector<Pos> ex((size_t)n_exits);
    for (int k = 0; k < n_exits; k++) ex[(size_t)k] = Pos{exits_ij[2 * k], exits_ij[2 * k + 1]};
    std::string why;
    const int rc = b->map.set_exits(ex, why);
    if (rc != 0) return rc;
    dispatch(b, M_OBSERVE, 0, 0, 0, 0, nullptr, nullptr, 0);
    return 0;
}
// LaserSource.set_colour / enable / disable per environment (lle_batch_set_sources; kernels.hip MODE_ENV_SOURCES)
void hs_set_sources(hs_batch* b, const uint8_t* colours, const uint32_t* enabled, const uint8_t* mask) {
    const MapHeader& h = b->map.header;
    const int A = (int)h.A, L = (int)h.L;
    if (!b->per_env) {
        b->per_env = true;
        b->src_colour.assign((size_t)b->n * 32, 0);
        b->src_enabled.assign((size_t)b->n, h.enabled_mask);
        for (int64_t env = 0; env < b->n; env++)
            for (int l = 0; l < L; l++) b->src_colour[env * 32 + l] = h.beam_colour[l];
    }
    // (the caller speaks of SOURCES -- colours [n][n_sources], bit s of the enabled mask --, the tables of beam WORDS: tables.h)
    const int S = (int)h.n_sources;
    for (int64_t env = 0; env < b->n; env++) {
        if (mask && !mask[env]) continue;
        bool bad = false, crosses = false;
        if (colours)
            for (int l = 0; l < L; l++) {
                if (!((h.word_mask >> l) & 1u)) continue;
                const int c = colours[env * S + h.word_source[l]];
                bad |= c >= A;
                crosses |= c < A && !((h.colour_ok[l] >> c) & 1u);
            }
        b->err[env] = bad ? ENV_INVALID_COLOUR : (crosses ? ENV_COLOUR_CROSSES_START : 0);
        if (bad || crosses) continue;
        uint32_t new_en = b->src_enabled[env];
        if (enabled) {
            new_en = 0u;
            for (int l = 0; l < L; l++) new_en |= ((enabled[env] >> h.word_source[l]) & 1u) << l;
            new_en &= h.word_mask;
        }
        for (int l = 0; l < L; l++) {
            const bool was = (b->src_enabled[env] >> l) & 1u, now = (new_en >> l) & 1u;
            if (was && !now) b->beams[env * L + l] = 0u;                 // LaserBeam::disable (laser.rs:74-77)
            if (!was && now) b->beams[env * L + l] = h.beam_full[l];     // LaserBeam::enable  (laser.rs:69-72)
            if (colours && ((h.word_mask >> l) & 1u)) b->src_colour[env * 32 + l] = colours[env * S + h.word_source[l]];
        }
        b->src_enabled[env] = new_en;
    }
    dispatch(b, M_OBSERVE, 0, 0, 0, 0, nullptr, nullptr, 0);
}

// ---- the other observation builders: the same per-element logic (observers_logic.hpp) and the same view tables
// (Map::compile_view) as observers.hip, evaluated on the host.  Output is the unpadded logical array.
static void view_rows(hs_batch* b, int kind, int param, int8_t* out, int64_t env_pitch, bool* supported) {
    std::vector<uint8_t> blob = b->map.compile_view(kind, param);
    ViewHeader v;
    std::memcpy(&v, blob.data(), sizeof v);
    *supported = v.supported != 0;
    const uint64_t* dyn = reinterpret_cast<const uint64_t*>(blob.data() + v.off_dyn);
    const int A = (int)v.A, L = (int)v.L;
    for (int64_t env = 0; env < b->n; env++) {
        std::vector<int8_t> row(blob.data() + v.off_template, blob.data() + v.off_template + v.obs_bytes);
        for (uint32_t d = 0; d < v.D; d++) {  // same evaluation as write_observations (obs_stream.hpp)
            const uint64_t e = dyn[d];
            const uint32_t refs = dyn_refs(e), gem = dyn_gem(e);
            const uint32_t r0 = dyn_ref0(e), r1 = dyn_ref1(e);
            uint32_t lit = 0;
            if (refs >= 1) lit |= (b->beams[env * (L ? L : 1) + ref_word(r0)] >> ref_bit(r0)) & 1u;
            if (refs >= 2) lit |= (b->beams[env * (L ? L : 1) + ref_word(r1)] >> ref_bit(r1)) & 1u;
            if (gem != NO_GEM) lit |= (~b->gems[env] >> gem) & 1u;
            row[dyn_index(e)] = (int8_t)(lit ? 1 : dyn_base(e));
        }
        for (int a = 0; a < A; a++) row[(uint32_t)v.agent_layer[a] * v.HW + cell_of(b->pos[env * A + a], (int)v.W)] = 1;
        std::memcpy(out + env * env_pitch, row.data(), v.obs_bytes);
    }
}

int hs_observe_as(hs_batch* b, int kind, int param, void* out) {
    const MapHeader& h = b->map.header;
    const int A = (int)h.A, L = (int)h.L, G = (int)h.G;
    bool ok = true;
    ObsTables T;
    T.cell_lay = reinterpret_cast<const uint64_t*>(b->map.blob.data() + h.off_cell_lay);
    T.cell_meta = reinterpret_cast<const uint32_t*>(b->map.blob.data() + h.off_cell_meta);
    T.beam_colour = h.beam_colour;
    T.A = A; T.H = (int)h.H; T.W = (int)h.W;
    switch (kind) {
        case OBS_LAYERED:
            view_rows(b, OBS_PERSPECTIVE, 0, static_cast<int8_t*>(out), (int64_t)h.obs_bytes, &ok);
            break;
        case OBS_LAYERED_PADDED:
            view_rows(b, kind, param, static_cast<int8_t*>(out), (int64_t)(2 * (A + param) + 4) * h.HW, &ok);
            break;
        case OBS_PERSPECTIVE:
            for (int k = 0; k < A; k++) {
                bool okk = true;
                view_rows(b, kind, k, static_cast<int8_t*>(out) + (int64_t)k * h.obs_bytes, (int64_t)A * h.obs_bytes, &okk);
                ok = ok && okk;
            }
            break;
        case OBS_PARTIAL: {
            for (const Source& s : b->map.sources)
                if (s.agent_id > A + 1) ok = false;
            if (!ok) break;
            const int k = param, row_bytes = A * (2 * A + 3) * k * k;
            for (int64_t env = 0; env < b->n; env++) {
                int8_t* row = static_cast<int8_t*>(out) + env * row_bytes;
                for (int a = 0; a < A; a++)
                    for (int w = 0; w < k * k; w++)
                        partial_cell(T, &b->pos[env * A], b->gems[env], &b->beams[env * (L ? L : 1)], a, w / k, w % k, k, row);
            }
            break;
        }
        default: {
            const int len = 3 * A + G;
            for (int64_t env = 0; env < b->n; env++)
                for (int e = 0; e < len; e++)
                    static_cast<float*>(out)[env * len + e] = state_elem(A, G, (int)h.H, (int)h.W, &b->pos[env * A], b->gems[env],
                                                                        (uint32_t)b->bits[env] & 0xFFFFu, e, kind == OBS_NORMALIZED_STATE);
        }
    }
    return ok ? 0 : -1;
}

void hs_available_actions(hs_batch* b, int walkable_lasers, uint8_t* out) {
    const MapHeader& h = b->map.header;
    const int A = (int)h.A, L = (int)h.L;
    ObsTables T;
    T.cell_lay = reinterpret_cast<const uint64_t*>(b->map.blob.data() + h.off_cell_lay);
    T.cell_meta = reinterpret_cast<const uint32_t*>(b->map.blob.data() + h.off_cell_meta);
    T.beam_colour = h.beam_colour;
    T.A = A; T.H = (int)h.H; T.W = (int)h.W;
    for (int64_t env = 0; env < b->n; env++)
        for (int a = 0; a < A; a++) {
            const uint32_t m = avail_bools(T, &b->pos[env * A], &b->beams[env * (L ? L : 1)], a, b->avail[env * A + a], walkable_lasers != 0);
            for (int act = 0; act < 5; act++) out[(env * A + a) * 5 + act] = (uint8_t)((m >> act) & 1u);
        }
}

// The static observation rebuilt from its bit form (tables.h off_tmpl_bits) with the arithmetic of the split-row prologue (step_kernel.hpp), next
// to the template itself: returns obs_stride, or 0 when the map carries no bit form.
int64_t hs_template_from_bits(hs_batch* b, int8_t* from_bits, int8_t* pristine) {
    const MapHeader& h = b->map.header;
    const uint8_t* blob = b->map.blob.data();
    std::memcpy(pristine, blob + h.off_template, h.obs_stride);
    if (!h.off_tmpl_bits) return 0;
    if (h.off_tmpl_bits != sizeof(MapHeader) || h.off_tmpl_bits + h.bits_bytes != h.off_cell_lay || h.tmpl_neg_n > TMPL_NEG_MAX) return -1;
    const uint16_t* bits = reinterpret_cast<const uint16_t*>(blob + h.off_tmpl_bits);
    for (uint32_t c = 0; c < h.n_chunks; c++) {
        const uint32_t v = bits[c];
        const uint32_t d[4] = {((v & 15u) * 0x00204081u) & 0x01010101u, (((v >> 4) & 15u) * 0x00204081u) & 0x01010101u,
                               (((v >> 8) & 15u) * 0x00204081u) & 0x01010101u, ((v >> 12) * 0x00204081u) & 0x01010101u};
        std::memcpy(from_bits + (size_t)c * 16, d, 16);
    }
    const uint32_t* neg = reinterpret_cast<const uint32_t*>(blob + h.off_tmpl_bits + tmpl_bits_bytes(h.n_chunks));
    for (uint32_t i = 0; i < h.tmpl_neg_n; i++) from_bits[neg[i]] = (int8_t)-1;
    return (int64_t)h.obs_stride;
}

// The table section [off_cell_lay, off_template) rebuilt from its packed image (tables.h off_packed) the way obs_stream.hpp expand_packed_tables fills
// LDS, next to the section itself: returns its size, 0 when the map carries no image, -1 when the image is malformed.
int64_t hs_tables_from_packed(hs_batch* b, uint8_t* from_packed, uint8_t* verbatim, int64_t cap, int* has_image) {
    const MapHeader& h = b->map.header;
    const uint8_t* blob = b->map.blob.data();
    const uint32_t bytes = h.off_template - h.off_cell_lay, HW = h.HW, n_lay = h.packed_n_lay;
    if ((int64_t)bytes > cap) return -1;
    std::memcpy(verbatim, blob + h.off_cell_lay, bytes);
    *has_image = h.off_packed != 0;
    if (!h.off_packed) return (int64_t)bytes;
    if (h.off_packed != h.off_bare + h.ext_bytes || (size_t)h.off_packed + h.packed_cap != b->map.blob.size() || h.packed_bytes > h.packed_cap) return -1;
    const uint8_t* pk = blob + h.off_packed;
    const uint16_t* m = reinterpret_cast<const uint16_t*>(pk);
    const uint16_t* idx = reinterpret_cast<const uint16_t*>(pk + packed_meta_bytes(HW));
    const uint64_t* lay = reinterpret_cast<const uint64_t*>(pk + packed_meta_bytes(HW) + packed_idx_bytes(n_lay));
    const uint8_t* tail = pk + packed_meta_bytes(HW) + packed_idx_bytes(n_lay) + ((n_lay * 8u + 15u) & ~15u);
    const uint32_t tail_bytes = h.off_template - h.off_dyn;
    if (packed_meta_bytes(HW) + packed_idx_bytes(n_lay) + ((n_lay * 8u + 15u) & ~15u) + tail_bytes != h.packed_bytes) return -1;
    std::memset(from_packed, 0, bytes);
    uint32_t* md = reinterpret_cast<uint32_t*>(from_packed + (h.off_cell_meta - h.off_cell_lay));
    for (uint32_t c = 0; c < HW; c++) md[c] = m[c];
    std::memcpy(from_packed + (h.off_dyn - h.off_cell_lay), tail, tail_bytes);
    uint64_t* ld = reinterpret_cast<uint64_t*>(from_packed);
    for (uint32_t i = 0; i < n_lay; i++) ld[idx[i]] = lay[i];
    return (int64_t)bytes;
}

void* hs_buffer(hs_batch* b, int which) {
    switch (which) {
        case LLE_BUF_POS: return b->pos.data();
        case LLE_BUF_BITS: return b->bits.data();
        case LLE_BUF_GEMS: return b->gems.data();
        case LLE_BUF_BEAMS: return b->beams.data();
        case LLE_BUF_AVAIL: return b->avail.data();
        case LLE_BUF_ACTIONS: return b->actions.data();
        case LLE_BUF_ERR: return b->err.data();
        case LLE_BUF_EVCOUNT: return b->evcount.data();
        case LLE_BUF_EVENTS: return b->events.data();
        case LLE_BUF_DONE: return b->done.data();
        case LLE_BUF_OBS: return b->obs.data();
        case LLE_BUF_STATS: return b->stats;
        case LLE_BUF_REQ_POS: return b->req_pos.data();
        case LLE_BUF_REQ_GEMS: return b->req_gems.data();
        case LLE_BUF_REQ_ALIVE: return b->req_alive.data();
    }
    return nullptr;
}
}
