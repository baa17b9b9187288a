// step_lanes.hpp -- World.step with one LANE PER AGENT: the state machine of step_kernel (step_kernel.hpp), written once
// for the device and for a host build.
//
// On the device a lane value is a register and the G lanes of an environment are neighbouring lanes of a wavefront;
// cross-lane traffic is DPP / ds_swizzle.  For the host (tests/hostsim, g++ with -fsanitize=address,undefined; never part of
// the product) the same source runs with every lane value held G times:
//   LV<G, T>            device: T.  host: T v[G], reads and writes go to the CURRENT lane.
//   LLE_LANES(G) { }    device: the block, once.  host: the block once per lane, in lane order.
//   lane_xor / grp_or   device: quad-perm DPP / ds_swizzle.  host: read the other lanes' values.
//   any_lane / uniform  device: __any / the value itself.  host: OR over the lanes / lane 0 (all lanes must agree).
// The host build is only faithful if every cross-lane read sees FINAL values, i.e. if a value is written in one
// LLE_LANES block and read across lanes in a LATER one -- the lock-step the wavefront provides for free.  That is the
// discipline of this file; on the device the block boundaries vanish.
//
// What makes the lane split legal (same results as the sequential reference, src/core/world.rs:477-505):
//   * leaves of one pass commute (each only turns bits ON, and a leave skipped because an earlier one already lit its
//     bit would have been a no-op): beam |= OR over the group of every lane's suffix;
//   * pre-enters commute (each only clears a suffix): beam &= AND over the group of every lane's prefix;
//   * enter reads the beams (final after leave + pre-enter) and touches only the agent's own flags, its own cell's gem
//     and the occupant slot of its own cell (agents never share a cell), so the enters of one pass are independent;
//     their events are ordered by agent id = lane order (prefix count inside the group);
//   * the three loops stay in the reference's order, and passes repeat while any agent of the environment died.
#pragma once
#include "step_logic.hpp"

namespace lle {

#if defined(__HIPCC__)
// ------------------------------------------------------------------------------------------------ device
#define LLE_LANE_FN __device__ __forceinline__
template <int G, typename T> using LV = T;
#define LLE_LANES(G)
#define LLE_LANE_INDEX 0

template <int J>
__device__ __forceinline__ uint32_t lane_xor_raw(uint32_t v) {
    static_assert(J >= 1 && J < 16, "group offsets only");
    if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    if (J == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, true);  // quad_perm [3,2,1,0]
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (J << 10));                     // lane ^ J within 32 lanes
}
template <int G, int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) { return lane_xor_raw<J>(v); }
template <int G>
__device__ __forceinline__ uint32_t grp_or(uint32_t v) {
    if (G > 1) v |= lane_xor_raw<1>(v);
    if (G > 2) v |= lane_xor_raw<2>(v);
    if (G > 4) v |= lane_xor_raw<4>(v);
    if (G > 8) v |= lane_xor_raw<8>(v);
    return v;
}
template <int G>
__device__ __forceinline__ uint64_t grp_or64(uint64_t v) {
    return (uint64_t)grp_or<G>((uint32_t)v) | ((uint64_t)grp_or<G>((uint32_t)(v >> 32)) << 32);
}
template <int G>
__device__ __forceinline__ bool any_lane(bool v) { return __any(v); }
template <int G>
__device__ __forceinline__ bool uniform(bool v) { return v; }
// beam words kept in the environment's LDS record (BM below): OR / AND without return (ds_or_b32 / ds_and_b32)
__device__ __forceinline__ void mem_or(uint32_t* p, uint32_t m) { (void)__hip_atomic_fetch_or(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void mem_and(uint32_t* p, uint32_t m) { (void)__hip_atomic_fetch_and(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

#else
// ------------------------------------------------------------------------------------------------ host (test builds only)
#define LLE_LANE_FN inline
struct LaneCtx {
    static int& lane() { static thread_local int l = 0; return l; }
};
#define LLE_LANES(G) for (::lle::LaneCtx::lane() = 0; ::lle::LaneCtx::lane() < (G); ::lle::LaneCtx::lane()++)
#define LLE_LANE_INDEX (::lle::LaneCtx::lane())

template <int G, typename T>
struct LV {
    T v[G];
    LV() { for (int i = 0; i < G; i++) v[i] = T(); }
    explicit LV(T x) { for (int i = 0; i < G; i++) v[i] = x; }
    operator T() const { return v[LaneCtx::lane()]; }
    LV& operator=(T x) { v[LaneCtx::lane()] = x; return *this; }
    LV& operator=(const LV& o) { v[LaneCtx::lane()] = o.v[LaneCtx::lane()]; return *this; }
    LV(const LV&) = default;
    LV& operator|=(T x) { v[LaneCtx::lane()] |= x; return *this; }
    LV& operator&=(T x) { v[LaneCtx::lane()] &= x; return *this; }
    LV& operator+=(T x) { v[LaneCtx::lane()] += x; return *this; }
};
template <int G, int J>
inline uint32_t lane_xor(const LV<G, uint32_t>& x) { return x.v[LaneCtx::lane() ^ J]; }
template <int G>
inline uint32_t grp_or(const LV<G, uint32_t>& x) { uint32_t r = 0; for (int i = 0; i < G; i++) r |= x.v[i]; return r; }
template <int G>
inline uint64_t grp_or64(const LV<G, uint64_t>& x) { uint64_t r = 0; for (int i = 0; i < G; i++) r |= x.v[i]; return r; }
template <int G>
inline bool any_lane(const LV<G, bool>& x) { bool r = false; for (int i = 0; i < G; i++) r |= x.v[i]; return r; }
// a value every lane of the group must agree on (anything derived from group reductions)
struct LaneDivergence {};
template <int G, typename T>
inline T uniform(const LV<G, T>& x) { for (int i = 1; i < G; i++) if (!(x.v[i] == x.v[0])) throw LaneDivergence(); return x.v[0]; }
inline void mem_or(uint32_t* p, uint32_t m) { *p |= m; }
inline void mem_and(uint32_t* p, uint32_t m) { *p &= m; }
#endif

// value of `v` in the group lane whose agent id is (a ^ J), for every J in 1..G-1, fed to f(J, value)
template <int G, int J = 1, typename V, typename F>
LLE_LANE_FN void for_each_other(const V& v, F&& f) {
    if constexpr (J < G) {
        f(J, lane_xor<G, J>(v));
        for_each_other<G, J + 1>(v, f);
    }
}

template <int LM, typename E>
LLE_LANE_FN uint32_t beam_get_lv(const E (&b)[LM], uint32_t idx) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < LM; k++) r = (idx == (uint32_t)k) ? (uint32_t)b[k] : r;
    return r;
}

// ---- World.step for the G lanes of one environment, after the action of every lane is known.
// In / out (all per lane; the env-wide words are replicated in the group's lanes and stay identical):
//   pos, avail          this agent's packed position (i | j << 8) and cached availability mask
//   alive, arrived, occ, gems, beams[]   the environment's packed state
// Out: err (0, or 1 + lowest agent whose action was not available: nothing else is then touched), evw[] / n_ev (the ordered
// event bytes, this lane's share: OR them over the group), meta_step (cell meta of the agent's final cell), stepped.
// SHORTCUT: skip a pass that cannot change anything (see below); always on in the product, switchable in test builds.
// BM ("beams in memory", maps with more than 8 sources): the beam masks are not registers of every lane but the L words
// `bm[0..L)` of the environment -- its hand-over record in LDS on the device -- and `full_tab[b]` = (1 << length) - 1.
// With the masks in registers a pass walks ALL LM of them against the layers of the lane's cell, twice, with a group
// reduction per beam each: O(LM x layers) instructions per pass, most of them for beams the lane has nothing to do
// with.  A cell carries at most four beams, so here a lane asks for its <= 4 re-lights (mem_or) and cuts (mem_and) by beam
// index and reads back the <= 4 masks its enter needs.  Legal for the reason the register form is (leaves commute,
// pre-enters commute, all leaves precede all pre-enters): the conditions of every leave are read in one LLE_LANES
// block, the ORs happen in the next, the ANDs in the one after -- on the device the LDS operations of a wavefront
// execute in order and the lanes of an environment share a wavefront.  `beams` / `h_beam_full` are then unused
// (instantiate with LM = 1).
template <int G, int LM, bool ML1, bool PES, int CWM, bool SHORTCUT = true, bool BM = false>
LLE_LANE_FN void step_lanes(const uint64_t* cell_lay, const uint32_t* cell_meta, int A, int L, int W, uint32_t table_max_layers,
                       const uint32_t (&h_beam_full)[LM], const LV<G, uint32_t>& a, const LV<G, bool>& me, const LV<G, bool>& env_ok,
                       const LV<G, uint32_t>& enabled, const LV<G, uint32_t> (&colw)[CWM > 0 ? CWM : 1], const LV<G, uint32_t>& act,
                       LV<G, uint32_t>& pos, const LV<G, uint32_t>& avail, LV<G, uint32_t>& alive, LV<G, uint32_t>& arrived,
                       LV<G, uint32_t>& occ, LV<G, uint32_t>& gems, LV<G, uint32_t> (&beams)[LM], LV<G, uint32_t>& err,
                       LV<G, uint64_t> (&evw)[(2 * G + 7) / 8], LV<G, uint32_t>& n_ev, LV<G, uint32_t>& meta_step, LV<G, bool>& stepped,
                       int64_t* passes_executed = nullptr /* test builds: counts the move_agents passes that ran */,
                       uint32_t* bm = nullptr, const uint32_t* full_tab = nullptr /* BM only */,
                       uint32_t chain = 0 /* BM only: bit b = word b continues the beam of word b - 1 (tables.h chain_mask) */) {
    constexpr int NW = (2 * G + 7) / 8;
    const uint32_t max_layers = ML1 ? 1u : table_max_layers;

    // ---- availability check (world.rs:444-453): lowest offending agent, before any mutation.  The cached list can
    // only disagree with the static walk mask after a failed set_state left it stale; such an action is refused.
    LV<G, uint64_t> lay_cur, lay_new, lay_from;
    LV<G, uint32_t> badbit, np, dupv, meta_new, bit;
    LLE_LANES(G) {
        bit = 1u << a;
        const uint32_t cur_cell = me ? cell_of(pos, W) : 0u;
        uint64_t lc = cell_lay[cur_cell];
        if (PES) {
            uint32_t cw[CWM > 0 ? CWM : 1];
#pragma unroll
            for (int q = 0; q < (CWM > 0 ? CWM : 1); q++) cw[q] = colw[q];
            lc = recolour_lay<(CWM > 0 ? CWM : 1), (ML1 ? 1 : MAX_CELL_LAYERS)>(lc, cw);
        }
        lay_cur = lc;
        const uint32_t walk_cur = meta_walk(cell_meta[cur_cell]) | 16u;
        const bool bad = me && (act > 4u || !(((avail & walk_cur) >> (act & 7u)) & 1u));
        badbit = bad ? (uint32_t)bit : 0u;
    }
    LLE_LANES(G) {
        const uint32_t badmask = grp_or<G>(badbit);
#if defined(__HIP_DEVICE_COMPILE__)
        err = badmask ? (uint32_t)__ffs((int)badmask) : 0u;
#else
        err = badmask ? (uint32_t)__builtin_ffs((int)badmask) : 0u;
#endif
#pragma unroll
        for (int k = 0; k < NW; k++) evw[k] = 0ull;
        n_ev = 0u;
        meta_step = 0u;
        stepped = false;
    }
    LV<G, bool> proceed;
    LLE_LANES(G) { proceed = env_ok && err == 0u; }
    if (!uniform<G>(proceed)) return;

    // target cell (src/action.rs:18-26 on the packed i | j << 8 form); lanes without an agent keep a unique sentinel
    LLE_LANES(G) { np = me ? apply_action(pos, act) : (uint32_t)pos; }
    // solve_vertex_conflicts (world.rs:365-378): every agent whose target is shared goes back to its cell
    LV<G, bool> again(true);
    while (any_lane<G>(again)) {
        LLE_LANES(G) {
            bool dup = false;
            const uint32_t mine = np;
            for_each_other<G>(np, [&](int, uint32_t other) { dup |= (other == mine); });
            dupv = dup ? 1u : 0u;
        }
        LLE_LANES(G) {
            np = dupv ? (uint32_t)pos : (uint32_t)np;
            again = grp_or<G>(dupv) != 0u;
        }
    }
    LV<G, uint32_t> kind, gbit;
    LLE_LANES(G) {
        const uint32_t new_cell = me ? cell_of(np, W) : 0u;
        uint64_t ln = cell_lay[new_cell];
        if (PES) {
            uint32_t cw[CWM > 0 ? CWM : 1];
#pragma unroll
            for (int q = 0; q < (CWM > 0 ? CWM : 1); q++) cw[q] = colw[q];
            ln = recolour_lay<(CWM > 0 ? CWM : 1), (ML1 ? 1 : MAX_CELL_LAYERS)>(ln, cw);
        }
        lay_new = ln;
        const uint32_t mn = cell_meta[new_cell];
        meta_new = mn;
        kind = meta_kind(mn);
        gbit = 1u << gem_bit(meta_index(mn));  // (cells without a gem carry NO_INDEX: the shift stays defined, the bit unused)
        lay_from = (uint64_t)lay_cur;  // pass 1 leaves the old cells, later passes the new ones
    }

    // move_agents passes (world.rs:464-472)
    LV<G, bool> go(true), me_alive, first_pass(true);
    LV<G, uint32_t> light[LM], lit[LM], cutv[LM], alive0, p1v, p2v, gemv;
    LV<G, bool> died, ev_exit, ev_gem, has_ev, inner;
    LV<G, uint32_t> rq_b[4], rq_m[4], rq_any;  // BM: this lane's re-light requests (beam, suffix), one per layer of its cell
    while (any_lane<G>(go)) {
        // leave (laser.rs:199-202,157-162): what the alive agents of the env re-light, per beam
        LLE_LANES(G) {
            alive0 = (uint32_t)alive;
            me_alive = go && me && (alive0 & bit);
            if constexpr (BM) {
                uint32_t any = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t rb = 0, rm = 0;
                    if ((uint32_t)k < max_layers) {
                        const uint32_t eo = lay_entry((uint64_t)lay_from, k);
                        const uint32_t b = lay_word(eo), off = lay_bit(eo);
                        const bool valid = (eo & LAY_VALID) != 0;  // (an empty layer slot names no beam: nothing is read for it)
                        const uint32_t cur = valid ? bm[b] : 0xFFFFFFFFu, full = valid ? full_tab[b] : 0u;
                        const bool lo = me_alive && valid && ((enabled >> b) & 1u) && !((cur >> off) & 1u);
                        rb = valid ? b : 0u;
                        rm = lo ? ((0xFFFFFFFFu << off) & full) : 0u;
                    }
                    rq_b[k] = rb;
                    rq_m[k] = rm;
                    any |= rm;
                }
                rq_any = any;
            } else {
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    uint32_t lt = 0;
                    if (b < L) {
                        for (uint32_t k = 0; k < max_layers; k++) {
                            const uint32_t eo = lay_entry((uint64_t)lay_from, k);
                            const bool lo = me_alive && (eo & LAY_VALID) && lay_word(eo) == (uint32_t)b &&
                                            !(((uint32_t)beams[b] >> lay_bit(eo)) & 1u);
                            lt |= lo ? (0xFFFFFFFFu << lay_bit(eo)) : 0u;
                        }
                        lt = ((enabled >> b) & 1u) ? lt : 0u;
                    }
                    light[b] = lt;
                }
            }
        }
        LLE_LANES(G) {
            uint32_t al = 0;
            if constexpr (BM) {
                al = grp_or<G>(rq_any);
            } else {
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    lit[b] = (b < L) ? grp_or<G>(light[b]) : 0u;
                    al |= lit[b];
                }
            }
            // A pass after the first one leaves and re-enters the SAME cells.  If no alive agent re-lights anything, the
            // beams cannot change (the owners' cuts are repeated as they are), so every enter repeats its outcome: alive
            // agents stay alive, occupants / arrivals / gems are already recorded, the dead stay blocked or buried.
            // The pass is then a no-op and `while agent_died` ends (world.rs:468-472).
            if (SHORTCUT && !first_pass && al == 0u) go = false;
        }
        if constexpr (BM) {
            LLE_LANES(G) {  // every leave's condition has been read: re-light
                if (go) {
#pragma unroll
                    for (int k = 0; k < 4; k++)
                        if ((uint32_t)rq_m[k]) {
                            mem_or(bm + (uint32_t)rq_b[k], (uint32_t)rq_m[k]);
                            // LaserBeam::turn_on runs to the end of the Vec (laser.rs:50-55): the following words of a chained beam whole
                            if (chain)
                                for (uint32_t w = (uint32_t)rq_b[k] + 1u; (chain >> w) & 1u; w++) mem_or(bm + w, full_tab[w]);
                        }
                }
            }
        }
        // pre_enter (laser.rs:173-182): what the alive agents of the beam's colour cut
        LLE_LANES(G) {
            if constexpr (BM) {
                if (go && me_alive) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if ((uint32_t)k < max_layers) {
                            const uint32_t en = lay_entry((uint64_t)lay_new, k);
                            const uint32_t b = lay_word(en);
                            if ((en & LAY_VALID) && lay_colour(en) == a && ((enabled >> b) & 1u)) {
                                mem_and(bm + b, (1u << lay_bit(en)) - 1u);
                                if (chain)  // LaserBeam::turn_off likewise (laser.rs:57-59)
                                    for (uint32_t w = b + 1u; (chain >> w) & 1u; w++) mem_and(bm + w, 0u);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < LM; b++) {
                    uint32_t keep = 0xFFFFFFFFu;
                    if (b < L) {
                        for (uint32_t k = 0; k < max_layers; k++) {
                            const uint32_t en = lay_entry((uint64_t)lay_new, k);
                            const bool pe = go && me_alive && (en & LAY_VALID) && lay_word(en) == (uint32_t)b && lay_colour(en) == a;
                            keep &= pe ? ((1u << lay_bit(en)) - 1u) : 0xFFFFFFFFu;
                        }
                    }
                    cutv[b] = ((b < L) && ((enabled >> b) & 1u)) ? ~keep : 0u;
                }
            }
        }
        LLE_LANES(G) {
            if (go) {
                if (passes_executed && LLE_LANE_INDEX == 0) ++*passes_executed;
                occ &= ~(uint32_t)alive0;  // Tile::leave: slot.take() for every alive agent
                if constexpr (!BM) {
#pragma unroll
                    for (int b = 0; b < LM; b++)
                        if (b < L) beams[b] = ((uint32_t)beams[b] | lit[b]) & ~grp_or<G>(cutv[b]);
                }
            }
        }
        // enter (tile.rs:29-50, laser.rs:184-197)
        LLE_LANES(G) {
            bool blocked = false;
            for (uint32_t k = 0; k < max_layers; k++) {
                const uint32_t en = lay_entry((uint64_t)lay_new, k);
                uint32_t m;
                if constexpr (BM) m = (en & LAY_VALID) ? bm[lay_word(en)] : 0u;
                else m = beam_get_lv<LM>(beams, lay_word(en));
                blocked |= (en & LAY_VALID) && ((m >> lay_bit(en)) & 1u) && (lay_colour(en) != a);
            }
            const bool is_alive = (alive & bit) != 0;
            inner = go && me && !blocked;
            ev_exit = inner && kind == K_EXIT && !(arrived & bit);
            ev_gem = inner && kind == K_GEM && !(gems & gbit);
            died = go && me && is_alive && (blocked || kind == K_VOID);
            has_ev = died || ev_exit || ev_gem;
            p1v = (died ? (uint32_t)bit : 0u) | (ev_exit ? (uint32_t)bit << 16 : 0u);
            p2v = (inner ? (uint32_t)bit : 0u) | (has_ev ? (uint32_t)bit << 16 : 0u);
            gemv = ev_gem ? (uint32_t)gbit : 0u;
        }
        LLE_LANES(G) {
            const uint32_t p1 = grp_or<G>(p1v), p2 = grp_or<G>(p2v);
            if (go) {
                gems |= grp_or<G>(gemv);
                alive &= ~(p1 & 0xFFFFu);
                arrived |= p1 >> 16;
                occ |= p2 & 0xFFFFu;
                const uint32_t evmask = p2 >> 16;  // agents with an event this pass: ordered by agent id
#if defined(__HIP_DEVICE_COMPILE__)
                const uint32_t slot = n_ev + (uint32_t)__popc(evmask & (bit - 1u));
                const uint32_t n_new = (uint32_t)__popc(evmask);
#else
                const uint32_t slot = n_ev + (uint32_t)__builtin_popcount(evmask & (bit - 1u));
                const uint32_t n_new = (uint32_t)__builtin_popcount(evmask);
#endif
                const uint64_t byte = has_ev ? (uint64_t)(((died ? EV_DIED : (ev_gem ? EV_GEM : EV_EXIT)) << 4) | a) : 0ull;
#pragma unroll
                for (int k = 0; k < NW; k++) evw[k] |= (NW == 1 || (slot >> 3) == (uint32_t)k) ? (byte << ((slot & 7u) * 8u)) : 0ull;
                n_ev += n_new;
                go = (p1 & 0xFFFFu) != 0;  // while agent_died
            }
            lay_from = (uint64_t)lay_new;
            first_pass = false;
        }
    }
    LLE_LANES(G) {
        pos = (uint32_t)np;
        if constexpr (!BM) {
#pragma unroll
            for (int b = 0; b < LM; b++)
                if (b < L) beams[b] &= h_beam_full[b];
        }
        meta_step = (uint32_t)meta_new;
        stepped = true;
    }
}

// compute_available_actions (world.rs:343-363) for this lane's agent after a step: bit = Action value, Stay always.
// `meta_step` = cell meta of the agent's cell (static walk mask in bits 8-11); only an OCCUPANT blocks (tile.rs:86-99).
template <int G>
LLE_LANE_FN void avail_lanes(const LV<G, uint32_t>& a, const LV<G, bool>& me, const LV<G, uint32_t>& pos, const LV<G, uint32_t>& occ,
                        const LV<G, uint32_t>& alive, const LV<G, uint32_t>& arrived, const LV<G, uint32_t>& meta_step,
                        LV<G, uint32_t>& avail) {
    LLE_LANES(G) {
        const uint32_t bit = 1u << a, mine = pos, aa = a, oc = occ;
        const bool can_move = me && (alive & bit) && !(arrived & bit);
        uint32_t blocked_dirs = 0;
        for_each_other<G>(pos, [&](int j, uint32_t other) {
            const int d = (int)other - (int)mine;
            uint32_t hit = (d == -1) ? 1u : 0u;
            hit |= (d == 1) ? 2u : 0u;
            hit |= (d == 256) ? 4u : 0u;
            hit |= (d == -256) ? 8u : 0u;
            blocked_dirs |= ((oc >> (aa ^ (uint32_t)j)) & 1u) ? hit : 0u;
        });
        avail = 16u | (can_move ? (meta_walk(meta_step) & ~blocked_dirs) : 0u);
    }
}

}  // namespace lle
