// step_logic.hpp -- per-environment state machine of the batched World (one lane = one environment).
//
// Bitmask restatement of the reference's sequential tile-object semantics:
//   World::step               src/core/world.rs:435-475
//   solve_vertex_conflicts    src/core/world.rs:365-378 + src/utils/mod.rs:18-36
//   move_agents               src/core/world.rs:477-505  (leave* -> pre_enter* -> enter*, each in agent order)
//   Tile/Laser/Gem/Void       src/core/tiles/{tile.rs:20-99, laser.rs:157-202, gem.rs:26-35, void.rs:13-22}
//   compute_available_actions src/core/world.rs:343-363
//   World::reset              src/core/world.rs:411-432
//   World::set_state          src/core/world.rs:515-597
//
// Representation: a beam (`LaserBeam{Vec<bool>}`) is a u32 mask, bit k = on at offset k; the occupant slot of a
// cell is one "occupant" bit per agent (agents never share a cell, so the slot of a cell is the agent standing
// on it with that bit set); gems are one collected-bit each.  A cell's stack of `Laser` wrappers is the static
// `cell_lay` entry (tables.h), outermost layer first.
//
// Everything is written against register arrays with compile-time indices (AM = max agents, LM = max sources of
// this instantiation): runtime-indexed private arrays would go to scratch memory on gfx950.
#pragma once
#include <stdint.h>

#include "tables.h"

#if defined(__HIPCC__)
#define LLE_HD __host__ __device__ __forceinline__
#else
#define LLE_HD inline
#endif

namespace lle {

// splitmix64 finaliser
LLE_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
// Counter-based action sampler (DESIGN.md "Action stream"): stateless in (seed, env, t, agent).
//   key   = mix64(seed + 0x9E3779B97F4A7C15 * (t + 1))                      uniform over a launch: scalar work
//   h     = lowbias32((lo(key) ^ lo(env) * 0x9E3779B1) + rotl(hi(key) ^ hi(env), 15) + rotl(0xC2B2AE35, 3 * pair + 1))
//   field = 16 bits of h per agent of the pair (agents 2 * pair and 2 * pair + 1)
// and the action is the k-th available one with k = (field16 * popcount(mask)) >> 16, uniform up to a 2^-16 bias.
// Three 32-bit multiplies per lane (integer multiplies run at quarter rate on the vector ALU; the 64-bit mixing this
// replaced cost twenty).
LLE_HD uint32_t rotl32(uint32_t x, uint32_t s) { return (x << (s & 31u)) | (x >> ((32u - s) & 31u)); }
LLE_HD uint64_t action_step_key(uint64_t seed, uint64_t t) { return mix64(seed + 0x9E3779B97F4A7C15ULL * (t + 1)); }
LLE_HD uint32_t action_hash_pair(uint64_t key, uint64_t env, uint32_t pair) {
    uint32_t h = ((uint32_t)key ^ ((uint32_t)env * 0x9E3779B1u)) + rotl32((uint32_t)(key >> 32) ^ (uint32_t)(env >> 32), 15u) +
                 rotl32(0xC2B2AE35u, 3u * pair + 1u);
    h ^= h >> 16; h *= 0x7FEB352Du;   // lowbias32 finaliser
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}
LLE_HD uint32_t action_field(uint32_t pair_hash, uint32_t agent) { return (pair_hash >> (16u * (agent & 1u))) & 0xFFFFu; }

LLE_HD uint32_t popc5(uint32_t m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popc(m);
#else
    return (uint32_t)__builtin_popcount(m);
#endif
}
// k-th set bit of the 5-bit availability mask in enum order N,S,E,W,STAY
LLE_HD uint32_t sample_action(uint32_t mask, uint32_t field16) {
    mask &= 31u;
    const uint32_t k = (field16 * popc5(mask)) >> 16;
    uint32_t act = 4;
#pragma unroll
    for (int b = 4; b >= 0; b--) {
        const uint32_t below = popc5(mask & ((1u << b) - 1u));
        act = (((mask >> b) & 1u) && below == k) ? (uint32_t)b : act;
    }
    return act;
}

// Static tables as seen by the logic (pointers may be LDS or global).
struct MapView {
    const uint64_t* cell_lay;
    const uint32_t* cell_meta;
    const MapHeader* hdr;  // uniform fields
    int W, A, L, G;
    uint32_t enabled;      // bit b: source b enabled
    uint32_t max_layers;
    // per-environment source colours (lle_batch_set_sources): 4 colours per word, colour of beam b = byte b; when
    // `per_env` is set they replace the colours baked into cell_lay
    bool per_env;
    uint32_t colw[MAX_SOURCES / 4];
    uint32_t chain = 0;    // bit b: beam word b continues the beam of word b - 1 (tables.h chain_mask)
};

// colour of beam b from the env's packed colour words (compile-time indexed selects: no private-array indexing)
template <int NWORDS>
LLE_HD uint32_t colour_get(const uint32_t (&colw)[NWORDS], uint32_t b) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < NWORDS; k++) w = ((b >> 2) == (uint32_t)k) ? colw[k] : w;
    return (w >> ((b & 3u) * 8u)) & 0xFFu;
}
// cell_lay entry with the colour field of every valid layer replaced by the env's colour of that layer's beam
// NL: layers that can be valid (1 on maps without crossing beams: the other three slots are empty and stay as they are)
template <int NWORDS, int NL = MAX_CELL_LAYERS>
LLE_HD uint64_t recolour_lay(uint64_t lay, const uint32_t (&colw)[NWORDS]) {
    uint64_t out = NL < MAX_CELL_LAYERS ? (lay & ~((1ull << (16 * NL)) - 1ull)) : 0ull;
#pragma unroll
    for (int q = 0; q < NL; q++) {
        uint32_t e = lay_entry(lay, (int)q);
        const uint32_t c = colour_get<NWORDS>(colw, lay_word(e));
        e = (e & LAY_VALID) ? ((e & 0x7FFu) | (c << 11)) : e;
        out |= (uint64_t)e << (16 * q);
    }
    return out;
}

template <int AM, int LM>
struct Env {
    uint32_t pos[AM];  // i | j << 8
    uint32_t alive, arrived, occ;  // bit per agent
    uint32_t gems;                 // bit per gem: collected
    uint32_t beams[LM];
};

// Ordered event list of one step, one byte per event (type << 4 | agent), packed little-endian into 64-bit words.
// Everything below is written predicated (selects instead of branches): the lanes of a wave are different
// environments, so a data-dependent `if` costs an exec-mask round trip for every lane.
template <int AM>
struct Events {
    static constexpr int NW = (2 * AM + 7) / 8;
    uint64_t w[NW];
    uint32_t n;
    LLE_HD void clear() {
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] = 0;
        n = 0;
    }
    LLE_HD void push_if(bool c, uint32_t type, uint32_t agent) {
        const uint64_t byte = c ? (uint64_t)((type << 4) | agent) : 0ull;
        const uint32_t word = n >> 3, sh = (n & 7u) * 8u;
#pragma unroll
        for (int k = 0; k < NW; k++) w[k] |= (NW == 1 || word == (uint32_t)k) ? (byte << sh) : 0ull;
        n += c ? 1u : 0u;
    }
};

LLE_HD uint32_t cell_of(uint32_t p, int W) { return (p & 0xFFu) * (uint32_t)W + (p >> 8); }

// Table entries of the cells the agents stand on.  Looked up for all agents at once: the lookups are LDS reads with
// a ~100-cycle round trip each, so they are issued back to back and waited for once instead of one by one.
template <int AM>
struct Cells {
    uint64_t lay[AM];
    uint32_t meta[AM];
};
template <int AM>
LLE_HD void load_cells(const MapView& mv, const uint32_t (&pos)[AM], Cells<AM>& out) {
#pragma unroll
    for (int a = 0; a < AM; a++) {
        const uint32_t c = (a < mv.A) ? cell_of(pos[a], mv.W) : 0u;
        out.lay[a] = mv.cell_lay[c];
        if (mv.per_env) out.lay[a] = recolour_lay<MAX_SOURCES / 4>(out.lay[a], mv.colw);
        out.meta[a] = mv.cell_meta[c];
    }
}

template <int LM>
LLE_HD uint32_t beam_get(const uint32_t (&b)[LM], uint32_t idx) {
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < LM; k++) r = (idx == (uint32_t)k) ? b[k] : r;
    return r;
}
template <int LM>
LLE_HD void beam_set_if(uint32_t (&b)[LM], uint32_t idx, bool c, uint32_t v) {
#pragma unroll
    for (int k = 0; k < LM; k++) b[k] = (c && idx == (uint32_t)k) ? v : b[k];
}

// A re-light / cut of beam word b that runs to the end of the beam (LaserBeam::turn_on / turn_off, laser.rs:50-59) takes the
// following words of a chained beam whole.  `c`: the operation happens; `fill`: re-light (bits beyond a word's length are trimmed by
// canonicalise()) or cut.  Maps without a beam longer than 32 cells have chain == 0: a uniform branch.
template <int LM>
LLE_HD void chain_rest(uint32_t (&beams)[LM], uint32_t b, bool c, bool fill, uint32_t chain) {
    if (!chain) return;
    bool carry = false;
#pragma unroll
    for (int k = 0; k < LM; k++) {
        carry = carry && ((chain >> k) & 1u);
        beams[k] = carry ? (fill ? 0xFFFFFFFFu : 0u) : beams[k];
        carry = carry || (c && b == (uint32_t)k);
    }
}

// Tile::leave on a laser stack (laser.rs:199-202 -> :157-162 -> :50-55): every layer whose bit is off is
// re-lit from its offset to the end, unless the source is disabled.  `doit`: the agent is alive (world.rs:483).
template <int LM>
LLE_HD void lasers_leave(uint32_t (&beams)[LM], uint64_t lay, bool doit, const MapView& mv) {
    for (uint32_t k = 0; k < mv.max_layers; k++) {  // uniform trip count; absent layers are predicated off
        const uint32_t e = lay_entry(lay, (int)k);
        const uint32_t b = lay_word(e), off = lay_bit(e);
        const uint32_t m = beam_get<LM>(beams, b);
        const bool c = doit && (e & LAY_VALID) && !((m >> off) & 1u) && ((mv.enabled >> b) & 1u);
        beam_set_if<LM>(beams, b, c, m | (0xFFFFFFFFu << off));  // bits beyond the length are trimmed by canonicalise()
        chain_rest<LM>(beams, b, c, true, mv.chain);
    }
}

// Tile::pre_enter on a laser stack (laser.rs:173-182): an ALIVE agent of the beam's colour switches the beam
// off from its offset on.  (Order over layers is irrelevant: each layer is a different beam.)
template <int LM>
LLE_HD void lasers_pre_enter(uint32_t (&beams)[LM], uint64_t lay, uint32_t agent, bool alive, const MapView& mv) {
    for (uint32_t k = 0; k < mv.max_layers; k++) {
        const uint32_t e = lay_entry(lay, (int)k);
        const uint32_t b = lay_word(e), off = lay_bit(e), colour = lay_colour(e);
        const bool c = alive && (e & LAY_VALID) && colour == agent && ((mv.enabled >> b) & 1u);
        const uint32_t m = beam_get<LM>(beams, b);
        beam_set_if<LM>(beams, b, c, m & ((1u << off) - 1u));
        chain_rest<LM>(beams, b, c, false, mv.chain);
    }
}

// Laser::enter (laser.rs:184-197): the first layer (outermost first) that is on and of another colour stops the
// agent; whether one exists does not depend on the order.
template <int LM>
LLE_HD bool lasers_block(const uint32_t (&beams)[LM], uint64_t lay, uint32_t agent, const MapView& mv) {
    bool blocked = false;
    for (uint32_t k = 0; k < mv.max_layers; k++) {
        const uint32_t e = lay_entry(lay, (int)k);
        const uint32_t b = lay_word(e), off = lay_bit(e), colour = lay_colour(e);
        const uint32_t m = beam_get<LM>(beams, b);
        blocked |= (e & LAY_VALID) && ((m >> off) & 1u) && (colour != agent);
    }
    return blocked;
}

// Tile::enter for agent a at its cell (tile.rs:29-50).  Returns true if the agent died in this call.
//   blocked by a lit beam of another colour: alive -> dies (AgentDied), dead -> nothing; the wrapped tile is NOT entered.
//   otherwise the innermost tile takes the agent as occupant; Exit: arrive once (AgentExit, no alive check);
//   Gem: collect once (GemCollected, no alive check); Void: dies if alive (AgentDied).
template <int AM, int LM, bool EMIT>
LLE_HD bool enter_agent(Env<AM, LM>& s, uint32_t a, uint64_t lay, uint32_t meta, const MapView& mv, Events<AM>& ev) {
    const uint32_t bit = 1u << a;
    const bool is_alive = (s.alive & bit) != 0;
    const bool blocked = lasers_block<LM>(s.beams, lay, a, mv);
    const uint32_t kind = meta_kind(meta);
    const uint32_t gbit = 1u << gem_bit(meta_index(meta));  // (cells without a gem carry NO_INDEX: the shift stays defined, the bit unused)
    const bool inner = !blocked;
    const bool ev_exit = inner && kind == K_EXIT && !(s.arrived & bit);
    const bool ev_gem = inner && kind == K_GEM && !(s.gems & gbit);
    const bool died = is_alive && (blocked || kind == K_VOID);
    s.occ |= inner ? bit : 0u;
    s.arrived |= ev_exit ? bit : 0u;
    s.gems |= ev_gem ? gbit : 0u;
    s.alive &= died ? ~bit : 0xFFFFFFFFu;
    if (EMIT) ev.push_if(died || ev_exit || ev_gem, died ? EV_DIED : (ev_gem ? EV_GEM : EV_EXIT), a);
    return died;
}

// One call of World::move_agents (world.rs:477-505).  old_lay: laser stacks of the cells alive agents leave from;
// nw: table entries of the cells every agent enters.
template <int AM, int LM>
LLE_HD bool move_agents(Env<AM, LM>& s, const uint64_t (&old_lay)[AM], const Cells<AM>& nw, const MapView& mv, Events<AM>& ev) {
    const uint32_t alive0 = s.alive;  // leave and pre_enter both see the flags of before this pass's enter loop
    s.occ &= ~alive0;                 // Tile::leave: `slot.take()` for every alive agent
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) lasers_leave<LM>(s.beams, old_lay[a], (alive0 >> a) & 1u, mv);
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) lasers_pre_enter<LM>(s.beams, nw.lay[a], (uint32_t)a, (alive0 >> a) & 1u, mv);
    bool died = false;
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) died |= enter_agent<AM, LM, true>(s, (uint32_t)a, nw.lay[a], nw.meta[a], mv, ev);
    return died;
}

// bits beyond each beam's length back to 0 (keeps the stored masks canonical)
template <int AM, int LM>
LLE_HD void canonicalise(Env<AM, LM>& s, const MapView& mv) {
#pragma unroll
    for (int b = 0; b < LM; b++)
        if (b < mv.L) s.beams[b] &= mv.hdr->beam_full[b];
}

// compute_available_actions (world.rs:343-363) as 5-bit masks (bit = Action value).
template <int AM, int LM>
LLE_HD void compute_avail(const Env<AM, LM>& s, const MapView& mv, const Cells<AM>& at, uint32_t (&avail)[AM]) {
#pragma unroll
    for (int a = 0; a < AM; a++) {
        if (a >= mv.A) { avail[a] = 0; continue; }
        const bool can_move = ((s.alive >> a) & 1u) && !((s.arrived >> a) & 1u);
        const uint32_t walk = meta_walk(at.meta[a]);  // `at`: table entries of the agents' current cells
        uint32_t blocked = 0;
#pragma unroll
        for (int o = 0; o < AM; o++) {
            if (o < mv.A && o != a) {
                const int d = (int)s.pos[o] - (int)s.pos[a];
                uint32_t hit = (d == -1) ? 1u : 0u;   // North: i - 1
                hit |= (d == 1) ? 2u : 0u;            // South: i + 1
                hit |= (d == 256) ? 4u : 0u;          // East:  j + 1
                hit |= (d == -256) ? 8u : 0u;         // West:  j - 1
                blocked |= ((s.occ >> o) & 1u) ? hit : 0u;  // only an occupant blocks (tile.rs:86-99)
            }
        }
        avail[a] = 16u | (can_move ? (walk & ~blocked) : 0u);  // Stay is always available
    }
}

LLE_HD uint32_t apply_action(uint32_t p, uint32_t act) {
    // action.rs:18-26 deltas on the packed i | j << 8 form
    const int d = (act == 0) ? -1 : (act == 1) ? 1 : (act == 2) ? 256 : (act == 3) ? -256 : 0;
    return (uint32_t)((int)p + d);
}

// World::step after the availability check.  cur: table entries of the agents' cells before the step; on return
// `fin` holds those of the cells they end on (for compute_avail).
template <int AM, int LM>
LLE_HD void step_env(Env<AM, LM>& s, const uint32_t (&actions)[AM], const MapView& mv, Events<AM>& ev, const Cells<AM>& cur,
                     Cells<AM>& fin) {
    uint32_t np[AM];
#pragma unroll
    for (int a = 0; a < AM; a++) np[a] = (a < mv.A) ? apply_action(s.pos[a], actions[a]) : 0xFFFF0000u + (uint32_t)a;
    // solve_vertex_conflicts: every agent whose target is shared goes back to its current cell, until stable
    bool conflict = true;
    while (conflict) {
        conflict = false;
        uint32_t dup = 0;
#pragma unroll
        for (int i = 0; i < AM; i++)
#pragma unroll
            for (int j = i + 1; j < AM; j++)
                dup |= (j < mv.A && np[i] == np[j]) ? ((1u << i) | (1u << j)) : 0u;
#pragma unroll
        for (int i = 0; i < AM; i++) np[i] = ((dup >> i) & 1u) ? s.pos[i] : np[i];
        conflict = dup != 0;
    }
    load_cells<AM>(mv, np, fin);
    bool died = move_agents<AM, LM>(s, cur.lay, fin, mv, ev);
#pragma unroll
    for (int a = 0; a < AM; a++) s.pos[a] = np[a];
    while (died) died = move_agents<AM, LM>(s, fin.lay, fin, mv, ev);
    canonicalise<AM, LM>(s, mv);
}

// Tile::reset over the whole grid (tile.rs:75-84): occupants cleared, gems uncollected, every enabled beam fully
// on (a disabled beam stays dark: laser.rs:50-53).
template <int AM, int LM>
LLE_HD void reset_tiles(Env<AM, LM>& s, const MapView& mv) {
    s.occ = 0;
    s.gems = 0;
#pragma unroll
    for (int b = 0; b < LM; b++) s.beams[b] = (b < mv.L && ((mv.enabled >> b) & 1u)) ? mv.hdr->beam_full[b] : 0u;
}

// World::reset (world.rs:411-432); events of the initial enter are dropped.  `at`: table entries of the start cells.
template <int AM, int LM>
LLE_HD void reset_env(Env<AM, LM>& s, const MapView& mv, Cells<AM>& at) {
    reset_tiles<AM, LM>(s, mv);
    s.alive = (mv.A >= 32) ? 0xFFFFFFFFu : ((1u << mv.A) - 1u);
    s.arrived = 0;
#pragma unroll
    for (int a = 0; a < AM; a++) s.pos[a] = (a < mv.A) ? (uint32_t)mv.hdr->start[a] : 0xFFFF0000u + (uint32_t)a;
    load_cells<AM>(mv, s.pos, at);
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) lasers_pre_enter<LM>(s.beams, at.lay[a], (uint32_t)a, true, mv);
    Events<AM> ev;
    ev.clear();
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) enter_agent<AM, LM, false>(s, (uint32_t)a, at.lay[a], at.meta[a], mv, ev);
    canonicalise<AM, LM>(s, mv);
}

// reset tiles, collect the requested direct gems, pre_enter with the agents' CURRENT alive flags, then
// agent.reset() + enter (+ die if requested dead) per agent: world.rs:543-586.  Returns whether the resulting
// get_state() equals the target (world.rs:588-589).
template <int AM, int LM, bool EMIT>
LLE_HD bool apply_state(Env<AM, LM>& s, const uint32_t (&tpos)[AM], uint32_t tgems, uint32_t talive, const MapView& mv,
                        Events<AM>& ev, Cells<AM>& at) {
    load_cells<AM>(mv, tpos, at);
    reset_tiles<AM, LM>(s, mv);
    s.gems = tgems & mv.hdr->direct_gems;  // only direct Tile::Gem are collected up-front (world.rs:550-554)
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) lasers_pre_enter<LM>(s.beams, at.lay[a], (uint32_t)a, (s.alive >> a) & 1u, mv);
#pragma unroll
    for (int a = 0; a < AM; a++)
        if (a < mv.A) s.pos[a] = tpos[a];
#pragma unroll
    for (int a = 0; a < AM; a++) {
        if (a < mv.A) {
            const uint32_t bit = 1u << a;
            s.alive |= bit;      // agent.reset()
            s.arrived &= ~bit;
            enter_agent<AM, LM, EMIT>(s, (uint32_t)a, at.lay[a], at.meta[a], mv, ev);
            if (!((talive >> a) & 1u)) s.alive &= ~bit;
        }
    }
    canonicalise<AM, LM>(s, mv);
    return s.gems == tgems && s.alive == talive;
}

// World::set_state (world.rs:515-597) for one request.  Returns ENV_OK, ENV_INVALID_WORLD_STATE (duplicate
// positions: nothing touched; or final mismatch: NO rollback, like the reference), ENV_OUT_OF_WORLD_POSITION
// (nothing touched) or ENV_INVALID_AGENT_POSITION (after the reference's rollback through set_state(current)).
// `avail_dirty`: whether compute_available_actions ran in the reference.
template <int AM, int LM>
LLE_HD uint8_t set_state_env(Env<AM, LM>& s, const uint32_t (&req_pos)[AM], uint32_t req_gems, uint32_t req_alive,
                             const MapView& mv, Events<AM>& ev, bool& avail_dirty, Cells<AM>& at) {
    avail_dirty = false;
    const uint32_t amask = (1u << mv.A) - 1u, gmask = (mv.G >= 32) ? 0xFFFFFFFFu : ((1u << mv.G) - 1u);
    req_alive &= amask;
    req_gems &= gmask;
    bool dup = false, oob = false, unwalkable = false;
#pragma unroll
    for (int i = 0; i < AM; i++) {
        if (i >= mv.A) continue;
#pragma unroll
        for (int j = i + 1; j < AM; j++)
            if (j < mv.A && req_pos[i] == req_pos[j]) dup = true;
        if ((req_pos[i] & 0xFFu) >= mv.hdr->H || (req_pos[i] >> 8) >= mv.hdr->W) oob = true;
    }
    if (dup) return ENV_INVALID_WORLD_STATE;
    if (oob) return ENV_OUT_OF_WORLD_POSITION;
#pragma unroll
    for (int a = 0; a < AM; a++) {
        if (a >= mv.A) continue;
        const uint32_t kind = meta_kind(mv.cell_meta[cell_of(req_pos[a], mv.W)]);
        if (kind == K_WALL || kind == K_SOURCE) unwalkable = true;
    }
    if (unwalkable) {
        // pre_enter failed -> self.set_state(&current_state).unwrap() -> Err(InvalidAgentPosition)
        uint32_t cur_pos[AM];
#pragma unroll
        for (int a = 0; a < AM; a++) cur_pos[a] = s.pos[a];
        Events<AM> dropped;
        dropped.clear();
        // the reference `.unwrap()`s this inner call: if the snapshot cannot be restored (a collected gem under a
        // beam) it panics before compute_available_actions; otherwise the availability lists were recomputed
        avail_dirty = apply_state<AM, LM, false>(s, cur_pos, s.gems, s.alive, mv, dropped, at);
        return ENV_INVALID_AGENT_POSITION;
    }
    if (!apply_state<AM, LM, true>(s, req_pos, req_gems, req_alive, mv, ev, at)) return ENV_INVALID_WORLD_STATE;
    avail_dirty = true;
    return ENV_OK;
}

}  // namespace lle
