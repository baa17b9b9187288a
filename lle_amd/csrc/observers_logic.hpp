// observers_logic.hpp -- per-element evaluation of the observation builders that are not the layered tensor
// (SURVEY.md section 8(f) rank 3), from the packed dynamic state of one environment and the static map tables.
//
//   PartialGenerator.observe      python/lle/observations.py:312-369   -> partial_cell()
//   StateGenerator.observe        python/lle/observations.py:137-159, src/bindings/world/pyworld_state.rs:79-101
//                                                                        -> state_elem()
//   LLE.available_actions         python/lle/env/env.py:146-163        -> avail_bools()
//
// Host + device (the test-only host build in tests/hostsim runs the same code against the CPU oracle).
#pragma once
#include <stdint.h>

#include "step_logic.hpp"
#include "tables.h"

namespace lle {

struct ObsTables {
    const uint64_t* cell_lay;    // [HW]
    const uint32_t* cell_meta;   // [HW]
    const uint8_t* beam_colour;  // [L] min(colour, 31)
    int A, H, W;
};

// One cell (wi, wj) of agent a's k x k window: every layer of that cell, written in the reference's order
// (agents, gems, exits, walls, lasers that are on, -1 at sources: observations.py:343-359) so that a laser colour
// A or A+1, whose layer index LASER_0 + colour falls on GEM / EXIT, overwrites exactly what it overwrites there.
// `row` is agent-major [A][2A+3][k][k]; the lane that owns the cell owns all of its layers.
// ZERO = false: the caller has cleared the row (the kernel does it with 16-byte LDS stores).
template <bool ZERO = true, typename Row>
LLE_HD void partial_cell(const ObsTables& T, const uint16_t* pos, uint32_t gems, const uint32_t* beams, int a, int wi,
                         int wj, int k, Row row) {
    const int A = T.A, layers = 2 * A + 3, kk = k * k, centre = k / 2;
    const int WALL = A, LASER_0 = A + 1, GEM = 2 * A + 1, EXIT = 2 * A + 2;   // observations.py:318-323
    Row cellp = row + (a * layers) * kk + wi * k + wj;
    if (ZERO)
        for (int l = 0; l < layers; l++) cellp[l * kk] = 0;
    const int i = (int)(pos[a] & 0xFFu) - centre + wi, j = (int)(pos[a] >> 8) - centre + wj;
    if (i < 0 || j < 0 || i >= T.H || j >= T.W) return;
    const uint32_t here = (uint32_t)i | ((uint32_t)j << 8);
    for (int a2 = 0; a2 < A; a2++)
        if (pos[a2] == here) cellp[a2 * kk] = 1;                 // dead agents included (agents_positions)
    const int c = i * T.W + j;
    const uint32_t meta = T.cell_meta[c], kind = meta_kind(meta), idx = meta_index(meta);
    if (kind == K_GEM && !((gems >> idx) & 1u)) cellp[GEM * kk] = 1;
    if (kind == K_EXIT) cellp[EXIT * kk] = 1;
    if (kind == K_WALL || kind == K_SOURCE) cellp[WALL * kk] = 1;  // wall_pos holds the sources too (parser_v1.rs:22-25)
    const uint64_t lay = T.cell_lay[c];
    for (int q = 0; q < 2; q++) {                                 // World.lasers(): two layers per cell (world.rs:159-172)
        const uint32_t e = lay_entry(lay, (int)q);
        if (!(e & LAY_VALID)) break;
        const uint32_t beam = lay_word(e), off = lay_bit(e);
        if ((beams[beam] >> off) & 1u) cellp[(LASER_0 + (int)T.beam_colour[beam]) * kk] = 1;
    }
    if (kind == K_SOURCE) cellp[(LASER_0 + (int)T.beam_colour[idx]) * kk] = -1;   // idx = laser_id of a source cell
}

// Element e of the state vector [i0, j0, ..., gems..., alive...] (length 3A + G).  The reference divides the f32
// positions by an int64/float64 `dimensions` array, i.e. in float64, and rounds the quotient to f32 on assignment.
LLE_HD float state_elem(int A, int G, int H, int W, const uint16_t* pos, uint32_t gems, uint32_t alive, int e, bool normalize) {
    if (e < 2 * A) {
        const int a = e >> 1;
        const double v = (e & 1) ? (double)(pos[a] >> 8) : (double)(pos[a] & 0xFFu);
        const double d = normalize ? ((e & 1) ? (double)W : (double)H) : 1.0;
        return (float)(v / d);
    }
    if (e < 2 * A + G) return ((gems >> (e - 2 * A)) & 1u) ? 1.0f : 0.0f;
    return ((alive >> (e - 2 * A - G)) & 1u) ? 1.0f : 0.0f;
}

// Availability of agent a as 5 bools in Action value order N,S,E,W,STAY.  With walkable_lasers = false an action is
// dropped when its target cell shows (World.lasers(): outer two layers) a laser that is on and of another colour --
// STAY included, whose target is the agent's own cell (env.py:155-162).
LLE_HD uint32_t avail_bools(const ObsTables& T, const uint16_t* pos, const uint32_t* beams, int a, uint32_t mask, bool walkable_lasers) {
    if (walkable_lasers) return mask & 31u;
    uint32_t out = 0;
    const int pi = (int)(pos[a] & 0xFFu), pj = (int)(pos[a] >> 8);
    const int DI[5] = {-1, 1, 0, 0, 0}, DJ[5] = {0, 0, 1, -1, 0};   // action.rs:18-26
    for (int act = 0; act < 5; act++) {
        if (!((mask >> act) & 1u)) continue;
        const int i = pi + DI[act], j = pj + DJ[act];
        bool blocked = false;
        if (i >= 0 && j >= 0 && i < T.H && j < T.W) {
            const uint64_t lay = T.cell_lay[i * T.W + j];
            for (int q = 0; q < 2; q++) {
                const uint32_t e = lay_entry(lay, (int)q);
                if (!(e & LAY_VALID)) break;
                const uint32_t beam = lay_word(e), off = lay_bit(e);
                if (((beams[beam] >> off) & 1u) && (int)T.beam_colour[beam] != a) blocked = true;
            }
        }
        if (!blocked) out |= 1u << act;
    }
    return out;
}

}  // namespace lle
