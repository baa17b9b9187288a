"""Decoding of the packed per-environment buffers (include/lle_hip.h, LLE_BUF_*) into the reference's vocabulary.

Pure numpy on host copies; shared by the `World` facade and by the tests.  Layouts:
  pos u8[A,2] (i,j); bits u64: alive 0-15 | arrived 16-31 | occupant 32-47; gems u32 bit g = collected;
  beams u32[Lw]: beam WORDS -- the tile at offset k of a beam is bit k % 32 of word (first word of its source) + k // 32
  (lle_laser_tile.word / .bit; word == laser_id, bit == offset unless a beam is longer than 32 cells); avail u8[A] bit a = Action a; events u8[2A] = type << 4 | agent.
"""
import numpy as np

# order of the reference's available-action lists (src/core/world.rs:349-351): Stay, then N, E, S, W
AVAIL_ORDER = (4, 0, 2, 1, 3)


def positions(pos):
    return [(int(p[0]), int(p[1])) for p in np.asarray(pos).reshape(-1, 2)]


def agent_bits(bits, n_agents):
    b = int(bits)
    alive = [bool((b >> a) & 1) for a in range(n_agents)]
    arrived = [bool((b >> (16 + a)) & 1) for a in range(n_agents)]
    occupant = [bool((b >> (32 + a)) & 1) for a in range(n_agents)]
    return alive, arrived, occupant


def gem_bits(gems, n_gems):
    g = int(gems)
    return [bool((g >> k) & 1) for k in range(n_gems)]


def avail_lists(avail):
    return [[a for a in AVAIL_ORDER if (int(m) >> a) & 1] for m in np.asarray(avail).reshape(-1)]


def events_list(evcount, events):
    n = int(evcount) & 0x7F
    ev = np.asarray(events).reshape(-1)
    return [(int(ev[k]) >> 4, int(ev[k]) & 15) for k in range(n)]


def beam_bits(beams, first_word, length):
    """on / off of the `length` tiles of the beam whose words start at `first_word` (== laser_id on maps without long beams)."""
    bm = np.asarray(beams).reshape(-1)
    return [bool((int(bm[first_word + k // 32]) >> (k % 32)) & 1) for k in range(length)]


def lasers_listing(laser_tiles, sources, beams):
    """World.lasers (src/core/world.rs:159-172): rows (i, j, laser_id, agent_id, is_on, is_enabled), outer layer first."""
    out = []
    bm = np.asarray(beams).reshape(-1)
    for t in laser_tiles:
        s = sources[t.laser_id]
        on = (int(bm[t.word]) >> t.bit) & 1
        out.append((int(t.i), int(t.j), int(t.laser_id), int(s.agent_id), int(on), int(s.enabled)))
    return out


def pack_positions(positions_ij):
    return np.asarray(positions_ij, dtype=np.uint8).reshape(-1, 2)


def pack_bits(flags):
    v = 0
    for k, f in enumerate(flags):
        if f:
            v |= 1 << k
    return v
