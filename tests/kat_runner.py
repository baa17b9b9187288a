"""Runs the known-answer scripts of tests/golden/kat_world.json against any implementation of the World surface.

An implementation is a factory `make(map_str=None, level=None)` returning an object with the small surface
of oracle.oracle.OracleWorld (positions(), alive(), step(), set_state(), lasers(), obs() ...) and raising an
exception carrying `.kind` (and `.agent`) on errors.  The same scripts drive the CPU oracle (tests -m "not gpu")
and the HIP-backed lle_amd.World facade (tests -m gpu).
"""
import json
import os
from collections import Counter

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_world.json")
DELTAS = {0: (-1, 0), 1: (0, 1), 2: (1, 0), 3: (0, -1)}  # direction codes N,E,S,W (direction.rs:20-27)


def load_cases():
    with open(GOLDEN) as f:
        return json.load(f)["cases"]


def _kind(exc):
    return getattr(exc, "kind", type(exc).__name__)


def _pairs(x):
    return [tuple(p) for p in x]


def _outer_lasers(w):
    """first listed laser per position = the outer layer (what the reference's get_laser helpers return)."""
    out = {}
    for (i, j, lid, col, on, en) in w.lasers():
        out.setdefault((i, j), (lid, col, bool(on), bool(en)))
    return out


def beam_positions(w, laser_id):
    i, j, d, _col, _en, ln = w.sources()[laser_id]
    di, dj = DELTAS[d]
    return [(i + di * (k + 1), j + dj * (k + 1)) for k in range(ln)]


def check_static(w, st):
    for key, val in st.items():
        if key == "start_pos":
            assert _pairs(w.start_pos) == _pairs(val)
        elif key == "exit_pos":
            assert _pairs(w.exit_pos) == _pairs(val)
        elif key == "n_exits":
            assert len(w.exit_pos) == val
        elif key == "gem_pos_contains":
            assert set(_pairs(val)) <= set(_pairs(w.gem_pos))
        elif key == "exit_pos_contains":
            assert set(_pairs(val)) <= set(_pairs(w.exit_pos))
        elif key == "wall_pos_contains":
            assert set(_pairs(val)) <= set(_pairs(w.wall_pos))
        elif key == "n_walls":
            assert len(w.wall_pos) == val
        elif key == "sources":
            assert [(s[0], s[1], s[3]) for s in w.sources()] == _pairs(val)
        elif key == "n_sources":
            assert w.n_sources == val
        elif key == "n_agents":
            assert w.n_agents == val
        elif key == "height":
            assert w.height == val
        elif key == "width":
            assert w.width == val
        elif key == "n_laser_colours":
            assert len({s[3] for s in w.sources()}) == val
        else:
            raise KeyError(f"unknown static key {key}")


def check_obs_consistent(w, obs):
    """python/tests/test_observations.py:126-159 restated: every listed object shows on its layer."""
    A = w.n_agents
    L0, WALL, VOID, GEM, EXIT = A, 2 * A, 2 * A + 1, 2 * A + 2, 2 * A + 3
    for (i, j) in w.wall_pos:
        if not any((s[0], s[1]) == (i, j) and L0 + s[3] == WALL for s in w.sources()):
            assert obs[WALL, i, j] == 1
    for (i, j), c in zip(w.gem_pos, w.gems_collected()):
        if not c:
            assert obs[GEM, i, j] == 1
    for (i, j) in w.exit_pos:
        assert obs[EXIT, i, j] == 1
    for (i, j, _lid, col, on, _en) in w.lasers():
        if on:
            assert obs[L0 + col, i, j] == 1
    for s in w.sources():
        assert obs[L0 + s[3], s[0], s[1]] == -1
    for a, (i, j) in enumerate(w.positions()):
        assert obs[a, i, j] == 1
        assert int(obs[a].sum()) == 1


def check_expect(w, ex, derived):
    for key, val in ex.items():
        if key == "op":
            continue
        if key.startswith("derived_") and not derived:
            continue
        k = key[len("derived_"):] if key.startswith("derived_") else key
        if k == "positions":
            assert _pairs(w.positions()) == _pairs(val), (w.positions(), val)
        elif k == "alive":
            assert w.alive() == val, (w.alive(), val)
        elif k == "alive_of":
            for a, v in val.items():
                assert w.alive()[int(a)] == v
        elif k == "arrived":
            assert w.arrived() == val
        elif k == "n_arrived":
            assert sum(w.arrived()) == val
        elif k == "gems":
            assert w.gems_collected() == val, (w.gems_collected(), val)
        elif k == "n_gems_collected":
            assert w.n_gems_collected() == val
        elif k == "avail_sets":
            got = [sorted(a) for a in w.available_actions()]
            assert got == [sorted(v) for v in val], (got, val)
            for a in w.available_actions():  # reference order: [Stay, N, E, S, W] filtered (world.rs:349-351)
                order = [4, 0, 2, 1, 3]
                assert a == [x for x in order if x in a]
        elif k == "avail_includes":
            for a, v in zip(w.available_actions(), val):
                assert set(v) <= set(a)
        elif k == "avail_excludes":
            for a, v in zip(w.available_actions(), val):
                assert not (set(v) & set(a))
        elif k == "n_joint_actions":
            assert int(np.prod([len(a) for a in w.available_actions()])) == val
        elif k == "lasers_on":
            outer = _outer_lasers(w)
            for (i, j, on) in val:
                assert outer[(i, j)][2] == on, ((i, j), outer[(i, j)], on)
        elif k == "lasers_enabled":
            outer = _outer_lasers(w)
            for (i, j, en) in val:
                assert outer[(i, j)][3] == en
        elif k == "all_lasers":
            assert all(bool(l[4]) == (val == "on") for l in w.lasers()), w.lasers()
        elif k == "n_lasers":
            assert len(w.lasers()) == val
        elif k == "laser_colours":
            outer = _outer_lasers(w)
            for (i, j, c) in val:
                assert outer[(i, j)][1] == c
        elif k == "all_laser_colour":
            assert all(l[3] == val for l in w.lasers())
        elif k == "laser_colour_by_row":
            for l in w.lasers():
                if str(l[0]) in val:
                    assert l[3] == val[str(l[0])]
        elif k == "laser_ids_by_row":
            for l in w.lasers():
                assert l[2] == val[str(l[0])]
        elif k == "lasers_per_id":
            cnt = Counter(l[2] for l in w.lasers())
            assert {str(a): b for a, b in cnt.items()} == val
        elif k == "no_laser_at":
            pos = {(l[0], l[1]) for l in w.lasers()}
            assert not (pos & set(_pairs(val)))
        elif k == "beam":
            for lid, cells in val.items():
                assert beam_positions(w, int(lid)) == _pairs(cells)
        elif k == "beam_len":
            for lid, ln in val.items():
                assert w.sources()[int(lid)][5] == ln
        elif k == "beam_bits":
            for lid, bits in val.items():
                assert w.beam_bits(int(lid)) == bits, (w.beam_bits(int(lid)), bits)
        elif k == "sources":
            assert [(s[0], s[1], s[3]) for s in w.sources()] == _pairs(val), (w.sources(), val)
        elif k == "sources_enabled":
            got = {(s[0], s[1]): bool(s[4]) for s in w.sources()}
            for (i, j, en) in val:
                assert got[(i, j)] == en
        elif k == "exit_pos":
            assert _pairs(w.exit_pos) == _pairs(val), (w.exit_pos, val)
        elif k == "n_exits":
            assert len(w.exit_pos) == val
        elif k == "exit_pos_contains":
            assert set(_pairs(val)) <= set(_pairs(w.exit_pos))
        elif k == "tile_agent":
            for (i, j, a) in val:
                assert w.tile_agent(i, j) == a, ((i, j), w.tile_agent(i, j), a)
        elif k == "obs_shape":
            assert list(w.obs().shape) == val
        elif k == "obs_shape_formula":
            assert w.obs().shape == (2 * w.n_agents + 4, w.height, w.width)
        elif k == "obs_cells":
            obs = w.obs()
            for (c, i, j, v) in val:
                assert obs[c, i, j] == v, ((c, i, j), int(obs[c, i, j]), v)
        elif k == "obs_layer_all":
            obs = w.obs()
            for (c, v) in val:
                assert np.all(obs[c] == v)
        elif k == "obs_layer_exact":
            obs = w.obs()
            for c, cells in val.items():
                want = np.zeros((w.height, w.width), np.int8)
                for (i, j) in cells:
                    want[i, j] = 1
                assert np.array_equal(obs[int(c)], want)
        elif k == "obs_consistent":
            check_obs_consistent(w, w.obs())
        else:
            raise KeyError(f"unknown expect key {key}")


def check_events(ev, spec, derived):
    ev = _pairs(ev)
    if "events" in spec:
        assert ev == _pairs(spec["events"]), (ev, spec["events"])
    if "n_events" in spec:
        assert len(ev) == spec["n_events"], (ev, spec["n_events"])
    if "event_types" in spec:
        assert [e[0] for e in ev] == spec["event_types"]
    if "event_multiset" in spec:
        assert Counter(ev) == Counter(_pairs(spec["event_multiset"])), ev
    if derived and "derived_events" in spec:
        assert ev == _pairs(spec["derived_events"]), (ev, spec["derived_events"])


class KatBindingError(Exception):
    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind


def _generic_clone(make, case, w):
    c = make(map_str=case.get("map"), level=case.get("level"))
    if _pairs(c.exit_pos) != _pairs(w.exit_pos):  # (the config the reference clones from carries the current exits)
        c.set_exits(_pairs(w.exit_pos))
    for k, s in enumerate(w.sources()):  # the config carries the sources' current colour / enabled flag
        c.set_source(k, enabled=bool(s[4]), colour=s[3])
    c.set_state(w.positions(), w.gems_collected(), w.alive())
    return c


def _binding_set_colour(w, laser_id, colour):
    """PyLaserSource.set_agent_id (src/bindings/tiles/pylaser_source.rs:107-141) on the core surface: usize conversion,
    colour < n_agents, the CORE colour is set (:114-119), and only then the beam (the tiles World::lasers() exposes for this
    laser id) is checked against the possible starts of every OTHER agent (:121-139)."""
    if colour < 0:
        raise KatBindingError("OverflowError")
    if colour >= w.n_agents:
        raise KatBindingError("ValueError")
    w.set_source(laser_id, colour=colour)
    cells = {(l[0], l[1]) for l in w.lasers() if l[2] == laser_id}
    for agent, start in enumerate(w.start_pos):
        if agent != colour and tuple(start) in cells:
            raise KatBindingError("ValueError")


def _set_agent_position(w, agent, pos):
    """PyWorld.set_agent_position (pyworld.rs:282-299): id check, then set_state(get_state() with one position replaced)."""
    if hasattr(w, "set_agent_position"):
        return w.set_agent_position(agent, pos)
    if agent >= w.n_agents:
        raise KatBindingError("AgentIdOutOfBounds")
    positions = [tuple(p) for p in w.positions()]
    positions[agent] = pos
    return w.set_state(positions, w.gems_collected(), w.alive())


def run_case(make, case, derived=True):
    """make(map_str=None, level=None) -> world.  Raises AssertionError on any mismatch."""
    if "parse_error" in case:
        try:
            make(map_str=case.get("map"), level=case.get("level"))
        except Exception as e:  # noqa: BLE001
            assert _kind(e) == case["parse_error"], (_kind(e), case["parse_error"])
            return
        raise AssertionError(f"expected parse error {case['parse_error']}")
    w = make(map_str=case.get("map"), level=case.get("level"))
    if "static" in case:
        check_static(w, case["static"])
    for op in case["script"]:
        kind = op["op"]
        if kind == "reset":
            w.reset()
        elif kind == "step":
            if "error" in op:
                before = (w.positions(), w.gems_collected(), w.alive())
                try:
                    w.step(op["actions"])
                except Exception as e:  # noqa: BLE001
                    assert _kind(e) in op["error"].split("|"), (_kind(e), op["error"])
                    if "error_agent" in op:
                        assert e.agent == op["error_agent"]
                    # step errors leave the world untouched (world.rs:436-453)
                    assert (w.positions(), w.gems_collected(), w.alive()) == before
                else:
                    raise AssertionError(f"expected {op['error']}")
            else:
                check_events(w.step(op["actions"]), op, derived)
        elif kind == "set_state":
            if "error" in op:
                try:
                    w.set_state(op["positions"], op["gems"], op["alive"])
                except Exception as e:  # noqa: BLE001
                    assert _kind(e) in op["error"].split("|"), (_kind(e), op["error"])
                else:
                    raise AssertionError(f"expected {op['error']}")
            else:
                check_events(w.set_state(op["positions"], op["gems"], op["alive"]), op, derived)
        elif kind == "source":
            w.set_source(op["laser_id"], enabled=op.get("enabled"), colour=op.get("colour"))
        elif kind == "set_exits":
            # World.exit_pos = [...] (pyworld.rs:203-209 -> world.rs:195-234); "Panic": a cell where the reference panics
            try:
                w.set_exits([tuple(p) for p in op["exits"]])
            except Exception as e:  # noqa: BLE001
                assert "error" in op and _kind(e) in op["error"].split("|"), (_kind(e), op.get("error"))
            else:
                assert "error" not in op, f"expected {op.get('error')}"
        elif kind == "collect_gem":
            # Gem.collect() (pygem.rs:52-66): no event, the gem simply counts as collected from now on
            try:
                w.collect_gem(*op["pos"])
            except Exception as e:  # noqa: BLE001
                assert "error" in op and _kind(e) in op["error"].split("|"), (_kind(e), op.get("error"))
            else:
                assert "error" not in op, f"expected {op.get('error')}"
        elif kind == "clone_check":
            # World::clone (world.rs:645-652) = a new world from the config + set_state(get_state()); deepcopy is clone
            c = w.clone() if hasattr(w, "clone") else _generic_clone(make, case, w)
            assert _pairs(c.positions()) == _pairs(w.positions()) and c.n_gems_collected() == w.n_gems_collected()
            assert (c.width, c.height) == (w.width, w.height)
            assert [(s[0], s[1], s[2], s[3]) for s in c.sources()] == [(s[0], s[1], s[2], s[3]) for s in w.sources()]
            if derived:
                assert (c.alive(), c.gems_collected(), c.arrived()) == (w.alive(), w.gems_collected(), w.arrived())
                before = (w.positions(), w.alive(), w.gems_collected())
                c.step([a[-1] for a in c.available_actions()])  # the clone moves on; the original must not
                assert (w.positions(), w.alive(), w.gems_collected()) == before
        elif kind == "set_agent_position":
            try:
                _set_agent_position(w, op["agent"], tuple(op["pos"]))
            except Exception as e:  # noqa: BLE001
                assert "error" in op and _kind(e) in op["error"].split("|"), (_kind(e), op.get("error"))
            else:
                assert "error" not in op, f"expected {op.get('error')}"
        elif kind == "colour":
            try:
                (w.set_colour_checked if hasattr(w, "set_colour_checked") else
                 lambda lid, c: _binding_set_colour(w, lid, c))(op["laser_id"], op["colour"])
            except Exception as e:  # noqa: BLE001
                assert "error" in op and _kind(e) in op["error"].split("|"), (_kind(e), op.get("error"))
            else:
                assert "error" not in op, f"expected {op.get('error')}"
        elif kind == "expect":
            check_expect(w, op, derived)
        else:
            raise KeyError(kind)
    if hasattr(w, "panics"):
        n, msg = w.panics()
        assert n == 0, f"reference panic site reached: {msg}"
    return w
