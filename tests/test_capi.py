"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/lle_hip.h declares;
the host-only map functions agree with the oracle's parser; the facade's value types behave like the reference's."""
import os
import re

import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from lle_amd import _capi

    header = open(os.path.join(ROOT, "include", "lle_hip.h")).read()
    declared = set(re.findall(r"\b(lle_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations found"
    L = _capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"liblle_hip.so lacks {missing}"
    assert declared == set(_capi.EXPORTS)
    assert L.lle_abi_version() == 3


def test_no_device_fails_loudly():
    import torch

    from lle_amd import BatchedWorld, World

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        BatchedWorld(LEVELS[1], 4)
    w = World("S0 . X")  # parsing needs no GPU ...
    assert w.exit_pos == [(0, 2)]
    with pytest.raises(RuntimeError, match="no HIP device"):  # ... the dynamics do, and say so
        w.reset()


@pytest.mark.parametrize("name", [f"level{k}" for k in range(1, 7)] + list(EXTRA_MAPS))
def test_map_compiler_matches_oracle_parser(oracle_mod, name):
    from lle_amd import _capi

    text = LEVELS[int(name[5:])] if name.startswith("level") else EXTRA_MAPS[name]
    m = _capi.Map(text)
    o = oracle_mod.OracleWorld(text)
    assert (m.height, m.width, m.n_agents, m.n_gems, m.n_sources) == (o.height, o.width, o.n_agents, o.n_gems, o.n_sources)
    assert m.positions(_capi.LLE_POS_START) == o.start_pos
    assert m.positions(_capi.LLE_POS_EXIT) == o.exit_pos
    assert m.positions(_capi.LLE_POS_WALL) == o.wall_pos
    assert m.positions(_capi.LLE_POS_VOID) == o.void_pos
    assert m.positions(_capi.LLE_POS_GEM) == o.gem_pos
    assert [(s.i, s.j, s.direction, s.agent_id, s.enabled, s.length) for s in m.sources()] == o.sources()
    assert sorted((t.i, t.j, t.laser_id) for t in m.laser_tiles()) == sorted((l[0], l[1], l[2]) for l in o.lasers())
    assert m.n_layers == 2 * o.n_agents + 4 and m.obs_bytes == m.n_layers * o.height * o.width


def test_level_maps_are_12x13():
    from lle_amd import _capi

    for lvl in range(1, 7):
        m = _capi.Map(level=lvl)
        assert (m.height, m.width) == (12, 13)
    m6 = _capi.Map(level=6)
    assert (m6.n_agents, m6.n_gems, m6.n_sources, m6.obs_bytes) == (4, 4, 3, 1872)
    assert sorted(s.length for s in m6.sources()) == [2, 6, 12]
    with pytest.raises(_capi.MapParseError) as e:
        _capi.Map(level=7)
    assert e.value.kind == "InvalidLevel"


def test_world_string_roundtrip():
    from lle_amd import World

    # tests/world_integration_tests.rs:467-479
    w = World("S0  L0S  X ")
    assert w.world_string == "S0  L0S  X "
    w._map.set_source(0, agent_id=1)
    assert w.world_string == "S0  L1S  X "


def test_parse_errors_are_parsing_errors():
    from lle_amd import InvalidLevelError, ParsingError, World

    for text in ("", "X G", "S0 S0 X X", "S1 S0 X", "X S0 .\n . ."):
        with pytest.raises(ParsingError):
            World(text)
    with pytest.raises(InvalidLevelError):
        World.level(9)
    with pytest.raises(FileNotFoundError):
        World.from_file("/nonexistent/level")
    assert World.from_file("lvl3").n_agents == 2 and World.from_file("level6").n_agents == 4


def test_action_value_type():
    from lle_amd import Action

    # python/tests/test_actions.py:58-85 (binding's (dx,dy) convention) and src/action.rs:18-26
    assert [a.value for a in Action.variants()] == [0, 1, 2, 3, 4] and Action.cardinality() == 5
    assert Action.NORTH.delta == (-1, 0) and Action.SOUTH.delta == (1, 0) and Action.EAST.delta == (0, 1)
    assert Action.WEST.delta == (0, -1) and Action.STAY.delta == (0, 0)
    assert Action.from_delta(-1, 0) == Action.WEST and Action.from_delta(0, -1) == Action.NORTH
    assert Action.from_delta(1, 0) == Action.EAST and Action.from_delta(0, 1) == Action.SOUTH
    assert Action.NORTH.opposite() == Action.SOUTH and Action.STAY.opposite() == Action.STAY
    assert Action(2) == Action.EAST
    with pytest.raises(ValueError):
        Action(5)
    with pytest.raises(ValueError):
        Action.from_delta(1, 1)


def test_world_state_value_type():
    from lle_amd import WorldState

    # python/tests/test_world.py:651-664
    s = WorldState([(0, 0)], [False])
    assert list(s.as_array()) == [0.0, 0.0, 0.0, 1.0]
    assert WorldState.from_array([0.0, 0.0, 0.0, 1.0], 1, 1) == s
    s = WorldState([(25, 17), (10, 30)], [True, False], agents_alive=[True, False])
    expected = [25.0, 17.0, 10.0, 30.0, 1.0, 0.0, 1.0, 0.0]
    assert list(s.as_array()) == expected and s.as_array().dtype == np.float32
    assert WorldState.from_array(expected, 2, 2) == s
    # python/tests/test_world.py:395-407
    assert WorldState([(0, 0)], [False], [True]) != WorldState([(0, 0)], [False], [False])
    assert hash(WorldState([(0, 0)], [False])) == hash(WorldState([(0, 0)], [False], [True]))
    assert all(WorldState([(0, 0)], [False]).agents_alive)


def test_row_alignment_is_validated_and_must_agree_inside_a_batch():
    """lle_map_set_row_align: pitch of an observation row; host-side only, so checked without a GPU."""
    import ctypes as C

    from lle_amd import _capi, mapgen

    m = _capi.Map(LEVELS[6], row_align=16)
    assert (m.obs_bytes, m.obs_stride) == (1872, 1872)
    m.set_row_align(128)
    assert (m.obs_bytes, m.obs_stride) == (1872, 1920)
    # automatic (the default): whole 128-byte lines when that pads the row by at most 1/32 of its size
    assert _capi.Map(LEVELS[6]).obs_stride == 1920 and _capi.Map(LEVELS[1]).obs_stride == 944
    m.set_row_align(0)
    assert m.obs_stride == 1920
    assert _capi.Map(LEVELS[1], row_align=256).obs_stride == 1024 and _capi.Map(LEVELS[1], row_align=16).obs_stride == 944
    with pytest.raises(ValueError):
        _capi.Map(LEVELS[6], row_align=48)
    assert m.positions(_capi.LLE_POS_WALL) == _capi.Map(LEVELS[6]).positions(_capi.LLE_POS_WALL)
    shape = dict(height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2)
    a, b = _capi.Map(mapgen.generate(seed=1, **shape), row_align=128), _capi.Map(mapgen.generate(seed=2, **shape), row_align=16)
    L = _capi.lib()
    assert L.lle_batch_arena_bytes_multi((C.c_void_p * 2)(a.h, b.h), 2, 64) < 0
    assert b"row alignment" in L.lle_last_error()
    b.set_row_align(128)
    assert L.lle_batch_arena_bytes_multi((C.c_void_p * 2)(a.h, b.h), 2, 64) > 0


def test_world_state_hash_eq_and_pickle():
    """python/tests/test_world.py:393-407 (hash / eq of hand-built states, dead flag included) and
    python/tests/test_serialization.py:9-16 (50 random WorldStates survive pickle)."""
    import pickle
    import random

    from lle_amd import WorldState

    s1, s2 = WorldState([(0, 0)], [False], [True]), WorldState([(0, 0)], [False], [False])
    assert hash(s1) != hash(s2) and s1 != s2                      # test_world_state_hash_eq_dead
    s1, s2 = WorldState([(0, 0)], [False]), WorldState([(0, 1)], [False])
    assert hash(s1) != hash(s2) and s1 != s2                      # test_world_state_hash_neq
    rng = random.Random(0)
    for _ in range(50):                                           # test_pickle_world_state
        s = WorldState(gems_collected=[rng.choice([True, False]) for _ in range(rng.randint(0, 10))],
                       agents_positions=[(rng.randint(0, 50), rng.randint(0, 90)) for _ in range(rng.randint(0, 10))])
        assert pickle.loads(pickle.dumps(s)) == s


@pytest.mark.parametrize("name", sorted(dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)))
def test_row_head_bytes_never_change(oracle_mod, name):
    """lle_map_row_head: whole 128-byte lines behind the agent layers whose bytes are the same in every environment after
    every step (the step kernel stores them before its state machine).  Checked against oracle rollouts with deaths,
    collected gems and beams switching, with and without resets."""
    from lle_amd import Map

    text = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)[name]
    m = Map(text, row_align=128)
    with pytest.raises(ValueError):
        m.set_head_lines(9)
    m.set_head_lines(8)
    first, nbytes = m.row_head
    assert first % 128 == 0 and nbytes % 128 == 0 and 0 <= nbytes <= 1024 and first + nbytes <= m.obs_stride
    if m.obs_supported:
        assert nbytes == 0 or first + 127 >= m.n_agents * m.height * m.width
    lo, hi = min(first, m.obs_bytes), min(first + nbytes, m.obs_bytes)
    n = 256
    ob = oracle_mod.OracleBatch(text, n)
    ref = None
    for t in range(60):
        rows = ob.step(None, auto_reset=(t // 20) % 2 == 0, seed=21, t=t)["obs"].reshape(n, -1)
        ref = rows[0, lo:hi].copy() if ref is None else ref
        assert (rows[:, lo:hi] == ref).all(), (name, t)


@pytest.mark.parametrize("name", sorted(dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)))
def test_static_lines_of_a_row_never_change(oracle_mod, name):
    """lle_map_row_dynamic_lines: the 128-byte lines that LLE_STEP_INCREMENTAL_OBS does NOT write hold the same bytes in every
    environment after every step of an oracle rollout (deaths, collected gems, beams switching, resets) -- the bytes of the
    reset observation; the row head is a run of them; rows that are not whole lines are all dynamic."""
    from lle_amd import Map

    text = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)[name]
    m = Map(text, row_align=128)
    lines = m.row_dynamic_lines
    assert len(lines) == m.obs_stride // 128 and sum(lines) * 128 == m.dyn_row_bytes
    first, nbytes = m.row_head
    assert not any(lines[first // 128: (first + nbytes) // 128])
    static = np.zeros(m.obs_stride, bool)
    for l, dynamic in enumerate(lines):
        static[l * 128:(l + 1) * 128] = not dynamic
    static = static[: m.obs_bytes]
    n = 256
    ob = oracle_mod.OracleBatch(text, n)
    ref = np.stack([ob.world(e).obs() for e in range(4)]).reshape(4, -1)[:, static]
    assert (ref == ref[0]).all()
    for t in range(60):
        rows = ob.step(None, auto_reset=(t // 20) % 2 == 0, seed=23, t=t)["obs"].reshape(n, -1)
        assert (rows[:, static] == ref[0]).all(), (name, t)
    packed = Map(text, row_align=16)
    assert packed.obs_stride % 128 == 0 or (all(packed.row_dynamic_lines) and packed.dyn_row_bytes == packed.obs_stride)


@pytest.mark.parametrize("name", sorted(dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)))
def test_reset_beam_table_matches_oracle_reset(oracle_mod, name):
    """lle_map_reset_beam(s, c) -- what LLE_STEP_RECOLOUR_RESETS stores as an env's reset beams -- against the oracle:
    colour c on source s, World.reset, read the beam (for every pair the binding's set_colour accepts)."""
    from lle_amd import Map

    text = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)[name]
    m = Map(text)
    for s in m.sources():
        for c in range(m.n_agents):
            with pytest.raises(ValueError):
                m.reset_beam(len(m.sources()), c)
            if not m.colour_allowed(s.laser_id, c):
                continue
            w = oracle_mod.OracleWorld(text)
            w.set_source(s.laser_id, colour=c)
            w.reset()
            bits = w.beam_bits(s.laser_id)
            want = sum(int(b) << k for k, b in enumerate(bits))
            assert m.reset_beam(s.laser_id, c) == want, (name, s.laser_id, c)


def test_map_set_exits_and_clone(oracle_mod, tmp_path):
    """lle_map_set_exits (World::set_exit_positions, world.rs:195-234) on the host object: the exit list, the world string and
    the row head's bytes follow; the refusals leave the map untouched; lle_map_clone is independent.  No GPU involved."""
    from lle_amd import World, _capi
    from lle_amd.world import ParsingError

    m = _capi.Map(LEVELS[6])
    o = oracle_mod.OracleWorld(LEVELS[6])
    old = m.positions(_capi.LLE_POS_EXIT)
    new = [(0, 2), (4, 3), (11, 0), (5, 5)]  # (4, 3) lies under the beam of L0E
    copy = m.clone()
    m.set_exits(new)
    o.set_exits(new)
    assert m.positions(_capi.LLE_POS_EXIT) == new == o.exit_pos and m.n_exits == 4
    assert copy.positions(_capi.LLE_POS_EXIT) == old and copy.world_string() != m.world_string()
    rows = [r.split() for r in m.world_string().split("\n")]
    assert all(rows[i][j] == "X" for i, j in new) and all(rows[i][j] == "." for i, j in old)
    assert _capi.Map(m.world_string()).positions(_capi.LLE_POS_EXIT) == sorted(new)  # (a re-parse lists them in row-major order)
    with pytest.raises(_capi.MapParseError) as e:
        m.set_exits(new[:3])
    assert e.value.kind == "NotEnoughExitTiles"
    for bad in ([(3, 0)] + new[1:], [(0, 0)] + new[1:], [(4, 0)] + new[1:], [(12, 0)] + new[1:], [(0, 2), (0, 2)] + new[2:]):
        with pytest.raises(ValueError):  # wall, gem, source, out of the world, the same floor twice
            m.set_exits(bad)
        assert m.positions(_capi.LLE_POS_EXIT) == new
    # the facade without a device: the property, the error classes, save
    w = World.level(6)
    w.exit_pos = new
    assert w.exit_pos == new and w.world_string == m.world_string()
    with pytest.raises(ParsingError) as pe:
        w.exit_pos = new[:2]
    assert pe.value.kind == "NotEnoughExitTiles" and w.exit_pos == new
    w.save(str(tmp_path / "lvl.txt"))
    assert World.from_file(str(tmp_path / "lvl.txt")).exit_pos == sorted(new)


def test_limits_are_refused_loudly(oracle_mod):
    """include/lle_hip.h LLE_MAX_*: a map beyond a static limit is refused with LLE_PARSE_LIMIT -- never
    clamped, never a silent wrong answer; the same map at the limit is accepted and matches the oracle's parser."""
    from lle_amd import World, _capi
    from lle_amd.world import ParsingError

    def corridor(n):  # a source followed by n free cells: a beam of n cells (agent 0's own colour, so its start may lie on it)
        return "L0E " + ". " * (n - 2) + "S0 X"
    ok = _capi.Map(corridor(32))
    assert ok.max_beam_len == 32 and ok.width == 33 and ok.n_beam_words == ok.n_sources == 1
    o = oracle_mod.OracleWorld(corridor(32))
    assert [s.length for s in ok.sources()] == [s[5] for s in o.sources()] == [32]
    # round 4: a beam is a CHAIN of 32-cell words (tables.h): any beam a map of at most 255 x 255 can hold is accepted ...
    for n in (33, 40, 64, 65, 253):
        m, o = _capi.Map(corridor(n)), oracle_mod.OracleWorld(corridor(n))
        assert [s.length for s in m.sources()] == [s[5] for s in o.sources()] == [n] and m.max_beam_len == n
        assert m.n_sources == 1 and m.n_beam_words == max(5, -(-n // 32)) and m.source_first_words() == [0]  # (padded to the record form's 5 words)
        tiles = m.laser_tiles()
        assert [(t.offset, t.word, t.bit) for t in tiles] == [(k, k // 32, k % 32) for k in range(n)]
    assert World(corridor(40)).n_agents == 1
    # ... and what is refused is the TOTAL number of words: 32 over all beams of a map
    rows_of_40 = lambda k: "\n".join("L0E " + ". " * 40 for _ in range(k)) + "\nS0 X" + " ." * 39  # k beams of 40 cells = 2 words each
    assert _capi.Map(rows_of_40(16)).n_beam_words == 32
    with pytest.raises(_capi.MapParseError) as e:
        _capi.Map(rows_of_40(17))
    assert e.value.kind == "Limit"
    with pytest.raises(ParsingError, match="static limit"):
        World(rows_of_40(17))
    with pytest.raises(_capi.MapParseError) as e:  # 17 agents
        _capi.Map(" ".join(f"S{k}" for k in range(17)) + "\n" + " ".join("X" for _ in range(17)))
    assert e.value.kind == "Limit"
    with pytest.raises(_capi.MapParseError) as e:  # 33 gems
        _capi.Map("S0 X " + "G " * 33)
    assert e.value.kind == "Limit"
    with pytest.raises(_capi.MapParseError) as e:  # a side of 256
        _capi.Map("S0 X " + ". " * 254)
    assert e.value.kind == "Limit"


@pytest.mark.parametrize("name", ["level3", "level5", "level6", "nested", "exit_under_beam", "three_beams", "many_agents", "colour_alias"])
def test_row_head_under_per_env_colours_never_changes(oracle_mod, name):
    """lle_map_row_head_env_sources: lines that no agent, gem or laser of ANY colour below n_agents can change -- what the
    per-env-sources step kernel (MODE 8) stores ahead of its state machine.  Oracle worlds re-coloured at random (legal
    colours only) and switched on / off along rollouts: the head's bytes are the same in every env at every step, and equal
    the bare template's.  A map whose own sources have a colour >= n_agents has no such head."""
    from lle_amd import Map
    from tests.parity_util import legal_colours

    text = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)[name]
    m = Map(text, row_align=128)
    m.set_head_lines(8)
    first, nbytes = m.row_head_env_sources
    A, L = m.n_agents, m.n_sources
    if any(s.agent_id >= A for s in m.sources()):
        assert nbytes == 0
        return
    assert first % 128 == 0 and nbytes % 128 == 0 and first + nbytes <= m.obs_stride
    assert nbytes == 0 or first + 127 >= 2 * A * m.height * m.width  # behind the agent and laser layers
    lo, hi = min(first, m.obs_bytes), min(first + nbytes, m.obs_bytes)
    # the second run of such lines (round 4): behind the first, disjoint from it, together at most the 8 lines asked for
    first2, nbytes2 = m.row_head_env_sources_second
    assert first2 % 128 == 0 and nbytes2 % 128 == 0 and first2 + nbytes2 <= m.obs_stride and nbytes + nbytes2 <= 8 * 128
    assert nbytes2 == 0 or (nbytes != 0 and first2 >= first + nbytes)
    lo2, hi2 = min(first2, m.obs_bytes), min(first2 + nbytes2, m.obs_bytes)
    n = 96
    ob = oracle_mod.OracleBatch(text, n)
    rng = np.random.default_rng(4)
    ref = None
    for t in range(45):
        if t % 9 == 0 and L:
            colours = legal_colours(m, rng.integers(0, A, size=(n, L), dtype=np.uint8))
            for e in range(n):
                w = ob.world(e)
                for l in range(L):
                    w.set_source(l, colour=int(colours[e, l]), enabled=bool(rng.integers(0, 2)))
        rows = ob.step(None, auto_reset=(t // 15) % 2 == 0, seed=8, t=t)["obs"].reshape(n, -1)
        both = np.concatenate([rows[:, lo:hi], rows[:, lo2:hi2]], axis=1)
        ref = both[0].copy() if ref is None else ref
        assert (both == ref).all(), (name, t)


def test_second_head_run_of_level6():
    """Level 6 under per-environment colours: lines 10-11 (WALL / VOID planes) and line 14 (the end of the EXIT plane)."""
    from lle_amd import Map

    m = Map(LEVELS[6])
    assert m.row_head_env_sources == (1280, 256) and m.row_head_env_sources_second == (1792, 128)
    m.set_head_lines(2)
    assert m.row_head_env_sources == (1280, 256) and m.row_head_env_sources_second == (0, 0)
    m.set_head_lines(0)
    assert m.row_head_env_sources[1] == 0 and m.row_head_env_sources_second[1] == 0


def test_laser_tokens():
    """src/unit_tests/test_laser_config.rs:6-26: `L<agent><direction>` tokens -- the colour digit, the four letters, ids in parse order."""
    from lle_amd import Map

    m = Map("L0E .   .  S0\n"
            ".   .   .  L1W\n"
            ".   .   X  .\n"
            ".   L2N .  .\n"
            "L3S .   .  .\n"
            ".   .   .  .")
    srcs = m.sources()
    assert [(s.laser_id, s.agent_id, s.direction) for s in srcs] == [(0, 0, 1), (1, 1, 3), (2, 2, 0), (3, 3, 2)]   # N=0 E=1 S=2 W=3
    assert [(s.i, s.j) for s in srcs] == [(0, 0), (1, 3), (3, 1), (4, 0)]


def test_every_compiled_kernel_is_reachable_and_every_reachable_one_is_compiled():
    """lle_debug_reachable walks the launchers' own dispatch (launches suppressed) over every agent count, beam-word count, crossing
    or not, and mode; tools/compiled_kernels.py reads the kernel descriptors out of the library's gfx950 code objects.  The two lists
    must be the same: a compiled kernel that no launch can reach is a kernel no test can check (round 4 shipped 50 of them:
    step_kernel<G,4,MODE,true,-1>), and tests/test_gpu_instantiations.py then launches every one of them against the oracle."""
    import importlib.util

    from lle_amd import _capi
    spec = importlib.util.spec_from_file_location("compiled_kernels", os.path.join(ROOT, "tools", "compiled_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    compiled, reachable = set(mod.compiled_kernels()), set(_capi.reachable_kernels())
    assert compiled == reachable, (sorted(compiled - reachable)[:8], sorted(reachable - compiled)[:8])
    assert len(compiled) == 555 and sum(n.startswith("step_kernel<") for n in compiled) == 520
    assert _capi.launched_kernels() == [] or set(_capi.launched_kernels()) <= reachable  # (nothing launches without a GPU)
    from tests import instantiation_maps as im
    # the coverage test's cases name exactly the step-kernel instantiations
    names = {im.kernel_name(A, L, cross, mode) for (A, _g) in im.AGENT_CLASSES for (L, cross) in im.SOURCE_CLASSES for mode in range(10)
             if mode < 6 or im.lm_of(L) <= 8}
    assert names == {n for n in reachable if n.startswith("step_kernel<")}
