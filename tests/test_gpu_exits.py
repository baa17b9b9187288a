"""World.exit_pos = [...] on live batches (World::set_exit_positions, src/core/world.rs:195-234; setter pyworld.rs:203-209):
lle_map_set_exits + lle_batch_update_map against the oracle, where every env is its own world object and takes the same
set_exit_positions call.  Bit-exact on state, ordered events, availability and the int8 observation, before and after
the change, with auto-reset (the reset state follows the new exits), per-environment sources and several maps."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, legal_colours, unpack_engine

pytestmark = pytest.mark.gpu

MAPS = {"level6": LEVELS[6], "level3": LEVELS[3], "level1": LEVELS[1], "nested": EXTRA_MAPS["nested"],
        "exit_under_beam": EXTRA_MAPS["exit_under_beam"], "four_layers": EXTRA_MAPS["four_layers"], "corridor": EXTRA_MAPS["corridor"],
        "many_agents": EXTRA_MAPS["many_agents"], "config5_32x32": EXTRA_MAPS["config5_32x32"], "gen_20_lasers": EXTRA_MAPS["gen_20_lasers"]}


def dims_of(ob):
    return (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)


def oracle_obs(ob, n):
    return np.stack([ob.world(e).obs() for e in range(n)])


def check(bw, ob, ostep, where, n_obs=0):
    eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
    if ostep is not None:
        assert_step_equal(eng, ostep, where)
    assert_state_equal(eng, ob.dump(), where)
    if n_obs:  # no step to take the observation from: ask the oracle worlds one by one
        assert np.array_equal(eng["obs"][:n_obs], oracle_obs(ob, n_obs)), f"{where}: observation"


def legal_exits(m, rng, n_exits):
    """n_exits distinct cells World::set_exit_positions accepts: floor tiles (starts, current exits and floors under a beam
    included), i.e. everything but walls (sources among them), voids and gems."""
    from lle_amd import _capi
    taken = set(m.positions(_capi.LLE_POS_WALL)) | set(m.positions(_capi.LLE_POS_VOID)) | set(m.positions(_capi.LLE_POS_GEM))
    free = [(i, j) for i in range(m.height) for j in range(m.width) if (i, j) not in taken]
    lasers = [p for p in {(t.i, t.j) for t in m.laser_tiles()} if p not in taken]
    pick = [free[k] for k in rng.choice(len(free), size=min(n_exits, len(free)), replace=False)]
    if lasers and lasers[0] not in pick:
        pick[0] = lasers[0]  # always one exit under a beam (Laser::set_tile, laser.rs:109-115)
    return pick


@pytest.mark.parametrize("name", list(MAPS))
def test_exits_change_mid_rollout(oracle_mod, name):
    from lle_amd import BatchedWorld

    text = MAPS[name]
    n = 200 if "config5" in name else 500
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    rng = np.random.default_rng(17)
    t = 0
    for round_ in range(3):
        auto = round_ != 1  # run-to-the-end once: corpses on former exits, arrived agents on floors
        for _ in range(12):
            bw.step(sample=True, auto_reset=auto, seed=77, t=t, env_offset=9)
            check(bw, ob, ob.step(None, auto_reset=auto, seed=77, t=t, env_offset=9), f"{name} round {round_} t={t}")
            t += 1
        exits = legal_exits(bw.map, rng, ob.A + round_)
        bw.set_exits(exits)
        for e in range(n):
            ob.world(e).set_exits(exits)
        assert bw.map.positions(1) == exits
        check(bw, ob, None, f"{name} after set_exits {round_}", n_obs=min(n, 64))  # nothing dynamic moved; the EXIT plane did
    for _ in range(12):
        bw.step(sample=True, auto_reset=True, seed=77, t=t, env_offset=9)
        check(bw, ob, ob.step(None, auto_reset=True, seed=77, t=t, env_offset=9), f"{name} final t={t}")
        t += 1


def test_reset_state_follows_the_exits(oracle_mod):
    """An agent whose start becomes an exit arrives at reset (world.rs:411-432): the InitRecord the auto-reset copies is recomputed."""
    from lle_amd import BatchedWorld

    text = "S0 . S1 .\n . G . X\n X . . ."
    n = 128
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    exits = [(0, 0), (2, 3)]
    bw.set_exits(exits)
    for e in range(n):
        ob.world(e).set_exits(exits)
    check(bw, ob, None, "after set_exits", n_obs=n)
    assert not bw.agents_arrived().any()
    bw.reset()
    ob.reset()
    check(bw, ob, None, "after reset", n_obs=n)
    assert bool(bw.agents_arrived()[:, 0].all()) and not bool(bw.agents_arrived()[:, 1].any())
    for t in range(40):
        bw.step(sample=True, auto_reset=True, seed=5, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=5, t=t), f"t={t}")


@pytest.mark.parametrize("name", ["level6", "nested", "many_agents"])
def test_exits_with_per_env_sources(oracle_mod, name):
    """Every env keeps ITS colours / flags; its own reset record (what auto-reset copies) is recomputed under the new exits."""
    import torch

    from lle_amd import BatchedWorld
    from tests.test_gpu_env_sources import Mirror

    text = MAPS[name]
    n = 256
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    A, L = ob.A, bw.map.n_sources
    rng = np.random.default_rng(3)
    mirror = Mirror(ob, n, L)
    colours = legal_colours(bw.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    enabled = rng.integers(0, 1 << L, size=n, dtype=np.int64).astype(np.int32)
    bw.set_sources(torch.from_numpy(colours), torch.from_numpy(enabled))
    mirror.apply(colours, enabled)
    t = 0
    for _ in range(10):
        bw.step(sample=True, auto_reset=True, seed=11, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=11, t=t), f"{name} before t={t}")
        t += 1
    exits = legal_exits(bw.map, rng, A + 1)
    bw.set_exits(exits)
    for e in range(n):
        ob.world(e).set_exits(exits)
    check(bw, ob, None, f"{name} after set_exits", n_obs=64)
    assert np.array_equal(bw.src_colour.cpu().numpy(), colours) and np.array_equal(bw.src_enabled.cpu().numpy(), enabled)
    for _ in range(25):
        bw.step(sample=True, auto_reset=True, seed=11, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=11, t=t), f"{name} after t={t}")
        t += 1
    # the map's own sources must be unchanged for update_map on such a batch; broadcasting them is update_sources' job
    if A > 1:
        nxt = next((c for c in range(A) if c != bw.map.sources()[0].agent_id and bw.map.colour_allowed(0, c)), None)
        if nxt is not None:
            bw.map.set_source(0, agent_id=nxt)
            with pytest.raises(RuntimeError, match="source colours"):
                bw.update_map()


@pytest.mark.parametrize("case", ["small_x128", "config5_x16", "config5_x8"])
def test_exits_of_one_map_of_several(oracle_mod, case):
    """(config 5's shape in blocks of 16 and 8: the split-row kernels, whose workgroups build their rows from the bit form of the static observation and --
    at 16 per map -- their tables from the packed image: both are recompiled with the exits and travel with the blob.)"""
    from lle_amd import BatchedWorld, mapgen

    if case == "small_x128":
        shape = dict(height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2)
        texts = [mapgen.generate(seed=200 + s, **shape) for s in range(3)]
        per = 128
    else:
        texts = [mapgen.config5(300 + s) for s in range(3)]
        per = 16 if case == "config5_x16" else 8
    bw = BatchedWorld(texts, per * len(texts))
    obs = [oracle_mod.OracleBatch(t, per) for t in texts]
    rng = np.random.default_rng(5)

    def check_blocks(osteps, where):
        bufs = bw.host_buffers()
        for m, ob in enumerate(obs):
            eng = unpack_engine({k: v[m * per:(m + 1) * per] for k, v in bufs.items()}, *dims_of(ob))
            if osteps is not None:
                assert_step_equal(eng, osteps[m], f"{where} map {m}")
            else:
                assert np.array_equal(eng["obs"][:min(32, per)], oracle_obs(ob, min(32, per))), f"{where} map {m}: observation"
            assert_state_equal(eng, ob.dump(), f"{where} map {m}")

    t = 0
    for which in (1, 2, 1):
        for _ in range(8):
            bw.step(sample=True, auto_reset=True, seed=2, t=t, env_offset=64)
            check_blocks([ob.step(None, auto_reset=True, seed=2, t=t, env_offset=64 + m * per) for m, ob in enumerate(obs)], f"t={t}")
            t += 1
        exits = legal_exits(bw.maps[which], rng, bw.map.n_agents + 1)
        bw.set_exits(exits, map_index=which)
        for e in range(per):
            obs[which].world(e).set_exits(exits)
        check_blocks(None, f"after set_exits on map {which}")
    for _ in range(8):
        bw.step(sample=True, auto_reset=True, seed=2, t=t, env_offset=64)
        check_blocks([ob.step(None, auto_reset=True, seed=2, t=t, env_offset=64 + m * per) for m, ob in enumerate(obs)], f"t={t}")
        t += 1


def test_update_map_refuses_what_is_not_a_recompilation():
    """a map with another row pitch (lle_map_set_row_align after lle_batch_create) or other tiles is LLE_ERR_ARG,
    for lle_batch_update_sources too; BatchedWorld(row_align=...) leaves the caller's Map alone."""
    import torch

    from lle_amd import BatchedWorld, _capi

    m = _capi.Map(LEVELS[6])
    assert m.obs_stride == 1920
    bw = BatchedWorld(m, 64)
    m.set_row_align(16)  # 1 872-byte rows: same blob capacity class, another pitch
    assert m.obs_stride == 1872
    for call in (bw.update_sources, bw.update_map):
        with pytest.raises(RuntimeError, match="row alignment"):
            call()
    m.set_row_align(0)
    bw.update_sources()
    small = BatchedWorld("S0 . @ X\nL0E . . .", 64)
    small.maps[0] = small.map = _capi.Map("S0 @ . X\nL0E . . .")  # same shape and counts, a wall moved
    with pytest.raises(RuntimeError, match="does not match"):
        small.update_map()
    # a plain void that became floor (or the reverse) is another map -- only under a beam may a void change
    # (Laser::set_tile, laser.rs:109-115), and the batch is left as it was
    voids = BatchedWorld("S0 . V X\nL0E . . .", 64)
    before = voids.obs.clone()
    voids.maps[0] = voids.map = _capi.Map("S0 . . X\nL0E . . .")
    with pytest.raises(RuntimeError, match="does not match"):
        voids.update_map()
    assert torch.equal(voids.obs, before)
    under = BatchedWorld("S0 . . X\nL0E . V .", 64)  # the void sits under the beam: it may become an exit
    under.set_exits([(1, 2)])
    assert under.map.positions(_capi.LLE_POS_EXIT) == [(1, 2)]
    # row_align on a batch is applied to a COPY of the caller's map
    mine = _capi.Map(LEVELS[6])
    bw2 = BatchedWorld(mine, 64, row_align=16)
    assert mine.obs_stride == 1920 and bw2.map.obs_stride == 1872 and bw2.map is not mine


def test_world_facade_exit_pos_save_copy_pickle(tmp_path):
    """The drop-in surface: `World.exit_pos` is a property whose setter acts on the device (it used to be a plain attribute:
    assignment changed nothing), `World.save` (pyworld.rs:183-193), and copies / pickles carry the new exits (world.rs:98-110)."""
    import copy
    import pickle

    from lle_amd import Action, World
    from lle_amd.world import ParsingError

    w = World("S0 . X")
    w.reset()
    assert w.exit_pos == [(0, 2)]
    w.exit_pos = [(0, 1)]
    assert w.exit_pos == [(0, 1)] and w.world_string.split() == ["S0", "X", "."]
    assert [e.event_type.name for e in w.step(Action.EAST)] == ["AGENT_EXIT"]
    for clone in (copy.deepcopy(w), pickle.loads(pickle.dumps(w))):
        assert clone.exit_pos == [(0, 1)] and clone.agents_positions == [(0, 1)]
    with pytest.raises(ParsingError, match="Not enough exit tiles"):
        w.exit_pos = []
    with pytest.raises(ValueError):
        w.exit_pos = [(0, 7)]
    with pytest.raises(OverflowError):
        w.exit_pos = [(0, -1)]
    assert w.exit_pos == [(0, 1)]
    path = tmp_path / "saved.txt"
    w.save(str(path))
    assert World.from_file(str(path)).exit_pos == [(0, 1)]
    with pytest.raises(ValueError, match="Could not write to file"):
        w.save(str(tmp_path / "no_such_dir" / "x.txt"))
    # a world that was never stepped (no device batch yet) takes the exits into its first batch
    v = World("S0 . X")
    v.exit_pos = [(0, 1)]
    v.reset()
    assert [e.event_type.name for e in v.step([Action.EAST])] == ["AGENT_EXIT"]
    # the generator protocol of the reference test (python/tests/test_observations.py:76-92)
    from lle_amd.observations import Layered
    u = World("S0 X . .")
    g = Layered(u)
    u.exit_pos = [(0, 2), (0, 3)]
    u.reset()
    g.reset()
    o = g.observe()
    assert np.all(o[:, g.EXIT, 0, 2] == 1) and np.all(o[:, g.EXIT, 0, 3] == 1) and np.all(o[:, g.EXIT, 0, 1] == 0)


def test_restore_after_an_exit_change_recomputes_the_reset_records(oracle_mod):
    """A snapshot taken BEFORE World.exit_pos changes, restored AFTER: the dynamic state comes back as it was, but with
    per-environment sources the snapshot also carries every env's own reset record -- computed under the old exits.  restore
    recomputes them under the current tables (an agent whose start is an exit now arrives at reset)."""
    import torch

    from lle_amd import BatchedWorld

    text = "S0 . S1 .\nL0E . . X\n X . . ."
    n = 128
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    colours = torch.zeros((n, 1), dtype=torch.uint8)
    bw.set_sources(colours=colours)  # per-environment sources on (the map's own colour)
    for t in range(5):
        bw.step(sample=True, seed=3, t=t)
        ob.step(None, seed=3, t=t)
    snap = bw.snapshot()
    saved = [(ob.world(e).positions(), ob.world(e).gems_collected(), ob.world(e).alive(), ob.world(e).arrived()) for e in range(n)]
    exits = [(0, 0), (2, 3)]
    bw.set_exits(exits)
    for e in range(n):
        ob.world(e).set_exits(exits)
    for t in range(5, 9):
        bw.step(sample=True, auto_reset=True, seed=3, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=3, t=t), f"t={t}")
    bw.restore(snap)
    assert bw.map.positions(1) == exits  # the map is not part of a snapshot
    got = unpack_engine(bw.host_buffers(), *dims_of(ob))
    assert [tuple(map(tuple, got["pos"][e].tolist())) for e in range(n)] == [tuple(s[0]) for s in saved]
    # every env over -> auto-reset copies its own reset record: agent 0's start is an exit now
    bw.bits.zero_()  # everybody dead: the next auto-reset step restarts every env
    bw.step(sample=True, auto_reset=True, seed=3, t=20)
    assert bool((bw.evcount >> 7).all())
    ob.reset()
    ostep = ob.step(None, seed=3, t=20)
    eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
    eng["ev_count"] = eng["ev_count"] & 0x7F  # (bit 7: the env was auto-reset first; the oracle side was reset by hand)
    assert_step_equal(eng, ostep, "after restore + reset")
    assert_state_equal(eng, ob.dump(), "after restore + reset")
    assert bool(bw.agents_arrived()[:, 0].all())
