"""Per-environment laser sources (lle_batch_set_sources; SURVEY.md section 8(f) rank 4: LLE.reset with randomize_lasers,
python/lle/env/env.py:198-200, and LaserSource.enable / disable, pylaser_source.rs:55-75) against the oracle, where every
env is its own world object and takes the same set_colour / enable / disable calls."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, LONG_MAPS, assert_state_equal, assert_step_equal, legal_colours, unpack_engine

pytestmark = pytest.mark.gpu

MAPS = {"level6": LEVELS[6], "level5": LEVELS[5], "level1_no_lasers": LEVELS[1], "nested": EXTRA_MAPS["nested"], "three_beams": EXTRA_MAPS["three_beams"],
        "four_layers": EXTRA_MAPS["four_layers"], "q1": EXTRA_MAPS["q1"], "many_agents": EXTRA_MAPS["many_agents"],
        "gen_20_lasers": EXTRA_MAPS["gen_20_lasers"],
        # beams longer than 32 cells (chains of beam words): the caller still speaks of sources, colours [n][n_sources]
        "long_q1": LONG_MAPS["long_q1"], "long_crossing": LONG_MAPS["long_crossing"]}


def dims_of(ob):
    return ob.dims


def check(bw, ob, ostep, where):
    eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
    if ostep is not None:
        assert_step_equal(eng, ostep, where)
    assert_state_equal(eng, ob.dump(), where)


class Mirror:
    """Applies the same per-env source changes to the oracle worlds, with the Python binding's semantics: enable /
    disable only act when the flag changes (pylaser_source.rs:55-75)."""

    def __init__(self, ob, n, L):
        self.ob, self.n, self.L = ob, n, L
        srcs = ob.world(0).sources()
        self.enabled = np.array([[bool(s[4]) for s in srcs]] * n)

    def apply(self, colours=None, enabled=None, mask=None):
        for e in range(self.n):
            if mask is not None and not mask[e]:
                continue
            w = self.ob.world(e)
            for l in range(self.L):
                if colours is not None:
                    w.set_source(l, colour=int(colours[e, l]))
                if enabled is not None:
                    want = bool((int(enabled[e]) >> l) & 1)
                    if want != self.enabled[e, l]:
                        w.set_source(l, enabled=want)
                        self.enabled[e, l] = want


@pytest.mark.parametrize("name", list(MAPS))
def test_random_colours_and_flags_per_env(oracle_mod, name):
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n = 300
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    A, L = ob.A, bw.map.n_sources
    mirror = Mirror(ob, n, L)
    rng = np.random.default_rng(7)
    t = 0
    for episode in range(4):
        # a few steps, then re-colour / switch sources of a random subset of envs mid-episode (beams keep their state)
        for _ in range(6):
            auto = episode % 2 == 1
            bw.step(sample=True, auto_reset=auto, seed=21, t=t, env_offset=5)
            ostep = ob.step(None, auto_reset=auto, seed=21, t=t, env_offset=5)
            check(bw, ob, ostep, f"{name} episode {episode} t={t}")
            t += 1
        colours = legal_colours(bw.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
        enabled = rng.integers(0, 1 << L, size=n, dtype=np.int64).astype(np.int32) if episode != 2 else None
        mask = (rng.random(n) < 0.7).astype(np.uint8) if episode != 0 else None
        bw.set_sources(torch.from_numpy(colours), None if enabled is None else torch.from_numpy(enabled),
                       None if mask is None else torch.from_numpy(mask))
        mirror.apply(colours, enabled, mask)
        check(bw, ob, None, f"{name} after set_sources {episode}")
        assert int(bw.err.max()) == 0
        if episode == 1:  # LLE.reset: world.reset() with the current colours, for a subset
            rmask = (rng.random(n) < 0.5).astype(np.uint8)
            bw.reset(torch.from_numpy(rmask))
            for e in np.nonzero(rmask)[0]:
                ob.world(int(e)).reset()
            check(bw, ob, None, f"{name} after masked reset {episode}")
    got_c = bw.src_colour.cpu().numpy()
    got_e = bw.src_enabled.cpu().numpy()
    fw = ob.first_words  # (LLE_BUF_SRC_COLOUR / _ENABLED are per beam WORD; every word of a source carries its colour and flag)
    words_of = [range(fw[l], fw[l + 1] if l + 1 < L else max(fw[l] + 1, -(-ob.world(0).sources()[l][5] // 32) + fw[l])) for l in range(L)]
    for e in range(0, n, 37):
        srcs = ob.world(e).sources()
        for l in range(L):
            assert all(int(got_c[e][w]) == srcs[l][3] for w in words_of[l]), (e, l)
            assert all((int(got_e[e]) >> w) & 1 == int(srcs[l][4]) for w in words_of[l]), (e, l)


def test_invalid_colour_is_refused_per_env(oracle_mod):
    import torch

    from lle_amd import BatchedWorld, _capi

    text = LEVELS[6]
    n = 64
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    A, L = ob.A, bw.map.n_sources
    colours = np.ones((n, L), np.uint8)
    colours[5, 1] = A          # "Agent ID is greater than the number of agents"
    colours[9, 0] = 200
    bw.set_sources(torch.from_numpy(colours))
    err = bw.err.cpu().numpy()
    assert err[5] == _capi.LLE_ENV_INVALID_COLOUR and err[9] == _capi.LLE_ENV_INVALID_COLOUR and err.sum() == 2 * 0x43
    for e in range(n):
        if e not in (5, 9):
            for l in range(L):
                ob.world(e).set_source(l, colour=1)
    check(bw, ob, None, "after partial refusal")


def test_other_builders_and_modes_with_per_env_sources(oracle_mod):
    """Every observation builder and the availability mask without foreign lasers use the env's colours; the map-wide update broadcasts; snapshots carry the sources; the fused rollout and the
    lane-per-env diagnostic kernel agree with single steps."""
    import torch

    from lle_amd import BatchedWorld, _capi
    from oracle import observers as oo

    text = EXTRA_MAPS["nested"]
    n = 128
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    A, L = ob.A, bw.map.n_sources
    mirror = Mirror(ob, n, L)
    rng = np.random.default_rng(3)
    colours = legal_colours(bw.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    enabled = rng.integers(0, 1 << L, size=n).astype(np.int32)
    bw.set_sources(torch.from_numpy(colours), torch.from_numpy(enabled))
    mirror.apply(colours, enabled)
    for t in range(5):
        bw.step(sample=True, seed=2, t=t)
        check(bw, ob, ob.step(None, seed=2, t=t), f"t={t}")
    part = bw.observe_as(_capi.LLE_OBS_PARTIAL, 5).cpu().numpy()
    lay = bw.observe_as(_capi.LLE_OBS_LAYERED).cpu().numpy()
    strict = bw.available_actions(False).cpu().numpy()
    for e in range(0, n, 11):
        w = ob.world(e)
        assert np.array_equal(part[e].astype(np.float32), oo.partial_observe(w, 5)), e
        assert np.array_equal(lay[e].astype(np.float32), oo.layered_observe(w)[0]), e
        assert np.array_equal(strict[e], oo.available_actions(w, False)), e
    persp = bw.observe_as(_capi.LLE_OBS_PERSPECTIVE).cpu().numpy()
    padded = bw.observe_as(_capi.LLE_OBS_LAYERED_PADDED, 2).cpu().numpy()
    for e in range(0, n, 11):
        w = ob.world(e)
        assert np.array_equal(persp[e].astype(np.float32), oo.perspective_observe(w)), e
        assert np.array_equal(padded[e].astype(np.float32), oo.layered_padded_observe(w, 2)[0]), e

    # snapshot / restore carry colours, flags and the per-env reset states
    snap = bw.snapshot()
    before = {k: getattr(bw, k).clone() for k in ("pos", "bits", "gems", "beams", "avail", "src_colour", "src_enabled", "obs")}
    bw.set_sources(torch.zeros((n, L), dtype=torch.uint8), torch.zeros(n, dtype=torch.int32))
    for t in range(3):
        bw.step(sample=True, auto_reset=True, seed=9, t=t)
    bw.restore(snap)
    for k, v in before.items():
        assert torch.equal(getattr(bw, k), v), k
    for t in range(5, 9):  # the restored batch continues exactly like the oracle (auto-reset uses each env's reset state)
        bw.step(sample=True, auto_reset=True, seed=2, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=2, t=t), f"after restore t={t}")

    # fused rollout == single steps, and the lane-per-env diagnostic kernel == the lane-per-agent kernel
    twin = BatchedWorld(text, n)
    twin.set_sources(bw.src_colour.clone(), bw.src_enabled.clone())
    twin.restore(bw.snapshot())
    diag = BatchedWorld(text, n, envs_per_wave=16)
    diag.set_sources(bw.src_colour.clone(), bw.src_enabled.clone())
    diag.restore(bw.snapshot())
    twin.rollout(6, auto_reset=True, seed=4, t=100)
    for t in range(100, 106):
        bw.step(sample=True, auto_reset=True, seed=4, t=t)
        diag.step(sample=True, auto_reset=True, seed=4, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=4, t=t), f"t={t}")
    for k in ("pos", "bits", "gems", "beams", "avail", "obs"):
        assert torch.equal(getattr(bw, k), getattr(twin, k)), k
        assert torch.equal(getattr(bw, k), getattr(diag, k)), k

    # the map-wide update broadcasts the map's sources to every env
    bw.map.set_source(0, enabled=False)
    bw.map.set_source(1, agent_id=0)
    bw.update_sources()
    for e in range(n):
        w = ob.world(e)
        for l, s in enumerate(bw.map.sources()):
            w.set_source(l, colour=int(s.agent_id))
            if bool(s.enabled) != mirror.enabled[e, l]:
                w.set_source(l, enabled=bool(s.enabled))
                mirror.enabled[e, l] = bool(s.enabled)
    check(bw, ob, None, "after the map-wide update")
    for t in range(200, 204):
        bw.step(sample=True, auto_reset=True, seed=6, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=6, t=t), f"t={t}")


@pytest.mark.parametrize("name", ["level6", "nested", "gen_16x16_12agents"])
def test_reset_sources_equals_reset_then_set_sources(name):
    """lle_batch_reset_sources (LLE.reset with randomize_lasers in one launch) leaves every buffer exactly as
    lle_batch_reset followed by lle_batch_set_sources on the same envs -- refused colours and `done` as the mask included."""
    import torch

    from lle_amd import BatchedWorld

    text = LEVELS[6] if name == "level6" else EXTRA_MAPS[name]
    n = 1000
    a, b = BatchedWorld(text, n), BatchedWorld(text, n)
    A, L = a.map.n_agents, a.map.n_sources
    g = torch.Generator(device="cuda").manual_seed(11)
    names = ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done", "obs", "src_colour", "src_enabled")
    for rnd in range(12):
        for t in range(4):
            for w in (a, b):
                w.step(sample=True, auto_reset=False, seed=5, t=4 * rnd + t)
        colours = legal_colours(a.map, torch.randint(0, A, (n, L), generator=g, device="cuda", dtype=torch.uint8))
        if rnd % 3 == 1:
            colours[::7, 0] = A + 1  # refused envs: reset all the same, sources unchanged
        enabled = torch.randint(0, 1 << L, (n,), generator=g, device="cuda", dtype=torch.int32) if rnd % 2 else None
        if rnd % 4 == 3:
            mask_a, mask_b = None, None
        elif rnd % 4 == 2:
            mask_a, mask_b = a.done.clone(), b.done  # the live buffer itself
        else:
            mask_a = (torch.rand(n, generator=g, device="cuda") < 0.4).to(torch.uint8)
            mask_b = mask_a
        a.reset(mask_a)
        a.set_sources(colours=colours, enabled=enabled, env_mask=mask_a)
        b.set_sources(colours=colours, enabled=enabled, env_mask=mask_b, reset_first=True)
        for k in names:
            assert torch.equal(getattr(a, k), getattr(b, k)), (rnd, k)


def test_reset_sources_without_observation_then_step():
    """LLE_STEP_NO_OBS on lle_batch_reset_sources: the state is the same, and the step that follows writes the same
    observation as after a reset that wrote one."""
    import torch

    from lle_amd import BatchedWorld

    n = 777
    a, b = BatchedWorld(LEVELS[6], n), BatchedWorld(LEVELS[6], n)
    g = torch.Generator(device="cuda").manual_seed(2)
    for rnd in range(6):
        colours = torch.randint(0, 4, (n, a.map.n_sources), generator=g, device="cuda", dtype=torch.uint8)
        mask = (torch.rand(n, generator=g, device="cuda") < 0.5).to(torch.uint8)
        a.set_sources(colours=colours, env_mask=mask, reset_first=True)
        b.set_sources(colours=colours, env_mask=mask, reset_first=True, write_obs=False)
        for k in ("pos", "bits", "gems", "beams", "avail", "done", "src_colour"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (rnd, k)
        for t in range(3):
            a.step(sample=True, seed=1, t=3 * rnd + t)
            b.step(sample=True, seed=1, t=3 * rnd + t)
            assert torch.equal(a.obs, b.obs), (rnd, t)


def test_colour_that_crosses_a_start_is_refused_per_env(oracle_mod):
    """The start check of the binding's LaserSource.set_colour (pylaser_source.rs:121-139; python/tests/test_world.py:537-545):
    such an env keeps its sources and gets LLE_ENV_COLOUR_CROSSES_START; BatchedLLE(randomize_lasers=True) refuses the map."""
    import torch

    from lle_amd import BatchedLLE, BatchedWorld, _capi

    text = "L0E X X . S0\n@ @ @ S1 ."       # S0 on the beam of source 0: only colour 0 is acceptable
    n = 64
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    colours = np.zeros((n, 1), np.uint8)
    colours[3::5, 0] = 1
    bw.set_sources(torch.from_numpy(colours))
    err = bw.err.cpu().numpy()
    assert np.array_equal(err, np.where(colours[:, 0] == 1, _capi.LLE_ENV_COLOUR_CROSSES_START, 0))
    check(bw, ob, None, "refused envs untouched, accepted ones keep colour 0")
    for t in range(6):
        bw.step(sample=True, seed=3, t=t)
        check(bw, ob, ob.step(None, seed=3, t=t), f"t={t}")
    with pytest.raises(ValueError, match="cross the start position"):
        BatchedLLE(text, n, randomize_lasers=True)
    BatchedLLE(LEVELS[6], n, randomize_lasers=True)  # no beam over a start: fine


# ---- LLE_STEP_RECOLOUR_RESETS: LLE.reset with randomize_lasers inside the step kernel
@pytest.mark.parametrize("name", ["level6", "level5", "three_beams", "q1", "many_agents", "gen_20_lasers", "config5_32x32"])
@pytest.mark.parametrize("with_flags", [False, True])
def test_recolour_resets_inside_the_step(oracle_mod, name, with_flags):
    """An env that the step auto-resets also draws a fresh colour per source (python/lle/env/env.py:189-203: world.reset()
    under the colours it had, then set_colour on the live world).  The oracle side does exactly that per env -- reset(),
    then the colours from the documented hash -- and every buffer is compared after every step; with_flags: some sources
    disabled per env (their reset beams stay off)."""
    import torch

    from lle_amd import BatchedWorld
    from lle_amd._capi import RECOLOUR_SALT

    text = dict(MAPS, config5_32x32=EXTRA_MAPS["config5_32x32"])[name]
    n, seed, off = 1200, 4242, 17
    bw = BatchedWorld(text, n)
    m = bw.map
    A, L = m.n_agents, m.n_sources
    assert m.max_cell_layers <= 2
    allowed = [[c for c in range(A) if m.colour_allowed(s, c)] for s in range(L)]
    ob = oracle_mod.OracleBatch(text, n)
    mirror = Mirror(ob, n, L)
    rng = np.random.default_rng(1)
    enabled = (rng.integers(0, 1 << min(L, 30), n) | (rng.integers(0, 3, n) > 0) * ((1 << L) - 1)).astype(np.int64) & ((1 << L) - 1) if with_flags else None
    start_colours = legal_colours(m, rng.integers(0, A, (n, L)).astype(np.uint8))
    bw.set_sources(colours=torch.from_numpy(start_colours).cuda(), enabled=None if enabled is None else torch.from_numpy(enabled.astype(np.int32)).cuda())
    mirror.apply(start_colours, enabled)
    colours = start_colours.copy()
    check(bw, ob, None, f"{name} after set_sources")
    seen = [set() for _ in range(L)]
    for t in range(70):
        over = bw.done.cpu().numpy().astype(bool)
        for e in np.nonzero(over)[0]:
            w = ob.world(int(e))
            w.reset()
            for s in range(L):
                if allowed[s]:
                    h = oracle_mod.action_hash(seed ^ RECOLOUR_SALT, off + int(e), t, s)
                    colours[e, s] = allowed[s][(h * len(allowed[s])) >> 16]
                    w.set_source(s, colour=int(colours[e, s]))
                    seen[s].add(int(colours[e, s]))
        bw.step(sample=True, auto_reset=True, recolour_resets=True, seed=seed, t=t, env_offset=off)
        ostep = ob.step(None, auto_reset=False, seed=seed, t=t, env_offset=off)
        eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
        assert np.array_equal(eng["ev_count"] >> 7, over.astype(np.uint8)), f"{name} t={t}: which envs were reset"
        eng["ev_count"] = eng["ev_count"] & 0x7F
        assert_step_equal(eng, ostep, f"{name} t={t}")
        assert_state_equal(eng, ob.dump(), f"{name} t={t}")
        assert np.array_equal(bw.src_colour.cpu().numpy()[:, :L], colours), f"{name} t={t}: LLE_BUF_SRC_COLOUR"
    if name in ("level6", "three_beams"):
        assert all(seen[s] == set(allowed[s]) for s in range(L)), seen  # every allowed colour is reached (test_env.py:381-400)
    # the reset record the kernel keeps per env follows the colours: a plain masked reset afterwards equals the oracle's
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    bw.reset(torch.from_numpy(mask).cuda())
    for e in np.nonzero(mask)[0]:
        ob.world(int(e)).reset()
    check(bw, ob, None, f"{name} masked reset after recoloured resets")


def test_recolour_resets_argument_checks():
    import torch

    from lle_amd import BatchedWorld

    bw = BatchedWorld(LEVELS[6], 256)
    with pytest.raises(Exception, match="per-environment sources"):
        bw.step(sample=True, auto_reset=True, recolour_resets=True)
    bw.set_sources(colours=torch.zeros((256, 3), dtype=torch.uint8, device="cuda") + torch.tensor([2, 0, 1], dtype=torch.uint8, device="cuda"))
    with pytest.raises(Exception, match="pass both"):
        bw.step(sample=True, recolour_resets=True)
    deep = BatchedWorld(EXTRA_MAPS["four_layers"], 256)
    A, L = deep.map.n_agents, deep.map.n_sources
    deep.set_sources(colours=torch.from_numpy(legal_colours(deep.map, np.zeros((256, L), np.uint8))).cuda())
    if deep.map.max_cell_layers > 2:
        with pytest.raises(Exception, match="more than two laser layers"):
            deep.step(sample=True, auto_reset=True, recolour_resets=True)


@pytest.mark.parametrize("name", ["gen_12x13_4agents_8lasers", "many_agents", "gen_20_lasers", "config5_32x32"])
def test_fused_rollout_with_per_env_sources_on_many_source_maps(name):
    """lle_batch_rollout on batches with per-environment sources (step_kernel MODE 3) on maps with 8, 14, 20 and 8 sources
    (LM = 8 / 16 / 32 beam words): the fused rollout -- in place and into a ring -- must leave what the same number of single
    steps leaves (MODE 5 / 8, compared with the oracle above), state and every ring slot."""
    import torch

    from lle_amd import BatchedWorld, mapgen

    text = mapgen.generate(12, 13, 4, 8, 4, seed=2) if name.startswith("gen_12x13") else EXTRA_MAPS[name]
    n = 384
    a, b, c = BatchedWorld(text, n), BatchedWorld(text, n), BatchedWorld(text, n)
    A, L = a.map.n_agents, a.map.n_sources
    rng = np.random.default_rng(12)
    colours = torch.from_numpy(legal_colours(a.map, rng.integers(0, A, size=(n, L), dtype=np.uint8)))
    enabled = torch.from_numpy((rng.integers(0, 1 << min(L, 30), size=n) | (rng.integers(0, 2, size=n) * ((1 << L) - 1))).astype(np.int64).astype(np.int32))
    for w in (a, b, c):
        w.set_sources(colours, enabled)
    T = 12
    ring = c.make_ring(4)
    a.rollout(T, auto_reset=True, seed=31, t=0)
    c.rollout(T, auto_reset=True, seed=31, t=0, ring=ring, ring_pos=0)
    obs_of_step = {}
    for t in range(T):
        b.step(sample=True, auto_reset=True, seed=31, t=t)
        if t >= T - 4:
            obs_of_step[t % 4] = (b.obs.clone(), b.actions.clone())
    for k in ("pos", "bits", "gems", "beams", "avail", "obs", "err", "evcount", "events", "done"):
        assert torch.equal(getattr(a, k), getattr(b, k)), (name, k)
        if k != "obs":
            assert torch.equal(getattr(c, k), getattr(b, k)), (name, "ring", k)
    for slot, (obs, acts) in obs_of_step.items():
        assert torch.equal(ring["obs"][slot], obs) and torch.equal(ring["actions"][slot], acts), (name, "ring slot", slot)
    assert a.stats()["env_steps"] == b.stats()["env_steps"] == n * T
    assert a.stats() == b.stats() == c.stats()  # every counter (the rollout keeps them in LDS across its steps: tables.h PES_WAVE_EXTRA_BYTES)


@pytest.mark.parametrize("name", ["level6", "level5", "nested", "three_beams", "many_agents"])
def test_incremental_rows_with_per_env_sources(oracle_mod, name):
    """LLE_STEP_INCREMENTAL_OBS on batches with per-environment sources (tables.h off_pes_dyn_chunks: every laser plane is dynamic, the
    WALL / VOID / EXIT lines are not written): the buffer's content equals that of a twin stepped without the flag and the oracle's,
    through re-colourings mid-episode, masked subsets, and resets re-coloured inside the step kernel."""
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n = 300
    ob = oracle_mod.OracleBatch(text, n)
    a, b = BatchedWorld(text, n), BatchedWorld(text, n)
    A, L = ob.A, a.map.n_sources
    mirror = Mirror(ob, n, L)
    rng = np.random.default_rng(9)
    t = 0
    for episode in range(3):
        colours = legal_colours(a.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
        enabled = rng.integers(0, 1 << L, size=n, dtype=np.int64).astype(np.int32)
        mask = (rng.random(n) < 0.7).astype(np.uint8) if episode else None
        for bw in (a, b):
            bw.set_sources(torch.from_numpy(colours), torch.from_numpy(enabled), None if mask is None else torch.from_numpy(mask))
        mirror.apply(colours, enabled, mask)
        for _ in range(8):
            a.step(sample=True, auto_reset=True, seed=13, t=t, incremental_obs=True)
            b.step(sample=True, auto_reset=True, seed=13, t=t)
            ostep = ob.step(None, auto_reset=True, seed=13, t=t)
            assert torch.equal(a.obs_rows, b.obs_rows), (name, t)
            check(a, ob, ostep, f"{name} incremental + per-env sources t={t}")
            t += 1
    if a.map.max_cell_layers <= 2:  # resets re-coloured inside the step kernel (LLE_STEP_RECOLOUR_RESETS)
        for _ in range(12):
            a.step(sample=True, auto_reset=True, recolour_resets=True, seed=13, t=t, incremental_obs=True)
            b.step(sample=True, auto_reset=True, recolour_resets=True, seed=13, t=t)
            assert torch.equal(a.obs_rows, b.obs_rows) and torch.equal(a.src_colour, b.src_colour), (name, t)
            t += 1
