"""BatchedLLE (lle_amd/env.py) against a per-env restatement of the reference's LLE host class on oracle worlds
(tests/oracle_env.py): observation, state, reward (both strategies), done, available actions, reset with recolouring."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.oracle_env import OracleLLE
from tests.parity_util import EXTRA_MAPS, legal_colours

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("level6", dict(obs_type="layered", state_type="state")),
    ("level6", dict(obs_type="partial5x5", state_type="normalized-state", walkable_lasers=False, multi_objective=True)),
    ("nested", dict(obs_type="perspective", state_type="layered", walkable_lasers=False)),
    ("level3", dict(obs_type="flattened", state_type="state", multi_objective=True)),
    ("exit_under_beam", dict(obs_type="layered-padded", padding_size=2, state_type="partial3x3")),
]


@pytest.mark.parametrize("name,kw", CONFIGS, ids=[f"{n}-{k['obs_type']}" for n, k in CONFIGS])
@pytest.mark.parametrize("randomize", [False, True])
def test_batched_lle_matches_per_env_restatement(oracle_mod, name, kw, randomize):
    import torch

    from lle_amd import BatchedLLE

    text = LEVELS[int(name[-1])] if name.startswith("level") else EXTRA_MAPS[name]
    n = 96
    env = BatchedLLE(text, n, randomize_lasers=randomize, seed=5, **kw)
    okw = {k: v for k, v in kw.items()}
    refs = [OracleLLE(oracle_mod.OracleWorld(text), **okw) for _ in range(n)]
    A, L = env.n_agents, env.world.map.n_sources
    rng = np.random.default_rng(0)

    def same_obs(got, want, where):
        got = got.cpu().numpy().astype(np.float32)
        if want.ndim == got.ndim + 1:            # the reference tiles n_agents identical copies; the batch carries one
            assert all(np.array_equal(want[0], want[k]) for k in range(1, want.shape[0])), where
            want = want[0]
        assert got.shape == want.shape and np.array_equal(got, want), where

    def compare(step_out, rewards, where):
        obs, state = env.get_observation(), env.get_state()
        avail = env.available_actions().cpu().numpy()
        done = env.done.cpu().numpy()
        for e in range(0, n, 7):
            same_obs(obs[e], refs[e].get_observation(), f"{where} obs env {e}")
            same_obs(state[e], refs[e].get_state(), f"{where} state env {e}")
            assert np.array_equal(avail[e], refs[e].available_actions()), f"{where} avail env {e}"
            assert bool(done[e]) == refs[e].done, f"{where} done env {e}"
        if step_out is not None:
            r = step_out["reward"].cpu().numpy()
            for e in range(n):
                assert np.array_equal(r[e], rewards[e]), f"{where} reward env {e}: {r[e]} != {rewards[e]}"

    colours = torch.from_numpy(legal_colours(text, rng.integers(0, A, size=(n, L), dtype=np.uint8))) if randomize else None
    env.reset(colours=colours)
    for e in range(n):
        refs[e].reset(None if colours is None else colours[e].numpy())
    compare(None, None, "after reset")
    for t in range(25):
        # finished envs are reset first (auto_reset), with fresh colours when lasers are randomised
        done = env.done.cpu().numpy()
        if randomize:
            colours = torch.from_numpy(legal_colours(text, rng.integers(0, A, size=(n, L), dtype=np.uint8)))
            env.reset(env_mask=torch.from_numpy(done.astype(np.uint8)), colours=colours)
        for e in np.nonzero(done)[0]:
            refs[e].reset(None if not randomize else colours[e].numpy())
        avail = env.available_actions().cpu().numpy() if not randomize else None
        # uniform over each agent's available actions (World.available_actions: the walkable_lasers filter is advisory)
        wa = env.world.available_actions(True).cpu().numpy()
        if not randomize:
            # the kernel's auto-reset happens inside step(): the availability of a finished env is that of its reset state
            for e in np.nonzero(done)[0]:
                wa[e] = np.array([[a in lst for a in range(5)] for lst in refs[e].w.available_actions()])
        actions = np.zeros((n, A), np.uint8)
        for e in range(n):
            for a in range(A):
                actions[e, a] = rng.choice(np.nonzero(wa[e, a])[0])
        out = env.step(torch.from_numpy(actions), auto_reset=not randomize)
        rewards = [refs[e].step(actions[e])[0] for e in range(n)]
        assert int(out["err"].max()) == 0
        compare(out, rewards, f"t={t}")


# ---- the reference's own LLE tests (tests/golden/kat_env.json) through BatchedLLE
from tests.kat_env_runner import load_cases, run_case  # noqa: E402

ENV_CASES = load_cases()


class _BatchedAdapter:
    def __init__(self, case):
        from lle_amd import BatchedLLE
        self.n = 70  # every env plays the same script; the first and the last are checked
        self.env = BatchedLLE(case["map"], self.n, multi_objective=case["multi_objective"])

    def reset(self):
        self.env.reset()

    def step(self, actions):
        import torch
        out = self.env.step(torch.tensor([actions] * self.n, dtype=torch.uint8))
        assert int(out["err"].max()) == 0
        r, d = out["reward"].cpu().numpy(), out["done"].cpu().numpy()
        assert np.array_equal(r[0], r[-1]) and d[0] == d[-1]
        return r[0], d[0]

    def set_state(self, positions, gems, alive):
        import torch
        n = self.n
        err = self.env.set_state(torch.tensor([positions] * n, dtype=torch.uint8), torch.tensor([gems] * n, dtype=torch.bool),
                                 torch.tensor([alive] * n, dtype=torch.bool))
        assert int(err.max()) == 0

    def metrics(self):
        return {"has-arrived": [bool(x) for x in self.env.agents_arrived()[-1].cpu().numpy()],
                "is-alive": [bool(x) for x in self.env.agents_alive()[-1].cpu().numpy()]}

    def done(self):
        d = self.env.done.cpu().numpy()
        assert d[0] == d[-1]
        return bool(d[-1])

    def available(self):
        a = self.env.available_actions().cpu().numpy()
        assert np.array_equal(a[0], a[-1])
        return a[-1]


@pytest.mark.parametrize("case", ENV_CASES, ids=[c["name"] for c in ENV_CASES])
def test_batched_lle_env_kat(case):
    run_case(_BatchedAdapter, case)


@pytest.mark.parametrize("name", ["level6", "nested"])
@pytest.mark.parametrize("per_env", [False, True])
def test_env_outputs_equals_separate_entry_points(name, per_env):
    """lle_batch_env_outputs (one launch) against the entry points it fuses: observe_as(state / normalized-state),
    available_actions(walkable_lasers), the two reward strategies, done, alive / arrived."""
    import torch

    from lle_amd import BatchedWorld, _capi

    text = LEVELS[6] if name == "level6" else EXTRA_MAPS[name]
    n = 1000
    w = BatchedWorld(text, n)
    A, G = w.map.n_agents, w.map.n_gems
    if per_env:
        g = torch.Generator(device="cuda").manual_seed(3)
        w.set_sources(colours=legal_colours(w.map, torch.randint(0, A, (n, w.map.n_sources), generator=g, device="cuda").to(torch.uint8)))
    for t in range(25):
        w.step(sample=True, auto_reset=(t % 5 == 4), seed=17, t=t)
        for normalize in (False, True):
            for multi in (False, True):
                for walkable in (False, True):
                    state = torch.full((n, 3 * A + G), -7.0, device="cuda")
                    reward = torch.full((n, 4 if multi else 1), -7.0, device="cuda")
                    done, avail = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty((n, A, 5), dtype=torch.uint8, device="cuda")
                    alive, arrived = torch.empty((n, A), dtype=torch.uint8, device="cuda"), torch.empty((n, A), dtype=torch.uint8, device="cuda")
                    w.env_outputs(state=state, normalize_state=normalize, reward=reward, multi_objective=multi, done=done,
                                  available=avail, walkable_lasers=walkable, alive=alive, arrived=arrived)
                    kind = _capi.LLE_OBS_NORMALIZED_STATE if normalize else _capi.LLE_OBS_STATE
                    assert torch.equal(state.view(torch.int32), w.observe_as(kind, 0).view(torch.int32)), (t, "state")
                    want_r = w.reward_multi_objective() if multi else w.reward_single_objective().unsqueeze(1)
                    assert torch.equal(reward, want_r), (t, "reward")
                    assert torch.equal(done, w.done), (t, "done")
                    assert torch.equal(avail.view(torch.bool), w.available_actions(walkable)), (t, "available")
                    assert torch.equal(alive.bool(), w.agents_alive()) and torch.equal(arrived.bool(), w.agents_arrived()), (t, "flags")
    # every output is optional
    w.env_outputs()
    only = torch.empty((n, 1), device="cuda")
    w.env_outputs(reward=only)
    assert torch.equal(only, w.reward_single_objective().unsqueeze(1))


# ---- python/tests/test_env.py:381-430: randomize_lasers (the reference's only pins of per-env source colours)
MAP_RANDOMIZED = "S0 S1 L0S\n.   . L1W\n.   . L0W\nX   X  ."


def test_randomized_lasers_reach_every_colour_and_refresh_the_source_markers():
    """test_randomized_lasers (:381-400): over repeated resets every source takes every colour.  
    test_randomized_lasers_updates_static_observation_layer (:403-430): after EVERY reset the layered observation holds -1
    at each source's cell on layer LASER_0 + the colour the source has now.  A batch draws a colour per env and source."""
    import torch

    from lle_amd import BatchedLLE

    n = 1024
    env = BatchedLLE(MAP_RANDOMIZED, n, obs_type="layered", randomize_lasers=True, seed=3)
    A, sources = env.n_agents, env.world.map.sources()
    L = len(sources)
    seen = torch.zeros(L, A, dtype=torch.bool)
    rows = torch.arange(n, device="cuda")
    for _ in range(50):
        obs, _state = env.reset()
        colours = env.world.src_colour[:, :L].long()
        assert int(colours.min()) >= 0 and int(colours.max()) < A
        for l, s in enumerate(sources):
            for c in range(A):
                seen[l, c] |= bool((colours[:, l] == c).any())
            assert bool((obs[rows, A + colours[:, l], s.i, s.j] == -1).all()), f"source {l}: marker missing"
            # ... and on no other laser layer of that cell (a stale marker of the previous episode's colour)
            others = obs[:, A:2 * A, s.i, s.j].clone()
            others[rows, colours[:, l]] = 0
            assert int(others.abs().sum()) == 0, f"source {l}: stale marker"
    assert bool(seen.all()), seen
    # the single-env restatement of the same loop through lle_amd.World (the facade the reference's LLE wraps)
    from lle_amd import World
    import random
    rng = random.Random(0)
    w = World(MAP_RANDOMIZED)
    encountered = [[False] * w.n_agents for _ in w.laser_sources]
    for _ in range(60):
        w.reset()
        for source in w.laser_sources:
            source.set_colour(rng.randint(0, w.n_agents - 1))
        for source in w.laser_sources:
            encountered[source.laser_id][source.agent_id] = True
    assert all(all(ce) for ce in encountered)


@pytest.mark.parametrize("map_name", ["randomized", "level6"])
def test_randomized_lasers_through_auto_reset_steps(map_name):
    """The same pins along ROLLOUTS: BatchedLLE.step(auto_reset=True) with randomize_lasers re-colours the envs it resets
    inside the step kernel (LLE_STEP_RECOLOUR_RESETS).  After every step: colours in range, the -1 marker of every source on
    the layer of its CURRENT colour and nowhere else, an env that was not reset keeps its colours, every colour is reached,
    and the one-launch step (fused=True) returns what the two-launch step returns."""
    import torch

    from lle_amd import BatchedLLE

    text = MAP_RANDOMIZED if map_name == "randomized" else LEVELS[6]
    n = 2048
    a = BatchedLLE(text, n, obs_type="layered", randomize_lasers=True, seed=11)
    b = BatchedLLE(text, n, obs_type="layered", randomize_lasers=True, seed=11)
    assert a._recolour_in_step
    a.reset(), b.reset()
    b.world.set_sources(colours=a.world.src_colour[:, : a.world.map.n_sources].clone())  # (reset() draws from the torch generator)
    A, sources = a.n_agents, a.world.map.sources()
    L = len(sources)
    seen = torch.zeros(L, A, dtype=torch.bool)
    rows = torch.arange(n, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(2)
    n_resets = 0
    for t in range(80):
        before = a.world.src_colour[:, :L].clone()
        over = a.done.clone()
        avail = a.available_actions()
        acts = torch.multinomial(avail.reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
        x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (k, t)
        colours = a.world.src_colour[:, :L].long()
        assert torch.equal(colours, b.world.src_colour[:, :L].long())
        assert int(colours.min()) >= 0 and int(colours.max()) < A
        assert torch.equal(colours[~over], before[~over].long()), "an env that was not reset changed colour"
        n_resets += int(over.sum())
        obs = x["obs"]
        for l, s in enumerate(sources):
            for c in range(A):
                seen[l, c] |= bool((colours[over, l] == c).any())
            assert bool((obs[rows, A + colours[:, l], s.i, s.j] == -1).all()), f"t={t} source {l}: marker missing"
            others = obs[:, A:2 * A, s.i, s.j].clone()
            others[rows, colours[:, l]] = 0
            assert int(others.abs().sum()) == 0, f"t={t} source {l}: stale marker"
    assert n_resets > n and bool(seen.all()), (n_resets, seen)


def test_envs_of_a_batch_are_independent_copies():
    """python/tests/test_core.py:127-141 (test_deep_copy): a copy of an env must not finish when the original does.
    In a batch the copies are the environments: one walks onto the exit, its neighbour stays."""
    import torch

    from lle_amd import BatchedLLE

    env = BatchedLLE("S0 X", 2)
    env.reset()
    out = env.step(torch.tensor([[2], [4]], dtype=torch.uint8))  # EAST | STAY
    assert out["done"].tolist() == [True, False]
    out = env.step(torch.tensor([[4], [2]], dtype=torch.uint8))
    assert out["done"].tolist() == [True, True]


def test_set_state_round_trip_along_a_rollout(oracle_mod):
    """python/tests/test_env.py:144-180 (test_set_state): every state visited by a random rollout (resets at episode ends
    included), handed back through LLE.set_state, is the state the world then reports.  The reference runs it on a
    laser-free TOML map with random spawns (v2 maps: out of scope); same property on a laser-free v1 map, and on
    level 6 for the states whose re-derivation is lossless (no agent dead: quirk Q6 covers the others)."""
    import torch

    from lle_amd import BatchedLLE

    for text, only_alive in (("S0 . . G X\nS1 . @ . X\n.  G . . .\nS2 . . G X", False), (LEVELS[6], True)):
        n = 128
        env, probe = BatchedLLE(text, n, state_type="state"), BatchedLLE(text, n, state_type="state")
        env.reset()
        probe.reset()
        A, G = env.n_agents, env.world.map.n_gems
        for t in range(60):
            env.world.step(sample=True, auto_reset=True, seed=4, t=t)
            pos, gems, alive = env.world.pos.clone(), env.world.gems_collected(), env.world.agents_alive()
            err = probe.set_state(pos, gems, alive)
            ok = alive.all(dim=1) if only_alive else torch.ones(n, dtype=torch.bool, device="cuda")
            assert int(err[ok].max()) == 0, f"t={t}"
            assert torch.equal(probe.world.pos[ok], pos[ok]) and torch.equal(probe.world.gems_collected()[ok], gems[ok])
            assert torch.equal(probe.world.agents_alive()[ok], alive[ok])
            assert torch.equal(probe.get_state()[ok], env.get_state()[ok])


def test_env_static_description():
    """python/tests/test_core.py:71-114 (width / height, default and flattened state shapes), :195-201 (agent_state_size),
    :238-241 (n_agents), :204-235 (observation types by string and by enum value)."""
    from lle_amd import BatchedLLE
    from lle_amd.observations import ObservationType

    env = BatchedLLE("S0 X .\n.  . .\n.  . .", 4)
    assert (env.width, env.height) == (3, 3)
    assert BatchedLLE("S0 X . .\n.  . . .\nG  . . .", 4).width == 4
    assert env.state_shape == (env.n_agents * 3 + env.world.map.n_gems,)
    env.reset()
    assert tuple(env.get_state().shape[1:]) == env.state_shape
    flat = BatchedLLE("S0 X .\n.  . .\n.  . .", 4, state_type=ObservationType.FLATTENED.value)
    assert flat.state_shape == (int(np.prod(flat.state_shape)),)
    flat.reset()
    assert tuple(flat.get_state().shape[1:]) == flat.state_shape == (6 * 3 * 3,)
    assert BatchedLLE(LEVELS[1], 4).agent_state_size == 2
    with pytest.raises((ValueError, NotImplementedError)):
        BatchedLLE(LEVELS[1], 4, state_type="flattened").agent_state_size
    assert BatchedLLE("S0 S1 X X", 4).n_agents == 2 and BatchedLLE(LEVELS[6], 4).n_agents == 4
    for name, member in (("layered", ObservationType.LAYERED), ("flattened", ObservationType.FLATTENED), ("partial3x3", ObservationType.PARTIAL_3x3),
                         ("partial5x5", ObservationType.PARTIAL_5x5), ("partial7x7", ObservationType.PARTIAL_7x7), ("state", ObservationType.STATE),
                         ("perspective", ObservationType.AGENT0_PERSPECTIVE_LAYERED)):
        a, b = BatchedLLE(LEVELS[1], 4, obs_type=name), BatchedLLE(LEVELS[1], 4, obs_type=member.value)
        assert a.observation_shape == b.observation_shape and a.state_shape == b.state_shape
        a.reset()
        assert tuple(a.get_observation().shape[1:]) == a.observation_shape, name


def test_builder_chain_names_and_the_small_members():
    """`lle.level(6).obs_type(...).build()` (python/lle/env/builder.py:12-166; env.py:222-243) for the batch; names as
    python/tests/test_env.py:310-331; `n_arrived` / `compute_done` (env.py:142-144,253-254) against the oracle-side env; the
    reference's own other pieces of the builder (pbrs, extras) refuse loudly."""
    import os
    import tempfile

    import torch

    import lle_amd
    from lle_amd import BatchedLLE, Builder, DeathStrategy
    from lle_amd.observations import ObservationType

    for lvl in range(1, 7):
        assert lle_amd.level(lvl).build(4).name == f"LLE-lvl{lvl}"
        assert BatchedLLE.level(lvl).multi_objective().build(4).name == f"LLE-lvl{lvl}-MO"
    assert lle_amd.from_str("S0 X").build(4).name == "LLE" and BatchedLLE.from_str("S0 X").multi_objective().multi_objective().build(4).name == "LLE-MO"
    with tempfile.NamedTemporaryFile(mode="w+") as f:
        f.write("S0 X")
        f.flush()
        assert lle_amd.from_file(f.name).build(4).name == f"LLE-{os.path.basename(f.name)}"
    with pytest.raises(FileNotFoundError):
        lle_amd.from_file("/no/such/map")
    assert BatchedLLE("S0 X", 4).name == "LLE" and BatchedLLE("S0 X", 4).death_strategy is DeathStrategy.END
    with pytest.raises(NotImplementedError):
        lle_amd.level(1).death_strategy("respawn").build(4)      # env.py:106-107
    with pytest.raises(ValueError):
        lle_amd.level(1).death_strategy("sleep").build(4)
    with pytest.raises(NotImplementedError):
        lle_amd.level(1).pbrs()
    with pytest.raises(NotImplementedError):
        lle_amd.level(1).add_extras("laser_subgoal")
    assert isinstance(lle_amd.level(1).add_extras(), Builder)
    a = lle_amd.level(1).obs_type("partial5x5").state_type(ObservationType.NORMALIZED_STATE).walkable_lasers(False).build(8)
    b = BatchedLLE(LEVELS[1], 8, obs_type=ObservationType.PARTIAL_5x5, state_type="normalized-state", walkable_lasers=False)
    assert (a.obs_type, a.state_type, a.walkable_lasers) == (b.obs_type, b.state_type, b.walkable_lasers) == ("partial5x5", "normalized-state", False)
    assert a.observation_shape == b.observation_shape
    # n_arrived / compute_done along a rollout, against one OracleLLE per env
    n = 64
    assert lle_amd.level(3).randomize_lasers().build(n, seed=3).randomize_lasers
    env = lle_amd.from_str("S0 . X\nS1 . X\n.  V .").build(n)   # exits two steps away, a void to die in
    env.reset()
    assert int(env.n_arrived.sum()) == 0 and not bool(env.compute_done().any())
    rng = np.random.default_rng(5)
    seen = 0
    for t in range(60):
        avail = env.available_actions().cpu().numpy()
        actions = np.array([[rng.choice(np.nonzero(avail[e, a])[0]) for a in range(env.n_agents)] for e in range(n)], dtype=np.uint8)
        env.step(torch.from_numpy(actions).cuda(), auto_reset=False)
        arrived = ((env.world.bits.cpu().numpy().view(np.uint64)[:, None] >> (np.arange(env.n_agents, dtype=np.uint64) + np.uint64(16))) & np.uint64(1)).sum(1)
        assert np.array_equal(env.n_arrived.cpu().numpy(), arrived.astype(np.int64))
        done = env.compute_done().cpu().numpy()
        alive = ((env.world.bits.cpu().numpy().view(np.uint64)[:, None] >> np.arange(env.n_agents, dtype=np.uint64)) & np.uint64(1)).sum(1)
        assert np.array_equal(done, (arrived == env.n_agents) | (alive < env.n_agents))
        seen += int(arrived.sum())
        env.reset(torch.from_numpy(done.astype(np.uint8)).cuda()) if done.any() else None
    assert seen > 0


@pytest.mark.parametrize("name", ["level1", "level6", "nested", "many_agents", "config5_32x32"])
@pytest.mark.parametrize("variant", ["plain", "normalized_multi", "per_env_sources", "two_maps"])
def test_one_launch_step_equals_step_plus_env_outputs(name, variant):
    """lle_batch_step_outputs (the step kernel writes LLE.step's state / reward / done / available / alive / arrived itself)
    against lle_batch_step followed by lle_batch_env_outputs, on twin batches: explicit actions with refused ones, sampled
    actions with auto-reset, every output bit for bit (float32 bit patterns), in the default, rollout-free general modes."""
    import torch

    from lle_amd import BatchedWorld

    text = LEVELS[int(name[-1])] if name.startswith("level") else EXTRA_MAPS[name]
    n = 640
    maps = [text, text] if variant == "two_maps" else text
    a, b = BatchedWorld(maps, n), BatchedWorld(maps, n)
    A, G, L = a.map.n_agents, a.map.n_gems, a.map.n_sources
    if variant == "per_env_sources" and L:
        g = torch.Generator(device="cuda").manual_seed(2)
        colours = legal_colours(a.map, torch.randint(0, A, (n, L), generator=g, device="cuda", dtype=torch.uint8))
        for w in (a, b):
            w.set_sources(colours=colours)
    norm, multi = variant == "normalized_multi", variant == "normalized_multi"

    def outs():
        return dict(state=torch.full((n, 3 * A + G), -7.0, device="cuda"), reward=torch.full((n, 4 if multi else 1), -7.0, device="cuda"),
                    done=torch.full((n,), 9, dtype=torch.uint8, device="cuda"), available=torch.full((n, A, 5), 9, dtype=torch.uint8, device="cuda"),
                    alive=torch.full((n, A), 9, dtype=torch.uint8, device="cuda"), arrived=torch.full((n, A), 9, dtype=torch.uint8, device="cuda"))
    ta, tb = outs(), outs()
    oa = a.make_env_outputs(normalize_state=norm, multi_objective=multi, **ta)
    rng = np.random.default_rng(1)
    for t in range(30):
        if t % 3 == 0:
            acts = torch.from_numpy(rng.integers(0, 6, size=(n, A), dtype=np.uint8)).cuda()  # unavailable and invalid ones included
            a.step(acts, env_out=oa)
            b.step(acts)
        else:
            a.step(sample=True, auto_reset=(t % 2 == 0), seed=3, t=t, env_out=oa)
            b.step(sample=True, auto_reset=(t % 2 == 0), seed=3, t=t)
        b.env_outputs(normalize_state=norm, multi_objective=multi, **tb)
        for k in ta:
            x, y = ta[k], tb[k]
            same = torch.equal(x.view(torch.int32), y.view(torch.int32)) if x.dtype == torch.float32 else torch.equal(x, y)
            assert same, (name, variant, t, k)
        for k in ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done", "obs", "reward"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (name, variant, t, k)


def test_batched_lle_one_launch_step():
    import torch

    from lle_amd import BatchedLLE

    n = 512
    a, b = BatchedLLE(LEVELS[6], n, multi_objective=True), BatchedLLE(LEVELS[6], n, multi_objective=True)
    a.reset()
    b.reset()
    for t in range(25):
        acts = a.available_actions().to(torch.uint8).argmax(dim=2).to(torch.uint8)  # first available action of every agent
        x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (t, k)
    with pytest.raises(ValueError):
        BatchedLLE(LEVELS[6], n, walkable_lasers=False).step(acts, fused=True)


def test_restore_does_not_bring_back_an_old_output_descriptor():
    """with per-env sources a snapshot used to include the device mirror of the lle_env_outputs descriptor,
    so restore() put back the pointers bound at snapshot time while the host still believed its copy current: the next
    fused step wrote into whatever those were.  snapshot -> fused step into NEW tensors -> restore -> fused step must equal
    the two-launch path on a twin."""
    import torch

    from lle_amd import BatchedLLE

    n = 1024
    a = BatchedLLE(LEVELS[6], n, obs_type="layered", randomize_lasers=True, seed=4)
    b = BatchedLLE(LEVELS[6], n, obs_type="layered", randomize_lasers=True, seed=4)
    a.reset(), b.reset()
    b.world.set_sources(colours=a.world.src_colour[:, : a.world.map.n_sources].clone())
    g = torch.Generator(device="cuda").manual_seed(9)

    def acts_of(env):
        return torch.multinomial(env.available_actions().reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)

    for _ in range(3):
        acts = acts_of(a)
        a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
    snap_a, snap_b = a.world.snapshot(), b.world.snapshot()
    old_fused = a._fused
    a._fused = None  # the next fused step binds fresh output tensors: the device descriptor changes after the snapshot
    acts = acts_of(a)
    x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
    assert all(torch.equal(x[k], y[k]) for k in ("obs", "state", "reward", "done", "available_actions"))
    del old_fused  # the tensors the snapshot-time descriptor pointed at are gone
    torch.cuda.empty_cache()
    a.world.restore(snap_a), b.world.restore(snap_b)
    a._t = b._t = 3
    for t in range(6):
        acts = acts_of(b)
        x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (k, t)


@pytest.mark.parametrize("kw", [dict(), dict(walkable_lasers=False, multi_objective=True), dict(obs_type="partial5x5", state_type="normalized-state"),
                                dict(obs_type="flattened", state_type="layered"), dict(obs_type="perspective", state_type="perspective"),
                                dict(randomize_lasers=True), dict(obs_type="layered-padded-2", state_type="partial3x3")])
def test_persistent_step_equals_the_allocating_step(kw):
    """BatchedLLE.step(persistent=True) -- bound C-ABI calls into persistent tensors (BatchedWorld.bound_step / bound_env_outputs /
    bound_observer) -- returns what the default step returns, for every observation / state pairing and option."""
    import torch

    from lle_amd import BatchedLLE

    n = 768
    a, b = BatchedLLE(LEVELS[6], n, seed=5, **kw), BatchedLLE(LEVELS[6], n, seed=5, **kw)
    a.reset(), b.reset()
    if kw.get("randomize_lasers"):
        b.world.set_sources(colours=a.world.src_colour[:, : a.world.map.n_sources].clone())
    g = torch.Generator(device="cuda").manual_seed(1)
    for t in range(20):
        avail = a.available_actions()
        acts = torch.multinomial(avail.reshape(-1, 5).float() + 1e-6, 1, generator=g).reshape(n, -1).to(torch.uint8)  # (now and then a refused one)
        x, y = a.step(acts, auto_reset=True, persistent=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert x[k].shape == y[k].shape and torch.equal(x[k], y[k]), (kw, t, k)
    # the bound calls on their own
    w = a.world
    from lle_amd import _capi
    f = w.bound_observer(_capi.LLE_OBS_PARTIAL, 3)
    assert torch.equal(f(), w.observe_as(_capi.LLE_OBS_PARTIAL, 3)) and f() is f.out
    h = w.bound_available_actions(False)
    assert torch.equal(h(), w.available_actions(False))


def test_seed_after_a_persistent_step_and_another_stream():
    """the bound calls of step(persistent=True) must not freeze the seed of the in-kernel colour draws nor the stream
    that was current when they were bound.  env.seed(x) after a persistent step gives the draws of the allocating path with the
    same seed; a persistent step issued inside `torch.cuda.stream(s)` lands on s (ordered with the caller's tensors there)."""
    import torch

    from lle_amd import BatchedLLE

    n = 640
    a, b = BatchedLLE(LEVELS[6], n, seed=5, randomize_lasers=True), BatchedLLE(LEVELS[6], n, seed=5, randomize_lasers=True)
    a.reset(), b.reset()
    b.world.set_sources(colours=a.world.src_colour[:, : a.world.map.n_sources].clone())
    g = torch.Generator(device="cuda").manual_seed(2)

    def acts_of(env):
        avail = env.available_actions()
        return torch.multinomial(avail.reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)

    for t in range(12):
        if t == 4:
            a.seed(77), b.seed(77)  # (after the step calls were bound with seed 5)
        acts = acts_of(b)
        x, y = a.step(acts, auto_reset=True, persistent=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (t, k)
        assert torch.equal(a.world.src_colour, b.world.src_colour), t
    assert a.world.stats()["auto_resets"] > 0
    # another stream: the producer of `acts` and the step are ordered on `side` alone
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for t in range(6):
            acts = acts_of(b)
            x, y = a.step(acts, auto_reset=True, persistent=True), b.step(acts, auto_reset=True, fused=False)
            for k in ("obs", "state", "reward", "done", "err"):
                assert torch.equal(x[k], y[k]), (t, k)
    side.synchronize()


@pytest.mark.parametrize("kw", [dict(), dict(randomize_lasers=True), dict(obs_type="partial7x7"), dict(state_type="layered"), dict(multi_objective=True, state_type="normalized-state"),
                                dict(walkable_lasers=False)])
def test_default_step_is_one_launch_into_fresh_tensors(kw):
    """The default BatchedLLE.step (round 4): the step kernel writes state / reward / available_actions itself, into tensors allocated for
    that step, wherever it can (walkable_lasers) -- same values as the two-launch step (fused=False), and every step's tensors its own."""
    import torch

    from lle_amd import BatchedLLE

    n = 1500
    a, b = BatchedLLE(LEVELS[6], n, seed=5, **kw), BatchedLLE(LEVELS[6], n, seed=5, **kw)
    a.reset(), b.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    earlier = []
    for t in range(25):
        probs = a.available_actions().reshape(-1, 5).float()
        probs[:, 4] += 1e-3  # (without walkable lasers an agent may have NO available action: multinomial asserts on an all-zero row)
        acts = torch.multinomial(probs, 1, generator=g).reshape(n, -1).to(torch.uint8)
        x, y = a.step(acts, auto_reset=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (kw, k, t)
        earlier.append((x["reward"], x["reward"].clone(), x["available_actions"], x["available_actions"].clone()))
    torch.cuda.synchronize()
    for r, rc, av, avc in earlier:  # no later step wrote into an earlier step's tensors
        assert torch.equal(r, rc) and torch.equal(av, avc)
    assert len({e[0].data_ptr() for e in earlier[-3:]}) == 3


@pytest.mark.parametrize("kw", [dict(), dict(obs_type="flattened"), dict(randomize_lasers=True), dict(state_type="layered", obs_type="partial3x3"),
                                dict(obs_type="partial7x7"), dict(obs_type="perspective", state_type="layered-padded-2"), dict(obs_type="partial5x5", randomize_lasers=True)])
def test_batched_lle_in_the_learner_dtype(kw):
    """BatchedLLE(obs_dtype=...): the layered-style observations (layered, flattened, padded, perspective, partial -- the step launch's own partial
    writer included) arrive from the kernels as float32 -- the reference's element type, python/lle/observations.py:223 --, float16 or bfloat16;
    every step path (default one-launch, two launches, fused, persistent), reset included: the values of the int8 environment stepped alongside, cast."""
    import torch

    from lle_amd import BatchedLLE

    n = 700
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        a, b = BatchedLLE(LEVELS[6], n, seed=5, obs_dtype=dt, **kw), BatchedLLE(LEVELS[6], n, seed=5, **kw)
        (oa, sa), (ob, sb) = a.reset(), b.reset()
        layered_obs = True  # (every observation type of these cases is a layered-style one)
        assert oa.dtype == dt and torch.equal(oa, ob.to(oa.dtype))
        assert torch.equal(sa, sb.to(sa.dtype))
        g = torch.Generator(device="cuda").manual_seed(3)
        for t, how in enumerate([dict(), dict(fused=False), dict(fused=True), dict(persistent=True)] * 4):
            acts = torch.multinomial(a.available_actions().reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
            x, y = a.step(acts, auto_reset=True, **how), b.step(acts, auto_reset=True, **how)
            if layered_obs:
                assert x["obs"].dtype == dt and x["obs"].shape == y["obs"].shape
            if kw.get("state_type", "state") != "state":
                assert x["state"].dtype == dt
            for k in ("obs", "state", "reward", "done", "available_actions", "err"):
                assert torch.equal(x[k], y[k].to(x[k].dtype)), (kw, dt, how, k, t)
    with pytest.raises(ValueError):
        BatchedLLE(LEVELS[6], 64, obs_type="state", obs_dtype=torch.float32)
