"""The observation builders of observers.hip (layered-padded, perspective, partial k x k, state, availability bools)
through the C ABI: the reference's own tests via the lle_amd generator classes, a differential against the oracle's
restatement of python/lle/observations.py along random rollouts, and size-independent properties at 65 536 envs."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.kat_observers_runner import _Base, load_cases, run_case
from tests.observer_checks import compare_all
from tests.parity_util import EXTRA_MAPS, LONG_MAPS

pytestmark = pytest.mark.gpu

MAPS = {f"level{k}": v for k, v in LEVELS.items()}
MAPS.update(EXTRA_MAPS)
MAPS.update(LONG_MAPS)  # beams longer than 32 cells: chains of beam words (tables.h)
CASES = load_cases()


class GpuAdapter(_Base):
    """The reference's calling convention on lle_amd: ObservationType(...).get_observation_generator(world).observe()."""

    def __init__(self, case):
        from lle_amd import World
        self.w = World.level(case["level"]) if "level" in case else World(case["map"])
        w = self.w
        self.n_agents, self.n_gems, self.height, self.width = w.n_agents, w.n_gems, w.height, w.width

    def reset(self):
        self.w.reset()

    def step(self, actions):
        from lle_amd import Action
        self.w.step([Action(a) for a in actions])

    def get_state(self):
        s = self.w.get_state()
        return s.agents_positions, s.gems_collected, s.agents_alive

    def _generator(self, kind, param):
        from lle_amd.observations import LayeredPadded, ObservationType, PartialGenerator
        if kind == "partial":
            return PartialGenerator(self.w, param)
        if kind == "layered-padded":
            return LayeredPadded(self.w, param)
        return ObservationType.from_str(kind).get_observation_generator(self.w)

    def announced_shape(self, kind, param):
        return self._generator(kind, param).shape

    def observe(self, kind, param):
        return self._generator(kind, param).observe()

    def avail(self, walkable):
        return self.w.available_actions_mask(walkable)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_observers_kat(case):
    run_case(GpuAdapter, case)


def _engine(bw):
    import torch

    def observe(kind, param):
        try:
            out = bw.observe_as(kind, param)
        except IndexError:
            return None
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def avail(walkable):
        out = bw.available_actions(walkable)
        torch.cuda.synchronize()
        return out.cpu().numpy()

    return observe, avail


@pytest.mark.parametrize("name", list(MAPS))
def test_observers_along_rollout(oracle_mod, name):
    from lle_amd import BatchedWorld

    text = MAPS[name]
    n, steps = 200, 24  # ragged last wave of the 16-env-per-wave observer kernels
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    observe, avail = _engine(bw)
    envs = list(range(0, n, 23)) + [n - 1]
    compare_all(observe, avail, ob, envs, f"{name} after reset")
    for t in range(steps):
        auto_reset = t >= steps // 2
        ob.step(None, auto_reset=auto_reset, seed=99, t=t, env_offset=3, want_obs=False)
        bw.step(sample=True, auto_reset=auto_reset, seed=99, t=t, env_offset=3)
        if t % 6 == 5:
            compare_all(observe, avail, ob, envs, f"{name} t={t}")


def test_observers_follow_source_updates(oracle_mod):
    """Views are compiled from the sources' current colours: recolour / disable through the map, push, compare."""
    from lle_amd import BatchedWorld

    text = EXTRA_MAPS["nested"]
    n = 64
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    observe, avail = _engine(bw)
    for t in range(6):
        ob.step(None, seed=5, t=t, want_obs=False)
        bw.step(sample=True, seed=5, t=t)
    for lid, kw in ((0, dict(colour=1)), (1, dict(enabled=False)), (0, dict(colour=3)), (1, dict(enabled=True, colour=5))):
        for e in range(n):
            ob.world(e).set_source(lid, **kw)
        bw.map.set_source(lid, enabled=kw.get("enabled"), agent_id=kw.get("colour"))
        bw.update_sources()
        compare_all(observe, avail, ob, range(0, n, 9), f"nested after set_source({lid}, {kw})")


def test_full_size_properties():
    """65 536 level-6 envs: relations between the builders that hold for any state."""
    import torch

    from lle_amd import BatchedWorld, _capi

    n = 65536
    bw = BatchedWorld(LEVELS[6], n)
    for t in range(12):
        bw.step(sample=True, auto_reset=(t % 3 == 0), seed=3, t=t)
    A, G, H, W = bw.map.n_agents, bw.map.n_gems, bw.map.height, bw.map.width
    layered = bw.observe_as(_capi.LLE_OBS_LAYERED)
    assert torch.equal(layered, bw.obs)                      # the view kernel reproduces the step kernel's tensor
    persp = bw.observe_as(_capi.LLE_OBS_PERSPECTIVE)
    assert persp.shape == (n, A, 2 * A + 4, H, W)
    assert torch.equal(persp[:, 0], layered)
    for k in range(1, A):
        perm = list(range(2 * A + 4))
        perm[0], perm[k] = perm[k], perm[0]
        perm[A], perm[A + k] = perm[A + k], perm[A]
        assert torch.equal(persp[:, k], layered[:, perm]), k
    p = 2
    padded = bw.observe_as(_capi.LLE_OBS_LAYERED_PADDED, p)
    assert padded.shape == (n, 2 * (A + p) + 4, H, W)
    assert torch.equal(padded[:, :A], layered[:, :A]) and torch.all(padded[:, A:A + p] == 0)
    assert torch.equal(padded[:, A + p:2 * A + p], layered[:, A:2 * A]) and torch.all(padded[:, 2 * A + p:2 * (A + p)] == 0)
    assert torch.equal(padded[:, 2 * (A + p):], layered[:, 2 * A:])
    state = bw.observe_as(_capi.LLE_OBS_STATE)
    assert state.shape == (n, 3 * A + G) and state.dtype == torch.float32
    assert torch.equal(state[:, :2 * A].reshape(n, A, 2), bw.pos.to(torch.float32))
    alive = ((bw.bits.unsqueeze(1) >> torch.arange(A, device="cuda")) & 1).to(torch.float32)
    assert torch.equal(state[:, 2 * A + G:], alive)
    norm = bw.observe_as(_capi.LLE_OBS_NORMALIZED_STATE)
    dims = torch.tensor([H, W] * A, device="cuda", dtype=torch.float64)
    assert torch.equal(norm[:, :2 * A], (state[:, :2 * A].to(torch.float64) / dims).to(torch.float32))
    for k in (3, 7):
        part = bw.observe_as(_capi.LLE_OBS_PARTIAL, k)
        assert part.shape == (n, A, 2 * A + 3, k, k)
        c = k // 2
        for a in range(A):
            assert torch.all(part[:, a, a, c, c] == 1)      # every agent is at the centre of its own window
        # the window of agent a is the crop of the full map around it: compare the agent layers with the layered tensor
        pos = bw.pos.to(torch.int64)
        e = torch.arange(n, device="cuda")
        for a in range(A):
            for a2 in range(A):
                di = pos[:, a2, 0] - pos[:, a, 0] + c
                dj = pos[:, a2, 1] - pos[:, a, 1] + c
                inside = (di >= 0) & (di < k) & (dj >= 0) & (dj < k)
                assert torch.equal(part[:, a, a2].sum((-1, -2)).to(torch.int64), inside.to(torch.int64))
                assert torch.all(part[e[inside], a, a2, di[inside], dj[inside]] == 1)
    walk = bw.available_actions(True)
    assert torch.equal(walk, ((bw.avail.unsqueeze(-1) >> torch.arange(5, device="cuda")) & 1).to(torch.bool))
    strict = bw.available_actions(False)
    assert torch.all(walk | ~strict)                          # the laser filter only removes actions


def test_perspective_output_beyond_4_GiB():
    """Maximum sizes: the perspective tensor of config 5 at 65 536 envs is 10.7 GB -- (env, agent) row offsets pass 2^32 in
    the kernel.  The relation to the layered tensor (agent k's view = layers 0 <-> k and A <-> A + k swapped) on every env."""
    import torch

    from lle_amd import BatchedWorld, _capi, mapgen

    free, _ = torch.cuda.mem_get_info()
    if free < 24 << 30:
        pytest.skip("needs 24 GB of free device memory")
    n = 65536
    bw = BatchedWorld(mapgen.config5(0), n)
    for t in range(4):
        bw.step(sample=True, auto_reset=True, seed=8, t=t)
    A, H, W = bw.map.n_agents, bw.map.height, bw.map.width
    persp = bw.observe_as(_capi.LLE_OBS_PERSPECTIVE)
    assert persp.shape == (n, A, 2 * A + 4, H, W) and persp.numel() > 1 << 33
    layered = bw.obs
    for k in range(A):
        perm = list(range(2 * A + 4))
        perm[0], perm[k] = perm[k], perm[0]
        perm[A], perm[A + k] = perm[A + k], perm[A]
        for lo in range(0, n, 16384):
            assert torch.equal(persp[lo:lo + 16384, k], layered[lo:lo + 16384][:, perm]), (k, lo)


def test_unsupported_colour_raises_index_error():
    """A laser colour without a layer: IndexError like the reference (python/lle/observations.py:229, 259, 356)."""
    from lle_amd import BatchedWorld, _capi

    bw = BatchedWorld("S0 . X\nL7E . .", 8)
    with pytest.raises(IndexError):
        bw.observe_as(_capi.LLE_OBS_PARTIAL, 3)
    with pytest.raises(IndexError):
        bw.observe_as(_capi.LLE_OBS_LAYERED)
    assert bw.observe_as(_capi.LLE_OBS_LAYERED_PADDED, 4).shape == (8, 14, 2, 3)  # 5 agent + 9 laser-and-fixed layers: colour 7 fits
    assert bw.observe_as(_capi.LLE_OBS_STATE).shape == (8, 3)


@pytest.mark.parametrize("project", ["0", "1", "lanes", "lanes:E=1", "lanes:E=2", "lanes:E=4,B=2", "lanes:WT"])
@pytest.mark.parametrize("name", ["level6", "nested", "colour_alias", "four_layers", "many_agents", "config5_32x32"])
def test_partial_window_and_projection_kernels_agree_with_the_oracle(oracle_mod, monkeypatch, name, project):
    """The partial k x k observation has three kernels -- one lane per (env, observer) over the non-empty cells of its window
    (partial_lanes_kernel, the default), per window cell (partial_observe_kernel) and per (entity, observer) pair
    (partial_project_kernel).  Force each (LLE_PARTIAL_PROJECT for the last two; the lane kernel also with few environments
    per batch, i.e. several lanes per observer, several batches per wavefront and written-through stores) on every map,
    window sizes 3 to 15, along a rollout with deaths, also with per-env source colours: all must match oracle/observers.py."""
    import torch

    from lle_amd import BatchedWorld, _capi
    from oracle import observers as oo
    from tests.parity_util import legal_colours

    if project in ("0", "1"):
        monkeypatch.setenv("LLE_PARTIAL_PROJECT", project)
    else:
        monkeypatch.delenv("LLE_PARTIAL_PROJECT", raising=False)
        for opt in project.split(":")[1].split(",") if ":" in project else []:
            key, _, val = opt.partition("=")
            monkeypatch.setenv({"E": "LLE_PARTIAL_E", "B": "LLE_PARTIAL_BATCHES", "WT": "LLE_PARTIAL_WT"}[key], val or "1")
    all_sizes = (3, 5, 7) if project in ("0", "1") else (3, 5, 7, 9, 15)
    text = MAPS[name]
    n = 200
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    A, L = bw.map.n_agents, bw.map.n_sources
    sizes = [k for k in all_sizes if bw.obs_desc(_capi.LLE_OBS_PARTIAL, k).supported]
    rng = np.random.default_rng(5)
    for t in range(12):
        if t == 6 and L:  # per-environment colours from here on
            colours = legal_colours(bw.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
            bw.set_sources(torch.from_numpy(colours))
            for e in range(n):
                for l in range(L):
                    ob.world(e).set_source(l, colour=int(colours[e, l]))
            sizes = [k for k in all_sizes if bw.obs_desc(_capi.LLE_OBS_PARTIAL, k).supported]
        bw.step(sample=True, auto_reset=(t % 4 == 3), seed=9, t=t)
        ob.step(None, auto_reset=(t % 4 == 3), seed=9, t=t)
        for k in sizes:
            got = bw.observe_as(_capi.LLE_OBS_PARTIAL, k).cpu().numpy()
            for e in range(0, n, 17):
                want = oo.partial_observe(ob.world(e), k)
                assert np.array_equal(got[e].astype(np.float32), want), (name, project, t, k, e)


@pytest.mark.parametrize("name", ["level6", "level1", "level5", "nested", "three_beams", "four_layers", "colour_alias", "many_agents", "gen_16x16_12agents"])
def test_partial_written_by_the_step_launch(oracle_mod, name):
    """lle_batch_step_outputs with `partial` set: the step launch writes the partial k x k observation itself (step kernel MODE 9,
    partial_stream.hpp) from the state machine's records.  Same bytes as lle_batch_observe_as(LLE_OBS_PARTIAL) behind a plain step and
    as the oracle's restatement of python/lle/observations.py:312-369, on a ragged batch, every window size the kernel serves, with
    auto-reset, given actions and refused actions; the other fused outputs unchanged."""
    import torch

    from lle_amd import BatchedWorld, _capi
    from oracle import observers as oo

    text = MAPS[name]
    n = 300 + 5
    a, b = BatchedWorld(text, n), BatchedWorld(text, n)
    ob = oracle_mod.OracleBatch(text, n)
    A = ob.A
    g = torch.Generator(device="cuda").manual_seed(4)
    served = 0
    for k in (3, 5, 7, 9, 15):
        try:
            a.obs_desc(_capi.LLE_OBS_PARTIAL, k)
            buf, view = a.partial_buffer(k)
        except IndexError:  # (colour_alias: a laser colour without a layer -- the reference raises too)
            continue
        state = torch.empty((n, 3 * A + a.map.n_gems), dtype=torch.float32, device="cuda")
        avail = torch.empty((n, A, 5), dtype=torch.uint8, device="cuda")
        out = a.make_env_outputs(state=state, available=avail, partial=buf, partial_k=k)
        try:
            a.step(sample=True, auto_reset=True, seed=6, t=1000 * k, env_out=out, write_obs=False)
        except RuntimeError as e:
            # refused, loudly: more than 8 beam words (many_agents: 14 sources), or 16 lanes per env and a window above 8 x 8
            assert "partial observation" in str(e) and (a.map.n_beam_words > 8 or (A > 8 and k > 8)), (name, k, str(e))
            continue
        served += 1
        b.restore(a.snapshot()) if False else None
        a.reset(), b.reset(), ob.reset()
        for t in range(10):
            if t % 3 == 2:  # given actions, now and then a refused one
                acts = torch.multinomial(b.available_actions().reshape(-1, 5).float() + 1e-6, 1, generator=g).reshape(n, A).to(torch.uint8)
                a.step(acts, auto_reset=t >= 4, env_out=out, write_obs=False)
                b.step(acts, auto_reset=t >= 4)
                ob.step(acts.cpu().numpy(), auto_reset=t >= 4, want_obs=False)
            else:
                a.step(sample=True, auto_reset=t >= 4, seed=6, t=t, env_out=out, write_obs=False)
                b.step(sample=True, auto_reset=t >= 4, seed=6, t=t)
                ob.step(None, auto_reset=t >= 4, seed=6, t=t, want_obs=False)
            want = b.observe_as(_capi.LLE_OBS_PARTIAL, k)
            assert torch.equal(view, want), (name, k, t)
            st = torch.empty_like(state)
            av = torch.empty_like(avail)
            b.env_outputs(state=st, available=av)
            assert torch.equal(state, st) and torch.equal(avail, av), (name, k, t)
            for e in (0, n // 2, n - 1):
                assert np.array_equal(view[e].cpu().numpy().astype(np.float32), oo.partial_observe(ob.world(e), k)), (name, k, t, e)
        assert torch.equal(a.pos, b.pos) and torch.equal(a.bits, b.bits) and torch.equal(a.beams, b.beams)
    assert served >= 3 or name == "colour_alias" or a.map.n_beam_words > 8, name  # (colour_alias: a colour without a layer in this observation -- IndexError, like the reference)


def test_batched_lle_partial_in_one_launch():
    """BatchedLLE(obs_type="partial7x7").step(fused=True): observation + state + reward + done + available actions from ONE launch;
    equal to the default (step + observer + outputs launches) along a rollout with auto-reset."""
    import torch

    from lle_amd import BatchedLLE

    for kw in (dict(obs_type="partial7x7"), dict(obs_type="partial3x3", state_type="normalized-state", multi_objective=True),
               dict(obs_type="partial5x5", state_type="partial5x5")):
        n = 640
        a, b = BatchedLLE(LEVELS[6], n, seed=1, **kw), BatchedLLE(LEVELS[6], n, seed=1, **kw)
        assert a._fused_partial
        a.reset(), b.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        for t in range(16):
            acts = torch.multinomial(b.available_actions().reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
            x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
            for key in ("obs", "state", "reward", "done", "available_actions", "err"):
                assert x[key].shape == y[key].shape and torch.equal(x[key], y[key]), (kw, t, key)
    assert not BatchedLLE(LEVELS[6], 64, obs_type="partial7x7", randomize_lasers=True)._fused_partial  # (per-env sources: two launches)
