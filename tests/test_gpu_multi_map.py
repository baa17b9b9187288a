"""Batches of several maps (lle_batch_create_multi): map m owns a block of environments.  Every block is compared with
its own oracle batch on the same global action stream, bit-exact, through step / auto-reset / reset / set_state /
per-env sources / every observation builder."""
import numpy as np
import pytest

from tests.observer_checks import compare_all
from tests.parity_util import assert_state_equal, assert_step_equal, legal_colours, unpack_engine

pytestmark = pytest.mark.gpu


def _maps(k, **kw):
    from lle_amd import mapgen
    return [mapgen.generate(seed=100 + s, **kw) for s in range(k)]


def dims_of(ob):
    return (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)


def check_blocks(bw, obs, osteps, per, where):
    bufs = bw.host_buffers()
    for m, ob in enumerate(obs):
        sl = slice(m * per, (m + 1) * per)
        eng = unpack_engine({k: v[sl] for k, v in bufs.items()}, *dims_of(ob))
        if osteps is not None:
            assert_step_equal(eng, osteps[m], f"{where} map {m}")
        assert_state_equal(eng, ob.dump(), f"{where} map {m}")


@pytest.mark.parametrize("shape", [dict(height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2),
                                   dict(height=16, width=16, n_agents=6, n_lasers=7, n_gems=5, n_voids=3),
                                   dict(height=6, width=7, n_agents=1, n_lasers=2, n_gems=2, n_voids=1)])
def test_blocks_of_maps_match_their_oracles(oracle_mod, shape):
    import torch

    from lle_amd import BatchedWorld

    texts = _maps(5, **shape)
    per = 192  # a multiple of 64, not of 256: workgroups are narrowed so that none straddles two maps
    n = per * len(texts)
    bw = BatchedWorld(texts, n)
    obs = [oracle_mod.OracleBatch(t, per) for t in texts]
    check_blocks(bw, obs, None, per, "after creation")
    for t in range(30):
        auto = t >= 10
        bw.step(sample=True, auto_reset=auto, seed=8, t=t, env_offset=1000)
        osteps = [ob.step(None, auto_reset=auto, seed=8, t=t, env_offset=1000 + m * per) for m, ob in enumerate(obs)]
        check_blocks(bw, obs, osteps, per, f"t={t}")
    # masked reset, then the fused rollout against single steps of a twin
    rng = np.random.default_rng(1)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    bw.reset(torch.from_numpy(mask))
    for m, ob in enumerate(obs):
        for e in np.nonzero(mask[m * per:(m + 1) * per])[0]:
            ob.world(int(e)).reset()
    check_blocks(bw, obs, None, per, "after masked reset")
    twin = BatchedWorld(texts, n)
    twin.restore(bw.snapshot())
    twin.rollout(5, auto_reset=True, seed=3, t=50)
    for t in range(50, 55):
        bw.step(sample=True, auto_reset=True, seed=3, t=t)
    for k in ("pos", "bits", "gems", "beams", "avail", "obs"):
        assert torch.equal(getattr(bw, k), getattr(twin, k)), k


def test_observers_and_per_env_sources_on_blocks_of_maps(oracle_mod):
    import torch

    from lle_amd import BatchedWorld

    shape = dict(height=8, width=9, n_agents=3, n_lasers=3, n_gems=2, n_voids=1)
    texts = _maps(4, **shape)
    per = 64
    n = per * len(texts)
    bw = BatchedWorld(texts, n)
    obs = [oracle_mod.OracleBatch(t, per) for t in texts]
    for t in range(8):
        bw.step(sample=True, seed=4, t=t)
        for m, ob in enumerate(obs):
            ob.step(None, seed=4, t=t, env_offset=m * per, want_obs=False)

    def engine(m):
        def observe(kind, param):
            try:
                out = bw.observe_as(kind, param)
            except IndexError:
                return None
            torch.cuda.synchronize()
            return out[m * per:(m + 1) * per].cpu().numpy()

        def avail(walkable):
            out = bw.available_actions(walkable)
            torch.cuda.synchronize()
            return out[m * per:(m + 1) * per].cpu().numpy()
        return observe, avail

    for m, ob in enumerate(obs):
        compare_all(*engine(m), ob, range(0, per, 13), f"map {m}")

    # per-environment sources on top: random colours / flags for every env of every map
    A, L = obs[0].A, bw.map.n_sources
    rng = np.random.default_rng(2)
    colours = legal_colours(bw.maps, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    enabled = rng.integers(0, 1 << L, size=n).astype(np.int32)
    bw.set_sources(torch.from_numpy(colours), torch.from_numpy(enabled))
    for m, ob in enumerate(obs):
        was = [bool(s[4]) for s in ob.world(0).sources()]
        for e in range(per):
            w = ob.world(e)
            for l in range(L):
                w.set_source(l, colour=int(colours[m * per + e, l]))
                want = bool((int(enabled[m * per + e]) >> l) & 1)
                if want != was[l]:
                    w.set_source(l, enabled=want)
    check_blocks(bw, obs, None, per, "after set_sources")
    for t in range(20, 30):
        bw.step(sample=True, auto_reset=True, seed=4, t=t)
        osteps = [ob.step(None, auto_reset=True, seed=4, t=t, env_offset=m * per) for m, ob in enumerate(obs)]
        check_blocks(bw, obs, osteps, per, f"per-env sources t={t}")
    for m, ob in enumerate(obs):
        compare_all(*engine(m), ob, range(0, per, 17), f"per-env sources map {m}")


def test_mismatched_maps_are_refused():
    from lle_amd import BatchedWorld

    with pytest.raises(RuntimeError, match="agree"):
        BatchedWorld(["S0 . X", "S0 . . X"], 128)
    with pytest.raises(Exception):  # (the envs do not divide among the maps)
        BatchedWorld(["S0 . X", "S0 X .", "S0 X ."], 100)


def test_env_outputs_on_blocks_of_maps():
    """lle_batch_env_outputs on a batch of several maps (each env's availability bools come from ITS map's tables),
    with per-env sources on top: equal to the separate entry points."""
    import torch

    from lle_amd import BatchedWorld, _capi

    texts = _maps(4, height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2)
    per = 192
    n = per * len(texts)
    w = BatchedWorld(texts, n)
    A, G, L = w.map.n_agents, w.map.n_gems, w.map.n_sources
    g = torch.Generator(device="cuda").manual_seed(1)
    for rnd in range(2):
        if rnd == 1:
            w.set_sources(colours=legal_colours(w.maps, torch.randint(0, A, (n, L), generator=g, device="cuda", dtype=torch.uint8)))
        for t in range(12):
            w.step(sample=True, auto_reset=(t % 4 == 3), seed=8, t=12 * rnd + t)
            for walkable in (False, True):
                state = torch.empty((n, 3 * A + G), device="cuda")
                reward = torch.empty((n, 4), device="cuda")
                done, avail = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty((n, A, 5), dtype=torch.uint8, device="cuda")
                w.env_outputs(state=state, normalize_state=True, reward=reward, multi_objective=True, done=done, available=avail,
                              walkable_lasers=walkable)
                assert torch.equal(state.view(torch.int32), w.observe_as(_capi.LLE_OBS_NORMALIZED_STATE, 0).view(torch.int32)), (rnd, t)
                assert torch.equal(reward, w.reward_multi_objective()) and torch.equal(done, w.done), (rnd, t)
                assert torch.equal(avail.view(torch.bool), w.available_actions(walkable)), (rnd, t, walkable)


@pytest.mark.parametrize("n_maps,per", [(1024, 64), (256, 16), (96, 48), (512, 8), (33, 24), (2048, 1), (700, 3), (512, 4)])
def test_one_map_per_block_at_scale(oracle_mod, n_maps, per):
    """SURVEY.md section 8(d), stretch variant of config 5: per-env distinct maps -- what a learner on generated maps trains on
    (python/lle/generator/world_builder.py:84-89).  1 024 distinct `mapgen.config5(seed)` maps x 64 envs (the bench's
    `cfg5_multi_map` block), 256 x 16, 96 x 48, and -- since round 5 -- 512 x 8 (ONE wavefront per map), 33 x 24, and 2 048 x 1 (a map per ENVIRONMENT: four wavefronts
    share one env's split row), 700 x 3, 512 x 4: envs_per_map may be anything: every block
    against its own oracle batch on the global action stream -- state after every step, the full check (events, observation) on a
    sample of the blocks."""
    from lle_amd import BatchedWorld, mapgen

    texts = [mapgen.config5(seed) for seed in range(n_maps)]
    n = n_maps * per
    bw = BatchedWorld(texts, n)
    sample = sorted(set([0, 1, n_maps // 2, n_maps - 1] + list(range(0, n_maps, max(1, n_maps // 24)))))
    obs = {m: oracle_mod.OracleBatch(texts[m], per) for m in sample}

    def check(osteps, where):
        bufs = bw.host_buffers()
        for m, ob in obs.items():
            sl = slice(m * per, (m + 1) * per)
            eng = unpack_engine({k: v[sl] for k, v in bufs.items()}, *ob.dims)
            if osteps is not None:
                assert_step_equal(eng, osteps[m], f"{where} map {m}")
            assert_state_equal(eng, ob.dump(), f"{where} map {m}")
    check(None, "after creation")
    for t in range(12):
        auto = t >= 4
        bw.step(sample=True, auto_reset=auto, seed=8, t=t, env_offset=5)
        check({m: ob.step(None, auto_reset=auto, seed=8, t=t, env_offset=5 + m * per) for m, ob in obs.items()}, f"t={t}")
    bw.rollout(4, auto_reset=True, seed=8, t=12, env_offset=5)
    for t in range(12, 16):
        osteps = {m: ob.step(None, auto_reset=True, seed=8, t=t, env_offset=5 + m * per) for m, ob in obs.items()}
    check(osteps, "fused rollout")
    st = bw.stats()
    assert st["env_steps"] == 16 * n and st["invalid"] == 0


@pytest.mark.parametrize("per", [8, 16, 64])
@pytest.mark.parametrize("packed", ["0", "1"])
def test_packed_table_image_either_way(oracle_mod, monkeypatch, per, packed):
    """tables.h off_packed: the split-row launches of a multi-map batch expand the packed image of the table section (16-bit cell words, the layer words
    of the cells under a beam scattered over zeros) instead of copying it -- by default where a map's block fills a four-wavefront workgroup.  Both
    ways forced (LLE_PACKED_TABLES) on blocks of 8 (two wavefronts per workgroup), 16 and 64: single steps, the fused outputs, a rollout; every
    block against its own oracle batch, observation included."""
    import torch

    from lle_amd import BatchedWorld, _capi, mapgen

    monkeypatch.setenv("LLE_PACKED_TABLES", packed)
    _capi.refresh_tuning()
    try:
        n_maps = 40
        texts = [mapgen.config5(100 + s) for s in range(n_maps)]
        n = n_maps * per
        bw = BatchedWorld(texts, n)
        obs = [oracle_mod.OracleBatch(t, per) for t in texts]
        check_blocks(bw, obs, None, per, "after creation")
        done = torch.empty(n, dtype=torch.uint8, device="cuda")
        eo = bw.make_env_outputs(done=done)
        for t in range(10):
            auto = t >= 3
            bw.step(sample=True, auto_reset=auto, seed=4, t=t, env_offset=7, env_out=eo if t % 3 == 2 else None)
            osteps = [ob.step(None, auto_reset=auto, seed=4, t=t, env_offset=7 + m * per) for m, ob in enumerate(obs)]
            check_blocks(bw, obs, osteps, per, f"packed={packed} t={t}")
        bw.rollout(4, auto_reset=True, seed=4, t=10, env_offset=7)
        for t in range(10, 14):
            osteps = [ob.step(None, auto_reset=True, seed=4, t=t, env_offset=7 + m * per) for m, ob in enumerate(obs)]
        check_blocks(bw, obs, osteps, per, f"packed={packed} fused rollout")
    finally:
        monkeypatch.delenv("LLE_PACKED_TABLES")
        _capi.refresh_tuning()


@pytest.mark.parametrize("shape", [dict(height=9, width=11, n_agents=3, n_lasers=4, n_gems=3, n_voids=2),
                                   dict(height=6, width=7, n_agents=1, n_lasers=2, n_gems=2, n_voids=1),
                                   dict(height=12, width=12, n_agents=12, n_lasers=6, n_gems=4, n_voids=2)])
@pytest.mark.parametrize("per", [8, 1, 2, 3, 5, 12])
def test_few_envs_per_map(oracle_mod, shape, per):
    """Round 5: a map may own ANY number of environments, down to one map per environment (the reference hands every env its own map,
    python/lle/generator/world_builder.py:84-89; 16 was the floor until round 5, then 8).  Lane groups of 1, 4 and 16 -- a wavefront serves one map,
    so it takes the largest power of two of environments that divides the block (1 for odd blocks), and LLE_BUF_STATS has a counter slot per
    wavefront --: single steps, a fused rollout, masked reset, every observation builder and per-environment sources, each block against its own
    oracle batch."""
    import torch

    from lle_amd import BatchedWorld

    texts = _maps(9, **shape)
    n = per * len(texts)
    bw = BatchedWorld(texts, n)
    obs = [oracle_mod.OracleBatch(t, per) for t in texts]
    check_blocks(bw, obs, None, per, "after creation")
    for t in range(24):
        auto = t >= 8
        bw.step(sample=True, auto_reset=auto, seed=8, t=t, env_offset=100)
        osteps = [ob.step(None, auto_reset=auto, seed=8, t=t, env_offset=100 + m * per) for m, ob in enumerate(obs)]
        check_blocks(bw, obs, osteps, per, f"t={t}")
    bw.rollout(5, auto_reset=True, seed=8, t=24, env_offset=100)
    for t in range(24, 29):
        osteps = [ob.step(None, auto_reset=True, seed=8, t=t, env_offset=100 + m * per) for m, ob in enumerate(obs)]
    check_blocks(bw, obs, osteps, per, "fused rollout")
    rng = np.random.default_rng(1)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    bw.reset(torch.from_numpy(mask))
    for m, ob in enumerate(obs):
        for e in np.nonzero(mask[m * per:(m + 1) * per])[0]:
            ob.world(int(e)).reset()
    check_blocks(bw, obs, None, per, "after masked reset")
    for t in range(4):
        bw.step(sample=True, seed=9, t=t)
        for m, ob in enumerate(obs):
            ob.step(None, seed=9, t=t, env_offset=m * per, want_obs=False)

    def engine(m):
        def observe(kind, param):
            try:
                out = bw.observe_as(kind, param)
            except IndexError:
                return None
            torch.cuda.synchronize()
            return out[m * per:(m + 1) * per].cpu().numpy()

        def avail(walkable):
            out = bw.available_actions(walkable)
            torch.cuda.synchronize()
            return out[m * per:(m + 1) * per].cpu().numpy()
        return observe, avail
    for m, ob in enumerate(obs):
        compare_all(*engine(m), ob, range(per), f"map {m}")
    A, L = obs[0].A, bw.map.n_sources
    colours = legal_colours(bw.maps, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    bw.set_sources(torch.from_numpy(colours))
    for m, ob in enumerate(obs):
        for e in range(per):
            for l in range(L):
                ob.world(e).set_source(l, colour=int(colours[m * per + e, l]))
    check_blocks(bw, obs, None, per, "after set_sources")
    for t in range(8):
        bw.step(sample=True, auto_reset=True, seed=4, t=t)
        osteps = [ob.step(None, auto_reset=True, seed=4, t=t, env_offset=m * per) for m, ob in enumerate(obs)]
        check_blocks(bw, obs, osteps, per, f"per-env sources t={t}")
    for m, ob in enumerate(obs):
        compare_all(*engine(m), ob, range(per), f"per-env sources map {m}")
