"""Parity proper: BatchedWorld (C ABI -> HIP kernels) vs the CPU oracle on identical action streams, bit-exact for
state, events (order included), availability masks and the int8 layered observation."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, LONG_MAPS, assert_state_equal, assert_step_equal, legal_colours, unpack_engine

pytestmark = pytest.mark.gpu

MAPS = {f"level{k}": v for k, v in LEVELS.items()}
MAPS.update(EXTRA_MAPS)


def dims_of(ob):
    return (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)


def check(bw, ob, ostep, where):
    eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
    if ostep is not None:
        assert_step_equal(eng, ostep, where)
    assert_state_equal(eng, ob.dump(), where)


@pytest.mark.parametrize("name", list(MAPS))
@pytest.mark.parametrize("auto_reset", [False, True])
def test_random_rollout(oracle_mod, name, auto_reset):
    from lle_amd import BatchedWorld

    text = MAPS[name]
    n, steps = 1000, 50  # not a multiple of 64: exercises the ragged last wave
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    check(bw, ob, None, f"{name} after reset")
    for t in range(steps):
        bw.step(sample=True, auto_reset=auto_reset, seed=1234, t=t, env_offset=3)
        ostep = ob.step(None, auto_reset=auto_reset, seed=1234, t=t, env_offset=3)
        check(bw, ob, ostep, f"{name} t={t}")


@pytest.mark.parametrize("epw", [8, 16, 32, 64])
def test_envs_per_wave_variants(oracle_mod, epw):
    from lle_amd import BatchedWorld

    n = 777
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(LEVELS[6], n, envs_per_wave=epw)
    for t in range(20):
        bw.step(sample=True, auto_reset=True, seed=5, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=5, t=t), f"epw={epw} t={t}")


def test_config2_level1_batch4096(oracle_mod):
    """BASELINE.json configs[1]: level 1, batch 4096, bit-exact over 256 steps (> 10^6 env-steps)."""
    from lle_amd import BatchedWorld

    n = 4096
    ob = oracle_mod.OracleBatch(LEVELS[1], n)
    bw = BatchedWorld(LEVELS[1], n)
    for t in range(256):
        bw.step(sample=True, auto_reset=True, seed=1234, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=1234, t=t), f"t={t}")


def test_config3_level6_batch65536(oracle_mod):
    """BASELINE.json configs[2] at full size: level 6, batch 65536 + layered obs, bit-exact for 32 steps
    (> 2 x 10^6 env-steps: state, ordered events, availability, the full int8 observation of every env, every step)."""
    from lle_amd import BatchedWorld

    n = 65536
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(LEVELS[6], n)
    for t in range(32):
        bw.step(sample=True, auto_reset=True, seed=1234, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=1234, t=t), f"t={t}")


@pytest.mark.parametrize("name", ["level6", "many_agents", "config5_32x32"])
def test_alternating_walk_changes_nothing(oracle_mod, monkeypatch, name):
    """Launches that rewrite an output larger than the Infinity Cache walk the environments alternately up and down
    (obs_stream.hpp xcd_block_dir; LLE_PINGPONG=1 forces it at any size, =0 switches it off): the order in which a launch serves
    its environments is not observable.  A ragged batch (the last wavefront partly empty, served FIRST on the way down),
    every buffer against the oracle along a rollout, and every observation builder against the one-directional run."""
    import torch

    from lle_amd import BatchedWorld, _capi

    text = MAPS[name] if name in MAPS else EXTRA_MAPS[name]
    n = 1000
    ob = oracle_mod.OracleBatch(text, n)
    monkeypatch.setenv("LLE_PINGPONG", "1")
    bw = BatchedWorld(text, n)
    monkeypatch.setenv("LLE_PINGPONG", "0")
    ref = BatchedWorld(text, n)
    kinds = [(_capi.LLE_OBS_LAYERED, 0), (_capi.LLE_OBS_LAYERED_PADDED, 2), (_capi.LLE_OBS_PERSPECTIVE, 0), (_capi.LLE_OBS_PARTIAL, 3),
             (_capi.LLE_OBS_PARTIAL, 7), (_capi.LLE_OBS_STATE, 0)]
    for t in range(9):
        monkeypatch.setenv("LLE_PINGPONG", "1")
        bw.step(sample=True, auto_reset=(t % 3 != 2), seed=19, t=t)
        check(bw, ob, ob.step(None, auto_reset=(t % 3 != 2), seed=19, t=t), f"{name} t={t}")
        got = [bw.observe_as(k, p).clone() for k, p in kinds for _ in (0, 1)]   # twice each: once up, once down
        bw.observe()
        rows_down = bw.obs_rows.clone()
        bw.observe()
        assert torch.equal(bw.obs_rows, rows_down)
        monkeypatch.setenv("LLE_PINGPONG", "0")
        ref.step(sample=True, auto_reset=(t % 3 != 2), seed=19, t=t)
        want = [ref.observe_as(k, p) for k, p in kinds for _ in (0, 1)]
        assert all(torch.equal(a, b) for a, b in zip(got, want)), t
        assert torch.equal(ref.obs_rows, rows_down)


def test_rows_beyond_4_GiB(oracle_mod):
    """Maximum sizes: config 5's map at 262 144 envs = 5.4 GB of observation rows in one launch, so row offsets pass 2^32
    (and 2^31) inside the kernel.  Windows of 64 envs -- the first, the ones astride the 2 GiB and 4 GiB offsets, the last --
    against oracle batches stepping the same global env ids; the rest by the size-independent properties."""
    import torch

    from lle_amd import BatchedWorld, mapgen
    from tests.parity_util import assert_state_equal, assert_step_equal, unpack_engine

    text = mapgen.config5(0)
    n = 262144
    free, _ = torch.cuda.mem_get_info()
    if free < 8 << 30:
        pytest.skip("needs 8 GB of free device memory")
    bw = BatchedWorld(text, n)
    pitch = bw.map.obs_stride
    assert n * pitch > 1 << 32
    starts = sorted({0, (1 << 31) // pitch - 32, (1 << 32) // pitch - 32, n - 64})
    obs = [oracle_mod.OracleBatch(text, 64) for _ in starts]
    names = ("pos", "bits", "gems", "beams", "avail", "actions", "err", "evcount", "events", "done")
    for t in range(3):
        bw.step(sample=True, auto_reset=True, seed=41, t=t)
        torch.cuda.synchronize()
        for lo, ob in zip(starts, obs):
            ostep = ob.step(None, auto_reset=True, seed=41, t=t, env_offset=lo)
            bufs = {k: getattr(bw, k)[lo:lo + 64].cpu().numpy() for k in names}
            bufs["obs"] = bw.obs_rows[lo:lo + 64].cpu().numpy()
            for k, dt in (("bits", np.uint64), ("gems", np.uint32), ("beams", np.uint32)):
                bufs[k] = np.ascontiguousarray(bufs[k]).view(dt)
            eng = unpack_engine(bufs, ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
            assert_step_equal(eng, ostep, f"envs {lo}.. t={t}")
            assert_state_equal(eng, ob.dump(), f"envs {lo}.. t={t}")
    A, H, W = bw.map.n_agents, bw.map.height, bw.map.width
    for lo in range(0, n, 32768):  # every env: one cell per agent layer, and it is the agent's cell
        sl = slice(lo, lo + 32768)
        layers = bw.obs[sl, :A].reshape(32768, A, H * W)
        cell = bw.pos[sl].to(torch.int64)
        cell = cell[..., 0] * W + cell[..., 1]
        assert torch.equal(layers.sum(-1).to(torch.int64), torch.ones(32768, A, dtype=torch.int64, device="cuda")), lo
        assert torch.all(layers.gather(2, cell.unsqueeze(-1)) == 1), lo


@pytest.mark.parametrize("rank", [3, 7])
def test_config4_one_shard_of_the_524288_env_job(oracle_mod, rank):
    """BASELINE.json configs[3]: level 6 x 524 288 envs sharded over 8 GPUs = 65 536 envs per rank, rank r sampling for the
    global env ids [r * 65 536, (r + 1) * 65 536) (lle_amd/distributed.py shard_range -> env_offset).  No 8-GPU box here:
    the shard of one rank on the one GPU, every buffer against the oracle stepping the same global ids."""
    from lle_amd import BatchedWorld
    from lle_amd.distributed import shard_range

    lo, hi = shard_range(524288, rank, 8)
    assert (lo, hi) == (rank * 65536, (rank + 1) * 65536)
    n = hi - lo
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(LEVELS[6], n)
    for t in range(8):
        bw.step(sample=True, auto_reset=True, seed=77, t=t, env_offset=lo)
        check(bw, ob, ob.step(None, auto_reset=True, seed=77, t=t, env_offset=lo), f"rank {rank} t={t}")


def test_explicit_and_invalid_actions(oracle_mod):
    import torch

    from lle_amd import BatchedWorld

    n = 512
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    bw = BatchedWorld(LEVELS[6], n)
    rng = np.random.default_rng(0)
    for t in range(40):
        actions = rng.integers(0, 6, size=(n, ob.A), dtype=np.uint8)  # unavailable and out-of-range (5) included
        bw.step(torch.from_numpy(actions).cuda())
        check(bw, ob, ob.step(actions), f"t={t}")


def test_masked_reset(oracle_mod):
    import torch

    from lle_amd import BatchedWorld

    n = 300
    ob = oracle_mod.OracleBatch(LEVELS[5], n)
    bw = BatchedWorld(LEVELS[5], n)
    for t in range(15):
        bw.step(sample=True, seed=9, t=t)
        ob.step(None, seed=9, t=t)
    mask = (np.arange(n) % 3 == 0)
    bw.reset(torch.from_numpy(mask.astype(np.uint8)).cuda())
    for e in np.nonzero(mask)[0]:
        ob.world(int(e)).reset()
    check(bw, ob, None, "after masked reset")


def test_batched_set_state(oracle_mod):
    """World.set_state semantics (lossy, with the reference's rollback rules) on random requests."""
    import torch

    from lle_amd import BatchedWorld

    text = EXTRA_MAPS["nested"]
    n = 600
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    rng = np.random.default_rng(1)
    for rnd in range(6):
        for t in range(5):
            bw.step(sample=True, seed=rnd, t=t)
            ob.step(None, seed=rnd, t=t)
        pos = np.stack([rng.integers(0, ob.H + (rnd == 5), size=(n, ob.A)), rng.integers(0, ob.W, size=(n, ob.A))], axis=-1).astype(np.uint8)
        gems = rng.integers(0, 2, size=(n, ob.G)).astype(bool)
        alive = rng.integers(0, 4, size=(n, ob.A)) > 0
        bw.set_state(torch.from_numpy(pos).cuda(), torch.from_numpy(gems).cuda(), torch.from_numpy(alive).cuda())
        host = bw.host_buffers()
        codes = {0: 0, -4: 0x40, -5: 0x41, -6: 0x42}
        for e in range(n):
            w = ob.world(e)
            try:
                ev = w.set_state([tuple(int(v) for v in p) for p in pos[e]], list(gems[e]), list(alive[e]))
                rc = 0
            except oracle_mod.OracleError as ex:
                rc = {"InvalidWorldState": 0x40, "OutOfWorldPosition": 0x41, "InvalidAgentPosition": 0x42}[ex.kind]
                ev = []
            assert int(host["err"][e]) == rc, (e, int(host["err"][e]), rc)
            cnt = int(host["evcount"][e]) & 0x7F
            got = [(int(b) >> 4, int(b) & 15) for b in host["events"][e][:cnt]]
            assert got == ev, (e, got, ev)
        check(bw, ob, None, f"set_state round {rnd}")
        bw.observe()
        eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
        obs = np.stack([ob.world(e).obs() for e in range(n)])
        assert np.array_equal(eng["obs"], obs)
        # a set_state that failed with InvalidWorldState leaves the reference with stale availability lists (its next
        # step may index out of the grid and panic): such worlds are reset before the rollout goes on
        poisoned = host["err"] == 0x40
        bw.reset(torch.from_numpy(poisoned.astype(np.uint8)).cuda())
        for e in np.nonzero(poisoned)[0]:
            ob.world(int(e)).reset()


def test_update_sources(oracle_mod):
    from lle_amd import BatchedWorld

    text = EXTRA_MAPS["three_beams"]
    n = 200
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    for t in range(6):
        bw.step(sample=True, seed=2, t=t)
        ob.step(None, seed=2, t=t)
    for (lid, en, col) in [(0, False, None), (1, None, 0), (0, True, None), (2, False, 2)]:
        bw.map.set_source(lid, enabled=en, agent_id=col)
        bw.update_sources()
        for e in range(n):
            ob.world(e).set_source(lid, enabled=en, colour=col)
        check(bw, ob, None, f"after set_source {lid}")
        eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
        obs = np.stack([ob.world(e).obs() for e in range(n)])
        assert np.array_equal(eng["obs"], obs)
        for t in range(4):
            bw.step(sample=True, seed=3 + lid, t=t)
            check(bw, ob, ob.step(None, seed=3 + lid, t=t), f"steps after set_source {lid}")


def test_full_size_properties():
    """Size-independent properties at BASELINE's full batch (65536): determinism, shard invariance (env_offset),
    observation consistent with the state it was built from, counters equal to the events emitted."""
    import torch

    from lle_amd import BatchedWorld

    n = 65536
    a = BatchedWorld(LEVELS[6], n)
    lo = BatchedWorld(LEVELS[6], n // 2)
    hi = BatchedWorld(LEVELS[6], n // 2)
    ev_total = torch.zeros(3, dtype=torch.int64, device="cuda")
    for t in range(30):
        a.step(sample=True, auto_reset=True, seed=77, t=t)
        lo.step(sample=True, auto_reset=True, seed=77, t=t, env_offset=0)
        hi.step(sample=True, auto_reset=True, seed=77, t=t, env_offset=n // 2)
        cnt = (a.evcount & 0x7F).to(torch.int64)
        valid = torch.arange(a.events.shape[1], device="cuda")[None, :] < cnt[:, None]
        ty = (a.events >> 4).to(torch.int64)
        for k in range(3):
            ev_total[k] += ((ty == k) & valid).sum()
    for name in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "obs"):
        whole = getattr(a, name)
        parts = torch.cat([getattr(lo, name), getattr(hi, name)], 0)
        assert torch.equal(whole, parts), name
    # observation vs state: agent layers hold exactly one 1 at the agent's cell
    A, H, W = a.map.n_agents, a.map.height, a.map.width
    pos = a.pos.to(torch.int64)
    cell = pos[..., 0] * W + pos[..., 1]
    agent_layers = a.obs[:, :A].reshape(n, A, H * W)
    assert torch.equal(agent_layers.sum(-1).to(torch.int64), torch.ones(n, A, dtype=torch.int64, device="cuda"))
    assert torch.all(agent_layers.gather(2, cell.unsqueeze(-1)) == 1)
    st = a.stats()
    assert st["env_steps"] == 30 * n and st["agent_steps"] == 30 * n * A
    assert (st["exits"], st["gems"], st["deaths"]) == tuple(int(v) for v in ev_total.tolist())


def test_world_facade_pickle_and_deepcopy():
    import copy
    import pickle
    import random

    from lle_amd import World

    random.seed(0)
    for lvl in range(1, 7):  # python/tests/test_serialization.py:19-38
        world = World.level(lvl)
        world.reset()
        for _ in range(10):
            world.step([random.choice(a) for a in world.available_actions()])
            clone = pickle.loads(pickle.dumps(world))
            assert clone.get_state() == world.get_state()
            assert clone.wall_pos == world.wall_pos and clone.exit_pos == world.exit_pos
        assert copy.deepcopy(world).get_state() == world.get_state()


def test_config5_generated_32x32(oracle_mod):
    """BASELINE.json configs[4]: generated 32x32 map, 8 agents, 8 lasers (crossing beams), bit-exact at n=4096, and
    the size-independent properties at the full batch of 65536."""
    import torch

    from lle_amd import BatchedWorld, mapgen

    text = mapgen.config5(0)
    n = 4096
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    assert bw.kernel_info()["kernel"] == "step_kernel<8,8>"
    for t in range(30):
        bw.step(sample=True, auto_reset=(t % 2 == 0), seed=11, t=t)
        check(bw, ob, ob.step(None, auto_reset=(t % 2 == 0), seed=11, t=t), f"t={t}")
    del bw, ob
    n = 65536
    a, b = BatchedWorld(text, n), BatchedWorld(text, n, envs_per_wave=8)
    for t in range(10):
        a.step(sample=True, auto_reset=True, seed=5, t=t)
        b.step(sample=True, auto_reset=True, seed=5, t=t)
    for name in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "obs"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    A, H, W = a.map.n_agents, a.map.height, a.map.width
    pos = a.pos.to(torch.int64)
    cell = pos[..., 0] * W + pos[..., 1]
    agent_layers = a.obs[:, :A].reshape(n, A, H * W)
    assert torch.equal(agent_layers.sum(-1).to(torch.int64), torch.ones(n, A, dtype=torch.int64, device="cuda"))
    assert torch.all(agent_layers.gather(2, cell.unsqueeze(-1)) == 1)


def test_reward_counts_and_snapshot(oracle_mod):
    """Reward epilogue (per-step gem/exit/death counts + all-arrived flag) against the oracle's events, and the exact
    snapshot/restore of the dynamic state mid-episode (corpses and stale beams included)."""
    import torch

    from lle_amd import BatchedWorld

    text = EXTRA_MAPS["q1"]
    n = 700
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    snap = None
    for t in range(25):
        bw.step(sample=True, auto_reset=(t > 12), seed=21, t=t)
        o = ob.step(None, auto_reset=(t > 12), seed=21, t=t)
        cnt = (o["ev_count"] & 0x7F).astype(np.int64)
        valid = np.arange(o["events"].shape[1])[None, :] < cnt[:, None]
        ty = o["events"][:, :, 0]
        want = np.stack([((ty == 1) & valid).sum(1), ((ty == 0) & valid).sum(1), ((ty == 2) & valid).sum(1),
                         ob.dump()["arrived"].all(1).astype(np.int64)], axis=1)
        got = bw.reward.cpu().numpy().astype(np.int64)
        assert np.array_equal(got, want), t
        single = bw.reward_single_objective().cpu().numpy()
        assert np.array_equal(single, want[:, 0] + want[:, 1] - want[:, 2] + want[:, 3])
        multi = bw.reward_multi_objective().cpu().numpy()
        assert np.all(multi[want[:, 2] > 0][:, [0, 1, 3]] == 0)
        if t == 9:
            snap = bw.snapshot()
            ref = {k: v.copy() for k, v in bw.host_buffers().items()}
    bw.restore(snap)
    now = bw.host_buffers()
    for k in ("pos", "bits", "gems", "beams", "avail", "obs"):
        assert np.array_equal(now[k], ref[k]), k


@pytest.mark.parametrize("name", ["level6", "nested", "many_agents", "gen_20_lasers", "gen_16x16_12agents"])
def test_fused_rollout_equals_single_steps(oracle_mod, name):
    """lle_batch_rollout: T steps in one launch == T single steps of the oracle, per-step outputs in the rings."""
    from lle_amd import BatchedWorld

    text = MAPS[name]
    n, T, R = 900, 12, 5
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    ring = bw.make_ring(R)
    t0 = 0
    for chunk in (T, 7):
        bw.rollout(chunk, auto_reset=True, seed=99, t=t0, env_offset=11, ring=ring, ring_pos=t0)
        steps = [ob.step(None, auto_reset=True, seed=99, t=t0 + j, env_offset=11) for j in range(chunk)]
        eng = unpack_engine(bw.host_buffers(), *dims_of(ob))
        assert_state_equal(eng, ob.dump(), f"{name} after rollout of {chunk}")
        for key in ("err", "ev_count"):  # events / err / evcount reflect the last step
            assert np.array_equal(eng[key], steps[-1][key]), key
        obs_ring = ring["obs"].cpu().numpy()
        act_ring = ring["actions"].cpu().numpy()
        rew_ring = ring["reward"].cpu().numpy()
        for j in range(max(0, chunk - R), chunk):  # the slots that were not overwritten within this chunk
            slot = (t0 + j) % R
            assert np.array_equal(obs_ring[slot], steps[j]["obs"]), (name, j)
            assert np.array_equal(act_ring[slot], steps[j]["actions"]), (name, j)
            cnt = (steps[j]["ev_count"] & 0x7F).astype(np.int64)
            valid = np.arange(steps[j]["events"].shape[1])[None, :] < cnt[:, None]
            ty = steps[j]["events"][:, :, 0]
            assert np.array_equal(rew_ring[slot][:, 0], ((ty == 1) & valid).sum(1)), (name, j)
            assert np.array_equal(rew_ring[slot][:, 2], ((ty == 2) & valid).sum(1)), (name, j)
        t0 += chunk


def test_fuzz_maps_gpu(oracle_mod):
    """The random small maps of tests/test_hostsim_parity.py::test_fuzz_maps through the HIP kernels (every lane-group
    size G = 1, 2, 4, 8 occurs)."""
    from lle_amd import BatchedWorld
    from tests.test_hostsim_parity import _fuzz_maps

    groups = set()
    for name, text in _fuzz_maps():
        n = 333
        ob = oracle_mod.OracleBatch(text, n)
        bw = BatchedWorld(text, n)
        groups.add(bw.kernel_info()["kernel"])
        for t in range(30):
            ar = t % 3 != 0
            bw.step(sample=True, auto_reset=ar, seed=5, t=t)
            check(bw, ob, ob.step(None, auto_reset=ar, seed=5, t=t), f"{name} t={t}")
    assert len(groups) >= 3, groups


def test_large_batch_byte_offsets_beyond_32_bits():
    """2.5 million envs = 4.7 GB of observations in one batch (a fraction of the 288 GB of HBM): a window of the big
    batch equals a 65 536-env batch stepped with the matching env_offset, on every buffer."""
    import torch

    from lle_amd import BatchedWorld, Map

    m = Map(level=6)
    first = 2_400_000
    big = BatchedWorld(m, 2_500_000)
    small = BatchedWorld(m, 65536)
    for t in range(6):
        big.step(sample=True, auto_reset=True, seed=3, t=t)
        small.step(sample=True, auto_reset=True, seed=3, t=t, env_offset=first)
    sl = slice(first, first + 65536)
    for k in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "done", "obs"):
        assert torch.equal(getattr(big, k)[sl], getattr(small, k)), k
    assert big.stats()["env_steps"] == 6 * 2_500_000


@pytest.mark.parametrize("policy", ["0", "1"])
def test_store_policies_agree_with_oracle(oracle_mod, policy, monkeypatch):
    """The observation rows are stored written-through (`sc1`) or plain depending on the bytes a launch writes
    (obs_stream.hpp: stream_store).  Force each policy (LLE_WRITE_THROUGH) on short rows (level 6: 1 872 B), on rows
    longer than 2 KiB (12 agents on 16x16: 7 168 B, the 4-deep stream_row path) and on the per-env-sources stream."""
    import torch

    from lle_amd import BatchedWorld

    monkeypatch.setenv("LLE_WRITE_THROUGH", policy)
    for name, n in (("level6", 2048), ("gen_16x16_12agents", 1000)):
        text = MAPS[name]
        ob = oracle_mod.OracleBatch(text, n)
        bw = BatchedWorld(text, n)
        check(bw, ob, None, f"{name} policy={policy} after reset")
        for t in range(12):
            bw.step(sample=True, auto_reset=True, seed=77, t=t)
            check(bw, ob, ob.step(None, auto_reset=True, seed=77, t=t), f"{name} policy={policy} t={t}")
    # per-env sources: same colours in every env == the map-wide stream
    text = MAPS["level6"]
    a, b = BatchedWorld(text, 2048), BatchedWorld(text, 2048)
    L = a.map.n_sources
    cols = torch.tensor([s.agent_id for s in a.map.sources()], dtype=torch.uint8, device="cuda").repeat(2048, 1)
    assert cols.shape == (2048, L)
    b.set_sources(colours=cols)
    for t in range(8):
        a.step(sample=True, auto_reset=True, seed=9, t=t)
        b.step(sample=True, auto_reset=True, seed=9, t=t)
        assert torch.equal(a.obs, b.obs), t


def test_config5_plain_store_path(oracle_mod):
    """config5 rows (20 480 B) at 16 384 envs = 335 MB per launch: beyond the write-through threshold, so the long-row
    path with plain stores, bit-exact against the oracle."""
    from lle_amd import BatchedWorld, mapgen

    text = mapgen.config5(0)
    n = 16384
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    for t in range(4):
        bw.step(sample=True, auto_reset=True, seed=21, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=21, t=t), f"t={t}")


def test_config5_full_batch_against_the_oracle(oracle_mod):
    """BASELINE configs[4] at its full batch of 65 536 envs (1.34 GB of rows per launch), every buffer against the oracle
    (until round 3 only the soak runs did this; two steps keep it to seconds)."""
    from lle_amd import BatchedWorld, mapgen

    text = mapgen.config5(0)
    n = 65536
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    assert bw.kernel_info()["kernel"] == "step_kernel<8,8>"
    for t in range(2):
        bw.step(sample=True, auto_reset=True, seed=33, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=33, t=t), f"t={t}")


@pytest.mark.parametrize("name", ["level6", "corridor", "nested", "many_agents", "gen_20_lasers", "config5_32x32"])
@pytest.mark.parametrize("mode", ["per_env_sources", "two_maps", "two_maps_per_env_sources"])
def test_rollout_with_rings_equals_steps_in_every_general_mode(name, mode):
    """The step kernel's general instantiations come in pairs (step_kernel.hpp): MODE 4 / 5 for single steps, MODE 2 / 3
    with the rollout loop and the trajectory rings.  Same batch, same action stream: T single steps (each checked
    against the oracle elsewhere) == one fused rollout, on every buffer and every ring slot -- over group sizes 1-16
    and 0-20 sources."""
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n, T, R = 640, 9, 4
    maps = [text, text] if mode.startswith("two_maps") else text
    a, b = BatchedWorld(maps, n), BatchedWorld(maps, n)
    L, A = a.map.n_sources, a.map.n_agents
    if mode.endswith("per_env_sources") and L > 0:
        g = torch.Generator(device="cuda").manual_seed(5)
        colours = legal_colours(a.maps, torch.randint(0, A, (n, L), generator=g, device="cuda", dtype=torch.uint8))
        enabled = torch.randint(0, 1 << min(L, 30), (n,), generator=g, device="cuda", dtype=torch.int32)
        for w in (a, b):
            w.set_sources(colours=colours, enabled=enabled)
    ring = b.make_ring(R)
    names = ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done")  # (obs / actions / reward: rings)
    t0 = 0
    for chunk in (T, 3):
        per_step = []
        for j in range(chunk):
            a.step(sample=True, auto_reset=True, seed=31, t=t0 + j, env_offset=7)
            per_step.append((a.obs.clone(), a.actions.clone(), a.reward.clone()))
        b.rollout(chunk, auto_reset=True, seed=31, t=t0, env_offset=7, ring=ring, ring_pos=t0)
        for k in names:
            assert torch.equal(getattr(a, k), getattr(b, k)), (name, mode, chunk, k)
        for j in range(max(0, chunk - R), chunk):
            slot = (t0 + j) % R
            assert torch.equal(ring["obs"][slot], per_step[j][0]), (name, mode, j, "ring obs")
            assert torch.equal(ring["actions"][slot], per_step[j][1]), (name, mode, j, "ring actions")
            assert torch.equal(ring["reward"][slot], per_step[j][2]), (name, mode, j, "ring reward")
        t0 += chunk
    assert a.stats() == b.stats()


@pytest.mark.parametrize("name", ["level6", "nested", "many_agents", "config5_32x32"])
def test_fused_rollout_with_a_caller_provided_action_ring(oracle_mod, name):
    """lle_batch_rollout without on-device sampling: step j reads its joint actions from the action ring (an open-loop
    plan).  The plan: what the sampler would have drawn, with every 7th env's plan corrupted at one step by an unavailable
    action -- that env refuses the step (err of the last step only), keeps its state and goes on.  Against the oracle
    stepped with the same actions, and against single steps."""
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n, T = 320, 8
    ob = oracle_mod.OracleBatch(text, n)
    planner, bw, single = BatchedWorld(text, n), BatchedWorld(text, n), BatchedWorld(text, n)
    ring = bw.make_ring(T)
    A = bw.map.n_agents
    plan = []
    for t in range(T):  # record the sampler's choices (no auto-reset: dead agents only ever STAY)
        planner.step(sample=True, seed=3, t=t)
        plan.append(planner.actions.clone())
    plan[T // 2][::7, 0] = 5  # not an Action
    for t in range(T):
        ring["actions_rows"][t].copy_(plan[t].new_zeros(ring["actions_rows"][t].shape))
        ring["actions"][t].copy_(plan[t][:, :A])
    bw.rollout(T, auto_reset=False, ring=ring, ring_pos=0, sample=False)
    for t in range(T):
        acts = plan[t][:, :A].contiguous()
        single.step(acts)
        ostep = ob.step(acts.cpu().numpy())
        assert torch.equal(ring["obs"][t], single.obs), (name, t)
        assert np.array_equal(ring["obs"][t].cpu().numpy(), ostep["obs"]), (name, t)
    check(bw, ob, None, f"{name} after the planned rollout")
    for k in ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done"):
        assert torch.equal(getattr(bw, k), getattr(single, k)), (name, k)
    # (an env thrown off its plan may refuse later steps as well: the counters say how many)
    assert bw.stats() == single.stats() and bw.stats()["invalid"] >= len(range(0, n, 7))
    # no ring: the same joint action at every step (LLE_BUF_ACTIONS), here STAY: nothing moves, nothing is refused
    bw.actions.fill_(4)
    before = bw.pos.clone()
    bw.rollout(3, auto_reset=False, sample=False)
    assert torch.equal(bw.pos, before) and int(bw.err.max()) == 0


@pytest.mark.gpu
def test_autotune_changes_rules_not_results(oracle_mod):
    """lle_batch_autotune times the launcher's alternatives on the batch's own arena and keeps the fastest in the handle; the
    LLE_* overrides are read once per process (lle_tuning_refresh re-reads them).  Whatever it chooses, the results are those of
    an untuned batch and of the oracle; the batch ends reset with its counters at zero."""
    from lle_amd import BatchedWorld, mapgen

    for text, n in ((LEVELS[6], 4096 + 37), (mapgen.config5(3), 512)):
        a, b = BatchedWorld(text, n, autotune_ms=0), BatchedWorld(text, n, autotune_ms=0)
        before = a.tuning()
        assert before["autotuned"] == 0 and before["log"] == ""
        assert BatchedWorld(text, 2048).tuning()["autotuned"] == 1 and BatchedWorld(text, 2047).tuning()["autotuned"] == 0  # (the constructor's default)
        for t in range(3):  # (mid-episode state: autotune must end on the reset state whatever came before)
            a.step(sample=True, auto_reset=True, seed=9, t=t)
        tuned = a.autotune(budget_ms=5.0)
        assert tuned["autotuned"] == 1 and "envs_per_wave:" in tuned["log"] and "write_through:" in tuned["log"]
        assert tuned["envs_per_wave"] in (1, 2, 4, 8, 16, 32, 64) and a.kernel_info()["envs_per_wave"] == tuned["envs_per_wave"]
        assert a.stats()["env_steps"] == 0
        ob = oracle_mod.OracleBatch(text, n)
        dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
        assert_state_equal(unpack_engine(a.host_buffers(), *dims), ob.dump(), "after autotune")
        for t in range(12):
            a.step(sample=True, auto_reset=True, seed=4, t=t)
            b.step(sample=True, auto_reset=True, seed=4, t=t)
            ostep = ob.step(None, auto_reset=True, seed=4, t=t)
            ea, eb = unpack_engine(a.host_buffers(), *dims), unpack_engine(b.host_buffers(), *dims)
            assert_step_equal(ea, ostep, f"tuned t={t}")
            assert_state_equal(ea, ob.dump(), f"tuned t={t}")
            assert_step_equal(eb, ostep, f"untuned t={t}")
        assert a.stats() == b.stats()


PATTERN = {1: 0x5A, 2: 0x5A5A, 4: 0x5A5A5A5A}  # a sentinel byte pattern per element size


def test_autotune_many_agents_keeps_the_counter_slots(oracle_mod):
    """lle_batch_autotune swept environments-per-wavefront down to 2 and 1 on maps with more than four agents, where
    LLE_BUF_STATS (one slot per wavefront of AT LEAST four environments) is too short: the trial launches wrote their counters over
    REQ_POS / REQ_GEMS / REQ_ALIVE / REWARD / SRC_COLOUR.  Now no launch runs below four environments per wavefront.  A 14-agent map
    (step_kernel<16, .>, cap 4) and config 5 (8 agents, cap 8) at 65 536 / 32 768 environments: the buffers behind LLE_BUF_STATS keep
    a sentinel pattern through the sweep, LLE_STEP_EPW=1 / 2 are ignored, and the counters of a rollout add up to the events."""
    import torch

    from lle_amd import BatchedWorld, _capi, mapgen

    for text, n in ((EXTRA_MAPS["many_agents"], 65536), (mapgen.config5(1), 32768)):
        bw = BatchedWorld(text, n, autotune_ms=0)
        assert bw.kernel_info()["envs_per_wave"] >= 4
        guard = [bw.req_pos, bw.req_gems, bw.req_alive, bw.src_colour, bw.src_enabled]
        for g in guard:
            g.fill_(PATTERN[g.element_size()])
        tuned = bw.autotune(budget_ms=8.0)
        assert tuned["envs_per_wave"] >= 4, tuned
        if "envs_per_wave:" in tuned["log"]:  # (16 lanes per environment: four environments per wavefront is the only choice, nothing is swept)
            swept = tuned["log"].split("envs_per_wave:")[1].split("->")[0]
            assert " 2=" not in swept and " 1=" not in swept, tuned["log"]
        torch.cuda.synchronize()
        for g in guard:
            assert bool((g == PATTERN[g.element_size()]).all()), "a trial launch wrote past LLE_BUF_STATS"
        assert bw.stats()["env_steps"] == 0
        ev = torch.zeros(3, dtype=torch.int64, device="cuda")
        for t in range(6):
            bw.step(sample=True, auto_reset=True, seed=3, t=t)
            cnt = (bw.evcount & 0x7F).to(torch.int64)
            valid = torch.arange(bw.events.shape[1], device="cuda")[None, :] < cnt[:, None]
            ty = (bw.events >> 4).to(torch.int64)
            for k in range(3):
                ev[k] += ((ty == k) & valid).sum()
        st = bw.stats()
        assert st["env_steps"] == 6 * n and (st["exits"], st["gems"], st["deaths"]) == tuple(int(v) for v in ev.tolist())
        for g in guard:
            assert bool((g == PATTERN[g.element_size()]).all())
        del bw
    import os
    os.environ["LLE_STEP_EPW"] = "1"
    _capi.refresh_tuning()
    try:
        bw = BatchedWorld(EXTRA_MAPS["many_agents"], 20000)
        assert bw.kernel_info()["envs_per_wave"] == 4  # (the override is below the floor: ignored)
        ob = oracle_mod.OracleBatch(EXTRA_MAPS["many_agents"], 20000)
        for t in range(3):
            bw.step(sample=True, auto_reset=True, seed=8, t=t)
            check(bw, ob, ob.step(None, auto_reset=True, seed=8, t=t), f"LLE_STEP_EPW=1 t={t}")
        assert bw.stats()["env_steps"] == 3 * 20000
    finally:
        del os.environ["LLE_STEP_EPW"]
        _capi.refresh_tuning()


@pytest.mark.parametrize("name", ["level6", "level1", "nested", "many_agents"])
def test_row_rotation_changes_nothing(oracle_mod, monkeypatch, name):
    """Every wavefront starts its observation stream at another one of its rows (obs_stream.hpp row_rotation, LLE_ROW_ROTATE):
    the order in which a wavefront writes its rows is not observable.  Forced off and on against the oracle, on a batch whose
    last wavefront is ragged (and therefore not rotated), with and without row heads, single steps and a fused rollout."""
    from lle_amd import BatchedWorld

    text = MAPS.get(name) or EXTRA_MAPS[name]
    n = 1000 + 7
    for rotate in ("0", "1"):
        for heads in ("0", "1"):
            monkeypatch.setenv("LLE_ROW_ROTATE", rotate)
            monkeypatch.setenv("LLE_ROW_HEADS", heads)
            bw = BatchedWorld(text, n)
            assert bw.tuning()["rotate_rows"] == int(rotate)
            ob = oracle_mod.OracleBatch(text, n)
            for t in range(10):
                bw.step(sample=True, auto_reset=True, seed=21, t=t)
                check(bw, ob, ob.step(None, auto_reset=True, seed=21, t=t), f"{name} rotate={rotate} heads={heads} t={t}")
            bw.rollout(6, auto_reset=True, seed=21, t=10)
            for t in range(10, 16):
                ostep = ob.step(None, auto_reset=True, seed=21, t=t)
            check(bw, ob, ostep, f"{name} rotate={rotate} heads={heads} rollout")


@pytest.mark.parametrize("name", list(LONG_MAPS))
def test_long_beams(oracle_mod, name):
    """Beams longer than 32 cells: a chain of 32-cell beam words (lle_amd/csrc/tables.h; the reference's LaserBeam is a Vec<bool>,
    src/core/tiles/laser.rs:15-21).  The step kernel takes the LDS-record form of the masks and walks the chains; reset / set_state /
    observe run the lane-per-env engine.  Random rollouts with and without auto-reset, a fused rollout, random set_state requests and
    a snapshot round trip against the oracle's Vec<bool> beams, on a ragged batch."""
    import torch

    from lle_amd import BatchedWorld

    text = LONG_MAPS[name]
    n = 1000 + 11
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    assert bw.map.max_beam_len > 32 and bw.map.n_beam_words >= 5 and bw.map.n_sources == ob.Ls and bw.beams.shape == (n, bw.map.n_beam_words)

    def chk(ostep, where):
        eng = unpack_engine(bw.host_buffers(), *ob.dims)
        if ostep is not None:
            assert_step_equal(eng, ostep, where)
        assert_state_equal(eng, ob.dump(), where)
    chk(None, f"{name} after reset")
    t = 0
    for auto in (False, True):
        for _ in range(40):
            bw.step(sample=True, auto_reset=auto, seed=31, t=t, env_offset=9)
            chk(ob.step(None, auto_reset=auto, seed=31, t=t, env_offset=9), f"{name} auto_reset={auto} t={t}")
            t += 1
    snap = bw.snapshot()
    bw.rollout(12, auto_reset=True, seed=31, t=t, env_offset=9)
    for k in range(12):
        ostep = ob.step(None, auto_reset=True, seed=31, t=t + k, env_offset=9)
    chk(ostep, f"{name} fused rollout")
    # World.set_state on random requests (the beams are re-derived: world.rs:515-597)
    rng = np.random.default_rng(5)
    A, G = ob.A, ob.G
    pos = np.stack([rng.integers(0, ob.H, size=(n, A)), rng.integers(0, ob.W, size=(n, A))], axis=-1).astype(np.uint8)
    gems = rng.random((n, G)) < 0.3
    alive = rng.random((n, A)) < 0.8
    bw.set_state(torch.from_numpy(pos), torch.from_numpy(gems), torch.from_numpy(alive))
    err = bw.err.cpu().numpy()
    codes = {"InvalidWorldState": 0x40, "OutOfWorldPosition": 0x41, "InvalidAgentPosition": 0x42}
    for e in range(n):
        w = ob.world(e)
        try:
            w.set_state([tuple(int(v) for v in p) for p in pos[e]], [bool(v) for v in gems[e]], [bool(v) for v in alive[e]])
            want = 0
        except oracle_mod.OracleError as ex:
            want = codes[str(ex)]
        assert int(err[e]) == want, (name, e, int(err[e]), want)
    chk(None, f"{name} after set_state")
    bw.restore(snap)
    bw.t = t
    ob2 = oracle_mod.OracleBatch(text, n)  # (replay to the snapshot point)
    for k in range(t):
        ob2.step(None, auto_reset=k >= 40, seed=31, t=k, env_offset=9, want_obs=False)
    eng = unpack_engine(bw.host_buffers(), *ob2.dims)
    assert_state_equal(eng, ob2.dump(), f"{name} after restore")


def test_placed_rings_and_observer_outputs():
    """lle_amd.placement: trajectory rings and observer outputs larger than the Infinity Cache may be sampled like the arena
    (BatchedWorld.make_ring / bound_observer, placement_candidates=k): k candidates timed with the step kernel's store pattern
    (lle_probe_fill_rows), the fastest kept and zeroed.  Results are those of unplaced buffers; small buffers are not sampled."""
    import torch

    from lle_amd import BatchedWorld, _capi, placement

    n = 65536
    a, b = BatchedWorld(LEVELS[6], n), BatchedWorld(LEVELS[6], n)
    ra, rb = a.make_ring(3, placement_candidates=2), b.make_ring(3)
    assert ra["placement"]["candidates"] == 2 and len(ra["placement"]["row_fill_us"]) == 2 and all(v > 0 for v in ra["placement"]["row_fill_us"])
    assert rb["placement"] is None and a.make_ring(1, placement_candidates=4)["placement"] is None  # (126 MB: inside the cache)
    a.rollout(5, seed=3, ring=ra, ring_pos=0), b.rollout(5, seed=3, ring=rb, ring_pos=0)
    assert torch.equal(ra["obs_rows"], rb["obs_rows"]) and torch.equal(ra["actions_rows"], rb["actions_rows"]) and torch.equal(ra["reward"], rb["reward"])
    fa = a.bound_observer(_capi.LLE_OBS_PERSPECTIVE, placement_candidates=2)  # 503 MB
    fb = b.bound_observer(_capi.LLE_OBS_PERSPECTIVE)
    assert fa.placement["candidates"] == 2 and fb.placement is None and torch.equal(fa(), fb())
    small = a.bound_observer(_capi.LLE_OBS_PARTIAL, 3, placement_candidates=4)
    assert small.placement is None
    t = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    assert placement.time_row_fill(t, 1024, rows_per_wave=8) > 0 and int(t.sum()) == 0


@pytest.mark.parametrize("name", ["level6", "level1", "level3", "level5", "nested", "three_beams", "colour_alias", "many_agents", "gen_16x16_12agents"])
def test_incremental_observation_is_the_full_observation(oracle_mod, name):
    """LLE_STEP_INCREMENTAL_OBS: a single step in place writes only the lines of a row that dynamic state can change (tables.h
    off_dyn_chunks); the rest are in the buffer from the last full write.  The CONTENT of LLE_BUF_OBS -- padding included -- equals
    that of a batch stepped without the flag and the oracle's tensor, every step: sampled and given actions, auto-reset, steps
    without observation in between, source and exit updates (they rewrite the rows in full), whole-row and split-row kernels."""
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n = 500 + 3
    a, b = BatchedWorld(text, n), BatchedWorld(text, n)
    ob = oracle_mod.OracleBatch(text, n)
    for t in range(24):
        if t % 5 == 3:  # a step without observation, then the incremental one must still be complete
            a.step(sample=True, auto_reset=True, seed=2, t=t, write_obs=False)
            b.step(sample=True, auto_reset=True, seed=2, t=t, write_obs=False)
            ob.step(None, auto_reset=True, seed=2, t=t, want_obs=False)
            continue
        a.step(sample=True, auto_reset=t >= 8, seed=2, t=t, incremental_obs=True)
        b.step(sample=True, auto_reset=t >= 8, seed=2, t=t)
        ostep = ob.step(None, auto_reset=t >= 8, seed=2, t=t)
        assert torch.equal(a.obs_rows, b.obs_rows), (name, t)
        check(a, ob, ostep, f"{name} incremental t={t}")
        if t == 12 and a.map.n_sources:  # LaserSource.disable / set_colour on the map: a full rewrite, then incremental again
            for bw in (a, b):
                bw.map.set_source(0, enabled=False)
                bw.update_sources()
            for e in range(n):
                ob.world(e).set_source(0, enabled=False)
            assert torch.equal(a.obs_rows, b.obs_rows)
    first, nbytes = a.map.row_head
    assert a.stats() == b.stats()


def test_incremental_observation_on_split_rows_and_several_maps(oracle_mod):
    from lle_amd import BatchedWorld, mapgen
    import torch

    texts = [mapgen.config5(s) for s in range(4)]
    for maps, n in ((texts[0], 256 + 8), (texts, 4 * 64)):
        a, b = BatchedWorld(maps, n), BatchedWorld(maps, n)
        for t in range(12):
            a.step(sample=True, auto_reset=True, seed=3, t=t, incremental_obs=True)
            b.step(sample=True, auto_reset=True, seed=3, t=t)
            assert torch.equal(a.obs_rows, b.obs_rows), t
            assert torch.equal(a.pos, b.pos) and torch.equal(a.beams, b.beams)
    ob = oracle_mod.OracleBatch(texts[0], 64)
    c = BatchedWorld(texts[0], 64)
    for t in range(8):
        c.step(sample=True, auto_reset=True, seed=5, t=t, incremental_obs=True)
        check(c, ob, ob.step(None, auto_reset=True, seed=5, t=t), f"cfg5 incremental t={t}")


def test_check_obs_finds_a_tampered_row_and_incremental_is_a_batch_option(oracle_mod, monkeypatch):
    """incremental rows are an opt-in because the caller promises not to write into `obs`.  The opt-in can be
    given once, at construction; `check_obs()` is the debug check of that promise (the rows rebuilt in full from the state by another
    kernel and compared), LLE_DEBUG_CHECK_OBS=1 runs it after every step."""
    import torch

    from lle_amd import BatchedWorld

    n = 300
    for dt in (None, torch.float16):
        a, b = BatchedWorld(LEVELS[6], n, incremental_obs=True, obs_dtype=dt), BatchedWorld(LEVELS[6], n, obs_dtype=dt)
        ob = oracle_mod.OracleBatch(LEVELS[6], n)
        for t in range(10):
            a.step(sample=True, auto_reset=True, seed=4, t=t)   # (incremental: the batch's own default)
            b.step(sample=True, auto_reset=True, seed=4, t=t)
            assert torch.equal(a.obs_rows, b.obs_rows) and a.check_obs() == 0 and b.check_obs() == 0
            ostep = ob.step(None, auto_reset=True, seed=4, t=t)
            if dt is None:
                check(a, ob, ostep, f"incremental batch t={t}")
        first, nbytes = a.map.row_head   # static lines of the row (never rewritten by an incremental step)
        assert nbytes > 0
        a.obs_rows[7, first] = 1         # the caller breaks the promise ...
        a.obs_rows[11, first + 5] = -1
        a.step(sample=True, auto_reset=True, seed=4, t=10)
        assert a.check_obs() == 2        # ... and the check says so (an incremental step does not repair static lines)
        a.observe()                      # a full rewrite does
        assert a.check_obs() == 0
    monkeypatch.setenv("LLE_DEBUG_CHECK_OBS", "1")
    c = BatchedWorld(LEVELS[6], n, incremental_obs=True)
    c.step(sample=True, seed=1, t=0)
    c.obs_rows[3, c.map.row_head[0]] = 1
    with pytest.raises(AssertionError, match="LLE_DEBUG_CHECK_OBS"):
        c.step(sample=True, seed=1, t=1)
