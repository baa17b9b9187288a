"""Split rows (step_kernel LAUNCH_SPLIT_ROWS, obs_stream.hpp write_observations_split): for big observation rows the
wavefronts of a workgroup each keep one slice of the row and stream that slice of every environment of the workgroup.
The launcher picks it for config 5 (20 KB rows); LLE_STEP_SPLIT=1 forces it on every map whose kernel carries it (more
than four agents), LLE_STEP_SPLIT=0 forces whole-row copies.  Both must match the oracle bit for bit, in single steps,
fused rollouts with trajectory rings, batches of several maps, ragged batches and every envs-per-wave setting."""
import pytest

from tests.parity_util import EXTRA_MAPS
from tests.test_gpu_parity import check

pytestmark = pytest.mark.gpu

BIG_GROUP_MAPS = ["many_agents", "gen_16x16_12agents", "config5_32x32"]  # 14, 12 and 8 agents: G = 16, 16, 8


@pytest.mark.parametrize("name", BIG_GROUP_MAPS)
@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_split_rows_random_rollout(oracle_mod, monkeypatch, name, split, auto_reset):
    from lle_amd import BatchedWorld

    monkeypatch.setenv("LLE_STEP_SPLIT", split)
    text = EXTRA_MAPS[name]
    n, steps = 1000, 40  # ragged: the last workgroup is partly empty, its wavefronts still meet at the barrier
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    check(bw, ob, None, f"{name} split={split} after reset")
    for t in range(steps):
        bw.step(sample=True, auto_reset=auto_reset, seed=1234, t=t, env_offset=3)
        check(bw, ob, ob.step(None, auto_reset=auto_reset, seed=1234, t=t, env_offset=3), f"{name} split={split} t={t}")


@pytest.mark.parametrize("n,epw", [(1, None), (7, None), (33, 1), (100, 2), (257, 4), (640, 8)])
def test_split_rows_small_and_ragged_batches(oracle_mod, monkeypatch, n, epw):
    from lle_amd import BatchedWorld

    monkeypatch.setenv("LLE_STEP_SPLIT", "1")
    if epw:
        monkeypatch.setenv("LLE_STEP_EPW", str(epw))  # environments per wavefront of the step kernel (tuning override)
    text = EXTRA_MAPS["config5_32x32"]
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    for t in range(12):
        bw.step(sample=True, auto_reset=True, seed=5, t=t)
        check(bw, ob, ob.step(None, auto_reset=True, seed=5, t=t), f"n={n} t={t}")


@pytest.mark.parametrize("name", BIG_GROUP_MAPS)
@pytest.mark.parametrize("two_maps", [False, True])
def test_split_rows_fused_rollout_with_rings(monkeypatch, name, two_maps):
    """Whole-row copies, single steps (checked against the oracle above) == split rows, fused rollout: every buffer, every ring slot."""
    import torch

    from lle_amd import BatchedWorld

    text = EXTRA_MAPS[name]
    n, R = 640, 4
    maps = [text, text] if two_maps else text
    monkeypatch.setenv("LLE_STEP_SPLIT", "0")
    a = BatchedWorld(maps, n)
    b = BatchedWorld(maps, n)
    ring = b.make_ring(R)
    names = ("pos", "bits", "gems", "beams", "avail", "err", "evcount", "events", "done")
    t0 = 0
    for chunk in (9, 3, 1):
        per_step = []
        monkeypatch.setenv("LLE_STEP_SPLIT", "0")
        for j in range(chunk):
            a.step(sample=True, auto_reset=True, seed=31, t=t0 + j, env_offset=7)
            per_step.append((a.obs.clone(), a.actions.clone(), a.reward.clone()))
        monkeypatch.setenv("LLE_STEP_SPLIT", "1")
        b.rollout(chunk, auto_reset=True, seed=31, t=t0, env_offset=7, ring=ring, ring_pos=t0)
        for k in names:
            assert torch.equal(getattr(a, k), getattr(b, k)), (name, chunk, k)
        for j in range(max(0, chunk - R), chunk):
            slot = (t0 + j) % R
            assert torch.equal(ring["obs"][slot], per_step[j][0]), (name, j, "ring obs")
            assert torch.equal(ring["actions"][slot], per_step[j][1]), (name, j, "ring actions")
            assert torch.equal(ring["reward"][slot], per_step[j][2]), (name, j, "ring reward")
        t0 += chunk
    assert a.stats() == b.stats()


def test_config5_uses_split_rows_by_default_and_fits_four_workgroups_per_cu(monkeypatch):
    from lle_amd import BatchedWorld, mapgen

    monkeypatch.delenv("LLE_STEP_SPLIT", raising=False)
    bw = BatchedWorld(mapgen.config5(0), 4096)
    info = bw.kernel_info()
    assert info["kernel"] == "step_kernel<8,8>" and info["lds_bytes"] <= 40 * 1024, info
    small = BatchedWorld(EXTRA_MAPS["many_agents"], 256)
    assert small.kernel_info()["lds_bytes"] < 64 * 1024
