"""The general single-step kernels with row heads (step_kernel MODE 7: several maps per batch, the fused LLE.step
outputs; MODE 8: per-environment sources, LLE.step with randomize_lasers): the existing tests of those paths, collected again
with LLE_ROW_HEADS=1 so that their small batches take the head path (the launcher's own choice needs 2 048+ wavefronts), on
128-byte aligned rows (rows without a head otherwise)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def heads_forced_on_aligned_rows(monkeypatch):
    from lle_amd import BatchedWorld

    monkeypatch.setenv("LLE_ROW_HEADS", "1")
    orig = BatchedWorld.__init__

    def init(self, map_or_text, n_envs, device=None, envs_per_wave=None, row_align=None, **kw):
        orig(self, map_or_text, n_envs, device=device, envs_per_wave=envs_per_wave, row_align=128 if row_align is None else row_align, **kw)

    monkeypatch.setattr(BatchedWorld, "__init__", init)


from tests.test_gpu_env import (test_batched_lle_env_kat, test_batched_lle_matches_per_env_restatement,  # noqa: E402,F401
                                test_batched_lle_one_launch_step, test_env_outputs_equals_separate_entry_points,
                                test_one_launch_step_equals_step_plus_env_outputs)
from tests.test_gpu_env import (test_persistent_step_equals_the_allocating_step, test_randomized_lasers_through_auto_reset_steps,  # noqa: E402,F401
                                test_restore_does_not_bring_back_an_old_output_descriptor)
from tests.test_gpu_env_sources import *  # noqa: E402,F401,F403  (every per-env-sources parity test: MODE 8 where the map has a head)
from tests.test_gpu_exits import test_exits_with_per_env_sources  # noqa: E402,F401
from tests.test_gpu_multi_map import test_blocks_of_maps_match_their_oracles, test_observers_and_per_env_sources_on_blocks_of_maps  # noqa: E402,F401
from tests.test_gpu_parity import test_explicit_and_invalid_actions, test_random_rollout, test_reward_counts_and_snapshot  # noqa: E402,F401


@pytest.mark.parametrize("name", ["level6", "gen_12x13_4agents_8lasers"])
def test_fused_step_outputs_with_heads_equal_two_launches(name):
    """BatchedLLE.step(fused=True) (one launch, MODE 7 here) against the two-launch path, every output; the generated map
    has 8 sources: beam masks in the LDS record, this lane's share read ahead of the head stores."""
    import torch

    from lle_amd import BatchedLLE, mapgen
    from oracle.levels import LEVELS

    text = LEVELS[6] if name == "level6" else mapgen.generate(12, 13, 4, 8, 4, seed=2)
    n = 3000
    a, b = BatchedLLE(text, n), BatchedLLE(text, n)
    a.reset(), b.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(30):
        avail = a.available_actions()
        acts = torch.multinomial(avail.reshape(-1, 5).float(), 1, generator=g).reshape(n, -1).to(torch.uint8)
        x, y = a.step(acts, auto_reset=True, fused=True), b.step(acts, auto_reset=True, fused=False)
        for k in ("obs", "state", "reward", "done", "available_actions", "err"):
            assert torch.equal(x[k], y[k]), (k, t)
