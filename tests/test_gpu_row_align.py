"""The parity suite again on 128-byte aligned observation rows (lle_map_set_row_align; level 6: 1 872 -> 1 920 B pitch).

Every test imported below runs unchanged, with BatchedWorld creating its maps at row_align = 128: bit-exact on the C*H*W
prefix of every row (the reference's tensor, python/lle/observations.py:254-266) exactly as on the default pitch -- the
comparisons of those tests slice rows to C*H*W -- plus, here, the padding of every row must be zero."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS

pytestmark = pytest.mark.gpu

ALIGN = 128


@pytest.fixture(autouse=True)
def padded_rows(monkeypatch):
    from lle_amd import BatchedWorld

    orig = BatchedWorld.__init__

    def init(self, map_or_text, n_envs, device=None, envs_per_wave=None, row_align=None, **kw):
        align = ALIGN if row_align is None else row_align
        orig(self, map_or_text, n_envs, device=device, envs_per_wave=envs_per_wave, row_align=align, **kw)
        assert self.map.obs_stride % align == 0 and self.obs_rows.shape[1] == self.map.obs_stride

    monkeypatch.setattr(BatchedWorld, "__init__", init)


# the same test functions, collected a second time under the fixture above
from tests.test_gpu_env import test_batched_lle_matches_per_env_restatement, test_env_outputs_equals_separate_entry_points  # noqa: E402,F401
from tests.test_gpu_env_sources import (test_other_builders_and_modes_with_per_env_sources, test_random_colours_and_flags_per_env,  # noqa: E402,F401
                                        test_reset_sources_equals_reset_then_set_sources)
from tests.test_gpu_multi_map import test_blocks_of_maps_match_their_oracles, test_observers_and_per_env_sources_on_blocks_of_maps  # noqa: E402,F401
from tests.test_gpu_observers import test_observers_along_rollout, test_observers_follow_source_updates  # noqa: E402,F401
from tests.test_gpu_parity import (test_batched_set_state, test_config2_level1_batch4096, test_config3_level6_batch65536,  # noqa: E402,F401
                                   test_config5_generated_32x32, test_envs_per_wave_variants, test_fused_rollout_equals_single_steps,
                                   test_masked_reset, test_random_rollout, test_reward_counts_and_snapshot,
                                   test_rollout_with_rings_equals_steps_in_every_general_mode, test_store_policies_agree_with_oracle,
                                   test_update_sources)


@pytest.mark.parametrize("name", ["level1", "level6", "nested", "colour_alias", "config5_32x32"])
@pytest.mark.parametrize("align", [32, 64, 128, 256])
def test_row_padding_stays_zero_and_prefix_matches_the_default_pitch(name, align):
    import torch

    from lle_amd import BatchedWorld, Map

    text = dict({f"level{k}": v for k, v in LEVELS.items()}, **EXTRA_MAPS)[name]
    n = 777
    ref = BatchedWorld(Map(text), n, row_align=16)
    pad = BatchedWorld(Map(text), n, row_align=align)
    nb = ref.map.obs_bytes
    assert pad.map.obs_stride == -(-nb // align) * align and pad.map.obs_bytes == nb
    ring = pad.make_ring(3)
    for t in range(24):
        ref.step(sample=True, auto_reset=True, seed=11, t=t)
        if t % 8 < 4:
            pad.step(sample=True, auto_reset=True, seed=11, t=t)
            rows = pad.obs_rows
        else:
            pad.rollout(1, auto_reset=True, seed=11, t=t, ring=ring, ring_pos=t)
            rows = ring["obs_rows"][t % 3]
        torch.cuda.synchronize()
        assert torch.equal(rows[:, :nb], ref.obs_rows[:, :nb]), f"{name} align={align} t={t}"
        assert int(rows[:, nb:].abs().sum()) == 0, f"{name} align={align} t={t}: padding written"
        assert torch.equal(pad.bits, ref.bits) and torch.equal(pad.pos, ref.pos)
    # the other layered-style builders share the pitch rule
    from lle_amd import _capi
    for kind, param in ((_capi.LLE_OBS_LAYERED, 0), (_capi.LLE_OBS_PERSPECTIVE, 0), (_capi.LLE_OBS_LAYERED_PADDED, 2)):
        d = pad.obs_desc(kind, param)
        if not d.supported:
            continue
        a, b = pad.observe_as(kind, param), ref.observe_as(kind, param)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (name, align, kind)
        assert int(d.stride[d.ndim - 4]) % align == 0
