"""Differential test on CPU: host build of the device state machine + map tables vs the oracle, random rollouts."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, unpack_engine

MAPS = {f"level{k}": v for k, v in LEVELS.items()}
MAPS.update(EXTRA_MAPS)


def sim_bufs(b, with_obs=True):
    names = ["pos", "bits", "gems", "beams", "avail", "actions", "err", "evcount", "events"] + (["obs"] if with_obs else [])
    return {k: b.buf(k) for k in names}


@pytest.mark.parametrize("name", list(MAPS))
@pytest.mark.parametrize("auto_reset", [False, True])
def test_random_rollout(oracle_mod, name, auto_reset):
    from lle_amd import _capi
    from tests import hostsim

    text = MAPS[name]
    n, steps = 96, 60
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    assert_state_equal(unpack_engine(sim_bufs(sb), *dims), ob.dump(), f"{name} after reset")
    flags = _capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto_reset else 0)
    for t in range(steps):
        ostep = ob.step(None, auto_reset=auto_reset, seed=1234, t=t, env_offset=7)
        sb.step(None, flags=flags, seed=1234, t=t, env_offset=7)
        eng = unpack_engine(sim_bufs(sb), *dims)
        assert_step_equal(eng, ostep, f"{name} t={t}")
        assert_state_equal(eng, ob.dump(), f"{name} t={t}")
    for e in range(0, n, 17):
        assert ob.world(e).panics()[0] == 0


def test_invalid_actions_leave_env_untouched(oracle_mod):
    from tests import hostsim

    text = LEVELS[6]
    n = 64
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    rng = np.random.default_rng(0)
    for t in range(40):
        actions = rng.integers(0, 6, size=(n, ob.A), dtype=np.uint8)  # includes unavailable and out-of-range (5) actions
        ostep = ob.step(actions)
        sb.step(actions)
        eng = unpack_engine(sim_bufs(sb), *dims)
        assert_step_equal(eng, ostep, f"t={t}")
        assert_state_equal(eng, ob.dump(), f"t={t}")


def test_sampler_matches_oracle(oracle_mod):
    from lle_amd import _capi

    for seed, env, t, agent in [(0, 0, 0, 0), (1234, 65535, 99, 3), (2**63, 2**40, 2**33, 15)]:
        assert _capi.lib().lle_action_hash(seed, env, t, agent) == oracle_mod.action_hash(seed, env, t, agent)
