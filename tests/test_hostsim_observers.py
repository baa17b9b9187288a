"""CPU differential: the observation builders of observers_logic.hpp / Map::compile_view (host build, tests/hostsim) vs
the oracle's restatement of python/lle/observations.py (oracle/observers.py), along random rollouts."""
import pytest

from oracle.levels import LEVELS
from tests.observer_checks import compare_all
from tests.parity_util import EXTRA_MAPS, LONG_MAPS

MAPS = {f"level{k}": v for k, v in LEVELS.items()}
MAPS.update(EXTRA_MAPS)
MAPS.update(LONG_MAPS)  # beams longer than 32 cells: chains of beam words (tables.h)


@pytest.mark.parametrize("name", list(MAPS))
def test_observers_along_rollout(oracle_mod, name):
    from lle_amd import _capi
    from tests import hostsim

    text = MAPS[name]
    n, steps = 12, 24
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    envs = range(0, n, 3)
    compare_all(sb.observe_as, sb.available_actions, ob, envs, f"{name} after reset")
    for t in range(steps):
        auto_reset = t >= steps // 2  # first half: dead agents / collected gems pile up; second half: resets
        ob.step(None, auto_reset=auto_reset, seed=99, t=t, env_offset=3, want_obs=False)
        sb.step(None, flags=_capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto_reset else 0), seed=99, t=t, env_offset=3)
        if t % 4 == 3 or t == steps - 1:
            compare_all(sb.observe_as, sb.available_actions, ob, envs, f"{name} t={t}")


def test_observers_after_recolouring(oracle_mod):
    """Views are compiled from the current colours: recolour / disable a source and compare again."""
    from tests import hostsim

    text = EXTRA_MAPS["nested"]
    ob = oracle_mod.OracleBatch(text, 4)
    sb = hostsim.SimBatch(text, 4)
    for lid, kw in ((0, dict(colour=1)), (1, dict(enabled=False)), (0, dict(colour=3)), (1, dict(enabled=True, colour=5))):
        for e in range(4):
            ob.world(e).set_source(lid, **kw)
        sb.set_source(lid, **kw)
        compare_all(sb.observe_as, sb.available_actions, ob, range(4), f"nested after set_source({lid}, {kw})")
