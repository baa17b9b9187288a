"""CPU differential for per-environment laser sources: the device state machine with per-env colours (recolour_lay in
step_logic.hpp), the second table section (bare template + element list) and the set_sources rules, built for the host,
vs oracle worlds that each take the same set_colour / enable / disable calls."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, LONG_MAPS, assert_state_equal, assert_step_equal, legal_colours, unpack_engine
from tests.test_hostsim_parity import sim_bufs

MAPS = {"level6": LEVELS[6], "nested": EXTRA_MAPS["nested"], "three_beams": EXTRA_MAPS["three_beams"],
        "four_layers": EXTRA_MAPS["four_layers"], "many_agents": EXTRA_MAPS["many_agents"], "gen_20_lasers": EXTRA_MAPS["gen_20_lasers"],
        "long_q1": LONG_MAPS["long_q1"], "long_crossing": LONG_MAPS["long_crossing"]}  # (beams of several words: colours stay per SOURCE)


@pytest.mark.parametrize("name", list(MAPS))
def test_random_colours_and_flags_per_env(oracle_mod, name):
    from lle_amd import _capi
    from tests import hostsim

    text = MAPS[name]
    n = 48
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    dims = ob.dims
    A, L = ob.A, sb.map.n_sources
    is_on = np.array([[bool(s[4]) for s in ob.world(0).sources()]] * n)
    rng = np.random.default_rng(11)
    t = 0
    for episode in range(4):
        auto = episode % 2 == 1
        for _ in range(8):
            ostep = ob.step(None, auto_reset=auto, seed=77, t=t, env_offset=2)
            sb.step(None, flags=_capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto else 0), seed=77, t=t, env_offset=2)
            eng = unpack_engine(sim_bufs(sb), *dims)
            assert_step_equal(eng, ostep, f"{name} t={t}")
            assert_state_equal(eng, ob.dump(), f"{name} t={t}")
            t += 1
        colours = legal_colours(text, rng.integers(0, A, size=(n, L), dtype=np.uint8))
        enabled = rng.integers(0, 1 << L, size=n, dtype=np.int64).astype(np.uint32) if episode != 2 else None
        mask = (rng.random(n) < 0.7).astype(np.uint8) if episode != 0 else None
        sb.set_sources(colours, enabled, mask)
        for e in range(n):
            if mask is not None and not mask[e]:
                continue
            w = ob.world(e)
            for l in range(L):
                w.set_source(l, colour=int(colours[e, l]))
                if enabled is not None:
                    want = bool((int(enabled[e]) >> l) & 1)
                    if want != is_on[e, l]:      # the binding acts only on a change (pylaser_source.rs:55-60)
                        w.set_source(l, enabled=want)
                        is_on[e, l] = want
        assert_state_equal(unpack_engine(sim_bufs(sb), *dims), ob.dump(), f"{name} after set_sources {episode}")


def test_invalid_colour_refused(oracle_mod):
    from tests import hostsim

    sb = hostsim.SimBatch(LEVELS[6], 4)
    before = sb.buf("obs").copy()
    colours = np.zeros((4, sb.map.n_sources), np.uint8)
    colours[2, 0] = sb.map.n_agents
    sb.set_sources(colours)
    assert list(sb.buf("err")) == [0, 0, 0x43, 0]
    assert np.array_equal(sb.buf("obs")[2], before[2])     # untouched
    assert not np.array_equal(sb.buf("obs")[0], before[0])  # recoloured: the -1 marks moved to layer LASER_0 + 0


def test_colour_that_crosses_a_start_is_refused(oracle_mod):
    """pylaser_source.rs:121-139: `L0E X X S0 S1` -- the beam of source 0 runs over both starts; it may keep colour 0 ...
    no: S1 lies on it too, so NO colour but ... the map itself is only legal because parse-time pruning compares each
    start with the beam's owner (world_config.rs:225-243).  python/tests/test_world.py:537-545 pins that colour 1 is refused."""
    from lle_amd import _capi
    from tests import hostsim

    m = _capi.Map("L0E X X . S0\n@ @ @ S1 .")           # S0 on the beam of source 0, S1 off it
    assert m.colour_allowed(0, 0) and not m.colour_allowed(0, 1)
    assert not _capi.Map("L0E X X S0 S1").colour_allowed(0, 1)   # the reference's map: agent 0's start is on the beam
    with pytest.raises(ValueError):
        m.colour_allowed(0, 2)
    sb = hostsim.SimBatch("L0E X X . S0\n@ @ @ S1 .", 4)
    before = (sb.buf("obs").copy(), sb.buf("beams").copy())
    sb.set_sources(np.array([[0], [1], [0], [1]], np.uint8))
    assert list(sb.buf("err")) == [0, _capi.LLE_ENV_COLOUR_CROSSES_START, 0, _capi.LLE_ENV_COLOUR_CROSSES_START]
    assert np.array_equal(sb.buf("obs")[1], before[0][1]) and np.array_equal(sb.buf("beams")[1], before[1][1])
    assert np.array_equal(legal_colours("L0E X X . S0\n@ @ @ S1 .", np.array([[1], [0]], np.uint8)), [[0], [0]])
