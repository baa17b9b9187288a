"""The lane-per-AGENT state machine (lle_amd/csrc/step_lanes.hpp: what a group of lanes of step_kernel executes, the
product's hot path) built for the host, against the reference's known-answer tests and against the oracle.

tests/hostsim runs the SAME source as the kernel with every lane value held G times (step_lanes.hpp explains how); the
suites below are the ones that already run on the lane-per-env restatement (step_logic.hpp), collected a second time
with the host simulator's engine switched to "lanes"."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, unpack_engine


@pytest.fixture(autouse=True)
def lanes_engine(monkeypatch):
    from tests import hostsim
    monkeypatch.setattr(hostsim, "DEFAULT_ENGINE", ["lanes"])


from tests.test_hostsim_env_sources import test_random_colours_and_flags_per_env  # noqa: E402,F401
from tests.test_hostsim_kat import test_hostsim_kat  # noqa: E402,F401
from tests.test_hostsim_parity import (test_config1_single_env_10k_steps, test_fuzz_maps, test_invalid_actions_leave_env_untouched,  # noqa: E402,F401
                                       test_random_rollout)


def _rollout(oracle_mod, text, n, steps, engine, auto_reset, seed):
    from lle_amd import _capi
    from tests import hostsim
    from tests.test_hostsim_parity import sim_bufs

    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    sb.set_engine(engine)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    flags = _capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto_reset else 0)
    for t in range(steps):
        ostep = ob.step(None, auto_reset=auto_reset, seed=seed, t=t)
        sb.step(None, flags=flags, seed=seed, t=t)
        eng = unpack_engine(sim_bufs(sb), *dims)
        assert_step_equal(eng, ostep, f"{engine} t={t}")
        assert_state_equal(eng, ob.dump(), f"{engine} t={t}")
    return sb


@pytest.mark.parametrize("engine", ["lanes", "lanes_no_shortcut"])
@pytest.mark.parametrize("name", ["q1", "nested", "three_beams", "four_layers", "exit_under_beam", "voids_gems", "many_agents",
                                  "gen_16x16_12agents", "config5_32x32"])
def test_shortcut_and_full_passes_agree_with_the_oracle(oracle_mod, name, engine):
    """The no-op-pass shortcut of step_lanes (`if (!first_pass && any_lit == 0) go = false`: a pass after the first that
    re-lights nothing cannot change anything, world.rs:468-472) switched on and off: both must match the oracle on maps
    with deaths every few steps (no auto-reset: corpses pile up on gems, exits and under beams, quirks Q1 / Q2 / Q4)."""
    _rollout(oracle_mod, EXTRA_MAPS[name], 96, 60, engine, auto_reset=False, seed=5)
    _rollout(oracle_mod, EXTRA_MAPS[name], 96, 60, engine, auto_reset=True, seed=6)


def test_the_shortcut_is_taken_and_not_taken():
    """Both branches of the shortcut on known scripts, observed through the count of executed move_agents passes.
    Q1 (tests/world_integration_tests.rs:279-309), step [East, North, Stay]: agent 1 dies in pass 1; in pass 2 agent 0
    leaves (0,1) with the bit of beam 1 off and re-lights it -> pass 2 is NOT skippable and kills agent 0; pass 3
    re-lights nothing -> skipped (2 passes with the shortcut, 3 without).
    A plain death -- agent 0 walks south into the lit beam of colour 1, nobody stands on a beam: pass 2 re-lights
    nothing -> skipped (1 pass with, 2 without)."""
    from tests import hostsim

    for engine, q1_passes, plain_passes in (("lanes", 2, 1), ("lanes_no_shortcut", 3, 2)):
        q1 = hostsim.SimWorld(EXTRA_MAPS["q1"])
        q1.b.set_engine(engine)
        q1.reset()
        p0 = q1.b.lane_passes()
        assert q1.step([2, 0, 4]) == [(2, 1), (2, 0)], engine     # Died(1) in pass 1, Died(0) in pass 2
        assert q1.beam_bits(1) == [True, False, True], engine     # the stale [on, off, on] beam
        assert q1.b.lane_passes() - p0 == q1_passes, engine
        plain = hostsim.SimWorld("S0 . X\n.  . .\nL1E . .\nS1 . X")
        plain.b.set_engine(engine)
        plain.reset()
        assert plain.step([2, 4]) == [] and plain.step([1, 4]) == []      # East, South: (1,1)
        p0 = plain.b.lane_passes()
        assert plain.step([1, 4]) == [(2, 0)], engine                    # South onto (2,1): the beam of colour 1
        assert plain.b.lane_passes() - p0 == plain_passes, engine
        assert plain.alive() == [False, True]
