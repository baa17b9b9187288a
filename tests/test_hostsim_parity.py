"""Differential test on CPU: host build of the device state machine + map tables vs the oracle, random rollouts."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, LONG_MAPS, assert_state_equal, assert_step_equal, unpack_engine

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


@pytest.mark.parametrize("name", list(LONG_MAPS))
@pytest.mark.parametrize("engine", ["env", "lanes", "lanes_no_shortcut"])
def test_long_beams_random_rollout(oracle_mod, name, engine):
    """Beams longer than 32 cells (chains of beam words, tables.h): both host engines -- step_logic.hpp (one lane per environment:
    reset / set_state / world_kernel) and step_lanes.hpp (one lane per agent, the LDS-record form that walks the chains) -- against the
    oracle's Vec<bool> beams on random rollouts with and without auto-reset, then random set_state requests."""
    from lle_amd import _capi
    from tests import hostsim

    text = LONG_MAPS[name]
    n, steps = 128, 80
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    sb.set_engine(engine)
    assert ob.beam_stride > 32 and sb.map.n_beam_words >= 5 and sb.map.n_sources == ob.Ls
    assert_state_equal(unpack_engine(sim_bufs(sb), *ob.dims), ob.dump(), f"{name} after reset")
    for auto_reset in (False, True):
        flags = _capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto_reset else 0)
        for t in range(steps):
            ostep = ob.step(None, auto_reset=auto_reset, seed=77, t=t, env_offset=3)
            sb.step(None, flags=flags, seed=77, t=t, env_offset=3)
            eng = unpack_engine(sim_bufs(sb), *ob.dims)
            assert_step_equal(eng, ostep, f"{name} {engine} auto_reset={auto_reset} t={t}")
            assert_state_equal(eng, ob.dump(), f"{name} {engine} auto_reset={auto_reset} t={t}")
    assert (sb.buf("beams")[:, sb.map.n_beam_words:] == 0).all() if sb.buf("beams").shape[1] > sb.map.n_beam_words else True


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


def test_set_state_random_requests(oracle_mod):
    """World.set_state (lossy re-derivation + rollback rules) on random, often invalid, requests."""
    from lle_amd import _decode
    from tests import hostsim
    from tests.parity_util import EXTRA_MAPS

    text = EXTRA_MAPS["nested"]
    n = 300
    ob = oracle_mod.OracleBatch(text, n)
    sb = hostsim.SimBatch(text, n)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    rng = np.random.default_rng(1)
    codes = {"InvalidWorldState": 0x40, "OutOfWorldPosition": 0x41, "InvalidAgentPosition": 0x42}
    for rnd in range(8):
        for t in range(5):
            sb.step(None, flags=1, seed=rnd, t=t)
            ob.step(None, seed=rnd, t=t)
        pos = np.stack([rng.integers(0, ob.H + (rnd >= 6), size=(n, ob.A)), rng.integers(0, ob.W, size=(n, ob.A))], axis=-1).astype(np.uint8)
        gems = rng.integers(0, 2, size=(n, ob.G)).astype(bool)
        alive = rng.integers(0, 4, size=(n, ob.A)) > 0
        sb.buf("req_pos")[:] = pos
        sb.buf("req_gems")[:] = [_decode.pack_bits(g) for g in gems]
        sb.buf("req_alive")[:] = [_decode.pack_bits(a) for a in alive]
        sb.set_state()
        for e in range(n):
            try:
                ev, rc = ob.world(e).set_state([tuple(int(v) for v in p) for p in pos[e]], list(gems[e]), list(alive[e])), 0
            except oracle_mod.OracleError as ex:
                ev, rc = [], codes[ex.kind]
            assert int(sb.buf("err")[e]) == rc, (e, int(sb.buf("err")[e]), rc)
            assert _decode.events_list(sb.buf("evcount")[e], sb.buf("events")[e]) == ev
        assert_state_equal(unpack_engine(sim_bufs(sb), *dims), ob.dump(), f"set_state round {rnd}")
        # a set_state that failed with InvalidWorldState leaves the reference with stale availability lists (its next
        # step may index out of the grid and panic): such worlds are reset before the rollout goes on
        poisoned = (sb.buf("err") == 0x40).astype(np.uint8)
        sb.L.hs_reset(sb.h, poisoned.ctypes.data)
        for e in np.nonzero(poisoned)[0]:
            ob.world(int(e)).reset()


def _fuzz_maps():
    from lle_amd import mapgen

    rng = np.random.default_rng(7)
    out = []
    for seed in range(24):
        h, w = int(rng.integers(4, 14)), int(rng.integers(4, 14))
        agents = int(rng.integers(1, 7))
        lasers = int(rng.integers(0, 7))
        try:
            out.append((f"fuzz{seed}_{h}x{w}_a{agents}_l{lasers}",
                        mapgen.generate(h, w, agents, lasers, n_gems=int(rng.integers(0, 5)), wall_fraction=0.08,
                                        n_voids=int(rng.integers(0, 3)), seed=100 + seed)))
        except RuntimeError:
            pass  # the rejection sampler found no valid placement for these sizes
    return out


@pytest.mark.parametrize("name,text", _fuzz_maps())
def test_fuzz_maps(oracle_mod, name, text):
    """Random small maps (crossing beams of random colours, beams over gems and exits, voids): the two independent
    formulations of the rules must agree on every step, with and without auto-reset."""
    from lle_amd import _capi
    from tests import hostsim

    n, steps = 48, 40
    for auto_reset in (False, True):
        ob = oracle_mod.OracleBatch(text, n)
        sb = hostsim.SimBatch(text, n)
        dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
        flags = _capi.LLE_STEP_SAMPLE_ACTIONS | (_capi.LLE_STEP_AUTO_RESET if auto_reset else 0)
        for t in range(steps):
            ostep = ob.step(None, auto_reset=auto_reset, seed=5, t=t)
            sb.step(None, flags=flags, seed=5, t=t)
            eng = unpack_engine(sim_bufs(sb), *dims)
            assert_step_equal(eng, ostep, f"{name} t={t}")
            assert_state_equal(eng, ob.dump(), f"{name} t={t}")


def test_config1_single_env_10k_steps(oracle_mod):
    """BASELINE.json configs[0]: World("S0 G X"), one env, 10 000 random-action steps (auto-reset when done):
    oracle vs the host build of the device state machine, every step."""
    from lle_amd import _capi
    from tests import hostsim

    text = "S0 G X"
    ob = oracle_mod.OracleBatch(text, 1)
    sb = hostsim.SimBatch(text, 1)
    dims = (ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
    stats = np.zeros(8, np.int64)
    for t in range(10_000):
        ostep = ob.step(None, auto_reset=True, seed=3, t=t, stats=stats)
        sb.step(None, flags=_capi.LLE_STEP_SAMPLE_ACTIONS | _capi.LLE_STEP_AUTO_RESET, seed=3, t=t)
        if t % 97 == 0 or t > 9_900:
            eng = unpack_engine(sim_bufs(sb), *dims)
            assert_step_equal(eng, ostep, f"t={t}")
            assert_state_equal(eng, ob.dump(), f"t={t}")
    assert np.array_equal(sb.buf("stats")[:7], stats[:7]) and stats[0] == 10_000 and stats[3] > 100  # many exits


def test_sampler_is_uniform_and_decorrelated(oracle_mod):
    """The 16-bit fields of the counter-based sampler (NOTEBOOK.md section 6): flat histogram, no visible dependence
    between neighbouring envs / steps / agents (chi-square bounds loose enough to be deterministic)."""
    f = np.array([[[oracle_mod.action_hash(1234, e, t, a) for a in range(4)] for t in range(16)] for e in range(512)], dtype=np.int64)
    assert f.min() >= 0 and f.max() < 65536
    n = f.size
    hist = np.bincount((f >> 10).ravel(), minlength=64)            # 64 bins
    chi2 = float(((hist - n / 64) ** 2 / (n / 64)).sum())
    assert chi2 < 120, chi2                                       # 63 dof: mean 63, 120 is far in the tail
    act = (f * 5) >> 16                                           # five available actions
    for axis in range(3):                                         # neighbours along env / t / agent
        a0 = np.take(act, range(0, act.shape[axis] - 1), axis=axis).ravel()
        a1 = np.take(act, range(1, act.shape[axis]), axis=axis).ravel()
        joint = np.bincount(a0 * 5 + a1, minlength=25).astype(float)
        exp = joint.sum() / 25
        assert float(((joint - exp) ** 2 / exp).sum()) < 60, axis  # 24 dof


def test_template_bit_form_and_packed_tables_rebuild_the_sections(oracle_mod):
    """MapHeader.off_packed (tables.h): the table section as 16-bit cell words + the layer words of the cells under a beam + dyn table and dynamic chunks,
    expanded (tests/hostsim: the kernel's arithmetic) gives the verbatim section byte for byte.  MapHeader.off_tmpl_bits (tables.h): the static observation as one bit per byte plus the list of -1 marks, which the split-row launch's
    wavefronts expand instead of copying the template -- the expansion (the kernel's arithmetic, restated in tests/hostsim) gives the template
    byte for byte, on every level, every extra map, the fuzz maps, config 5's shape, and after a source is re-coloured / disabled (the table
    is recompiled), and the template is the oracle's observation of a world without agents' and dynamic bytes where both are static."""
    from lle_amd import mapgen
    from tests import hostsim

    texts = dict(MAPS)
    texts.update(dict(_fuzz_maps()[:40]))
    texts.update({f"config5_{s}": mapgen.config5(s) for s in range(3)})
    seen = recoloured = 0
    for name, text in texts.items():
        sb = hostsim.SimBatch(text, 1)
        got, tmpl = sb.template_from_bits()
        assert got is not None, name  # (every value of a static observation is -1 / 0 / 1, at most one mark per source)
        assert np.array_equal(got, tmpl), name
        packed, section = sb.tables_from_packed()  # (tables.h off_packed: the table section as the multi-map split-row launches read it)
        assert packed is not None and np.array_equal(packed, section), name
        seen += int((tmpl == -1).sum() > 0)
        if sb.map.n_sources:
            cur = int(sb.map.sources()[0].agent_id)
            for c in range(sb.map.n_agents):
                if c == cur:
                    continue
                try:
                    sb.set_source(0, colour=c, enabled=False)
                except Exception:
                    continue  # (a colour whose beam would cross another agent's start is refused)
                got, tmpl2 = sb.template_from_bits()
                assert got is not None and np.array_equal(got, tmpl2), (name, "after set_source")
                packed, section = sb.tables_from_packed()
                assert packed is not None and np.array_equal(packed, section), (name, "after set_source")
                recoloured += int(not np.array_equal(tmpl, tmpl2))
                break
    assert seen > 10 and recoloured > 5
    # ... and on a few hundred generated maps of every shape class (1 - 14 agents, 0 - 20 sources, beams up to the map's width, voids, many gems)
    rng = np.random.default_rng(11)
    checked = 0
    for k in range(400):
        h, w = int(rng.integers(3, 20)), int(rng.integers(3, 20))
        agents = int(rng.integers(1, min(14, max(1, h * w // 6)) + 1))
        try:
            text = mapgen.generate(h, w, agents, int(rng.integers(0, 21)), n_gems=int(rng.integers(0, 12)), wall_fraction=float(rng.uniform(0.0, 0.2)),
                                   n_voids=int(rng.integers(0, 4)), seed=7000 + k, max_beam=int(rng.integers(3, 40)))
            sb = hostsim.SimBatch(text, 1)
        except (RuntimeError, hostsim.SimError):
            continue  # (no placement for these sizes, or beyond the static limits)
        got, tmpl = sb.template_from_bits()
        packed, section = sb.tables_from_packed()
        assert got is not None and np.array_equal(got, tmpl), text
        assert packed is not None and np.array_equal(packed, section), text
        checked += 1
    assert checked > 200, checked
