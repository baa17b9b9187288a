"""Instantiation coverage: EVERY kernel the library's dispatch can reach is launched here and checked against the oracle.

The library is 555 kernels: step_kernel<G, LM, MODE, ML1, LX> in 520 instantiations (lanes per environment x beam registers x launch
mode x single-layer maps x exact source count), world_kernel<AM, LM, MODE> in 24, eleven observers.  Round 4 met a COMPILER fault that hit
one of them (a live-range-split copy ahead of an exec restore in step_kernel<4,4,4,false,-1>; DESIGN.md "ISA tripwire"): "bit-exact" has
to be a property of every instantiation, not of the ones the hand-picked maps of the other suites happen to take.

* one test per step-kernel instantiation: a map built to hit it (tests/instantiation_maps.py), driven into the launch mode, >= 20 steps,
  every buffer against the oracle (world.rs:435-505 semantics: state, ordered events, availability, the int8 observation) -- and the
  instantiation asserted to be among lle_debug_launched();
* the world kernels (reset / set_state / observe / source updates / the lane-per-env diagnostic step) and the observers likewise;
* last: lle_debug_reachable() -- the launchers' own dispatch walked with launches suppressed -- minus lle_debug_launched() must be empty.
  The report goes to gpurun_out/r05_instantiation_coverage.md (committed as profiles/r05_instantiation_coverage.md)."""
import os

import numpy as np
import pytest

from tests import instantiation_maps as im
from tests.parity_util import assert_state_equal, assert_step_equal, legal_colours, unpack_engine

pytestmark = pytest.mark.gpu

N = 203          # ragged: not a multiple of any wavefront's environment count
PER_MAP = 112    # two-map batches: a multiple of 16, not of 64 (workgroups narrowed so that none straddles the maps)
STEPS = 22

CASES = [(A, L, cross, mode) for (A, _g) in im.AGENT_CLASSES for (L, cross) in im.SOURCE_CLASSES for mode in range(10)
         if mode < 6 or im.lm_of(L) <= 8]  # (row heads and the partial writer serve maps with at most 8 beam words)
assert len(CASES) == 520


def _id(case):
    A, L, cross, mode = case
    return f"A{A}-L{L}{'x' if cross else ''}-mode{mode}"


def _text(A, L, cross, variant=0):
    return im.build(A, L, cross, seed=100 * A + L, variant=variant)


def _check(bw, ob, ostep, where, lo=None, hi=None):
    bufs = bw.host_buffers()
    if lo is not None:
        bufs = {k: v[lo:hi] for k, v in bufs.items()}
    eng = unpack_engine(bufs, *ob.dims)
    if ostep is not None:
        assert_step_equal(eng, ostep, where)
    assert_state_equal(eng, ob.dump(), where)


def _random_actions(rng, n, A):
    return rng.integers(0, 6, size=(n, A), dtype=np.uint8)  # unavailable ones and 5 (not an Action) included: refused per env


def _single_steps(oracle_mod, bw, obs, per, tag, pes_mirrors=None):
    """STEPS single steps of `bw` (one map: obs = [oracle batch], per = n; two maps: one oracle batch per block) against the oracle:
    sampled actions with and without auto-reset, given actions (some refused) now and then."""
    import torch
    rng = np.random.default_rng(5)
    n, A = bw.n_envs, bw.map.n_agents
    for t in range(STEPS):
        auto = t >= 5
        if t % 6 == 4:
            acts = _random_actions(rng, n, A)
            bw.step(torch.from_numpy(acts).cuda(), auto_reset=auto)
            osteps = [ob.step(acts[m * per:(m + 1) * per], auto_reset=auto) for m, ob in enumerate(obs)]
        else:
            bw.step(sample=True, auto_reset=auto, seed=77, t=t, env_offset=3)
            osteps = [ob.step(None, auto_reset=auto, seed=77, t=t, env_offset=3 + m * per) for m, ob in enumerate(obs)]
        for m, ob in enumerate(obs):
            _check(bw, ob, osteps[m], f"{tag} t={t} map {m}", m * per, (m + 1) * per)


def _rollouts(oracle_mod, bw, obs, per, tag):
    """Fused rollouts (chunks of 7, 6 and 9 steps = 22) into a ring of four slots: the final state of every chunk and the ring's
    surviving slots (observation, actions) against the oracle stepped one step at a time."""
    R, t0 = 4, 0
    ring = bw.make_ring(R)
    for chunk in (7, 6, 9):
        bw.rollout(chunk, auto_reset=True, seed=91, t=t0, env_offset=11, ring=ring, ring_pos=t0)
        steps = [[ob.step(None, auto_reset=True, seed=91, t=t0 + j, env_offset=11 + m * per) for j in range(chunk)] for m, ob in enumerate(obs)]
        obs_ring, act_ring = ring["obs"].cpu().numpy(), ring["actions"].cpu().numpy()
        for m, ob in enumerate(obs):
            sl = slice(m * per, (m + 1) * per)
            bufs = {k: v[sl] for k, v in bw.host_buffers().items()}
            eng = unpack_engine(bufs, *ob.dims)
            assert_state_equal(eng, ob.dump(), f"{tag} after a rollout of {chunk}, map {m}")
            for key in ("err", "ev_count"):
                assert np.array_equal(eng[key], steps[m][-1][key]), (tag, key)
            for j in range(chunk - R, chunk):
                slot = (t0 + j) % R
                assert np.array_equal(obs_ring[slot][sl], steps[m][j]["obs"]), (tag, chunk, j, "ring obs")
                assert np.array_equal(act_ring[slot][sl], steps[m][j]["actions"]), (tag, chunk, j, "ring actions")
        t0 += chunk


def _set_random_sources(bw, ob, n, seed):
    """Per-environment colours and enabled flags (lle_batch_set_sources), mirrored onto the oracle's worlds."""
    import torch

    from tests.test_gpu_env_sources import Mirror
    A, L = bw.map.n_agents, bw.map.n_sources
    rng = np.random.default_rng(seed)
    if L == 0:
        bw.set_sources(enabled=torch.zeros(n, dtype=torch.int32))  # (no source to colour: the batch still takes its per-env kernels)
        return
    colours = legal_colours(bw.map, rng.integers(0, A, size=(n, L), dtype=np.uint8))
    enabled = rng.integers(0, 1 << min(L, 31), size=n, dtype=np.int64).astype(np.int32)
    if L > 31:
        enabled |= np.int32(-(2 ** 31)) * (rng.integers(0, 2, size=n) == 1)
    bw.set_sources(torch.from_numpy(colours), torch.from_numpy(enabled))
    Mirror(ob, n, L).apply(colours, enabled.astype(np.int64) & 0xFFFFFFFF, None)
    assert int(bw.err.max()) == 0


@pytest.mark.parametrize("case", CASES, ids=_id)
def test_step_kernel_instantiation(oracle_mod, monkeypatch, case):
    import torch

    from lle_amd import BatchedWorld, _capi
    from oracle import observers as oo

    A, L, cross, mode = case
    tag = _id(case)
    # modes 6 / 7 / 8 are 0 / 4 / 5 with the rows' head lines stored ahead of the state machine: forced on / off (the rule picks by launch size)
    monkeypatch.setenv("LLE_ROW_HEADS", "1" if mode in (6, 7, 8) else "0")
    two_maps = mode in (2, 4, 7)
    pes = mode in (3, 5, 8)
    texts = [_text(A, L, cross, 0), _text(A, L, cross, 1)] if two_maps else [_text(A, L, cross)]
    per = PER_MAP if two_maps else N
    n = per * len(texts)
    bw = BatchedWorld(texts if two_maps else texts[0], n, row_align=128)
    assert bw.map.n_agents == A and bw.map.n_beam_words == L and (bw.map.max_cell_layers > 1) == cross
    obs = [oracle_mod.OracleBatch(t, per) for t in texts]
    for m, ob in enumerate(obs):
        _check(bw, ob, None, f"{tag} after reset map {m}", m * per, (m + 1) * per)
    if pes:
        _set_random_sources(bw, obs[0], n, seed=A + L)
        _check(bw, obs[0], None, f"{tag} after set_sources")
    if mode in (1, 2, 3):
        _rollouts(oracle_mod, bw, obs, per, tag)
    elif mode == 9:
        # the partial k x k observation written by the step launch (python/lle/observations.py:312-369), k = 3 and 5 alternating
        ob = obs[0]
        bufs = {k: bw.partial_buffer(k) for k in (3, 5)}
        state = torch.empty((n, 3 * A + bw.map.n_gems), dtype=torch.float32, device="cuda")
        rng = np.random.default_rng(2)
        for t in range(STEPS):
            k = (3, 5)[t & 1]
            out = bw.make_env_outputs(state=state, partial=bufs[k][0], partial_k=k)
            auto = t >= 5
            if t % 6 == 4:
                acts = _random_actions(rng, n, A)
                bw.step(torch.from_numpy(acts).cuda(), auto_reset=auto, env_out=out, write_obs=False)
                ostep = ob.step(acts, auto_reset=auto, want_obs=False)
            else:
                bw.step(sample=True, auto_reset=auto, seed=77, t=t, env_offset=3, env_out=out, write_obs=False)
                ostep = ob.step(None, auto_reset=auto, seed=77, t=t, env_offset=3, want_obs=False)
            eng = unpack_engine(bw.host_buffers(("pos", "bits", "gems", "beams", "avail", "actions", "err", "evcount", "events", "done")), *ob.dims)
            assert_step_equal(eng, ostep, f"{tag} t={t}", check_obs=False)
            assert_state_equal(eng, ob.dump(), f"{tag} t={t}")
            view = bufs[k][1].cpu().numpy()
            st = state.cpu().numpy()
            for e in range(0, n, 29):
                assert np.array_equal(view[e].astype(np.float32), oo.partial_observe(ob.world(e), k)), (tag, t, k, e)
                assert np.array_equal(st[e], oo.state_array(ob.world(e))), (tag, t, e)
    else:
        _single_steps(oracle_mod, bw, obs, per, tag)
    want = im.kernel_name(A, L, cross, mode)
    assert want in _capi.launched_kernels(), f"{tag} was meant to launch {want}"


# ---- world_kernel<AM, LM, MODE>: one environment per lane -- reset, set_state, observe, source updates, and (as a diagnostic) step
WORLD_CASES = [(3, 3, True, "world_kernel<4,4"), (7, 7, True, "world_kernel<8,8"), (13, 12, True, "world_kernel<16,16"), (3, 20, False, "world_kernel<16,32")]


@pytest.mark.parametrize("case", WORLD_CASES, ids=lambda c: c[3])
def test_world_kernel_instantiations(oracle_mod, case):
    import torch

    from lle_amd import BatchedWorld, _capi

    A, L, cross, prefix = case
    text = im.build(A, L, cross, seed=100 * A + L)
    n = N
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)                      # MODE 1: reset (at creation)
    _check(bw, ob, None, f"{prefix} reset")
    bw.set_envs_per_wave(8)                         # MODE 0: the lane-per-env step
    for t in range(STEPS):
        bw.step(sample=True, auto_reset=t >= 5, seed=13, t=t)
        _check(bw, ob, ob.step(None, auto_reset=t >= 5, seed=13, t=t), f"{prefix} step t={t}")
    rng = np.random.default_rng(1)                  # MODE 2: set_state on random requests
    pos = np.stack([rng.integers(0, ob.H, size=(n, A)), rng.integers(0, ob.W, size=(n, A))], axis=-1).astype(np.uint8)
    gems = rng.random((n, ob.G)) < 0.3
    alive = rng.random((n, A)) < 0.8
    bw.set_state(torch.from_numpy(pos), torch.from_numpy(gems), torch.from_numpy(alive))
    err = bw.err.cpu().numpy()
    codes = {"InvalidWorldState": 0x40, "OutOfWorldPosition": 0x41, "InvalidAgentPosition": 0x42}
    for e in range(n):
        try:
            ob.world(e).set_state([tuple(int(v) for v in p) for p in pos[e]], [bool(v) for v in gems[e]], [bool(v) for v in alive[e]])
            want = 0
        except oracle_mod.OracleError as ex:
            want = codes[ex.kind]
        assert int(err[e]) == want, (prefix, e)
    _check(bw, ob, None, f"{prefix} set_state")
    poisoned = err == 0x40  # (the reference keeps stale availability lists after a failed set_state: reset those worlds)
    bw.reset(torch.from_numpy(poisoned.astype(np.uint8)))
    for e in np.nonzero(poisoned)[0]:
        ob.world(int(e)).reset()
    bw.observe()                                    # MODE 3: observe
    eng = unpack_engine(bw.host_buffers(), *ob.dims)
    assert np.array_equal(eng["obs"], np.stack([ob.world(e).obs() for e in range(n)]))
    bw.map.set_source(0, enabled=False)             # MODE 4: the map's sources pushed to the batch
    bw.update_sources()
    for e in range(n):
        ob.world(e).set_source(0, enabled=False)
    _check(bw, ob, None, f"{prefix} update_sources")
    _set_random_sources(bw, ob, n, seed=3)          # MODE 5: per-environment sources
    # (the disabled flag of source 0 was broadcast above; the Mirror starts from the oracle's own flags)
    _check(bw, ob, None, f"{prefix} set_sources")
    for t in range(4):
        bw.step(sample=True, auto_reset=True, seed=14, t=t)
        _check(bw, ob, ob.step(None, auto_reset=True, seed=14, t=t), f"{prefix} per-env step t={t}")
    launched = _capi.launched_kernels()
    for mode in range(6):
        assert f"{prefix},{mode}>" in launched, (prefix, mode)


def test_observer_kernels(oracle_mod, monkeypatch):
    """Every kernel of observers.hip, each against oracle/observers.py (python/lle/observations.py:137-395, env.py:146-163): the three
    partial writers are forced in turn (LLE_PARTIAL_KERNEL), the row-fill probe under both store policies."""
    import torch

    from lle_amd import BatchedWorld, _capi
    from tests.observer_checks import compare_all

    text = im.build(3, 3, True, seed=303)
    n = N
    ob = oracle_mod.OracleBatch(text, n)
    bw = BatchedWorld(text, n)
    for t in range(9):
        bw.step(sample=True, auto_reset=t > 5, seed=2, t=t)
        ob.step(None, auto_reset=t > 5, seed=2, t=t, want_obs=False)

    def engine_observe(kind, param):
        try:
            return bw.observe_as(kind, param).cpu().numpy()
        except IndexError:
            return None
    for which in ("window", "project", "lanes"):
        monkeypatch.setenv("LLE_PARTIAL_KERNEL", which)
        compare_all(engine_observe, lambda w: bw.available_actions(w).cpu().numpy(), ob, range(0, n, 7), which)
    monkeypatch.delenv("LLE_PARTIAL_KERNEL")
    A = ob.A
    state = torch.empty((n, 3 * A + bw.map.n_gems), dtype=torch.float32, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    bw.env_outputs(state=state, done=done)
    from oracle import observers as oo
    for e in range(0, n, 7):
        assert np.array_equal(state[e].cpu().numpy(), oo.state_array(ob.world(e)))
    assert bw.stats()["env_steps"] == 9 * n
    rows = bw.obs_rows.clone()
    for policy in ("0", "1"):
        monkeypatch.setenv("LLE_WRITE_THROUGH", policy)
        bw.row_fill_prober(value=0x01010101)()
        assert bool((bw.obs_rows == 1).all())
    monkeypatch.delenv("LLE_WRITE_THROUGH")
    bw.observe()
    assert torch.equal(bw.obs_rows, rows)
    f16 = torch.empty(rows.numel(), dtype=torch.float16, device="cuda")
    assert _capi.lib().lle_probe_read_rows(rows.data_ptr(), f16.data_ptr(), rows.numel() // 16 * 16, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(f16[: rows.numel() // 16 * 16], rows.flatten()[: rows.numel() // 16 * 16].to(torch.float16))
    launched = set(_capi.launched_kernels())
    for name in ("view_observe_kernel", "partial_observe_kernel", "partial_project_kernel", "partial_lanes_kernel", "state_observe_kernel", "avail_kernel",
                 "env_outputs_kernel", "row_fill_probe_kernel<true>", "row_fill_probe_kernel<false>", "cast_rows_kernel", "stats_sum_kernel"):
        assert name in launched, name


def test_zz_every_reachable_instantiation_was_launched():
    """Runs last in this file: what the dispatch can reach (lle_debug_reachable: the launchers' own switch statements, walked for every
    agent count 1..16, beam-word count 0..32, crossing or not, mode) against what this process launched."""
    from lle_amd import _capi

    reachable, launched = set(_capi.reachable_kernels()), set(_capi.launched_kernels())
    missing = sorted(reachable - launched)
    stray = sorted(launched - reachable)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    by_kind = {}
    for name in sorted(reachable):
        by_kind.setdefault(name.split("<")[0], []).append(name)
    lines = ["# Instantiation coverage (tests/test_gpu_instantiations.py)", "",
             f"reachable through the dispatch (lle_debug_reachable): **{len(reachable)}** kernels; launched by this test process and checked against "
             f"the oracle: **{len(reachable & launched)}**; reachable but never launched: **{len(missing)}**; launched but not in the walk: {len(stray)}.", ""]
    lines += ["| kernel | reachable | launched |", "|---|---|---|"]
    for kind, names in by_kind.items():
        lines.append(f"| `{kind}` | {len(names)} | {sum(n in launched for n in names)} |")
    lines += ["", "Step-kernel instantiations by mode (reachable / launched):", ""]
    for mode in range(10):
        names = [n for n in by_kind.get("step_kernel", []) if n.split(",")[2] == str(mode)]
        lines.append(f"* MODE {mode}: {len(names)} / {sum(n in launched for n in names)}")
    if missing:
        lines += ["", "## never launched", ""] + [f"* `{n}`" for n in missing]
    with open(os.path.join(out_dir, "r05_instantiation_coverage.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not stray, stray
    assert not missing, f"{len(missing)} reachable instantiations were never launched: {missing[:12]}"
