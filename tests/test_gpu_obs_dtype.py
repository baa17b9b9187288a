"""The layered observation in the learner's element type (lle_batch_options.obs_dtype: fp16 / bf16 / fp32 = the reference's,
python/lle/observations.py:223).  The kernels build the row as int8 in LDS and widen at the store; the bar: the CONTENT of LLE_BUF_OBS --
padding included -- equals the int8 tensor cast to that type, every step, in every kernel that writes the batch's rows (step kernel in
every mode that carries the stream, world kernel: reset / observe / set_state / source updates), and the oracle's tensor directly."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.parity_util import EXTRA_MAPS, legal_colours

pytestmark = pytest.mark.gpu

MAPS = {"level6": LEVELS[6], "level3": LEVELS[3], "level1": LEVELS[1], "nested": EXTRA_MAPS["nested"], "colour_alias": EXTRA_MAPS["colour_alias"],
        "many_agents": EXTRA_MAPS["many_agents"], "gen_20_lasers": EXTRA_MAPS["gen_20_lasers"], "config5_32x32": EXTRA_MAPS["config5_32x32"]}


def _dtypes():
    import torch
    return [torch.float16, torch.bfloat16, torch.float32]


def _same(wide, narrow, where):
    import torch
    assert wide.obs_rows.dtype == wide.obs_dtype and wide.obs_rows.shape == narrow.obs_rows.shape
    assert torch.equal(wide.obs_rows, narrow.obs_rows.to(wide.obs_dtype)), where


@pytest.mark.parametrize("name", list(MAPS))
def test_widened_rows_equal_the_int8_rows_cast(oracle_mod, name):
    """Single steps (sampled / given actions, auto-reset, incremental rows, steps without observation), reset with a mask, observe,
    set_state, the map's sources updated, per-environment sources, a fused rollout into a ring, two maps: every path that writes the rows."""
    import torch

    from lle_amd import BatchedWorld

    text = MAPS[name]
    n = 600 + 7
    for dt in _dtypes():
        a, b = BatchedWorld(text, n, obs_dtype=dt), BatchedWorld(text, n)
        assert a.obs.dtype == dt and a.obs.shape == b.obs.shape
        _same(a, b, f"{name} {dt} after creation")
        ob = oracle_mod.OracleBatch(text, n) if dt == torch.float32 else None
        rng = np.random.default_rng(3)
        for t in range(14):
            if t == 5:
                acts = torch.from_numpy(rng.integers(0, 6, size=(n, a.map.n_agents), dtype=np.uint8)).cuda()
                for w in (a, b):
                    w.step(acts, auto_reset=True)
                if ob:
                    ob.step(acts.cpu().numpy(), auto_reset=True, want_obs=False)
            else:
                kw = dict(sample=True, auto_reset=t >= 3, seed=5, t=t, incremental_obs=(t % 4 == 2), write_obs=(t != 7))
                for w in (a, b):
                    w.step(**kw)
                if ob:
                    ostep = ob.step(None, auto_reset=t >= 3, seed=5, t=t, want_obs=(t != 7))
                    if t != 7:  # the reference's own dtype against the oracle's tensor directly
                        C, H, W = ob.C, ob.H, ob.W
                        got = a.obs_rows[:, : C * H * W].cpu().numpy().reshape(n, C, H, W)
                        assert got.dtype == np.float32 and np.array_equal(got, ostep["obs"].astype(np.float32)), (name, t)
            _same(a, b, f"{name} {dt} t={t}")
        mask = torch.from_numpy((rng.random(n) < 0.5).astype(np.uint8))
        for w in (a, b):
            w.reset(mask)
        _same(a, b, f"{name} {dt} masked reset")
        for w in (a, b):
            w.step(sample=True, seed=6, t=0)
            w.observe()
        _same(a, b, f"{name} {dt} observe")
        A, G = a.map.n_agents, a.map.n_gems
        pos = torch.from_numpy(np.stack([rng.integers(0, a.map.height, size=(n, A)), rng.integers(0, a.map.width, size=(n, A))], axis=-1).astype(np.uint8))
        gems, alive = torch.from_numpy(rng.random((n, G)) < 0.3), torch.from_numpy(rng.random((n, A)) < 0.8)
        for w in (a, b):
            w.set_state(pos, gems, alive)
            w.observe()
        _same(a, b, f"{name} {dt} set_state")
        for w in (a, b):
            w.reset()
        if a.map.n_sources:
            for w in (a, b):
                w.map.set_source(0, enabled=False)
                w.update_sources()
            _same(a, b, f"{name} {dt} update_sources")
            L = a.map.n_sources
            colours = legal_colours(a.map, torch.from_numpy(rng.integers(0, A, size=(n, L), dtype=np.uint8)))
            enabled = torch.from_numpy(rng.integers(0, 1 << min(L, 30), size=n).astype(np.int32))
            for w in (a, b):
                w.set_sources(colours=colours, enabled=enabled)
            _same(a, b, f"{name} {dt} set_sources")
            for t in range(6):
                for w in (a, b):
                    w.step(sample=True, auto_reset=True, seed=8, t=t, incremental_obs=(t == 4))
                _same(a, b, f"{name} {dt} per-env sources t={t}")
        ra, rb = a.make_ring(3), b.make_ring(3)
        assert ra["obs_rows"].dtype == dt
        for w, r in ((a, ra), (b, rb)):
            w.rollout(7, auto_reset=True, seed=9, t=0, ring=r, ring_pos=0)
        assert torch.equal(ra["obs_rows"], rb["obs_rows"].to(dt)), (name, dt, "ring")
        for w in (a, b):
            w.rollout(3, auto_reset=True, seed=9, t=7)   # in place
        _same(a, b, f"{name} {dt} rollout in place")
        for k in ("pos", "bits", "gems", "beams", "avail", "events", "evcount", "done", "reward"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (name, dt, k)
        assert a.stats() == b.stats()


def test_two_maps_and_split_rows_widened():
    import torch

    from lle_amd import BatchedWorld, mapgen

    texts = [mapgen.config5(s) for s in range(2)]
    n = 2 * 128
    for dt in _dtypes():
        a, b = BatchedWorld(texts, n, obs_dtype=dt), BatchedWorld(texts, n)
        assert a.kernel_info()["lds_bytes"] == b.kernel_info()["lds_bytes"]
        for t in range(8):
            for w in (a, b):
                w.step(sample=True, auto_reset=True, seed=2, t=t, incremental_obs=(t % 3 == 1))
            _same(a, b, f"two maps {dt} t={t}")
        for w in (a, b):
            w.rollout(4, auto_reset=True, seed=2, t=8)
        _same(a, b, f"two maps {dt} rollout")


@pytest.mark.parametrize("config", ["cfg3", "cfg5"])
def test_full_size_configs_widened(config):
    """BASELINE configs[2] and [4] at their full batch of 65 536 environments, fp16 and fp32: every byte of the rows against the int8 batch
    stepped alongside (itself checked against the oracle at this size in tests/test_gpu_parity.py), every step."""
    import torch

    from lle_amd import BatchedWorld, mapgen

    text = LEVELS[6] if config == "cfg3" else mapgen.config5(0)
    n = 65536
    b = BatchedWorld(text, n)
    for dt in (torch.float16, torch.float32):
        a = BatchedWorld(text, n, obs_dtype=dt)
        b.reset()
        for t in range(6 if config == "cfg3" else 3):
            a.step(sample=True, auto_reset=True, seed=1234, t=t)
            b.step(sample=True, auto_reset=True, seed=1234, t=t)
            for lo in range(0, n, 16384):  # (compared in slices: the cast of 1.3 GB of int8 rows to fp32 need not exist whole)
                assert torch.equal(a.obs_rows[lo:lo + 16384], b.obs_rows[lo:lo + 16384].to(dt)), (config, dt, t, lo)
        assert torch.equal(a.pos, b.pos) and torch.equal(a.beams, b.beams) and a.stats()["env_steps"] == (6 if config == "cfg3" else 3) * n
        del a
        torch.cuda.empty_cache()


def test_the_facade_returns_float32_from_the_device(oracle_mod):
    """lle_amd.observations.Layered.observe: float32 like the reference (python/lle/observations.py:223,266), and the tensor is float32
    when it leaves the kernels -- `World.layered_observation` hands over what the device wrote."""
    from lle_amd import World
    from lle_amd.observations import Layered
    from oracle import observers as oo

    w = World.level(6)
    gen = Layered(w)
    raw = w.layered_observation()
    assert raw.dtype == np.float32 and w._batch.obs.dtype.is_floating_point and w._batch.obs.element_size() == 4
    obs = gen.observe()
    assert obs.dtype == np.float32 and obs.shape == (4, 12, 12, 13)
    ow = oracle_mod.OracleWorld.level(6)
    assert np.array_equal(obs, oo.layered_observe(ow))
    import random
    random.seed(1)
    for _ in range(12):
        acts = [random.choice(a) for a in w.available_actions()]
        w.step(acts)
        ow.step([a.value for a in acts])
        assert np.array_equal(gen.observe(), oo.layered_observe(ow))


def test_options_are_validated():
    import ctypes as C

    from lle_amd import Map, _capi

    L = _capi.lib()
    m = Map(level=1)
    handles = (C.c_void_p * 1)(m.h)
    bad = _capi.BatchOptions(7)
    assert L.lle_batch_arena_bytes_opt(handles, 1, 64, C.byref(bad)) < 0
    assert not L.lle_batch_create_opt(handles, 1, 64, 0, None, 0, C.byref(bad), None)
    zero = _capi.BatchOptions(1)
    zero.struct_bytes = 0
    assert not L.lle_batch_create_opt(handles, 1, 64, 0, None, 0, C.byref(zero), None)
    i8 = L.lle_batch_arena_bytes(m.h, 64)
    assert L.lle_batch_arena_bytes_opt(handles, 1, 64, None) == i8
    f32 = L.lle_batch_arena_bytes_opt(handles, 1, 64, C.byref(_capi.BatchOptions(_capi.LLE_DTYPE_F32)))
    assert f32 - i8 >= 3 * 64 * m.obs_stride and f32 - i8 < 3 * 64 * m.obs_stride + 1024
    h = L.lle_batch_create_opt(handles, 1, 64, 0, None, 0, C.byref(_capi.BatchOptions(_capi.LLE_DTYPE_BF16)), None)
    assert h and L.lle_batch_obs_dtype(h) == _capi.LLE_DTYPE_BF16
    d = _capi.BufferDesc()
    assert L.lle_batch_get_buffer(h, _capi.BUFFER_NAMES.index("obs"), C.byref(d)) == 0 and d.elem_bytes == 2 and d.bytes == 2 * 64 * m.obs_stride
    L.lle_batch_free(h)


@pytest.mark.parametrize("name", ["level6", "level3", "nested", "many_agents", "config5_32x32"])
def test_every_observation_builder_in_the_batch_dtype(name, monkeypatch):
    """lle_batch_observe_as on a batch created with obs_dtype: layered, layered-padded, perspective (one launch and one launch per observer) and
    partial 3x3 / 5x5 / 7x7 / 9x9 -- the lane kernel in its window-table and bitmap forms, the window and projection kernels -- come in the
    batch's element type, the values of the int8 batch's outputs; per-environment sources; the partial observation written by the STEP launch
    (lle_env_outputs.partial, step kernel MODE 9); the state vector stays float32."""
    import torch

    from lle_amd import BatchedWorld, _capi

    text = MAPS[name]
    n = 333
    kinds = [(_capi.LLE_OBS_LAYERED, 0), (_capi.LLE_OBS_LAYERED_PADDED, 2), (_capi.LLE_OBS_PERSPECTIVE, 0)] + [(_capi.LLE_OBS_PARTIAL, k) for k in (3, 5, 7, 9)]
    for dt in _dtypes():
        a, b = BatchedWorld(text, n, obs_dtype=dt), BatchedWorld(text, n)
        for t in range(9):
            for w in (a, b):
                w.step(sample=True, auto_reset=t >= 2, seed=3, t=t)

        def compare(where):
            for kind, param in kinds:
                da, db = a.obs_desc(kind, param), b.obs_desc(kind, param)
                assert da.supported == db.supported and int(da.elem_bytes) == dt.itemsize and int(db.elem_bytes) == 1 and int(da.bytes) == int(db.bytes) * dt.itemsize
                assert [int(da.stride[q]) for q in range(da.ndim)] == [int(db.stride[q]) for q in range(db.ndim)]  # (strides are in elements)
                if not da.supported:
                    continue
                xa, xb = a.observe_as(kind, param), b.observe_as(kind, param)
                assert xa.dtype == dt and xb.dtype == torch.int8 and torch.equal(xa, xb.to(dt)), (name, dt, kind, param, where)
            sa, sb = a.observe_as(_capi.LLE_OBS_STATE), b.observe_as(_capi.LLE_OBS_STATE)
            assert sa.dtype == torch.float32 and torch.equal(sa, sb)
        compare("rule")
        for env in (dict(LLE_PARTIAL_KERNEL="window"), dict(LLE_PARTIAL_KERNEL="project"), dict(LLE_PARTIAL_NO_SETS="1"), dict(LLE_PARTIAL_SETS="1", LLE_PARTIAL_E="2")):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            _capi.refresh_tuning()
            try:
                compare(str(env))
            finally:
                for k in env:
                    monkeypatch.delenv(k)
                _capi.refresh_tuning()
        if a.map.n_beam_words <= 8:  # the step launch writes the partial observation itself (MODE 9)
            for k in (3, 7):
                (ba, va), (bb, vb) = a.partial_buffer(k), b.partial_buffer(k)
                done = torch.empty(n, dtype=torch.uint8, device="cuda")
                ea, eb = a.make_env_outputs(done=done, partial=ba, partial_k=k), b.make_env_outputs(done=done, partial=bb, partial_k=k)
                for t in range(9, 13):
                    a.step(sample=True, auto_reset=True, seed=3, t=t, env_out=ea)
                    b.step(sample=True, auto_reset=True, seed=3, t=t, env_out=eb)
                    assert va.dtype == dt and torch.equal(va, vb.to(dt)), (name, dt, k, t)
                    assert torch.equal(va, a.observe_as(_capi.LLE_OBS_PARTIAL, k))
        if a.map.n_sources:
            rng = np.random.default_rng(5)
            colours = legal_colours(a.map, torch.from_numpy(rng.integers(0, a.map.n_agents, size=(n, a.map.n_sources), dtype=np.uint8)))
            for w in (a, b):
                w.set_sources(colours=colours)
                w.step(sample=True, auto_reset=True, seed=4, t=0)
            compare("per-env sources")


def test_observation_builders_widened_on_blocks_of_maps():
    import torch

    from lle_amd import BatchedWorld, _capi, mapgen

    texts = [mapgen.generate(9, 11, 3, 4, 3, n_voids=2, seed=40 + s) for s in range(6)]
    n = 6 * 24
    for dt in (torch.float16, torch.float32):
        a, b = BatchedWorld(texts, n, obs_dtype=dt), BatchedWorld(texts, n)
        for t in range(6):
            for w in (a, b):
                w.step(sample=True, auto_reset=True, seed=6, t=t)
        for kind, param in ((_capi.LLE_OBS_LAYERED, 0), (_capi.LLE_OBS_LAYERED_PADDED, 1), (_capi.LLE_OBS_PERSPECTIVE, 0), (_capi.LLE_OBS_PARTIAL, 3), (_capi.LLE_OBS_PARTIAL, 5)):
            if not a.obs_desc(kind, param).supported:
                continue
            assert torch.equal(a.observe_as(kind, param), b.observe_as(kind, param).to(dt)), (dt, kind, param)
