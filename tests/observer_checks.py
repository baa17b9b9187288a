"""Shared comparison of an engine's observation builders (host simulator or HIP kernels) with oracle/observers.py."""
import numpy as np

from lle_amd import _capi
from oracle import observers as oo

KINDS = [("layered", _capi.LLE_OBS_LAYERED, 0), ("padded1", _capi.LLE_OBS_LAYERED_PADDED, 1),
         ("padded3", _capi.LLE_OBS_LAYERED_PADDED, 3), ("perspective", _capi.LLE_OBS_PERSPECTIVE, 0),
         ("partial3", _capi.LLE_OBS_PARTIAL, 3), ("partial5", _capi.LLE_OBS_PARTIAL, 5), ("partial7", _capi.LLE_OBS_PARTIAL, 7),
         ("state", _capi.LLE_OBS_STATE, 0), ("normalized-state", _capi.LLE_OBS_NORMALIZED_STATE, 0)]


def oracle_observe(w, kind, param):
    """The reference's tensor for one env, reduced to the part the engine materialises (one copy where the reference
    tiles n_agents identical copies), or None when the reference raises IndexError."""
    try:
        if kind == _capi.LLE_OBS_LAYERED:
            return oo.layered_observe(w)[0]
        if kind == _capi.LLE_OBS_LAYERED_PADDED:
            full = oo.layered_padded_observe(w, param)
            assert all(np.array_equal(full[0], full[k]) for k in range(1, full.shape[0]))
            return full[0]
        if kind == _capi.LLE_OBS_PERSPECTIVE:
            return oo.perspective_observe(w)
        if kind == _capi.LLE_OBS_PARTIAL:
            return oo.partial_observe(w, param)
        full = oo.state_observe(w, kind == _capi.LLE_OBS_NORMALIZED_STATE)
        return full[0]
    except IndexError:
        return None


def compare_all(engine_observe, engine_avail, oracle_batch, envs, where=""):
    """engine_observe(kind, param) -> array [n, ...] or None; engine_avail(walkable) -> bool [n, A, 5]."""
    for name, kind, param in KINDS:
        got = engine_observe(kind, param)
        for e in envs:
            want = oracle_observe(oracle_batch.world(e), kind, param)
            if want is None:
                assert got is None, f"{where} {name}: the reference raises IndexError here"
                continue
            assert got is not None, f"{where} {name}: engine refused a supported observation"
            g = np.asarray(got[e])
            assert g.shape == want.shape, f"{where} {name} env {e}: shape {g.shape} != {want.shape}"
            if kind in (_capi.LLE_OBS_STATE, _capi.LLE_OBS_NORMALIZED_STATE):
                # bit-exact: the reference rounds the float64 quotient to float32
                assert g.dtype == np.float32 and np.array_equal(g.view(np.uint32), want.view(np.uint32)), f"{where} {name} env {e}"
            else:
                assert np.array_equal(g.astype(np.float32), want), f"{where} {name} env {e}"
    for walkable in (True, False):
        got = engine_avail(walkable)
        for e in envs:
            want = oo.available_actions(oracle_batch.world(e), walkable)
            assert np.array_equal(np.asarray(got[e]).astype(bool), want), f"{where} avail walkable={walkable} env {e}"
