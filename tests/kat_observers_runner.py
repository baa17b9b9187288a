"""Runs the cases of tests/golden/kat_observers.json against an adapter (oracle, host simulator, HIP kernels).

Adapter protocol:  n_agents, n_gems, height, width; reset(); step(actions); get_state() -> (positions, gems, alive);
    observe(kind, param) -> float32 array with the reference's leading (n_agents, ...) axis (raises IndexError like
    the reference); announced_shape(kind, param) -> tuple; avail(walkable_lasers) -> bool (n_agents, 5)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "kat_observers.json")) as f:
        return json.load(f)["cases"]


def layer_table(kind, param, A):
    if kind == "partial":
        return {"A0": 0, "WALL": A, "LASER_0": A + 1, "GEM": 2 * A + 1, "EXIT": 2 * A + 2}
    n = A + (param if kind == "layered-padded" else 0)
    return {"A0": 0, "LASER_0": n, "WALL": 2 * n, "VOID": 2 * n + 1, "GEM": 2 * n + 2, "EXIT": 2 * n + 3}


def _index(entry, layers):
    out = []
    for x in entry:
        if isinstance(x, str):
            out.append(layers[x])
        elif isinstance(x, list) and len(x) == 2 and isinstance(x[0], str):
            out.append(layers[x[0]] + x[1])
        elif isinstance(x, list):
            out.append(slice(x[0], x[1]))
        else:
            out.append(x)
    return tuple(out)


def expected_shape(kind, param, A, G, H, W):
    """Per-agent shapes announced by the reference generators (observations.py:161-164, 212, 285, 317)."""
    if kind in ("state", "normalized-state"):
        return (3 * A + G,)
    if kind == "partial":
        return (2 * A + 3, param, param)
    n = A + (param if kind == "layered-padded" else 0)
    if kind == "flattened":
        return ((2 * A + 4) * H * W,)
    return (2 * n + 4, H, W)


def run_check(ad, op, name):
    kind, param = op["obs"], op.get("param", 0)
    A = ad.n_agents
    layers = layer_table(kind, param, A)
    if "shape" in op:
        assert tuple(ad.announced_shape(kind, param)) == tuple(op["shape"]), f"{name}: announced shape"
    needs_obs = any(k in op for k in ("at", "all", "equal", "allclose", "count_nonzero", "roundtrip_state",
                                      "perspective_of_layered", "shape_consistent"))
    if not needs_obs:
        return
    obs = ad.observe(kind, param)
    assert obs.dtype == np.float32, f"{name}: dtype {obs.dtype}"
    if op.get("shape_consistent"):
        shape = tuple(ad.announced_shape(kind, param))
        assert shape == expected_shape(kind, param, A, ad.n_gems, ad.height, ad.width), f"{name}: {kind} announced {shape}"
        for a in range(A):
            assert obs[a].shape == shape, f"{name}: {kind} shape is not consistent: announced {shape} but returned {obs[a].shape}"
    for entry in op.get("at", []):
        idx = _index(entry[:-1], layers)
        assert obs[idx] == entry[-1], f"{name}: {kind}{list(idx)} = {obs[idx]}, expected {entry[-1]}"
    for entry in op.get("all", []):
        idx = _index(entry[:-1], layers)
        assert np.all(obs[idx] == entry[-1]), f"{name}: {kind}{list(idx)} not all {entry[-1]}"
    for entry in op.get("count_nonzero", []):
        idx = _index(entry[:-1], layers)
        assert np.count_nonzero(obs[idx]) == entry[-1], f"{name}: {kind}{list(idx)} nonzero count"
    if "equal" in op:
        assert np.array_equal(np.array(op["equal"]), obs), f"{name}: {obs}"
    if "allclose" in op:
        assert np.allclose(np.array(op["allclose"]), obs), f"{name}: {obs}"
    if op.get("roundtrip_state"):
        # StateGenerator.to_world_state (observations.py:152-154) + WorldState.from_array (pyworld_state.rs:103-131)
        data = obs[0].copy()
        dims = np.array(([ad.height, ad.width] if kind == "normalized-state" else [1.0, 1.0]) * A)
        data[: 2 * A] = data[: 2 * A] * dims
        vals = data.tolist()
        pos = [(int(vals[2 * i]), int(vals[2 * i + 1])) for i in range(A)]
        gems = [vals[2 * A + i] == 1.0 for i in range(ad.n_gems)]
        alive = [vals[2 * A + ad.n_gems + i] == 1.0 for i in range(A)]
        spos, sgems, salive = ad.get_state()
        assert (pos, gems, alive) == ([tuple(p) for p in spos], list(sgems), list(salive)), f"{name}: roundtrip"
    if op.get("perspective_of_layered"):
        layered = ad.observe("layered", 0)
        assert obs.shape == (A, *ad.announced_shape("perspective", 0))
        positions = ad.get_state()[0]
        A0, L0 = layers["A0"], layers["LASER_0"]
        for k, p in enumerate(positions):
            expected = np.copy(layered[k])
            expected[[A0, A0 + k]] = expected[[A0 + k, A0]]
            expected[[L0, L0 + k]] = expected[[L0 + k, L0]]
            np.testing.assert_array_equal(obs[k], expected)
            assert obs[k, A0, p[0], p[1]] == 1.0


def run_case(make_adapter, case):
    ad = make_adapter(case)
    name = case["name"]
    for op in case["script"]:
        if op["op"] == "reset":
            ad.reset()
        elif op["op"] == "step":
            ad.step(op["actions"])
        elif op["op"] == "check":
            run_check(ad, op, name)
        elif op["op"] == "check_avail":
            av = ad.avail(op["walkable_lasers"])
            assert av.shape == (ad.n_agents, 5) and av.dtype == bool
            for a, act in op["true"]:
                assert av[a, act], f"{name}: agent {a} action {act} should be available"
            for a, act in op["false"]:
                assert not av[a, act], f"{name}: agent {a} action {act} should not be available"
        else:
            raise ValueError(op)


# ---------------------------------------------------------------------------------------------------- adapters
_KIND_CODE = {"layered": 0, "flattened": 0, "layered-padded": 1, "perspective": 2, "partial": 3, "state": 4, "normalized-state": 5}


class _Base:
    def announced_shape(self, kind, param):
        return expected_shape(kind, param, self.n_agents, self.n_gems, self.height, self.width)

    def _tile(self, kind, param, single):
        """Give an engine's single copy the reference's leading axis (np.tile, observations.py:151,266)."""
        A = self.n_agents
        single = np.asarray(single).astype(np.float32)
        if kind in ("partial", "perspective"):
            return single
        if kind == "flattened":
            return np.tile(single.reshape(1, -1), (A, 1))
        reps = A + (param if kind == "layered-padded" else 0)
        return np.tile(single, (reps,) + (1,) * single.ndim)


class OracleAdapter(_Base):
    """oracle/observers.py on an OracleWorld."""

    def __init__(self, oracle_mod, case):
        from oracle.levels import LEVELS
        self.w = oracle_mod.OracleWorld(LEVELS[case["level"]] if "level" in case else case["map"])
        w = self.w
        self.n_agents, self.n_gems, self.height, self.width = w.n_agents, w.n_gems, w.height, w.width

    def reset(self):
        self.w.reset()

    def step(self, actions):
        self.w.step(actions)

    def get_state(self):
        return self.w.get_state()

    def observe(self, kind, param):
        from oracle import observers as oo
        w = self.w
        return {"state": lambda: oo.state_observe(w, False), "normalized-state": lambda: oo.state_observe(w, True),
                "layered": lambda: oo.layered_observe(w), "flattened": lambda: oo.flattened_observe(w),
                "layered-padded": lambda: oo.layered_padded_observe(w, param), "partial": lambda: oo.partial_observe(w, param),
                "perspective": lambda: oo.perspective_observe(w)}[kind]()

    def avail(self, walkable):
        from oracle import observers as oo
        return oo.available_actions(self.w, walkable)


class HostsimAdapter(_Base):
    """Host build of observers_logic.hpp / compile_view (tests/hostsim)."""

    def __init__(self, case):
        from tests import hostsim
        if "level" in case:
            self.w = hostsim.SimWorld(level=case["level"])
        else:
            self.w = hostsim.SimWorld(case["map"])
        w = self.w
        self.n_agents, self.n_gems, self.height, self.width = w.n_agents, w.n_gems, w.height, w.width

    def reset(self):
        self.w.reset()

    def step(self, actions):
        self.w.step(actions)

    def get_state(self):
        return self.w.positions(), self.w.gems_collected(), self.w.alive()

    def observe(self, kind, param):
        out = self.w.b.observe_as(_KIND_CODE[kind], param)
        if out is None:
            raise IndexError(kind)
        return self._tile(kind, param, out[0])

    def avail(self, walkable):
        return self.w.b.available_actions(walkable)[0]
