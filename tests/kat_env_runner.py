"""Runs tests/golden/kat_env.json (the reference's LLE-level tests) against an adapter.

Adapter protocol: reset(); step(actions) -> (reward float32 array [1] or [4], done bool); set_state(positions, gems, alive);
metrics() -> {"has-arrived": [...], "is-alive": [...]}; done() -> bool; available() -> bool array [A, 5]."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "kat_env.json")) as f:
        return json.load(f)["cases"]


def run_case(make_adapter, case):
    ad = make_adapter(case)
    name = case["name"]
    for k, op in enumerate(case["script"]):
        if op["op"] == "reset":
            ad.reset()
        elif op["op"] == "set_state":
            alive = op["alive"] if op["alive"] is not None else [True] * len(op["positions"])
            ad.set_state([tuple(p) for p in op["positions"]], list(op["gems"]), alive)
        elif op["op"] == "step":
            reward, done = ad.step(op["actions"])
            if op["reward"] is not None:
                want = np.atleast_1d(np.array(op["reward"], dtype=np.float32))
                assert np.array_equal(np.asarray(reward, dtype=np.float32).reshape(-1), want), f"{name} op {k}: reward {reward} != {want}"
            if op["done"] is not None:
                assert bool(done) == op["done"], f"{name} op {k}: done {done}"
            if op["metrics"] is not None:
                assert ad.metrics() == op["metrics"], f"{name} op {k}: {ad.metrics()}"
        elif op["op"] == "expect":
            if op.get("done") is not None:
                assert bool(ad.done()) == op["done"], f"{name} op {k}: done {ad.done()}"
            for key in ("available", "derived_available"):
                if op.get(key) is not None:
                    want = np.zeros((len(op[key]), 5), dtype=bool)
                    for a, acts in enumerate(op[key]):
                        want[a, acts] = True
                    assert np.array_equal(np.asarray(ad.available(), dtype=bool), want), f"{name} op {k}: {key} {ad.available()}"
        else:
            raise ValueError(op)
