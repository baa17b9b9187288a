"""BatchedLLE (lle_amd/env.py) against a per-env restatement of the reference's LLE host class on oracle worlds
(tests/oracle_env.py): observation, state, reward (both strategies), done, available actions, reset with recolouring."""
import numpy as np
import pytest

from oracle.levels import LEVELS
from tests.oracle_env import OracleLLE
from tests.parity_util import EXTRA_MAPS

pytestmark = pytest.mark.gpu

CONFIGS = [
    ("level6", dict(obs_type="layered", state_type="state")),
    ("level6", dict(obs_type="partial5x5", state_type="normalized-state", walkable_lasers=False, multi_objective=True)),
    ("nested", dict(obs_type="perspective", state_type="layered", walkable_lasers=False)),
    ("level3", dict(obs_type="flattened", state_type="state", multi_objective=True)),
    ("exit_under_beam", dict(obs_type="layered-padded", padding_size=2, state_type="partial3x3")),
]


@pytest.mark.parametrize("name,kw", CONFIGS, ids=[f"{n}-{k['obs_type']}" for n, k in CONFIGS])
@pytest.mark.parametrize("randomize", [False, True])
def test_batched_lle_matches_per_env_restatement(oracle_mod, name, kw, randomize):
    import torch

    from lle_amd import BatchedLLE

    text = LEVELS[int(name[-1])] if name.startswith("level") else EXTRA_MAPS[name]
    n = 96
    env = BatchedLLE(text, n, randomize_lasers=randomize, seed=5, **kw)
    okw = {k: v for k, v in kw.items()}
    refs = [OracleLLE(oracle_mod.OracleWorld(text), **okw) for _ in range(n)]
    A, L = env.n_agents, env.world.map.n_sources
    rng = np.random.default_rng(0)

    def same_obs(got, want, where):
        got = got.cpu().numpy().astype(np.float32)
        if want.ndim == got.ndim + 1:            # the reference tiles n_agents identical copies; the batch carries one
            assert all(np.array_equal(want[0], want[k]) for k in range(1, want.shape[0])), where
            want = want[0]
        assert got.shape == want.shape and np.array_equal(got, want), where

    def compare(step_out, rewards, where):
        obs, state = env.get_observation(), env.get_state()
        avail = env.available_actions().cpu().numpy()
        done = env.done.cpu().numpy()
        for e in range(0, n, 7):
            same_obs(obs[e], refs[e].get_observation(), f"{where} obs env {e}")
            same_obs(state[e], refs[e].get_state(), f"{where} state env {e}")
            assert np.array_equal(avail[e], refs[e].available_actions()), f"{where} avail env {e}"
            assert bool(done[e]) == refs[e].done, f"{where} done env {e}"
        if step_out is not None:
            r = step_out["reward"].cpu().numpy()
            for e in range(n):
                assert np.array_equal(r[e], rewards[e]), f"{where} reward env {e}: {r[e]} != {rewards[e]}"

    colours = torch.from_numpy(rng.integers(0, A, size=(n, L), dtype=np.uint8)) if randomize else None
    env.reset(colours=colours)
    for e in range(n):
        refs[e].reset(None if colours is None else colours[e].numpy())
    compare(None, None, "after reset")
    for t in range(25):
        # finished envs are reset first (auto_reset), with fresh colours when lasers are randomised
        done = env.done.cpu().numpy()
        if randomize:
            colours = torch.from_numpy(rng.integers(0, A, size=(n, L), dtype=np.uint8))
            env.reset(env_mask=torch.from_numpy(done.astype(np.uint8)), colours=colours)
        for e in np.nonzero(done)[0]:
            refs[e].reset(None if not randomize else colours[e].numpy())
        avail = env.available_actions().cpu().numpy() if not randomize else None
        # uniform over each agent's available actions (World.available_actions: the walkable_lasers filter is advisory)
        wa = env.world.available_actions(True).cpu().numpy()
        if not randomize:
            # the kernel's auto-reset happens inside step(): the availability of a finished env is that of its reset state
            for e in np.nonzero(done)[0]:
                wa[e] = np.array([[a in lst for a in range(5)] for lst in refs[e].w.available_actions()])
        actions = np.zeros((n, A), np.uint8)
        for e in range(n):
            for a in range(A):
                actions[e, a] = rng.choice(np.nonzero(wa[e, a])[0])
        out = env.step(torch.from_numpy(actions), auto_reset=not randomize)
        rewards = [refs[e].step(actions[e])[0] for e in range(n)]
        assert int(out["err"].max()) == 0
        compare(out, rewards, f"t={t}")


# ---- the reference's own LLE tests (tests/golden/kat_env.json) through BatchedLLE
from tests.kat_env_runner import load_cases, run_case  # noqa: E402

ENV_CASES = load_cases()


class _BatchedAdapter:
    def __init__(self, case):
        from lle_amd import BatchedLLE
        self.n = 70  # every env plays the same script; the first and the last are checked
        self.env = BatchedLLE(case["map"], self.n, multi_objective=case["multi_objective"])

    def reset(self):
        self.env.reset()

    def step(self, actions):
        import torch
        out = self.env.step(torch.tensor([actions] * self.n, dtype=torch.uint8))
        assert int(out["err"].max()) == 0
        r, d = out["reward"].cpu().numpy(), out["done"].cpu().numpy()
        assert np.array_equal(r[0], r[-1]) and d[0] == d[-1]
        return r[0], d[0]

    def set_state(self, positions, gems, alive):
        import torch
        n = self.n
        err = self.env.set_state(torch.tensor([positions] * n, dtype=torch.uint8), torch.tensor([gems] * n, dtype=torch.bool),
                                 torch.tensor([alive] * n, dtype=torch.bool))
        assert int(err.max()) == 0

    def metrics(self):
        return {"has-arrived": [bool(x) for x in self.env.agents_arrived()[-1].cpu().numpy()],
                "is-alive": [bool(x) for x in self.env.agents_alive()[-1].cpu().numpy()]}


@pytest.mark.parametrize("case", ENV_CASES, ids=[c["name"] for c in ENV_CASES])
def test_batched_lle_env_kat(case):
    run_case(_BatchedAdapter, case)
