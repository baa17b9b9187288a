"""The reference's LLE-level tests (tests/golden/kat_env.json, from python/tests/test_env.py) against the per-env
restatement of env.py + reward_strategy.py on the oracle (tests/oracle_env.py): this pins that restatement."""
import pytest

from tests.kat_env_runner import load_cases, run_case
from tests.oracle_env import OracleLLE

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_env_kat(oracle_mod, case):
    def make(c):
        env = OracleLLE(oracle_mod.OracleWorld(c["map"]), multi_objective=c["multi_objective"])
        env.free_running = bool(c.get("free_running"))  # strategy-level cases step past the end of the episode
        return _Adapter(env)
    run_case(make, case)


class _Adapter:
    """kat_env_runner protocol over OracleLLE (whose `done` is an attribute and whose mask method is available_actions)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def done(self):
        return self.env.done

    def available(self):
        return self.env.available_actions()
