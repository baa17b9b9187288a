"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/kat_world.json)."""
import pytest

from tests.kat_runner import load_cases, run_case

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_kat(oracle_mod, case):
    def make(map_str=None, level=None):
        if level is not None:
            return oracle_mod.OracleWorld.level(level)
        return oracle_mod.OracleWorld(map_str)

    run_case(make, case, derived=True)
