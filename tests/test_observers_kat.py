"""The reference's own observation tests (tests/golden/kat_observers.json) against the oracle's restatement
(oracle/observers.py) -- this is what pins it -- and against the host build of the device logic."""
import pytest

from tests.kat_observers_runner import HostsimAdapter, OracleAdapter, load_cases, run_case

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_observers_kat(oracle_mod, case):
    run_case(lambda c: OracleAdapter(oracle_mod, c), case)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hostsim_observers_kat(oracle_mod, case):
    run_case(HostsimAdapter, case)
