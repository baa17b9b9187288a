"""The device state machine (step_logic.hpp) and the compiled map tables, built for the host (tests/hostsim),
against the reference's known-answer tests.  Catches logic/table bugs without a GPU; the GPU tests repeat the
same scripts through the real kernels."""
import pytest

from tests.kat_runner import load_cases, run_case

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hostsim_kat(case):
    from tests import hostsim

    run_case(lambda map_str=None, level=None: hostsim.SimWorld(map_str, level), case, derived=True)
