"""The reference's known-answer tests through the product path: lle_amd.World -> C ABI -> HIP kernels on the MI355X."""
import pytest

from tests.kat_runner import load_cases, run_case

pytestmark = pytest.mark.gpu
CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_kat(case):
    from tests.gpu_adapter import GpuWorld

    run_case(lambda map_str=None, level=None: GpuWorld(map_str, level), case, derived=True)
