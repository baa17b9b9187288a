"""The C ABI from a plain-C host (examples/c_abi_rollout.c; built by __graft_entry__.build()): no Python, no torch in the
process that drives the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_c_host_runs_a_rollout_through_the_c_abi():
    exe = os.path.join(ROOT, "examples", "c_abi_rollout")
    if not os.path.exists(exe):
        from lle_amd.build import build_c_example
        build_c_example()
    res = subprocess.run([exe, "8192", "50"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert res.returncode == 0, res.stdout
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "ok", res.stdout
    stats = dict(zip(lines[-2].split()[0::2], map(int, lines[-2].split()[1::2])))
    assert stats["env_steps"] == 8192 * 50 and stats["agent_steps"] == 4 * 8192 * 50 and stats["invalid"] == 0


def test_c_example_builds_and_fails_loudly_without_a_device():
    """CPU side: the example compiles against include/lle_hip.h with gcc and links liblle_hip.so; without a GPU it stops
    at the first HIP call with a message (there is no CPU path to fall back to)."""
    import torch

    from lle_amd.build import build_c_example
    exe = build_c_example()
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    res = subprocess.run([exe, "64", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert res.returncode != 0 and "failed" in res.stdout
