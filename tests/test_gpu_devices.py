"""One process, several handles (BASELINE north_star: "a thin host that owns device buffers"): every C-ABI call runs on its
batch's device whatever device is current, and puts the caller's device back (lle_hip.h "Threading"); the > 64 KiB LDS opt-in
is remembered per device.  The cross-device half needs two GPUs and is skipped on a one-GPU box (examples/c_abi_multi_gpu.c
does the same from C on whatever is visible)."""
import pytest

from oracle.levels import LEVELS

pytestmark = pytest.mark.gpu


def test_calls_leave_the_current_device_alone():
    import torch

    from lle_amd import BatchedWorld
    dev = torch.cuda.current_device()
    bw = BatchedWorld(LEVELS[6], 512)
    bw.step(sample=True, auto_reset=True, seed=1, t=0)
    bw.stats(), bw.snapshot(), bw.observe(), bw.available_actions()
    assert torch.cuda.current_device() == dev


def test_two_handles_in_one_process_are_independent(oracle_mod):
    """Interleaved steps of two batches (different maps, different streams) from one thread: each matches its own oracle."""
    import torch

    from lle_amd import BatchedWorld
    from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, unpack_engine
    texts = [LEVELS[6], EXTRA_MAPS["config5_32x32"]]  # the second needs the > 64 KiB LDS opt-in
    n = 256
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bws, obs = [], []
    for text, st in zip(texts, streams):
        with torch.cuda.stream(st):
            bws.append(BatchedWorld(text, n))
        obs.append(oracle_mod.OracleBatch(text, n))
    for t in range(12):
        for bw, st in zip(bws, streams):
            with torch.cuda.stream(st):
                bw.step(sample=True, auto_reset=True, seed=6, t=t)
        for bw, ob in zip(bws, obs):
            ostep = ob.step(None, auto_reset=True, seed=6, t=t)
            eng = unpack_engine(bw.host_buffers(), ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
            assert_step_equal(eng, ostep, f"t={t}")
            assert_state_equal(eng, ob.dump(), f"t={t}")


def test_handles_on_two_devices_with_the_wrong_device_current(oracle_mod):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's 8-GPU node; examples/c_abi_multi_gpu.c covers it from C)")
    from lle_amd import BatchedWorld
    from tests.parity_util import EXTRA_MAPS, assert_state_equal, assert_step_equal, unpack_engine
    text = EXTRA_MAPS["config5_32x32"]  # > 64 KiB of LDS: the opt-in must be made on BOTH devices
    n = 256
    bws = [BatchedWorld(text, n, device=f"cuda:{d}") for d in (0, 1)]
    obs = [oracle_mod.OracleBatch(text, n) for _ in (0, 1)]
    for t in range(8):
        for d, (bw, ob) in enumerate(zip(bws, obs)):
            torch.cuda.set_device(1 - d)  # deliberately the OTHER device (the launch goes to bw.device's current stream)
            bw.step(sample=True, auto_reset=True, seed=6, t=t, env_offset=d * n)
            assert torch.cuda.current_device() == 1 - d
            ostep = ob.step(None, auto_reset=True, seed=6, t=t, env_offset=d * n)
            eng = unpack_engine(bw.host_buffers(), ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
            assert_step_equal(eng, ostep, f"device {d} t={t}")
            assert_state_equal(eng, ob.dump(), f"device {d} t={t}")
    assert bws[0].stats()["env_steps"] == bws[1].stats()["env_steps"] == 8 * n


def test_placement_candidates_keep_one_arena_and_change_nothing(oracle_mod):
    """BatchedWorld(placement_candidates=k): k arenas timed with the row-fill probe, the fastest kept, back in the start
    state -- then the batch is the batch it would have been (steps against the oracle)."""
    import torch

    from lle_amd import BatchedWorld
    from tests.parity_util import assert_state_equal, assert_step_equal, unpack_engine
    n = 1024
    bw = BatchedWorld(LEVELS[6], n, placement_candidates=3)
    p = bw.placement
    assert p["candidates"] == 3 and len(p["row_fill_us"]) == 3 and 0 <= p["chosen"] < 3 and min(p["row_fill_us"]) > 0
    assert p["row_fill_us"][p["chosen"]] == min(p["row_fill_us"])
    plain = BatchedWorld(LEVELS[6], n)
    assert plain.placement is None
    torch.cuda.synchronize()
    assert torch.equal(bw.obs, plain.obs) and torch.equal(bw.pos, plain.pos) and torch.equal(bw.avail, plain.avail)
    ob = oracle_mod.OracleBatch(LEVELS[6], n)
    for t in range(6):
        bw.step(sample=True, auto_reset=True, seed=11, t=t)
        ostep = ob.step(None, auto_reset=True, seed=11, t=t)
        eng = unpack_engine(bw.host_buffers(), ob.A, ob.G, ob.Ls, ob.beam_stride, ob.C, ob.H, ob.W)
        assert_step_equal(eng, ostep, f"placed t={t}")
        assert_state_equal(eng, ob.dump(), f"placed t={t}")
