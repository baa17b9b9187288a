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


@pytest.mark.gpu
def test_c_host_places_its_arena():
    """The same host with three candidate arenas: the row-fill probe times each, the rollout runs on the one kept
    and takes the same steps (the counters are a function of seed and step numbers only)."""
    exe = os.path.join(ROOT, "examples", "c_abi_rollout")
    if not os.path.exists(exe):
        from lle_amd.build import build_c_example
        build_c_example()
    outs = []
    for extra in ([], ["3"]):
        res = subprocess.run([exe, "8192", "50"] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        assert res.returncode == 0 and res.stdout.strip().splitlines()[-1] == "ok", res.stdout
        outs.append(res.stdout)
    assert sum(line.startswith("arena ") and "row fill" in line for line in outs[1].splitlines()) == 3
    stats = [next(line for line in o.splitlines() if line.startswith("env_steps")) for o in outs]
    assert stats[0] == stats[1]


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


@pytest.mark.gpu
def test_c_host_owning_one_handle_per_gpu_reduces_over_rccl():
    """examples/c_abi_multi_gpu.c: ONE process, a batch + stream + RCCL communicator per visible GPU (lle_comm_create_all), the
    current device deliberately wrong for every handle but the last, counters reduced by lle_batch_stats_allreduce_group."""
    exe = os.path.join(ROOT, "examples", "c_abi_multi_gpu")
    if not os.path.exists(exe):
        from lle_amd.build import build_c_example
        build_c_example(name="c_abi_multi_gpu")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([exe, "8192", "40"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "ok", res.stdout
    n_gpus = sum(1 for ln in lines if ln.startswith("gpu "))
    row = next(ln for ln in lines if ln.startswith("env_steps"))
    stats = dict(zip(row.split()[0::2], map(int, row.split()[1::2])))
    assert n_gpus >= 1 and stats["env_steps"] == n_gpus * 8192 * 40 and stats["invalid"] == 0
    # round 4: before anything is timed, every GPU's first 64 envs hashed, all-reduced over RCCL (lle_comm_allreduce_i64_group), replayed on GPU 0
    assert any(ln.startswith("shard check: ok") for ln in lines), res.stdout
    # more GPUs than visible: refused, never a smaller run under the same name
    res = subprocess.run([exe, "64", "2", "99"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60, env=env)
    assert res.returncode == 2 and "refusing" in res.stdout


@pytest.mark.gpu
def test_comm_api_with_one_rank_matches_the_local_counters():
    """lle_comm_unique_id / lle_comm_create (one process per GPU) and lle_comm_create_all with a world of one: the reduced
    counters are the batch's own; the calls leave the current device alone."""
    import ctypes as C

    import torch

    from lle_amd import BatchedWorld, Map, _capi

    L = _capi.lib()
    bw = BatchedWorld(Map(level=6), 4096)
    for t in range(12):
        bw.step(sample=True, auto_reset=True, seed=3, t=t)
    local = bw.stats()
    ident = (C.c_uint8 * 128)()
    assert L.lle_comm_unique_id(ident) == 0, L.lle_last_error()
    comm = L.lle_comm_create(ident, 1, 0, bw.device.index or 0)
    assert comm, L.lle_last_error()
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.lle_comm_rank(comm, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    out = (C.c_int64 * 8)()
    assert L.lle_batch_stats_allreduce(bw.h, comm, out, 0, bw._stream()) == 0, L.lle_last_error()
    assert list(out) == list(local.values()) and out[0] == 4096 * 12
    buf = torch.tensor([5, 7], dtype=torch.int64, device=bw.device)
    assert L.lle_comm_allreduce_i64(comm, buf.data_ptr(), 2, 1, bw._stream()) == 0
    torch.cuda.synchronize()
    assert buf.tolist() == [5, 7]
    L.lle_comm_free(comm)
    comms = (C.c_void_p * 1)()
    assert L.lle_comm_create_all(comms, 1, None) == 0, L.lle_last_error()
    batches, streams = (C.c_void_p * 1)(bw.h), (C.c_void_p * 1)(bw._stream())
    assert L.lle_batch_stats_allreduce_group(batches, comms, streams, 1, out, 1) == 0, L.lle_last_error()
    assert list(out) == list(local.values())
    assert bw.stats()["env_steps"] == 0  # reset_counters
    grp = torch.tensor([11, -3, 2**40], dtype=torch.int64, device=bw.device)
    bufs = (C.c_void_p * 1)(grp.data_ptr())
    assert L.lle_comm_allreduce_i64_group(comms, bufs, streams, 1, 3, 0) == 0, L.lle_last_error()
    torch.cuda.synchronize()
    assert grp.tolist() == [11, -3, 2**40]
    L.lle_comm_free(comms[0])
    assert L.lle_comm_create(ident, 2, 5, 0) is None and L.lle_last_status() == -2
