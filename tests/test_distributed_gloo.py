"""The N>1 path on CPU (gloo, world_size 2): envs shard over ranks with env_offset, no data-path collective, one
all-reduce of the counters.  The per-rank engine here is the TEST-ONLY host build of the device logic
(tests/hostsim); on the GPU box the same sharding code drives BatchedWorld over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.levels import LEVELS

N_PER_RANK, STEPS, SEED = 64, 25, 1234


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rollout(text, n, offset):
    from lle_amd import _capi
    from tests import hostsim

    sb = hostsim.SimBatch(text, n)
    for t in range(STEPS):
        sb.step(None, flags=_capi.LLE_STEP_SAMPLE_ACTIONS | _capi.LLE_STEP_AUTO_RESET, seed=SEED, t=t, env_offset=offset)
    return sb


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lle_amd.distributed import STAT_KEYS, allreduce_max, allreduce_stats, shard_offset

    sb = _rollout(LEVELS[6], N_PER_RANK, shard_offset(N_PER_RANK, rank))
    stats = dict(zip(STAT_KEYS, [int(v) for v in sb.buf("stats")]))
    total = allreduce_stats(stats, torch.device("cpu"))
    slowest = allreduce_max(float(rank + 1), torch.device("cpu"))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=sb.buf("pos"), bits=sb.buf("bits"), obs=sb.buf("obs"),
             total=np.array([total[k] for k in STAT_KEYS]), slowest=slowest)
    dist.destroy_process_group()


def test_two_ranks_reproduce_one_big_batch(tmp_path):
    from lle_amd.distributed import shard_range

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    whole = _rollout(LEVELS[6], N_PER_RANK * world, 0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for key in ("pos", "bits", "obs"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), whole.buf(key)), key
    for p in parts:
        assert np.array_equal(p["total"], whole.buf("stats"))
        assert float(p["slowest"]) == 2.0
    assert [shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def _check_worker(rank, world, port, out_dir, broken_rank):
    import json

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lle_amd.distributed import shard_check, tensor_hash

    def window_hash(env_offset):
        # the checksum of a real window: 16 envs of the host build of the device logic stepped from env_offset
        sb = _rollout(LEVELS[6], 16, env_offset)
        h = 0
        for key in ("pos", "bits", "obs"):
            h = (h * 1099511628211 + tensor_hash(torch.from_numpy(np.ascontiguousarray(sb.buf(key))))) & (2**64 - 1)
        # a rank whose shard diverged (here: forced) must show up as a mismatch on rank 0, which recomputes every window itself
        return h ^ 1 if rank == broken_rank and env_offset != 0 else h
    res = shard_check(window_hash, N_PER_RANK, rank, world, torch.device("cpu"))
    with open(os.path.join(out_dir, f"check{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_shard_check_over_the_collective(tmp_path):
    """bench.py's first-SCALE-run evidence (lle_amd.distributed.shard_check): every rank hashes the window at the head of its
    shard, the hashes are all-gathered, rank 0 recomputes all windows with the matching env_offset.  Two gloo ranks agree; a rank
    that diverges is named."""
    import json

    for broken, want in ((-1, ("ok", [])), (1, ("fail", [1]))):
        d = tmp_path / f"b{broken}"
        d.mkdir()
        mp.spawn(_check_worker, args=(2, _free_port(), str(d), broken), nprocs=2, join=True)
        r0, r1 = (json.load(open(d / f"check{r}.json")) for r in range(2))
        assert r1 == {"status": "n/a"}
        assert (r0["status"], r0["mismatching_ranks"]) == want and r0["ranks"] == 2 and r0["distinct_windows"] == 2
