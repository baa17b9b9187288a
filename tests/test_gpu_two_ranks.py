"""Two processes, each with its own batch on the card, reproduce one big batch: env-range sharding with env_offset through the
real kernels, the counters summed by an all-reduce.  One MI355X here, so both ranks share cuda:0 and
the collective runs over gloo on host tensors -- RCCL refuses two ranks on one device; the RCCL leg with one rank is
tests/test_bench_spawn.py, and bench.py --gpus N is the same code with one device per rank."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_PER_RANK, STEPS, SEED = 4096, 24, 1234


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from lle_amd import BatchedWorld, Map
    from lle_amd.distributed import allreduce_max, allreduce_stats, shard_offset

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bw = BatchedWorld(Map(level=6), N_PER_RANK, device="cuda:0")
    off = shard_offset(N_PER_RANK, rank)
    for t in range(STEPS):
        bw.step(sample=True, auto_reset=True, seed=SEED, t=t, env_offset=off)
    total = allreduce_stats(bw.stats(), torch.device("cpu"))
    slowest = allreduce_max(float(rank + 1), torch.device("cpu"))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pos=bw.pos.cpu().numpy(), bits=bw.bits.cpu().numpy(), obs=bw.obs.cpu().numpy(),
             total=np.array(list(total.values())), slowest=slowest)
    dist.destroy_process_group()


def test_two_processes_reproduce_one_big_batch(tmp_path):
    import torch
    import torch.multiprocessing as mp

    from lle_amd import BatchedWorld, Map

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    whole = BatchedWorld(Map(level=6), N_PER_RANK * world, device="cuda:0")
    for t in range(STEPS):
        whole.step(sample=True, auto_reset=True, seed=SEED, t=t)
    torch.cuda.synchronize()
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for key in ("pos", "bits", "obs"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), getattr(whole, key).cpu().numpy()), key
    stats = np.array(list(whole.stats().values()))
    for p in parts:
        assert np.array_equal(p["total"], stats) and float(p["slowest"]) == 2.0
    assert stats[0] == N_PER_RANK * world * STEPS
