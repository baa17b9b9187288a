"""bench.py as the driver runs it: `python bench.py --gpus N` must start the N ranks itself, print ONE JSON line, and
refuse (exit code 2) to measure fewer GPUs than it was asked for (BASELINE configs[3]).

CPU: the spawn / rendezvous / one-line plumbing over gloo (no GPU, `--plumbing-only`: not a measurement) and the
refusal.  GPU (one card): the same entry point through torch.distributed.run + RCCL with one rank, and the RCCL
all-reduce of BatchedWorld's counters in this process."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run([sys.executable, BENCH, *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def _one_line(res):
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}:\n{res.stdout}\n{res.stderr[-2000:]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 3])
def test_parent_spawns_n_ranks_and_prints_one_line(world):
    res = _run("--gpus", str(world), "--plumbing-only")
    assert res.returncode == 0, res.stderr[-2000:]
    out = _one_line(res)
    assert out["n_gpus"] == world and out["rccl_ranks"] == world and out["plumbing_only"] is True
    from lle_amd.distributed import STAT_KEYS
    tri = world * (world + 1) // 2  # rank r contributes (r + 1) * (i + 1) to counter i
    assert out["rollout_stats"] == {k: tri * (i + 1) for i, k in enumerate(STAT_KEYS)}
    assert out["elapsed_max"] == float(world)
    # the per-rank record of a real N-rank line (gathered over the same collective helpers), in rank order
    assert [r["rank"] for r in out["per_rank"]] == list(range(world))
    assert [round(r["kernel_ms"], 6) for r in out["per_rank"]] == [round(0.02 * (k + 1), 6) for k in range(world)]
    assert all(r["env_steps"] == 65536 * 20 for r in out["per_rank"])
    reg = out["region"]
    assert abs(reg["kernel_ms_max_over_ranks"] - 0.02 * world) < 1e-9 and abs(reg["kernel_ms_min_over_ranks"] - 0.02) < 1e-9
    assert abs(reg["wall_ms_per_step_max_over_ranks"] - world / 20 * 1e3) < 1e-6
    assert abs(reg["wall_ms_per_step_without_closing_barrier_max_over_ranks"] - 0.9 * world / 20 * 1e3) < 1e-6
    assert abs(reg["agent_steps_per_s_by_slowest_rank_events"] - 4 * 65536 * world / (0.02e-3 * world)) < 1.0
    assert abs(reg["sustained_kernel_ms_max_over_ranks"] - 0.019 * world) < 1e-9
    # the shard check's plumbing: one 64-bit hash per rank all-gathered, rank 0 recomputed every window with its env_offset
    chk = out["shard_check"]
    assert chk["status"] == "ok" and chk["ranks"] == world and chk["mismatching_ranks"] == []
    assert len(chk["hashes"]) == world and chk["distinct_windows"] == world


def test_scaling_block_and_gpu_count_helpers():
    """scaling_block on made-up rows (one slow rank must show), the N = 1 reference round trip, visible_gpus without HIP."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = [[0.0005, 0.0200, 65536 * 20, 0.04, 0.0195, 0.00045], [0.0009, 0.0400, 65536 * 20, 0.08, 0.0390, 0.00085]]  # rank 1 twice as slow
    blk = bench.scaling_block(rows, steps=20, sustained_steps=2000, n_envs=65536, agents=4, n1_reference=0.0195)
    assert blk["region"]["kernel_ms_max_over_ranks"] == 0.04 and blk["per_rank"][1]["kernel_ms"] == 0.04
    assert abs(blk["weak_scaling_vs_n1"]["ratio"] - 0.5) < 1e-12
    assert abs(blk["region"]["host_share_of_wall"] - (1 - 0.04e-3 * 20 / 0.0009)) < 1e-12
    assert "weak_scaling_vs_n1" not in bench.scaling_block(rows, 20, 2000, 65536, 4)
    # both clocks of the region: the contract's (closing barrier inside) and each rank's own
    assert abs(blk["region"]["wall_ms_per_step_without_closing_barrier_max_over_ranks"] - 0.00085 / 20 * 1e3) < 1e-12
    assert abs(blk["per_rank"][0]["wall_ms_per_step_without_closing_barrier"] - 0.00045 / 20 * 1e3) < 1e-12
    # the N = 1 reference is keyed on (user, envs per GPU, sustained launches): another workload's figure is never picked up
    assert bench.n1_reference(4321, 77, write=0.5) == 0.5 and bench.n1_reference(4321, 77) == 0.5
    assert bench.n1_reference(4321, 78) is None and bench.n1_reference(4322, 77) is None
    import os
    os.unlink(bench.n1_cache_path(4321, 77))
    import torch
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0
    if not torch.cuda.is_available():
        assert n == 0


def test_refuses_to_measure_fewer_gpus_than_asked_for():
    import torch
    asked = torch.cuda.device_count() + 1
    if asked < 2:
        asked = 2
    res = _run("--gpus", str(asked), "--steps", "2", "--warmup", "1")
    assert res.returncode == 2, (res.returncode, res.stderr[-2000:])
    assert res.stdout.strip() == ""
    assert "refusing" in res.stderr


def test_rank_count_mismatch_under_the_launcher_is_an_error():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=env, timeout=300, cwd=ROOT)
    assert res.returncode == 2 and res.stdout.strip() == "" and "WORLD_SIZE=1" in res.stderr


@pytest.mark.gpu
def test_one_rank_through_the_launcher_and_rccl():
    steps = 20
    res = _run("--gpus", "1", "--force-spawn", "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-configs",
               "--no-fused", "--sustained-steps", "0")
    assert res.returncode == 0, res.stderr[-3000:]
    out = _one_line(res)
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1
    assert out["rollout_stats"]["env_steps"] == 65536 * steps
    assert out["rollout_stats"]["agent_steps"] == 4 * 65536 * steps
    assert out["roofline"]["bound"] == "infinity-cache-absorbed" and 0.0 < out["roofline"]["frac"] < 1.0
    assert out["shard_check"]["status"] == "ok" and out["shard_check"]["ranks"] == 1


@pytest.mark.gpu
def test_allreduce_of_batched_world_counters_over_rccl():
    import torch
    import torch.distributed as dist

    from lle_amd import BatchedWorld, Map
    from lle_amd.distributed import allreduce_max, allreduce_stats, shard_offset

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        n, steps = 4096, 16
        bw = BatchedWorld(Map(level=6), n, device=dev)
        for t in range(steps):
            bw.step(sample=True, auto_reset=True, seed=7, t=t, env_offset=shard_offset(n, 0))
        local = bw.stats()
        total = allreduce_stats(local, dev)
        assert total == local and total["env_steps"] == n * steps
        assert allreduce_max(1.5, dev) == 1.5
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_code_path_rehearsed_on_one_gpu():
    """`bench.py --gpus 2 --rehearse-on-one-gpu`: the parent spawns two ranks through torch.distributed.run, both step their own
    env range on cuda:0, the collectives run over gloo -- every line of the N > 1 path (sharding, barrier-bracketed timed
    region, max over ranks, counter all-reduce, per-rank gather, the one JSON line) except RCCL itself, which refuses two
    ranks on one device.  Not a measurement, and the line says so."""
    steps = 20
    res = _run("--gpus", "2", "--rehearse-on-one-gpu", "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-configs",
               "--fused-steps", "4", "--sustained-steps", "40")
    assert res.returncode == 0, res.stderr[-3000:]
    out = _one_line(res)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and "NOT a measurement" in out["rehearsal"]
    assert out["rollout_stats"]["env_steps"] == 2 * 65536 * steps and out["rollout_stats"]["agent_steps"] == 4 * 2 * 65536 * steps
    assert out["config"]["global_batch"] == 2 * 65536 and out["config"]["parallelism"] == "env-shard x2"
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["env_steps"] == 65536 * steps for r in out["per_rank"])
    assert all(r["kernel_ms"] > 0 and r["sustained_kernel_ms"] > 0 for r in out["per_rank"])
    assert out["region"]["kernel_ms_max_over_ranks"] >= out["region"]["kernel_ms_min_over_ranks"] > 0
    assert out["sustained"]["steps"] == 40 and out["fused_rollout"]["steps_per_launch"] == 4
    assert "cpu_baseline" not in out and "configs" not in out  # N = 1 only
    # shard invariance across the two ranks: rank 1's window (env_offset 65 536) hashed by rank 1 and recomputed by rank 0
    chk = out["shard_check"]
    assert chk["status"] == "ok" and chk["ranks"] == 2 and chk["distinct_windows"] == 2 and len(set(chk["hashes"])) == 2
    assert out["tuning"]["autotuned"] == 1 and out["host_issue_us_per_step"] > 0
    assert out["region"]["wall_ms_per_step_max_over_ranks"] >= out["region"]["wall_ms_per_step_without_closing_barrier_max_over_ranks"] > 0
