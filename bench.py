"""bench.py -- env-steps/s of the batched World.step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: on-device action sampling (uniform over available actions),
auto-reset of finished envs, World.step, event / availability emission and the int8 layered observation, for
65 536 level-6 environments per GPU (BASELINE.json configs[2], the configuration the metric is quoted on).
All inputs are resident in HBM before the timed region.  Envs shard over GPUs with no data-path collective
(weak scaling); the only collective is one RCCL all-reduce of the rollout counters after the timed region.

Prints ONE JSON line (rank 0).  value = agent-steps/s over the whole job (agents x envs x steps / s);
env-steps/s is reported next to it.  roofline / cpu_baseline objects: see DESIGN.md section "Measurement".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
LEVEL = 6
SEED = 1234
# algorithmic bytes per env-step of level 6
#   obs C*H*W = 12*12*13 = 1872, state r/w 2*24 = 48, actions 4, avail 4, events 1 + 2*4 = 9
ALGO_BYTES_PER_ENV_STEP = 1937
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(n_envs, seconds_target=12.0):
    """The CPU oracle (C restatement of the reference algorithm, kind "port") on a bounded sample of the same workload:
    the same n_envs level-6 environments, sampled actions + auto-reset + int8 layered observation, for about
    `seconds_target` seconds on every host core (one thread per core over disjoint env ranges)."""
    import numpy as np

    from oracle import oracle
    from oracle.levels import LEVELS

    level_text = LEVELS[LEVEL]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))  # the GPU box gives one GPU a share of the host cores; more threads only thrash
    ob = oracle.OracleBatch(level_text, n_envs)
    obs = np.zeros((n_envs, ob.C * ob.H * ob.W), np.int8)
    ob.rollout(4, SEED, threads, obs)  # warm-up
    cal = 16
    t0 = time.perf_counter()
    ob.rollout(cal, SEED, threads, obs)
    rate = n_envs * cal / (time.perf_counter() - t0)
    steps = max(8, int(seconds_target * rate / n_envs))
    t0 = time.perf_counter()
    ob.rollout(steps, SEED, threads, obs)
    dt = time.perf_counter() - t0
    steps1 = max(2, steps // (4 * threads))
    t1 = time.perf_counter()
    ob.rollout(steps1, SEED, 1, obs)
    dt1 = time.perf_counter() - t1
    return {
        "value": ob.A * n_envs * steps / dt, "unit": "agent-steps/s", "cores": threads, "kind": "port",
        "env_steps_per_s": n_envs * steps / dt, "seconds": dt,
        "single_thread_env_steps_per_s": n_envs * steps1 / dt1,
        "sample": f"level {LEVEL}, {n_envs} envs x {steps} steps ({dt:.1f} s), sampled actions + auto-reset + int8 layered obs, "
                  f"C restatement of the Rust reference algorithm (oracle/lle_oracle.c), {threads} threads",
    }


def load_traffic():
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--envs-per-wave", type=int, default=0, help="0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--fused-steps", type=int, default=16, help="steps per launch of the fused rollout")
    ap.add_argument("--ring-slots", type=int, default=8, help="trajectory ring slots of the fused rollout")
    args = ap.parse_args()

    # Everything but the final JSON line goes to stderr, at the file-descriptor level: RCCL and the HIP runtime print
    # banners to fd 1 from C code (e.g. "Librccl path : ...") and the contract is ONE line on stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from lle_amd import BatchedWorld, Map
    from lle_amd.distributed import allreduce_max, allreduce_stats, shard_offset

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU execution path)"
    # one process per GPU under torch.distributed.run (RCCL = backend "nccl"); a plain `python bench.py` is rank 0 of 1
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    dev = torch.device("cuda", local_rank if use_dist else 0)
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    n = args.envs_per_gpu
    bw = BatchedWorld(Map(level=LEVEL), n, device=dev, envs_per_wave=args.envs_per_wave or None)
    offset = shard_offset(n, rank)

    def run(k, t0):
        for t in range(t0, t0 + k):
            bw.step(sample=True, auto_reset=True, seed=SEED, t=t, env_offset=offset)

    run(args.warmup, 0)
    torch.cuda.synchronize(dev)
    bw.stats(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_start = time.perf_counter()
    ev0.record()
    run(args.steps, args.warmup)
    ev1.record()
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch-to-launch duration on the launch stream

    elapsed = allreduce_max(elapsed, dev) if use_dist else elapsed
    stats = bw.stats()
    stats = allreduce_stats(stats, dev) if use_dist else stats

    # ---- secondary measurement: the same random rollout with lle_batch_rollout (several steps per launch, every
    # step's observation / actions / reward counts written to a trajectory ring larger than the caches)
    fused = []
    if not args.no_fused:
        # (T, R): the ring of args.ring_slots slots is larger than the 256 MB Infinity Cache (true HBM writes); the
        # two-slot ring is a double buffer -- a consumer reads slot t while step t + 1 is written -- and stays in it
        for T, R in ((args.fused_steps, args.ring_slots), (args.fused_steps, 2)):
            ring = bw.make_ring(R)
            launches = max(1, args.steps // T)
            for _ in range(2):
                bw.rollout(T, auto_reset=True, seed=SEED, env_offset=offset, ring=ring, ring_pos=bw.t)
            torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(dev)
            tf = time.perf_counter()
            for _ in range(launches):
                bw.rollout(T, auto_reset=True, seed=SEED, env_offset=offset, ring=ring, ring_pos=bw.t)
            torch.cuda.synchronize(dev)
            if use_dist:
                dist.barrier()
            fused_elapsed = time.perf_counter() - tf
            fused_elapsed = allreduce_max(fused_elapsed, dev) if use_dist else fused_elapsed
            fused.append((T, R, launches, fused_elapsed))
            del ring

    if rank == 0:
        A = bw.map.n_agents
        total_envs = n * world
        env_steps_s = total_envs * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        info = bw.kernel_info()
        out = {
            "metric": "env-steps/s (agents x envs x steps/s), level-6 batch 65536",
            "value": A * env_steps_s, "unit": "agent-steps/s",
            "env_steps_per_s": env_steps_s,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"World.level({LEVEL}) 4 agents 12x13, {n} envs/GPU, sampled actions + auto-reset + "
                                   "int8 layered obs (C=12)", "envs_per_gpu": n, "global_batch": total_envs,
                       "parallelism": f"env-shard x{world}", "kernel": info["kernel"], "envs_per_wave": info["envs_per_wave"],
                       "lds_bytes_per_workgroup": info["lds_bytes"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(),
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n, "kernel_ms": kernel_ms},
            "rollout_stats": stats,
        }
        for key, (T, R, launches, fe) in zip(("fused_rollout", "fused_rollout_double_buffer"), fused):
            # per env-step: obs 1872 + actions 4 + reward 4 + err/evcount/done 3 + events 8, state r/w (48 B) once per launch
            fused_bytes = 1891 + 48.0 / T
            ring_mb = R * n * 1872 / 1e6
            out[key] = {
                "what": "lle_batch_rollout: same random rollout, several steps per launch, per-step obs/actions/reward to a trajectory ring"
                        + (" larger than the 256 MB Infinity Cache (true HBM writes)" if ring_mb > 256 else
                           " of two slots (double buffer, stays in the 256 MB Infinity Cache: NOT an HBM figure)"),
                "steps_per_launch": T, "ring_slots": R, "ring_MB": ring_mb, "steps": launches * T, "ms_per_step": fe / (launches * T) * 1e3,
                "env_steps_per_s": total_envs * launches * T / fe, "agent_steps_per_s": A * total_envs * launches * T / fe,
                "algorithmic_bytes_per_env_step": fused_bytes,
                "achieved_GBps_per_gpu": fused_bytes * n * launches * T / fe / 1e9,
                "frac_of_hbm_peak": fused_bytes * n * launches * T / fe / 1e9 / HBM_PEAK_GBS,
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
