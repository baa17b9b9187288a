"""bench.py -- env-steps/s of the batched World.step() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 starts the N ranks ITSELF: this parent process never touches a GPU, checks that N devices are
visible (exit code 2 otherwise -- it never measures fewer GPUs than asked for) and runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` as a
child process, one rank per GPU over RCCL.  Launched under torch.distributed.run directly (RANK / WORLD_SIZE in the
environment) it is one of those ranks.

A "step" is one pass of the hot path over one batch: on-device action sampling (uniform over available actions),
auto-reset of finished envs, World.step, event / availability emission and the int8 layered observation, for
65 536 level-6 environments per GPU (BASELINE.json configs[2], the configuration the metric is quoted on).
All inputs are resident in HBM before the timed region.  Envs shard over GPUs with no data-path collective
(weak scaling); the only collective is one RCCL all-reduce of the rollout counters after the timed region.

Prints ONE JSON line (rank 0).  value = agent-steps/s over the whole job (agents x envs x steps / s);
env-steps/s is reported next to it.  Objects of the line (NOTEBOOK.md section 7):
  roofline       the dominant kernel of the --steps region (HIP events on the launch stream).  Each step rewrites the
                 same 122.7 MB of rows, which the 256 MB Infinity Cache absorbs: bound = "infinity-cache-absorbed".
  roofline_hbm   the same kernel on a batch whose rows (503 MB per launch) do not fit the Infinity Cache.  Since round 3 such launches
                 walk the environments alternately up and down, so each one finds the last cache-full of its predecessor's rows
                 in the Infinity Cache (`infinity_cache_share`); the rest are HBM writes.
  sustained      the same launches as the --steps region, >= 1000 of them.
  configs        BASELINE.json configs[1] (level 1 x 4 096) and configs[4] (32x32, 8 agents, 8 lasers x 65 536).
  lle_step       BatchedLLE.step (the reference's LLE host class, batched) on the headline workload: us per step.
  observers      the other observation builders (partial k x k, state, availability) through bound calls: us per launch, GB/s.
  fused_rollout  lle_batch_rollout, 16 steps per launch into a trajectory ring larger than the caches.
  cpu_baseline   the C restatement of the reference algorithm (oracle/) on the host cores, N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 65536
LEVEL = 6
SEED = 1234
# algorithmic bytes per env-step
#   level 6: obs C*H*W = 12*12*13 = 1872, state r/w 2*24 = 48, actions 4, avail 4, events 1 + 2*4 = 9
ALGO_BYTES_PER_ENV_STEP = 1937
ALGO_BYTES_CFG2 = 953      # level 1: 936 + 12 + 1 + 1 + 3
ALGO_BYTES_CFG5 = 20617    # 32x32, 8 agents, 8 sources: 20480 + 104 + 8 + 8 + 17
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20
PLACEMENT_CANDIDATES = 4  # arenas tried for the blocks whose rows go to HBM (transient: 4 x 1.4 GB for config 5)
HBM_REGIME_ENVS = 262144   # rows of one launch: 491 MB > Infinity Cache
PREROLL_SECONDS = 1.0      # untimed launches that bring the clocks up before the warm-up (reported as preroll_steps)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(n_envs, seconds_target=20.0):
    """The CPU oracle (C restatement of the reference algorithm, kind "port") on a bounded sample of the same workload:
    the same n_envs level-6 environments, sampled actions + auto-reset + int8 layered observation.  The thread count is
    SWEPT -- 64, 128 and every core this process may run on (deduplicated, capped at the visible cores), threads pinned to
    distinct CPUs spread evenly over the allowed set, disjoint env ranges; then the best count once more with placement left
    to the scheduler -- for about `seconds_target` seconds in total; every point is reported and the best one is quoted as
    `value`."""
    import numpy as np

    from oracle import oracle
    from oracle.levels import LEVELS

    level_text = LEVELS[LEVEL]
    cores = host_cores()
    counts = sorted({max(1, min(c, cores)) for c in (64, 128, cores)})
    ob = oracle.OracleBatch(level_text, n_envs)
    obs = np.zeros((n_envs, ob.C * ob.H * ob.W), np.int8)
    oracle.set_thread_pinning(True)
    ob.rollout(2, SEED, counts[-1], obs)  # warm-up: pages of `obs` touched, worlds in cache
    per_point = seconds_target / (len(counts) + 1.25)
    sweep = []
    for threads, pinned in [(c, True) for c in counts] + [(None, False)]:
        if threads is None:  # one more point: the best pinned thread count again, placement left to the scheduler
            threads = max(sweep, key=lambda p: p["env_steps_per_s"])["threads"]
        oracle.set_thread_pinning(pinned)
        cal = 4
        t0 = time.perf_counter()
        ob.rollout(cal, SEED, threads, obs)
        rate = n_envs * cal / (time.perf_counter() - t0)
        steps = max(4, int(per_point * rate / n_envs))
        t0 = time.perf_counter()
        ob.rollout(steps, SEED, threads, obs)
        dt = time.perf_counter() - t0
        sweep.append({"threads": threads, "pinned": pinned, "steps": steps, "seconds": dt, "env_steps_per_s": n_envs * steps / dt,
                      "agent_steps_per_s": ob.A * n_envs * steps / dt})
    best = max(sweep, key=lambda p: p["env_steps_per_s"])
    steps1 = max(2, int(0.25 * per_point * best["env_steps_per_s"] / best["threads"] / n_envs))
    t1 = time.perf_counter()
    ob.rollout(steps1, SEED, 1, obs)
    dt1 = time.perf_counter() - t1
    oracle.set_thread_pinning(False)
    return {
        "value": best["agent_steps_per_s"], "unit": "agent-steps/s", "cores": best["threads"], "kind": "port",
        "threads_used": best["threads"], "threads_pinned": best["pinned"], "host_cores_visible": cores, "host_cores_total": os.cpu_count(),
        "env_steps_per_s": best["env_steps_per_s"], "seconds": sum(p["seconds"] for p in sweep), "thread_sweep": sweep,
        "single_thread_env_steps_per_s": n_envs * steps1 / dt1,
        "sample": f"level {LEVEL}, {n_envs} envs, sampled actions + auto-reset + int8 layered obs, C restatement of the Rust reference "
                  f"algorithm (oracle/lle_oracle.c); thread sweep {counts} (pinned, spread evenly over the allowed CPUs) + the best count unpinned, on {cores} visible host cores, "
                  f"{'/'.join(str(p['steps']) for p in sweep)} steps per point; best: {best['threads']} threads{' pinned' if best['pinned'] else ' unpinned'}, "
                  f"{best['steps']} steps in {best['seconds']:.1f} s",
    }


def load_traffic(key="hbm_bytes_per_launch"):
    """HBM bytes per launch from the committed PMC profile of this same command (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except Exception:  # noqa: BLE001
        return None


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs this process would see, WITHOUT loading HIP: the parent of an N-rank run must stay off the GPU (on ROCm,
    torch.cuda.device_count() may run hsa_init and keep /dev/kfd open for the whole run).  KFD topology nodes with SIMDs are
    GPUs -- those whose DRM render node this process can open --; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES narrow them.  Falls back to a short-lived child
    process that asks torch when sysfs is not there."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        count = 0
        for d in sorted(os.listdir(nodes)):
            with open(os.path.join(nodes, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                # a container may be given some of the host's GPUs only: the render node must be there and usable
                minor = props.get("drm_render_minor")
                if minor is None or os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
                    count += 1
    except OSError:
        count = None
    if count is None:
        return gpus_seen_by_a_child()
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def gpus_seen_by_a_child():
    """torch.cuda.device_count() asked in a short-lived child process (the parent stays off the GPU)."""
    try:
        res = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True, timeout=300)
        return int(res.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return 0


def spawn_ranks(args, argv):
    """Parent of an N-rank run.  Never touches a GPU (visible_gpus() reads sysfs) and never execs: the ranks are child
    processes of torch.distributed.run, and this process exits with their code.  The ranks check their own device again."""
    if not args.plumbing_only and not args.rehearse_on_one_gpu:
        visible = visible_gpus()
        if visible < args.gpus:  # (sysfs says too few: before refusing, ask the runtime itself -- in a child)
            visible = max(visible, gpus_seen_by_a_child())
        if visible < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {visible} GPU(s) visible; refusing to measure fewer GPUs than asked for",
                  file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port or free_port()),
           os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def plumbing_only(args, real_stdout):
    """Spawn / rendezvous / one-line check without any GPU (tests/test_bench_spawn.py): every rank contributes known
    counters through the same allreduce helpers over gloo; rank 0 prints one line.  Not a measurement."""
    import torch
    import torch.distributed as dist

    from lle_amd import _capi
    from lle_amd.distributed import STAT_KEYS, allreduce_max, allreduce_stats, gather_rows, shard_check
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    stats = {k: (rank + 1) * (i + 1) for i, k in enumerate(STAT_KEYS)}
    dev = torch.device("cpu")
    # (the same region record every rank of a real run contributes: wall s, kernel ms per launch, env-steps, sustained wall s, kernel ms)
    mine = [float(rank + 1), 0.02 * (rank + 1), 65536.0 * 20, 2.0 * (rank + 1), 0.019 * (rank + 1), 0.9 * (rank + 1)]
    if dist.is_initialized():
        stats = allreduce_stats(stats, dev)
        elapsed = allreduce_max(float(rank + 1), dev)
    else:
        elapsed = 1.0
    rows = gather_rows(mine, dev)
    # the shard check's plumbing (all-gather of 64-bit hashes, rank 0's recomputation): the "window" is the host-side action hash
    # of the shard's first environment -- a function of env_offset alone, like the real window's checksum
    check = shard_check(lambda off: int(_capi.lib().lle_action_hash(SEED, off, 8, 0)) ^ (off * 0x9E3779B97F4A7C15 & (2**64 - 1)),
                        65536, rank, world, dev)
    if rank == 0:
        line = {"plumbing_only": True, "shard_check": check, "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                "backend": "gloo", "rollout_stats": stats, "elapsed_max": elapsed,
                **scaling_block(rows, steps=20, sustained_steps=2000, n_envs=65536, agents=4)}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def scaling_block(rows, steps, sustained_steps, n_envs, agents, n1_reference=None):
    """What a reader needs to diagnose an N-rank line: every rank's own numbers and the region both ways.
    rows[r] = [wall seconds of the --steps region, HIP-event ms per launch in it, env-steps stepped in it,
               wall seconds of the sustained region (0 = none), HIP-event ms per launch in it,
               wall seconds of the --steps region WITHOUT the closing barrier].
    `value` of the line follows the contract (wall clock around barrier + synchronize, max over ranks); the HIP-event figures
    say what the GPUs did inside it: at --steps 20 the region is 0.4 ms long and the rank skew and the host's wake-up are a
    visible share of it."""
    world = len(rows)
    per_rank = [{"rank": r, "wall_ms_per_step": row[0] / steps * 1e3, "wall_ms_per_step_without_closing_barrier": row[5] / steps * 1e3,
                 "kernel_ms": row[1], "env_steps": int(row[2]),
                 "agent_steps_per_s_by_events": agents * n_envs / (row[1] * 1e-3) if row[1] > 0 else None,
                 "sustained_wall_ms_per_step": row[3] / sustained_steps * 1e3 if sustained_steps and row[3] > 0 else None,
                 "sustained_kernel_ms": row[4] if sustained_steps and row[4] > 0 else None} for r, row in enumerate(rows)]
    k_max, k_min = max(row[1] for row in rows), min(row[1] for row in rows)
    wall_max = max(row[0] for row in rows)
    out = {"per_rank": per_rank,
           "region": {"wall_ms_per_step_max_over_ranks": wall_max / steps * 1e3,
                      "wall_ms_per_step_without_closing_barrier_max_over_ranks": max(row[5] for row in rows) / steps * 1e3,
                      "kernel_ms_max_over_ranks": k_max,
                      "kernel_ms_min_over_ranks": k_min,
                      "agent_steps_per_s_by_slowest_rank_events": agents * n_envs * world / (k_max * 1e-3) if k_max > 0 else None,
                      "host_share_of_wall": 1.0 - k_max * steps * 1e-3 / wall_max if wall_max > 0 else None,
                      "note": "wall = the contract's clock (barrier + synchronize on both sides, max over ranks): what `value` is computed from; "
                              "without_closing_barrier = each rank's clock stopped at its own synchronize, before the closing barrier (the same clock at N = 1); "
                              "kernel = HIP events on each rank's launch stream INSIDE the same run of launches (behind the first call and ahead of the last: K - 2 launches timed, the records themselves hidden behind running kernels)"}}
    if sustained_steps and all(row[4] > 0 for row in rows):
        s_k = max(row[4] for row in rows)
        out["region"]["sustained_kernel_ms_max_over_ranks"] = s_k
        if n1_reference:
            # weak scaling: every rank steps the same n_envs; N ranks at the speed of one alone would take the same time per launch
            out["weak_scaling_vs_n1"] = {"n1_sustained_kernel_ms": n1_reference, "sustained_kernel_ms_max_over_ranks": s_k,
                                         "ratio": n1_reference / s_k,
                                         "note": "the slowest rank's sustained launch time against the N = 1 sustained launch time recorded "
                                                 "on this host by an earlier `bench.py --gpus 1` (a diagnostic; the driver computes efficiency "
                                                 "from the per-N `value`s itself)"}
    return out


def n1_cache_path(n_envs, sustained_steps):
    """One file per (user, envs per GPU, sustained launches): another user's or another workload's N = 1 figure is never picked up."""
    uid = os.getuid() if hasattr(os, "getuid") else 0
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"lle_amd_bench_n1_u{uid}_e{n_envs}_s{sustained_steps}.json")


def n1_reference(n_envs, sustained_steps, write=None):
    """The N = 1 sustained launch time (ms) of an earlier run of the same workload by this user on this host, for the diagnostic of
    N > 1 runs; `write`: record it."""
    path = n1_cache_path(n_envs, sustained_steps)
    try:
        if write is not None:
            with open(path, "w") as f:
                json.dump({"sustained_kernel_ms": write, "host": socket.gethostname(), "time": time.time(), "envs_per_gpu": n_envs,
                           "sustained_steps": sustained_steps}, f)
            return write
        with open(path) as f:
            d = json.load(f)
        if (d.get("host") == socket.gethostname() and time.time() - d.get("time", 0) < 6 * 3600 and d.get("envs_per_gpu") == n_envs
                and d.get("sustained_steps") == sustained_steps):
            return float(d["sustained_kernel_ms"])
    except Exception:  # noqa: BLE001
        pass
    return None


class Timer:
    """K launches bracketed the way the contract asks: barrier + synchronize on both sides, every rank's own wall clock
    (the caller takes the MAX over ranks) and HIP events on the launch stream (torch's current stream is the one every
    launch uses; since the end of round 4 the two records sit behind the first and ahead of the last call, see run()).  Opening: synchronize, barrier, synchronize, clock starts -- the ranks start together.  Closing: synchronize
    (`wall_open` stops here: this rank's own launches are done), barrier, synchronize, `wall` stops -- the contract's clock, the
    one `value` is computed from.  At N = 1 there is no barrier and the two are the same clock; at N > 1 `wall` carries one
    small all-reduce (tens of microseconds: 5-10 % of the driver's 0.4-ms region of 20 steps), `wall_open` does not but is
    optimistic by the start skew after the opening barrier: both are reported (region.*_without_closing_barrier).
    (Round 3 quoted `wall_open` as the value; rounds 1-2 and this one the contract's clock.)"""

    def __init__(self, torch, dist, dev, use_dist, sync_dev=None):
        self.torch, self.dist, self.use_dist = torch, dist, use_dist
        self.dev = sync_dev if sync_dev is not None else dev  # the GPU this rank launches on
        self.wall_open = 0.0  # the last region's clock without the closing barrier
        self.issue = 0.0      # ... and the host time its K calls took to issue

    def sync(self):
        self.torch.cuda.synchronize(self.dev)
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def run(self, fn, k):
        torch = self.torch
        self.sync()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # The two event records are harness work, not work of the K steps: issued at the region's edges they cost it 0.5 us per step at
        # K = 20 (tools/region_edges.py: 21.0 us per step with them, 20.5 without).  So they sit INSIDE the run of launches -- ev0 behind
        # the first call, ev1 ahead of the last one, where the GPU is busy and the host's microseconds are hidden -- and time the K - 2
        # calls between them; the clock still brackets exactly K calls with a synchronize (and barrier) on both sides.
        inner = k >= 4
        t0 = time.perf_counter()
        if inner:
            fn()
            ev0.record()
            for _ in range(k - 2):
                fn()
            ev1.record()
            fn()
        else:
            ev0.record()
            for _ in range(k):
                fn()
            ev1.record()
        self.issue = time.perf_counter() - t0  # the host's own time: K calls (and the two records) issued, nothing waited for
        torch.cuda.synchronize(self.dev)  # (blocking: a spin on ev1.query() cost 0.5 us per step more)
        wall = self.wall_open = time.perf_counter() - t0
        if self.use_dist:
            self.dist.barrier()
            torch.cuda.synchronize(self.dev)
            wall = time.perf_counter() - t0
        return wall, ev0.elapsed_time(ev1) / ((k - 2) if inner else k)  # seconds (the contract's clock), ms per launch


def stepper(bw, offset=0):
    """One step per call, arguments and stream bound once (BatchedWorld.sampled_stepper): the host side of a step is the C-ABI call."""
    return bw.sampled_stepper(auto_reset=True, seed=SEED, env_offset=offset)


def preroll(torch, dev, fn, seconds=PREROLL_SECONDS):
    """Untimed launches until `seconds` of wall time have passed: the clocks of a fresh box ramp up over the first tens of
    milliseconds, and a 20-step timed region is 0.5 ms long."""
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(256):
            fn()
        torch.cuda.synchronize(dev)
        n += 256
    return n


def measure_config(torch, timer, dev, map_or_text, n_envs, algo_bytes, steps, label, traffic_key=None, fused=None, fill_ceiling=False):
    """One secondary configuration (N = 1): K single-step launches after a warm-up, HIP-event timed."""
    from lle_amd import BatchedWorld
    # past the Infinity Cache the write rate depends on where the arena landed (profiles/r03_hbm_fronts.md): where the line reports a
    # fill ceiling, the batch is created the way a long-running host would create it -- a few candidate arenas, the fastest kept
    bw = BatchedWorld(map_or_text, n_envs, device=dev, placement_candidates=PLACEMENT_CANDIDATES if fill_ceiling else None)
    fn = stepper(bw)
    for _ in range(max(20, steps // 10)):
        fn()
    wall, ms = timer.run(fn, steps)
    info = bw.kernel_info()
    m = bw.map
    achieved = algo_bytes * n_envs / (ms * 1e-3) / 1e9
    rows = n_envs * m.obs_stride
    out = {
        "workload": label, "n_envs": n_envs, "agents": m.n_agents, "steps": steps, "ms_per_step": wall / steps * 1e3, "kernel_ms": ms,
        "env_steps_per_s": n_envs * steps / wall, "agent_steps_per_s": m.n_agents * n_envs * steps / wall,
        "algorithmic_bytes_per_env_step": algo_bytes, "achieved_GBps": achieved, "frac_of_hbm_peak": achieved / HBM_PEAK_GBS,
        "row_bytes": m.obs_bytes, "row_stride": m.obs_stride, "rows_MB_per_launch": rows / 1e6,
        "bound": ("hbm" if os.environ.get("LLE_PINGPONG", "1") == "0" else "hbm + infinity cache (alternating walk)")
                 if rows > INFINITY_CACHE_BYTES else "infinity-cache-absorbed",
        # rows > cache: launches walk the envs alternately up and down (LLE_PINGPONG, obs_stream.hpp xcd_block_dir): at most this
        # share of a launch's rows is still in the 256 MB Infinity Cache from the launch before and is rewritten there
        "walk": "alternating" if rows > INFINITY_CACHE_BYTES and os.environ.get("LLE_PINGPONG", "1") != "0" else "ascending",
        "infinity_cache_share": min(1.0, INFINITY_CACHE_BYTES / rows),
        "kernel": info["kernel"], "envs_per_wave": info["envs_per_wave"], "lds_bytes_per_workgroup": info["lds_bytes"],
        "traffic": load_traffic(traffic_key) if traffic_key else None,
        "rollout_stats": bw.stats(),
    }
    if rows > INFINITY_CACHE_BYTES:
        # the same launches walked one way only (LLE_PINGPONG=0): nothing of a launch's rows is found in the Infinity Cache, every
        # row goes to DRAM -- the only figure of this block that may be quoted against the HBM peak
        from lle_amd import _capi
        old = os.environ.get("LLE_PINGPONG")
        os.environ["LLE_PINGPONG"] = "0"
        _capi.refresh_tuning()
        for _ in range(6):
            fn()
        _, d_ms = timer.run(fn, max(20, steps // 2))
        if old is None:
            os.environ.pop("LLE_PINGPONG")
        else:
            os.environ["LLE_PINGPONG"] = old
        _capi.refresh_tuning()
        for _ in range(4):
            fn()
        d_ach = algo_bytes * n_envs / (d_ms * 1e-3) / 1e9
        out["dram_only"] = {"walk": "ascending (LLE_PINGPONG=0)", "kernel_ms": d_ms, "achieved_GBps": d_ach, "frac_of_hbm_peak": d_ach / HBM_PEAK_GBS,
                            "note": "every row of every launch is written to DRAM: the figure to read against the 8 TB/s HBM peak"}
    if bw.placement:
        out["placement"] = dict(bw.placement, note="BatchedWorld(placement_candidates=k): k arenas allocated side by side, the step kernel's store "
                                                   "pattern timed on each (us per launch), the fastest kept, the rest released before the timed region")
    if fill_ceiling:
        # what THIS box gives a writer of the same shape: (a) the step kernel's own store pattern without a state machine
        # (lle_batch_probe_row_fill), (b) a memset-class fill of the same bytes (a narrow write front); ~10 ms each.  Boxes
        # differ by up to 15 % past the Infinity Cache (NOTEBOOK.md section 4 "Two kinds of box"): read `frac_of_fill`.
        probe = bw.row_fill_prober()
        launches = max(20, min(400, int(10e-3 / (ms * 1e-3))))
        for _ in range(5):
            probe()
        _, p_ms = timer.run(probe, launches)
        rows_t = bw.obs_rows

        def memset():
            rows_t.zero_()
        for _ in range(5):
            memset()
        _, m_ms = timer.run(memset, launches)
        bw.observe()
        out["fill_ceiling"] = {"row_fill_us": p_ms * 1e3, "row_fill_GBps": rows / (p_ms * 1e-3) / 1e9,
                               "memset_us": m_ms * 1e3, "memset_GBps": rows / (m_ms * 1e-3) / 1e9, "launches": launches,
                               "frac_of_fill": p_ms / ms, "frac_of_memset": m_ms / ms,
                               "note": "row_fill = the step kernel's stores, rows per wavefront and block mapping with no state machine "
                                       "(lle_batch_probe_row_fill); memset = torch fill of the same bytes; frac = that time / the step kernel's"}
    if fused:  # (T, R): the same rollout as lle_batch_rollout, T steps per launch into a ring of R slots
        T, R = fused
        ring = bw.make_ring(R)

        def roll():
            bw.rollout(T, auto_reset=True, seed=SEED, ring=ring, ring_pos=bw.t)
        for _ in range(3):
            roll()
        launches = max(8, steps // T)
        fw, _ = timer.run(roll, launches)
        out["fused_rollout"] = {"steps_per_launch": T, "ring_slots": R, "ring_MB": R * rows / 1e6, "steps": launches * T,
                                "ms_per_step": fw / (launches * T) * 1e3, "env_steps_per_s": n_envs * launches * T / fw,
                                "agent_steps_per_s": m.n_agents * n_envs * launches * T / fw,
                                "note": "a launch of 4 096 envs is launch-bound (min 4.7 us); several steps per launch are not"}
        del ring
    del bw
    torch.cuda.empty_cache()
    return out


def measure_multi_map(torch, timer, dev, steps, single_map_kernel_ms, single_map_frac_of_fill=None):
    """SURVEY.md section 8(d), stretch variant of configs[4]: per-env DISTINCT maps -- 1 024 (and 4 096) different `mapgen.config5(seed)`
    maps in one batch of 65 536 envs, 64 (16) envs per map (lle_batch_create_multi: what a learner on generated maps steps,
    python/lle/generator/world_builder.py:84-89).  Same launches as the cfg5 block: sampled actions + auto-reset + int8 layered obs."""
    from lle_amd import BatchedWorld, mapgen
    out = {"what": "BASELINE configs[4] with a different generated 32x32 map per block of envs; kernel_ms by HIP events", "n_envs": 65536,
           "single_map_kernel_ms": single_map_kernel_ms, "single_map_frac_of_fill": single_map_frac_of_fill,
           "note": "every batch sits on its own arena, and past the Infinity Cache an arena's write rate is a lottery (row_fill_us: the same stores with no state "
                   "machine, 175-230 us by arena): vs_single_map compares raw times across arenas, vs_single_map_at_equal_fill each launch against its OWN arena's "
                   "fill (profiles/r05_multi_map.md)"}
    for n_maps in (1024, 4096, 8192):  # (8 192 x 8: one wavefront per map, supported since round 5)
        per = 65536 // n_maps
        t0 = time.perf_counter()
        # (placed like the single-map block it is compared with: past the Infinity Cache the arena's write rate is a lottery)
        bw = BatchedWorld([mapgen.config5(seed) for seed in range(n_maps)], 65536, device=dev,
                          placement_candidates=PLACEMENT_CANDIDATES if n_maps <= 1024 else 2)
        create_s = time.perf_counter() - t0
        fn = stepper(bw)
        for _ in range(max(10, steps // 10)):
            fn()
        wall, ms = timer.run(fn, steps)
        probe = bw.row_fill_prober()
        for _ in range(4):
            probe()
        _, p_ms = timer.run(probe, max(10, steps // 4))
        bw.observe()
        achieved = ALGO_BYTES_CFG5 * 65536 / (ms * 1e-3) / 1e9
        out[f"maps{n_maps}_x{per}"] = {"n_maps": n_maps, "envs_per_map": per, "steps": steps, "kernel_ms": ms, "ms_per_step": wall / steps * 1e3,
                                       "achieved_GBps": achieved, "vs_single_map": ms / single_map_kernel_ms if single_map_kernel_ms else None,
                                       "row_fill_us": p_ms * 1e3, "frac_of_fill": p_ms / ms,
                                       "vs_single_map_at_equal_fill": (single_map_frac_of_fill / (p_ms / ms)) if single_map_frac_of_fill else None,
                                       "placement": bw.placement,
                                       "create_s": create_s, "table_MB": n_maps * bw.maps[0].table_bytes / 1e6, "kernel": bw.kernel_info(),
                                       "rollout_stats": bw.stats()}
        del bw, fn
        torch.cuda.empty_cache()
    return out


def measure_incremental(torch, timer, dev, steps):
    """LLE_STEP_INCREMENTAL_OBS (opt-in; NOT the headline): single steps that write only the 128-byte lines of each row that dynamic state
    can change -- the WALL / VOID / EXIT planes and the beam-less parts of the laser planes hold the same bytes after every step and are
    already in LLE_BUF_OBS from the last full write.  The buffer's content after every step is identical (tests/test_gpu_parity.py
    test_incremental_observation_is_the_full_observation); the bytes WRITTEN per env-step are `written_bytes_per_env_step`, and the
    fractions below are computed from those, not from the 1 937 / 20 617 algorithmic bytes of the full rewrite."""
    from lle_amd import BatchedWorld, Map, mapgen
    out = {"what": "same workload as the headline / roofline_hbm / cfg5 blocks, rows written incrementally (opt-in flag); us per step by HIP events",
           "steps": steps}
    for label, m, n, small in (("level6_65536", Map(level=LEVEL), 65536, 65), ("level6_262144", Map(level=LEVEL), 262144, 65),
                               ("cfg5_65536", Map(mapgen.config5(0)), 65536, 137)):
        bw = BatchedWorld(m, n, device=dev)
        full, incr = bw.sampled_stepper(auto_reset=True, seed=SEED), bw.sampled_stepper(auto_reset=True, seed=SEED, incremental_obs=True)
        k = steps if n == 65536 and label.startswith("level6") else max(20, steps // 2)
        for _ in range(10):
            full()
        _, f_ms = timer.run(full, k)
        for _ in range(10):
            incr()
        wall, i_ms = timer.run(incr, k)
        written = m.dyn_row_bytes + small  # rows + state r/w, actions, availability, events (SURVEY section 8(d) without the observation)
        out[label] = {"full_rewrite_us": f_ms * 1e3, "incremental_us": i_ms * 1e3, "speedup": f_ms / i_ms,
                      "row_bytes": m.obs_stride, "dyn_row_bytes": m.dyn_row_bytes, "written_bytes_per_env_step": written,
                      "written_GBps": written * n / (i_ms * 1e-3) / 1e9, "written_frac_of_8TBps": written * n / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "env_steps_per_s": n / (i_ms * 1e-3), "agent_steps_per_s": m.n_agents * n / (i_ms * 1e-3)}
        del bw, full, incr
        torch.cuda.empty_cache()
    return out


def measure_consumer_loop(torch, timer, dev, steps):
    """step -> reader -> step on the launch stream: what a policy does between two steps is READ the observation (its first layer),
    which changes what the Infinity Cache holds when the next step rewrites the rows.  Reader = an int8 -> fp16 cast of the whole
    observation into a separate buffer (python/lle/env/env.py:165-189: the caller of LLE.step consumes `obs` before it steps again).
    Per workload: step alone, reader alone, the pair -- with the alternating walk on and off where the rows exceed the cache.
    The walk's default (capi.cpp pingpong_pays) is read off the `pair_us` columns of this block."""
    from lle_amd import BatchedWorld, Map, _capi, mapgen
    out = {"what": "us per launch (HIP events): World.step alone, an int8->fp16 cast of the whole observation alone, step + cast alternating on one stream; "
                   "step_fp16_us / step_fp32_us: the step of a batch created with obs_dtype = fp16 / fp32 (rows widened at the store: no cast pass)",
           "steps": steps}
    for label, m, n in (("level6_65536", Map(level=LEVEL), 65536), ("level6_262144", Map(level=LEVEL), 262144),
                        ("cfg5_65536", Map(mapgen.config5(0)), 65536)):
        bw = BatchedWorld(m, n, device=dev)
        rows = bw.obs_rows
        big = rows.numel() > INFINITY_CACHE_BYTES
        half = torch.empty(rows.shape, dtype=torch.float16, device=dev)
        step = stepper(bw)
        read_rows, st = _capi.lib().lle_probe_read_rows, torch.cuda.current_stream(dev).cuda_stream
        r_ptr, h_ptr, r_bytes = rows.data_ptr(), half.data_ptr(), rows.numel()

        def reader():  # (torch's own int8 -> fp16 copy_ runs at 0.5 TB/s of reads: a stand-in that slow would hide the step)
            read_rows(r_ptr, h_ptr, r_bytes, st)

        def pair():
            step()
            reader()
        blk = {"rows_MB": rows.numel() / 1e6, "reader_out_MB": half.numel() * 2 / 1e6}
        k = max(20, steps if not big else steps // 2)
        for walk in (("on", "off") if big else ("default",)):
            if walk != "default":
                os.environ["LLE_PINGPONG"] = "1" if walk == "on" else "0"
                _capi.refresh_tuning()
            for _ in range(8):
                pair()
            _, s_ms = timer.run(step, k)
            _, r_ms = timer.run(reader, k)
            _, p_ms = timer.run(pair, k)
            blk["walk_" + walk] = {"step_us": s_ms * 1e3, "reader_us": r_ms * 1e3, "pair_us": p_ms * 1e3, "step_in_pair_us": (p_ms - r_ms) * 1e3}
        if big:
            os.environ.pop("LLE_PINGPONG", None)
            _capi.refresh_tuning()
            blk["walk_that_wins_the_pair"] = "on" if blk["walk_on"]["pair_us"] <= blk["walk_off"]["pair_us"] else "off"
        # the same step with the rows leaving the kernel in the learner's type (lle_batch_options.obs_dtype: the kernels widen at the store):
        # what replaces the step + cast pair above.  Bytes = what the launch writes (rows x element size + the small outputs).
        del half
        for dt_name, dt in (("fp16", torch.float16), ("fp32", torch.float32)):
            if rows.numel() * dt.itemsize > 6 << 30:
                continue
            wide = BatchedWorld(m, n, device=dev, obs_dtype=dt)
            wstep = stepper(wide)
            for _ in range(8):
                wstep()
            _, w_ms = timer.run(wstep, k)
            wbytes = rows.numel() * dt.itemsize + n * (ALGO_BYTES_PER_ENV_STEP - 1872 if label.startswith("level6") else 137)
            blk[f"step_{dt_name}_us"] = w_ms * 1e3
            blk[f"step_{dt_name}"] = {"us": w_ms * 1e3, "rows_MB": rows.numel() * dt.itemsize / 1e6, "written_GBps": wbytes / (w_ms * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": wbytes / (w_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "vs_int8_step_plus_cast_us": blk[("walk_on" if big else "walk_default")]["pair_us"] if dt_name == "fp16" else None}
            del wide, wstep
            torch.cuda.empty_cache()
        out[label] = blk
        del bw, rows
        torch.cuda.empty_cache()
    # the fused rollout into a two-slot ring with the reader trailing one slot (a double buffer: slot t is cast while t + 1 is written)
    bw = BatchedWorld(Map(level=LEVEL), 65536, device=dev)
    ring = bw.make_ring(2)
    half = torch.empty(ring["obs_rows"][0].shape, dtype=torch.float16, device=dev)
    state = {"t": 0}

    read_rows, st = _capi.lib().lle_probe_read_rows, torch.cuda.current_stream(dev).cuda_stream
    slots = [ring["obs_rows"][k].data_ptr() for k in range(2)]
    h_ptr, r_bytes = half.data_ptr(), ring["obs_rows"][0].numel()

    def ring_pair():
        t = state["t"]
        bw.rollout(1, auto_reset=True, seed=SEED, ring=ring, ring_pos=t)
        read_rows(slots[(t + 1) % 2], h_ptr, r_bytes, st)  # the slot written by the launch before this one
        state["t"] = t + 1
    for _ in range(8):
        ring_pair()
    _, rp_ms = timer.run(ring_pair, max(20, steps))
    out["level6_65536"]["ring2_reader_trailing_one_slot_pair_us"] = rp_ms * 1e3
    del bw, ring, half
    torch.cuda.empty_cache()
    return out


def measure_lle_step(torch, timer, dev, n, steps):
    """The reference's host class around the path, batched (lle_amd.BatchedLLE): LLE.step =
    World.step + observation + state + reward + done + available_actions, with auto-reset, on level 6 x n envs.  The joint
    actions are a recorded random rollout of the same world (valid along the same trajectory); with randomize_lasers the
    trajectories part at the first re-coloured reset and a recorded action may be refused (the kernels do the same work)."""
    from lle_amd import BatchedLLE, BatchedWorld, Map
    W = 32                      # untimed steps per variant
    K = W + steps               # every step of a variant replays its own slot: on the recorded trajectory throughout
    rec = BatchedWorld(Map(level=LEVEL), n, device=dev)
    ring = rec.make_ring(K)
    rec.rollout(K, auto_reset=True, seed=SEED, ring=ring, ring_pos=0)
    actions = ring["actions"].clone()  # [K, n, A]
    del ring, rec
    out = {"what": f"lle_amd.BatchedLLE.step(actions, auto_reset=True) on World.level({LEVEL}) x {n} envs: obs (layered) + state + reward + "
                   "done + available_actions per step; us per step, launch-to-launch", "steps": steps}
    for key, kw, mode in (("default_us", {}, {}),  # (since round 4 the default step is the one launch, into tensors allocated per step)
                          ("two_launches_us", {}, {"fused": False}), ("two_launches_persistent_us", {}, {"persistent": True}), ("one_launch_us", {}, {"fused": True}),
                          ("randomize_lasers_default_us", {"randomize_lasers": True}, {}),
                          ("randomize_lasers_two_launches_us", {"randomize_lasers": True}, {"fused": False}),
                          ("randomize_lasers_one_launch_us", {"randomize_lasers": True}, {"fused": True}),
                          # opt-in incremental rows (LLE_STEP_INCREMENTAL_OBS): not the default path
                          ("one_launch_incremental_us", {"incremental_obs": True}, {"fused": True}),
                          ("randomize_lasers_one_launch_incremental_us", {"randomize_lasers": True, "incremental_obs": True}, {"fused": True}),
                          # obs_type="partial7x7": step + partial observer + outputs (persistent: bound calls), and all of it in the step launch
                          ("partial7x7_three_launches_persistent_us", {"obs_type": "partial7x7"}, {"persistent": True}),
                          ("partial7x7_one_launch_us", {"obs_type": "partial7x7"}, {"fused": True})):
        env = BatchedLLE(Map(level=LEVEL), n, device=dev, seed=SEED, **kw)
        env.reset()
        state = {"t": 0}

        def step():
            env.step(actions[state["t"] % K], auto_reset=True, **mode)
            state["t"] += 1
        for _ in range(W):
            step()
        wall, _ = timer.run(step, steps)
        assert state["t"] == K
        out[key] = wall / steps * 1e6
        del env
    torch.cuda.empty_cache()
    return out


def measure_observers(torch, timer, dev, n, steps):
    """The other observation builders on the headline batch and on config 5: launch-to-launch time of the
    bound calls (BatchedWorld.bound_*: the C-ABI call and nothing else per launch) and bytes written / time."""
    from lle_amd import BatchedWorld, Map, _capi, mapgen
    out = {"what": "observers.hip builders through bound calls, us per launch (HIP events) and GB/s of the bytes they write", "steps": steps}
    for label, m in (("level6", Map(level=LEVEL)), ("cfg5", Map(mapgen.config5(0)))):
        bw = BatchedWorld(m, n, device=dev)
        fn = stepper(bw)
        for _ in range(16):
            fn()
        # (outputs larger than the Infinity Cache are placed: k candidate buffers, the one the row stream writes fastest kept)
        k_place = PLACEMENT_CANDIDATES
        calls = {f"partial{k}x{k}": bw.bound_observer(_capi.LLE_OBS_PARTIAL, k, placement_candidates=k_place) for k in (3, 5, 7)}
        if label == "level6":  # (config 5's perspective tensor is 10.7 GB: left to tools/lle_prof.py observers)
            calls["perspective"] = bw.bound_observer(_capi.LLE_OBS_PERSPECTIVE, placement_candidates=k_place)
        calls["layered_padded2"] = bw.bound_observer(_capi.LLE_OBS_LAYERED_PADDED, 2, placement_candidates=k_place)
        calls["state"] = bw.bound_observer(_capi.LLE_OBS_STATE)
        calls["available_actions"] = bw.bound_available_actions(True)
        calls["available_actions_no_walkable_lasers"] = bw.bound_available_actions(False)
        blk = {}
        for name, call in calls.items():
            for _ in range(10):
                call()
            _, ms = timer.run(call, steps)
            nbytes = call.buffer.numel() * call.buffer.element_size()
            blk[name] = {"us": ms * 1e3, "MB": nbytes / 1e6, "GBps": nbytes / (ms * 1e-3) / 1e9}
            if getattr(call, "placement", None):
                blk[name]["placement"] = call.placement
        out[label] = blk
        del calls, bw
        torch.cuda.empty_cache()
    # the same builders on a batch whose element type is fp16 (lle_batch_options.obs_dtype: every layered-style builder widens at the store)
    bw = BatchedWorld(Map(level=LEVEL), n, device=dev, obs_dtype=torch.float16)
    fn = stepper(bw)
    for _ in range(16):
        fn()
    blk = {}
    for name, kind, param in (("partial7x7", _capi.LLE_OBS_PARTIAL, 7), ("layered_padded2", _capi.LLE_OBS_LAYERED_PADDED, 2), ("perspective", _capi.LLE_OBS_PERSPECTIVE, 0)):
        call = bw.bound_observer(kind, param)
        for _ in range(10):
            call()
        _, ms = timer.run(call, steps)
        nbytes = call.buffer.numel() * call.buffer.element_size()
        blk[name] = {"us": ms * 1e3, "MB": nbytes / 1e6, "GBps": nbytes / (ms * 1e-3) / 1e9}
    out["level6_fp16"] = blk
    del bw, fn
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--envs-per-wave", type=int, default=0, help="0 = library default")
    ap.add_argument("--sustained-steps", type=int, default=2000, help="launches of the `sustained` block (0 = skip)")
    ap.add_argument("--config-steps", type=int, default=200, help="launches per secondary configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--autotune-ms", type=float, default=None,
                    help="GPU time lle_batch_autotune may spend on the headline batch choosing its launch rules (default: what a plain "
                         "BatchedWorld(map, n) spends at construction -- BatchedWorld.AUTOTUNE_MS; 0 = the library's default rules)")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary fused-rollout measurement")
    ap.add_argument("--no-configs", action="store_true", help="skip roofline_hbm and the cfg2 / cfg5 blocks")
    ap.add_argument("--no-multi-map", action="store_true",
                    help="skip the cfg5_multi_map block (thousands of tiny launches at batch creation: rocprofv3 --pmc crashes on them)")
    ap.add_argument("--fused-steps", type=int, default=16, help="steps per launch of the fused rollout")
    ap.add_argument("--ring-slots", type=int, default=8, help="trajectory ring slots of the fused rollout")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of a self-spawned N-rank run (0 = pick a free one)")
    ap.add_argument("--force-spawn", action="store_true", help="go through torch.distributed.run (and RCCL) even for --gpus 1")
    ap.add_argument("--plumbing-only", action="store_true", help="spawn / rendezvous / one-line check over gloo, no GPU, no measurement")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N ranks that SHARE cuda:0, collectives over gloo: every line of the N > 1 path but RCCL, on a one-GPU box (not a measurement)")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (args.gpus > 1 or args.force_spawn):
        sys.exit(spawn_ranks(args, [a for a in sys.argv[1:] if a != "--force-spawn"]))

    # Everything but the final JSON line goes to stderr, at the file-descriptor level: RCCL, gloo and the HIP runtime
    # print banners to fd 1 from C code (e.g. "Librccl path : ...") and the contract is ONE line on stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.plumbing_only:
        sys.exit(plumbing_only(args, real_stdout))

    import torch
    import torch.distributed as dist

    from lle_amd import BatchedWorld, Map
    from lle_amd.distributed import allreduce_max, allreduce_stats, gather_rows, shard_check, shard_offset, world_hash

    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a rank count that was not asked for", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU execution path)"
    rehearsal = args.rehearse_on_one_gpu and under_launcher
    if rehearsal:
        local_rank = 0  # every rank on the one card; RCCL refuses two ranks on a device, so the collectives run over gloo
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU {local_rank} ({torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    # one process per GPU under torch.distributed.run (RCCL = backend "nccl"); a plain `python bench.py` is rank 0 of 1
    use_dist = under_launcher
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        rccl_ranks = dist.get_world_size()
    # the small tensors of the collectives live where the backend can reduce them: on the GPU for RCCL, on the host for gloo
    cdev = torch.device("cpu") if rehearsal else dev
    timer = Timer(torch, dist, cdev if rehearsal else dev, use_dist, sync_dev=dev)

    n = args.envs_per_gpu
    # ---- shard invariance over the real collective, before anything is timed: every rank steps the 64
    # environments at the head of its shard for 8 steps, the ranks all-gather a 64-bit checksum of (pos, bits, beams, obs), rank 0
    # recomputes all N windows with the matching env_offset.  No oracle involved: the product against itself, across ranks.
    def window_hash(env_offset, envs=64, steps=8):
        w = BatchedWorld(Map(level=LEVEL), envs, device=dev)
        for t in range(steps):
            w.step(sample=True, auto_reset=True, seed=SEED, t=t, env_offset=env_offset)
        return world_hash(w)
    check = shard_check(window_hash, n, rank, world, cdev)
    if rank == 0 and check["status"] != "ok":
        print(f"bench.py: shard check FAILED: {check}", file=sys.stderr)

    # the headline batch is what a user gets from the constructor: since round 5 BatchedWorld autotunes at construction (10 ms by default)
    bw = BatchedWorld(Map(level=LEVEL), n, device=dev, envs_per_wave=args.envs_per_wave or None, autotune_ms=args.autotune_ms)
    offset = shard_offset(n, rank)
    tuning = dict(bw.tuning(), constructor_default=args.autotune_ms is None)
    step = stepper(bw, offset)

    bw.stats_blocks.zero_()  # (the first torch fill of the process loads a code object: not right in front of the timed region)
    preroll_steps = preroll(torch, dev, step)
    for _ in range(args.warmup):
        step()
    # A dress rehearsal of the timed region, untimed: the first pass through Timer.run in a process pays one-off host costs
    # (first timing events, first elapsed_time, cold Python / ctypes paths, and the first torch fill kernel above: ~100 us, a fifth of the driver's 20-step region;
    # measured 26.0 us per step for the first region of a process against 21.1-21.2 for every later one).
    timer.run(step, max(1, min(args.steps, 32)))
    # counters of the timed region only: zeroed by a fill on the launch stream (bw.stats(reset=True) would read them back
    # first -- a host round trip right in front of a timed region that is 0.5 ms long at the driver's 20 steps)
    bw.stats_blocks.zero_()
    my_wall, kernel_ms = timer.run(step, args.steps)
    my_wall_open, my_issue = timer.wall_open, timer.issue
    elapsed = allreduce_max(my_wall, cdev) if use_dist else my_wall
    local_stats = bw.stats()
    stats = allreduce_stats(local_stats, cdev) if use_dist else local_stats

    sustained = None
    my_s_wall = s_ms = 0.0
    if args.sustained_steps > 0:
        my_s_wall, s_ms = timer.run(step, args.sustained_steps)
        s_issue = timer.issue
        s_wall = allreduce_max(my_s_wall, cdev) if use_dist else my_s_wall
        sustained = (args.sustained_steps, s_wall, s_ms, s_issue)
    # every rank's own numbers (rank order): a poor aggregate can then be traced to the rank, or to the host side, that caused it
    rank_rows = gather_rows([my_wall, kernel_ms, local_stats["env_steps"], my_s_wall, s_ms, my_wall_open], cdev)

    # ---- secondary measurement: the same random rollout with lle_batch_rollout (several steps per launch, every
    # step's observation / actions / reward counts written to a trajectory ring larger than the caches)
    fused = []
    if not args.no_fused:
        # (T, R): the ring of args.ring_slots slots is larger than the 256 MB Infinity Cache (true HBM writes); the
        # two-slot ring is a double buffer -- a consumer reads slot t while step t + 1 is written -- and stays in it
        for T, R in ((args.fused_steps, args.ring_slots), (args.fused_steps, 2)):
            # (a ring larger than the Infinity Cache is placed like the arenas of the HBM-regime blocks: lle_amd.placement)
            ring = bw.make_ring(R, placement_candidates=PLACEMENT_CANDIDATES if world == 1 else None)
            ring_placement = ring.get("placement")
            launches = max(4, max(args.steps, 512) // T)

            def roll():
                bw.rollout(T, auto_reset=True, seed=SEED, env_offset=offset, ring=ring, ring_pos=bw.t)
            for _ in range(2):
                roll()
            fe, _ = timer.run(roll, launches)
            fe = allreduce_max(fe, cdev) if use_dist else fe
            fused.append((T, R, launches, fe, ring_placement))
            del ring

    A = bw.map.n_agents
    info = bw.kernel_info()
    row_stride = bw.map.obs_stride
    del bw
    torch.cuda.empty_cache()

    hbm = cfgs = None
    if world == 1 and not args.no_configs:
        from lle_amd import mapgen
        k = args.config_steps
        hbm = measure_config(torch, timer, dev, Map(level=LEVEL), HBM_REGIME_ENVS, ALGO_BYTES_PER_ENV_STEP, k,
                             f"World.level({LEVEL}) x {HBM_REGIME_ENVS} envs: rows of one launch exceed the 256 MB Infinity Cache",
                             "hbm_regime_bytes_per_launch", fill_ceiling=True)
        cfgs = {
            "cfg2_level1_4096": measure_config(torch, timer, dev, Map(level=1), 4096, ALGO_BYTES_CFG2, max(k, 1000),
                                               "BASELINE configs[1]: World.level(1), 1 agent, 4096 envs", "cfg2_bytes_per_launch",
                                               fused=(64, 8)),
            "cfg5_32x32_a8_l8_65536": measure_config(torch, timer, dev, mapgen.config5(0), 65536, ALGO_BYTES_CFG5, k,
                                                     "BASELINE configs[4]: generated 32x32, 8 agents, 8 lasers (mapgen.config5(0)), 65536 envs",
                                                     "cfg5_bytes_per_launch", fill_ceiling=True),
        }

    lle_step = observers = consumer = multi = incremental = None
    if world == 1 and not args.no_configs:
        incremental = measure_incremental(torch, timer, dev, max(args.config_steps, 200))
        if not args.no_multi_map:
            multi = measure_multi_map(torch, timer, dev, max(args.config_steps // 2, 50), cfgs["cfg5_32x32_a8_l8_65536"]["kernel_ms"],
                                          (cfgs["cfg5_32x32_a8_l8_65536"].get("fill_ceiling") or {}).get("frac_of_fill"))
        lle_step = measure_lle_step(torch, timer, dev, n, max(args.config_steps, 200))
        observers = measure_observers(torch, timer, dev, n, max(args.config_steps, 200))
        consumer = measure_consumer_loop(torch, timer, dev, max(args.config_steps, 200))

    if rank == 0:
        total_envs = n * world
        env_steps_s = total_envs * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n / (kernel_ms * 1e-3) / 1e9
        rows_bytes = n * row_stride
        out = {
            "metric": "env-steps/s (agents x envs x steps/s), level-6 batch 65536",
            "value": A * env_steps_s, "unit": "agent-steps/s",
            "env_steps_per_s": env_steps_s,
            "n_gpus": world, "rccl_ranks": rccl_ranks, **({"rehearsal": "N ranks SHARING one GPU, collectives over gloo: the N > 1 code path, NOT a measurement"} if rehearsal else {}),
            "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll_steps,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"World.level({LEVEL}) 4 agents 12x13, {n} envs/GPU, sampled actions + auto-reset + "
                                   "int8 layered obs (C=12)", "envs_per_gpu": n, "global_batch": total_envs,
                       "parallelism": f"env-shard x{world}", "kernel": info["kernel"], "envs_per_wave": info["envs_per_wave"],
                       "lds_bytes_per_workgroup": info["lds_bytes"]},
            # `bound`: what the stores of this launch hit.  The rows of one launch (122.7 MB, rewritten in place every
            # step) stay in the 256 MB Infinity Cache; the DRAM figure is `roofline_hbm`.
            "roofline": {"bound": "hbm" if rows_bytes > INFINITY_CACHE_BYTES else "infinity-cache-absorbed",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(),
                         "traffic_note": "L2<->fabric bytes (WRITE_SIZE + 2 x FETCH_SIZE) from the committed profile, profiles/traffic.json",
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n, "kernel_ms": kernel_ms,
                         # `kernel_ms` is a STEADY-STATE figure since the end of round 4 (events behind the first and ahead of the last of
                         # the K calls: K - 2 launches, without the cold first launch and the drain of the last; K < 4: all K) -- compare
                         # rounds on `value` / `frac_wall` (the contract's wall clock over all K calls), not on `kernel_ms`
                         "kernel_ms_kind": "steady-state (K-2 launches between the first and the last call)" if args.steps >= 4 else "all K launches",
                         "frac_wall": ALGO_BYTES_PER_ENV_STEP * n / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "rows_MB_per_launch": rows_bytes / 1e6},
            "rollout_stats": stats,
            "shard_check": dict(check, what="every rank steps the 64 envs at the head of its shard for 8 steps; the ranks all-gather a 64-bit checksum of (pos, bits, "
                                             "beams, obs) over the run's collective backend; rank 0 recomputes the N windows with the matching env_offset"),
            # the launch rules of the headline batch (lle_batch_tuning): chosen by lle_batch_autotune on this batch's own arena when
            # `autotuned`, the library's default rules otherwise; `log` = the trials (us per launch)
            "tuning": tuning,
            # host side of the timed region: the K C-ABI calls took this long to ISSUE (nothing waited for); the rest of
            # (ms_per_step - kernel_ms) is the closing synchronize's wake-up spread over K steps
            "host_issue_us_per_step": my_issue / args.steps * 1e6,
        }
        if world == 1 and sustained:
            n1_reference(n, args.sustained_steps, write=sustained[2])
        out.update(scaling_block(rank_rows, args.steps, args.sustained_steps, n, A,
                                 n1_reference(n, args.sustained_steps) if world > 1 else None))
        if sustained:
            k, s_wall, s_ms, s_issue = sustained
            s_ach = ALGO_BYTES_PER_ENV_STEP * n / (s_ms * 1e-3) / 1e9
            out["sustained"] = {"steps": k, "ms_per_step": s_wall / k * 1e3, "kernel_ms": s_ms, "host_issue_us_per_step": s_issue / k * 1e6,
                                "env_steps_per_s": total_envs * k / s_wall, "agent_steps_per_s": A * total_envs * k / s_wall,
                                "achieved_GBps_per_gpu": s_ach, "frac_of_hbm_peak": s_ach / HBM_PEAK_GBS}
        if hbm:
            out["roofline_hbm"] = {"bound": hbm["bound"], "achieved": hbm["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": hbm["frac_of_hbm_peak"],
                                   "frac_note": "with the alternating walk part of every launch's rows is rewritten inside the Infinity Cache: NOT a fraction of "
                                                "the HBM peak -- `dram_only` (same kernel, one-directional walk) is",
                                   "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * hbm["n_envs"], **hbm}
        if cfgs:
            out["configs"] = cfgs
        if lle_step:
            out["lle_step"] = lle_step
        if observers:
            out["observers"] = observers
        if consumer:
            out["consumer_loop"] = consumer
        if multi:
            out["cfg5_multi_map"] = multi
        if incremental:
            out["incremental_obs"] = incremental
        for key, (T, R, launches, fe, ring_placement) in zip(("fused_rollout", "fused_rollout_double_buffer"), fused):
            # per env-step: obs 1872 + actions 4 + reward 4 + err/evcount/done 3 + events 8, state r/w (48 B) once per launch
            fused_bytes = 1891 + 48.0 / T
            ring_mb = R * n * row_stride / 1e6
            out[key] = {
                "what": "lle_batch_rollout: same random rollout, several steps per launch, per-step obs/actions/reward to a trajectory ring"
                        + (" larger than the 256 MB Infinity Cache (true HBM writes)" if ring_mb * 1e6 > INFINITY_CACHE_BYTES else
                           " of two slots (double buffer, stays in the 256 MB Infinity Cache: NOT an HBM figure)"),
                "steps_per_launch": T, "ring_slots": R, "ring_MB": ring_mb, "steps": launches * T, "ms_per_step": fe / (launches * T) * 1e3,
                "env_steps_per_s": total_envs * launches * T / fe, "agent_steps_per_s": A * total_envs * launches * T / fe,
                "algorithmic_bytes_per_env_step": fused_bytes,
                "achieved_GBps_per_gpu": fused_bytes * n * launches * T / fe / 1e9,
                "frac_of_hbm_peak": fused_bytes * n * launches * T / fe / 1e9 / HBM_PEAK_GBS,
                **({"placement": ring_placement} if ring_placement else {}),
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
