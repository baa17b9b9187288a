"""Multi-GPU plumbing: environments shard over ranks with no data-path collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).  Each rank owns a contiguous
range of environment ids and steps them independently; the only exchange is one small all-reduce of the rollout
counters at the end of a rollout (64 B: latency-bound, the xGMI link bandwidth is irrelevant)."""
import torch
import torch.distributed as dist

STAT_KEYS = ["env_steps", "agent_steps", "gems", "exits", "deaths", "invalid", "auto_resets", "reward_sum"]


def shard_offset(envs_per_rank, rank):
    """Global id of the first environment of `rank` (used as env_offset by the action sampler, so that N ranks of
    n envs reproduce one batch of N*n envs bit for bit)."""
    return int(envs_per_rank) * int(rank)


def shard_range(n_total, rank, world):
    """Contiguous [lo, hi) split of n_total environments over `world` ranks (remainder to the first ranks)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_stats(stats, device):
    """Sum the per-rank rollout counters (dict of STAT_KEYS) over all ranks."""
    t = torch.tensor([int(stats[k]) for k in STAT_KEYS], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(STAT_KEYS, t.tolist())}


def allreduce_max(value, device):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(row, device):
    """Every rank's list of floats, in rank order, on every rank (one all_gather of a small float64 tensor): the per-rank
    timings of a benchmark region, so that a poor aggregate can be traced to the rank that caused it."""
    t = torch.tensor([float(v) for v in row], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def gather_i64(values, device):
    """Every rank's list of 64-bit integers, in rank order, on every rank (one all_gather of a small int64 tensor)."""
    t = torch.tensor([_signed64(v) for v in values], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [[int(v) & _M64 for v in t.tolist()]]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(v) & _M64 for v in o.tolist()] for o in out]


_M64 = (1 << 64) - 1


def _signed64(v):
    v = int(v) & _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def tensor_hash(t):
    """64-bit checksum of a tensor's logical bytes (position-weighted wrap-around sum of its 64-bit words, computed where the
    tensor lives): equal tensors hash equal on every device and rank."""
    b = t.contiguous().view(torch.uint8).flatten()
    pad = (-b.numel()) % 8
    if pad:
        b = torch.cat([b, b.new_zeros(pad)])
    w = b.view(torch.int64)
    idx = torch.arange(w.numel(), device=w.device, dtype=torch.int64)
    mixed = (w ^ (w >> 29)) * ((2 * idx + 1) * _signed64(0x9E3779B97F4A7C15))
    return int(mixed.sum().item()) & _M64


def world_hash(bw):
    """Checksum of a BatchedWorld's dynamic state and observation: positions, alive / arrived / occupant bits, beam masks, the
    int8 layered observation (logical bytes: no row padding, no agent-stride padding)."""
    h = 0
    for k, t in enumerate((bw.pos, bw.bits, bw.beams, bw.obs)):
        h = (h * 0x100000001B3 + tensor_hash(t) + k) & _M64
    return h


def shard_check(window_hash, envs_per_rank, rank, world, device):
    """Shard invariance over the real collective, before a timed N-rank run: `window_hash(env_offset)`
    steps a small window of environments whose global ids start at env_offset and returns a 64-bit checksum of the result.
    Every rank hashes the window at the head of ITS shard (env_offset = rank * envs_per_rank); the hashes are all-gathered; rank 0
    recomputes all N windows locally with the matching env_offset and compares.  N ranks reproduce one big batch bit for bit iff
    every pair agrees.  Returns {"status": "ok" | "fail", ...} on rank 0, {"status": "n/a"} elsewhere."""
    mine = window_hash(shard_offset(envs_per_rank, rank))
    gathered = [row[0] for row in gather_i64([mine], device)]
    if rank != 0:
        return {"status": "n/a"}
    local = [window_hash(shard_offset(envs_per_rank, r)) for r in range(world)]
    bad = [r for r in range(world) if local[r] != gathered[r]]
    return {"status": "fail" if bad else "ok", "ranks": world, "mismatching_ranks": bad,
            "hashes": [f"{v:016x}" for v in gathered], "distinct_windows": len(set(local))}
