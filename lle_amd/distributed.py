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
