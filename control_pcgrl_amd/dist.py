"""Multi-GPU: envs shard trivially (one process per GPU, no data-path collective); the only exchange is the
episodic-return reduction that the reference's logging does at episode end (rl/callbacks.py:91-117), done here
with one small all-reduce per reporting interval (torch.distributed: "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests)."""
import torch
import torch.distributed as dist


def shard_env_range(total_envs, rank, world_size):
    """Contiguous env-id range [lo, hi) owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(base_seed, total_envs, rank, world_size):
    """Global env g always gets seed base_seed + g, whatever the sharding."""
    lo, hi = shard_env_range(total_envs, rank, world_size)
    return [int(base_seed) + g for g in range(lo, hi)]


class EpisodeStatsReducer:
    """Accumulates (sum return, sum length, n episodes, sum final stats) locally -- on device, no sync -- and
    all-reduces the short vector on demand."""

    def __init__(self, n_stats, device):
        self.n_stats = int(n_stats)
        self.acc = torch.zeros(3 + self.n_stats, dtype=torch.float64, device=device)

    def update(self, done, ep_return, ep_len, final_stats):
        """done: bool [N]; the others are the per-env values of the episodes that just finished."""
        d = done.to(torch.float64)
        self.acc[0] += (ep_return.to(torch.float64) * d).sum()
        self.acc[1] += (ep_len.to(torch.float64) * d).sum()
        self.acc[2] += d.sum()
        self.acc[3:] += (final_stats.to(torch.float64) * d[:, None]).sum(0)

    def update_from_env(self, env):
        """Add the episodes a VecPcgrlEnv finished (auto-reset mode) since the last call: one pcgrl_reduce_episodes
        launch (fixed-order sum on the device, accumulators cleared), no host synchronisation."""
        self._buf = env.reduce_episodes(clear=True, out=getattr(self, "_buf", None))
        self.acc += self._buf

    def reduce(self, group=None, device=None):
        """Returns dict of global means; collective over `group` when torch.distributed is initialised.
        `device`: where the all-reduce runs (default: the accumulator's device; "cpu" for a gloo group)."""
        v = self.acc.clone()
        if device is not None:
            v = v.to(device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
        h = v.tolist()  # the one device -> host copy (and synchronisation) of a reporting interval
        n = max(h[2], 1.0)
        return {"episodes": h[2], "mean_return": h[0] / n, "mean_length": h[1] / n,
                "mean_final_stats": [x / n for x in h[3:]]}

    def reset(self):
        self.acc.zero_()
