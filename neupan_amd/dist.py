"""Scene-batch sharding across the GPUs of one node (one process per GPU, torch.distributed
over RCCL/xGMI on the GPU box, gloo in the CPU tests).

Scenes are independent planning instances (the reference plans one scene per `neupan`
object and keeps no cross-scene state), so the path shards with NO data-path collective:
rank r plans the contiguous block of scenes `shard_range(total, r, world)` from inputs it
generates or loads itself.  The only exchange is one all-gather of the control outputs
(2*T floats per scene; 80 B at T=10) so that every rank -- or just the consumer -- sees the
whole batch: latency-bound, far below the per-link xGMI bandwidth, so one flat all-gather
is the right shape (no bucketing, no ring-vs-tree tuning).
"""
from __future__ import annotations

import torch


def shard_range(total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of `total` scenes owned by `rank`; blocks differ by at most 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_controls(opt_u: torch.Tensor, dist=None, world: int = 1, equal_shards: bool = False):
    """All-gather per-scene controls (B_local, 2, T) -> (sum B_local, 2, T) in rank order.
    Equal shard sizes use one all_gather_into_tensor (pass equal_shards=True when the caller knows
    that every rank holds the same number of scenes: it saves the size exchange); ragged shards
    are padded to the largest shard and trimmed."""
    if dist is None or not dist.is_initialized():
        return opt_u                     # no process group: the single-process case (a group of ONE rank does gather)
    if equal_shards and dist.get_backend() != "gloo":
        send = opt_u.contiguous()
        out = torch.empty((world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(out, send)
        return out
    b_local = torch.tensor([opt_u.shape[0]], device=opt_u.device, dtype=torch.int64)
    sizes = [torch.zeros_like(b_local) for _ in range(world)]
    dist.all_gather(sizes, b_local)
    sizes = [int(s.item()) for s in sizes]
    bmax = max(sizes)
    send = opt_u.contiguous()
    if send.shape[0] != bmax:
        pad = torch.zeros((bmax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    out = torch.empty((world * bmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
        out = torch.cat(parts, dim=0)
    else:
        dist.all_gather_into_tensor(out, send)
    if all(s == bmax for s in sizes):
        return out
    return torch.cat([out[r * bmax:r * bmax + sizes[r]] for r in range(world)], dim=0)


def plan_sharded(planner, make_inputs, total_scenes: int, dist=None, rank: int = 0, world: int = 1):
    """Plan `total_scenes` scenes across `world` ranks.  `make_inputs(lo, hi)` returns the
    positional arguments of `planner.forward_batch` for scenes [lo, hi); returns the
    all-gathered controls (total_scenes, 2, T) on every rank."""
    lo, hi = shard_range(total_scenes, rank, world)
    out = planner.forward_batch(*make_inputs(lo, hi))
    return gather_controls(out["opt_u"], dist, world)
