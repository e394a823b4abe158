"""Multi-GPU plumbing: one process per GPU (``torch.distributed``; backend "nccl" is RCCL on
ROCm, "gloo" in the CPU tests).

The path shards over *image pairs* (SURVEY.md §8e): every pair's forward is independent
(InstanceNorm is per sample, BatchNorm is frozen), so inference partitions the pair list over
the ranks with NO data-path collective.  Training adds exactly one collective per step: a
SUM all-reduce of one flat fp32 gradient buffer (8 337 646 elements = 33.35 MB) over xGMI,
replacing ``nn.DataParallel``'s broadcast / scatter / gather / reduce_add (train_flow.py:96).
SUM, not mean: the reference's loss is a sum over samples (train_flow.py:69), and
``reduce_add_coalesced`` adds the replicas' gradients.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch


def shard_seed(base_seed: int, rank: int) -> int:
    """Distinct synthetic data per rank, reproducible per (seed, rank)."""
    return int(base_seed) + 7919 * int(rank)


def shard_indices(num_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin partition of a pair list: rank r owns items r, r+W, r+2W, ..."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_items, world_size))


def gather_results(local: Sequence[torch.Tensor], num_items: int, rank: int, world_size: int,
                   group=None) -> List[torch.Tensor]:
    """Control-plane helper for evaluation: reassemble per-pair results (e.g. EPE scalars) in
    the original order on every rank.  Not on the data path of the forward."""
    import torch.distributed as dist
    if world_size == 1:
        return list(local)
    gathered: List[list] = [None] * world_size   # type: ignore[list-item]
    dist.all_gather_object(gathered, [t.cpu() for t in local], group=group)
    out: List[torch.Tensor] = [None] * num_items  # type: ignore[list-item]
    for r in range(world_size):
        for t, idx in zip(gathered[r], shard_indices(num_items, r, world_size)):
            out[idx] = t
    return out


class FlatGradAllReduce:
    """One bucket for the whole model: a flat fp32 buffer that aliases nothing; gradients are
    packed, SUM-all-reduced once, and unpacked.  33 MB over 7 xGMI links is ~0.1-0.4 ms, far
    below a training step, so no overlap / bucketing machinery is warranted."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        first = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=first.device)

    @torch.no_grad()
    def __call__(self, group=None) -> torch.Tensor:
        import torch.distributed as dist
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
        return self.flat
