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


def flatten_parameters(params: Iterable[torch.nn.Parameter]):
    """Move every trainable parameter and its gradient into ONE flat fp32 buffer each and make ``p.data`` /
    ``p.grad`` views of them.  Returns ``(params, flat, grad)``.  One buffer = one all-reduce per step and one
    fused optimizer launch; nothing is packed or unpacked afterwards."""
    plist = [p for p in params if p.requires_grad]
    if not plist:
        raise ValueError("got an empty parameter list")
    dev = plist[0].device
    n = sum(p.numel() for p in plist)
    flat = torch.empty(n, device=dev, dtype=torch.float32)
    grad = torch.zeros(n, device=dev, dtype=torch.float32)
    realias(plist, flat, grad, keep_values=True)
    return plist, flat, grad


def realias(plist: Sequence[torch.nn.Parameter], flat: torch.Tensor, grad: torch.Tensor, keep_values: bool) -> int:
    """(Re-)bind ``p.data`` / ``p.grad`` to their slices of ``flat`` / ``grad``.  With ``keep_values`` the
    current contents of a parameter (and of a gradient that lives elsewhere) are copied in first -- that is what
    repairs the aliasing after ``model.to()`` / ``.cuda()`` / ``zero_grad(set_to_none=True)`` re-bound the tensors.
    Returns how many tensors had to be re-bound."""
    fixed, o, esz = 0, 0, flat.element_size()
    with torch.no_grad():
        for p in plist:
            k = p.numel()
            if p.data.data_ptr() != flat.data_ptr() + o * esz or p.data.dtype != flat.dtype:
                if keep_values:
                    flat[o:o + k].copy_(p.data.reshape(-1))
                p.data = flat[o:o + k].view_as(p)
                fixed += 1
            if p.grad is None or p.grad.data_ptr() != grad.data_ptr() + o * esz:
                if p.grad is not None and keep_values:
                    grad[o:o + k].copy_(p.grad.reshape(-1))
                elif p.grad is None:
                    grad[o:o + k].zero_()
                p.grad = grad[o:o + k].view_as(p)
                fixed += 1
            o += k
    return fixed


def _world(group=None) -> int:
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def all_reduce_sum_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """THE collective of the training step: SUM of the flat gradient buffer over the ranks, in place
    (RCCL over xGMI with backend "nccl"; 33 MB, far below a step, so no bucketing / overlap machinery)."""
    if _world(group) > 1:
        import torch.distributed as dist
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def sync_replicas(flat_params: torch.Tensor, buffers: Iterable[torch.Tensor] = (), group=None, src: int = 0) -> None:
    """Make every rank start from rank ``src``'s weights and buffers (BatchNorm running statistics): what
    ``nn.DataParallel`` does implicitly by re-broadcasting the module every step (train_flow.py:96).  Needed
    once -- identical gradients (all-reduce) and identical optimizer state keep the replicas identical."""
    if _world(group) <= 1:
        return
    import torch.distributed as dist
    dist.broadcast(flat_params, src=src, group=group)
    for b in buffers:
        dist.broadcast(b, src=src, group=group)


def replica_checksum(flat_params: torch.Tensor) -> torch.Tensor:
    """[sum, sum |.|] in fp64: cheap fingerprint of a replica's weights."""
    x = flat_params.double()
    return torch.stack([x.sum(), x.abs().sum()])


def assert_replicas_in_sync(flat_params: torch.Tensor, group=None) -> None:
    """Raises if the ranks' weights have diverged (different init without ``sync_replicas``, a rank that
    skipped a step, ...).  One 16-byte all-gather; call it every few hundred steps."""
    world = _world(group)
    if world <= 1:
        return
    import torch.distributed as dist
    mine = replica_checksum(flat_params)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    for r, other in enumerate(every):
        if not torch.equal(other, every[0]):
            raise RuntimeError(f"data-parallel replicas diverged: rank {r} checksum {other.tolist()} "
                               f"!= rank 0 checksum {every[0].tolist()}")
