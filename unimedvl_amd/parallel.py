"""Data-parallel sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference has no distributed code at all; samples are independent
(varlen packing has no cross-sample term), so the path shards by samples with a full weight
replica per GPU and NO data-path collective.  The only exchange is the gather of results:
generated token ids (or, on request, the bf16 logits of a step) with one RCCL all-gather over
xGMI (backend "nccl" on ROCm); on CPU (tests) the same code runs over gloo.
"""
from typing import List, Sequence

import torch


def shard_bounds(n_items: int, world_size: int, rank: int):
    """Contiguous split; the first n % world ranks take one extra item."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_list(items: Sequence, world_size: int, rank: int):
    s, e = shard_bounds(len(items), world_size, rank)
    return list(items[s:e])


def balanced_order(lengths: Sequence[int], world_size: int):
    """Length-sorted round-robin assignment (limits EOS-straggler skew): returns, per rank, the
    list of item indices it owns."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    per_rank = [[] for _ in range(world_size)]
    for pos, idx in enumerate(order):
        lap, k = divmod(pos, world_size)
        per_rank[k if lap % 2 == 0 else world_size - 1 - k].append(idx)
    return per_rank


def all_gather_ragged(local: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather tensors whose first dim differs per rank (e.g. [steps_r, B_r] token ids).
    Two collectives: the shapes, then the padded payload."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    shape = torch.tensor(list(local.shape), dtype=torch.int64, device=local.device)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    maxshape = torch.stack(shapes).max(0).values.tolist()
    padded = torch.zeros(maxshape, dtype=local.dtype, device=local.device)
    padded[tuple(slice(0, s) for s in local.shape)] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return [b[tuple(slice(0, int(s)) for s in sh.tolist())] for b, sh in zip(bufs, shapes)]


class DataParallelVQA:
    """Runs `engine_fn(images, prompts) -> List[result]` on this rank's shard and gathers every
    rank's results in the original order.  `engine_fn` is the single-GPU path (e.g. a batched
    Bagel.chat); results must be picklable (strings) or tensors."""

    def __init__(self, engine_fn, group=None):
        import torch.distributed as dist
        self.engine_fn = engine_fn
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, images: Sequence, prompts: Sequence[str], lengths: Sequence[int] = None):
        import torch.distributed as dist
        n = len(prompts)
        if lengths is None:
            owned = list(range(*shard_bounds(n, self.world, self.rank)))
        else:
            owned = balanced_order(lengths, self.world)[self.rank]
        local = self.engine_fn([images[i] for i in owned], [prompts[i] for i in owned]) if owned else []
        if self.world == 1:
            out = [None] * n
            for i, r in zip(owned, local):
                out[i] = r
            return out
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (owned, local), group=self.group)
        out = [None] * n
        for idxs, res in gathered:
            for i, r in zip(idxs, res):
                out[i] = r
        return out
