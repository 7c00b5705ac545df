"""Molecule-sharded data parallelism: one process per GPU, RCCL all-reduce of gradients over xGMI.

The reference is single-device (ggpm/nnutils.py:9-10); batches shard naturally because molecules in a
batch are disjoint graphs (ggpm/mol_graph.py:247-250) and the losses are means over the batch
(ggpm/property_vae.py:30).  Each rank therefore consumes its own stream of batches with replicated
parameters and the only exchange is ONE sum of the flat gradient per step, before clipping/Adam
(vae_train.py:82-83) so that every rank applies the same update.

All parameter gradients are packed into a single flat fp32 buffer by one multi-tensor copy: the all-reduce
is ONE collective of ~18-20 MB at H=300, and afterwards ``p.grad`` are views into the reduced buffer.
Works with any torch.distributed backend ("nccl" = RCCL on ROCm, "gloo" for the CPU tests).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin shard of item (batch) indices: rank r takes r, r+W, r+2W, ..."""
    return list(range(rank, n_items, world_size))


class FlatGradSync:
    """Flat gradient buffer + averaged all-reduce for a replicated module.

    ``zero_grad()`` drops the gradients (``p.grad = None``) so that backward ASSIGNS fresh gradient tensors
    instead of launching one accumulate kernel per parameter; ``all_reduce()`` packs them into the flat buffer
    with one multi-tensor copy, issues ONE collective and re-points every ``p.grad`` at its slice of the
    reduced buffer (no unpack copy).  With a single rank nothing is packed at all.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        seen, uniq = set(), []
        for p in params:                         # tied embeddings appear once
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal switch: run the pack + collective even with one rank (exercises the RCCL path on a 1-GPU box)
        import os
        self.force = dist.is_initialized() and os.environ.get("GGPM_FORCE_ALLREDUCE") == "1"

    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    def check_views(self) -> None:
        """Kept for API compatibility: gradients are (re)pointed at the flat buffer by all_reduce()."""
        return None

    def pack(self) -> None:
        grads, views = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                grads.append(p.grad)
                views.append(v)
        if views:
            torch._foreach_copy_(views, grads)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, async_op: bool = False):
        """Sum over ranks then divide by world size (mean of per-rank batch-mean losses)."""
        if self.world_size == 1 and not self.force:
            return None
        self.pack()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return work
        self.flat.div_(self.world_size)
        return None

    def finish(self, work) -> None:
        if work is not None:
            work.wait()
            self.flat.div_(self.world_size)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Replicate rank ``src``'s parameters (every rank must start from the same weights)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
