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

from . import _dev


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin shard of item (batch) indices: rank r takes r, r+W, r+2W, ..."""
    return list(range(rank, n_items, world_size))


class FlatGradSync:
    """Flat gradient buffer + averaged all-reduce for a replicated module.

    ``zero_grad()`` drops the gradients (``p.grad = None``) so that backward ASSIGNS fresh gradient tensors
    instead of launching one accumulate kernel per parameter; ``all_reduce()`` packs them into the flat buffer
    with one multi-tensor copy, issues ONE collective and re-points every ``p.grad`` at its slice of the
    reduced buffer (no unpack copy).  With a single rank nothing is packed at all.

    With ``encoder=`` (a HierMPNEncoder on the whole-encoder C++ drivers) the encoder's parameters come first in
    the flat buffer, in the drivers' slot order, and the backward writes its gradients straight into the buffer (no
    pack copy).  With GGPM_BUCKETED_ALLREDUCE=1 the slice in front of the atom level's parameters (the last slots,
    whose backward is the last and longest part) is additionally all-reduced on the second stream WHILE that part
    still runs.  Off by default: on one rank (RCCL, forced) every collective costs ~0.45 ms of fixed overhead, so two
    collectives were slower than one (6.38 vs 5.98 ms/step) and the multi-GPU balance could not be measured here.
    """

    ALIGN = 64      # floats

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, encoder=None, keep_flat: bool = False):
        import os
        import weakref
        seen, uniq = set(), []
        for p in params:                         # tied embeddings appear once
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.encoder_params, self.early_numel = [], 0
        self.bucketed = os.environ.get("GGPM_BUCKETED_ALLREDUCE", "0") != "0"
        if encoder is not None and _dev.GRAD_SINK:
            from . import fused
            from .rnn import LSTM
            named = dict(encoder.named_parameters())
            order = fused.param_order("LSTM" if isinstance(encoder.graph_encoder.rnn, LSTM) else "GRU")
            enc = [named[k] for k in order if k in named]
            ids = set(id(p) for p in uniq)
            if len(enc) == len(order) and all(id(p) in ids for p in enc) and len(set(id(p) for p in enc)) == len(enc):
                enc_ids = set(id(p) for p in enc)
                uniq = enc + [p for p in uniq if id(p) not in enc_ids]
                self.encoder_params = enc
                first_graph = next(i for i, k in enumerate(order) if k.startswith("graph_encoder."))
                self.early_numel = sum(p.numel() for p in enc[:first_graph])
                encoder._grad_sink = weakref.ref(self)
        self.params = uniq
        # Every parameter starts on a 256-byte boundary of the flat buffer (zero padding in between: Adam leaves zeros at
        # zero, an all-reduce of zeros is zeros).  Packed tightly, the first parameter with an element count that is not a
        # multiple of four -- the decoder's topoNN ends in a Linear(H, 1): a bias of ONE float -- leaves everything behind it
        # 4-byte aligned, and ggpm_amd.optim.FlatAdam makes the module's parameters views of a buffer with this layout: every
        # GEMM that reads such a weight then fails the 16-byte test of the vector-load kernels and runs on the scalar-load
        # fallback (round 5: the whole decoder side of the bench's VAE row did, 17-21 us per product instead of 9-12).
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        total = off
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        if self.encoder_params:
            self.early_numel = self.offsets[first_graph]
        self.encoder_views = self.views[:len(self.encoder_params)]
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # rehearsal switch: run the pack + collective even with one rank (exercises the RCCL path on a 1-GPU box)
        self.force = dist.is_initialized() and os.environ.get("GGPM_FORCE_ALLREDUCE") == "1"
        # keep_flat: gradients live in the flat buffer on ONE rank too (no collective): the encoder backward writes
        # there and ggpm_amd.optim.FlatAdam reads it as the gradient of its single flat parameter
        self.keep_flat = keep_flat
        self._early_work = None
        if self.active():
            # gradients this package forms itself (functional._defer_flush) are written straight into their slice of the
            # buffer: pack() then has nothing to copy for them
            # (not the encoder's: the C++ backward driver OVERWRITES its slices -- a tied embedding's decoder-side contribution
            # must be ADDED behind it, which the fresh-tensor path does)
            enc = set(id(p) for p in self.encoder_params)
            for p, v in zip(self.params, self.views):
                if id(p) not in enc:
                    p._ggpm_grad_view = v
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"             # RCCL averages in the collective; gloo sums, then one division

    # ------------------------------------------------------------------ interface used by fused._HierEncoder.backward
    def active(self) -> bool:
        return self.world_size > 1 or self.force or self.keep_flat

    def collective(self) -> bool:
        return self.world_size > 1 or self.force

    def accepts(self, params) -> bool:
        return (self.active() and len(params) == len(self.encoder_params)
                and all(p is q for p, q in zip(params, self.encoder_params)))

    def wants_early_bucket(self) -> bool:
        return self.bucketed and self.early_numel > 0 and self.collective()

    def _join_writers(self) -> None:
        """Order the CURRENT stream behind every stream of this package that writes gradients (the second stream: weight-
        gradient contractions of the drivers and of the deferred queue; the atom-level stream: the decode loop's backward and
        its stacked contractions), explicitly and HERE -- a collective reads the flat buffer on a stream of its own that is
        ordered behind the current stream only (RCCL: its internal stream waits for an event recorded on the current one;
        gloo: its device-to-host copy likewise).  The end-of-pass callbacks of the backward have made the same joins already;
        this one does not depend on the collective being called from inside the pass that queued them.

        A host-side (CPU) backend additionally gets an IDLE device: the current stream is synchronised on the host before the
        copy is issued.  gloo blocks the host for the whole collective anyway, so this costs nothing, and the device-to-host
        copy then waits for no event at all: with two rank processes on one GPU, each with five streams on shared hardware
        queues, an event wait queued behind another process's packets is the one ingredient of the round-3 stall
        that is still there by construction -- with this it is gone for the rehearsal backend."""
        if not self.flat.is_cuda:
            return
        from . import functional as F_
        cur = torch.cuda.current_stream(self.flat.device)
        for s in F_.writer_streams(self.flat.device):
            if s.cuda_stream != cur.cuda_stream:
                cur.wait_stream(s)
        if not self._avg:
            cur.synchronize()

    def _reduce(self, t: torch.Tensor, async_op: bool):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)

    def reduce_early_bucket(self) -> None:
        """Called on the second stream between the two phases of the encoder backward (the slice's writers are that stream
        and the main stream, which it has just waited for)."""
        if self.flat.is_cuda and not self._avg:
            torch.cuda.current_stream(self.flat.device).synchronize()      # CPU backend: see _join_writers
        self._early_work = self._reduce(self.flat[:self.early_numel], async_op=True)

    # ------------------------------------------------------------------ step interface
    def zero_grad(self) -> None:
        for p in self.params:
            p.grad = None

    def pack(self, start: int = 0) -> None:
        grads, views = [], []
        for p, v in zip(self.params[start:], self.views[start:]):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                grads.append(p.grad)
                views.append(v)
        if views:
            from .functional import mark
            mark("pack: before the copy")
            torch._foreach_copy_(views, grads)
            mark("pack: after the copy")
        for p, v in zip(self.params, self.views):
            p.grad = v
        if views:
            mark("pack: grads re-pointed")

    def all_reduce(self, async_op: bool = False):
        """Sum over ranks then divide by world size (mean of per-rank batch-mean losses)."""
        if not self.active():
            return None
        if not self.collective():         # one rank, keep_flat: only gather what autograd left outside the buffer
            in_place = len(self.encoder_params) > 0 and all(
                p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                for p, v in zip(self.encoder_params, self.encoder_views))
            self.pack(start=len(self.encoder_params) if in_place else 0)
            return None
        self._join_writers()
        early, self._early_work = self._early_work, None
        if early is not None:
            # the encoder backward wrote its gradients into the flat buffer and the front slice is already being
            # reduced: pack what autograd produced for the remaining parameters, reduce the tail, join
            self.pack(start=len(self.encoder_params))
            tail = self.flat[self.early_numel:]
            self._reduce(tail, async_op=False)
            early.wait()
            if not self._avg:
                self.flat.div_(self.world_size)
            return None
        in_place = len(self.encoder_params) > 0 and all(
            p.grad is not None and p.grad.data_ptr() == v.data_ptr()
            for p, v in zip(self.encoder_params, self.encoder_views))
        self.pack(start=len(self.encoder_params) if in_place else 0)
        work = self._reduce(self.flat, async_op=async_op)
        if async_op:
            return work
        if not self._avg:
            self.flat.div_(self.world_size)
        return None

    def finish(self, work) -> None:
        if work is not None:
            work.wait()
            if not self._avg:
                self.flat.div_(self.world_size)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Replicate rank ``src``'s parameters (every rank must start from the same weights)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
