"""Host -> device input path.

The reference moves every batch to the device tensor by tensor, through Python lists
(``make_tensor``: ``ndarray -> .tolist() -> torch.tensor -> .cuda()``, ggpm/nnutils.py:201-214; nine tensors per
batch, each its own pageable copy).  ``DevicePrefetcher`` packs the nine A0 index arrays of a batch into ONE pinned
int64 staging buffer, uploads it with ONE asynchronous copy on its own stream while the previous step still computes,
and hands out int64 device views in the exact ``make_cuda`` layout (``[fnode, fmess, agraph, bgraph, cgraph, scope]``,
``[fnode, fmess, agraph, bgraph, scope]``).  The consumer's stream is ordered behind the copy by an event; the staging
buffers rotate, and a buffer is only rewritten after the copy that read it has completed.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Sequence, Tuple

import numpy as np
import torch


def _as_int64(x) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.int64)


def batch_layout(tensors):
    """(tree_tensors, graph_tensors) -> (the 9 index arrays, [(offset, shape)], total int64 elements)."""
    tree, graph = tensors
    arrays = [x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
              for x in list(tree[:5]) + list(graph[:4])]
    layout, off = [], 0
    for a in arrays:
        layout.append((off, tuple(a.shape)))
        off += (a.size + 1) // 2 * 2            # keep every array 16-byte aligned
    return arrays, layout, off


def pack_into(dst: np.ndarray, arrays, layout) -> None:
    """Write the arrays (any integer dtype) into the flat int64 destination; plain single-threaded numpy copies --
    a torch CPU copy_ of this size fans out over every visible core and gets the process throttled on a CPU quota."""
    for a, (o, _) in zip(arrays, layout):
        dst[o:o + a.size] = a.reshape(-1)


def pack_batch(tensors) -> Tuple[np.ndarray, List[Tuple[int, Tuple[int, ...]]], list, list]:
    """(tree_tensors, graph_tensors) -> (flat int64 array, [(offset, shape)] for the 9 arrays, tree scope, graph scope)."""
    arrays, layout, total = batch_layout(tensors)
    flat = np.zeros(total, dtype=np.int64)
    pack_into(flat, arrays, layout)
    return flat, layout, tensors[0][-1], tensors[1][-1]


def unpack_views(flat: torch.Tensor, layout, tree_scope, graph_scope):
    views = []
    for off, shape in layout:
        n = int(np.prod(shape)) if len(shape) else 1
        views.append(flat[off:off + n].view(shape))
    return views[:5] + [tree_scope], views[5:] + [graph_scope]


class DevicePrefetcher:
    """Iterate over host batches, yielding device-resident ``(tree_tensors, graph_tensors)`` one batch ahead."""

    def __init__(self, batches: Iterable, device=None, depth: int = 2):
        self.batches = batches
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.depth = max(1, depth)
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._staging: List[torch.Tensor] = []          # pinned ring
        self._staging_events: List = []

    def _stage(self, slot: int, total: int) -> torch.Tensor:
        while len(self._staging) <= slot:
            self._staging.append(torch.empty(0, dtype=torch.int64))
            self._staging_events.append(None)
        ev = self._staging_events[slot]
        if ev is not None:
            ev.synchronize()                             # the copy that read this buffer has finished
        buf = self._staging[slot]
        if buf.numel() < total:
            buf = torch.empty(int(total * 1.25) + 64, dtype=torch.int64)
            if self.cuda:
                buf = buf.pin_memory()
            self._staging[slot] = buf
        return buf[:total]

    def _upload(self, slot: int, tensors):
        from .nnutils import attach_hint, tree_chain_length
        arrays, layout, total = batch_layout(tensors)
        tscope, gscope = tensors[0][-1], tensors[1][-1]
        chain = tree_chain_length(tensors[0][3])         # host data here: lets the tree-side levels stop at their fixed point
        B = len(tscope)
        host = self._stage(slot, total + (B + 1) // 2)   # + the molecules' root node ids, int32, behind the 9 arrays
        pack_into(host.numpy(), arrays, layout)          # straight into the (pinned) staging buffer
        host.numpy().view(np.int32)[2 * total:2 * total + B] = [st for st, _ in tscope]

        def views(flat):
            tree, graph = unpack_views(flat, layout, tscope, gscope)
            if chain:
                attach_hint(tree[3], "ggpm_chain", chain)
            attach_hint(tree[0], "ggpm_roots", flat.view(torch.int32)[2 * total:2 * total + B])     # what embed_root gathers by
            return tree, graph

        if not self.cuda:
            return views(host.clone()), None
        with torch.cuda.stream(self.stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._staging_events[slot] = ev
        return views(dev), (ev, dev)

    def __iter__(self) -> Iterator:
        pending = []
        it = iter(self.batches)
        slot = 0
        nslots = self.depth + 1
        exhausted = False
        while True:
            while not exhausted and len(pending) < self.depth:
                try:
                    b = next(it)
                except StopIteration:
                    exhausted = True
                    break
                pending.append(self._upload(slot % nslots, b))
                slot += 1
            if not pending:
                return
            (tree, graph), sync = pending.pop(0)
            if sync is not None:
                ev, dev = sync
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)                       # consumer stream ordered behind the upload
                dev.record_stream(cur)
            yield tree, graph


class ScheduledGraphs(tuple):
    """``graphs`` of a batch tuple with the batch's decode schedule riding along (``.ggpm_schedule``): still the
    ``(tree_batchG, graph_batchG)`` pair for every reader of the reference's batch shape."""
    ggpm_schedule = None


class ScheduleAhead:
    """Iterator wrapper for the training loop (vae_train.py:71, ``for batch in dataset``): yields the SAME batch tuples
    ``(mols, graphs, tensors, orders, homos, lumos)``, with the decoder's integer bookkeeping of batch k+1
    (``DecodeSchedule.from_graphs``: label walk over the networkx nodes + csrc/schedule.hip, which releases the GIL)
    built on a worker thread while step k runs.  ``HierPropertyVAE.forward`` picks the schedule up from ``graphs``.

        for batch in ScheduleAhead(dataset, model):          # the one changed line
            loss, metrics = model(*batch, beta=beta)

    ``depth`` batches are in flight (default 1).  Exceptions of the builder surface at the batch they belong to; a
    batch whose ``graphs`` is None passes through untouched.  Host work only: uploads stay in the step (two copies).
    """

    def __init__(self, batches: Iterable, model, depth: int = 1):
        self.batches, self.depth = batches, max(1, int(depth))
        dec = model.decoder
        self._vocab, self._hints = dec.vocab, dec.schedule_hints()

    def _prepare(self, batch):
        from .decoder import DecodeSchedule
        mols, graphs, tensors, orders = batch[:4]
        if graphs is None or getattr(graphs, "ggpm_schedule", None) is not None:
            return batch
        sch = DecodeSchedule.from_graphs(graphs, tensors, orders, self._vocab, **self._hints)
        g = ScheduledGraphs(graphs)
        g.ggpm_schedule = sch
        return (mols, g, tensors, orders) + tuple(batch[4:])

    def __iter__(self) -> Iterator:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ggpm-schedule")
        try:
            pending = []
            it = iter(self.batches)
            exhausted = False
            while True:
                while not exhausted and len(pending) < self.depth + 1:
                    try:
                        b = next(it)
                    except StopIteration:
                        exhausted = True
                        break
                    pending.append(pool.submit(self._prepare, b))
                if not pending:
                    return
                yield pending.pop(0).result()
        finally:
            pool.shutdown(wait=True, cancel_futures=True)

    def __len__(self):
        return len(self.batches)
