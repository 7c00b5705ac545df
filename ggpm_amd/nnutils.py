"""Device-move helpers with the reference's names and meaning (reference ggpm/nnutils.py:9-10,156-158,201-214)."""
from __future__ import annotations

import numpy as np
import torch

is_cuda = torch.cuda.is_available()
device = torch.device("cuda:0") if is_cuda else torch.device("cpu")   # the reference's import-time choice


def current_device() -> torch.device:
    """The reference pins ``cuda:0`` at import (ggpm/nnutils.py:9-10); with one process per GPU the target is
    whatever ``torch.cuda.set_device(local_rank)`` selected, which on a single GPU is the same ``cuda:0``."""
    return torch.device("cuda", torch.cuda.current_device()) if is_cuda else torch.device("cpu")


def to_cuda(inputs):
    return inputs.to(current_device())


def make_tensor(x):
    """ndarray / list / tensor -> tensor on the device.

    The reference goes ndarray -> .tolist() -> torch.tensor (ggpm/nnutils.py:201-207); torch.from_numpy
    yields the same values without the Python-list round trip.
    """
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else torch.tensor(x)
    return to_cuda(x)


def tree_chain_length(bgraph) -> int:
    """Longest dependency chain among the tree messages, from the padded predecessor table ``tree bgraph`` (host data).

    Message e depends on the messages listed in row e (0 = padding).  On a tree that relation is acyclic, and after
    ``chain`` message-passing steps no message changes any more (chain(e) = 1 + max over its predecessors, 1 without
    any): the encoder then runs ``chain + 1`` of its ``depthT`` steps and replicates the result.  Returns 0 ("unknown",
    run every step) if the table is not acyclic or is not host data."""
    import numpy as np
    if isinstance(bgraph, torch.Tensor):
        if bgraph.is_cuda:
            return 0
        bgraph = bgraph.numpy()
    bg = np.asarray(bgraph)
    if bg.ndim != 2 or bg.shape[0] <= 1:
        return 0
    mask = bg > 0
    has = mask.any(axis=1)
    d = np.zeros(bg.shape[0], dtype=np.int64)
    for _ in range(bg.shape[0] + 1):
        nd = np.where(has, 1 + (d[bg] * mask).max(axis=1), 1)
        nd[0] = 0
        if np.array_equal(nd, d):
            return int(d.max())
        d = nd
    return 0            # never settled: a cycle


def attach_hint(t: torch.Tensor, name: str, value) -> None:
    """Leave a value derived from ``t``'s CONTENTS on the tensor object, stamped with its in-place version counter."""
    setattr(t, name, value)
    setattr(t, name + "_version", t._version)


def read_hint(t, name: str, default=None):
    """The value ``attach_hint`` left, unless the tensor was written in place since (a resident index tensor refilled
    with the next batch must not carry the previous batch's chain length / root ids)."""
    v = getattr(t, name, None)
    if v is None or getattr(t, name + "_version", None) != getattr(t, "_version", None):
        return default
    return v


def make_cuda(tensors):
    """(tree_tensors, graph_tensors) -> int64 device tensors, host ``scope`` list kept last.  While the predecessor
    table is still host data its longest dependency chain is measured and rides along as an attribute of the device
    tensor (``ggpm_chain``), which lets the encoder stop the tree-side levels at their fixed point."""
    tree_tensors, graph_tensors = tensors
    chain = tree_chain_length(tree_tensors[3]) if len(tree_tensors) > 4 else 0
    host = lambda x: isinstance(x, np.ndarray) or (isinstance(x, torch.Tensor) and not x.is_cuda)
    if is_cuda and len(tree_tensors) == 6 and len(graph_tensors) == 5 and all(host(x) for x in list(tree_tensors[:5]) + list(graph_tensors[:4])):
        # a batch as the loader delivers it (host arrays, vae_train.py:78): the nine index arrays packed into ONE int64
        # buffer, one asynchronous copy through the pinned staging ring, device views -- instead of nine blocking
        # pageable copies and as many int32 -> int64 conversion launches
        from . import functional as F_
        from .dataloader import batch_layout, pack_into, unpack_views
        arrays, layout, total = batch_layout(tensors)
        B = len(tree_tensors[-1])
        flat = np.zeros(total + (B + 1) // 2, dtype=np.int64)
        pack_into(flat, arrays, layout)
        flat.view(np.int32)[2 * total:2 * total + B] = [st for st, _ in tree_tensors[-1]]      # root node ids, int32
        dev = F_.upload(flat, current_device())
        tree_tensors, graph_tensors = unpack_views(dev, layout, tree_tensors[-1], graph_tensors[-1])
        if chain:
            attach_hint(tree_tensors[3], "ggpm_chain", chain)
        attach_hint(tree_tensors[0], "ggpm_roots", dev.view(torch.int32)[2 * total:2 * total + B])
        return tree_tensors, graph_tensors
    tree_tensors = [make_tensor(x).long() for x in tree_tensors[:-1]] + [tree_tensors[-1]]
    graph_tensors = [make_tensor(x).long() for x in graph_tensors[:-1]] + [graph_tensors[-1]]
    if chain and isinstance(tree_tensors[3], torch.Tensor):
        attach_hint(tree_tensors[3], "ggpm_chain", chain)
    # the molecules' root node ids (scope starts, what embed_root gathers by) ride along as a device tensor: batches
    # that stay resident then never pay the per-forward upload of that list (0.3 ms of host time per step)
    if isinstance(tree_tensors[0], torch.Tensor) and read_hint(tree_tensors[0], "ggpm_roots") is None and len(tree_tensors) > 4:
        attach_hint(tree_tensors[0], "ggpm_roots", make_tensor(np.asarray([st for st, _ in tree_tensors[-1]], dtype=np.int32)))
    return tree_tensors, graph_tensors
