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


def make_cuda(tensors):
    """(tree_tensors, graph_tensors) -> int64 device tensors, host ``scope`` list kept last."""
    tree_tensors, graph_tensors = tensors
    tree_tensors = [make_tensor(x).long() for x in tree_tensors[:-1]] + [tree_tensors[-1]]
    graph_tensors = [make_tensor(x).long() for x in graph_tensors[:-1]] + [graph_tensors[-1]]
    return tree_tensors, graph_tensors
