"""ORACLE -- test infrastructure only, never the product path.

CPU restatement (PyTorch, any float dtype) of the reference's hierarchical
message-passing hot path, written functionally over a ``state_dict``-style
parameter mapping so that the very same parameter names/shapes the reference
checkpoints use drive it.  It keeps the reference's *padded*
formulation and op order (gather to [E,K,H], per-slot U_r / W_f, no hoisting),
which makes it both the parity checker and the "port" CPU baseline.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Parity status: PINNED against outputs of the
reference itself (tests/golden/*.npz, produced by tests/golden/make_golden.py
importing /root/reference in the build container).

Every function cites the reference lines it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

from typing import Dict, Mapping, Sequence, Tuple

import torch

Tensor = torch.Tensor
Params = Mapping[str, Tensor]

NUM_BOND_TYPES = 4   # len(MolGraph.BOND_LIST), ggpm/mol_graph.py:14-15
MAX_POS = 20         # MolGraph.MAX_POS, ggpm/mol_graph.py:16


def gather_rows(src: Tensor, index: Tensor) -> Tensor:
    """index_select_ND(source, 0, index) -- ggpm/nnutils.py:65-70."""
    flat = src.index_select(0, index.reshape(-1))
    return flat.reshape(tuple(index.shape) + tuple(src.shape[1:]))


def _affine(p: Params, name: str, x: Tensor) -> Tensor:
    w = p[name + ".weight"]
    b = p.get(name + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def _row0_mask(n: int, like: Tensor) -> Tensor:
    m = torch.ones(n, 1, dtype=like.dtype, device=like.device)
    m[0, 0] = 0
    return m


# ---------------------------------------------------------------- GRU (ggpm/rnn.py:5-59)
def gru_cell(p: Params, pre: str, x: Tensor, h_nei: Tensor) -> Tensor:
    """GRU.GRU -- ggpm/rnn.py:25-39."""
    H = h_nei.shape[-1]
    s = h_nei.sum(dim=1)
    z = torch.sigmoid(_affine(p, pre + "W_z", torch.cat([x, s], dim=1)))
    r = torch.sigmoid(_affine(p, pre + "W_r", x).view(-1, 1, H) + _affine(p, pre + "U_r", h_nei))
    g = (r * h_nei).sum(dim=1)
    cand = torch.tanh(_affine(p, pre + "W_h", torch.cat([x, g], dim=1)))
    return (1.0 - z) * s + z * cand


def gru_forward(p: Params, pre: str, fmess: Tensor, bgraph: Tensor, depth: int,
                trace: list | None = None) -> Tensor:
    """GRU.forward -- ggpm/rnn.py:41-50."""
    H = p[pre + "U_r.weight"].shape[0]
    h = torch.zeros(fmess.shape[0], H, dtype=fmess.dtype, device=fmess.device)
    mask = _row0_mask(h.shape[0], h)
    for _ in range(depth):
        h = gru_cell(p, pre, fmess, gather_rows(h, bgraph)) * mask
        if trace is not None:
            trace.append(h)
    return h


# ---------------------------------------------------------------- LSTM (ggpm/rnn.py:61-121)
def lstm_cell(p: Params, pre: str, x: Tensor, h_nei: Tensor, c_nei: Tensor) -> Tuple[Tensor, Tensor]:
    """LSTM.LSTM -- ggpm/rnn.py:85-94 (each gate is Sequential(Linear, act): keys '<gate>.0.*')."""
    s = h_nei.sum(dim=1)
    xs = torch.cat([x, s], dim=-1)
    i = torch.sigmoid(_affine(p, pre + "W_i.0", xs))
    o = torch.sigmoid(_affine(p, pre + "W_o.0", xs))
    x_rep = x.unsqueeze(1).expand(-1, h_nei.shape[1], -1)
    f = torch.sigmoid(_affine(p, pre + "W_f.0", torch.cat([x_rep, h_nei], dim=-1)))
    u = torch.tanh(_affine(p, pre + "W.0", xs))
    c = i * u + (f * c_nei).sum(dim=1)
    h = o * torch.tanh(c)
    return h, c


def lstm_forward(p: Params, pre: str, fmess: Tensor, bgraph: Tensor, depth: int,
                 trace: list | None = None) -> Tuple[Tensor, Tensor]:
    """LSTM.forward -- ggpm/rnn.py:96-108."""
    H = p[pre + "W_i.0.weight"].shape[0]
    h = torch.zeros(fmess.shape[0], H, dtype=fmess.dtype, device=fmess.device)
    c = torch.zeros_like(h)
    mask = _row0_mask(h.shape[0], h)
    for _ in range(depth):
        h, c = lstm_cell(p, pre, fmess, gather_rows(h, bgraph), gather_rows(c, bgraph))
        h = h * mask
        c = c * mask
        if trace is not None:
            trace.append(h)
    return h, c


def rnn_forward(p: Params, pre: str, rnn_type: str, fmess: Tensor, bgraph: Tensor, depth: int,
                trace: list | None = None) -> Tensor:
    """rnn(...) followed by get_hidden_state -- ggpm/encoder.py:29-30, rnn.py:22-23,82-83."""
    if rnn_type == "GRU":
        return gru_forward(p, pre, fmess, bgraph, depth, trace)
    if rnn_type == "LSTM":
        return lstm_forward(p, pre, fmess, bgraph, depth, trace)[0]
    raise ValueError("unsupported rnn cell type " + rnn_type)


# ---------------------------------------------------------------- incremental form (decoder side)
def index_scatter(sub: Tensor, full: Tensor, index: Tensor) -> Tensor:
    """index_scatter -- ggpm/nnutils.py:124-128: rows `index` of `full` replaced by `sub`."""
    mask = torch.ones(full.shape[0], dtype=full.dtype, device=full.device)
    mask[index] = 0
    buf = torch.zeros_like(full)
    buf = buf.index_copy(0, index, sub)
    return full * mask.unsqueeze(-1) + buf


def gru_sparse_forward(p: Params, pre: str, h: Tensor, fmess: Tensor, submess: Tensor, bgraph: Tensor,
                       depth: int) -> Tensor:
    """GRU.sparse_forward -- ggpm/rnn.py:52-59."""
    mask = torch.ones(h.shape[0], dtype=h.dtype, device=h.device)
    mask[submess] = 0
    h = h * mask.unsqueeze(1)
    for _ in range(depth):
        sub_h = gru_cell(p, pre, fmess, gather_rows(h, bgraph))
        h = index_scatter(sub_h, h, submess)
    return h


def lstm_sparse_forward(p: Params, pre: str, h: Tensor, c: Tensor, fmess: Tensor, submess: Tensor, bgraph: Tensor,
                        depth: int) -> Tuple[Tensor, Tensor]:
    """LSTM.sparse_forward -- ggpm/rnn.py:110-121."""
    mask = torch.ones(h.shape[0], dtype=h.dtype, device=h.device)
    mask[submess] = 0
    h = h * mask.unsqueeze(1)
    c = c * mask.unsqueeze(1)
    for _ in range(depth):
        sub_h, sub_c = lstm_cell(p, pre, fmess, gather_rows(h, bgraph), gather_rows(c, bgraph))
        h = index_scatter(sub_h, h, submess)
        c = index_scatter(sub_c, c, submess)
    return h, c


# ---------------------------------------------------------------- MPNEncoder (ggpm/encoder.py:8-38)
def mpn_forward(p: Params, pre: str, rnn_type: str, depth: int, fnode: Tensor, fmess: Tensor,
                agraph: Tensor, bgraph: Tensor, trace: list | None = None) -> Tuple[Tensor, Tensor]:
    """MPNEncoder.forward -- ggpm/encoder.py:28-38 (dropout inactive: eval / p=0)."""
    h = rnn_forward(p, pre + "rnn.", rnn_type, fmess, bgraph, depth, trace)
    nei = gather_rows(h, agraph).sum(dim=1)
    node = torch.relu(_affine(p, pre + "W_o.0", torch.cat([fnode, nei], dim=1)))
    return node * _row0_mask(node.shape[0], node), h


# ---------------------------------------------------------------- HierMPNEncoder (ggpm/encoder.py:41-157)
def _eye(n: int, like: Tensor) -> Tensor:
    return torch.eye(n, dtype=like.dtype, device=like.device)


def embed_graph(p: Params, graph_tensors, atom_size: int, dtype) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """HierMPNEncoder.embed_graph -- ggpm/encoder.py:119-126."""
    fnode, fmess, agraph, bgraph = graph_tensors[:4]
    ref = torch.zeros((), dtype=dtype, device=fnode.device)
    hnode = _eye(atom_size, ref).index_select(0, fnode)
    hmess = torch.cat([hnode.index_select(0, fmess[:, 0]),
                       _eye(NUM_BOND_TYPES, ref).index_select(0, fmess[:, 2]),
                       _eye(MAX_POS, ref).index_select(0, fmess[:, 3])], dim=-1)
    return hnode, hmess, agraph, bgraph


def embed_inter(p: Params, tree_tensors, hatom: Tensor):
    """HierMPNEncoder.embed_inter -- ggpm/encoder.py:96-107."""
    fnode, fmess, agraph, bgraph, cgraph = tree_tensors[:5]
    finput = p["E_i.0.weight"].index_select(0, fnode[:, 1])
    pooled = gather_rows(hatom, cgraph).sum(dim=1)
    hnode = torch.relu(_affine(p, "W_i.0", torch.cat([finput, pooled], dim=-1)))
    hmess = torch.cat([hnode.index_select(0, fmess[:, 0]),
                       _eye(MAX_POS, hnode).index_select(0, fmess[:, 2])], dim=-1)
    return hnode, hmess, agraph, bgraph


def embed_tree(p: Params, tree_tensors, hinter: Tensor):
    """HierMPNEncoder.embed_tree -- ggpm/encoder.py:109-117."""
    fnode, fmess, agraph, bgraph, cgraph = tree_tensors[:5]
    finput = p["E_c.0.weight"].index_select(0, fnode[:, 0])
    hnode = torch.relu(_affine(p, "W_c.0", torch.cat([finput, hinter], dim=-1)))
    hmess = torch.cat([hnode.index_select(0, fmess[:, 0]),
                       _eye(MAX_POS, hnode).index_select(0, fmess[:, 2])], dim=-1)
    return hnode, hmess, agraph, bgraph


def embed_root(p: Params, hmess: Tensor, tree_inputs, roots: Sequence[int]) -> Tensor:
    """HierMPNEncoder.embed_root -- ggpm/encoder.py:128-138 (uses the *pre-MPN* node features)."""
    idx = torch.as_tensor(list(roots), dtype=torch.long, device=hmess.device)
    fnode = tree_inputs[0].index_select(0, idx)
    agraph = tree_inputs[2].index_select(0, idx)
    nei = gather_rows(hmess, agraph).sum(dim=1)
    return torch.tanh(_affine(p, "W_root.0", torch.cat([fnode, nei], dim=1)))


def hier_encoder_forward(p: Params, rnn_type: str, depthT: int, depthG: int, tree_tensors,
                         graph_tensors, atom_size: int = 38, trace: Dict[str, list] | None = None):
    """HierMPNEncoder.forward -- ggpm/encoder.py:140-157.

    ``tree_tensors`` / ``graph_tensors`` are the A0 tuples as int64 tensors with the
    host ``scope`` list last.  Returns (hroot, hnode, hinter, hatom).
    """
    dtype = p["W_root.0.weight"].dtype
    tr = (lambda k: None) if trace is None else (lambda k: trace.setdefault(k, []))
    t = embed_graph(p, graph_tensors, atom_size, dtype)
    hatom, _ = mpn_forward(p, "graph_encoder.", rnn_type, depthG, *t, trace=tr("atom"))
    t = embed_inter(p, tree_tensors, hatom)
    hinter, _ = mpn_forward(p, "inter_encoder.", rnn_type, depthT, *t, trace=tr("inter"))
    t = embed_tree(p, tree_tensors, hinter)
    hnode, hmess = mpn_forward(p, "tree_encoder.", rnn_type, depthT, *t, trace=tr("tree"))
    hroot = embed_root(p, hmess, t, [st for st, _ in tree_tensors[-1]])
    return hroot, hnode, hinter, hatom


# ---------------------------------------------------------------- MotifEncoder (ggpm/encoder.py:252-341)
def motif_encoder_forward(p: Params, rnn_type: str, depthT: int, tree_tensors):
    """MotifEncoder.embed_tree / forward -- ggpm/encoder.py:299-341. Returns (root, node)."""
    fnode, fmess, agraph, bgraph = tree_tensors[:4]
    hnode = p["E_c.0.weight"].index_select(0, fnode[:, 0])
    hatt = p["E_i.0.weight"].index_select(0, fnode[:, 1])
    hmess = torch.cat([hatt.index_select(0, fmess[:, 0]), _eye(MAX_POS, hnode).index_select(0, fmess[:, 2])], dim=-1)
    node, mess = mpn_forward(p, "tree_encoder.", rnn_type, depthT, hnode, hmess, agraph, bgraph)
    root = embed_root(p, mess, (hnode, hmess, agraph, bgraph), [st for st, _ in tree_tensors[-1]])
    return root, node


# ---------------------------------------------------------------- KL (ggpm/property_vae.py:26-33)
def rsample_kl(p: Params, hroot: Tensor, pre_mean: str = "R_mean", pre_var: str = "R_var",
               eps: Tensor | None = None) -> Tuple[Tensor, Tensor]:
    """HierPropertyVAE.rsample -- ggpm/property_vae.py:26-33; eps=None means perturb=False."""
    B = hroot.shape[0]
    z_mean = _affine(p, pre_mean, hroot)
    z_log_var = -torch.abs(_affine(p, pre_var, hroot))
    kl = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / B
    z = z_mean if eps is None else z_mean + torch.exp(z_log_var / 2) * eps
    return z, kl


def to_long_tensors(tensors, device="cpu"):
    """make_cuda -- ggpm/nnutils.py:210-214: every array to int64, host scope list kept last."""
    import numpy as np
    return [torch.as_tensor(np.asarray(x)).long().to(device) for x in tensors[:-1]] + [tensors[-1]]
