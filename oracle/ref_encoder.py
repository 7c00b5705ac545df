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


# ---------------------------------------------------------------- bf16 gate products (BASELINE configs[4])
# ``gate_dtype`` of the functions below: "f32" = the reference's arithmetic.  "bf16" / "bf16w" restate the SAME cells
# (ggpm/rnn.py:27-36 GRU, :88-91 LSTM) with the operands of every hidden x hidden product rounded to bf16
# (round-to-nearest-even, what v_cvt_pk_bf16_f32 does) and the sum accumulated in the working dtype -- the arithmetic of
# v_mfma_f32_16x16x32_bf16 -- so that the HIP bf16 path can be checked against an oracle that differs from it only by
# summation order, instead of against itself.  Everything else (input halves, gate math, state) stays in the working
# dtype.  Rounding points, per product y = x W^T:
#   forward   y  = rne(x) rne(W)^T
#   backward  dx = rne(dy) rne(W)            (the products of the backward depth kernels)
#             dW = dy^T x                    "bf16":  weight gradients contracted from the unrounded stashes
#             dW = rne(dy)^T rne(x)          "bf16w": bf16 operands in the tall weight-gradient contractions as well
#                                            "bf16s": ... and bf16 STORAGE of the depth loop's arrays (see _Round below)
# The per-slot recurrent products of the reference, U_r(h_nei)[e,k] (rnn.py:31) and the hidden half of W_f([x, h_nei])[e,k]
# (rnn.py:90), are applied once per MESSAGE and gathered, (U_r h)[bgraph[e,k]] -- the same numbers element for
# element, but the gradient then reaches the product summed over a message's uses, which is where the kernels round it.
def rne_bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


class _Bf16Product(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, round_wgrad):
        wr = rne_bf16(w)
        ctx.save_for_backward(x, wr)
        ctx.round_wgrad = round_wgrad
        return rne_bf16(x) @ wr.t()

    @staticmethod
    def backward(ctx, dy):
        x, wr = ctx.saved_tensors
        dyr = rne_bf16(dy)
        dw = (dyr.t() @ rne_bf16(x)) if ctx.round_wgrad else (dy.t() @ x)
        return dyr @ wr, dw, None


def bf16_product(x: Tensor, w: Tensor, gate_dtype: str) -> Tensor:
    return _Bf16Product.apply(x, w, gate_dtype in ("bf16w", "bf16s"))


# "bf16s": bf16 STORAGE on top of "bf16w" (BASELINE configs[4]: "bf16 storage / MFMA with fp32 accumulate").  Every array of
# the depth loop that the kernels keep in bf16 is rounded where the kernels write it -- the value is then the same wherever it
# is read, in the forward and in the backward:
#   state h' and the per-message recurrent product q (Hs, Qs), s = sum_p h_p, the gates z / m (GRU) and i / o / u (LSTM)
#       rounded in the FORWARD (what the kernels stash is what their own epilogue uses, and what the backward differentiates
#       the activations with: _StoredGate);
#   the gradients the backward stores between its launches -- d(pre-activation) of every gate (DZP / DMP, DI / DO / DU), dq
#       (DQ) and dS, dG -- rounded in the BACKWARD, once, where autograd has summed them.
# Kept in fp32 by the kernels and therefore untouched here: the LSTM cell state c and dFC, the reset / forget coefficient
# sums R / F, the gate inputs X and their gradients.
class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rne_bf16(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        return (rne_bf16(dy) if ctx.bwd else dy), None, None


def _stored(t: Tensor, gate_dtype: str, fwd: bool, bwd: bool) -> Tensor:
    return _Round.apply(t, fwd, bwd) if gate_dtype == "bf16s" else t


class _StoredGate(torch.autograd.Function):
    """A gate activation whose OUTPUT is what the kernels stash in bf16: y = rne(act(x)), and the derivative is formed from
    that stored y (y (1 - y) resp. 1 - y^2), as the backward kernels form it -- not from the unrounded activation."""

    @staticmethod
    def forward(ctx, x, tanh):
        y = rne_bf16(torch.tanh(x) if tanh else torch.sigmoid(x))
        ctx.save_for_backward(y)
        ctx.tanh = tanh
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return dy * ((1.0 - y * y) if ctx.tanh else (y * (1.0 - y))), None


def _gate(x: Tensor, gate_dtype: str, tanh: bool = False) -> Tensor:
    if gate_dtype == "bf16s":
        return _StoredGate.apply(x, tanh)
    return torch.tanh(x) if tanh else torch.sigmoid(x)


def gru_cell_bf16(p: Params, pre: str, x: Tensor, h: Tensor, bgraph: Tensor, gate_dtype: str) -> Tensor:
    """GRU.GRU (ggpm/rnn.py:25-39) on the message state ``h`` [E, H] with bf16 gate products (see above)."""
    I = x.shape[1]
    Wz, Wh = p[pre + "W_z.weight"], p[pre + "W_h.weight"]
    st = lambda t, fwd, bwd: _stored(t, gate_dtype, fwd, bwd)
    h_nei = gather_rows(h, bgraph)
    q_nei = gather_rows(st(bf16_product(h, p[pre + "U_r.weight"], gate_dtype) + p[pre + "U_r.bias"], True, True), bgraph)
    s = st(h_nei.sum(dim=1), True, True)
    z = _gate(st(x @ Wz[:, :I].t() + bf16_product(s, Wz[:, I:], gate_dtype) + p[pre + "W_z.bias"], False, True), gate_dtype)
    r = torch.sigmoid(_affine(p, pre + "W_r", x).unsqueeze(1) + q_nei)
    g = st((r * h_nei).sum(dim=1), False, True)      # (forwards g only feeds the product, which rounds it anyway)
    cand = _gate(st(x @ Wh[:, :I].t() + bf16_product(g, Wh[:, I:], gate_dtype) + p[pre + "W_h.bias"], False, True), gate_dtype,
                 tanh=True)
    return st((1.0 - z) * s + z * cand, True, False)


def lstm_cell_bf16(p: Params, pre: str, x: Tensor, h: Tensor, c: Tensor, bgraph: Tensor, gate_dtype: str):
    """LSTM.LSTM (ggpm/rnn.py:85-94) on the message state (h, c) with bf16 gate products (see above)."""
    I = x.shape[1]
    st = lambda t, fwd, bwd: _stored(t, gate_dtype, fwd, bwd)
    h_nei, c_nei = gather_rows(h, bgraph), gather_rows(c, bgraph)
    s = st(h_nei.sum(dim=1), True, True)

    def gate(name):
        W = p[pre + name + ".0.weight"]
        return st(x @ W[:, :I].t() + bf16_product(s, W[:, I:], gate_dtype) + p[pre + name + ".0.bias"], False, True)

    i, o, u = _gate(gate("W_i"), gate_dtype), _gate(gate("W_o"), gate_dtype), _gate(gate("W"), gate_dtype, tanh=True)
    Wf = p[pre + "W_f.0.weight"]
    xf = x @ Wf[:, :I].t() + p[pre + "W_f.0.bias"]
    f = torch.sigmoid(xf.unsqueeze(1) + gather_rows(st(bf16_product(h, Wf[:, I:], gate_dtype), True, True), bgraph))
    c_new = i * u + (f * c_nei).sum(dim=1)
    return st(o * torch.tanh(c_new), True, False), c_new


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
                trace: list | None = None, gate_dtype: str = "f32") -> Tensor:
    """GRU.forward -- ggpm/rnn.py:41-50."""
    H = p[pre + "U_r.weight"].shape[0]
    h = torch.zeros(fmess.shape[0], H, dtype=fmess.dtype, device=fmess.device)
    mask = _row0_mask(h.shape[0], h)
    for _ in range(depth):
        if gate_dtype == "f32":
            h = gru_cell(p, pre, fmess, gather_rows(h, bgraph)) * mask
        else:
            h = gru_cell_bf16(p, pre, fmess, h, bgraph, gate_dtype) * mask
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
                 trace: list | None = None, gate_dtype: str = "f32") -> Tuple[Tensor, Tensor]:
    """LSTM.forward -- ggpm/rnn.py:96-108."""
    H = p[pre + "W_i.0.weight"].shape[0]
    h = torch.zeros(fmess.shape[0], H, dtype=fmess.dtype, device=fmess.device)
    c = torch.zeros_like(h)
    mask = _row0_mask(h.shape[0], h)
    for _ in range(depth):
        if gate_dtype == "f32":
            h, c = lstm_cell(p, pre, fmess, gather_rows(h, bgraph), gather_rows(c, bgraph))
        else:
            h, c = lstm_cell_bf16(p, pre, fmess, h, c, bgraph, gate_dtype)
        h = h * mask
        c = c * mask
        if trace is not None:
            trace.append(h)
    return h, c


def rnn_forward(p: Params, pre: str, rnn_type: str, fmess: Tensor, bgraph: Tensor, depth: int,
                trace: list | None = None, gate_dtype: str = "f32") -> Tensor:
    """rnn(...) followed by get_hidden_state -- ggpm/encoder.py:29-30, rnn.py:22-23,82-83."""
    if gate_dtype not in ("f32", "bf16", "bf16w", "bf16s"):
        raise ValueError("gate_dtype " + gate_dtype)
    if rnn_type == "GRU":
        return gru_forward(p, pre, fmess, bgraph, depth, trace, gate_dtype)
    if rnn_type == "LSTM":
        return lstm_forward(p, pre, fmess, bgraph, depth, trace, gate_dtype)[0]
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
def _drop(x: Tensor, masks, site: str) -> Tensor:
    """nn.Dropout in training mode with an INJECTED mask (already scaled by 1 / (1 - p)); ``masks`` None = inactive."""
    return x if masks is None else x * masks[site].to(x.dtype)


def mpn_forward(p: Params, pre: str, rnn_type: str, depth: int, fnode: Tensor, fmess: Tensor,
                agraph: Tensor, bgraph: Tensor, trace: list | None = None, masks=None,
                gate_dtype: str = "f32") -> Tuple[Tensor, Tensor]:
    """MPNEncoder.forward -- ggpm/encoder.py:28-38; W_o = Linear + ReLU + Dropout (:15-19)."""
    h = rnn_forward(p, pre + "rnn.", rnn_type, fmess, bgraph, depth, trace, gate_dtype)
    nei = gather_rows(h, agraph).sum(dim=1)
    node = _drop(torch.relu(_affine(p, pre + "W_o.0", torch.cat([fnode, nei], dim=1))), masks, pre + "W_o")
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


def embed_inter(p: Params, tree_tensors, hatom: Tensor, masks=None):
    """HierMPNEncoder.embed_inter -- ggpm/encoder.py:96-107 (E_i = Embedding + Dropout, W_i = Linear + ReLU + Dropout)."""
    fnode, fmess, agraph, bgraph, cgraph = tree_tensors[:5]
    finput = _drop(p["E_i.0.weight"].index_select(0, fnode[:, 1]), masks, "E_i")
    pooled = gather_rows(hatom, cgraph).sum(dim=1)
    hnode = _drop(torch.relu(_affine(p, "W_i.0", torch.cat([finput, pooled], dim=-1))), masks, "W_i")
    hmess = torch.cat([hnode.index_select(0, fmess[:, 0]),
                       _eye(MAX_POS, hnode).index_select(0, fmess[:, 2])], dim=-1)
    return hnode, hmess, agraph, bgraph


def embed_tree(p: Params, tree_tensors, hinter: Tensor, masks=None):
    """HierMPNEncoder.embed_tree -- ggpm/encoder.py:109-117 (E_c = Embedding + Dropout, W_c = Linear + ReLU + Dropout)."""
    fnode, fmess, agraph, bgraph, cgraph = tree_tensors[:5]
    finput = _drop(p["E_c.0.weight"].index_select(0, fnode[:, 0]), masks, "E_c")
    hnode = _drop(torch.relu(_affine(p, "W_c.0", torch.cat([finput, hinter], dim=-1))), masks, "W_c")
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
                         graph_tensors, atom_size: int = 38, trace: Dict[str, list] | None = None, masks=None,
                         gate_dtype: str = "f32"):
    """HierMPNEncoder.forward -- ggpm/encoder.py:140-157.

    ``tree_tensors`` / ``graph_tensors`` are the A0 tuples as int64 tensors with the
    host ``scope`` list last.  Returns (hroot, hnode, hinter, hatom).  ``masks``: training-mode dropout with injected
    masks (scaled keep masks) for the seven Dropout modules, keyed "E_i", "E_c", "W_i", "W_c" and
    "<level>_encoder.W_o"; None = dropout inactive.  ``gate_dtype``: "f32", or "bf16" / "bf16w" for the bf16 gate
    products of BASELINE configs[4] (see ``bf16_product``); a dict keyed "graph_encoder." / "inter_encoder." /
    "tree_encoder." gives one per level.
    """
    dtype = p["W_root.0.weight"].dtype
    tr = (lambda k: None) if trace is None else (lambda k: trace.setdefault(k, []))
    gd = (lambda pre: gate_dtype) if isinstance(gate_dtype, str) else (lambda pre: gate_dtype[pre])    # (one per level)
    t = embed_graph(p, graph_tensors, atom_size, dtype)
    hatom, _ = mpn_forward(p, "graph_encoder.", rnn_type, depthG, *t, trace=tr("atom"), masks=masks,
                           gate_dtype=gd("graph_encoder."))
    t = embed_inter(p, tree_tensors, hatom, masks)
    hinter, _ = mpn_forward(p, "inter_encoder.", rnn_type, depthT, *t, trace=tr("inter"), masks=masks,
                            gate_dtype=gd("inter_encoder."))
    t = embed_tree(p, tree_tensors, hinter, masks)
    hnode, hmess = mpn_forward(p, "tree_encoder.", rnn_type, depthT, *t, trace=tr("tree"), masks=masks,
                               gate_dtype=gd("tree_encoder."))
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


# ---------------------------------------------------------------- incremental encoders (ggpm/encoder.py:160-249, 343-394)
class IncState:
    """HTuple -- ggpm/decoder.py:13-16."""

    def __init__(self, node=None, mess=None, vmask=None, emask=None):
        self.node, self.mess, self.vmask, self.emask = node, mess, vmask, emask


def _hidden(rnn_type: str, h):
    return h if rnn_type == "GRU" else h[0]


def rnn_init_state(rnn_type: str, n_mess: int, H: int, like: Tensor, init_state: Tensor | None = None):
    """GRU/LSTM.get_init_state -- ggpm/rnn.py:18-20, 74-80 (root vectors appended as extra message rows)."""
    h = torch.zeros(n_mess, H, dtype=like.dtype, device=like.device)
    c = torch.zeros(n_mess, H, dtype=like.dtype, device=like.device)
    if init_state is not None:
        h = torch.cat([h, init_state], dim=0)
        c = torch.cat([c, torch.zeros_like(init_state)], dim=0)
    return h if rnn_type == "GRU" else (h, c)


def apply_tree_mask(tensors, cur: IncState, prev: IncState):
    """HierMPNDecoder.apply_tree_mask -- ggpm/decoder.py:72-77."""
    fnode, fmess, agraph, bgraph, cgraph, scope = tensors
    return (fnode, fmess, agraph * cur.emask[agraph], bgraph * cur.emask[bgraph], cgraph * prev.vmask[cgraph], scope)


def apply_graph_mask(tensors, hgraph: IncState):
    """HierMPNDecoder.apply_graph_mask -- ggpm/decoder.py:79-83."""
    fnode, fmess, agraph, bgraph, scope = tensors
    return fnode, fmess, agraph * hgraph.emask[agraph], bgraph * hgraph.emask[bgraph], scope


def init_decoder_tensors(tree_tensors, batch: int):
    """agraph/bgraph edits of init_decoder_state -- ggpm/decoder.py:103-118 (root vector i lives in row num_mess+i)."""
    fmess, agraph, bgraph = tree_tensors[1], tree_tensors[2].clone(), tree_tensors[3].clone()
    num_mess = fmess.shape[0]
    for i, (root, _) in enumerate(tree_tensors[-1]):
        agraph[root, -1] = num_mess + i
        for m in range(1, num_mess):
            if int(fmess[m, 0]) == root:
                bgraph[m, -1] = num_mess + i
    return list(tree_tensors[:2]) + [agraph, bgraph] + list(tree_tensors[4:])


def inc_mpn_forward(p: Params, pre: str, rnn_type: str, depth: int, tensors, h, num_nodes: int, subset):
    """IncMPNEncoder.forward -- ggpm/encoder.py:165-179 (no pad-row mask on this path)."""
    fnode, fmess, agraph, bgraph = tensors
    subnode, submess = subset
    if len(submess) > 0:
        if rnn_type == "GRU":
            h = gru_sparse_forward(p, pre + "rnn.", h, fmess, submess, bgraph, depth)
        else:
            h = lstm_sparse_forward(p, pre + "rnn.", h[0], h[1], fmess, submess, bgraph, depth)
    nei = gather_rows(_hidden(rnn_type, h), agraph).sum(dim=1)
    node = torch.relu(_affine(p, pre + "W_o.0", torch.cat([fnode, nei], dim=1)))
    buf = torch.zeros(num_nodes, node.shape[1], dtype=node.dtype, device=node.device)
    return index_scatter(node, buf, subnode), h


def _sub_tensor(tensors, subset):
    """get_sub_tensor -- ggpm/encoder.py:195-206."""
    subnode, submess = subset
    out = [tensors[0].index_select(0, subnode), tensors[1].index_select(0, submess),
           tensors[2].index_select(0, subnode), tensors[3].index_select(0, submess)]
    if len(tensors) == 6:
        out.append(tensors[4].index_select(0, subnode))
    return out


def _sub_messages(hnode: Tensor, subnode: Tensor, fmess: Tensor, num_nodes: int):
    buf = torch.zeros(num_nodes, hnode.shape[1], dtype=hnode.dtype, device=hnode.device)
    buf = index_scatter(hnode, buf, subnode)
    return torch.cat([buf.index_select(0, fmess[:, 0]), _eye(MAX_POS, hnode).index_select(0, fmess[:, 2])], dim=-1)


def embed_sub_tree(p: Params, tree_tensors, hinput: Tensor, subtree, is_inter_layer: bool):
    """IncHierMPNEncoder.embed_sub_tree -- ggpm/encoder.py:208-230."""
    subnode, submess = subtree
    fnode, fmess, agraph, bgraph, cgraph = _sub_tensor(tree_tensors, subtree)
    if is_inter_layer:
        finput = p["E_i.0.weight"].index_select(0, fnode[:, 1])
        pooled = gather_rows(hinput, cgraph).sum(dim=1)
        hnode = torch.relu(_affine(p, "W_i.0", torch.cat([finput, pooled], dim=-1)))
    else:
        finput = p["E_c.0.weight"].index_select(0, fnode[:, 0])
        hnode = torch.relu(_affine(p, "W_c.0", torch.cat([finput, hinput.index_select(0, subnode)], dim=-1)))
    hmess = fmess if len(submess) == 0 else _sub_messages(hnode, subnode, fmess, tree_tensors[0].shape[0])
    return hnode, hmess, agraph, bgraph


def inc_hier_forward(p: Params, rnn_type: str, depthT: int, depthG: int, tree_tensors, inter_tensors, graph_tensors,
                     htree: IncState, hinter: IncState, hgraph: IncState, subtree, subgraph):
    """IncHierMPNEncoder.forward -- ggpm/encoder.py:232-249 (``graph_tensors`` already embedded)."""
    n_tree, n_graph = tree_tensors[0].shape[0], graph_tensors[0].shape[0]
    if len(subgraph[0]) + len(subgraph[1]) > 0:
        sub = _sub_tensor(graph_tensors[:4], subgraph)
        hgraph.node, hgraph.mess = inc_mpn_forward(p, "graph_encoder.", rnn_type, depthG, sub, hgraph.mess, n_graph,
                                                   subgraph)
    if len(subtree[0]) + len(subtree[1]) > 0:
        sub = embed_sub_tree(p, inter_tensors, hgraph.node, subtree, True)
        hinter.node, hinter.mess = inc_mpn_forward(p, "inter_encoder.", rnn_type, depthT, sub, hinter.mess, n_tree,
                                                   subtree)
        sub = embed_sub_tree(p, tree_tensors, hinter.node, subtree, False)
        htree.node, htree.mess = inc_mpn_forward(p, "tree_encoder.", rnn_type, depthT, sub, htree.mess, n_tree, subtree)
    return htree, hinter, hgraph


def inc_tree_forward(p: Params, rnn_type: str, depthT: int, tree_tensors, htree: IncState, subtree):
    """IncEncoder.embed_sub_tree / forward -- ggpm/encoder.py:364-394."""
    if len(subtree[0]) + len(subtree[1]) > 0:
        subnode, submess = subtree
        fnode, fmess, agraph, bgraph, _ = _sub_tensor(tree_tensors, subtree)
        hnode = p["E_c.0.weight"].index_select(0, fnode[:, 0])
        hmess = fmess if len(submess) == 0 else _sub_messages(hnode, subnode, fmess, tree_tensors[0].shape[0])
        htree.node, htree.mess = inc_mpn_forward(p, "tree_encoder.", rnn_type, depthT, (hnode, hmess, agraph, bgraph),
                                                 htree.mess, tree_tensors[0].shape[0], subtree)
    return htree


def inc_teacher_forced(p: Params, kind: str, rnn_type: str, depthT: int, depthG: int, tree_tensors, graph_tensors,
                       init_vecs: Tensor, schedule, atom_size: int = 38):
    """The state handling of the teacher-forced decoder loop around the incremental encoder --
    ggpm/decoder.py:165-222 (hier) / 640-683 (tree-only).  ``schedule`` is a list of steps
    ``(subnode, submess, new_atoms, new_bonds)`` (int64 tensors); returns the vectors the decoder reads after every
    step (``htree.node[xid]``, hidden ``htree.mess[mess_idx]``) and the final states."""
    B, H = init_vecs.shape
    dev = init_vecs.device
    n_mess_t, n_mess_g = tree_tensors[1].shape[0], graph_tensors[1].shape[0]
    inter_tensors = tree_tensors
    dec_tensors = init_decoder_tensors(tree_tensors, B)
    z = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
    htree = IncState(mess=rnn_init_state(rnn_type, n_mess_t, H, init_vecs, init_vecs),
                     emask=torch.cat([z(n_mess_t), torch.ones(B, dtype=torch.long, device=dev)]))
    hinter = IncState(mess=rnn_init_state(rnn_type, n_mess_t, H, init_vecs), emask=z(n_mess_t))
    hgraph = IncState(mess=rnn_init_state(rnn_type, n_mess_g, H, init_vecs), vmask=z(graph_tensors[0].shape[0]),
                      emask=z(n_mess_g))
    if kind == "hier":
        graph_emb = tuple(embed_graph(p, graph_tensors, atom_size, init_vecs.dtype)) + (graph_tensors[-1],)
    topo, cls = [], []
    for subnode, submess, atoms, bonds in schedule:
        hgraph.vmask[atoms] = 1                              # update_graph_mask, ggpm/decoder.py:85-101
        hgraph.emask[bonds] = 1
        htree.emask[submess] = 1
        cur_tree = apply_tree_mask(dec_tensors, htree, hgraph)
        if kind == "hier":
            hinter.emask[submess] = 1
            cur_inter = apply_tree_mask(inter_tensors, hinter, hgraph)
            cur_graph = apply_graph_mask(graph_emb, hgraph)
            htree, hinter, hgraph = inc_hier_forward(p, rnn_type, depthT, depthG, cur_tree, cur_inter, cur_graph,
                                                     htree, hinter, hgraph, (subnode, submess), (atoms, bonds))
        else:
            htree = inc_tree_forward(p, rnn_type, depthT, cur_tree, htree, (subnode, submess))
        topo.append(htree.node.index_select(0, subnode))
        if len(submess) > 0:
            cls.append(_hidden(rnn_type, htree.mess).index_select(0, submess))
    out = {"topo": torch.cat(topo), "cls": torch.cat(cls), "tree_mess": _hidden(rnn_type, htree.mess)}
    if kind == "hier":
        out.update(inter_mess=_hidden(rnn_type, hinter.mess), graph_mess=_hidden(rnn_type, hgraph.mess),
                   graph_node=hgraph.node, inter_node=hinter.node)
    return out, dec_tensors


# ---------------------------------------------------------------- decoder score heads + losses (ggpm/decoder.py:35-69, 136-164, 262-283)
def _head(p: Params, name: str, x: Tensor) -> Tensor:
    """Sequential(Linear, ReLU, Dropout(inactive), Linear) -- ggpm/decoder.py:35-52."""
    return _affine(p, name + ".3", torch.relu(_affine(p, name + ".0", x)))


def vocab_mask(n_motif: int, n_attach: int, owner: Tensor, dtype) -> Tensor:
    """PairVocab.mask -- ggpm/vocab.py:34-41: 0 where attachment `idx` belongs to motif `owner[idx]`, -1000 elsewhere."""
    mask = torch.zeros(n_motif, n_attach, dtype=dtype)
    mask[owner, torch.arange(n_attach)] = 1000.0
    return mask - 1000.0


def score_heads(p: Params, mask: Tensor, src_tree_vecs: Tensor, src_graph_vecs: Tensor, topo_vecs: Tensor,
                topo_idx: Tensor, topo_labels: Tensor, cls_vecs: Tensor, cls_idx: Tensor, cls_labs: Tensor,
                icls_labs: Tensor, assm_vecs: Tensor, assm_idx: Tensor, assm_labels: Tensor, batch_size: int):
    """get_topo_score / get_cls_score / get_assm_score and the loss of HierMPNDecoder.forward --
    ggpm/decoder.py:136-164, 262-283 (attention off, losses with size_average=False)."""
    topo = _head(p, "topoNN", torch.cat([topo_vecs, src_tree_vecs.index_select(0, topo_idx)], dim=-1)).squeeze(-1)
    x = torch.cat([cls_vecs, src_tree_vecs.index_select(0, cls_idx)], dim=-1)
    cls = _head(p, "clsNN", x)
    icls = _head(p, "iclsNN", x) + mask.index_select(0, cls_labs)
    cxt = src_graph_vecs.index_select(0, assm_idx.reshape(-1)).view(assm_idx.shape + (-1,))
    assm = (_affine(p, "W_assm", assm_vecs) * cxt).sum(dim=-1)
    F = torch.nn.functional
    topo_loss = F.binary_cross_entropy_with_logits(topo, topo_labels.to(topo.dtype), reduction="sum")
    cls_loss = F.cross_entropy(cls, cls_labs, reduction="sum") + F.cross_entropy(icls, icls_labs, reduction="sum")
    assm_loss = F.cross_entropy(assm, assm_labels, reduction="sum")
    loss = (topo_loss + cls_loss + assm_loss) / batch_size
    return {"topo": topo, "cls": cls, "icls": icls, "assm": assm, "loss": loss}


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
