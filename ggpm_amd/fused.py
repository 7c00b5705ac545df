"""Whole-encoder autograd node on the C++ drivers of csrc/encoder.hip (ggpm_encoder_forward / _backward).

One ctypes call per direction instead of ~150: at MI355X speeds the op-by-op host path (functional.py) needs about as
long to ENQUEUE a step as the GPU needs to run it.  Same kernels, same order, same results; used by
``HierMPNEncoder.forward_padded`` for both message functions, with or without dropout (counter-based masks generated
in the drivers, ``ggpm_dropout``).  GGPM_FUSED_ENCODER=0 switches it off.
"""
from __future__ import annotations

import ctypes
import os
from typing import List

import torch

from . import _dev, _lib
from . import functional as F_
from .nnutils import read_hint

_HEAD = ["E_c.0.weight", "E_i.0.weight", "W_c.0.weight", "W_c.0.bias", "W_i.0.weight", "W_i.0.bias", "W_root.0.weight",
         "W_root.0.bias"]
_CELL = {"GRU": ("rnn.W_z.weight", "rnn.W_z.bias", "rnn.W_r.weight", "rnn.U_r.weight", "rnn.U_r.bias", "rnn.W_h.weight",
                 "rnn.W_h.bias"),
         "LSTM": ("rnn.W_i.0.weight", "rnn.W_i.0.bias", "rnn.W_o.0.weight", "rnn.W_o.0.bias", "rnn.W.0.weight",
                  "rnn.W.0.bias", "rnn.W_f.0.weight", "rnn.W_f.0.bias")}


def param_order(cell: str):
    """Parameter names in the slot order of include/ggpm_hip.h (ggpm_encoder_forward)."""
    return _HEAD + [lvl + "." + k for lvl in ("tree_encoder", "inter_encoder", "graph_encoder")
                    for k in ("W_o.0.weight", "W_o.0.bias") + _CELL[cell]]


PARAM_ORDER = param_order("GRU")


class EncDims(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("H", "He", "depthT", "depthG", "atom_size", "n_motif", "n_attach", "N1g",
                                            "E1g", "Kg_a", "Kg_b", "N1t", "E1t", "Kt_a", "Kt_b", "Kt_c", "B", "rnn_type",
                                            "tree_chain")] + \
               [("dropout", ctypes.c_float), ("seed_lo", ctypes.c_uint), ("seed_hi", ctypes.c_uint),
                ("gate_dtype", ctypes.c_int)]

GATE_DTYPES = F_.GATE_DTYPES      # "f32" (default: split operands where they pay) | "bf16" | "f32_mfma" | "f32_split" (A/B forms)


def _dropout_seed():
    """64 bits from torch's CPU generator (so ``torch.manual_seed`` makes dropout runs repeatable)."""
    v = torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int64)
    return int(v[0]), int(v[1])


def enabled() -> bool:
    return os.environ.get("GGPM_FUSED_ENCODER", "1") != "0"


def _arena_size(nbytes: int) -> int:
    """Arena sizes follow the batch's node / message counts, so every batch asks for a slightly different size; rounded up
    to 128 MiB the second batch already finds the first one's block in PyTorch's cache instead of paying a
    several-hundred-MB hipMalloc whenever a new maximum shows up (≈1.5 % of the default bench run before)."""
    step = 1 << 27
    return (nbytes + step - 1) // step * step if nbytes > (1 << 24) else nbytes


def _ptr_array(tensors) -> ctypes.Array:
    arr = _lib.array_type(ctypes.c_void_p, len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def _side_ptr(device):
    if not F_.side_stream_enabled():
        return None, ctypes.c_void_p(0)
    side = F_._side_stream(device)
    return side, ctypes.c_void_p(side.cuda_stream)


# True while the caller runs the encoder beside another chain of launches (property_vae: the decoder's atom level): the
# level kernels then take half as many workgroups (ggpm_level_prefer_narrow), forwards and backwards.
NARROW = [False]


class _HierEncoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dims: EncDims, tree_tensors, graph_tensors, roots, grad_sink, *params):
        lib = _lib.load()
        dev = params[0].device
        tfnode, tfmess, tagraph, tbgraph, tcgraph = [t.contiguous() for t in tree_tensors[:5]]
        gfnode, gfmess, gagraph, gbgraph = [t.contiguous() for t in graph_tensors[:4]]
        Hp = F_.padded_hidden(dims.H)
        f32 = dict(dtype=torch.float32, device=dev)
        saved_bytes = int(lib.ggpm_encoder_saved_bytes(ctypes.byref(dims)))
        saved = torch.empty(_arena_size(saved_bytes), dtype=torch.uint8, device=dev)
        out = torch.empty(dims.B + 2 * dims.N1t + dims.N1g, Hp, **f32)          # one allocation, four row ranges
        hroot, hnode, hinter, hatom = out.split([dims.B, dims.N1t, dims.N1t, dims.N1g])
        params = [p if p.is_contiguous() else p.contiguous() for p in params]
        side, side_p = _side_ptr(dev)
        if side is not None:
            saved.record_stream(side)
            roots.record_stream(side)
        P = F_._p
        beside = bool(NARROW[0])
        narrow = beside and _dev.ENC_NARROW in (True, "fwd")
        if narrow:
            lib.ggpm_level_prefer_narrow(1)
        try:
            _lib.check(lib.ggpm_encoder_forward(ctypes.byref(dims), _ptr_array(params), P(tfnode), P(tfmess), P(tagraph),
                                                P(tbgraph), P(tcgraph), P(gfnode), P(gfmess), P(gagraph), P(gbgraph),
                                                P(roots), P(saved), saved_bytes, P(hroot), P(hnode), P(hinter), P(hatom),
                                                F_._stream(), side_p), "encoder_forward")
        finally:
            if narrow:
                lib.ggpm_level_prefer_narrow(0)
        if any(ctx.needs_input_grad):
            # everything the backward reads goes through save_for_backward: autograd then owns the arena and the
            # outputs (no ctx -> output -> grad_fn -> ctx cycle that would keep a 128 MiB-rounded arena alive after a
            # forward without backward) and a second backward raises autograd's own "backward through the graph a
            # second time" error instead of failing on a cleared attribute
            ctx.save_for_backward(saved, roots, hroot, hnode, hinter, hatom, *params)
            ctx.dims, ctx.grad_sink = dims, grad_sink
            ctx.narrow = beside and _dev.ENC_NARROW in (True, "bwd")
        return hroot, hnode, hinter, hatom

    @staticmethod
    def backward(ctx, d_hroot, d_hnode, d_hinter, d_hatom):
        lib = _lib.load()
        F_.mark("bwd: encoder's node reached")
        dims = ctx.dims
        saved, roots, hroot, hnode, hinter, hatom, *params = ctx.saved_tensors
        dev = saved.device
        sink = ctx.grad_sink() if ctx.grad_sink is not None else None
        if sink is not None and not sink.accepts(params):
            sink = None
        # The driver OVERWRITES every gradient element.  If a parameter's .grad already is its slice of the flat
        # buffer, an earlier encoder backward of this step (gradient accumulation, the encoder called twice) has
        # written there: run into a scratch buffer and add it on top, so that nothing is lost.
        accumulate = sink is not None and any(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                                              for p, v in zip(params, sink.encoder_views))
        if accumulate and sink.wants_early_bucket():
            raise RuntimeError("GGPM_BUCKETED_ALLREDUCE=1 starts reducing the encoder's gradients inside its backward and "
                               "cannot accumulate a second encoder backward in the same step")
        if sink is not None and not accumulate:      # data parallel: write straight into the flat all-reduce buffer
            flat, grads = sink.flat, sink.encoder_views
        else:
            flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
            grads: List[torch.Tensor] = []
            off = 0
            for p in params:
                grads.append(flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
        work_bytes = int(lib.ggpm_encoder_work_bytes(ctypes.byref(dims)))
        work = torch.empty(_arena_size(work_bytes), dtype=torch.uint8, device=dev)
        side, side_p = _side_ptr(dev)
        if side is not None:
            for t in (flat, work):
                t.record_stream(side)
        douts = [None if g is None else g.contiguous() for g in (d_hroot, d_hnode, d_hinter, d_hatom)]
        P = F_._p
        parr, garr = _ptr_array(params), _ptr_array(grads)

        def run(phase):
            if getattr(ctx, "narrow", False):
                lib.ggpm_level_prefer_narrow(1)
            try:
                _lib.check(lib.ggpm_encoder_backward(ctypes.byref(dims), parr, garr, P(roots), P(saved), saved.numel(),
                                                     P(hroot), P(hnode), P(hinter), P(hatom), P(douts[0]), P(douts[1]),
                                                     P(douts[2]), P(douts[3]), P(work), work_bytes, phase, F_._stream(),
                                                     side_p), "encoder_backward")
            finally:
                if getattr(ctx, "narrow", False):
                    lib.ggpm_level_prefer_narrow(0)

        if sink is not None and side is not None and sink.wants_early_bucket():
            run(1)                      # everything but the atom level; its gradients complete on the second stream
            with torch.cuda.stream(side):
                sink.reduce_early_bucket()          # all-reduce them while the atom level's depth loop runs
            run(2)
        else:
            run(0)
        if accumulate:
            torch._foreach_add_(list(sink.encoder_views), grads)
            return (None,) * (5 + len(params))
        if sink is not None:
            # hand the gradients over in place: returning the buffer's own views would make autograd clone each one
            for p, v in zip(params, grads):
                if p.grad is None:
                    p.grad = v
                elif p.grad.data_ptr() != v.data_ptr():
                    p.grad.add_(v)
            return (None,) * (5 + len(params))
        return (None, None, None, None, None, *grads)


def hier_encoder(encoder, tree_tensors, graph_tensors, roots):
    """-> (hroot, hnode, hinter, hatom) as [rows, Hp] tensors; ``encoder`` is a HierMPNEncoder with GRU levels."""
    from .rnn import LSTM
    lstm = isinstance(encoder.graph_encoder.rnn, LSTM)
    params = getattr(encoder, "_fused_params", None)
    if params is None or any(p is not q for p, q in zip(params, encoder._fused_check())):
        sd = dict(encoder.named_parameters())
        params = [sd[k] for k in param_order("LSTM" if lstm else "GRU")]
        encoder._fused_params = params
    tf, gf = tree_tensors, graph_tensors
    dims = EncDims(encoder.hidden_size, encoder.embed_size, encoder.tree_encoder.depth, encoder.graph_encoder.depth,
                   encoder.atom_size, encoder.E_c[0].weight.shape[0], encoder.E_i[0].weight.shape[0],
                   gf[0].shape[0], gf[1].shape[0], gf[2].shape[1], gf[3].shape[1],
                   tf[0].shape[0], tf[1].shape[0], tf[2].shape[1], tf[3].shape[1], tf[4].shape[1], roots.numel(), int(lstm),
                   int(read_hint(tf[3], "ggpm_chain", 0)), 0.0, 0, 0,
                   GATE_DTYPES[getattr(encoder, "gate_dtype", None) or os.environ.get("GGPM_GATE_DTYPE", "f32")])
    if encoder.training and encoder.dropout > 0:      # nn.Dropout semantics: active in training mode only
        seed = getattr(encoder, "_dropout_seed", None) or _dropout_seed()      # (tests pin the seed)
        dims.dropout, dims.seed_lo, dims.seed_hi = float(encoder.dropout), seed[0], seed[1]
    return _HierEncoder.apply(dims, tree_tensors, graph_tensors, roots, getattr(encoder, "_grad_sink", None), *params)
