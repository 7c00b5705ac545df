"""Decoder score heads and their losses -- drop-in for the score-head part of
``HierMPNDecoder`` (reference ggpm/decoder.py:35-69, 136-164, 262-283).

Same sub-module names and shapes as the reference (``topoNN``, ``clsNN``, ``iclsNN``, ``matchNN``, ``W_assm``), so the
matching slice of a reference decoder ``state_dict`` loads unchanged; same ``get_topo_score`` / ``get_cls_score`` /
``get_assm_score`` signatures (attention off, the reference's default).  The linear layers run on the library's GEMM,
the losses (``reduction='sum'`` as the reference's ``size_average=False``) on ``ggpm_softmax_ce`` / ``ggpm_bce_logits``
with the additive vocabulary mask of ``PairVocab.get_mask`` fused into the cross entropy instead of materialised.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from . import functional as F_

MAX_POS = 20


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, label, mask, mask_row, N):
        F_._need_gpu(logits, label)
        lib = _lib.load()
        M = logits.shape[0]
        dev = logits.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        work = torch.empty(M, dtype=torch.float32, device=dev)
        argmax = torch.empty(M, dtype=torch.int32, device=dev)
        need = ctx.needs_input_grad[0]
        dlog = torch.empty(M, F_._ld(logits), dtype=torch.float32, device=dev) if need else None
        if dlog is not None and dlog.shape[1] > N:
            dlog[:, N:].zero_()
        lab = label.to(torch.int32).contiguous()
        mrow = mask_row.to(torch.int32).contiguous() if mask_row is not None else None
        _lib.check(lib.ggpm_softmax_ce(F_._p(logits), F_._ld(logits), M, N, F_._p(mask),
                                       0 if mask is None else F_._ld(mask), F_._p(mrow), F_._p(lab), F_._p(loss),
                                       F_._p(dlog), 0 if dlog is None else dlog.shape[1], F_._p(argmax), F_._p(work),
                                       F_._stream()), "softmax_ce")
        ctx.dlog, ctx.N = dlog, N
        ctx.mark_non_differentiable(argmax)
        return loss.reshape(()), argmax

    @staticmethod
    def backward(ctx, dloss, _dargmax):
        d, ctx.dlog = ctx.dlog, None
        scale = dloss.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.load().ggpm_scale_rows(F_._p(d), d.shape[1], d.shape[0], ctx.N, F_._p(scale), F_._stream()),
                   "scale_rows")
        return d[:, :ctx.N], None, None, None, None


class _BCELogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        F_._need_gpu(x, y)
        x = x.contiguous()
        y = y.to(torch.float32).contiguous()
        M = x.numel()
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        work = torch.empty(M, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().ggpm_bce_logits(F_._p(x), F_._p(y), M, F_._p(loss), F_._p(dx), F_._p(work), F_._stream()),
                   "bce_logits")
        ctx.dx = dx
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        d, ctx.dx = ctx.dx, None
        return d * dloss, None


def cross_entropy_sum(scores: torch.Tensor, labels: torch.Tensor, n_classes: int = None, mask: torch.Tensor = None,
                      mask_row: torch.Tensor = None):
    """nn.CrossEntropyLoss(size_average=False)(scores [+ mask[mask_row]], labels) -> (loss, argmax per row)."""
    N = scores.shape[1] if n_classes is None else n_classes
    return _SoftmaxCE.apply(scores, labels, mask, mask_row, N)


def bce_with_logits_sum(scores: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """nn.BCEWithLogitsLoss(size_average=False)"""
    return _BCELogits.apply(scores, labels)


def _mlp(seq: nn.Sequential, parts, widths):
    """Sequential(Linear, ReLU, Dropout, Linear) on the concatenation of `parts` (never materialised)."""
    l1, l2 = seq[0], seq[3]
    H = l1.weight.shape[0]
    h = F_.linear(parts, widths, l1.weight, l1.bias, act=F_.ACT_RELU)
    h = seq[2](h)
    n = l2.weight.shape[0]
    return F_.linear([h], [H], l2.weight, l2.bias, ld_out=(n + 3) // 4 * 4)[:, :n]


class ScoreHeads(nn.Module):
    """topoNN / clsNN / iclsNN / matchNN / W_assm of reference ggpm/decoder.py:35-58 with their score functions."""

    def __init__(self, vocab, embed_size, hidden_size, latent_size, dropout):
        super().__init__()
        self.vocab, self.hidden_size, self.latent_size, self.embed_size = vocab, hidden_size, latent_size, embed_size
        n_cls, n_icls = vocab.size()

        def head(n_out):
            return nn.Sequential(nn.Linear(hidden_size + latent_size, hidden_size), nn.ReLU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_size, n_out))
        self.topoNN, self.clsNN, self.iclsNN = head(1), head(n_cls), head(n_icls)
        self.matchNN = nn.Sequential(nn.Linear(hidden_size + embed_size + MAX_POS, hidden_size), nn.ReLU())
        self.W_assm = nn.Linear(hidden_size, latent_size)

    def _context(self, src_vecs, batch_idx):
        """``src_vecs.index_select(0, batch_idx)`` (ggpm/decoder.py:138,146,161) through the library's gather: its
        backward sums the rows of one molecule in a fixed order (transposed CSR), where index_select's backward uses
        float atomics -- every prediction of a molecule adds into the same row of d(latent vector), so with atomics
        the whole encoder's gradient changes in its last bits from run to run."""
        if batch_idx.dtype != torch.int32:
            batch_idx = batch_idx.to(torch.int32)
        idx = batch_idx if batch_idx.dim() == 1 else batch_idx.reshape(-1)    # (a resident 1-D index keeps its CSR memo)
        L = src_vecs.shape[1]
        src = src_vecs if src_vecs.stride(1) == 1 else src_vecs.contiguous()
        return F_.gather_rows(src, idx, F_.csr_from_index(idx, ncols=src.shape[0]), L, (L + 3) // 4 * 4)

    def _parts(self, src_vecs, batch_idx, vecs):
        # (a [rows, H] view of a row-padded buffer goes to the GEMM as it is: its leading dimension is the buffer's)
        v = vecs if vecs.dim() == 2 and vecs.stride(1) == 1 and vecs.stride(0) % 4 == 0 else vecs.contiguous()
        return [v, self._context(src_vecs, batch_idx)], [self.hidden_size, self.latent_size]

    def get_topo_score(self, src_tree_vecs, batch_idx, topo_vecs):
        """reference ggpm/decoder.py:136-141"""
        return _mlp(self.topoNN, *self._parts(src_tree_vecs, batch_idx, topo_vecs)).squeeze(-1)

    def get_cls_score(self, src_tree_vecs, batch_idx, cls_vecs, cls_labs):
        """reference ggpm/decoder.py:143-157; the returned icls scores carry the vocabulary mask like the reference's."""
        parts = self._parts(src_tree_vecs, batch_idx, cls_vecs)
        cls_scores = _mlp(self.clsNN, *parts)
        icls_scores = _mlp(self.iclsNN, *parts)
        if cls_labs is not None:
            icls_scores = icls_scores + self.vocab.get_mask(cls_labs).to(icls_scores.device)
        return cls_scores, icls_scores

    def cls_losses(self, src_tree_vecs, batch_idx, cls_vecs, cls_labs, icls_labs):
        """cls_loss + icls_loss of ggpm/decoder.py:268-271 with the mask fused into the cross entropy."""
        parts = self._parts(src_tree_vecs, batch_idx, cls_vecs)
        cls_scores = _mlp(self.clsNN, *parts)
        icls_scores = _mlp(self.iclsNN, *parts)
        l1, a1 = cross_entropy_sum(cls_scores, cls_labs)
        vocab = self.vocab
        mask = vocab.mask_on(icls_scores.device) if hasattr(vocab, "mask_on") else vocab.mask.to(icls_scores.device)
        l2, a2 = cross_entropy_sum(icls_scores, icls_labs, mask=mask, mask_row=cls_labs)
        return l1 + l2, a1, a2

    def get_assm_score(self, src_graph_vecs, batch_idx, assm_vecs, rows_padded=None):
        """reference ggpm/decoder.py:159-164.  ``rows_padded``: ``assm_vecs`` as a contiguous [P * C, ld >= H] buffer (then
        ``assm_vecs`` itself is only read for its shape [P, C, H]: no slice-and-copy of the padded rows)."""
        shape = assm_vecs.shape
        flat = rows_padded if rows_padded is not None else assm_vecs.reshape(-1, shape[-1]).contiguous()
        proj = F_.linear([flat], [self.hidden_size], self.W_assm.weight, self.W_assm.bias)[:, :self.latent_size]
        cxt = self._context(src_graph_vecs, batch_idx)[:, :self.latent_size]
        return (proj * cxt).sum(dim=-1).view(shape[:-1])
