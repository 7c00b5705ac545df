"""The decoder's four score heads and their losses as ONE autograd node.

Reference: ``HierMPNDecoder.get_topo_score / get_cls_score / get_assm_score / enum_attach`` and the loss block of
``HierMPNDecoder.forward`` (ggpm/decoder.py:136-164, 261-301) -- what ``decoder_heads.ScoreHeads`` and
``HierMPNDecoder._losses`` compose op by op out of ~30 autograd nodes (``_Linear``, ``_GatherRows``, ``_SoftmaxCE``, torch's
slices / adds / index_copy / mul / sum).  Every one of those nodes costs the autograd engine 20-50 us of HOST time each way,
and the backward of this block sits on the step's critical path: the atom level's 2.7 ms backward chain cannot be posted
before the heads' and the two tree-side levels' backward have been ISSUED (profiles/r05_vae_gru_phase_times.txt: 1.65 ms of
host time for 0.95 ms of GPU time).  Here the same launches -- the same library calls with the same operands, in the same
order per head -- are issued straight-line from one ``torch.autograd.Function``: no tape inside, parameter gradients
through the deferred queue (``functional._defer_linear`` / ``_defer_gather``) exactly as the op-by-op nodes queue them.

Results: losses, arg-maxes, accuracies and every parameter gradient are the op-by-op path's (same kernels, same operands);
the gradient of the latent vectors is the same three per-head scatters added in a fixed order (autograd adds them in the order
its nodes happen to finish).  ``_dev.HEADS_COMPOSITE = False`` keeps the op-by-op path (the checker of
tests/test_gpu_parity.py::test_heads_composite_equals_the_op_by_op_heads).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as TF

from . import _lib
from . import functional as F_

MAX_POS = 20
RELU = F_.ACT_RELU


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()


def _gather(table: torch.Tensor, idx32: torch.Tensor, width: int, ld_out: int) -> torch.Tensor:
    rows = idx32.numel()
    out = torch.empty(rows, ld_out, dtype=torch.float32, device=table.device)
    _lib.check(_lib.load().ggpm_gather_rows(F_._p(table), F_._ld(table), F_._p(idx32), rows, width, F_._p(out), ld_out, 0, ld_out,
                                            F_._stream()), "gather_rows")
    return out


def _mlp_forward(seq, x, cxt, H: int, L: int):
    """Sequential(Linear(H + L, H), ReLU, Dropout(inactive), Linear(H, n)) on [x | cxt] -> (hidden [rows, Hp], scores [rows, ld])"""
    l1, l2 = seq[0], seq[3]
    rows, Hp = x.shape[0], F_.padded_hidden(H)
    f32 = dict(dtype=torch.float32, device=x.device)
    h = torch.empty(rows, Hp, **f32)
    ldw = l1.weight.stride(0)
    F_.gemm_ksegments(1, rows, H, [x, cxt], [F_._ld(x), F_._ld(cxt)], [l1.weight, l1.weight[:, H:]], [ldw, ldw], [H, L], h, Hp, Hp,
                      bias=l1.bias, act=RELU)
    n = l2.weight.shape[0]
    ld = (n + 3) // 4 * 4
    s = torch.empty(rows, ld, **f32)
    F_.gemm(0, 1, rows, n, H, h, Hp, l2.weight, l2.weight.stride(0), s, ld, ld, bias=l2.bias)
    return h, s


def _mlp_backward(seq, x, cxt, h, ds, H: int, L: int, dx: Optional[torch.Tensor], accumulate: bool):
    """ds [rows, ld] (pad columns zero) -> (dx [rows, H] dense -- written or accumulated --, dcxt [rows, ld of cxt]); queues the
    two Linears' parameter gradients."""
    l1, l2 = seq[0], seq[3]
    rows, Hp = x.shape[0], F_.padded_hidden(H)
    n = l2.weight.shape[0]
    f32 = dict(dtype=torch.float32, device=x.device)
    dh = torch.empty(rows, Hp, **f32)
    F_.gemm(0, 0, rows, H, n, ds, F_._ld(ds), l2.weight, l2.weight.stride(0), dh, Hp, Hp)
    F_._defer_linear(l2.weight, l2.bias, ds, [h], (H,))
    dpre = torch.empty(rows, Hp, **f32)
    _lib.check(_lib.load().ggpm_act_backward(F_._p(dh), F_._p(h), rows, H, Hp, RELU, 0, F_._p(dpre), F_._stream()), "act_backward")
    ldw = l1.weight.stride(0)
    if dx is None:
        dx = torch.empty(rows, H, **f32)
    F_.gemm(0, 0, rows, H, H, dpre, Hp, l1.weight, ldw, dx, H, H, accumulate=accumulate)
    dcxt = torch.empty_like(cxt)
    F_.gemm(0, 0, rows, L, H, dpre, Hp, l1.weight[:, H:], ldw, dcxt, F_._ld(dcxt), cxt.shape[1])
    F_._defer_linear(l1.weight, l1.bias, dpre, [x, cxt], (H, L))
    return dx, dcxt


def _scatter_context(dcxt: torch.Tensor, idx32: torch.Tensor, B: int, L: int, csr=None) -> torch.Tensor:
    """backward of z.index_select(0, idx): the rows of one molecule summed in a fixed order (transposed CSR; ``csr``: the one
    that came with the schedule's upload)"""
    dz = torch.empty(B, L, dtype=torch.float32, device=dcxt.device)
    if csr is None or csr.rows != idx32.numel() or csr.ncols != B:
        csr = F_.csr_from_index(idx32, ncols=B)
    F_._segment_sum_raw(dcxt, csr.T, L, dz)
    return dz


class AssmBlock:
    """candidates of `k` atoms each: rows [base, base + n) of the candidate-atom buffer, embedding ids, child positions and
    the rows of the padded [P * C] score buffer the candidates go to"""

    def __init__(self, k: int, base: int, n: int, icls32, nth, dest):
        self.k, self.base, self.n, self.icls32, self.nth, self.dest = k, base, n, icls32, nth, dest


class _Heads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heads, spec: dict, z, topo_x, cls_x, cand, *params):
        """heads: the ScoreHeads module (topoNN, clsNN, iclsNN, matchNN, W_assm, E_assm); spec: index tensors of the batch;
        `params`: the same parameters once more, so that autograd knows the node depends on them (their gradients go
        through the deferred queue, never through the return value)."""
        lib = _lib.load()
        H, L, B = heads.hidden_size, heads.latent_size, z.shape[0]
        dev = z.device
        f32 = dict(dtype=torch.float32, device=dev)
        Lp = (L + 3) // 4 * 4
        zc = z if z.stride(1) == 1 else z.contiguous()
        saved = {}
        # ---- topology head: BCE over every (step, node) visit
        cxt_t = _gather(zc, spec["topo_idx"], L, Lp)
        h_t, s_t = _mlp_forward(heads.topoNN, topo_x, cxt_t, H, L)
        x_t = s_t[:, 0].contiguous()
        loss_t = torch.empty(1, **f32)
        dx_t = torch.empty_like(x_t)
        _lib.check(lib.ggpm_bce_logits(F_._p(x_t), F_._p(spec["topo_y"]), x_t.numel(), F_._p(loss_t), F_._p(dx_t),
                                       F_._p(torch.empty_like(x_t)), F_._stream()), "bce_logits")
        # ---- motif-class and attachment-class heads: cross entropy over every cluster prediction (the roots first)
        n_c = cls_x.shape[0]
        cxt_c = _gather(zc, spec["cls_idx"], L, Lp)
        h_c, s_c = _mlp_forward(heads.clsNN, cls_x, cxt_c, H, L)
        h_i, s_i = _mlp_forward(heads.iclsNN, cls_x, cxt_c, H, L)

        def ce(s, N, labels, mask=None, mask_row=None):
            M = s.shape[0]
            loss, arg = torch.empty(1, **f32), torch.empty(M, dtype=torch.int32, device=dev)
            d = torch.empty(M, F_._ld(s), **f32)
            if d.shape[1] > N:
                d[:, N:].zero_()
            _lib.check(lib.ggpm_softmax_ce(F_._p(s), F_._ld(s), M, N, F_._p(mask), 0 if mask is None else F_._ld(mask),
                                           F_._p(mask_row), F_._p(labels), F_._p(loss), F_._p(d), d.shape[1], F_._p(arg),
                                           F_._p(torch.empty(M, **f32)), F_._stream()), "softmax_ce")
            return loss, d, arg

        loss_c, d_c, arg_c = ce(s_c, heads.clsNN[3].weight.shape[0], spec["cls_lab"])
        vocab = heads.vocab
        mask = vocab.mask_on(dev) if hasattr(vocab, "mask_on") else vocab.mask.to(dev)
        loss_i, d_i, arg_i = ce(s_i, heads.iclsNN[3].weight.shape[0], spec["icls_lab"], mask, spec["cls_lab"])
        loss = loss_t + (loss_c + loss_i)
        # ---- attachment head: enum_attach over all candidates, W_assm, dot with the latent vector, cross entropy (label 0)
        blocks: List[AssmBlock] = spec["assm_blocks"]
        P, C = spec["n_assm"], spec["max_cls_size"]
        scores = None
        if P > 0:
            Hp, He = F_.padded_hidden(H), heads.embed_size
            l1 = heads.matchNN[0]
            ldw = l1.weight.stride(0)
            buf = torch.zeros(P * C, Hp, **f32)
            keep = []
            for b in blocks:
                rows = cand[b.base:b.base + b.n]
                emb = _gather(heads.E_assm[0].weight, b.icls32, He, F_.padded_hidden(He))
                order = TF.one_hot(b.nth, MAX_POS).to(torch.float32)
                v = torch.empty(b.n, Hp, **f32)
                F_.gemm_ksegments(1, b.n, H, [rows, emb, order], [F_._ld(rows), F_._ld(emb), MAX_POS],
                                  [l1.weight, l1.weight[:, H:], l1.weight[:, H + He:]], [ldw] * 3, [H, He, MAX_POS], v, Hp, Hp,
                                  bias=l1.bias, act=RELU)
                vs = v if b.k == 1 else v.view(-1, b.k, Hp).sum(dim=1)
                buf.index_copy_(0, b.dest, vs)
                keep.append((rows, emb, order, v))
            wa = heads.W_assm
            ldp = F_.padded_hidden(L)
            proj = torch.empty(P * C, ldp, **f32)
            F_.gemm(0, 1, P * C, L, H, buf, Hp, wa.weight, wa.weight.stride(0), proj, ldp, ldp, bias=wa.bias)
            cxt_a = _gather(zc, spec["assm_idx"], L, Lp)
            scores = (proj[:, :L] * cxt_a[:, :L]).sum(dim=-1).view(P, C).contiguous()
            loss_a, d_a, _ = ce(scores, C, spec["assm_lab"])
            loss = loss + loss_a
            saved.update(buf=buf, keep=keep, proj=proj, cxt_a=cxt_a, d_a=d_a)
        acc = F_.head_accuracies(arg_c, spec["cls_lab_raw"], arg_i, spec["icls_lab_raw"], x_t, spec["topo_lab_raw"], scores)
        saved.update(cxt_t=cxt_t, h_t=h_t, dx_t=dx_t, ld_st=s_t.shape[1], cxt_c=cxt_c, h_c=h_c, h_i=h_i, d_c=d_c, d_i=d_i,
                     topo_x=topo_x, cls_x=cls_x, cand=cand)
        ctx.heads, ctx.spec, ctx.saved_ = heads, spec, saved
        ctx.dims = (H, L, B, P, C)
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)
        return loss.reshape(()), acc

    @staticmethod
    def backward(ctx, dloss, _dacc):
        lib = _lib.load()
        heads, spec, S = ctx.heads, ctx.spec, ctx.saved_
        H, L, B, P, C = ctx.dims
        ctx.saved_ = None
        if dloss is None:
            return (None,) * len(ctx.needs_input_grad)
        dev = dloss.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = dloss.reshape(1).to(torch.float32).contiguous()

        def scale(d, N):
            _lib.check(lib.ggpm_scale_rows(F_._p(d), d.shape[1], d.shape[0], N, F_._p(g), F_._stream()), "scale_rows")
            return d

        # ---- topology head
        n_t = S["dx_t"].numel()
        ds_t = torch.zeros(n_t, S["ld_st"], **f32)
        ds_t[:, 0] = S["dx_t"] * g
        dtopo_x, dcxt_t = _mlp_backward(heads.topoNN, S["topo_x"], S["cxt_t"], S["h_t"], ds_t, H, L, None, False)
        dz = _scatter_context(dcxt_t, spec["topo_idx"], B, L, spec.get("idx_csr", {}).get("topo"))
        # ---- class heads (both read the same rows: one input gradient, two scatters into the same context rows)
        dcls_x, dcxt_c = _mlp_backward(heads.clsNN, S["cls_x"], S["cxt_c"], S["h_c"],
                                       scale(S["d_c"], heads.clsNN[3].weight.shape[0]), H, L, None, False)
        dcls_x, dcxt_i = _mlp_backward(heads.iclsNN, S["cls_x"], S["cxt_c"], S["h_i"],
                                       scale(S["d_i"], heads.iclsNN[3].weight.shape[0]), H, L, dcls_x, True)
        dz = dz + _scatter_context(dcxt_c.add_(dcxt_i), spec["cls_idx"], B, L, spec.get("idx_csr", {}).get("cls"))
        # ---- attachment head
        dcand = None
        if P > 0:
            Hp, He = F_.padded_hidden(H), heads.embed_size
            d_a = scale(S["d_a"], C).reshape(P * C, 1)
            proj, cxt_a, buf = S["proj"], S["cxt_a"], S["buf"]
            dproj = torch.zeros_like(proj)
            dproj[:, :L] = d_a * cxt_a[:, :L]
            dcxt_a = torch.zeros_like(cxt_a)
            dcxt_a[:, :L] = d_a * proj[:, :L]
            dz = dz + _scatter_context(dcxt_a, spec["assm_idx"], B, L, spec.get("idx_csr", {}).get("assm"))
            wa = heads.W_assm
            dbuf = torch.empty(P * C, Hp, **f32)
            F_.gemm(0, 0, P * C, H, L, dproj, F_._ld(dproj), wa.weight, wa.weight.stride(0), dbuf, Hp, Hp)
            F_._defer_linear(wa.weight, wa.bias, dproj, [buf], (H,))
            l1 = heads.matchNN[0]
            ldw = l1.weight.stride(0)
            cand = S["cand"]
            dcand = torch.zeros(cand.shape[0], F_._ld(cand), **f32)[:, :cand.shape[1]] if ctx.needs_input_grad[5] else None
            for b, (rows, emb, order, v) in zip(spec["assm_blocks"], S["keep"]):
                dvs = dbuf.index_select(0, b.dest)
                dv = dvs if b.k == 1 else dvs.unsqueeze(1).expand(-1, b.k, -1).reshape(-1, Hp)
                dpre = torch.empty(b.n, Hp, **f32)
                _lib.check(lib.ggpm_act_backward(F_._p(dv), F_._p(v), b.n, H, Hp, RELU, 0, F_._p(dpre), F_._stream()), "act_backward")
                if dcand is not None:
                    drows = dcand[b.base:b.base + b.n]
                    F_.gemm(0, 0, b.n, H, H, dpre, Hp, l1.weight, ldw, drows, F_._ld(drows), drows.shape[1])
                demb = torch.empty_like(emb)
                F_.gemm(0, 0, b.n, He, H, dpre, Hp, l1.weight[:, H:], ldw, demb, F_._ld(demb), emb.shape[1])
                F_._defer_gather(heads.E_assm[0].weight, He, demb, b.icls32)
                F_._defer_linear(l1.weight, l1.bias, dpre, [rows, emb, order], (H, He, MAX_POS))
        n_params = len(ctx.needs_input_grad) - 6
        return (None, None, dz if ctx.needs_input_grad[2] else None, dtopo_x if ctx.needs_input_grad[3] else None,
                dcls_x if ctx.needs_input_grad[4] else None, dcand) + (None,) * n_params


def usable(heads) -> bool:
    """Dropout inactive in every head, parameter gradients deferrable and publishable (the op-by-op nodes' own conditions)."""
    from . import _dev
    if not _dev.HEADS_COMPOSITE or not F_.defer_wgrads_enabled():
        return False
    for seq in (heads.topoNN, heads.clsNN, heads.iclsNN):
        if seq[2].training and seq[2].p > 0:
            return False
    e = heads.E_assm
    if len(e) > 1 and e[1].training and e[1].p > 0:
        return False
    return F_.can_publish(*[p for p in heads_parameters(heads)])


def heads_parameters(heads):
    ps = []
    for m in (heads.topoNN, heads.clsNN, heads.iclsNN, heads.matchNN, heads.W_assm):
        ps += list(m.parameters())
    ps.append(heads.E_assm[0].weight)
    return ps


def heads_losses(heads, spec: dict, z, topo_x, cls_x, cand):
    """-> (topo_loss + cls_loss + icls_loss + assm_loss  [sum, not yet divided by the batch size], accuracies [4])"""
    return _Heads.apply(heads, spec, z, topo_x, cls_x, cand, *heads_parameters(heads))
