"""Encoder half of the reference's HierPropertyVAE (ggpm/property_vae.py:11-62): encoder + latent heads + KL.

The decoder is outside this build's scope.  ``rsample`` restates
ggpm/property_vae.py:26-33; the two [B,H]x[H,latent] products run through the library GEMM, the
[B,latent] elementwise tail (|.|, exp, KL sum, reparameterisation) is one HIP launch each way (csrc/losses.hip).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _dev
from . import functional as F_
from .encoder import HierMPNEncoder
from .nnutils import make_cuda


class _RsampleTail(torch.autograd.Function):
    """(mean, pv, eps) -> (z, kl): the elementwise part of rsample in one launch each way (csrc/losses.hip)."""

    @staticmethod
    def forward(ctx, mean, pv, eps):
        from . import _lib
        B, L = mean.shape
        mean, pv = mean.contiguous(), pv.contiguous()
        eps = eps.contiguous() if eps is not None else None
        z = torch.empty_like(mean)
        kl = torch.empty(1, dtype=torch.float32, device=mean.device)
        _lib.check(_lib.load().ggpm_rsample_forward(F_._p(mean), F_._p(pv), F_._p(eps), B, L, F_._p(z), F_._p(kl),
                                                    F_._stream()), "rsample_forward")
        ctx.save_for_backward(mean, pv)
        ctx.eps = eps
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        from . import _lib
        mean, pv = ctx.saved_tensors
        B, L = mean.shape
        dz = dz.contiguous() if dz is not None else None
        dkl = dkl.reshape(1).contiguous() if dkl is not None else None
        dmean, dpv = torch.empty_like(mean), torch.empty_like(pv)
        _lib.check(_lib.load().ggpm_rsample_backward(F_._p(mean), F_._p(pv), F_._p(ctx.eps), F_._p(dz), F_._p(dkl), B, L,
                                                     F_._p(dmean), F_._p(dpv), F_._stream()), "rsample_backward")
        return dmean, dpv, None


class _KLHead(torch.autograd.Function):
    """The whole of rsample as ONE autograd node: both [B,H]x[H,latent] products in one grouped launch, the elementwise
    tail in one launch, and on the way back one launch for d(z_vecs) (two K segments), one grouped launch for the two
    weight gradients (second stream, straight into ``.grad`` like the level functions do)."""

    @staticmethod
    def forward(ctx, z_vecs, Wm, bm, Wv, bv, eps):
        from . import _lib
        B, (L, H) = z_vecs.shape[0], Wm.shape
        dev = z_vecs.device
        mean = torch.empty(B, L, dtype=torch.float32, device=dev)
        pv = torch.empty(B, L, dtype=torch.float32, device=dev)
        F_.gemm_grouped(0, 1, B, L, H, [
            dict(A=z_vecs, lda=F_._ld(z_vecs), B=Wm, ldb=Wm.stride(0), C=mean, ldc=L, n_pad=L, bias=bm),
            dict(A=z_vecs, lda=F_._ld(z_vecs), B=Wv, ldb=Wv.stride(0), C=pv, ldc=L, n_pad=L, bias=bv)])
        z = torch.empty_like(mean)
        kl = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(_lib.load().ggpm_rsample_forward(F_._p(mean), F_._p(pv), F_._p(eps), B, L, F_._p(z), F_._p(kl),
                                                    F_._stream()), "rsample_forward")
        ctx.save_for_backward(z_vecs, Wm, Wv, mean, pv)
        ctx.eps = eps
        ctx.params = (Wm, bm, Wv, bv)           # the Parameter objects themselves (their .grad is assigned)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        from . import _lib
        z_vecs, Wm, Wv, mean, pv = ctx.saved_tensors
        B, (L, H) = z_vecs.shape[0], Wm.shape
        dz = dz.contiguous() if dz is not None else None
        dkl = dkl.reshape(1).contiguous() if dkl is not None else None
        dmean, dpv = torch.empty_like(mean), torch.empty_like(pv)
        _lib.check(_lib.load().ggpm_rsample_backward(F_._p(mean), F_._p(pv), F_._p(ctx.eps), F_._p(dz), F_._p(dkl), B, L,
                                                     F_._p(dmean), F_._p(dpv), F_._stream()), "rsample_backward")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(z_vecs)
            F_.gemm_ksegments(0, B, H, [dmean, dpv], [L, L], [Wm, Wv], [Wm.stride(0), Wv.stride(0)], [L, L], dx,
                              F_._ld(dx), z_vecs.shape[1])

        pWm, pbm, pWv, pbv = ctx.params

        def param_grads():
            dWm, dWv = torch.empty_like(Wm), torch.empty_like(Wv)
            F_.gemm_grouped(1, 0, L, H, B, [
                dict(A=dmean, lda=L, B=z_vecs, ldb=F_._ld(z_vecs), C=dWm, ldc=dWm.stride(0), n_pad=H),
                dict(A=dpv, lda=L, B=z_vecs, ldb=F_._ld(z_vecs), C=dWv, ldc=dWv.stride(0), n_pad=H)])
            return dWm, F_.colsum(dmean, B, L), dWv, F_.colsum(dpv, B, L)

        if F_.side_stream_enabled() and F_.can_publish(*ctx.params) and all(ctx.needs_input_grad[1:5]):
            main = torch.cuda.current_stream()
            side = F_._side_stream(z_vecs.device)
            side.wait_stream(main)
            for t in (dmean, dpv, z_vecs):
                t.record_stream(side)
            with torch.cuda.stream(side):
                for q, g in zip(ctx.params, param_grads()):
                    F_._accumulate_grad(q, g, main)
            F_._join_later(main, side)
            return dx, None, None, None, None, None
        dWm, dbm, dWv, dbv = param_grads()
        return dx, dWm, dbm, dWv, dbv, None


def rsample(z_vecs, W_mean: nn.Linear, W_var: nn.Linear, perturb: bool = True, z_width=None):
    """(z, kl) -- reference ggpm/property_vae.py:26-33. ``z_vecs`` may carry zero pad columns."""
    B, L = z_vecs.shape[0], W_mean.weight.shape[0]
    # the reference draws epsilon with torch's generator too
    eps = torch.randn(B, L, dtype=torch.float32, device=z_vecs.device) if perturb else None
    if W_mean.bias is None or W_var.bias is None:
        H = W_mean.weight.shape[1]
        z_mean = F_.linear([z_vecs], [H], W_mean.weight, W_mean.bias, ld_out=L)
        pre_var = F_.linear([z_vecs], [H], W_var.weight, W_var.bias, ld_out=L)
        return _RsampleTail.apply(z_mean, pre_var, eps)
    return _KLHead.apply(z_vecs, W_mean.weight, W_mean.bias, W_var.weight, W_var.bias, eps)


class HierEncoderVAE(nn.Module):
    """``encoder`` / ``R_mean`` / ``R_var`` exactly as HierPropertyVAE names them (state_dict compatible)."""

    def __init__(self, args):
        super().__init__()
        self.encoder = HierMPNEncoder(args.vocab, args.atom_vocab, args.rnn_type, args.embed_size, args.hidden_size,
                                      args.depthT, args.depthG, args.dropout)
        self.latent_size = args.latent_size
        self.R_mean = nn.Linear(args.hidden_size, args.latent_size)
        self.R_var = nn.Linear(args.hidden_size, args.latent_size)

    def forward(self, tensors, beta=0.0, perturb_z=True, prep=None):
        tree_tensors, graph_tensors = make_cuda(tensors)
        hroot, hnode, hinter, hatom = self.encoder.forward_padded(tree_tensors, graph_tensors, prep)
        z, kl = rsample(hroot, self.R_mean, self.R_var, perturb_z)
        H = self.encoder.hidden_size
        return z, kl, (hroot[:, :H], hnode[:, :H], hinter[:, :H], hatom[:, :H])


class StepMetrics(dict):
    """The metrics dictionary ``HierPropertyVAE.forward`` returns (ggpm/property_vae.py:60-62 builds it with ``.item()``:
    python floats).  Same keys, same values -- read on FIRST ACCESS: the six device scalars are stacked by one kernel when
    the forward ends, their copy into pinned host memory is enqueued right behind it, and a value is taken from there
    when it is asked for, which in ``vae_train.py`` is after ``loss.backward(); optimizer.step()`` (:81-86).  Reading them
    inside the forward, as the reference does, stops the host in the middle of the step: the backward cannot be issued
    while the forward's tail still runs.  The read waits for the copy's event, not for the stream: like the reference's
    loop, the host goes on to the next batch while the GPU still finishes this step's backward and optimizer."""

    _KEYS = ('Loss', 'KL:', 'Word', 'I-Word', 'Topo', 'Assm')

    def __init__(self, values):
        super().__init__()
        vals = [v.detach().reshape(()).float() if isinstance(v, torch.Tensor) else None for v in values]
        self._const = [None if isinstance(v, torch.Tensor) else float(v) for v in values]
        dev = next((v.device for v in vals if v is not None), None)
        self._dev = torch.stack([v if v is not None else torch.zeros((), device=dev) for v in vals]) if dev is not None else None
        self._ready = False
        self._host = self._event = None
        if self._dev is not None and self._dev.is_cuda and _dev.METRICS_ASYNC:
            # the copy to the host is ENQUEUED here, behind the forward; a reader waits for this event only -- not for
            # whatever the stream has been given since (backward, optimizer), so the loop can go on to the next batch
            # while the step's tail still runs, as it does around the reference's .item() calls
            self._host = torch.empty(len(self._KEYS), dtype=torch.float32, pin_memory=True)
            self._host.copy_(self._dev, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()

    def _fill(self):
        if not self._ready:
            if self._event is not None:
                self._event.synchronize()
                host = self._host.tolist()
            else:
                host = self._dev.tolist() if self._dev is not None else [0.0] * len(self._KEYS)
            for k, h, c in zip(self._KEYS, host, self._const):
                dict.__setitem__(self, k, h if c is None else c)
            self._ready = True

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        return iter(self._KEYS)

    def __len__(self):
        return len(self._KEYS)

    def __contains__(self, k):
        return k in self._KEYS

    def keys(self):
        return list(self._KEYS)

    def values(self):
        self._fill()
        return [dict.__getitem__(self, k) for k in self._KEYS]

    def items(self):
        self._fill()
        return [(k, dict.__getitem__(self, k)) for k in self._KEYS]

    def get(self, k, default=None):
        return self[k] if k in self._KEYS else default

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


class HierPropertyVAE(nn.Module):
    """reference ggpm/property_vae.py:11-62 -- encoder, latent heads, teacher-forced decoder; the class
    ``OPVNet.get_model('hier-prop')`` returns (ggpm/opvnet.py:4-9) and ``vae_train.py:78`` calls as
    ``model(*batch, beta=beta)``.  Same constructor argument bag, sub-module names and ``state_dict`` keys.

    ``forward(mols, graphs, tensors, orders, homos, lumos, beta, perturb_z=True)`` returns ``(loss, metrics)`` like the
    reference; ``schedule=`` optionally passes a prepared :class:`ggpm_amd.decoder.DecodeSchedule` (the decoder's
    integer bookkeeping, otherwise derived from ``graphs`` on every call).
    """

    def __init__(self, args):
        super().__init__()
        from .decoder import HierMPNDecoder
        self.encoder = HierMPNEncoder(args.vocab, args.atom_vocab, args.rnn_type, args.embed_size, args.hidden_size,
                                      args.depthT, args.depthG, args.dropout)
        self.decoder = HierMPNDecoder(args.vocab, args.atom_vocab, args.rnn_type, args.embed_size, args.hidden_size,
                                      args.latent_size, args.diterT, args.diterG, args.dropout)
        if getattr(args, "tie_embedding", False):
            self.encoder.tie_embedding(self.decoder.hmpn)
        self.latent_size = args.latent_size
        self.R_mean = nn.Linear(args.hidden_size, args.latent_size)
        self.R_var = nn.Linear(args.hidden_size, args.latent_size)

    def rsample(self, z_vecs, W_mean, W_var, perturb=True):
        return rsample(z_vecs, W_mean, W_var, perturb)

    def forward(self, mols, graphs, tensors, orders, homos=None, lumos=None, beta=0.0, perturb_z=True, schedule=None):
        if schedule is None:
            schedule = getattr(graphs, "ggpm_schedule", None)       # dataloader.ScheduleAhead: built one batch ahead
        if schedule is None and graphs is not None:
            # the reference's call shape, ``model(*batch, beta=beta)`` (vae_train.py:78): derive the decoder's integer
            # bookkeeping HERE, from the batch as it arrives (host arrays: no read-back), so that the atom level can be
            # issued beside the encoder exactly as with a prepared schedule
            from .decoder import DecodeSchedule
            schedule = DecodeSchedule.from_graphs(graphs, tensors, orders, self.decoder.vocab, **self.decoder.schedule_hints())
        tree_tensors, graph_tensors = tensors = make_cuda(tensors)
        F_.mark("fwd: inputs on the device")
        self.decoder.start_atom_level(schedule, tensors)       # independent of the latent vector: issued beside the encoder
        F_.mark("fwd: atom level posted")
        # beside the atom level's chain of small launches the encoder's levels take half as many (twice as large)
        # workgroups: the chain's launches then find free compute units instead of waiting for the encoder's to drain
        from . import fused
        beside = getattr(self.decoder, "_atom_ahead", None) is not None and _dev.ENC_NARROW
        fused.NARROW[0] = beside
        try:
            root_vecs = self.encoder.forward_padded(tree_tensors, graph_tensors)[0]
        finally:
            fused.NARROW[0] = False
        root_vecs, kl_div = rsample(root_vecs, self.R_mean, self.R_var, perturb_z)
        F_.mark("fwd: encoder + rsample issued")
        loss, wacc, iacc, tacc, sacc = self.decoder(mols, (root_vecs, root_vecs, root_vecs), graphs, tensors, orders,
                                                    schedule=schedule)
        loss = loss + beta * kl_div
        F_.mark("fwd: heads and losses issued")
        if os.environ.get("GGPM_LAZY_METRICS", "1") != "0":
            return loss, StepMetrics((loss, kl_div, wacc, iacc, tacc, sacc))
        return loss, {'Loss': loss.item(), 'KL:': kl_div.item(), 'Word': float(wacc), 'I-Word': float(iacc),
                      'Topo': float(tacc), 'Assm': float(sacc)}
