"""Encoder half of the reference's HierPropertyVAE (ggpm/property_vae.py:11-62): encoder + latent heads + KL.

The decoder is outside this build's scope.  ``rsample`` restates
ggpm/property_vae.py:26-33; the two [B,H]x[H,latent] products run through the library GEMM, the
[B,latent] elementwise tail (|.|, exp, KL sum, reparameterisation) is one HIP launch each way (csrc/losses.hip).
"""
from __future__ import annotations

import torch
import torch.nn as nn

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


def rsample(z_vecs, W_mean: nn.Linear, W_var: nn.Linear, perturb: bool = True, z_width=None):
    """(z, kl) -- reference ggpm/property_vae.py:26-33. ``z_vecs`` may carry zero pad columns."""
    H = W_mean.weight.shape[1]
    L = W_mean.weight.shape[0]
    z_mean = F_.linear([z_vecs], [H], W_mean.weight, W_mean.bias, ld_out=L)
    pre_var = F_.linear([z_vecs], [H], W_var.weight, W_var.bias, ld_out=L)
    eps = torch.randn_like(z_mean) if perturb else None        # the reference draws epsilon with torch's generator too
    return _RsampleTail.apply(z_mean, pre_var, eps)


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
