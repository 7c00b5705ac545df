"""Encoder half of the reference's HierPropertyVAE (ggpm/property_vae.py:11-62): encoder + latent heads + KL.

The decoder is outside this build's scope.  ``rsample`` restates
ggpm/property_vae.py:26-33; the two [B,H]x[H,latent] products run through the library GEMM, the
[B,latent] elementwise tail is plain torch on the same stream (negligible, listed for parity only).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_
from .encoder import HierMPNEncoder
from .nnutils import make_cuda


def rsample(z_vecs, W_mean: nn.Linear, W_var: nn.Linear, perturb: bool = True, z_width=None):
    """(z, kl) -- reference ggpm/property_vae.py:26-33. ``z_vecs`` may carry zero pad columns."""
    batch_size = z_vecs.size(0)
    H = W_mean.weight.shape[1]
    L = W_mean.weight.shape[0]
    z_mean = F_.linear([z_vecs], [H], W_mean.weight, W_mean.bias, ld_out=L)
    z_log_var = -torch.abs(F_.linear([z_vecs], [H], W_var.weight, W_var.bias, ld_out=L))
    kl_loss = -0.5 * torch.sum(1.0 + z_log_var - z_mean * z_mean - torch.exp(z_log_var)) / batch_size
    if perturb:
        epsilon = torch.randn_like(z_mean)
        z = z_mean + torch.exp(z_log_var / 2) * epsilon
    else:
        z = z_mean
    return z, kl_loss


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
