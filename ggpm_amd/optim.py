"""Adam over one flat view of all parameters.

``vae_train.py:60`` uses ``torch.optim.Adam(model.parameters())``.  The update is elementwise, so running it on ONE
flat fp32 tensor that all parameters are views of gives the same numbers with one optimizer "parameter": one fused
launch and none of the per-parameter Python bookkeeping of a 39-tensor parameter list (~0.2 ms of host time per step,
which matters once the encoder step itself is ~2.5 ms and host bound).  Gradients come from
``ggpm_amd.parallel.FlatGradSync(keep_flat=True)``, whose flat buffer has the same layout: the C++ encoder backward
writes into it directly, so nothing is packed or copied for the optimizer either.
"""
from __future__ import annotations

import os

import torch

from . import _dev


class FlatAdam:
    def __init__(self, sync, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.sync = sync
        params = sync.params
        # the gradient buffer's layout: every parameter on a 256-byte boundary (parallel.FlatGradSync says why), zeros between
        flat = torch.zeros(sync.flat.numel(), dtype=params[0].dtype, device=params[0].device)
        with torch.no_grad():
            for p, off in zip(params, sync.offsets):
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view_as(p)          # the module's parameters are views of the flat buffer now
        self.flat = torch.nn.Parameter(flat)
        self.opt = torch.optim.Adam([self.flat], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=flat.is_cuda)
        # on the GPU the step is ONE launch of ggpm_adam_step over the flat buffer (torch's fused Adam issues three
        # multi-tensor launches for it); learning-rate schedulers keep working through ``param_groups``
        self._hip = flat.is_cuda and _dev.HIP_ADAM
        if self._hip:
            self._m, self._v, self._t = torch.zeros_like(flat), torch.zeros_like(flat), 0

    @property
    def param_groups(self):
        return self.opt.param_groups

    def step(self) -> None:
        """Call after ``sync.all_reduce()`` (which also gathers stray gradients into the flat buffer on one rank)."""
        if self._hip:
            from . import _lib
            from . import functional as F_
            g, grp = self.sync.flat, self.opt.param_groups[0]
            self._t += 1
            _lib.check(_lib.load().ggpm_adam_step(F_._p(self.flat.data), F_._p(g), F_._p(self._m), F_._p(self._v),
                                                  self.flat.numel(), float(grp["lr"]), float(grp["betas"][0]),
                                                  float(grp["betas"][1]), float(grp["eps"]), float(grp["weight_decay"]),
                                                  self._t, F_._stream()), "adam_step")
            return
        self.flat.grad = self.sync.flat
        self.opt.step()

    def zero_grad(self) -> None:
        self.sync.zero_grad()
