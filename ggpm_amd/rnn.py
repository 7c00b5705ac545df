"""GRU / LSTM message functions -- drop-in for reference ggpm/rnn.py (same ctor, parameter names, shapes).

``forward(fmess, bgraph)`` keeps the reference signature: ``fmess`` is the dense message input [E+1, I]
(row stride may be padded) and ``bgraph`` either the reference's zero-padded int64 predecessor table or an
already converted :class:`ggpm_amd.functional.CSR`.  The depth loop runs in the fused HIP kernels
(csrc/mpn_gru.hip, csrc/mpn_lstm.hip); the x-halves of the gate weights are hoisted into one GEMM each.
``module.gate_dtype = "bf16"`` (attribute, default fp32) runs the hidden x hidden products of ``forward`` on bf16
operands with fp32 accumulate (BASELINE configs[4]; not the 1e-4 parity mode).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_


def _as_csr(bgraph, E1):
    if isinstance(bgraph, F_.CSR):
        return bgraph
    return F_.csr_from_padded(bgraph, ncols=E1)


class GRU(nn.Module):
    """reference ggpm/rnn.py:5-59"""

    def __init__(self, input_size, hidden_size, depth):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.depth = depth
        self.W_z = nn.Linear(input_size + hidden_size, hidden_size)
        self.W_r = nn.Linear(input_size, hidden_size, bias=False)
        self.U_r = nn.Linear(hidden_size, hidden_size)
        self.W_h = nn.Linear(input_size + hidden_size, hidden_size)

    def get_init_state(self, fmess, init_state=None):
        h = torch.zeros(len(fmess), self.hidden_size, device=fmess.device)
        return h if init_state is None else torch.cat((h, init_state), dim=0)

    def get_hidden_state(self, h):
        return h

    def forward_padded(self, fmess, bgraph):
        """h_D as a [E+1, Hp] tensor (pad columns zero)."""
        I, H = self.input_size, self.hidden_size
        pred = _as_csr(bgraph, fmess.shape[0])
        return F_.gru_level(fmess, self.W_z.weight, self.W_z.bias, self.W_r.weight, self.U_r.weight,
                            self.U_r.bias, self.W_h.weight, self.W_h.bias, pred, self.depth, I, H,
                            gate_dtype=getattr(self, "gate_dtype", None))

    def forward(self, fmess, bgraph):
        return self.forward_padded(fmess, bgraph)[:, :self.hidden_size]

    def sparse_forward(self, h, fmess, submess, bgraph):
        """reference ggpm/rnn.py:52-59: recompute rows ``submess`` of the state ``h`` (``fmess``/``bgraph`` are the
        sub-tensors of those rows, ``bgraph`` holding GLOBAL predecessor ids)."""
        return F_.gru_sparse(h, fmess, submess.long(), bgraph.long(), self.W_z.weight, self.W_z.bias, self.W_r.weight,
                             self.U_r.weight, self.U_r.bias, self.W_h.weight, self.W_h.bias, self.depth,
                             self.input_size, self.hidden_size)


class LSTM(nn.Module):
    """reference ggpm/rnn.py:61-121"""

    def __init__(self, input_size, hidden_size, depth):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.depth = depth
        self.W_i = nn.Sequential(nn.Linear(input_size + hidden_size, hidden_size), nn.Sigmoid())
        self.W_o = nn.Sequential(nn.Linear(input_size + hidden_size, hidden_size), nn.Sigmoid())
        self.W_f = nn.Sequential(nn.Linear(input_size + hidden_size, hidden_size), nn.Sigmoid())
        self.W = nn.Sequential(nn.Linear(input_size + hidden_size, hidden_size), nn.Tanh())

    def get_init_state(self, fmess, init_state=None):
        h = torch.zeros(len(fmess), self.hidden_size, device=fmess.device)
        c = torch.zeros(len(fmess), self.hidden_size, device=fmess.device)
        if init_state is not None:
            h = torch.cat((h, init_state), dim=0)
            c = torch.cat((c, torch.zeros_like(init_state)), dim=0)
        return h, c

    def get_hidden_state(self, h):
        return h[0]

    def forward_padded(self, fmess, bgraph):
        I, H = self.input_size, self.hidden_size
        pred = _as_csr(bgraph, fmess.shape[0])
        i, o, u, f = self.W_i[0], self.W_o[0], self.W[0], self.W_f[0]
        return F_.lstm_level(fmess, i.weight, i.bias, o.weight, o.bias, u.weight, u.bias, f.weight, f.bias,
                             pred, self.depth, I, H, gate_dtype=getattr(self, "gate_dtype", None))

    def forward(self, fmess, bgraph):
        h, c = self.forward_padded(fmess, bgraph)
        return h[:, :self.hidden_size], c[:, :self.hidden_size]

    def sparse_forward(self, h, fmess, submess, bgraph):
        """reference ggpm/rnn.py:110-121: ``h`` is the (h, c) pair; returns the updated pair."""
        h, c = h
        i, o, u, f = self.W_i[0], self.W_o[0], self.W[0], self.W_f[0]
        return F_.lstm_sparse(h, c, fmess, submess.long(), bgraph.long(), i.weight, i.bias, o.weight, o.bias,
                              u.weight, u.bias, f.weight, f.bias, self.depth, self.input_size, self.hidden_size)
