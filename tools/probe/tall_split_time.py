"""The encoder's end-of-backward contraction by itself (dev probe): ggpm_gru_weight_grads_stacked on configs[1]-sized stashes
(20 depth slots x 2848 messages, H = 300) -- the three H x H products of one gemm_tn_tall_split launch + reduce + db_u."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ggpm_amd import _lib, functional as F_

lib = _lib.load()
P = F_._p
H, rows = 300, int(os.environ.get("ROWS", 20 * 2848))
Hp = F_.padded_hidden(H)
arr = [torch.randn(rows, Hp, device="cuda") for _ in range(6)]
dW = [torch.empty(H, H, device="cuda") for _ in range(3)]
dbu = torch.empty(H, device="cuda")
wsb = int(lib.ggpm_weight_grads_stacked_workspace_bytes(H, rows))
ws = torch.empty(wsb // 4 + 64, device="cuda")
s = F_._stream()


def run():
    _lib.check(lib.ggpm_gru_weight_grads_stacked(rows, rows, H, P(arr[0]), P(arr[1]), P(arr[2]), P(arr[3]), P(arr[4]), P(arr[5]),
                                                 P(dW[0]), H, P(dW[1]), H, P(dbu), P(dW[2]), H, P(ws), ws.numel() * 4, s), "stacked")


for _ in range(5):
    run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
n = 50
for _ in range(n):
    run()
ev[1].record()
torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) / n * 1e3
print("%s: colsum + tall split (3 products, K = %d) + reduce: %.1f us per call = %.1f TFLOP/s fp32-equivalent"
      % (os.environ.get("GGPM_LIB_PATH", "default library").split("/")[-1], rows, us, 3 * 2.0 * H * H * rows / us / 1e6))
