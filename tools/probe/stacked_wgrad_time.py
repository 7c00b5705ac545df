"""Times ggpm_gru_weight_grads_stacked / ggpm_lstm_weight_grads_stacked in isolation (dev probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ggpm_amd import _lib, functional as F_

lib = _lib.load()
dev = torch.device("cuda:0")
P = F_._p
H = int(os.environ.get("H", "300"))
Hp = F_.padded_hidden(H)
for rows in (38500, 54200):
    rq = rows + rows // 5
    f32 = dict(dtype=torch.float32, device=dev)
    A = [torch.randn(rows, Hp, **f32) for _ in range(3)]
    S, G = torch.randn(rows, Hp, **f32), torch.randn(rows, Hp, **f32)
    DQ, Hs = torch.randn(rq, Hp, **f32), torch.randn(rq, Hp, **f32)
    out = [torch.empty(H, H, **f32) for _ in range(4)]
    dbu = torch.empty(H, **f32)
    wsb = int(lib.ggpm_weight_grads_stacked_workspace_bytes(H, max(rows, rq)))
    ws = torch.empty((wsb + 3) // 4, **f32)
    s = F_._stream()

    def gru():
        _lib.check(lib.ggpm_gru_weight_grads_stacked(rows, rq, H, P(A[0]), P(G), P(A[1]), P(S), P(DQ), P(Hs), P(out[0]), H,
                                                     P(out[1]), H, P(dbu), P(out[2]), H, P(ws), ws.numel() * 4, s), "gru")

    def lstm():
        _lib.check(lib.ggpm_lstm_weight_grads_stacked(rows, rq, H, P(A[0]), P(A[1]), P(A[2]), P(S), P(DQ), P(Hs), P(out[0]), H,
                                                      P(out[1]), H, P(out[2]), H, P(out[3]), H, P(ws), ws.numel() * 4, s), "lstm")

    for name, fn, nm, dt in (("GRU (3 products)", gru, 3, 0), ("LSTM (4 products)", lstm, 4, 0),
                             ("GRU bf16 operands", gru, 3, 1), ("LSTM bf16 operands", lstm, 4, 1)):
        lib.ggpm_level_gate_dtype(dt)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        fl = 2.0 * H * H * (rows * (nm - 1) + rq)
        print("rows %d  %-18s %.3f ms  %.1f TFLOP/s" % (rows, name, ms, fl / ms / 1e9))
    lib.ggpm_level_gate_dtype(0)
