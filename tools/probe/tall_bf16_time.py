"""fp32 tall contraction vs the bf16-operand one, single product (dev probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ggpm_amd import _lib, functional as F_

lib = _lib.load()
P = F_._p
for H in (300, 600):
    Hp = F_.padded_hidden(H)
    for K in (54200, 400000 if H == 600 else 160000):
        A = torch.randn(K, Hp, device="cuda")
        B = torch.randn(K, Hp, device="cuda")
        C = torch.empty(H, H, device="cuda")
        wsb = int(lib.ggpm_gemm_workspace_bytes(H, H, K))
        ws = torch.empty(wsb // 4 + 64, device="cuda")
        s = F_._stream()

        def f32():
            _lib.check(lib.ggpm_gemm(1, 0, H, H, K, P(A), Hp, P(B), Hp, P(C), H, H, None, 0, 0, 0, P(ws), ws.numel() * 4, s), "gemm")

        def bf16():
            _lib.check(lib.ggpm_gemm_tn_bf16(H, H, K, P(A), Hp, P(B), Hp, P(C), H, P(ws), ws.numel() * 4, s), "bf16")

        for name, fn in (("fp32 MFMA", f32), ("bf16 operands", bf16)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            print("H=%d K=%d %-14s %.3f ms  %.1f TFLOP/s  %.2f TB/s of operands" % (H, K, name, ms, 2.0 * H * H * K / ms / 1e9, 2.0 * K * Hp * 4 / ms / 1e9))
