"""Times ggpm_gemm on the shapes of one configs[1] training step (dev tool): python tools/time_gemm.py

Each line: transposes, M N K, microseconds per call (HIP events around 20 calls), TFLOP/s.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggpm_amd import _lib  # noqa: E402

SHAPES = [  # (trans_a, trans_b, M, N, K, label)
    (1, 0, 300, 300, 56860, "atom level dW_h / dW_z / dU_r (K = E*depth)"),
    (1, 0, 300, 300, 11460, "motif level dW (K = E*depth)"),
    (1, 0, 340, 912, 2843, "atom level dW_x (K = E)"),
    (1, 0, 600, 900, 573, "motif level dW_x (K = E)"),
    (0, 1, 2843, 912, 340, "atom level input projection"),
    (0, 1, 573, 912, 600, "motif level input projection"),
    (0, 0, 2843, 340, 912, "atom level dX"),
    (0, 0, 573, 600, 912, "motif level dX"),
    (0, 1, 1225, 300, 600, "W_o readout"),
    (0, 0, 1225, 600, 300, "W_o readout dX"),
    (0, 1, 310, 300, 600, "tree-node readout (25 tiles of 64 x 64)"),
    (0, 0, 310, 300, 300, "tree-node readout dX"),
    (0, 0, 573, 340, 900, "motif level dx over the gate slabs"),
    (0, 1, 32, 300, 600, "root readout"),
]


def main():
    lib = _lib.load(build_if_missing=False)
    dev = torch.device("cuda")
    s = torch.cuda.current_stream().cuda_stream
    only = os.environ.get("SHAPE")
    for ta, tb, M, N, K, label in (SHAPES if only is None else [SHAPES[int(only)]]):
        make = torch.zeros if os.environ.get("ZEROS") else torch.randn
        pad = lambda n: (n + 15) // 16 * 16 if n in (300, 340, 600, 900) else n     # the stashes are padded to Hp
        A = make((K, pad(M)) if ta else (M, K), device=dev)
        B = make((N, K) if tb else (K, pad(N)), device=dev)
        if ta:
            A = A[:, :M]
        if not tb:
            B = B[:, :N]
        C = torch.empty(M, N, device=dev)
        nbytes = lib.ggpm_gemm_workspace_bytes(M, N, K)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)

        def call():
            rc = lib.ggpm_gemm(ta, tb, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), C.data_ptr(), N, N,
                               None, 0, 0, 0, ws.data_ptr() if nbytes else None, nbytes, s)
            assert rc == 0, rc
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        ref = (A.t() if ta else A).double() @ (B.t() if tb else B).double()
        err = float((C.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
        print("%d%d  %5d %5d %6d  %8.1f us  %6.1f TF  err %.1e  %s" % (ta, tb, M, N, K, us, 2.0 * M * N * K / us / 1e6,
                                                                          err, label), flush=True)


if __name__ == "__main__":
    main()
