#!/usr/bin/env python3
"""Host cost of the small things a ctypes launch is made of, on the GPU box (dev probe)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ggpm_amd import functional as F_, _lib
lib = _lib.load()
t = torch.zeros(64, 304, device="cuda")
N = 20000
def bench(name, fn, n=N):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print("%-52s %.2f us" % (name, (time.perf_counter() - t0) / n * 1e6))
bench("F_._p(cuda tensor)", lambda: F_._p(t))
bench("t.data_ptr()", lambda: t.data_ptr())
bench("ctypes.c_void_p(int)", lambda: ctypes.c_void_p(140000000000000))
bench("F_._stream()", lambda: F_._stream())
bench("F_._ld(t)", lambda: F_._ld(t))
bench("torch.empty(64, 304, cuda)", lambda: torch.empty(64, 304, dtype=torch.float32, device=t.device))
bench("torch.empty_like(t)", lambda: torch.empty_like(t))
bench("_lib.load()", lambda: _lib.load())
bench("lib.ggpm_padded_hidden(300) [ctypes call, 1 int arg]", lambda: lib.ggpm_padded_hidden(300))
bench("lib.ggpm_gemm_workspace_bytes(300,300,500)", lambda: lib.ggpm_gemm_workspace_bytes(300, 300, 500))
A = torch.randn(592, 320, device="cuda"); W = torch.randn(300, 320, device="cuda"); C = torch.empty(592, 304, device="cuda")
torch.cuda.synchronize()
def g(): F_.gemm(0, 1, 592, 300, 320, A, 320, W, 320, C, 304, 304)
bench("F_.gemm (one launch, host side; queue drained every 200)", g, 200)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    g()
    if i % 100 == 99: torch.cuda.synchronize()
print("F_.gemm incl. periodic sync: %.2f us per call" % ((time.perf_counter() - t0) / 2000 * 1e6))
s = F_._stream()
pa, pw, pc = F_._p(A), F_._p(W), F_._p(C)
def raw(): lib.ggpm_gemm(0, 1, 592, 300, 320, pa, 320, pw, 320, pc, 304, 304, None, 0, 0, 0, None, 0, s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    raw()
    if i % 100 == 99: torch.cuda.synchronize()
print("raw lib.ggpm_gemm with prebuilt args incl. periodic sync: %.2f us per call" % ((time.perf_counter() - t0) / 2000 * 1e6))
