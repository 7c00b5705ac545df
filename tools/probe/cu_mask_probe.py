"""What a CU-masked stream does on this device (dev probe): time of a compute-bound GEMM and of a chain of tiny kernels on
streams created with different masks."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ggpm_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
n = ctypes.c_int(0)
lib.ggpm_device_cu_count(ctypes.byref(n))
total = n.value
print("CUs:", total)
a = torch.randn(4096, 4096, device=dev)
small = torch.randn(64, 64, device=dev)


def masked(bits):
    words = (total + 31) // 32
    m = (ctypes.c_uint32 * words)()
    for i in bits:
        m[i // 32] |= 1 << (i % 32)
    out = ctypes.c_void_p(0)
    rc = lib.ggpm_stream_create_cu_mask(m, words, ctypes.byref(out))
    if rc != 0:
        return None
    return torch.cuda.ExternalStream(out.value, device=dev)


def timeit(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3):
            a @ a
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()
        g = (time.perf_counter() - t0) / 10 * 1e3
        t0 = time.perf_counter()
        x = small
        for _ in range(500):
            x = x * 1.0001
        torch.cuda.synchronize()
        c = (time.perf_counter() - t0) / 500 * 1e6
    print("%-34s gemm 4096^3 %.3f ms (%.0f TFLOP/s)   tiny-kernel chain %.2f us/launch" % (label, g, 2 * 4096 ** 3 / g / 1e9, c))


timeit(torch.cuda.Stream(device=dev), "plain stream")
for label, bits in (("all %d bits" % total, range(total)), ("bits 0..127", range(128)), ("bits 128..255", range(128, total)),
                    ("even bits", range(0, total, 2)), ("bits 0..31", range(32)), ("bits 0..63", range(64)),
                    ("every 8th bit", range(0, total, 8))):
    s = masked(list(bits))
    if s is None:
        print(label, "-> declined")
        continue
    timeit(s, label)
