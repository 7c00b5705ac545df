"""Times the two halves of StepMetrics._fill inside the pipelined VAE loop (dev probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from ggpm_amd import property_vae as PV


class A:
    steps, pool, host_input = 10, 4, False


wl = bench.VaeWorkload(bench.CONFIGS[1], "GRU", A, torch.device("cuda:0"))
for i in range(12):
    wl.step(i)
torch.cuda.synchronize()
rec = []
orig = PV.StepMetrics._fill


def fill(self):
    if not self._ready and self._event is not None:
        t0 = time.perf_counter()
        q = self._event.query()
        t1 = time.perf_counter()
        self._event.synchronize()
        t2 = time.perf_counter()
        self._host.tolist()
        t3 = time.perf_counter()
        rec.append((q, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    return orig(self)


PV.StepMetrics._fill = fill
t0 = time.perf_counter()
for i in range(20):
    wl.step(i)
torch.cuda.synchronize()
print("%.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
for r in rec[5:12]:
    print("event done at read: %s; query %.3f ms, synchronize %.3f ms, tolist %.3f ms" % r)
