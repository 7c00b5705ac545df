"""Wall time of the phases of the full VAE step with a device sync between them (dev tool, GPU box):
forward (encoder + decoder + losses), backward, optimizer; RNN=GRU|LSTM, and the GGPM_* switches apply."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


class A:
    steps, pool, host_input = 30, 8, False


wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
bench._settle_gc()
acc = [0.0, 0.0, 0.0]
N = 24
for i in range(N):
    _, dev_tensors, sch = wl.items[i % len(wl.items)]
    wl.opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, metrics = wl.model(None, None, dev_tensors, wl.orders, None, None, beta=0.1, perturb_z=True, schedule=sch)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    wl.opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2)):
        acc[k] += d
print("forward %.2f ms, backward %.2f ms, optimizer %.2f ms (sum %.2f)" % tuple([1e3 * a / N for a in acc] + [1e3 * sum(acc) / N]))
