"""cProfile of the full VAE step's host side (dev tool, GPU box): where the Python time of the op-by-op decoder goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


class A:
    steps, pool, host_input = 10, 4, False


cfg = bench.CONFIGS[1]
wl = bench.VaeWorkload(cfg, os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
if os.environ.get("IN_LOOP"):         # the vae_train.py call shape: host batch + networkx graphs in, schedule built in the step
    wl.step = wl.step_in_loop
for i in range(12):          # three passes over the pool: plans, uploads and memoised index structures exist
    wl.step(i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
print("unprofiled: %.2f ms/step" % ((time.perf_counter() - t0) / 8 * 1e3))
torch.autograd.set_multithreading_enabled(False)      # backward functions on this thread, so that cProfile sees them
pr = cProfile.Profile()
pr.enable()
for i in range(8):
    wl.step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
print("(8 profiled steps)")
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))
