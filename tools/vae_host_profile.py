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
for i in range(3):
    wl.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(4):
    wl.step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "45")))
