"""Synchronising calls inside one VAE step, with their Python stacks (torch.cuda.set_sync_debug_mode; dev probe)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


class A:
    steps, pool, host_input = 10, 4, False


wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
if os.environ.get("IN_LOOP"):
    wl.step = wl.step_in_loop
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
import traceback
torch.cuda.set_sync_debug_mode("warn")
orig = warnings.showwarning


def show(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", message)
    for fr in traceback.extract_stack()[:-1]:
        if "ggpm_amd" in fr.filename or "bench.py" in fr.filename:
            print("    %s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.line))


warnings.showwarning = show
warnings.simplefilter("always")
wl.step(0)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
