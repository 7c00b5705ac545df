"""Where the full VAE step's time goes, WITHOUT a profiler attached (dev tool, GPU box): host time and GPU time of the
phase marks the package records when ``functional.MARKS`` is a list (forward: inputs, atom level posted, encoder issued,
atom level joined, tree-side levels issued, heads issued; backward: atom node reached / loop posted / returns, encoder
node reached, end-of-pass callback).  Per mark: host clock when the mark was taken and the time its event completed on
the GPU stream it was recorded on, both relative to the step's start, averaged over the timed steps.  A phase whose GPU
time trails its host time by little is issued as fast as the GPU consumes it (host-bound); one that trails by a lot is
GPU-bound.

    python tools/vae_phase_times.py [NAME=VALUE ...]     # RNN=LSTM, IN_LOOP=1, STEPS=20, PIPE=1; NAME: a ggpm_amd/_dev.py setting
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ggpm_amd import functional as F_


class A:
    steps, pool, host_input = 10, 4, False


def main():
    import ast
    from ggpm_amd import _dev
    for arg in sys.argv[1:]:
        name, value = arg.split("=", 1)
        assert hasattr(_dev, name), name
        setattr(_dev, name, ast.literal_eval(value))
        print("_dev.%s = %r" % (name, getattr(_dev, name)))
    cfg = bench.CONFIGS[1]
    wl = bench.VaeWorkload(cfg, os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
    if os.environ.get("IN_LOOP"):
        wl.step = wl.step_in_loop
    for i in range(12):
        wl.step(i)
    torch.cuda.synchronize()
    n = int(os.environ.get("STEPS", "20"))
    t0 = time.perf_counter()
    for i in range(n):
        wl.step(i)
    torch.cuda.synchronize()
    print("unmarked: %.2f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
    if os.environ.get("PIPE"):
        # no synchronisation between the steps (the loop as it runs): host time of every mark relative to its step's start
        # -- a phase that takes longer here than in the synchronised table below is where the host waits for the GPU
        acc, order = {}, []
        torch.cuda.synchronize()
        F_.MARKS = []
        starts = []
        t0 = time.perf_counter()
        for i in range(n):
            starts.append((len(F_.MARKS), time.perf_counter()))
            wl.step(i)
            F_.mark("step returns (metrics read)")
        torch.cuda.synchronize()
        print("pipelined: %.2f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
        marks, F_.MARKS = F_.MARKS, None
        starts.append((len(marks), None))
        for (a, h0), (b, _) in zip(starts[2:], starts[3:]):          # (the first two steps fill the pipeline)
            first = marks[a][2]            # "inputs on the device": reached by the main stream behind the previous step
            for name, h, ev in marks[a:b]:
                if name not in acc:
                    acc[name] = [0.0, 0.0, 0]
                    order.append(name)
                acc[name][0] += (h - h0) * 1e3
                acc[name][1] += first.elapsed_time(ev)
                acc[name][2] += 1
        print("%-36s %10s %22s" % ("mark", "host ms", "gpu ms after 1st mark"))
        for name in order:
            print("%-36s %10.3f %22.3f" % (name, acc[name][0] / acc[name][2], acc[name][1] / acc[name][2]))
    acc, order = {}, []
    total = 0.0
    for i in range(n):
        torch.cuda.synchronize()
        F_.MARKS = []
        start = torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        start.record()
        wl.step(i)
        F_.mark("step returns (metrics read)")
        torch.cuda.synchronize()
        total += time.perf_counter() - h0
        marks, F_.MARKS = F_.MARKS, None
        for name, h, ev in marks:
            if name not in acc:
                acc[name] = [0.0, 0.0, 0]
                order.append(name)
            a = acc[name]
            a[0] += (h - h0) * 1e3
            a[1] += start.elapsed_time(ev)
            a[2] += 1
    print("marked:   %.2f ms/step" % (total / n * 1e3))
    print("%-36s %10s %10s %8s" % ("mark", "host ms", "gpu ms", "lag"))
    for name in order:
        h, g, c = acc[name]
        print("%-36s %10.3f %10.3f %8.3f" % (name, h / c, g / c, (g - h) / c))


if __name__ == "__main__":
    main()
