#!/usr/bin/env python3
"""Per-step host times of the encoder training step (dev tool): where do the slow timed regions come from?

    python tools/step_jitter.py [--rnn LSTM] [--steps 300]
Prints the distribution of host time per step (enqueue only) and of step-to-step completion times (an event per step),
and every step that took more than 3x the median with what the allocator did during it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench          # noqa: E402
import numpy as np    # noqa: E402
import torch          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rnn", default="LSTM")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--config", type=int, default=1)
    a0 = ap.parse_args()
    a = bench.parse_args(["--config", str(a0.config)])
    if a0.config == 4:
        a.pool = 2
    cfg = dict(bench.CONFIGS[a0.config])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.Workload(cfg, a0.rnn, a, 0, 1, dev)
    for i in range(16):
        wl.step(i)
    torch.cuda.synchronize()
    bench._settle_gc()
    host, evs, allocs = [], [], []
    stats0 = torch.cuda.memory_stats()
    for i in range(a0.steps):
        t0 = time.perf_counter()
        wl.step(16 + i)
        host.append(1e3 * (time.perf_counter() - t0))
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        evs.append(ev)
        s = torch.cuda.memory_stats()
        allocs.append((s["num_device_alloc"], s["num_device_free"], s["num_alloc_retries"]))
    torch.cuda.synchronize()
    gpu = [evs[i - 1].elapsed_time(evs[i]) for i in range(1, len(evs))]
    for name, v in (("host enqueue ms/step", host), ("GPU ms between step ends", gpu)):
        v = np.asarray(v)
        print("%s: median %.3f  mean %.3f  p99 %.3f  max %.3f (step %d)" % (name, np.median(v), v.mean(), np.percentile(v, 99),
                                                                            v.max(), int(v.argmax())))
    med = float(np.median(host))
    base = (stats0["num_device_alloc"], stats0["num_device_free"], stats0["num_alloc_retries"])
    prev = base
    for i, h in enumerate(host):
        if h > 3 * med:
            print("  step %d: host %.2f ms; device allocs/frees/retries during the step: %s" % (
                i, h, tuple(x - y for x, y in zip(allocs[i], prev))))
        prev = allocs[i]
    print("device allocs/frees/retries over the run:", tuple(x - y for x, y in zip(allocs[-1], base)))


if __name__ == "__main__":
    main()
