"""Does releasing tensors that were used on several streams hold up the NEXT launch on the current stream?  (dev probe)
A kernel, then (variant) release 60 tensors, then a second kernel; GPU time between the two kernels from events."""
import sys, time
import torch

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
a = torch.randn(1 << 20, device=dev)


def trial(variant):
    torch.cuda.synchronize()
    ts = []
    if variant != "none":
        with torch.cuda.stream(side):
            ts = [torch.randn(70000, device=dev) for _ in range(60)]      # allocated on the side stream
        main.wait_stream(side)
        if variant in ("recorded", "recorded+busy"):
            for t in ts:
                t.record_stream(main)
        outs = [t * 2 for t in ts]                                        # read on main
    if variant == "recorded+busy":
        with torch.cuda.stream(side):
            for _ in range(40):
                b = a * 1.0001                                             # the side stream still has work queued
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x = a * 2
    e0.record()
    t0 = time.perf_counter()
    ts = None
    outs = None                                                            # <- the release
    t1 = time.perf_counter()
    y = a * 3
    e1.record()
    torch.cuda.synchronize()
    print("%-14s release took %.3f ms on the host; GPU time between the two kernels %.3f ms" % (
        variant, (t1 - t0) * 1e3, e0.elapsed_time(e1)))


for v in ("none", "plain", "recorded", "recorded+busy", "none", "recorded+busy"):
    trial(v)
