"""Does waiting for an event recorded EARLY in a stream return before the work queued behind it has run?  (dev probe)"""
import time
import torch

dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev)
x = torch.zeros(6, device=dev)
for _ in range(3):
    (a @ a).sum().item()


def trial(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = torch.empty(6, dtype=torch.float32, pin_memory=True)
    host.copy_(x, non_blocking=True)
    ev = torch.cuda.Event(blocking=(kind == "blocking"))
    ev.record()
    for _ in range(200):
        b = a @ a                       # ~20 ms of queued work behind the event
    t1 = time.perf_counter()
    if kind == "query":
        while not ev.query():
            pass
    else:
        ev.synchronize()
    v = host.tolist()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-9s enqueue %.2f ms, wait for the early event %.2f ms, rest of the queue %.2f ms" % (
        kind, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))


for k in ("default", "blocking", "query", "default"):
    trial(k)
