"""How long does the caching allocator take for the encoder's big arenas? (dev tool)"""
import time, torch
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
sizes = [700_000_000 + 7_000_000 * (i % 5) for i in range(40)]
for mode in ("plain", "record_stream"):
    torch.cuda.synchronize()
    ts = []
    for i, n in enumerate(sizes):
        t0 = time.perf_counter()
        a = torch.empty(n, dtype=torch.uint8, device=dev)
        t1 = time.perf_counter()
        if mode == "record_stream":
            a.record_stream(side)
        a[:1024].zero_()
        del a
        ts.append(t1 - t0)
    torch.cuda.synchronize()
    ts2 = sorted(ts[10:])
    print("%-14s torch.empty(~700 MB): median %.1f us, max %.1f us" % (mode, 1e6 * ts2[len(ts2) // 2], 1e6 * ts2[-1]))
print(torch.cuda.memory_reserved() / 1e9, "GB reserved")
