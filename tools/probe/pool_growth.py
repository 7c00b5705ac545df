"""Does the caching allocator's pool stop growing in the bench loop?  (dev probe)  Runs the encoder step of bench.py for
STEPS steps and prints, every 30 steps, the device allocations so far, the reserved bytes and the sizes of the segments that
appeared since the last line (torch.cuda.memory_snapshot)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench


class A:
    pool, host_input, steps, warmup = int(os.environ.get("POOL", "16")), False, 30, 0


def segs():
    return collections.Counter((s["total_size"], s["stream"]) for s in torch.cuda.memory_snapshot())


def main():
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS[1]
    wl = bench.Workload(cfg, os.environ.get("RNN", "GRU"), A, 0, 1, dev)
    before = segs()
    for i in range(int(os.environ.get("STEPS", "600"))):
        wl.step(i)
        if (i + 1) % 30 == 0:
            torch.cuda.synchronize()
            st = torch.cuda.memory_stats(dev)
            now = segs()
            new = now - before
            before = now
            print("step %4d: %4d device allocations, %.2f GB reserved, %.2f GB allocated; new segments: %s"
                  % (i + 1, st.get("num_device_alloc", 0), st["reserved_bytes.all.current"] / 1e9,
                     st["allocated_bytes.all.current"] / 1e9,
                     ", ".join("%d x %.1f MB (stream %s)" % (n, sz / 1e6, "main" if not stm else hex(stm)[-5:])
                               for (sz, stm), n in sorted(new.items())) or "-"), flush=True)


if __name__ == "__main__":
    main()
