"""How many Tensor.record_stream calls one VAE step makes, by target stream (dev probe).  On ROCm every recorded
(tensor, stream) pair costs an event record (~4.7 us of queue time) ON THAT STREAM when the tensor is released."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


class A:
    steps, pool, host_input = 10, 4, False


if os.environ.get("ENCODER"):          # the headline workload: encoder + KL heads + optimizer
    class B:
        steps, pool, host_input, warmup, no_full_depth = 10, 4, False, 4, True
    wl = bench.Workload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), B, 0, 1, torch.device("cuda:0"))
else:
    wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
if os.environ.get("IN_LOOP"):
    wl.step = wl.step_in_loop
for i in range(6):
    wl.step(i)
torch.cuda.synchronize()
main = torch.cuda.current_stream().cuda_stream
cnt, sites = collections.Counter(), collections.Counter()
orig = torch.Tensor.record_stream


def rs(self, stream):
    alloc_on_target = False
    cnt[stream.cuda_stream] += 1
    fr = traceback.extract_stack(limit=3)[0]
    sites[(os.path.basename(fr.filename), fr.lineno, "main" if stream.cuda_stream == main else hex(stream.cuda_stream)[-5:])] += 1
    return orig(self, stream)


torch.Tensor.record_stream = rs
for i in range(4):
    wl.step(i)
torch.cuda.synchronize()
torch.Tensor.record_stream = orig
print("record_stream calls per step by stream:", {("main" if k == main else hex(k)[-5:]): v / 4 for k, v in cnt.items()})
for (f, ln, st), v in sites.most_common(25):
    print("  %5.1f per step  -> %-6s %s:%d" % (v / 4, st, f, ln))
