"""cProfile of HierMPNDecoder.start_atom_level only (the serial host prefix of a VAE step; dev probe, GPU box)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench


class A:
    steps, pool, host_input = 10, 4, False


wl = bench.VaeWorkload(bench.CONFIGS[1], os.environ.get("RNN", "GRU"), A, torch.device("cuda:0"))
for i in range(12):
    wl.step(i)
torch.cuda.synchronize()
dec = wl.model.decoder
orig = dec.start_atom_level
pr = cProfile.Profile()
spent = [0.0, 0]


def timed(*a, **k):
    t0 = time.perf_counter()
    r = orig(*a, **k)
    spent[0] += time.perf_counter() - t0
    spent[1] += 1
    return r


dec.start_atom_level = timed
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
print("start_atom_level: %.3f ms per call (unprofiled)" % (1e3 * spent[0] / spent[1]))


def prof(*a, **k):
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()


dec.start_atom_level = prof
for i in range(16):
    wl.step(i)
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
