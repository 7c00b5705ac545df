import cProfile, pstats, io, sys, os, itertools, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from ggpm_amd.dataloader import DevicePrefetcher
pool = bench.make_batches(8, 32, seed0=1000, motifs=(8, 12), n_motif=500, n_attach=1500)
it = iter(DevicePrefetcher(itertools.cycle(pool), depth=2))
for i in range(5): next(it)
torch.cuda.synchronize()
t = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for i in range(20): next(it)
pr.disable()
print("ms per batch", (time.perf_counter() - t) / 20 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:1600])
