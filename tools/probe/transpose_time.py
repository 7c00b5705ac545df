"""time csr_from_index(idx).T for an index with few ids (dev probe)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from ggpm_amd import functional as F_, _dev
_dev.INDEX_MEMO = False
dev = torch.device("cuda:0")
for rows, ncols in ((2832, 32), (600, 6214), (3000, 600), (9000, 1)):
    idx = torch.from_numpy(np.random.RandomState(0).randint(0, ncols, size=rows).astype(np.int32)).to(dev)
    for _ in range(3):
        F_.csr_from_index(idx, ncols).T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        F_.csr_from_index(idx, ncols)._build_T()
    e1.record(); torch.cuda.synchronize()
    print("rows %5d ncols %5d: %.1f us per transpose (incl. 3 allocations)" % (rows, ncols, 1e3 * e0.elapsed_time(e1) / 20))
