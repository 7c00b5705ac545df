"""Per-launch timing of the fused depth-step kernels on synthetic levels (dev tool)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggpm_amd import _lib, synth, rnn as R, functional as F_
lib = _lib.load()
dev = torch.device("cuda:0")
H, depth = int(os.environ.get("H", 300)), 20
rnn = os.environ.get("RNN", "GRU")
specs = synth.random_batch(1, 32, motifs=(8, 12))
tree, graph = synth.tensorize(specs)
for name, t, I in (("atom", graph, 62), ("tree", tree, H + 20)):
    bg = torch.from_numpy(t[3].astype(np.int64)).to(dev)
    E1 = bg.shape[0]
    x = torch.randn(E1, I, device=dev)
    mod = (R.GRU if rnn == "GRU" else R.LSTM)(I, H, depth).to(dev)
    xg = x.clone().requires_grad_(True)
    for it in range(3):
        out = mod(xg, bg); h = out if rnn == "GRU" else out[0]
        h.sum().backward()
    torch.cuda.synchronize()
    lib.ggpm_timing_enable(1)
    for it in range(5):
        out = mod(xg, bg); h = out if rnn == "GRU" else out[0]
        h.sum().backward()
    torch.cuda.synchronize()
    lib.ggpm_timing_enable(0)
    for which, nm in enumerate(["gru_fwd_a", "gru_bwd_a", "lstm_fwd_a", "lstm_bwd_a", "gru_fwd_b", "gru_bwd_b", "lstm_fwd_b", "lstm_bwd_b"]):
        n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        lib.ggpm_timing_collect(which, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
        if n.value:
            print("%s E1=%d %s: %d launches avg %.2f us  %.2f TFLOP/s" % (name, E1, nm, n.value, 1e3 * ms.value / n.value, fl.value / ms.value / 1e9), flush=True)
