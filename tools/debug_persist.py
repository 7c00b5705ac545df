"""Compare the persistent forward's stashes with the stepwise kernels' (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggpm_amd import _lib, synth, functional as F_
lib = _lib.load()
dev = torch.device("cuda:0")
H, depth, B = int(os.environ.get("H", 16)), int(os.environ.get("DEPTH", 3)), int(os.environ.get("B", 3))
I = 62
mot = tuple(int(v) for v in os.environ.get("MOT", "2,4").split(","))
specs = synth.random_batch(H + depth, B, motifs=mot, n_motif_vocab=11, n_attach_vocab=33)
tree, graph = synth.tensorize(specs)
bg = torch.from_numpy((tree if os.environ.get("LEVEL") == "tree" else graph)[3].astype(np.int64)).to(dev)
E1 = bg.shape[0]
pred = F_.csr_from_padded(bg, ncols=E1)
Hp = F_.padded_hidden(H)
torch.manual_seed(0)
X = torch.randn(3, E1, Hp, device=dev) * 0.5
X[:, :, H:] = 0
Wz, Wh, Ur = (torch.randn(H, H, device=dev) * 0.3 for _ in range(3))
bu = torch.randn(H, device=dev) * 0.1
f32 = dict(dtype=torch.float32, device=dev)
P = lambda t: F_._p(t)
out = {}
for mode in ("step", "persist"):
    Hs = torch.full((depth + 1, E1, Hp), float("nan"), **f32); Qs = torch.full((depth, E1, Hp), float("nan"), **f32)
    St = torch.zeros(5, depth, E1, Hp, **f32)
    wpack = torch.empty(int(lib.ggpm_gru_pack_floats(H)), **f32)
    if mode == "step":
        rc = lib.ggpm_gru_forward(E1, H, depth, P(X[0]), P(X[1]), P(X[2]), P(Wz), H, P(Ur), H, P(bu), P(Wh), H,
                                  P(pred.rowptr), P(pred.col), P(Hs), P(Qs), P(St[0]), P(St[1]), P(St[2]), P(St[3]),
                                  P(St[4]), P(wpack), 1, F_._stream())
    else:
        target = int(lib.ggpm_gru_persistent_target_rows(E1, H))
        table = pred.clusters(target)
        xwork = torch.full((int(lib.ggpm_gru_persistent_workspace_floats(E1, H)),), float("nan"), **f32)
        sync = torch.empty(8 + E1 // target + 4, dtype=torch.int32, device=dev)
        rc = lib.ggpm_gru_forward_persistent(E1, H, depth, P(X[0]), P(X[1]), P(X[2]), P(Wz), H, P(Ur), H, P(bu), P(Wh),
                                             H, P(pred.rowptr), P(pred.col), P(table), target, P(Hs), P(Qs), P(St[0]),
                                             P(St[1]), P(St[2]), P(St[3]), P(St[4]), P(wpack), P(xwork), P(sync),
                                             F_._stream())
        torch.cuda.synchronize()
        print("clusters", table[: int(table[0]) + 2].tolist(), "target", target, "timeout", int(sync[1]))
    torch.cuda.synchronize()
    assert rc == 0, rc
    out[mode] = (Hs, Qs, St)
names = ["S", "G", "Z", "M", "R"]
for mode in out:
    Hs, Qs, St = out[mode]
    print(mode, "nan rows in Hs per slot", [sorted(set(torch.isnan(Hs[t]).nonzero()[:, 0].tolist()))[:10] for t in range(depth + 1)])
    print(mode, "nan rows in Qs per slot", [sorted(set(torch.isnan(Qs[t]).nonzero()[:, 0].tolist()))[:10] for t in range(depth)])
    print(mode, "nan cols in Qs[1]", sorted(set(torch.isnan(Qs[1]).nonzero()[:, 1].tolist()))[:20])
for d in range(depth):
    for k, nm in enumerate(names):
        a, b = out["step"][2][k, d], out["persist"][2][k, d]
        err = (a - b).abs()
        print("depth %d %s max err %.3e" % (d, nm, float(err.max())), end="")
        if float(err.max()) > 1e-5:
            r, c = np.unravel_index(int(err.argmax()), err.shape)
            bad = (err > 1e-5).nonzero()
            print("  first bad row %d col %d; #bad %d rows %s cols %s" % (r, c, bad.shape[0], sorted(set(bad[:, 0].tolist()))[:12],
                                                                       sorted(set(bad[:, 1].tolist()))[:12]), end="")
        print()
    for nm, idx in (("H", 0), ("Q", 1)):
        if idx == 1 and d + 1 >= depth:
            continue
        a, b = out["step"][idx][d + 1], out["persist"][idx][d + 1]
        print("depth %d %s' max err %.3e" % (d, nm, float((a - b).abs().max())))
