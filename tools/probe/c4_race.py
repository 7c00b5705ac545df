#!/usr/bin/env python3
"""configs[4] B=32: is the encoder's backward deterministic, and does it depend on the second stream? (dev probe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from ggpm_amd import synth
from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
from ggpm_amd.property_vae import HierEncoderVAE
import bench
H, depth, B, latent = 600, 30, int(os.environ.get("B", "32")), 32
specs = synth.random_batch(515, 32, motifs=(46, 58), n_motif_vocab=500, n_attach_vocab=1500, chain=1.0)[:B]
sd = seeded_state_dict(encoder_param_shapes("GRU", H, 500, 1500), 5)
sd.update(seeded_state_dict(vae_head_shapes(H, latent), 6))
a = bench.make_args("GRU", H, depth, latent, 500, 1500)
tree, graph = synth.tensorize(specs)
def run():
    model = HierEncoderVAE(a).to("cuda:0")
    model.load_state_dict({(k if k.startswith("R_") else "encoder." + k): torch.from_numpy(v) for k, v in sd.items()})
    z, kl, outs = model((tree, graph), perturb_z=False)
    (kl + sum((o * o).sum() for o in outs)).backward()
    torch.cuda.synchronize()
    return {k: v.grad.detach().cpu().numpy().astype(np.float64) for k, v in model.named_parameters() if v.grad is not None}
runs = [run() for _ in range(3)]
for i in (1, 2):
    bad = [(k, float(np.abs(runs[0][k] - runs[i][k]).max() / max(np.abs(runs[0][k]).max(), 1e-30))) for k in runs[0] if not np.array_equal(runs[0][k], runs[i][k])]
    print("run 0 vs run %d: %d tensors differ %s" % (i, len(bad), sorted(bad, key=lambda t: -t[1])[:6]))
np.savez(os.environ.get("OUT", "/tmp/c4_race.npz"), **{k.replace(".", "__"): v for k, v in runs[0].items()})
prev = os.environ.get("CMP")
if prev:
    z = np.load(prev)
    bad = []
    for k, v in runs[0].items():
        w = z[k.replace(".", "__")]
        e = float(np.abs(v - w).max() / max(np.abs(w).max(), 1e-30))
        if e > 0: bad.append((k, e))
    print("vs %s: %d tensors differ; worst %s" % (prev, len(bad), sorted(bad, key=lambda t: -t[1])[:8]))
