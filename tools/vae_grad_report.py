#!/usr/bin/env python3
"""Per-parameter gradient errors of the full VAE step against one tests/golden/vae_*.npz fixture (dev tool, GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from golden_utils import VaeGolden
from ggpm_amd import synth
from ggpm_amd.decoder import DecodeSchedule
from ggpm_amd.property_vae import HierPropertyVAE
from ggpm_amd.vocab import IndexPairVocab

name = sys.argv[1] if len(sys.argv) > 1 else "vae_lstm_s43"
g = VaeGolden(name)
specs = g.specs()
tensors = synth.tensorize(specs)
model = HierPropertyVAE(g.args(IndexPairVocab(g.n_motif, g.n_attach))).to("cuda:0")
model.load_state_dict({k: torch.from_numpy(v) for k, v in g.state_dict().items()}, strict=False)
sch = DecodeSchedule.from_specs(specs, tensors)
loss, metrics = model(None, None, tensors, [None] * g.B, None, None, beta=g.beta, perturb_z=False, schedule=sch)
loss.backward()
torch.cuda.synchronize()
print(name, "loss", float(loss.detach()), "ref", float(g.z["loss"]), "kl", metrics["KL:"], float(g.z["kl"]))
for k, v in model.named_parameters():
    grad = v.grad.cpu().numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
    if "grad/" + k in g.z.files:
        want = g.z["grad/" + k]
        err = np.abs(grad - want).max() / max(np.abs(want).max(), 1e-30)
        print("%-50s full  rel err %.3e  (max |ref| %.3e)" % (k, err, np.abs(want).max()))
    else:
        idx = g.probe_indices(k, grad.size)
        st = g.z["gstat/" + k]
        err = np.abs(grad.reshape(-1)[idx] - g.z["gprobe/" + k]).max() / max(st[2], 1e-30)
        print("%-50s probe rel err %.3e  l2 %.4e vs %.4e" % (k, err, np.sqrt((grad.astype(np.float64) ** 2).sum()), st[1]))
