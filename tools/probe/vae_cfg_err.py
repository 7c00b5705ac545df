#!/usr/bin/env python3
"""Per-parameter gradient distances of the full VAE step at configs[1] shapes: HIP vs the oracle's fp32 run, and both vs
the oracle's fp64 run (dev tool, GPU box).  Args: [rnn H L depth B seed]"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from ggpm_amd import synth
from ggpm_amd.decoder import DecodeSchedule
from ggpm_amd.params import vae_param_shapes, seeded_state_dict
from ggpm_amd.property_vae import HierPropertyVAE
from ggpm_amd.vocab import IndexPairVocab
from oracle import ref_encoder as ref, ref_decoder as refd

rnn, H, L, depth, B, seed = (sys.argv[1:] + ["GRU", 300, 32, 20, 32, 4242][len(sys.argv) - 1:])
H, L, depth, B, seed = int(H), int(L), int(depth), int(B), int(seed)
n_motif, n_attach = 500, 1500
specs = synth.random_batch(seed, B, motifs=(8, 12), n_motif_vocab=n_motif, n_attach_vocab=n_attach)
tensors = synth.tensorize(specs)
sch = DecodeSchedule.from_specs(specs, tensors)
sd = seeded_state_dict(vae_param_shapes(rnn, H, L, n_motif, n_attach), seed)
voc = IndexPairVocab(n_motif, n_attach)
a = types.SimpleNamespace(vocab=voc, rnn_type=rnn, embed_size=H, hidden_size=H, atom_vocab=types.SimpleNamespace(size=lambda: 38),
                          depthT=depth, depthG=depth, diterT=1, diterG=5, dropout=0.0, latent_size=L, tie_embedding=False)
model = HierPropertyVAE(a).to("cuda:0")
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
loss, metrics = model(None, None, tensors, [None] * B, None, None, beta=0.1, perturb_z=False, schedule=sch)
loss.backward()
torch.cuda.synchronize()
runs = {}
for dt in (torch.float32, torch.float64):
  try:
    p = {k: torch.from_numpy(v).to(dt).requires_grad_(True) for k, v in sd.items()}
    tt, gt = ref.to_long_tensors(tensors[0]), ref.to_long_tensors(tensors[1])
    rl, rkl, accs, _ = refd.vae_forward(p, rnn, depth, depth, 1, 5, tt, gt, sch, voc.mask if dt == torch.float32 else voc.mask.double(), 0.1)
    rl.backward()
    runs[dt] = (float(rl.detach()), {k: (v.grad.numpy().astype(np.float64) if v.grad is not None else None) for k, v in p.items()})
  except Exception as exc:
    print("oracle run in", dt, "failed:", repr(exc))
    runs[dt] = runs[torch.float32]
print("loss HIP %.8f  oracle f32 %.8f  f64 %.8f" % (float(loss.detach()), runs[torch.float32][0], runs[torch.float64][0]))
print("%-44s %10s %10s %10s" % ("grad", "HIP-f32", "HIP-f64", "f32-f64"))
for k, v in model.named_parameters():
    w32, w64 = runs[torch.float32][1][k], runs[torch.float64][1][k]
    if w64 is None:
        continue
    got = v.grad.cpu().numpy().astype(np.float64) if v.grad is not None else np.zeros_like(w64)
    sc = max(np.abs(w64).max(), 1e-30)
    print("%-44s %10.2e %10.2e %10.2e" % (k, np.abs(got - w32).max() / sc, np.abs(got - w64).max() / sc, np.abs(w32 - w64).max() / sc))
