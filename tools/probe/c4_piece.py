#!/usr/bin/env python3
"""configs[4] B=32 vs its B=4 pieces, piece by piece: the full batch's gradient of (sum of squares over ONE piece's rows)
against that piece run alone (dev probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ggpm_amd import synth
from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
from ggpm_amd.property_vae import HierEncoderVAE
import bench
H, depth, B, latent, sub = 600, 30, 32, 32, 4
specs = synth.random_batch(515, B, motifs=(46, 58), n_motif_vocab=500, n_attach_vocab=1500, chain=1.0)
sd = seeded_state_dict(encoder_param_shapes("GRU", H, 500, 1500), 5)
sd.update(seeded_state_dict(vae_head_shapes(H, latent), 6))
a = bench.make_args("GRU", H, depth, latent, 500, 1500)
keys = ["encoder.E_i.0.weight", "encoder.inter_encoder.W_o.0.weight", "encoder.inter_encoder.W_o.0.bias", "encoder.E_c.0.weight",
        "encoder.tree_encoder.W_o.0.weight", "encoder.graph_encoder.W_o.0.weight", "encoder.W_i.0.weight", "encoder.inter_encoder.rnn.W_z.weight"]
def run(group, rows_of=None, which=(0, 1, 2, 3)):
    tree, graph = synth.tensorize(group)
    model = HierEncoderVAE(a).to("cuda:0")
    model.load_state_dict({(k if k.startswith("R_") else "encoder." + k): torch.from_numpy(v) for k, v in sd.items()})
    z, kl, outs = model((tree, graph), perturb_z=False)
    loss = 0
    for n in which:
        o = outs[n]
        if rows_of is None:
            loss = loss + (o * o).sum()
        else:
            m = torch.zeros(o.shape[0], 1, device=o.device)
            for i in rows_of:
                (t0, tn), (a0, an) = tree[-1][i], graph[-1][i]
                lo, ln = ((i, 1), (t0, tn), (t0, tn), (a0, an))[n]
                m[lo:lo + ln] = 1
            loss = loss + (o * o * m).sum()
    loss.backward()
    torch.cuda.synchronize()
    return {k: v.grad.detach().double().cpu().numpy() for k, v in model.named_parameters() if v.grad is not None and k in keys}
for which in ((0, 1, 2, 3), (1,), (2,), (3,), (0,)):
    print("loss over outputs", which)
    for j in range(0, 3):
        big = run(specs, rows_of=list(range(j * sub, (j + 1) * sub)), which=which)
        piece = run(specs[j * sub:(j + 1) * sub], which=which)
        print("  piece %d: " % j + "  ".join("%s %.1e" % (k.split("encoder.")[1][:22], float(np.abs(big[k] - piece[k]).max() / max(np.abs(piece[k]).max(), 1e-30))) for k in keys if k in big and k in piece))
