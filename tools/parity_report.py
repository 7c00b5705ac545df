#!/usr/bin/env python3
"""Parity report (dev tool, GPU box): HIP path vs the oracle in fp32 AND fp64 on one synthetic batch of a BASELINE config.

For the four encoder outputs, the KL and every parameter gradient prints the norm-wise error max|a-b| / max|b| and the
per-element error of SURVEY.md section 8(d) (tests/golden_utils.elem_rel_err, floor ELEM_FLOOR) for three pairs:
HIP vs oracle-fp32, HIP vs oracle-fp64, oracle-fp32 vs oracle-fp64.  The last pair is the rounding noise of the
reference's own arithmetic; the HIP path is "as good as the reference" where its distance to fp64 is of that size.

    python tools/parity_report.py [--config 1] [--rnn GRU] [--batch N]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import bench
from golden_utils import ELEM_FLOOR, elem_rel_err, rel_err
from ggpm_amd import synth
from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
from ggpm_amd.property_vae import HierEncoderVAE
from oracle import ref_encoder as ref


def main():
    from ggpm_amd.launcher import host_cores
    torch.set_num_threads(host_cores())        # (the oracle's CPU runs: this process's CPU share, not every visible core)
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--rnn", default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--orders", action="store_true",
                    help="also evaluate the oracle's fp32 arithmetic in four equivalent orders (tests/golden_utils."
                         "oracle_fp32_orders) and print every order's distance to the fp64 run beside the HIP path's")
    a = ap.parse_args()
    cfg = bench.CONFIGS[a.config]
    rnn = a.rnn or cfg["rnn"]
    H, depth, latent = cfg["hidden"], cfg["depth"], cfg["latent"]
    n_motif, n_attach = cfg["vocab"]
    tree, graph = bench.make_batches(1, a.batch or cfg["batch"], 4242, cfg["gen"], n_motif, n_attach)[0]
    sd = seeded_state_dict(encoder_param_shapes(rnn, H, n_motif, n_attach), 5)
    sd.update(seeded_state_dict(vae_head_shapes(H, latent), 6))

    model = HierEncoderVAE(bench.make_args(rnn, H, depth, latent, n_motif, n_attach)).to("cuda:0")
    model.load_state_dict({(k if k.startswith("R_") else "encoder." + k): torch.from_numpy(v) for k, v in sd.items()})
    z, kl, outs = model((tree, graph), perturb_z=False)
    (kl + sum((o * o).sum() for o in outs)).backward()
    hip = {n: o.detach().cpu().numpy() for n, o in zip(("hroot", "hnode", "hinter", "hatom"), outs)}
    hip["kl"] = np.asarray(float(kl.detach()))
    for k, v in model.named_parameters():
        hip["grad " + (k[8:] if k.startswith("encoder.") else k)] = v.grad.cpu().numpy()

    res = {}
    for dtype in (torch.float32, torch.float64):
        p = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in sd.items()}
        tt, gt = ref.to_long_tensors(tree), ref.to_long_tensors(graph)
        routs = ref.hier_encoder_forward(p, rnn, depth, depth, tt, gt)
        _, rkl = ref.rsample_kl(p, routs[0])
        (rkl + sum((o * o).sum() for o in routs)).backward()
        r = {n: o.detach().numpy() for n, o in zip(("hroot", "hnode", "hinter", "hatom"), routs)}
        r["kl"] = np.asarray(float(rkl.detach()))
        for k, v in p.items():
            r["grad " + k] = v.grad.numpy()
        res[dtype] = r
    o32, o64 = res[torch.float32], res[torch.float64]
    print("config %d %s H=%d depth=%d batch=%d; per-element floor %g" % (a.config, rnn, H, depth, len(tree[-1]), ELEM_FLOOR))
    print("%-44s | %-21s | %-21s | %-21s" % ("tensor", "HIP vs oracle32", "HIP vs oracle64", "oracle32 vs oracle64"))
    print("%-44s | %10s %10s | %10s %10s | %10s %10s" % ("", "norm", "elem", "norm", "elem", "norm", "elem"))
    worst = [0.0] * 6
    for k in hip:
        pairs = ((hip[k], o32[k]), (hip[k], o64[k]), (o32[k], o64[k]))
        vals = []
        for x, y in pairs:
            x, y = np.atleast_1d(x), np.atleast_1d(y)
            vals += [rel_err(x, y), elem_rel_err(x, y, ELEM_FLOOR)]
        worst = [max(w, v) for w, v in zip(worst, vals)]
        print("%-44s | %10.2e %10.2e | %10.2e %10.2e | %10.2e %10.2e" % ((k,) + tuple(vals)))
    print("%-44s | %10.2e %10.2e | %10.2e %10.2e | %10.2e %10.2e" % (("WORST",) + tuple(worst)))
    if a.orders:
        from golden_utils import oracle_fp32_orders
        orders = oracle_fp32_orders(rnn, depth, sd, tree, graph)
        names = list(orders)
        print()
        print("norm-wise distance to the oracle's fp64 run: the HIP path and the oracle's fp32 arithmetic in %d equivalent "
              "evaluation orders" % len(names))
        print("%-44s | %10s | %s | %8s" % ("tensor", "HIP", " ".join("%14s" % n for n in names), "HIP/worst"))
        top = 0.0
        for k in hip:
            if np.abs(np.atleast_1d(o64[k])).max() == 0:
                continue
            e = [rel_err(np.atleast_1d(orders[n][k]), np.atleast_1d(o64[k])) for n in names]
            eh = rel_err(np.atleast_1d(hip[k]), np.atleast_1d(o64[k]))
            ratio = eh / max(max(e), 1e-30)
            top = max(top, ratio if max(e) > 5e-5 else 0.0)
            print("%-44s | %10.2e | %s | %8.2f" % (k, eh, " ".join("%14.2e" % x for x in e), ratio))
        print("largest HIP / worst-order ratio among the tensors whose fp32 orders are themselves > 5e-5 from fp64: %.2f" % top)


if __name__ == "__main__":
    main()
