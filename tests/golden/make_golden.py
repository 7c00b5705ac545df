#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

What it does
------------
* imports the reference's hot-path modules (ggpm/rnn.py, encoder.py, mol_graph.py,
  property_vae.py) from /root/reference.  ``rdkit`` is not installed here and is never
  called on this path, so empty stand-in modules are registered for the import
  statements only (SURVEY.md section 8c recipe);
* builds synthetic molecules with ``ggpm_amd.synth`` and pushes them through the
  reference's OWN ``MolGraph.tensorize`` (only ``MolGraph.__init__``, the rdkit part,
  is replaced by a constructor that fills the attributes from the synthetic spec);
* loads seeded parameters into the reference's ``HierMPNEncoder`` and records, per case:
  the A0 input tensors, the four encoder outputs, message states after depth 1 and D,
  KL from the reference's ``HierPropertyVAE.rsample`` and parameter gradients of
  ``beta*KL + sum_k <c_k, out_k>`` (c_k seeded).

Fixtures are DATA (inputs / expected outputs); no reference source text is stored.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

REF = "/root/reference"


def import_reference():
    for name in ["rdkit", "rdkit.Chem", "rdkit.Chem.AllChem", "rdkit.Chem.rdchem", "rdkit.DataStructs",
                 "rdkit.RDLogger", "rdkit.Chem.Descriptors", "rdkit.Chem.rdmolops"]:
        m = MagicMock()
        m.__path__ = []
        sys.modules[name] = m
    pkg = types.ModuleType("ggpm")
    pkg.__path__ = [os.path.join(REF, "ggpm")]
    sys.modules["ggpm"] = pkg
    import ggpm.rnn, ggpm.encoder, ggpm.mol_graph, ggpm.vocab, ggpm.property_vae  # noqa
    return sys.modules["ggpm"]


import torch  # noqa: E402
import networkx as nx  # noqa: E402

from ggpm_amd import synth  # noqa: E402
from ggpm_amd.params import (encoder_param_shapes, vae_head_shapes, seeded_state_dict,  # noqa: E402
                             motif_encoder_param_shapes)

CASES = [
    # name, rnn, H, latent, depthT, depthG, B, motifs, n_motif, n_attach, seed, full_grads
    ("tiny_gru_s0", "GRU", 16, 8, 3, 3, 2, (2, 4), 11, 33, 0, True),
    ("tiny_gru_s1", "GRU", 16, 8, 3, 4, 3, (1, 5), 11, 33, 1, True),
    ("tiny_lstm_s0", "LSTM", 16, 8, 3, 3, 2, (2, 4), 11, 33, 0, True),
    ("tiny_lstm_s2", "LSTM", 24, 8, 2, 5, 3, (1, 5), 11, 33, 2, True),
    ("cfg_gru_s0", "GRU", 300, 32, 20, 20, 4, (7, 11), 50, 150, 0, False),
    ("cfg_gru_s1", "GRU", 300, 32, 20, 20, 4, (7, 11), 50, 150, 1, False),
    ("cfg_lstm_s0", "LSTM", 300, 32, 20, 20, 4, (7, 11), 50, 150, 0, False),
    ("cfg_lstm_s2", "LSTM", 250, 24, 20, 20, 5, (7, 11), 50, 150, 2, False),
    # edge cases: a single molecule of a single motif at depth 1 (no tree messages at all), hidden sizes that are not
    # multiples of 4 or 16 (unaligned GEMM operands), one larger molecule alone in its batch
    ("edge_gru_s30", "GRU", 18, 6, 1, 1, 1, (1, 1), 11, 33, 30, True),
    ("edge_lstm_s31", "LSTM", 21, 10, 1, 2, 2, (1, 2), 11, 33, 31, True),
    ("edge_gru_s32", "GRU", 300, 32, 2, 3, 1, (12, 12), 50, 150, 32, False),
]
ONLY = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
if ONLY:
    CASES = [c for c in CASES if c[0] in ONLY[0]]
MOTIF_CASES = [
    # name, rnn, H, depthT, B, motifs, n_motif, n_attach, seed
    ("motif_gru_s3", "GRU", 24, 4, 3, (2, 6), 11, 33, 3),
    ("motif_lstm_s4", "LSTM", 300, 20, 4, (7, 11), 50, 150, 4),
]
BETA = 0.1
N_PROBE = 64


class FakePairVocab:
    """dict-backed stand-in for PairVocab (only __getitem__/size are used on this path)."""

    def __init__(self, n_motif, n_attach):
        self.n = (n_motif, n_attach)

    def __getitem__(self, label):
        return int(label[0][1:]), int(label[1][1:])

    def size(self):
        return self.n


def patched_init(self, spec, mol=None):
    """Replacement for MolGraph.__init__ (the rdkit part): fill attributes from a MolSpec."""
    from ggpm.vocab import COMMON_ATOMS
    self.smiles = "synthetic"
    g = nx.DiGraph()
    for a, lab in enumerate(spec.atom_label):
        g.add_node(a, label=COMMON_ATOMS[lab])
    adj = spec.atom_adj()
    for u in range(spec.n_atoms):
        for v in adj[u]:
            bt = spec.bond_type(u, v)
            pos = spec.bond_pos.get((u, v))
            g.add_edge(u, v, label=(bt, pos) if pos is not None else bt)
    self.mol_graph = g
    t = nx.DiGraph()
    for i, cls in enumerate(spec.clusters):
        m, a = spec.motif_label[i]
        t.add_node(i, label=("m%d" % m, "a%d" % a), smiles="m%d" % m, ismiles="a%d" % a,
                   inter_label=[], cluster=list(cls), assm_cands=[])
    tadj = spec.tree_adj()
    for u in range(spec.n_motifs):
        for v in tadj[u]:
            t.add_edge(u, v, label=spec.tree_edge_label[(u, v)])
    self.mol_tree = t
    self.clusters = [list(c) for c in spec.clusters]
    self.order = list(spec.order)


def probe_indices(name, numel, seed):
    rs = np.random.RandomState((hash_name(name) + seed) % (2 ** 31))
    return rs.randint(0, numel, size=min(N_PROBE, numel))


def hash_name(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % (2 ** 31)
    return h


def loss_coeffs(shapes, seed):
    rs = np.random.RandomState(seed + 1000)
    return [rs.standard_normal(s).astype(np.float32) for s in shapes]


def main():
    ggpm = import_reference()
    from ggpm.mol_graph import MolGraph
    from ggpm.encoder import HierMPNEncoder
    from ggpm.vocab import common_atom_vocab
    from ggpm.property_vae import HierPropertyVAE
    from ggpm.nnutils import make_cuda
    MolGraph.__init__ = patched_init

    for (name, rnn, H, latent, dT, dG, B, motifs, n_motif, n_attach, seed, full) in CASES:
        torch.manual_seed(seed)
        specs = synth.random_batch(seed, B, motifs=motifs, n_motif_vocab=n_motif, n_attach_vocab=n_attach)
        vocab = FakePairVocab(n_motif, n_attach)
        batch = [[s, 0.0, 0.0] for s in specs]
        _, _, (tree_t, graph_t), orders, _, _ = MolGraph.tensorize(batch, vocab, common_atom_vocab)
        tree_np = [np.asarray(x.numpy()) for x in tree_t[:-1]] + [tree_t[-1]]
        graph_np = [np.asarray(x.numpy()) for x in graph_t[:-1]] + [graph_t[-1]]

        # our own layout restatement must agree with the reference's tensorize
        mine_tree, mine_graph = synth.tensorize(specs)
        for a, b in zip(tree_np[:-1], mine_tree[:-1]):
            assert a.shape == b.shape and (a == b).all(), "tree layout mismatch in " + name
        for a, b in zip(graph_np[:-1], mine_graph[:-1]):
            assert a.shape == b.shape and (a == b).all(), "graph layout mismatch in " + name
        assert [tuple(x) for x in tree_np[-1]] == [tuple(x) for x in mine_tree[-1]]
        assert [tuple(x) for x in graph_np[-1]] == [tuple(x) for x in mine_graph[-1]]

        shapes = encoder_param_shapes(rnn, H, n_motif, n_attach)
        sd = seeded_state_dict(shapes, seed)
        head = seeded_state_dict(vae_head_shapes(H, latent), seed + 7)

        out = {}
        for dtype, tag in ((torch.float32, ""), (torch.float64, "_f64")):
            torch.set_default_dtype(dtype)   # the reference allocates its states with the default dtype
            enc = HierMPNEncoder(vocab, common_atom_vocab, rnn, H, H, dT, dG, 0.0)
            missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            enc = enc.to(dtype)
            for attr in ("E_a", "E_b", "E_apos", "E_pos"):
                setattr(enc, attr, getattr(enc, attr).to(dtype))
            R_mean = torch.nn.Linear(H, latent).to(dtype)
            R_var = torch.nn.Linear(H, latent).to(dtype)
            with torch.no_grad():
                R_mean.weight.copy_(torch.from_numpy(head["R_mean.weight"])); R_mean.bias.copy_(torch.from_numpy(head["R_mean.bias"]))
                R_var.weight.copy_(torch.from_numpy(head["R_var.weight"])); R_var.bias.copy_(torch.from_numpy(head["R_var.bias"]))
            tt, gt = make_cuda((tree_np, graph_np))

            captured = {}
            hk = [enc.graph_encoder.rnn.register_forward_hook(lambda m, i, o: captured.__setitem__("atom", o)),
                  enc.inter_encoder.rnn.register_forward_hook(lambda m, i, o: captured.__setitem__("inter", o)),
                  enc.tree_encoder.rnn.register_forward_hook(lambda m, i, o: captured.__setitem__("tree", o))]
            hroot, hnode, hinter, hatom = enc(tt, gt)
            for h in hk:
                h.remove()
            outs = [hroot, hnode, hinter, hatom]
            z, kl = HierPropertyVAE.rsample(None, hroot, R_mean, R_var, perturb=False)
            coeffs = loss_coeffs([tuple(o.shape) for o in outs], seed)
            loss = BETA * kl
            for c, o in zip(coeffs, outs):
                loss = loss + (torch.from_numpy(c).to(dtype) * o).sum()
            loss.backward()

            for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs):
                out[k + tag] = o.detach().numpy()
            for lvl in ("atom", "inter", "tree"):
                o = captured[lvl]
                o = o[0] if isinstance(o, tuple) else o
                out[lvl + "_hD" + tag] = o.detach().numpy()
            # state after ONE depth on the atom level (depth attribute drives the loop)
            with torch.no_grad():
                emb = enc.embed_graph(gt)
                saved = enc.graph_encoder.rnn.depth
                enc.graph_encoder.rnn.depth = 1
                h1 = enc.graph_encoder.rnn(emb[1], emb[3])
                enc.graph_encoder.rnn.depth = saved
                h1 = h1[0] if isinstance(h1, tuple) else h1
            out["atom_h1" + tag] = h1.numpy()
            out["kl" + tag] = kl.detach().numpy()
            out["z" + tag] = z.detach().numpy()
            out["loss" + tag] = loss.detach().numpy()

            named = list(enc.named_parameters()) + [("R_mean.weight", R_mean.weight), ("R_mean.bias", R_mean.bias),
                                                    ("R_var.weight", R_var.weight), ("R_var.bias", R_var.bias)]
            for pname, prm in named:
                g = prm.grad.detach().numpy() if prm.grad is not None else np.zeros(tuple(prm.shape))
                if full:
                    out["grad/" + pname + tag] = g
                else:
                    idx = probe_indices(pname, g.size, seed)
                    out["gprobe/" + pname + tag] = g.reshape(-1)[idx]
                    out["gstat/" + pname + tag] = np.array([g.sum(dtype=np.float64), np.sqrt((g.astype(np.float64) ** 2).sum()),
                                                            np.abs(g).max()])

        torch.set_default_dtype(torch.float32)
        for i, k in enumerate(("fnode", "fmess", "agraph", "bgraph", "cgraph")):
            out["tree_" + k] = tree_np[i].astype(np.int32)
        out["tree_scope"] = np.asarray(tree_np[-1], dtype=np.int32)
        for i, k in enumerate(("fnode", "fmess", "agraph", "bgraph")):
            out["graph_" + k] = graph_np[i].astype(np.int32)
        out["graph_scope"] = np.asarray(graph_np[-1], dtype=np.int32)
        out["meta"] = np.array([H, latent, dT, dG, B, n_motif, n_attach, seed, motifs[0], motifs[1], int(full)], dtype=np.int64)
        out["rnn"] = np.array(rnn)
        out["beta"] = np.array(BETA)
        # keep fixtures small: the float64 pass is stored rounded to fp32 ("best possible fp32 answer"),
        # and for the config-shaped cases the bulky per-message states are kept for the fp32 pass only
        for k in list(out.keys()):
            if k.endswith("_f64"):
                if not full and (k.startswith("atom_h") or k.startswith("inter_h") or k.startswith("tree_h")):
                    del out[k]
                elif k not in ("kl_f64", "loss_f64"):
                    out[k] = out[k].astype(np.float32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-14s E_atom=%d E_tree=%d  kl=%.6f loss=%.6f  -> %s (%.1f KB)" % (
            name, graph_np[1].shape[0] - 1, tree_np[1].shape[0] - 1, float(out["kl"]), float(out["loss"]),
            os.path.basename(path), os.path.getsize(path) / 1024))


def main_motif():
    """MotifEncoder (ggpm/encoder.py:252-341): (root, node) outputs and full parameter gradients."""
    from ggpm.mol_graph import MolGraph
    from ggpm.encoder import MotifEncoder
    from ggpm.vocab import common_atom_vocab
    from ggpm.nnutils import make_cuda
    MolGraph.__init__ = patched_init
    for (name, rnn, H, dT, B, motifs, n_motif, n_attach, seed) in MOTIF_CASES:
        torch.set_default_dtype(torch.float32)
        specs = synth.random_batch(seed, B, motifs=motifs, n_motif_vocab=n_motif, n_attach_vocab=n_attach)
        vocab = FakePairVocab(n_motif, n_attach)
        _, _, (tree_t, graph_t), _, _, _ = MolGraph.tensorize([[s, 0.0, 0.0] for s in specs], vocab, common_atom_vocab)
        tree_np = [np.asarray(x.numpy()) for x in tree_t[:-1]] + [tree_t[-1]]
        graph_np = [np.asarray(x.numpy()) for x in graph_t[:-1]] + [graph_t[-1]]
        sd = seeded_state_dict(motif_encoder_param_shapes(rnn, H, n_motif, n_attach), seed)
        enc = MotifEncoder(vocab, common_atom_vocab, rnn, H, H, dT, dT, 0.0)
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        tt, _ = make_cuda((tree_np, graph_np))
        root, node = enc(tt)
        coeffs = loss_coeffs([tuple(root.shape), tuple(node.shape)], seed)
        loss = (torch.from_numpy(coeffs[0]) * root).sum() + (torch.from_numpy(coeffs[1]) * node).sum()
        loss.backward()
        out = {"root": root.detach().numpy(), "node": node.detach().numpy(), "loss": loss.detach().numpy()}
        full = H <= 32
        for pname, prm in enc.named_parameters():
            g = prm.grad.detach().numpy()
            if full:
                out["grad/" + pname] = g
            else:
                idx = probe_indices(pname, g.size, seed)
                out["gprobe/" + pname] = g.reshape(-1)[idx]
                out["gstat/" + pname] = np.array([g.sum(dtype=np.float64), np.sqrt((g.astype(np.float64) ** 2).sum()),
                                                  np.abs(g).max()])
        for i, k in enumerate(("fnode", "fmess", "agraph", "bgraph", "cgraph")):
            out["tree_" + k] = tree_np[i].astype(np.int32)
        out["tree_scope"] = np.asarray(tree_np[-1], dtype=np.int32)
        for i, k in enumerate(("fnode", "fmess", "agraph", "bgraph")):
            out["graph_" + k] = graph_np[i].astype(np.int32)
        out["graph_scope"] = np.asarray(graph_np[-1], dtype=np.int32)
        out["meta"] = np.array([H, 0, dT, dT, B, n_motif, n_attach, seed, motifs[0], motifs[1], int(full)], dtype=np.int64)
        out["rnn"] = np.array(rnn)
        out["beta"] = np.array(0.0)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-14s E_tree=%d loss=%.6f -> %s (%.1f KB)" % (name, tree_np[1].shape[0] - 1, float(out["loss"]),
                                                             os.path.basename(path), os.path.getsize(path) / 1024))


SPARSE_CASES = [
    # name, rnn, E1, I, H, depth, ms, K, seed
    ("sparse_gru_s5", "GRU", 60, 13, 24, 3, 17, 4, 5),
    ("sparse_lstm_s6", "LSTM", 60, 13, 24, 3, 17, 4, 6),
    ("sparse_gru_s7", "GRU", 300, 62, 100, 5, 80, 5, 7),
    ("sparse_lstm_s8", "LSTM", 300, 62, 100, 5, 80, 5, 8),
]


def main_sparse():
    """GRU/LSTM.sparse_forward (ggpm/rnn.py:52-59, 110-121) on seeded states/subsets: outputs and all gradients."""
    from ggpm.rnn import GRU, LSTM
    from ggpm_amd.params import rnn_param_shapes
    from golden_utils import sparse_inputs
    for (name, rnn, E1, I, H, depth, ms, K, seed) in SPARSE_CASES:
        torch.set_default_dtype(torch.float32)
        h, c, submess, x, bg, coef = sparse_inputs(E1, I, H, ms, K, seed)
        sd = seeded_state_dict(rnn_param_shapes(rnn, I, H), seed)
        mod = (GRU if rnn == "GRU" else LSTM)(I, H, depth)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        ht = torch.from_numpy(h).requires_grad_(True)
        ct = torch.from_numpy(c).requires_grad_(True)
        xt = torch.from_numpy(x).requires_grad_(True)
        if rnn == "GRU":
            ho = mod.sparse_forward(ht, xt, torch.from_numpy(submess), torch.from_numpy(bg))
            loss = (torch.from_numpy(coef[0]) * ho).sum()
            co = None
        else:
            ho, co = mod.sparse_forward((ht, ct), xt, torch.from_numpy(submess), torch.from_numpy(bg))
            loss = (torch.from_numpy(coef[0]) * ho).sum() + (torch.from_numpy(coef[1]) * co).sum()
        loss.backward()
        out = {"h_out": ho.detach().numpy(), "dh_in": ht.grad.numpy(), "dx": xt.grad.numpy(), "loss": loss.detach().numpy()}
        if co is not None:
            out["c_out"] = co.detach().numpy()
            out["dc_in"] = ct.grad.numpy()
        for pname, prm in mod.named_parameters():
            out["grad/" + pname] = prm.grad.numpy()
        out["meta"] = np.array([E1, I, H, depth, ms, K, seed], dtype=np.int64)
        out["rnn"] = np.array(rnn)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-14s loss=%.6f -> %s (%.1f KB)" % (name, float(out["loss"]), os.path.basename(path),
                                                   os.path.getsize(path) / 1024))


if __name__ == "__main__":
    if "--sparse-only" in sys.argv:
        import_reference()
        main_sparse()
        sys.exit(0)
    if "--motif-only" not in sys.argv:
        main()
    else:
        import_reference()
    if ONLY:
        sys.exit(0)
    main_motif()
    main_sparse()
