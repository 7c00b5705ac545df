#!/usr/bin/env python3
"""Golden fixtures for the decoder score heads and losses, produced by RUNNING THE
REFERENCE:  python tests/golden/make_golden_heads.py   (build container only; needs /root/reference)

The reference's ``HierMPNDecoder`` is constructed with seeded weights; its own ``get_topo_score``, ``get_cls_score``
(with the additive ``vocab.get_mask``), ``get_assm_score`` and loss modules (ggpm/decoder.py:35-69, 136-164, 262-283)
run on seeded prediction vectors / labels.  Recorded: inputs, the four score tensors, the loss and the gradients with
respect to every head parameter and every input vector.  Fixtures are DATA; no reference source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

import torch  # noqa: E402

from ggpm_amd.params import score_head_shapes, seeded_state_dict  # noqa: E402

HEAD_CASES = [
    # name, H, L, n_motif, n_attach, B, n_topo, n_cls, n_assm, max_cls, seed
    ("heads_s20", 24, 24, 11, 33, 3, 17, 13, 5, 4, 20),
    ("heads_s21", 300, 300, 200, 600, 8, 150, 80, 30, 8, 21),
]


class MaskVocab(mg.FakePairVocab):
    """FakePairVocab with the mask table / get_mask of ggpm/vocab.py:34-41,56-58 (attachment idx belongs to motif owner[idx])."""

    def __init__(self, n_motif, n_attach, owner):
        super().__init__(n_motif, n_attach)
        m = torch.zeros(n_motif, n_attach)
        m[torch.from_numpy(owner), torch.arange(n_attach)] = 1000.0
        self.mask = m - 1000.0

    def get_mask(self, cls_idx):
        return self.mask.index_select(index=cls_idx, dim=0)


def main():
    mg.import_reference()
    import ggpm.decoder as D
    from ggpm.vocab import common_atom_vocab
    for (name, H, L, n_motif, n_attach, B, n_topo, n_cls, n_assm, max_cls, seed) in HEAD_CASES:
        torch.set_default_dtype(torch.float32)
        rs = np.random.RandomState(seed)
        owner = rs.randint(0, n_motif, size=n_attach)
        owner[:n_motif] = np.arange(n_motif)                 # every motif owns at least one attachment
        vocab = MaskVocab(n_motif, n_attach, owner)
        dec = D.HierMPNDecoder(vocab, common_atom_vocab, "GRU", H, H, L, 2, 2, 0.0)
        sd = seeded_state_dict(score_head_shapes(H, L, H, n_motif, n_attach), seed)
        missing = dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        assert not missing.unexpected_keys
        f = lambda *s: (0.5 * rs.standard_normal(s)).astype(np.float32)
        inp = {"src_tree_vecs": f(B, L), "src_graph_vecs": f(B, L), "topo_vecs": f(n_topo, H), "cls_vecs": f(n_cls, H),
               "assm_vecs": f(n_assm, max_cls, H)}
        idx = {"topo_idx": rs.randint(0, B, n_topo), "topo_labels": rs.randint(0, 2, n_topo),
               "cls_idx": rs.randint(0, B, n_cls), "icls_labs": rs.randint(0, n_attach, n_cls),
               "assm_idx": np.repeat(rs.randint(0, B, (n_assm, 1)), max_cls, axis=1), "assm_labels": np.zeros(n_assm, np.int64)}
        idx["cls_labs"] = owner[idx["icls_labs"]]
        t = {k: torch.from_numpy(v).requires_grad_(True) for k, v in inp.items()}
        ti = {k: torch.from_numpy(np.asarray(v, dtype=np.int64)) for k, v in idx.items()}
        topo = dec.get_topo_score(t["src_tree_vecs"], ti["topo_idx"], t["topo_vecs"])
        cls, icls = dec.get_cls_score(t["src_tree_vecs"], ti["cls_idx"], t["cls_vecs"], ti["cls_labs"])
        assm = dec.get_assm_score(t["src_graph_vecs"], ti["assm_idx"], t["assm_vecs"])
        loss = (dec.topo_loss(topo, ti["topo_labels"].float()) + dec.cls_loss(cls, ti["cls_labs"]) +
                dec.icls_loss(icls, ti["icls_labs"]) + dec.assm_loss(assm, ti["assm_labels"])) / B
        loss.backward()
        out = {"topo": topo.detach().numpy(), "cls": cls.detach().numpy(), "icls": icls.detach().numpy(),
               "assm": assm.detach().numpy(), "loss": loss.detach().numpy(), "owner": owner.astype(np.int64)}
        for k, v in inp.items():
            out["in/" + k] = v
            out["din/" + k] = t[k].grad.numpy()
        for k, v in idx.items():
            out["idx/" + k] = np.asarray(v, dtype=np.int64)
        named = dict(dec.named_parameters())
        for k in sd:
            g = named[k].grad
            g = g.numpy() if g is not None else np.zeros(sd[k].shape, np.float32)
            if g.size > 20000:                          # big class projections: probes + statistics
                pi = mg.probe_indices(k, g.size, seed)
                out["gprobe/" + k] = g.reshape(-1)[pi]
                out["gstat/" + k] = np.array([g.sum(dtype=np.float64), np.sqrt((g.astype(np.float64) ** 2).sum()),
                                              np.abs(g).max()])
            else:
                out["grad/" + k] = g
        out["meta"] = np.array([H, L, n_motif, n_attach, B, seed], dtype=np.int64)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("%-10s loss=%.6f -> %s (%.1f KB)" % (name, float(loss.detach()), os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
