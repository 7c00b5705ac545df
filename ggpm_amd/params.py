"""Parameter names / shapes of the hierarchical encoder and a portable seeded initialiser.

The names and shapes are the reference's ``state_dict`` contract (SURVEY.md section 8b;
reference ggpm/encoder.py:43-90, ggpm/rnn.py:13-16,69-72): checkpoints written by the
reference load into the drop-in unchanged.  ``seeded_state_dict`` draws values from
numpy's legacy MT19937 stream, which is stable across numpy/torch versions, so golden
fixtures only need to record a seed instead of megabytes of weights.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

MAX_POS = 20
NUM_BOND_TYPES = 4


def rnn_param_shapes(rnn_type: str, input_size: int, hidden: int) -> "OrderedDict[str, Tuple[int, ...]]":
    I, H = input_size, hidden
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if rnn_type == "GRU":
        s["W_z.weight"] = (H, I + H); s["W_z.bias"] = (H,)
        s["W_r.weight"] = (H, I)
        s["U_r.weight"] = (H, H); s["U_r.bias"] = (H,)
        s["W_h.weight"] = (H, I + H); s["W_h.bias"] = (H,)
    elif rnn_type == "LSTM":
        for g in ("W_i", "W_o", "W_f", "W"):
            s[g + ".0.weight"] = (H, I + H); s[g + ".0.bias"] = (H,)
    else:
        raise ValueError("unsupported rnn cell type " + rnn_type)
    return s


def encoder_param_shapes(rnn_type: str, hidden: int, n_motif: int, n_attach: int,
                         atom_size: int = 38, embed: int | None = None):
    H = hidden
    He = H if embed is None else embed
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["E_c.0.weight"] = (n_motif, He)
    s["E_i.0.weight"] = (n_attach, He)
    s["W_c.0.weight"] = (H, He + H); s["W_c.0.bias"] = (H,)
    s["W_i.0.weight"] = (H, 2 * He); s["W_i.0.bias"] = (H,)
    s["W_root.0.weight"] = (H, 2 * H); s["W_root.0.bias"] = (H,)
    levels = (("tree_encoder", H + MAX_POS, H), ("inter_encoder", H + MAX_POS, H),
              ("graph_encoder", atom_size + NUM_BOND_TYPES + MAX_POS, atom_size))
    for name, I, F in levels:
        s[name + ".W_o.0.weight"] = (H, F + H); s[name + ".W_o.0.bias"] = (H,)
        for k, v in rnn_param_shapes(rnn_type, I, H).items():
            s[name + ".rnn." + k] = v
    return s


def motif_encoder_param_shapes(rnn_type: str, hidden: int, n_motif: int, n_attach: int):
    """MotifEncoder (reference ggpm/encoder.py:252-341): embeddings + W_root + one tree-level MPNEncoder."""
    H = hidden
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["E_c.0.weight"] = (n_motif, H)
    s["E_i.0.weight"] = (n_attach, H)
    s["W_root.0.weight"] = (H, 2 * H); s["W_root.0.bias"] = (H,)
    s["tree_encoder.W_o.0.weight"] = (H, 2 * H); s["tree_encoder.W_o.0.bias"] = (H,)
    for k, v in rnn_param_shapes(rnn_type, H + MAX_POS, H).items():
        s["tree_encoder.rnn." + k] = v
    return s


def vae_head_shapes(hidden: int, latent: int):
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["R_mean.weight"] = (latent, hidden); s["R_mean.bias"] = (latent,)
    s["R_var.weight"] = (latent, hidden); s["R_var.bias"] = (latent,)
    return s


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, bias_scale: float = 0.1
                      ) -> "OrderedDict[str, np.ndarray]":
    """xavier-normal matrices (as vae_train.py:48-53 does) and small random vectors, fp32."""
    rs = np.random.RandomState(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in shapes.items():
        if len(shape) == 1:
            v = bias_scale * rs.standard_normal(shape)
        else:
            std = np.sqrt(2.0 / (shape[0] + shape[1]))
            v = std * rs.standard_normal(shape)
        out[name] = v.astype(np.float32)
    return out


def score_head_shapes(hidden: int, latent: int, embed: int, n_motif: int, n_attach: int):
    """topoNN / clsNN / iclsNN / matchNN / W_assm of HierMPNDecoder (reference ggpm/decoder.py:35-58)."""
    H, L = hidden, latent
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, n_out in (("topoNN", 1), ("clsNN", n_motif), ("iclsNN", n_attach)):
        s[name + ".0.weight"] = (H, H + L); s[name + ".0.bias"] = (H,)
        s[name + ".3.weight"] = (n_out, H); s[name + ".3.bias"] = (n_out,)
    s["matchNN.0.weight"] = (H, H + embed + MAX_POS); s["matchNN.0.bias"] = (H,)
    s["W_assm.weight"] = (L, H); s["W_assm.bias"] = (L,)
    return s


def vae_param_shapes(rnn_type: str, hidden: int, latent: int, n_motif: int, n_attach: int, embed: int | None = None):
    """Every parameter of HierPropertyVAE (reference ggpm/property_vae.py:11-24) under its state_dict name, without the
    alias entries the reference's decoder registers twice (``decoder.rnn_cell.*`` = ``decoder.hmpn.tree_encoder.rnn.*``,
    ``decoder.E_assm.*`` = ``decoder.hmpn.E_i.*``).  With ``tie_embedding`` the encoder's ``E_c`` / ``E_i`` ARE the
    decoder's: load ``tied_state_dict`` of the result."""
    He = hidden if embed is None else embed
    enc = encoder_param_shapes(rnn_type, hidden, n_motif, n_attach, embed=He)
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for k, v in enc.items():
        s["encoder." + k] = v
    for k, v in enc.items():
        if not k.startswith("W_root"):
            s["decoder.hmpn." + k] = v
    for k, v in score_head_shapes(hidden, latent, He, n_motif, n_attach).items():
        s["decoder." + k] = v
    if latent != hidden:
        s["decoder.W_root.weight"] = (hidden, latent); s["decoder.W_root.bias"] = (hidden,)
    for k, v in vae_head_shapes(hidden, latent).items():
        s[k] = v
    return s


def tied_state_dict(sd):
    """tie_embedding (ggpm/encoder.py:92-94): the encoder's embeddings are the decoder's."""
    out = OrderedDict(sd)
    for k in ("E_c.0.weight", "E_i.0.weight"):
        out["encoder." + k] = out["decoder.hmpn." + k]
    return out
