"""The two ``PairVocab`` contracts the model reads (reference ggpm/vocab.py:22-58), for vocabularies given as index
pairs: ``size() -> (n_motif, n_attach)`` (encoder / decoder constructors) and the additive mask table
``mask[motif, attachment] = 0 if the attachment belongs to the motif else -1000`` with ``get_mask(cls_idx)``
(decoder cluster loss, ggpm/decoder.py:155-156).  The reference builds both from SMILES strings with rdkit; the
synthetic vocabularies of the benchmarks and fixtures are defined by an ``owner`` array instead.
"""
from __future__ import annotations

import numpy as np
import torch


class IndexPairVocab:
    def __init__(self, n_motif: int, n_attach: int, owner=None):
        if owner is None:                      # ggpm_amd.synth: attachment a of motif m is m * per + r
            assert n_attach % n_motif == 0, "give `owner` when attachments are not laid out per motif"
            owner = np.arange(n_attach) // (n_attach // n_motif)
        self.n = (n_motif, n_attach)
        self.owner = np.asarray(owner, dtype=np.int64)
        m = torch.zeros(n_motif, n_attach)
        m[torch.from_numpy(self.owner), torch.arange(n_attach)] = 1000.0
        self._mask = {"cpu": m - 1000.0}

    def size(self):
        return self.n

    def __getitem__(self, label):
        """(motif id, attachment id) of a label given as a pair of ids or of 'm<i>' / 'a<j>' strings."""
        f = lambda v: int(v[1:]) if isinstance(v, str) else int(v)
        return f(label[0]), f(label[1])

    @property
    def mask(self) -> torch.Tensor:
        return self._mask["cpu"]

    def mask_on(self, device) -> torch.Tensor:
        key = str(device)
        if key not in self._mask:
            self._mask[key] = self._mask["cpu"].to(device)
        return self._mask[key]

    def get_mask(self, cls_idx: torch.Tensor) -> torch.Tensor:
        return self.mask_on(cls_idx.device).index_select(0, cls_idx)
