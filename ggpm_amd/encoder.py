"""MPNEncoder / HierMPNEncoder -- drop-in for reference ggpm/encoder.py:8-157.

Same constructor signatures, parameter names and shapes (state_dict compatible), same
``forward(tree_tensors, graph_tensors) -> (hroot, hnode, hinter, hatom)`` on the ``MolGraph.tensorize()``
input layout after ``make_cuda``.  Underneath, every level runs as hand-written HIP kernels on gfx950.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F_
from .nnutils import read_hint, to_cuda
from .rnn import GRU, LSTM

NUM_BOND_TYPES = 4   # len(MolGraph.BOND_LIST), reference ggpm/mol_graph.py:14-15
MAX_POS = 20         # MolGraph.MAX_POS,        reference ggpm/mol_graph.py:16


class LevelGraph:
    """Device-side CSR view of one level's (fmess, agraph, bgraph[, cgraph]) tensors."""

    def __init__(self, fmess: torch.Tensor, agraph: torch.Tensor, bgraph: torch.Tensor,
                 cgraph: Optional[torch.Tensor] = None, n_lower: int = 0):
        self.E1, self.N1 = fmess.shape[0], agraph.shape[0]
        self.pred = F_.csr_from_padded(bgraph, ncols=self.E1)          # message -> predecessor messages
        self.agr = F_.csr_from_padded(agraph, ncols=self.E1)           # node -> incoming messages
        self.src = F_.extract_column(fmess, 0)                         # message -> source node
        self.src_csr = F_.csr_from_index(self.src, ncols=self.N1)
        self.attr0 = F_.extract_column(fmess, 2)                       # tree levels: position label
        self.cgr = F_.csr_from_padded(cgraph, ncols=n_lower) if cgraph is not None else None


class _PinnedRing:
    """Small ring of pinned int32 staging buffers: host lists (the ``scope`` roots) reach the GPU with an async
    copy, so building a batch never blocks the host behind the GPU work already queued (a pageable
    ``torch.tensor(list, device=...)`` copy waits for the stream and would serialise consecutive steps)."""

    def __init__(self, slots: int = 8):        # (every slot pins its buffer on first use: ~45 us of host time each)
        self.slots, self.bufs, self.events, self.i = slots, [None] * slots, [None] * slots, 0

    def upload(self, values, device) -> torch.Tensor:
        n = len(values)
        k = self.i
        self.i = (self.i + 1) % self.slots
        if self.events[k] is not None:
            self.events[k].synchronize()          # slot reuse: its previous copy must have been consumed
        if self.bufs[k] is None or self.bufs[k].numel() < n:
            self.bufs[k] = torch.empty(max(n, 64), dtype=torch.int32).pin_memory()
        self.bufs[k][:n] = torch.as_tensor(values, dtype=torch.int32)
        out = torch.empty(n, dtype=torch.int32, device=device)
        out.copy_(self.bufs[k][:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


_RING = _PinnedRing()


class PreparedBatch:
    """CSR structures of one tensorized batch; build once, reuse across forward calls of the same batch."""

    def __init__(self, tree_tensors, graph_tensors, roots: Optional[torch.Tensor] = None):
        tfnode, tfmess, tagraph, tbgraph, tcgraph, tscope = tree_tensors
        gfnode, gfmess, gagraph, gbgraph, gscope = graph_tensors
        self.graph = LevelGraph(gfmess, gagraph, gbgraph)
        self.tree = LevelGraph(tfmess, tagraph, tbgraph, tcgraph, n_lower=gfnode.shape[0])
        self.motif_id = F_.extract_column(tfnode, 0)
        self.attach_id = F_.extract_column(tfnode, 1)
        self.roots = roots if roots is not None else _RING.upload([st for st, _ in tscope], tfnode.device)
        self.root_csr = F_.csr_from_index(self.roots, ncols=tfnode.shape[0])
        self.motif_csr = self.attach_csr = None      # set by the encoder (they need the vocabulary sizes)

    def prefetch_backward_structures(self):
        """Transposed CSRs are only read by the backward: build them beside the forward, off the critical path."""
        F_.prefetch_transposes([self.graph.pred, self.graph.agr, self.tree.pred, self.tree.agr, self.tree.src_csr,
                                self.tree.cgr, self.root_csr, self.motif_csr, self.attach_csr])


class MPNEncoder(nn.Module):
    """reference ggpm/encoder.py:8-38"""

    def __init__(self, rnn_type, input_size, node_fdim, hidden_size, depth, dropout):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.node_fdim = node_fdim
        self.depth = depth
        self.W_o = nn.Sequential(nn.Linear(node_fdim + hidden_size, hidden_size), nn.ReLU(), nn.Dropout(dropout))
        if rnn_type == 'GRU':
            self.rnn = GRU(input_size, hidden_size, depth)
        elif rnn_type == 'LSTM':
            self.rnn = LSTM(input_size, hidden_size, depth)
        else:
            raise ValueError('unsupported rnn cell type ' + rnn_type)

    def forward_padded(self, fnode, fmess, agr: F_.CSR, pred: F_.CSR):
        """-> (node_hiddens [N1,Hp], h [E1,Hp], nei [N1,Hp]) with zero pad columns."""
        H = self.hidden_size
        h = self.rnn.forward_padded(fmess, pred)
        h = self.rnn.get_hidden_state(h)
        nei = F_.segment_sum(h, agr, H)
        node = F_.linear([fnode, nei], [self.node_fdim, H], self.W_o[0].weight, self.W_o[0].bias,
                         act=F_.ACT_RELU, zero_row0=True)
        node = self.W_o[2](node)        # Dropout (identity when p = 0 / eval)
        return node, h, nei

    def forward(self, fnode, fmess, agraph, bgraph):
        agr = agraph if isinstance(agraph, F_.CSR) else F_.csr_from_padded(agraph, ncols=fmess.shape[0])
        pred = bgraph if isinstance(bgraph, F_.CSR) else F_.csr_from_padded(bgraph, ncols=fmess.shape[0])
        node, h, _ = self.forward_padded(fnode, fmess, agr, pred)
        H = self.hidden_size
        return node[:, :H], h[:, :H]


class HierMPNEncoder(nn.Module):
    """reference ggpm/encoder.py:41-157"""

    def __init__(self, vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout):
        super().__init__()
        self.vocab = vocab
        self.hidden_size = hidden_size
        self.embed_size = embed_size
        self.dropout = dropout
        self.atom_size = atom_size = avocab.size()
        self.bond_size = bond_size = NUM_BOND_TYPES + MAX_POS

        self.E_c = nn.Sequential(nn.Embedding(vocab.size()[0], embed_size), nn.Dropout(dropout))
        self.E_i = nn.Sequential(nn.Embedding(vocab.size()[1], embed_size), nn.Dropout(dropout))
        self.W_c = nn.Sequential(nn.Linear(embed_size + hidden_size, hidden_size), nn.ReLU(), nn.Dropout(dropout))
        self.W_i = nn.Sequential(nn.Linear(embed_size * 2, hidden_size), nn.ReLU(), nn.Dropout(dropout))

        # constant one-hot tables: plain attributes (not in state_dict), kept for the decoder's tie_embedding
        self.E_a = to_cuda(torch.eye(atom_size))
        self.E_b = to_cuda(torch.eye(NUM_BOND_TYPES))
        self.E_apos = to_cuda(torch.eye(MAX_POS))
        self.E_pos = to_cuda(torch.eye(MAX_POS))

        self.W_root = nn.Sequential(nn.Linear(hidden_size * 2, hidden_size), nn.Tanh())
        self.tree_encoder = MPNEncoder(rnn_type, hidden_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)
        self.inter_encoder = MPNEncoder(rnn_type, hidden_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)
        self.graph_encoder = MPNEncoder(rnn_type, atom_size + bond_size, atom_size, hidden_size, depthG, dropout)

    def tie_embedding(self, other):
        self.E_c, self.E_i = other.E_c, other.E_i
        self.E_a, self.E_b = other.E_a, other.E_b

    #: dtype of the H x H gate products inside the depth loops of the one-call drivers: "f32" (default; the 1e-4 parity
    #: contract) or "bf16" (operands rounded to bf16, fp32 accumulate on v_mfma_f32_16x16x32_bf16 -- BASELINE configs[4];
    #: everything else stays fp32).  None: the environment variable GGPM_GATE_DTYPE, else "f32".
    gate_dtype = None

    # ------------------------------------------------------------------ embeddings (padded tensors)
    def _tree_mess(self, hnode, lvl: LevelGraph):
        H = self.hidden_size
        ld = (H + MAX_POS + 3) // 4 * 4
        return F_.tree_message_input(hnode, lvl.src, lvl.src_csr, lvl.attr0, H, MAX_POS, ld)

    def embed_graph_padded(self, graph_tensors):
        fnode, fmess = graph_tensors[0], graph_tensors[1]
        return F_.embed_graph(fnode, fmess, self.atom_size, NUM_BOND_TYPES, MAX_POS)

    def embed_graph(self, graph_tensors):
        """reference ggpm/encoder.py:119-126: (one-hot atoms, [atom | bond type | position] messages, agraph, bgraph)."""
        hnode, hmess = self.embed_graph_padded(graph_tensors)
        return (hnode[:, :self.atom_size], hmess[:, :self.atom_size + self.bond_size], graph_tensors[2],
                graph_tensors[3])

    def embed_inter_padded(self, prep: PreparedBatch, hatom):
        H, He = self.hidden_size, self.embed_size
        emb = self.E_i[0].weight
        ids = prep.attach_id
        finput = F_.gather_rows(emb, ids, prep.attach_csr, He, F_.padded_hidden(He))
        finput = self.E_i[1](finput)
        pooled = F_.segment_sum(hatom, prep.tree.cgr, H)
        hnode = F_.linear([finput, pooled], [He, H], self.W_i[0].weight, self.W_i[0].bias, act=F_.ACT_RELU)
        hnode = self.W_i[2](hnode)
        return hnode, self._tree_mess(hnode, prep.tree)

    def embed_tree_padded(self, prep: PreparedBatch, hinter):
        H, He = self.hidden_size, self.embed_size
        emb = self.E_c[0].weight
        ids = prep.motif_id
        finput = F_.gather_rows(emb, ids, prep.motif_csr, He, F_.padded_hidden(He))
        finput = self.E_c[1](finput)
        hnode = F_.linear([finput, hinter], [He, H], self.W_c[0].weight, self.W_c[0].bias, act=F_.ACT_RELU)
        hnode = self.W_c[2](hnode)
        return hnode, self._tree_mess(hnode, prep.tree)

    def embed_root_padded(self, prep: PreparedBatch, hnode_in, nei):
        """tanh(W_root [hnode_in[root], sum hmess[agraph[root]]]) -- reference ggpm/encoder.py:128-138."""
        H = self.hidden_size
        roots = prep.roots
        rcsr = prep.root_csr
        f = F_.gather_rows(hnode_in, roots, rcsr, H, F_.padded_hidden(H))
        n = F_.gather_rows(nei, roots, rcsr, H, F_.padded_hidden(H))
        return F_.linear([f, n], [H, H], self.W_root[0].weight, self.W_root[0].bias, act=F_.ACT_TANH)

    # ------------------------------------------------------------------ forward
    def _fused_ok(self, tree_tensors, graph_tensors) -> bool:
        """The one-call C++ driver covers both message functions (dropout included) on int64 device tensors."""
        from . import fused
        if not fused.enabled() or not isinstance(self.graph_encoder.rnn, (GRU, LSTM)):
            return False
        if type(self.tree_encoder) is not MPNEncoder or self.atom_size + 24 > 252:
            return False
        ts = list(tree_tensors[:5]) + list(graph_tensors[:4])
        return all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 for t in ts)

    def _fused_check(self):
        """The tensors whose identity invalidates the cached parameter list (tie_embedding / load into new modules)."""
        return (self.E_c[0].weight, self.E_i[0].weight)

    def forward_padded(self, tree_tensors, graph_tensors, prep: Optional[PreparedBatch] = None,
                       roots: Optional[torch.Tensor] = None):
        if prep is None and self._fused_ok(tree_tensors, graph_tensors):
            from . import fused
            if roots is None:
                roots = read_hint(tree_tensors[0], "ggpm_roots")         # make_cuda / DevicePrefetcher leave it there
                if roots is None or roots.numel() != len(tree_tensors[-1]):
                    roots = _RING.upload([st for st, _ in tree_tensors[-1]], tree_tensors[0].device)
            return fused.hier_encoder(self, tree_tensors, graph_tensors, roots)
        if prep is None:
            prep = PreparedBatch(tree_tensors, graph_tensors, roots)
        if prep.motif_csr is None:
            prep.motif_csr = F_.csr_from_index(prep.motif_id, ncols=self.E_c[0].weight.shape[0])
            prep.attach_csr = F_.csr_from_index(prep.attach_id, ncols=self.E_i[0].weight.shape[0])
        if torch.is_grad_enabled():
            prep.prefetch_backward_structures()
        hnode_a, hmess_a = self.embed_graph_padded(graph_tensors)
        hatom, _, _ = self.graph_encoder.forward_padded(hnode_a, hmess_a, prep.graph.agr, prep.graph.pred)

        hnode_i, hmess_i = self.embed_inter_padded(prep, hatom)
        hinter, _, _ = self.inter_encoder.forward_padded(hnode_i, hmess_i, prep.tree.agr, prep.tree.pred)

        hnode_t, hmess_t = self.embed_tree_padded(prep, hinter)
        hnode, _, nei = self.tree_encoder.forward_padded(hnode_t, hmess_t, prep.tree.agr, prep.tree.pred)

        hroot = self.embed_root_padded(prep, hnode_t, nei)
        return hroot, hnode, hinter, hatom

    def forward(self, tree_tensors, graph_tensors, prep: Optional[PreparedBatch] = None):
        H = self.hidden_size
        outs = self.forward_padded(tree_tensors, graph_tensors, prep)
        return tuple(o[:, :H] for o in outs)


class MotifEncoder(nn.Module):
    """reference ggpm/encoder.py:252-341 (tree-only models 'prop' / 'prop-opt', ggpm/opvnet.py:4-9).

    ``forward(tree_tensors) -> (root, node)``; one motif-level MPNEncoder on
    ``hmess = [E_i(attachment)[src] | onehot(position)]`` with the motif embedding as node feature
    (which, as in the reference, requires ``embed_size == hidden_size``).
    """

    def __init__(self, vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout):
        super().__init__()
        self.vocab = vocab
        self.hidden_size = hidden_size
        self.embed_size = embed_size
        self.dropout = dropout
        self.atom_size = avocab.size()
        self.bond_size = NUM_BOND_TYPES + MAX_POS
        self.E_c = nn.Sequential(nn.Embedding(vocab.size()[0], embed_size), nn.Dropout(dropout))
        self.E_i = nn.Sequential(nn.Embedding(vocab.size()[1], embed_size), nn.Dropout(dropout))
        self.W_root = nn.Sequential(nn.Linear(hidden_size * 2, hidden_size), nn.Tanh())
        self.E_a = to_cuda(torch.eye(self.atom_size))
        self.E_pos = to_cuda(torch.eye(MAX_POS))
        self.tree_encoder = MPNEncoder(rnn_type, hidden_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)

    def tie_embedding(self, other):
        self.E_c, self.E_i = other.E_c, other.E_i
        self.E_a = other.E_a

    def forward_padded(self, tree_tensors):
        H, He = self.hidden_size, self.embed_size
        tfnode, tfmess, tagraph, tbgraph = tree_tensors[:4]
        lvl = LevelGraph(tfmess, tagraph, tbgraph)
        motif_id, attach_id = F_.extract_column(tfnode, 0), F_.extract_column(tfnode, 1)
        roots = _RING.upload([st for st, _ in tree_tensors[-1]], tfnode.device)
        ec, ei = self.E_c[0].weight, self.E_i[0].weight
        hnode = F_.gather_rows(ec, motif_id, F_.csr_from_index(motif_id, ncols=ec.shape[0]), He, F_.padded_hidden(He))
        hnode = self.E_c[1](hnode)
        hatt = F_.gather_rows(ei, attach_id, F_.csr_from_index(attach_id, ncols=ei.shape[0]), He, F_.padded_hidden(He))
        hatt = self.E_i[1](hatt)
        ld = (H + MAX_POS + 3) // 4 * 4
        hmess = F_.tree_message_input(hatt, lvl.src, lvl.src_csr, lvl.attr0, H, MAX_POS, ld)
        node, _, nei = self.tree_encoder.forward_padded(hnode, hmess, lvl.agr, lvl.pred)
        rcsr = F_.csr_from_index(roots, ncols=hnode.shape[0])
        f = F_.gather_rows(hnode, roots, rcsr, H, F_.padded_hidden(H))
        n = F_.gather_rows(nei, roots, rcsr, H, F_.padded_hidden(H))
        root = F_.linear([f, n], [H, H], self.W_root[0].weight, self.W_root[0].bias, act=F_.ACT_TANH)
        return root, node

    def forward(self, tree_tensors):
        H = self.hidden_size
        root, node = self.forward_padded(tree_tensors)
        return root[:, :H], node[:, :H]
