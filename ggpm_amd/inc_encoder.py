"""Incremental encoders used inside decoder training -- drop-in for reference ggpm/encoder.py:160-249, 343-394.

``IncMPNEncoder.forward(tensors, h, num_nodes, subset)`` recomputes only the message rows ``subset[1]`` of the
state ``h`` (GRU/LSTM.sparse_forward on the level kernels) and the node vectors of ``subset[0]``;
``IncHierMPNEncoder`` / ``IncEncoder`` stack it over the atom / attachment / motif levels exactly as the
reference does, including its conventions (node buffers are rebuilt from zeros on every call, the pad row
is not masked on this path).  ``HTuple`` and the mask helpers are the decoder-side state the reference keeps
in ggpm/decoder.py:13-16,72-101; they are integer index plumbing and stay in torch.
"""
from __future__ import annotations

import torch

from . import functional as F_
from .encoder import MPNEncoder, HierMPNEncoder, MotifEncoder, MAX_POS


class HTuple:
    """reference ggpm/decoder.py:13-16"""

    def __init__(self, node=None, mess=None, vmask=None, emask=None):
        self.node, self.mess = node, mess
        self.vmask, self.emask = vmask, emask


def index_scatter(sub_data, all_data, index):
    """reference ggpm/nnutils.py:124-128: ``all_data`` with rows ``index`` replaced by ``sub_data``."""
    out = all_data.clone()
    out.index_copy_(0, index, sub_data)
    return out


def _masked(graph, mask):
    return graph * mask[graph]


def apply_tree_mask(tensors, cur, prev):
    """reference ggpm/decoder.py:72-77: keep agraph/bgraph entries of live messages, cgraph entries of live atoms."""
    fnode, fmess, agraph, bgraph, cgraph, scope = tensors
    return fnode, fmess, _masked(agraph, cur.emask), _masked(bgraph, cur.emask), _masked(cgraph, prev.vmask), scope


def apply_graph_mask(tensors, hgraph):
    """reference ggpm/decoder.py:79-83"""
    fnode, fmess, agraph, bgraph, scope = tensors
    return fnode, fmess, _masked(agraph, hgraph.emask), _masked(bgraph, hgraph.emask), scope


def init_decoder_state(rnn_cell, tree_tensors, src_root_vecs):
    """reference ggpm/decoder.py:103-124, from the tensors alone: one extra message row per molecule holds its
    root vector; the root's agraph row and the bgraph rows of the messages leaving the root point at it."""
    fmess, agraph, bgraph = tree_tensors[1], tree_tensors[2].clone(), tree_tensors[3].clone()
    num_mess, batch = fmess.shape[0], src_root_vecs.shape[0]
    roots = torch.tensor([st for st, _ in tree_tensors[-1]], dtype=torch.long, device=fmess.device)
    extra = torch.arange(num_mess, num_mess + batch, dtype=agraph.dtype, device=fmess.device)
    agraph[roots, -1] = extra
    owner = torch.zeros(tree_tensors[0].shape[0], dtype=agraph.dtype, device=fmess.device)
    owner[roots] = extra
    from_root = owner[fmess[:, 0]]
    from_root[0] = 0
    bgraph[:, -1] = torch.where(from_root > 0, from_root, bgraph[:, -1])
    htree = HTuple()
    htree.mess = rnn_cell.get_init_state(fmess, src_root_vecs)
    htree.emask = torch.cat([bgraph.new_zeros(num_mess), bgraph.new_ones(batch)], dim=0)
    return htree, list(tree_tensors[:2]) + [agraph, bgraph] + list(tree_tensors[4:])


class IncMPNEncoder(MPNEncoder):
    """reference ggpm/encoder.py:160-179"""

    def forward(self, tensors, h, num_nodes, subset):
        fnode, fmess, agraph, bgraph = tensors
        subnode, submess = subset
        H, Hp = self.hidden_size, F_.padded_hidden(self.hidden_size)
        if len(submess) > 0:
            h = self.rnn.sparse_forward(h, fmess, submess, bgraph)
        node_buf = torch.zeros(num_nodes, Hp, dtype=torch.float32, device=agraph.device)
        if len(subnode) > 0:
            hid = self.rnn.get_hidden_state(h)
            nei = F_.segment_sum(hid, F_.csr_from_padded(agraph.long(), ncols=hid.shape[0]), H)
            node = F_.linear([fnode, nei], [self.node_fdim, H], self.W_o[0].weight, self.W_o[0].bias,
                             act=F_.ACT_RELU)
            node = self.W_o[2](node)
            node_buf = node_buf.index_copy(0, subnode.long(), node)
        return node_buf[:, :H], h


def _get_sub_tensor(tensors, subset):
    subnode, submess = subset
    fnode, fmess, agraph, bgraph = tensors[:4]
    fnode, fmess = fnode.index_select(0, subnode), fmess.index_select(0, submess)
    agraph, bgraph = agraph.index_select(0, subnode), bgraph.index_select(0, submess)
    if len(tensors) == 6:
        return fnode, fmess, agraph, bgraph, tensors[4].index_select(0, subnode), tensors[-1]
    return fnode, fmess, agraph, bgraph, tensors[-1]


def _sub_tree_messages(hnode, subnode, fmess, num_nodes, H):
    """hmess = [node_buf[fmess[:,0]] | onehot(fmess[:,2])], node_buf = zeros with the subset's hnode scattered in
    (``hnode`` is the padded [ns, Hp] tensor the producing kernel wrote)."""
    node_buf = torch.zeros(num_nodes, hnode.shape[1], dtype=torch.float32, device=hnode.device)
    node_buf = node_buf.index_copy(0, subnode, hnode)
    src = F_.extract_column(fmess, 0)
    pos = F_.extract_column(fmess, 2)
    ld = (H + MAX_POS + 3) // 4 * 4
    out = F_.tree_message_input(node_buf, src, F_.csr_from_index(src, ncols=num_nodes), pos, H, MAX_POS, ld)
    return out[:, :H + MAX_POS]


def _embedding_rows(seq, ids):
    emb = seq[0].weight
    He = emb.shape[1]
    out = F_.gather_rows(emb, ids, F_.csr_from_index(ids, ncols=emb.shape[0]), He, F_.padded_hidden(He))
    return seq[1](out)


class IncHierMPNEncoder(HierMPNEncoder):
    """reference ggpm/encoder.py:182-249"""

    def __init__(self, vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout):
        super().__init__(vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout)
        self.tree_encoder = IncMPNEncoder(rnn_type, hidden_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)
        self.inter_encoder = IncMPNEncoder(rnn_type, hidden_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)
        self.graph_encoder = IncMPNEncoder(rnn_type, self.atom_size + self.bond_size, self.atom_size, hidden_size,
                                           depthG, dropout)
        del self.W_root

    def get_sub_tensor(self, tensors, subset):
        return _get_sub_tensor(tensors, subset)

    def embed_sub_tree(self, tree_tensors, hinput, subtree, is_inter_layer):
        subnode, submess = subtree
        H, He = self.hidden_size, self.embed_size
        num_nodes = tree_tensors[0].size(0)
        fnode, fmess, agraph, bgraph, cgraph, _ = self.get_sub_tensor(tree_tensors, subtree)
        if is_inter_layer:
            finput = _embedding_rows(self.E_i, F_.extract_column(fnode, 1))
            pooled = F_.segment_sum(hinput, F_.csr_from_padded(cgraph.long(), ncols=hinput.shape[0]), H)
            hnode = F_.linear([finput, pooled], [He, H], self.W_i[0].weight, self.W_i[0].bias, act=F_.ACT_RELU)
            hnode = self.W_i[2](hnode)
        else:
            finput = _embedding_rows(self.E_c, F_.extract_column(fnode, 0))
            hsel = hinput.index_select(0, subnode)
            hnode = F_.linear([finput, hsel], [He, H], self.W_c[0].weight, self.W_c[0].bias, act=F_.ACT_RELU)
            hnode = self.W_c[2](hnode)
        if len(submess) == 0:
            hmess = fmess
        else:
            hmess = _sub_tree_messages(hnode, subnode, fmess, num_nodes, H)
        return hnode[:, :H], hmess, agraph, bgraph

    def forward(self, tree_tensors, inter_tensors, graph_tensors, htree, hinter, hgraph, subtree, subgraph):
        num_tree_nodes = tree_tensors[0].size(0)
        num_graph_nodes = graph_tensors[0].size(0)
        if len(subgraph[0]) + len(subgraph[1]) > 0:
            sub_graph_tensors = self.get_sub_tensor(graph_tensors, subgraph)[:-1]   # graph tensors arrive embedded
            hgraph.node, hgraph.mess = self.graph_encoder(sub_graph_tensors, hgraph.mess, num_graph_nodes, subgraph)
        if len(subtree[0]) + len(subtree[1]) > 0:
            sub_inter_tensors = self.embed_sub_tree(inter_tensors, hgraph.node, subtree, is_inter_layer=True)
            hinter.node, hinter.mess = self.inter_encoder(sub_inter_tensors, hinter.mess, num_tree_nodes, subtree)
            sub_tree_tensors = self.embed_sub_tree(tree_tensors, hinter.node, subtree, is_inter_layer=False)
            htree.node, htree.mess = self.tree_encoder(sub_tree_tensors, htree.mess, num_tree_nodes, subtree)
        return htree, hinter, hgraph


class IncEncoder(MotifEncoder):
    """reference ggpm/encoder.py:343-394 (tree-only decoder state)"""

    def __init__(self, vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout):
        super().__init__(vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout)
        self.tree_encoder = IncMPNEncoder(rnn_type, embed_size + MAX_POS, hidden_size, hidden_size, depthT, dropout)
        del self.W_root

    def get_sub_tensor(self, tensors, subset):
        return _get_sub_tensor(tensors, subset)

    def embed_sub_tree(self, tree_tensors, subtree):
        subnode, submess = subtree
        num_nodes = tree_tensors[0].size(0)
        fnode, fmess, agraph, bgraph, cgraph, _ = self.get_sub_tensor(tree_tensors, subtree)
        hnode = _embedding_rows(self.E_c, F_.extract_column(fnode, 0))
        if len(submess) == 0:
            hmess = fmess
        else:
            hmess = _sub_tree_messages(hnode, subnode, fmess, num_nodes, self.hidden_size)
        return hnode[:, :self.hidden_size], hmess, agraph, bgraph

    def forward(self, tree_tensors, htree, subtree):
        num_tree_nodes = tree_tensors[0].size(0)
        if len(subtree[0]) + len(subtree[1]) > 0:
            sub_tree_tensors = self.embed_sub_tree(tree_tensors, subtree)
            htree.node, htree.mess = self.tree_encoder(sub_tree_tensors, htree.mess, num_tree_nodes, subtree)
        return htree
