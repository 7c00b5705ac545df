"""HierMPNDecoder -- the teacher-forced training forward of reference ggpm/decoder.py:19-301 (decode/beam search and
the rdkit assembly are inference-time chemistry and stay out of scope).

Same constructor, sub-module names and ``state_dict`` keys as the reference (``hmpn.*``, ``topoNN``, ``clsNN``,
``iclsNN``, ``matchNN``, ``W_assm``, ``W_root``, the aliases ``rnn_cell`` and ``E_assm``), same
``forward(mols, src_mol_vecs, graphs, tensors, orders) -> (loss, cls_acc, icls_acc, topo_acc, assm_acc)``.

The reference interleaves host bookkeeping (networkx look-ups, Python lists of prediction tuples) with device work on
every one of its ``maxt`` steps.  Here the bookkeeping is separated out: :class:`DecodeSchedule` derives every index
list of the loop (ggpm/decoder.py:186-259) from the tensorized batch, ``orders`` and two per-motif labels once per
batch, as integer arrays, and uploads them with ONE copy; the device loop then only slices that buffer.  Device work:
``IncHierMPNEncoder`` (``sparse_forward`` on the level kernels), the score heads on the library GEMM, ``enum_attach``
batched per step (one ``matchNN`` product for all candidates of a step), the losses in ``csrc/losses.hip``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import _dev
from . import functional as F_
from . import inc_encoder as IE
from .decoder_heads import ScoreHeads, bce_with_logits_sum, cross_entropy_sum

MAX_POS = 20


class DecodeSchedule:
    """Integer bookkeeping of ``HierMPNDecoder.forward`` for one tensorized batch (reference ggpm/decoder.py:186-259).

    Per step ``t`` (the reference's loop variable): ``subnode`` / ``submess`` (the ``subtree`` pair), ``atoms`` /
    ``bonds`` (the ``subgraph`` pair: what the previous ``update_graph_mask`` revealed), the topology predictions
    (``htree.node[xid]``, batch index, label), the cluster predictions made on messages, and the attachment
    predictions (``enum_attach`` arguments).  Built from plain arrays: the tree / graph tensors (host copies),
    ``orders``, and per tree node the attachment ids of its ``inter_label`` and its ``assm_cands``
    (``from_graphs`` reads those two from the reference's networkx batch).
    """

    def __init__(self):
        self._steps: Optional[List[dict]] = []
        self._native = None                  # schedule_native.NativeTables when the tables were built by csrc/schedule.hip
        self._labels = None
        self.plan: Optional[dict] = None
        self._atom_plan = None
        self.root_clab: List[int] = []
        self.root_ilab: List[int] = []
        self.max_cls_size = 0
        self.batch_size = 0
        self._dev = None

    def __getstate__(self):
        """Host tables only.  A natively built schedule converts itself: its tables become plain numpy copies (they are
        views into the library's buffers otherwise) and the atom plan keeps what it needs to serve the device path."""
        st = dict(self.__dict__)
        st["_dev"] = None          # (device views: rebuilt by to_device in the receiving process)
        if self._native is not None:
            nt = self._native
            st["_native"] = _FrozenTables(nt)
            st["plan"] = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in self.plan.items()}
        return st

    # ``steps``: the per-step lists of the reference's loop.  The numpy builder fills them as it goes; a natively built
    # schedule reconstructs them on first use (only the step-by-step fallback paths, the oracle and the tests read them).
    @property
    def steps(self) -> List[dict]:
        if self._steps is None:
            self._steps = self._steps_from_native()
        return self._steps

    def _steps_from_native(self) -> List[dict]:
        nt, B = self._native, self.batch_size
        g = lambda k: nt.get(k)
        ioff, aoff, boff = self.plan["inst_off"], self.plan["atom_off"], self.plan["bond_off"]
        soff, coff = g("submess_off").tolist(), g("cls_off").tolist()
        inter_icls, assm_cands = self._labels
        a_step, a_yid, a_nth, a_b = (g(k).tolist() for k in ("assm_step", "assm_yid", "assm_nth", "assm_bidx"))
        by_step: Dict[int, list] = {}
        for t, y, nth, i in zip(a_step, a_yid, a_nth, a_b):
            cands = np.asarray(assm_cands[int(y)], dtype=np.int64)
            by_step.setdefault(t, []).append((cands.reshape(len(cands), -1), tuple(int(a) for a in inter_icls[int(y)]), int(nth), int(i)))
        out = []
        for t in range(len(ioff) - 1):
            sl = lambda k, off: g(k)[off[t]:off[t + 1]].tolist()
            out.append(dict(subnode=sl("inst_node", ioff), submess=sl("submess_all", soff), atoms=sl("atoms_all", aoff),
                            bonds=sl("bonds_all", boff), topo_batch=sl("topo_batch", ioff), topo_label=sl("topo_label", ioff),
                            cls_mess=sl("cls_mess", coff), cls_batch=g("cls_batch")[B + coff[t]:B + coff[t + 1]].tolist(),
                            cls_clab=g("cls_clab")[B + coff[t]:B + coff[t + 1]].tolist(),
                            cls_ilab=g("cls_ilab")[B + coff[t]:B + coff[t + 1]].tolist(), assm=by_step.get(t, [])))
        return out

    # ------------------------------------------------------------------ construction (host)
    @staticmethod
    def from_graphs(graphs, tensors, orders, vocab, **kw) -> "DecodeSchedule":
        """From the reference's batch tuple: ``graphs = (tree_batchG, graph_batchG)`` (networkx, as produced by
        ``MolGraph.tensorize``).  Only the two labels that are not in the tensors are read from the node attributes."""
        tree_batch = graphs[0]
        from . import schedule_native as SN
        if kw.get("native") is not False and DecodeSchedule._native_wanted():
            # one pass over the nodes straight into the flat arrays the C++ builder reads (no per-node tuples / dicts)
            flat = SN.labels_from_graph(tree_batch, vocab, int(np.shape(tensors[0][0])[0]))
            if flat is not None:
                S = DecodeSchedule._build_native(tensors, orders, flat, kw.get("depth"), kw.get("gates"))
                if S is not None:
                    return S
        inter_icls, assm_cands = {}, {}
        for v, attr in tree_batch.nodes(data=True):
            cls = attr["smiles"]
            inter_icls[v] = tuple(vocab[(cls, icls)][1] for _, icls in attr["inter_label"])
            assm_cands[v] = list(attr["assm_cands"])
        return DecodeSchedule.from_tensors(tensors, orders, inter_icls, assm_cands, **kw)

    @staticmethod
    def _native_wanted() -> bool:
        from . import schedule_native as SN
        return bool(SN.enabled() and _dev.DECODER_BATCHED and _dev.ATOM_DECODE and _dev.ATOM_COMPACT)

    @staticmethod
    def _build_native(tensors, orders, labels, depth, gates) -> "Optional[DecodeSchedule]":
        """The schedule over tables built by csrc/schedule.hip, or None (the library declined the batch, or the batch is one
        of the degenerate ones that take the step-by-step forms: the numpy builder handles both)."""
        from . import schedule_native as SN
        nt = SN.build_tables(tensors, orders, labels, None, depth or 0, gates or 0)
        if nt is None:
            return None
        sc = nt.scalars()
        if not (sc["ok"] and sc["all_live"] and sc["E1"] > 1):
            return None
        inter_icls, assm_cands = labels.views() if isinstance(labels, SN.FlatLabels) else labels
        return DecodeSchedule._from_native(nt, tensors, inter_icls, assm_cands)

    @staticmethod
    def from_specs(specs, tensors, orders=None, **kw) -> "DecodeSchedule":
        """From synthetic molecules (ggpm_amd.synth.MolSpec) and their ``synth.tensorize`` output."""
        tree_scope, graph_scope = tensors[0][-1], tensors[1][-1]
        inter_icls, assm_cands = {}, {}
        if orders is None:
            orders = synth_orders(specs, tree_scope)
        for b, m in enumerate(specs):
            toff, aoff = tree_scope[b][0], graph_scope[b][0]
            for i in range(m.n_motifs):
                inter_icls[toff + i] = tuple(a for _, a in m.inter_label[i])
                assm_cands[toff + i] = [x + aoff for x in m.assm_cands[i]]
        return DecodeSchedule.from_tensors(tensors, orders, inter_icls, assm_cands, **kw)

    @staticmethod
    def from_tensors(tensors, orders, inter_icls: Dict[int, Tuple[int, ...]], assm_cands: Dict[int, list],
                     depth: Optional[int] = None, gates: Optional[int] = None, native: Optional[bool] = None
                     ) -> "DecodeSchedule":
        """``depth`` / ``gates`` (the decoder's diterG and 3 for GRU / 4 for LSTM), when known, let the native builder
        prepare the tables that depend on them as well; ``native=False`` forces the numpy builder (the checker)."""
        from . import schedule_native as SN
        if native is None:
            native = DecodeSchedule._native_wanted()
        if native:
            flat = SN.labels_from_dicts(inter_icls, assm_cands, int(np.shape(tensors[0][0])[0]))
            if flat is not None:
                nt = SN.build_tables(tensors, orders, flat, None, depth or 0, gates or 0)
                if nt is not None:
                    sc = nt.scalars()
                    if sc["ok"] and sc["all_live"] and sc["E1"] > 1:      # (anything else takes the step-by-step forms)
                        return DecodeSchedule._from_native(nt, tensors, inter_icls, assm_cands)
        tree_tensors, graph_tensors = tensors
        host = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        tfnode, tfmess, cgraph = host(tree_tensors[0]), host(tree_tensors[1]), host(tree_tensors[4])
        tagraph, tbgraph = host(tree_tensors[2]), host(tree_tensors[3])
        gfmess = host(graph_tensors[1])
        gagraph, gbgraph = host(graph_tensors[2]).astype(np.int64), host(graph_tensors[3]).astype(np.int64)
        bond_live = np.zeros(gfmess.shape[0], dtype=bool)     # hgraph.emask as the loop evolves it
        g_ag, g_bg = [], []
        tree_scope = tree_tensors[-1]
        S = DecodeSchedule()
        B = S.batch_size = len(orders)
        E1 = tfmess.shape[0]
        mess_time = np.full(E1, -1, dtype=np.int64)          # step at which a tree message is computed (each exactly once)
        mess_inst = np.zeros(E1, dtype=np.int64)             # the (step, node) visit whose vector is its input
        inst_node, inst_step, pool_rows = [], [], []
        revealed = set()
        tmess = {(int(u), int(v)): e for e, (u, v) in enumerate(tfmess[:, :2]) if e > 0}
        gadj: Dict[int, List[Tuple[int, int]]] = {}
        for e in range(1, gfmess.shape[0]):
            gadj.setdefault(int(gfmess[e, 0]), []).append((int(gfmess[e, 1]), e))
        cluster = lambda v: [int(a) for a in cgraph[v] if a > 0]

        def reveal(new_atoms):          # update_graph_mask, ggpm/decoder.py:85-100: the bonds induced by the new atoms
            aset = set(new_atoms)
            return list(new_atoms), [e for z in new_atoms for (n, e) in gadj.get(z, ()) if n in aset]

        new_atoms: List[int] = []
        for i in range(B):
            root = tree_scope[i][0]
            S.root_clab.append(int(tfnode[root, 0]))
            S.root_ilab.append(int(tfnode[root, 1]))
            new_atoms.extend(cluster(root))
        subgraph = reveal(new_atoms)
        S.max_cls_size = 2 * int(cgraph.shape[1])          # max(len(cluster) * 2), ggpm/decoder.py:199
        maxt = max(len(x) for x in orders)
        for t in range(maxt):
            st = dict(subnode=[], submess=[], atoms=subgraph[0], bonds=subgraph[1], topo_batch=[], topo_label=[],
                      cls_mess=[], cls_batch=[], cls_clab=[], cls_ilab=[], assm=[])
            batch_list = [i for i in range(B) if t < len(orders[i])]
            revealed.update(subgraph[0])
            # apply_graph_mask + get_sub_tensor of this step (ggpm/decoder.py:79-83, encoder.py:195-206), on the host:
            # the rows of the step's atoms / bonds with the entries that are revealed by now
            bond_live[np.asarray(subgraph[1], dtype=np.int64)] = True
            ra, rb = gagraph[np.asarray(subgraph[0], dtype=np.int64)], gbgraph[np.asarray(subgraph[1], dtype=np.int64)]
            g_ag.append(np.where(bond_live[ra], ra, 0))
            g_bg.append(np.where(bond_live[rb], rb, 0))
            for i in batch_list:
                xid, yid, tlab = orders[i][t]
                st["subnode"].append(int(xid))
                pool_rows.append([int(a) if (a > 0 and int(a) in revealed) else 0 for a in cgraph[xid]])
                inst_node.append(int(xid))
                inst_step.append(t)
                if yid is not None:
                    m = tmess[(int(xid), int(yid))]
                    st["submess"].append(m)
                    mess_time[m], mess_inst[m] = t, len(inst_node) - 1
            new_atoms = []
            for i in batch_list:
                xid, yid, tlab = orders[i][t]
                st["topo_batch"].append(i)
                st["topo_label"].append(int(tlab))
                if yid is not None:
                    new_atoms.extend(cluster(yid))          # "regardless of tlab", ggpm/decoder.py:230
                if tlab == 0:
                    continue
                st["cls_mess"].append(tmess[(int(xid), int(yid))])
                st["cls_batch"].append(i)
                st["cls_clab"].append(int(tfnode[yid, 0]))
                st["cls_ilab"].append(int(tfnode[yid, 1]))
                if len(cluster(xid)) > 2:                   # attachment is ambiguous only inside a ring
                    nth_child = int(tfmess[tmess[(int(yid), int(xid))], 2])
                    cands = np.asarray(assm_cands[int(yid)], dtype=np.int64)
                    icls = tuple(int(a) for a in inter_icls[int(yid)])
                    cands = cands.reshape(len(cands), -1)
                    assert cands.shape[1] == len(icls) and len(cands) <= S.max_cls_size
                    st["assm"].append((cands, icls, nth_child, i))
            subgraph = reveal(new_atoms)
            S.steps.append(st)
        S._level_plan(tfnode, tfmess, tagraph, tbgraph, tree_scope, mess_time, mess_inst, inst_node, inst_step, pool_rows)
        S.plan.update(g_agraph=np.concatenate(g_ag).reshape(-1, gagraph.shape[1]),
                      g_bgraph=np.concatenate(g_bg).reshape(-1, gbgraph.shape[1]),
                      atoms_all=np.asarray([a for st in S.steps for a in st["atoms"]], dtype=np.int64),
                      bonds_all=np.asarray([b for st in S.steps for b in st["bonds"]], dtype=np.int64),
                      atom_off=np.cumsum([0] + [len(st["atoms"]) for st in S.steps]).tolist(),
                      bond_off=np.cumsum([0] + [len(st["bonds"]) for st in S.steps]).tolist())
        return S

    @staticmethod
    def _from_native(nt, tensors, inter_icls, assm_cands) -> "DecodeSchedule":
        """The schedule object over tables built by csrc/schedule.hip (numpy views into the library's buffers)."""
        from .atom_decode import AtomPlan
        tree_tensors, graph_tensors = tensors
        sc, g = nt.scalars(), nt.get
        S = DecodeSchedule()
        S._native, S._steps, S._labels = nt, None, (inter_icls, assm_cands)
        S.batch_size, S.max_cls_size = sc["B"], sc["max_cls_size"]
        S.root_clab, S.root_ilab = g("root_clab").tolist(), g("root_ilab").tolist()
        Kt, At, C = tree_tensors[3].shape[1], tree_tensors[2].shape[1], tree_tensors[4].shape[1]
        Ag, Kg = graph_tensors[2].shape[1], graph_tensors[3].shape[1]
        S.plan = dict(chain=sc["chain"], n_inst=sc["n_inst"], E1=sc["E1"], all_live=bool(sc["all_live"]),
                      inst_motif=g("inst_motif"), inst_attach=g("inst_attach"), mess_inst=g("mess_inst"), mess_pos=g("mess_pos"),
                      dag_tree=g("dag_tree").reshape(-1, Kt), dag_inter=g("dag_inter").reshape(-1, Kt),
                      in_tree=g("in_tree").reshape(-1, At), in_inter=g("in_inter").reshape(-1, At),
                      pool=g("pool").reshape(-1, C), cls_mess=g("cls_mess"), inst_off=g("inst_off").tolist(),
                      g_agraph=g("g_agraph").reshape(-1, Ag), g_bgraph=g("g_bgraph").reshape(-1, Kg),
                      atoms_all=g("atoms_all"), bonds_all=g("bonds_all"), atom_off=g("atom_off").tolist(),
                      bond_off=g("bond_off").tolist())
        S._atom_plan = AtomPlan.from_native(S, nt, sc["Ng1"], sc["Eg1"])
        return S

    def _level_plan(self, tfnode, tfmess, tagraph, tbgraph, tree_scope, mess_time, mess_inst, inst_node, inst_step,
                    pool_rows) -> None:
        """Index tables of the BATCHED form of the two tree-side levels (HierMPNDecoder.forward_batched).

        Teacher forcing fixes the whole schedule: every tree message (x -> y) is computed exactly once, at the step that
        traverses it, from predecessor messages that were computed at earlier steps and never change afterwards.  The
        loop over steps therefore evaluates, for these two levels, a feed-forward computation on a DAG: message m reads
        the predecessors p of the padded table with time(p) < time(m) (plus, on the motif level, the constant pseudo
        message that carries its molecule's root vector).  Synchronous message passing on that DAG from h = 0 -- the
        encoder's own level kernels -- reaches exactly those values after `chain` iterations (chain = longest path).
        Node vectors exist once per VISIT (step, node): a visit sums the incoming messages revealed up to its step."""
        E1, B = tfmess.shape[0], self.batch_size
        dec_ag, dec_bg = tagraph.astype(np.int64).copy(), tbgraph.astype(np.int64).copy()
        for i, (root, _) in enumerate(tree_scope):           # init_decoder_state, ggpm/decoder.py:108-115
            dec_ag[root, -1] = E1 + i
            dec_bg[(tfmess[:, 0] == root) & (np.arange(E1) > 0), -1] = E1 + i
        live = mess_time >= 0
        live[0] = False
        mt = np.where(mess_time >= 0, mess_time, np.iinfo(np.int64).max)

        def dag(table, pseudo):
            t = table[1:]
            keep = (t > 0) & (t < E1) & (mt[np.minimum(t, E1 - 1)] < mt[1:, None])
            if pseudo:
                keep |= t >= E1
            return np.where(keep, t, 0)

        def incoming(table, pseudo):
            t = table[np.asarray(inst_node, dtype=np.int64)]
            keep = (t > 0) & (t < E1) & (mt[np.minimum(t, E1 - 1)] <= np.asarray(inst_step, dtype=np.int64)[:, None])
            if pseudo:
                keep |= t >= E1
            return np.where(keep, t, 0)

        dag_tree, dag_inter = dag(dec_bg, True), dag(tbgraph.astype(np.int64), False)
        chain = np.zeros(E1, dtype=np.int64)
        for m in np.argsort(mt[1:], kind="stable") + 1:      # longest dependency chain, in time order
            if mess_time[m] < 0:
                break
            preds = dag_inter[m - 1]
            chain[m] = 1 + (chain[preds[preds > 0]].max() if (preds > 0).any() else 0)
        nodes = np.asarray(inst_node, dtype=np.int64)
        self.plan = dict(
            chain=int(chain.max()) if E1 > 1 else 0, n_inst=len(inst_node), E1=int(E1), all_live=bool(live[1:].all()),
            inst_motif=tfnode[nodes, 0].astype(np.int64), inst_attach=tfnode[nodes, 1].astype(np.int64),
            mess_inst=mess_inst[1:], mess_pos=tfmess[1:, 2].astype(np.int64), dag_tree=dag_tree, dag_inter=dag_inter,
            in_tree=incoming(dec_ag, True), in_inter=incoming(tagraph.astype(np.int64), False),
            pool=np.asarray(pool_rows, dtype=np.int64).reshape(len(inst_node), -1),
            cls_mess=np.asarray([m for st in self.steps for m in st["cls_mess"]], dtype=np.int64),
            inst_off=np.cumsum([0] + [len(st["subnode"]) for st in self.steps]).tolist())

    # ------------------------------------------------------------------ flat views used by both the HIP path and the oracle
    def atom_plan(self, n_gnodes: int, n_gmess: int):
        """Index tables of the atom-level decode loop (ggpm_amd.atom_decode.AtomPlan), built once per batch."""
        if self._atom_plan is None:
            from .atom_decode import AtomPlan
            self._atom_plan = AtomPlan(self, n_gnodes, n_gmess)
        return self._atom_plan

    def topo(self):
        """(batch index, label) of every topology prediction in the reference's order (step major)."""
        return ([i for st in self.steps for i in st["topo_batch"]], [v for st in self.steps for v in st["topo_label"]])

    def cls(self):
        """(batch index, cluster label, attachment label) of every cluster prediction: the B roots first."""
        B = self.batch_size
        return (list(range(B)) + [i for st in self.steps for i in st["cls_batch"]],
                self.root_clab + [v for st in self.steps for v in st["cls_clab"]],
                self.root_ilab + [v for st in self.steps for v in st["cls_ilab"]])

    def assm_batch(self):
        return [i for st in self.steps for (_, _, _, i) in st["assm"]]

    def _native_to_device(self, device) -> "DecodeSchedule":
        """Two uploads (the int64 and the int32 pack of csrc/schedule.hip); every device table is a view into them."""
        nt = self._native
        d64, d32 = F_.upload(nt.packs[1], device), F_.upload(nt.packs[2], device)

        def view(name, shape=None):
            pack, off, cnt, el = nt.dir[name]
            v = (d64 if pack == 1 else d32)[off // el:off // el + cnt]
            return v.view(shape) if shape is not None else v

        P = self.plan
        plan = {k: view(k, P[k].shape) for k in ("inst_motif", "inst_attach", "mess_inst", "mess_pos", "dag_tree",
                                                  "dag_inter", "in_tree", "in_inter", "cls_mess", "atoms_all")}
        self._dev = dict(device=device, steps=None, n_assm=nt.scalars()["n_assm"], host=None, plan=plan, native=(d64, d32),
                         **{k: view(k) for k in ("topo_label", "cls_clab", "cls_ilab", "topo_batch32", "cls_batch32",
                                                 "assm_batch32")})
        if nt.has("pred_tree_rp"):
            # the CSRs, transposes and frozen masks of the two tree-side levels and the transposes the heads' backward sums
            # over came with the upload (csrc/schedule.hip): nothing is derived on the device for a batch seen for the first time
            sc = nt.scalars()
            E1, n_inst, B = sc["E1"], sc["n_inst"], sc["B"]
            iota = view("iota")

            def pair(rp, col, rows, ncols, rpT, colT):
                c, t = F_.CSR(rp, col, rows, ncols), F_.CSR(rpT, colT, ncols, rows)
                c._T, t._back = t, weakref.ref(c)
                return c

            def index_csr(idx, ncols, nameT):
                return pair(iota[:idx.numel() + 1], idx, idx.numel(), ncols, view(nameT + "_rp"), view(nameT + "_col"))

            def bytes_of(name):
                _, off, cnt, _ = nt.dir[name]
                return d32.view(torch.uint8)[off:off + cnt]

            src = index_csr(plan["mess_inst"], n_inst, "srcT")
            structs = {}
            for tag, n_extra in (("inter", 0), ("tree", B)):
                Etot = E1 + n_extra
                structs[tag] = (bytes_of("frozen_" + tag),
                                pair(view("pred_%s_rp" % tag), view("pred_%s_col" % tag), Etot, Etot,
                                     view("succ_%s_rp" % tag), view("succ_%s_col" % tag)),
                                pair(view("in_%s_rp" % tag), view("in_%s_col" % tag), n_inst, Etot,
                                     view("inT_%s_rp" % tag), view("inT_%s_col" % tag)), src)
            self._dev["level_structs"] = structs
            self._dev["head_csr"] = {k: index_csr(self._dev[k + "_batch32"], B, k + "T") for k in ("topo", "cls", "assm")}
        return self

    # ------------------------------------------------------------------ device copy (one upload)
    def to_device(self, device) -> "DecodeSchedule":
        """All index lists packed into one pinned int64 buffer, one asynchronous copy; per-step views of it."""
        if self._dev is not None and self._dev["device"] == device:
            return self
        if self._native is not None:
            return self._native_to_device(device)
        chunks: List[np.ndarray] = []
        where: List[Tuple[int, int]] = []

        fill = [0]

        def put(values) -> int:
            a = np.asarray(values, dtype=np.int64).reshape(-1)
            where.append((fill[0], len(a)))
            fill[0] += len(a)
            chunks.append(a)
            return len(where) - 1

        plan, pred = [], 0
        for st in self.steps:
            e = {k: put(st[k]) for k in ("subnode", "submess", "atoms", "bonds", "cls_mess")}
            groups: Dict[int, dict] = {}
            for (cands, icls, nth, i) in st["assm"]:
                g = groups.setdefault(len(icls), dict(atoms=[], icls=[], nth=[], dest=[]))
                n = len(cands)
                g["atoms"].extend(cands.reshape(-1).tolist())
                g["icls"].extend(list(icls) * n)
                g["nth"].extend([nth] * (n * len(icls)))
                g["dest"].extend(range(pred * self.max_cls_size, pred * self.max_cls_size + n))
                pred += 1
            e["assm"] = [(k, {n: put(v) for n, v in g.items()}) for k, g in sorted(groups.items())]
            plan.append(e)
        tb, tl = self.topo()
        cb, cc, ci = self.cls()
        ab = self.assm_batch()
        P = self.plan
        ptab = {k: (put(P[k]), P[k].shape) for k in ("inst_motif", "inst_attach", "mess_inst", "mess_pos", "dag_tree",
                                                      "dag_inter", "in_tree", "in_inter", "pool", "cls_mess", "g_agraph",
                                                      "g_bgraph", "atoms_all", "bonds_all")}
        tail = dict(topo_batch=put(tb), topo_label=put(tl), cls_batch=put(cb), cls_clab=put(cc), cls_ilab=put(ci),
                    assm_batch=put(np.repeat(np.asarray(ab, dtype=np.int64), self.max_cls_size)))
        flat = np.concatenate(chunks) if chunks else np.zeros(0, np.int64)
        hostbuf = None
        devbuf = F_.upload(flat, device)
        dev32 = devbuf.to(torch.int32)          # embedding ids are int32 for the gather kernels (one conversion per batch)
        view = lambda k: devbuf[where[k][0]:where[k][0] + where[k][1]]
        view32 = lambda k: dev32[where[k][0]:where[k][0] + where[k][1]]
        steps = []
        for e in plan:
            d = {k: view(e[k]) for k in ("subnode", "submess", "atoms", "bonds", "cls_mess")}
            d["assm"] = [(k, {n: (view32(v) if n == "icls" else view(v)) for n, v in g.items()}) for k, g in e["assm"]]
            steps.append(d)
        plan = {k: ((view32 if k in ("inst_motif", "inst_attach", "mess_inst", "mess_pos") else view)(i)).view(shape)
                for k, (i, shape) in ptab.items()}
        self._dev = dict(device=device, steps=steps, n_assm=len(ab), host=hostbuf, plan=plan, packs=(devbuf, dev32),
                         **{k: view(v) for k, v in tail.items()},
                         **{k + "32": view32(tail[k]) for k in ("topo_batch", "cls_batch", "assm_batch")})
        return self


def _note_stream(D: dict, stream) -> None:
    """The schedule's device tables (two tensors, everything else is a view) are about to be read on ``stream``, which is
    not the stream they were uploaded on: tell the allocator once per stream."""
    seen = D.setdefault("streams_seen", set())
    if stream.cuda_stream in seen:
        return
    seen.add(stream.cuda_stream)
    for t in D.get("native") or D.get("packs") or ():
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(stream)


class _FrozenTables:
    """What a pickled, natively built schedule carries instead of the library handle: plain numpy copies of the tables,
    the two device packs and the directory -- the same interface as schedule_native.NativeTables."""

    def __init__(self, nt):
        self.names, self.dir = list(nt.names), dict(nt.dir)
        self._arr = {k: np.array(nt.get(k)) for k in nt.names}
        self.packs = {k: np.array(v) for k, v in nt.packs.items()}
        self._scalars = nt.scalars()

    def get(self, name):
        return self._arr[name]

    def has(self, name):
        return name in self._arr

    def scalars(self):
        return dict(self._scalars)


def synth_orders(specs, tree_scope):
    """``orders`` as ``MolGraph.tensorize`` builds them (ggpm/mol_graph.py:225-231): per molecule the DFS order with
    the batch offset of its tree nodes."""
    out = []
    for m, (off, _) in zip(specs, tree_scope):
        out.append([(x + off, y + off, z) for x, y, z in m.order[:-1]] + [(m.order[-1][0] + off, None, 0)])
    return out


def _memo(D: dict, key: str, make):
    """``D[key]``, made on first use: tensors derived from a decode schedule's device tables live as long as the tables."""
    v = D.get(key)
    if v is None:
        v = D[key] = make()
    return v


def _accuracy(pred: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """get_accuracy (ggpm/nnutils.py:84-87) from the arg-max the loss kernel already produced."""
    return (pred.long() == labels).float().sum() / labels.numel()


class HierMPNDecoder(ScoreHeads):
    """reference ggpm/decoder.py:19-301 (training forward)"""

    def __init__(self, vocab, avocab, rnn_type, embed_size, hidden_size, latent_size, depthT, depthG, dropout,
                 attention=False):
        super().__init__(vocab, embed_size, hidden_size, latent_size, dropout)
        if attention:
            raise NotImplementedError("attention is off in every shipped configuration (ggpm/decoder.py:20)")
        self.avocab = avocab
        self.use_attention = False
        self.hmpn = IE.IncHierMPNEncoder(vocab, avocab, rnn_type, embed_size, hidden_size, depthT, depthG, dropout)
        self.rnn_cell = self.hmpn.tree_encoder.rnn          # aliases, as the reference registers them
        self.E_assm = self.hmpn.E_i
        if latent_size != hidden_size:
            self.W_root = nn.Linear(latent_size, hidden_size)

    def schedule_hints(self) -> dict:
        """What ``DecodeSchedule.from_*`` can use of this decoder: diterG and the number of gates of its message function,
        so that the native builder prepares the tables that depend on them in the same call."""
        from .rnn import LSTM
        rnn = self.hmpn.graph_encoder.rnn
        return dict(depth=rnn.depth, gates=4 if isinstance(rnn, LSTM) else 3)

    # ------------------------------------------------------------------ enum_attach, batched over one step
    def enum_attach_batched(self, hgraph_node, k: int, atoms, icls, nth) -> torch.Tensor:
        """``enum_attach`` (ggpm/decoder.py:286-301) for all predictions of one step whose candidates consist of ``k``
        atoms: ``matchNN([node[cand] | E_assm(icls) | onehot(nth_child)])``, summed over the ``k`` atoms of a candidate.
        ``atoms`` / ``nth`` int64, ``icls`` int32 (embedding ids, as the gather kernel reads them).
        -> [candidates, Hp] (zero pad columns)."""
        return self.enum_attach_rows(hgraph_node.index_select(0, atoms).contiguous(), k, icls, nth)

    def enum_attach_rows(self, cand, k: int, icls, nth) -> torch.Tensor:
        """As ``enum_attach_batched`` with the candidates' atom vectors already gathered (one row per candidate atom)."""
        H, He = self.hidden_size, self.embed_size
        emb = IE._embedding_rows(self.E_assm, icls)
        order = TF.one_hot(nth, MAX_POS).to(torch.float32)
        l1 = self.matchNN[0]
        vec = F_.linear([cand, emb, order], [H, He, MAX_POS], l1.weight, l1.bias, act=F_.ACT_RELU)
        return vec if k == 1 else vec.view(-1, k, vec.shape[1]).sum(dim=1)

    def enum_attach(self, hgraph, cands, icls, nth_child):
        """reference signature (one prediction): ``cands`` a list of atoms or of atom tuples, ``icls`` the attachment ids."""
        dev = hgraph.node.device
        c = torch.as_tensor(np.asarray(cands, dtype=np.int64).reshape(len(cands), -1), device=dev)
        k, n = c.shape[1], c.shape[0]
        ic = torch.as_tensor(list(icls) * n, dtype=torch.int32, device=dev)
        nth = torch.full((n * k,), int(nth_child), dtype=torch.long, device=dev)
        return self.enum_attach_batched(hgraph.node, k, c.reshape(-1), ic, nth)[:, :self.hidden_size]

    # ------------------------------------------------------------------ forward
    def forward(self, mols, src_mol_vecs, graphs, tensors, orders, schedule: Optional[DecodeSchedule] = None):
        tree_tensors, graph_tensors = tensors
        B, H, L = len(orders), self.hidden_size, self.latent_size
        dev = tree_tensors[0].device
        if schedule is None:
            schedule = DecodeSchedule.from_graphs(graphs, tensors, orders, self.vocab, **self.schedule_hints())
        D = schedule.to_device(dev)._dev
        src_root_vecs, src_tree_vecs, src_graph_vecs = src_mol_vecs
        if L == H:
            init_vecs = src_root_vecs
        else:
            init_vecs = F_.linear([src_root_vecs.contiguous()], [L], self.W_root.weight, self.W_root.bias)[:, :H]
        self._heads_in = None
        if _dev.DECODER_BATCHED and schedule.plan["all_live"] and schedule.plan["E1"] > 1:
            topo_vecs, cls_vecs, assm_vecs, assm_dest = self._states_batched(schedule, D, tree_tensors, graph_tensors,
                                                                            init_vecs)
        else:
            topo_vecs, cls_vecs, assm_vecs, assm_dest = self._states_stepwise(D, tree_tensors, graph_tensors, init_vecs)
        heads_in, self._heads_in = self._heads_in, None
        if heads_in is not None and src_tree_vecs is src_graph_vecs:
            return self._losses_composite(schedule, D, src_tree_vecs, topo_vecs, cls_vecs, heads_in, B, dev)
        if heads_in is not None:                     # (distinct tree / graph context vectors: the op-by-op heads)
            cand, blocks = heads_in
            for b in blocks:
                assm_vecs.append(self.enum_attach_rows(cand[b.base:b.base + b.n], b.k, b.icls32, b.nth))
                assm_dest.append(b.dest)
        return self._losses(schedule, D, src_tree_vecs, src_graph_vecs, topo_vecs, cls_vecs, assm_vecs, assm_dest, B, dev)

    def _losses_composite(self, schedule, D, z, topo_vecs, cls_vecs, heads_in, B, dev):
        """The four heads, their losses and accuracies as ONE autograd node (ggpm_amd/heads_fused.py): the same launches as
        ``_losses`` issues through ~30 nodes (ggpm/decoder.py:136-164, 261-301)."""
        from . import heads_fused
        cand, blocks = heads_in
        spec = D.get("heads_spec")
        if spec is None:
            i32 = lambda t: t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()
            spec = D["heads_spec"] = dict(
                topo_idx=i32(D["topo_batch32"]), topo_y=_memo(D, "topo_label_f32", lambda: D["topo_label"].to(torch.float32)),
                cls_idx=i32(D["cls_batch32"]),
                cls_lab=_memo(D, "cls_clab32", lambda: D["cls_clab"].to(torch.int32).contiguous()),
                icls_lab=_memo(D, "cls_ilab32", lambda: D["cls_ilab"].to(torch.int32).contiguous()),
                cls_lab_raw=D["cls_clab"], icls_lab_raw=D["cls_ilab"], topo_lab_raw=D["topo_label"],
                n_assm=D["n_assm"], max_cls_size=schedule.max_cls_size,
                assm_idx=i32(D["assm_batch32"]) if D["n_assm"] > 0 else None,
                assm_lab=_memo(D, "assm_labels32", lambda: torch.zeros(max(D["n_assm"], 1), dtype=torch.int32, device=dev)),
                idx_csr=D.get("head_csr") or {})          # molecule -> its prediction rows, when the schedule's builder made them
        spec = dict(spec, assm_blocks=blocks)
        tv = topo_vecs if (topo_vecs.dim() == 2 and topo_vecs.stride(1) == 1 and topo_vecs.stride(0) % 4 == 0) else topo_vecs.contiguous()
        cv = cls_vecs if (cls_vecs.dim() == 2 and cls_vecs.stride(1) == 1 and cls_vecs.stride(0) % 4 == 0) else cls_vecs.contiguous()
        loss_sum, acc = heads_fused.heads_losses(self, spec, z, tv, cv, cand)
        return loss_sum / B, acc[0], acc[1], acc[2], acc[3]

    def _states_stepwise(self, D, tree_tensors, graph_tensors, init_vecs):
        """The reference's loop, step by step (ggpm/decoder.py:175-259): three incremental encoder calls per step."""
        dev = tree_tensors[0].device

        hmpn, rnn_cell = self.hmpn, self.rnn_cell
        inter_tensors = tree_tensors
        htree, tree_tensors = IE.init_decoder_state(rnn_cell, tree_tensors, init_vecs)
        izeros = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
        hinter = IE.HTuple(mess=rnn_cell.get_init_state(inter_tensors[1]), emask=izeros(inter_tensors[1].size(0)))
        hgraph = IE.HTuple(mess=rnn_cell.get_init_state(graph_tensors[1]), vmask=izeros(graph_tensors[0].size(0)),
                           emask=izeros(graph_tensors[1].size(0)))
        graph_tensors = hmpn.embed_graph(graph_tensors) + (graph_tensors[-1],)

        topo_vecs, cls_vecs = [], [init_vecs]
        assm_vecs, assm_dest = [], []
        for st in D["steps"]:
            hgraph.vmask[st["atoms"]] = 1
            hgraph.emask[st["bonds"]] = 1
            htree.emask[st["submess"]] = 1
            hinter.emask[st["submess"]] = 1
            cur_tree = IE.apply_tree_mask(tree_tensors, htree, hgraph)
            cur_inter = IE.apply_tree_mask(inter_tensors, hinter, hgraph)
            cur_graph = IE.apply_graph_mask(graph_tensors, hgraph)
            htree, hinter, hgraph = hmpn(cur_tree, cur_inter, cur_graph, htree, hinter, hgraph,
                                         (st["subnode"], st["submess"]), (st["atoms"], st["bonds"]))
            topo_vecs.append(htree.node.index_select(0, st["subnode"]))
            if st["cls_mess"].numel():
                cls_vecs.append(rnn_cell.get_hidden_state(htree.mess).index_select(0, st["cls_mess"]))
            for k, g in st["assm"]:
                assm_vecs.append(self.enum_attach_batched(hgraph.node, k, g["atoms"], g["icls"], g["nth"]))
                assm_dest.append(g["dest"])

        return torch.cat(topo_vecs, dim=0), torch.cat(cls_vecs, dim=0), assm_vecs, assm_dest

    def _level_states(self, rnn, h0, hmess, dag, depth):
        """All messages of one tree-side level at once: ``sparse_forward`` over every real message row with the
        time-ordered predecessor table, iterated ``depth`` = longest chain times (see DecodeSchedule._level_plan)."""
        E1 = dag.shape[0] + 1
        rows = getattr(dag, "_ggpm_rows", None)            # (the same object every step: F_._sparse_structure's key)
        if rows is None:
            rows = torch.arange(1, E1, dtype=torch.long, device=hmess.device)
            try:
                dag._ggpm_rows = rows
            except AttributeError:
                pass
        I, H = rnn.input_size, rnn.hidden_size
        if isinstance(h0, tuple):
            i, o, u, f = rnn.W_i[0], rnn.W_o[0], rnn.W[0], rnn.W_f[0]
            return F_.lstm_sparse(h0[0], h0[1], hmess, rows, dag, i.weight, i.bias, o.weight, o.bias, u.weight, u.bias,
                                  f.weight, f.bias, depth, I, H)
        return F_.gru_sparse(h0, hmess, rows, dag, rnn.W_z.weight, rnn.W_z.bias, rnn.W_r.weight, rnn.U_r.weight,
                             rnn.U_r.bias, rnn.W_h.weight, rnn.W_h.bias, depth, I, H)

    # ---- the atom level ahead of the encoder ---------------------------------------------------------------------------
    # Teacher forcing makes the decoder's atom level (and its backward) independent of the latent vector: it reads the
    # molecule's own bonds and the decoder's parameters only.  ``start_atom_level`` therefore issues it on its own stream
    # BEFORE the encoder runs (HierPropertyVAE.forward), so that two chains of small latency-bound launches share the GPU
    # instead of queueing behind each other; autograd runs a node's backward on the stream of its forward, so the backward
    # overlaps the encoder's backward the same way.  _dev.ATOM_AHEAD = False switches it off.
    _ATOM_STREAMS = {}

    def start_atom_level(self, schedule, tensors) -> bool:
        self._atom_ahead = None
        if schedule is None or not (_dev.ATOM_AHEAD and _dev.ATOM_DECODE and _dev.DECODER_BATCHED):
            return False
        tree_tensors, graph_tensors = tensors
        dev = tree_tensors[0].device
        if dev.type != "cuda" or not (schedule.plan["all_live"] and schedule.plan["E1"] > 1):
            return False
        ap = schedule.atom_plan(graph_tensors[0].size(0), graph_tensors[1].size(0))
        if not ap.ok:
            return False
        D = schedule.to_device(dev)._dev
        main = torch.cuda.current_stream(dev)
        side = self._ATOM_STREAMS.get(dev.index)
        if side is None:
            # high priority: this chain of small dependent launches is the step's critical path, the encoder beside it
            # has slack -- where both have a kernel waiting for CUs, this one goes first (_dev.ATOM_PRIORITY = False: default priority)
            prio = -1 if _dev.ATOM_PRIORITY else 0
            side = self._ATOM_STREAMS[dev.index] = torch.cuda.Stream(device=dev, priority=prio)
        side.wait_stream(main)
        from .atom_decode import compact_enabled
        with torch.cuda.stream(side):
            if compact_enabled():       # issued now; its autograd node is created at the join, behind the encoder's
                pre = self._atom_level(schedule, D, graph_tensors, prelaunch=True)
                self._atom_ahead = (schedule, None, None, side, None, pre)
            else:
                pooled_all, cand, _, finish = self._atom_level(schedule, D, graph_tensors, defer_finish=True)
                self._atom_ahead = (schedule, pooled_all, cand, side, finish, None)
        return True

    def _atom_level(self, schedule, D, graph_tensors, defer_finish: bool = False, prelaunch: bool = False):
        """(pooled cluster vectors of all visits, attachment-candidate atom vectors, plan[, finish]) through atom_decode;
        ``prelaunch``: -> the state ``atom_decode_node`` turns into the autograd node later."""
        from .atom_decode import atom_decode
        hmpn, T = self.hmpn, D["plan"]
        graph_emb = hmpn.embed_graph(graph_tensors)
        fnode_all = graph_emb[0].index_select(0, T["atoms_all"])
        ap = schedule.atom_plan(graph_tensors[0].size(0), graph_tensors[1].size(0))
        out = atom_decode(ap, hmpn.graph_encoder, graph_emb[0], graph_emb[1], fnode_all, defer_finish=defer_finish,
                          prelaunch=prelaunch)
        if prelaunch:
            return out
        return (out[0], out[1], ap) + tuple(out[2:])

    def _states_batched(self, schedule, D, tree_tensors, graph_tensors, init_vecs):
        """Same vectors as ``_states_stepwise`` with the two tree-side levels de-sequentialised: only the atom level
        (diterG interacting iterations per step) keeps the step loop; the attachment and motif levels are ONE call each
        over all their messages (a DAG in decode time, DecodeSchedule._level_plan) and ONE read-out over all visits."""
        hmpn, rnn_cell = self.hmpn, self.rnn_cell
        H, He, P, T = self.hidden_size, self.embed_size, schedule.plan, D["plan"]
        dev = tree_tensors[0].device
        Hp = F_.padded_hidden(H)
        izeros = lambda n: torch.zeros(n, dtype=torch.long, device=dev)
        n_gnodes = graph_tensors[0].size(0)
        hgraph = IE.HTuple(mess=rnn_cell.get_init_state(graph_tensors[1]))
        pooled, assm_vecs, assm_dest = [], [], []
        off, aoff, boff = P["inst_off"], P["atom_off"], P["bond_off"]
        ahead, self._atom_ahead = getattr(self, "_atom_ahead", None), None
        ap = schedule.atom_plan(n_gnodes, graph_tensors[1].size(0)) if _dev.ATOM_DECODE else None
        if ap is not None and ap.ok:                        # ---- atom level as ONE autograd node (atom_decode.py)
            if ahead is not None and ahead[0] is schedule:  # issued before the encoder on its own stream: join it here
                _, pooled_all, cand, side, finish, pre = ahead
                if pre is not None:                         # prelaunched: join the worker, read-out, autograd node NOW (on
                    from .atom_decode import atom_decode_node          # the level's stream: its backward runs there)
                    with torch.cuda.stream(side):
                        pooled_all, cand = atom_decode_node(pre)
                else:
                    finish()                                # (worker-issued step loop: join it, enqueue the read-out behind it)
                main = torch.cuda.current_stream(dev)
                main.wait_stream(side)
                pooled_all.record_stream(main); cand.record_stream(main)
                F_.mark("fwd: atom level joined")
            else:
                if ahead is not None:                       # (a stale ahead run of another schedule: let its loop drain)
                    stale = ahead[4] if ahead[5] is None else ahead[5]["state"]["finish"]
                    if stale is not None:
                        stale()
                pooled_all, cand, _ = self._atom_level(schedule, D, graph_tensors)
            meta = ap.to_device(dev)["meta"]

            def attach_rows():
                for k, base, n in ap.cand_blocks:
                    assm_vecs.append(self.enum_attach_rows(cand[base:base + n], k, meta[k]["icls"], meta[k]["nth"]))
                    assm_dest.append(meta[k]["dest"])

            from . import heads_fused
            self._heads_in = None
            if heads_fused.usable(self):       # enum_attach moves into the heads' one autograd node (heads_fused.py)
                self._heads_in = (cand, [heads_fused.AssmBlock(k, base, n, meta[k]["icls"], meta[k]["nth"], meta[k]["dest"])
                                         for k, base, n in ap.cand_blocks])
            else:
                attach_rows()
            steps = []
        else:
            # the masked sub-tensors of every step come from the schedule (host-built, one upload); the constant one-hot
            # feature rows of all steps are selected by one gather each
            graph_emb = hmpn.embed_graph(graph_tensors)
            fnode_all = graph_emb[0].index_select(0, T["atoms_all"])
            fmess_all = graph_emb[1].index_select(0, T["bonds_all"])
            steps = D["steps"]
        for t, st in enumerate(steps):                      # ---- atom level, step by step (fallback)
            if st["atoms"].numel() + st["bonds"].numel() > 0:
                sub = (fnode_all[aoff[t]:aoff[t + 1]], fmess_all[boff[t]:boff[t + 1]],
                       T["g_agraph"][aoff[t]:aoff[t + 1]], T["g_bgraph"][boff[t]:boff[t + 1]])
                hgraph.node, hgraph.mess = hmpn.graph_encoder(sub, hgraph.mess, n_gnodes, (st["atoms"], st["bonds"]))
            pooled.append(F_.segment_sum(hgraph.node, F_.csr_from_padded(T["pool"][off[t]:off[t + 1]], ncols=n_gnodes), H))
            for k, g in st["assm"]:
                assm_vecs.append(self.enum_attach_batched(hgraph.node, k, g["atoms"], g["icls"], g["nth"]))
                assm_dest.append(g["dest"])
        pooled = torch.cat(pooled, dim=0) if steps else pooled_all          # [visits, Hp]
        n_inst, E1, depth = P["n_inst"], P["E1"], max(P["chain"], 1)
        ld = (H + MAX_POS + 3) // 4 * 4
        src_csr = F_.csr_from_index(T["mess_inst"], ncols=n_inst)

        def messages(hnode):                                 # [hnode[visit of the message] | onehot(position)]
            return F_.tree_message_input(hnode, T["mess_inst"], src_csr, T["mess_pos"], H, MAX_POS, ld)[:, :H + MAX_POS]

        def readout(enc, hnode, state, table):               # W_o([visit vector | sum of the incoming messages revealed])
            hid = enc.rnn.get_hidden_state(state)
            nei = F_.segment_sum(hid, F_.csr_from_padded(table, ncols=hid.shape[0]), H)
            node = F_.linear([hnode, nei], [H, H], enc.W_o[0].weight, enc.W_o[0].bias, act=F_.ACT_RELU)
            return enc.W_o[2](node)

        fm = tree_tensors[1]
        from . import tree_decode as TD
        ie, te = hmpn.inter_encoder, hmpn.tree_encoder
        mods = (hmpn.E_i[1], hmpn.E_c[1], hmpn.W_i[2], hmpn.W_c[2], ie.W_o[2], te.W_o[2])
        prms = [q for m in (hmpn.E_i, hmpn.E_c, hmpn.W_i, hmpn.W_c, ie.W_o, te.W_o, ie.rnn, te.rnn) for q in m.parameters()]
        if TD.usable(mods, prms):
            # ---- both levels as ONE autograd node each (tree_decode.py): same arithmetic, a seventh of the host work
            specs = D.get("level_specs")
            if specs is None:
                B_ = init_vecs.shape[0]
                pre = D.get("level_structs") or {}
                specs = D["level_specs"] = (
                    TD.LevelSpec(T["inst_attach"], T["mess_inst"], T["mess_pos"], T["dag_inter"], T["in_inter"], E1, 0, depth,
                                 prebuilt=pre.get("inter")),
                    TD.LevelSpec(T["inst_motif"], T["mess_inst"], T["mess_pos"], T["dag_tree"], T["in_tree"], E1, B_, depth,
                                 prebuilt=pre.get("tree") if pre.get("tree") is not None and pre["tree"][1].rows == E1 + B_ else None))
            hinter_node, _ = TD.tree_level(specs[0], ie.rnn, hmpn.E_i, hmpn.W_i, ie.W_o, pooled, None)
            htree_node, hid_t = TD.tree_level(specs[1], te.rnn, hmpn.E_c, hmpn.W_c, te.W_o, hinter_node,
                                              init_vecs.contiguous())
            cls_vecs = torch.cat([init_vecs, hid_t[:, :H].index_select(0, T["cls_mess"])], dim=0)
            F_.mark("fwd: tree-side levels issued")
            return htree_node[:, :H], cls_vecs, assm_vecs, assm_dest
        # ---- attachment level (embed_sub_tree(is_inter_layer=True) + inter_encoder, ggpm/encoder.py:208-245)
        finput = IE._embedding_rows(hmpn.E_i, T["inst_attach"])
        hnode_i = hmpn.W_i[2](F_.linear([finput, pooled], [He, H], hmpn.W_i[0].weight, hmpn.W_i[0].bias, act=F_.ACT_RELU))
        h_i = self._level_states(hmpn.inter_encoder.rnn, rnn_cell.get_init_state(fm), messages(hnode_i), T["dag_inter"], depth)
        hinter_node = readout(hmpn.inter_encoder, hnode_i, h_i, T["in_inter"])
        # ---- motif level: the root vectors ride as B extra, frozen message rows (init_decoder_state, :102-122)
        finput = IE._embedding_rows(hmpn.E_c, T["inst_motif"])
        hnode_t = hmpn.W_c[2](F_.linear([finput, hinter_node], [He, H], hmpn.W_c[0].weight, hmpn.W_c[0].bias,
                                        act=F_.ACT_RELU))
        h_t = self._level_states(hmpn.tree_encoder.rnn, rnn_cell.get_init_state(fm, init_vecs), messages(hnode_t),
                                 T["dag_tree"], depth)
        htree_node = readout(hmpn.tree_encoder, hnode_t, h_t, T["in_tree"])
        cls_vecs = torch.cat([init_vecs, rnn_cell.get_hidden_state(h_t).index_select(0, T["cls_mess"])], dim=0)
        return htree_node[:, :H], cls_vecs, assm_vecs, assm_dest

    def _assm_head(self, schedule, D, src_graph_vecs, assm_vecs, assm_dest, dev, want_scores=False):
        """(attachment loss, accuracy) of the batch -- ggpm/decoder.py:252-254, 276-281"""
        H = self.hidden_size
        P, C = D["n_assm"], schedule.max_cls_size
        if P <= 0:
            return 0, (None if want_scores else 1)
        vec = torch.cat(assm_vecs, dim=0)
        # (index tables of a resident schedule: concatenated once per ordering -- the batched and the step-by-step forms list
        # the predictions in different orders -- and ONE entry, so a caller that rebuilds the pieces every step keeps nothing)
        key = tuple(id(t) for t in assm_dest)
        held = D.get("assm_dest_all")
        if held is None or held[0] != key:
            held = D["assm_dest_all"] = (key, torch.cat(assm_dest), list(assm_dest))      # (the pieces are kept: ids stay theirs)
        dest = held[1]
        buf = torch.zeros(P * C, vec.shape[1], dtype=torch.float32, device=dev)
        buf.index_copy_(0, dest, vec)                             # F.pad to max_cls_size rows, ggpm/decoder.py:252-254
        scores = self.get_assm_score(src_graph_vecs, D["assm_batch32"], buf.view(P, C, -1)[:, :, :H], rows_padded=buf)
        # "the label is always the first of assm_cands"
        labels = _memo(D, "assm_labels32", lambda: torch.zeros(P, dtype=torch.int32, device=dev))
        scores = scores.contiguous()
        assm_loss, _ = cross_entropy_sum(scores, labels)
        s = scores.detach()
        if want_scores:                 # (the caller forms all four accuracies in one launch)
            return assm_loss, s
        assm_acc = (s[:, 0] == s.max(dim=-1)[0]).float().sum() / P      # get_accuracy_sym
        return assm_loss, assm_acc

    def _losses(self, schedule, D, src_tree_vecs, src_graph_vecs, topo_vecs, cls_vecs, assm_vecs, assm_dest, B, dev):
        """The three batched heads and their losses / accuracies (ggpm/decoder.py:261-284)."""
        H = self.hidden_size
        topo_scores = self.get_topo_score(src_tree_vecs, D["topo_batch32"], topo_vecs)
        # the loss kernels read float targets / int32 labels: converted once per schedule, not once per step
        topo_loss = bce_with_logits_sum(topo_scores, _memo(D, "topo_label_f32", lambda: D["topo_label"].to(torch.float32)))
        cls_loss, cls_pred, icls_pred = self.cls_losses(
            src_tree_vecs, D["cls_batch32"], cls_vecs, _memo(D, "cls_clab32", lambda: D["cls_clab"].to(torch.int32).contiguous()),
            _memo(D, "cls_ilab32", lambda: D["cls_ilab"].to(torch.int32).contiguous()))
        labs = (D["topo_label"], D["cls_clab"], D["cls_ilab"])
        if topo_scores.is_cuda and len({t.dtype for t in labs}) == 1 and labs[0].dtype in (torch.int64, torch.int32):
            # the four accuracies in ONE launch (ggpm_head_accuracies) instead of ~19 elementwise / reduction launches
            assm_loss, assm_scores = self._assm_head(schedule, D, src_graph_vecs, assm_vecs, assm_dest, dev, want_scores=True)
            acc = F_.head_accuracies(cls_pred, D["cls_clab"], icls_pred, D["cls_ilab"], topo_scores.detach(), D["topo_label"],
                                     assm_scores)
            cls_acc, icls_acc, topo_acc, assm_acc = acc[0], acc[1], acc[2], acc[3]
        else:
            topo_acc = ((topo_scores.detach() >= 0).long() == D["topo_label"]).float().sum() / D["topo_label"].numel()
            cls_acc, icls_acc = _accuracy(cls_pred, D["cls_clab"]), _accuracy(icls_pred, D["cls_ilab"])
            assm_loss, assm_acc = self._assm_head(schedule, D, src_graph_vecs, assm_vecs, assm_dest, dev)
        loss = (topo_loss + cls_loss + assm_loss) / B
        return loss, cls_acc, icls_acc, topo_acc, assm_acc
