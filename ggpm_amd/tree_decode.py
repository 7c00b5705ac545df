"""One tree-side level of the teacher-forced decoder as ONE autograd node.

``HierMPNDecoder._states_batched`` evaluates the attachment level and the motif level of the decoder each as a chain of
seven coarse ops (reference: ``IncHierMPNEncoder.embed_sub_tree`` + ``IncMPNEncoder.forward``, ggpm/encoder.py:208-245,
165-179, called once per decode step from ggpm/decoder.py:201-222; de-sequentialised over the decode-time DAG, see
``DecodeSchedule._level_plan``):

    finput = E[ids]                               embedding rows of every visit
    hnode  = relu([finput | lower] W^T + b)       W_i / W_c           (``lower``: pooled atom vectors / attachment-level nodes)
    hmess  = [hnode[visit of the message] | onehot(position)]
    h      = sparse_forward(h0, hmess, all real messages, DAG, chain)  (GRU / LSTM message function)
    node   = relu([hnode | sum of the incoming messages revealed] W_o^T + b_o)

Through ``functional.py`` that is seven autograd nodes per level and direction, each with its own argument checks,
allocations and ctypes marshalling -- the full VAE step spends more host time issuing these small ops than the GPU needs
to run them.  ``tree_level`` is the same arithmetic in the same order on the same C entry points (results bit-identical to
the op-by-op path up to the summation order of two fused launches) issued from ONE ``torch.autograd.Function``: the gate
inputs are written straight into the rows they belong to (the op-by-op form scatters them), the parameter gradients
go through the same deferred contractions (``functional._defer_*``).  Used when dropout is inactive and the parameters can be published
(``functional.can_publish``); ``GGPM_TREE_COMPOSITE=0`` keeps the op-by-op path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _dev, _lib
from . import functional as F_

MAX_POS = 20


def enabled() -> bool:
    return _dev.TREE_COMPOSITE


class LevelSpec:
    """The index tables of one level of one schedule on one device (built once per schedule, see decoder.py)."""

    def __init__(self, ids, mess_inst, mess_pos, dag, in_table, E1: int, n_extra: int, depth: int, prebuilt=None):
        self.ids, self.mess_inst, self.mess_pos, self.dag, self.in_table = ids, mess_inst, mess_pos, dag, in_table
        self.E1, self.n_extra, self.depth = E1, n_extra, depth
        self.prebuilt = prebuilt          # the four structures below, when the schedule's builder made them (csrc/schedule.hip)
        self.rows = None if prebuilt is not None else torch.arange(1, E1, dtype=torch.long, device=dag.device)

    def structures(self):
        """(frozen mask, predecessor CSR, incoming CSR, message -> visit CSR): part of the schedule's upload, or derived on
        the device and memoised on the index tensors"""
        if self.prebuilt is not None:
            return self.prebuilt
        Etot = self.E1 + self.n_extra
        frozen, pred, _ = F_._sparse_structure(Etot, self.rows, self.dag)
        return (frozen, pred, F_.csr_from_padded(self.in_table, ncols=Etot),
                F_.csr_from_index(self.mess_inst, ncols=self.ids.numel()))


class _TreeLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S: LevelSpec, lstm: bool, H: int, He: int, lower, extra, emb, W, b, Wo, bo, *rp):
        lib = _lib.load()
        P, dev = F_._p, lower.device
        f32 = dict(dtype=torch.float32, device=dev)
        Hp, Hep = F_.padded_hidden(H), F_.padded_hidden(He)
        I = H + MAX_POS
        ldm = (I + 3) // 4 * 4
        E1, depth = S.E1, S.depth
        Etot, ms, n_inst = E1 + S.n_extra, E1 - 1, S.ids.numel()
        G = 4 if lstm else 3
        s = F_._stream()
        frozen, pred, in_csr, src_csr = S.structures()
        # 1-2. visit vectors: relu([E[ids] | lower] W^T + b)
        finput = torch.empty(n_inst, Hep, **f32)
        _lib.check(lib.ggpm_gather_rows(P(emb), F_._ld(emb), P(S.ids), n_inst, He, P(finput), Hep, 0, Hep, s), "gather_rows")
        hnode = torch.empty(n_inst, Hp, **f32)
        ldw = W.stride(0)
        F_.gemm_ksegments(1, n_inst, H, [finput, lower], [Hep, F_._ld(lower)], [W, W[:, He:]], [ldw, ldw], [He, H], hnode, Hp, Hp,
                          bias=b, act=F_.ACT_RELU)
        # 3. message inputs
        hmess = torch.empty(ms, ldm, **f32)
        _lib.check(lib.ggpm_gather_rows(P(hnode), Hp, P(S.mess_inst), ms, H, P(hmess), ldm, 0, 0, s), "gather_rows")
        _lib.check(lib.ggpm_onehot(P(S.mess_pos), ms, MAX_POS, P(hmess), ldm, H, ldm, s), "onehot")
        # 4. hoisted gate inputs, straight into the rows 1 .. E1-1 they belong to (row 0 / the extra rows: zero)
        if lstm:
            Wi, bi, Wog, bog, Wu, bu_, Wf, bf = rp
            gates = ((Wi, bi), (Wog, bog), (Wu, bu_), (Wf, bf))
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh = rp
            gates = ((Wz, bz), (Wr, None), (Wh, bh))
        X = torch.zeros(G, Etot, Hp, **f32)
        for k, (Wk, bk) in enumerate(gates):
            F_.gemm(0, 1, ms, H, I, hmess, ldm, Wk, Wk.stride(0), X[k][1:], Hp, Hp, bias=bk)
        # 5. start state: zero, the extra (frozen) rows carry `extra`
        hp = torch.zeros(Etot, Hp, **f32)
        if extra is not None:
            hp[E1:, :H] = extra
        save = True
        Hs = torch.empty(depth + 1, Etot, Hp, **f32)
        Qs = torch.empty(depth, Etot, Hp, **f32)
        St = torch.empty(5, depth, Etot, Hp, **f32)
        if lstm:
            cp = torch.zeros(Etot, Hp, **f32)
            Cs = torch.empty(depth + 1, Etot, Hp, **f32)
            wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H)), **f32)
            Wh = [w[:, I:] for w, _ in gates]
            _lib.check(lib.ggpm_lstm_sparse_forward(
                Etot, H, depth, P(hp), P(cp), P(frozen), P(X[0]), P(X[1]), P(X[2]), P(X[3]), P(Wh[0]), Wi.stride(0), P(Wh[1]),
                Wog.stride(0), P(Wh[2]), Wu.stride(0), P(Wh[3]), Wf.stride(0), P(pred.rowptr), P(pred.col), P(Hs), P(Cs), P(Qs),
                P(St[0]), P(St[1]), P(St[2]), P(St[3]), P(St[4]), P(wpack), int(save), s), "lstm_sparse_forward")
        else:
            Cs = None
            wpack = torch.empty(int(lib.ggpm_gru_pack_floats(H)), **f32)
            _lib.check(lib.ggpm_gru_sparse_forward(
                Etot, H, depth, P(hp), P(frozen), P(X[0]), P(X[1]), P(X[2]), P(Wz[:, I:]), Wz.stride(0), P(Ur), Ur.stride(0),
                P(bu), P(Wh[:, I:]), Wh.stride(0), P(pred.rowptr), P(pred.col), P(Hs), P(Qs), P(St[0]), P(St[1]), P(St[2]),
                P(St[3]), P(St[4]), P(wpack), int(save), s), "gru_sparse_forward")
        hid = Hs[depth]
        # 6. read-out of every visit
        nei = torch.empty(n_inst, Hp, **f32)
        F_._segment_sum_raw(hid, in_csr, H, nei)
        node = torch.empty(n_inst, Hp, **f32)
        ldo = Wo.stride(0)
        F_.gemm_ksegments(1, n_inst, H, [hnode, nei], [Hp, Hp], [Wo, Wo[:, H:]], [ldo, ldo], [H, H], node, Hp, Hp, bias=bo,
                          act=F_.ACT_RELU)
        ctx.S, ctx.meta = S, (lstm, H, He, extra is not None)
        ctx.save_for_backward(lower, node)          # (an input and an output: through autograd, so that no ctx -> output cycle forms)
        ctx.stash = (finput, hnode, hmess, X[G - 1] if lstm else X[1], Hs, Cs, Qs, St, nei)
        ctx.prm = (emb, W, b, Wo, bo) + tuple(rp)
        ctx.structs = (frozen, pred, in_csr, src_csr)
        ctx.set_materialize_grads(False)            # (an unused output arrives as None, not as a zero tensor to be added)
        return node, hid

    @staticmethod
    def backward(ctx, d_node, d_hid):
        lib = _lib.load()
        S = ctx.S
        lstm, H, He, has_extra = ctx.meta
        finput, hnode, hmess, Xg, Hs, Cs, Qs, St, nei = ctx.stash
        lower, node = ctx.saved_tensors
        ctx.stash = None
        emb, W, b, Wo, bo = ctx.prm[:5]
        rp = ctx.prm[5:]
        frozen, pred, in_csr, src_csr = ctx.structs
        P, dev = F_._p, node.device
        f32 = dict(dtype=torch.float32, device=dev)
        Hp, Hep = F_.padded_hidden(H), F_.padded_hidden(He)
        I = H + MAX_POS
        ldm = hmess.shape[1]
        E1, depth = S.E1, S.depth
        Etot, ms, n_inst = E1 + S.n_extra, E1 - 1, S.ids.numel()
        G = 4 if lstm else 3
        s = F_._stream()
        succ = pred.T
        # ---- read-out: dpre_o -> d(hnode), d(nei) -> d(final state)
        if d_node is None:
            d_node = torch.zeros(n_inst, Hp, **f32)
        d_node = d_node.contiguous()
        dpre_o = torch.empty(n_inst, Hp, **f32)
        _lib.check(lib.ggpm_act_backward(P(d_node), P(node), n_inst, H, Hp, F_.ACT_RELU, 0, P(dpre_o), s), "act_backward")
        d_hnode = torch.empty(n_inst, Hp, **f32)
        d_nei = torch.empty(n_inst, Hp, **f32)
        ldo = Wo.stride(0)
        F_.gemm_grouped(0, 0, n_inst, H, H, [
            dict(A=dpre_o, lda=Hp, B=Wo, ldb=ldo, C=d_hnode, ldc=Hp, n_pad=Hp),
            dict(A=dpre_o, lda=Hp, B=Wo[:, H:], ldb=ldo, C=d_nei, ldc=Hp, n_pad=Hp)])
        if d_hid is not None:
            dHD = d_hid.clone() if d_hid.is_contiguous() else d_hid.contiguous()
            acc = 1
        else:
            dHD = torch.empty(Etot, Hp, **f32)
            acc = 0
        in_T = in_csr.T
        _lib.check(lib.ggpm_segment_sum(P(d_nei), Hp, P(in_T.rowptr), P(in_T.col), Etot, H, P(dHD), Hp, acc, 0 if acc else Hp, s),
                   "segment_sum")
        # ---- the level
        dHin = torch.empty(Etot, Hp, **f32)
        dX = torch.empty(G, Etot, Hp, **f32)
        if lstm:
            Wi, bi, Wog, bog, Wu, bu_, Wf, bf = rp
            Ws = (Wi, Wog, Wu, Wf)
            dWs = [torch.empty(w.shape, **f32) for w in Ws]
            Wh = [w[:, I:] for w in Ws]
            dWh = [w[:, I:] for w in dWs]
            dCD = torch.zeros(Etot, Hp, **f32)
            dCin = torch.empty(Etot, Hp, **f32)
            wb = int(lib.ggpm_lstm_backward_workspace_bytes(Etot, H, depth))
            work = torch.empty((wb + 3) // 4, **f32)
            _lib.check(lib.ggpm_lstm_sparse_backward(
                Etot, H, depth, P(frozen), P(Xg), P(Wh[0]), Wi.stride(0), P(Wh[1]), Wog.stride(0), P(Wh[2]), Wu.stride(0),
                P(Wh[3]), Wf.stride(0), P(pred.rowptr), P(pred.col), P(succ.rowptr), P(succ.col), P(Hs), P(Cs), P(Qs), P(St[0]),
                P(St[1]), P(St[2]), P(St[3]), P(St[4]), P(dHD), P(dCD), P(dHin), P(dCin), P(dX[0]), P(dX[1]), P(dX[2]), P(dX[3]),
                P(dWh[0]), dWs[0].stride(0), P(dWh[1]), dWs[1].stride(0), P(dWh[2]), dWs[2].stride(0), P(dWh[3]),
                dWs[3].stride(0), P(work), work.numel() * 4, s), "lstm_sparse_backward")
            xw = [(Ws[k], dWs[k]) for k in range(4)]
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh = rp
            dWz, dWr, dUr, dWh_ = (torch.empty(Wz.shape, **f32), torch.empty(Wr.shape, **f32), torch.empty(H, H, **f32),
                                   torch.empty(Wh.shape, **f32))
            dbu = torch.empty(H, **f32)
            wb = int(lib.ggpm_gru_backward_workspace_bytes(Etot, H, depth))
            work = torch.empty((wb + 3) // 4, **f32)
            _lib.check(lib.ggpm_gru_sparse_backward(
                Etot, H, depth, P(frozen), P(Xg), P(Wz[:, I:]), Wz.stride(0), P(Ur), Ur.stride(0), P(Wh[:, I:]), Wh.stride(0),
                P(pred.rowptr), P(pred.col), P(succ.rowptr), P(succ.col), P(Hs), P(Qs), P(St[0]), P(St[1]), P(St[2]), P(St[3]),
                P(St[4]), P(dHD), P(dHin), P(dX[0]), P(dX[1]), P(dX[2]), P(dWz[:, I:]), dWz.stride(0), P(dUr), H, P(dbu),
                P(dWh_[:, I:]), dWh_.stride(0), P(work), work.numel() * 4, s), "gru_sparse_backward")
            xw = [(Wz, dWz), (Wr, dWr), (Wh, dWh_)]
        dXs = [dX[k][1:E1] for k in range(G)]          # the rows of the real messages: contiguous
        # input halves of the gate weights, gate biases
        dbs = []
        for k, (Wk, dWk) in enumerate(xw):
            F_.gemm(1, 0, H, I, ms, dXs[k], Hp, hmess, ldm, dWk, dWk.stride(0), I, splitk=True)
            dbs.append(F_.colsum(dXs[k], ms, H) if (lstm or k != 1) else None)       # (W_r has no bias)
        if lstm:
            pg = (dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2], dWs[3], dbs[3])
        else:
            pg = (dWz, dbs[0], dWr, dUr, dbu, dWh_, dbs[2])
        for q, g in zip(rp, pg):
            F_._defer_sum(q, g)
        # ---- message inputs -> visit vectors
        dhmess = torch.empty(ms, ldm, **f32)
        F_.gemm_ksegments(0, ms, I, dXs, [Hp] * G, [w for w, _ in xw], [w.stride(0) for w, _ in xw], [H] * G, dhmess, ldm, ldm)
        src_T = src_csr.T
        _lib.check(lib.ggpm_segment_sum(P(dhmess), ldm, P(src_T.rowptr), P(src_T.col), n_inst, H, P(d_hnode), Hp, 1, 0, s),
                   "segment_sum")
        dpre_w = torch.empty(n_inst, Hp, **f32)
        _lib.check(lib.ggpm_act_backward(P(d_hnode), P(hnode), n_inst, H, Hp, F_.ACT_RELU, 0, P(dpre_w), s), "act_backward")
        ldw = W.stride(0)
        d_finput = torch.empty(n_inst, Hep, **f32)
        F_.gemm(0, 0, n_inst, He, H, dpre_w, Hp, W, ldw, d_finput, Hep, Hep)
        d_lower = None
        if ctx.needs_input_grad[4]:
            d_lower = F_._empty_same_layout(lower)
            F_.gemm(0, 0, n_inst, H, H, dpre_w, Hp, W[:, He:], ldw, d_lower, F_._ld(d_lower), lower.shape[1])
        # ---- parameter gradients: one contraction per Linear / one scatter per table at the end of the pass
        F_._defer_linear(W, b, dpre_w, [finput, lower], (He, H))
        F_._defer_linear(Wo, bo, dpre_o, [hnode, nei], (H, H))
        F_._defer_gather(emb, He, d_finput, S.ids)
        d_extra = dHin[E1:, :H] if (has_extra and ctx.needs_input_grad[5]) else None
        return (None, None, None, None, d_lower, d_extra) + (None,) * len(ctx.prm)


class TreeLevelC(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_tree_level"""
    _fields_ = ([(k, ctypes.c_int) for k in ("lstm", "H", "He", "E1", "n_extra", "depth", "n_inst")] +
                [(k, ctypes.c_void_p) for k in ("ids", "mess_inst", "mess_pos", "frozen", "pred_rowptr", "pred_col",
                                                "succ_rowptr", "succ_col", "in_rowptr", "in_col", "inT_rowptr", "inT_col",
                                                "srcT_rowptr", "srcT_col")] +
                [("emb", ctypes.c_void_p), ("ld_emb", ctypes.c_int), ("W", ctypes.c_void_p), ("b", ctypes.c_void_p),
                 ("ld_w", ctypes.c_int), ("Wo", ctypes.c_void_p), ("bo", ctypes.c_void_p), ("ld_wo", ctypes.c_int),
                 ("gate_w", ctypes.c_void_p * 4), ("ld_gate", ctypes.c_int * 4), ("gate_b", ctypes.c_void_p * 4),
                 ("Ur", ctypes.c_void_p), ("bu", ctypes.c_void_p), ("ld_ur", ctypes.c_int),
                 ("lower", ctypes.c_void_p), ("ld_lower", ctypes.c_int), ("extra", ctypes.c_void_p), ("ld_extra", ctypes.c_int)])


class TreeLevelViews(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_tree_level_views"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("finput", "hnode", "hmess", "X", "hp", "cp", "Hs", "Cs", "Qs", "St", "wpack",
                                               "nei", "node")]


class TreeLevelGrads(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_tree_level_grads"""
    _fields_ = [("dgate_w", ctypes.c_void_p * 4), ("ld_dgate", ctypes.c_int * 4), ("dgate_b", ctypes.c_void_p * 4),
                ("dUr", ctypes.c_void_p), ("dbu", ctypes.c_void_p), ("dpre_w", ctypes.c_void_p), ("dpre_o", ctypes.c_void_p),
                ("d_finput", ctypes.c_void_p), ("d_lower", ctypes.c_void_p), ("ld_dlower", ctypes.c_int),
                ("n_pad_dlower", ctypes.c_int), ("dHin", ctypes.c_void_p)]


def _addr(t) -> int:
    return 0 if t is None else t.data_ptr()


class _TreeLevelNative(torch.autograd.Function):
    """``_TreeLevel`` with each direction as ONE call into csrc/tree_level.hip (ggpm_tree_level_forward / _backward): the same
    launches in the same order (bit-identical results), no per-launch ctypes marshalling and two allocations per direction."""

    @staticmethod
    def forward(ctx, S: LevelSpec, lstm: bool, H: int, He: int, lower, extra, emb, W, b, Wo, bo, *rp):
        lib = _lib.load()
        dev = lower.device
        Hp = F_.padded_hidden(H)
        frozen, pred, in_csr, src_csr = S.structures()
        succ, in_T, src_T = pred.T, in_csr.T, src_csr.T
        n_inst, Etot = S.ids.numel(), S.E1 + S.n_extra
        if lstm:
            Wi, bi, Wog, bog, Wu, bu_, Wf, bf = rp
            gw, gb, Ur, bu = (Wi, Wog, Wu, Wf), (bi, bog, bu_, bf), None, None
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh = rp
            gw, gb = (Wz, Wr, Wh), (bz, None, bh)
        L = TreeLevelC()
        L.lstm, L.H, L.He, L.E1, L.n_extra, L.depth, L.n_inst = int(lstm), H, He, S.E1, S.n_extra, S.depth, n_inst
        for k, t in (("ids", S.ids), ("mess_inst", S.mess_inst), ("mess_pos", S.mess_pos), ("frozen", frozen),
                     ("pred_rowptr", pred.rowptr), ("pred_col", pred.col), ("succ_rowptr", succ.rowptr), ("succ_col", succ.col),
                     ("in_rowptr", in_csr.rowptr), ("in_col", in_csr.col), ("inT_rowptr", in_T.rowptr), ("inT_col", in_T.col),
                     ("srcT_rowptr", src_T.rowptr), ("srcT_col", src_T.col)):
            setattr(L, k, _addr(t))
        L.emb, L.ld_emb = _addr(emb), F_._ld(emb)
        L.W, L.b, L.ld_w = _addr(W), _addr(b), W.stride(0)
        L.Wo, L.bo, L.ld_wo = _addr(Wo), _addr(bo), Wo.stride(0)
        for k, (w, bb) in enumerate(zip(gw, gb)):
            L.gate_w[k], L.ld_gate[k], L.gate_b[k] = _addr(w), w.stride(0), _addr(bb)
        L.Ur, L.bu, L.ld_ur = _addr(Ur), _addr(bu), (Ur.stride(0) if Ur is not None else 0)
        L.lower, L.ld_lower = _addr(lower), F_._ld(lower)
        L.extra, L.ld_extra = _addr(extra), (F_._ld(extra) if extra is not None else 0)
        n_saved = int(lib.ggpm_tree_level_saved_floats(ctypes.byref(L)))
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        V = TreeLevelViews()
        _lib.check(lib.ggpm_tree_level_forward(ctypes.byref(L), F_._p(saved), n_saved, ctypes.byref(V), F_._stream()),
                   "tree_level_forward")
        base = saved.data_ptr()

        def view(addr, rows, cols):
            off = (addr - base) // 4
            return saved[off:off + rows * cols].view(rows, cols)

        Hep = F_.padded_hidden(He)
        node = view(V.node, n_inst, Hp)
        hid = view(V.Hs, (S.depth + 1) * Etot, Hp)[S.depth * Etot:]
        ctx.S, ctx.meta = S, (lstm, H, He, extra is not None)
        ctx.save_for_backward(lower, saved)
        ctx.keep = (L, V, extra, (frozen, pred, succ, in_csr, in_T, src_csr, src_T))      # (the tables the descriptor names)
        ctx.views = (view(V.finput, n_inst, Hep), view(V.hnode, n_inst, Hp), view(V.nei, n_inst, Hp))
        ctx.prm = (emb, W, b, Wo, bo) + tuple(rp)
        ctx.set_materialize_grads(False)
        return node, hid

    @staticmethod
    def backward(ctx, d_node, d_hid):
        lib = _lib.load()
        S = ctx.S
        lstm, H, He, has_extra = ctx.meta
        lower, saved = ctx.saved_tensors
        L, V, extra, _tables = ctx.keep
        finput, hnode, nei = ctx.views
        emb, W, b, Wo, bo = ctx.prm[:5]
        rp = ctx.prm[5:]
        dev = saved.device
        f32 = dict(dtype=torch.float32, device=dev)
        Hp, Hep = F_.padded_hidden(H), F_.padded_hidden(He)
        n_inst, Etot, E1 = S.ids.numel(), S.E1 + S.n_extra, S.E1
        if lstm:
            Wi, bi, Wog, bog, Wu, bu_, Wf, bf = rp
            gw, gb = (Wi, Wog, Wu, Wf), (bi, bog, bu_, bf)
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh = rp
            gw, gb = (Wz, Wr, Wh), (bz, None, bh)
        dgw = [torch.empty(w.shape, **f32) for w in gw]
        dgb = [torch.empty(H, **f32) if bb is not None else None for bb in gb]
        dUr = torch.empty(H, H, **f32) if not lstm else None
        dbu = torch.empty(H, **f32) if not lstm else None
        rows = torch.empty(2, n_inst, Hp, **f32)
        dpre_w, dpre_o = rows[0], rows[1]
        d_finput = torch.empty(n_inst, Hep, **f32)
        dHin = torch.empty(Etot, Hp, **f32)
        d_lower = F_._empty_same_layout(lower) if ctx.needs_input_grad[4] else None
        g = TreeLevelGrads()
        for k in range(len(gw)):
            g.dgate_w[k], g.ld_dgate[k], g.dgate_b[k] = _addr(dgw[k]), dgw[k].stride(0), _addr(dgb[k])
        g.dUr, g.dbu = _addr(dUr), _addr(dbu)
        g.dpre_w, g.dpre_o, g.d_finput = _addr(dpre_w), _addr(dpre_o), _addr(d_finput)
        g.d_lower = _addr(d_lower)
        g.ld_dlower, g.n_pad_dlower = (F_._ld(d_lower), lower.shape[1]) if d_lower is not None else (0, 0)
        g.dHin = _addr(dHin)
        if d_node is not None:
            d_node = d_node.contiguous()
        if d_hid is not None:
            d_hid = d_hid.contiguous()
        wb = int(lib.ggpm_tree_level_work_bytes(ctypes.byref(L)))
        work = torch.empty((wb + 3) // 4, **f32)
        # the level's parameter gradients on the second stream (csrc/tree_level.hip): what flows on -- d_lower, on the
        # attachment level the input of the atom level's 2.8 ms backward chain -- is then not queued behind ~145 us of
        # contractions per level.  The deferred-gradient queue (functional._DEFER) sums them: on that same stream when it is
        # flushed early, on the main stream once it has waited for this one ("early") at the end of the pass.
        side = F_._side_stream(dev) if (F_.side_stream_enabled() and _dev.TREE_WGRADS_ASIDE) else None
        _lib.check(lib.ggpm_tree_level_backward(ctypes.byref(L), ctypes.byref(V), F_._p(d_node), F_._p(d_hid), ctypes.byref(g),
                                                F_._p(work), work.numel() * 4, F_._stream(),
                                                ctypes.c_void_p(side.cuda_stream) if side is not None else None),
                   "tree_level_backward")
        if side is not None:
            for t in (work, saved):        # read there after this node has returned
                t.record_stream(side)
        if lstm:
            pg = (dgw[0], dgb[0], dgw[1], dgb[1], dgw[2], dgb[2], dgw[3], dgb[3])
        else:
            pg = (dgw[0], dgb[0], dgw[1], dUr, dbu, dgw[2], dgb[2])
        for q, gr in zip(rp, pg):
            F_._defer_sum(q, gr)
        if side is not None:
            F_._DEFER["early"] = side      # the end-of-pass flush waits for the second stream before it reads what was queued
        F_._defer_linear(W, b, dpre_w, [finput, lower], (He, H))
        F_._defer_linear(Wo, bo, dpre_o, [hnode, nei], (H, H))
        F_._defer_gather(emb, He, d_finput, S.ids)
        d_extra = dHin[E1:, :H] if (has_extra and ctx.needs_input_grad[5]) else None
        ctx.keep = ctx.views = None
        return (None, None, None, None, d_lower, d_extra) + (None,) * len(ctx.prm)


def usable(modules, params) -> bool:
    """Dropout inactive on every module of the level, deferral on, parameters publishable."""
    if not enabled() or not F_.defer_wgrads_enabled():
        return False
    if any(m.training and m.p > 0 for m in modules):
        return False
    return F_.can_publish(*params)


def tree_level(S: LevelSpec, rnn, emb_seq, lin_seq, wo_seq, lower, extra: Optional[torch.Tensor]):
    """-> (node [n_inst, Hp], hidden state [E1 + extra rows, Hp]) of one tree-side decoder level."""
    from .rnn import LSTM
    lstm = isinstance(rnn, LSTM)
    if lstm:
        rp = (rnn.W_i[0].weight, rnn.W_i[0].bias, rnn.W_o[0].weight, rnn.W_o[0].bias, rnn.W[0].weight, rnn.W[0].bias,
              rnn.W_f[0].weight, rnn.W_f[0].bias)
    else:
        rp = (rnn.W_z.weight, rnn.W_z.bias, rnn.W_r.weight, rnn.U_r.weight, rnn.U_r.bias, rnn.W_h.weight, rnn.W_h.bias)
    emb = emb_seq[0].weight
    lower = lower if lower.stride(1) == 1 else lower.contiguous()
    node_fn = _TreeLevelNative if _dev.TREE_DRIVER else _TreeLevel      # (the Python composite stays as the checker)
    return node_fn.apply(S, lstm, rnn.hidden_size, emb.shape[1], lower, extra, emb, lin_seq[0].weight, lin_seq[0].bias,
                         wo_seq[0].weight, wo_seq[0].bias, *rp)
