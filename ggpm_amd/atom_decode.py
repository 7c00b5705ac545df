"""The atom level of the teacher-forced decoder as ONE autograd node.

Inside ``HierMPNDecoder.forward`` the reference calls ``IncHierMPNEncoder.forward`` once per decode step
(ggpm/decoder.py:201-222); its first block (ggpm/encoder.py:235-239) recomputes, on the atom level, the bond messages
among the atoms the previous step revealed (``diterG`` interacting iterations of ``sparse_forward``) and the vectors of
those atoms (``IncMPNEncoder.forward``, ggpm/encoder.py:165-179).  This level is the part of the decoder that is truly
sequential in the step index (the two tree-side levels are not, see ``DecodeSchedule._level_plan``).

Teacher forcing fixes every index of that loop in advance, so ``AtomPlan`` builds all of them on the host, once per
batch: per step the frozen mask, the masked predecessor CSR of the step's bonds and its transpose, the masked
incoming-message CSR of the step's atoms and its transpose, and -- composed with the scatter into the zeroed node
buffer the reference rebuilds at every step -- where each pooled cluster vector (``embed_sub_tree``) and each
attachment candidate (``enum_attach``) reads the step's atom vectors.  ``atom_decode`` then runs the whole loop as one
``torch.autograd.Function`` whose forward and backward call the C ABI directly (``ggpm_*_sparse_forward/backward``,
``ggpm_segment_sum``, ``ggpm_gemm*``): no per-step autograd graph, no per-step index kernels, the gate input
projections of ALL bonds computed once (they are depth- and step-invariant one-hot look-ups), and every weight gradient
formed once per call (the read-out's from the stacked rows of all steps, the input halves from the summed ``dX``).

Compact steps (default, GGPM_ATOM_COMPACT=0 switches them off): a step recomputes a few hundred of the level's thousands
of bond messages, so its ``sparse_forward`` runs on the step's COMPACT row set -- the step's bonds plus the frozen
older bonds they read, renumbered 0..n-1 with host-built local CSRs -- instead of on every row of the level with a
frozen mask: the rows are gathered from / scattered back into the level-wide state (``ggpm_gather_rows`` /
``ggpm_scatter_rows``), the depth kernels launch ~1/8 of the row tiles and the hidden-half weight-gradient
contractions shrink by the same factor.  Same arithmetic per row, so the results are bit-identical to the full-level
form except for the weight gradients' summation order.
"""
from __future__ import annotations

import ctypes
import os
from typing import List

import numpy as np
import torch

from . import _lib
from . import functional as F_


def _csr_from_lists(counts: np.ndarray, flat: np.ndarray):
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), flat.astype(np.int32)


def _transpose(rows: np.ndarray, cols: np.ndarray, ncols: int):
    """(row, col) entry lists -> CSR of the transpose (rows of the result = columns), entries ascending."""
    order = np.lexsort((rows, cols))
    return (np.concatenate([[0], np.cumsum(np.bincount(cols, minlength=ncols))]).astype(np.int32),
            rows[order].astype(np.int32))


class AtomPlan:
    """Host-built index tables of the atom-level decode loop (see the module docstring)."""

    def __init__(self, schedule, n_gnodes: int, n_gmess: int):
        P, steps = schedule.plan, schedule.steps
        self.T, self.N1, self.E1 = len(steps), n_gnodes, n_gmess
        self.ok = all(len(st["atoms"]) > 0 for st in steps)          # (the reference keeps a stale node buffer otherwise)
        ints: List[np.ndarray] = []
        self.where = {}
        fill = [0]

        def put(key, a):
            a = np.asarray(a, dtype=np.int32).reshape(-1)
            self.where[key] = (fill[0], len(a))
            fill[0] += len(a)
            ints.append(a)

        frozen = np.ones((self.T, n_gmess), dtype=np.uint8)
        frozen_loc: List[np.ndarray] = []                     # compact steps: mask over the step's local rows
        self.nloc, self.floc_off = [], [0]
        aoff, boff, ioff = P["atom_off"], P["bond_off"], P["inst_off"]
        self.aoff, self.ioff = aoff, ioff
        # candidate atoms of all steps, grouped by atoms-per-candidate k first, then by step: one contiguous block per k
        groups = sorted({len(icls) for st in steps for (_, icls, _, _) in st["assm"]})
        cand_pos = {k: [] for k in groups}
        cand_meta = {k: dict(icls=[], nth=[], dest=[]) for k in groups}
        per_step_cands = []
        pred_i = 0
        for t, st in enumerate(steps):
            bonds = np.asarray(st["bonds"], dtype=np.int64)
            atoms = np.asarray(st["atoms"], dtype=np.int64)
            frozen[t, bonds] = 0
            # predecessor CSR over all E1 rows (only this step's bonds have entries) and its transpose
            order = np.argsort(bonds, kind="stable")
            tab = P["g_bgraph"][boff[t]:boff[t + 1]][order]
            cnt = (tab > 0).sum(axis=1)
            counts = np.zeros(n_gmess, dtype=np.int64)
            counts[bonds[order]] = cnt
            rp, col = _csr_from_lists(counts, tab[tab > 0])
            put(("pred_rp", t), rp); put(("pred_col", t), col)
            rpT, colT = _transpose(np.repeat(bonds[order], cnt), tab[tab > 0], n_gmess)
            put(("succ_rp", t), rpT); put(("succ_col", t), colT)
            # the same two CSRs over the step's compact row set: its bonds + the frozen rows they read + the null row
            rows = np.union1d(np.union1d(bonds, tab[tab > 0]), [0]).astype(np.int64)
            n = len(rows)
            lpos = np.full(n_gmess, -1, dtype=np.int64)
            lpos[rows] = np.arange(n)
            lcounts = np.zeros(n, dtype=np.int64)
            lcounts[lpos[bonds[order]]] = cnt                  # (ascending global id = ascending local id)
            rp, col = _csr_from_lists(lcounts, lpos[tab[tab > 0]])
            put(("lpred_rp", t), rp); put(("lpred_col", t), col)
            rpT, colT = _transpose(np.repeat(lpos[bonds[order]], cnt), lpos[tab[tab > 0]], n)
            put(("lsucc_rp", t), rpT); put(("lsucc_col", t), colT)
            fl = np.ones(n, dtype=np.uint8)
            fl[lpos[bonds]] = 0
            frozen_loc.append(fl)
            self.nloc.append(n)
            self.floc_off.append(self.floc_off[-1] + (n + 15) // 16 * 16)
            put(("rows", t), rows)                             # local -> level row
            put(("live", t), np.where(fl == 0, rows, -1))      # ... of the recomputed rows only (-1: skip)
            # incoming messages of the step's atoms (rows local to the step) and the transpose (rows = messages)
            atab = P["g_agraph"][aoff[t]:aoff[t + 1]]
            acnt = (atab > 0).sum(axis=1)
            rp, col = _csr_from_lists(acnt, atab[atab > 0])
            put(("agr_rp", t), rp); put(("agr_col", t), col)
            rpT, colT = _transpose(np.repeat(np.arange(len(atoms)), acnt), atab[atab > 0], n_gmess)
            put(("agrT_rp", t), rpT); put(("agrT_col", t), colT)
            # pooled cluster vectors of the step's visits read the step's atom vectors (other atoms' rows are zero in
            # the node buffer the reference rebuilds every step)
            pos = np.full(n_gnodes, -1, dtype=np.int64)
            pos[atoms] = np.arange(len(atoms))
            ptab = P["pool"][ioff[t]:ioff[t + 1]]
            loc = np.where(ptab > 0, pos[ptab], -1)
            pcnt = (loc >= 0).sum(axis=1)
            rp, col = _csr_from_lists(pcnt, loc[loc >= 0])
            put(("pool_rp", t), rp); put(("pool_col", t), col)
            rpT, colT = _transpose(np.repeat(np.arange(len(ptab)), pcnt), loc[loc >= 0], len(atoms))
            put(("poolT_rp", t), rpT); put(("poolT_col", t), colT)
            here = []
            for (cands, icls, nth, _) in st["assm"]:
                k, n = len(icls), len(cands)
                start = len(cand_pos[k])
                cand_pos[k].extend(pos[cands.reshape(-1)].tolist())
                m = cand_meta[k]
                m["icls"].extend(list(icls) * n)
                m["nth"].extend([nth] * (n * k))
                m["dest"].extend(range(pred_i * schedule.max_cls_size, pred_i * schedule.max_cls_size + n))
                here.append((k, start, n * k))
                pred_i += 1
            per_step_cands.append(here)
        # flat candidate layout: block of k = 1 rows, then k = 2 ...
        base, self.cand_blocks = 0, []
        for k in groups:
            self.cand_blocks.append((k, base, len(cand_pos[k])))
            base += len(cand_pos[k])
        self.n_cand = base
        kbase = {k: b for k, b, _ in self.cand_blocks}
        self.step_cands = []                                  # per step: [(flat row offset, count)] + transposed scatter
        for t, here in enumerate(per_step_cands):
            segs = [(kbase[k] + start, n) for (k, start, n) in here]
            self.step_cands.append(segs)
            ns = aoff[t + 1] - aoff[t]
            for j, (k, start, n) in enumerate(here):
                p = np.asarray(cand_pos[k][start:start + n], dtype=np.int64)
                put(("cand_pos", t, j), p)
                ok = p >= 0
                rpT, colT = _transpose(np.arange(n)[ok], p[ok], ns)
                put(("candT_rp", t, j), rpT); put(("candT_col", t, j), colT)
        self.cand_meta = {k: {n: np.asarray(v, dtype=np.int64) for n, v in m.items()} for k, m in cand_meta.items()}
        self.ints = np.concatenate(ints) if ints else np.zeros(0, np.int32)
        self.frozen = frozen
        # stacked per-step blocks: depth * n rows (stashes) / (depth + 1) * n rows (states), see row_offsets()
        self.frozen_loc = np.ones(self.floc_off[-1], dtype=np.uint8)
        for t, fl in enumerate(frozen_loc):
            self.frozen_loc[self.floc_off[t]:self.floc_off[t] + len(fl)] = fl
        self._dev = None

    def row_offsets(self, depth: int):
        """-> (offsets of the steps' [depth, n] blocks, offsets of their [depth + 1, n] blocks), each of length T + 1."""
        n = np.asarray(self.nloc, dtype=np.int64)
        return (np.concatenate([[0], np.cumsum(depth * n)]).tolist(),
                np.concatenate([[0], np.cumsum((depth + 1) * n)]).tolist())

    def gate_rows(self, t: int, gates: int) -> np.ndarray:
        """Rows of the step's compact set inside the stacked [gates * E1, Hp] gate-input matrix."""
        off, n = self.where[("rows", t)]
        rows = self.ints[off:off + n].astype(np.int64)
        return np.concatenate([k * self.E1 + rows for k in range(gates)]).astype(np.int32)

    def to_device(self, device):
        if self._dev is None or self._dev["device"] != device:
            cuda = torch.device(device).type == "cuda"
            hi, hf = torch.from_numpy(self.ints), torch.from_numpy(self.frozen)
            hl = torch.from_numpy(self.frozen_loc)
            if cuda:
                hi, hf, hl = hi.pin_memory(), hf.pin_memory(), hl.pin_memory()
            di, df = hi.to(device, non_blocking=True), hf.to(device, non_blocking=True)
            dl = hl.to(device, non_blocking=True)
            base = di.data_ptr()
            ptr = {k: base + 4 * off for k, (off, n) in self.where.items()}
            meta = {k: {n: torch.from_numpy(v).to(device, non_blocking=True) for n, v in m.items()}
                    for k, m in self.cand_meta.items()}
            for m in meta.values():
                m["icls"] = m["icls"].to(torch.int32)
            self._dev = dict(device=device, ints=di, frozen=df, frozen_loc=dl, ptr=ptr, meta=meta, keep=(hi, hf, hl),
                             gate_rows={})
        return self._dev

    def gate_rows_device(self, gates: int):
        """Per step: device pointer of gate_rows(t, gates) (3 gates for the GRU, 4 for the LSTM); built on first use."""
        D = self._dev
        if gates not in D["gate_rows"]:
            parts = [self.gate_rows(t, gates) for t in range(self.T)]
            offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
            h = torch.from_numpy(np.concatenate(parts))
            if D["ints"].is_cuda:
                h = h.pin_memory()
            d = h.to(D["device"], non_blocking=True)
            D["gate_rows"][gates] = (d, h, [d.data_ptr() + 4 * int(o) for o in offs[:-1]])
        return D["gate_rows"][gates][2]


def compact_enabled() -> bool:
    return os.environ.get("GGPM_ATOM_COMPACT", "1") != "0"


def _vp(addr: int) -> ctypes.c_void_p:
    return ctypes.c_void_p(addr)


class _AtomDecode(torch.autograd.Function):
    """(pooled cluster vectors of all visits [n_inst, Hp], attachment-candidate atom vectors [n_cand, Hp])."""

    @staticmethod
    def forward(ctx, plan: AtomPlan, cell: str, depth: int, H: int, Fdim: int, I: int, fn_all, hmess, drop, *params):
        lib = _lib.load()
        dev = hmess.device
        D = plan.to_device(dev)
        ptr, P = D["ptr"], F_._p
        Hp = F_.padded_hidden(H)
        E1, T = plan.E1, plan.T
        f32 = dict(dtype=torch.float32, device=dev)
        lstm = cell == "LSTM"
        G = 4 if lstm else 3
        if lstm:
            Wi, bi, Wo_g, bo_g, Wu, bu_g, Wf, bf, Wout, bout = params
            gates = ((Wi, bi), (Wo_g, bo_g), (Wu, bu_g), (Wf, bf))
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh, Wout, bout = params
            gates = ((Wz, bz), (Wr, None), (Wh, bh))
        # hoisted gate input projections of ALL bond messages (step and depth invariant)
        X = torch.empty(G, E1, Hp, **f32)
        for k, (W, b) in enumerate(gates):
            F_.gemm(0, 1, E1, H, I, hmess, F_._ld(hmess), W, W.stride(0), X[k], Hp, Hp, bias=b)
        compact = compact_enabled()
        zero = torch.zeros(E1, Hp, **f32)
        if compact:
            # level-wide state, updated in place step by step; the stashes live per step at the step's own row count
            Hg, Cg = zero, (torch.zeros(E1, Hp, **f32) if lstm else None)
            xrows = plan.gate_rows_device(G)
            Xflat = X.view(G * E1, Hp)
            # the steps' stashes stacked row-wise, one buffer per kind: the backward contracts them in one go
            roff, qoff = plan.row_offsets(depth)
            Hs_all = torch.empty(qoff[-1], Hp, **f32)
            Cs_all = torch.empty(qoff[-1], Hp, **f32) if lstm else None
            Qs_all = torch.empty(roff[-1], Hp, **f32)
            St_all = torch.empty(5, roff[-1], Hp, **f32)
            Xl = []
        else:
            Hs = torch.empty(T, depth + 1, E1, Hp, **f32)
            Cs = torch.empty(T, depth + 1, E1, Hp, **f32) if lstm else None
            Qs = torch.empty(T, depth, E1, Hp, **f32)
            St = torch.empty(T, 5, depth, E1, Hp, **f32)
        ns_tot, n_inst = plan.aoff[-1], plan.ioff[-1]
        NODE = torch.empty(ns_tot, Hp, **f32)
        NEI = torch.empty(ns_tot, Hp, **f32)
        pooled = torch.empty(n_inst, Hp, **f32)
        cand = torch.zeros(max(plan.n_cand, 1), Hp, **f32)
        wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H) if lstm else lib.ggpm_gru_pack_floats(H)), **f32)
        s = F_._stream()
        frz = D["frozen"]
        frz_loc = D["frozen_loc"].data_ptr()
        ldF, ldwo = F_._ld(fn_all), Wout.stride(0)
        h_prev, c_prev = zero, zero

        def sparse_forward(n, h_in, c_in, fz, x, rp, col, hs, cs, qs, st):
            if lstm:
                _lib.check(lib.ggpm_lstm_sparse_forward(
                    n, H, depth, P(h_in), P(c_in), fz, P(x[0]), P(x[1]), P(x[2]), P(x[3]), P(Wi[:, I:]),
                    Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0), P(Wu[:, I:]), Wu.stride(0), P(Wf[:, I:]), Wf.stride(0),
                    rp, col, P(hs), P(cs), P(qs), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(wpack), 1, s),
                    "lstm_sparse_forward")
            else:
                _lib.check(lib.ggpm_gru_sparse_forward(
                    n, H, depth, P(h_in), fz, P(x[0]), P(x[1]), P(x[2]), P(Wz[:, I:]), Wz.stride(0), P(Ur),
                    Ur.stride(0), P(bu), P(Wh[:, I:]), Wh.stride(0), rp, col, P(hs), P(qs), P(st[0]), P(st[1]), P(st[2]),
                    P(st[3]), P(st[4]), P(wpack), 1, s), "gru_sparse_forward")

        for t in range(T):
            a0, a1, i0, i1 = plan.aoff[t], plan.aoff[t + 1], plan.ioff[t], plan.ioff[t + 1]
            ns, ni = a1 - a0, i1 - i0
            if compact:
                n = plan.nloc[t]
                rows = _vp(ptr[("rows", t)])
                h_in = torch.empty(n, Hp, **f32)
                _lib.check(lib.ggpm_gather_rows(P(Hg), Hp, rows, n, Hp, P(h_in), Hp, 0, 0, s), "gather_rows")
                c_in = None
                if lstm:
                    c_in = torch.empty(n, Hp, **f32)
                    _lib.check(lib.ggpm_gather_rows(P(Cg), Hp, rows, n, Hp, P(c_in), Hp, 0, 0, s), "gather_rows")
                x = torch.empty(G, n, Hp, **f32)
                _lib.check(lib.ggpm_gather_rows(P(Xflat), Hp, _vp(xrows[t]), G * n, Hp, P(x), Hp, 0, 0, s), "gather_rows")
                hs = Hs_all[qoff[t]:qoff[t + 1]].view(depth + 1, n, Hp)
                cs = Cs_all[qoff[t]:qoff[t + 1]].view(depth + 1, n, Hp) if lstm else None
                qs = Qs_all[roff[t]:roff[t + 1]].view(depth, n, Hp)
                st = St_all[:, roff[t]:roff[t + 1]].view(5, depth, n, Hp)
                sparse_forward(n, h_in, c_in, _vp(frz_loc + plan.floc_off[t]), x, _vp(ptr[("lpred_rp", t)]),
                               _vp(ptr[("lpred_col", t)]), hs, cs, qs, st)
                live = _vp(ptr[("live", t)])
                _lib.check(lib.ggpm_scatter_rows(P(hs[depth]), Hp, live, n, Hp, P(Hg), Hp, 0, s), "scatter_rows")
                if lstm:
                    _lib.check(lib.ggpm_scatter_rows(P(cs[depth]), Hp, live, n, Hp, P(Cg), Hp, 0, s), "scatter_rows")
                Xl.append(x[3] if lstm else x[1])           # the backward reads the forget / reset gate's input only
                h_prev = Hg
            else:
                sparse_forward(E1, h_prev, c_prev, P(frz[t]), X, _vp(ptr[("pred_rp", t)]), _vp(ptr[("pred_col", t)]),
                               Hs[t], Cs[t] if lstm else None, Qs[t], St[t])
                if lstm:
                    c_prev = Cs[t, depth]
                h_prev = Hs[t, depth]
            nei, node = NEI[a0:a1], NODE[a0:a1]
            _lib.check(lib.ggpm_segment_sum(P(h_prev), Hp, _vp(ptr[("agr_rp", t)]), _vp(ptr[("agr_col", t)]), ns, H,
                                            P(nei), Hp, 0, Hp, s), "segment_sum")
            F_.gemm_ksegments(1, ns, H, [fn_all[a0:a1], nei], [ldF, Hp], [Wout, Wout[:, Fdim:]], [ldwo, ldwo], [Fdim, H],
                              node, Hp, Hp, bias=bout, act=F_.ACT_RELU)
            if drop is not None:
                _lib.check(lib.ggpm_dropout(P(node), ns, H, Hp, drop[0], drop[1], drop[2], t, s), "dropout")
            _lib.check(lib.ggpm_segment_sum(P(node), Hp, _vp(ptr[("pool_rp", t)]), _vp(ptr[("pool_col", t)]), ni, H,
                                            P(pooled[i0:i1]), Hp, 0, Hp, s), "segment_sum")
            for j, (row0, n) in enumerate(plan.step_cands[t]):
                _lib.check(lib.ggpm_gather_rows(P(node), Hp, _vp(ptr[("cand_pos", t, j)]), n, H, P(cand[row0:row0 + n]),
                                                Hp, 0, Hp, s), "gather_rows")
        ctx.plan, ctx.meta, ctx.drop, ctx.compact = plan, (cell, depth, H, Fdim, I), drop, compact
        if compact:
            ctx.save_for_backward(fn_all, hmess, NODE, NEI, Hs_all, Qs_all, St_all, *([Cs_all] if lstm else []), *params, *Xl)
        else:
            ctx.save_for_backward(fn_all, hmess, X, Hs, Qs, St, NODE, NEI, *([Cs] if lstm else []), *params)
        ctx.keep = D
        return pooled, cand

    @staticmethod
    def backward(ctx, d_pooled, d_cand):
        lib = _lib.load()
        plan, (cell, depth, H, Fdim, I), drop = ctx.plan, ctx.meta, ctx.drop
        lstm = cell == "LSTM"
        sv = list(ctx.saved_tensors)
        compact, T = ctx.compact, plan.T
        if compact:
            fn_all, hmess, NODE, NEI, Hs_all, Qs_all, St_all = sv[:7]
            Cs_all = sv[7] if lstm else None
            k0 = 8 if lstm else 7
            npar = 10 if lstm else 9
            params, Xl = sv[k0:k0 + npar], sv[k0 + npar:]
        else:
            fn_all, hmess, X, Hs, Qs, St, NODE, NEI = sv[:8]
            Cs = sv[8] if lstm else None
            params = sv[9:] if lstm else sv[8:]
        D = ctx.keep
        ptr, P = D["ptr"], F_._p
        dev = hmess.device
        Hp = F_.padded_hidden(H)
        E1 = plan.E1
        f32 = dict(dtype=torch.float32, device=dev)
        G = 4 if lstm else 3
        if lstm:
            Wi, bi, Wo_g, bo_g, Wu, bu_g, Wf, bf, Wout, bout = params
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh, Wout, bout = params
        d_pooled = d_pooled.contiguous()
        d_cand = d_cand.contiguous()
        s = F_._stream()
        frz = D["frozen"]
        ns_tot = plan.aoff[-1]
        DPRE = torch.empty(ns_tot, Hp, **f32)
        dX_tot = torch.zeros(G, E1, Hp, **f32)
        dH = torch.zeros(E1, Hp, **f32)
        dC = torch.zeros(E1, Hp, **f32) if lstm else None
        if compact:
            nmax = max(plan.nloc)
            xrows = plan.gate_rows_device(G)
            dXflat = dX_tot.view(G * E1, Hp)
            frz_loc = D["frozen_loc"].data_ptr()
            roff, qoff = plan.row_offsets(depth)
            # gate-gradient stashes of all steps, stacked like the forward's (DQ with a zero slot per step at the end so
            # that it lines up with the depth + 1 state slots): contracted once behind the loop
            stacked = os.environ.get("GGPM_ATOM_STACKED_WGRADS", "1") != "0"        # (0: contract per step; dev A/B)
            if stacked:
                DG_all = torch.empty(3 if lstm else 2, roff[-1], Hp, **f32)
                DQ_all = torch.zeros(qoff[-1], Hp, **f32)
        else:
            nmax = E1
            dX = torch.empty(G, E1, Hp, **f32)
            dH2 = torch.empty(E1, Hp, **f32)
            dC2 = torch.empty(E1, Hp, **f32) if lstm else None
        nh = 4 if lstm else 3                                   # hidden-half weight gradients (+ GRU: b_u)
        acc = [torch.zeros(H, H, **f32) for _ in range(nh)] + ([] if lstm else [torch.zeros(H, **f32)])
        tmp = [torch.empty(H, H, **f32) for _ in range(nh)] + ([] if lstm else [torch.empty(H, **f32)])
        wb = int((lib.ggpm_lstm_backward_workspace_bytes if lstm else lib.ggpm_gru_backward_workspace_bytes)(nmax, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        ldwo = Wout.stride(0)

        def sparse_backward(n, fz, xg, rp, col, rpT, colT, hs, cs, qs, st, dhd, dcd, dhin, dcin, dx):
            if lstm:
                _lib.check(lib.ggpm_lstm_sparse_backward(
                    n, H, depth, fz, P(xg), P(Wi[:, I:]), Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0),
                    P(Wu[:, I:]), Wu.stride(0), P(Wf[:, I:]), Wf.stride(0), rp, col, rpT, colT, P(hs), P(cs),
                    P(qs), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(dhd), P(dcd), P(dhin), P(dcin), P(dx[0]),
                    P(dx[1]), P(dx[2]), P(dx[3]), P(tmp[0]), H, P(tmp[1]), H, P(tmp[2]), H, P(tmp[3]), H, P(work),
                    work.numel() * 4, s), "lstm_sparse_backward")
            else:
                _lib.check(lib.ggpm_gru_sparse_backward(
                    n, H, depth, fz, P(xg), P(Wz[:, I:]), Wz.stride(0), P(Ur), Ur.stride(0), P(Wh[:, I:]),
                    Wh.stride(0), rp, col, rpT, colT, P(hs), P(qs), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]),
                    P(dhd), P(dhin), P(dx[0]), P(dx[1]), P(dx[2]), P(tmp[0]), H, P(tmp[1]), H, P(tmp[3]), P(tmp[2]), H,
                    P(work), work.numel() * 4, s), "gru_sparse_backward")

        for t in range(T - 1, -1, -1):
            a0, a1, i0, i1 = plan.aoff[t], plan.aoff[t + 1], plan.ioff[t], plan.ioff[t + 1]
            ns, ni = a1 - a0, i1 - i0
            d_node = torch.empty(ns, Hp, **f32)
            _lib.check(lib.ggpm_segment_sum(P(d_pooled[i0:i1]), Hp, _vp(ptr[("poolT_rp", t)]), _vp(ptr[("poolT_col", t)]),
                                            ns, H, P(d_node), Hp, 0, Hp, s), "segment_sum")
            for j, (row0, n) in enumerate(plan.step_cands[t]):
                _lib.check(lib.ggpm_segment_sum(P(d_cand[row0:row0 + n]), Hp, _vp(ptr[("candT_rp", t, j)]),
                                                _vp(ptr[("candT_col", t, j)]), ns, H, P(d_node), Hp, 1, 0, s), "segment_sum")
            dpre = DPRE[a0:a1]
            _lib.check(lib.ggpm_act_backward(P(d_node), P(NODE[a0:a1]), ns, H, Hp, F_.ACT_RELU, 0, P(dpre), s), "act_backward")
            if drop is not None:            # d(dropout . relu) = mask * scale * relu' (the saved output is the dropped one)
                _lib.check(lib.ggpm_dropout(P(dpre), ns, H, Hp, drop[0], drop[1], drop[2], t, s), "dropout")
            d_nei = torch.empty(ns, Hp, **f32)
            F_.gemm(0, 0, ns, H, H, dpre, Hp, Wout[:, Fdim:], ldwo, d_nei, Hp, Hp)
            # d(state after step t) = what step t+1 passed back + the read-out's share
            _lib.check(lib.ggpm_segment_sum(P(d_nei), Hp, _vp(ptr[("agrT_rp", t)]), _vp(ptr[("agrT_col", t)]), E1, H,
                                            P(dH), Hp, 1, 0, s), "segment_sum")
            if compact:
                n = plan.nloc[t]
                rows = _vp(ptr[("rows", t)])
                dhd, dhin = torch.empty(n, Hp, **f32), torch.empty(n, Hp, **f32)
                _lib.check(lib.ggpm_gather_rows(P(dH), Hp, rows, n, Hp, P(dhd), Hp, 0, 0, s), "gather_rows")
                dcd = dcin = None
                if lstm:
                    dcd, dcin = torch.empty(n, Hp, **f32), torch.empty(n, Hp, **f32)
                    _lib.check(lib.ggpm_gather_rows(P(dC), Hp, rows, n, Hp, P(dcd), Hp, 0, 0, s), "gather_rows")
                dx = torch.empty(G, n, Hp, **f32)
                if stacked:
                    dg = [DG_all[k, roff[t]:roff[t + 1]] for k in range(DG_all.shape[0])] + [DQ_all[qoff[t]:qoff[t + 1]]]
                    lib.ggpm_backward_defer_stash(P(dg[0]), P(dg[1]), P(dg[2]), P(dg[3]) if lstm else None)
                sparse_backward(n, _vp(frz_loc + plan.floc_off[t]), Xl[t], _vp(ptr[("lpred_rp", t)]),
                                _vp(ptr[("lpred_col", t)]), _vp(ptr[("lsucc_rp", t)]), _vp(ptr[("lsucc_col", t)]),
                                Hs_all[qoff[t]:qoff[t + 1]], Cs_all[qoff[t]:qoff[t + 1]] if lstm else None,
                                Qs_all[roff[t]:roff[t + 1]], St_all[:, roff[t]:roff[t + 1]], dhd, dcd, dhin, dcin, dx)
                # d(state before step t): the frozen rows' carried gradient, zero on the recomputed rows
                _lib.check(lib.ggpm_scatter_rows(P(dhin), Hp, rows, n, Hp, P(dH), Hp, 0, s), "scatter_rows")
                if lstm:
                    _lib.check(lib.ggpm_scatter_rows(P(dcin), Hp, rows, n, Hp, P(dC), Hp, 0, s), "scatter_rows")
                _lib.check(lib.ggpm_scatter_rows(P(dx), Hp, _vp(xrows[t]), G * n, Hp, P(dXflat), Hp, 1, s), "scatter_rows")
                if not stacked:
                    torch._foreach_add_(acc, tmp)
            else:
                sparse_backward(E1, P(frz[t]), X[3] if lstm else X[1], _vp(ptr[("pred_rp", t)]), _vp(ptr[("pred_col", t)]),
                                _vp(ptr[("succ_rp", t)]), _vp(ptr[("succ_col", t)]), Hs[t], Cs[t] if lstm else None, Qs[t],
                                St[t], dH, dC, dH2, dC2, dX)
                if lstm:
                    dC, dC2 = dC2, dC
                dH, dH2 = dH2, dH
                torch._foreach_add_([dX_tot] + acc, [dX] + tmp)
        # ---- parameter gradients, once
        if compact and stacked:
            R, RQ = roff[-1], qoff[-1]
            wsb = int(lib.ggpm_weight_grads_stacked_workspace_bytes(H, max(R, RQ)))
            ws = torch.empty((wsb + 3) // 4, **f32)
            if lstm:        # acc: Wi_h, Wo_h, Wu_h, Wf_h
                _lib.check(lib.ggpm_lstm_weight_grads_stacked(
                    R, RQ, H, P(DG_all[0]), P(DG_all[1]), P(DG_all[2]), P(St_all[0]), P(DQ_all), P(Hs_all), P(acc[0]), H,
                    P(acc[1]), H, P(acc[2]), H, P(acc[3]), H, P(ws), ws.numel() * 4, s), "lstm_weight_grads_stacked")
            else:           # acc: Wz_h, U_r, Wh_h, b_u; St_all: S, G, Z, M, R
                _lib.check(lib.ggpm_gru_weight_grads_stacked(
                    R, RQ, H, P(DG_all[0]), P(St_all[1]), P(DG_all[1]), P(St_all[0]), P(DQ_all), P(Hs_all), P(acc[0]), H,
                    P(acc[1]), H, P(acc[3]), P(acc[2]), H, P(ws), ws.numel() * 4, s), "gru_weight_grads_stacked")
        x_ld = F_._ld(hmess)

        def full(W, k, hidden):                 # [input half from the summed dX | accumulated hidden half]
            dW = torch.empty_like(W)
            F_.gemm(1, 0, H, I, E1, dX_tot[k], Hp, hmess, x_ld, dW, dW.stride(0), I, splitk=True)
            if hidden is not None:
                dW[:, I:] = hidden
            return dW

        dWout = torch.empty_like(Wout)
        F_.gemm(1, 0, H, Fdim, ns_tot, DPRE, Hp, fn_all, F_._ld(fn_all), dWout, dWout.stride(0), Fdim, splitk=True)
        F_.gemm(1, 0, H, H, ns_tot, DPRE, Hp, NEI, Hp, dWout[:, Fdim:], dWout.stride(0), H, splitk=True)
        dbout = F_.colsum(DPRE, ns_tot, H)
        if lstm:
            grads = []
            for k, W in enumerate((Wi, Wo_g, Wu, Wf)):
                grads += [full(W, k, acc[k]), F_.colsum(dX_tot[k], E1, H)]
            grads += [dWout, dbout]
        else:          # tmp / acc order of the GRU: Wz_h, U_r, Wh_h, b_u
            grads = [full(Wz, 0, acc[0]), F_.colsum(dX_tot[0], E1, H), full(Wr, 1, None), acc[1], acc[3],
                     full(Wh, 2, acc[2]), F_.colsum(dX_tot[2], E1, H), dWout, dbout]
        return (None,) * 9 + tuple(grads)


def atom_decode(plan: AtomPlan, graph_encoder, hnode_a: torch.Tensor, hmess_a: torch.Tensor, fn_all: torch.Tensor):
    """-> (pooled [n_inst, Hp], cand [n_cand, Hp]) for ``graph_encoder`` = the decoder's atom-level ``IncMPNEncoder``."""
    from .rnn import LSTM
    rnn, wo = graph_encoder.rnn, graph_encoder.W_o
    lstm = isinstance(rnn, LSTM)
    if lstm:
        params = (rnn.W_i[0].weight, rnn.W_i[0].bias, rnn.W_o[0].weight, rnn.W_o[0].bias, rnn.W[0].weight, rnn.W[0].bias,
                  rnn.W_f[0].weight, rnn.W_f[0].bias)
    else:
        params = (rnn.W_z.weight, rnn.W_z.bias, rnn.W_r.weight, rnn.U_r.weight, rnn.U_r.bias, rnn.W_h.weight, rnn.W_h.bias)
    drop = None
    if graph_encoder.training and wo[2].p > 0:
        seed = torch.randint(0, 2 ** 31 - 1, (2,), dtype=torch.int64)
        drop = (float(wo[2].p), int(seed[0]), int(seed[1]))
    return _AtomDecode.apply(plan, "LSTM" if lstm else "GRU", rnn.depth, rnn.hidden_size, graph_encoder.node_fdim,
                             rnn.input_size, fn_all, hmess_a, drop, *params, wo[0].weight, wo[0].bias)
