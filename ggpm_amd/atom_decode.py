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

Compact steps (default, _dev.ATOM_COMPACT = False switches them off): a step recomputes a few hundred of the level's thousands
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
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import _dev, _lib
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

    def __init__(self, schedule, n_gnodes: int, n_gmess: int, full: Optional[bool] = None):
        """``full``: also build the level-wide per-step tables of the full-level form (``_AtomDecode``; default: only when
        _dev.ATOM_COMPACT = False -- the compact form does not read them and they are the larger half of the build and upload)."""
        P, steps = schedule.plan, schedule.steps
        self._native, self._schedule, self._raw_cache = None, None, None
        self.T, self.N1, self.E1 = len(steps), n_gnodes, n_gmess
        self.full = (not compact_enabled()) if full is None else bool(full)
        full = self.full
        self.ok = all(len(st["atoms"]) > 0 for st in steps)          # (the reference keeps a stale node buffer otherwise)
        ints: List[np.ndarray] = []
        self.where = {}
        fill = [0]

        def put(key, a):
            a = np.asarray(a, dtype=np.int32).reshape(-1)
            self.where[key] = (fill[0], len(a))
            fill[0] += len(a)
            ints.append(a)

        frozen = np.ones((self.T if full else 0, n_gmess), dtype=np.uint8)
        frozen_loc: List[np.ndarray] = []                     # compact steps: mask over the step's local rows
        self.nloc, self.floc_off = [], [0]
        self._raw, self._ct = [], {}                          # per step (rows, mask, incoming-message table, pool table)
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
            # predecessor CSR over all E1 rows (only this step's bonds have entries) and its transpose
            order = np.argsort(bonds, kind="stable")
            tab = P["g_bgraph"][boff[t]:boff[t + 1]][order]
            cnt = (tab > 0).sum(axis=1)
            if full:
                frozen[t, bonds] = 0
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
            # incoming messages of the step's atoms (rows local to the step) and the transpose (rows = messages)
            atab = P["g_agraph"][aoff[t]:aoff[t + 1]]
            if full:
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
            if full:
                pcnt = (loc >= 0).sum(axis=1)
                rp, col = _csr_from_lists(pcnt, loc[loc >= 0])
                put(("pool_rp", t), rp); put(("pool_col", t), col)
                rpT, colT = _transpose(np.repeat(np.arange(len(ptab)), pcnt), loc[loc >= 0], len(atoms))
                put(("poolT_rp", t), rpT); put(("poolT_col", t), colT)
            self._raw.append((rows, fl, atab, loc))
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
        self._cand_pos, self._per_step_cands, self._kbase = cand_pos, per_step_cands, kbase
        self.step_cands = []                                  # per step: [(flat row offset, count)] + transposed scatter
        for t, here in enumerate(per_step_cands):
            segs = [(kbase[k] + start, n) for (k, start, n) in here]
            self.step_cands.append(segs)
            ns = aoff[t + 1] - aoff[t]
            for j, (k, start, n) in enumerate(here if full else ()):
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
        self._dev = {}

    def __getstate__(self):
        """Host tables only: device copies and the descriptors that name them stay with the process that made them."""
        st = dict(self.__dict__)
        st["_dev"] = {}
        st["_schedule"] = self._schedule() if self._schedule is not None else None       # (the weak reference back: __setstate__)
        st["_ct"] = {k: dict(v, dev={}) for k, v in self._ct.items()}
        if self._native is not None:
            from .decoder import _FrozenTables
            st["_native"] = self._native if isinstance(self._native, _FrozenTables) else _FrozenTables(self._native)
            st["ints"], st["frozen_loc"] = np.array(self.ints), np.array(self.frozen_loc)
            st["cand_meta"] = {k: {n: np.array(v) for n, v in m.items()} for k, m in self.cand_meta.items()}
            st["_raw_cache"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        if self._schedule is not None:
            self._schedule = weakref.ref(self._schedule)

    # ------------------------------------------------------------------ tables built by csrc/schedule.hip
    @classmethod
    def from_native(cls, schedule, nt, n_gnodes: int, n_gmess: int) -> "AtomPlan":
        """The same object over the tables of ``ggpm_schedule_build``: ``ints`` IS the int32 device pack (so the
        addresses handed to the C drivers are offsets into one upload shared with the schedule), the compact tables for the
        (depth, gates) the build was given are already there."""
        self = cls.__new__(cls)
        sc, g = nt.scalars(), nt.get
        # (weak: the schedule owns this plan -- a strong reference back would make the pair, and with it every device table of
        # the batch, a reference cycle that only Python's collector frees)
        self._native, self._schedule = nt, weakref.ref(schedule)
        self.T, self.N1, self.E1, self.full, self.ok = sc["T"], n_gnodes, n_gmess, False, bool(sc["ok"])
        self.nloc, self.floc_off = g("nloc").tolist(), g("floc_off").tolist()
        self.aoff, self.ioff = schedule.plan["atom_off"], schedule.plan["inst_off"]
        self.ints, self.frozen_loc = nt.packs[2], g("frozen_loc")
        self.frozen = np.ones((0, n_gmess), dtype=np.uint8)
        d = nt.dir
        e32 = lambda name: d[name][1] // 4                      # element offset inside the int32 pack
        rp_off, col_off, rows_off = g("lpred_rp_off").tolist(), g("lpred_col_off").tolist(), g("rows_off").tolist()
        self.where = {}
        for t in range(self.T):
            for name, off in (("lpred_rp", rp_off), ("lsucc_rp", rp_off), ("lpred_col", col_off), ("lsucc_col", col_off),
                              ("rows", rows_off)):
                self.where[(name, t)] = (e32(name) + off[t], off[t + 1] - off[t])
        cb = g("cand_blocks").reshape(-1, 3).tolist()
        self.cand_blocks = [tuple(x) for x in cb]
        self.n_cand = sc["n_cand"]
        self.cand_meta = {k: dict(icls=g("meta_icls/%d" % k), nth=g("meta_nth/%d" % k), dest=g("meta_dest/%d" % k))
                          for k, _, _ in self.cand_blocks}
        self._kbase = {k: b for k, b, _ in self.cand_blocks}
        self._cand_pos = {k: g("cand_pos/%d" % k) for k, _, _ in self.cand_blocks}
        psc = g("step_cands").reshape(-1, 4).tolist()
        self._per_step_cands = [[] for _ in range(self.T)]
        for t, k, start, n in psc:
            self._per_step_cands[t].append((k, start, n))
        self.step_cands = [[(self._kbase[k] + start, n) for (k, start, n) in here] for here in self._per_step_cands]
        self._raw_cache, self._ct, self._dev = None, {}, {}
        if sc["depth"] > 0 and sc["gates"] > 0:
            names = ["srcF", "srcH", "agr_rp", "agr_col", "agrT_rp", "agrT_col", "pool_rp", "pool_col", "poolT_rp", "poolT_col",
                     "cand_idx", "candT_rp", "candT_col", "xrows", "xT_rp", "xT_col"]
            foff = g("foff").tolist()
            where = {}
            for name in names:
                if name in ("srcF", "srcH"):
                    for t in range(self.T):
                        where[(name, t)] = (e32(name) + foff[t], foff[t + 1] - foff[t])
                else:
                    where[name] = (e32(name), d[name][2])
            self._ct[(sc["depth"], sc["gates"])] = dict(ints=nt.packs[2], where=where, foff=foff, Ftot=sc["Ftot"], dev={},
                                                        native=True)
        return self

    def _owner(self):
        sch = self._schedule() if self._schedule is not None else None
        if sch is None:
            raise RuntimeError("AtomPlan: the DecodeSchedule these tables belong to is gone")
        return sch

    @property
    def _raw(self):
        """per step (rows, local frozen mask, incoming-message table, pool table) -- what compact_tables reads"""
        if self._raw_cache is None:
            nt, P = self._native, self._owner().plan
            rows_all, rows_off = nt.get("rows").astype(np.int64), nt.get("rows_off").tolist()
            loc = nt.get("loc").reshape(-1, P["pool"].shape[1])
            self._raw_cache = []
            for t in range(self.T):
                rows = rows_all[rows_off[t]:rows_off[t + 1]]
                fl = np.array(self.frozen_loc[self.floc_off[t]:self.floc_off[t] + len(rows)])
                self._raw_cache.append((rows, fl, P["g_agraph"][self.aoff[t]:self.aoff[t + 1]], loc[self.ioff[t]:self.ioff[t + 1]]))
        return self._raw_cache

    @_raw.setter
    def _raw(self, value):
        self._raw_cache = value

    def row_offsets(self, depth: int):
        """-> (offsets of the steps' [depth, n] blocks, offsets of their [depth + 1, n] blocks), each of length T + 1."""
        n = np.asarray(self.nloc, dtype=np.int64)
        return (np.concatenate([[0], np.cumsum(depth * n)]).tolist(),
                np.concatenate([[0], np.cumsum((depth + 1) * n)]).tolist())

    def compact_tables(self, depth: int, gates: int):
        """Host tables of the compact form that depend on the iteration count / the number of gates (cached).

        Every step leaves the final states of its n rows in slot ``depth`` of its block of the stacked state buffer.  ``F
        id`` = foff[t] + i numbers those rows over all steps; ``H id`` is the row of the same state in the stacked
        [(depth + 1) * n_t]-row blocks.  A message's state at time t is the final state of the last step <= t that
        recomputed it (zero if none did): that look-up is resolved here, once, so the device never needs a level-wide
        state -- a step's frozen rows are gathered straight from the earlier steps' blocks (``srcH``) and the read-out of
        ALL steps (incoming messages -> atoms -> pooled clusters / attachment candidates) is one CSR over the stacked
        blocks (``agr``), i.e. a handful of launches behind the loop instead of a handful per step."""
        key = (depth, gates)
        if key in self._ct:
            return self._ct[key]
        T, E1 = self.T, self.E1
        n = np.asarray(self.nloc, dtype=np.int64)
        foff = np.concatenate([[0], np.cumsum(n)])
        Ftot = int(foff[-1])
        stepof = np.repeat(np.arange(T), n)
        Hid = (depth + 1) * foff[stepof] + depth * n[stepof] + (np.arange(Ftot) - foff[stepof])
        last = np.full(E1, -1, dtype=np.int64)
        tabs, where, fill = [], {}, [0]

        def put(key_, a):
            a = np.asarray(a, dtype=np.int32).reshape(-1)
            where[key_] = (fill[0], len(a))
            fill[0] += len(a)
            tabs.append(a)

        agr_r, agr_c, pool_r, pool_c = [], [], [], []
        for t, (rows, fl, atab, loc) in enumerate(self._raw):
            sf = np.where(fl == 1, last[rows], -1)
            put(("srcF", t), sf)
            put(("srcH", t), np.where(sf >= 0, Hid[np.maximum(sf, 0)], -1))
            live = np.nonzero(fl == 0)[0]
            last[rows[live]] = foff[t] + live
            r, c = np.nonzero(atab > 0)
            f = last[atab[r, c]]
            agr_r.append(self.aoff[t] + r[f >= 0]); agr_c.append(f[f >= 0])
            r, c = np.nonzero(loc >= 0)
            pool_r.append(self.ioff[t] + r); pool_c.append(self.aoff[t] + loc[r, c])
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, np.int64)
        agr_r, agr_c, pool_r, pool_c = cat(agr_r), cat(agr_c), cat(pool_r), cat(pool_c)
        ns_tot, n_inst = int(self.aoff[-1]), int(self.ioff[-1])
        rp = lambda r_, nrows: np.concatenate([[0], np.cumsum(np.bincount(r_, minlength=nrows))])
        put("agr_rp", rp(agr_r, ns_tot)); put("agr_col", Hid[agr_c])                 # rows ascend already (step order)
        a, b = _transpose(agr_r, agr_c, Ftot); put("agrT_rp", a); put("agrT_col", b)
        put("pool_rp", rp(pool_r, n_inst)); put("pool_col", pool_c)
        a, b = _transpose(pool_r, pool_c, ns_tot); put("poolT_rp", a); put("poolT_col", b)
        cidx = np.full(max(self.n_cand, 1), -1, dtype=np.int64)
        for t, here in enumerate(self._per_step_cands):
            for (k, start, cnt) in here:
                pos = np.asarray(self._cand_pos[k][start:start + cnt], dtype=np.int64)
                cidx[self._kbase[k] + start:self._kbase[k] + start + cnt] = np.where(pos >= 0, self.aoff[t] + pos, -1)
        put("cand_idx", cidx)
        ok = np.nonzero(cidx >= 0)[0]
        a, b = _transpose(ok, cidx[ok], ns_tot); put("candT_rp", a); put("candT_col", b)
        xrows = np.concatenate([self.gate_rows(t, gates) for t in range(T)]).astype(np.int64)
        put("xrows", xrows)
        a, b = _transpose(np.arange(len(xrows)), xrows, gates * E1); put("xT_rp", a); put("xT_col", b)
        ct = dict(ints=np.concatenate(tabs), where=where, foff=foff.tolist(), Ftot=Ftot, dev={})
        self._ct[key] = ct
        return ct

    def compact_device(self, depth: int, gates: int, device):
        """-> (tables dict, {key: device address}) of compact_tables on ``device`` (uploaded once)."""
        ct = self.compact_tables(depth, gates)
        device = torch.device(device)
        if device not in ct["dev"]:
            # (natively built tables live in the int32 pack the plan's own tables are views of: no second upload)
            d = self.to_device(device)["ints"] if ct.get("native") else F_.upload(ct["ints"], device)
            ct["dev"][device] = (d, None, {k: d.data_ptr() + 4 * off for k, (off, _) in ct["where"].items()})
        return ct, ct["dev"][device][2]

    def gate_rows(self, t: int, gates: int) -> np.ndarray:
        """Rows of the step's compact set inside the stacked [gates * E1, Hp] gate-input matrix."""
        off, n = self.where[("rows", t)]
        rows = self.ints[off:off + n].astype(np.int64)
        return np.concatenate([k * self.E1 + rows for k in range(gates)]).astype(np.int32)

    def to_device(self, device):
        """Device copies of the tables, one set PER DEVICE (kept for the life of the plan: the raw addresses handed to
        the C drivers -- ``ptr``, the cached ``ggpm_decode_steps`` descriptors under ``desc`` -- name exactly these
        tensors and live in the same dict, so they can never outlast them)."""
        device = torch.device(device)
        D = self._dev.get(device)
        if D is None and self._native is not None:
            sd = self._owner().to_device(device)._dev
            d64, d32 = sd["native"]
            nd = self._native.dir

            def view(name):
                pack, off, cnt, el = nd[name]
                return (d64 if pack == 1 else d32)[off // el:off // el + cnt]
            floff = nd["frozen_loc"][1]
            dl = d32.view(torch.uint8)[floff:floff + nd["frozen_loc"][2]]
            base = d32.data_ptr()
            ptr = {k: base + 4 * off for k, (off, n) in self.where.items()}
            meta = {k: dict(icls=view("meta_icls/%d" % k), nth=view("meta_nth/%d" % k), dest=view("meta_dest/%d" % k))
                    for k, _, _ in self.cand_blocks}
            D = self._dev[device] = dict(device=device, ints=d32, frozen=None, frozen_loc=dl, ptr=ptr, meta=meta, desc={},
                                         seen=set(), pack64=d64)
        if D is None:
            di, dl = F_.upload(self.ints, device), F_.upload(self.frozen_loc, device)
            df = F_.upload(self.frozen, device) if self.full else None
            base = di.data_ptr()
            ptr = {k: base + 4 * off for k, (off, n) in self.where.items()}
            meta = {k: {n: F_.upload(v, device) for n, v in m.items()} for k, m in self.cand_meta.items()}
            for m in meta.values():
                m["icls"] = m["icls"].to(torch.int32)
            D = self._dev[device] = dict(device=device, ints=di, frozen=df, frozen_loc=dl, ptr=ptr, meta=meta, desc={},
                                         seen=set())
        if device.type == "cuda":
            # uploaded on whatever stream was current then (the atom-ahead stream, usually) and read on others: tell the
            # allocator once per reading stream
            cur = torch.cuda.current_stream(device)
            if cur.cuda_stream not in D["seen"]:
                D["seen"].add(cur.cuda_stream)
                ts = [D["ints"], D["frozen_loc"]] + ([D["frozen"]] if D["frozen"] is not None else [])
                ts += [v for m in D["meta"].values() for v in m.values()] + ([D["pack64"]] if "pack64" in D else [])
                for t in ts:
                    t.record_stream(cur)
        return D


class DecodeSteps(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_decode_steps (host arrays of per-step sizes / device pointers)."""
    _fields_ = [("T", ctypes.c_int), ("H", ctypes.c_int), ("depth", ctypes.c_int), ("lstm", ctypes.c_int),
                ("n", ctypes.c_void_p), ("foff", ctypes.c_void_p), ("roff", ctypes.c_void_p), ("qoff", ctypes.c_void_p),
                ("srcH", ctypes.c_void_p), ("srcF", ctypes.c_void_p), ("frozen", ctypes.c_void_p),
                ("pred_rowptr", ctypes.c_void_p), ("pred_col", ctypes.c_void_p), ("succ_rowptr", ctypes.c_void_p),
                ("succ_col", ctypes.c_void_p)]


def _decode_steps(plan: "AtomPlan", D, ct, cp, H: int, depth: int, lstm: bool):
    """The ggpm_decode_steps descriptor of (plan, device, depth, cell), built once (it only names resident tables)."""
    key = ("steps", depth, bool(lstm), H)
    hit = D["desc"].get(key)
    if hit is not None:
        return hit
    T, ptr = plan.T, D["ptr"]
    roff, qoff = plan.row_offsets(depth)
    arr32 = _lib.array_type(ctypes.c_int32, T)(*plan.nloc)
    a64 = lambda v: _lib.array_type(ctypes.c_int64, T + 1)(*[int(x) for x in v])
    pa = lambda vals: _lib.array_type(ctypes.c_void_p, T)(*[int(v) for v in vals])
    frz = D["frozen_loc"].data_ptr()
    keep = dict(n=arr32, foff=a64(ct["foff"]), roff=a64(roff), qoff=a64(qoff),
                srcH=pa(cp[("srcH", t)] for t in range(T)), srcF=pa(cp[("srcF", t)] for t in range(T)),
                frozen=pa(frz + plan.floc_off[t] for t in range(T)),
                pred_rowptr=pa(ptr[("lpred_rp", t)] for t in range(T)), pred_col=pa(ptr[("lpred_col", t)] for t in range(T)),
                succ_rowptr=pa(ptr[("lsucc_rp", t)] for t in range(T)), succ_col=pa(ptr[("lsucc_col", t)] for t in range(T)))
    d = DecodeSteps(T, H, depth, int(lstm), *[ctypes.addressof(keep[k]) for k in
                                               ("n", "foff", "roff", "qoff", "srcH", "srcF", "frozen", "pred_rowptr",
                                                "pred_col", "succ_rowptr", "succ_col")])
    D["desc"][key] = (d, keep)      # beside the tensors whose addresses it holds
    return d, keep


# _dev.DECODE_DRIVER (False: the step loops are issued from Python; dev A/B, tests) -- read at CALL time, like every
# _dev setting: a tool or test that flips the attribute after this module was imported gets the other form.
# The two step loops are ~280 launches = 1.5-1.8 ms of host time each, inside one C call.  _dev.ATOM_ASYNC (default)
# hands them to a worker thread of the library (ggpm_decode_steps_*_async): the forward loop is then issued beside the
# encoder's forward, the backward loop beside the encoder's backward -- the autograd engine reaches the two nodes at about
# the same time and would otherwise issue one chain only after the other, although they do not depend on each other.
# What follows a loop on its stream (read-out / parameter gradients) is enqueued after ggpm_decode_join.
_INFLIGHT: list = []         # buffers named by loops the worker may still be issuing (released by the next join)
_PENDING: dict = {}          # id(plan) -> the forward's `finish` closure, taken by atom_decode()


def _join_worker(what: str) -> None:
    _lib.check(_lib.load().ggpm_decode_join(), what)
    del _INFLIGHT[:]


# _dev.PACK_ONCE = False: every decode step packs its weights again (dev A/B)


def compact_enabled() -> bool:
    return _dev.ATOM_COMPACT


def _vp(addr: int) -> ctypes.c_void_p:
    return ctypes.c_void_p(addr)


class _AtomDecode(torch.autograd.Function):
    """(pooled cluster vectors of all visits [n_inst, Hp], attachment-candidate atom vectors [n_cand, Hp]) -- the
    full-level form: every step runs over all E1 rows of the level with a frozen mask (_dev.ATOM_COMPACT = False)."""

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
        Hs = torch.empty(T, depth + 1, E1, Hp, **f32)
        Cs = torch.empty(T, depth + 1, E1, Hp, **f32) if lstm else None
        Qs = torch.empty(T, depth, E1, Hp, **f32)
        St = torch.empty(T, 5, depth, E1, Hp, **f32)
        zero = torch.zeros(E1, Hp, **f32)
        ns_tot, n_inst = plan.aoff[-1], plan.ioff[-1]
        NODE = torch.empty(ns_tot, Hp, **f32)
        NEI = torch.empty(ns_tot, Hp, **f32)
        pooled = torch.empty(n_inst, Hp, **f32)
        cand = torch.zeros(max(plan.n_cand, 1), Hp, **f32)
        wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H) if lstm else lib.ggpm_gru_pack_floats(H)), **f32)
        s = F_._stream()
        frz = D["frozen"]
        ldF, ldwo = F_._ld(fn_all), Wout.stride(0)
        h_prev, c_prev = zero, zero
        for t in range(T):
            a0, a1, i0, i1 = plan.aoff[t], plan.aoff[t + 1], plan.ioff[t], plan.ioff[t + 1]
            ns, ni = a1 - a0, i1 - i0
            st = St[t]
            if lstm:
                _lib.check(lib.ggpm_lstm_sparse_forward(
                    E1, H, depth, P(h_prev), P(c_prev), P(frz[t]), P(X[0]), P(X[1]), P(X[2]), P(X[3]), P(Wi[:, I:]),
                    Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0), P(Wu[:, I:]), Wu.stride(0), P(Wf[:, I:]), Wf.stride(0),
                    _vp(ptr[("pred_rp", t)]), _vp(ptr[("pred_col", t)]), P(Hs[t]), P(Cs[t]), P(Qs[t]), P(st[0]), P(st[1]),
                    P(st[2]), P(st[3]), P(st[4]), P(wpack), 1, s), "lstm_sparse_forward")
                c_prev = Cs[t, depth]
            else:
                _lib.check(lib.ggpm_gru_sparse_forward(
                    E1, H, depth, P(h_prev), P(frz[t]), P(X[0]), P(X[1]), P(X[2]), P(Wz[:, I:]), Wz.stride(0), P(Ur),
                    Ur.stride(0), P(bu), P(Wh[:, I:]), Wh.stride(0), _vp(ptr[("pred_rp", t)]), _vp(ptr[("pred_col", t)]),
                    P(Hs[t]), P(Qs[t]), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(wpack), 1, s),
                    "gru_sparse_forward")
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
        ctx.plan, ctx.meta, ctx.drop = plan, (cell, depth, H, Fdim, I), drop
        ctx.save_for_backward(fn_all, hmess, X, Hs, Qs, St, NODE, NEI, *([Cs] if lstm else []), *params)
        ctx.keep = D
        return pooled, cand

    @staticmethod
    def backward(ctx, d_pooled, d_cand):
        lib = _lib.load()
        plan, (cell, depth, H, Fdim, I), drop = ctx.plan, ctx.meta, ctx.drop
        lstm = cell == "LSTM"
        sv = list(ctx.saved_tensors)
        fn_all, hmess, X, Hs, Qs, St, NODE, NEI = sv[:8]
        Cs = sv[8] if lstm else None
        params = sv[9:] if lstm else sv[8:]
        D = ctx.keep
        ptr, P = D["ptr"], F_._p
        dev = hmess.device
        Hp = F_.padded_hidden(H)
        E1, T = plan.E1, plan.T
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
        dX = torch.empty(G, E1, Hp, **f32)
        dH, dH2 = torch.zeros(E1, Hp, **f32), torch.empty(E1, Hp, **f32)
        dC, dC2 = (torch.zeros(E1, Hp, **f32), torch.empty(E1, Hp, **f32)) if lstm else (None, None)
        nh = 4 if lstm else 3                                   # hidden-half weight gradients (+ GRU: b_u)
        acc = [torch.zeros(H, H, **f32) for _ in range(nh)] + ([] if lstm else [torch.zeros(H, **f32)])
        tmp = [torch.empty(H, H, **f32) for _ in range(nh)] + ([] if lstm else [torch.empty(H, **f32)])
        wb = int((lib.ggpm_lstm_backward_workspace_bytes if lstm else lib.ggpm_gru_backward_workspace_bytes)(E1, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        ldwo = Wout.stride(0)
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
            st = St[t]
            if lstm:
                _lib.check(lib.ggpm_lstm_sparse_backward(
                    E1, H, depth, P(frz[t]), P(X[3]), P(Wi[:, I:]), Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0),
                    P(Wu[:, I:]), Wu.stride(0), P(Wf[:, I:]), Wf.stride(0), _vp(ptr[("pred_rp", t)]),
                    _vp(ptr[("pred_col", t)]), _vp(ptr[("succ_rp", t)]), _vp(ptr[("succ_col", t)]), P(Hs[t]), P(Cs[t]),
                    P(Qs[t]), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(dH), P(dC), P(dH2), P(dC2), P(dX[0]),
                    P(dX[1]), P(dX[2]), P(dX[3]), P(tmp[0]), H, P(tmp[1]), H, P(tmp[2]), H, P(tmp[3]), H, P(work),
                    work.numel() * 4, s), "lstm_sparse_backward")
                dC, dC2 = dC2, dC
            else:
                _lib.check(lib.ggpm_gru_sparse_backward(
                    E1, H, depth, P(frz[t]), P(X[1]), P(Wz[:, I:]), Wz.stride(0), P(Ur), Ur.stride(0), P(Wh[:, I:]),
                    Wh.stride(0), _vp(ptr[("pred_rp", t)]), _vp(ptr[("pred_col", t)]), _vp(ptr[("succ_rp", t)]),
                    _vp(ptr[("succ_col", t)]), P(Hs[t]), P(Qs[t]), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(dH),
                    P(dH2), P(dX[0]), P(dX[1]), P(dX[2]), P(tmp[0]), H, P(tmp[1]), H, P(tmp[3]), P(tmp[2]), H, P(work),
                    work.numel() * 4, s), "gru_sparse_backward")
            dH, dH2 = dH2, dH
            torch._foreach_add_([dX_tot] + acc, [dX] + tmp)
        # ---- parameter gradients, once
        return (None,) * 9 + _param_grads(lstm, params, acc, dX_tot, hmess, DPRE, NEI, fn_all, H, Hp, I, Fdim, E1, ns_tot)


class _AtomDecodeCompact(torch.autograd.Function):
    """Same outputs as ``_AtomDecode`` on compact row sets (module docstring, ``AtomPlan.compact_tables``): inside the step
    loop only ``gather frozen rows -> sparse_forward`` (backward: ``sparse_backward -> scatter-add to the rows' sources``);
    gate inputs of all steps gathered once, read-out of all steps batched behind the loop, weight gradients contracted
    once over the stacked stashes."""

    @staticmethod
    def launch(plan: AtomPlan, cell: str, depth: int, H: int, Fdim: int, I: int, fn_all, hmess, drop, params) -> dict:
        """Everything the forward computes, issued now (no autograd node): -> state for ``forward(..., state, *params)``.
        ``HierMPNDecoder.start_atom_level`` calls this BEFORE the encoder runs and creates the node later, behind the
        encoder's: the engine then reaches this node first in the backward pass and -- its loop being issued by the worker
        thread -- goes straight on to the encoder's backward, so that the longer chain starts first."""
        lib = _lib.load()
        dev = hmess.device
        D = plan.to_device(dev)
        ptr, P = D["ptr"], F_._p
        Hp = F_.padded_hidden(H)
        E1, T = plan.E1, plan.T
        f32 = dict(dtype=torch.float32, device=dev)
        lstm = cell == "LSTM"
        G = 4 if lstm else 3
        ct, cp = plan.compact_device(depth, G, dev)
        if lstm:
            Wi, bi, Wo_g, bo_g, Wu, bu_g, Wf, bf, Wout, bout = params
            gates = ((Wi, bi), (Wo_g, bo_g), (Wu, bu_g), (Wf, bf))
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh, Wout, bout = params
            gates = ((Wz, bz), (Wr, None), (Wh, bh))
        s = F_._stream()
        # hoisted gate input projections of ALL bond messages, then the rows every step needs, step by step
        X = torch.empty(G, E1, Hp, **f32)
        for k, (W, b) in enumerate(gates):
            F_.gemm(0, 1, E1, H, I, hmess, F_._ld(hmess), W, W.stride(0), X[k], Hp, Hp, bias=b)
        foff, Ftot = ct["foff"], ct["Ftot"]
        X_all = torch.empty(G * Ftot, Hp, **f32)
        _lib.check(lib.ggpm_gather_rows(P(X), Hp, _vp(cp["xrows"]), G * Ftot, Hp, P(X_all), Hp, 0, 0, s), "gather_rows")
        roff, qoff = plan.row_offsets(depth)
        Hs_all = torch.empty(qoff[-1], Hp, **f32)
        Cs_all = torch.empty(qoff[-1], Hp, **f32) if lstm else None
        Qs_all = torch.empty(roff[-1], Hp, **f32)
        St_all = torch.empty(5, roff[-1], Hp, **f32)
        wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H) if lstm else lib.ggpm_gru_pack_floats(H)), **f32)
        frz_loc = D["frozen_loc"].data_ptr()
        if lstm:
            hw = ((Wi, I), (Wo_g, I), (Wu, I), (Wf, I))
        else:
            hw = ((Wz, I), (Ur, 0), (Wh, I))
        W_arr = _lib.array_type(ctypes.c_void_p, 4)(*[w[:, c:].data_ptr() for w, c in hw])
        ld_arr = _lib.array_type(ctypes.c_int, 4)(*[w.stride(0) for w, _ in hw])
        deferred = False
        if _dev.DECODE_DRIVER:         # the whole step loop as one C call (csrc/decode.hip)
            desc, _keep = _decode_steps(plan, D, ct, cp, H, depth, lstm)
            tmp = torch.empty(2 * max(plan.nloc), Hp, **f32)
            fn = lib.ggpm_decode_steps_forward_async if _dev.ATOM_ASYNC else lib.ggpm_decode_steps_forward
            _lib.check(fn(
                ctypes.byref(desc), W_arr, ld_arr, None if lstm else P(bu), P(X_all), P(Hs_all), P(Cs_all) if lstm else None,
                P(Qs_all), P(St_all), St_all.stride(0), P(wpack), P(tmp), s), "decode_steps_forward")
            if _dev.ATOM_ASYNC:
                deferred = True
                _INFLIGHT.append((desc, _keep, X_all, Hs_all, Cs_all, Qs_all, St_all, wpack, tmp, params))
        for t in (() if _dev.DECODE_DRIVER else range(T)):
            n = plan.nloc[t]
            src = _vp(cp[("srcH", t)])
            h_in = torch.empty(n, Hp, **f32)
            _lib.check(lib.ggpm_gather_rows(P(Hs_all), Hp, src, n, Hp, P(h_in), Hp, 0, 0, s), "gather_rows")
            x = X_all[G * foff[t]:G * foff[t + 1]].view(G, n, Hp)
            hs, qs = Hs_all[qoff[t]:qoff[t + 1]], Qs_all[roff[t]:roff[t + 1]]
            st = St_all[:, roff[t]:roff[t + 1]]
            fz, rp, col = _vp(frz_loc + plan.floc_off[t]), _vp(ptr[("lpred_rp", t)]), _vp(ptr[("lpred_col", t)])
            if t > 0 and _dev.PACK_ONCE:
                lib.ggpm_weights_packed(1)          # same weights, same `wpack`: packed by the first step
            if lstm:
                c_in = torch.empty(n, Hp, **f32)
                _lib.check(lib.ggpm_gather_rows(P(Cs_all), Hp, src, n, Hp, P(c_in), Hp, 0, 0, s), "gather_rows")
                _lib.check(lib.ggpm_lstm_sparse_forward(
                    n, H, depth, P(h_in), P(c_in), fz, P(x[0]), P(x[1]), P(x[2]), P(x[3]), P(Wi[:, I:]),
                    Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0), P(Wu[:, I:]), Wu.stride(0), P(Wf[:, I:]), Wf.stride(0),
                    rp, col, P(hs), P(Cs_all[qoff[t]:qoff[t + 1]]), P(qs), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]),
                    P(wpack), 1, s), "lstm_sparse_forward")
            else:
                _lib.check(lib.ggpm_gru_sparse_forward(
                    n, H, depth, P(h_in), fz, P(x[0]), P(x[1]), P(x[2]), P(Wz[:, I:]), Wz.stride(0), P(Ur),
                    Ur.stride(0), P(bu), P(Wh[:, I:]), Wh.stride(0), rp, col, P(hs), P(qs), P(st[0]), P(st[1]), P(st[2]),
                    P(st[3]), P(st[4]), P(wpack), 1, s), "gru_sparse_forward")
        # ---- read-out of all steps at once: incoming messages (at their step's time) -> atoms -> clusters / candidates
        ns_tot, n_inst = plan.aoff[-1], plan.ioff[-1]
        NEI = torch.empty(ns_tot, Hp, **f32)
        NODE = torch.empty(ns_tot, Hp, **f32)
        pooled = torch.empty(n_inst, Hp, **f32)
        cand = torch.empty(max(plan.n_cand, 1), Hp, **f32)
        ldF, ldwo = F_._ld(fn_all), Wout.stride(0)
        stream_obj = torch.cuda.current_stream(dev)

        def readout():
            s_ = F_._stream()
            _lib.check(lib.ggpm_segment_sum(P(Hs_all), Hp, _vp(cp["agr_rp"]), _vp(cp["agr_col"]), ns_tot, H, P(NEI), Hp, 0, Hp, s_),
                       "segment_sum")
            F_.gemm_ksegments(1, ns_tot, H, [fn_all, NEI], [ldF, Hp], [Wout, Wout[:, Fdim:]], [ldwo, ldwo], [Fdim, H], NODE, Hp,
                              Hp, bias=bout, act=F_.ACT_RELU)
            if drop is not None:
                _lib.check(lib.ggpm_dropout(P(NODE), ns_tot, H, Hp, drop[0], drop[1], drop[2], 0, s_), "dropout")
            _lib.check(lib.ggpm_segment_sum(P(NODE), Hp, _vp(cp["pool_rp"]), _vp(cp["pool_col"]), n_inst, H, P(pooled), Hp, 0, Hp,
                                            s_), "segment_sum")
            _lib.check(lib.ggpm_gather_rows(P(NODE), Hp, _vp(cp["cand_idx"]), max(plan.n_cand, 1), H, P(cand), Hp, 0, Hp, s_),
                       "gather_rows")

        finish = None
        if deferred:        # the loop is still being issued by the worker: the read-out follows it after the join
            def finish():
                _join_worker("decode_join (forward)")
                with torch.cuda.stream(stream_obj):
                    readout()
        else:
            readout()
        return dict(pooled=pooled, cand=cand, finish=finish, keep=(D, ct, cp), lstm=lstm,
                    saved=(fn_all, hmess, NODE, NEI, X_all, Hs_all, Qs_all, St_all) + ((Cs_all,) if lstm else ()))

    @staticmethod
    def forward(ctx, plan: AtomPlan, cell: str, depth: int, H: int, Fdim: int, I: int, fn_all, hmess, drop, state, *params):
        if state is None:
            state = _AtomDecodeCompact.launch(plan, cell, depth, H, Fdim, I, fn_all, hmess, drop, params)
            if state["finish"] is not None:
                _PENDING[id(plan)] = state["finish"]
        ctx.plan, ctx.meta, ctx.drop = plan, (cell, depth, H, Fdim, I), drop
        ctx.save_for_backward(*state["saved"], *params)
        ctx.keep = state["keep"]
        ctx.params_ref = params            # the Parameter objects themselves (the async backward assigns their .grad)
        return state["pooled"], state["cand"]

    @staticmethod
    def backward(ctx, d_pooled, d_cand):
        lib = _lib.load()
        # everything upstream of the atom level (heads, the two tree-side levels) has run its backward: their queued
        # weight-gradient contractions can start beside this node and the encoder's backward.  They are ~45 launches of
        # host time: with the step loop handed to the worker thread they are issued AFTER the loop has been posted (below),
        # so that the longest chain of the pass starts first; without the worker, here.
        F_.mark("bwd: atom level's node reached")
        flush_after_post = _dev.DECODE_DRIVER and _dev.ATOM_ASYNC
        if not flush_after_post:
            F_.flush_deferred_early()
        plan, (cell, depth, H, Fdim, I), drop = ctx.plan, ctx.meta, ctx.drop
        lstm = cell == "LSTM"
        sv = list(ctx.saved_tensors)
        fn_all, hmess, NODE, NEI, X_all, Hs_all, Qs_all, St_all = sv[:8]
        Cs_all = sv[8] if lstm else None
        params = sv[9:] if lstm else sv[8:]
        D, ct, cp = ctx.keep
        ptr, P = D["ptr"], F_._p
        dev = hmess.device
        Hp = F_.padded_hidden(H)
        E1, T = plan.E1, plan.T
        f32 = dict(dtype=torch.float32, device=dev)
        G = 4 if lstm else 3
        if lstm:
            Wi, bi, Wo_g, bo_g, Wu, bu_g, Wf, bf, Wout, bout = params
        else:
            Wz, bz, Wr, Ur, bu, Wh, bh, Wout, bout = params
        d_pooled, d_cand = d_pooled.contiguous(), d_cand.contiguous()
        s = F_._stream()
        foff, Ftot = ct["foff"], ct["Ftot"]
        roff, qoff = plan.row_offsets(depth)
        ns_tot, ldwo = plan.aoff[-1], Wout.stride(0)
        # ---- read-out of all steps, backwards: clusters / candidates -> atoms -> the final states they read
        d_node = torch.empty(ns_tot, Hp, **f32)
        _lib.check(lib.ggpm_segment_sum(P(d_pooled), Hp, _vp(cp["poolT_rp"]), _vp(cp["poolT_col"]), ns_tot, H, P(d_node), Hp, 0,
                                        Hp, s), "segment_sum")
        if plan.n_cand > 0:
            _lib.check(lib.ggpm_segment_sum(P(d_cand), Hp, _vp(cp["candT_rp"]), _vp(cp["candT_col"]), ns_tot, H, P(d_node), Hp,
                                            1, 0, s), "segment_sum")
        DPRE = torch.empty(ns_tot, Hp, **f32)
        _lib.check(lib.ggpm_act_backward(P(d_node), P(NODE), ns_tot, H, Hp, F_.ACT_RELU, 0, P(DPRE), s), "act_backward")
        if drop is not None:                # d(dropout . relu) = mask * scale * relu' (the saved output is the dropped one)
            _lib.check(lib.ggpm_dropout(P(DPRE), ns_tot, H, Hp, drop[0], drop[1], drop[2], 0, s), "dropout")
        d_nei = torch.empty(ns_tot, Hp, **f32)
        F_.gemm(0, 0, ns_tot, H, H, DPRE, Hp, Wout[:, Fdim:], ldwo, d_nei, Hp, Hp)
        dF = torch.empty(Ftot, Hp, **f32)            # gradient of every step's final states (zero where nothing reads them)
        _lib.check(lib.ggpm_segment_sum(P(d_nei), Hp, _vp(cp["agrT_rp"]), _vp(cp["agrT_col"]), Ftot, H, P(dF), Hp, 0, Hp, s),
                   "segment_sum")
        dCF = torch.zeros(Ftot, Hp, **f32) if lstm else None
        dX_all = torch.empty(G * Ftot, Hp, **f32)
        # gate-gradient stashes of all steps, stacked like the forward's (DQ with a zero slot per step at the end so that it
        # lines up with the depth + 1 state slots): contracted once behind the loop
        DG_all = torch.empty(3 if lstm else 2, roff[-1], Hp, **f32)
        DQ_all = torch.zeros(qoff[-1], Hp, **f32)
        nh = 4 if lstm else 3                                   # hidden-half weight gradients (+ GRU: b_u)
        # the [H, I + H] gradient buffers of the gate weights exist from the start: the stacked contractions write their
        # hidden halves in place (leading dimension I + H), the grouped input-half launch fills the rest -- no copies
        gate_ws = (Wi, Wo_g, Wu, Wf) if lstm else (Wz, Wr, Wh)
        bufs = [torch.empty_like(w) for w in gate_ws]
        if lstm:
            acc = [b[:, I:] for b in bufs]
        else:           # acc order of the GRU: Wz_h, U_r, Wh_h, b_u (W_r has no hidden half: U_r is its own matrix)
            acc = [bufs[0][:, I:], torch.empty(H, H, **f32), bufs[2][:, I:], torch.empty(H, **f32)]
        nmax = max(plan.nloc)
        wb = int((lib.ggpm_lstm_backward_workspace_bytes if lstm else lib.ggpm_gru_backward_workspace_bytes)(nmax, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        frz_loc = D["frozen_loc"].data_ptr()
        params_ref = ctx.params_ref
        go_async = (_dev.DECODE_DRIVER and _dev.ATOM_ASYNC and F_.can_publish(*params_ref) and all(ctx.needs_input_grad[10:]))
        if _dev.DECODE_DRIVER:         # the whole step loop as one C call (csrc/decode.hip)
            if lstm:
                hw = ((Wi, I), (Wo_g, I), (Wu, I), (Wf, I))
            else:
                hw = ((Wz, I), (Ur, 0), (Wh, I))
            W_arr = _lib.array_type(ctypes.c_void_p, 4)(*[w[:, c:].data_ptr() for w, c in hw])
            ld_arr = _lib.array_type(ctypes.c_int, 4)(*[w.stride(0) for w, _ in hw])
            desc, _keep = _decode_steps(plan, D, ct, cp, H, depth, lstm)
            tmp = torch.empty(2 * nmax, Hp, **f32)
            dW_arr = _lib.array_type(ctypes.c_void_p, 4)(*([a.data_ptr() for a in acc] + ([] if len(acc) == 4 else [0])))
            fn = lib.ggpm_decode_steps_backward_async if go_async else lib.ggpm_decode_steps_backward
            _lib.check(fn(
                ctypes.byref(desc), W_arr, ld_arr, P(X_all), P(Hs_all), P(Cs_all) if lstm else None, P(Qs_all), P(St_all),
                St_all.stride(0), P(dF), P(dCF) if lstm else None, P(dX_all), P(DG_all), DG_all.stride(0), P(DQ_all), dW_arr,
                P(work), work.numel() * 4, P(tmp), s), "decode_steps_backward")
            if go_async:
                _INFLIGHT.append((desc, _keep, sv, dF, dCF, dX_all, DG_all, DQ_all, acc, work, tmp))
        F_.mark("bwd: atom loop posted")
        if flush_after_post:
            F_.flush_deferred_early()
        F_.mark("bwd: atom node returns")
        for t in (() if _dev.DECODE_DRIVER else range(T - 1, -1, -1)):
            n = plan.nloc[t]
            dhd, dhin = dF[foff[t]:foff[t + 1]], torch.empty(n, Hp, **f32)
            dx = dX_all[G * foff[t]:G * foff[t + 1]].view(G, n, Hp)
            xg = X_all[G * foff[t]:G * foff[t + 1]].view(G, n, Hp)[3 if lstm else 1]
            hs, qs, st = Hs_all[qoff[t]:qoff[t + 1]], Qs_all[roff[t]:roff[t + 1]], St_all[:, roff[t]:roff[t + 1]]
            fz = _vp(frz_loc + plan.floc_off[t])
            csr = (_vp(ptr[("lpred_rp", t)]), _vp(ptr[("lpred_col", t)]), _vp(ptr[("lsucc_rp", t)]), _vp(ptr[("lsucc_col", t)]))
            srcF = _vp(cp[("srcF", t)])
            if t < T - 1 and _dev.PACK_ONCE:
                lib.ggpm_weights_packed(1)          # same weights, same `work`: the transposes were packed by the first call
            if lstm:
                dcin = torch.empty(n, Hp, **f32)
                lib.ggpm_backward_defer_stash(P(DG_all[0, roff[t]:]), P(DG_all[1, roff[t]:]), P(DG_all[2, roff[t]:]),
                                              P(DQ_all[qoff[t]:]))
                _lib.check(lib.ggpm_lstm_sparse_backward(
                    n, H, depth, fz, P(xg), P(Wi[:, I:]), Wi.stride(0), P(Wo_g[:, I:]), Wo_g.stride(0), P(Wu[:, I:]),
                    Wu.stride(0), P(Wf[:, I:]), Wf.stride(0), *csr, P(hs), P(Cs_all[qoff[t]:qoff[t + 1]]), P(qs), P(st[0]),
                    P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(dhd), P(dCF[foff[t]:foff[t + 1]]), P(dhin), P(dcin), P(dx[0]),
                    P(dx[1]), P(dx[2]), P(dx[3]), P(acc[0]), H, P(acc[1]), H, P(acc[2]), H, P(acc[3]), H, P(work),
                    work.numel() * 4, s), "lstm_sparse_backward")
                _lib.check(lib.ggpm_scatter_rows(P(dcin), Hp, srcF, n, Hp, P(dCF), Hp, 1, s), "scatter_rows")
            else:
                lib.ggpm_backward_defer_stash(P(DG_all[0, roff[t]:]), P(DG_all[1, roff[t]:]), P(DQ_all[qoff[t]:]), None)
                _lib.check(lib.ggpm_gru_sparse_backward(
                    n, H, depth, fz, P(xg), P(Wz[:, I:]), Wz.stride(0), P(Ur), Ur.stride(0), P(Wh[:, I:]), Wh.stride(0),
                    *csr, P(hs), P(qs), P(st[0]), P(st[1]), P(st[2]), P(st[3]), P(st[4]), P(dhd), P(dhin), P(dx[0]),
                    P(dx[1]), P(dx[2]), P(acc[0]), H, P(acc[1]), H, P(acc[3]), P(acc[2]), H, P(work), work.numel() * 4, s),
                    "gru_sparse_backward")
            # the frozen rows' gradient goes to the step that produced their state (rows recomputed here: none)
            _lib.check(lib.ggpm_scatter_rows(P(dhin), Hp, srcF, n, Hp, P(dF), Hp, 1, s), "scatter_rows")
        # ---- parameter gradients, once
        def tail():
            s_ = F_._stream()
            dX_tot = torch.empty(G, E1, Hp, **f32)
            _lib.check(lib.ggpm_segment_sum(P(dX_all), Hp, _vp(cp["xT_rp"]), _vp(cp["xT_col"]), G * E1, Hp, P(dX_tot), Hp, 0, 0, s_),
                       "segment_sum")
            R, RQ = roff[-1], qoff[-1]
            wsb = int(lib.ggpm_weight_grads_stacked_workspace_bytes(H, max(R, RQ)))
            ws = torch.empty((wsb + 3) // 4, **f32)
            ld = [a.stride(0) for a in acc[:nh]]
            if lstm:        # acc: Wi_h, Wo_h, Wu_h, Wf_h; St_all[0] = S
                _lib.check(lib.ggpm_lstm_weight_grads_stacked(
                    R, RQ, H, P(DG_all[0]), P(DG_all[1]), P(DG_all[2]), P(St_all[0]), P(DQ_all), P(Hs_all), P(acc[0]), ld[0],
                    P(acc[1]), ld[1], P(acc[2]), ld[2], P(acc[3]), ld[3], P(ws), ws.numel() * 4, s_), "lstm_weight_grads_stacked")
            else:           # acc: Wz_h, U_r, Wh_h, b_u; St_all: S, G, Z, M, R
                _lib.check(lib.ggpm_gru_weight_grads_stacked(
                    R, RQ, H, P(DG_all[0]), P(St_all[1]), P(DG_all[1]), P(St_all[0]), P(DQ_all), P(Hs_all), P(acc[0]), ld[0],
                    P(acc[1]), ld[1], P(acc[3]), P(acc[2]), ld[2], P(ws), ws.numel() * 4, s_), "gru_weight_grads_stacked")
            return _param_grads(lstm, params, acc, dX_tot, hmess, DPRE, NEI, fn_all, H, Hp, I, Fdim, E1, ns_tot, bufs=bufs)

        if go_async:
            # The loop is being issued by the worker; this node returns now so that the engine can issue the encoder's
            # backward beside it.  At the end of the pass: join the worker, enqueue the tail behind the loop on the atom
            # level's stream, hand the gradients to .grad on the stream the rest of the step consumes them on.
            atom_stream = torch.cuda.current_stream(dev)

            def finish():
                F_.mark("bwd: end-of-pass callback")
                _join_worker("decode_join (backward)")
                main = torch.cuda.current_stream(dev)
                with torch.cuda.stream(atom_stream):
                    F_.mark("bwd: atom loop done (its stream)")
                    grads = tail()
                    F_.mark("bwd: atom level's tail issued")
                main.wait_stream(atom_stream)
                for prm, g in zip(params_ref, grads):
                    F_.hand_to(g, main)
                    F_._add_to_grad(prm, g)
            torch.autograd.Variable._execution_engine.queue_callback(finish)
            return (None,) * (10 + len(params))
        return (None,) * 10 + tail()


def _param_grads(lstm, params, acc, dX_tot, hmess, DPRE, NEI, fn_all, H, Hp, I, Fdim, E1, ns_tot, bufs=None):
    """Parameter gradients of the atom-level decode from the summed gate-input gradients, the accumulated hidden halves
    and the stacked read-out rows (shared by both forms)."""
    if lstm:
        Wi, bi, Wo_g, bo_g, Wu, bu_g, Wf, bf, Wout, bout = params
    else:
        Wz, bz, Wr, Ur, bu, Wh, bh, Wout, bout = params
    x_ld = F_._ld(hmess)

    # input halves dW_k[:, :I] = dX_k^T hmess of all gates in ONE grouped launch (as the encoder's drivers form them)
    gate_ws = (Wi, Wo_g, Wu, Wf) if lstm else (Wz, Wr, Wh)
    in_place = bufs is not None             # the hidden halves are already there (written by the stacked contractions)
    if bufs is None:
        bufs = [torch.empty_like(W) for W in gate_ws]
    F_.gemm_grouped(1, 0, H, I, E1, [dict(A=dX_tot[k], lda=Hp, B=hmess, ldb=x_ld, C=b, ldc=b.stride(0), n_pad=I)
                                      for k, b in enumerate(bufs)], splitk=True)

    def full(W, k, hidden):                 # [input half from the summed dX | accumulated hidden half]
        dW = bufs[k]
        if hidden is not None and not in_place:
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
    else:          # acc order of the GRU: Wz_h, U_r, Wh_h, b_u
        grads = [full(Wz, 0, acc[0]), F_.colsum(dX_tot[0], E1, H), full(Wr, 1, None), acc[1], acc[3],
                 full(Wh, 2, acc[2]), F_.colsum(dX_tot[2], E1, H), dWout, dbout]
    return tuple(grads)


def atom_decode_node(pre: dict):
    """The autograd node of a prelaunched atom level (``atom_decode(..., prelaunch=True)``), created where the decoder
    joins it: -> (pooled, cand).  Joins the worker and enqueues the read-out first."""
    fin = pre["state"]["finish"]
    if fin is not None:
        fin()
        pre["state"]["finish"] = None
    return _AtomDecodeCompact.apply(*pre["args"], pre["state"], *pre["params"])


def atom_decode(plan: AtomPlan, graph_encoder, hnode_a: torch.Tensor, hmess_a: torch.Tensor, fn_all: torch.Tensor,
                defer_finish: bool = False, prelaunch: bool = False):
    """-> (pooled [n_inst, Hp], cand [n_cand, Hp]) for ``graph_encoder`` = the decoder's atom-level ``IncMPNEncoder``.
    ``defer_finish``: -> (pooled, cand, finish); the caller calls ``finish()`` before it reads the two tensors or lets any
    other stream wait on the current one (with _dev.ATOM_ASYNC the step loop is still being issued by a worker thread
    when this returns, and the read-out behind it is enqueued by ``finish``)."""
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
    fn = _AtomDecodeCompact if compact_enabled() else _AtomDecode
    if fn is _AtomDecode and not plan.full:
        raise RuntimeError("this AtomPlan was built without the level-wide tables (_dev.ATOM_COMPACT changed after the plan "
                           "was built?); build it with full=True")
    args = (plan, "LSTM" if lstm else "GRU", rnn.depth, rnn.hidden_size, graph_encoder.node_fdim, rnn.input_size, fn_all,
            hmess_a, drop)
    params = params + (wo[0].weight, wo[0].bias)
    if prelaunch:           # issue now, create the autograd node later (atom_decode_node)
        assert fn is _AtomDecodeCompact
        with torch.no_grad():
            return dict(args=args, params=params, state=fn.launch(*args, params))
    pooled, cand = fn.apply(*args, *params) if fn is _AtomDecode else fn.apply(*args, None, *params)
    finish = _PENDING.pop(id(plan), None)        # set when the step loop was handed to the library's worker thread
    if defer_finish:
        return pooled, cand, (finish or (lambda: None))
    if finish is not None:
        finish()
    return pooled, cand
