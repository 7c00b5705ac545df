"""ctypes front end of csrc/schedule.hip: the decoder's per-batch integer bookkeeping built in C++ (host only).

``build_tables`` takes what ``DecodeSchedule.from_tensors`` takes and returns a :class:`NativeTables`: named integer
tables as numpy views into the library's buffers, the two device packs (one int64, one int32 upload per batch) and the
directory that says where each device table sits inside its pack.  The numpy builders in ``decoder.py`` /
``atom_decode.py`` remain the statement of what the tables mean and the checker (tests/test_schedule_native.py).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Tuple

import numpy as np

from . import _lib

_P64 = ctypes.POINTER(ctypes.c_int64)
_view = ctypes.pythonapi.PyMemoryView_FromMemory          # (address, bytes, PyBUF_WRITE = 0x200) -> memoryview, no copy
_view.restype, _view.argtypes = ctypes.py_object, (ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int)


class SchedIn(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_sched_in"""
    _fields_ = [(k, ctypes.c_int) for k in ("B", "Nt1", "Et1", "At", "Kt", "C", "Ng1", "Eg1", "Ag", "Kg", "depth", "gates")] + \
               [(k, ctypes.c_void_p) for k in ("tfnode", "tfmess", "tagraph", "tbgraph", "cgraph", "tree_scope", "gfmess",
                                               "gagraph", "gbgraph", "orders", "order_off", "icls_off", "icls", "cand_off",
                                               "cand_atom_off", "cands")]


def enabled() -> bool:
    return os.environ.get("GGPM_NATIVE_SCHEDULE", "1") != "0"


def _i64(x) -> np.ndarray:
    if hasattr(x, "detach"):                      # a torch tensor (host): no copy when it already is int64
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=np.int64)


class NativeTables:
    """Owner of one ``ggpm_schedule_build`` result.  ``get(name)`` -> numpy view (valid while this object lives)."""

    _ELEM = {8: np.int64, 4: np.int32, 1: np.uint8}

    def __init__(self, handle, lib):
        self._h, self._lib = handle, lib
        cap = 1 << 13
        buf = _lib.array_type(ctypes.c_char, cap)()       # (create_string_buffer builds the array type anew per call)
        _lib.check(lib.ggpm_schedule_names(handle, buf, cap), "schedule_names")
        self.names = buf.value.decode().split("\n")[:-1]
        table = np.empty(4 * len(self.names), dtype=np.int64)
        n = lib.ggpm_schedule_directory(handle, table.ctypes.data, table.size)
        if n != len(self.names):
            raise RuntimeError("ggpm_schedule_directory: %d" % n)
        rows = table.reshape(-1, 4).tolist()
        self.dir: Dict[str, Tuple[int, int, int, int]] = {k: tuple(r) for k, r in zip(self.names, rows)}   # (pack, byte offset, count, elem)
        self._arr: Dict[str, np.ndarray] = {}
        self._bytes = {}
        ptr, nb = ctypes.c_void_p(), ctypes.c_int64()
        for which in (0, 1, 2):
            _lib.check(lib.ggpm_schedule_pack(handle, which, ctypes.byref(ptr), ctypes.byref(nb)), "schedule_pack")
            if nb.value and ptr.value:
                # (a memoryview over the library's buffer; ``(c_uint8 * n).from_address`` would create a new ctypes array TYPE
                # per distinct size -- three classes per batch, each a reference cycle left to Python's collector)
                self._bytes[which] = np.frombuffer(_view(ptr.value, nb.value, 0x200), dtype=np.uint8)
            else:
                self._bytes[which] = np.zeros(0, dtype=np.uint8)
        self.packs = {1: self._bytes[1].view(np.int64), 2: self._bytes[2].view(np.int32)}

    def get(self, name: str) -> np.ndarray:
        a = self._arr.get(name)
        if a is None:
            pack, off, cnt, el = self.dir[name]
            a = self._arr[name] = self._bytes[pack][off:off + cnt * el].view(self._ELEM[el])
        return a

    def has(self, name: str) -> bool:
        return name in self.dir

    def scalars(self) -> dict:
        keys = ("T", "n_inst", "E1", "chain", "all_live", "max_cls_size", "n_assm", "n_cand", "Ftot", "ok", "B", "depth",
                "gates", "Ng1", "Eg1")
        return dict(zip(keys, self.get("scalars").tolist()))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                self._lib.ggpm_schedule_free(h)
            except Exception:
                pass

    def __getstate__(self):
        raise TypeError("NativeTables holds a library handle; pickle the DecodeSchedule (it converts itself)")


class _View:
    """read-only ``mapping[v]`` over a function of the node id (what DecodeSchedule keeps as its labels)"""

    def __init__(self, fn):
        self._fn = fn

    def __getitem__(self, v):
        return self._fn(int(v))

    def get(self, v, default=None):
        try:
            return self._fn(int(v))
        except IndexError:
            return default


class FlatLabels:
    """Per tree node v the two labels that are not in the tensors, flat, as ``ggpm_schedule_build`` reads them: the attachment
    ids of its inter_label (``icls[icls_off[v]:icls_off[v+1]]``, k_v of them) and its assm_cands as ``cand_off[v+1] -
    cand_off[v]`` candidates of k_v atoms each from ``cands[cand_atom_off[v]:]``."""

    def __init__(self, icls_n, cand_n, atom_n, icls_flat, cand_flat):
        self.icls_off, self.cand_off, self.cand_atom_off = (np.cumsum(np.asarray(x, dtype=np.int64)) for x in (icls_n, cand_n, atom_n))
        self.icls, self.cands = np.asarray(icls_flat, dtype=np.int64), np.asarray(cand_flat, dtype=np.int64)
        self.ok = self.cands.size == self.cand_atom_off[-1] and self.icls.size == self.icls_off[-1]

    def icls_of(self, v: int) -> tuple:
        return tuple(self.icls[self.icls_off[v]:self.icls_off[v + 1]].tolist())

    def cands_of(self, v: int) -> np.ndarray:
        n, k = int(self.cand_off[v + 1] - self.cand_off[v]), int(self.icls_off[v + 1] - self.icls_off[v])
        a0 = int(self.cand_atom_off[v])
        return self.cands[a0:a0 + n * k].reshape(n, max(k, 1)) if n else np.zeros((0, max(k, 1)), dtype=np.int64)

    def views(self):
        """(inter_icls, assm_cands) as mappings over the node id"""
        return _View(self.icls_of), _View(self.cands_of)


def labels_from_dicts(inter_icls: Dict[int, Tuple[int, ...]], assm_cands: Dict[int, list], Nt1: int) -> "FlatLabels | None":
    """None: a node whose candidates do not have one atom per attachment id (the numpy builder raises the informative error)."""
    icls_n, cand_n, atom_n = [0] * (Nt1 + 1), [0] * (Nt1 + 1), [0] * (Nt1 + 1)
    icls_flat, cand_flat = [], []
    for v in range(Nt1):
        ic = inter_icls.get(v)
        if ic:
            icls_n[v + 1] = len(ic)
            icls_flat.extend(ic)
        c = assm_cands.get(v)
        if c is not None and len(c):
            k = icls_n[v + 1]
            if isinstance(c, np.ndarray):
                c = c.reshape(len(c), -1).tolist()
            if isinstance(c[0], (list, tuple)):
                if len(c[0]) != k:
                    return None
                for tup in c:
                    cand_flat.extend(tup)
            else:
                if k != 1:
                    return None
                cand_flat.extend(c)
            cand_n[v + 1] = len(c)
            atom_n[v + 1] = len(c) * k
    out = FlatLabels(icls_n, cand_n, atom_n, icls_flat, cand_flat)
    return out if out.ok else None


def labels_from_graph(tree_batch, vocab, Nt1: int) -> "FlatLabels | None":
    """The same arrays read straight off the networkx nodes of a batch (``MolGraph.tensorize``'s ``tree_batchG``: attributes
    ``smiles``, ``inter_label``, ``assm_cands``) in ONE pass -- no per-node tuples and dicts in between.  None when the nodes
    do not come in ascending id order or a node is not of the expected shape: the caller then takes the dictionary path."""
    from itertools import chain
    icls_n, cand_n, atom_n = [0] * (Nt1 + 1), [0] * (Nt1 + 1), [0] * (Nt1 + 1)
    icls_flat, cand_flat = [], []
    add_icls, add_cand = icls_flat.extend, cand_flat.extend
    prev = -1
    for v, attr in tree_batch.nodes(data=True):
        if not (prev < v < Nt1):
            return None
        prev = v
        il = attr["inter_label"]
        k = len(il)
        if k:
            cls = attr["smiles"]
            add_icls([vocab[(cls, ic)][1] for _, ic in il])
            icls_n[v + 1] = k
        c = attr["assm_cands"]
        n = len(c)
        if n:
            if isinstance(c, np.ndarray):
                return None
            if isinstance(c[0], (list, tuple)):
                if len(c[0]) != k:
                    return None
                add_cand(chain.from_iterable(c))
            else:
                if k != 1:
                    return None
                add_cand(c)
            cand_n[v + 1] = n
            atom_n[v + 1] = n * k
    out = FlatLabels(icls_n, cand_n, atom_n, icls_flat, cand_flat)
    return out if out.ok else None


def build_tables(tensors, orders, inter_icls, assm_cands=None, depth: int = 0, gates: int = 0) -> "NativeTables | None":
    """-> NativeTables, or None when the library declines the batch (malformed orders / labels: the numpy builder then
    raises the informative error).  ``inter_icls``: a :class:`FlatLabels`, or the two dictionaries of the numpy builder."""
    lib = _lib.load()
    tree, graph = tensors
    tfnode, tfmess, tagraph, tbgraph, cgraph = (_i64(x) for x in tree[:5])
    gfmess, gagraph, gbgraph = _i64(graph[1]), _i64(graph[2]), _i64(graph[3])
    if tfnode.ndim != 2 or tfnode.shape[1] != 2 or tfmess.ndim != 2 or tfmess.shape[1] != 4 or gfmess.shape[1] != 4:
        return None
    scope = np.ascontiguousarray(np.asarray(tree[-1], dtype=np.int64).reshape(-1, 2))
    B, Nt1 = len(orders), tfnode.shape[0]
    flat = [-1 if v is None else v for o in orders for step in o for v in step]
    od = np.asarray(flat, dtype=np.int64).reshape(-1, 3)
    ooff = np.zeros(B + 1, dtype=np.int64)
    np.cumsum([len(o) for o in orders], out=ooff[1:])
    labels = inter_icls if isinstance(inter_icls, FlatLabels) else labels_from_dicts(inter_icls, assm_cands, Nt1)
    if labels is None or labels.icls_off.shape[0] != Nt1 + 1:
        return None
    icls_off, cand_off, cand_atom_off, icls, cands = labels.icls_off, labels.cand_off, labels.cand_atom_off, labels.icls, labels.cands
    keep = (tfnode, tfmess, tagraph, tbgraph, cgraph, scope, gfmess, gagraph, gbgraph, od, ooff, icls_off, icls, cand_off,
            cand_atom_off, cands)
    p = lambda a: a.ctypes.data if a.size else 0
    si = SchedIn(B, Nt1, tfmess.shape[0], tagraph.shape[1], tbgraph.shape[1], cgraph.shape[1], gagraph.shape[0],
                 gfmess.shape[0], gagraph.shape[1], gbgraph.shape[1], int(depth), int(gates), *[p(a) for a in keep])
    h = lib.ggpm_schedule_build(ctypes.byref(si))
    del keep
    return NativeTables(h, lib) if h else None
