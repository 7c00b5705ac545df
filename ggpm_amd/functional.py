"""torch.autograd wrappers over the C ABI (include/ggpm_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the stream and the autograd tape;
every number on the hot path is produced by a hand-written HIP kernel from libggpm_hip.so.  There is no CPU
or eager fallback: on a tensor that is not on a ROCm device these functions raise.

Conventions: feature matrices are 2-D fp32 tensors whose row stride may exceed the logical width; pad
columns are zero.  Index tensors are int32 (CSR) unless stated.
"""
from __future__ import annotations

import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch

from . import _dev, _lib

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3


def _stream() -> ctypes.c_void_p:
    # raw handle of torch's current stream on the current device (torch.cuda.current_stream() builds a Python
    # Stream object: ~9 us per call, and a step makes ~1000 of them)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _need_gpu(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ggpm_amd: tensors must live on the MI355X (got a %s tensor); there is no CPU path"
                               % t.device.type)


def padded_hidden(H: int) -> int:
    return (H + 15) // 16 * 16


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D tensor expected"
    return t.stride(0)


def _empty_same_layout(x: torch.Tensor) -> torch.Tensor:
    """An uninitialised [rows, cols] tensor with x's leading dimension.  (``torch.empty_like`` of a column slice of a
    padded buffer -- e.g. the [rows, H + 20] view of a [rows, ld] message-input buffer when H + 20 is not a multiple
    of 4 -- returns a DENSE tensor, whose leading dimension is not x's.)"""
    return torch.empty(x.shape[0], _ld(x), dtype=x.dtype, device=x.device)[:, :x.shape[1]]


# ----------------------------------------------------------------------------- graph layout
class CSR:
    """Device CSR (int32 rowptr[rows+1], col[cap]) plus its lazily built transpose."""

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, rows: int, ncols: int):
        self.rowptr, self.col, self.rows, self.ncols = rowptr, col, rows, ncols
        self._T: Optional["CSR"] = None
        self._back = None           # on a transpose: weak reference to the CSR it was built from (a strong one would make the
        self._T_event = None        # pair a reference cycle, and its device tensors would wait for Python's collector)
        self._keep = None

    def _build_T(self) -> "CSR":
        lib = _lib.load()
        dev = self.col.device
        rowptrT = torch.empty(self.ncols + 1, dtype=torch.int32, device=dev)
        colT = torch.empty(max(self.col.numel(), 1), dtype=torch.int32, device=dev)
        cursor = torch.empty(self.ncols, dtype=torch.int32, device=dev)
        _lib.check(lib.ggpm_csr_transpose(_p(self.rowptr), _p(self.col), self.rows, self.ncols, _p(rowptrT),
                                          _p(colT), _p(cursor), _stream()), "csr_transpose")
        t = CSR(rowptrT, colT, self.ncols, self.rows)
        t._back = weakref.ref(self)
        t._keep = cursor
        return t

    @property
    def T(self) -> "CSR":
        if self._T is None:
            back = self._back() if self._back is not None else None
            if back is not None:
                return back
            self._T = self._build_T()
        ev = self._T_event
        if ev is not None:          # built ahead of time on the second stream: order this stream behind it once
            torch.cuda.current_stream().wait_event(ev)
            self._T_event = None
        return self._T


def prefetch_transposes(csrs: Sequence["CSR"]) -> None:
    """Build the transposes the BACKWARD will need on the second stream while the forward runs (each is a
    single-workgroup integer kernel of ~10 us that would otherwise sit on the backward's critical path)."""
    todo = [c for c in csrs if c is not None and c._T is None and (c._back is None or c._back() is None)]
    if not todo or not side_stream_enabled():
        return
    main = torch.cuda.current_stream()
    side = _side_stream(todo[0].col.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for c in todo:
            c._T = c._build_T()
            for t in (c.rowptr, c.col):
                t.record_stream(side)
        ev = torch.cuda.Event()
        ev.record(side)
    for c in todo:
        for t in (c._T.rowptr, c._T.col, c._T._keep):
            t.record_stream(main)
        c._T_event = ev


class _StagingRing:
    """Pinned staging for host -> device uploads of numpy tables (decode schedules, atom plans): ``pin_memory()`` on a
    fresh tensor is a cudaHostAlloc of several milliseconds per call -- more than the copy -- so the bytes go through a
    small ring of pinned buffers that are allocated once and grow on demand."""

    def __init__(self, slots: int = 8):      # (three uploads per batch in the vae_train.py call shape: a slot is reused 2-3 batches later)
        self.slots, self.bufs, self.events, self.i = slots, [None] * slots, [None] * slots, 0

    def upload(self, a, device) -> torch.Tensor:
        import numpy as np
        a = np.ascontiguousarray(a)
        out = torch.empty(a.shape, dtype=torch.from_numpy(a[:0].reshape(0)).dtype, device=device)
        if a.size == 0:
            return out
        if torch.device(device).type != "cuda":
            out.copy_(torch.from_numpy(a))
            return out
        k = self.i
        self.i = (self.i + 1) % self.slots
        if self.events[k] is not None:
            self.events[k].synchronize()          # slot reuse: its previous copy must have left the buffer
        if self.bufs[k] is None or self.bufs[k].numel() < a.nbytes:
            self.bufs[k] = torch.empty(max(a.nbytes, 1 << 20), dtype=torch.uint8).pin_memory()
        stage = self.bufs[k][:a.nbytes]
        # a plain memcpy into the pinned buffer: torch's CPU copy_ fans a 1 MB copy out over every visible core, whose
        # OpenMP team then spins beside the training thread (on a box with a 16-CPU quota the whole process gets
        # throttled: 35 instead of 12 ms per step when the decode schedule is uploaded inside the step)
        np.copyto(stage.numpy(), a.reshape(-1).view(np.uint8))
        out.view(-1).view(torch.uint8).copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[k] = ev
        return out


_STAGING = _StagingRing()


def upload(a, device) -> torch.Tensor:
    """numpy array -> device tensor of the same dtype and shape through the pinned staging ring (asynchronous)."""
    return _STAGING.upload(a, device)


_MEMO_ON = None        # None: _dev.INDEX_MEMO, read at call time; True / False: forced (bench.py's "index structures rebuilt" leg)


def _memo_get(t: torch.Tensor, slot: str, key):
    """Index structures derived from an index tensor are remembered ON that tensor object, keyed by its in-place
    version counter: a batch whose tables stay resident -- a DecodeSchedule on the device, the encoder's graph
    tensors -- pays for their CSRs and transposes once, not once per step."""
    if not (_dev.INDEX_MEMO if _MEMO_ON is None else _MEMO_ON):
        return None
    m = getattr(t, slot, None)
    # `_version` counts in-place writes: a resident index tensor that is refilled (``t.copy_(next_batch)``) must not
    # be served the structures of its old contents
    return m[2] if m is not None and m[0] == key and m[1] == t._version else None


def _memo_put(t: torch.Tensor, slot: str, key, value):
    try:
        setattr(t, slot, (key, t._version, value))
    except AttributeError:
        pass
    return value


def csr_from_padded(padded: torch.Tensor, ncols: int) -> CSR:
    """agraph/bgraph/cgraph (int64 [rows, width], 0 = no entry) -> CSR over the real entries."""
    _need_gpu(padded)
    assert padded.dtype == torch.int64 and padded.dim() == 2
    hit = _memo_get(padded, "_ggpm_csr", ncols)
    if hit is not None:
        return hit
    owner = padded
    padded = padded.contiguous()
    rows, width = padded.shape
    rowptr = torch.empty(rows + 1, dtype=torch.int32, device=padded.device)
    col = torch.empty(max(rows * width, 1), dtype=torch.int32, device=padded.device)
    _lib.check(_lib.load().ggpm_padded_to_csr(_p(padded), rows, width, _p(rowptr), _p(col), _stream()),
               "padded_to_csr")
    return _memo_put(owner, "_ggpm_csr", ncols, CSR(rowptr, col, rows, ncols))


def csr_from_index(idx: torch.Tensor, ncols: int) -> CSR:
    """One entry per row (col = idx[row]); its transpose lists, per id, the rows that use it."""
    hit = _memo_get(idx, "_ggpm_csr_index", ncols)
    if hit is not None:
        return hit
    rows = idx.numel()
    rowptr = torch.arange(rows + 1, dtype=torch.int32, device=idx.device)
    # (the remembered CSR holds an alias of `idx`, not the object the memo hangs on: no reference cycle)
    return _memo_put(idx, "_ggpm_csr_index", ncols, CSR(rowptr, idx.detach(), rows, ncols))


def extract_column(mat: torch.Tensor, column: int) -> torch.Tensor:
    _need_gpu(mat)
    assert mat.dtype == torch.int64
    if mat.dim() == 1:
        mat = mat.unsqueeze(1)
    mat = mat.contiguous()
    out = torch.empty(mat.shape[0], dtype=torch.int32, device=mat.device)
    _lib.check(_lib.load().ggpm_extract_column(_p(mat), mat.shape[0], mat.shape[1], column, _p(out), _stream()),
               "extract_column")
    return out


# ----------------------------------------------------------------------------- raw launches
def gemm(ta: int, tb: int, M: int, N: int, K: int, A: torch.Tensor, lda: int, B: torch.Tensor, ldb: int,
         C: torch.Tensor, ldc: int, n_pad: int, bias: Optional[torch.Tensor] = None, accumulate: bool = False,
         act: int = ACT_NONE, zero_row0: bool = False, splitk: bool = False) -> None:
    lib = _lib.load()
    ws, wsb = None, 0
    if splitk:
        wsb = int(lib.ggpm_gemm_workspace_bytes(M, N, K))
        if wsb:
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=C.device)
    _lib.check(lib.ggpm_gemm(ta, tb, M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc, n_pad, _p(bias),
                             int(accumulate), act, int(zero_row0), _p(ws), wsb, _stream()), "gemm")


class WgradItem(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_wgrad_item"""
    _fields_ = [("dpre", ctypes.c_void_p), ("ld_dpre", ctypes.c_int), ("x", ctypes.c_void_p), ("ld_x", ctypes.c_int),
                ("dW", ctypes.c_void_p), ("ld_dw", ctypes.c_int), ("db", ctypes.c_void_p), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int)]


class GemmProblem(ctypes.Structure):
    """include/ggpm_hip.h: ggpm_gemm_problem"""
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int),
                ("C", ctypes.c_void_p), ("ldc", ctypes.c_int), ("n_pad", ctypes.c_int), ("bias", ctypes.c_void_p),
                ("accumulate", ctypes.c_int), ("act", ctypes.c_int), ("zero_row0", ctypes.c_int)]


def gemm_grouped(ta: int, tb: int, M: int, N: int, K: int, problems, splitk: bool = False) -> None:
    """problems: list of dicts(A, lda, B, ldb, C, ldc, n_pad, bias=None, accumulate=False, act=ACT_NONE, zero_row0=False);
    up to four independent products of one shape in one launch.  ``splitk``: the K range in chunks where the group has few
    output tiles and a long K (ggpm_gemm_grouped_splitk: weight gradients over all rows of a level)."""
    lib = _lib.load()
    arr = _lib.array_type(GemmProblem, len(problems))()
    for i, q in enumerate(problems):
        arr[i] = GemmProblem(_p(q["A"]), q["lda"], _p(q["B"]), q["ldb"], _p(q["C"]), q["ldc"], q["n_pad"],
                             _p(q.get("bias")), int(q.get("accumulate", False)), q.get("act", ACT_NONE),
                             int(q.get("zero_row0", False)))
    wsb = int(lib.ggpm_gemm_grouped_splitk_workspace_bytes(M, N, K, len(problems))) if splitk else 0
    if wsb:
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=problems[0]["C"].device)
        _lib.check(lib.ggpm_gemm_grouped_splitk(ta, tb, M, N, K, len(problems), ctypes.addressof(arr), _p(ws), wsb,
                                                _stream()), "gemm_grouped_splitk")
        return
    _lib.check(lib.ggpm_gemm_grouped(ta, tb, M, N, K, len(problems), ctypes.addressof(arr), _stream()),
               "gemm_grouped")


def head_accuracies(cls_pred, cls_lab, icls_pred, icls_lab, topo, topo_lab, assm) -> torch.Tensor:
    """-> [4] float tensor {motif-class, attachment-class, topology, attachment accuracy} (ggpm_head_accuracies: one launch).
    ``cls_pred`` / ``icls_pred``: int32 arg-max per row; ``topo``: 1-D scores (any stride); ``assm``: [P, C] scores or None."""
    _need_gpu(topo, cls_pred)
    out = torch.empty(4, dtype=torch.float32, device=topo.device)
    lab64 = int(cls_lab.dtype == torch.int64)
    labs = [t if t.is_contiguous() else t.contiguous() for t in (cls_lab, icls_lab, topo_lab)]
    preds = [t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous() for t in (cls_pred, icls_pred)]
    n_topo = topo.numel()
    ld_topo = topo.stride(0) if n_topo > 1 else 1
    P_, C_, ld_assm = (assm.shape[0], assm.shape[1], assm.stride(0)) if assm is not None else (0, 1, 1)
    _lib.check(_lib.load().ggpm_head_accuracies(_p(preds[0]), _p(labs[0]), _p(preds[1]), _p(labs[1]), preds[0].numel(), _p(topo),
                                                ld_topo, _p(labs[2]), n_topo, _p(assm), ld_assm, P_, C_, lab64, _p(out),
                                                _stream()), "head_accuracies")
    return out


def gemm_ksegments(tb: int, M: int, N: int, As, ldas, Bs, ldbs, Ks, C: torch.Tensor, ldc: int, n_pad: int,
                   bias: Optional[torch.Tensor] = None, accumulate: bool = False, act: int = ACT_NONE,
                   zero_row0: bool = False) -> None:
    """C = act(sum_s A_s B_s' + bias (+ C)) in one launch (up to four K segments)."""
    n = len(As)
    vp, ci = _lib.array_type(ctypes.c_void_p, n), _lib.array_type(ctypes.c_int, n)
    pa, pb = vp(*[_p(a) for a in As]), vp(*[_p(b) for b in Bs])
    la, lb, kk = ci(*ldas), ci(*ldbs), ci(*Ks)
    cast = ctypes.addressof          # (ctypes.cast would leave each array in a reference cycle with itself)
    _lib.check(_lib.load().ggpm_gemm_ksegments(tb, M, N, n, cast(pa), cast(la), cast(pb), cast(lb), cast(kk), _p(C), ldc,
                                               n_pad, _p(bias), int(accumulate), act, int(zero_row0), _stream()),
               "gemm_ksegments")


def colsum(A: torch.Tensor, M: int, N: int) -> torch.Tensor:
    out = torch.empty(N, dtype=torch.float32, device=A.device)
    ws = torch.empty(256 * N, dtype=torch.float32, device=A.device)
    _lib.check(_lib.load().ggpm_colsum(_p(A), _ld(A), M, N, _p(out), _p(ws), _stream()), "colsum")
    return out


def _segment_sum_raw(src: torch.Tensor, csr: CSR, width: int, out: torch.Tensor) -> None:
    """out[:, :width] = segmented sums; the kernel also zeroes the pad columns [width, out.shape[1])."""
    _lib.check(_lib.load().ggpm_segment_sum(_p(src), _ld(src), _p(csr.rowptr), _p(csr.col), csr.rows, width,
                                            _p(out), _ld(out), 0, out.shape[1], _stream()), "segment_sum")


# ----------------------------------------------------------------------------- autograd functions
class _Linear(torch.autograd.Function):
    """y[:, :N] = act( sum_i x_i[:, :K_i] W[:, off_i:off_i+K_i]^T + b ), pad columns of y zero."""

    @staticmethod
    def forward(ctx, weight, bias, act, zero_row0, ld_out, Ks, *xs):
        _need_gpu(weight, *xs)
        N = weight.shape[0]
        M = xs[0].shape[0]
        assert sum(Ks) == weight.shape[1] and weight.stride(1) == 1
        y = torch.empty(M, ld_out, dtype=torch.float32, device=weight.device)
        if 1 < len(xs) <= 4:        # the inputs are K segments of ONE product: no cat, no read-modify-write of y
            offs = [sum(Ks[:i]) for i in range(len(Ks))]
            gemm_ksegments(1, M, N, list(xs), [_ld(x) for x in xs], [weight[:, o:] for o in offs],
                           [weight.stride(0)] * len(xs), list(Ks), y, ld_out, ld_out, bias=bias, act=act,
                           zero_row0=zero_row0)
        else:
            off = 0
            last = len(xs) - 1
            for i, (x, K) in enumerate(zip(xs, Ks)):
                wv = weight[:, off:]
                gemm(0, 1, M, N, K, x, _ld(x), wv, weight.stride(0), y, ld_out, ld_out,
                     bias=bias if i == 0 else None, accumulate=i > 0, act=act if i == last else ACT_NONE,
                     zero_row0=zero_row0 and i == last)
                off += K
        ctx.save_for_backward(weight, y, *xs)
        ctx.meta = (act, zero_row0, Ks, bias is not None)
        ctx.weight_ref, ctx.bias_ref = weight, bias       # the Parameter objects themselves (their .grad is assigned)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, y, *xs = ctx.saved_tensors
        act, zero_row0, Ks, has_bias = ctx.meta
        N = weight.shape[0]
        M = y.shape[0]
        dy = dy.contiguous() if dy.stride(1) != 1 else dy
        lib = _lib.load()
        if act != ACT_NONE or zero_row0:
            dpre = torch.empty_like(y)
            assert _ld(dy) == _ld(y)
            _lib.check(lib.ggpm_act_backward(_p(dy), _p(y), M, N, _ld(y), act, int(zero_row0), _p(dpre), _stream()),
                       "act_backward")
        else:
            dpre = dy
        dxs: List[Optional[torch.Tensor]] = []
        off = 0
        for i, (x, K) in enumerate(zip(xs, Ks)):     # input gradients are needed upstream right away: main stream
            if ctx.needs_input_grad[6 + i]:
                dx = torch.empty_like(x)
                gemm(0, 0, M, K, N, dpre, _ld(dpre), weight[:, off:], weight.stride(0), dx, _ld(dx), x.shape[1])
                dxs.append(dx)
            else:
                dxs.append(None)
            off += K

        def param_grads():
            dW_ = torch.empty_like(weight) if ctx.needs_input_grad[0] else None
            o = 0
            for x, K in zip(xs, Ks):
                if dW_ is not None:
                    gemm(1, 0, N, K, M, dpre, _ld(dpre), x, _ld(x), dW_[:, o:], dW_.stride(0), K, splitk=True)
                o += K
            db_ = colsum(dpre, M, N) if (has_bias and ctx.needs_input_grad[1]) else None
            return dW_, db_

        wref, bias = ctx.weight_ref, ctx.bias_ref
        leaf = (ctx.needs_input_grad[0] and (bias is None or ctx.needs_input_grad[1]) and can_publish(wref, bias))
        if leaf and defer_wgrads_enabled():       # one contraction per parameter at the end of the pass (see _DEFER)
            _defer_linear(wref, bias, dpre, list(xs), Ks)
            return (None, None, None, None, None, None, *dxs)
        use_side = side_stream_enabled() and leaf
        if use_side:      # weight / bias gradients: second stream, straight into param.grad (as the level functions do)
            main = torch.cuda.current_stream()
            side = _side_stream(weight.device)
            side.wait_stream(main)
            for tns in (dpre, *xs):
                tns.record_stream(side)
            with torch.cuda.stream(side):
                dW, db = param_grads()
                _accumulate_grad(wref, dW, main)
                if db is not None:
                    _accumulate_grad(bias, db, main)
            _join_later(main, side)
            return (None, None, None, None, None, None, *dxs)
        dW, db = param_grads()
        return (dW, db, None, None, None, None, *dxs)


def linear(xs: Sequence[torch.Tensor], Ks: Sequence[int], weight: torch.Tensor, bias: Optional[torch.Tensor],
           act: int = ACT_NONE, zero_row0: bool = False, ld_out: Optional[int] = None) -> torch.Tensor:
    N = weight.shape[0]
    if ld_out is None:
        ld_out = padded_hidden(N)
    return _Linear.apply(weight, bias, act, zero_row0, ld_out, tuple(Ks), *xs)


class _SegmentSum(torch.autograd.Function):
    """out[r] = sum_{j in csr row r} src[col[j]]  (index_select_ND(...).sum(1) over real entries)."""

    @staticmethod
    def forward(ctx, src, csr, width):
        _need_gpu(src)
        out = torch.empty(csr.rows, _ld(src), dtype=torch.float32, device=src.device)
        _segment_sum_raw(src, csr, width, out)
        ctx.csr, ctx.width, ctx.src_rows, ctx.src_cols = csr, width, src.shape[0], src.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous() if dout.stride(1) != 1 else dout
        csrT = ctx.csr.T
        assert csrT.rows == ctx.src_rows
        dsrc = torch.empty(ctx.src_rows, _ld(dout), dtype=torch.float32, device=dout.device)
        _segment_sum_raw(dout, csrT, ctx.width, dsrc)
        return dsrc[:, :ctx.src_cols], None, None     # src may have been a [rows, H] view of a padded buffer


def segment_sum(src: torch.Tensor, csr: CSR, width: int) -> torch.Tensor:
    return _SegmentSum.apply(src, csr, width)


class _GatherRows(torch.autograd.Function):
    """out[r, :width] = table[idx[r], :width] (nn.Embedding / index_select); backward through idx^T."""

    @staticmethod
    def forward(ctx, table, idx, idx_csr, width, ld_out):
        _need_gpu(table, idx)
        rows = idx.numel()
        out = torch.empty(rows, ld_out, dtype=torch.float32, device=table.device)
        _lib.check(_lib.load().ggpm_gather_rows(_p(table), _ld(table), _p(idx), rows, width, _p(out), ld_out, 0,
                                                ld_out, _stream()), "gather_rows")
        ctx.idx_csr, ctx.width, ctx.tshape, ctx.tld = idx_csr, width, table.shape, _ld(table)
        ctx.table_ref, ctx.idx = table, idx
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous() if dout.stride(1) != 1 else dout
        if ctx.needs_input_grad[0] and can_publish(ctx.table_ref) and defer_wgrads_enabled():
            _defer_gather(ctx.table_ref, ctx.width, dout, ctx.idx)      # one scatter over the rows of all visits at the end of the pass
            return None, None, None, None, None
        csrT = ctx.idx_csr.T
        dtable = torch.empty(ctx.tshape, dtype=torch.float32, device=dout.device)
        _segment_sum_raw(dout, csrT, ctx.width, dtable)
        return dtable, None, None, None, None


def gather_rows(table: torch.Tensor, idx: torch.Tensor, idx_csr: CSR, width: int, ld_out: int) -> torch.Tensor:
    return _GatherRows.apply(table, idx, idx_csr, width, ld_out)


class _TreeMessInput(torch.autograd.Function):
    """hmess = [hnode[src] | onehot(pos)]  (embed_inter/embed_tree, ggpm/encoder.py:103-106,114-116)."""

    @staticmethod
    def forward(ctx, hnode, src, src_csr, pos, H, n_pos, ld_out):
        _need_gpu(hnode, src, pos)
        lib = _lib.load()
        rows = src.numel()
        out = torch.empty(rows, ld_out, dtype=torch.float32, device=hnode.device)
        _lib.check(lib.ggpm_gather_rows(_p(hnode), _ld(hnode), _p(src), rows, H, _p(out), ld_out, 0, 0, _stream()),
                   "gather_rows")
        _lib.check(lib.ggpm_onehot(_p(pos), rows, n_pos, _p(out), ld_out, H, ld_out, _stream()), "onehot")
        ctx.src_csr, ctx.H, ctx.nshape = src_csr, H, hnode.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous() if dout.stride(1) != 1 else dout
        csrT = ctx.src_csr.T
        dh = torch.empty(ctx.nshape, dtype=torch.float32, device=dout.device)
        _segment_sum_raw(dout, csrT, ctx.H, dh)
        return dh, None, None, None, None, None, None


def tree_message_input(hnode, src, src_csr, pos, H, n_pos, ld_out):
    return _TreeMessInput.apply(hnode, src, src_csr, pos, H, n_pos, ld_out)


def embed_graph(fnode: torch.Tensor, fmess: torch.Tensor, atom_size: int, bond_types: int, max_pos: int
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One-hot atom-level inputs (constants: no gradient), ggpm/encoder.py:119-126."""
    _need_gpu(fnode, fmess)
    N1, E1 = fnode.shape[0], fmess.shape[0]
    ld_n = (atom_size + 3) // 4 * 4
    ld_m = (atom_size + bond_types + max_pos + 3) // 4 * 4
    hnode = torch.empty(N1, ld_n, dtype=torch.float32, device=fnode.device)
    hmess = torch.empty(E1, ld_m, dtype=torch.float32, device=fnode.device)
    _lib.check(_lib.load().ggpm_embed_graph(_p(fnode.contiguous()), N1, _p(fmess.contiguous()), E1, atom_size,
                                            bond_types, max_pos, _p(hnode), ld_n, _p(hmess), ld_m, _stream()),
               "embed_graph")
    return hnode, hmess


# ----------------------------------------------------------------------------- second stream for weight gradients
# The depth loops are latency bound (small dependent kernels); the weight-gradient GEMMs over the stashes are
# throughput bound and independent of the remaining backward.  With GGPM_SIDE_STREAM=1 (default) they run on
# a second HIP stream beside the next level's depth loop and accumulate straight into ``param.grad``; the main
# stream re-joins at the end of the backward pass (autograd engine callback), so after ``loss.backward()``
# returns every later main-stream consumer (clip_grad_norm_, all-reduce, optimizer) is ordered behind them.
# Contract: a parameter whose gradient is published this way must be consumed ONLY by the ops of this module inside the
# backward pass (true for every parameter of the encoder / decoder classes of this package, tied embeddings included:
# both users go through these ops).  A stock torch op on the same leaf would have autograd's AccumulateGrad add to
# ``.grad`` on the main stream while the second stream may still be writing it -- route such a use through
# GGPM_SIDE_STREAM=0, which keeps every gradient on the main stream and inside autograd.
_SIDE = {}


def side_stream_enabled() -> bool:
    import os
    return os.environ.get("GGPM_SIDE_STREAM", "1") != "0"


def _side_stream(device) -> torch.cuda.Stream:
    key = (device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)      # (ROCm offers priorities 0 and -1 only: nothing below normal)
    return _SIDE[key]


def writer_streams(device) -> list:
    """Every helper stream of the package on `device` that writes parameter gradients: the second stream and the decoder's
    atom-level stream (parallel.FlatGradSync._join_writers)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    out = [_SIDE[key]] if key in _SIDE else []
    try:
        from .decoder import HierMPNDecoder
        s = HierMPNDecoder._ATOM_STREAMS.get(key)
        if s is not None:
            out.append(s)
    except Exception:
        pass
    return out


# A gradient tensor computed on a helper stream (second stream, atom-level stream) and then read on the main stream
# (optimizer, clipping, the flat-buffer pack) would ordinarily be marked with ``record_stream(main)``.  On ROCm every such
# mark costs an event record on that stream when the tensor is released -- ~4.7 us of queue time each, and the ~55
# gradients of a VAE step are released together right in front of the optimizer launch (0.25 ms of idle GPU, measured:
# tools/probe/free_stall_probe.py).  The mark is not needed here: a published gradient stays referenced by ``param.grad``
# until the optimizer side releases it, i.e. after everything that reads it has been ENQUEUED on the main stream, and every
# helper stream of this package waits for the main stream (``wait_stream(main)``) before the first allocation of its next use
# -- the block cannot be handed out again in front of its readers.  _dev.RECORD_GRADS restores the marks.


def hand_to(g: torch.Tensor, main: torch.cuda.Stream) -> None:
    if _dev.RECORD_GRADS:
        g.record_stream(main)


def _accumulate_grad(param: torch.Tensor, g: torch.Tensor, main: torch.cuda.Stream) -> None:
    """param.grad (+)= g on the CURRENT (side) stream."""
    if param.grad is None:
        hand_to(g, main)
        param.grad = g
    else:
        param.grad.add_(g)


def _join_later(main: torch.cuda.Stream, side: torch.cuda.Stream) -> None:
    torch.autograd.Variable._execution_engine.queue_callback(lambda: main.wait_stream(side))


# ----------------------------------------------------------------------------- phase marks (dev instrumentation)
# tools/vae_phase_times.py sets MARKS = [] and reads (name, host time, event on the current stream) triples back; None
# (default) makes mark() a no-op.
MARKS = None


def mark(name: str) -> None:
    if MARKS is not None:
        import time
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        MARKS.append((name, time.perf_counter(), ev))


# ----------------------------------------------------------------------------- deferred parameter gradients
# The teacher-forced decoder uses the same parameters on every one of its ~20 steps, so a backward pass meets each
# Linear ~20 times.  Forming dW = dpre^T x (+ column sum, + the accumulate kernel autograd adds) per visit is ~40 tiny
# launches per parameter and pass.  With GGPM_DEFER_WGRADS=1 (default) a visit only queues its (dpre, x) rows; when the
# backward pass ends (autograd engine callback) every parameter gets ONE contraction over the stacked rows of all its
# visits -- dW = [dpre_1; dpre_2; ...]^T [x_1; x_2; ...], the same sum in a different order -- and other per-visit
# parameter gradients (the message functions', the embedding tables') are summed by one stacked reduction each.
_DEFER = {"linear": {}, "sum": {}, "gather": {}, "task": None, "stream": None, "early": None, "pending": []}


def defer_wgrads_enabled() -> bool:
    return os.environ.get("GGPM_DEFER_WGRADS", "1") != "0"


def _has_hooks(p) -> bool:
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


_PUBLISH = [True]


def publish_gradients(enabled: bool) -> bool:
    """Process-wide switch for gradients written to ``.grad`` by this package itself (deferred contractions, second
    stream, flat gradient buffer) instead of being returned through autograd.  -> the previous setting.

    Switch it OFF before wrapping a model in ``torch.nn.parallel.DistributedDataParallel`` or anything else that hooks the
    AccumulateGrad NODE of a parameter: such hooks live in the C++ node and cannot be seen from Python (``can_publish``
    only sees ``Tensor.register_hook`` and ``register_post_accumulate_grad_hook``), and a reducer whose hook never fires
    waits for a gradient that was written around it.  The data-parallel path this package supports and measures is
    ``ggpm_amd.parallel.FlatGradSync`` (one all-reduce of one flat buffer), which needs no hooks."""
    prev, _PUBLISH[0] = _PUBLISH[0], bool(enabled)
    return prev


def can_publish(*params) -> bool:
    """True when the gradients of these parameters may be written to ``.grad`` by this module itself (deferred
    contraction / second stream) instead of being returned through autograd: publishing is on (``publish_gradients``)
    and every parameter is a leaf that requires grad and carries NO Python-visible hook.  A parameter with a tensor hook
    or a post-accumulate-grad hook (hook-based clippers / reducers) gets its gradient the ordinary way, through
    AccumulateGrad, so that the hooks fire.  Hooks on the AccumulateGrad node itself (stock DDP's reducer) are not visible
    here: see ``publish_gradients``."""
    return _PUBLISH[0] and all(p is None or (getattr(p, "is_leaf", False) and p.requires_grad and not _has_hooks(p))
                               for p in params)


def _defer_register() -> None:
    """Queue the end-of-backward flush once per backward pass.  A pass is identified by the autograd engine's graph
    task id: a pass that RAISED never ran its callbacks, so whatever it left queued is dropped when the next pass
    registers (nothing sticky survives a failed backward)."""
    task = torch._C._current_graph_task_id()
    if _DEFER["task"] != task or task < 0:
        _DEFER["linear"].clear()
        _DEFER["sum"].clear()
        _DEFER["gather"].clear()
        _DEFER["pending"], _DEFER["early"] = [], None
        _DEFER["task"] = task
        _DEFER["stream"] = torch.cuda.current_stream()
        torch.autograd.Variable._execution_engine.queue_callback(_defer_flush)


def _defer_linear(weight, bias, dpre, xs, Ks) -> None:
    _defer_register()
    # keyed by the parameter AND the column split it was visited with: a Linear used with two different K splits in one
    # pass gets one contraction per split (both land in the same .grad)
    key = (id(weight), tuple(Ks), id(bias) if bias is not None else 0)
    _DEFER["linear"].setdefault(key, (weight, bias, tuple(Ks), []))[3].append((dpre, xs))


def _defer_gather(table, width, dout, idx) -> None:
    _defer_register()
    _DEFER["gather"].setdefault((id(table), int(width)), (table, width, []))[2].append((dout, idx))


def _defer_sum(param, grad) -> None:
    """param.grad += grad, summed with the pass's other contributions to the same parameter by ONE reduction at the end."""
    if grad is None:
        return
    _defer_register()
    _DEFER["sum"].setdefault(id(param), (param, []))[1].append(grad)


def _add_to_grad(param, g) -> None:
    if param.grad is None:
        param.grad = g
    else:
        param.grad.add_(g)


def _defer_flush(side: Optional[torch.cuda.Stream] = None) -> None:
    """Form the queued parameter gradients.  ``side`` = None: the end-of-backward callback, on the stream the entries
    were queued on.  ``side`` given (flush_deferred_early): on that stream, ordered behind everything the queueing
    stream has been given so far; the queueing stream re-joins at the end of the backward pass."""
    lin, sums, gath = dict(_DEFER["linear"]), dict(_DEFER["sum"]), dict(_DEFER["gather"])
    _DEFER["linear"].clear()
    _DEFER["sum"].clear()
    _DEFER["gather"].clear()
    main = _DEFER["stream"]
    if side is None:
        _DEFER["task"] = None
        mark("bwd: end-of-pass flush starts")
        if main is not None and _DEFER["early"] is not None:
            # what the early flush computed on the second stream is handed to .grad HERE, on the queueing stream, once
            # that stream is ordered behind it: every mutation of .grad stays on one stream, whatever order the engine
            # ran the nodes in (autograd's own AccumulateGrad for tied / shared parameters included)
            main.wait_stream(_DEFER["early"])
            _DEFER["early"] = None
            with torch.cuda.stream(main):
                for param, g in _DEFER["pending"]:
                    _add_to_grad(param, g)
        _DEFER["pending"] = []
    if not (lin or sums or gath) or main is None:
        return
    stream = main
    if side is not None:
        side.wait_stream(main)
        stream = side
        _DEFER["early"] = side

    def use(t):                            # queued on one stream, read on another: keep the allocator from recycling it early
        if side is not None and isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(stream)
        return t

    def publish(param, g):
        if side is not None:               # computed early: published by the end-of-backward flush (see above)
            hand_to(g, main)
            _DEFER["pending"].append((param, g))
        else:
            _add_to_grad(param, g)

    def target(param):
        """Where a parameter's gradient is formed: its slice of the flat gradient buffer (parallel.FlatGradSync) when it has one
        and nothing has been written there in this pass -- the optimizer / all-reduce side then finds it in place (no pack copy:
        one multi-tensor launch right in front of the optimizer, and ~35 `.grad` re-pointings, less per step) -- else a fresh
        tensor.  The first contribution of a pass writes, later ones are added (`_add_to_grad`)."""
        v = getattr(param, "_ggpm_grad_view", None)
        if v is not None and param.grad is None and id(param) not in taken:
            taken.add(id(param))
            return v
        return torch.empty_like(param)

    taken = set(id(q) for q, _ in _DEFER["pending"])
    with torch.cuda.stream(stream):
        # every Linear's weight gradient (one contraction per K segment) and bias gradient in ONE library call
        # (ggpm_linear_wgrads_batch: the same launches in the same order as ~60 separate gemm / colsum calls)
        items, keep, n_max, ws_max = [], [], 0, 0
        lib = _lib.load()
        for weight, bias, Ks, visits in lin.values():
            N = weight.shape[0]
            if len(visits) == 1:
                dpre, xs = use(visits[0][0]), [use(x) for x in visits[0][1]]
            else:
                dpre = torch.cat([use(v[0]) for v in visits], dim=0)
                xs = [torch.cat([use(v[1][i])[:, :K] for v in visits], dim=0) for i, K in enumerate(Ks)]
            M = dpre.shape[0]
            dW = target(weight)
            db = target(bias) if bias is not None else None
            o = 0
            for i, (x, K) in enumerate(zip(xs, Ks)):
                items.append((dpre.data_ptr(), _ld(dpre), x.data_ptr(), _ld(x), dW.data_ptr() + 4 * o, dW.stride(0),
                              db.data_ptr() if (db is not None and i == 0) else 0, M, N, K))
                ws_max = max(ws_max, int(lib.ggpm_gemm_workspace_bytes(N, K, M)))
                o += K
            n_max = max(n_max, N)
            keep.append((dpre, xs, dW, db))
            publish(weight, dW)
            if db is not None:
                publish(bias, db)
        if items:
            arr = _lib.array_type(WgradItem, len(items))(*[WgradItem(*it) for it in items])
            dev = keep[0][0].device
            ws = torch.empty(ws_max // 4, dtype=torch.float32, device=dev) if ws_max else None
            csws = torch.empty(256 * n_max, dtype=torch.float32, device=dev)
            _lib.check(lib.ggpm_linear_wgrads_batch(len(items), ctypes.addressof(arr), _p(ws), ws_max, _p(csws),
                                                    _stream()), "linear_wgrads_batch")
        for param, grads in sums.values():
            publish(param, use(grads[0]) if len(grads) == 1 else torch.stack([use(g) for g in grads], dim=0).sum(dim=0))
        for table, width, visits in gath.values():       # embedding tables: d(table)[id] = sum of the rows that used id
            dout = use(visits[0][0]) if len(visits) == 1 else torch.cat([use(v[0]) for v in visits], dim=0)
            idx = use(visits[0][1]) if len(visits) == 1 else torch.cat([use(v[1]).reshape(-1) for v in visits], dim=0)
            csrT = csr_from_index(idx.reshape(-1), ncols=table.shape[0]).T
            dtable = target(table)
            _segment_sum_raw(dout, csrT, width, dtable)
            publish(table, dtable)


def flush_deferred_early() -> None:
    """Called from inside a backward pass at a point after which only nodes WITHOUT deferred gradients have much left to
    do (the decoder's atom level and the encoder, once the heads and the tree-side levels have run): the queued
    contractions start now on the second stream, beside that work, instead of behind it.  Whatever is queued later still
    goes through the end-of-backward flush.  _dev.DEFER_EARLY = False switches it off."""
    if not _dev.DEFER_EARLY or not side_stream_enabled():
        return
    main = _DEFER["stream"]
    if main is None or _DEFER["task"] is None or _DEFER["task"] != torch._C._current_graph_task_id():
        return
    _defer_flush(side=_side_stream(main.device))       # (the end-of-backward callback stays registered: it publishes)


def _split_cols(W: torch.Tensor, I: int):
    """(x-half view, h-half view) of a [H, I+H] gate weight; both share W's row stride."""
    return W[:, :I], W[:, I:]


class _GruLevel(torch.autograd.Function):
    """GRU.forward (ggpm/rnn.py:41-50) for one level: hoisted input GEMMs + fused depth loop.

    One autograd node per level: the backward writes the x-half and the h-half gradient of every gate weight
    straight into ONE full-shape gradient tensor (no slice/cat/add kernels from autograd).
    Returns h_D as [E1, Hp] (pad columns zero).
    """

    @staticmethod
    def forward(ctx, x, W_z, b_z, W_r, U_r, b_u, W_h, b_h, pred, depth, I, H, gate_dtype=0):
        _need_gpu(x, W_z, b_z, W_r, U_r, b_u, W_h, b_h)
        lib = _lib.load()
        ctx.gate_dtype = gate_dtype
        E1, Hp = x.shape[0], padded_hidden(H)
        f32 = dict(dtype=torch.float32, device=x.device)
        save = any(ctx.needs_input_grad)
        X = torch.empty(3, E1, Hp, **f32)
        Wz_x, Wz_h = _split_cols(W_z, I)
        Wh_x, Wh_h = _split_cols(W_h, I)
        ldx = _ld(x)
        gemm(0, 1, E1, H, I, x, ldx, Wz_x, W_z.stride(0), X[0], Hp, Hp, bias=b_z)
        gemm(0, 1, E1, H, I, x, ldx, W_r, W_r.stride(0), X[1], Hp, Hp)
        gemm(0, 1, E1, H, I, x, ldx, Wh_x, W_h.stride(0), X[2], Hp, Hp, bias=b_h)
        wpack = torch.empty(int(lib.ggpm_gru_pack_floats(H)), **f32)
        if save:
            Hs = torch.empty(depth + 1, E1, Hp, **f32)
            Qs = torch.empty(depth, E1, Hp, **f32)
            St = torch.empty(5, depth, E1, Hp, **f32)
            Ss, Gs, Zs, Ms, Rs = St[0], St[1], St[2], St[3], St[4]
        else:
            Hs = torch.empty(2, E1, Hp, **f32)
            Qs = torch.empty(2, E1, Hp, **f32)
            Ss = Gs = Zs = Ms = Rs = None
        with _gate_dtype(gate_dtype):
            _lib.check(lib.ggpm_gru_forward(E1, H, depth, _p(X[0]), _p(X[1]), _p(X[2]), _p(Wz_h), W_z.stride(0),
                                            _p(U_r), U_r.stride(0), _p(b_u), _p(Wh_h), W_h.stride(0), _p(pred.rowptr),
                                            _p(pred.col), _p(Hs), _p(Qs), _p(Ss), _p(Gs), _p(Zs), _p(Ms), _p(Rs),
                                            _p(wpack), int(save), _stream()), "gru_forward")
        if save:
            ctx.save_for_backward(x, W_z, W_r, U_r, W_h)
            ctx.stash = (X[1], Hs, Qs, Ss, Gs, Zs, Ms, Rs)
            ctx.meta = (pred, depth, I, H)
            ctx.params = (W_z, b_z, W_r, U_r, b_u, W_h, b_h)
            return Hs[depth]
        return Hs[depth & 1]

    @staticmethod
    def backward(ctx, dHD):
        x, W_z, W_r, U_r, W_h = ctx.saved_tensors
        Xr, Hs, Qs, Ss, Gs, Zs, Ms, Rs = ctx.stash
        pred, depth, I, H = ctx.meta
        lib = _lib.load()
        E1, Hp = x.shape[0], padded_hidden(H)
        succ = pred.T
        dHD = dHD.contiguous()
        f32 = dict(dtype=torch.float32, device=x.device)
        dX = torch.empty(3, E1, Hp, **f32)
        dW_z, dW_r, dU_r, dW_h = (torch.empty(W_z.shape, **f32), torch.empty(W_r.shape, **f32),
                                  torch.empty(H, H, **f32), torch.empty(W_h.shape, **f32))
        db_u = torch.empty(H, **f32)
        Wz_x, Wz_h = _split_cols(W_z, I)
        Wh_x, Wh_h = _split_cols(W_h, I)
        dWz_x, dWz_h = _split_cols(dW_z, I)
        dWh_x, dWh_h = _split_cols(dW_h, I)
        wb = int(lib.ggpm_gru_backward_workspace_bytes(E1, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        use_side = side_stream_enabled() and can_publish(*ctx.params)
        with _gate_dtype(ctx.gate_dtype):
            _lib.check(lib.ggpm_gru_backward(E1, H, depth, _p(Xr), _p(Wz_h), W_z.stride(0), _p(U_r), U_r.stride(0),
                                             _p(Wh_h), W_h.stride(0), _p(pred.rowptr), _p(pred.col), _p(succ.rowptr),
                                             _p(succ.col), _p(Hs), _p(Qs), _p(Ss), _p(Gs), _p(Zs), _p(Ms), _p(Rs),
                                             _p(dHD), _p(dX[0]), _p(dX[1]), _p(dX[2]), _p(dWz_h), dW_z.stride(0),
                                             _p(dU_r), H, _p(db_u), _p(dWh_h), dW_h.stride(0), _p(work),
                                             work.numel() * 4, 0 if use_side else 1, _stream()), "gru_backward")
        ctx.stash = None
        ldx = _ld(x)
        dx = None
        if ctx.needs_input_grad[0]:      # needed upstream right away: stays on the main stream
            dx = _empty_same_layout(x)
            gemm(0, 0, E1, I, H, dX[0], Hp, Wz_x, W_z.stride(0), dx, ldx, x.shape[1])
            gemm(0, 0, E1, I, H, dX[1], Hp, W_r, W_r.stride(0), dx, ldx, I, accumulate=True)
            gemm(0, 0, E1, I, H, dX[2], Hp, Wh_x, W_h.stride(0), dx, ldx, I, accumulate=True)

        def weight_grads():
            if use_side:
                with _gate_dtype(ctx.gate_dtype):
                    _lib.check(lib.ggpm_gru_weight_grads(E1, H, depth, _p(Hs), _p(Ss), _p(Gs), _p(work), work.numel() * 4,
                                                         _p(dWz_h), dW_z.stride(0), _p(dU_r), H, _p(db_u), _p(dWh_h),
                                                         dW_h.stride(0), _stream()), "gru_weight_grads")
            # x-halves of the gate weights and the gate biases
            gemm(1, 0, H, I, E1, dX[0], Hp, x, ldx, dWz_x, dW_z.stride(0), I, splitk=True)
            gemm(1, 0, H, I, E1, dX[1], Hp, x, ldx, dW_r, dW_r.stride(0), I, splitk=True)
            gemm(1, 0, H, I, E1, dX[2], Hp, x, ldx, dWh_x, dW_h.stride(0), I, splitk=True)
            return colsum(dX[0], E1, H), colsum(dX[2], E1, H)

        if use_side:
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)
            for tns in (work, dX, Hs, Ss, x, dW_z, dW_r, dU_r, dW_h, db_u):
                tns.record_stream(side)
            with torch.cuda.stream(side):
                db_z, db_h = weight_grads()
                P_z, Pb_z, P_r, P_u, Pb_u, P_h, Pb_h = ctx.params
                for prm, g in ((P_z, dW_z), (Pb_z, db_z), (P_r, dW_r), (P_u, dU_r), (Pb_u, db_u), (P_h, dW_h),
                               (Pb_h, db_h)):
                    _accumulate_grad(prm, g, main)
            _join_later(main, side)
            return dx, None, None, None, None, None, None, None, None, None, None, None, None
        db_z, db_h = weight_grads()
        return dx, dW_z, db_z, dW_r, dU_r, db_u, dW_h, db_h, None, None, None, None, None


GATE_DTYPES = {"f32": 0, "fp32": 0, "bf16": 1, "f32_mfma": 2, "f32_split": 3, None: 0, 0: 0, 1: 1, 2: 2, 3: 3}


class _gate_dtype:
    """``with _gate_dtype(1):`` -- the level calls issued inside run their hidden x hidden products on bf16 operands
    (ggpm_level_gate_dtype is per thread, so this holds for exactly the calls made here, on whichever thread autograd
    runs the function)."""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        self.prev = _lib.load().ggpm_level_gate_dtype(self.dt) if self.dt else 0

    def __exit__(self, *exc):
        if self.dt:
            _lib.load().ggpm_level_gate_dtype(self.prev)
        return False


def gru_level(x, W_z, b_z, W_r, U_r, b_u, W_h, b_h, pred: CSR, depth: int, I: int, H: int, gate_dtype=None) -> torch.Tensor:
    return _GruLevel.apply(x, W_z, b_z, W_r, U_r, b_u, W_h, b_h, pred, depth, I, H, GATE_DTYPES[gate_dtype])


def _as_padded_state(h: torch.Tensor, H: int, Hp: int) -> torch.Tensor:
    """[E, H] public state -> [E, Hp] buffer with zero pad columns (no copy when it already is a view of one)."""
    if h.dim() == 2 and h.shape[1] == H and h.stride(1) == 1 and h.stride(0) == Hp and Hp != H \
            and h.storage_offset() % Hp == 0:
        return h.as_strided((h.shape[0], Hp), (Hp, 1), h.storage_offset())     # our own earlier output
    if Hp == H:
        return h.contiguous()
    out = torch.zeros(h.shape[0], Hp, dtype=h.dtype, device=h.device)
    out[:, :H] = h
    return out


def _scatter_rows(full_rows: int, sub: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """zeros([full_rows, ...]) with rows `index` := sub (the subset tensors of get_sub_tensor back in place)."""
    out = torch.zeros((full_rows,) + tuple(sub.shape[1:]), dtype=sub.dtype, device=sub.device)
    out.index_copy_(0, index, sub)
    return out


def _sparse_structure(E1: int, submess: torch.Tensor, bgraph_sub: torch.Tensor):
    """(frozen mask [E1] uint8, predecessor CSR over all E1 rows) of a sparse_forward call, remembered on ``bgraph_sub``."""
    key = (E1, submess.data_ptr(), submess.numel(), submess._version)
    hit = _memo_get(bgraph_sub, "_ggpm_sparse", key)
    if hit is not None:
        return hit
    frozen = torch.ones(E1, dtype=torch.uint8, device=submess.device)
    frozen.index_fill_(0, submess, 0)
    pred = csr_from_padded(_scatter_rows(E1, bgraph_sub, submess), ncols=E1)
    return _memo_put(bgraph_sub, "_ggpm_sparse", key, (frozen, pred, submess))      # (keeps `submess` alive: the key names it)


class _GruSparse(torch.autograd.Function):
    """GRU.sparse_forward (ggpm/rnn.py:52-59): recompute the rows `submess` of the message state `depth` times."""

    @staticmethod
    def forward(ctx, h_in, x_sub, submess, bgraph_sub, W_z, b_z, W_r, U_r, b_u, W_h, b_h, depth, I, H):
        _need_gpu(h_in, x_sub, submess, bgraph_sub, W_z)
        lib = _lib.load()
        E1, Hp, ms = h_in.shape[0], padded_hidden(H), submess.numel()
        f32 = dict(dtype=torch.float32, device=h_in.device)
        save = any(ctx.needs_input_grad)
        hp = _as_padded_state(h_in, H, Hp)
        frozen, pred, _ = _sparse_structure(E1, submess, bgraph_sub)
        Wz_x, Wz_h = _split_cols(W_z, I)
        Wh_x, Wh_h = _split_cols(W_h, I)
        ldx = _ld(x_sub)
        Xs = torch.empty(3, ms, Hp, **f32)
        gemm(0, 1, ms, H, I, x_sub, ldx, Wz_x, W_z.stride(0), Xs[0], Hp, Hp, bias=b_z)
        gemm(0, 1, ms, H, I, x_sub, ldx, W_r, W_r.stride(0), Xs[1], Hp, Hp)
        gemm(0, 1, ms, H, I, x_sub, ldx, Wh_x, W_h.stride(0), Xs[2], Hp, Hp, bias=b_h)
        X = torch.zeros(3, E1, Hp, **f32)
        X.index_copy_(1, submess, Xs)
        wpack = torch.empty(int(lib.ggpm_gru_pack_floats(H)), **f32)
        if save:
            Hs = torch.empty(depth + 1, E1, Hp, **f32)
            Qs = torch.empty(depth, E1, Hp, **f32)
            St = torch.empty(5, depth, E1, Hp, **f32)
            Ss, Gs, Zs, Ms, Rs = St[0], St[1], St[2], St[3], St[4]
        else:
            Hs = torch.empty(2, E1, Hp, **f32)
            Qs = torch.empty(2, E1, Hp, **f32)
            Ss = Gs = Zs = Ms = Rs = None
        _lib.check(lib.ggpm_gru_sparse_forward(E1, H, depth, _p(hp), _p(frozen), _p(X[0]), _p(X[1]), _p(X[2]), _p(Wz_h),
                                               W_z.stride(0), _p(U_r), U_r.stride(0), _p(b_u), _p(Wh_h), W_h.stride(0),
                                               _p(pred.rowptr), _p(pred.col), _p(Hs), _p(Qs), _p(Ss), _p(Gs), _p(Zs),
                                               _p(Ms), _p(Rs), _p(wpack), int(save), _stream()), "gru_sparse_forward")
        out = Hs[depth] if save else Hs[depth & 1]
        if save:
            ctx.save_for_backward(x_sub, submess, W_z, W_r, U_r, W_h)
            ctx.stash = (X[1], frozen, pred, Hs, Qs, Ss, Gs, Zs, Ms, Rs)
            ctx.meta = (depth, I, H)
            ctx.param_refs = (W_z, b_z, W_r, U_r, b_u, W_h, b_h)
        return out[:, :H]

    @staticmethod
    def backward(ctx, dH):
        x_sub, submess, W_z, W_r, U_r, W_h = ctx.saved_tensors
        Xr, frozen, pred, Hs, Qs, Ss, Gs, Zs, Ms, Rs = ctx.stash
        depth, I, H = ctx.meta
        lib = _lib.load()
        E1, Hp, ms = Hs.shape[1], padded_hidden(H), submess.numel()
        f32 = dict(dtype=torch.float32, device=x_sub.device)
        succ = pred.T
        dHD = torch.zeros(E1, Hp, **f32)
        dHD[:, :H] = dH
        dHin = torch.empty(E1, Hp, **f32)
        dX = torch.empty(3, E1, Hp, **f32)
        dW_z, dW_r, dU_r, dW_h = (torch.empty(W_z.shape, **f32), torch.empty(W_r.shape, **f32),
                                  torch.empty(H, H, **f32), torch.empty(W_h.shape, **f32))
        db_u = torch.empty(H, **f32)
        Wz_x, Wz_h = _split_cols(W_z, I)
        Wh_x, Wh_h = _split_cols(W_h, I)
        dWz_x, dWz_h = _split_cols(dW_z, I)
        dWh_x, dWh_h = _split_cols(dW_h, I)
        wb = int(lib.ggpm_gru_backward_workspace_bytes(E1, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        _lib.check(lib.ggpm_gru_sparse_backward(E1, H, depth, _p(frozen), _p(Xr), _p(Wz_h), W_z.stride(0), _p(U_r),
                                                U_r.stride(0), _p(Wh_h), W_h.stride(0), _p(pred.rowptr), _p(pred.col),
                                                _p(succ.rowptr), _p(succ.col), _p(Hs), _p(Qs), _p(Ss), _p(Gs), _p(Zs),
                                                _p(Ms), _p(Rs), _p(dHD), _p(dHin), _p(dX[0]), _p(dX[1]), _p(dX[2]),
                                                _p(dWz_h), dW_z.stride(0), _p(dU_r), H, _p(db_u), _p(dWh_h),
                                                dW_h.stride(0), _p(work), work.numel() * 4, _stream()),
                   "gru_sparse_backward")
        ctx.stash = None
        dXs = dX.index_select(1, submess)             # [3, ms, Hp]: only the recomputed rows carry input gradients
        ldx = _ld(x_sub)
        gemm(1, 0, H, I, ms, dXs[0], Hp, x_sub, ldx, dWz_x, dW_z.stride(0), I, splitk=True)
        gemm(1, 0, H, I, ms, dXs[1], Hp, x_sub, ldx, dW_r, dW_r.stride(0), I, splitk=True)
        gemm(1, 0, H, I, ms, dXs[2], Hp, x_sub, ldx, dWh_x, dW_h.stride(0), I, splitk=True)
        db_z, db_h = colsum(dXs[0], ms, H), colsum(dXs[2], ms, H)
        dx = None
        if ctx.needs_input_grad[1]:
            dx = _empty_same_layout(x_sub)
            gemm(0, 0, ms, I, H, dXs[0], Hp, Wz_x, W_z.stride(0), dx, ldx, x_sub.shape[1])
            gemm(0, 0, ms, I, H, dXs[1], Hp, W_r, W_r.stride(0), dx, ldx, I, accumulate=True)
            gemm(0, 0, ms, I, H, dXs[2], Hp, Wh_x, W_h.stride(0), dx, ldx, I, accumulate=True)
        pgrads = (dW_z, db_z, dW_r, dU_r, db_u, dW_h, db_h)
        if defer_wgrads_enabled() and can_publish(*ctx.param_refs):
            for q, g in zip(ctx.param_refs, pgrads):      # summed once per parameter at the end of the pass (see _DEFER)
                _defer_sum(q, g)
            pgrads = (None,) * 7
        return (dHin[:, :H], dx, None, None, *pgrads, None, None, None)


def gru_sparse(h_in, x_sub, submess, bgraph_sub, W_z, b_z, W_r, U_r, b_u, W_h, b_h, depth: int, I: int, H: int):
    return _GruSparse.apply(h_in, x_sub, submess, bgraph_sub, W_z, b_z, W_r, U_r, b_u, W_h, b_h, depth, I, H)


class _LstmSparse(torch.autograd.Function):
    """LSTM.sparse_forward (ggpm/rnn.py:110-121): recompute rows `submess` of the (h, c) state `depth` times."""

    @staticmethod
    def forward(ctx, h_in, c_in, x_sub, submess, bgraph_sub, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, depth, I, H):
        _need_gpu(h_in, c_in, x_sub, submess, bgraph_sub, W_i)
        lib = _lib.load()
        E1, Hp, ms = h_in.shape[0], padded_hidden(H), submess.numel()
        f32 = dict(dtype=torch.float32, device=h_in.device)
        save = any(ctx.needs_input_grad)
        hp, cp = _as_padded_state(h_in, H, Hp), _as_padded_state(c_in, H, Hp)
        frozen, pred, _ = _sparse_structure(E1, submess, bgraph_sub)
        Ws, bs = (W_i, W_o, W_u, W_f), (b_i, b_o, b_u, b_f)
        ldx = _ld(x_sub)
        Xs = torch.empty(4, ms, Hp, **f32)
        for k in range(4):
            gemm(0, 1, ms, H, I, x_sub, ldx, Ws[k][:, :I], Ws[k].stride(0), Xs[k], Hp, Hp, bias=bs[k])
        X = torch.zeros(4, E1, Hp, **f32)
        X.index_copy_(1, submess, Xs)
        wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H)), **f32)
        if save:
            Hs = torch.empty(depth + 1, E1, Hp, **f32)
            Cs = torch.empty(depth + 1, E1, Hp, **f32)
            Qs = torch.empty(depth, E1, Hp, **f32)
            St = torch.empty(5, depth, E1, Hp, **f32)
            Ss, Is, Os, Us, Fs = St[0], St[1], St[2], St[3], St[4]
        else:
            Hs = torch.empty(2, E1, Hp, **f32)
            Cs = torch.empty(2, E1, Hp, **f32)
            Qs = torch.empty(2, E1, Hp, **f32)
            Ss = Is = Os = Us = Fs = None
        Wh = [w[:, I:] for w in Ws]
        _lib.check(lib.ggpm_lstm_sparse_forward(E1, H, depth, _p(hp), _p(cp), _p(frozen), _p(X[0]), _p(X[1]), _p(X[2]),
                                                _p(X[3]), _p(Wh[0]), W_i.stride(0), _p(Wh[1]), W_o.stride(0), _p(Wh[2]),
                                                W_u.stride(0), _p(Wh[3]), W_f.stride(0), _p(pred.rowptr), _p(pred.col),
                                                _p(Hs), _p(Cs), _p(Qs), _p(Ss), _p(Is), _p(Os), _p(Us), _p(Fs), _p(wpack),
                                                int(save), _stream()), "lstm_sparse_forward")
        k = depth if save else depth & 1
        if save:
            ctx.save_for_backward(x_sub, submess, W_i, W_o, W_u, W_f)
            ctx.stash = (X[3], frozen, pred, Hs, Cs, Qs, Ss, Is, Os, Us, Fs)
            ctx.meta = (depth, I, H)
            ctx.param_refs = (W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f)
        return Hs[k][:, :H], Cs[k][:, :H]

    @staticmethod
    def backward(ctx, dH, dC):
        x_sub, submess, W_i, W_o, W_u, W_f = ctx.saved_tensors
        Xf, frozen, pred, Hs, Cs, Qs, Ss, Is, Os, Us, Fs = ctx.stash
        depth, I, H = ctx.meta
        lib = _lib.load()
        E1, Hp, ms = Hs.shape[1], padded_hidden(H), submess.numel()
        f32 = dict(dtype=torch.float32, device=x_sub.device)
        succ = pred.T
        dHD = torch.zeros(E1, Hp, **f32)
        dCD = torch.zeros(E1, Hp, **f32)
        if dH is not None:
            dHD[:, :H] = dH
        if dC is not None:
            dCD[:, :H] = dC
        dHin, dCin = torch.empty(E1, Hp, **f32), torch.empty(E1, Hp, **f32)
        Ws = (W_i, W_o, W_u, W_f)
        dX = torch.empty(4, E1, Hp, **f32)
        dWs = [torch.empty(w.shape, **f32) for w in Ws]
        Wh = [w[:, I:] for w in Ws]
        dWh = [w[:, I:] for w in dWs]
        wb = int(lib.ggpm_lstm_backward_workspace_bytes(E1, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        _lib.check(lib.ggpm_lstm_sparse_backward(E1, H, depth, _p(frozen), _p(Xf), _p(Wh[0]), W_i.stride(0), _p(Wh[1]),
                                                 W_o.stride(0), _p(Wh[2]), W_u.stride(0), _p(Wh[3]), W_f.stride(0),
                                                 _p(pred.rowptr), _p(pred.col), _p(succ.rowptr), _p(succ.col), _p(Hs),
                                                 _p(Cs), _p(Qs), _p(Ss), _p(Is), _p(Os), _p(Us), _p(Fs), _p(dHD),
                                                 _p(dCD), _p(dHin), _p(dCin), _p(dX[0]), _p(dX[1]), _p(dX[2]), _p(dX[3]),
                                                 _p(dWh[0]), dWs[0].stride(0), _p(dWh[1]), dWs[1].stride(0), _p(dWh[2]),
                                                 dWs[2].stride(0), _p(dWh[3]), dWs[3].stride(0), _p(work),
                                                 work.numel() * 4, _stream()), "lstm_sparse_backward")
        ctx.stash = None
        dXs = dX.index_select(1, submess)
        ldx = _ld(x_sub)
        dbs = []
        for k in range(4):
            gemm(1, 0, H, I, ms, dXs[k], Hp, x_sub, ldx, dWs[k][:, :I], dWs[k].stride(0), I, splitk=True)
            dbs.append(colsum(dXs[k], ms, H))
        dx = None
        if ctx.needs_input_grad[2]:
            dx = _empty_same_layout(x_sub)
            for k in range(4):
                gemm(0, 0, ms, I, H, dXs[k], Hp, Ws[k][:, :I], Ws[k].stride(0), dx, ldx,
                     x_sub.shape[1] if k == 0 else I, accumulate=k > 0)
        pgrads = (dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2], dWs[3], dbs[3])
        if defer_wgrads_enabled() and can_publish(*ctx.param_refs):
            for q, g in zip(ctx.param_refs, pgrads):      # summed once per parameter at the end of the pass (see _DEFER)
                _defer_sum(q, g)
            pgrads = (None,) * 8
        return (dHin[:, :H], dCin[:, :H], dx, None, None, *pgrads, None, None, None)


def lstm_sparse(h_in, c_in, x_sub, submess, bgraph_sub, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, depth, I, H):
    return _LstmSparse.apply(h_in, c_in, x_sub, submess, bgraph_sub, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, depth, I, H)


class _LstmLevel(torch.autograd.Function):
    """LSTM.forward (ggpm/rnn.py:96-108) for one level; returns (h_D, c_D) as [E1, Hp] tensors."""

    @staticmethod
    def forward(ctx, x, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, pred, depth, I, H, gate_dtype=0):
        _need_gpu(x, W_i, W_o, W_u, W_f)
        lib = _lib.load()
        ctx.gate_dtype = gate_dtype
        E1, Hp = x.shape[0], padded_hidden(H)
        f32 = dict(dtype=torch.float32, device=x.device)
        save = any(ctx.needs_input_grad)
        Ws, bs = (W_i, W_o, W_u, W_f), (b_i, b_o, b_u, b_f)
        X = torch.empty(4, E1, Hp, **f32)
        ldx = _ld(x)
        for k in range(4):
            gemm(0, 1, E1, H, I, x, ldx, Ws[k][:, :I], Ws[k].stride(0), X[k], Hp, Hp, bias=bs[k])
        wpack = torch.empty(int(lib.ggpm_lstm_pack_floats(H)), **f32)
        if save:
            Hs = torch.empty(depth + 1, E1, Hp, **f32)
            Cs = torch.empty(depth + 1, E1, Hp, **f32)
            Qs = torch.empty(depth, E1, Hp, **f32)
            St = torch.empty(5, depth, E1, Hp, **f32)
            Ss, Is, Os, Us, Fs = St[0], St[1], St[2], St[3], St[4]
        else:
            Hs = torch.empty(2, E1, Hp, **f32)
            Cs = torch.empty(2, E1, Hp, **f32)
            Qs = torch.empty(2, E1, Hp, **f32)
            Ss = Is = Os = Us = Fs = None
        Wh = [w[:, I:] for w in Ws]
        with _gate_dtype(gate_dtype):
            _lib.check(lib.ggpm_lstm_forward(E1, H, depth, _p(X[0]), _p(X[1]), _p(X[2]), _p(X[3]), _p(Wh[0]), W_i.stride(0),
                                             _p(Wh[1]), W_o.stride(0), _p(Wh[2]), W_u.stride(0), _p(Wh[3]), W_f.stride(0),
                                             _p(pred.rowptr), _p(pred.col), _p(Hs), _p(Cs), _p(Qs), _p(Ss), _p(Is), _p(Os),
                                             _p(Us), _p(Fs), _p(wpack), int(save), _stream()), "lstm_forward")
        k = depth if save else depth & 1
        if save:
            ctx.save_for_backward(x, W_i, W_o, W_u, W_f)
            ctx.stash = (X[3], Hs, Cs, Qs, Ss, Is, Os, Us, Fs)
            ctx.meta = (pred, depth, I, H)
            ctx.params = (W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f)
        c_out = Cs[k]
        ctx.mark_non_differentiable(c_out)
        return Hs[k], c_out

    @staticmethod
    def backward(ctx, dHD, _dC):
        x, W_i, W_o, W_u, W_f = ctx.saved_tensors
        Xf, Hs, Cs, Qs, Ss, Is, Os, Us, Fs = ctx.stash
        pred, depth, I, H = ctx.meta
        lib = _lib.load()
        E1, Hp = x.shape[0], padded_hidden(H)
        succ = pred.T
        dHD = dHD.contiguous()
        f32 = dict(dtype=torch.float32, device=x.device)
        Ws = (W_i, W_o, W_u, W_f)
        dX = torch.empty(4, E1, Hp, **f32)
        dWs = [torch.empty(w.shape, **f32) for w in Ws]
        Wh = [w[:, I:] for w in Ws]
        dWh = [w[:, I:] for w in dWs]
        wb = int(lib.ggpm_lstm_backward_workspace_bytes(E1, H, depth))
        work = torch.empty((wb + 3) // 4, **f32)
        use_side = side_stream_enabled() and can_publish(*ctx.params)
        with _gate_dtype(ctx.gate_dtype):
            _lib.check(lib.ggpm_lstm_backward(E1, H, depth, _p(Xf), _p(Wh[0]), W_i.stride(0), _p(Wh[1]), W_o.stride(0),
                                              _p(Wh[2]), W_u.stride(0), _p(Wh[3]), W_f.stride(0), _p(pred.rowptr),
                                              _p(pred.col), _p(succ.rowptr), _p(succ.col), _p(Hs), _p(Cs), _p(Qs), _p(Ss),
                                              _p(Is), _p(Os), _p(Us), _p(Fs), _p(dHD), _p(dX[0]), _p(dX[1]), _p(dX[2]),
                                              _p(dX[3]), _p(dWh[0]), dWs[0].stride(0), _p(dWh[1]), dWs[1].stride(0),
                                              _p(dWh[2]), dWs[2].stride(0), _p(dWh[3]), dWs[3].stride(0), _p(work),
                                              work.numel() * 4, 0 if use_side else 1, _stream()), "lstm_backward")
        ctx.stash = None
        ldx = _ld(x)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty_same_layout(x)
            for k in range(4):
                gemm(0, 0, E1, I, H, dX[k], Hp, Ws[k][:, :I], Ws[k].stride(0), dx, ldx, x.shape[1] if k == 0 else I,
                     accumulate=k > 0)

        def weight_grads():
            if use_side:
                with _gate_dtype(ctx.gate_dtype):
                    _lib.check(lib.ggpm_lstm_weight_grads(E1, H, depth, _p(Hs), _p(Ss), _p(work), work.numel() * 4,
                                                          _p(dWh[0]), dWs[0].stride(0), _p(dWh[1]), dWs[1].stride(0),
                                                          _p(dWh[2]), dWs[2].stride(0), _p(dWh[3]), dWs[3].stride(0),
                                                          _stream()), "lstm_weight_grads")
            out = []
            for k in range(4):
                gemm(1, 0, H, I, E1, dX[k], Hp, x, ldx, dWs[k][:, :I], dWs[k].stride(0), I, splitk=True)
                out.append(colsum(dX[k], E1, H))
            return out

        if use_side:
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)
            for tns in [work, dX, Hs, Ss, x] + dWs:
                tns.record_stream(side)
            with torch.cuda.stream(side):
                dbs = weight_grads()
                for k in range(4):
                    _accumulate_grad(ctx.params[2 * k], dWs[k], main)
                    _accumulate_grad(ctx.params[2 * k + 1], dbs[k], main)
            _join_later(main, side)
            return (dx,) + (None,) * 13
        dbs = weight_grads()
        return (dx, dWs[0], dbs[0], dWs[1], dbs[1], dWs[2], dbs[2], dWs[3], dbs[3], None, None, None, None, None)


def lstm_level(x, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, pred: CSR, depth: int, I: int, H: int, gate_dtype=None):
    return _LstmLevel.apply(x, W_i, b_i, W_o, b_o, W_u, b_u, W_f, b_f, pred, depth, I, H, GATE_DTYPES[gate_dtype])
