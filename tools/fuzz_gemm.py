"""Random-shape check of every GEMM entry point against fp64 (dev tool, needs an MI355X): python tools/fuzz_gemm.py [cases]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ggpm_amd import functional as F_

dev = torch.device("cuda")
rs = np.random.RandomState(int(os.environ.get("SEED", "0")))
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
worst = 0.0


def rnd_ld(n, aligned):
    return (n + 3) // 4 * 4 + 4 * rs.randint(0, 3) if aligned else n + rs.randint(0, 4)


def mat(rows, cols, ld):
    a = rs.standard_normal((rows, ld)).astype(np.float32)
    return a, torch.from_numpy(a).to(dev)


def check(got, ref, what, pre=None):
    """Error relative to the size of the PRE-activation values: tanh / relu pass an absolute error of the sum through
    one to one near 0, while their outputs are O(1)."""
    global worst
    scale = max(np.abs(ref if pre is None else pre).max(), 1e-6)
    err = np.abs(got - ref).max() / scale
    worst = max(worst, err)
    assert err < 2e-5, (what, err)


def act_np(x, act):
    return np.maximum(x, 0) if act == F_.ACT_RELU else (np.tanh(x) if act == F_.ACT_TANH else x)


for case in range(ncases):
    kind = rs.choice(["gemm", "grouped", "ksegments", "tall"])
    aligned = rs.rand() < 0.8
    ta, tb = int(rs.randint(2)), int(rs.randint(2))
    M, N = int(rs.randint(1, 700)), int(rs.randint(1, 700))
    K = int(rs.randint(1, 1500))
    act = int(rs.choice([F_.ACT_NONE, F_.ACT_RELU, F_.ACT_TANH]))
    zr0 = bool(rs.randint(2))
    if kind == "tall":
        ta, tb, K = 1, 0, int(rs.randint(6200, 20000))
        M, N = int(rs.randint(100, 500)), int(rs.randint(100, 500))
    ldc = (N + 15) // 16 * 16 + 4 * rs.randint(0, 3)
    n_pad = int(rs.randint(N, ldc + 1))
    bias = rs.standard_normal(N).astype(np.float32)
    tbias = torch.from_numpy(bias).to(dev)
    if kind in ("gemm", "tall"):
        lda, ldb = rnd_ld(M if ta else K, aligned), rnd_ld(K if tb else N, aligned)
        A, tA = mat(K if ta else M, M if ta else K, lda)
        B, tB = mat(N if tb else K, K if tb else N, ldb)
        C0 = rs.standard_normal((M, ldc)).astype(np.float32)
        C = torch.from_numpy(C0.copy()).to(dev)
        acc = bool(rs.randint(2))
        F_.gemm(ta, tb, M, N, K, tA, lda, tB, ldb, C, ldc, n_pad, bias=tbias, accumulate=acc, act=act, zero_row0=zr0,
                splitk=bool(rs.randint(2)) or kind == "tall")
        Am = (A[:, :M].T if ta else A[:, :K]).astype(np.float64)
        Bm = (B[:, :K].T if tb else B[:, :N]).astype(np.float64)
        pre = Am @ Bm + bias + (C0[:, :N] if acc else 0)
        ref = act_np(pre, act)
        if zr0:
            ref[0] = 0
        got = C.cpu().numpy()
        check(got[:, :N], ref, (kind, ta, tb, M, N, K, aligned), pre)
        assert (got[:, N:n_pad] == 0).all() and (got[:, n_pad:] == C0[:, n_pad:]).all(), (kind, "pads", M, N, K)
    elif kind == "grouped":
        count = int(rs.randint(1, 5))
        probs, refs, c0s, pres = [], [], [], []
        for i in range(count):
            lda, ldb = rnd_ld(M if ta else K, aligned), rnd_ld(K if tb else N, aligned)
            A, tA = mat(K if ta else M, M if ta else K, lda)
            B, tB = mat(N if tb else K, K if tb else N, ldb)
            C0 = rs.standard_normal((M, ldc)).astype(np.float32)
            acc = bool(rs.randint(2))
            Am = (A[:, :M].T if ta else A[:, :K]).astype(np.float64)
            Bm = (B[:, :K].T if tb else B[:, :N]).astype(np.float64)
            pres.append(Am @ Bm + bias + (C0[:, :N] if acc else 0))
            refs.append(act_np(pres[-1], act))
            c0s.append(C0)
            probs.append(dict(A=tA, lda=lda, B=tB, ldb=ldb, C=torch.from_numpy(C0.copy()).to(dev), ldc=ldc, n_pad=n_pad,
                              bias=tbias, accumulate=acc, act=act))
        F_.gemm_grouped(ta, tb, M, N, K, probs)
        for q, ref, C0, pre in zip(probs, refs, c0s, pres):
            got = q["C"].cpu().numpy()
            check(got[:, :N], ref, (kind, ta, tb, M, N, K, count, aligned), pre)
            assert (got[:, N:n_pad] == 0).all() and (got[:, n_pad:] == C0[:, n_pad:]).all(), (kind, "pads")
    else:
        nseg = int(rs.randint(1, 5))
        Ks = [int(rs.randint(1, 700)) for _ in range(nseg)]
        As, Bs, ldas, ldbs = [], [], [], []
        ref = np.zeros((M, N))
        for Kk in Ks:
            lda, ldb = rnd_ld(Kk, aligned), rnd_ld(Kk if tb else N, aligned)
            A, tA = mat(M, Kk, lda)
            B, tB = mat(N if tb else Kk, Kk if tb else N, ldb)
            ref += A[:, :Kk].astype(np.float64) @ (B[:, :Kk].T if tb else B[:, :N]).astype(np.float64)
            As.append(tA); Bs.append(tB); ldas.append(lda); ldbs.append(ldb)
        C0 = rs.standard_normal((M, ldc)).astype(np.float32)
        C = torch.from_numpy(C0.copy()).to(dev)
        acc = bool(rs.randint(2))
        F_.gemm_ksegments(tb, M, N, As, ldas, Bs, ldbs, Ks, C, ldc, n_pad, bias=tbias, accumulate=acc, act=act, zero_row0=zr0)
        pre = ref + bias + (C0[:, :N] if acc else 0)
        ref = act_np(pre, act)
        if zr0:
            ref[0] = 0
        got = C.cpu().numpy()
        check(got[:, :N], ref, (kind, tb, M, N, Ks, aligned), pre)
        assert (got[:, N:n_pad] == 0).all() and (got[:, n_pad:] == C0[:, n_pad:]).all(), (kind, "pads")
print("%d cases ok, worst relative error %.2e" % (ncases, worst))
