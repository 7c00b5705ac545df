"""Load the golden fixtures (tests/golden/*.npz, produced by make_golden.py from the reference)."""
import glob
import os

import numpy as np
import torch

from ggpm_amd.params import (encoder_param_shapes, vae_head_shapes, seeded_state_dict,
                             motif_encoder_param_shapes, score_head_shapes)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_PROBE = 64


def case_names(prefix="", motif=False):
    """HierMPNEncoder fixtures by default; ``motif=True`` lists the MotifEncoder fixtures instead."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    names = [n for n in names if not n.startswith(("sparse_", "inc_", "heads_", "vae_", "attach_"))]   # own tests
    return [n for n in names if n.startswith("motif_") == motif]


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        m = self.z["meta"]
        (self.H, self.latent, self.depthT, self.depthG, self.B, self.n_motif, self.n_attach, self.seed,
         m0, m1, full) = [int(x) for x in m]
        self.motifs = (m0, m1)
        self.full = bool(full)
        self.rnn = str(self.z["rnn"])
        self.beta = float(self.z["beta"])

    def tensors(self, device="cpu"):
        z = self.z
        tree = [torch.from_numpy(z["tree_" + k].astype(np.int64)).to(device)
                for k in ("fnode", "fmess", "agraph", "bgraph", "cgraph")]
        tree.append([tuple(int(v) for v in r) for r in z["tree_scope"]])
        graph = [torch.from_numpy(z["graph_" + k].astype(np.int64)).to(device)
                 for k in ("fnode", "fmess", "agraph", "bgraph")]
        graph.append([tuple(int(v) for v in r) for r in z["graph_scope"]])
        return tree, graph

    def numpy_tensors(self):
        z = self.z
        tree = [z["tree_" + k] for k in ("fnode", "fmess", "agraph", "bgraph", "cgraph")]
        tree.append([tuple(int(v) for v in r) for r in z["tree_scope"]])
        graph = [z["graph_" + k] for k in ("fnode", "fmess", "agraph", "bgraph")]
        graph.append([tuple(int(v) for v in r) for r in z["graph_scope"]])
        return tree, graph

    def params(self, dtype=torch.float32, device="cpu", requires_grad=False):
        if self.name.startswith("motif_"):
            sd = seeded_state_dict(motif_encoder_param_shapes(self.rnn, self.H, self.n_motif, self.n_attach), self.seed)
        else:
            sd = seeded_state_dict(encoder_param_shapes(self.rnn, self.H, self.n_motif, self.n_attach), self.seed)
            sd.update(seeded_state_dict(vae_head_shapes(self.H, self.latent), self.seed + 7))
        out = {}
        for k, v in sd.items():
            t = torch.from_numpy(v).to(dtype).to(device)
            if requires_grad:
                t.requires_grad_(True)
            out[k] = t
        return out

    def loss_coeffs(self, shapes):
        rs = np.random.RandomState(self.seed + 1000)
        return [rs.standard_normal(s).astype(np.float32) for s in shapes]

    def probe_indices(self, pname, numel):
        h = 0
        for ch in pname:
            h = (h * 131 + ord(ch)) % (2 ** 31)
        rs = np.random.RandomState((h + self.seed) % (2 ** 31))
        return rs.randint(0, numel, size=min(N_PROBE, numel))

    def check_grad(self, pname, g, rel, tag=""):
        """Compare a gradient array with the fixture (full array or probes + statistics)."""
        g = np.asarray(g, dtype=np.float64)
        if self.full:
            ref = self.z["grad/" + pname + tag].astype(np.float64)
            scale = max(np.abs(ref).max(), 1e-12)
            err = np.abs(g - ref).max() / scale
            assert err <= rel, "%s grad %s: rel err %.3e" % (self.name, pname, err)
        else:
            idx = self.probe_indices(pname, g.size)
            ref = self.z["gprobe/" + pname + tag].astype(np.float64)
            stat = self.z["gstat/" + pname + tag].astype(np.float64)
            scale = max(stat[2], 1e-12)
            err = np.abs(g.reshape(-1)[idx] - ref).max() / scale
            assert err <= rel, "%s grad probe %s: rel err %.3e" % (self.name, pname, err)
            l2 = np.sqrt((g ** 2).sum())
            assert abs(l2 - stat[1]) <= rel * max(stat[1], 1e-12) * 4, "%s grad l2 %s: %g vs %g" % (self.name, pname, l2, stat[1])


def rel_err(a, b):
    """max|a-b| / max|b| (the 1e-4 bar of BASELINE.json is on this quantity)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def elem_rel_err(a, b, floor=1e-6):
    """SURVEY.md section 8(d)'s per-element form: max_i |a_i - b_i| / max(|b_i|, floor * ||b||_inf).

    With the survey's floor of 1e-6 a 1e-4 bound asks elements a millionth of the tensor's scale to carry four correct
    digits, i.e. an absolute error of 1e-10 of the scale -- below fp32's own resolution (6e-8) of the sums those
    elements come from.  The reference cannot meet that against itself: its fp32 run differs from its fp64 run by up
    to 3.5e-2 in this measure on the committed fixtures (``test_per_element_criterion_is_calibrated``, CPU), and with
    a floor of 1 % of the scale still by 5e-4 on outputs and 3e-3 on gradients at configs[1] size
    (profiles/r02_parity_report_*.txt).  The tests therefore apply the per-element form RELATIVE TO THAT NOISE:
    ``assert_close(a, b32, what, b64=...)`` demands that no element of the HIP result is further from the fp64 answer
    than ELEM_SLACK times what the reference's own fp32 run is (floor ELEM_FLOOR), next to the norm-wise 1e-4 bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), floor * max(float(np.abs(b).max()), 1e-300))
    return float((np.abs(a - b) / den).max())


ELEM_FLOOR = 1e-2
ELEM_SLACK = 8.0
ELEM_TOL = 1e-3      # the reference's fp32 vs fp64 outputs reach 5e-4 in this measure


def assert_close(a, b, what, tol=1e-4, elem_tol=ELEM_TOL, b64=None, slack=None):
    """``a`` (HIP) against ``b`` (the reference / oracle in fp32): the norm-wise 1e-4 bar of BASELINE.json (where the
    fp32 reference is itself further than that from its fp64 run -- GRU at depth 30 on 50-motif trees,
    profiles/r02_parity_report_configs4_gru.txt -- the bar becomes ELEM_SLACK x the reference's own distance), plus the
    per-element form of SURVEY section 8(d) (see elem_rel_err): against the fp64 run ``b64`` when given -- at most
    ELEM_SLACK x the fp32 reference's own per-element distance to fp64 (never asked to be below ELEM_TOL: column sums
    of ~1e4 terms with cancellation land anywhere within a few 1e-4 of each other, HIP or reference) -- else against
    ``b`` with the absolute bound ``elem_tol`` (None: skipped)."""
    slack = ELEM_SLACK if slack is None else slack
    e = rel_err(a, b)
    if b64 is None or rel_err(b, b64) * slack < tol:
        assert e < tol, "%s: norm-wise rel err %.3e" % (what, e)
    else:       # ill-conditioned case: the fp32 reference itself is further than tol / ELEM_SLACK from the fp64 answer
        n64, e64 = rel_err(b, b64), rel_err(a, b64)
        assert e64 <= slack * n64, "%s: norm-wise err vs fp64 %.3e, the fp32 reference's own is %.3e" % (what, e64, n64)
    if np.abs(np.asarray(b)).max() == 0:
        return e
    if b64 is not None:
        noise = elem_rel_err(b, b64, ELEM_FLOOR)
        pe = elem_rel_err(a, b64, ELEM_FLOOR)
        assert pe <= max(slack * noise, ELEM_TOL), \
            "%s: per-element err vs fp64 %.3e, the fp32 reference's own is %.3e (floor %g)" % (what, pe, noise, ELEM_FLOOR)
    elif elem_tol is not None:
        pe = elem_rel_err(a, b, ELEM_FLOOR)
        assert pe < elem_tol, "%s: per-element rel err %.3e (floor %g)" % (what, pe, ELEM_FLOOR)
    return e


def dropout_keep(rows, cols, p, seed_lo, seed_hi, site):
    """The keep mask of ggpm_dropout (include/ggpm_hip.h), restated in numpy: [rows, cols] bool."""
    M = np.uint64(0xFFFFFFFF)

    def fmix(h):
        h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85EBCA6B)) & M
        h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xC2B2AE35)) & M
        return h ^ (h >> np.uint64(16))

    idx = np.arange(rows * cols, dtype=np.uint64)
    h = fmix((idx * np.uint64(0x9E3779B1) + np.uint64(seed_lo)) & M)
    h = fmix(h ^ ((np.uint64(seed_hi) + np.uint64(site) * np.uint64(0x7F4A7C15)) & M))
    return ((h >> np.uint64(8)) >= np.uint64(int(p * 16777216.0))).reshape(rows, cols)


def sparse_inputs(E1, I, H, ms, K, seed):
    """Seeded inputs of a sparse_forward call (shared with the tests): state, subset, its inputs and predecessors."""
    rs = np.random.RandomState(seed)
    h = (0.5 * rs.standard_normal((E1, H))).astype(np.float32)
    c = (0.5 * rs.standard_normal((E1, H))).astype(np.float32)
    h[0] = 0; c[0] = 0
    submess = rs.choice(np.arange(1, E1), size=ms, replace=False).astype(np.int64)
    x = rs.standard_normal((ms, I)).astype(np.float32)
    bg = np.zeros((ms, K + 1), dtype=np.int64)
    for i in range(ms):
        k = rs.randint(0, K + 1)
        bg[i, :k] = rs.choice(np.arange(1, E1), size=k, replace=False)     # inside AND outside the subset
    coef = rs.standard_normal((2, E1, H)).astype(np.float32)
    return h, c, submess, x, bg, coef




def inc_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "inc_*.npz")))


class IncGolden(Golden):
    """Teacher-forced incremental-encoder fixtures (tests/golden/make_golden_inc.py)."""

    def __init__(self, name):
        super().__init__(name)
        self.kind = str(self.z["kind"])

    def params(self, dtype=torch.float32, device="cpu", requires_grad=False):
        shapes = (encoder_param_shapes if self.kind == "hier" else motif_encoder_param_shapes)(
            self.rnn, self.H, self.n_motif, self.n_attach)
        out = {}
        for k, v in seeded_state_dict(shapes, self.seed).items():
            if k.startswith("W_root"):
                continue
            t = torch.from_numpy(v).to(dtype).to(device)
            out[k] = t.requires_grad_(True) if requires_grad else t
        return out

    def schedule(self, device="cpu"):
        z = self.z
        cols = []
        for k in ("subnode", "submess", "atoms", "bonds"):
            flat, off = z["sched_" + k].astype(np.int64), z["sched_" + k + "_off"]
            cols.append([torch.from_numpy(flat[off[i]:off[i + 1]]).to(device) for i in range(len(off) - 1)])
        return list(zip(*cols))

    def init_vecs(self, device="cpu", dtype=torch.float32):
        return torch.from_numpy(self.z["init_vecs"]).to(dtype).to(device).requires_grad_(True)

    def output_keys(self):
        keys = ["topo", "cls", "tree_mess"]
        if self.kind == "hier":
            keys += ["inter_mess", "graph_mess", "graph_node", "inter_node"]
        return keys


def heads_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "heads_*.npz")))


class HeadsGolden:
    """Decoder score-head fixtures (tests/golden/make_golden_heads.py)."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.H, self.L, self.n_motif, self.n_attach, self.B, self.seed = [int(v) for v in self.z["meta"]]

    def params(self, dtype=torch.float32, device="cpu", requires_grad=False):
        sd = seeded_state_dict(score_head_shapes(self.H, self.L, self.H, self.n_motif, self.n_attach), self.seed)
        out = {}
        for k, v in sd.items():
            t = torch.from_numpy(v).to(dtype).to(device)
            out[k] = t.requires_grad_(True) if requires_grad else t
        return out

    def inputs(self, dtype=torch.float32, device="cpu"):
        fl = {k[3:]: torch.from_numpy(self.z[k]).to(dtype).to(device).requires_grad_(True)
              for k in self.z.files if k.startswith("in/")}
        ix = {k[4:]: torch.from_numpy(self.z[k]).to(device) for k in self.z.files if k.startswith("idx/")}
        return fl, ix

    def check_grad(self, pname, g, rel):
        g = np.asarray(g, dtype=np.float64)
        if "grad/" + pname in self.z.files:
            want = self.z["grad/" + pname]
            if np.abs(want).max() < 1e-6:      # analytically zero (W_assm.bias: the candidates of a row share one
                assert np.abs(g).max() < 1e-5, pname      # context and softmax gradients sum to 0): rounding noise only
                return
            assert rel_err(g, want) < rel, pname
            return
        h = 0
        for ch in pname:
            h = (h * 131 + ord(ch)) % (2 ** 31)
        rs = np.random.RandomState((h + self.seed) % (2 ** 31))
        idx = rs.randint(0, g.size, size=min(N_PROBE, g.size))
        stat = self.z["gstat/" + pname]
        scale = max(float(stat[2]), 1e-12)
        assert np.abs(g.reshape(-1)[idx] - self.z["gprobe/" + pname]).max() <= rel * scale, pname
        assert abs(np.sqrt((g ** 2).sum()) - stat[1]) <= rel * max(stat[1], 1e-12) * 10, pname


def vae_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "vae_*.npz")))


class VaeGolden(Golden):
    """Full-VAE-step fixtures (tests/golden/make_golden_vae.py): the reference's HierPropertyVAE forward + backward."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        (self.H, self.latent, self.depthT, self.depthG, self.diterT, self.diterG, self.B, self.n_motif, self.n_attach,
         self.seed, m0, m1, tie, full) = [int(v) for v in self.z["meta"]]
        self.motifs, self.tie, self.full = (m0, m1), bool(tie), bool(full)
        self.rnn = str(self.z["rnn"])
        self.beta = float(self.z["beta"])

    def specs(self):
        from ggpm_amd import synth
        return synth.random_batch(self.seed, self.B, motifs=self.motifs, n_motif_vocab=self.n_motif,
                                  n_attach_vocab=self.n_attach)

    def state_dict(self):
        from ggpm_amd.params import vae_param_shapes, tied_state_dict
        sd = seeded_state_dict(vae_param_shapes(self.rnn, self.H, self.latent, self.n_motif, self.n_attach), self.seed)
        return tied_state_dict(sd) if self.tie else sd

    def args(self, vocab):
        class A:
            pass
        a = A()
        a.vocab, a.rnn_type, a.embed_size, a.hidden_size = vocab, self.rnn, self.H, self.H
        a.atom_vocab = type("V", (), {"size": lambda s: 38})()
        a.depthT, a.depthG, a.diterT, a.diterG = self.depthT, self.depthG, self.diterT, self.diterG
        a.dropout, a.latent_size, a.tie_embedding = 0.0, self.latent, self.tie
        return a

    def ref_steps(self):
        z = self.z
        cols = []
        for k in ("subnode", "submess", "atoms", "bonds"):
            flat, off = z["ref_" + k], z["ref_" + k + "_off"]
            cols.append([flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)])
        return list(zip(*cols))

    def check_grad(self, pname, g, rel):
        g = np.asarray(g, dtype=np.float64)
        if "grad/" + pname in self.z.files:
            want = self.z["grad/" + pname].astype(np.float64)
            scale = np.abs(want).max()
            if scale < 1e-7 or pname.endswith("W_assm.bias"):
                # analytically zero (W_assm.bias: the candidates of a row share one context vector and the softmax
                # gradients of a row sum to 0): both sides hold rounding noise only
                assert np.abs(g).max() < 1e-4 and scale < 1e-4, pname
                return
            err = np.abs(g - want).max() / scale
            assert err <= rel, "%s grad %s: rel err %.3e" % (self.name, pname, err)
            return
        idx = self.probe_indices(pname, g.size)
        stat = self.z["gstat/" + pname].astype(np.float64)
        scale = max(stat[2], 1e-12)
        err = np.abs(g.reshape(-1)[idx] - self.z["gprobe/" + pname]).max() / scale
        assert err <= rel, "%s grad probe %s: rel err %.3e" % (self.name, pname, err)
        l2 = np.sqrt((g ** 2).sum())
        assert abs(l2 - stat[1]) <= rel * max(stat[1], 1e-12) * 4, "%s grad l2 %s: %g vs %g" % (self.name, pname, l2, stat[1])


# ---------------------------------------------------------------------------------------------- fp32 evaluation orders
def reversed_slots(table):
    """A zero-padded neighbour table with the real entries of every row in REVERSED order (still left-packed, trailing
    zero column kept): the same sets, summed by the reference's ops in another order."""
    t = np.asarray(table)
    f = t[:, ::-1]
    order = np.argsort(f == 0, axis=1, kind="stable")
    return np.ascontiguousarray(np.take_along_axis(f, order, axis=1))


def oracle_encoder_result(rnn, depth, sd, tree, graph, dtype=torch.float32, hoisted=False, threads=None):
    """One oracle evaluation of the test loss  kl + sum_o |o|^2  -> {name: array} (outputs, kl, 'grad <param>')."""
    from oracle import ref_encoder as ref
    old_threads, old_rne = torch.get_num_threads(), ref.rne_bf16
    try:
        if threads:
            torch.set_num_threads(threads)
        if hoisted:
            ref.rne_bf16 = lambda t: t            # the per-message / split-halves restatement, nothing rounded
        p = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in sd.items()}
        tt, gt = ref.to_long_tensors(tree), ref.to_long_tensors(graph)
        routs = ref.hier_encoder_forward(p, rnn, depth, depth, tt, gt, gate_dtype="bf16w" if hoisted else "f32")
        _, rkl = ref.rsample_kl(p, routs[0])
        (rkl + sum((o * o).sum() for o in routs)).backward()
        r = {n: o.detach().numpy() for n, o in zip(("hroot", "hnode", "hinter", "hatom"), routs)}
        for k, v in p.items():
            r["grad " + k] = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))
        r["kl"] = np.asarray(float(rkl.detach()))
        return r
    finally:
        torch.set_num_threads(old_threads)
        ref.rne_bf16 = old_rne


def reversed_batch(tree, graph):
    """(tree, graph) with the predecessor / incoming / cluster lists of every row reversed."""
    rev = lambda t, idx: tuple(reversed_slots(x) if i in idx else x for i, x in enumerate(t[:-1])) + (t[-1],)
    return rev(tree, (2, 3, 4)), rev(graph, (2, 3))


# The equivalent fp32 evaluation orders: name -> arguments of a job (oracle_worker.py / oracle_encoder_result).  Four
# restatements -- 'padded' (the reference's op order), 'hoisted' (recurrent products applied once per message, gate weights
# split into input / hidden halves: exact algebra, other summation order), 'slots_reversed' (the predecessor / incoming /
# cluster lists of every row reversed), 'hoisted_reversed' (both) -- each with 1, 2, 3 and 4 BLAS threads (another blocking
# of every product): sixteen evaluations.  Thread counts are explicit so that the family is the same on every host; on the
# ill-conditioned configs[4] GRU case single tensors land anywhere between 7e-4 and 6e-3 from the fp64 run across it
# (gradient of the attachment level's W_r, 20 evaluations on one CPU), which is why five samples were too few: the worst of
# five moved by 2x with the thread count alone.
FP32_ORDER_BASES = {"padded": {}, "hoisted": {"hoisted": True}, "slots_reversed": {"reverse": True},
                    "hoisted_reversed": {"hoisted": True, "reverse": True}}
FP32_ORDERS = {"%s@%d" % (name, t): dict(j, threads=t) for name, j in FP32_ORDER_BASES.items() for t in (1, 2, 3, 4)}


def oracle_fp32_orders(rnn, depth, sd, tree, graph):
    """The oracle's fp32 arithmetic in the equivalent evaluation orders of FP32_ORDERS -- what "the reference's fp32
    result" is known up to -- one after the other in this process (OracleRuns: side by side)."""
    tree_r, graph_r = reversed_batch(tree, graph)
    out = {}
    for name, j in FP32_ORDERS.items():
        t, g = (tree_r, graph_r) if j.get("reverse") else (tree, graph)
        out[name] = oracle_encoder_result(rnn, depth, sd, t, g, hoisted=bool(j.get("hoisted")), threads=j.get("threads"))
    return out


class OracleRuns:
    """Several oracle evaluations of one batch, each in its OWN process with a share of the host's cores
    (tests/oracle_worker.py): the constructor returns at once -- the caller runs the HIP path meanwhile -- and ``results()``
    waits.  The big parity cases spend nearly all their time in these CPU runs (fp32 + fp64, the calibrated case four
    more fp32 orders); side by side they take as long as the slowest one.  At most MAX_PROCS run at a time: autograd's
    first backward() asks every backend for its device count, which OPENS the GPU in a CPU-only process too
    (tools/probe/gpu_open_probe.sh: /dev/kfd and the render node appear after a CPU backward, visible-devices variables
    or not), and a GPU box allows six such processes -- the test process and four workers stay below that.
    A worker that fails raises here with its stderr."""
    MAX_PROCS = 4

    def __init__(self, rnn, depth, sd, tree, graph, jobs):
        import pickle
        import tempfile
        self.dir = tempfile.TemporaryDirectory(prefix="ggpm_oracle_")
        from ggpm_amd.launcher import host_cores          # (affinity capped by the cgroup quota: a GPU box shows 256 CPUs and
        cores = host_cores()                               # grants 16 -- workers with 128 threads each took 70-110 s per case)
        share = max(1, min(16, cores // max(1, min(len(jobs), self.MAX_PROCS))))      # (the oracle's small ops gain nothing past 16 threads)
        self.pending, self.procs = [], {}
        for name, j in jobs.items():
            job = dict(rnn=rnn, depth=depth, sd=sd, tree=tree, graph=graph, **j)
            job.setdefault("threads", share)
            src, dst = os.path.join(self.dir.name, name + ".job"), os.path.join(self.dir.name, name + ".out")
            with open(src, "wb") as f:
                pickle.dump(job, f, protocol=pickle.HIGHEST_PROTOCOL)
            # (the slow ones first: fp64, then the jobs with the fewest threads)
            self.pending.append((0 if j.get("dtype") == "f64" else job["threads"], name, src, dst, job["threads"]))
        self.pending.sort()
        self._pump()

    def _pump(self):
        import subprocess
        import sys
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_worker.py")
        running = sum(1 for p, _, _ in self.procs.values() if p.poll() is None)
        while self.pending and running < self.MAX_PROCS:
            _, name, src, dst, threads = self.pending.pop(0)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
            err = open(os.path.join(self.dir.name, name + ".err"), "wb")
            self.procs[name] = (subprocess.Popen([sys.executable, worker, src, dst], env=env, stdout=err, stderr=err), dst, err)
            running += 1

    def cancel(self):
        self.pending = []
        for p, _, err in self.procs.values():
            if p.poll() is None:
                p.kill()
            p.wait()
            err.close()
        self.dir.cleanup()

    def results(self, timeout=900):
        import pickle
        import time
        out, failed = {}, None
        t_wait = time.time()
        t_end = time.time() + timeout
        try:
            while self.pending or any(p.poll() is None for p, _, _ in self.procs.values()):
                if time.time() > t_end:
                    raise RuntimeError("oracle workers: no result after %d s" % timeout)
                self._pump()
                time.sleep(0.05)
            print("oracle runs: waited %.1f s for %d workers" % (time.time() - t_wait, len(self.procs)))
            for name, (p, dst, err) in self.procs.items():
                rc = p.returncode
                err.close()
                if rc != 0 and failed is None:
                    with open(err.name, "rb") as f:
                        failed = "oracle worker %r exited with %s:\n%s" % (name, rc, f.read().decode(errors="replace")[-2000:])
                if rc == 0:
                    with open(dst, "rb") as f:
                        out[name] = pickle.load(f)
                    with open(err.name, "rb") as f:          # (pytest -s shows where a worker's time went)
                        print("  %s: %s" % (name, f.read().decode(errors="replace").strip().splitlines()[-1:]))
        finally:
            self.pending = []
            for p, _, _ in self.procs.values():
                if p.poll() is None:
                    p.kill()
                    p.wait()
            self.dir.cleanup()
        if failed:
            raise RuntimeError(failed)
        return out
