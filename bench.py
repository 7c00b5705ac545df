#!/usr/bin/env python3
"""bench.py -- molecules/s of the hierarchical encoder training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 0..4] [--rnn GRU|LSTM]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

``--gpus N`` (N > 1) without WORLD_SIZE in the environment starts the N rank processes itself (ggpm_amd/launcher.py: the
parent makes no HIP call, every rank is a fresh process with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set); under
torch.distributed.run the ranks exist already and this file is one of them.

Default workload = BASELINE.json configs[1]: synthetic random-motif molecules (~40 atoms, motif vocab 500),
hidden = embed = 300, depthT = depthG = 20, batch 32 per GPU, fp32, GRU message function (the LSTM run of the same
workload is reported in the same line under "lstm").  ``--config N`` selects configs[N] of BASELINE.json (CONFIGS
below).  One "step" = zero_grad + HierMPNEncoder forward + KL heads + backward (+ gradient all-reduce over RCCL for
N > 1) + Adam update on one batch whose tensorized index tensors are already resident in HBM.  Rank r consumes its own
stream of batches (weak scaling: per-GPU work fixed); value = all molecules processed by all ranks / max-over-ranks
time.

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline"     -- the dominant kernel (the fused depth step of the ATOM level: one 16-wave workgroup per CU) timed with
                    HIP events on its own stream, ALGORITHMIC flops per launch / mean launch time vs the fp32 MFMA
                    peak; the same kernel's launches on the two small tree-side levels are reported beside it
                    ("tree_levels"), never averaged into it;
  "cpu_baseline" -- the oracle (padded reference op order, PyTorch CPU) timed on this box's host cores on a bounded
                    sample of the same batches (rank 0, N = 1 only; BASELINE.md section 3 protocol: 2 warm-up steps,
                    >= 10 timed, median, all cores and 8 threads, CPU model stated).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import ggpm_amd          # noqa: E402  (sets GPU_MAX_HW_QUEUES=8 before the HIP runtime starts; see ggpm_amd/__init__.py)
import numpy as np       # noqa: E402
import torch             # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_MFMA_BF16_TFLOPS = 2500.0   # dense bf16 (never the 2:1 sparsity figure)
PEAK_HBM_GBS = 8000.0

# BASELINE.json configs[i] -> concrete synthetic workload (C1..C5).  `gen`: motifs-per-molecule
# range of ggpm_amd.synth.random_molecule, or "mix" = the chem-trio size mix (synth.SIZE_MIX, true proportions).
CONFIGS = {
    0: dict(tag="configs[0]: pretrained_wo_tie_embedding shape on HOPV-15-sized synthetic molecules", rnn="LSTM",
            hidden=250, depth=20, latent=24, batch=20, vocab=(721, 6214), gen=(6, 14)),
    1: dict(tag="configs[1]: synthetic random-motif graphs (~40 atoms)", rnn="GRU", hidden=300, depth=20, latent=32,
            batch=32, vocab=(500, 1500), gen=(8, 12)),
    2: dict(tag="configs[2]: QM9-shaped synthetic molecules (1-3 motifs, ~9 atoms; data/qm9 is a missing blob)",
            rnn="GRU", hidden=300, depth=20, latent=32, batch=64, vocab=(500, 1500), gen=(1, 3)),
    3: dict(tag="configs[3]: pretrained_600_hidden_w_tie_embedding shape, chem-trio size mix, one GPU's shard of 32x8",
            rnn="LSTM", hidden=600, depth=20, latent=24, batch=32, vocab=(721, 6214), gen="mix"),
    4: dict(tag="configs[4]: synthetic large polymers (~200 atoms), one GPU's shard", rnn="GRU", hidden=600, depth=30,
            latent=32, batch=32, vocab=(500, 1500), gen=(46, 58)),
}
KERNELS = ["gru_fwd_a", "gru_bwd_a", "lstm_fwd_a", "lstm_bwd_a", "gru_fwd_b", "gru_bwd_b", "lstm_fwd_b", "lstm_bwd_b"]
LEVEL_TAGS = {1: "atom", 2: "attachment", 3: "motif"}


class _Vocab:
    def __init__(self, n):
        self._n = n

    def size(self):
        return self._n


class _Args:
    pass


def make_args(rnn, hidden, depth, latent, n_motif, n_attach):
    a = _Args()
    a.vocab, a.atom_vocab = _Vocab((n_motif, n_attach)), _Vocab(38)
    a.rnn_type, a.embed_size, a.hidden_size = rnn, hidden, hidden
    a.depthT = a.depthG = depth
    a.dropout, a.latent_size = 0.0, latent
    return a


def make_batches(n_batches, batch_size, seed0, gen, n_motif, n_attach):
    from ggpm_amd import synth
    out = []
    for i in range(n_batches):
        if gen == "mix":
            specs = synth.size_mix_batch(seed0 + i, batch_size, one_of_each=False, n_motif_vocab=n_motif,
                                         n_attach_vocab=n_attach)
        else:
            specs = synth.random_batch(seed0 + i, batch_size, motifs=tuple(gen), n_motif_vocab=n_motif,
                                       n_attach_vocab=n_attach)
        out.append(synth.tensorize(specs))
    return out


def algorithmic_work(batches, H, depth, gates, chains=None):
    """FLOPs_fwd(level) = D*2*G*E*H^2 (+ small terms); fwd+bwd = 3x.

    -> (full-depth flops per step, executed flops per step, atoms per batch).  "Executed" credits the tree-side levels
    only with the depth steps the fixed-point shortcut really issues (chain + 1 forward, chain backward steps)."""
    from ggpm_amd import synth
    full = execd = 0.0
    atoms = 0
    for i, (tree, graph) in enumerate(batches):
        st = synth.batch_stats(tree, graph)
        c = chains[i] if chains else 0
        for lvl, I, Fd in (("atom", 62, 38), ("tree", H + 20, H), ("tree", H + 20, H)):
            E, N = st[lvl]["E"], st[lvl]["N"]
            gate = 2.0 * gates * E * H * H
            small = 2.0 * gates * E * I * H + 2.0 * N * (Fd + H) * H
            full += 3.0 * (depth * gate + small)
            if lvl == "tree" and 0 < c:
                execd += (min(depth, c + 1) + 2.0 * min(depth, c)) * gate + 3.0 * small
            else:
                execd += 3.0 * (depth * gate + small)
        atoms += st["atom"]["N"]
    n = len(batches)
    return full / n, execd / n, atoms / n


from ggpm_amd.launcher import host_cores      # noqa: E402  (CPU share of this process: affinity capped by the cgroup quota)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def log(msg):
    print("[bench %.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def cpu_baseline(batches, rnn, H, depth, latent, n_motif, n_attach, budget_s=24.0, warm=2, timed=10):
    """Oracle (reference op order, PyTorch CPU) fwd+bwd on the first batches; BASELINE.md section 3: 2 warm-up + >= 10
    timed steps, median, with all host cores and with 8 threads.  Bounded: a thread setting stops early once it has
    used its share of `budget_s` (large configs), and says so in `sample`."""
    from oracle import ref_encoder as ref
    from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
    sd = seeded_state_dict(encoder_param_shapes(rnn, H, n_motif, n_attach), 0)
    sd.update(seeded_state_dict(vae_head_shapes(H, latent), 7))
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
    B = len(batches[0][0][-1])
    ts = [(ref.to_long_tensors(tree), ref.to_long_tensors(graph)) for tree, graph in batches]

    def run(threads, share):
        torch.set_num_threads(threads)
        times, t_begin = [], time.time()
        for i in range(warm + timed):
            tt, gt = ts[i % len(ts)]
            t0 = time.time()
            outs = ref.hier_encoder_forward(p, rnn, depth, depth, tt, gt)
            _, kl = ref.rsample_kl(p, outs[0])
            loss = 0.1 * kl + 1e-3 * sum(o.sum() for o in outs)
            for v in p.values():
                v.grad = None
            loss.backward()
            dt = time.time() - t0
            if i >= warm or (time.time() - t_begin > share and i >= 1):
                times.append(dt)
            if time.time() - t_begin > share and len(times) >= 2:
                break
        log("cpu baseline %s, %d threads: %d timed steps, median %.3f s" % (rnn, threads, len(times), np.median(times)))
        return float(np.median(times)), len(times)

    cores = host_cores()
    med_all, n_all = run(cores, 0.6 * budget_s)
    out = {"value": round(B / med_all, 2), "unit": "molecules/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model(),
           "sample": "oracle/ref_encoder.py (reference padded op order, torch CPU) fwd+bwd of batch %d: %d warm-up + %d "
                     "timed steps, median %.3f s, %d threads" % (B, warm, n_all, med_all, cores)}
    if cores != 8:
        med8, n8 = run(min(8, cores), 0.4 * budget_s)
        out["threads_8"] = {"value": round(B / med8, 2), "cores": min(8, cores), "timed_steps": n8,
                            "median_step_s": round(med8, 4)}
    torch.set_num_threads(cores)
    return out


def depth_kernel_bytes(kname, E, H, dbar, st16):
    """HBM bytes ONE launch of a depth kernel reads + writes on a level of E messages (algorithmic: every array once, the
    gathered rows dbar times; weights stream from L2 and are left out), at the stored element sizes: `s` = 2 B for the arrays
    ggpm_level_bf16_storage keeps in bf16 (state h / q, stashes S G Z M resp. S I O U, dS / dG, DQ / DZP / DMP resp. DQ / DI /
    DO / DU), 4 B for the rest (gate inputs X and their gradients, R / F, the LSTM cell state, ds_dir).  Per message row and
    hidden column (csrc/mpn_gru.hip, mpn_lstm.hip; the unfused forms, as two-row-tile levels launch them):
      gru_fwd_a   reads Xr Xz Xh (12), gathers h q (2 s dbar); writes h' (s), S G Z M (4 s), R (4)
      gru_fwd_b   reads h' (s); writes q' (s)
      gru_bwd_a   reads own h q (2 s), per successor Xr dS dG ((4 + 2 s) dbar), S Z M (3 s); writes DQ DZP DMP (3 s), ds_dir (4)
      gru_bwd_b   reads DZP DMP (2 s), ds_dir R (8); writes dS dG (2 s); dXr read-modify-write (8)
      lstm_fwd_a  reads Xf Xi Xo Xu (16), gathers h qf (2 s dbar) and c (4 dbar); writes h' (s), c' (4), S I O U (4 s), F (4)
      lstm_fwd_b  reads h' (s); writes qf' (s)
      lstm_bwd_a  reads own h qf (2 s), c (4), per successor Xf dS (4 + s) dbar and dFC (4 dbar), S I O U (4 s), c' (4);
                  writes DQ DI DO DU (4 s), dFC (4), ds_dir (4)
      lstm_bwd_b  reads DI DO DU (3 s), ds_dir F (8); writes dS (s); dXf read-modify-write (8)"""
    s = 2.0 if st16 else 4.0
    per_elem = {
        "gru_fwd_a": 12 + 2 * s * dbar + s + 4 * s + 4,
        "gru_fwd_b": 2 * s,
        "gru_bwd_a": 2 * s + (4 + 2 * s) * dbar + 3 * s + 3 * s + 4,
        "gru_bwd_b": 2 * s + 8 + 2 * s + 8,
        "lstm_fwd_a": 16 + (2 * s + 4) * dbar + s + 4 + 4 * s + 4,
        "lstm_fwd_b": 2 * s,
        "lstm_bwd_a": 2 * s + 4 + (4 + s + 4) * dbar + 4 * s + 4 + 4 * s + 8,
        "lstm_bwd_b": 3 * s + 8 + s + 8,
    }[kname]
    return per_elem * E * H


class _SumOutputs(torch.autograd.Function):
    """hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum() -- the harness term that sends a gradient into every encoder
    output -- without the harness costing ~20 tiny launches per step: the four outputs are row ranges of one buffer, so the
    forward is ONE reduction over it and the backward ONE fill (autograd's own form: four reductions, three adds, four
    expanded gradients that the encoder's backward then has to make contiguous)."""

    @staticmethod
    def forward(ctx, *outs):
        base = outs[0]._base
        whole = (base is not None and all(o._base is base for o in outs) and base.is_contiguous()
                 and sum(o.numel() for o in outs) == base.numel())
        ctx.shapes = [o.shape for o in outs]
        return base.sum() if whole else sum(o.sum() for o in outs)

    @staticmethod
    def backward(ctx, g):
        rows, width = sum(s[0] for s in ctx.shapes), ctx.shapes[0][1]
        buf = g.expand(rows, width).contiguous()
        return buf.split([s[0] for s in ctx.shapes])


def _sum_outputs(*outs):
    same_width = all(o.dim() == 2 and o.shape[1] == outs[0].shape[1] for o in outs)
    return _SumOutputs.apply(*outs) if same_width else sum(o.sum() for o in outs)


class Workload:
    """One (config, message function) pair on this rank: model, optimizer, device-resident batches."""

    def __init__(self, cfg, rnn, a, rank, world, dev, gate_dtype="f32"):
        from ggpm_amd.nnutils import make_cuda
        from ggpm_amd.parallel import FlatGradSync, broadcast_parameters
        from ggpm_amd.property_vae import HierEncoderVAE
        self.cfg, self.rnn, self.a, self.world, self.dev, self.gate_dtype = cfg, rnn, a, world, dev, gate_dtype
        n_motif, n_attach = cfg["vocab"]
        # rank r draws the batches r, r+W, ... of the seed-indexed synthetic set (10 000 molecules = 313 batches)
        self.pool = make_batches(a.pool, cfg["batch"], seed0=1000 + rank * 313, gen=cfg["gen"], n_motif=n_motif,
                                 n_attach=n_attach)
        if world > 1:
            # balance the ranks by message count, not molecule count -- every rank steps through ITS batches in
            # size order, so the batches the ranks work on at the same time are of similar size and the all-reduce waits
            # less for the straggler (one rank: order unchanged, the same batches either way)
            self.pool.sort(key=lambda tg: int(tg[1][1].shape[0]))
        self.dev_batches = [make_cuda(b) for b in self.pool]      # int64 index tensors resident in HBM before timing
        torch.manual_seed(0)
        self.model = HierEncoderVAE(make_args(rnn, cfg["hidden"], cfg["depth"], cfg["latent"], n_motif, n_attach)).to(dev)
        for p in self.model.parameters():                       # vae_train.py:48-53
            if p.dim() == 1:
                torch.nn.init.constant_(p, 0)
            else:
                torch.nn.init.xavier_normal_(p)
        self.model.encoder.gate_dtype = gate_dtype
        broadcast_parameters(self.model)
        # Adam (vae_train.py:60) on one flat view of the parameters; gradients in the flat buffer the all-reduce uses anyway
        from ggpm_amd.optim import FlatAdam
        self.sync = FlatGradSync(self.model.parameters(), encoder=self.model.encoder, keep_flat=True)
        self.opt = FlatAdam(self.sync, lr=1e-3)
        self.host_iter = None
        if a.host_input:
            import itertools
            from ggpm_amd.dataloader import DevicePrefetcher
            self.host_iter = iter(DevicePrefetcher(itertools.cycle(self.pool), device=dev, depth=2))

    # steps the host may be ahead of the GPU.  This loop reads nothing back, so without a bound the host (2.0 ms of enqueue
    # per 2.8 ms step) gets 8-10 steps ahead inside a 30-step region, every step in flight holds its own ~1 GB arena of saved
    # states, and whenever the lead reaches a new maximum the caching allocator goes to the device for one more (5 ms each;
    # tools/probe/pool_growth.py: 22 arenas = 21.6 GB reserved for 0.06 GB of live tensors, still growing at step 210).  A real
    # loop is bounded by its own metrics read (vae_train.py:86-94 reads them every step); here: step i waits for step i - 2.
    MAX_LEAD = 2

    def step(self, i):
        from ggpm_amd.property_vae import rsample
        m = self.model
        tree, graph = next(self.host_iter) if self.host_iter is not None else self.dev_batches[i % len(self.dev_batches)]
        evs = getattr(self, "_step_events", None)
        if evs is None:
            evs = self._step_events = [torch.cuda.Event() for _ in range(self.MAX_LEAD)]
            self._step_no = 0
        ev = evs[self._step_no % self.MAX_LEAD]
        self._step_no += 1
        t_w = time.perf_counter()
        ev.synchronize()                  # (the step MAX_LEAD back; returns at once while the host is not that far ahead)
        self._lead_wait = getattr(self, "_lead_wait", 0.0) + (time.perf_counter() - t_w)
        self.sync.zero_grad()
        hroot, hnode, hinter, hatom = m.encoder.forward_padded(tree, graph)
        _, kl = rsample(hroot, m.R_mean, m.R_var, perturb=False)
        loss = 0.1 * kl + 1e-3 * _sum_outputs(hroot, hnode, hinter, hatom)
        loss.backward()
        self.sync.all_reduce()
        self.opt.step()
        ev.record()
        return loss

    def _any_rank(self, flag: bool) -> bool:
        """`flag` on any rank (one rank: itself) -- for decisions that change how many steps, i.e. collectives, a rank runs."""
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist
        t = torch.tensor([1.0 if flag else 0.0], device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def fence(self):
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(self, steps, first, what="timed"):
        """EXACTLY `steps` steps between two fences; max over ranks.  -> (elapsed s, host enqueue s)

        A region during which PyTorch's caching allocator went to the device for memory (``num_device_alloc`` moved) is
        measured again, at most three times: such a call stalls ONE step by 5 ms (configs[1] sizes) to 300 ms (configs[4])
        however long the loop has run before (tools/step_jitter.py) -- a one-time cost of the process, not throughput of the
        step.  Nothing is hidden by that: ``self.region_log[what]`` keeps, for the REPORTED region, the reading of every
        attempt (the first one included) and the device allocations inside the attempt that was kept.  The same rule on every
        world size: the ranks decide together (`_any_rank`), since each attempt holds collectives."""
        import torch.distributed as dist
        readings = []
        for attempt in range(4):
            _settle_gc()
            self.fence()
            allocs0 = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0)
            self._lead_wait = 0.0
            t0 = time.perf_counter()
            for i in range(steps):
                self.step(first + i)
            host = time.perf_counter() - t0 - self._lead_wait      # (enqueue time proper: without the waits of MAX_LEAD)
            self.fence()
            elapsed = time.perf_counter() - t0
            grew = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0) - allocs0
            if self.world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            readings.append(round(1e3 * elapsed / steps, 4))
            if attempt == 3 or not self._any_rank(grew > 0):
                break
            self.regions_repeated = getattr(self, "regions_repeated", 0) + 1
            log("  (%d device allocation(s) inside the timed region, %.3f ms/step: measuring it again)" % (grew, 1e3 * elapsed / steps))
        if not hasattr(self, "region_log"):
            self.region_log = {}
        self.region_log[what] = {"attempts": len(readings), "ms_per_step_of_each_attempt": readings,
                                 "device_allocs_in_reported_attempt": int(grew)}
        return elapsed, host

    def measure(self, lib, rank):
        a, cfg = self.a, self.cfg
        H, depth, batch = cfg["hidden"], cfg["depth"], cfg["batch"]
        log("%s: model + %d batches resident; warm-up" % (self.rnn, len(self.dev_batches)))
        # W untimed steps, and at least one pass over the pool: every batch shape has then been seen by the caching allocator,
        # on each of the package's streams -- a shape met for the first time inside the timed region costs a device allocation
        # there (measured with tools/step_jitter.py: 5 ms at configs[1] sizes, 250-300 ms at configs[4] sizes, in ONE step)
        # ... and for at least a second of wall time: the first GPU process on a freshly acquired box ran its first timed region
        # 10-60 % slow (3.2 / 4.8 ms per step where every later process measured 2.9) -- clocks, page cache, lazily loaded code
        # Every rank runs the SAME number of warm-up steps -- each step holds a collective -- so what ends the warm-up (a
        # second of wall time, an allocation-free pass) is decided by all ranks together (`_any_rank`), in chunks of 8 steps.
        warm = max(a.warmup, len(self.dev_batches))
        t_warm, i = time.perf_counter(), 0
        while i < warm:
            self.step(i)
            if i == 0:
                torch.cuda.synchronize()
                log("first step done")
            i += 1
        torch.cuda.synchronize()
        while i < 4000 and self._any_rank(time.perf_counter() - t_warm < 1.0):
            for _k in range(8):
                self.step(i)
                i += 1
            torch.cuda.synchronize()              # (the host runs ahead of the GPU: count GPU time, not enqueue time)
        # ... and until one more pass over the pool (at least as many steps as the timed region has) goes by without a device
        # allocation, three extra passes at most: the allocator's pool has then stopped growing for these shapes
        extra = max(len(self.dev_batches), min(a.steps, 8))
        for _ in range(3):
            torch.cuda.synchronize()
            n0 = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0)
            for _k in range(extra):
                self.step(i)
                i += 1
            torch.cuda.synchronize()
            if not self._any_rank(torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0) != n0):
                break
        # more than one rank: ONE all-reduce of the flat gradient behind the backward, or the slice in front of the atom level's
        # parameters reduced on the second stream while that level's backward still runs (FlatGradSync.bucketed) -- which is
        # faster depends on the fabric and on what a collective costs beside the depth kernels, so both forms are timed here
        # (8 steps each, max over ranks: every rank sees the same two numbers and makes the same choice)
        self.allreduce_choice = None
        if self.world > 1 and self.sync.early_numel > 0 and "GGPM_BUCKETED_ALLREDUCE" not in os.environ:
            t_form = {}
            for form in (False, True):
                self.sync.bucketed = form
                for _k in range(4):
                    self.step(i)
                    i += 1
                t_form[form] = self.timed(8, i, "allreduce_trial")[0] / 8
                i += 8
            self.sync.bucketed = t_form[True] < t_form[False]
            self.allreduce_choice = {"form": "bucketed" if self.sync.bucketed else "single",
                                     "ms_per_step_single": round(1e3 * t_form[False], 4),
                                     "ms_per_step_bucketed": round(1e3 * t_form[True], 4)}
            log("all-reduce: %s" % self.allreduce_choice)
        warm = i
        self.warmup_run = warm
        log("warm-up done (%d steps); timing %d steps" % (warm, a.steps))
        elapsed, host_enqueue = self.timed(a.steps, warm, "value")
        log("timed region done: %.3f ms/step (host enqueue %.3f ms/step)" % (1e3 * elapsed / a.steps,
                                                                             1e3 * host_enqueue / a.steps))
        chains = [int(getattr(t[0][3], "ggpm_chain", 0)) for t in self.dev_batches]
        # The same K steps once more WITHOUT the fixed-point hint (make_cuda measures the longest dependency chain of
        # the tree messages; with it the two tree-side levels stop at their fixed point, forward and backward,
        # bit-identical outputs): every level runs all `depth` launches both ways.  Reported beside `value`.
        full_elapsed = None
        if self.host_iter is None and not a.no_full_depth and any(chains):
            hinted = self.dev_batches
            self.dev_batches = [(list(tree[:3]) + [tree[3].view_as(tree[3])] + list(tree[4:]), graph) for tree, graph in hinted]
            for i in range(max(min(a.warmup, 4), min(len(self.dev_batches), 4))):
                self.step(i)
            full_elapsed, _ = self.timed(a.steps, warm, "full_depth_loops")
            self.dev_batches = hinted
            log("without the tree fixed-point hint: %.3f ms/step" % (1e3 * full_elapsed / a.steps))
        gates = 3 if self.rnn == "GRU" else 4
        fl_full, fl_exec, atoms = algorithmic_work(self.pool, H, depth, gates, chains)
        mols = a.steps * batch * self.world
        res = {"ms_per_step": round(1e3 * elapsed / a.steps, 4), "value": round(mols / elapsed, 2),
               "unit": "molecules/s", "rnn_type": self.rnn, "warmup_steps_run": warm,
               "timed_regions_repeated": getattr(self, "regions_repeated", 0),
               "timed_region": self.region_log.get("value"),
               "host_enqueue_ms_per_step": round(1e3 * host_enqueue / a.steps, 4),
               "host_lead_bound_steps": self.MAX_LEAD,
               "allreduce": getattr(self, "allreduce_choice", None),
               "algorithmic_gflop_per_step_per_gpu": round(fl_full / 1e9, 2),
               "executed_gflop_per_step_per_gpu": round(fl_exec / 1e9, 2),
               # executed flops over the measured time: the tree-side levels are credited only with the depth steps
               # the fixed-point shortcut really issues
               "step_tflops_executed": round(fl_exec * self.world / (elapsed / a.steps) / 1e12, 3),
               "atoms_per_molecule": round(atoms / batch, 1)}
        if full_elapsed is not None:
            res["tree_fixed_point"] = (
                "motif-tree messages settle after their longest dependency chain C (%d-%d in these batches): the two "
                "tree-side levels run C+1 of the %d forward and C of the %d backward steps; outputs bit-identical to the "
                "full loops, gradients to fp32 summation order (tests/test_gpu_parity.py::"
                "test_tree_fixed_point_shortcut_is_bit_identical)" % (min(chains), max(chains), depth, depth))
            res["full_depth_loops"] = {"ms_per_step": round(1e3 * full_elapsed / a.steps, 4),
                                       "value": round(mols / full_elapsed, 2), "unit": "molecules/s",
                                       "timed_region": self.region_log.get("full_depth_loops"),
                                       "step_tflops_algorithmic": round(fl_full * self.world / (full_elapsed / a.steps) / 1e12, 3)}
        if not a.no_roofline:
            roof = self.roofline(lib)
            if roof and rank == 0:
                res["roofline"] = roof
        return res

    def roofline(self, lib):
        """Second, instrumented pass over the same steps: HIP events recorded on the launch stream around every fused
        depth-step launch (not part of `value`), per kernel kind and per level."""
        from ggpm_amd.property_vae import rsample
        a, m = self.a, self.model

        def eager_step(i):
            tree, graph = self.dev_batches[i % len(self.dev_batches)]
            for p in m.parameters():
                p.grad = None
            hroot, hnode, hinter, hatom = m.encoder.forward_padded(tree, graph)
            _, kl = rsample(hroot, m.R_mean, m.R_var, perturb=False)
            (0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())).backward()

        eager_step(0)
        torch.cuda.synchronize()
        lib.ggpm_timing_enable(1)
        for i in range(min(a.steps, 4)):
            eager_step(i)
        torch.cuda.synchronize()
        lib.ggpm_timing_enable(0)
        per = {}
        peak_tf = PEAK_MFMA_BF16_TFLOPS if self.gate_dtype == "bf16" else PEAK_MFMA_F32_TFLOPS      # (per-kernel "frac")
        for tag, lname in LEVEL_TAGS.items():
            for which, kname in enumerate(KERNELS):
                n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
                lib.ggpm_timing_collect(which + 8 * tag, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
                if n.value:
                    per.setdefault(lname, {})[kname] = {
                        "launches": n.value, "avg_launch_us": round(1e3 * ms.value / n.value, 3),
                        "gflop_per_launch": round(fl.value / n.value / 1e9, 4),
                        "tflops": round(fl.value / (ms.value * 1e-3) / 1e12, 3),
                        "frac": round(fl.value / (ms.value * 1e-3) / 1e12 / peak_tf, 4),
                        "total_ms": round(ms.value, 3)}
        if "atom" not in per:
            return None
        kname = max(per["atom"], key=lambda k: per["atom"][k]["total_ms"])      # dominant kernel of the step
        k = per["atom"][kname]
        traffic = None
        try:      # HBM bytes per ATOM-LEVEL launch from the committed rocprofv3 --pmc passes of this same command
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tj = json.load(f)
            same = lambda n: all(self.cfg[k] == CONFIGS[n][k] for k in ("hidden", "depth", "batch", "gen"))
            if same(1) and self.gate_dtype != "bf16":
                traffic = tj.get(self.rnn + "_atom_level", {}).get(kname, {}).get("bytes_per_launch")
            elif same(4) and self.rnn == "GRU":
                key = "configs4_%s_atom_level" % ("bf16" if self.gate_dtype == "bf16" else "f32")
                traffic = tj.get(key, {}).get(kname, {}).get("bytes_per_launch")
        except Exception:
            pass
        tree = {lv: per[lv][kname] for lv in ("attachment", "motif") if kname in per.get(lv, {})}
        rocprof = None
        try:      # the committed rocprofv3 --kernel-trace summary of the same command: its mean duration of this kernel class
            import glob
            same1 = all(self.cfg[q] == CONFIGS[1][q] for q in ("hidden", "depth", "batch", "gen"))
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_kernel_stats.txt" % self.rnn.lower())))
            if same1 and self.gate_dtype != "bf16" and files:
                for line in open(files[-1]):
                    if line.startswith(kname + "[atom level]"):
                        us = float(line.split()[3])
                        # (an event bracket reads ~2 us longer than the kernel itself: the trace's figure is the kernel's)
                        rocprof = {"avg_launch_us": us, "frac": round(k["gflop_per_launch"] * 1e9 / (us * 1e-6) / 1e12 / peak_tf, 4),
                                   "source": os.path.relpath(files[-1], ROOT)}
                        break
        except Exception:
            pass
        out = {"kernel": kname, "level": "atom (one 16-wave workgroup per 16 messages, all gate columns)",
               "bound": "mfma", "achieved": k["tflops"], "peak": peak_tf, "unit": "TFLOP/s",
               "frac": k["frac"], "traffic": traffic, "launches": k["launches"], "avg_launch_us": k["avg_launch_us"],
               "flops_per_launch_avg": round(k["gflop_per_launch"] * 1e9, 1), "rocprof_trace": rocprof, "tree_levels": tree,
               "all_depth_kernels": per}
        if self.gate_dtype == "bf16":
            # The bf16 leg is priced against HBM (bf16 gate products leave the matrix pipe 16x faster than fp32).
            # Bytes of a launch = the arrays THAT kernel reads and writes, at the element size they are stored in
            # (`depth_kernel_bytes`: 2 B where ggpm_level_bf16_storage keeps a depth-loop array in bf16, 4 B otherwise); the
            # A + B pair of a depth step stands beside the dominant kernel.  Weights come from L2 and are not counted.
            from ggpm_amd import synth
            st = [synth.batch_stats(t, g)["atom"] for t, g in self.pool]
            E, dbar = float(np.mean([x["E"] for x in st])), float(np.mean([x["dbar"] for x in st]))
            H = self.cfg["hidden"]
            st16 = bool(lib.ggpm_level_bf16_storage(int(E) + 1, H))
            nbytes = depth_kernel_bytes(kname, E, H, dbar, st16)
            gbs = nbytes / (k["avg_launch_us"] * 1e-6) / 1e9
            pair = {}
            for direction in ("fwd", "bwd"):
                names = [n for n in per["atom"] if direction in n]
                if names:
                    us = sum(per["atom"][n]["avg_launch_us"] for n in names)
                    by = sum(depth_kernel_bytes(n, E, H, dbar, st16) for n in names)
                    pair[direction] = {"kernels": names, "us_per_depth_step": round(us, 2), "bytes_per_depth_step": round(by, 1),
                                       "achieved": round(by / (us * 1e-6) / 1e9, 1), "unit": "GB/s",
                                       "frac": round(by / (us * 1e-6) / 1e9 / PEAK_HBM_GBS, 4)}
            out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_launch": round(nbytes, 1),
                        "stored_element_bytes": 2 if st16 else 4, "depth_step_pairs": pair,
                        "regime": "latency / phase bound: gather, gate products and epilogue follow each other inside one "
                                  "workgroup per CU, so neither HBM (this figure) nor the bf16 matrix pipe (`mfma`) binds "
                                  "(DESIGN 12.5)",
                        "mfma": {"achieved": k["tflops"], "peak": PEAK_MFMA_BF16_TFLOPS, "unit": "TFLOP/s",
                                 "frac": round(k["tflops"] / PEAK_MFMA_BF16_TFLOPS, 4),
                                 "note": "the gate products' flops against the dense bf16 MFMA peak"}})
        return out


class VaeWorkload:
    """The "reported row" of SURVEY.md section 8(d): the FULL VAE training step -- HierPropertyVAE.forward (encoder,
    rsample with perturbation, teacher-forced HierMPNDecoder with diterT = 1 / diterG = 5 as every shipped config has
    them, the four losses) + backward (+ gradient all-reduce for N > 1) + Adam -- on the configs[1] workload.

    Two ways of feeding it, both reported:
      * ``ms_per_step``            the batches' index tensors AND decode schedules resident on the device, like the index
                                   tensors of the encoder row (``schedule=`` passed in);
      * ``schedule_in_loop``       ``model(*batch, beta=beta)`` exactly as vae_train.py:78 calls it: the batch arrives as
                                   host arrays + the networkx graphs, ``make_cuda`` and the decode schedule
                                   (DecodeSchedule.from_graphs -> csrc/schedule.hip, two uploads) happen INSIDE the step;
      * ``schedule_ahead``         that loop with its iterator wrapped, ``for batch in ScheduleAhead(dataset, model)``: the
                                   schedule of batch k+1 is built on a worker thread while step k runs.
    """

    DITER_T, DITER_G, TIE = 1, 5, False

    def __init__(self, cfg, rnn, a, dev, rank=0, world=1):
        from ggpm_amd import synth
        from ggpm_amd.decoder import DecodeSchedule
        from ggpm_amd.nnutils import make_cuda
        from ggpm_amd.parallel import FlatGradSync, broadcast_parameters
        from ggpm_amd.property_vae import HierPropertyVAE
        from ggpm_amd.vocab import IndexPairVocab
        self.cfg, self.rnn, self.a, self.dev, self.rank, self.world = cfg, rnn, a, dev, rank, world
        n_motif, n_attach = cfg["vocab"]
        self.vocab = IndexPairVocab(n_motif, n_attach)
        args = make_args(rnn, cfg["hidden"], cfg["depth"], cfg["latent"], n_motif, n_attach)
        args.vocab, args.diterT, args.diterG, args.tie_embedding = self.vocab, self.DITER_T, self.DITER_G, self.TIE
        torch.manual_seed(0)
        self.model = HierPropertyVAE(args).to(dev)
        for p in self.model.parameters():
            if p.dim() == 1:
                torch.nn.init.constant_(p, 0)
            else:
                torch.nn.init.xavier_normal_(p)
        hints = self.model.decoder.schedule_hints()
        self.items = []
        for i in range(min(a.pool, 8)):
            specs = synth.random_batch(1000 + rank * 313 + i, cfg["batch"], motifs=tuple(cfg["gen"]), n_motif_vocab=n_motif,
                                       n_attach_vocab=n_attach)
            tensors = synth.tensorize(specs)
            batch6 = synth.train_batch(specs, tensors)            # what a DataFolder batch holds (vae_train.py:75-78)
            sch = DecodeSchedule.from_graphs(batch6[1], tensors, batch6[3], self.vocab, **hints)
            self.items.append((tensors, make_cuda(tensors), sch.to_device(dev), batch6))
        broadcast_parameters(self.model)
        # Adam (vae_train.py:60) on one flat view of the parameters, gradients in the flat buffer the all-reduce uses anyway
        # and the encoder's backward writes into directly
        from ggpm_amd.optim import FlatAdam
        self.sync = FlatGradSync(self.model.parameters(), encoder=self.model.encoder, keep_flat=True)
        self.opt = FlatAdam(self.sync, lr=1e-3)
        self.orders = [None] * cfg["batch"]

    def _finish(self, loss, metrics):
        from ggpm_amd.functional import mark          # (no-ops unless tools/vae_phase_times.py switched them on)
        mark("bwd: starts")
        loss.backward()
        mark("bwd: engine returns")
        if self.sync is not None:
            self.sync.all_reduce()
        mark("gradients packed")
        self.opt.step()
        mark("optimizer issued")
        self._issued_at = time.perf_counter()      # everything of this step has been handed to the GPU (or to the worker)
        metrics["Loss"]      # the training loop reads the metrics here (vae_train.py:86): one host read-back per step
        return metrics

    def _zero(self):
        if self.sync is not None:
            self.sync.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=True)

    def step(self, i):
        _, dev_tensors, sch, _ = self.items[i % len(self.items)]
        self._zero()
        loss, metrics = self.model(None, None, dev_tensors, self.orders, None, None, beta=0.1, perturb_z=True, schedule=sch)
        return self._finish(loss, metrics)

    def step_in_loop(self, i):
        batch6 = self.items[i % len(self.items)][3]
        self._zero()
        loss, metrics = self.model(*batch6, beta=0.1)             # vae_train.py:78, unchanged
        return self._finish(loss, metrics)

    def step_ahead(self, i):
        """the loop with its one changed line, ``for batch in ScheduleAhead(dataset, model)``: batch i+1's schedule is
        built on a worker thread while this step runs"""
        if getattr(self, "_ahead", None) is None:
            from ggpm_amd.dataloader import ScheduleAhead

            def forever():
                k = i
                while True:
                    yield self.items[k % len(self.items)][3]
                    k += 1
            self._ahead = iter(ScheduleAhead(forever(), self.model))
        batch6 = next(self._ahead)
        self._zero()
        loss, metrics = self.model(*batch6, beta=0.1)
        return self._finish(loss, metrics)

    def _fence(self):
        torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def _any_rank(self, flag: bool) -> bool:
        if self.world <= 1:
            return bool(flag)
        import torch.distributed as dist
        t = torch.tensor([1.0 if flag else 0.0], device=self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def _timed(self, fn, steps, first, what="resident"):
        """EXACTLY `steps` calls between two fences, max over ranks -> (ms per step, per-step host marks).  A region with a
        device allocation inside is measured again (Workload.timed: same rule, same record in ``self.region_log[what]``)."""
        readings = []
        for attempt in range(4):
            _settle_gc()
            self._fence()
            allocs0 = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0)
            self._gc_events = []
            if os.environ.get("GGPM_BENCH_TRACE_STEPS"):
                import gc
                gc.callbacks[:] = []
                st = {}

                def _cb(phase, info, st=st, ev=self._gc_events):
                    if phase == "start":
                        st["t"] = time.perf_counter()
                    else:
                        ev.append((info["generation"], round(1e3 * (time.perf_counter() - st["t"]), 2), info["collected"], st["t"]))
                gc.callbacks.append(_cb)
            t0 = time.perf_counter()
            marks, issue = [t0], []
            for i in range(steps):
                fn(first + i)                     # (ends with the metrics' host read: the host is in step with the GPU)
                issue.append(1e3 * (self._issued_at - marks[-1]))
                marks.append(time.perf_counter())
            self.host_issue_ms = float(np.median(issue)) if issue else None      # host time to ISSUE one step (median)
            self._fence()
            dt = time.perf_counter() - t0
            grew = torch.cuda.memory_stats(self.dev).get("num_device_alloc", 0) - allocs0
            if self.world > 1:
                import torch.distributed as dist
                t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            readings.append(round(1e3 * dt / steps, 4))
            if attempt == 3 or not self._any_rank(grew > 0):
                break
            self.regions_repeated = getattr(self, "regions_repeated", 0) + 1
            log("  (%d device allocation(s) inside the timed region, %.3f ms/step: measuring it again)" % (grew, 1e3 * dt / steps))
        if not hasattr(self, "region_log"):
            self.region_log = {}
        self.region_log[what] = {"attempts": len(readings), "ms_per_step_of_each_attempt": readings,
                                 "device_allocs_in_reported_attempt": int(grew)}
        self._step_trace = [(1e3 * (b - a), [e for e in self._gc_events if a <= e[3] < b]) for a, b in zip(marks, marks[1:])]
        return 1e3 * dt / steps, [1e3 * (b - a) for a, b in zip(marks, marks[1:])]

    def work(self):
        """Executed / algorithmic flops per step (fwd + bwd = 3 x fwd), from the batches' schedules.

        Per (row, depth step) of a message function: G gate products of 2 H^2 flops.  Encoder: the three levels as in
        `algorithmic_work` (executed = with the tree fixed point).  Decoder atom level: every decode step runs its compact
        row set (its bonds + the frozen older bonds they read) through diterG depth steps -- `executed` counts all rows of
        the set, `algorithmic` the rows that are really recomputed.  Decoder tree-side levels: all messages, `chain`
        synchronous steps each (DecodeSchedule._level_plan).  Heads: topoNN / clsNN / iclsNN / matchNN / W_assm and the
        read-outs as dense products over their rows."""
        from ggpm_amd import synth
        cfg = self.cfg
        H, L, depth = cfg["hidden"], cfg["latent"], cfg["depth"]
        G = 3 if self.rnn == "GRU" else 4
        n_motif, n_attach = cfg["vocab"]
        chains = [int(getattr(it[1][0][3], "ggpm_chain", 0)) for it in self.items]
        enc_full, enc_exec, _ = algorithmic_work([it[0] for it in self.items], H, depth, G, chains)
        ex = al = 0.0
        for tensors, _, sch, _ in self.items:
            P = sch.plan
            plan = sch.atom_plan(tensors[1][0].shape[0], tensors[1][1].shape[0])
            gate = 2.0 * G * H * H
            rows_exec = float(sum(plan.nloc))
            rows_live = float(P["bond_off"][-1])
            E1g, E1t = tensors[1][1].shape[0] - 1, P["E1"] - 1
            x_atom = 2.0 * G * E1g * 62 * H                          # hoisted gate inputs of all bonds
            ex += self.DITER_G * rows_exec * gate + x_atom
            al += self.DITER_G * rows_live * gate + x_atom
            tree = 2 * (max(P["chain"], 1) * E1t * gate + 2.0 * G * E1t * (H + 20) * H)
            n_inst, ns_tot = P["n_inst"], P["atom_off"][-1]
            readouts = ns_tot * 2.0 * (38 + H) * H + n_inst * (2 * 2.0 * 2 * H * H + 2 * 2.0 * 2 * H * H)
            n_topo, n_cls = len(sch.topo()[0]), len(sch.cls()[0])
            heads = (n_topo * 2.0 * (H + L) * H + n_cls * (2 * 2.0 * (H + L) * H + 2.0 * H * (n_motif + n_attach))
                     + plan.n_cand * 2.0 * (H + H + 20) * H + len(sch.assm_batch()) * sch.max_cls_size * 2.0 * H * L)
            ex += tree + readouts + heads
            al += tree + readouts + heads
        n = len(self.items)
        return enc_exec + 3.0 * ex / n, enc_full + 3.0 * al / n

    def measure(self):
        B, steps = self.cfg["batch"], min(self.a.steps, 30)
        warm = 2 * len(self.items)        # two passes over the pool: every batch shape seen, clocks up
        for i in range(warm):
            m = self.step(i)
        ms, raw = self._timed(self.step, steps, warm, "resident")
        host_issue = self.host_issue_ms
        per = sorted(raw)
        log("full VAE step (%s): %.2f ms/step (per step: min %.2f, median %.2f, max %.2f at step %d; reserved %.1f GB), loss %.3f"
            % (self.rnn, ms, per[0], per[len(per) // 2], per[-1], raw.index(per[-1]), torch.cuda.memory_reserved() / 1e9,
               m["Loss"]))
        if getattr(self.a, "vae_profile", None) == "resident":      # (rocprofv3 runs: the trace ends with the resident steps)
            return {"ms_per_step": round(ms, 3), "steps": steps, "rnn_type": self.rnn}
        if getattr(self.a, "vae_profile", None) == "gc":          # what cyclic garbage does one step leave?
            from tools.gc_cycles import cycles_of
            for label, fn in (("resident", self.step), ("in_loop", self.step_in_loop)):
                k = [0]

                def one(fn=fn, k=k):
                    fn(k[0])
                    k[0] += 1
                log("cyclic garbage of 4 %s steps: " % label + "\n".join(cycles_of(one, repeat=4)))
            return {"ms_per_step": round(ms, 3), "steps": steps, "rnn_type": self.rnn}
        if getattr(self.a, "vae_profile", None) == "in_loop":
            for i in range(len(self.items)):
                self.step_in_loop(i)
            loop_ms, _ = self._timed(self.step_in_loop, steps, 0, "in_loop")
            return {"ms_per_step": round(ms, 3), "schedule_in_loop_ms": round(loop_ms, 3), "steps": steps, "rnn_type": self.rnn}
        # the same steps with the index structures derived from the resident decode tables (CSRs, transposes, frozen masks)
        # rebuilt on the device every step, as a stream of never-seen batches would have it
        from ggpm_amd import functional as F_
        fresh = None
        try:
            F_._MEMO_ON = False
            for i in range(len(self.items)):
                self.step(i)
            fresh, _ = self._timed(self.step, min(steps, 20), 0, "rebuilt")
        finally:
            F_._MEMO_ON = None
        log("  ... %.2f ms/step with the index structures rebuilt every step" % fresh)
        # vae_train.py:78 unchanged: host batch in, make_cuda + decode schedule (C++ builder, two uploads) inside the step
        for i in range(len(self.items)):
            self.step_in_loop(i)
        loop_ms, loop_raw = self._timed(self.step_in_loop, min(steps, 20), 0, "in_loop")
        loop_issue = self.host_issue_ms
        loop_trace = self._step_trace
        # ... and with the loop's iterator wrapped (dataloader.ScheduleAhead): the schedule of batch k+1 built during step k
        self._ahead = None
        for i in range(len(self.items)):
            self.step_ahead(i)
        ahead_ms, _ = self._timed(self.step_ahead, min(steps, 20), 0, "ahead")
        self._ahead.close()
        self._ahead = None
        t0 = time.perf_counter()
        from ggpm_amd.decoder import DecodeSchedule
        hints = self.model.decoder.schedule_hints()
        for k in range(16):
            b6 = self.items[k % len(self.items)][3]
            DecodeSchedule.from_graphs(b6[1], b6[2], b6[3], self.vocab, **hints)
        build_ms = 1e3 * (time.perf_counter() - t0) / 16
        lr = sorted(loop_raw)
        if os.environ.get("GGPM_BENCH_TRACE_STEPS"):
            log("  in-loop per step (ms; of which gc ms @ highest generation): " + " ".join(
                "%.2f(%.2f@%s)" % (x, sum(e[1] for e in evs), max([e[0] for e in evs], default="-")) for x, evs in loop_trace))
        log("  ... %.2f ms/step as vae_train.py calls it (host batch + networkx graphs in; schedule build %.2f ms of host time; per "
            "step: min %.2f, median %.2f, max %.2f); %.2f ms/step with the loop's iterator wrapped in ScheduleAhead"
            % (loop_ms, build_ms, lr[0], lr[len(lr) // 2], lr[-1], ahead_ms))
        fl_exec, fl_alg = self.work()
        tf = fl_exec * self.world / (ms * 1e-3) / 1e12
        out = {"ms_per_step": round(ms, 3), "value": round(B * self.world / (ms * 1e-3), 2), "unit": "molecules/s",
               "host_issue_ms": round(host_issue, 3),      # median host time from the start of a step to its last launch
               "timed_regions_repeated": getattr(self, "regions_repeated", 0),
               "timed_region": self.region_log.get("resident"),
               "ms_per_step_index_structures_rebuilt": round(fresh, 3),
               "schedule_in_loop": {"ms_per_step": round(loop_ms, 3), "timed_region": self.region_log.get("in_loop"), "value": round(B * self.world / (loop_ms * 1e-3), 2),
                                    "ratio_to_resident": round(loop_ms / ms, 3),
                                    "host_issue_ms": round(loop_issue, 3),
                                    "host_schedule_build_ms": round(build_ms, 3),
                                    "what": "model(*batch, beta=beta) as vae_train.py:78: numpy tensors + networkx graphs in, "
                                            "make_cuda and DecodeSchedule.from_graphs (csrc/schedule.hip) inside the step"},
               "schedule_ahead": {"ms_per_step": round(ahead_ms, 3), "timed_region": self.region_log.get("ahead"), "value": round(B * self.world / (ahead_ms * 1e-3), 2),
                                  "ratio_to_resident": round(ahead_ms / ms, 3),
                                  "what": "the same loop with ONE changed line, `for batch in ScheduleAhead(dataset, model)` "
                                          "(ggpm_amd/dataloader.py): batch k+1's schedule is built on a worker thread during "
                                          "step k; make_cuda and the two table uploads stay in the step"},
               "steps": steps, "warmup": warm, "rnn_type": self.rnn, "n_gpus": self.world,
               "roofline": {"bound": "mfma", "achieved": round(tf, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / PEAK_MFMA_F32_TFLOPS, 4), "traffic": self._counter_traffic(),
                            "executed_gflop_per_step_per_gpu": round(fl_exec / 1e9, 2),
                            "algorithmic_gflop_per_step_per_gpu": round(fl_alg / 1e9, 2),
                            "note": "whole-step figure: executed flops (every row the kernels process; fwd + bwd = 3 x fwd) over "
                                    "the step time; the step is a chain of small dependent launches (profiles/r05_vae_*), "
                                    "not one kernel"},
               "workload": "HierPropertyVAE fwd (perturb_z) + bwd%s + Adam on the configs[1] batches: latent=%d, diterT=%d, "
                           "diterG=%d, tie_embedding=%s, metrics read back on the host every step (after optimizer.step(), where "
                           "vae_train.py uses them); ms_per_step: index tensors AND decode schedules resident on the device, "
                           "derived index structures kept across steps (ms_per_step_index_structures_rebuilt: rebuilt every "
                           "step); schedule_in_loop: nothing resident"
                           % (" + all-reduce" if self.world > 1 else "", self.cfg["latent"], self.DITER_T, self.DITER_G, self.TIE)}
        try:        # launches per step and per-kernel-class time from the committed rocprofv3 trace of `bench.py --only-vae`
            import glob
            with open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_vae_launches.json")))[-1]) as f:      # (the latest round's)
                lj = json.load(f)
            out["launches_per_step"] = lj.get(self.rnn, {}).get("launches_per_step")
            out["kernel_classes"] = lj.get(self.rnn, {}).get("classes")
            out["launches_source"] = lj.get("source")
        except Exception:
            pass
        return out

    def _counter_traffic(self):
        """Bytes per STEP at the L2 <-> fabric boundary from the committed FETCH_SIZE / WRITE_SIZE passes of `bench.py --only-vae`
        (profiles/pmc_traffic.json, tools/pmc_step_total.py; Infinity-Cache hits included).  Roughly half of it is the packed gate
        weights: each of the step's ~700 small depth launches pulls them into eight L2s again (8 x 1.1 MB)."""
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                return json.load(f).get("vae_step_" + self.rnn, {}).get("bytes_per_step")
        except Exception:
            return None

    def cpu_baseline(self, budget_s=14.0):
        """The oracle's full VAE step (oracle/ref_decoder.py, reference op order) on the same batches."""
        from oracle import ref_encoder as ref, ref_decoder as refd
        cores = host_cores()
        torch.set_num_threads(cores)
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in self.model.state_dict().items()
             if not k.startswith(("decoder.rnn_cell.", "decoder.E_assm."))}
        cfg, times, t_begin = self.cfg, [], time.time()
        for i in range(2 + 6):
            tensors, _, sch, _ = self.items[i % len(self.items)]
            tt, gt = ref.to_long_tensors(tensors[0]), ref.to_long_tensors(tensors[1])
            t0 = time.time()
            loss, kl, _, _ = refd.vae_forward(p, self.rnn, cfg["depth"], cfg["depth"], self.DITER_T, self.DITER_G, tt, gt,
                                              sch, self.vocab.mask, 0.1)
            for v in p.values():
                v.grad = None
            loss.backward()
            dt = time.time() - t0
            if i >= 2 or time.time() - t_begin > budget_s:
                times.append(dt)
            if time.time() - t_begin > budget_s and times:
                break
        med = float(np.median(times))
        return {"value": round(cfg["batch"] / med, 2), "unit": "molecules/s", "cores": cores, "kind": "port",
                "cpu_model": cpu_model(),
                "sample": "oracle/ref_decoder.py vae_forward + backward of batch %d: 2 warm-up + %d timed steps, median "
                          "%.3f s, %d threads" % (cfg["batch"], len(times), med, cores)}


def configs4_leg(a, lib, dev, budget_s=90.0):
    """BASELINE configs[4] as a leg of the default line: {"fp32": ..., "bf16": ...}, <= 6 timed steps each."""
    import copy
    import gc
    if time.time() - _T0 > budget_s:
        return {"skipped": "the run was %.0f s old when this leg came up (budget %.0f s); `bench.py --config 4` measures it"
                           % (time.time() - _T0, budget_s)}
    cfg = dict(CONFIGS[4])
    b = copy.copy(a)
    b.pool, b.steps, b.warmup, b.no_full_depth, b.host_input = 2, min(a.steps, 6), 8, True, False
    out = {"workload": "BASELINE %s: hidden=%d depth=%d batch=%d, %s cell, %d timed steps on a pool of %d batches"
                       % (cfg["tag"], cfg["hidden"], cfg["depth"], cfg["batch"], cfg["rnn"], b.steps, b.pool)}
    try:
        for key, dt in (("fp32", "f32"), ("bf16", "bf16")):
            gc.unfreeze()
            gc.collect()
            # these batches are 50 x the size of the rows before: hand the cached blocks of those rows back first, or every
            # large request is carved out of whatever block fits and the pool keeps growing inside the timed regions (one
            # device allocation per 6-step region, 250-300 ms each: 55-66 ms/step where a fresh process measures 40.5)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            wl = Workload(cfg, cfg["rnn"], b, 0, 1, dev, gate_dtype=dt)
            m = wl.measure(lib, 0)
            leg = {k: m[k] for k in ("ms_per_step", "value", "unit", "step_tflops_executed", "atoms_per_molecule",
                                     "warmup_steps_run", "timed_regions_repeated", "timed_region") if k in m}
            r = m.get("roofline")
            if r:
                leg["roofline"] = {k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                     "algorithmic_bytes_per_launch", "stored_element_bytes", "depth_step_pairs",
                                                     "regime", "mfma") if k in r}
                leg["roofline"]["all_depth_kernels"] = {"atom": r.get("all_depth_kernels", {}).get("atom")}
            out[key] = leg
            del wl
    except Exception as exc:          # (a leg is a reported number, never a reason to lose the line)
        out["error"] = repr(exc)
    return out


_REAL_STDOUT = None


def _settle_gc():
    """Before a timed region: collect what the warm-up left behind and move the survivors (model, batches, the decode
    schedules' thousands of small index tables) to the permanent generation, so that a full collection falling into the
    timed steps does not walk them again (one such pass cost 90 ms of a 30-step VAE measurement)."""
    import gc
    gc.collect()
    gc.freeze()


def _claim_stdout():
    """The contract is ONE JSON line on stdout; libraries write there too (RCCL prints a version banner when the first
    communicator comes up).  From here on fd 1 points at stderr and the result line goes to the saved descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=16,
                    help="untimed steps; 16 = one pass over the pool of pre-tensorized batches, so that every batch shape has "
                         "been seen (allocator blocks, lazily created streams and events) before the timed region")
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS), help="index into BASELINE.json configs[]")
    ap.add_argument("--rnn", default=None, choices=["GRU", "LSTM"], help="message function (default: the config's)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32_mfma", "f32_split"],
                    help="gate products of the depth loops: fp32 MFMA (parity contract) or bf16 operands with fp32 "
                         "accumulate (configs[4]; everything else stays fp32).  --config 4 reports both.")
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--latent", type=int, default=None)
    ap.add_argument("--pool", type=int, default=16, help="distinct pre-tensorized batches per rank (cycled)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--host-input", action="store_true",
                    help="diagnostic: every step takes its batch from HOST memory through ggpm_amd.dataloader."
                         "DevicePrefetcher (pinned staging + async copy); the PCIe-inclusive rate, never `value`")
    ap.add_argument("--no-full-depth", action="store_true",
                    help="skip the extra timed pass without the tree fixed-point hint (profiling runs)")
    ap.add_argument("--no-second-cell", action="store_true",
                    help="skip the run of the other message function (configs[1] reports GRU and, under \"lstm\", LSTM)")
    ap.add_argument("--no-vae", action="store_true", help="skip the full-VAE-step row (\"vae_step\", configs[1], N = 1)")
    ap.add_argument("--only-vae", action="store_true", help="profiling: run ONLY the full-VAE-step row and print it")
    ap.add_argument("--vae-profile", default=None, choices=["resident", "in_loop", "gc"],
                    help="with --only-vae under rocprofv3: stop after the resident steps / after the vae_train.py-shaped steps, so "
                         "that the trace ends with the steps to be cut out (tools/prof_summary.py --steps)")
    ap.add_argument("--no-configs4", action="store_true", help="skip the configs[4] leg of the default line (\"configs4\": fp32 + bf16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args(argv)


def launch_ranks(a, argv):
    """``python bench.py --gpus N`` with nobody having set up the ranks: this process -- which has made no HIP call and
    makes none, torch.cuda included -- starts N fresh copies of itself, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set (ggpm_amd/launcher.py), relays rank 0's JSON line and exits non-zero if any rank does."""
    from ggpm_amd import launcher
    have = launcher.visible_gpu_count()              # the driver's topology files (-1: unreadable -- rank 0 will say)
    if 0 <= have < a.gpus and not os.environ.get("GGPM_BENCH_ONE_DEVICE"):
        raise SystemExit("--gpus %d but this node shows %d GPU(s) (rehearsal on one GPU: GGPM_BENCH_ONE_DEVICE=1 "
                         "--backend gloo)" % (a.gpus, have))
    log("starting %d rank processes (backend %s)" % (a.gpus, a.backend))
    return launcher.run_ranks(os.path.abspath(__file__), argv, a.gpus,
                              timeout=float(os.environ.get("GGPM_LAUNCH_TIMEOUT", "1500")))


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, sys.argv[1:]))
    _claim_stdout()
    cfg = dict(CONFIGS[a.config])
    for k in ("hidden", "depth", "batch", "latent"):
        if getattr(a, k) is not None:
            cfg[k] = getattr(a, k)
    rnn = a.rnn or cfg["rnn"]
    if a.config == 4:
        a.pool = min(a.pool, 4)          # ~15 K atom messages per batch: keep tensorizing (host, untimed) short

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=1" % a.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if os.environ.get("GGPM_BENCH_ONE_DEVICE"):      # rehearsal of N ranks on a 1-GPU box (use --backend gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    if world > 1 or os.environ.get("GGPM_FORCE_ALLREDUCE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    from ggpm_amd import _lib
    lib = _lib.load(build_if_missing=False)

    if a.only_vae:
        vae = VaeWorkload(cfg, rnn, a, dev, rank, world)
        res = vae.measure()
        if rank == 0:
            _emit({"vae_step": res})
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    main_wl = Workload(cfg, rnn, a, rank, world, dev, gate_dtype=a.dtype)
    m = main_wl.measure(lib, rank)
    n_motif, n_attach = cfg["vocab"]
    result = {
        "metric": "HierMPNEncoder fwd+bwd target row (SURVEY 8(d): encoder + KL heads + optimizer, tree fixed-point hint on; "
                  "full depth loops in full_depth_loops, the full VAE fwd+bwd row in vae_step) of BASELINE's "
                  "molecules/sec VAE fwd+bwd (hidden=300, depth=20, batch=32)",
        "value": m["value"], "unit": "molecules/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "warmup_steps_run": m.get("warmup_steps_run", a.warmup),
        "timed_regions_repeated": m.get("timed_regions_repeated", 0), "timed_region": m.get("timed_region"),
        "ms_per_step": m["ms_per_step"],
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic" + (" (host-resident batches, PCIe-inclusive diagnostic)" if a.host_input else ""),
        "config": {"workload": "BASELINE %s: %.1f atoms/molecule, motif vocab %d/%d, hidden=%d depth=%d latent=%d "
                               "batch=%d per GPU, %s cell; step = zero_grad + encoder fwd + KL + bwd%s + Adam"
                               % (cfg["tag"], m["atoms_per_molecule"], n_motif, n_attach, cfg["hidden"], cfg["depth"],
                                  cfg["latent"], cfg["batch"], rnn,
                                  "" if world == 1 else " + RCCL all-reduce" if a.backend == "nccl" else
                                  " + %s all-reduce (rehearsal backend, not RCCL)" % a.backend),
                   "baseline_config_index": a.config, "rnn_type": rnn, "global_batch": cfg["batch"] * world,
                   "parallelism": "dp%d" % world,
                   "arithmetic": "fp32 state, stashes, gate math and accumulation; gate products of the depth loops: the atom "
                                 "level (one column group) on split operands -- every fp32 operand the exact sum of three "
                                 "bf16 values, six of the nine partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate, "
                                 "within fp32 rounding of the fp32 product (tests: test_gate_products_on_split_operands_keep_"
                                 "fp32_accuracy) -- the tree-side levels and two-row-tile launches on v_mfma_f32_16x16x4_f32%s; "
                                 "tall weight-gradient contractions %s" % (
                                     " (bf16 operands under --dtype bf16 / the \"bf16\" leg)" if a.config == 4 else "",
                                     "with every fp32 operand split exactly into three bf16 terms, six of the nine partial "
                                     "products on v_mfma_f32_16x16x32_bf16, fp32 accumulate (fp32 accuracy: 3e-6 from fp64)"),
                   "algorithmic_gflop_per_step_per_gpu": m["algorithmic_gflop_per_step_per_gpu"],
                   "executed_gflop_per_step_per_gpu": m["executed_gflop_per_step_per_gpu"]},
        "host_enqueue_ms_per_step": m["host_enqueue_ms_per_step"],
        "host_lead_bound_steps": m["host_lead_bound_steps"],       # step i waits for step i - 2 (Workload.MAX_LEAD)
        "allreduce": m.get("allreduce"),                           # N > 1: which form of the gradient exchange was faster here
        "step_tflops_executed": m["step_tflops_executed"],
    }
    for k in ("full_depth_loops", "roofline"):
        if k in m:
            result[k] = m[k]
    if "tree_fixed_point" in m:
        result["config"]["tree_fixed_point"] = m["tree_fixed_point"]

    cpu_runs = [(rnn, main_wl.pool)]
    # the other message function on the same workload, in the same line (all 32 shipped model configs use LSTM)
    if a.config == 1 and a.rnn is None and not a.no_second_cell and not a.host_input:
        del main_wl      # (no empty_cache(): 288 GB of HBM leave nothing to make room for, and cached blocks spare device allocations)
        other = Workload(cfg, "LSTM", a, rank, world, dev)
        mo = other.measure(lib, rank)
        result["lstm"] = {k: mo[k] for k in ("ms_per_step", "value", "unit", "host_enqueue_ms_per_step",
                                             "algorithmic_gflop_per_step_per_gpu", "step_tflops_executed",
                                             "full_depth_loops", "roofline") if k in mo}
        cpu_runs.append(("LSTM", other.pool))

    # configs[4] names bf16: the same workload with bf16 gate products, in the same line
    if a.config == 4 and a.dtype == "f32" and not a.no_second_cell and not a.host_input:
        del main_wl
        other = Workload(cfg, rnn, a, rank, world, dev, gate_dtype="bf16")
        mo = other.measure(lib, rank)
        result["bf16"] = {k: mo[k] for k in ("ms_per_step", "value", "unit", "host_enqueue_ms_per_step",
                                             "step_tflops_executed", "full_depth_loops", "roofline") if k in mo}
        result["bf16"]["dtype"] = ("bf16 operands / fp32 accumulate (v_mfma_f32_16x16x32_bf16) for the H x H gate products of "
                                   "the depth loops; state, stashes, gate math, input projections and weight-gradient "
                                   "contractions fp32; tolerance: tests/test_gpu_parity.py::test_bf16_gate_products")

    # the reported row: full VAE training step (never allowed to cost the line)
    vae = None
    if a.config == 1 and not a.no_vae and not a.host_input:
        try:
            main_wl = other = None               # (their models, device batches and cached blocks are not the VAE row's business)
            import gc
            gc.unfreeze()
            gc.collect()
            vae = VaeWorkload(cfg, rnn, a, dev, rank, world)
            result["vae_step"] = vae.measure()
            if a.rnn is None and not a.no_second_cell:          # the LSTM leg of the same row
                vl = VaeWorkload(cfg, "LSTM", a, dev, rank, world)
                ml = vl.measure()
                result["vae_step"]["lstm"] = {k: ml[k] for k in ("ms_per_step", "value", "unit", "host_issue_ms", "schedule_in_loop",
                                                                 "schedule_ahead", "ms_per_step_index_structures_rebuilt", "roofline",
                                                                 "launches_per_step") if k in ml}
                del vl
        except Exception as exc:
            if world > 1:
                raise                            # (ranks must not diverge around collectives)
            result["vae_step"] = {"error": repr(exc)}
            vae = None

    # configs[4] (polymers, H = 600, depth 30) next to the headline row, so that its numbers are timed by whoever runs the
    # default command: fp32 and the bf16 leg, a few steps each on a pool of two batches; skipped, with the reason in the
    # line, once the run is past its time budget
    if a.config == 1 and world == 1 and not a.no_configs4 and not a.host_input and a.rnn is None:
        result["configs4"] = configs4_leg(a, lib, dev)

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if vae is not None:
            try:
                result["vae_step"]["cpu_baseline"] = vae.cpu_baseline()
            except Exception as exc:
                result["vae_step"]["cpu_baseline"] = {"error": repr(exc)}
        budget = 24.0 / len(cpu_runs)
        for cell, pool in cpu_runs:
            log("cpu baseline (%s) on %d threads" % (cell, host_cores()))
            try:
                cb = cpu_baseline(pool, cell, cfg["hidden"], cfg["depth"], cfg["latent"], n_motif, n_attach, budget_s=budget)
            except Exception as exc:          # the baseline is a reported number, never a reason to lose the line
                cb = {"error": repr(exc)}
            if cell == rnn:
                result["cpu_baseline"] = cb
            else:
                result["lstm"]["cpu_baseline"] = cb
    if rank == 0:
        _emit(result)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
