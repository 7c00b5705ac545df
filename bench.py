#!/usr/bin/env python3
"""bench.py -- molecules/s of the hierarchical encoder training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rnn GRU|LSTM]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic random-motif molecules (~40 atoms, motif vocab 500),
hidden = embed = 300, depthT = depthG = 20, batch 32 per GPU, fp32.  One "step" = zero_grad + HierMPNEncoder
forward + KL heads + backward (+ gradient all-reduce over RCCL for N > 1) + Adam update on one batch of 32
molecules whose tensorized index tensors are already resident in HBM.  Rank r consumes its own stream of
batches (weak scaling: per-GPU work fixed); value = all molecules processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline"     -- the dominant kernel (fused depth step) timed with HIP events on its own stream,
                    ALGORITHMIC flops per launch / mean launch time vs the fp32 MFMA peak;
  "cpu_baseline" -- the oracle (padded reference op order, PyTorch CPU) timed on this box's host cores on a
                    bounded sample of the same batches (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# The encoder uses two HIP streams per process and RCCL brings its own; with the default of 4 hardware queues an
# eagerly created communicator takes them first and the encoder's second stream ends up sharing a queue with the main
# one (measured: 5.94 instead of 5.45 ms/step).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0


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


def make_batches(n_batches, batch_size, seed0, motifs, n_motif, n_attach):
    from ggpm_amd import synth
    out = []
    for i in range(n_batches):
        specs = synth.random_batch(seed0 + i, batch_size, motifs=motifs, n_motif_vocab=n_motif,
                                   n_attach_vocab=n_attach)
        out.append(synth.tensorize(specs))
    return out


def algorithmic_work(batches, H, depth, gates):
    """FLOPs_fwd(level) = D*2*G*E*H^2 (+ small terms); fwd+bwd = 3x."""
    from ggpm_amd import synth
    fl = 0.0
    atoms = 0
    for tree, graph in batches:
        st = synth.batch_stats(tree, graph)
        for lvl, I, Fd in (("atom", 62, 38), ("tree", H + 20, H), ("tree", H + 20, H)):
            E, N = st[lvl]["E"], st[lvl]["N"]
            fl += depth * 2.0 * gates * E * H * H + 2.0 * gates * E * I * H + 2.0 * N * (Fd + H) * H
        atoms += st["atom"]["N"]
    return 3.0 * fl / len(batches), atoms / len(batches)


def host_cores():
    """CPU share of this process: affinity, capped by the cgroup quota (a GPU box grants ~16 per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("GGPM_CPU_THREADS", "16"))))


def log(msg):
    print("[bench %.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def cpu_baseline(batches, rnn, H, depth, latent, n_motif, n_attach, budget_s=20.0, max_steps=6):
    """Oracle (reference op order, PyTorch CPU, all host cores) fwd+bwd on the first batches."""
    from oracle import ref_encoder as ref
    from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = seeded_state_dict(encoder_param_shapes(rnn, H, n_motif, n_attach), 0)
    sd.update(seeded_state_dict(vae_head_shapes(H, latent), 7))
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in sd.items()}
    times, mols = [], 0
    t_begin = time.time()
    for i, (tree, graph) in enumerate(batches[:max_steps + 1]):
        tt, gt = ref.to_long_tensors(tree), ref.to_long_tensors(graph)
        t0 = time.time()
        outs = ref.hier_encoder_forward(p, rnn, depth, depth, tt, gt)
        _, kl = ref.rsample_kl(p, outs[0])
        loss = 0.1 * kl + 1e-3 * sum(o.sum() for o in outs)
        for v in p.values():
            v.grad = None
        loss.backward()
        dt = time.time() - t0
        log("cpu baseline step %d: %.2f s" % (i, dt))
        if i > 0:           # first step is warm-up
            times.append(dt)
            mols += len(tree[-1])
        if time.time() - t_begin > budget_s and len(times) >= 2:
            break
    if not times:
        return None
    return {"value": round(mols / sum(times), 2), "unit": "molecules/s", "cores": cores, "kind": "port",
            "sample": "%d fwd+bwd steps of batch %d after 1 warm-up, oracle/ref_encoder.py (reference padded op "
                      "order, torch CPU %d threads), median step %.3f s" % (len(times), len(batches[0][0][-1]),
                                                                            cores, float(np.median(times)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=16,
                    help="untimed steps; 16 = one pass over the pool of pre-tensorized batches, so that every batch shape has "
                         "been seen (allocator blocks, lazily created streams and events) before the timed region")
    ap.add_argument("--rnn", default="GRU", choices=["GRU", "LSTM"])
    ap.add_argument("--hidden", type=int, default=300)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--pool", type=int, default=16, help="distinct pre-tensorized batches per rank (cycled)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--host-input", action="store_true",
                    help="diagnostic: every step takes its batch from HOST memory through ggpm_amd.dataloader."
                         "DevicePrefetcher (pinned staging + async copy); the PCIe-inclusive rate, never `value`")
    ap.add_argument("--no-full-depth", action="store_true",
                    help="skip the extra timed pass without the tree fixed-point hint (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
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
    from ggpm_amd.encoder import PreparedBatch
    from ggpm_amd.nnutils import make_cuda
    from ggpm_amd.parallel import FlatGradSync, broadcast_parameters
    from ggpm_amd.property_vae import HierEncoderVAE, rsample
    lib = _lib.load(build_if_missing=False)

    n_motif, n_attach, motifs = 500, 1500, (8, 12)
    # rank r draws the batches r, r+W, ... of the seed-indexed synthetic set (10 000 molecules = 313 batches)
    pool = make_batches(a.pool, a.batch, seed0=1000 + rank * 313, motifs=motifs, n_motif=n_motif, n_attach=n_attach)
    dev_batches = [make_cuda(b) for b in pool]      # int64 index tensors resident in HBM before timing

    torch.manual_seed(0)
    model = HierEncoderVAE(make_args(a.rnn, a.hidden, a.depth, a.latent, n_motif, n_attach)).to(dev)
    for p in model.parameters():                       # vae_train.py:48-53
        if p.dim() == 1:
            torch.nn.init.constant_(p, 0)
        else:
            torch.nn.init.xavier_normal_(p)
    broadcast_parameters(model)
    H = a.hidden
    sync = FlatGradSync(model.parameters(), encoder=model.encoder)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)

    host_iter = None
    if a.host_input:
        import itertools
        from ggpm_amd.dataloader import DevicePrefetcher
        host_iter = iter(DevicePrefetcher(itertools.cycle(pool), device=dev, depth=2))

    def step(i):
        tree, graph = next(host_iter) if host_iter is not None else dev_batches[i % len(dev_batches)]
        sync.zero_grad()
        hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
        _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
        loss = 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())
        loss.backward()
        sync.all_reduce()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    log("model + %d batches resident; warm-up" % len(dev_batches))
    for i in range(a.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    sync.check_views()
    fence()
    log("warm-up done; timing %d steps" % a.steps)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    host_enqueue = time.perf_counter() - t0     # host time to enqueue K steps (diagnostic: host- vs GPU-bound)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    log("timed region done: %.3f ms/step (host enqueue %.3f ms/step)" % (1e3 * elapsed / a.steps,
                                                                         1e3 * host_enqueue / a.steps))

    # The same K steps once more WITHOUT the fixed-point hint (make_cuda measures the longest dependency chain of the
    # tree messages; with it the two tree-side levels stop at their fixed point, bit-identical results): every level
    # runs all `depth` launches.  Reported beside `value`, never instead of it.
    full_elapsed = None
    if host_iter is None and not a.no_full_depth and any(hasattr(t[0][3], "ggpm_chain") for t in dev_batches):
        hinted = dev_batches
        dev_batches = [(list(tree[:3]) + [tree[3].view_as(tree[3])] + list(tree[4:]), graph) for tree, graph in hinted]
        for i in range(min(a.warmup, 4)):
            step(i)
        fence()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)
        fence()
        full_elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([full_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            full_elapsed = float(t.item())
        dev_batches = hinted
        log("without the tree fixed-point hint: %.3f ms/step" % (1e3 * full_elapsed / a.steps))
    flops_step, atoms = algorithmic_work(pool, H, a.depth, 3 if a.rnn == "GRU" else 4)
    mols = a.steps * a.batch * world
    result = {
        "metric": "molecules/sec VAE fwd+bwd (hidden=300, depth=20, batch=32) -- HierMPNEncoder fwd+bwd target row",
        "value": round(mols / elapsed, 2), "unit": "molecules/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" + (" (host-resident batches, PCIe-inclusive diagnostic)" if a.host_input else ""),
        "config": {"workload": "BASELINE configs[1]: synthetic random-motif graphs, %.1f atoms/molecule, motif vocab "
                               "%d/%d, hidden=%d depth=%d batch=%d per GPU, %s cell; step = zero_grad + encoder "
                               "fwd + KL + bwd%s + Adam" % (atoms / a.batch, n_motif, n_attach, H, a.depth, a.batch,
                                                             a.rnn, " + RCCL all-reduce" if world > 1 else ""),
                   "rnn_type": a.rnn, "global_batch": a.batch * world, "parallelism": "dp%d" % world,
                   "algorithmic_gflop_per_step_per_gpu": round(flops_step / 1e9, 2)},
        "step_tflops_algorithmic": round(flops_step * world / (elapsed / a.steps) / 1e12, 3),
    }
    chains = [getattr(t[0][3], "ggpm_chain", 0) for t in dev_batches]
    if full_elapsed is not None:
        result["config"]["tree_fixed_point"] = (
            "motif-tree messages settle after their longest dependency chain (%d-%d steps in these batches); the two "
            "tree-side levels run chain+1 of the %d steps and replicate the last stash slot, outputs and gradients "
            "bit-identical to the full loops (tests/test_gpu_parity.py::test_tree_fixed_point_shortcut_is_bit_identical)"
            % (min(chains), max(chains), a.depth))
        result["full_depth_loops"] = {"ms_per_step": round(1e3 * full_elapsed / a.steps, 4),
                                      "value": round(mols / full_elapsed, 2), "unit": "molecules/s"}

    # ---- roofline of the dominant kernel: a second, instrumented pass over the same steps (HIP events
    # recorded on the launch stream around every fused depth-step launch; not part of `value`).
    if not a.no_roofline:
        def eager_step(i):      # instrumented launches are issued eagerly (HIP events bracket each one)
            tree, graph = dev_batches[i % len(dev_batches)]
            for p in model.parameters():
                p.grad = None
            hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
            _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
            (0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())).backward()

        eager_step(0)
        torch.cuda.synchronize()
        lib.ggpm_timing_enable(1)
        nprobe = min(a.steps, 4)
        for i in range(nprobe):
            eager_step(i)
        torch.cuda.synchronize()
        lib.ggpm_timing_enable(0)
        names = ["gru_fwd_a", "gru_bwd_a", "lstm_fwd_a", "lstm_bwd_a", "gru_fwd_b", "gru_bwd_b", "lstm_fwd_b",
                 "lstm_bwd_b"]
        per_kernel, best = {}, None
        for which, kname in enumerate(names):
            n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
            lib.ggpm_timing_collect(which, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
            if n.value:
                per_kernel[kname] = {"launches": n.value, "avg_launch_us": round(1e3 * ms.value / n.value, 3),
                                     "tflops": round(fl.value / (ms.value * 1e-3) / 1e12, 3)}
                if best is None or ms.value > best[2]:
                    best = (kname, n.value, ms.value, fl.value)
        if best and rank == 0:
            kname, n, ms, fl = best
            ach = fl / (ms * 1e-3) / 1e12
            traffic = None
            try:      # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    traffic = json.load(f).get(a.rnn, {}).get(kname)
            except Exception:
                pass
            result["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 3),
                                  "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": traffic,
                                  "launches": n, "avg_launch_us": round(1e3 * ms / n, 3),
                                  "flops_per_launch_avg": round(fl / n, 1), "all_depth_kernels": per_kernel}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("roofline pass done; cpu baseline on %d threads" % host_cores())
        try:
            result["cpu_baseline"] = cpu_baseline(pool, a.rnn, H, a.depth, a.latent, n_motif, n_attach)
        except Exception as exc:          # the baseline is a reported number, never a reason to lose the line
            result["cpu_baseline"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
