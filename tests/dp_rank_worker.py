"""One data-parallel rank of tests/test_aa_data_parallel_gpu.py (a helper script, not a test module).

Two of these run as gloo ranks sharing the one MI355X of the GPU box and train the real encoder through the C++
drivers with the gradient sink (``FlatGradSync(encoder=...)``: the backward writes the encoder's gradients straight
into the flat all-reduce buffer):

* parameters must stay BIT-identical across ranks after 6 steps on different batches, with GGPM_BUCKETED_ALLREDUCE at
  1 and at 0, and both settings must give the same parameters;
* two encoder backwards in one step (gradient accumulation) must give the gradients of the path without the sink
  (the driver overwrites its output buffers, so the second backward has to be added, not written, into the flat buffer);
* the FULL model (``HierPropertyVAE`` with ``tie_embedding=True``: the tied ``E_c`` / ``E_i`` of ggpm/encoder.py:92-94
  receive gradients from the encoder's driver AND from the decoder's deferred scatter) trained in the order of
  vae_train.py:78-83 -- ``model(*batch, beta=beta)``, backward, all-reduce, ``clip_grad_norm_``, Adam -- for 4 steps on
  different batches per rank: parameters stay BIT-identical across ranks, and the all-reduced gradient of the first step
  equals the single-process gradient of the two ranks' batches concatenated.

Prints one line per check and "DP-RANK-OK" at the end; any failed assertion ends the process with a non-zero code.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

import bench
from ggpm_amd.nnutils import make_cuda
from ggpm_amd.parallel import FlatGradSync, broadcast_parameters
from ggpm_amd.property_vae import HierEncoderVAE, rsample


def loss_of(model, batch):
    tree, graph = batch
    hroot, hnode, hinter, hatom = model.encoder.forward_padded(tree, graph)
    _, kl = rsample(hroot, model.R_mean, model.R_var, perturb=False)
    return 0.1 * kl + 1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() + hatom.sum())


def full_vae_section(rank, world):
    from ggpm_amd import synth
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(50, 150)

    def specs_of(r, i):
        return synth.random_batch(2000 + 313 * r + i, 6, motifs=(3, 7), n_motif_vocab=50, n_attach_vocab=150)

    def build(rnn):
        a = bench.make_args(rnn, 100, 5, 16, 50, 150)
        a.vocab, a.diterT, a.diterG, a.tie_embedding = vocab, 1, 3, True
        torch.manual_seed(0)
        m = HierPropertyVAE(a).cuda()
        for p in m.parameters():                       # vae_train.py:48-53
            if p.dim() == 1:
                torch.nn.init.constant_(p, 0)
            else:
                torch.nn.init.xavier_normal_(p)
        return m

    for rnn in ("GRU", "LSTM"):
        model = build(rnn)
        assert model.encoder.E_c[0].weight is model.decoder.hmpn.E_c[0].weight      # tied
        broadcast_parameters(model)
        sync = FlatGradSync(model.parameters(), encoder=model.encoder)
        assert sync.encoder_params, "the gradient sink must be installed"
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        start = {k: v.detach().clone() for k, v in model.state_dict().items()}
        for i in range(4):
            batch = synth.train_batch(specs_of(rank, i))
            sync.zero_grad()
            model.train()
            loss, metrics = model(*batch, beta=0.1, perturb_z=False)
            loss.backward()
            sync.all_reduce()
            if i == 0:
                reduced = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
            torch.nn.utils.clip_grad_norm_(model.parameters(), 20.0)
            opt.step()
            assert all(torch.isfinite(p).all() for p in model.parameters())
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "full VAE: parameters differ across ranks"
        print("VAE %s tie_embedding: ranks bit-identical after 4 steps (loss %.4f)" % (rnn, metrics["Loss"]), flush=True)

        # the reduced gradient of step 0 vs ONE process on the concatenation of all ranks' batches (no sink, no collective)
        ref = build(rnn)
        ref.load_state_dict(start)
        ref.train()
        cat = [m for r in range(world) for m in specs_of(r, 0)]
        loss, _ = ref(*synth.train_batch(cat), beta=0.1, perturb_z=False)
        loss.backward()
        torch.cuda.synchronize()
        # (the reference pads every attachment prediction to the BATCH's max_cls_size candidates and the padded rows take
        # part in the softmax, ggpm/decoder.py:199,252-254: the comparison needs batches that agree on it)
        from ggpm_amd.decoder import DecodeSchedule
        sizes = {DecodeSchedule.from_specs(b, synth.tensorize(b)).max_cls_size for b in [cat] + [specs_of(r, 0) for r in range(world)]}
        assert len(sizes) == 1, sizes
        gmax = max(float(p.grad.abs().max()) for p in ref.parameters())
        worst, worst_k = 0.0, ""
        for k, p in ref.named_parameters():
            want, got = p.grad, reduced[k]
            # norm-wise per tensor; a gradient that is analytically zero (W_assm.bias: the candidates of a prediction share
            # one context vector and a softmax row's gradients sum to 0) holds rounding noise on both sides
            if k.endswith("W_assm.bias"):
                assert float(want.abs().max()) < 1e-4 * gmax and float(got.abs().max()) < 1e-4 * gmax, k
                continue
            scale = max(float(want.abs().max()), 1e-5 * gmax)
            err = float((want - got).abs().max()) / scale
            if err > worst:
                worst, worst_k = err, k
        assert worst <= 1e-5, (worst, worst_k)
        print("VAE %s 2-rank all-reduced gradient vs 1-rank gradient of the concatenated batch: worst norm-wise diff %.3e"
              % (rnn, worst), flush=True)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full_vae_section(rank, world)
    for rnn in ("GRU", "LSTM"):
        pool = bench.make_batches(4, 8, seed0=1000 + rank * 313, gen=(4, 7), n_motif=50, n_attach=150)
        batches = [make_cuda(b) for b in pool]
        results = {}
        for bucketed in ("1", "0"):
            os.environ["GGPM_BUCKETED_ALLREDUCE"] = bucketed
            torch.manual_seed(0)
            model = HierEncoderVAE(bench.make_args(rnn, 100, 5, 16, 50, 150)).cuda()
            broadcast_parameters(model)
            sync = FlatGradSync(model.parameters(), encoder=model.encoder)
            assert sync.encoder_params, "the gradient sink must be installed"
            opt = torch.optim.SGD(model.parameters(), lr=0.05)
            for i in range(6):
                sync.zero_grad()
                loss_of(model, batches[i % len(batches)]).backward()
                # the C++ backward really wrote into the flat buffer (no pack copy of the encoder's gradients)
                assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.encoder_params, sync.encoder_views))
                sync.all_reduce()
                opt.step()
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            gathered = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(gathered, flat)
            assert all(torch.equal(gathered[0], g) for g in gathered), "parameters differ across ranks"
            results[bucketed] = flat.cpu()
            print("%s bucketed=%s early_numel=%d: ranks bit-identical" % (rnn, bucketed, sync.early_numel), flush=True)
        d = float((results["1"] - results["0"]).abs().max())
        assert d <= 1e-6, d
        print("%s max |bucketed - single collective| over parameters: %.3e" % (rnn, d), flush=True)

        # ---- two backwards per step: sink path vs the path without the sink
        os.environ["GGPM_BUCKETED_ALLREDUCE"] = "0"
        grads = {}
        from ggpm_amd import _dev as dev_settings
        for sink in ("1", "0"):
            dev_settings.GRAD_SINK = sink == "1"
            torch.manual_seed(0)
            model = HierEncoderVAE(bench.make_args(rnn, 100, 5, 16, 50, 150)).cuda()
            broadcast_parameters(model)
            sync = FlatGradSync(model.parameters(), encoder=model.encoder)
            assert bool(sync.encoder_params) == (sink == "1")
            sync.zero_grad()
            loss_of(model, batches[0]).backward()
            loss_of(model, batches[1]).backward()
            sync.all_reduce()
            grads[sink] = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
        dev_settings.GRAD_SINK = True
        scale = float(grads["0"].abs().max())
        d = float((grads["1"] - grads["0"]).abs().max())
        assert d <= 2e-6 * scale, (d, scale)
        print("%s two backwards per step: sink vs plain path max diff %.3e (scale %.3e)" % (rnn, d, scale), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("DP-RANK-OK", flush=True)


if __name__ == "__main__":
    main()
