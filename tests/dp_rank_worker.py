"""One data-parallel rank of tests/test_aa_data_parallel_gpu.py (a helper script, not a test module).

Two of these run as gloo ranks sharing the one MI355X of the GPU box and train the real encoder through the C++
drivers with the gradient sink (``FlatGradSync(encoder=...)``: the backward writes the encoder's gradients straight
into the flat all-reduce buffer):

* parameters must stay BIT-identical across ranks after 6 steps on different batches, with GGPM_BUCKETED_ALLREDUCE at
  1 and at 0, and both settings must give the same parameters;
* two encoder backwards in one step (gradient accumulation) must give the gradients of the path without the sink
  (the driver overwrites its output buffers, so the second backward has to be added, not written, into the flat buffer).

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


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
        for sink in ("1", "0"):
            os.environ["GGPM_GRAD_SINK"] = sink
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
        os.environ.pop("GGPM_GRAD_SINK")
        scale = float(grads["0"].abs().max())
        d = float((grads["1"] - grads["0"]).abs().max())
        assert d <= 2e-6 * scale, (d, scale)
        print("%s two backwards per step: sink vs plain path max diff %.3e (scale %.3e)" % (rnn, d, scale), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    print("DP-RANK-OK", flush=True)


if __name__ == "__main__":
    main()
