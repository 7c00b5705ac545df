"""CPU, world_size 2 over gloo: molecule sharding + flat-gradient all-reduce (ggpm_amd/parallel.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ggpm_amd.parallel import FlatGradSync, broadcast_parameters, shard_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _data(i):
    g = torch.Generator().manual_seed(100 + i)
    return torch.randn(4, 6, generator=g)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model(seed=rank)                 # different init per rank ...
    broadcast_parameters(model, src=0)        # ... replicated from rank 0
    sync = FlatGradSync(model.parameters())
    out = []
    for step in range(2):
        mine = shard_indices(4, rank, world)  # batches 0..3 round-robin
        sync.zero_grad()
        for b in mine[step::2]:
            model(_data(b)).pow(2).mean().backward()
        sync.all_reduce()
        out.append(torch.cat([v.reshape(-1) for v in sync.views]).clone())       # (the buffer pads every parameter to 256 bytes)
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    q.put((rank, [o.numpy() for o in out], [p.detach().numpy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_indices_partition():
    parts = [shard_indices(10, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(10))
    assert parts[1] == [1, 5, 9]


def test_flat_grad_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # replicated parameters and identical reduced gradients on both ranks
    for a, b in zip(res[0][2], res[1][2]):
        assert (a == b).all()
    for a, b in zip(res[0][1], res[1][1]):
        assert (a == b).all()
    # equals the single-process mean over the two ranks' batches
    ref = _model(seed=0)
    for step in range(2):
        grads = []
        for rank in range(world):
            ref.zero_grad()
            for bidx in shard_indices(4, rank, world)[step::2]:
                ref(_data(bidx)).pow(2).mean().backward()
            grads.append(torch.cat([p.grad.reshape(-1) for p in ref.parameters()]))
        mean = (grads[0] + grads[1]) / 2
        assert torch.allclose(torch.from_numpy(res[0][1][step]), mean, atol=1e-7)


def test_flat_adam_equals_adam_over_the_parameter_list():
    """ggpm_amd.optim.FlatAdam (one flat view of all parameters, gradients in FlatGradSync's flat buffer) gives exactly
    the parameters torch.optim.Adam over model.parameters() gives (vae_train.py:60)."""
    from ggpm_amd.optim import FlatAdam
    a, b = _model(seed=3), _model(seed=3)
    oa = torch.optim.Adam(a.parameters(), lr=1e-2)
    sync = FlatGradSync(b.parameters(), keep_flat=True)
    ob = FlatAdam(sync, lr=1e-2)
    for i in range(4):
        x = _data(i)
        oa.zero_grad()
        a(x).pow(2).mean().backward()
        oa.step()
        ob.zero_grad()
        b(x).pow(2).mean().backward()
        sync.all_reduce()
        ob.step()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync.views))
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)


def test_step_metrics_reads_on_first_access_and_behaves_like_the_reference_dict():
    """ggpm_amd.property_vae.StepMetrics: the reference's metrics dictionary (ggpm/property_vae.py:60-62) with its values
    copied from the tensors on first access; vae_train.py:86 builds ``np.array([metrics['Loss'], ...])`` from it."""
    import numpy as np
    import torch
    from ggpm_amd.property_vae import StepMetrics
    vals = (torch.tensor(98.875), torch.tensor(32.5), torch.tensor(0.25), torch.tensor(0.125), torch.tensor(0.5), 1)
    m = StepMetrics(vals)
    assert not m._ready and list(m) == ['Loss', 'KL:', 'Word', 'I-Word', 'Topo', 'Assm'] and len(m) == 6 and 'KL:' in m
    assert not m._ready                                   # nothing above touched a value
    arr = np.array([m['Loss'], m['KL:'], m['Word'], m['I-Word'], m['Topo'], m['Assm']])
    assert m._ready and arr.tolist() == [98.875, 32.5, 0.25, 0.125, 0.5, 1.0]
    assert dict(m.items()) == {'Loss': 98.875, 'KL:': 32.5, 'Word': 0.25, 'I-Word': 0.125, 'Topo': 0.5, 'Assm': 1.0}
    assert all(isinstance(v, float) for v in m.values()) and m.get('nope', 7) == 7


def test_flat_buffers_keep_every_parameter_on_a_256_byte_boundary():
    """FlatGradSync / FlatAdam: a parameter with an odd element count (the decoder's topoNN ends in Linear(H, 1): a bias of one
    float) must not leave the parameters behind it 4-byte aligned -- the vector-load GEMM kernels need 16-byte aligned
    weights and fall back to scalar loads otherwise.  Same numbers as torch.optim.Adam on separately allocated tensors."""
    import copy
    from ggpm_amd.optim import FlatAdam
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(5, 1), torch.nn.Linear(1, 7), torch.nn.Linear(7, 3))
    ref = copy.deepcopy(model)
    sync = FlatGradSync(model.parameters(), keep_flat=True)
    assert sync.offsets == sorted(sync.offsets) and all(o % 64 == 0 for o in sync.offsets)
    assert sync.flat.numel() % 64 == 0 and sync.flat.numel() >= sum(p.numel() for p in model.parameters())
    opt = FlatAdam(sync, lr=1e-2)
    base = opt.flat.data_ptr()
    assert all((p.data_ptr() - base) % 256 == 0 for p in model.parameters())
    assert all((v.data_ptr() - sync.flat.data_ptr()) % 256 == 0 for v in sync.views)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    x = torch.randn(11, 5)
    for _ in range(3):
        sync.zero_grad()
        model(x).pow(2).sum().backward()
        sync.all_reduce()
        opt.step()
        ropt.zero_grad()
        ref(x).pow(2).sum().backward()
        ropt.step()
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7)
    pad = torch.ones(sync.flat.numel(), dtype=torch.bool)
    for o, p in zip(sync.offsets, sync.params):
        pad[o:o + p.numel()] = False
    assert float(opt.flat.data[pad].abs().max()) == 0.0 and float(sync.flat[pad].abs().max()) == 0.0      # the padding stays zero
