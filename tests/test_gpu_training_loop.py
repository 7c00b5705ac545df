"""GPU: the training-loop SHAPE of the reference's vae_train.py:48-99 on the drop-in model -- ``model(*batch, beta=beta)``
with the networkx ``graphs`` (no pre-built schedule), ``clip_grad_norm_``, Adam, metric accumulation, ``param_norm`` /
``grad_norm``, ``torch.save(state_dict)`` -- plus the host-state guards that only show in a loop of several steps (index
memo vs in-place refill, a backward pass that raises)."""
import io
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


class _Configs:
    """The argument bag of vae_train.py (``configs``) for the fields the model reads, with the values of
    configs/pretrained_wo_tie_embedding_configs.json:14-23,39 (LSTM, H = He = 250, latent 24, depth 20, diter 1 / 5,
    dropout 0.1) unless overridden."""

    def __init__(self, vocab, **kw):
        self.vocab = vocab
        self.atom_vocab = type("V", (), {"size": lambda s: 38})()
        self.rnn_type, self.embed_size, self.hidden_size, self.latent_size = "LSTM", 250, 250, 24
        self.depthT = self.depthG = 20
        self.diterT, self.diterG, self.dropout, self.tie_embedding = 1, 5, 0.1, False
        self.lr, self.clip_norm, self.beta = 1e-3, 20.0, 0.1
        for k, v in kw.items():
            setattr(self, k, v)


def _init_like_vae_train(model):
    for param in model.parameters():             # vae_train.py:48-53
        if param.dim() == 1:
            nn.init.constant_(param, 0)
        else:
            nn.init.xavier_normal_(param)


def param_norm(m):                               # vae_train.py:63
    return math.sqrt(sum([p.norm().item() ** 2 for p in m.parameters()]))


def grad_norm(m):                                # vae_train.py:64
    return math.sqrt(sum([p.grad.norm().item() ** 2 for p in m.parameters() if p.grad is not None]))


@pytest.mark.parametrize("rnn,tie", [("LSTM", False), ("GRU", True)])
def test_vae_train_loop_on_the_drop_in_model(rnn, tie):
    """vae_train.py:71-99 unchanged: zero_grad / model.train() / ``model(*batch, beta=beta)`` (6-tuple with the networkx
    graphs, numpy tensors: the decode schedule is derived inside the forward, ``DecodeSchedule.from_graphs``) / backward /
    clip_grad_norm_ / Adam / metric accumulation over ``metrics_.items()`` / param_norm, grad_norm / save, load."""
    from ggpm_amd import synth
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(60, 180)
    configs = _Configs(vocab, rnn_type=rnn, tie_embedding=tie)
    torch.manual_seed(3)
    model = HierPropertyVAE(configs).to(_dev())
    _init_like_vae_train(model)
    optimizer = torch.optim.Adam(model.parameters(), lr=configs.lr)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, 0.9)
    # a "DataFolder" of two pickled-style batches (numpy tensors + networkx graphs), cycled
    dataset = [synth.train_batch(synth.random_batch(100 + i, 8, motifs=(3, 8), n_motif_vocab=60, n_attach_vocab=180))
               for i in range(2)]
    total_step, beta, metrics, losses = 0, configs.beta, {}, []
    for epoch in range(3):
        for batch in dataset:
            total_step += 1
            model.zero_grad()
            model.train()
            loss, metrics_ = model(*batch, beta=beta)
            loss.backward()
            nn.utils.clip_grad_norm_(model.parameters(), configs.clip_norm)
            optimizer.step()
            for k, v in metrics_.items():
                metrics[k] = v if k not in metrics else metrics[k] + v
            losses.append(metrics_["Loss"])
            gn, pn = grad_norm(model), param_norm(model)
            assert math.isfinite(gn) and gn > 0 and math.isfinite(pn), (total_step, gn, pn)
            assert all(p.grad is not None for p in model.parameters()), [k for k, p in model.named_parameters() if p.grad is None]
        scheduler.step()
    assert total_step == 6 and all(math.isfinite(x) for x in losses), losses
    assert set(metrics) == {"Loss", "KL:", "Word", "I-Word", "Topo", "Assm"}
    assert all(isinstance(v, float) for v in metrics.values())
    # each batch is seen three times: the loss on it goes down (dropout noise is far smaller than three Adam steps)
    assert losses[4] < losses[0] and losses[5] < losses[1], losses

    # torch.save(model.state_dict()) -> a fresh model -> the same loss (eval mode: no dropout; no latent noise)
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    fresh = HierPropertyVAE(configs).to(_dev())
    fresh.load_state_dict(torch.load(buf))
    assert set(fresh.state_dict()) == set(model.state_dict())
    model.eval(); fresh.eval()
    with torch.no_grad():
        a, _ = model(*dataset[0], beta=beta, perturb_z=False)
        b, _ = fresh(*dataset[0], beta=beta, perturb_z=False)
    assert float(a) == float(b), (float(a), float(b))


def test_forward_with_graphs_equals_forward_with_a_prebuilt_schedule():
    """``model(*batch)`` builds the decode schedule from the networkx batch inside the forward (the reference's call
    shape); passing ``schedule=`` only moves that work out of the step.  Same loss, same gradients, bit for bit."""
    from ggpm_amd import synth
    from ggpm_amd.decoder import DecodeSchedule
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    configs = _Configs(vocab, rnn_type="GRU", hidden_size=64, embed_size=64, latent_size=16, depthT=6, depthG=6, dropout=0.0)
    torch.manual_seed(5)
    model = HierPropertyVAE(configs).to(_dev())
    _init_like_vae_train(model)
    specs = synth.random_batch(9, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120)
    batch = synth.train_batch(specs)

    def run(**kw):
        model.zero_grad()
        loss, m = model(*batch, beta=0.3, perturb_z=False, **kw)
        loss.backward()
        return float(loss), {k: p.grad.clone() for k, p in model.named_parameters()}

    la, ga = run()
    lb, gb = run(schedule=DecodeSchedule.from_specs(specs, batch[2]))
    assert la == lb
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


def test_schedule_ahead_loop_equals_the_plain_loop():
    """``for batch in ScheduleAhead(dataset, model)`` (schedules built one batch ahead on a worker thread) against the
    unchanged loop: same losses and the same parameters after four Adam steps, bit for bit."""
    from ggpm_amd import synth
    from ggpm_amd.dataloader import ScheduleAhead
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    configs = _Configs(vocab, rnn_type="GRU", hidden_size=64, embed_size=64, latent_size=16, depthT=6, depthG=6, dropout=0.0)
    dataset = [synth.train_batch(synth.random_batch(70 + i, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120))
               for i in range(4)]

    def loop(wrap):
        torch.manual_seed(11)
        model = HierPropertyVAE(configs).to(_dev())
        _init_like_vae_train(model)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        losses = []
        for batch in (ScheduleAhead(dataset, model) if wrap else dataset):
            model.zero_grad()
            loss, m = model(*batch, beta=0.3, perturb_z=False)
            loss.backward()
            opt.step()
            losses.append(m["Loss"])
        return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}

    la, pa = loop(False)
    lb, pb = loop(True)
    assert la == lb, (la, lb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k


def test_metrics_read_through_the_event_equal_the_blocking_read(monkeypatch):
    """StepMetrics: the six values copied to pinned memory behind the forward and read through the copy's event are the
    floats the blocking read returns (_dev.METRICS_ASYNC = False) and the reference's ``.item()`` values (GGPM_LAZY_METRICS=0)."""
    from ggpm_amd import synth
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    configs = _Configs(vocab, rnn_type="GRU", hidden_size=64, embed_size=64, latent_size=16, depthT=6, depthG=6, dropout=0.0)
    torch.manual_seed(5)
    model = HierPropertyVAE(configs).to(_dev())
    _init_like_vae_train(model)
    batch = synth.train_batch(synth.random_batch(21, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120))
    got = []
    from ggpm_amd import _dev as dev_settings
    for lazy, asyn in (("1", True), ("1", False), ("0", False)):
        monkeypatch.setenv("GGPM_LAZY_METRICS", lazy)
        monkeypatch.setattr(dev_settings, "METRICS_ASYNC", asyn)
        loss, m = model(*batch, beta=0.3, perturb_z=False)
        loss.backward()
        model.zero_grad()
        got.append({k: float(v) for k, v in m.items()})
    assert set(got[0]) == {"Loss", "KL:", "Word", "I-Word", "Topo", "Assm"}
    assert got[0] == got[1] == got[2], got


def test_refilling_a_resident_index_tensor_rebuilds_its_csr():
    """the CSR memo hangs on the index tensor object; an in-place refill must invalidate it."""
    from ggpm_amd import functional as F_
    a = np.array([[0, 0, 0], [2, 3, 0], [1, 0, 0], [1, 2, 0]], dtype=np.int64)
    b = np.array([[0, 0, 0], [3, 0, 0], [3, 1, 0], [0, 0, 0]], dtype=np.int64)
    t = torch.from_numpy(a).to(_dev())
    c1 = F_.csr_from_padded(t, ncols=4)
    assert F_.csr_from_padded(t, ncols=4) is c1                  # resident batch: built once
    assert c1.col.cpu().numpy()[:5].tolist() == [2, 3, 1, 1, 2]
    t.copy_(torch.from_numpy(b))                                   # next batch into the same buffer
    c2 = F_.csr_from_padded(t, ncols=4)
    assert c2 is not c1
    assert c2.rowptr.cpu().numpy().tolist() == [0, 0, 1, 3, 3] and c2.col.cpu().numpy()[:3].tolist() == [3, 3, 1]


def test_a_failed_backward_does_not_poison_the_next_steps(monkeypatch):
    """after a backward pass that raises (here: a hook on an activation), the next pass must publish every
    deferred parameter gradient again, equal to the path without deferral."""
    from ggpm_amd import functional as F_
    dev = _dev()
    torch.manual_seed(0)
    lin1, lin2 = nn.Linear(24, 32).to(dev), nn.Linear(32, 8).to(dev)
    emb = nn.Embedding(11, 24).to(dev)
    idx = torch.tensor([1, 5, 5, 7, 0, 10, 3, 3, 3], dtype=torch.int32, device=dev)
    idx_csr = F_.csr_from_index(idx, ncols=11)

    def forward(boom):
        x = F_.gather_rows(emb.weight, idx, idx_csr, 24, 24)
        h = F_.linear([x], [24], lin1.weight, lin1.bias, act=F_.ACT_RELU)
        h2 = F_.linear([h[:, :32].contiguous()], [32], lin2.weight, lin2.bias)
        if boom:
            h.register_hook(lambda g: (_ for _ in ()).throw(RuntimeError("injected")))
        return (h2[:, :8] ** 2).sum() + h[:, :32].sum()

    params = [emb.weight, lin1.weight, lin1.bias, lin2.weight, lin2.bias]

    def grads(boom=False):
        for p in params:
            p.grad = None
        forward(boom).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params]

    monkeypatch.setenv("GGPM_DEFER_WGRADS", "0")
    want = grads()
    monkeypatch.setenv("GGPM_DEFER_WGRADS", "1")
    first = grads()
    with pytest.raises(RuntimeError, match="injected"):
        grads(boom=True)
    again = grads()
    third = grads()
    for w, f, a, t3 in zip(want, first, again, third):
        scale = float(w.abs().max()) + 1e-12
        assert float((f - w).abs().max()) <= 1e-5 * scale
        assert torch.equal(a, f) and torch.equal(t3, f)            # nothing lost, nothing doubled
    assert not F_._DEFER["linear"] and not F_._DEFER["gather"] and not F_._DEFER["sum"]


def test_a_parameter_with_a_hook_gets_its_gradient_through_autograd():
    from ggpm_amd import functional as F_
    dev = _dev()
    torch.manual_seed(1)
    lin = nn.Linear(16, 16).to(dev)
    x = torch.randn(5, 16, device=dev)
    seen = []
    h = lin.weight.register_hook(lambda g: seen.append(float(g.abs().sum())) or g)
    y = F_.linear([x], [16], lin.weight, lin.bias)
    y[:, :16].sum().backward()
    torch.cuda.synchronize()
    h.remove()
    assert len(seen) == 1 and seen[0] > 0
    want = torch.ones(5, 16, device=dev).t() @ x
    assert float((lin.weight.grad - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("rnn", ["GRU", "LSTM"])
def test_a_training_step_leaves_nothing_for_the_cyclic_collector(rnn):
    """vae_train.py's loop body, repeated: every object a step creates is freed by reference counting when the step is over.
    Reference cycles (a CSR and its transpose naming each other, an index memo that leads back to its own tensor, the atom
    plan and its schedule, a ``ctypes.cast`` result, a ctypes array type built per call) would hold a finished batch's device
    tables until Python's collector comes round and cost its passes on the thread that issues the launches."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from tools.gc_cycles import cycles_of
    from ggpm_amd import synth
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    configs = _Configs(vocab, rnn_type=rnn, hidden_size=64, embed_size=64, latent_size=16, depthT=6, depthG=6, dropout=0.1)
    torch.manual_seed(5)
    model = HierPropertyVAE(configs).to(_dev())
    _init_like_vae_train(model)
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3)
    dataset = [synth.train_batch(synth.random_batch(200 + i, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120))
               for i in range(2)]
    k = [0]

    def step():
        batch = dataset[k[0] % 2]
        k[0] += 1
        model.zero_grad()
        loss, metrics = model(*batch, beta=0.1)
        loss.backward()
        optimizer.step()
        assert math.isfinite(metrics["Loss"])

    step()
    step()
    report = cycles_of(step, repeat=4)
    assert report[0].startswith("0 objects"), "\n".join(report)


def test_index_structures_that_come_with_the_schedule_equal_the_device_built_ones():
    """csrc/schedule.hip puts the CSRs, transposes and frozen masks of the two tree-side levels and the transposes of the
    heads' molecule indices into the schedule's upload; the device kernels (ggpm_padded_to_csr, ggpm_csr_transpose) derive
    the same structures from the same tables: entry for entry."""
    from ggpm_amd import functional as F_, synth, tree_decode as TD
    from ggpm_amd.decoder import DecodeSchedule
    specs = synth.random_batch(31, 9, motifs=(3, 9), n_motif_vocab=40, n_attach_vocab=120)
    tensors = synth.tensorize(specs)
    sch = DecodeSchedule.from_specs(specs, tensors, depth=5, gates=3)
    assert sch._native is not None
    D = sch.to_device(_dev())._dev
    T, E1, B = D["plan"], sch.plan["E1"], sch.batch_size
    pre = D["level_structs"]

    def same_csr(a, b, what):
        assert (a.rows, a.ncols) == (b.rows, b.ncols), what
        nnz = int(b.rowptr[-1])
        assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col[:nnz], b.col[:nnz]), what
        at, bt = a.T, b.T
        assert torch.equal(at.rowptr, bt.rowptr) and torch.equal(at.col[:nnz], bt.col[:nnz]), what + " (transposed)"

    for tag, ids, n_extra in (("inter", "inst_attach", 0), ("tree", "inst_motif", B)):
        plain = TD.LevelSpec(T[ids], T["mess_inst"], T["mess_pos"], T["dag_" + tag], T["in_" + tag], E1, n_extra, 5)
        frozen, pred, inc, src = plain.structures()
        f2, p2, i2, s2 = pre[tag]
        assert torch.equal(frozen, f2), tag
        same_csr(p2, pred, tag + " pred")
        same_csr(i2, inc, tag + " incoming")
        same_csr(s2, src, tag + " source")
    for k in ("topo", "cls", "assm"):
        idx = D[k + "_batch32"]
        if idx.numel():
            same_csr(D["head_csr"][k], F_.csr_from_index(idx.clone(), ncols=B), k)
