"""CPU: host-side bookkeeping of ggpm_amd.functional / nnutils that needs no kernel -- the index-structure memo's
staleness guard, the hints make_cuda leaves on index tensors, and the deferred-gradient queue's behaviour around a
backward pass that raises."""
import contextlib

import pytest
import torch

from ggpm_amd import functional as F_
from ggpm_amd import nnutils


def test_index_memo_is_invalidated_by_an_in_place_refill():
    """A resident index tensor that is refilled with the next batch (``t.copy_(...)``) must not be served the CSR of its
    old contents: the memo is keyed by the tensor's in-place version counter."""
    t = torch.arange(12, dtype=torch.int64).view(4, 3)
    assert F_._memo_get(t, "_ggpm_csr", 5) is None
    F_._memo_put(t, "_ggpm_csr", 5, "old-structure")
    assert F_._memo_get(t, "_ggpm_csr", 5) == "old-structure"
    assert F_._memo_get(t, "_ggpm_csr", 6) is None            # other key (ncols)
    t.copy_(torch.zeros(4, 3, dtype=torch.int64))               # refill in place
    assert F_._memo_get(t, "_ggpm_csr", 5) is None
    F_._memo_put(t, "_ggpm_csr", 5, "new-structure")
    assert F_._memo_get(t, "_ggpm_csr", 5) == "new-structure"
    t[0, 0] = 3                                                  # any in-place write counts
    assert F_._memo_get(t, "_ggpm_csr", 5) is None


def test_tensor_hints_do_not_survive_an_in_place_refill():
    """``ggpm_chain`` (how many steps the tree-side levels need) and ``ggpm_roots`` describe the tensor's CONTENTS."""
    t = torch.zeros(6, 4, dtype=torch.int64)
    nnutils.attach_hint(t, "ggpm_chain", 7)
    assert nnutils.read_hint(t, "ggpm_chain", 0) == 7
    v = t.view_as(t)
    assert nnutils.read_hint(v, "ggpm_chain", 0) == 0           # a fresh view object carries no hint (bench.py relies on it)
    t.add_(1)
    assert nnutils.read_hint(t, "ggpm_chain", 0) == 0           # stale: the full depth loop runs
    nnutils.attach_hint(t, "ggpm_chain", 3)
    assert nnutils.read_hint(t, "ggpm_chain", 0) == 3


def test_tree_chain_length_is_bounded_by_the_motif_count():
    from ggpm_amd import synth
    specs = synth.random_batch(3, 4, motifs=(3, 6), n_motif_vocab=30, n_attach_vocab=90)
    tree, graph = synth.tensorize(specs)
    chain = nnutils.tree_chain_length(tree[3])          # (the host half of make_cuda; the upload needs the GPU)
    assert 1 <= chain <= 6


class _FakeStream:
    device = torch.device("cpu")

    def wait_stream(self, other):
        pass


class _Visit(torch.autograd.Function):
    """Stands in for a decoder op: its backward queues a deferred parameter gradient and may raise afterwards."""

    @staticmethod
    def forward(ctx, x, param, boom):
        ctx.param, ctx.boom = param, boom
        ctx.save_for_backward(x)
        return x * param.detach().sum()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        F_._defer_sum(ctx.param, torch.full_like(ctx.param, float((g * x).sum())))
        if ctx.boom:
            raise RuntimeError("injected failure inside the backward pass")
        return g * ctx.param.detach().sum(), None, None


@pytest.fixture
def fake_streams(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    for k in ("linear", "sum", "gather"):
        F_._DEFER[k].clear()
    F_._DEFER.update(task=None, stream=None, early=None, pending=[])
    yield
    for k in ("linear", "sum", "gather"):
        F_._DEFER[k].clear()
    F_._DEFER.update(task=None, stream=None, early=None, pending=[])


def test_deferred_gradients_survive_a_failed_backward(fake_streams):
    """the autograd engine does not run queued callbacks when a backward raises.  The next pass must start
    from a clean queue, flush at its end, and publish exactly its own gradients."""
    p = torch.nn.Parameter(torch.ones(3))
    x = torch.arange(4.0, requires_grad=True)

    def step(boom):
        p.grad = None
        y = _Visit.apply(_Visit.apply(x, p, False), p, boom)      # two visits of the same parameter in one pass
        y.sum().backward()

    step(False)
    want = p.grad.clone()
    assert want.abs().sum() > 0 and not F_._DEFER["sum"] and F_._DEFER["task"] is None
    with pytest.raises(RuntimeError, match="injected"):
        step(True)
    assert F_._DEFER["sum"], "the failed pass left its visit queued (the engine skipped the callback)"
    leftovers = F_._DEFER["task"]
    assert leftovers is not None
    step(False)                                                    # a normal step after the failure
    assert torch.equal(p.grad, want), (p.grad, want)               # not doubled by the failed pass's leftovers, not missing
    assert not F_._DEFER["sum"] and F_._DEFER["task"] is None
    step(False)
    assert torch.equal(p.grad, want)


def test_parameters_with_hooks_are_not_published_behind_autograd():
    """A parameter with a tensor hook / post-accumulate-grad hook (hook-based clippers, reducers) must get its gradient
    through AccumulateGrad so that the hooks fire.  Hooks on the AccumulateGrad NODE (stock DDP's reducer) cannot be seen
    from Python: for those the whole mechanism is switched off with ``publish_gradients(False)``."""
    p = torch.nn.Parameter(torch.ones(2))
    q = torch.nn.Parameter(torch.ones(2))
    assert F_.can_publish(p, q, None)
    h = q.register_hook(lambda g: g)
    assert not F_.can_publish(p, q)
    h.remove()
    assert F_.can_publish(p, q)
    h = p.register_post_accumulate_grad_hook(lambda t: None)
    assert not F_.can_publish(p)
    h.remove()
    frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)
    assert not F_.can_publish(frozen)
    assert not F_.can_publish(p * 2)                               # not a leaf
    # a hook on the accumulate node (what stock DDP registers) is invisible ...
    acc = p.expand_as(p).grad_fn.next_functions[0][0]
    h = acc.register_hook(lambda *a: None)
    assert F_.can_publish(p)
    # ... hence the process-wide switch
    assert F_.publish_gradients(False) is True
    try:
        assert not F_.can_publish(p) and not F_.can_publish(p, q, None)
    finally:
        assert F_.publish_gradients(True) is False
    assert F_.can_publish(p)
    h.remove()


def test_deferred_linear_is_keyed_by_its_column_split(fake_streams):
    """Two visits of one weight with different K splits must not be contracted with the first visit's split."""
    w = torch.nn.Parameter(torch.ones(2, 6))

    class _Q(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            F_._defer_linear(w, None, torch.ones(3, 2), [torch.ones(3, 6)], (6,))
            F_._defer_linear(w, None, torch.ones(3, 2), [torch.ones(3, 2), torch.ones(3, 4)], (2, 4))
            F_._defer_linear(w, None, torch.ones(5, 2), [torch.ones(5, 6)], (6,))
            keys = sorted(k[1] for k in F_._DEFER["linear"])
            assert keys == [(2, 4), (6,)]
            assert sorted(len(v[3]) for v in F_._DEFER["linear"].values()) == [1, 2]
            F_._DEFER["linear"].clear()                           # (the flush would need the GPU GEMM)
            return g

    x = torch.ones(2, requires_grad=True)
    _Q.apply(x).sum().backward()


@pytest.mark.parametrize("name", ["vae_lstm_s41", "vae_gru_s42"])
def test_state_dict_keys_are_the_references(name):
    """The parameter names of the reference's own HierPropertyVAE (recorded with its gradients in the fixtures) are
    exactly the drop-in's ``named_parameters``; ``state_dict`` adds only the two aliases the reference registers too
    (``decoder.rnn_cell``, ``decoder.E_assm``, ggpm/decoder.py:31-32)."""
    from golden_utils import VaeGolden
    from ggpm_amd.property_vae import HierPropertyVAE
    from ggpm_amd.vocab import IndexPairVocab
    g = VaeGolden(name)
    ref_names = {k.split("/", 1)[1] for k in g.z.files if k.startswith(("grad/", "gstat/"))}
    model = HierPropertyVAE(g.args(IndexPairVocab(g.n_motif, g.n_attach)))
    ours = {k for k, _ in model.named_parameters()}
    assert ours == ref_names, (sorted(ours - ref_names), sorted(ref_names - ours))
    extra = set(model.state_dict()) - ours
    assert extra and all(k.startswith(("decoder.rnn_cell.", "decoder.E_assm.")) or (g.tie and k.startswith("decoder.hmpn.E_"))
                         for k in extra), sorted(extra)


def test_published_gradients_are_not_stream_marked_by_default():
    """functional.hand_to: a gradient handed from a helper stream to the main stream gets no record_stream mark (on ROCm
    each mark is an event record on that stream when the tensor is released); _dev.RECORD_GRADS = True restores it"""
    from ggpm_amd import functional as F_

    class Probe:
        marks = 0

        def record_stream(self, stream):
            Probe.marks += 1

    from ggpm_amd import _dev
    F_.hand_to(Probe(), object())
    assert Probe.marks == (1 if _dev.RECORD_GRADS else 0)
    # the setting is read at call time: flipping it after import takes effect
    was, _dev.RECORD_GRADS = _dev.RECORD_GRADS, not _dev.RECORD_GRADS
    try:
        before = Probe.marks
        F_.hand_to(Probe(), object())
        assert Probe.marks - before == (1 if _dev.RECORD_GRADS else 0)
    finally:
        _dev.RECORD_GRADS = was



def test_index_structures_and_ctypes_marshalling_create_no_reference_cycles():
    """The host objects the step builds per call are freed by reference counting: the CSR remembered on an index
    tensor does not lead back to that tensor object, a transpose names its source weakly, array types are remembered."""
    import ctypes
    import gc
    import weakref
    import torch
    from ggpm_amd import _lib
    from ggpm_amd import functional as F_
    assert _lib.array_type(ctypes.c_void_p, 3) is _lib.array_type(ctypes.c_void_p, 3)
    assert _lib.array_type(ctypes.c_int, 3) is not _lib.array_type(ctypes.c_int, 4)
    gc.collect()
    gc.disable()
    try:
        idx = torch.tensor([2, 0, 1, 2], dtype=torch.int32)
        csr = F_.csr_from_index(idx, 3)
        assert F_.csr_from_index(idx, 3) is csr                      # remembered on the tensor ...
        assert csr.col is not idx and csr.col.data_ptr() == idx.data_ptr()      # ... through an alias, not the object itself
        t = F_.CSR(torch.zeros(4, dtype=torch.int32), torch.zeros(4, dtype=torch.int32), 3, 4)
        t._back = weakref.ref(csr)                                   # (what _build_T sets on a transpose)
        assert t.T is csr
        probe = weakref.ref(csr)
        del csr, idx, t
        assert probe() is None, "the CSR of an index tensor is kept alive by a reference cycle"
        arr = _lib.array_type(ctypes.c_void_p, 2)(1, 2)
        assert ctypes.addressof(arr) and arr._objects is None        # (ctypes.cast(arr, c_void_p) would store arr in arr._objects)
    finally:
        gc.enable()


def test_bench_harness_term_equals_the_plain_sum_of_the_outputs():
    """bench.py drives a gradient into every encoder output with ``1e-3 * (hroot.sum() + hnode.sum() + hinter.sum() +
    hatom.sum())``; over outputs that are row ranges of one buffer it forms that term as ONE reduction and its gradient as ONE
    fill (bench._SumOutputs).  Same value, same gradients -- also when the outputs are not views of one buffer."""
    import torch
    import bench
    torch.manual_seed(0)

    class Four(torch.autograd.Function):         # four outputs that are row ranges of one buffer, like fused._HierEncoder
        @staticmethod
        def forward(ctx, x):
            out = x * 2.0
            return out.split([3, 5, 5, 11])

        @staticmethod
        def backward(ctx, *g):
            assert all(t.is_contiguous() for t in g)
            return torch.cat(g, dim=0) * 2.0

    for views in (True, False):
        x = torch.randn(24, 8, dtype=torch.float64, requires_grad=True)
        outs = Four.apply(x) if views else tuple(t * 2.0 for t in x.split([3, 5, 5, 11]))
        a = 1e-3 * bench._sum_outputs(*outs)
        ga, = torch.autograd.grad(a, x)
        x2 = x.detach().clone().requires_grad_(True)
        outs2 = tuple(t * 2.0 for t in x2.split([3, 5, 5, 11]))
        b = 1e-3 * (outs2[0].sum() + outs2[1].sum() + outs2[2].sum() + outs2[3].sum())
        gb, = torch.autograd.grad(b, x2)
        assert abs(float(a) - float(b)) <= 1e-12 * abs(float(b)) and torch.equal(ga, gb), views
    ragged = (torch.ones(2, 3, requires_grad=True), torch.ones(2, 4, requires_grad=True))
    assert float(bench._sum_outputs(*ragged)) == 14.0
