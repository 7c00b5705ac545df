"""CPU: pin the oracle (oracle/ref_encoder.py) and the layout restatement against the reference's outputs."""
import numpy as np
import pytest
import torch

from ggpm_amd import synth
from oracle import ref_encoder as ref
from golden_utils import Golden, case_names, rel_err

CASES = case_names()


def _run_oracle(g, dtype):
    p = g.params(dtype=dtype, requires_grad=True)
    tree, graph = g.tensors()
    trace = {}
    outs = ref.hier_encoder_forward(p, g.rnn, g.depthT, g.depthG, tree, graph, trace=trace)
    z, kl = ref.rsample_kl(p, outs[0])
    coeffs = g.loss_coeffs([tuple(o.shape) for o in outs])
    loss = g.beta * kl
    for c, o in zip(coeffs, outs):
        loss = loss + (torch.from_numpy(c).to(dtype) * o).sum()
    loss.backward()
    return p, outs, trace, z, kl, loss


@pytest.mark.parametrize("name", CASES)
def test_layout_restatement_matches_reference_tensorize(name):
    g = Golden(name)
    specs = synth.random_batch(g.seed, g.B, motifs=g.motifs, n_motif_vocab=g.n_motif, n_attach_vocab=g.n_attach)
    tree, graph = synth.tensorize(specs)
    gt, gg = g.numpy_tensors()
    for a, b in zip(tree[:-1], gt[:-1]):
        assert a.shape == b.shape and (a == b).all()
    for a, b in zip(graph[:-1], gg[:-1]):
        assert a.shape == b.shape and (a == b).all()
    assert [tuple(x) for x in tree[-1]] == gt[-1]
    assert [tuple(x) for x in graph[-1]] == gg[-1]
    # invariants the reference relies on (decoder.py:110-115): pad row and trailing zero column
    for t in (tree, graph):
        assert (t[2][0] == 0).all() and (t[3][0] == 0).all()
        assert (t[2][:, -1] == 0).all() and (t[3][:, -1] == 0).all()


@pytest.mark.parametrize("name", CASES)
def test_oracle_fp32_matches_reference_outputs(name):
    g = Golden(name)
    p, outs, trace, z, kl, loss = _run_oracle(g, torch.float32)
    tol = 2e-6 if g.H <= 32 else 2e-5
    for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs):
        assert rel_err(o.detach().numpy(), g.z[k]) <= tol, k
    assert rel_err(trace["atom"][0].detach().numpy(), g.z["atom_h1"]) <= tol
    assert rel_err(trace["atom"][-1].detach().numpy(), g.z["atom_hD"]) <= tol
    assert rel_err(trace["inter"][-1].detach().numpy(), g.z["inter_hD"]) <= tol
    assert rel_err(trace["tree"][-1].detach().numpy(), g.z["tree_hD"]) <= tol
    assert abs(float(kl.detach()) - float(g.z["kl"])) <= tol * max(1.0, abs(float(g.z["kl"])))
    assert abs(float(loss.detach()) - float(g.z["loss"])) <= 1e-5 * max(1.0, abs(float(g.z["loss"])))
    for k, v in p.items():
        grad = v.grad if v.grad is not None else torch.zeros_like(v)
        g.check_grad(k, grad.numpy(), rel=5e-5)


@pytest.mark.parametrize("name", [c for c in CASES if c.startswith("tiny")])
def test_oracle_fp64_matches_reference_fp64(name):
    g = Golden(name)
    p, outs, trace, z, kl, loss = _run_oracle(g, torch.float64)
    for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs):
        assert rel_err(o.detach().numpy(), g.z[k + "_f64"]) <= 2e-7, k   # fixture stores the f64 pass rounded to fp32
    assert abs(float(kl.detach()) - float(g.z["kl_f64"])) <= 1e-10
    for k, v in p.items():
        grad = v.grad if v.grad is not None else torch.zeros_like(v)
        g.check_grad(k, grad.numpy(), rel=2e-7, tag="_f64")


@pytest.mark.parametrize("name", case_names(motif=True))
def test_oracle_motif_encoder_matches_reference(name):
    """MotifEncoder: oracle restatement vs the reference's outputs and gradients."""
    g = Golden(name)
    p = g.params(requires_grad=True)
    tree, _ = g.tensors()
    root, node = ref.motif_encoder_forward(p, g.rnn, g.depthT, tree)
    c = g.loss_coeffs([tuple(root.shape), tuple(node.shape)])
    loss = (torch.from_numpy(c[0]) * root).sum() + (torch.from_numpy(c[1]) * node).sum()
    loss.backward()
    assert rel_err(root.detach().numpy(), g.z["root"]) <= 2e-5
    assert rel_err(node.detach().numpy(), g.z["node"]) <= 2e-5
    for k, v in p.items():
        g.check_grad(k, v.grad.numpy(), rel=5e-5)
