"""CPU: pin the oracle (oracle/ref_encoder.py) and the layout restatement against the reference's outputs."""
import numpy as np
import pytest
import torch

from ggpm_amd import synth
from oracle import ref_encoder as ref
import os

from golden_utils import (ELEM_FLOOR, ELEM_TOL, GOLDEN_DIR, Golden, HeadsGolden, IncGolden, case_names, elem_rel_err,
                          heads_case_names, inc_case_names, rel_err, sparse_inputs)

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
        # per-element form of SURVEY section 8(d) (golden_utils.elem_rel_err): even with the reference's op order the
        # values next to a zero crossing of the final tanh differ by 1.6e-4 of themselves between two fp32 runs
        assert elem_rel_err(o.detach().numpy(), g.z[k], ELEM_FLOOR) <= ELEM_TOL, k
    assert rel_err(trace["atom"][0].detach().numpy(), g.z["atom_h1"]) <= tol
    assert rel_err(trace["atom"][-1].detach().numpy(), g.z["atom_hD"]) <= tol
    assert rel_err(trace["inter"][-1].detach().numpy(), g.z["inter_hD"]) <= tol
    assert rel_err(trace["tree"][-1].detach().numpy(), g.z["tree_hD"]) <= tol
    assert abs(float(kl.detach()) - float(g.z["kl"])) <= tol * max(1.0, abs(float(g.z["kl"])))
    assert abs(float(loss.detach()) - float(g.z["loss"])) <= 1e-5 * max(1.0, abs(float(g.z["loss"])))
    for k, v in p.items():
        grad = v.grad if v.grad is not None else torch.zeros_like(v)
        g.check_grad(k, grad.numpy(), rel=5e-5)


def test_per_element_criterion_is_calibrated():
    """SURVEY section 8(d) writes the parity bar per element with a floor of 1e-6 of the tensor's scale.  Pinned here on
    the reference's OWN two runs (fp32 and fp64 of the same code on the same inputs, both stored in the fixtures): in that
    measure the reference disagrees with itself by more than 1e-2 (near-zero ReLU outputs after 20 depths), with a
    floor of 1 % of the scale by a few 1e-4, and norm-wise by ~1e-5.  Hence ``golden_utils.assert_close``: norm-wise
    1e-4 plus the per-element form with ELEM_FLOOR / ELEM_TOL."""
    worst = {1e-6: 0.0, ELEM_FLOOR: 0.0, "norm": 0.0}
    for name in CASES:
        g = Golden(name)
        for k in ("hroot", "hnode", "hinter", "hatom"):
            a, b = g.z[k], g.z[k + "_f64"]
            for fl in (1e-6, ELEM_FLOOR):
                worst[fl] = max(worst[fl], elem_rel_err(a, b, fl))
            worst["norm"] = max(worst["norm"], rel_err(a, b))
    assert worst[1e-6] > 1e-2                # unattainable in fp32, by the reference itself
    assert 1e-4 < worst[ELEM_FLOOR] < ELEM_TOL
    assert worst["norm"] < 1e-4


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


def _sparse_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    E1, I, H, depth, ms, K, seed = [int(v) for v in z["meta"]]
    return z, str(z["rnn"]), (E1, I, H, depth, ms, K, seed)


@pytest.mark.parametrize("name", ["sparse_gru_s5", "sparse_lstm_s6", "sparse_gru_s7", "sparse_lstm_s8"])
def test_oracle_sparse_forward_matches_reference(name):
    """GRU/LSTM.sparse_forward restatement vs the reference's outputs and gradients."""
    from ggpm_amd.params import rnn_param_shapes, seeded_state_dict
    z, rnn, (E1, I, H, depth, ms, K, seed) = _sparse_case(name)
    h, c, submess, x, bg, coef = sparse_inputs(E1, I, H, ms, K, seed)
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in seeded_state_dict(rnn_param_shapes(rnn, I, H), seed).items()}
    ht, ct, xt = (torch.from_numpy(a).requires_grad_(True) for a in (h, c, x))
    sm, bgt = torch.from_numpy(submess), torch.from_numpy(bg)
    if rnn == "GRU":
        ho = ref.gru_sparse_forward(p, "", ht, xt, sm, bgt, depth)
        loss = (torch.from_numpy(coef[0]) * ho).sum()
    else:
        ho, co = ref.lstm_sparse_forward(p, "", ht, ct, xt, sm, bgt, depth)
        loss = (torch.from_numpy(coef[0]) * ho).sum() + (torch.from_numpy(coef[1]) * co).sum()
        assert rel_err(co.detach().numpy(), z["c_out"]) <= 2e-5
    loss.backward()
    assert rel_err(ho.detach().numpy(), z["h_out"]) <= 2e-5
    assert rel_err(ht.grad.numpy(), z["dh_in"]) <= 5e-5
    assert rel_err(xt.grad.numpy(), z["dx"]) <= 5e-5
    if rnn == "LSTM":
        assert rel_err(ct.grad.numpy(), z["dc_in"]) <= 5e-5
    for k, v in p.items():
        assert rel_err(v.grad.numpy(), z["grad/" + k]) <= 5e-5, k


# ---------------------------------------------------------------- incremental encoders
@pytest.mark.parametrize("name", inc_case_names())
def test_oracle_incremental_encoder_matches_reference(name):
    g = IncGolden(name)
    dtype = torch.float64
    p = g.params(dtype=dtype, requires_grad=True)
    tree, graph = g.tensors()
    init = g.init_vecs(dtype=dtype)
    outs, dec_tensors = ref.inc_teacher_forced(p, g.kind, g.rnn, g.depthT, g.depthG, tree, graph, init, g.schedule())
    assert (dec_tensors[2].numpy() == g.z["dec_agraph"]).all() and (dec_tensors[3].numpy() == g.z["dec_bgraph"]).all()
    keys = g.output_keys()
    coeffs = g.loss_coeffs([tuple(outs[k].shape) for k in keys])
    loss = sum((torch.from_numpy(c).to(dtype) * outs[k]).sum() for c, k in zip(coeffs, keys))
    loss.backward()
    for k in keys:
        assert rel_err(outs[k].detach().numpy(), g.z[k]) < 2e-5, k
    assert abs(float(loss.detach()) - float(g.z["loss"])) < 2e-4 * max(1.0, abs(float(g.z["loss"])))
    assert rel_err(init.grad.numpy(), g.z["d_init_vecs"]) < 1e-4
    for k, t in p.items():
        want = g.z["grad/" + k]
        got = t.grad.numpy() if t.grad is not None else np.zeros_like(want)
        assert rel_err(got, want) < 1e-4, k


# ---------------------------------------------------------------- host -> device input path
def test_prefetcher_layout_matches_make_cuda_on_cpu():
    from ggpm_amd.dataloader import DevicePrefetcher, pack_batch, unpack_views
    specs = [synth.random_batch(s, 3, motifs=(2, 5), n_motif_vocab=11, n_attach_vocab=33) for s in (1, 2, 3)]
    host = [synth.tensorize(b) for b in specs]
    out = list(DevicePrefetcher(host, device="cpu", depth=2))
    assert len(out) == 3
    for (tree, graph), (htree, hgraph) in zip(out, host):
        for a, b in zip(tree[:5], htree[:5]):
            assert a.dtype == torch.int64 and tuple(a.shape) == tuple(b.shape) and (a.numpy() == b).all()
        for a, b in zip(graph[:4], hgraph[:4]):
            assert a.dtype == torch.int64 and tuple(a.shape) == tuple(b.shape) and (a.numpy() == b).all()
        assert tree[-1] == htree[-1] and graph[-1] == hgraph[-1]
    flat, layout, ts, gs = pack_batch(host[0])
    assert all(off % 2 == 0 for off, _ in layout)


# ---------------------------------------------------------------- decoder score heads + losses
@pytest.mark.parametrize("name", heads_case_names())
def test_oracle_score_heads_match_reference(name):
    g = HeadsGolden(name)
    dtype = torch.float64
    p = g.params(dtype=dtype, requires_grad=True)
    fl, ix = g.inputs(dtype=dtype)
    mask = ref.vocab_mask(g.n_motif, g.n_attach, torch.from_numpy(g.z["owner"]), dtype)
    out = ref.score_heads(p, mask, fl["src_tree_vecs"], fl["src_graph_vecs"], fl["topo_vecs"], ix["topo_idx"],
                          ix["topo_labels"], fl["cls_vecs"], ix["cls_idx"], ix["cls_labs"], ix["icls_labs"],
                          fl["assm_vecs"], ix["assm_idx"], ix["assm_labels"], g.B)
    out["loss"].backward()
    for k in ("topo", "cls", "icls", "assm"):
        assert rel_err(out[k].detach().numpy(), g.z[k]) < 2e-5, k
    assert abs(float(out["loss"].detach()) - float(g.z["loss"])) < 2e-5 * abs(float(g.z["loss"]))
    for k, t in fl.items():
        assert rel_err(t.grad.numpy(), g.z["din/" + k]) < 1e-4, k
    for k, t in p.items():
        if t.grad is not None:
            g.check_grad(k, t.grad.numpy(), 1e-4)


def test_tree_chain_length_matches_the_fixed_point_of_the_oracle():
    """nnutils.tree_chain_length: after `chain` steps of the reference recurrence no tree message changes any more (and
    it still changes at step `chain`), a table with a cycle reports 0."""
    import numpy as np
    from ggpm_amd import synth
    from ggpm_amd.nnutils import tree_chain_length
    for seed, motifs in ((0, (2, 6)), (1, (8, 12)), (2, (1, 1)), (3, (1, 3))):
        specs = synth.random_batch(seed, 5, motifs=motifs, n_motif_vocab=11, n_attach_vocab=33)
        tree, _ = synth.tensorize(specs)
        bg = np.asarray(tree[3])
        chain = tree_chain_length(bg)
        # brute force: propagate "generation" sets -- message values as random hashes of (own id, predecessor values)
        rs = np.random.RandomState(seed)
        x = rs.standard_normal(bg.shape[0])
        h = np.zeros(bg.shape[0])
        last_change = 0
        for t in range(1, bg.shape[0] + 3):
            nh = np.tanh(x + (h[bg] * (bg > 0)).sum(axis=1))
            nh[0] = 0
            if not np.array_equal(nh, h):
                last_change = t
            h = nh
        assert chain == last_change, (seed, chain, last_change)
    cyc = np.array([[0, 0], [2, 0], [1, 0]])          # 1 <- 2 <- 1
    assert tree_chain_length(cyc) == 0
    import torch
    assert tree_chain_length(torch.from_numpy(bg)) == chain


# ------------------------------------------------------------------------------------------ full VAE step (R1)
from golden_utils import VaeGolden, vae_case_names  # noqa: E402


@pytest.mark.parametrize("name", vae_case_names())
def test_decode_schedule_matches_reference_bookkeeping(name):
    """ggpm_amd.decoder.DecodeSchedule (host integer plumbing of the product) against the index lists the reference's
    own decoder loop produced (ggpm/decoder.py:186-259), captured by make_golden_vae.py where the reference hands them
    to IncHierMPNEncoder / zip_tensors."""
    from ggpm_amd.decoder import DecodeSchedule
    g = VaeGolden(name)
    specs = g.specs()
    tensors = synth.tensorize(specs)
    for a, b in zip(tensors[0][:-1], g.numpy_tensors()[0][:-1]):
        assert (a == b).all()
    sch = DecodeSchedule.from_specs(specs, tensors)
    ref = g.ref_steps()
    assert len(sch.steps) == len(ref)
    for st, (sn, sm, at, bo) in zip(sch.steps, ref):
        assert st["subnode"] == sn and st["submess"] == sm
        assert sorted(st["atoms"]) == sorted(at) and sorted(st["bonds"]) == sorted(bo)
    tb, tl = sch.topo()
    assert tb == g.z["ref_topo_batch"].tolist() and tl == g.z["ref_topo_label"].tolist()
    cb, cc, ci = sch.cls()
    assert cb == g.z["ref_cls_batch"].tolist() and cc == g.z["ref_cls_clab"].tolist() and ci == g.z["ref_cls_ilab"].tolist()
    assert sch.assm_batch() == g.z["ref_assm_batch"].tolist()


def test_decode_schedule_from_networkx_batch():
    """The drop-in entry (``graphs`` as MolGraph.tensorize returns them): node attributes -> the same schedule."""
    import networkx as nx
    from ggpm_amd.decoder import DecodeSchedule, synth_orders
    from ggpm_amd.vocab import IndexPairVocab
    g = VaeGolden("vae_gru_s40")
    specs = g.specs()
    tensors = synth.tensorize(specs)
    tree = nx.DiGraph()
    for b, m in enumerate(specs):
        toff, aoff = tensors[0][-1][b][0], tensors[1][-1][b][0]
        for i in range(m.n_motifs):
            tree.add_node(toff + i, smiles="m%d" % m.motif_label[i][0],
                          inter_label=[(a + aoff, "a%d" % att) for a, att in m.inter_label[i]],
                          assm_cands=[x + aoff for x in m.assm_cands[i]])
    orders = synth_orders(specs, tensors[0][-1])
    a = DecodeSchedule.from_graphs((tree, None), tensors, orders, IndexPairVocab(g.n_motif, g.n_attach))
    b = DecodeSchedule.from_specs(specs, tensors)
    assert len(a.steps) == len(b.steps)
    for x, y in zip(a.steps, b.steps):
        assert all(x[k] == y[k] for k in x if k != "assm")
        assert [(c.tolist(), i, n, bi) for c, i, n, bi in x["assm"]] == [(c.tolist(), i, n, bi) for c, i, n, bi in y["assm"]]


def test_atom_plan_compact_row_sets_match_the_level_wide_tables():
    """AtomPlan's per-step compact row sets (host logic of ggpm_amd/atom_decode.py): mapped back through ``rows`` the local
    predecessor / successor CSRs and the local frozen mask are exactly the level-wide ones restricted to the step."""
    from ggpm_amd.decoder import DecodeSchedule
    specs = synth.random_batch(77, 6, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120)
    tensors = synth.tensorize(specs)
    sch = DecodeSchedule.from_specs(specs, tensors)
    E1 = tensors[1][1].shape[0]
    from ggpm_amd.atom_decode import AtomPlan
    plan = AtomPlan(sch, tensors[1][0].shape[0], E1, full=True)        # (the level-wide tables are what this test compares with)
    lean = AtomPlan(sch, tensors[1][0].shape[0], E1, full=False)       # what the compact form builds by default
    assert lean.nloc == plan.nloc and np.array_equal(lean.frozen_loc, plan.frozen_loc) and len(lean.ints) < len(plan.ints)
    assert np.array_equal(lean.compact_tables(5, 3)["ints"], plan.compact_tables(5, 3)["ints"])

    def arr(key):
        off, n = plan.where[key]
        return plan.ints[off:off + n].astype(np.int64)

    assert plan.T == len(sch.steps) and len(plan.nloc) == plan.T
    for t in range(plan.T):
        rows, n = arr(("rows", t)), plan.nloc[t]
        assert len(rows) == n and rows[0] == 0 and np.all(np.diff(rows) > 0)
        fl = plan.frozen_loc[plan.floc_off[t]:plan.floc_off[t] + n]
        assert np.array_equal(fl, plan.frozen[t][rows])                       # same mask on the rows of the set
        assert set(np.nonzero(plan.frozen[t] == 0)[0]) <= set(rows.tolist())  # every recomputed row is in the set
        for loc, glob in (("lpred", "pred"), ("lsucc", "succ")):
            rp, col = arr((loc + "_rp", t)), arr((loc + "_col", t))
            grp, gcol = arr((glob + "_rp", t)), arr((glob + "_col", t))
            assert len(rp) == n + 1 and len(grp) == E1 + 1
            for i in range(n):
                assert rows[col[rp[i]:rp[i + 1]]].tolist() == gcol[grp[rows[i]]:grp[rows[i] + 1]].tolist()
            inside = np.zeros(E1, dtype=bool)
            inside[rows] = True
            assert all(grp[r + 1] == grp[r] for r in np.nonzero(~inside)[0])  # rows outside the set have no entries
        gr = plan.gate_rows(t, 3)
        assert np.array_equal(gr, np.concatenate([k * E1 + rows for k in range(3)]))

    # compact_tables: "the state of message r at time t" resolved on the host.  Emulate the level-wide loop with random
    # states (a step overwrites the rows it recomputes) and compare what the tables read with what the loop would read.
    depth, G = 5, 3
    ct = plan.compact_tables(depth, G)

    def tab(key):
        off, n = ct["where"][key]
        return ct["ints"][off:off + n].astype(np.int64)

    rng = np.random.default_rng(0)
    foff, Ftot = ct["foff"], ct["Ftot"]
    assert Ftot == sum(plan.nloc)
    Fval = rng.standard_normal(Ftot)                       # final state of (step, local row); only live rows are ever read
    state = np.zeros(E1)
    nei_loop = np.zeros(plan.aoff[-1])
    n = np.asarray(plan.nloc)
    for t in range(plan.T):
        rows = arr(("rows", t))
        fl = plan.frozen_loc[plan.floc_off[t]:plan.floc_off[t] + n[t]]
        srcF, srcH = tab(("srcF", t)), tab(("srcH", t))
        want_in = np.where(fl == 1, state[rows], 0.0)       # frozen rows enter with the level-wide state, the others with 0
        got_in = np.where(srcF >= 0, Fval[np.maximum(srcF, 0)], 0.0)
        assert np.array_equal(np.where(fl == 1, got_in, 0.0), want_in)
        assert np.all(srcF[fl == 0] == -1)
        st = np.searchsorted(foff, srcF[srcF >= 0], side="right") - 1          # F id -> row of the stacked state blocks
        assert np.array_equal(srcH[srcF >= 0], (depth + 1) * np.asarray(foff)[st] + depth * n[st] + srcF[srcF >= 0] - np.asarray(foff)[st])
        live = np.nonzero(fl == 0)[0]
        state[rows[live]] = Fval[foff[t] + live]
        grp, gcol = arr(("agr_rp", t)), arr(("agr_col", t))                   # per-step table of the full-level form
        for a in range(plan.aoff[t + 1] - plan.aoff[t]):
            nei_loop[plan.aoff[t] + a] = state[gcol[grp[a]:grp[a + 1]]].sum()
    rpT, colT = tab("agrT_rp"), tab("agrT_col")
    nei_tab = np.zeros(plan.aoff[-1])
    for f in range(Ftot):
        nei_tab[colT[rpT[f]:rpT[f + 1]]] += Fval[f]
    assert np.allclose(nei_tab, nei_loop, rtol=0, atol=1e-12)
    xr = tab("xrows")
    assert np.array_equal(xr, np.concatenate([plan.gate_rows(t, G) for t in range(plan.T)]))
    rpT, colT = tab("xT_rp"), tab("xT_col")
    assert len(rpT) == G * E1 + 1 and all(np.all(xr[colT[rpT[r]:rpT[r + 1]]] == r) for r in range(G * E1))


def test_decode_schedule_with_its_plans_survives_pickling():
    """The decoder's host bookkeeping (DecodeSchedule + AtomPlan + compact tables, ~40 ms of numpy per batch of 32) is
    meant to be built by data-loader workers: it must cross a process boundary unchanged."""
    import pickle
    from ggpm_amd.decoder import DecodeSchedule
    specs = synth.random_batch(5, 4, motifs=(2, 6), n_motif_vocab=30, n_attach_vocab=90)
    tensors = synth.tensorize(specs)
    sch = DecodeSchedule.from_specs(specs, tensors)
    plan = sch.atom_plan(tensors[1][0].shape[0], tensors[1][1].shape[0])
    ct = plan.compact_tables(5, 4)
    back = pickle.loads(pickle.dumps(sch))
    plan2 = back.atom_plan(tensors[1][0].shape[0], tensors[1][1].shape[0])
    assert plan2 is back._atom_plan and plan2.T == plan.T and plan2.nloc == plan.nloc
    assert np.array_equal(plan2.ints, plan.ints) and np.array_equal(plan2.frozen_loc, plan.frozen_loc)
    ct2 = plan2.compact_tables(5, 4)
    assert np.array_equal(ct2["ints"], ct["ints"]) and ct2["where"] == ct["where"] and ct2["foff"] == ct["foff"]
    assert all(np.array_equal(back.plan[k], v) if isinstance(v, np.ndarray) else True for k, v in sch.plan.items())


@pytest.mark.parametrize("name", vae_case_names())
def test_oracle_vae_step_matches_reference(name):
    """oracle/ref_decoder.py (HierPropertyVAE.forward: encoder, rsample, teacher-forced decoder with enum_attach, the
    four losses) against the reference's loss, KL, metric tuple and parameter gradients."""
    from ggpm_amd.decoder import DecodeSchedule
    from ggpm_amd.vocab import IndexPairVocab
    from oracle import ref_decoder as refd
    g = VaeGolden(name)
    specs = g.specs()
    tensors = synth.tensorize(specs)
    sch = DecodeSchedule.from_specs(specs, tensors)
    sd = g.state_dict()
    p = {}
    for k, v in sd.items():
        p[k] = torch.from_numpy(v).requires_grad_(True)
    if g.tie:
        for k in ("E_c.0.weight", "E_i.0.weight"):
            p["encoder." + k] = p["decoder.hmpn." + k]
    tt, gt = ref.to_long_tensors(tensors[0]), ref.to_long_tensors(tensors[1])
    mask = IndexPairVocab(g.n_motif, g.n_attach).mask
    loss, kl, accs, recon = refd.vae_forward(p, g.rnn, g.depthT, g.depthG, g.diterT, g.diterG, tt, gt, sch, mask, g.beta)
    loss.backward()
    assert abs(float(loss) - float(g.z["loss"])) <= 2e-5 * abs(float(g.z["loss"]))
    assert abs(float(kl) - float(g.z["kl"])) <= 2e-5 * max(1.0, abs(float(g.z["kl"])))
    assert np.allclose([float(a) for a in accs], g.z["metrics"], atol=1e-6)
    seen = set()
    for k, v in p.items():
        if id(v) in seen:
            continue
        seen.add(id(v))
        key = k if not (g.tie and k.startswith("encoder.E_")) else "decoder.hmpn." + k[len("encoder."):]
        grad = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), np.float32)
        g.check_grad(_ref_param_name(key, g), grad, rel=1e-4)


def _ref_param_name(key, g):
    """named_parameters() of the reference lists a shared tensor under the FIRST name it meets: the encoder's for tied
    embeddings, ``decoder.hmpn.tree_encoder.rnn`` before the alias ``decoder.rnn_cell``."""
    if g.tie and key.startswith("decoder.hmpn.E_"):
        return "encoder." + key[len("decoder.hmpn."):]
    return key


def _attach_names():
    import glob
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "attach_*.npz")))


@pytest.mark.parametrize("name", _attach_names())
def test_oracle_enum_attach_matches_reference(name):
    """oracle/ref_decoder.enum_attach against the reference's HierMPNDecoder.enum_attach (single atoms and atom pairs)."""
    from ggpm_amd.params import seeded_state_dict
    from oracle import ref_decoder as refd
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    H, L, n_motif, n_attach, n_atoms, n_cands, k, nth, seed = [int(v) for v in z["meta"]]
    sd = seeded_state_dict({"matchNN.0.weight": (H, 2 * H + 20), "matchNN.0.bias": (H,), "hmpn.E_i.0.weight": (n_attach, H)},
                           seed)
    p = {kk: torch.from_numpy(v).requires_grad_(True) for kk, v in sd.items()}
    node = torch.from_numpy(z["node"]).requires_grad_(True)
    out = refd.enum_attach(p, node, torch.from_numpy(z["cands"]), z["icls"].tolist(), nth)
    (torch.from_numpy(z["coef"]) * out).sum().backward()
    assert rel_err(out.detach().numpy(), z["out"]) < 2e-6
    assert rel_err(node.grad.numpy(), z["d_node"]) < 2e-6
    for kk in p:
        assert rel_err(p[kk].grad.numpy(), z["grad/" + kk]) < 2e-6, kk


@pytest.mark.parametrize("name", ["tiny_gru_s1", "tiny_lstm_s2", "cfg_gru_s0", "cfg_lstm_s2"])
def test_bf16_cells_of_the_oracle_are_the_reference_cells_up_to_the_rounding(name, monkeypatch):
    """oracle/ref_encoder.py ``gate_dtype="bf16"`` restates the GRU / LSTM cells with the recurrent products applied per
    message and the hidden halves split off, so that operands can be rounded where the kernels round them.  With the
    rounding replaced by the identity that restatement must reproduce the reference's outputs and gradients (fixtures)
    like the plain oracle does -- which pins the bf16 oracle's algebra to the reference; and with the rounding on it must
    move the result by a bf16-sized amount, no more."""
    g = Golden(name)
    tree, graph = g.tensors()

    def run(mode):
        p = g.params(requires_grad=True)
        outs = ref.hier_encoder_forward(p, g.rnn, g.depthT, g.depthG, tree, graph, gate_dtype=mode)
        _, kl = ref.rsample_kl(p, outs[0])
        loss = g.beta * kl
        for c, o in zip(g.loss_coeffs([tuple(o.shape) for o in outs]), outs):
            loss = loss + (torch.from_numpy(c) * o).sum()
        loss.backward()
        return outs, p

    monkeypatch.setattr(ref, "rne_bf16", lambda t: t)
    for mode in ("bf16w", "bf16s"):       # ("bf16s": the storage roundings of ref._Round are identities too)
        outs, p = run(mode)
        for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs):
            assert rel_err(o.detach().numpy(), g.z[k]) <= 2e-5, (mode, k)
        for k, v in p.items():
            g.check_grad(k, (v.grad if v.grad is not None else torch.zeros_like(v)).numpy(), rel=1e-4)
    monkeypatch.undo()
    outs_b, _ = run("bf16")
    shift = max(rel_err(o.detach().numpy(), g.z[k]) for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs_b))
    assert 1e-6 < shift < 0.3, shift      # (GRU at depth 20 amplifies a 2^-9 operand error to ~0.1: sum aggregation)
    outs_s, _ = run("bf16s")
    shift_s = max(rel_err(o.detach().numpy(), g.z[k]) for k, o in zip(("hroot", "hnode", "hinter", "hatom"), outs_s))
    assert 1e-6 < shift_s < 0.5, shift_s



def test_oracle_runs_side_by_side_equal_the_in_process_runs():
    """golden_utils.OracleRuns (one CPU process per oracle evaluation; the full-size GPU parity cases use it): the same
    numbers as the in-process evaluation with the same thread count -- fp32, fp64 and a reversed-slot order."""
    import torch
    from ggpm_amd import synth
    from ggpm_amd.params import encoder_param_shapes, vae_head_shapes, seeded_state_dict
    import golden_utils as G
    specs = synth.random_batch(12, 3, motifs=(2, 5), n_motif_vocab=11, n_attach_vocab=33)
    tree, graph = synth.tensorize(specs)
    sd = seeded_state_dict(encoder_param_shapes("GRU", 24, 11, 33), 5)
    sd.update(seeded_state_dict(vae_head_shapes(24, 16), 6))
    jobs = {k: dict(v, threads=2) for k, v in G.FP32_ORDER_BASES.items()}       # six jobs: more than OracleRuns.MAX_PROCS
    jobs["f64"] = {"dtype": "f64", "threads": 2}
    jobs["hoisted@1"] = dict(G.FP32_ORDERS["hoisted@1"])
    runs = G.OracleRuns("GRU", 3, sd, tree, graph, jobs)
    here = {"padded": G.oracle_encoder_result("GRU", 3, sd, tree, graph, threads=2),
            "f64": G.oracle_encoder_result("GRU", 3, sd, tree, graph, dtype=torch.float64, threads=2),
            "slots_reversed": G.oracle_encoder_result("GRU", 3, sd, *G.reversed_batch(tree, graph), threads=2)}
    got = runs.results(timeout=300)
    assert set(got) == set(jobs)
    for name in here:
        assert set(got[name]) == set(here[name])
        for k, v in here[name].items():
            np.testing.assert_allclose(np.asarray(got[name][k], dtype=np.float64), np.asarray(v, dtype=np.float64),
                                       rtol=1e-6, atol=1e-9, err_msg="%s %s" % (name, k))
