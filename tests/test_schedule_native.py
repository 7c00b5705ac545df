"""CPU: the decode schedule built by csrc/schedule.hip (host-only C++) against the numpy builders it restates --
``DecodeSchedule.from_tensors`` / ``_level_plan`` (ggpm_amd/decoder.py) and ``AtomPlan`` / ``compact_tables``
(ggpm_amd/atom_decode.py).  Every table, element for element."""
import pickle

import numpy as np
import pytest

from ggpm_amd import synth
from ggpm_amd.atom_decode import AtomPlan
from ggpm_amd.decoder import DecodeSchedule

CASES = {
    "configs1": dict(seed=1000, B=32, motifs=(8, 12), vocab=(500, 1500)),
    "qm9_like": dict(seed=7, B=64, motifs=(1, 3), vocab=(60, 180)),
    "one_molecule": dict(seed=3, B=1, motifs=(4, 9), vocab=(30, 90)),
    "small": dict(seed=5, B=4, motifs=(2, 6), vocab=(30, 90)),
    "polymers": dict(seed=606, B=3, motifs=(46, 58), vocab=(60, 180)),
    "mix": dict(seed=11, B=12, motifs="mix", vocab=(721, 6214)),
}
# a seeded sweep over ragged batches: batch sizes 1 ... 9, molecules from a single motif (no tree message at all) to 14
_rs = np.random.RandomState(2026)
for _i in range(16):
    _lo = int(_rs.randint(1, 4))
    CASES["sweep_%02d" % _i] = dict(seed=int(_rs.randint(1, 10 ** 6)), B=int(_rs.randint(1, 10)),
                                    motifs=(_lo, _lo + int(_rs.randint(0, 12))), vocab=(40, 120))


def _batch(c):
    if c["motifs"] == "mix":
        specs = synth.size_mix_batch(c["seed"], c["B"], one_of_each=True, n_motif_vocab=c["vocab"][0], n_attach_vocab=c["vocab"][1])
    else:
        specs = synth.random_batch(c["seed"], c["B"], motifs=c["motifs"], n_motif_vocab=c["vocab"][0], n_attach_vocab=c["vocab"][1])
    return specs, synth.tensorize(specs)


def _same(a, b, what):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and np.array_equal(a, b), what
    else:
        assert a == b, (what, a, b)


def _same_steps(sa, sb):
    assert len(sa) == len(sb)
    for t, (x, y) in enumerate(zip(sa, sb)):
        for k in x:
            if k != "assm":
                assert list(x[k]) == list(y[k]), (t, k)
        assert len(x["assm"]) == len(y["assm"]), t
        for (c1, i1, n1, b1), (c2, i2, n2, b2) in zip(x["assm"], y["assm"]):
            assert np.array_equal(c1, c2) and tuple(i1) == tuple(i2) and n1 == n2 and b1 == b2, t


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("depth,gates", [(5, 3), (2, 4)])
def test_native_schedule_equals_the_numpy_builders(case, depth, gates):
    specs, tensors = _batch(CASES[case])
    nat = DecodeSchedule.from_specs(specs, tensors, depth=depth, gates=gates, native=True)
    ref = DecodeSchedule.from_specs(specs, tensors, native=False)
    if ref.plan["E1"] <= 1 or not ref.plan["all_live"]:
        assert nat._native is None            # degenerate batches take the numpy builder and the step-by-step forms
        return
    assert nat._native is not None and ref._native is None
    assert (nat.batch_size, nat.max_cls_size, nat.root_clab, nat.root_ilab) == (ref.batch_size, ref.max_cls_size,
                                                                                 ref.root_clab, ref.root_ilab)
    assert set(nat.plan) == set(ref.plan)
    for k in ref.plan:
        _same(nat.plan[k], ref.plan[k], "plan." + k)
    _same_steps(nat.steps, ref.steps)
    assert nat.topo() == ref.topo() and nat.cls() == ref.cls() and nat.assm_batch() == ref.assm_batch()
    # the int32 molecule indices the heads gather by
    tb, _ = ref.topo()
    cb, _, _ = ref.cls()
    _same(nat._native.get("topo_batch32"), np.asarray(tb, dtype=np.int32), "topo_batch32")
    _same(nat._native.get("cls_batch32"), np.asarray(cb, dtype=np.int32), "cls_batch32")
    _same(nat._native.get("assm_batch32"), np.repeat(np.asarray(ref.assm_batch(), dtype=np.int32), ref.max_cls_size), "assm_batch32")

    N1, E1 = tensors[1][0].shape[0], tensors[1][1].shape[0]
    pa, pb = nat.atom_plan(N1, E1), AtomPlan(ref, N1, E1, full=False)
    for k in ("T", "N1", "E1", "ok", "nloc", "floc_off", "n_cand", "full"):
        _same(getattr(pa, k), getattr(pb, k), "AtomPlan." + k)
    assert list(pa.aoff) == list(pb.aoff) and list(pa.ioff) == list(pb.ioff)
    assert [tuple(x) for x in pa.cand_blocks] == [tuple(x) for x in pb.cand_blocks]
    assert [[tuple(s) for s in st] for st in pa.step_cands] == [[tuple(s) for s in st] for st in pb.step_cands]
    _same(pa.frozen_loc, pb.frozen_loc, "frozen_loc")
    assert set(pa.where) == set(pb.where)
    for key, (off, n) in pb.where.items():
        o2, n2 = pa.where[key]
        assert n2 == n, key
        _same(pa.ints[o2:o2 + n2], pb.ints[off:off + n], key)
    assert set(pa.cand_meta) == set(pb.cand_meta)
    for k, m in pb.cand_meta.items():
        for name, v in m.items():
            _same(np.asarray(pa.cand_meta[k][name], dtype=np.int64), v, ("cand_meta", k, name))
    for dg in ((depth, gates), (3, 3)):        # the pair the build was given (native tables) and another one (numpy over the native rows)
        ca, cb_ = pa.compact_tables(*dg), pb.compact_tables(*dg)
        assert ca["foff"] == cb_["foff"] and ca["Ftot"] == cb_["Ftot"] and set(ca["where"]) == set(cb_["where"])
        assert bool(ca.get("native")) == (dg == (depth, gates))
        for key, (off, n) in cb_["where"].items():
            o2, n2 = ca["where"][key]
            assert n2 == n, key
            _same(ca["ints"][o2:o2 + n2], cb_["ints"][off:off + n], ("compact", dg, key))
    assert pa.row_offsets(depth) == pb.row_offsets(depth)


def _csr_of_padded(table, first_row, total_rows):
    """what ggpm_padded_to_csr makes of ``table`` placed at rows first_row.. of a [total_rows, width] zero table"""
    rp, col = np.zeros(total_rows + 1, dtype=np.int64), []
    for r, row in enumerate(np.asarray(table)):
        nz = [int(v) for v in row if v != 0]
        col.extend(nz)
        rp[first_row + r + 1] = len(nz)
    return np.cumsum(rp), np.asarray(col, dtype=np.int64)


def _transpose(rp, col, ncols):
    """ggpm_csr_transpose: per column the rows that name it, ascending"""
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    order = np.lexsort((rows, col))
    return np.concatenate([[0], np.cumsum(np.bincount(col, minlength=ncols))]), rows[order]


@pytest.mark.parametrize("case", sorted(CASES))
def test_native_level_and_head_index_structures(case):
    """The CSRs, transposes and frozen masks of the two tree-side levels and the transposes of the heads' molecule indices
    that csrc/schedule.hip puts into the upload, against the statement of what the device kernels derive from the NUMPY
    builder's tables (padded -> CSR keeps the in-row order of the non-zero entries; a transpose lists rows ascending)."""
    specs, tensors = _batch(CASES[case])
    nat = DecodeSchedule.from_specs(specs, tensors, depth=5, gates=3, native=True)
    ref = DecodeSchedule.from_specs(specs, tensors, native=False)
    if nat._native is None:
        return
    g, P = nat._native.get, ref.plan
    E1, n_inst, B = P["E1"], P["n_inst"], ref.batch_size

    def same(name, exp, pad=False):
        got = np.asarray(g(name), dtype=np.int64)
        if pad:                                            # (col tables carry one trailing 0 so that none is empty)
            assert got[-1] == 0
            got = got[:-1]
        assert got.shape == np.asarray(exp).shape and np.array_equal(got, exp), (case, name)

    for tag, n_extra in (("inter", 0), ("tree", B)):
        Etot = E1 + n_extra
        rp, col = _csr_of_padded(P["dag_" + tag], 1, Etot)
        same("pred_%s_rp" % tag, rp)
        same("pred_%s_col" % tag, col, pad=True)
        rpT, colT = _transpose(rp, col, Etot)
        same("succ_%s_rp" % tag, rpT)
        same("succ_%s_col" % tag, colT, pad=True)
        rp, col = _csr_of_padded(P["in_" + tag], 0, n_inst)
        same("in_%s_rp" % tag, rp)
        same("in_%s_col" % tag, col, pad=True)
        rpT, colT = _transpose(rp, col, Etot)
        same("inT_%s_rp" % tag, rpT)
        same("inT_%s_col" % tag, colT, pad=True)
        frozen = np.ones(Etot, dtype=np.int64)
        frozen[1:E1] = 0
        same("frozen_" + tag, frozen)
    heads = {"srcT": (np.asarray(P["mess_inst"], dtype=np.int64), n_inst), "topoT": (np.asarray(ref.topo()[0], dtype=np.int64), B),
             "clsT": (np.asarray(ref.cls()[0], dtype=np.int64), B),
             "assmT": (np.repeat(np.asarray(ref.assm_batch(), dtype=np.int64), ref.max_cls_size), B)}
    longest = 0
    for name, (idx, ncols) in heads.items():
        rpT, colT = _transpose(np.arange(len(idx) + 1), idx, ncols)
        same(name + "_rp", rpT)
        same(name + "_col", colT, pad=True)
        longest = max(longest, len(idx))
    same("iota", np.arange(longest + 1))


def test_native_schedule_survives_pickling():
    """Loader workers build schedules and ship them: a natively built one converts its tables to plain arrays."""
    specs, tensors = _batch(CASES["small"])
    nat = DecodeSchedule.from_specs(specs, tensors, depth=5, gates=4)
    assert nat._native is not None
    back = pickle.loads(pickle.dumps(nat))
    for k in nat.plan:
        _same(back.plan[k], nat.plan[k], k)
    _same_steps(back.steps, nat.steps)
    N1, E1 = tensors[1][0].shape[0], tensors[1][1].shape[0]
    pa, pb = back.atom_plan(N1, E1), nat.atom_plan(N1, E1)
    _same(pa.ints, pb.ints, "ints")
    assert pa.where == pb.where and pa.nloc == pb.nloc
    ca, cb_ = pa.compact_tables(5, 4), pb.compact_tables(5, 4)
    assert ca["where"] == cb_["where"]
    _same(back._native.packs[1], nat._native.packs[1], "pack64")
    _same(back._native.packs[2], nat._native.packs[2], "pack32")


@pytest.mark.parametrize("case", ["configs1", "small", "one_molecule", "mix", "sweep_03", "sweep_11"])
def test_schedule_from_the_networkx_batch_equals_the_one_from_dictionaries(case):
    """``from_graphs`` reads the labels off the networkx nodes in one pass straight into the flat arrays of the C++ builder
    (schedule_native.labels_from_graph); ``from_specs`` goes through the per-node dictionaries.  Same packs, same labels (the
    lazily rebuilt per-step lists read them), and the label views survive pickling."""
    from ggpm_amd import schedule_native as SN
    from ggpm_amd.vocab import IndexPairVocab
    c = CASES[case]
    specs, tensors = _batch(c)
    nm, na = c["vocab"]
    vocab = IndexPairVocab(nm, na, owner=None if na % nm == 0 else np.arange(na) % nm)
    b6 = synth.train_batch(specs, tensors)
    a = DecodeSchedule.from_graphs(b6[1], b6[2], b6[3], vocab, depth=5, gates=3)
    b = DecodeSchedule.from_specs(specs, tensors, depth=5, gates=3)
    assert (a._native is None) == (b._native is None)
    if a._native is None:
        return
    flat = SN.labels_from_graph(b6[1][0], vocab, tensors[0][0].shape[0])
    assert flat is not None                                 # (the fast path was the one taken)
    assert a._native.names == b._native.names
    _same(a._native.packs[1], b._native.packs[1], "pack64")
    _same(a._native.packs[2], b._native.packs[2], "pack32")
    for name in a._native.names:
        _same(a._native.get(name), b._native.get(name), name)
    _same_steps(a.steps, b.steps)
    back = pickle.loads(pickle.dumps(DecodeSchedule.from_graphs(b6[1], b6[2], b6[3], vocab, depth=5, gates=3)))
    _same_steps(back.steps, b.steps)


def test_labels_from_graph_declines_what_it_does_not_understand():
    """nodes out of order, or a node whose candidates do not have one atom per attachment id: None (the dictionary path and
    the numpy builder then take the batch and raise the informative error)"""
    import networkx as nx
    from ggpm_amd import schedule_native as SN
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(30, 90)
    g = nx.DiGraph()
    g.add_node(1, smiles=0, inter_label=[(0, 1)], assm_cands=[3, 4])
    g.add_node(0, smiles=0, inter_label=[], assm_cands=[])
    assert SN.labels_from_graph(g, vocab, 2) is None        # 1 before 0
    g = nx.DiGraph()
    g.add_node(0, smiles=0, inter_label=[], assm_cands=[])
    g.add_node(1, smiles=0, inter_label=[(0, 1), (0, 2)], assm_cands=[(3,), (4,)])
    assert SN.labels_from_graph(g, vocab, 2) is None        # two attachment ids, one atom per candidate
    g = nx.DiGraph()
    g.add_node(0, smiles=0, inter_label=[], assm_cands=[])
    g.add_node(1, smiles=0, inter_label=[(0, 1), (0, 2)], assm_cands=[(3, 5), (4, 6)])
    ok = SN.labels_from_graph(g, vocab, 2)
    assert ok is not None and ok.cands.tolist() == [3, 5, 4, 6] and ok.cands_of(1).tolist() == [[3, 5], [4, 6]]
    assert ok.icls_of(1) == tuple(vocab[(0, k)][1] for k in (1, 2)) and ok.icls_of(0) == ()


def test_native_builder_refuses_tables_that_index_outside_themselves():
    from ggpm_amd import schedule_native as SN
    from ggpm_amd.decoder import synth_orders
    specs, tensors = _batch(CASES["small"])
    tree, graph = tensors
    orders = synth_orders(specs, tree[-1])
    icls, cands = {}, {}
    for b, m in enumerate(specs):                 # (the labels DecodeSchedule.from_specs reads)
        toff, aoff = tree[-1][b][0], graph[-1][b][0]
        for i in range(m.n_motifs):
            icls[toff + i] = tuple(a for _, a in m.inter_label[i])
            cands[toff + i] = [x + aoff for x in m.assm_cands[i]]
    assert SN.build_tables(tensors, orders, icls, cands) is not None
    bad = [list(o) for o in orders]
    bad[0][0] = (10 ** 6, bad[0][0][1], bad[0][0][2])            # a node id beyond the tree tensors
    assert SN.build_tables(tensors, bad, icls, cands) is None
    g2 = list(graph)
    g2[3] = np.array(graph[3]).copy()
    g2[3][1, 0] = 10 ** 6                                          # a predecessor id beyond the bond table
    assert SN.build_tables((tree, tuple(g2)), orders, icls, cands) is None


class _FakeDecoder:
    def __init__(self, vocab, depth, gates):
        self.vocab, self._h = vocab, dict(depth=depth, gates=gates)

    def schedule_hints(self):
        return dict(self._h)


class _FakeModel:
    def __init__(self, vocab, depth=4, gates=3):
        self.decoder = _FakeDecoder(vocab, depth, gates)


def test_schedule_ahead_yields_the_same_batches_with_their_schedules():
    """dataloader.ScheduleAhead: the loop's batches come out in order, unchanged but for ``graphs`` carrying the decode
    schedule the forward would otherwise build (same tables as a direct DecodeSchedule.from_graphs)."""
    from ggpm_amd.dataloader import ScheduleAhead, ScheduledGraphs
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    model = _FakeModel(vocab)
    dataset = [synth.train_batch(synth.random_batch(30 + i, 5, motifs=(2, 7), n_motif_vocab=40, n_attach_vocab=120))
               for i in range(4)]
    out = list(ScheduleAhead(dataset, model))
    assert len(out) == len(dataset) == len(ScheduleAhead(dataset, model))
    for got, want in zip(out, dataset):
        assert got[0] is want[0] and got[2] is want[2] and got[3] is want[3] and got[4] is want[4] and got[5] is want[5]
        assert isinstance(got[1], ScheduledGraphs) and got[1][0] is want[1][0] and got[1][1] is want[1][1]
        direct = DecodeSchedule.from_graphs(want[1], want[2], want[3], vocab, depth=4, gates=3)
        _same_steps(got[1].ggpm_schedule.steps, direct.steps)
    # a batch without graphs (the prepared-schedule call shape) passes through untouched
    bare = (None, None, dataset[0][2], dataset[0][3], None, None)
    assert list(ScheduleAhead([bare], model))[0] is bare


def test_schedule_ahead_raises_at_the_batch_that_is_malformed():
    from ggpm_amd.dataloader import ScheduleAhead
    from ggpm_amd.vocab import IndexPairVocab
    vocab = IndexPairVocab(40, 120)
    good = synth.train_batch(synth.random_batch(3, 4, motifs=(2, 6), n_motif_vocab=40, n_attach_vocab=120))
    orders = [list(o) for o in good[3]]
    orders[0][0] = (10 ** 6, orders[0][0][1], orders[0][0][2])                          # a node id beyond the tree tensors
    bad = (good[0], good[1], good[2], orders, good[4], good[5])
    it = iter(ScheduleAhead([good, bad, good], _FakeModel(vocab)))
    assert next(it)[1].ggpm_schedule is not None
    with pytest.raises(Exception):
        next(it)
