"""More than 255 thresholds per feature (k_wide.hip) against the oracle: QuickRank's
default `--num-thresholds 0` on real-valued columns -- every distinct value a
candidate, mart.cc:147-158, the setup of the reference's own forest tests
(catch-unit-tests/learning/forests/test-lambdamart.cc "all thresholds") -- and
--num-thresholds above 255 (the equal-width branch, mart.cc:159-169)."""
import numpy as np
import pytest

from datagen import make_dataset
from parity_util import assert_split_log_parity, assert_tree_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qr():
    import quickrank_amd
    from quickrank_amd import build
    build.build()
    return quickrank_amd


@pytest.fixture(scope="module")
def ora(oracle_lib):
    return oracle_lib


CASES = [
    dict(nq=40, docs_per_query=30, F=16, seed=0),
    dict(nq=25, docs_per_query=60, F=136, seed=1, ragged=True),
    dict(nq=30, docs_per_query=40, F=70, seed=2, adversarial=True),
    dict(nq=12, docs_per_query=300, F=33, seed=3, ragged=True, adversarial=True),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr", [0, 256, 1000, 5000])
def test_wide_thresholds_and_bins(qr, ora, case, nthr):
    x, labels, qoff = make_dataset(**case)
    col = np.ascontiguousarray(x.T)
    othr, ots = ora.thresholds(col, nthr)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    thr, ts = c.build_bins(nthr)
    assert c.wide
    assert np.array_equal(ts.astype(np.uint64), ots)
    for f in range(x.shape[1]):
        n = int(ots[f])
        assert np.array_equal(thr[f, :n].view(np.uint32), othr[f, :n].view(np.uint32)), f
    stmap, _ = ora.binmap(col, othr, ots)
    assert np.array_equal(c.read_bins_u32().T, stmap)
    c.close()


def test_wide_is_chosen_only_when_needed(qr):
    x, labels, qoff = make_dataset(nq=20, docs_per_query=30, F=8, seed=0)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    with pytest.raises(qr.QrError):      # 600 distinct values per column do not fit u8 bins ...
        c.build_bins(0, wide=False)
    c.build_bins(0)                      # ... so nthresholds = 0 takes the wide path
    assert c.wide
    c.close()
    c = qr.Context(0)
    c.upload(np.floor(x * 100) / 100, labels, qoff)
    c.build_bins(0)                      # <= 255 distinct values: the u8 path, as before
    assert not c.wide
    c.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr,nleaves,minls", [(0, 10, 1), (1000, 16, 5), (300, 4, 1)])
def test_wide_root_histogram_and_tree(qr, ora, case, nthr, nleaves, minls):
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(11)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(nthr)
    c.set_pseudo(olam, ow)
    nodes = c.fit_tree(nleaves, minls, True)
    tr = ora.Trainer(x, nthr)
    ot = tr.fit_tree(olam, nleaves=nleaves, minls=minls)
    tr.update_output(ot, olam, ow)
    hs, hc = c.node_hist_ragged(0)
    os_, oc, _ = ora.hist_build(tr.stmap, tr.thr_size, tr.cap, olam)
    tol = 2.0 ** -30 * max(1.0, np.abs(olam).max()) * np.sqrt(len(olam))
    for f in range(x.shape[1]):
        n = int(tr.thr_size[f])
        assert np.array_equal(hc[f], oc[f, :n]), f
        assert np.allclose(hs[f], os_[f, :n], rtol=0, atol=tol), f
    on = ot["nodes"]
    ties = assert_tree_parity(tr.stmap, on, nodes, value_rtol=1e-9)
    assert_split_log_parity(c.split_log(), ot["splits"], ties)
    for li, on_leaf in enumerate(ot["leaf_nodes"]):
        ids = c.node_samples(ties.node_map[int(on_leaf)])
        assert np.array_equal(ids, np.nonzero(ot["leaf_of_doc"] == li)[0].astype(np.uint32))
    c.set_scores(scores)
    c.update_scores(0.1)
    s2 = scores.copy()
    tr.update_scores(ot, 0.1, s2)
    assert np.allclose(c.get_scores(), s2, rtol=1e-12, atol=1e-13)
    c.close()


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
@pytest.mark.parametrize("case,nthr,nleaves", [(CASES[0], 0, 10), (CASES[1], 0, 10), (CASES[2], 1000, 8),
                                               (CASES[3], 0, 16)])
def test_wide_training_loop(qr, ora, algo, case, nthr, nleaves):
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(**case)
    ntrees = 8
    kw = dict(ntrees=ntrees, shrinkage=0.1, nthresholds=nthr, nleaves=nleaves, minls=1, esr=0)
    om = ora.train(x, labels, qoff, algo=algo, **kw)
    gm = Mart(algo=algo, **kw).learn(x, labels, qoff)
    assert gm.ctx.wide and len(gm.ensemble) == om["ntrees_built"]
    tr = ora.Trainer(x, nthr)
    # The adversarial sets carry a duplicated, a constant and two quantised columns: with
    # every distinct value a threshold, two features often cut a node into the same two
    # sets (MART's residuals are discrete on top), also in nodes of a few thousand
    # documents -- the ceiling is the whole set there, 1000 documents otherwise.
    cap = len(labels) if case.get("adversarial") else 1000
    for t in range(ntrees):
        n = int(om["nnodes"][t])
        assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n], tie_max_docs=cap)
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-9)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-8, atol=1e-10)
    gm.ctx.close()


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
def test_wide_training_loop_medium(qr, ora, algo):
    """The reference's default flags (`--num-thresholds 0`) on 20k documents x 136
    real-valued features: up to 20001 slots per feature.  Every split must cut its
    node into the oracle's two sets.  The SLOT cannot be asked for bit for bit here:
    with every distinct value a threshold, most slots of a node are empty, runs of
    adjacent slots induce one and the same partition, and on a sibling histogram
    (parent - child in f64, rtnode_histogram.cc:72-87) the reference tells them apart
    by the rounding noise of that subtraction -- the device's integer histograms make
    them exactly equal and it keeps the first.  Checked instead: a deviating split names
    the SAME feature, and (by the walker) induces the same partition."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=200, docs_per_query=100, F=136, seed=21)
    kw = dict(ntrees=5, shrinkage=0.1, nthresholds=0, nleaves=10, minls=50, esr=0)
    om = ora.train(x, labels, qoff, algo=algo, **kw)
    gm = Mart(algo=algo, **kw).learn(x, labels, qoff)
    tr = ora.Trainer(x, 0)
    for t in range(kw["ntrees"]):
        n = int(om["nnodes"][t])
        o, g = om["nodes"][t][:n], gm.ensemble.trees[t][:n]
        ties = assert_tree_parity(tr.stmap, o, g, tie_max_docs=len(labels))
        for oi in ties.onodes:
            gi = ties.node_map[oi]
            assert g[gi]["feature"] == o[oi]["feature"] and g[gi]["thr_id"] < o[oi]["thr_id"], (t, oi)
            assert oi not in ties.mirrored
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-10)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-9, atol=1e-11)
    gm.ctx.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr,depth,minls", [(0, 3, 1), (1000, 4, 5), (300, 6, 1)])
def test_wide_oblivious_tree(qr, ora, case, nthr, depth, minls):
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(13)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(nthr)
    c.set_pseudo(olam, ow)
    nodes = c.fit_oblivious(depth, minls, True)
    tr = ora.Trainer(x, nthr)
    ot = tr.fit_tree(olam, minls=minls, oblivious_depth=depth)
    tr.update_output(ot, olam, ow)
    on = ot["nodes"]
    assert len(nodes) == len(on)
    assert np.array_equal(nodes["feature"] == -2, on["feature"] == -2)
    ties = assert_tree_parity(tr.stmap, on, nodes, value_rtol=1e-9)
    log, olog = c.split_log(), ot["splits"]
    assert len(log) == len(olog)
    if ties == 0:
        assert np.array_equal(log["feature"].astype(np.uint64), olog["feature"])
        assert np.array_equal(log["thr_id"].astype(np.uint64), olog["thr_id"])
    assert np.allclose(log["score"], olog["score"], rtol=1e-9)
    c.close()


@pytest.mark.parametrize("algo", ["OBVLAMBDAMART", "OBVMART"])
def test_wide_oblivious_training_loop(qr, ora, algo):
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=120, docs_per_query=60, F=40, seed=23)
    kw = dict(ntrees=6, shrinkage=0.1, nthresholds=0, minls=10, esr=0)
    om = ora.train(x, labels, qoff, algo=algo, depth=4, **kw)
    gm = Mart(algo=algo, depth=4, **kw).learn(x, labels, qoff)
    assert gm.ctx.wide
    tr = ora.Trainer(x, 0)
    for t in range(kw["ntrees"]):
        n = int(om["nnodes"][t])
        assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n])
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-9)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-8, atol=1e-10)
    gm.ctx.close()


@pytest.mark.parametrize("case", [CASES[1], CASES[3]])
@pytest.mark.parametrize("nthr", [300, 1100])
def test_wide_fast_rows_equal_the_general_kernel(qr, ora, case, nthr, monkeypatch):
    """Rows of up to 1152 slots take k_whist16 (blocked u16 bins, [slot][16] LDS histogram);
    QR_WIDE_NO_FAST=1 sends the same context through the general k_whist.  Integer cells:
    every node histogram, every tree record and the scores must be the same bits."""
    x, labels, qoff = make_dataset(**case)

    def run():
        c = qr.Context(0)
        c.upload(x, labels, qoff)
        _, ts = c.build_bins(nthr)
        assert c.wide and int(ts.max()) <= 1152
        c.reset_scores()
        trees, hists = [], []
        for it in range(3):
            c.compute_lambdas("NDCG", 10)
            t = c.fit_tree(8, 2, True) if it < 2 else c.fit_oblivious(3, 2, True)
            trees.append(t)
            hists.append([c.node_hist_ragged(n) for n in range(min(len(t), 5))])
            c.update_scores(0.1)
        s = c.get_scores()
        c.close()
        return trees, hists, s

    monkeypatch.delenv("QR_WIDE_NO_FAST", raising=False)
    ta, ha, sa = run()
    monkeypatch.setenv("QR_WIDE_NO_FAST", "1")
    tb, hb, sb = run()
    for a, b in zip(ta, tb):
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    for la, lb in zip(ha, hb):
        for (s1, c1), (s2, c2) in zip(la, lb):
            for f in range(x.shape[1]):
                assert np.array_equal(c1[f], c2[f]) and np.array_equal(s1[f], s2[f]), f
    assert np.array_equal(sa, sb)


@pytest.mark.parametrize("case", [CASES[1], CASES[2]])
@pytest.mark.parametrize("nthr", [0, 1000, 5000])
def test_wide_batched_growth_equals_one_split_per_step(qr, case, nthr, monkeypatch):
    """Wide-bin contexts whose rows fit the one-launch scan grow leaf-wise trees two splits per
    step with the control step inside the partition launch (round 3), like the u8 path;
    QR_NO_BATCH=1 (read when the context is created) keeps one split per step.  Same trees --
    both take the child sums from the partition pass here, so every field bit for bit -- and
    the same scores."""
    x, labels, qoff = make_dataset(**case)

    def run():
        c = qr.Context(0)
        c.upload(x, labels, qoff)
        c.build_bins(nthr)
        assert c.wide
        c.reset_scores()
        trees = []
        for it in range(4):
            c.compute_lambdas("NDCG", 10)
            trees.append(c.fit_tree(12 if it < 3 else 40, 2, True))
            c.update_scores(0.1)
        s = c.get_scores()
        c.close()
        return trees, s

    monkeypatch.delenv("QR_NO_BATCH", raising=False)
    ta, sa = run()
    monkeypatch.setenv("QR_NO_BATCH", "1")
    tb, sb = run()
    for a, b in zip(ta, tb):
        assert len(a) == len(b)
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb)


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr,minls", [(0, 1), (0, 7), (1000, 2)])
def test_presorted_lists_equal_the_slot_histograms(qr, algo, case, nthr, minls, monkeypatch):
    """k_exact.hip (round 4): rows of more than 16384 slots -- the reference's default
    `--num-thresholds 0` on real-valued columns -- grow their trees on per-feature lists of the
    documents sorted by slot instead of slot-indexed node histograms.  Same exact integers, same
    first maximum in (feature, slot) order: forced on small sets (QR_WIDE_EXACT=1, read when the
    bins are built) every record of every tree and every score must be the bits of the
    histogram path (QR_WIDE_NO_EXACT=1, one split per step)."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(**case)
    kw = dict(ntrees=4, shrinkage=0.1, nthresholds=nthr, nleaves=12, minls=minls, esr=0)

    def run():
        m = Mart(algo=algo, **kw).learn(x, labels, qoff)
        assert m.ctx.wide
        out = [t.copy() for t in m.ensemble.trees], m.ctx.get_scores(), list(m.train_metric)
        m.ctx.close()
        return out

    monkeypatch.setenv("QR_NO_BATCH", "1")
    monkeypatch.setenv("QR_WIDE_NO_EXACT", "1")
    ta, sa, ma = run()
    monkeypatch.delenv("QR_WIDE_NO_EXACT")
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    tb, sb, mb = run()
    for a, b in zip(ta, tb):
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb) and ma == mb


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
@pytest.mark.parametrize("subsample", [0.5, 0.13])
def test_presorted_lists_under_subsample(qr, algo, subsample, monkeypatch):
    """--subsample on the pre-sorted lists (round 5; mart.cc:287-329): every feature's root list is
    cut down to the iteration's sample, and the tree grows on it.  The same seeded draws on the
    slot-indexed path (QR_WIDE_NO_EXACT=1, what a sampled wide context fell back to in round 4) must
    give the same records and scores, bit for bit -- and the lists must really be what ran (a
    context on them has no node histograms to read)."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(**CASES[1])
    kw = dict(ntrees=4, shrinkage=0.1, nthresholds=0, nleaves=10, minls=2, esr=0, subsample=subsample, seed=11)

    def run(expect_lists):
        m = Mart(algo=algo, **kw).learn(x, labels, qoff)
        assert m.ctx.wide
        k = int(np.floor(np.float32(subsample) * np.float32(len(labels))))
        assert all(t[0]["nsamples"] == k for t in m.ensemble.trees)
        if expect_lists:
            with pytest.raises(Exception, match="pre-sorted"):
                m.ctx.node_hist_ragged(0)
        else:
            m.ctx.node_hist_ragged(0)
        out = [t.copy() for t in m.ensemble.trees], m.ctx.get_scores(), list(m.train_metric)
        m.ctx.close()
        return out

    monkeypatch.setenv("QR_NO_BATCH", "1")
    monkeypatch.setenv("QR_WIDE_NO_EXACT", "1")
    ta, sa, ma = run(False)
    monkeypatch.delenv("QR_WIDE_NO_EXACT")
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    tb, sb, mb = run(True)
    for a, b in zip(ta, tb):
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb) and ma == mb


def test_presorted_lists_refuse_node_histogram_reads(qr, monkeypatch):
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    x, labels, qoff = make_dataset(**CASES[0])
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(0)
    c.reset_scores()
    c.compute_lambdas("NDCG", 10)
    c.fit_tree(6, 1, True)
    with pytest.raises(Exception, match="pre-sorted"):
        c.node_hist_ragged(0)
    c.close()


def test_presorted_split_search_at_the_pop_equals_the_eager_order(qr, monkeypatch):
    """qr_tree_fit on pre-sorted lists searches a node's split when the loop pops it (rt.cc:58-90's
    own order; k_xpop / k_xapply); the phase API -- and QR_X_EAGER=1 -- search both children behind
    every split.  Same pops, same trees, same bits."""
    from quickrank_amd.trainer import Mart
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    x, labels, qoff = make_dataset(**CASES[1])
    kw = dict(ntrees=5, shrinkage=0.1, nthresholds=0, nleaves=9, minls=3, esr=0)

    def run():
        m = Mart(algo="LAMBDAMART", **kw).learn(x, labels, qoff)
        out = [t.copy() for t in m.ensemble.trees], m.ctx.get_scores()
        m.ctx.close()
        return out

    ta, sa = run()
    monkeypatch.setenv("QR_X_EAGER", "1")
    tb, sb = run()
    for a, b in zip(ta, tb):
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb)
    # the phase API on the same lists
    monkeypatch.delenv("QR_X_EAGER")
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(0)
    c.reset_scores()
    c.compute_lambdas("NDCG", 10)
    c.tree_begin(9, 3)
    for _ in range(8):
        c.tree_decide()
        c.tree_apply()
    c.tree_decide()
    t = c.tree_end(9, True)
    for k in t.dtype.names:     # (the ensemble's records are padded to 2 L + 1 entries)
        assert np.array_equal(t[k], ta[0][k][:len(t)], equal_nan=(t[k].dtype.kind == "f")), k
    c.close()


@pytest.mark.parametrize("nleaves,minls,nrows", [(3, 1, 2), (6, 1, 2), (6, 1, 3), (10, 2, 4), (4, 40, 4)])
def test_presorted_steps_without_a_split_are_carried_on(qr, nleaves, minls, nrows, monkeypatch, capfd):
    """Documents that share their whole feature row: a node of them has deviance > 0 and no valid
    split, which the search at the pop only finds out after it used one of the enqueued steps.  The
    last control call reports the tree unfinished and the host carries it on (tree_settle): the
    trees are the histogram path's."""
    rng = np.random.default_rng(5)
    rows = rng.random((nrows, 12), dtype=np.float32)
    nq, dpq = 30, 20
    pick = rng.integers(0, nrows, nq * dpq)
    x = rows[pick].copy()
    labels = rng.integers(0, 5, nq * dpq).astype(np.float32)
    qoff = np.arange(nq + 1, dtype=np.uint64) * dpq

    def run():
        c = qr.Context(0)
        c.upload(x, labels, qoff)
        c.build_bins(0, wide=True)
        assert c.wide
        c.reset_scores()
        trees = []
        for it in range(4):
            c.compute_lambdas("NDCG", 10)
            trees.append(c.fit_tree(nleaves, minls, True).copy())
            c.update_scores(0.1)
        s = c.get_scores()
        c.close()
        return trees, s

    monkeypatch.setenv("QR_NO_BATCH", "1")
    monkeypatch.setenv("QR_WIDE_NO_EXACT", "1")
    ta, sa = run()
    monkeypatch.delenv("QR_WIDE_NO_EXACT")
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    monkeypatch.setenv("QR_SPEC_DEBUG", "1")
    capfd.readouterr()
    tb, sb = run()
    err = capfd.readouterr().err
    for a, b in zip(ta, tb):
        assert len(a) == len(b)
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb)
    if nrows == 2 and nleaves == 3:   # root split, then two children nobody can split: three pops, two enqueued steps
        import re
        m = re.search(r"(\d+) trees with a guessed step count, (\d+) continued", err)
        assert m and int(m.group(2)) > 0, err      # the carried-on path has run


@pytest.mark.parametrize("nleaves", [40, 64, 100])
def test_presorted_lists_large_trees(qr, nleaves, monkeypatch):
    """Trees whose node records and heap outgrow the control kernels' LDS copies (more than 47
    leaves: k_xpop works on the device-resident state): the pre-sorted path with the split search
    at the pop against the slot histograms, every record of every tree."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(**CASES[3])
    kw = dict(ntrees=3, shrinkage=0.1, nthresholds=0, nleaves=nleaves, minls=1, esr=0)

    def run():
        m = Mart(algo="LAMBDAMART", **kw).learn(x, labels, qoff)
        assert m.ctx.wide
        out = [t.copy() for t in m.ensemble.trees], m.ctx.get_scores()
        m.ctx.close()
        return out

    monkeypatch.setenv("QR_NO_BATCH", "1")
    monkeypatch.setenv("QR_WIDE_NO_EXACT", "1")
    ta, sa = run()
    monkeypatch.delenv("QR_WIDE_NO_EXACT")
    monkeypatch.setenv("QR_WIDE_EXACT", "1")
    tb, sb = run()
    for a, b in zip(ta, tb):
        for k in a.dtype.names:
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k
    assert np.array_equal(sa, sb)
