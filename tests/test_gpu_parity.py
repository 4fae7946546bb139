"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle.

Bar (BASELINE.json north_star): split (feature, threshold-slot) sequences and
all integer/index work bit-exact; leaf outputs / NDCG within 1e-5 relative (the
tests ask for far tighter where the arithmetic allows it).
"""
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


def _ctx(qr, x, labels, qoff, nthr):
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    thr, ts = c.build_bins(nthr)
    return c, thr, ts


CASES = [
    dict(nq=40, docs_per_query=30, F=16, seed=0),
    dict(nq=25, docs_per_query=60, F=136, seed=1, ragged=True),
    dict(nq=30, docs_per_query=40, F=70, seed=2, adversarial=True),
    dict(nq=12, docs_per_query=300, F=33, seed=3, ragged=True, adversarial=True),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr", [255, 16, 0])
def test_thresholds_and_bins(qr, ora, case, nthr):
    x, labels, qoff = make_dataset(**case)
    if nthr == 0:
        x = np.floor(x * 200) / 200  # <= 255 distinct values per column
    col = np.ascontiguousarray(x.T)
    othr, ots = ora.thresholds(col, nthr)
    if nthr == 0:
        assert ots.max() <= 256
    c, thr, ts = _ctx(qr, x, labels, qoff, nthr)
    assert np.array_equal(ts.astype(np.uint64), ots)
    for f in range(x.shape[1]):
        n = int(ots[f])
        assert np.array_equal(thr[f, :n].view(np.uint32), othr[f, :n].view(np.uint32)), f
    stmap, _ = ora.binmap(col, othr, ots)
    bins = c.read_bins()
    assert np.array_equal(bins.T.astype(np.uint32), stmap)
    c.close()


def _scores_for(kind, n, rng):
    if kind == "zero":
        return np.zeros(n)
    if kind == "few":
        return rng.integers(0, 5, n) * 0.1
    if kind == "huge":      # scores that span more than exp() can take: the pair term's exp(s_hi - s_lo) path
        return rng.standard_normal(n) * 800.0
    if kind == "mixed":
        s = rng.standard_normal(n)
        s[rng.integers(0, n, n // 4)] = 0.25
        return s
    return rng.standard_normal(n)


def _check_ranks(ora, ranks, scores, qoff, cutoff, exact_tail, tag=None):
    """The rank permutation against the oracle's std::sort (pinned to libstdc++ through
    oracle/_ref).  QR_EXACT_TAIL=1: the whole permutation, bit for bit, tie order included.
    Default: the first `cutoff` ranks bit for bit -- the ones the metric and the lambdas can
    see (ranks beyond the cutoff carry no discount) -- and beyond them the same documents in
    non-increasing score order (the std::sort emulation does not partition ranges that lie
    entirely beyond the cutoff, so equal scores there keep an unspecified order)."""
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        want = ora.rank_by_score(scores[a:b]).astype(np.uint64)
        got = ranks[a:b].astype(np.uint64)
        size = b - a if (exact_tail or cutoff == 0) else min(cutoff, b - a)
        assert np.array_equal(got[:size], want[:size]), (tag, q)
        if size < b - a:
            assert np.array_equal(np.sort(got[size:]), np.sort(want[size:])), (tag, q)
            tail = scores[a:b][got[size:].astype(np.int64)]
            assert np.all(np.diff(tail) <= 0), (tag, q)
            assert np.array_equal(tail, scores[a:b][want[size:].astype(np.int64)]), (tag, q)


@pytest.mark.parametrize("exact_tail", [False, True])
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("kind", ["zero", "few", "mixed", "random", "huge"])
@pytest.mark.parametrize("metric,cutoff", [("NDCG", 10), ("NDCG", 3), ("NDCG", 0), ("DCG", 10)])
def test_ranks_metric_lambdas(qr, ora, case, kind, metric, cutoff, exact_tail, monkeypatch):
    if exact_tail:
        monkeypatch.setenv("QR_EXACT_TAIL", "1")
    else:
        monkeypatch.delenv("QR_EXACT_TAIL", raising=False)
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(5)
    scores = _scores_for(kind, len(labels), rng)
    c = qr.Context(0)
    c.upload(x, labels, qoff)
    c.set_scores(scores)
    c.compute_lambdas(metric, cutoff)
    m = 1 if metric == "NDCG" else 0
    # rank permutation: bit-exact incl. the std::sort tie order (see _check_ranks)
    _check_ranks(ora, c.ranks(), scores, qoff, cutoff, exact_tail)
    # per-query metric: bit-exact (same ranking, host-built log2 table)
    pq = c.metric_per_query()
    L = ora.lib()
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        fn = L.qro_ndcg_query if m else L.qro_dcg_query
        want = fn(np.ascontiguousarray(labels[a:b]), np.ascontiguousarray(scores[a:b]), b - a, cutoff)
        assert pq[q] == want, q
    assert c.metric_last() == pytest.approx(ora.eval_dataset(labels, scores, qoff, cutoff, m), rel=1e-13)
    assert c.metric_eval(0, metric, cutoff) == pytest.approx(
        ora.eval_dataset(labels, scores, qoff, cutoff, m), rel=1e-13)
    lam, w = c.get_pseudo()
    olam, ow = ora.lambdas(labels, scores, qoff, cutoff, m)
    scale = max(1.0, np.abs(olam).max())
    assert np.allclose(lam, olam, rtol=1e-11, atol=1e-13 * scale)
    assert np.allclose(w, ow, rtol=1e-11, atol=1e-13 * scale)
    c.close()



def _oracle_tree(ora, x, nthr, lam, w, nleaves, minls):
    tr = ora.Trainer(x, nthr)
    t = tr.fit_tree(lam, nleaves=nleaves, minls=minls)
    tr.update_output(t, lam, w)
    return tr, t


@pytest.mark.parametrize("case", [CASES[0], CASES[-1]])
@pytest.mark.parametrize("nthr,nleaves", [(255, 10), (32, 24)])
def test_every_node_histogram_against_the_oracle(qr, ora, case, nthr, nleaves):
    """RTNodeHistogram of EVERY node of a fitted tree (rtnode_histogram.cc:41-87): the
    directly built child (the SMALLER one here, the left one in the reference) and the
    sibling by subtraction -- counts exact, sums to the fixed-point resolution -- against the
    oracle's histogram of the node's documents (the documents of the leaves below it)."""
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(17)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c, thr, ts = _ctx(qr, x, labels, qoff, nthr)
    c.set_pseudo(olam, ow)
    nodes = c.fit_tree(nleaves, 1, True)
    tr = ora.Trainer(x, nthr)
    docs = {}

    def below(i):
        if i not in docs:
            n = nodes[i]
            docs[i] = c.node_samples(i).astype(np.uint64) if n["feature"] < 0 else \
                np.sort(np.concatenate([below(int(n["left"])), below(int(n["right"]))]))
        return docs[i]

    tol = 2.0 ** -30 * max(1.0, np.abs(olam).max()) * np.sqrt(len(olam))
    for i in range(len(nodes)):
        ids = below(i)
        assert len(ids) == nodes[i]["nsamples"]
        os_, oc, _ = ora.hist_build(tr.stmap, tr.thr_size, tr.cap, olam, sampleids=ids)
        hs, hc = c.node_hist(i)
        for f in range(x.shape[1]):
            n = int(tr.thr_size[f])
            assert np.array_equal(hc[f, :n], oc[f, :n]), (i, f)
            assert np.allclose(hs[f, :n], os_[f, :n], rtol=0, atol=tol), (i, f)
    c.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr,nleaves,minls", [(255, 10, 1), (32, 16, 5), (255, 4, 1), (8, 31, 2)])
def test_root_histogram_and_tree(qr, ora, case, nthr, nleaves, minls):
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(11)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c, thr, ts = _ctx(qr, x, labels, qoff, nthr)
    c.set_pseudo(olam, ow)
    nodes = c.fit_tree(nleaves, minls, True)
    tr, ot = _oracle_tree(ora, x, nthr, olam, ow, nleaves, minls)
    # root histogram: counts exact, sums to fixed-point resolution
    hs, hc = c.node_hist(0)
    os_, oc, _ = ora.hist_build(tr.stmap, tr.thr_size, tr.cap, olam)
    for f in range(x.shape[1]):
        n = int(tr.thr_size[f])
        assert np.array_equal(hc[f, :n], oc[f, :n]), f
        tol = 2.0 ** -30 * max(1.0, np.abs(olam).max()) * np.sqrt(len(olam))
        assert np.allclose(hs[f, :n], os_[f, :n], rtol=0, atol=tol), f
    # split sequence and structure
    on = ot["nodes"]
    ties = assert_tree_parity(tr.stmap, on, nodes, value_rtol=1e-9)
    # ties (equal-partition candidates in nodes of <= TIE_MAX_DOCS documents) are
    # counted by the walker; everything below is compared modulo them, never skipped
    assert_split_log_parity(c.split_log(), ot["splits"], ties)
    for oi, gi in ties.node_map.items():
        assert np.isclose(nodes[gi]["deviance"], on[oi]["deviance"], rtol=1e-6, atol=1e-9), (oi, gi)
    # leaf membership (stable partition => ascending doc ids), through the node map
    for li, on_leaf in enumerate(ot["leaf_nodes"]):
        ids = c.node_samples(ties.node_map[int(on_leaf)])
        assert np.all(np.diff(ids.astype(np.int64)) > 0)
        want = np.nonzero(ot["leaf_of_doc"] == li)[0]
        assert np.array_equal(ids, want.astype(np.uint32))
    # score update through the leaf membership
    c.set_scores(scores)
    c.update_scores(0.1)
    s2 = scores.copy()
    tr.update_scores(ot, 0.1, s2)
    assert np.allclose(c.get_scores(), s2, rtol=1e-12, atol=1e-13)
    c.close()


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
@pytest.mark.parametrize("case,nthr,nleaves", [(CASES[0], 255, 10), (CASES[1], 255, 10),
                                               (CASES[2], 64, 8), (CASES[3], 255, 16)])
def test_training_loop(qr, ora, algo, case, nthr, nleaves):
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(**case)
    ntrees = 12
    om = ora.train(x, labels, qoff, algo=algo, ntrees=ntrees, shrinkage=0.1, nthresholds=nthr,
                   nleaves=nleaves, minls=1, esr=0)
    gm = Mart(algo=algo, ntrees=ntrees, shrinkage=0.1, nthresholds=nthr, nleaves=nleaves, minls=1,
              esr=0).learn(x, labels, qoff)
    assert len(gm.ensemble) == om["ntrees_built"]
    tr = ora.Trainer(x, nthr)
    for t in range(ntrees):
        n = int(om["nnodes"][t])
        assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n])
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-9)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-8, atol=1e-10)
    # inference: bit-exact given the same model
    nodes, w = gm.ensemble.arrays()
    gm.ctx.upload_ensemble(nodes, w)
    got, _ = gm.ctx.score(x)
    model = dict(nodes=nodes, nnodes=np.full(len(nodes), nodes.shape[1], np.uint64),
                 ntrees=len(nodes), max_nodes=nodes.shape[1], shrinkage=0.1)
    want = ora.ensemble_score(model, x)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    gm.ctx.close()


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
def test_training_loop_medium_exact(qr, ora, algo):
    """Nodes of thousands of documents: no equal-partition candidates, so the
    (feature, slot) sequence must match the oracle bit for bit."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=300, docs_per_query=100, F=136, seed=21)
    kw = dict(ntrees=8, shrinkage=0.1, nthresholds=255, nleaves=10, minls=50, esr=0)
    om = ora.train(x, labels, qoff, algo=algo, **kw)
    gm = Mart(algo=algo, **kw).learn(x, labels, qoff)
    tr = ora.Trainer(x, 255)
    for t in range(kw["ntrees"]):
        n = int(om["nnodes"][t])
        assert assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n],
                                  exact=True) == 0
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-10)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-9, atol=1e-11)
    gm.ctx.close()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("nthr,depth,minls", [(255, 3, 1), (32, 4, 5), (255, 6, 1), (16, 2, 2)])
def test_oblivious_tree(qr, ora, case, nthr, depth, minls):
    """ObliviousRT::fit (ot.cc:32-201): one (feature, slot) per level."""
    x, labels, qoff = make_dataset(**case)
    rng = np.random.default_rng(13)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c, thr, ts = _ctx(qr, x, labels, qoff, nthr)
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
    # one (feature, slot) per level: a tie-resolved level may name another candidate
    same = (log["feature"].astype(np.uint64) == olog["feature"]) & (log["thr_id"].astype(np.uint64) == olog["thr_id"])
    assert int((~same).sum()) <= (1 if ties else 0) * len(log)
    if ties == 0:
        assert same.all()
    assert np.allclose(log["score"], olog["score"], rtol=1e-9)
    c.set_scores(scores)
    c.update_scores(0.1)
    s2 = scores.copy()
    tr.update_scores(ot, 0.1, s2)
    assert np.allclose(c.get_scores(), s2, rtol=1e-12, atol=1e-13)
    c.close()


@pytest.mark.parametrize("algo", ["OBVLAMBDAMART", "OBVMART"])
def test_oblivious_training_loop(qr, ora, algo):
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=120, docs_per_query=60, F=40, seed=23)
    kw = dict(ntrees=8, shrinkage=0.1, nthresholds=64, minls=10, esr=0)
    om = ora.train(x, labels, qoff, algo=algo, depth=4, **kw)
    gm = Mart(algo=algo, depth=4, **kw).learn(x, labels, qoff)
    tr = ora.Trainer(x, 64)
    for t in range(kw["ntrees"]):
        n = int(om["nnodes"][t])
        assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n])
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-9)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-8, atol=1e-10)
    gm.ctx.close()


def _random_tree(rng, nleaves, F, pool, chain=False):
    """(nodes in creation order) a random binary tree of `nleaves` leaves; chain: every split
    keeps splitting its right child (depth = nleaves - 1)."""
    from quickrank_amd._capi import NODE_DTYPE
    n = np.zeros(2 * nleaves - 1, NODE_DTYPE)
    n["feature"] = -1
    n["left"] = n["right"] = -1
    n["value"] = rng.standard_normal(len(n))
    leaves, used = [0], 1
    while len(leaves) < nleaves:
        i = leaves.pop(-1 if chain else int(rng.integers(len(leaves))))
        n[i]["feature"] = int(rng.integers(F))
        n[i]["threshold"] = np.float32(rng.choice(pool))
        n[i]["left"], n[i]["right"] = used, used + 1
        leaves += [used, used + 1]
        used += 2
    return n


@pytest.mark.parametrize("kind", ["p4", "self8", "u16", "wide_rows", "big_trees"])
def test_ensemble_scoring_every_record_format(qr, ora, kind):
    """Ensemble::score_instance (ensemble.cc:111-118) through each compact form of the model --
    4-byte records (u8 bins, trees of up to 255 nodes), 8-byte records with self-looping
    leaves, u16 bins, wide rows (fewer document blocks per workgroup), trees of more than 255 nodes -- with
    single-leaf trees, chains and ragged shapes mixed in: bit-exact against the oracle's walk,
    NaN / inf / -0.0 features included."""
    rng = np.random.default_rng({"p4": 1, "self8": 2, "u16": 3, "wide_rows": 4, "big_trees": 5}[kind])
    F = 400 if kind == "wide_rows" else 37      # (25 KB of u8 bins per 64 documents: fewer waves per workgroup)
    npool = 5000 if kind == "u16" else 200
    pool = np.unique(np.concatenate([rng.standard_normal(npool).astype(np.float32),
                                     np.array([0.0, -0.0, 1.0], np.float32)]))
    sizes = [1, 2, 3, 17, 64, 100, 128] * 3
    if kind in ("self8", "big_trees"):
        sizes += [200, 300]          # > 255 nodes: the 4-byte records do not apply
    trees = [_random_tree(rng, m, F, pool, chain=(k % 5 == 0 and m <= 40)) for k, m in enumerate(sizes)]
    maxn = max(len(t) for t in trees)
    from quickrank_amd._capi import NODE_DTYPE
    nodes = np.zeros((len(trees), maxn), NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    for k, t in enumerate(trees):
        nodes[k, :len(t)] = t
    w = rng.random(len(trees)) + 0.5
    x = rng.choice(pool, size=(3000, F)).astype(np.float32)
    x[rng.integers(0, 3000, 40), rng.integers(0, F, 40)] = np.nan
    x[rng.integers(0, 3000, 40), rng.integers(0, F, 40)] = np.inf
    x[rng.integers(0, 3000, 40), rng.integers(0, F, 40)] = -np.inf
    x[rng.integers(0, 3000, 40), rng.integers(0, F, 40)] = -0.0
    c = qr.Context(0)
    c.upload_ensemble(nodes, w)
    got, _ = c.score(x)
    # the reference's walk, restated: x[f] <= threshold goes left (NaN goes right)
    want = np.zeros(len(x))
    for k, t in enumerate(trees):
        cur = np.zeros(len(x), np.int64)
        while True:
            nd = t[cur]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
            cur[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        want = want + t["value"][cur] * w[k]
    assert np.array_equal(got, want)
    c.close()


@pytest.mark.parametrize("kind", ["p4", "self8"])
def test_ensemble_scoring_depth_order_is_a_reordered_sum(qr, kind):
    """qr_ensemble_set_depth_order (opt-in): the trees are walked and ADDED in ascending order of
    depth.  Every tree's contribution is the model-order walk's, bit for bit (partial scores,
    returned in the model's columns); the total is those contributions summed in depth order --
    restated here -- and equals the model-order total to f64 rounding."""
    rng = np.random.default_rng(11 if kind == "p4" else 12)
    F = 37
    pool = np.unique(rng.standard_normal(200).astype(np.float32))
    sizes = [1, 64, 2, 40, 3, 17, 64, 100, 5, 128, 9, 33] * 3 + ([300] if kind == "self8" else [])
    trees = [_random_tree(rng, m, F, pool, chain=(k % 4 == 0 and m <= 40)) for k, m in enumerate(sizes)]
    maxn = max(len(t) for t in trees)
    from quickrank_amd._capi import NODE_DTYPE
    nodes = np.zeros((len(trees), maxn), NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    for k, t in enumerate(trees):
        nodes[k, :len(t)] = t
    w = rng.random(len(trees)) + 0.5
    x = rng.choice(pool, size=(2000, F)).astype(np.float32)

    def depth(t):
        d, best, stack = 0, 0, [(0, 0)]
        while stack:
            i, d = stack.pop()
            if t[i]["feature"] >= 0:
                best = max(best, d + 1)
                stack += [(int(t[i]["left"]), d + 1), (int(t[i]["right"]), d + 1)]
        return best

    c = qr.Context(0)
    c.upload_ensemble(nodes, w)
    strict, _ = c.score(x)
    per_tree = c.partial_scores(x, len(trees))
    c.upload_ensemble(nodes, w, depth_order=True)
    got, _ = c.score(x)
    per_tree_d = c.partial_scores(x, len(trees))
    assert np.array_equal(per_tree, per_tree_d)              # the model's columns, the same bits
    order = sorted(range(len(trees)), key=lambda k: depth(trees[k]))   # (stable)
    want = np.zeros(len(x))
    for k in order:
        want = want + per_tree[:, k]
    assert np.array_equal(got, want)
    assert np.allclose(got, strict, rtol=1e-13, atol=1e-13) and not np.array_equal(got, strict)
    c.upload_ensemble(nodes, w)                               # back to the model's order: bit for bit again
    again, _ = c.score(x)
    assert np.array_equal(again, strict)
    c.close()


@pytest.mark.parametrize("F,N,T,leaves", [(1, 1, 1, 2), (2, 63, 15, 5), (3, 65, 16, 8), (5, 129, 17, 3),
                                          (7, 64, 33, 1), (200, 700, 48, 64), (9, 1000, 1, 128)])
def test_scoring_walk_edge_shapes(qr, F, N, T, leaves):
    """k_score_p4 on the shapes its layout has edges at: a feature count that is not a multiple of
    four (the bins are one dword per lane and feature QUAD), fewer documents than a wave, document
    counts around a block boundary, tree counts around the batch of sixteen (padding trees), single
    leaves (a group of depth 0), the largest trees the 4-byte records hold.  Bit-exact against
    the reference's walk restated in numpy (ensemble.cc:111-118)."""
    rng = np.random.default_rng(1000 * F + N + T)
    pool = np.unique(rng.standard_normal(60).astype(np.float32))
    trees = [_random_tree(rng, leaves if (k % 3) else max(1, leaves // 2), F, pool, chain=(k % 4 == 1 and leaves <= 40))
             for k in range(T)]
    maxn = max(len(t) for t in trees)
    from quickrank_amd._capi import NODE_DTYPE
    nodes = np.zeros((T, maxn), NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    for k, t in enumerate(trees):
        nodes[k, :len(t)] = t
    w = rng.random(T) + 0.5
    x = rng.choice(pool, size=(N, F)).astype(np.float32)
    c = qr.Context(0)
    c.upload_ensemble(nodes, w)
    got, _ = c.score(x)
    want = np.zeros(N)
    for k, t in enumerate(trees):
        cur = np.zeros(N, np.int64)
        while True:
            nd = t[cur]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
            cur[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        want = want + t["value"][cur] * w[k]
    assert np.array_equal(got, want)
    c.close()


@pytest.mark.parametrize("T,D,F", [(1, 1, 1), (3, 2, 2), (5, 7, 11), (33, 8, 40), (64, 3, 136), (7, 5, 1)])
def test_oblivious_scoring_scalar_kernel_shapes(qr, T, D, F):
    """k_obl_score_s (level tests in scalar registers) at the edges of its layout: every depth it
    takes (1 .. 8), tree counts that are not a multiple of the four trees in flight or of the
    batch (all-zero padding trees), one feature, mixed actual depths."""
    rng = np.random.default_rng(77 * T + D)
    N = 64 * 3 + 7
    feat = rng.integers(0, F, (T, D)).astype(np.uint32)
    thr = rng.random((T, D)).astype(np.float32)
    leaves = rng.standard_normal((T, 1 << D))
    w = (rng.random(T) * 0.2 + 0.01).astype(np.float32)
    depths = np.sort(rng.integers(1, D + 1, T)).astype(np.uint32)
    x = rng.random((N, F), dtype=np.float32)
    c = qr.Context(0)
    for dp in (None, depths):
        c.upload_oblivious(feat, thr, leaves, w, dp)
        got, _ = c.score_oblivious(x)
        want = np.zeros(N)
        for t in range(T):
            m = D if dp is None else int(dp[t])
            idx = np.zeros(N, np.int64)
            for l in range(m):
                idx |= (x[:, feat[t, l]] > thr[t, l]).astype(np.int64) << (m - 1 - l)
            want = want + np.float64(w[t]) * leaves[t, idx]
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), dp is None
    c.close()


def test_oblivious_bit_interleaved_scoring(qr, ora):
    """generate_oblivious.cc:237-324 semantics (f32 tree weights, `>` = right)."""
    rng = np.random.default_rng(3)
    T, D, F, N = 37, 6, 50, 1000
    feat = rng.integers(0, F, (T, D)).astype(np.uint32)
    thr = rng.random((T, D)).astype(np.float32)
    leaves = rng.standard_normal((T, 1 << D))
    w = np.full(T, 0.1, np.float32)
    x = rng.random((N, F), dtype=np.float32)
    want = np.zeros(N)
    ora.lib().qro_oblivious_score(feat, thr, leaves, w, T, D, x, N, F, want)
    c = qr.Context(0)
    c.upload_oblivious(feat, thr, leaves, w)
    got, _ = c.score_oblivious(x)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    c.close()


@pytest.mark.parametrize("T,D,F,pool", [(41, 6, 50, None), (300, 4, 20, None), (64, 8, 9, 3),
                                        (1000, 6, 136, None), (1200, 6, 5, None)])   # the last one: u16 indices
def test_oblivious_scoring_binned_edge_cases(qr, T, D, F, pool):
    """The binned document-parallel scorer against a numpy statement of
    generate_oblivious.cc:305-324: NaN / +-inf / -0.0 inputs (`x > thr` is false for
    NaN), values equal to a threshold, u16 threshold indices (more than 255 distinct
    thresholds on a feature), trees shallower than the array depth, rows wider
    than the features the model tests, a document count that is not a multiple of 64."""
    rng = np.random.default_rng(T + D)
    N = 64 * 37 + 5
    feat = rng.integers(0, F, (T, D)).astype(np.uint32)
    if pool:
        thr = rng.choice(np.array([0.25, 0.5, 0.75], np.float32), (T, D))
    else:
        thr = rng.random((T, D)).astype(np.float32)
    leaves = rng.standard_normal((T, 1 << D))
    w = (rng.random(T) * 0.2).astype(np.float32)
    depths = np.sort(rng.integers(1, D + 1, T)).astype(np.uint32)      # stably ordered by depth
    x = rng.random((N, F + 3), dtype=np.float32)
    x[::7, 1] = np.nan
    x[1::11, 2] = np.inf
    x[2::13, 0] = -np.inf
    x[3::5, 3] = thr[0, 0]                                             # equal to a threshold
    x[4::9, 4] = -0.0
    c = qr.Context(0)
    for dp in (None, depths):
        c.upload_oblivious(feat, thr, leaves, w, dp)
        got, _ = c.score_oblivious(x)
        want = np.zeros(N)
        for t in range(T):
            m = D if dp is None else int(dp[t])
            idx = np.zeros(N, np.int64)
            with np.errstate(invalid="ignore"):
                for l in range(m):
                    idx |= (x[:, feat[t, l]] > thr[t, l]).astype(np.int64) << (m - 1 - l)
            want = want + np.float64(w[t]) * leaves[t, idx]
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), dp is None
    c.close()


def test_validation_early_stop_and_rollback(qr, ora):
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=40, docs_per_query=30, F=20, seed=7)
    vx, vl, vq = make_dataset(nq=15, docs_per_query=25, F=20, seed=8)
    kw = dict(ntrees=30, shrinkage=0.3, nthresholds=64, nleaves=8, minls=20, esr=3)
    om = ora.train(x, labels, qoff, algo="LAMBDAMART", valid=(vx, vl, vq), **kw)
    gm = Mart(algo="LAMBDAMART", **kw).learn(x, labels, qoff, valid=(vx, vl, vq))
    assert len(gm.train_metric) == om["ntrees_built"]
    assert gm.best_model == om["best_model"]
    assert len(gm.ensemble) == om["ntrees"]
    assert np.allclose(gm.valid_metric, om["valid_metric"], rtol=1e-9)
    gm.ctx.close()


def test_no_device_index_is_an_error(qr):
    with pytest.raises(qr.QrError):
        qr.Context(63)


# ---- --max-features (rt.cc:222-243) ------------------------------------------------
M64 = (1 << 64) - 1


def _mf_key(seed, node, f):
    z = (seed + 0x9E3779B97F4A7C15 * ((node * 0x100000001 + f + 1) & M64)) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def _mf_allowed(user_seed, tree_no, node, F, k):
    seed = (user_seed * 0xD1342543DE82EF95 + 0x2545F4914F6CDD1D + tree_no * 0x9E3779B97F4A7C15) & M64
    keys = [(_mf_key(seed, node, f), f) for f in range(F)]
    return {f for _, f in sorted(keys)[:k]}


def test_max_features_subset_per_node(qr, ora):
    x, labels, qoff = make_dataset(nq=80, docs_per_query=50, F=40, seed=31)
    rng = np.random.default_rng(3)
    lam, w = ora.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff, 10, 1)
    c, thr, ts = _ctx(qr, x, labels, qoff, 64)
    c.set_pseudo(lam, w)
    base = c.fit_tree(12, 5, True)
    c.set_max_features(1.0, seed=9)                 # all features: no sampling
    same = c.fit_tree(12, 5, True)
    for k in base.dtype.names:
        assert np.array_equal(same[k], base[k]), k
    F, K = 40, 12                                   # 0.3 * 40 = 12 features per node
    c.set_max_features(0.3, seed=9)
    t1 = c.fit_tree(12, 5, True)                    # tree number 1 of this stream
    log = c.split_log()
    # every split uses a feature of its node's subset, and it is the best split
    # among that subset (first maximum: lowest feature, then lowest slot)
    tsz = ts.astype(np.uint64)
    for n in np.nonzero(t1["feature"] >= 0)[0]:
        allowed = _mf_allowed(9, 1, int(n), F, K)
        assert int(t1[n]["feature"]) in allowed, n
        hs, hc = c.node_hist(int(n))
        best = None
        for f in sorted(allowed):
            sp = ora.split_find(hs, hc, tsz, 5, f, f + 1)
            if sp.feature != 2 ** 64 - 1 and (best is None or sp.score > best[0]):
                best = (sp.score, int(sp.feature), int(sp.thr_id))
        assert best is not None
        assert (int(t1[n]["feature"]), int(t1[n]["thr_id"])) == best[1:], n
    assert len(set(int(f) for f in t1["feature"] if f >= 0) - set(int(f) for f in base["feature"])) >= 0
    # the stream is reproducible: same seed, same sequence of trees
    c.set_max_features(0.3, seed=9)
    again = c.fit_tree(12, 5, True)
    for k in t1.dtype.names:
        assert np.array_equal(again[k], t1[k]), k
    second = c.fit_tree(12, 5, True)                # tree number 2: other subsets
    c.set_max_features(0.3, seed=10)
    other = c.fit_tree(12, 5, True)
    assert not (np.array_equal(second["feature"], t1["feature"]) and np.array_equal(other["feature"], t1["feature"]))
    # a count instead of a fraction; more than F means all
    c.set_max_features(1000.0, seed=1)
    full = c.fit_tree(12, 5, True)
    assert np.array_equal(full["feature"], base["feature"])
    c.close()


# ---- --subsample (mart.cc:287-329, lambdamart.cc:85-102) ----------------------------
def _subset_trainer(ora, full, x, S):
    """Oracle tree machinery over the documents S with the thresholds of the WHOLE
    set (Mart::init runs before any sampling)."""
    t = ora.Trainer.__new__(ora.Trainer)
    t.N, t.F = len(S), full.F
    t.col = np.ascontiguousarray(x[S].T.astype(np.float32))
    t.thr, t.thr_size, t.cap = full.thr, full.thr_size, full.cap
    t.stmap, t.count0 = ora.binmap(t.col, t.thr, t.thr_size)
    t._td = ora.TrainData(t.N, t.F, t.cap, t.col.ctypes.data, t.stmap.ctypes.data, t.thr.ctypes.data,
                          t.thr_size.ctypes.data)
    return t


@pytest.mark.parametrize("algo", ["LAMBDAMART", "MART"])
@pytest.mark.parametrize("subsample", [0.5, 700.0])
def test_subsample_iteration(qr, ora, algo, subsample):
    x, labels, qoff = make_dataset(nq=60, docs_per_query=40, F=24, seed=37, ragged=True)
    N = len(labels)
    c, thr, ts = _ctx(qr, x, labels, qoff, 64)
    full = ora.Trainer(x, 64)
    rng = np.random.default_rng(5)
    scores = np.round(rng.standard_normal(N), 1)            # ties inside queries
    c.set_scores(scores)
    c.set_subsample(subsample, seed=3)
    k = int(subsample) if subsample > 1 else int(np.floor(np.float32(subsample) * np.float32(N)))
    seen = []
    for it in range(3):
        s0 = c.get_scores()
        if algo == "LAMBDAMART":
            c.compute_lambdas("NDCG", 10)
        else:
            c.compute_residuals()
        lam, w = c.get_pseudo()
        nodes = c.fit_tree(8, 2, algo == "LAMBDAMART")
        # the leaves partition the sample (the root's own list is recycled by its
        # grandchildren: the two document lists ping-pong)
        parts = [c.node_samples(int(i)).astype(np.int64) for i in np.nonzero(nodes["feature"] < 0)[0]]
        assert all(np.all(np.diff(p) > 0) for p in parts)
        S = np.sort(np.concatenate(parts))
        assert len(S) == k == nodes[0]["nsamples"] and np.all(np.diff(S) > 0)
        seen.append(S)
        if algo == "LAMBDAMART":
            # the queries cleaned of the other documents (lambdamart.cc:85-102)
            present = np.zeros(N, bool)
            present[S] = True
            cq = np.concatenate([[0], np.cumsum([present[int(qoff[q]):int(qoff[q + 1])].sum()
                                                 for q in range(len(qoff) - 1)])]).astype(np.uint64)
            olam, ow = ora.lambdas(labels[S], s0[S], cq, 10, 1)
            assert np.allclose(lam[S], olam, rtol=1e-10, atol=1e-14)
            assert np.allclose(w[S], ow, rtol=1e-10, atol=1e-14)
            assert not lam[~present].any() and not w[~present].any()
            pl, pw = olam, ow
        else:
            assert np.allclose(lam, labels.astype(np.float64) - s0, rtol=0, atol=0)   # every document
            pl, pw = lam[S], None
        st = _subset_trainer(ora, full, x, S)
        ot = st.fit_tree(pl, nleaves=8, minls=2)
        st.update_output(ot, pl, pw)
        assert_tree_parity(st.stmap, ot["nodes"], nodes, value_rtol=1e-9)
        # every training document is updated, in or out of the sample (mart.cc:345)
        c.update_scores(0.1)
        walk = np.zeros(N, np.int64)
        while True:
            nd = nodes[walk]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
            walk[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        assert np.array_equal(c.get_scores(), s0 + 0.1 * nodes["value"][walk])
    assert not np.array_equal(seen[0], seen[1])             # a fresh sample every iteration
    # reproducible stream
    c.set_scores(scores)
    c.set_subsample(subsample, seed=3)
    c.compute_residuals() if algo == "MART" else c.compute_lambdas("NDCG", 10)
    again = c.fit_tree(8, 2, algo == "LAMBDAMART")
    assert np.array_equal(np.sort(np.concatenate([c.node_samples(int(i)).astype(np.int64)
                                                  for i in np.nonzero(again["feature"] < 0)[0]])), seen[0])
    # subsample 1 (or >= N documents) switches it off
    c.set_subsample(1.0)
    c.compute_lambdas("NDCG", 10)
    assert c.fit_tree(8, 2, True)[0]["nsamples"] == N
    c.close()


def _sample_keys(seed, draw, N, mask):
    """k_sample.hip's key of every document for the draw-th sample of a context seeded with `seed`
    (splitmix64 of the document's index; the stream advances by the golden ratio per draw)."""
    m = (1 << 64) - 1
    s = ((seed * 0xD1342543DE82EF95 + 0x632BE59BD9B4E019) + draw * 0x9E3779B97F4A7C15) & m
    with np.errstate(over="ignore"):
        z = np.uint64(s) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, N + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(32)).astype(np.uint32) & np.uint32(mask)


@pytest.mark.parametrize("mask", [0xFFFFFFFF, 0xF0000000, 0x00000003, 0x001FFC00, 0],
                         ids=["full_keys", "16_keys", "4_keys_low_digit", "middle_digit_only", "one_key"])
@pytest.mark.parametrize("subsample", [0.37, 2049.0])
def test_subsample_is_the_k_smallest_keys(qr, mask, subsample):
    """The sample of a draw is EXACTLY the k documents with the smallest keys, equal keys by ascending
    document -- what a stable sort of (key, document) pairs keeps (rounds 2-5 sorted them; the radix
    select of round 6 must pick the same set).  Narrow keys make the k-th smallest one the key of
    hundreds of documents: the tie rule decides most of the sample."""
    x, labels, qoff = make_dataset(nq=170, docs_per_query=40, F=8, seed=43, ragged=True)
    N = len(labels)
    assert N > 4200 and N % 8                   # (several workgroups of the scatter pass and a ragged tail)
    c, thr, ts = _ctx(qr, x, labels, qoff, 16)
    c.set_subsample(subsample, seed=21)
    c.debug_sample_key_mask(mask)
    k = int(subsample) if subsample > 1 else int(np.floor(np.float32(subsample) * np.float32(N)))
    for draw in (1, 2, 3):
        c.compute_residuals()
        nodes = c.fit_tree(6, 1, False)
        S = np.sort(np.concatenate([c.node_samples(int(i)).astype(np.int64)
                                    for i in np.nonzero(nodes["feature"] < 0)[0]]))
        want = np.sort(np.argsort(_sample_keys(21, draw, N, mask), kind="stable")[:k])
        assert len(S) == k and np.array_equal(S, want), (draw, hex(mask))
        c.update_scores(0.1)
    c.close()


@pytest.mark.parametrize("algo", ["OBVLAMBDAMART", "OBVMART"])
def test_subsample_oblivious_iteration(qr, ora, algo):
    """--subsample with oblivious trees (ObliviousMart inherits Mart::learn's sampling,
    mart.cc:287-329): the level-wise tree is grown on the sample's list; given the sample,
    the tree equals the oracle's on those documents and every document's score is updated."""
    x, labels, qoff = make_dataset(nq=60, docs_per_query=40, F=24, seed=41, ragged=True)
    N = len(labels)
    c, thr, ts = _ctx(qr, x, labels, qoff, 64)
    full = ora.Trainer(x, 64)
    rng = np.random.default_rng(6)
    c.set_scores(np.round(rng.standard_normal(N), 1))
    c.set_subsample(0.4, seed=8)
    k = int(np.floor(np.float32(0.4) * np.float32(N)))
    newton = algo == "OBVLAMBDAMART"
    for it in range(3):
        s0 = c.get_scores()
        c.compute_lambdas("NDCG", 10) if newton else c.compute_residuals()
        lam, w = c.get_pseudo()
        nodes = c.fit_oblivious(4, 2, newton)
        parts = [c.node_samples(int(i)).astype(np.int64) for i in np.nonzero(nodes["feature"] == -1)[0]]
        S = np.sort(np.concatenate(parts))
        assert len(S) == k == nodes[0]["nsamples"] and np.all(np.diff(S) > 0)
        st = _subset_trainer(ora, full, x, S)
        ot = st.fit_tree(lam[S], minls=2, oblivious_depth=4)
        st.update_output(ot, lam[S], w[S] if newton else None)
        assert_tree_parity(st.stmap, ot["nodes"], nodes, value_rtol=1e-9)
        c.update_scores(0.1)
        walk = np.zeros(N, np.int64)
        while True:
            nd = nodes[walk]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
            walk[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        assert np.array_equal(c.get_scores(), s0 + 0.1 * nodes["value"][walk])
    c.close()


def test_error_paths_return_codes_not_crashes(qr):
    """Every misuse of the C-ABI comes back as a QrError with a message (the host
    prints it and exits, as the reference does): nothing asserts or segfaults."""
    x, labels, qoff = make_dataset(nq=10, docs_per_query=20, F=6, seed=1)
    c = qr.Context(0)
    with pytest.raises(qr.QrError, match="no dataset"):
        c.build_bins(16)
    with pytest.raises(qr.QrError):
        c.fit_tree(4, 1, True)
    with pytest.raises(qr.QrError, match="follows qr_lambda_compute"):
        c.metric_last()
    c.upload(x, labels, qoff)
    with pytest.raises(qr.QrError, match="wide path"):
        c.build_bins(300, wide=False)      # more than 255 thresholds are the wide path's (test_gpu_wide.py)
    with pytest.raises(qr.QrError, match="nthresholds <= 255"):
        c._ck(c.L.qr_bins_build(c.h, 300, None, None))
    with pytest.raises(qr.QrError, match="bins not built"):
        c.set_subsample(0.5)
    c.build_bins(16)
    with pytest.raises(qr.QrError, match="already built"):
        c.build_bins(16)
    with pytest.raises(qr.QrError, match="nleaves must be"):
        c.fit_tree(5000, 1, True)
    with pytest.raises(qr.QrError, match="tree depth"):
        c.fit_oblivious(12, 1, True)
    with pytest.raises(qr.QrError, match="no fitted tree"):
        c.update_scores(0.1)
    with pytest.raises(qr.QrError, match="must be > 0"):
        c.set_max_features(0.0)
    with pytest.raises(qr.QrError, match="no validation set"):
        c.set_valid_scores(np.zeros(0))
    with pytest.raises(qr.QrError, match="metric must be"):
        c._ck(c.L.qr_lambda_compute(c.h, 7, 10))
    bad_q = qoff.copy()
    bad_q[3] = bad_q[2] - 1
    with pytest.raises(qr.QrError, match="non-decreasing"):
        c.upload(x, labels, bad_q)
    # a sharded context must be driven through the phase calls
    s = qr.Context(0, rank=0, world=2)
    s.upload(x, labels, qoff)
    s.build_bins(16)
    s.set_pseudo(np.ones(len(labels)), np.ones(len(labels)))
    with pytest.raises(qr.QrError, match="phase|begin/decide"):
        s.fit_tree(4, 1, True)
    with pytest.raises(qr.QrError, match="phase by phase"):
        s.fit_oblivious(3, 1, True)
    d = qr.Context(0, rank=0, world=1, doc_shard=(len(labels), len(qoff) - 1))
    d.upload(x, labels, qoff)
    d.build_bins_with(*qr._capi.thresholds_from_stats(x.shape[1], 16, *[a[None] for a in d.bins_stats(16)]))
    with pytest.raises(qr.QrError, match="document-sharded contexts"):
        c.set_subsample(0.5, first_doc=0)            # ... and only they
    with pytest.raises(qr.QrError, match="qr_obl_level_exchange follows"):
        c.obl_level_exchange(0)
    with pytest.raises(qr.QrError, match="qr_subsample_set_doc"):
        d.set_subsample(0.5)             # document-sharded ranks say where their documents start
    with pytest.raises(qr.QrError, match="qr_obl_level_exchange follows"):
        d.obl_level_exchange(0)          # before qr_obl_begin
    d.obl_begin(3, 1)
    with pytest.raises(qr.QrError, match="no mask to exchange"):
        d.obl_mark(0)
    with pytest.raises(qr.QrError, match="outside the global range"):
        d.set_subsample(0.5, first_doc=7)
    d.set_subsample(0.5, first_doc=0)
    d.close()
    with pytest.raises(qr.QrError):
        qr.Context(0, rank=3, world=2)
    s.close()
    c.close()


@pytest.mark.parametrize("exact_tail", [False, True])
def test_queries_longer_than_the_lds(qr, ora, exact_tail, monkeypatch):
    """Queries whose working set does not fit the LDS (> ~3300 documents) run out of
    a global scratch slice: same ranks (ties included), metric, lambdas and weights;
    short queries of the same set keep the LDS path."""
    if exact_tail:
        monkeypatch.setenv("QR_EXACT_TAIL", "1")
    else:
        monkeypatch.delenv("QR_EXACT_TAIL", raising=False)
    rng = np.random.default_rng(8)
    lens = [50, 5000, 7, 3400, 120]
    qoff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    N = int(qoff[-1])
    x = rng.random((N, 5), dtype=np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    scores = np.round(rng.standard_normal(N), 2)            # plenty of ties
    c, _, _ = _ctx(qr, x, labels, qoff, 16)
    c.set_scores(scores)
    for metric, cutoff in (("NDCG", 10), ("DCG", 0)):
        c.compute_lambdas(metric, cutoff)
        lam, w = c.get_pseudo()
        olam, ow = ora.lambdas(labels, scores, qoff, cutoff, 1 if metric == "NDCG" else 0)
        assert np.allclose(lam, olam, rtol=1e-10, atol=1e-14)
        assert np.allclose(w, ow, rtol=1e-10, atol=1e-14)
        _check_ranks(ora, c.ranks(), scores, qoff, cutoff, exact_tail, metric)
        assert c.metric_last() == pytest.approx(
            ora.eval_dataset(labels, scores, qoff, cutoff, 1 if metric == "NDCG" else 0), rel=1e-13)
    c.close()


@pytest.mark.parametrize("exact_tail", [False, True])
@pytest.mark.parametrize("kind", ["zero", "few", "mixed", "random"])
@pytest.mark.parametrize("metric,cutoff", [("NDCG", 10), ("NDCG", 3), ("NDCG", 0), ("DCG", 40), ("NDCG", 300)])
def test_ragged_set_single_launch_roles(qr, ora, kind, metric, cutoff, exact_tail, monkeypatch):
    """The one launch of a ragged query set (k_lambda_u, round 5): query lengths on both sides of
    every role boundary (one packed wave up to 128 / up to 256 documents, eight waves beyond, the
    launch's own capacity ~1700 documents at cutoff 10 -- fewer with more top ranks kept -- and the
    size-class / global-scratch launches beside it for what is longer), empty-ish queries, every
    kind of tie, cutoffs from 3 to "none" (the pair sweep's rounds of five ranks then run over all
    of a query's ranks).  Ranks bit for bit, the metric bit for bit per query, lambdas to 1e-10.
    The size-class launches of round 4 stay for what the one launch does not take (a sample's cleaned
    lists, queries beyond its capacity) and are covered by the --subsample tests."""
    if exact_tail:
        monkeypatch.setenv("QR_EXACT_TAIL", "1")
    else:
        monkeypatch.delenv("QR_EXACT_TAIL", raising=False)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 130, 200, 255, 256, 257, 258, 300, 511, 512, 513, 600, 1100,
            1152, 1153, 1700, 1750, 2500, 3400, 17, 40, 100, 128, 256, 5]
    qoff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    N = int(qoff[-1])
    rng = np.random.default_rng(len(kind) + cutoff)
    x = rng.random((N, 5), dtype=np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    labels[int(qoff[3]):int(qoff[4])] = 0           # a query without a relevant document
    scores = _scores_for(kind, N, rng)
    c, _, _ = _ctx(qr, x, labels, qoff, 16)
    c.set_scores(scores)
    m = 1 if metric == "NDCG" else 0
    c.compute_lambdas(metric, cutoff)
    _check_ranks(ora, c.ranks(), scores, qoff, cutoff, exact_tail, (kind, metric, cutoff))
    pq = c.metric_per_query()
    L = ora.lib()
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        fn = L.qro_ndcg_query if m else L.qro_dcg_query
        assert pq[q] == fn(np.ascontiguousarray(labels[a:b]), np.ascontiguousarray(scores[a:b]), b - a, cutoff), q
    lam, w = c.get_pseudo()
    olam, ow = ora.lambdas(labels, scores, qoff, cutoff, m)
    scale = max(1.0, np.abs(olam).max())
    assert np.allclose(lam, olam, rtol=1e-10, atol=1e-13 * scale)
    assert np.allclose(w, ow, rtol=1e-10, atol=1e-13 * scale)
    assert c.metric_last() == pytest.approx(ora.eval_dataset(labels, scores, qoff, cutoff, m), rel=1e-13)
    # another cutoff on the SAME context is another plan (the slices hold the top ranks' sums)
    for cut2 in (cutoff + 2, 5):
        c.compute_lambdas(metric, cut2)
        lam2, w2 = c.get_pseudo()
        ol2, ow2 = ora.lambdas(labels, scores, qoff, cut2, m)
        sc2 = max(1.0, np.abs(ol2).max())
        assert np.allclose(lam2, ol2, rtol=1e-10, atol=1e-13 * sc2) and np.allclose(w2, ow2, rtol=1e-10, atol=1e-13 * sc2), cut2
    # the validation set takes the same launch (metric only)
    c.upload_valid(x, labels, qoff)
    c.set_valid_scores(scores)
    assert c.metric_eval(1, metric, cutoff) == pytest.approx(ora.eval_dataset(labels, scores, qoff, cutoff, m), rel=1e-13)
    c.close()


def test_score_update_left_to_the_lambda_pass(qr, ora):
    """qr_scores_update on one GPU leaves the update to the next lambda pass (mart.cc:464-467 ->
    lambdamart.cc:70; k_lambda.hip `upd_leaf`): whoever looks at the scores in between gets them
    updated all the same -- a read, a metric evaluation, a second tree without a lambda pass, a
    second update -- and the pass itself computes on the updated scores.  Bit for bit against the
    walk of the tree on the raw rows, on a uniform and on a ragged set (the short-queries launch
    and the ragged set's single launch)."""
    for case in (dict(nq=80, docs_per_query=50, F=12, seed=3), dict(nq=70, docs_per_query=90, F=12, seed=4, ragged=True)):
        x, labels, qoff = make_dataset(**case)
        N = len(labels)
        c, _, _ = _ctx(qr, x, labels, qoff, 32)
        rng = np.random.default_rng(9)
        s0 = np.round(rng.standard_normal(N), 1)
        c.set_scores(s0)

        def walk(nodes):
            at = np.zeros(N, np.int64)
            while True:
                nd = nodes[at]
                idx = np.nonzero(nd["feature"] >= 0)[0]
                if not len(idx):
                    return nodes["value"][at]
                go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
                at[idx] = np.where(go, nd["left"][idx], nd["right"][idx])

        want = s0.copy()
        for it in range(4):
            c.compute_lambdas("NDCG", 10)
            # the pass ran on the scores as they should be by now
            olam, ow = ora.lambdas(labels, want, qoff, 10, 1)
            lam, w = c.get_pseudo()
            sc = max(1.0, np.abs(olam).max())
            assert np.allclose(lam, olam, rtol=1e-10, atol=1e-13 * sc) and np.allclose(w, ow, rtol=1e-10, atol=1e-13 * sc)
            nodes = c.fit_tree(8, 2, True)
            c.update_scores(0.1)
            want = want + 0.1 * walk(nodes)
            if it == 0:
                assert np.array_equal(c.get_scores(), want)                       # a read in between
            elif it == 1:
                assert c.metric_eval(0, "NDCG", 10) == pytest.approx(ora.eval_dataset(labels, want, qoff, 10, 1), rel=1e-13)
            elif it == 2:
                nodes2 = c.fit_tree(8, 2, True)                                   # a tree without a lambda pass
                c.update_scores(0.1)                                              # ... and a second update
                want = want + 0.1 * walk(nodes2)
        assert np.array_equal(c.get_scores(), want)
        c.close()


@pytest.mark.parametrize("nleaves", [22, 23, 64, 65, 255, 256])
def test_leaf_counts_around_the_batched_growth_limits(qr, ora, nleaves):
    """Two splits per step with the control step inside the partition launch and its state
    in the small (<= 22 leaves) or the large (<= 64) LDS copy, as its own launch on device
    memory (<= 255), one split per step beyond: same trees, and
    the split log in the reference's order."""
    x, labels, qoff = make_dataset(nq=150, docs_per_query=40, F=20, seed=nleaves)
    rng = np.random.default_rng(nleaves)
    scores = rng.standard_normal(len(labels)) * 0.3
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    c, thr, ts = _ctx(qr, x, labels, qoff, 64)
    c.set_pseudo(olam, ow)
    nodes = c.fit_tree(nleaves, 1, True)
    tr = ora.Trainer(x, 64)
    ot = tr.fit_tree(olam, nleaves=nleaves, minls=1)
    tr.update_output(ot, olam, ow)
    on = ot["nodes"][:ot["nnodes"]] if "nnodes" in ot else ot["nodes"]
    assert len(nodes) == len(on) and (nodes["feature"] < 0).sum() == nleaves
    ties = assert_tree_parity(tr.stmap, on, nodes, value_rtol=1e-9)
    log, olog = c.split_log(), ot["splits"]
    assert len(log) == len(olog) == nleaves - 1
    assert_split_log_parity(log, olog, ties)
    c.close()


@pytest.mark.parametrize("hint", ["1", "2", "4"])
def test_guessed_step_count_and_continuation(qr, ora, monkeypatch, hint):
    """Batched growth enqueues a GUESSED number of steps per tree (the previous tree's + 1)
    and carries a tree on from the host when the device reports that the guess was too low
    (qr_k_tree_fit_batch / qr_k_tree_continue).  QR_STEPS_HINT forces a guess that is too
    low for every tree: every tree then goes through the continuation, with the score
    update enqueued behind it repeated -- same trees, metrics and scores as the oracle's."""
    from quickrank_amd.trainer import Mart
    monkeypatch.setenv("QR_STEPS_HINT", hint)
    x, labels, qoff = make_dataset(nq=150, docs_per_query=60, F=40, seed=91)
    kw = dict(ntrees=6, shrinkage=0.1, nthresholds=64, nleaves=12, minls=5, esr=0)
    om = ora.train(x, labels, qoff, algo="LAMBDAMART", **kw)
    gm = Mart(algo="LAMBDAMART", **kw).learn(x, labels, qoff)
    monkeypatch.delenv("QR_STEPS_HINT")
    tr = ora.Trainer(x, 64)
    for t in range(kw["ntrees"]):
        n = int(om["nnodes"][t])
        assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n])
    assert np.allclose(gm.train_metric, om["train_metric"], rtol=1e-9)
    assert np.allclose(gm.ctx.get_scores(), om["train_scores"], rtol=1e-8, atol=1e-10)
    gm.ctx.close()


@pytest.mark.parametrize("algo,subsample,ragged", [("LAMBDAMART", 1.0, False), ("LAMBDAMART", 1.0, True),
                                                    ("LAMBDAMART", 0.6, True), ("OBVLAMBDAMART", 1.0, True)])
def test_scalars_finished_in_the_root_scan_launch_equal_the_launch_of_their_own(qr, monkeypatch, algo,
                                                                                subsample, ragged):
    """qr_lambda_compute defers the launch that finishes the iteration's scalars (quantisation
    scale, the root's sums, the metric): batched and level-wise growth let its workgroups ride in
    the tree's root scan launch, and the root histogram / root scan take the scale from the slot
    set the lambda pass filled (csrc/qr_prep.h).  QR_NO_DEFER_PREP=1 keeps the launch of its own.
    Same additions in the same order: trees, metrics and scores bit for bit -- over enough
    iterations for both slot sets to be reused, on ragged query sets (several lambda launches
    fill one slot set) and with a sample."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=180, docs_per_query=70, F=30, seed=57, ragged=ragged)
    kw = dict(ntrees=7, shrinkage=0.1, nthresholds=64, minls=2, esr=0)
    if algo.startswith("OBV"):
        kw["depth"] = 4
    else:
        kw["nleaves"] = 9
    if subsample != 1.0:
        kw["subsample"] = subsample
    got = Mart(algo=algo, **kw).learn(x, labels, qoff)
    monkeypatch.setenv("QR_NO_DEFER_PREP", "1")
    want = Mart(algo=algo, **kw).learn(x, labels, qoff)
    monkeypatch.delenv("QR_NO_DEFER_PREP")
    for t in range(kw["ntrees"]):
        g, w = got.ensemble.trees[t], want.ensemble.trees[t]
        assert len(g) == len(w)
        for f in g.dtype.names:
            assert np.array_equal(g[f], w[f]), (t, f)
    assert np.array_equal(np.asarray(got.train_metric), np.asarray(want.train_metric))
    assert np.array_equal(got.ctx.get_scores(), want.ctx.get_scores())
    # a lambda pass that no tree follows: the scalars are finished on demand
    got.ctx.compute_lambdas("NDCG", 10)
    m1 = got.ctx.metric_last()
    want.ctx.compute_lambdas("NDCG", 10)
    assert m1 == want.ctx.metric_last()
    got.ctx.close()
    want.ctx.close()


def test_deferred_scalars_survive_any_call_order(qr, monkeypatch):
    """The scalars of a lambda pass stay unfinished until a tree's root scan launch or any call
    that needs them (csrc/qr_prep.h).  Call orders a host is free to choose -- two lambda passes in
    a row, the pseudo-responses read back, the metric asked for at once, a MART residual pass in
    between, a tree through the phase calls, an oblivious tree, the scores replaced -- must give
    what the launch of its own (QR_NO_DEFER_PREP=1) gives, bit for bit."""
    from quickrank_amd._capi import Context
    x, labels, qoff = make_dataset(nq=90, docs_per_query=40, F=18, seed=5, ragged=True)

    def run():
        c = Context(0)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        c.reset_scores()
        out = []
        c.compute_lambdas("NDCG", 10)
        c.compute_lambdas("NDCG", 10)                 # the first pass's scalars: nobody's
        out.append(c.fit_tree(8, 2, True)); c.update_scores(0.1)
        c.compute_lambdas("NDCG", 10)
        out.append(np.asarray(c.metric_last()))       # asked for before any tree
        lam, w = c.get_pseudo()
        out += [lam, w]
        out.append(c.fit_tree(8, 2, True)); c.update_scores(0.1)
        c.compute_lambdas("NDCG", 10)
        c.compute_residuals()                         # MART pass over an unfinished lambda pass
        out.append(c.fit_tree(6, 2, False)); c.update_scores(0.1)
        c.compute_lambdas("NDCG", 10)
        out.append(c.fit_oblivious(3, 2, True)); c.update_scores(0.1)
        out.append(np.asarray(c.metric_last()))
        c.compute_lambdas("NDCG", 10)
        c.tree_begin(7, 2)                            # one split per step, through the phase calls
        for _ in range(6):
            c.tree_decide(); c.tree_apply()
        c.tree_decide()
        out.append(c.tree_end(7, True)); c.update_scores(0.1)
        c.compute_lambdas("NDCG", 10)
        c.set_scores(np.zeros(len(labels)))           # the scores replaced under an unfinished pass
        c.compute_lambdas("NDCG", 10)
        out.append(c.fit_tree(8, 2, True)); c.update_scores(0.1)
        out.append(np.asarray(c.metric_eval(0)))
        out.append(c.get_scores())
        c.close()
        return out

    got = run()
    monkeypatch.setenv("QR_NO_DEFER_PREP", "1")
    want = run()
    monkeypatch.delenv("QR_NO_DEFER_PREP")
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        if g.dtype.names:
            for f in g.dtype.names:
                assert np.array_equal(g[f], w[f]), (i, f)
        else:
            assert np.array_equal(g, w), i


@pytest.mark.parametrize("ragged", [False, True])
def test_lazy_score_update_equals_its_own_launch(qr, monkeypatch, ragged):
    """Mart::update_modelscores (mart.cc:447-468) rides in the NEXT lambda pass on a
    single-GPU context -- between qr_scores_update and that pass the score array on the device is one
    tree behind, and every entry point that reads or writes it has to settle the pending update
    first.  QR_LAZY_SCORES=0 (read at context creation) makes every update a launch of its own.
    The two must agree bit for bit, with the readers a host may call in between -- scores read back,
    the metric asked for, the pseudo-responses read, the scores replaced, a MART residual pass, an
    oblivious tree, two updates in a row -- interleaved with the iterations."""
    from quickrank_amd._capi import Context
    x, labels, qoff = make_dataset(nq=150, docs_per_query=60, F=24, seed=91, ragged=ragged)

    def run():
        c = Context(0)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        c.reset_scores()
        out = []
        for it in range(9):
            c.compute_lambdas("NDCG", 10)
            out.append(np.asarray(c.metric_last()))
            out.append(c.fit_tree(10, 2, True))
            c.update_scores(0.1)                       # pending from here on (lazy contexts)
            if it == 1:
                out.append(c.get_scores())             # a reader right behind the update
            if it == 2:
                out.append(np.asarray(c.metric_eval(0)))
            if it == 3:
                out += list(c.get_pseudo())            # (does not touch the scores: stays pending)
                out.append(c.get_scores())
            if it == 4:
                c.compute_residuals()                  # MART's pass reads the scores
                out.append(c.fit_tree(6, 2, False)); c.update_scores(0.1)
            if it == 5:
                out.append(c.fit_oblivious(3, 2, True)); c.update_scores(0.05)   # two updates, no pass between
            if it == 6:
                s = c.get_scores(); c.set_scores(s * 0.5)                        # a writer
            if it == 7:
                c.update_scores(0.1)                   # the same tree once more: a second update behind a pending one
        c.compute_lambdas("NDCG", 10)
        out.append(np.asarray(c.metric_last()))
        out.append(c.get_scores())
        c.close()
        return out

    monkeypatch.delenv("QR_LAZY_SCORES", raising=False)
    got = run()
    monkeypatch.setenv("QR_LAZY_SCORES", "0")
    want = run()
    monkeypatch.delenv("QR_LAZY_SCORES")
    assert len(got) == len(want)
    for i, (g, w) in enumerate(zip(got, want)):
        if g.dtype.names:
            for f in g.dtype.names:
                assert np.array_equal(g[f], w[f]), (i, f)
        else:
            assert np.array_equal(g, w), i


@pytest.mark.parametrize("depth,minls,subsample", [(1, 1, 1.0), (3, 1, 1.0), (6, 1, 1.0), (7, 2, 1.0),
                                                    (6, 400, 1.0), (5, 3000, 1.0), (5, 1, 0.5)])
def test_oblivious_launches_folded_into_their_neighbours(qr, monkeypatch, depth, minls, subsample):
    """Single-GPU level-wise growth resets the tree state in the last workgroup of the root scan
    launch, lets k_obl_plan number the leaves at the level the tree ENDS at (the last one, or the
    first that finds no split: a large min-leaf-support ends trees early) and updates the scores of
    a level-order tree without a walk.  QR_OBL_OWN_LAUNCHES=1 brings k_obl_reset and k_finish
    back.  Same trees, metrics and scores bit for bit, whatever the depth the tree reaches."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=120, docs_per_query=50, F=20, seed=77, ragged=True)
    kw = dict(ntrees=5, shrinkage=0.1, nthresholds=32, minls=minls, esr=0, depth=depth)
    if subsample != 1.0:
        kw["subsample"] = subsample
    got = Mart(algo="OBVLAMBDAMART", **kw).learn(x, labels, qoff)
    monkeypatch.setenv("QR_OBL_OWN_LAUNCHES", "1")
    want = Mart(algo="OBVLAMBDAMART", **kw).learn(x, labels, qoff)
    monkeypatch.delenv("QR_OBL_OWN_LAUNCHES")
    shallow = 0
    for t in range(kw["ntrees"]):
        g, w = got.ensemble.trees[t], want.ensemble.trees[t]
        assert len(g) == len(w)
        shallow += int(np.count_nonzero(g["nsamples"])) < (1 << (depth + 1)) - 1  # (padding: 0 samples)
        for f in g.dtype.names:
            assert np.array_equal(g[f], w[f]), (t, f)
    if minls >= 400:
        assert shallow > 0, "the case is meant to end trees before their last level"
    assert np.array_equal(np.asarray(got.train_metric), np.asarray(want.train_metric))
    assert np.array_equal(got.ctx.get_scores(), want.ctx.get_scores())
    got.ctx.close()
    want.ctx.close()


@pytest.mark.parametrize("algo,nleaves,subsample", [("LAMBDAMART", 10, 1.0), ("LAMBDAMART", 16, 0.5),
                                                     ("MART", 7, 1.0), ("OBVLAMBDAMART", 16, 1.0)])
def test_leaf_sums_in_document_order_equal_the_position_order(qr, monkeypatch, algo, nleaves, subsample):
    """Trees of up to 16 leaves take their leaf sums in document order (k_leaf_sums_doc: every
    document finds its leaf by the tree's tests, lambda / weight stream in coalesced) and update
    the scores from the leaf bytes; larger ones through the leaves' document lists (k_leaf_sums,
    which QR_LEAF_BY_POSITION=1 forces).  Same trees -- structure bit for bit, leaf values and
    scores to the rounding of two summation orders -- over several boosting iterations."""
    from quickrank_amd.trainer import Mart
    x, labels, qoff = make_dataset(nq=200, docs_per_query=53, F=24, seed=23, ragged=True)
    kw = dict(ntrees=5, shrinkage=0.1, nthresholds=64, minls=3, esr=0)
    if algo.startswith("OBV"):
        kw["depth"] = 4
    else:
        kw["nleaves"] = nleaves
    if subsample != 1.0:
        kw["subsample"] = subsample
    got = Mart(algo=algo, **kw).learn(x, labels, qoff)
    monkeypatch.setenv("QR_LEAF_BY_POSITION", "1")
    want = Mart(algo=algo, **kw).learn(x, labels, qoff)
    monkeypatch.delenv("QR_LEAF_BY_POSITION")
    for t in range(kw["ntrees"]):
        g, w = got.ensemble.trees[t], want.ensemble.trees[t]
        assert len(g) == len(w)
        for k in ("feature", "thr_id", "threshold", "left", "right", "nsamples"):
            assert np.array_equal(g[k], w[k]), (t, k)
        # (a leaf's sum of lambdas cancels: the two orders agree to ~1e-16 of the sum of the
        # magnitudes, not of the result -- the tolerance the oracle comparison uses)
        assert np.allclose(g["value"], w["value"], rtol=1e-9, atol=1e-12), (t, np.abs(g["value"] - w["value"]).max())
    assert np.allclose(got.ctx.get_scores(), want.ctx.get_scores(), rtol=1e-9, atol=1e-12)
    assert np.allclose(got.train_metric, want.train_metric, rtol=1e-12)
    got.ctx.close()
    want.ctx.close()


@pytest.mark.parametrize("n", [600, 2000, 3250, 3300])
def test_one_query_at_the_lds_boundary(qr, ora, n):
    """A single long query next to a short one: 600 documents take the sixteen-wave class,
    3250-3300 sit at the largest working set the LDS holds (dynamic + the sixteen-wave
    kernel's static part must stay inside 160 KB: a launch that asked for more was refused
    with `invalid argument`), beyond that the global-scratch launch takes over."""
    rng = np.random.default_rng(n)
    N = n + 37
    x = rng.random((N, 4), dtype=np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    qoff = np.array([0, n, N], np.uint64)
    scores = np.round(rng.standard_normal(N), 1)
    c, _, _ = _ctx(qr, x, labels, qoff, 16)
    c.set_scores(scores)
    c.compute_lambdas("NDCG", 10)
    lam, w = c.get_pseudo()
    olam, ow = ora.lambdas(labels, scores, qoff, 10, 1)
    assert np.allclose(lam, olam, rtol=1e-10, atol=1e-14)
    assert np.allclose(w, ow, rtol=1e-10, atol=1e-14)
    assert c.metric_last() == pytest.approx(ora.eval_dataset(labels, scores, qoff, 10, 1), rel=1e-13)
    c.close()
