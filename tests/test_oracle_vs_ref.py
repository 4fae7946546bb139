"""Pin the C restatement (oracle/qr_oracle.c) against the reference itself.

oracle/_ref/libqr_ref.so is the reference's own translation units (dataset,
queryresults, rankedresults, dcg, ndcg, rtnode_histogram, radix) compiled
unmodified -- see oracle/Makefile.  Everything here is bit-exact.
"""
import ctypes as C

import numpy as np
import pytest

from datagen import make_dataset

pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def libs(oracle_lib):
    R = oracle_lib.ref()
    if R is None:
        pytest.skip("oracle/_ref/libqr_ref.so not present")
    return oracle_lib.lib(), R


def _tied_scores(rng, n, kind):
    if kind == "equal":
        return np.zeros(n)
    if kind == "few":
        return rng.integers(0, 4, n).astype(np.float64) * 0.25
    if kind == "some":
        s = rng.standard_normal(n)
        if n:
            s[rng.integers(0, n, n // 3)] = s[0]
        return s
    if kind == "sorted":
        return -np.arange(n, dtype=np.float64)
    if kind == "reverse":
        return np.arange(n, dtype=np.float64)
    if kind == "organ":
        return np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.float64)
    return rng.standard_normal(n)


@pytest.mark.parametrize("kind", ["equal", "few", "some", "random", "sorted", "reverse", "organ"])
def test_rank_sort_matches_std_sort(libs, kind):
    L, R = libs
    rng = np.random.default_rng(7)
    sizes = list(range(0, 70)) + [100, 127, 128, 129, 300, 1000, 1023, 4097]
    for n in sizes:
        s = np.ascontiguousarray(_tied_scores(rng, n, kind), np.float64)
        a = np.zeros(n, np.uint64)
        b = np.zeros(n, np.uint64)
        L.qro_rank_by_score(s, n, a)
        R.ref_rank_by_score(s, n, b)
        assert np.array_equal(a, b), (kind, n)


def test_known_tie_permutations(libs):
    """SURVEY.md Appendix A probes: all-equal keys, n = 17."""
    L, _ = libs
    a = np.zeros(17, np.uint64)
    L.qro_rank_by_score(np.zeros(17), 17, a)
    assert a.tolist() == [8, 16, 15, 14, 13, 12, 11, 10, 9, 0, 7, 6, 5, 4, 3, 2, 1]
    b = np.zeros(100, np.uint64)
    L.qro_rank_by_score(np.zeros(100), 100, b)
    assert b[:15].tolist() == [62, 74, 73, 72, 71, 70, 69, 68, 67, 66, 65, 64, 63, 75, 61]
    c = np.zeros(16, np.uint64)
    L.qro_rank_by_score(np.zeros(16), 16, c)
    assert c.tolist() == list(range(16))


def test_heapsort_fallback_matches_partial_sort(libs):
    L, R = libs
    rng = np.random.default_rng(3)
    for n in [0, 1, 2, 3, 17, 18, 64, 65, 257, 1000]:
        for kind in ["equal", "few", "random"]:
            s = np.ascontiguousarray(_tied_scores(rng, n, kind), np.float64)
            a = np.zeros(n, np.uint64)
            b = np.zeros(n, np.uint64)
            L.qro_heapsort_by_score(s, n, a)
            R.ref_heapsort_by_score(s, n, b)
            assert np.array_equal(a, b), (kind, n)


def test_label_sort(libs):
    L, R = libs
    rng = np.random.default_rng(5)
    for n in [0, 1, 5, 16, 17, 40, 100, 333]:
        lab = rng.integers(0, 5, n).astype(np.float32)
        lab2 = (lab + rng.random(n).astype(np.float32) * 0.9).astype(np.float32)  # non-integral: int-ties
        for x in (lab, lab2):
            a, b = x.copy(), x.copy()
            L.qro_sort_labels_desc_int(a, n)
            R.ref_sort_labels_desc_int(b, n)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("cutoff", [0, 1, 3, 10, 200])
def test_query_metric_and_jacobian(libs, metric, cutoff):
    L, R = libs
    rng = np.random.default_rng(11 + cutoff)
    for n in [1, 2, 5, 10, 11, 17, 40, 100]:
        for tied in (False, True):
            labels = rng.integers(0, 5, n).astype(np.float32)
            if n > 3 and rng.random() < 0.2:
                labels[:] = 0
            scores = rng.standard_normal(n)
            if tied:
                scores = np.round(scores)
            scores = np.ascontiguousarray(scores, np.float64)
            fn = L.qro_ndcg_query if metric else L.qro_dcg_query
            a = fn(labels, scores, n, cutoff)
            b = R.ref_eval_query(metric, labels, scores, n, cutoff)
            assert a == b, (n, tied, a, b)
            tri = n * (n + 1) // 2
            ja = np.zeros(tri)
            jb = np.zeros(tri)
            sl = np.zeros(n, np.float32)
            um = np.zeros(n, np.uint64)
            R.ref_jacobian(metric, labels, scores, n, cutoff, jb, sl, um)
            L.qro_jacobian(metric, sl, n, cutoff, ja)
            assert np.array_equal(ja.view(np.uint64), jb.view(np.uint64)), (n, tied)


@pytest.mark.parametrize("vertical", [0, 1])
def test_dataset_metric(libs, vertical, oracle_lib):
    L, R = libs
    x, labels, qoff = make_dataset(nq=37, docs_per_query=23, F=4, seed=2, ragged=True, adversarial=True)
    rng = np.random.default_rng(0)
    scores = np.round(rng.standard_normal(len(labels)), 1)
    for cutoff in (0, 5, 10):
        a = L.qro_eval_dataset(1, labels, scores, qoff, len(qoff) - 1, cutoff)
        b = R.ref_eval_dataset(1, labels, scores, qoff, len(qoff) - 1, cutoff, vertical)
        assert a == b


def test_argsort(libs):
    L, R = libs
    rng = np.random.default_rng(9)
    for n in [1, 2, 100, 5000]:
        v = rng.standard_normal(n).astype(np.float32)
        v[rng.integers(0, n, n // 4)] = 0.0
        v[rng.integers(0, n, n // 8)] = -0.0
        a = np.zeros(n, np.uint64)
        b = np.zeros(n, np.uint64)
        L.qro_argsort_f32(v, n, a)
        R.ref_argsort_f32(v, n, b)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nthr", [0, 8, 255])
@pytest.mark.parametrize("transform", [0, 1])
def test_histograms(libs, oracle_lib, nthr, transform):
    """RTRootHistogram (stmap, counts), update, child ctor, sibling subtraction."""
    L, R = libs
    x, labels, qoff = make_dataset(nq=30, docs_per_query=20, F=9, seed=4, adversarial=True)
    N, F = x.shape
    col = np.ascontiguousarray(x.T)
    thr, ts = oracle_lib.thresholds(col, nthr)
    cap = thr.shape[1]
    stmap, count0 = oracle_lib.binmap(col, thr, ts)
    rng = np.random.default_rng(1)
    lam = rng.standard_normal(N)
    left = np.sort(rng.choice(N, N // 3, replace=False)).astype(np.uint64)
    shp = (F, cap)
    r_stmap = np.zeros((F, N), np.uint32)
    r_c0 = np.zeros(shp, np.uint64)
    rs, rc = np.zeros(shp), np.zeros(shp, np.uint64)
    ls, lc = np.zeros(shp), np.zeros(shp, np.uint64)
    gs, gc = np.zeros(shp), np.zeros(shp, np.uint64)
    rss, lss, gss = C.c_double(), C.c_double(), C.c_double()
    R.ref_histograms(x, N, F, thr, ts, cap, lam, left, len(left), transform, r_stmap, r_c0,
                     rs, rc, C.byref(rss), ls, lc, C.byref(lss), gs, gc, C.byref(gss))
    assert np.array_equal(stmap, r_stmap)
    mask = np.arange(cap)[None, :] < ts[:, None]
    assert np.array_equal(count0[mask], r_c0[mask])
    os_, oc, oss = oracle_lib.hist_build(stmap, ts, cap, lam)
    assert np.array_equal(os_[mask].view(np.uint64), rs[mask].view(np.uint64))
    assert np.array_equal(oc[mask], rc[mask]) and oss == rss.value
    ols, olc, olss = oracle_lib.hist_build(stmap, ts, cap, lam, left)
    assert np.array_equal(ols[mask].view(np.uint64), ls[mask].view(np.uint64))
    assert np.array_equal(olc[mask], lc[mask]) and olss == lss.value
    ors, orc = np.zeros(shp), np.zeros(shp, np.uint64)
    L.qro_hist_subtract(F, ts, cap, os_, oc, ols, olc, ors, orc)
    assert np.array_equal(ors[mask].view(np.uint64), gs[mask].view(np.uint64))
    assert np.array_equal(orc[mask], gc[mask]) and (oss - olss) == gss.value


def _heap_ops(rng, n, kind):
    """A push / pop sequence like RegressionTree::fit's (two pushes per pop), with
    keys of the given tie structure.  ops[i] >= 0: push value ops[i]; -1: pop."""
    if kind == "equal":
        pool = np.zeros(n)
    elif kind == "few":
        pool = rng.integers(0, 3, n).astype(np.float64)
    elif kind == "zero_and_pos":     # many pure nodes (deviance 0) next to positive ones
        pool = np.where(rng.random(n) < 0.5, 0.0, rng.random(n))
    elif kind == "ascending":
        pool = np.arange(n, dtype=np.float64)
    elif kind == "descending":
        pool = -np.arange(n, dtype=np.float64)
    else:
        pool = rng.standard_normal(n)
    keys, ops, val, size = [], [], 0, 0
    for k in pool:
        keys.append(k)
        ops.append(val)
        val += 1
        size += 1
        if size >= 2 and rng.random() < 0.4:   # pop now and then (and never on empty)
            keys.append(0.0)
            ops.append(-1)
            size -= 1
    while size > 0:                            # drain: the full pop order
        keys.append(0.0)
        ops.append(-1)
        size -= 1
    return np.asarray(keys, np.float64), np.asarray(ops, np.int32)


@pytest.mark.parametrize("kind", ["equal", "few", "zero_and_pos", "ascending", "descending", "random"])
def test_maxheap_push_pop_order(libs, kind):
    """maxheap.h:58-88 decides the growth order of RegressionTree::fit, including
    among equal deviances: the restatement's heap against MaxHeap<int> itself, top and
    size after every operation."""
    L, R = libs
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 7, 10, 31, 64, 200, 1023):
        for initsize in (0, 10, n):
            keys, ops = _heap_ops(rng, n, kind)
            ta, tb = np.zeros(len(ops), np.int32), np.zeros(len(ops), np.int32)
            sa, sb = np.zeros(len(ops), np.uint64), np.zeros(len(ops), np.uint64)
            L.qro_heap_trace(keys, ops, len(ops), initsize, ta, sa)
            R.ref_heap_trace(keys, ops, len(ops), initsize, tb, sb)
            assert np.array_equal(ta, tb) and np.array_equal(sa, sb), (kind, n, initsize)


def test_symmatrix_index(libs):
    """symmatrix.h:29-89: the packed position of every (i, j), both triangles."""
    L, R = libs
    for size in (1, 2, 3, 16, 17, 100, 257):
        a, b = np.zeros((size, size), np.uint64), np.zeros((size, size), np.uint64)
        L.qro_sym_index(size, a)
        R.ref_sym_index(size, b)
        assert np.array_equal(a, b), size
        assert np.array_equal(a, a.T) and a.max() == size * (size + 1) // 2 - 1
