"""The HIP path (through the C-ABI) against OUTPUTS OF THE REFERENCE ITSELF.

tests/golden/g*.npz hold, for seeded inputs, what the reference's own translation
units produced in the build container (tests/golden/make_golden.py drives
oracle/_ref, which never travels): rank permutations incl. the std::sort tie order
(queryresults.cc:47-53), per-query and dataset NDCG and DCG (ndcg.cc:49-93, dcg.cc:41-57,
metric.h:77-106),
the bin map `stmap` (rtnode_histogram.cc:227-253), the root histogram
(rtnode_histogram.cc:172-204), a child built from an id list and its sibling by
subtraction (rtnode_histogram.cc:41-87), the stable argsort of every column
(radix.cc:35-73).  Every other `-m gpu` test compares the device with the C
restatement; these compare it with the reference's bits directly -- no oracle call
on the way (the `oracle_lib` fixture is not requested)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "g[0-9]_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]
QR_MAX_BINS = 256


@pytest.fixture(scope="module")
def qr():
    import quickrank_amd
    from quickrank_amd import build
    build.build()
    return quickrank_amd


def _padded_thr(g):
    thr, ts = g["thr"], g["thr_size"].astype(np.uint32)
    out = np.full((thr.shape[0], QR_MAX_BINS), np.finfo(np.float32).max, np.float32)
    for f in range(thr.shape[0]):
        out[f, :ts[f]] = thr[f, :ts[f]]
    return out, ts


def _ctx(qr, g, given_thresholds):
    c = qr.Context(0)
    c.upload(g["x"], g["labels"], g["qoff"])
    if given_thresholds:
        thr, ts = _padded_thr(g)
        c.build_bins_with(thr, ts)
    else:
        c.build_bins(int(g["nthresholds"]))
    return c


def test_fixtures_present():
    assert len(GOLDEN) >= 4, "tests/golden/g*.npz did not travel"


@pytest.mark.parametrize("given", [True, False], ids=["thresholds_given", "thresholds_built"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_bin_map_is_the_references_stmap(qr, path, given):
    """RTRootHistogram's stmap (rtnode_histogram.cc:227-253), bit for bit, through
    qr_bins_build_with (the fixture's thresholds handed over) and through qr_bins_build
    (thresholds made on the device from the raw columns)."""
    g = np.load(path)
    c = _ctx(qr, g, given)
    bins = c.read_bins()
    assert np.array_equal(bins.T.astype(np.uint32), g["stmap"])
    if not given:
        thr, ts = c.thresholds()
        assert np.array_equal(ts.astype(np.uint64), g["thr_size"])
        for f in range(len(ts)):
            n = int(ts[f])
            assert np.array_equal(thr[f, :n].view(np.uint32), g["thr"][f, :n].view(np.uint32)), f
    c.close()


@pytest.mark.parametrize("exact_tail", [True, False], ids=["whole_permutation", "visible_ranks"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_ranks_and_ndcg_are_the_references(qr, path, exact_tail, monkeypatch):
    """queryresults.cc:47-53 (std::sort of the index array, tie order included: the fixtures
    hold all-equal, few-valued and distinct scores on queries of 1..257 documents, among
    them SURVEY Appendix A's probe sizes 17 / 40 / 100), ndcg.cc:49-93 per query bitwise,
    metric.h:93-106 dataset mean to 1e-13."""
    if exact_tail:
        monkeypatch.setenv("QR_EXACT_TAIL", "1")
    else:
        monkeypatch.delenv("QR_EXACT_TAIL", raising=False)
    g = np.load(path)
    labels, scores, qoff, cutoff = g["labels"], g["scores"], g["qoff"], int(g["cutoff"])
    c = qr.Context(0)
    c.upload(g["x"], labels, qoff)
    c.set_scores(scores)
    c.compute_lambdas("NDCG", cutoff)
    got, want = c.ranks().astype(np.uint64), g["ranks"]
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        size = b - a if exact_tail else min(cutoff, b - a)
        assert np.array_equal(got[a:a + size], want[a:a + size]), q
        if size < b - a:    # beyond the cutoff: the same documents, the same scores in order
            assert np.array_equal(np.sort(got[a + size:b]), np.sort(want[a + size:b])), q
            assert np.array_equal(scores[a:b][got[a + size:b].astype(np.int64)],
                                  scores[a:b][want[a + size:b].astype(np.int64)]), q
    pq = c.metric_per_query()
    assert np.array_equal(pq.view(np.uint64), g["ndcg_per_query"].view(np.uint64))
    assert c.metric_last() == pytest.approx(float(g["ndcg_dataset"]), rel=1e-13)
    assert c.metric_eval(0, "NDCG", cutoff) == pytest.approx(float(g["ndcg_dataset"]), rel=1e-13)
    c.close()


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_dcg_is_the_references(qr, path):
    """The DCG metric (`--train-metric DCG`, dcg.cc:41-57) on the fixtures' rankings: per query
    bitwise, the dataset mean to 1e-13 -- through the lambda pass (which emits the metric of the
    scores it ranks) and through qr_metric_eval."""
    g = np.load(path)
    cutoff = int(g["cutoff"])
    c = qr.Context(0)
    c.upload(g["x"], g["labels"], g["qoff"])
    c.set_scores(g["scores"])
    c.compute_lambdas("DCG", cutoff)
    pq = c.metric_per_query()
    assert np.array_equal(pq.view(np.uint64), g["dcg_per_query"].view(np.uint64))
    assert c.metric_last() == pytest.approx(float(g["dcg_dataset"]), rel=1e-13)
    assert c.metric_eval(0, "DCG", cutoff) == pytest.approx(float(g["dcg_dataset"]), rel=1e-13)
    c.close()


def _cmp_hist(c, node, want_sum, want_count, ts, tol, tag):
    hs, hc = c.node_hist(node)
    for f in range(len(ts)):
        n = int(ts[f])
        assert np.array_equal(hc[f, :n], want_count[f, :n]), (tag, f)
        assert np.allclose(hs[f, :n], want_sum[f, :n], rtol=0, atol=tol), (tag, f)


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_histograms_are_the_references(qr, path):
    """RTRootHistogram::update (rtnode_histogram.cc:172-204) and the two children of the
    best root split: the directly built one (rtnode_histogram.cc:41-70) and its sibling by
    subtraction (72-87).  The device grows a two-leaf tree on the fixture's pseudo-responses;
    its left child must hold exactly the fixture's id list, and all three histograms the
    reference's cumulative counts (exact) and sums (to the 2^-30 fixed-point resolution)."""
    g = np.load(path)
    lam = g["lam"]
    ts = g["thr_size"]
    c = _ctx(qr, g, True)
    c.set_pseudo(lam, np.ones_like(lam))
    nodes = c.fit_tree(2, 1, True)
    tol = 2.0 ** -30 * max(1.0, np.abs(lam).max()) * np.sqrt(len(lam))
    _cmp_hist(c, 0, g["root_sum"], g["root_count"], ts, tol, "root")
    assert len(nodes) == 3 and nodes[0]["feature"] >= 0
    left, right = int(nodes[0]["left"]), int(nodes[0]["right"])
    ids = np.sort(c.node_samples(left).astype(np.uint64))
    assert np.array_equal(ids, g["split_left_ids"])
    if (int(nodes[0]["feature"]), int(nodes[0]["thr_id"])) != (int(g["split_feature"]), int(g["split_slot"])):
        # an equal-partition candidate named differently: same two sets
        f, t = int(nodes[0]["feature"]), int(nodes[0]["thr_id"])
        assert np.array_equal(np.flatnonzero(g["stmap"][f] <= t).astype(np.uint64), g["split_left_ids"])
    assert nodes[left]["nsamples"] == len(ids) and nodes[right]["nsamples"] == len(lam) - len(ids)
    _cmp_hist(c, left, g["split_left_sum"], g["split_left_count"], ts, tol, "left")
    _cmp_hist(c, right, g["split_right_sum"], g["split_right_count"], ts, tol, "right")
    # node statistics the reference keeps beside a histogram: squares_sum_ -> deviance
    # (rtnode.h:97-107): sum(lam^2) - sum(lam)^2 / n on the same id lists
    for node, ss, hsum in ((0, float(g["root_ss"]), g["root_sum"]),
                           (left, float(g["split_left_ss"]), g["split_left_sum"]),
                           (right, float(g["split_right_ss"]), g["split_right_sum"])):
        n = float(nodes[node]["nsamples"])
        tot = float(hsum[0, int(ts[0]) - 1])
        want = ss - tot * tot / n
        assert nodes[node]["deviance"] == pytest.approx(want, rel=1e-6, abs=1e-6 * ss), node
    c.close()


@pytest.mark.parametrize("path", GOLDEN[:3], ids=IDS[:3])
def test_presorted_lists_follow_the_references_argsort(qr, path):
    """idx_radixsort (radix.cc:35-73) is a stable ascending argsort.  The device has no
    stand-alone argsort; its consumers are the thresholds (above) and -- for
    `--num-thresholds 0` -- the wide bin map, whose slot of a document is the rank of its
    value among the column's distinct values.  Walking the reference's permutation must
    therefore meet the device's slots in non-decreasing order, stepping exactly where the
    value changes."""
    g = np.load(path)
    x = g["x"]
    c = qr.Context(0)
    c.upload(x, g["labels"], g["qoff"])
    c.build_bins(0, wide=True)
    bins = c.read_bins_u32()
    for f in range(x.shape[1]):
        order = g["argsort"][f].astype(np.int64)
        slots = bins[order, f].astype(np.int64)
        vals = x[order, f]
        step = np.diff(slots)
        assert np.all(step >= 0) and np.array_equal(step > 0, np.diff(vals) > 0), f
        assert slots[0] == 0 and np.all(step <= 1), f
    c.close()
