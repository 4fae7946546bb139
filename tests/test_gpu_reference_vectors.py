"""The reference's OWN known-answer vectors, fed to the DEVICE through the C-ABI.

catch-unit-tests/metric/ir/test-dcg.cc:28-99 and test-ndcg.cc:31-106 hold one list -- labels
{3,2,1,0,0}, scores {5,4,3,2,1} -- and assert on it: DCG / NDCG with the cutoff above the list's
length, at 2, at 0 (= none) and NO_CUTOFF; that `jacobian->at(0,2)` equals the change of the metric
when the scores of ranks 0 and 2 are swapped, without a cutoff and with the cutoff in the middle of
the swap; and the closed form of the latter, `2^l2 - 2^l0` (over the ideal DCG for NDCG).

tests/test_oracle_golden.py restates them against the oracle on the CPU; here the same numbers go
through qr_lambda_compute / qr_metric_eval / qr_metric_per_query, and the device's pseudo-responses
(lambdamart.cc:104-141) are checked against the swap deltas the tests define the jacobian by.
Then the fixtures' jacobians -- every query of tests/golden/g*.npz, both metrics, the REFERENCE's
bits (ndcg.cc:60-93, dcg.cc:59-84 compiled from its own files) -- and the reference's rank order feed
the pair loop, restated below in twenty lines of numpy: the device's lambdas and weights have to be
what that loop makes of the reference's own intermediate results.  No oracle call on the way."""
import glob
import os
from math import exp, log2

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "g[0-9]_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]

LABELS = np.array([3, 2, 1, 0, 0], np.float32)
SCORES = np.array([5, 4, 3, 2, 1], np.float64)
FULL = (2 ** 3 - 1) + (2 ** 2 - 1) / log2(3) + (2 ** 1 - 1) / 2      # test-dcg.cc:39-41
AT2 = (2 ** 3 - 1) + (2 ** 2 - 1) / log2(3)                          # test-dcg.cc:45-47
NO_CUTOFF = 0          # Metric::set_cutoff(0) == NO_CUTOFF (metric.h:65-67); the C-ABI takes 0


@pytest.fixture(scope="module")
def qr():
    import quickrank_amd
    from quickrank_amd import build
    build.build()
    return quickrank_amd


def _one_query(qr, labels, scores):
    c = qr.Context(0)
    n = len(labels)
    x = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    c.upload(x, np.asarray(labels, np.float32), np.array([0, n], np.uint64))
    c.set_scores(np.asarray(scores, np.float64))
    return c


def _metric(c, metric, cutoff, scores):
    """The metric of `scores` three ways: the lambda pass's own figure, its per-query array, qr_metric_eval."""
    c.set_scores(scores)
    c.compute_lambdas(metric, cutoff)
    a, b = c.metric_last(), float(c.metric_per_query()[0])
    e = c.metric_eval(0, metric, cutoff)
    assert a == b == e, (a, b, e)
    return e


@pytest.mark.parametrize("cutoff,want", [(5, FULL), (2, AT2), (0, FULL), (7, FULL)],
                         ids=["at5", "at2", "cutoff0_is_none", "beyond_the_list"])
def test_dcg_known_answers_on_the_device(qr, cutoff, want):
    """test-dcg.cc:36-58."""
    c = _one_query(qr, LABELS, SCORES)
    assert _metric(c, "DCG", cutoff, SCORES) == pytest.approx(want, rel=1e-15)
    c.close()


@pytest.mark.parametrize("cutoff", [5, 0, 2], ids=["at5", "cutoff0_is_none", "at2"])
def test_ndcg_known_answers_on_the_device(qr, cutoff):
    """test-ndcg.cc:39-66: the list is in ideal order, NDCG is DCG / IDCG = 1 at every cutoff."""
    c = _one_query(qr, LABELS, SCORES)
    assert _metric(c, "NDCG", cutoff, SCORES) == pytest.approx(1.0, rel=1e-15)
    c.close()


def _swap_delta(c, metric, cutoff, i, j):
    s = SCORES.copy()
    base = _metric(c, metric, cutoff, s)
    s[i], s[j] = s[j], s[i]
    return _metric(c, metric, cutoff, s) - base


@pytest.mark.parametrize("metric", ["DCG", "NDCG"])
def test_swap_delta_closed_form_on_the_device(qr, metric):
    """test-dcg.cc:78-97 / test-ndcg.cc:86-104: cutoff 2, ranks 0 and 2 swapped -- rank 2 lies beyond
    the cutoff (no discount), rank 0 has discount 1: the change is 2^l2 - 2^l0 (over IDCG@2)."""
    c = _one_query(qr, LABELS, SCORES)
    want = (2 ** 1 - 2 ** 3) / (AT2 if metric == "NDCG" else 1.0)
    assert _swap_delta(c, metric, 2, 0, 2) == pytest.approx(want, rel=1e-14)
    c.close()


@pytest.mark.parametrize("cutoff", [NO_CUTOFF, 2, 5, 1], ids=["no_cutoff", "at2", "at5", "at1"])
@pytest.mark.parametrize("metric", ["DCG", "NDCG"])
def test_lambdas_are_the_swap_deltas(qr, metric, cutoff):
    """The reference's tests DEFINE jacobian->at(j, k) as the metric's change under a swap of ranks j
    and k (test-ndcg.cc:70-84); lambdamart.cc:104-141 turns |at(j, k)| into pseudo-responses.  Here the
    swap deltas are measured on the device (metric after - metric before, every pair), the pair loop is
    run over them on the host, and the device's own lambdas / weights of the list have to be that."""
    c = _one_query(qr, LABELS, SCORES)
    n = len(LABELS)
    k = n if cutoff == 0 else cutoff
    lam, w = np.zeros(n), np.zeros(n)
    for j in range(n):           # (the list is in rank order: rank == position)
        for q in range(n):
            if q == j or (j >= k and q >= k) or not LABELS[j] > LABELS[q]:
                continue
            d = abs(_swap_delta(c, metric, cutoff, j, q))
            rho = 1.0 / (1.0 + exp(SCORES[j] - SCORES[q]))
            lam[j] += rho * d
            lam[q] -= rho * d
            w[j] += rho * (1.0 - rho) * d
            w[q] += rho * (1.0 - rho) * d
    c.set_scores(SCORES)
    c.compute_lambdas(metric, cutoff)
    got_l, got_w = c.get_pseudo()
    assert np.allclose(got_l, lam, rtol=1e-12, atol=1e-14), (got_l, lam)
    assert np.allclose(got_w, w, rtol=1e-12, atol=1e-14), (got_w, w)
    assert np.any(lam != 0.0)
    c.close()


@pytest.mark.parametrize("metric", ["DCG", "NDCG"])
def test_two_document_query_lambda(qr, metric):
    """The smallest pair: labels (3, 1), cutoff 1.  Rank 1 lies beyond the cutoff, so the swap delta is
    the closed form the reference's tests assert, 2^l1 - 2^l0 (over IDCG@1 = 2^l0 - 1 for NDCG), and
    the pseudo-responses are +-rho |delta|, the weights rho (1 - rho) |delta| (lambdamart.cc:127-136)."""
    s = np.array([0.3, -0.2])
    c = _one_query(qr, [3, 1], s)
    c.compute_lambdas(metric, 1)
    lam, w = c.get_pseudo()
    delta = abs(2 ** 1 - 2 ** 3) / ((2 ** 3 - 1) if metric == "NDCG" else 1.0)
    rho = 1.0 / (1.0 + exp(s[0] - s[1]))
    assert lam[0] == pytest.approx(rho * delta, rel=1e-14) and lam[1] == pytest.approx(-rho * delta, rel=1e-14)
    assert w[0] == pytest.approx(rho * (1 - rho) * delta, rel=1e-14) and w[0] == w[1]
    # the worse document on top: the same pair, the other way round in rank
    c.set_scores(-s)
    c.compute_lambdas(metric, 1)
    lam2, _ = c.get_pseudo()
    rho2 = 1.0 / (1.0 + exp(-s[0] + s[1]))
    assert lam2[0] == pytest.approx(rho2 * delta, rel=1e-14) and lam2[1] == pytest.approx(-rho2 * delta, rel=1e-14)
    c.close()


def _pair_loop(labels, scores, ranks, jac, cutoff):
    """lambdamart.cc:104-141 over a jacobian given as SymMatrix's packed upper triangle
    (symmatrix.h:29-89), ranks = RankedResults::pos_of_rank."""
    n = len(labels)
    lam, w = np.zeros(n), np.zeros(n)
    sl = labels[ranks]
    for j in range(n):
        ja = int(ranks[j])
        for k in range(n):
            if k == j:
                continue
            if j >= cutoff and k >= cutoff:
                break
            if sl[j] > sl[k]:
                a, b = (j, k) if j < k else (k, j)
                d = abs(jac[a * n - (a - 1) * a // 2 + b - a])
                ka = int(ranks[k])
                rho = 1.0 / (1.0 + exp(scores[ja] - scores[ka]))
                lam[ja] += rho * d
                lam[ka] -= rho * d
                w[ja] += rho * (1.0 - rho) * d
                w[ka] += rho * (1.0 - rho) * d
    return lam, w


@pytest.mark.parametrize("metric", ["NDCG", "DCG"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_lambdas_from_the_references_jacobian(qr, path, metric):
    """Every query of the golden fixtures: the device's pseudo-responses and weights against the pair
    loop run over the REFERENCE's jacobian and rank permutation (queries of 1..257 documents, equal /
    few-valued / distinct scores).  Documents of equal score beyond the cutoff may stand in another
    order on the device; their pairs' deltas do not depend on it (no discount beyond the cutoff)."""
    g = np.load(path)
    labels, scores, qoff, cutoff = g["labels"], g["scores"], g["qoff"], int(g["cutoff"])
    jac_all, jac_off = g["jac_ndcg_all" if metric == "NDCG" else "jac_dcg_all"], g["jac_off"]
    c = qr.Context(0)
    c.upload(g["x"], labels, qoff)
    c.set_scores(scores)
    c.compute_lambdas(metric, cutoff)
    got_l, got_w = c.get_pseudo()
    c.close()
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        lam, w = _pair_loop(labels[a:b], scores[a:b], g["ranks"][a:b].astype(np.int64),
                            jac_all[int(jac_off[q]):int(jac_off[q + 1])], cutoff)
        scale = max(np.abs(w).max(), 1e-300)
        assert np.allclose(got_l[a:b], lam, rtol=1e-11, atol=1e-13 * scale), (q, b - a)
        assert np.allclose(got_w[a:b], w, rtol=1e-11, atol=1e-13 * scale), (q, b - a)


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_sorted_labels_with_cutoff_are_the_references(qr, path):
    """QueryResults::sorted_labels(scores, dest, cutoff) (queryresults.cc:55-62): the labels of the
    first `cutoff` ranks, from the device's rank permutation."""
    g = np.load(path)
    labels, qoff, cutoff = g["labels"], g["qoff"], int(g["cutoff"])
    c = qr.Context(0)
    c.upload(g["x"], labels, qoff)
    c.set_scores(g["scores"])
    c.compute_lambdas("NDCG", cutoff)
    ranks = c.ranks().astype(np.int64)
    c.close()
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        m = min(cutoff, b - a)
        assert np.array_equal(labels[a:b][ranks[a:a + m]], g["sorted_labels_cut"][a:a + m]), q
