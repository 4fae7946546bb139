"""The classifiers of the randomised sweep (tests/tools/fuzz_parity.py) on hand-made trees: they
decide which differences between a device tree and the oracle's are "the reference's rounding
noise" -- a checker that accepted too much would hide device errors, so each is shown a case it
must accept and the nearest case it must refuse.  No GPU, no oracle library."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fuzz_parity import (deviance_order_tie, oblivious_level_gain_tie, parting_gains_exact,  # noqa: E402
                         score_tie_before, upstream_gain_tie)
from quickrank_amd._capi import NODE_DTYPE  # noqa: E402


def tree(spec):
    """spec: list of (feature, thr_id, left, right, deviance, nsamples); feature -1 = leaf."""
    n = np.zeros(len(spec), NODE_DTYPE)
    for i, (f, t, l, r, dev, cnt) in enumerate(spec):
        n[i]["feature"], n[i]["thr_id"], n[i]["left"], n[i]["right"] = f, t, l, r
        n[i]["deviance"], n[i]["nsamples"] = dev, cnt
    return n


# eight documents, three features (bin ids): feature 0 cuts {0..3 | 4..7}, feature 1 the same set
# mirrored, feature 2 cuts {0,1,4,5 | 2,3,6,7}
STMAP = np.array([[0, 0, 0, 0, 1, 1, 1, 1],
                  [1, 1, 1, 1, 0, 0, 0, 0],
                  [0, 0, 1, 1, 0, 0, 1, 1]], np.uint32)


def test_upstream_gain_tie_accepts_equal_child_deviances_only():
    o = tree([(0, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 3.0, 4), (-1, 0, -1, -1, 2.0, 4)])
    same_gain = tree([(2, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 1.5, 4), (-1, 0, -1, -1, 3.5, 4)])
    worse = tree([(2, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 1.5, 4), (-1, 0, -1, -1, 3.6, 4)])
    mirrored = tree([(1, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 2.0, 4), (-1, 0, -1, -1, 3.0, 4)])
    assert upstream_gain_tie(STMAP, o, same_gain)          # another partition, the same total
    assert not upstream_gain_tie(STMAP, o, worse)          # another partition with less gain: an error
    assert not upstream_gain_tie(STMAP, o, mirrored)       # the same partition: nothing parted here


def test_upstream_gain_tie_finds_the_node_below_a_mirrored_split():
    # the root is cut alike (mirrored on the device); the trees part in the child holding documents 0..3
    o = tree([(0, 0, 1, 2, 10.0, 8), (2, 0, 3, 4, 4.0, 4), (-1, 0, -1, -1, 2.0, 4),
              (-1, 0, -1, -1, 1.0, 2), (-1, 0, -1, -1, 1.0, 2)])
    g = tree([(1, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 2.0, 4), (-1, 0, -1, -1, 4.0, 4)])
    assert not upstream_gain_tie(STMAP, o, g)               # a split on one side only is not a gain tie ...
    assert not deviance_order_tie(STMAP, o, g)              # ... and nothing on the other side pairs with it


def test_deviance_order_tie_pairs_the_one_sided_splits():
    # both cut the root alike; the oracle spent its last split on the left child, the device on the right
    o = tree([(0, 0, 1, 2, 10.0, 8), (2, 0, 3, 4, 3.0, 4), (-1, 0, -1, -1, 3.0, 4),
              (-1, 0, -1, -1, 1.0, 2), (-1, 0, -1, -1, 1.0, 2)])
    g = tree([(0, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 3.0, 4), (2, 0, 3, 4, 3.0 + 1e-13, 4),
              (-1, 0, -1, -1, 1.0, 2), (-1, 0, -1, -1, 1.0, 2)])
    assert deviance_order_tie(STMAP, o, g)
    g[2]["deviance"] = 2.9                                  # the device split the LESS deviant node: an error
    assert not deviance_order_tie(STMAP, o, g)
    same = tree([(0, 0, 1, 2, 10.0, 8), (2, 0, 3, 4, 3.0, 4), (-1, 0, -1, -1, 3.0, 4),
                 (-1, 0, -1, -1, 1.0, 2), (-1, 0, -1, -1, 1.0, 2)])
    assert not deviance_order_tie(STMAP, o, same)           # identical trees: nothing to explain


def test_oblivious_level_gain_tie_prices_both_candidates_exactly():
    # one level: feature 0 against feature 2 on pseudo-responses that make the two gains equal / unequal
    o = tree([(0, 0, 1, 2, 0.0, 8), (-1, 0, -1, -1, 0.0, 4), (-1, 0, -1, -1, 0.0, 4)])
    g = tree([(2, 0, 1, 2, 0.0, 8), (-1, 0, -1, -1, 0.0, 4), (-1, 0, -1, -1, 0.0, 4)])
    tie = np.array([1.0, 0.0, 0.0, 1.0, 0.0, 1.0, 1.0, 0.0])     # both candidates: left sum 2, right sum 2
    assert oblivious_level_gain_tie(STMAP, o, g, tie, 1)
    notie = np.array([1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])   # feature 0: 2 | 0, feature 2: 2 | 0 ... on other sets
    notie[4] = 0.5                                               # feature 0: 2 | 0.5, feature 2: 2.5 | 0
    assert not oblivious_level_gain_tie(STMAP, o, g, notie, 1)
    assert not oblivious_level_gain_tie(STMAP, o, o, tie, 1)     # the same candidate: nothing parted


def test_score_tie_before_wants_a_rounding_sized_gap_inside_one_query():
    # one tree of two leaves; documents 0..3 get value a, 4..7 value b
    def model(a, b):
        t = tree([(0, 0, 1, 2, 1.0, 8), (-1, 0, -1, -1, 0.0, 4), (-1, 0, -1, -1, 0.0, 4)])
        t[1]["value"], t[2]["value"] = a, b
        return {"nodes": t[None], "nnodes": np.array([3])}
    qoff = np.array([0, 8], np.uint64)
    assert score_tie_before(STMAP, model(1.0, 1.0 + 2.3e-16), 1, 1.0, qoff)       # one ulp apart
    assert not score_tie_before(STMAP, model(1.0, 1.0), 1, 1.0, qoff)             # exactly equal: same ranks on both sides
    assert not score_tie_before(STMAP, model(1.0, 1.0 + 1e-9), 1, 1.0, qoff)      # a real difference
    two = np.array([0, 4, 8], np.uint64)                                          # the pair sits in DIFFERENT queries
    assert not score_tie_before(STMAP, model(1.0, 1.0 + 2.3e-16), 1, 1.0, two)
    assert not score_tie_before(STMAP, model(1.0, 1.0 + 2.3e-16), 0, 1.0, qoff)   # going into tree 0 all scores are 0


def test_parting_gains_exact_prices_the_node_where_the_trees_part():
    o = tree([(0, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 3.0, 4), (-1, 0, -1, -1, 2.0, 4)])
    g = tree([(2, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 1.5, 4), (-1, 0, -1, -1, 3.5, 4)])
    tie = np.array([0.5, 0.0, 1.5, 1.0, 0.5, 0.0, 0.5, 0.0])       # feature 0 cuts 3 | 1, feature 2 cuts 1 | 3
    rel, docs = parting_gains_exact(STMAP, o, g, tie)
    assert rel == 0.0 and docs == 8
    near = tie.copy()
    near[0] += 2.0 ** -40                                           # (3 + e)^2 + 1 against (1 + e)^2 + 9: apart by 4e
    rel, _ = parting_gains_exact(STMAP, o, g, near)
    assert 0.0 < rel < 1e-9
    far = tie.copy()
    far[0] = 1.0
    rel, _ = parting_gains_exact(STMAP, o, g, far)
    assert rel > 1e-3
    assert parting_gains_exact(STMAP, o, o, tie) is None            # the same cuts everywhere: nothing parted
    mirrored = tree([(1, 0, 1, 2, 10.0, 8), (-1, 0, -1, -1, 2.0, 4), (-1, 0, -1, -1, 3.0, 4)])
    assert parting_gains_exact(STMAP, o, mirrored, tie) is None     # the same partition, mirrored
