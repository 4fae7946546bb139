"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref: the
reference's own translation units, compiled unmodified by oracle/Makefile).
Only runs where /root/reference exists; the fixtures are data (inputs and the
reference's outputs), no reference source.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle  # noqa: E402
from datagen import make_dataset  # noqa: E402


def make(name, cutoff, nthr, score_kind, **case):
    R = oracle.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    x, labels, qoff = make_dataset(**case)
    N, F = x.shape
    rng = np.random.default_rng(123)
    if score_kind == "zero":
        scores = np.zeros(N)
    elif score_kind == "few":
        scores = rng.integers(0, 6, N) * 0.125
    else:
        scores = rng.standard_normal(N)
    ranks = np.zeros(N, np.uint64)
    per_q = np.zeros(len(qoff) - 1)
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        r = np.zeros(b - a, np.uint64)
        R.ref_rank_by_score(np.ascontiguousarray(scores[a:b]), b - a, r)
        ranks[a:b] = r
        per_q[q] = R.ref_eval_query(1, np.ascontiguousarray(labels[a:b]),
                                    np.ascontiguousarray(scores[a:b]), b - a, cutoff)
    ds = R.ref_eval_dataset(1, labels, scores, qoff, len(qoff) - 1, cutoff, 1)
    n0 = int(qoff[1])
    jac = np.zeros(n0 * (n0 + 1) // 2)
    sl = np.zeros(n0, np.float32)
    um = np.zeros(n0, np.uint64)
    R.ref_jacobian(1, np.ascontiguousarray(labels[:n0]), np.ascontiguousarray(scores[:n0]), n0,
                   cutoff, jac, sl, um)
    # thresholds come from Mart::init, which is not buildable here (pugixml): the
    # restatement supplies them; the bin map and histograms are the reference's.
    col = np.ascontiguousarray(x.T)
    thr, ts = oracle.thresholds(col, nthr)
    cap = thr.shape[1]
    lam = rng.standard_normal(N)
    left = np.sort(rng.choice(N, N // 3, replace=False)).astype(np.uint64)
    shp = (F, cap)
    stmap = np.zeros((F, N), np.uint32)
    c0 = np.zeros(shp, np.uint64)
    rs, rc = np.zeros(shp), np.zeros(shp, np.uint64)
    ls, lc = np.zeros(shp), np.zeros(shp, np.uint64)
    gs, gc = np.zeros(shp), np.zeros(shp, np.uint64)
    rss, lss, gss = C.c_double(), C.c_double(), C.c_double()
    R.ref_histograms(x, N, F, thr, ts, cap, lam, left, len(left), 0, stmap, c0, rs, rc,
                     C.byref(rss), ls, lc, C.byref(lss), gs, gc, C.byref(gss))
    np.savez_compressed(os.path.join(HERE, name), x=x, labels=labels, qoff=qoff, scores=scores,
                        cutoff=cutoff, nthresholds=nthr, ranks=ranks, ndcg_per_query=per_q,
                        ndcg_dataset=ds, jacobian_q0=jac, jac_sorted_labels=sl, stmap=stmap,
                        lam=lam, left_ids=left, root_sum=rs, root_count=rc, root_ss=rss.value,
                        left_sum=ls, left_count=lc, left_ss=lss.value)


def make_heap(name):
    """MaxHeap<int> traces (maxheap.h:58-88): push / pop sequences with equal, few-valued
    and random keys; top and size after every operation, from the reference's header."""
    R = oracle.ref()
    rng = np.random.default_rng(321)
    out = {}
    for kind in ("equal", "few", "random"):
        n = 40
        pool = {"equal": np.zeros(n), "few": rng.integers(0, 3, n).astype(np.float64),
                "random": rng.standard_normal(n)}[kind]
        keys, ops, size = [], [], 0
        for v, k in enumerate(pool):
            keys.append(k); ops.append(v); size += 1
            if size >= 2 and rng.random() < 0.4:
                keys.append(0.0); ops.append(-1); size -= 1
        keys += [0.0] * size
        ops += [-1] * size
        keys, ops = np.asarray(keys, np.float64), np.asarray(ops, np.int32)
        top, sz_ = np.zeros(len(ops), np.int32), np.zeros(len(ops), np.uint64)
        R.ref_heap_trace(keys, ops, len(ops), 10, top, sz_)
        out.update({f"{kind}_keys": keys, f"{kind}_ops": ops, f"{kind}_top": top, f"{kind}_size": sz_})
    idx = np.zeros((17, 17), np.uint64)
    R.ref_sym_index(17, idx)
    out["sym17"] = idx
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    oracle.build(ref=True)
    make_heap("heap_sym.npz")
    make("g1_ties_zero.npz", 10, 16, "zero", nq=12, docs_per_query=40, F=6, seed=31, ragged=True)
    make("g2_ties_few.npz", 10, 255, "few", nq=10, docs_per_query=50, F=8, seed=32, adversarial=True)
    make("g3_random.npz", 3, 8, "random", nq=8, docs_per_query=100, F=5, seed=33, ragged=True)
    print("golden fixtures written")
