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
    # the DCG train / test metric (dcg.cc:41-57) on the same rankings
    dcg_q = np.array([R.ref_eval_query(0, np.ascontiguousarray(labels[int(qoff[q]):int(qoff[q + 1])]),
                                       np.ascontiguousarray(scores[int(qoff[q]):int(qoff[q + 1])]),
                                       int(qoff[q + 1] - qoff[q]), cutoff) for q in range(len(qoff) - 1)])
    dcg_ds = R.ref_eval_dataset(0, labels, scores, qoff, len(qoff) - 1, cutoff, 1)
    n0 = int(qoff[1])
    jac = np.zeros(n0 * (n0 + 1) // 2)
    sl = np.zeros(n0, np.float32)
    um = np.zeros(n0, np.uint64)
    R.ref_jacobian(1, np.ascontiguousarray(labels[:n0]), np.ascontiguousarray(scores[:n0]), n0,
                   cutoff, jac, sl, um)
    # the jacobian of EVERY query under both metrics (ndcg.cc:60-93, dcg.cc:59-84: packed upper
    # triangles, one after the other) and QueryResults::sorted_labels with the cutoff
    # (queryresults.cc:55-62): what the pair loop of lambdamart.cc:104-141 consumes
    jac_off = np.zeros(len(qoff), np.uint64)
    jn, jd, slc_ = [], [], np.full(N, -1.0, np.float32)
    for q in range(len(qoff) - 1):
        a, b = int(qoff[q]), int(qoff[q + 1])
        n = b - a
        tri = n * (n + 1) // 2
        jac_off[q + 1] = jac_off[q] + tri
        for m, dst in ((1, jn), (0, jd)):
            j = np.zeros(max(tri, 1))
            R.ref_jacobian(m, np.ascontiguousarray(labels[a:b]), np.ascontiguousarray(scores[a:b]), n, cutoff, j,
                           np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.uint64))
            dst.append(j[:tri])
        d = np.full(max(n, 1), -1.0, np.float32)
        R.ref_sorted_labels(np.ascontiguousarray(labels[a:b]), np.ascontiguousarray(scores[a:b]), n, cutoff, d)
        slc_[a:b] = d[:n]
    jac_ndcg_all, jac_dcg_all = np.concatenate(jn), np.concatenate(jd)
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
    # A left child that a tree can actually produce: the documents on the left of the
    # best root split of the REFERENCE's root histogram under `lam` (the scan of
    # rt.cc:257-312 is not buildable here -- the restatement names (feature, slot); the
    # histograms of that id list and of its sibling are again the reference's own:
    # rtnode_histogram.cc:41-70 and 72-87).  tests/test_gpu_golden.py grows a two-leaf
    # tree on the device and compares ITS children with these.
    sp = oracle.split_find(rs, rc, ts, minls=1)
    sf, st = int(sp.feature), int(sp.thr_id)
    sleft = np.flatnonzero(stmap[sf] <= st).astype(np.uint64)
    assert len(sleft) == sp.lcount and 0 < len(sleft) < N
    sls, slc = np.zeros(shp), np.zeros(shp, np.uint64)
    srs, src = np.zeros(shp), np.zeros(shp, np.uint64)
    st2, c02 = np.zeros((F, N), np.uint32), np.zeros(shp, np.uint64)
    rs2, rc2 = np.zeros(shp), np.zeros(shp, np.uint64)
    rss2, slss, srss = C.c_double(), C.c_double(), C.c_double()
    R.ref_histograms(x, N, F, thr, ts, cap, lam, sleft, len(sleft), 0, st2, c02, rs2, rc2,
                     C.byref(rss2), sls, slc, C.byref(slss), srs, src, C.byref(srss))
    assert np.array_equal(st2, stmap) and np.array_equal(rs2.view(np.uint64), rs.view(np.uint64))
    # idx_radixsort of every column (radix.cc:35-73): stable ascending argsort
    argsort = np.zeros((F, N), np.uint64)
    for f in range(F):
        R.ref_argsort_f32(np.ascontiguousarray(col[f]), N, argsort[f])
    np.savez_compressed(os.path.join(HERE, name), x=x, labels=labels, qoff=qoff, scores=scores,
                        cutoff=cutoff, nthresholds=nthr, ranks=ranks, ndcg_per_query=per_q,
                        ndcg_dataset=ds, jacobian_q0=jac, jac_sorted_labels=sl, stmap=stmap,
                        lam=lam, left_ids=left, root_sum=rs, root_count=rc, root_ss=rss.value,
                        left_sum=ls, left_count=lc, left_ss=lss.value,
                        thr=thr, thr_size=ts, count0=c0, right_sum=gs, right_count=gc,
                        right_ss=gss.value, split_feature=sf, split_slot=st, split_left_ids=sleft,
                        split_left_sum=sls, split_left_count=slc, split_left_ss=slss.value,
                        split_right_sum=srs, split_right_count=src, split_right_ss=srss.value,
                        argsort=argsort.astype(np.uint32), dcg_per_query=dcg_q, dcg_dataset=dcg_ds,
                        jac_off=jac_off, jac_ndcg_all=jac_ndcg_all, jac_dcg_all=jac_dcg_all, sorted_labels_cut=slc_)


def make_svml(name):
    """An SVMLight text (every grammar case of svml.cc:38-161 the reference accepts:
    comment lines, ragged sparse rows, `#` trailers, repeated and non-monotone qids,
    exponents, negative zero) and what the REFERENCE's reader makes of it; plus a
    generated set as the reference's WRITER prints it (svml.cc:163-188)."""
    import tempfile
    R = oracle.ref()
    rng = np.random.default_rng(77)
    lines = ["# golden svml fixture", "2 qid:1 1:0.5 3:1.25 # doc one", "0 qid:1 2:-3e-2\t4:7",
             "   1   qid:1    1:1 2:2 3:3 4:4 5:5", "# another comment", "3 qid:2 5:0.125", "0 qid:2",
             "1 qid:7 1:1e10 2:-0", "2 qid:2 3:9.5 #trailing description 6:1"]
    qid = 10
    for _ in range(120):
        if rng.random() < 0.2:
            qid += int(rng.integers(1, 4))
        feats = np.sort(rng.choice(np.arange(1, 24), int(rng.integers(0, 9)), replace=False))
        toks = [f"{f}:{v:.7g}" for f, v in zip(feats, rng.standard_normal(len(feats)) * 10.0 ** int(rng.integers(-3, 4)))]
        tail = " # d%d" % _ if rng.random() < 0.3 else ""
        lines.append(f"{int(rng.integers(0, 5))} qid:{qid} " + " ".join(toks) + tail)
    text = ("\n".join(lines) + "\n").encode()
    sz_ = C.c_size_t
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "g.svml")
        open(p, "wb").write(text)
        N, F, Q = sz_(), sz_(), sz_()
        R.ref_svml_read(p.encode(), C.byref(N), C.byref(F), C.byref(Q), None, None, None)
        x = np.zeros((N.value, F.value), np.float32)
        lab = np.zeros(N.value, np.float32)
        qoff = np.zeros(Q.value + 1, np.uint64)
        R.ref_svml_read(p.encode(), C.byref(N), C.byref(F), C.byref(Q), x.ctypes.data, lab.ctypes.data,
                        qoff.ctypes.data)
        wx, wl, wq = make_dataset(nq=9, docs_per_query=7, F=11, seed=4, ragged=True, adversarial=True)
        p2 = os.path.join(d, "w.svml")
        R.ref_svml_write(p2.encode(), wx, wl, wq, len(wq) - 1, wx.shape[1])
        written = np.frombuffer(open(p2, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(HERE, name), text=np.frombuffer(text, np.uint8), x=x, labels=lab,
                        qoff=qoff, w_x=wx, w_labels=wl, w_qoff=wq, w_text=written)


def make_layout(name):
    """Dataset::addInstance over a qid column with repeated and non-monotone ids (dataset.cc:63-87)
    and the VerticalDataset built from it (vertical_dataset.cc): query offsets and the
    feature-major matrix as the REFERENCE lays them out."""
    R = oracle.ref()
    rng = np.random.default_rng(99)
    N, F = 211, 7
    x = rng.standard_normal((N, F)).astype(np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    runs = rng.integers(1, 9, 60)
    ids = rng.integers(1, 12, 60)             # few values: ids repeat and go up and down
    ids[5] = ids[4]                           # two consecutive runs of ONE id are one query
    qids = np.repeat(ids, runs)[:N].astype(np.uint32)
    assert len(qids) == N
    offs = np.zeros(N + 1, np.uint64)
    vert = np.zeros((F, N), np.float32)
    voffs = np.zeros(N + 1, np.uint64)
    nq = R.ref_dataset_layout(x, labels, qids, N, F, offs, vert.ctypes.data, voffs.ctypes.data)
    np.savez_compressed(os.path.join(HERE, name), x=x, labels=labels, qids=qids, qoff=offs[:nq + 1],
                        vertical=vert, vertical_qoff=voffs[:nq + 1])


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
    make_layout("layout.npz")
    make_svml("svml.npz")
    make("g1_ties_zero.npz", 10, 16, "zero", nq=12, docs_per_query=40, F=6, seed=31, ragged=True)
    make("g2_ties_few.npz", 10, 255, "few", nq=10, docs_per_query=50, F=8, seed=32, adversarial=True)
    make("g3_random.npz", 3, 8, "random", nq=8, docs_per_query=100, F=5, seed=33, ragged=True)
    # the query sizes of SURVEY.md Appendix A's tie-order probes (17, 40, 100) with equal,
    # few-valued and distinct scores side by side, 100 features in three blocks
    make("g4_probe_sizes.npz", 10, 63, "few", nq=9, docs_per_query=0, F=100, seed=34, sizes=(17, 40, 100, 17, 40, 100, 16, 1, 257))
    print("golden fixtures written")
