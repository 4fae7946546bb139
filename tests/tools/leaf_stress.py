"""Many-leaved first trees on fresh random sets, every leaf value checked against the device's OWN
pseudo-responses summed on the host over the documents that reach the leaf (TEST TOOL, GPU box; no
oracle needed) -- the shape of the one mismatch round 5's hunt met (64 leaves: the leaf sums by
position lists), thousands of times in one process:
    [QR_DEBUG=1] python tests/tools/leaf_stress.py TREES [nleaves]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from datagen import make_dataset
from quickrank_amd import Context

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 500
nleaves = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rng = np.random.default_rng(12345)
bad = 0
done = 0
while done < trees:
    nq, dpq, F = int(rng.integers(100, 500)), int(rng.choice([16, 17, 40])), int(rng.choice([5, 9, 16]))
    x, labels, qoff = make_dataset(nq=nq, docs_per_query=dpq, F=F, seed=int(rng.integers(1 << 30)), ragged=True)
    N = len(labels)
    c = Context(0); c.upload(x, labels, qoff); c.build_bins(64)
    bins = c.read_bins().astype(np.int64)                       # [N][F]
    for rep in range(8):                                        # a few trees per set: scores move on
        if rep == 0:
            c.reset_scores()
        c.compute_lambdas("NDCG", 10)
        nodes = c.fit_tree(nleaves, 2, True)
        lam, w = c.get_pseudo()
        at = np.zeros(N, np.int64)
        while True:
            nd = nodes[at]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = bins[idx, nd["feature"][idx]] <= nd["thr_id"][idx]
            at[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        for k in np.nonzero(nodes["feature"] < 0)[0]:
            ids = np.nonzero(at == k)[0]
            s1, s2 = lam[ids].sum(), w[ids].sum()
            want = s1 / s2 if s2 >= 2.220446049250313e-16 else 0.0
            if len(ids) != nodes[k]["nsamples"] or not np.isclose(nodes[k]["value"], want, rtol=1e-9, atol=1e-12):
                bad += 1
                print(f"MISMATCH tree {done} (N {N} F {F}) leaf {k}: device {float(nodes[k]['value'])!r} n {int(nodes[k]['nsamples'])}; "
                      f"host over its own lambdas {float(want)!r} n {len(ids)}; list {np.array_equal(np.sort(c.node_samples(int(k))), ids)}", flush=True)
        c.update_scores(0.1)
        done += 1
        if done >= trees:
            break
    c.close()
print(f"{done} trees of {nleaves} leaves: {bad} leaf mismatches", flush=True)
