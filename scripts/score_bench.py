"""Ensemble-scoring micro-benchmark (BASELINE.json config 5 shape, scaled):
T trees x 64 leaves (depth 6, thresholds drawn from a 255-value pool per feature,
like a model trained with --num-thresholds 255) over N docs x 200 features."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from quickrank_amd._capi import Context, NODE_DTYPE


def make_model(T, depth, F, rng, pool=255):
    thr_pool = np.sort(rng.random((F, pool), dtype=np.float32), axis=1)
    nn = (1 << (depth + 1)) - 1
    nodes = np.zeros((T, nn), NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    ni = (1 << depth) - 1
    for i in range(ni):
        f = rng.integers(0, F, T)
        nodes["feature"][:, i] = f
        nodes["threshold"][:, i] = thr_pool[f, rng.integers(0, pool, T)]
        nodes["left"][:, i] = 2 * i + 1
        nodes["right"][:, i] = 2 * i + 2
    nodes["value"][:, ni:] = rng.standard_normal((T, nn - ni))
    return nodes, np.full(T, 0.1)


def make_leafwise_model(T, leaves, F, rng, pool=255):
    """The shape a leaf-wise learner leaves (rt.cc:58-90 pops one leaf at a time): `leaves - 1`
    splits of a randomly chosen leaf, nodes numbered in creation order; features / thresholds as
    make_model.  Returns nodes, weights and the shape (max leaf depth per tree, mean leaf depth)."""
    thr_pool = np.sort(rng.random((F, pool), dtype=np.float32), axis=1)
    nn = 2 * leaves - 1
    nodes = np.zeros((T, nn), NODE_DTYPE)
    nodes["value"] = rng.standard_normal(nodes.shape)
    # (the draws of a whole model at once, the growth itself on plain lists: config 5's 10,000
    # trees take seconds, not a minute)
    feat = np.full((T, nn), -1, np.int32)
    left = np.full((T, nn), -1, np.int32)
    pick = rng.random((T, leaves - 1))
    fdraw = rng.integers(0, F, (T, leaves - 1)).astype(np.int32)
    tdraw = rng.integers(0, pool, (T, leaves - 1))
    thr = np.zeros((T, nn), np.float32)
    maxd, meand = [], []
    for t in range(T):
        open_, used, depth = [0], 1, [0] * nn
        ft, lt, tt, pk = feat[t], left[t], thr[t], pick[t]
        for s_ in range(leaves - 1):
            i = open_.pop(int(pk[s_] * len(open_)))
            f = fdraw[t, s_]
            ft[i] = f
            tt[i] = thr_pool[f, tdraw[t, s_]]
            lt[i] = used
            depth[used] = depth[used + 1] = depth[i] + 1
            open_ += [used, used + 1]
            used += 2
        d = [depth[i] for i in open_]
        maxd.append(max(d))
        meand.append(sum(d) / len(d))
    nodes["feature"], nodes["threshold"], nodes["left"] = feat, thr, left
    nodes["right"] = np.where(left >= 0, left + 1, -1)
    shape = {"max_depth_mean": round(float(np.mean(maxd)), 2), "max_depth_max": int(max(maxd)),
             "leaf_depth_mean": round(float(np.mean(meand)), 2)}
    return nodes, np.full(T, 0.1), shape


def numpy_score(nodes, w, x):
    """ensemble.cc:111-118 in numpy: sum += tree(x) * weight, in tree order, f64."""
    out = np.zeros(len(x))
    for t in range(len(nodes)):
        at = np.zeros(len(x), np.int64)
        while True:
            nd = nodes[t][at]
            idx = np.nonzero(nd["feature"] >= 0)[0]
            if not len(idx):
                break
            go = x[idx, nd["feature"][idx]] <= nd["threshold"][idx]
            at[idx] = np.where(go, nd["left"][idx], nd["right"][idx])
        out = out + nodes[t]["value"][at] * w[t]
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--docs", type=int, default=1000000)
    ap.add_argument("--features", type=int, default=200)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--check", type=int, default=2000)
    a = ap.parse_args()
    rng = np.random.default_rng(43)
    nodes, w = make_model(a.trees, a.depth, a.features, rng)
    x = rng.random((a.docs, a.features), dtype=np.float32)
    c = Context(0)
    c.upload_ensemble(nodes, w)
    s, ms = c.score(x)
    s, ms = c.score(x)
    visits = a.docs * a.trees * a.depth
    print(f"trees {a.trees} docs {a.docs} F {a.features}: kernel {ms:.2f} ms -> {a.docs / ms * 1e3:.3e} docs/s, "
          f"{visits / ms * 1e3:.3e} node visits/s")
    if a.check:
        want = numpy_score(nodes, w, x[:a.check])
        print("bit-exact vs a numpy walk on", a.check, "docs:", bool(np.array_equal(s[:a.check], want)))
