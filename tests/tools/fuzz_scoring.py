"""Randomised parity sweep of the scoring kernels (k_doc_bins + k_score_p4 / k_score_bin /
k_ensemble_score, k_obl_score_s / k_obl_score_bin) against the reference's walks restated in
numpy: Ensemble::score_instance (ensemble.cc:111-118, `x <= threshold` goes left, sum of
tree(x) * weight in tree order) and the bit-interleaved oblivious scorer
(generate_oblivious.cc:305-324, `x > threshold` sets the level's bit, f32 weights).  Every
configuration must be bit-exact.

    python tests/tools/fuzz_scoring.py [count] [seed]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))


def random_tree(rng, nleaves, F, pool, chain=False):
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


def special_values(rng, x, pool):
    n, F = x.shape
    for v in (np.nan, np.inf, -np.inf, -0.0, np.float32(pool[0]), np.float32(pool[-1])):
        k = max(1, n // 50)
        x[rng.integers(0, n, k), rng.integers(0, F, k)] = v


def one_ensemble(ctx, rng):
    from quickrank_amd._capi import NODE_DTYPE
    F = int(rng.choice([1, 2, 3, 4, 5, 7, 16, 37, 136, 200, 301]))
    T = int(rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 64, 100]))
    maxleaves = int(rng.choice([1, 2, 3, 10, 31, 64, 128, 200]))
    npool = int(rng.choice([2, 5, 40, 250, 300, 2000]))
    N = int(rng.choice([1, 5, 63, 64, 65, 200, 1000, 3000]))
    pool = np.unique(rng.standard_normal(npool).astype(np.float32))
    trees = []
    for k in range(T):
        m = int(rng.integers(1, maxleaves + 1))
        trees.append(random_tree(rng, m, F, pool, chain=bool(rng.integers(4) == 0 and m <= 60)))
    maxn = max(len(t) for t in trees)
    nodes = np.zeros((T, maxn), NODE_DTYPE)
    nodes["feature"] = -1
    nodes["left"] = nodes["right"] = -1
    for k, t in enumerate(trees):
        nodes[k, :len(t)] = t
    w = rng.random(T) + 0.25
    x = rng.choice(pool, size=(N, F)).astype(np.float32)
    special_values(rng, x, pool)
    ctx.upload_ensemble(nodes, w)
    got, _ = ctx.score(x)
    want = np.zeros(N)
    with np.errstate(invalid="ignore"):
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
    desc = f"ensemble F={F} T={T} leaves<={maxleaves} pool={len(pool)} N={N}"
    return desc, bool(np.array_equal(got.view(np.uint64), want.view(np.uint64)))


def one_oblivious(ctx, rng):
    F = int(rng.choice([1, 2, 5, 40, 136, 300]))
    T = int(rng.choice([1, 3, 4, 5, 31, 32, 33, 100, 300]))
    D = int(rng.integers(1, 10))
    npool = int(rng.choice([2, 3, 50, 400]))
    N = int(rng.choice([1, 63, 64, 65, 500, 2500]))
    pool = np.unique(rng.random(npool).astype(np.float32))
    feat = rng.integers(0, F, (T, D)).astype(np.uint32)
    thr = rng.choice(pool, (T, D)).astype(np.float32)
    leaves = rng.standard_normal((T, 1 << D))
    w = (rng.random(T) * 0.3 + 0.01).astype(np.float32)
    depths = np.sort(rng.integers(1, D + 1, T)).astype(np.uint32) if rng.integers(2) else None
    x = rng.choice(np.concatenate([pool, rng.random(8).astype(np.float32)]), size=(N, F)).astype(np.float32)
    special_values(rng, x, pool)
    ctx.upload_oblivious(feat, thr, leaves, w, depths)
    got, _ = ctx.score_oblivious(x)
    want = np.zeros(N)
    with np.errstate(invalid="ignore"):
        for t in range(T):
            m = D if depths is None else int(depths[t])
            idx = np.zeros(N, np.int64)
            for l in range(m):
                idx |= (x[:, feat[t, l]] > thr[t, l]).astype(np.int64) << (m - 1 - l)
            want = want + np.float64(w[t]) * leaves[t, idx]
    desc = f"oblivious F={F} T={T} D={D} pool={len(pool)} N={N} depths={'mixed' if depths is not None else 'full'}"
    return desc, bool(np.array_equal(got.view(np.uint64), want.view(np.uint64)))


def sweep(count, seed, verbose=True):
    from quickrank_amd._capi import Context
    rng = np.random.default_rng(seed)
    ctx = Context(0)
    out = []
    for i in range(count):
        desc, ok = (one_oblivious if i % 3 == 2 else one_ensemble)(ctx, rng)
        out.append({"i": i, "desc": desc, "ok": ok})
        if verbose and not ok:
            print("MISMATCH", i, desc)
    ctx.close()
    return out


if __name__ == "__main__":
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    res = sweep(count, seed)
    bad = [r for r in res if not r["ok"]]
    print(f"scoring fuzz seed {seed}: {len(res)} configurations, {len(bad)} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
