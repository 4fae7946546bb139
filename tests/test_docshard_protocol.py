"""Document-sharded N > 1 path on CPU: world_size 2 and 3 over gloo.  Each process
runs the real quickrank_amd.dist.DocShardedTrainer (int64 sum all-reduces of the
scalar / histogram / leaf buffers) over a host stand-in context that holds only
its own queries; the trees must reproduce the unsharded oracle training."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q, cuts, ntrees, mode="batched"):
    """mode: "one_split" (one histogram exchange per split), "batched" (up to two splits per
    exchange, trees read at once) or "lazy" (batched, read=False, one step enqueued whatever the
    tree: every tree is carried on when the trainer settles it)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    os.environ["QR_DOC_BATCH"] = "0" if mode == "one_split" else "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from datagen import make_dataset
    from parity_util import assert_tree_parity
    from quickrank_amd.dist import DocShardedTrainer
    from docshard_standin import DocStandinContext
    x, labels, qoff = make_dataset(nq=24, docs_per_query=40, F=20, seed=31, adversarial=True)
    # ("wide": more than 255 thresholds per feature -- ragged rows through the same one-split protocol)
    whole = oracle.Trainer(x, 300 if mode == "wide" else 64)   # thresholds of the WHOLE set
    edges = [0] + list(cuts) + [len(qoff) - 1]
    q0, q1 = edges[rank], edges[rank + 1]
    d0, d1 = int(qoff[q0]), int(qoff[q1])
    ctx = DocStandinContext(x[d0:d1], labels[d0:d1], qoff[q0:q1 + 1] - qoff[q0], whole.thr,
                            whole.thr_size, rank, world, len(labels), len(qoff) - 1)
    tr = DocShardedTrainer(ctx)
    scores = np.zeros(len(labels))
    ok = True
    exchanges = []
    if mode == "lazy":
        ctx.steps_force = 1
    for it in range(ntrees):
        tr.compute_lambdas("NDCG", 10)
        if mode == "lazy":
            assert tr.fit_tree(8, 2, True, read=False) is None
            ctx.update_scores(0.1)            # (enqueued behind an incomplete tree: repeated when it is settled)
            tr.settle()
            nodes = ctx.last_tree
        else:
            nodes = tr.fit_tree(8, 2, True)
            ctx.update_scores(0.1)
        if mode not in ("one_split", "wide"):
            exchanges.append(tr.collectives)
        # unsharded oracle iteration on the same scores
        lam, w = oracle.lambdas(labels, scores, qoff)
        t = whole.fit_tree(lam, nleaves=8, minls=2)
        whole.update_output(t, lam, w)
        o = t["nodes"]
        assert_tree_parity(whole.stmap, o, nodes, value_rtol=1e-9)
        # the metric of the scores the lambdas ranked (summed over the ranks' queries)
        assert ctx.metric == pytest.approx(oracle.eval_dataset(labels, scores, qoff), rel=1e-12)
        scores = scores + 0.1 * o["value"][_leaf_of(whole, o)]
        assert np.allclose(ctx.scores, scores[d0:d1], rtol=1e-9, atol=1e-12)
        # every rank must hold the same tree, bit for bit
        mine = torch.from_numpy(np.frombuffer(nodes.tobytes(), np.uint8).copy())
        ref = mine.clone()
        dist.broadcast(ref, 0)
        ok = ok and torch.equal(mine, ref)
    if mode == "batched":
        # root + steps: the first tree enqueues the worst case, the next ones what the last needed
        ok = ok and exchanges[0] == 8 and all(4 <= e <= 8 for e in exchanges[1:]) and min(exchanges) < 8
        # the trainer's own account (bench.py's collectives_per_tree / collective_bytes_per_tree): scalars +
        # histograms + leaf sums per tree, int64 words of the buffers it handed over
        b = ctx.doc_exchange_buffers()
        ok = ok and tr.traffic[0] == sum(exchanges) + 2 * ntrees
        ok = ok and tr.traffic[1] >= 8 * (ntrees * (b["scal_n"] + b["hist_n"]) + (sum(exchanges) - ntrees) * b["hist_n"])
    if mode == "lazy":
        ok = ok and all(e > 2 for e in exchanges)      # one step was never enough for 8 leaves
    if rank == 0:
        q.put(bool(ok))
    dist.destroy_process_group()


def _worker_obl(rank, world, port, q, cuts, ntrees, depth):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from datagen import make_dataset
    from parity_util import assert_tree_parity
    from quickrank_amd.dist import DocShardedTrainer
    from docshard_standin import DocStandinContext
    x, labels, qoff = make_dataset(nq=24, docs_per_query=40, F=20, seed=37)
    whole = oracle.Trainer(x, 64)
    edges = [0] + list(cuts) + [len(qoff) - 1]
    q0, q1 = edges[rank], edges[rank + 1]
    d0, d1 = int(qoff[q0]), int(qoff[q1])
    ctx = DocStandinContext(x[d0:d1], labels[d0:d1], qoff[q0:q1 + 1] - qoff[q0], whole.thr,
                            whole.thr_size, rank, world, len(labels), len(qoff) - 1)
    tr = DocShardedTrainer(ctx)
    scores = np.zeros(len(labels))
    ok = True
    for it in range(ntrees):
        tr.compute_lambdas("NDCG", 10)
        nodes = tr.fit_oblivious(depth, 2, True)
        ctx.update_scores(0.1)
        lam, w = oracle.lambdas(labels, scores, qoff)
        t = whole.fit_tree(lam, minls=2, oblivious_depth=depth)
        whole.update_output(t, lam, w)
        o = t["nodes"]
        assert len(nodes) == len(o), (len(nodes), len(o))
        assert_tree_parity(whole.stmap, o, nodes, value_rtol=1e-9)
        scores = scores + 0.1 * o["value"][_leaf_of(whole, o)]
        assert np.allclose(ctx.scores, scores[d0:d1], rtol=1e-9, atol=1e-12)
        mine = torch.from_numpy(np.frombuffer(nodes.tobytes(), np.uint8).copy())
        ref = mine.clone()
        dist.broadcast(ref, 0)
        ok = ok and torch.equal(mine, ref)
    if rank == 0:
        q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts", [(2, [9]), (3, [3, 15])])
def test_doc_sharded_oblivious_equals_unsharded(world, cuts):
    """DocShardedTrainer.fit_oblivious over gloo: root exchange, one exchange of the level's
    child cells per level, leaf exchange -- the unsharded oracle's oblivious tree."""
    import oracle
    oracle.build(ref=False)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 33500 + (os.getpid() + world * 13) % 2000
    procs = [ctxm.Process(target=_worker_obl, args=(r, world, port, q, cuts, 3, 3)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _leaf_of(tr, nodes):
    """leaf node index of every document: walk on the bin ids"""
    at = np.zeros(tr.N, np.int64)
    while True:
        nd = nodes[at]
        internal = nd["feature"] >= 0
        if not internal.any():
            return at
        idx = np.nonzero(internal)[0]
        go = tr.stmap[nd["feature"][idx], idx] <= nd["thr_id"][idx]
        at[idx] = np.where(go, nd["left"][idx], nd["right"][idx])


@pytest.mark.parametrize("world,cuts,mode", [(2, [9], "batched"), (3, [3, 15], "batched"), (2, [9], "one_split"),
                                             (3, [3, 15], "one_split"), (2, [9], "lazy"), (3, [3, 15], "lazy"),
                                             (2, [9], "wide")])
def test_doc_sharded_training_equals_unsharded(world, cuts, mode):
    import oracle
    oracle.build(ref=False)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 31500 + (os.getpid() + world * 11 + len(mode) * 101) % 2000
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q, cuts, 4, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_thresholds_from_stats_symbol_is_pure_host():
    """qr_thresholds_from_stats needs no GPU: merging the column statistics of two
    document shards gives the thresholds of their union (mart.cc:147-169)."""
    import oracle
    from quickrank_amd import build
    from quickrank_amd._capi import thresholds_from_stats
    if not os.path.exists(build.LIB):
        pytest.skip("HIP library not built")
    rng = np.random.default_rng(0)
    F, nthr = 6, 16
    x = rng.random((400, F), dtype=np.float32)
    x[:, 1] = np.floor(x[:, 1] * 5)            # few uniques -> the uniques branch
    x[:, 2] = 0.25                             # constant
    x[:200, 3] = np.floor(x[:200, 3] * 10)     # few uniques on one shard only
    limit = nthr + 1

    def flip(u):                               # radix key (radix.cc:28-30)
        return np.where(u >> 31, ~u, u | np.uint32(0x80000000)).astype(np.uint32)

    vals = np.zeros((2, F, limit + 1), np.uint32)
    cnt = np.zeros((2, F), np.uint32)
    mm = np.zeros((2, F, 2), np.uint32)
    for r, part in enumerate((x[:200], x[200:])):
        for f in range(F):
            u = np.unique(part[:, f])
            k = min(len(u), limit + 1)
            vals[r, f, :k] = u[:k].view(np.uint32)
            cnt[r, f] = k
            mm[r, f] = flip(np.array([part[:, f].min(), part[:, f].max()], np.float32).view(np.uint32))
    thr, ts = thresholds_from_stats(F, nthr, vals, cnt, mm)
    othr, ots = oracle.thresholds(np.ascontiguousarray(x.T), nthr)
    assert np.array_equal(ts, ots.astype(np.uint32))
    for f in range(F):
        assert np.array_equal(thr[f, :ts[f]], othr[f, :ts[f]]), f


@pytest.mark.parametrize("nthr", [300, 1024, 0])
def test_wide_thresholds_from_stats_are_the_whole_sets(nthr):
    """qr_thresholds_from_stats_wide (pure host code): merging the ranks' sorted distinct values /
    min / max gives the thresholds mart.cc:140-169 computes on the whole set -- the distinct values
    where there are at most `nthr` of them (or nthr == 0), `nthr` equal f32 steps otherwise."""
    import oracle
    from quickrank_amd import build
    from quickrank_amd._capi import thresholds_from_stats_wide
    if not os.path.exists(build.LIB):
        pytest.skip("HIP library not built")
    oracle.build(ref=False)
    rng = np.random.default_rng(3)
    F, N = 7, 3000
    x = rng.standard_normal((N, F)).astype(np.float32)
    x[:, 1] = np.floor(x[:, 1] * 40)                 # ~250 distinct values: below 300, above 255
    x[:, 2] = 0.25                                   # constant
    x[:1000, 3] = np.floor(x[:1000, 3] * 3)          # few distinct values on one shard only
    x[:, 4] = np.where(rng.random(N) < 0.5, -0.0, 0.0)   # one value by `<` (mart.cc:149-151)
    x[:, 5] = np.floor(rng.random(N) * 900)          # more than 255, fewer than 1024
    limit = nthr + 1 if nthr else 4096
    cuts = [0, 1000, 1700, N]

    def flip(u):                                     # radix key (radix.cc:28-30)
        return np.where(u >> 31, ~u, u | np.uint32(0x80000000)).astype(np.uint32)

    vals = np.zeros((3, F, limit), np.uint32)
    cnt = np.zeros((3, F), np.uint32)
    mm = np.zeros((3, F, 2), np.uint32)
    for r in range(3):
        part = x[cuts[r]:cuts[r + 1]]
        for f in range(F):
            keys = np.sort(flip(part[:, f].view(np.uint32)))
            col = np.where(keys >> 31, keys & np.uint32(0x7FFFFFFF), ~keys).astype(np.uint32).view(np.float32)
            u = [col[0]]
            for v in col[1:]:                        # what qr_bins_stats_wide keeps: strict `<` on the sorted column
                if u[-1] < v:
                    u.append(v)
                if len(u) > limit:
                    break
            k = min(len(u), limit)
            vals[r, f, :k] = np.array(u[:k], np.float32).view(np.uint32)
            cnt[r, f] = min(len(u), limit + 1)
            mm[r, f] = [keys[0], keys[-1]]
    flat, ts = thresholds_from_stats_wide(F, nthr, limit, vals, cnt, mm)
    othr, ots = oracle.thresholds(np.ascontiguousarray(x.T), nthr)
    assert np.array_equal(ts, ots.astype(np.uint32))
    o = 0
    for f in range(F):
        assert np.array_equal(flat[o:o + ts[f]].view(np.uint32), othr[f, :ts[f]].view(np.uint32)), f
        o += int(ts[f])
