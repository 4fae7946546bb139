"""Document-sharded device protocol on ONE GPU: P contexts, each holding its own
queries (all features), the int64 sum all-reduces of quickrank_amd/dist.py
replaced by explicit sums over the contexts' exchange buffers.  Histograms are
exact integers, so the tree STRUCTURE must equal the single-context tree on the
whole dataset bit for bit; f64 node sums are added in rank order instead of
document order, so leaf values / scores agree to rounding (1e-12 relative, far
inside the 1e-5 bar)."""
import numpy as np
import pytest

from datagen import make_dataset
from parity_util import assert_same_tree_records

pytestmark = pytest.mark.gpu


def _split_queries(qoff, cuts):
    """cuts: query indices where a new rank starts -> list of (q0, q1)"""
    edges = [0] + list(cuts) + [len(qoff) - 1]
    return [(edges[i], edges[i + 1]) for i in range(len(edges) - 1)]


class _Emu:
    """all_reduce(sum) over the contexts' int64 exchange buffers, on one device"""

    def __init__(self, torch, ctxs):
        from quickrank_amd.dist import _DevArray
        self.torch, self.ctxs, self.DevArray = torch, ctxs, _DevArray
        self.dev = torch.device("cuda", 0)

    def view(self, c, key):
        b = c.doc_exchange_buffers()
        return self.torch.as_tensor(self.DevArray(b[key], b[key + "_n"] * 8, "<i8", 8),
                                    device=self.dev)

    def sync(self):
        for c in self.ctxs:
            c.synchronize()
        self.torch.cuda.synchronize()

    def allreduce(self, key):
        self.sync()
        views = [self.view(c, key) for c in self.ctxs]
        tot = views[0].clone()
        for v in views[1:]:
            tot += v
        for v in views:
            v.copy_(tot)
        self.sync()


    def allreduce_raw(self, bufs):
        """bufs: one (device pointer, int64 count) per context"""
        self.sync()
        views = [self.torch.as_tensor(self.DevArray(p, n * 8, "<i8", 8), device=self.dev) for p, n in bufs]
        tot = views[0].clone()
        for v in views[1:]:
            tot += v
        for v in views:
            v.copy_(tot)
        self.sync()


def _make_ctxs(qr, x, labels, qoff, parts, nthr):
    from quickrank_amd._capi import thresholds_from_stats
    world = len(parts)
    N, Q = len(labels), len(qoff) - 1
    ctxs, stats = [], []
    for r, (q0, q1) in enumerate(parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        c = qr.Context(0, rank=r, world=world, doc_shard=(N, Q))
        c.upload(x[d0:d1], labels[d0:d1], qoff[q0:q1 + 1] - qoff[q0])
        stats.append(c.bins_stats(nthr))
        ctxs.append(c)
    thr, ts = thresholds_from_stats(x.shape[1], nthr, np.stack([s[0] for s in stats]),
                                    np.stack([s[1] for s in stats]), np.stack([s[2] for s in stats]))
    for c in ctxs:
        c.build_bins_with(thr, ts)
    return ctxs, thr, ts


def _doc_fit(emu, ctxs, nleaves, minls, newton):
    for c in ctxs:
        c.tree_begin(nleaves, minls)
    emu.allreduce("hist")
    for _ in range(nleaves - 1):
        for c in ctxs:
            c.tree_decide()
            c.tree_apply()
        emu.allreduce("hist")
    for c in ctxs:
        c.tree_decide()
        c.tree_end_local(newton)
    emu.allreduce("leaf")
    return [c.tree_leaves_finish(nleaves, newton) for c in ctxs]


@pytest.mark.parametrize("world,cuts,nthr,F", [
    (2, [30], 255, 136),
    (3, [5, 41], 255, 70),       # a small first shard
    (4, [15, 30, 45], 16, 20),   # equal-width thresholds from the GLOBAL min/max
    (8, [7, 14, 22, 30, 38, 45, 52], 255, 136),
])
def test_doc_sharded_training_equals_single(world, cuts, nthr, F):
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=23, adversarial=True)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    thr1, ts1 = single.build_bins(nthr)
    single.reset_scores()
    parts = _split_queries(qoff, cuts)
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, nthr)
    assert np.array_equal(ts, ts1) and np.array_equal(thr, thr1)   # same thresholds, bit for bit
    off = 0
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.array_equal(c.read_bins(), single.read_bins()[d0:d1])
    emu = _Emu(torch, ctxs)
    for c in ctxs:
        c.reset_scores()
    for it in range(6):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(10, 2, True)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10)
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_fit(emu, ctxs, 10, 2, True)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
            assert np.allclose(g["deviance"], want["deviance"], rtol=1e-9, atol=1e-12), it
        for g in got[1:]:                      # every rank: the same bits
            for k in want.dtype.names:
                assert np.array_equal(g[k], got[0][k]), (it, k)
        # the training metric of the scores the lambdas ranked
        m1 = single.metric_last()
        for c in ctxs:
            assert c.metric_last() == pytest.approx(m1, rel=1e-12)
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


def _doc_obl_fit(emu, ctxs, depth, minls, newton):
    for c in ctxs:
        c.obl_begin(depth, minls)
    emu.allreduce("hist")
    for level in range(depth):
        for c in ctxs:
            c.obl_propose(level)
            c.obl_apply(level)
        if level + 1 < depth:
            emu.allreduce_raw([c.obl_level_exchange(level) for c in ctxs])
    for c in ctxs:
        c.tree_end_local(newton)
    emu.allreduce("leaf")
    return [c.tree_leaves_finish(1 << depth, newton) for c in ctxs]


@pytest.mark.parametrize("world,cuts,depth,minls,F,algo", [
    (2, [30], 4, 1, 136, "lambda"),
    (3, [5, 41], 6, 2, 70, "lambda"),      # a small first shard, nodes that run empty on a rank
    (4, [15, 30, 45], 3, 1, 20, "mart"),
    (8, [7, 14, 22, 30, 38, 45, 52], 5, 1, 136, "lambda"),
    (2, [30], 1, 1, 40, "lambda"),         # depth 1: no level exchange at all
])
def test_doc_sharded_oblivious_equals_single(world, cuts, depth, minls, F, algo):
    """ObliviousRT::fit (ot.cc:32-201) over document shards: one exchange per level; the
    tree's structure is the single-context tree bit for bit, leaf values to f64 rounding."""
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=29, adversarial=True)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.reset_scores()
    parts = _split_queries(qoff, cuts)
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, 255)
    emu = _Emu(torch, ctxs)
    for c in ctxs:
        c.reset_scores()
    newton = algo == "lambda"
    for it in range(4):
        if newton:
            single.compute_lambdas("NDCG", 10)
        else:
            single.compute_residuals()
        want = single.fit_oblivious(depth, minls, newton)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10) if newton else c.compute_residuals()
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_obl_fit(emu, ctxs, depth, minls, newton)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
        for g in got[1:]:
            for k in want.dtype.names:
                assert np.array_equal(g[k], got[0][k]), (it, k)
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


@pytest.mark.parametrize("batched", [False, True])
@pytest.mark.parametrize("world,cuts,algo,subsample,key_mask", [
    (2, [30], "lambda", 0.5, None),
    (3, [5, 41], "lambda", 0.3, None),       # a small first shard: few (or none) of the sample's documents
    (4, [15, 30, 45], "mart", 900.0, None),  # a number of documents instead of a fraction
    (3, [7, 33], "mart", 0.45, 0xE0000000),  # eight distinct keys: the k-th smallest is shared by hundreds of
    (4, [15, 30, 45], "lambda", 0.6, 0x3),   # documents of EVERY rank, taken in ascending global order
])
def test_doc_sharded_subsample_equals_single(world, cuts, algo, subsample, key_mask, batched):
    """--subsample over document shards: the key of a document is a function of its GLOBAL
    index, so every rank finds the sample a single GPU draws from the whole set and keeps
    its own part; trees (structure bit for bit) and the scores of ALL documents (mart.cc:345)
    equal the single-context run."""
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=40, seed=31)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    single.reset_scores()
    single.set_subsample(subsample, seed=11)
    parts = _split_queries(qoff, cuts)
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, 64)
    emu = _Emu(torch, ctxs)
    for c, (q0, q1) in zip(ctxs, parts):
        c.reset_scores()
        c.set_subsample(subsample, seed=11, first_doc=int(qoff[q0]))
    if key_mask is not None:
        for c in ctxs + [single]:
            c.debug_sample_key_mask(key_mask)
    newton = algo == "lambda"
    for it in range(4):
        single.compute_lambdas("NDCG", 10) if newton else single.compute_residuals()
        want = single.fit_tree(8, 2, newton)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10) if newton else c.compute_residuals()
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        # (batched: two splits per exchange, the sample's lists as the root -- _doc_fit_batched below)
        got = (_doc_fit_batched if batched else _doc_fit)(emu, ctxs, 8, 2, newton)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
        for g in got[1:]:
            for k in want.dtype.names:
                assert np.array_equal(g[k], got[0][k]), (it, k)
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


def test_doc_sharded_max_features_equals_single():
    """--max-features over document shards: the same per-node feature subsets on every rank."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=40, seed=35)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    single.reset_scores()
    single.set_max_features(0.3, seed=7)
    parts = _split_queries(qoff, [20, 45])
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, 64)
    emu = _Emu(torch, ctxs)
    for c in ctxs:
        c.reset_scores()
        c.set_max_features(0.3, seed=7)
    for it in range(3):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(8, 2, True)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10)
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_fit(emu, ctxs, 8, 2, True)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
    for c in ctxs + [single]:
        c.close()


@pytest.mark.parametrize("world,cuts", [(2, [30]), (3, [5, 41])])
def test_doc_sharded_oblivious_subsample_equals_single(world, cuts):
    """Oblivious trees on a sample over document shards (both round-2 additions at once)."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=40, seed=33)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    single.reset_scores()
    single.set_subsample(0.5, seed=13)
    parts = _split_queries(qoff, cuts)
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, 64)
    emu = _Emu(torch, ctxs)
    for c, (q0, q1) in zip(ctxs, parts):
        c.reset_scores()
        c.set_subsample(0.5, seed=13, first_doc=int(qoff[q0]))
    for it in range(3):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_oblivious(4, 2, True)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10)
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_obl_fit(emu, ctxs, 4, 2, True)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


def test_doc_sharded_mart_residuals():
    """MART (mean leaves, residual pseudo-responses) through the same protocol."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=40, docs_per_query=30, F=24, seed=4)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    single.reset_scores()
    parts = _split_queries(qoff, [13, 29])
    ctxs, _, _ = _make_ctxs(qr, x, labels, qoff, parts, 64)
    emu = _Emu(torch, ctxs)
    for c in ctxs:
        c.reset_scores()
    for it in range(4):
        single.compute_residuals()
        want = single.fit_tree(8, 1, False)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_residuals()
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_fit(emu, ctxs, 8, 1, False)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            for k in ("feature", "thr_id", "left", "right", "nsamples"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14)
    for c in ctxs:
        c.close()
    single.close()


def test_doc_sharded_trainer_over_rccl_world1():
    """The real DocShardedTrainer over torch.distributed/nccl (RCCL) with one rank:
    int64 all-reduces on zero-copy views of the exchange buffers, enqueued on the
    context's stream.  With one rank every sum has one term: bit-identical."""
    import os
    import torch
    import torch.distributed as dist
    import quickrank_amd as qr
    from quickrank_amd.dist import DocShardedTrainer, gather_thresholds
    x, labels, qoff = make_dataset(nq=50, docs_per_query=40, F=40, seed=5, adversarial=True)
    ref = qr.Context(0)
    ref.upload(x, labels, qoff)
    thr1, ts1 = ref.build_bins(64)
    ref.reset_scores()
    torch.cuda.set_device(0)
    port = 29600 + (os.getpid() + 7) % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = qr.Context(0, stream=torch.cuda.current_stream().cuda_stream,
                       doc_shard=(len(labels), len(qoff) - 1))
        c.upload(x, labels, qoff)
        thr, ts = gather_thresholds(c, 64)
        assert np.array_equal(thr, thr1) and np.array_equal(ts, ts1)
        c.build_bins_with(thr, ts)
        c.reset_scores()
        tr = DocShardedTrainer(c)
        for it in range(3):
            ref.compute_lambdas("NDCG", 10)
            want = ref.fit_tree(8, 1, True)
            ref.update_scores(0.1)
            tr.compute_lambdas("NDCG", 10)
            got = tr.fit_tree(8, 1, True)
            c.update_scores(0.1)
            assert_same_tree_records(got, want, node_sums_exact=False, where=it)
            assert c.metric_last() == ref.metric_last()
        for it in range(2):                       # oblivious trees through the same trainer
            ref.compute_lambdas("NDCG", 10)
            want = ref.fit_oblivious(4, 1, True)
            ref.update_scores(0.1)
            tr.compute_lambdas("NDCG", 10)
            got = tr.fit_oblivious(4, 1, True)
            c.update_scores(0.1)
            for k in want.dtype.names:
                assert np.array_equal(got[k], want[k]), (it, k)
        assert np.array_equal(c.get_scores(), ref.get_scores())
        c.close()
    finally:
        dist.destroy_process_group()
    ref.close()


def _doc_fit_batched(emu, ctxs, nleaves, minls, newton, stats=None):
    """the document-sharded protocol with up to two splits per exchange (include/qr_hip.h,
    qr_tree_batch_*), the all-reduces replaced by explicit sums; what
    DocShardedTrainer._fit_tree_batched drives"""
    steps = [c.tree_batch_begin(nleaves, minls) for c in ctxs]
    assert len(set(steps)) == 1            # every rank guesses the same number of steps
    emu.allreduce("hist")
    for c in ctxs:
        c.tree_batch_root()
    exchanges = 1

    def run(k):
        for s in range(k):
            for c in ctxs:
                c.tree_batch_apply()
            emu.allreduce_raw([c.tree_batch_exchange() for c in ctxs])
            for c in ctxs:
                c.tree_batch_decide(s == k - 1)

    run(steps[0])
    exchanges += steps[0]
    done, piece, misses = steps[0], 1, 0
    while True:
        res = [c.tree_batch_settle() for c in ctxs]
        assert len(set(res)) == 1          # ... and settles alike
        if not res[0][0]:
            break
        misses += 1
        k = max(1, min(piece, nleaves - 1 - done))
        run(k)
        exchanges += k
        done += k
        piece *= 2
    for c in ctxs:
        c.tree_end_local(newton)
    emu.allreduce("leaf")
    if stats is not None:
        stats.append((exchanges, res[0][1], misses))
    return [c.tree_leaves_finish(nleaves, newton) for c in ctxs]


@pytest.mark.parametrize("world,cuts,nthr,F,nleaves,minls,force", [
    (2, [30], 255, 136, 10, 2, None),
    (3, [5, 41], 255, 70, 10, 1, None),        # a small first shard: nodes that run empty on a rank
    (4, [15, 30, 45], 16, 20, 16, 2, None),
    (8, [7, 14, 22, 30, 38, 45, 52], 255, 136, 10, 2, None),
    (2, [30], 255, 136, 10, 2, "1"),           # one step enqueued whatever the tree: every tree is carried on
    (3, [11, 37], 255, 70, 10, 2, "unfused"),  # the control step in a launch of its own (what runs above 4M documents per rank)
    (2, [30], 255, 40, 12, 1, "unfused1"),     # ... with one step enqueued whatever the tree
    (3, [20, 41], 64, 40, 31, 0, None),        # larger trees, empty leaves allowed
    (2, [30], 255, 40, 2, 1, None),            # a stump
    (2, [30], 64, 40, 64, 1, None),            # the larger LDS copy of the control step
    (3, [20, 41], 64, 40, 100, 1, None),       # node records beyond the LDS copies: the device-resident control step
])
def test_doc_sharded_batched_training_equals_single(world, cuts, nthr, F, nleaves, minls, force, monkeypatch):
    """Two splits per exchange on document shards: the trees are the single-context trees
    (structure bit for bit, f64 sums to rounding: they are added in rank order), every rank holds
    the same bits, and a tree costs 1 + steps histogram exchanges instead of nleaves."""
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=37, adversarial=True)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(nthr)
    single.reset_scores()
    if force in ("unfused", "unfused1"):
        monkeypatch.setenv("QR_FUSE_MAX_DOCS", "0")
        force = "1" if force == "unfused1" else None
    if force is not None:
        monkeypatch.setenv("QR_STEPS_HINT", force)
    parts = _split_queries(qoff, cuts)
    ctxs, thr, ts = _make_ctxs(qr, x, labels, qoff, parts, nthr)
    monkeypatch.delenv("QR_STEPS_HINT", raising=False)
    monkeypatch.delenv("QR_FUSE_MAX_DOCS", raising=False)
    emu = _Emu(torch, ctxs)
    for c in ctxs:
        c.reset_scores()
        assert c.tree_batch_supported(nleaves)
    stats = []
    for it in range(6):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(nleaves, minls, True)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10)
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_fit_batched(emu, ctxs, nleaves, minls, True, stats)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
            assert np.allclose(g["deviance"], want["deviance"], rtol=1e-9, atol=1e-12), it
        for g in got[1:]:                      # every rank: the same bits
            for k in want.dtype.names:
                assert np.array_equal(g[k], got[0][k]), (it, k)
        m1 = single.metric_last()
        for c in ctxs:
            assert c.metric_last() == pytest.approx(m1, rel=1e-12)
    # exchanges per tree: root + steps (+ what a wrong guess adds), never more than nleaves
    for ex, used, misses in stats:
        assert ex <= nleaves, stats
    if force is not None and nleaves > 3:
        assert any(m > 0 for _, _, m in stats), stats      # the carried-on path has run
    if force is None and nleaves >= 10:
        assert min(ex for ex, _, _ in stats[1:]) < nleaves - 1, stats   # fewer exchanges than one per split
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


def test_doc_sharded_batched_equals_one_split_protocol():
    """the two document-sharded protocols grow the same trees (structure bit for bit; the f64 sums
    of a directly built child come from the partition's slices in one and from the histogram
    workgroups in the other: rounding)"""
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=70, seed=41, adversarial=True)
    parts = _split_queries(qoff, [11, 37])
    a, _, _ = _make_ctxs(qr, x, labels, qoff, parts, 255)
    b, _, _ = _make_ctxs(qr, x, labels, qoff, parts, 255)
    ea, eb = _Emu(torch, a), _Emu(torch, b)
    for c in a + b:
        c.reset_scores()
    for it in range(5):
        for emu, ctxs in ((ea, a), (eb, b)):
            for c in ctxs:
                c.compute_lambdas("NDCG", 10)
            emu.allreduce("scal")
            for c in ctxs:
                c.lambda_finish()
        ta = _doc_fit(ea, a, 12, 2, True)
        tb = _doc_fit_batched(eb, b, 12, 2, True)
        for c in a + b:
            c.update_scores(0.1)
        for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
            assert np.array_equal(ta[0][k], tb[0][k]), (it, k)
        assert np.allclose(ta[0]["value"], tb[0]["value"], rtol=1e-11, atol=1e-14), it
    for ca, cb in zip(a, b):
        assert np.allclose(ca.get_scores(), cb.get_scores(), rtol=1e-10, atol=1e-13)
        ca.close()
        cb.close()


@pytest.mark.parametrize("world,cuts,nthr,F,minls,subsample", [
    (2, [30], 1024, 40, 2, None),        # rows of up to 1025 slots: the LDS-tiled histogram kernel
    (3, [5, 41], 0, 24, 1, None),        # every distinct value a threshold (rows of ~3000 slots), a small first shard
    (4, [15, 30, 45], 4096, 20, 2, None),  # the general histogram kernel, the chunked scan
    (3, [11, 37], 1024, 40, 2, 0.5),     # --subsample: the sample's lists as the root
])
def test_doc_sharded_wide_bins_equal_single(world, cuts, nthr, F, minls, subsample):
    """More than 255 thresholds per feature on document shards (round 4): every rank bins its own
    documents against the thresholds of the WHOLE set (qr_bins_build_wide_with) and the node
    histograms -- ragged rows -- go through the same ONE int64 all-reduce per split.  Trees: the
    single wide context's structure bit for bit, values to f64 rounding; every rank the same bits."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=47, adversarial=(nthr != 0))
    N, Q = len(labels), len(qoff) - 1
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    thr, ts = single.build_bins(nthr, wide=True)
    single.reset_scores()
    parts = _split_queries(qoff, cuts)
    ctxs = []
    for r, (q0, q1) in enumerate(parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        c = qr.Context(0, rank=r, world=world, doc_shard=(N, Q))
        c.upload(x[d0:d1], labels[d0:d1], qoff[q0:q1 + 1] - qoff[q0])
        c.build_bins_wide_with(thr, ts)
        assert np.array_equal(c.read_bins_u32(), single.read_bins_u32()[d0:d1])
        c.reset_scores()
        if subsample is not None:
            c.set_subsample(subsample, seed=13, first_doc=d0)
        ctxs.append(c)
    if subsample is not None:
        single.set_subsample(subsample, seed=13)
    emu = _Emu(torch, ctxs)
    for it in range(4):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(8, minls, True)
        single.update_scores(0.1)
        for c in ctxs:
            c.compute_lambdas("NDCG", 10)
        emu.allreduce("scal")
        for c in ctxs:
            c.lambda_finish()
        got = _doc_fit(emu, ctxs, 8, minls, True)
        for c in ctxs:
            c.update_scores(0.1)
        for g in got:
            assert len(g) == len(want), it
            for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
                assert np.array_equal(g[k], want[k]), (it, k)
            assert np.allclose(g["value"], want["value"], rtol=1e-11, atol=1e-14), it
        for g in got[1:]:
            for k in want.dtype.names:
                assert np.array_equal(g[k], got[0][k]), (it, k)
    s1 = single.get_scores()
    for c, (q0, q1) in zip(ctxs, parts):
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(c.get_scores(), s1[d0:d1], rtol=1e-10, atol=1e-13)
        c.close()
    single.close()


@pytest.mark.parametrize("nthr", [1024, 0])
def test_doc_sharded_wide_thresholds_over_rccl_world1(nthr):
    """quickrank_amd.dist.build_doc_bins + DocShardedTrainer with more than 255 thresholds per
    feature over real RCCL (one rank): the device's column statistics (qr_bins_stats_wide) merged by
    qr_thresholds_from_stats_wide are the single context's thresholds bit for bit, and so are the
    trees (one term per sum)."""
    import os
    import torch
    import torch.distributed as dist
    import quickrank_amd as qr
    from quickrank_amd.dist import DocShardedTrainer, build_doc_bins
    x, labels, qoff = make_dataset(nq=50, docs_per_query=40, F=24, seed=53)
    ref = qr.Context(0)
    ref.upload(x, labels, qoff)
    thr1, ts1 = ref.build_bins(nthr, wide=True)
    ref.reset_scores()
    torch.cuda.set_device(0)
    port = 29600 + (os.getpid() + 11 + nthr) % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = qr.Context(0, stream=torch.cuda.current_stream().cuda_stream, doc_shard=(len(labels), len(qoff) - 1))
        c.upload(x, labels, qoff)
        thr, ts = build_doc_bins(c, nthr)
        assert c.wide and np.array_equal(ts, ts1) and np.array_equal(thr.view(np.uint32), thr1.view(np.uint32))
        c.reset_scores()
        tr = DocShardedTrainer(c)
        for it in range(3):
            ref.compute_lambdas("NDCG", 10)
            want = ref.fit_tree(8, 1, True)
            ref.update_scores(0.1)
            tr.compute_lambdas("NDCG", 10)
            got = tr.fit_tree(8, 1, True)
            c.update_scores(0.1)
            assert_same_tree_records(got, want, node_sums_exact=False, where=it)
        assert np.allclose(c.get_scores(), ref.get_scores(), rtol=1e-12, atol=1e-14)
        c.close()
    finally:
        dist.destroy_process_group()
    ref.close()
