"""Feature-sharded device protocol on ONE GPU: P contexts (rank r of P) on the
same device, the two collectives of quickrank_amd/dist.py replaced by explicit
copies between the contexts' exchange buffers.  The sharded trees must be
bit-identical to the single-context tree (every feature's histogram lives on
exactly one rank, integer accumulation)."""
import numpy as np
import pytest

from datagen import make_dataset
from parity_util import assert_same_tree_records

pytestmark = pytest.mark.gpu


def _views(torch, ctx, world):
    from quickrank_amd.dist import _DevArray
    b = ctx.exchange_buffers()
    dev = torch.device("cuda", 0)
    return dict(
        loc=torch.as_tensor(_DevArray(b["recs_local"], b["rec_bytes"]), device=dev),
        all=torch.as_tensor(_DevArray(b["recs_all"], b["rec_bytes"] * world), device=dev),
        mask=torch.as_tensor(_DevArray(b["mask"], b["mask_bytes"], "<i4", 4), device=dev))


def _sharded_fit(torch, ctxs, nleaves, minls):
    world = len(ctxs)
    v = [_views(torch, c, world) for c in ctxs]

    def sync():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    def gather():
        sync()
        cat = torch.cat([x["loc"] for x in v])
        for x in v:
            x["all"].copy_(cat)
        sync()

    def reduce_mask():
        sync()
        tot = v[0]["mask"].clone()
        for x in v[1:]:
            tot += x["mask"]
        for x in v:
            x["mask"].copy_(tot)
        sync()

    for c in ctxs:
        c.tree_begin(nleaves, minls)
    gather()
    for _ in range(nleaves - 1):
        for c in ctxs:
            c.tree_decide()
        reduce_mask()
        for c in ctxs:
            c.tree_apply()
        gather()
    for c in ctxs:
        c.tree_decide()
    return [c.tree_end(nleaves, True) for c in ctxs]


@pytest.mark.parametrize("world,F", [(2, 136), (3, 70), (8, 136), (2, 9)])
def test_sharded_contexts_equal_single(world, F, oracle_lib):
    import torch
    import quickrank_amd as qr
    from quickrank_amd import build
    build.build()
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=17, adversarial=True)
    rng = np.random.default_rng(2)
    lam, w = oracle_lib.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.set_pseudo(lam, w)
    want = single.fit_tree(12, 3, True)
    single.set_scores(np.zeros(len(labels)))
    single.update_scores(0.1)
    want_scores = single.get_scores()
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(255)
        c.set_pseudo(lam, w)
        ctxs.append(c)
    got = _sharded_fit(torch, ctxs, 12, 3)
    for g in got:   # (internal nodes' f64 sums: two fixed summation orders, see parity_util)
        assert_same_tree_records(g, want, node_sums_exact=False)
    for c in ctxs:
        c.set_scores(np.zeros(len(labels)))
        c.update_scores(0.1)
        assert np.array_equal(c.get_scores(), want_scores)
        c.close()
    single.close()


def test_torch_distributed_fitter_world1(oracle_lib):
    """The real ShardedTreeFitter over torch.distributed/nccl (RCCL) with one
    rank: zero-copy tensor views of the exchange buffers, collectives enqueued on
    the context's stream (= torch's current stream), no host sync inside a tree."""
    import os
    import torch
    import torch.distributed as dist
    import quickrank_amd as qr
    from quickrank_amd.dist import ShardedTreeFitter
    x, labels, qoff = make_dataset(nq=50, docs_per_query=40, F=40, seed=5)
    lam, w = oracle_lib.lambdas(labels, np.zeros(len(labels)), qoff)
    ref = qr.Context(0)
    ref.upload(x, labels, qoff)
    ref.build_bins(64)
    ref.set_pseudo(lam, w)
    want = ref.fit_tree(8, 1, True)
    ref.close()
    torch.cuda.set_device(0)
    port = 29600 + os.getpid() % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = qr.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        c.set_pseudo(lam, w)
        got = ShardedTreeFitter(c).fit_tree(c, 8, 1, True)
        assert_same_tree_records(got, want, node_sums_exact=False)
        c.close()
    finally:
        dist.destroy_process_group()


def _sharded_obl_fit(torch, ctxs, depth, minls):
    """The oblivious protocol of ShardedTreeFitter.fit_oblivious with the two collectives of
    a level replaced by explicit copies / sums between the contexts' buffers."""
    from quickrank_amd.dist import _DevArray
    world = len(ctxs)
    dev = torch.device("cuda", 0)
    v = []
    for c in ctxs:
        b = c.obl_exchange_buffers()
        v.append(dict(loc=torch.as_tensor(_DevArray(b["recs_local"], b["rec_bytes"]), device=dev),
                      all=torch.as_tensor(_DevArray(b["recs_all"], b["rec_bytes"] * world), device=dev),
                      mask=torch.as_tensor(_DevArray(b["mask"], b["mask_bytes"], "<i4", 4), device=dev)))

    def sync():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    for c in ctxs:
        c.obl_begin(depth, minls)
    for level in range(depth):
        for c in ctxs:
            c.obl_propose(level)
        sync()
        cat = torch.cat([x["loc"] for x in v])
        for x in v:
            x["all"].copy_(cat)
        sync()
        for c in ctxs:
            c.obl_mark(level)
        sync()
        tot = v[0]["mask"].clone()
        for x in v[1:]:
            tot += x["mask"]
        for x in v:
            x["mask"].copy_(tot)
        sync()
        for c in ctxs:
            c.obl_apply(level)
    return [c.obl_end(depth, True) for c in ctxs]


@pytest.mark.parametrize("world,F,depth,minls", [(2, 136, 4, 1), (3, 70, 6, 3), (8, 136, 5, 1), (2, 9, 3, 20)])
def test_sharded_oblivious_equal_single(world, F, depth, minls, oracle_lib):
    """Feature-sharded oblivious trees (qr_obl_begin / propose / mark / apply): every rank's
    tree is bit-identical to the single-context one -- level splits, node counts, leaf
    values -- and so are the updated scores."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=F, seed=19, adversarial=True)
    rng = np.random.default_rng(3)
    lam, w = oracle_lib.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.set_pseudo(lam, w)
    want = single.fit_oblivious(depth, minls, True)
    single.set_scores(np.zeros(len(labels)))
    single.update_scores(0.1)
    want_scores = single.get_scores()
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(255)
        c.set_pseudo(lam, w)
        ctxs.append(c)
    got = _sharded_obl_fit(torch, ctxs, depth, minls)
    for g in got:
        assert len(g) == len(want)
        for k in want.dtype.names:
            assert np.array_equal(g[k], want[k]), k
    for c in ctxs:
        c.set_scores(np.zeros(len(labels)))
        c.update_scores(0.1)
        assert np.array_equal(c.get_scores(), want_scores)
        c.close()
    single.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_max_features_equal_single(world, oracle_lib):
    """--max-features on feature-sharded ranks: a node's feature subset is a function of
    (seed, tree, node, feature), the same on every rank; trees equal the single-context ones."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=70, docs_per_query=40, F=33, seed=31)
    rng = np.random.default_rng(4)
    lam, w = oracle_lib.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff)

    def run(ctxs):
        out = []
        for c in ctxs:
            c.set_pseudo(lam, w)
            c.set_max_features(0.3, seed=5)
        for it in range(3):
            out.append([ctxs[0].fit_tree(8, 2, True)] if ctxs[0].world == 1 else _sharded_fit(torch, ctxs, 8, 2))
        return out

    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    want = run([single])
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        ctxs.append(c)
    got = run(ctxs)
    for it in range(3):
        for g in got[it]:
            for k in want[it][0].dtype.names:
                assert np.array_equal(g[k], want[it][0][k]), (it, k)
    for c in ctxs + [single]:
        c.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_oblivious_subsample_equal_single(world, oracle_lib):
    """Oblivious trees on a sample, feature-sharded: every rank draws the same sample and grows
    the single-context tree; every document's score is updated."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=70, docs_per_query=40, F=33, seed=29)

    def run(ctxs):
        trees = []
        for c in ctxs:
            c.reset_scores()
            c.set_subsample(0.5, 21)
        for it in range(3):
            for c in ctxs:
                c.compute_lambdas("NDCG", 10)
            t = [ctxs[0].fit_oblivious(4, 2, True)] if ctxs[0].world == 1 else _sharded_obl_fit(torch, ctxs, 4, 2)
            for c in ctxs:
                c.update_scores(0.1)
            trees.append(t)
        return trees, [c.get_scores() for c in ctxs]

    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    want_t, want_s = run([single])
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        ctxs.append(c)
    got_t, got_s = run(ctxs)
    for it in range(3):
        for g in got_t[it]:
            for k in ("feature", "thr_id", "left", "right", "nsamples"):
                assert np.array_equal(g[k], want_t[it][0][k]), (it, k)
            assert np.allclose(g["value"], want_t[it][0]["value"], rtol=1e-12, atol=1e-15)
    for s_ in got_s:
        assert np.allclose(s_, want_s[0], rtol=1e-12, atol=1e-15)
    for c in ctxs + [single]:
        c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_subsample_equal_single(world, oracle_lib):
    """--subsample on feature-sharded ranks: every rank holds every document and draws the
    same sample (a pure function of seed and iteration); trees and the scores of ALL
    documents (mart.cc:345) equal the single-context run over three iterations."""
    import torch
    import quickrank_amd as qr
    x, labels, qoff = make_dataset(nq=80, docs_per_query=40, F=33, seed=23)

    def run(ctxs):
        trees = []
        for c in ctxs:
            c.reset_scores()
            c.set_subsample(0.5, 77)
        for it in range(3):
            for c in ctxs:
                c.compute_lambdas("NDCG", 10)
            if len(ctxs) == 1 and ctxs[0].world == 1:
                t = [ctxs[0].fit_tree(8, 2, True)]
            else:
                t = _sharded_fit(torch, ctxs, 8, 2)
            for c in ctxs:
                c.update_scores(0.1)
            trees.append(t)
        return trees, [c.get_scores() for c in ctxs]

    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(64)
    want_t, want_s = run([single])
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        ctxs.append(c)
    got_t, got_s = run(ctxs)
    for it in range(3):
        for g in got_t[it]:
            for k in ("feature", "thr_id", "left", "right", "nsamples"):
                assert np.array_equal(g[k], want_t[it][0][k]), (it, k)
            assert np.allclose(g["value"], want_t[it][0]["value"], rtol=1e-12, atol=1e-15)
    for s in got_s:
        assert np.allclose(s, want_s[0], rtol=1e-12, atol=1e-15)
    for c in ctxs + [single]:
        c.close()


def test_torch_distributed_oblivious_world1(oracle_lib):
    """ShardedTreeFitter.fit_oblivious over RCCL with one rank == the single-context tree."""
    import os
    import torch
    import torch.distributed as dist
    import quickrank_amd as qr
    from quickrank_amd.dist import ShardedTreeFitter
    x, labels, qoff = make_dataset(nq=50, docs_per_query=40, F=40, seed=6)
    lam, w = oracle_lib.lambdas(labels, np.zeros(len(labels)), qoff)
    ref = qr.Context(0)
    ref.upload(x, labels, qoff)
    ref.build_bins(64)
    ref.set_pseudo(lam, w)
    want = ref.fit_oblivious(4, 2, True)
    ref.close()
    torch.cuda.set_device(0)
    port = 29700 + os.getpid() % 1000
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        c = qr.Context(0, stream=torch.cuda.current_stream().cuda_stream)
        c.upload(x, labels, qoff)
        c.build_bins(64)
        c.set_pseudo(lam, w)
        got = ShardedTreeFitter(c).fit_oblivious(c, 4, 2, True)
        for k in want.dtype.names:
            assert np.array_equal(got[k], want[k]), k
        c.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lists", [False, True], ids=["slot_histograms", "presorted_lists"])
@pytest.mark.parametrize("world,F,nthr", [(2, 33, 0), (3, 70, 1000), (2, 16, 5000), (4, 21, 0)])
def test_sharded_contexts_with_more_than_255_thresholds(world, F, nthr, lists, oracle_lib, monkeypatch):
    """The reference's default `--num-thresholds 0` (every distinct value a threshold) and any
    value above 255 on FEATURE-sharded contexts (round 3): every rank builds the wide bins of
    its own features, the protocol is the one of the u8 path -- records all-gather, go-left mask
    all-reduce -- and the winning slot's threshold VALUE, which only its owner holds, rides behind
    the mask.  Trees bit-identical to the single wide context (incl. the thresholds), scores too.
    `presorted_lists` (round 6): every rank grows on the pre-sorted lists of ITS features
    (k_exact.hip; QR_WIDE_EXACT=1 forces them on rows this short) -- the segments of all its lists
    are partitioned by the go-left bits that came back with the reduced mask (k_xflag_mask),
    whoever owns the split feature; the single context beside it stays on slot histograms."""
    import torch
    import quickrank_amd as qr
    monkeypatch.setenv("QR_WIDE_NO_EXACT", "1")
    x, labels, qoff = make_dataset(nq=40, docs_per_query=50, F=F, seed=29, adversarial=True)
    rng = np.random.default_rng(4)
    lam, w = oracle_lib.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(nthr)
    assert single.wide
    single.set_pseudo(lam, w)
    want = single.fit_tree(10, 2, True)
    single.set_scores(np.zeros(len(labels)))
    single.update_scores(0.1)
    want_scores = single.get_scores()
    single.close()
    if lists:
        monkeypatch.delenv("QR_WIDE_NO_EXACT")
        monkeypatch.setenv("QR_WIDE_EXACT", "1")
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world)
        c.upload(x, labels, qoff)
        c.build_bins(nthr)
        assert c.wide
        c.set_pseudo(lam, w)
        ctxs.append(c)
    if lists:   # (a context on the lists has no node histograms to hand out: that is how one tells)
        _sharded_fit(torch, ctxs, 2, 2)
        with pytest.raises(qr.QrError, match="pre-sorted lists"):
            ctxs[0].node_hist_ragged(0)
    got = _sharded_fit(torch, ctxs, 10, 2)
    for g in got:
        assert_same_tree_records(g, want)       # both sides grow one split per step: every field exact
        assert np.all(g["threshold"][g["feature"] >= 0] != 0.0) or nthr == 0
    for c in ctxs:
        c.set_scores(np.zeros(len(labels)))
        c.update_scores(0.1)
        assert np.array_equal(c.get_scores(), want_scores)
        c.close()
