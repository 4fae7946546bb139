"""The PRODUCTION multi-GPU drivers (quickrank_amd/dist.py: ShardedTreeFitter,
DocShardedTrainer) on several rank contexts of ONE process, each in its own thread with its
own HIP stream, their collectives carried by a lockstep transport that has RcclComm's
interface.  The transport

  * records every call (kind, element type, count) per rank and asserts that all ranks make
    the SAME call at every step -- a mismatch is what deadlocks or corrupts a real RCCL run;
  * asserts that the stream it was created with is still the context's stream at every call
    (RcclComm enqueues on exactly that stream: a rebound context would race its buffers);
  * performs the exchange on the contexts' device buffers only after every rank's stream has
    drained (what stream order guarantees a real collective).

So a run on real hardware can only differ in RCCL itself.  Checked against the single-context
run (feature layout: bit-identical trees; document layout: identical structure) and against
the expected call sequence of each driver loop."""
import threading

import numpy as np
import pytest

from datagen import make_dataset
from parity_util import assert_same_tree_records

pytestmark = pytest.mark.gpu


class Hub:
    def __init__(self, torch, world):
        self.torch, self.world = torch, world
        self.bar = threading.Barrier(world)
        self.calls = [[] for _ in range(world)]
        self.slot = [None] * world
        self.errors = []
        self.dev = torch.device("cuda", 0)


class LockstepTransport:
    """RcclComm's interface over the rank contexts of one process."""

    def __init__(self, hub, rank, ctx):
        self.hub, self.rank, self.world, self.nranks, self.ctx = hub, rank, hub.world, hub.world, ctx
        self.stream = ctx.stream_handle()
        assert self.stream != 0, "every rank context runs on a stream of its own"

    def _view(self, ptr, nbytes, typestr, itemsize):
        from quickrank_amd.dist import _DevArray
        return self.hub.torch.as_tensor(_DevArray(ptr, nbytes, typestr, itemsize), device=self.hub.dev)

    def _step(self, call, payload, exchange):
        hub = self.hub
        assert self.ctx.stream_handle() == self.stream, "the context left the stream its communicator uses"
        hub.calls[self.rank].append(call)
        self.ctx.synchronize()          # stream order: what was enqueued before the collective has run
        hub.slot[self.rank] = (call, payload)
        try:
            if hub.bar.wait(timeout=120) == 0:
                calls = [s[0] for s in hub.slot]
                if any(c != calls[0] for c in calls):
                    hub.errors.append(f"ranks disagree on a collective: {calls}")
                else:
                    exchange([s[1] for s in hub.slot])
                    hub.torch.cuda.synchronize()
            hub.bar.wait(timeout=120)
        except threading.BrokenBarrierError:
            raise RuntimeError("a rank left the collective sequence: " + "; ".join(hub.errors))
        if hub.errors:
            raise RuntimeError(hub.errors[0])

    def _sum(self, typestr, itemsize, ptr, count):
        def exchange(ptrs):
            views = [self._view(p, count * itemsize, typestr, itemsize) for p in ptrs]
            tot = views[0].clone()
            for v in views[1:]:
                tot += v
            for v in views:
                v.copy_(tot)
        self._step(("all_reduce_sum", typestr, int(count)), ptr, exchange)

    def all_reduce_i64(self, ptr, count):
        self._sum("<i8", 8, ptr, count)

    def all_reduce_i32(self, ptr, count):
        self._sum("<i4", 4, ptr, count)

    def all_gather_bytes(self, send_ptr, recv_ptr, nbytes):
        def exchange(pairs):
            cat = self.hub.torch.cat([self._view(s, nbytes, "|u1", 1) for s, _ in pairs])
            for _, r in pairs:
                self._view(r, nbytes * self.world, "|u1", 1).copy_(cat)
        self._step(("all_gather", "|u1", int(nbytes)), (send_ptr, recv_ptr), exchange)

    def all_gather_host(self, obj):
        def exchange(objs):
            self.hub.gathered = list(objs)
        self._step(("all_gather_host", "obj", 0), obj, exchange)
        # (read before returning: the next exchange cannot start until every rank has come back)
        return list(self.hub.gathered)

    def close(self):
        pass


def _run_ranks(world, body):
    """body(rank) on one thread per rank; re-raises the first failure."""
    out, errs = [None] * world, []

    def run(r):
        try:
            out[r] = body(r)
        except BaseException as e:  # noqa: BLE001 -- reported to the main thread
            errs.append(e)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    if errs:
        raise errs[0]
    return out


def _equal_trees(a, b, values=True):
    if values:   # (internal nodes' f64 sums: two fixed summation orders, see parity_util)
        assert_same_tree_records(a, b, node_sums_exact=False)
        return
    assert len(a) == len(b)
    for k in a.dtype.names:
        if a[k].dtype.kind != "f" or k == "threshold":
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k


@pytest.mark.parametrize("world", [2, 3])
def test_feature_sharded_drivers_through_lockstep_transport(world, oracle_lib):
    import torch
    import quickrank_amd as qr
    from quickrank_amd.dist import ShardedTreeFitter
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=70, seed=17, adversarial=True)
    rng = np.random.default_rng(2)
    lam, w = oracle_lib.lambdas(labels, rng.standard_normal(len(labels)) * 0.3, qoff)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.set_pseudo(lam, w)
    want = single.fit_tree(12, 3, True)
    want_obl = single.fit_oblivious(4, 2, True)
    single.close()
    hub = Hub(torch, world)
    streams = [torch.cuda.Stream() for _ in range(world)]
    ctxs = []
    for r in range(world):
        c = qr.Context(0, rank=r, world=world, stream=streams[r].cuda_stream)
        c.upload(x, labels, qoff)
        c.build_bins(255)
        c.set_pseudo(lam, w)
        ctxs.append(c)

    def body(r):
        torch.cuda.set_device(0)
        f = ShardedTreeFitter(ctxs[r], transport=LockstepTransport(hub, r, ctxs[r]))
        assert (f.world, f.rank) == (world, r) and f.dist is None
        t = f.fit_tree(ctxs[r], 12, 3, True)
        n_leafwise = len(hub.calls[r])
        o = f.fit_oblivious(ctxs[r], 4, 2, True)
        return t, o, n_leafwise
    res = _run_ranks(world, body)
    for t, o, _ in res:
        _equal_trees(t, want)          # every feature's histogram lives on one rank: bit-identical
        _equal_trees(o, want_obl)
    # the call sequence: identical on every rank, and the one the driver loops prescribe
    assert all(c == hub.calls[0] for c in hub.calls)
    b = ctxs[0].exchange_buffers()
    gather = ("all_gather", "|u1", b["rec_bytes"])
    mask = ("all_reduce_sum", "<i4", b["mask_bytes"] // 4)
    n_lw = res[0][2]
    assert hub.calls[0][:n_lw] == [gather] + [mask, gather] * 11
    ob = ctxs[0].obl_exchange_buffers()
    omask = ("all_reduce_sum", "<i4", ob["mask_bytes"] // 4)
    assert hub.calls[0][n_lw:] == [gather, omask] * 4
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("world,cuts,batched", [(2, [30], True), (3, [5, 41], True), (2, [30], False), (3, [5, 41], False)])
def test_document_sharded_drivers_through_lockstep_transport(world, cuts, batched, monkeypatch):
    import torch
    monkeypatch.setenv("QR_DOC_BATCH", "1" if batched else "0")
    import quickrank_amd as qr
    from quickrank_amd.dist import DocShardedTrainer
    from test_gpu_docshard import _make_ctxs, _split_queries
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=40, seed=23, adversarial=True)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.reset_scores()
    want, want_ndcg = [], []
    for it in range(3):
        single.compute_lambdas("NDCG", 10)
        want.append(single.fit_tree(8, 2, True) if it < 2 else single.fit_oblivious(3, 2, True))
        single.update_scores(0.1)
        want_ndcg.append(single.metric_last())
    want_final = single.metric_eval(0, "NDCG", 10)
    single.close()
    parts = _split_queries(qoff, cuts)
    ctxs, _, _ = _make_ctxs(qr, x, labels, qoff, parts, 255)
    streams = [torch.cuda.Stream() for _ in range(world)]
    for c, s in zip(ctxs, streams):
        c.set_stream(s.cuda_stream)
        c.reset_scores()
    hub = Hub(torch, world)

    def body(r):
        torch.cuda.set_device(0)
        tr = DocShardedTrainer(ctxs[r], transport=LockstepTransport(hub, r, ctxs[r]))
        trees, marks = [], []
        for it in range(3):
            tr.compute_lambdas("NDCG", 10)
            trees.append(tr.fit_tree(8, 2, True) if it < 2 else tr.fit_oblivious(3, 2, True))
            ctxs[r].update_scores(0.1)
            marks.append(len(hub.calls[r]))
        return trees, marks, tr.metric_eval(0, "NDCG", 10)
    res = _run_ranks(world, body)
    for trees, _, final in res:
        for t, wt in zip(trees, want):
            _equal_trees(t, wt, values=False)      # structure bit for bit
            assert np.allclose(t["value"], wt["value"], rtol=1e-9, atol=1e-12, equal_nan=True)   # (the root: a sum of lambdas that cancel)
        assert abs(final - want_final) < 1e-12
    assert all(c == hub.calls[0] for c in hub.calls)
    b = ctxs[0].doc_exchange_buffers()
    scal = ("all_reduce_sum", "<i8", b["scal_n"])
    hist = ("all_reduce_sum", "<i8", b["hist_n"])
    leaf = ("all_reduce_sum", "<i8", b["leaf_n"])
    marks = res[0][1]
    if batched:
        # a leaf-wise iteration, two splits per exchange: scalars, the root's histogram, one buffer of
        # batch cells per step, leaves.  The first tree of a context enqueues the worst case (one
        # split per step), the second as many steps as the first turned out to need (+ what a low
        # guess adds): never more exchanges than the one-split protocol
        p, n = ctxs[0].tree_batch_exchange()
        cells = ("all_reduce_sum", "<i8", n)
        assert hub.calls[0][:marks[0]] == [scal, hist] + [cells] * 7 + [leaf]
        second = hub.calls[0][marks[0]:marks[1]]
        assert second[:2] == [scal, hist] and second[-1] == leaf and all(c == cells for c in second[2:-1])
        assert 4 <= len(second) - 3 <= 7, second
    else:
        # a leaf-wise iteration: scalars, one histogram per node (root + 7 splits), leaves
        assert hub.calls[0][:marks[0]] == [scal] + [hist] * 8 + [leaf]
    # the oblivious one: scalars, root, one LEVEL buffer per level but the last (ot.cc:127), leaves
    obl = hub.calls[0][marks[1]:marks[2]]
    assert obl[0] == scal and obl[1] == hist and len(obl) == 2 + 2 + 1
    assert all(c[0] == "all_reduce_sum" and c[1] == "<i8" for c in obl)
    assert hub.calls[0][marks[2]:] == [("all_gather_host", "obj", 0)]
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("world,cuts,hint", [(2, [30], "1"), (3, [5, 41], "2"), (2, [30], None)])
def test_document_sharded_lazy_trees_through_lockstep_transport(world, cuts, hint, monkeypatch):
    """fit_tree(read=False) on document shards ends a tree behind a GUESSED number of steps, leaf
    kernels and score update enqueued at once; the next call of the trainer looks at the tree's last
    control step and carries the tree on if the guess was too low (QR_STEPS_HINT forces that for
    every tree here).  The trees read afterwards, the scores and the metric are the single
    context's; all ranks make the same calls."""
    import torch
    import quickrank_amd as qr
    from quickrank_amd.dist import DocShardedTrainer
    from test_gpu_docshard import _make_ctxs, _split_queries
    x, labels, qoff = make_dataset(nq=60, docs_per_query=50, F=40, seed=43, adversarial=True)
    single = qr.Context(0)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.reset_scores()
    want, want_ndcg = [], []
    for it in range(5):
        single.compute_lambdas("NDCG", 10)
        want.append(single.fit_tree(10, 2, True))
        single.update_scores(0.1)
        want_ndcg.append(single.metric_last())
    want_final = single.metric_eval(0, "NDCG", 10)
    s1 = single.get_scores()
    single.close()
    parts = _split_queries(qoff, cuts)
    if hint is not None:
        monkeypatch.setenv("QR_STEPS_HINT", hint)
    ctxs, _, _ = _make_ctxs(qr, x, labels, qoff, parts, 255)
    monkeypatch.delenv("QR_STEPS_HINT", raising=False)
    streams = [torch.cuda.Stream() for _ in range(world)]
    for c, s in zip(ctxs, streams):
        c.set_stream(s.cuda_stream)
        c.reset_scores()
    hub = Hub(torch, world)

    def body(r):
        torch.cuda.set_device(0)
        ctx = ctxs[r]
        tr = DocShardedTrainer(ctx, transport=LockstepTransport(hub, r, ctx))
        trees, ndcg = [], []
        for it in range(5):
            tr.compute_lambdas("NDCG", 10)           # (settles tree it - 1)
            if it:
                trees.append(ctx.tree_nodes())
            assert tr.fit_tree(10, 2, True, read=False) is None
            ctx.update_scores(0.1)
            ndcg.append(ctx.metric_last())
        final = tr.metric_eval(0, "NDCG", 10)         # (settles the last tree)
        trees.append(ctx.tree_nodes())
        return trees, ndcg, final, ctx.get_scores(), ctx.spec_stats() if hasattr(ctx, "spec_stats") else None
    res = _run_ranks(world, body)
    assert all(c == hub.calls[0] for c in hub.calls)
    for (trees, ndcg, final, scores, _), (q0, q1) in zip(res, parts):
        for t, wt in zip(trees, want):
            _equal_trees(t, wt, values=False)
            assert np.allclose(t["value"], wt["value"], rtol=1e-9, atol=1e-12, equal_nan=True)
        assert np.allclose(ndcg, want_ndcg, rtol=1e-12)
        assert abs(final - want_final) < 1e-12
        d0, d1 = int(qoff[q0]), int(qoff[q1])
        assert np.allclose(scores, s1[d0:d1], rtol=1e-10, atol=1e-13)
    if hint is not None:
        # every tree needs more than `hint` steps: the carried-on path ran, and its second leaf
        # exchange shows in the call sequence (two per tree)
        b = ctxs[0].doc_exchange_buffers()
        leaf = ("all_reduce_sum", "<i8", b["leaf_n"])
        assert sum(1 for c in hub.calls[0] if c == leaf) == 10
    for c in ctxs:
        c.close()
