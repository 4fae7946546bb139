"""Feature-block sharding across the GPUs of one node.

Rank r owns the contiguous feature range [r*ceil(F/P), (r+1)*ceil(F/P)) of the
bin matrix for ALL documents (17 features per GPU for F = 136, P = 8); labels, scores, pseudo-responses and document lists are replicated,
so every rank makes the same control decisions.  Per split there are exactly two
exchanges, both enqueued on the context's stream with no host synchronisation:

  1. all_gather of the per-rank best-split records (2 x 32 B per rank: left and
     right child of the split just applied) -> every rank merges them with the
     same deterministic rule (max gain, ties -> lowest feature) in k_decide;
  2. all_reduce(sum) of the go-left bit mask of the node being split: only the
     owner of the winning feature contributes non-zero words, so the sum is the
     owner's mask (a broadcast whose root need not be known on the host).

Each feature's histogram is accumulated on exactly one GPU, so the sums -- and
therefore the trees -- are bit-identical to the 1-GPU run.

The transport is torch.distributed (backend "nccl" == RCCL over xGMI on GPUs;
"gloo" in the CPU protocol tests, which drive the same class with a host-side
stand-in context).
"""
import numpy as np


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface)."""

    def __init__(self, ptr, nbytes, typestr="|u1", itemsize=1):
        self.__cuda_array_interface__ = {
            "shape": (nbytes // itemsize,), "typestr": typestr,
            "data": (int(ptr), False), "version": 2}


def _bind_stream(ctx, torch):
    """The collectives below are enqueued on torch's current stream, the kernels on the
    context's: both must be ONE stream, or the exchange buffers race with the kernels
    that fill them.  Real contexts are moved onto torch's current stream here."""
    if hasattr(ctx, "host_buffers") or not torch.cuda.is_available():
        return None
    s = torch.cuda.current_stream().cuda_stream
    if getattr(ctx, "stream", None) != s:
        ctx.set_stream(s)
    return s


def _direct_comm(ctx, torch, dist, group, stream):
    """RCCL straight on the context's stream (RcclComm) unless QR_DIRECT_RCCL=0 or the
    communicator cannot be created -- then torch.distributed carries the collectives."""
    import os
    import sys
    if stream is None or os.environ.get("QR_DIRECT_RCCL", "1") == "0" or dist.get_backend(group) != "nccl":
        return None
    try:
        return RcclComm(dist.get_rank(group), dist.get_world_size(group), stream, group)
    except Exception as e:  # noqa: BLE001 -- any failure falls back to torch's path, loudly
        print(f"quickrank_amd.dist: direct RCCL unavailable ({e}); using torch.distributed", file=sys.stderr)
        return None


class ShardedTreeFitter:
    """Drives qr_tree_begin/decide/apply/end with the collectives in between."""

    def __init__(self, ctx, group=None, device=None, transport=None):
        """`transport`: a ready-made object with RcclComm's interface (rank, world, nranks,
        all_reduce_i64 / all_reduce_i32 / all_gather_bytes on device pointers, close) -- the
        collectives then go through it on the context's OWN stream and no process group is
        touched (tests drive several rank contexts of one process in lockstep that way)."""
        import torch
        self.torch, self.group = torch, group
        # what this rank has handed to collectives so far: [calls, payload bytes] (an all-gather counts what
        # every rank receives: world x the record; an all-reduce the buffer once) -- bench.py's per-tree figures
        self.traffic = [0, 0]
        if transport is not None:
            self.dist = None
            self.world, self.rank = transport.world, transport.rank
            b = ctx.exchange_buffers()
            self.rec_bytes = b["rec_bytes"]
            self.direct, self._b = transport, b
            self.recs_local = self.recs_all = self.mask = None
            return
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        b = ctx.exchange_buffers()
        self.rec_bytes = b["rec_bytes"]
        self.direct = None
        if hasattr(ctx, "host_buffers"):        # CPU protocol stand-in (tests)
            hb = ctx.host_buffers()
            self.recs_local = torch.from_numpy(hb["recs_local"])
            self.recs_all = torch.from_numpy(hb["recs_all"])
            self.mask = torch.from_numpy(hb["mask"])
        else:
            stream = _bind_stream(ctx, torch)
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            self.recs_local = torch.as_tensor(_DevArray(b["recs_local"], self.rec_bytes), device=dev)
            self.recs_all = torch.as_tensor(_DevArray(b["recs_all"], self.rec_bytes * self.world),
                                            device=dev)
            self.mask = torch.as_tensor(_DevArray(b["mask"], b["mask_bytes"], "<i4", 4), device=dev)
            self.direct = _direct_comm(ctx, torch, dist, group, stream)
            self._b = b

    def _gather_records(self):
        self.traffic[0] += 1
        self.traffic[1] += self.rec_bytes * self.world
        if self.direct is not None:
            self.direct.all_gather_bytes(self._b["recs_local"], self._b["recs_all"], self.rec_bytes)
            return
        self.dist.all_gather_into_tensor(self.recs_all, self.recs_local, group=self.group)

    def _reduce_mask(self):
        self.traffic[0] += 1
        self.traffic[1] += self._b["mask_bytes"] if getattr(self, "_b", None) else self.mask.numel() * 4
        if self.direct is not None:
            self.direct.all_reduce_i32(self._b["mask"], self._b["mask_bytes"] // 4)
            return
        self.dist.all_reduce(self.mask, op=self.dist.ReduceOp.SUM, group=self.group)

    def fit_tree(self, ctx, nleaves, minls, newton, read=True):
        ctx.tree_begin(nleaves, minls)
        self._gather_records()
        for _ in range(nleaves - 1):
            ctx.tree_decide()
            self._reduce_mask()
            ctx.tree_apply()
            self._gather_records()
        ctx.tree_decide()
        if not read:
            ctx.tree_end_local(newton)        # records later: ctx.tree_nodes()
            return None
        return ctx.tree_end(nleaves, newton)

    def fit_oblivious(self, ctx, depth, minls, newton, read=True):
        """ObliviousRT::fit (ot.cc:32-201) on feature-sharded ranks: per level an
        all-gather of the ranks' best (feature, slot) of the level and a sum all-reduce
        of the owner's go-left bits (one per DOCUMENT: every node of a level takes the
        same split) + the level's left counts."""
        if not hasattr(self, "_ob"):
            b = ctx.obl_exchange_buffers()
            self._ob = b
            if hasattr(ctx, "host_buffers"):
                self._omask = self.torch.from_numpy(ctx.host_buffers()["obl_mask"])
            elif self.dist is None:
                self._omask = None          # (injected transport: device pointers only)
            else:
                dev = self.recs_local.device
                self._omask = self.torch.as_tensor(_DevArray(b["mask"], b["mask_bytes"], "<i4", 4), device=dev)
        ctx.obl_begin(depth, minls)
        for level in range(depth):
            ctx.obl_propose(level)
            self._gather_records()
            ctx.obl_mark(level)
            self.traffic[0] += 1
            self.traffic[1] += self._ob["mask_bytes"]
            if self.direct is not None:
                self.direct.all_reduce_i32(self._ob["mask"], self._ob["mask_bytes"] // 4)
            else:
                self.dist.all_reduce(self._omask, op=self.dist.ReduceOp.SUM, group=self.group)
            ctx.obl_apply(level)
        return ctx.obl_end(depth, newton, read=read)


class RcclComm:
    """ncclAllReduce / ncclAllGather straight from the RCCL torch already loaded, on the
    context's OWN stream: no hop to torch's collective stream, hence none of the
    cross-stream event waits a torch.distributed call brackets every collective with
    (~6 us of bubble each on this GPU).  torch.distributed is used once, to hand out the
    ncclUniqueId.  The default on GPU runs (QR_DIRECT_RCCL=0 turns it off)."""

    def __init__(self, rank, world, stream, group=None):
        import ctypes as C
        import os
        import torch
        import torch.distributed as dist
        self.C, self.stream = C, C.c_void_p(stream)
        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.L = L = C.CDLL(lib)
        L.ncclGetErrorString.restype = C.c_char_p
        uid = (C.c_byte * 128)()
        if rank == 0:
            self._ck(L.ncclGetUniqueId(C.byref(uid)))
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = (C.c_byte * 128).from_buffer_copy(box[0])
        self.comm = C.c_void_p()
        # ncclUniqueId is a 128-byte struct passed by value
        class Uid(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
        u = Uid()
        C.memmove(C.byref(u), uid, 128)
        self._ck(L.ncclCommInitRank(C.byref(self.comm), world, u, rank))
        L.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                    C.c_void_p]
        L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        n = C.c_int(0)
        L.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._ck(L.ncclCommCount(self.comm, C.byref(n)))
        self.nranks = n.value

    def _ck(self, rc):
        if rc:
            raise RuntimeError(f"RCCL: {self.L.ncclGetErrorString(rc).decode()} (code {rc})")

    def all_reduce_i64(self, ptr, count):
        NCCL_INT64, NCCL_SUM = 4, 0
        self._ck(self.L.ncclAllReduce(self.C.c_void_p(ptr), self.C.c_void_p(ptr), count, NCCL_INT64,
                                      NCCL_SUM, self.comm, self.stream))

    def all_reduce_i32(self, ptr, count):
        NCCL_INT32, NCCL_SUM = 2, 0
        self._ck(self.L.ncclAllReduce(self.C.c_void_p(ptr), self.C.c_void_p(ptr), count, NCCL_INT32,
                                      NCCL_SUM, self.comm, self.stream))

    def all_gather_bytes(self, send_ptr, recv_ptr, nbytes):
        NCCL_INT8 = 0
        self._ck(self.L.ncclAllGather(self.C.c_void_p(send_ptr), self.C.c_void_p(recv_ptr), nbytes,
                                      NCCL_INT8, self.comm, self.stream))

    def close(self):
        if self.comm:
            self.L.ncclCommDestroy(self.comm)
            self.comm = None


class DocShardedTrainer:
    """Document sharding: rank r holds its own queries (all features of them).

    The exchange is the histogram itself.  Its cells are exact fixed-point
    integers, so ONE int64 sum all-reduce per node histogram leaves every rank
    with the bits a single GPU would have computed over all documents; scan, gain
    and the heap-driven growth then run redundantly on every rank and agree without
    any further exchange.  The partition is local (every rank has every feature of
    its documents): no mask.  f64 quantities (sum and sum of squares of the
    pseudo-responses, leaf sums, the metric) ride as bit patterns in per-rank
    slots of int64 buffers -- sum == gather -- and are added in rank order.
    Collectives per boosting iteration: 1 (scalars) + nleaves (histograms) + 1
    (leaves), none of which needs a host synchronisation.

    This is the layout that scales with the number of documents (weak scaling):
    the per-rank work is the single-GPU work on its own shard.  The feature-
    sharded ShardedTreeFitter keeps every per-document step replicated.
    """

    def __init__(self, ctx, group=None, device=None, transport=None):
        """`transport`: see ShardedTreeFitter -- RcclComm's interface on the context's own
        stream, no process group."""
        import torch
        self.torch, self.group, self.ctx = torch, group, ctx
        self.device = device
        self.traffic = [0, 0]          # [collectives, payload bytes] handed over so far (see ShardedTreeFitter)
        if transport is not None:
            self.dist = None
            self.world, self.rank = transport.world, transport.rank
            b = ctx.doc_exchange_buffers()
            self.hist = self.scal = self.leaf = None
            self.leaf_n = 0
            self._ptr = {"hist": (b["hist"], b["hist_n"]), "scal": (b["scal"], b["scal_n"])}
            self.direct = transport
            return
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        stream = _bind_stream(ctx, torch)
        b = ctx.doc_exchange_buffers()
        self.hist = self._view(b["hist"], b["hist_n"], "hist")
        self.scal = self._view(b["scal"], b["scal_n"], "scal")
        self.leaf = None
        self.leaf_n = 0
        # device (pointer, count) of the named exchange buffers, for the direct path
        self._ptr = {"hist": (b["hist"], b["hist_n"]), "scal": (b["scal"], b["scal_n"])}
        self.direct = None if hasattr(ctx, "host_buffers") else _direct_comm(ctx, torch, dist, group, stream)

    def _view(self, ptr, n, name):
        torch = self.torch
        if self.dist is None:                        # injected transport: device pointers only
            return None
        if hasattr(self.ctx, "host_buffers"):        # CPU protocol stand-in (tests)
            return torch.from_numpy(self.ctx.host_buffers()[name])
        dev = self.device if self.device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        return torch.as_tensor(_DevArray(ptr, n * 8, "<i8", 8), device=dev)

    def _sum(self, t, name=None):
        """Sum all-reduce of one of the context's exchange buffers (`name`) or of a
        temporary tensor (name None: always through torch.distributed)."""
        self.traffic[0] += 1
        self.traffic[1] += self._ptr[name][1] * 8 if name is not None else t.numel() * t.element_size()
        if self.direct is not None and name is not None:
            self.direct.all_reduce_i64(*self._ptr[name])
            return
        if self.dist is None:
            raise RuntimeError("an injected transport carries the context's exchange buffers only")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def compute_lambdas(self, metric="NDCG", cutoff=10):
        self.settle()
        self.ctx.compute_lambdas(metric, cutoff)
        self._sum(self.scal, "scal")
        self.ctx.lambda_finish()

    def compute_residuals(self):
        self.settle()
        self.ctx.compute_residuals()
        self._sum(self.scal, "scal")
        self.ctx.lambda_finish()

    def fit_tree(self, nleaves, minls, newton, read=True, batched=None):
        """`batched` (default: whenever the context can, QR_DOC_BATCH=0 turns it off): up to two
        splits per exchange -- 1 + steps histogram all-reduces per tree instead of nleaves."""
        ctx = self.ctx
        self.settle()
        if batched is None:
            import os
            batched = os.environ.get("QR_DOC_BATCH", "1") != "0"
        if batched and hasattr(ctx, "tree_batch_supported") and ctx.tree_batch_supported(nleaves):
            return self._fit_tree_batched(nleaves, minls, newton, read)
        ctx.tree_begin(nleaves, minls)
        self._sum(self.hist, "hist")
        for _ in range(nleaves - 1):
            ctx.tree_decide()
            ctx.tree_apply()
            self._sum(self.hist, "hist")
        ctx.tree_decide()
        ctx.tree_end_local(newton)
        b = ctx.doc_exchange_buffers()
        if self.leaf is None or self._ptr.get("leaf") != (b["leaf"], b["leaf_n"]):
            self.leaf = self._view(b["leaf"], b["leaf_n"], "leaf")
            self.leaf_n = b["leaf_n"]
            self._ptr["leaf"] = (b["leaf"], b["leaf_n"])
        self._sum(self.leaf, "leaf")
        return ctx.tree_leaves_finish(nleaves, newton, read=read)


    def _leaf_exchange(self, nleaves, newton, read):
        ctx = self.ctx
        ctx.tree_end_local(newton)
        b = ctx.doc_exchange_buffers()
        if self.leaf is None or self._ptr.get("leaf") != (b["leaf"], b["leaf_n"]):
            self.leaf = self._view(b["leaf"], b["leaf_n"], "leaf")
            self.leaf_n = b["leaf_n"]
            self._ptr["leaf"] = (b["leaf"], b["leaf_n"])
        self._sum(self.leaf, "leaf")
        return ctx.tree_leaves_finish(nleaves, newton, read=read)

    def _batch_steps(self, k):
        ctx = self.ctx
        for s in range(k):
            ctx.tree_batch_apply()
            self._sum(self.batch, "batch")
            ctx.tree_batch_decide(s == k - 1)
        self.collectives += k

    def _carry_on(self, nleaves, done):
        """the enqueued steps did not suffice (the last control step said so): more of them, in
        doubling pieces, looking at the last one of each"""
        piece = 1
        while True:
            k = max(1, min(piece, nleaves - 1 - done))
            self._batch_steps(k)
            done += k
            piece *= 2
            if not self.ctx.tree_batch_settle()[0]:
                return

    def settle(self):
        """A tree enqueued with read=False ends behind a GUESSED number of steps, with its leaf
        kernels and score update already in the queue (they leave at once on an incomplete tree).
        Whatever consumes the tree or the scores next looks at the tree's last control step here
        and, if the guess was too low, carries the tree on and ends it again.  Every rank grows
        the same trees, so every rank takes the same path."""
        p, self._pending = getattr(self, "_pending", None), None
        if p is None:
            return
        nleaves, newton, done = p
        if self.ctx.tree_batch_settle()[0]:
            self._carry_on(nleaves, done)
            self._leaf_exchange(nleaves, newton, read=False)   # (repeats the score update if one was enqueued)

    def tree_nodes(self):
        """records of the last tree (settled first)"""
        self.settle()
        return self.ctx.tree_nodes()

    def _fit_tree_batched(self, nleaves, minls, newton, read):
        """RegressionTree::fit (rt.cc:58-90) with up to two splits per exchange: the one-GPU
        batched growth (qr_tree_fit) cut at the all-reduces.  The number of steps enqueued is a
        guess (the previous tree's); the last control step tells whether it sufficed.  read=False
        does not wait for it: see settle()."""
        ctx = self.ctx
        steps = ctx.tree_batch_begin(nleaves, minls)
        self._sum(self.hist, "hist")
        ctx.tree_batch_root()
        ptr, n = ctx.tree_batch_exchange()
        if self._ptr.get("batch") != (ptr, n):
            self._ptr["batch"] = (ptr, n)
            self.batch = self._view(ptr, n, "batch")
        self.collectives = 1
        self._batch_steps(steps)
        if read:
            if ctx.tree_batch_settle()[0]:
                self._carry_on(nleaves, steps)
        else:
            self._pending = (nleaves, newton, steps)
        return self._leaf_exchange(nleaves, newton, read)

    def fit_oblivious(self, depth, minls, newton, read=True):
        """ObliviousRT::fit (ot.cc:32-201) over document shards: ONE int64 all-reduce per
        level -- the cells of all its directly built children -- after the root's."""
        ctx = self.ctx
        self.settle()
        ctx.obl_begin(depth, minls)
        self._sum(self.hist, "hist")
        for level in range(depth):
            ctx.obl_propose(level)
            ctx.obl_apply(level)
            if level + 1 < depth:           # ot.cc:127: no histograms for the leaves
                ptr, n = ctx.obl_level_exchange(level)
                self._ptr["level"] = (ptr, n)
                self._sum(self._view(ptr, n, "level"), "level")
        ctx.tree_end_local(newton)
        b = ctx.doc_exchange_buffers()
        if self.leaf is None or self._ptr.get("leaf") != (b["leaf"], b["leaf_n"]):
            self.leaf = self._view(b["leaf"], b["leaf_n"], "leaf")
            self.leaf_n = b["leaf_n"]
            self._ptr["leaf"] = (b["leaf"], b["leaf_n"])
        self._sum(self.leaf, "leaf")
        return ctx.tree_leaves_finish((1 << depth), newton, read=read)

    def metric_eval(self, which=0, metric="NDCG", cutoff=10):
        """Metric::evaluate_dataset (metric.h:77-106) over ALL ranks' queries of the
        training (0) or validation (1) set: every rank evaluates its own queries on
        the device; the per-rank (sum, count) pairs are all-gathered and added in
        rank order."""
        self.settle()
        local = self.ctx.metric_eval(which, metric, cutoff)
        nq = self.ctx.Q if which == 0 else self.ctx.vQ
        if self.dist is None:   # injected transport: the per-rank pairs through its host-side gather
            pairs = self.direct.all_gather_host((local * nq, float(nq)))
            total = sum(float(a) for a, _ in pairs)
            count = sum(float(b_) for _, b_ in pairs)
            return total / count if count else 0.0
        t = self.torch.zeros(2 * self.world, dtype=self.torch.float64)
        t[2 * self.rank] = local * nq
        t[2 * self.rank + 1] = nq
        if self.hist.is_cuda:
            t = t.to(self.hist.device)
        self._sum(t)
        t = t.cpu().numpy().reshape(self.world, 2)
        total = sum(float(a) for a in t[:, 0])
        count = sum(float(a) for a in t[:, 1])
        return total / count if count else 0.0


class FeatureShardedTrainer:
    """The interface of DocShardedTrainer over north_star's FEATURE layout: every rank holds every
    document (so the lambda pass, the metric and the score update are the rank's own, redundantly)
    and the bin columns / pre-sorted lists of its own feature range; per split the ranks all-gather
    their best records and all-reduce the go-left mask (ShardedTreeFitter).  What a document-sharded
    run falls back to when `--num-thresholds 0` meets columns whose every-distinct-value rows a
    document-sharded node histogram cannot hold (scripts/train_multi_gpu.py): the best split over
    slots is a function of prefix sums over ALL documents, which shards by feature."""

    def __init__(self, ctx, group=None, device=None, transport=None):
        self.ctx = ctx
        self.fitter = ShardedTreeFitter(ctx, group=group, device=device, transport=transport)
        self.traffic = self.fitter.traffic

    def compute_lambdas(self, metric="NDCG", cutoff=10):
        self.ctx.compute_lambdas(metric, cutoff)

    def compute_residuals(self):
        self.ctx.compute_residuals()

    def fit_tree(self, nleaves, minls, newton, read=True):
        return self.fitter.fit_tree(self.ctx, nleaves, minls, newton, read)

    def fit_oblivious(self, depth, minls, newton, read=True):
        return self.fitter.fit_oblivious(self.ctx, depth, minls, newton, read)

    def metric_eval(self, which=0, metric="NDCG", cutoff=10):
        return self.ctx.metric_eval(which, metric, cutoff)


def gather_thresholds(ctx, nthresholds, group=None):
    """Thresholds of the whole (document-sharded) training set: every rank's column
    statistics are all-gathered and merged with the reference's rule."""
    import torch.distributed as dist
    from ._capi import thresholds_from_stats
    vals, cnt, mm = ctx.bins_stats(nthresholds)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (vals, cnt, mm), group=group)
    return thresholds_from_stats(ctx.F, nthresholds, np.stack([p[0] for p in parts]),
                                 np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]))


def build_doc_bins(ctx, nthresholds, group=None, distinct_limit=65536):
    """Mart::init on a document-sharded rank: the thresholds of the WHOLE training set from every
    rank's column statistics, then this rank's bins.  Up to 255 thresholds per feature take the u8
    path; more -- nthresholds > 255, or 0 on a column with more distinct values -- the wide one
    (ragged rows; `distinct_limit` bounds the distinct values gathered per column when
    nthresholds == 0).  Returns (thresholds, thr_size) as Context.thresholds() does."""
    import torch.distributed as dist
    from ._capi import QR_ERR_UNSUPPORTED, QrError, thresholds_from_stats_wide
    if 0 <= nthresholds <= 255:
        # u8 or wide is decided by the MERGED statistics, which every rank holds alike: all ranks
        # take the same branch.  Only "more than 255 distinct values" (nthresholds == 0) leads to
        # the wide path; any other failure -- and any failure of this rank's own bin build --
        # is raised as it is (a rank that fell through would wait in the next all-gather alone).
        merged = None
        try:
            merged = gather_thresholds(ctx, nthresholds, group)
        except QrError as e:
            if nthresholds != 0 or e.code != QR_ERR_UNSUPPORTED:
                raise
        if merged is not None:
            ctx.build_bins_with(*merged)
            return ctx.thresholds()
    limit = nthresholds + 1 if nthresholds else distinct_limit
    vals, cnt, mm = ctx.bins_stats_wide(limit)
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, (vals, cnt, mm), group=group)
    flat, ts = thresholds_from_stats_wide(ctx.F, nthresholds, limit, np.stack([p[0] for p in parts]),
                                          np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]))
    ctx.build_bins_wide_with(flat, ts)
    return ctx.thresholds()


def owned_features(F, rank, world):
    """Global feature indices rank owns (same rule as qr_ctx_set_shard)."""
    per = (F + world - 1) // world
    lo = min(F, per * rank)
    return np.arange(lo, min(F, lo + per), dtype=np.int64)
