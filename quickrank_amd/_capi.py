"""ctypes binding of include/qr_hip.h (the C-ABI of libqr_hip.so).

Fails loudly when the HIP library is missing or no GPU is visible: there is no
CPU fallback anywhere in this package.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

from . import build as _build

NODE_DTYPE = np.dtype([("feature", np.int32), ("thr_id", np.int32),
                       ("threshold", np.float32), ("left", np.int32),
                       ("right", np.int32), ("value", np.float64),
                       ("deviance", np.float64), ("nsamples", np.uint64)],
                      align=True)
SPLIT_DTYPE = np.dtype([("score", np.float64), ("feature", np.uint32),
                        ("thr_id", np.uint32), ("lcount", np.uint64),
                        ("rcount", np.uint64)], align=True)
assert NODE_DTYPE.itemsize == 48 and SPLIT_DTYPE.itemsize == 32

QR_MAX_BINS = 256
METRICS = {"DCG": 0, "NDCG": 1}

# every symbol include/qr_hip.h declares
SYMBOLS = [
    "qr_ctx_create", "qr_ctx_destroy", "qr_last_error", "qr_ctx_set_stream",
    "qr_ctx_set_shard", "qr_synchronize", "qr_dataset_upload", "qr_valid_upload",
    "qr_bins_build", "qr_bins_read", "qr_bins_read_fm", "qr_bins_verify", "qr_debug_bins_clobber", "qr_debug_sample_key_mask", "qr_scores_reset", "qr_scores_set",
    "qr_scores_get", "qr_valid_scores_get", "qr_pseudo_get", "qr_pseudo_set",
    "qr_lambda_compute", "qr_residual_compute", "qr_metric_eval", "qr_metric_last",
    "qr_tree_fit", "qr_oblivious_fit", "qr_scores_update", "qr_tree_begin",
    "qr_tree_decide", "qr_tree_apply", "qr_tree_end", "qr_exchange_buffers",
    "qr_node_hist_read", "qr_node_samples_read", "qr_tree_split_log",
    "qr_metric_per_query", "qr_ranks_read", "qr_ensemble_upload",
    "qr_ensemble_score", "qr_ensemble_score_device", "qr_prof_reset",
    "qr_prof_get", "qr_prof_enable", "qr_oblivious_upload", "qr_oblivious_score",
    "qr_ctx_set_doc_shard", "qr_bins_stats", "qr_thresholds_from_stats",
    "qr_bins_build_with", "qr_lambda_finish", "qr_tree_leaves_finish",
    "qr_doc_exchange_buffers", "qr_tree_nodes", "qr_valid_scores_set",
    "qr_tree_set_max_features", "qr_subsample_set", "qr_subsample_set_doc", "qr_ensemble_partial_scores",
    "qr_prof_get_child", "qr_bins_build_wide", "qr_thresholds_read", "qr_bins_read_u32",
    "qr_node_hist_read_ragged", "qr_ctx_stream", "qr_obl_begin", "qr_obl_propose", "qr_obl_mark",
    "qr_obl_apply", "qr_obl_exchange_buffers", "qr_obl_level_exchange", "qr_prof_lds_atomic",
    "qr_tree_batch_supported", "qr_tree_batch_begin", "qr_tree_batch_root", "qr_tree_batch_apply",
    "qr_tree_batch_decide", "qr_tree_batch_settle", "qr_tree_batch_exchange",
    "qr_ensemble_set_depth_order", "qr_bins_build_wide_with",
    "qr_bins_stats_wide", "qr_thresholds_from_stats_wide", "qr_tree_pending", "qr_debug_check",
    "qr_readback_retries",
]

_LIB = None
# re-reads of polled read-backs over every context this process has closed (Context.close)
READBACK_RETRIES = 0
# Live contexts, closed by an atexit hook while the interpreter and the HIP runtime are both
# still whole: a Context that is only collected during interpreter shutdown (a global, a frame
# of a failed test) would otherwise call qr_ctx_destroy -- stream syncs, hipFree, hipHostFree --
# at a point where the order against the runtime's own exit handlers is not defined.
_LIVE = None


def _close_all():
    for ctx in list(_LIVE or ()):
        try:
            ctx.close()
        except Exception:
            pass


class QrError(RuntimeError):
    """`code`: the C-ABI status (include/qr_hip.h), 0 when the error did not come with one."""

    def __init__(self, msg, code=0):
        super().__init__(msg)
        self.code = code


QR_ERR_UNSUPPORTED = 5
_DEBUG = bool(os.environ.get("QR_DEBUG"))


def lib():
    """Load libqr_hip.so (never builds implicitly on import; see build.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if not os.path.exists(path):
        raise QrError(
            f"{path} is missing: build the HIP extension first "
            "(python -m quickrank_amd.build, or __graft_entry__.build()). "
            "quickrank_amd has no CPU fallback.")
    # torch ships its own HIP runtime under the SONAME of /opt/rocm's: whichever copy
    # is mapped first serves the whole process, and torch does not find its GPUs
    # through the system copy (nor this library its own afterwards).  So if torch is
    # there, it goes first -- whatever order the caller imports things in.
    if not os.environ.get("QR_NO_TORCH"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
    L = C.CDLL(path)
    global _LIVE
    _LIVE = weakref.WeakSet()
    atexit.register(_close_all)   # (registered after torch's own hooks: runs before them)
    vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
    L.qr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.qr_ctx_destroy.argtypes = [vp]
    L.qr_ctx_destroy.restype = None
    L.qr_last_error.argtypes = [vp]
    L.qr_last_error.restype = C.c_char_p
    L.qr_ctx_set_stream.argtypes = [vp, vp]
    L.qr_ctx_stream.argtypes = [vp, C.POINTER(vp)]
    L.qr_ctx_set_shard.argtypes = [vp, C.c_int, C.c_int]
    L.qr_synchronize.argtypes = [vp]
    L.qr_tree_pending.argtypes = [vp, C.POINTER(C.c_int)]
    L.qr_debug_check.argtypes = [vp]
    L.qr_readback_retries.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.qr_dataset_upload.argtypes = [vp, vp, sz, sz, vp, vp, sz]
    L.qr_valid_upload.argtypes = [vp, vp, sz, sz, vp, vp, sz]
    L.qr_bins_build.argtypes = [vp, sz, vp, vp]
    L.qr_bins_read.argtypes = [vp, vp]
    L.qr_bins_read_fm.argtypes = [vp, vp]
    L.qr_bins_verify.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.qr_debug_bins_clobber.argtypes = [vp, C.c_int, sz, sz]
    L.qr_debug_sample_key_mask.argtypes = [vp, C.c_uint32]
    L.qr_bins_build_wide.argtypes = [vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.qr_bins_build_wide_with.argtypes = [vp, vp, vp, C.POINTER(sz), C.POINTER(sz)]
    L.qr_bins_stats_wide.argtypes = [vp, sz, vp, vp, vp]
    L.qr_thresholds_from_stats_wide.argtypes = [sz, sz, sz, sz, vp, vp, vp, vp, sz, vp, C.POINTER(sz)]
    L.qr_thresholds_read.argtypes = [vp, vp, vp]
    L.qr_bins_read_u32.argtypes = [vp, vp]
    L.qr_node_hist_read_ragged.argtypes = [vp, C.c_int, vp, vp]
    L.qr_scores_reset.argtypes = [vp]
    L.qr_scores_set.argtypes = [vp, vp]
    L.qr_scores_get.argtypes = [vp, vp]
    L.qr_valid_scores_get.argtypes = [vp, vp]
    L.qr_valid_scores_set.argtypes = [vp, vp]
    L.qr_pseudo_get.argtypes = [vp, vp, vp]
    L.qr_pseudo_set.argtypes = [vp, vp, vp]
    L.qr_lambda_compute.argtypes = [vp, C.c_int, sz]
    L.qr_residual_compute.argtypes = [vp]
    L.qr_metric_eval.argtypes = [vp, C.c_int, C.c_int, sz, C.POINTER(C.c_double)]
    L.qr_metric_last.argtypes = [vp, C.POINTER(C.c_double)]
    L.qr_tree_fit.argtypes = [vp, sz, u64, C.c_int, vp, C.POINTER(sz)]
    L.qr_oblivious_fit.argtypes = [vp, sz, u64, C.c_int, vp, C.POINTER(sz)]
    L.qr_scores_update.argtypes = [vp, C.c_double]
    L.qr_tree_begin.argtypes = [vp, sz, u64]
    L.qr_obl_begin.argtypes = [vp, sz, u64]
    L.qr_obl_propose.argtypes = [vp, sz]
    L.qr_obl_mark.argtypes = [vp, sz]
    L.qr_obl_apply.argtypes = [vp, sz]
    L.qr_obl_exchange_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz),
                                          C.POINTER(vp), C.POINTER(sz)]
    L.qr_obl_level_exchange.argtypes = [vp, sz, C.POINTER(vp), C.POINTER(sz)]
    L.qr_tree_batch_supported.argtypes = [vp, sz]
    L.qr_tree_batch_begin.argtypes = [vp, sz, C.c_uint64, C.POINTER(sz)]
    L.qr_tree_batch_root.argtypes = [vp]
    L.qr_tree_batch_apply.argtypes = [vp]
    L.qr_tree_batch_decide.argtypes = [vp, C.c_int]
    L.qr_tree_batch_settle.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(sz)]
    L.qr_tree_batch_exchange.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.qr_tree_decide.argtypes = [vp]
    L.qr_tree_apply.argtypes = [vp]
    L.qr_tree_end.argtypes = [vp, C.c_int, vp, C.POINTER(sz)]
    L.qr_exchange_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz),
                                      C.POINTER(vp), C.POINTER(sz)]
    L.qr_node_hist_read.argtypes = [vp, C.c_int, vp, vp]
    L.qr_node_samples_read.argtypes = [vp, C.c_int, vp, C.POINTER(sz)]
    L.qr_tree_split_log.argtypes = [vp, vp, C.POINTER(sz)]
    L.qr_metric_per_query.argtypes = [vp, vp]
    L.qr_ranks_read.argtypes = [vp, vp]
    L.qr_ensemble_upload.argtypes = [vp, vp, sz, sz, vp]
    L.qr_ensemble_set_depth_order.argtypes = [vp, C.c_int]
    L.qr_ensemble_score.argtypes = [vp, vp, sz, sz, vp, C.POINTER(C.c_float)]
    L.qr_ensemble_score_device.argtypes = [vp, vp, sz, sz, vp]
    L.qr_ensemble_partial_scores.argtypes = [vp, vp, sz, sz, C.c_int, vp]
    L.qr_oblivious_upload.argtypes = [vp, vp, vp, vp, vp, vp, sz, sz]
    L.qr_oblivious_score.argtypes = [vp, vp, sz, sz, vp, C.POINTER(C.c_float)]
    L.qr_prof_reset.argtypes = [vp]
    L.qr_prof_get.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.qr_prof_enable.argtypes = [vp, C.c_int]
    L.qr_prof_get_child.argtypes = [vp, C.POINTER(u64), C.POINTER(C.c_double)]
    L.qr_prof_lds_atomic.argtypes = [vp] + [C.POINTER(C.c_double)] * 4
    L.qr_ctx_set_doc_shard.argtypes = [vp, C.c_int, C.c_int, u64, u64]
    L.qr_bins_stats.argtypes = [vp, sz, vp, vp, vp]
    L.qr_thresholds_from_stats.argtypes = [sz, sz, sz, vp, vp, vp, vp, vp]
    L.qr_bins_build_with.argtypes = [vp, vp, vp]
    L.qr_tree_nodes.argtypes = [vp, vp, C.POINTER(sz)]
    L.qr_tree_set_max_features.argtypes = [vp, C.c_float, u64]
    L.qr_subsample_set.argtypes = [vp, C.c_float, u64]
    L.qr_subsample_set_doc.argtypes = [vp, C.c_float, u64, sz]
    L.qr_lambda_finish.argtypes = [vp]
    L.qr_tree_leaves_finish.argtypes = [vp, C.c_int, vp, C.POINTER(sz)]
    L.qr_doc_exchange_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp),
                                          C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
    for s in SYMBOLS:
        fn = getattr(L, s)
        if s not in ("qr_ctx_destroy", "qr_last_error"):
            fn.restype = C.c_int
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data if a is not None else None


def thresholds_from_stats(F, nthresholds, vals, cnt, mm):
    """Thresholds of the union of `nranks` document shards (mart.cc:147-169).
    vals/cnt/mm: the per-rank arrays of Context.bins_stats stacked on axis 0."""
    vals = np.ascontiguousarray(vals, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    mm = np.ascontiguousarray(mm, np.uint32)
    nranks = cnt.shape[0]
    thr = np.empty((F, QR_MAX_BINS), np.float32)
    ts = np.empty(F, np.uint32)
    rc = lib().qr_thresholds_from_stats(F, nthresholds, nranks, _ptr(vals), _ptr(cnt), _ptr(mm),
                                        _ptr(thr), _ptr(ts))
    if rc:
        raise QrError(f"qr_thresholds_from_stats failed (code {rc}): nthresholds == 0 needs "
                      "<= 255 distinct values per feature", rc)
    return thr, ts


def thresholds_from_stats_wide(F, nthresholds, limit, vals, cnt, mm):
    """Ragged thresholds (flat f32, thr_size [F]) of the union of the ranks' shards, more than 255
    per feature allowed (mart.cc:140-169); vals/cnt/mm: Context.bins_stats_wide stacked on axis 0."""
    vals = np.ascontiguousarray(vals, np.uint32)
    cnt = np.ascontiguousarray(cnt, np.uint32)
    mm = np.ascontiguousarray(mm, np.uint32)
    nranks = cnt.shape[0]
    ts = np.empty(F, np.uint32)
    cells = C.c_size_t()
    rc = lib().qr_thresholds_from_stats_wide(F, nthresholds, nranks, limit, _ptr(vals), _ptr(cnt), _ptr(mm),
                                             None, 0, _ptr(ts), C.byref(cells))
    if rc:
        raise QrError(f"qr_thresholds_from_stats_wide failed (code {rc}): nthresholds == 0 with a column of "
                      f"more than {limit} distinct values", rc)
    flat = np.empty(cells.value, np.float32)
    rc = lib().qr_thresholds_from_stats_wide(F, nthresholds, nranks, limit, _ptr(vals), _ptr(cnt), _ptr(mm),
                                             _ptr(flat), cells.value, _ptr(ts), C.byref(cells))
    if rc:
        raise QrError(f"qr_thresholds_from_stats_wide failed (code {rc})", rc)
    return flat, ts


class Context:
    """One device context (= one GPU, one stream).  Thin, 1:1 over the C-ABI."""

    def __init__(self, device=0, rank=0, world=1, stream=None, doc_shard=None):
        """doc_shard=(n_global, q_global): document-sharded context (this rank holds
        its own queries and all features); otherwise world > 1 = feature-sharded."""
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.qr_ctx_create(device, C.byref(h))
        if rc:
            raise QrError(f"qr_ctx_create: {self.L.qr_last_error(None).decode()} (code {rc})")
        self.h = h
        _LIVE.add(self)
        self.N = self.F = self.Q = 0
        self.vN = self.vQ = 0
        self.stream = None          # None: the context's own stream
        if stream is not None:
            self.set_stream(stream)
        if doc_shard is not None:
            self._ck(self.L.qr_ctx_set_doc_shard(self.h, rank, world, int(doc_shard[0]),
                                                 int(doc_shard[1])))
        elif world > 1:
            self._ck(self.L.qr_ctx_set_shard(self.h, rank, world))
        self.rank, self.world = rank, world
        self.doc_shard = doc_shard is not None

    def _ck(self, rc):
        if rc:
            raise QrError(f"{self.L.qr_last_error(self.h).decode()} (code {rc})", rc)
        if _DEBUG and self.h:   # QR_DEBUG=1: every call is drained and checked (qr_debug_check)
            rc = self.L.qr_debug_check(self.h)
            if rc:
                raise QrError(f"{self.L.qr_last_error(self.h).decode()} (code {rc})", rc)

    def set_stream(self, stream):
        """Run every launch on the caller's HIP stream (qr_ctx_set_stream)."""
        self._ck(self.L.qr_ctx_set_stream(self.h, C.c_void_p(stream)))
        self.stream = stream

    def stream_handle(self):
        """The HIP stream the context launches on (qr_ctx_stream), as an integer."""
        s = C.c_void_p()
        self._ck(self.L.qr_ctx_stream(self.h, C.byref(s)))
        return s.value or 0

    def readback_retries(self):
        """Re-reads of polled read-backs that did not fit their sequence number at first sight
        (qr_readback_retries); expected 0."""
        n = C.c_ulonglong(0)
        self._ck(self.L.qr_readback_retries(self.h, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "h", None):
            global READBACK_RETRIES
            n = C.c_ulonglong(0)
            if self.L.qr_readback_retries(self.h, C.byref(n)) == 0:
                READBACK_RETRIES += int(n.value)   # (tests/conftest.py asserts the process total is 0)
            self.L.qr_ctx_destroy(self.h)
            self.h = None
            _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ---------------------------------------------------------------
    def upload(self, x, labels, qoff):
        x = np.ascontiguousarray(x, np.float32)
        labels = np.ascontiguousarray(labels, np.float32)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        self.N, self.F = x.shape
        self.Q = len(qoff) - 1
        self._ck(self.L.qr_dataset_upload(self.h, _ptr(x), self.N, self.F, _ptr(labels),
                                          _ptr(qoff), self.Q))

    def upload_device(self, dev_ptr, n, f, labels, qoff):
        """qr_dataset_upload with rows that already live on the context's device (a raw device
        pointer to f32 [n][f]); labels and query offsets come from host arrays as always."""
        labels = np.ascontiguousarray(labels, np.float32)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        self.N, self.F, self.Q = int(n), int(f), len(qoff) - 1
        self._ck(self.L.qr_dataset_upload(self.h, C.c_void_p(int(dev_ptr)), self.N, self.F, _ptr(labels),
                                          _ptr(qoff), self.Q))

    def upload_valid(self, x, labels, qoff):
        x = np.ascontiguousarray(x, np.float32)
        labels = np.ascontiguousarray(labels, np.float32)
        qoff = np.ascontiguousarray(qoff, np.uint64)
        self.vN = x.shape[0]
        self.vQ = len(qoff) - 1
        self._ck(self.L.qr_valid_upload(self.h, _ptr(x), self.vN, x.shape[1], _ptr(labels),
                                        _ptr(qoff), len(qoff) - 1))

    def build_bins(self, nthresholds, wide=None):
        """Thresholds + bin map (Mart::init, mart.cc:117-176).  Up to 255 thresholds per
        feature take the u8 path; more -- `nthresholds` > 255, or 0 ("every distinct
        value") on a column with more than 255 of them -- the wide one (u32 bins, ragged
        rows).  Returns (thr [F][cap] padded with FLT_MAX, thr_size [F]); cap is 256 on
        the u8 path.  wide=True / False forces the path."""
        self.wide = False
        if wide is not True and nthresholds <= 255:
            thr = np.empty((self.F, QR_MAX_BINS), np.float32)
            ts = np.empty(self.F, np.uint32)
            rc = self.L.qr_bins_build(self.h, nthresholds, _ptr(thr), _ptr(ts))
            if rc == 0:
                return thr, ts
            if rc != 5 or nthresholds != 0 or wide is False:   # QR_ERR_UNSUPPORTED: too many distinct values
                self._ck(rc)
        elif wide is False:
            raise QrError("more than 255 thresholds need the wide path")
        cells, cap = C.c_size_t(), C.c_size_t()
        self._ck(self.L.qr_bins_build_wide(self.h, nthresholds, C.byref(cells), C.byref(cap)))
        self.wide = True
        return self.thresholds()

    def build_bins_wide_with(self, thr, ts):
        """Wide bins for given thresholds: thr [F][cap] padded (or the ragged concatenation), ts [F]
        -- on a document-sharded context the thresholds of the WHOLE set."""
        ts = np.ascontiguousarray(ts, np.uint32)
        thr = np.asarray(thr, np.float32)
        flat = np.concatenate([thr[f, :ts[f]] for f in range(self.F)]) if thr.ndim == 2 else thr
        flat = np.ascontiguousarray(flat, np.float32)
        cells, cap = C.c_size_t(), C.c_size_t()
        self._ck(self.L.qr_bins_build_wide_with(self.h, _ptr(flat), _ptr(ts), C.byref(cells), C.byref(cap)))
        self.wide = True
        return cells.value, cap.value

    def thresholds(self):
        """(thr [F][cap] padded with FLT_MAX, thr_size [F]) of a binned context."""
        ts = np.empty(self.F, np.uint32)
        self._ck(self.L.qr_thresholds_read(self.h, None, _ptr(ts)))
        flat = np.empty(int(ts.sum()), np.float32)
        self._ck(self.L.qr_thresholds_read(self.h, _ptr(flat), _ptr(ts)))
        cap = max(int(ts.max()), QR_MAX_BINS)
        thr = np.full((self.F, cap), np.finfo(np.float32).max, np.float32)
        o = 0
        for f in range(self.F):
            thr[f, :ts[f]] = flat[o:o + ts[f]]
            o += int(ts[f])
        return thr, ts

    def read_bins_u32(self):
        out = np.empty((self.N, self.F), np.uint32)
        self._ck(self.L.qr_bins_read_u32(self.h, _ptr(out)))
        return out

    def node_hist_ragged(self, node):
        """(sum, count) of a node as lists of per-feature rows (any context)."""
        ts = np.empty(self.F, np.uint32)
        self._ck(self.L.qr_thresholds_read(self.h, None, _ptr(ts)))
        n = int(ts.sum())
        s, c = np.zeros(n, np.float64), np.zeros(n, np.uint64)
        self._ck(self.L.qr_node_hist_read_ragged(self.h, node, _ptr(s), _ptr(c)))
        off = np.concatenate([[0], np.cumsum(ts)]).astype(np.int64)
        return [s[off[f]:off[f + 1]] for f in range(self.F)], [c[off[f]:off[f + 1]] for f in range(self.F)]

    def bins_stats_wide(self, limit):
        """Column statistics for more than 255 thresholds per feature (see qr_bins_stats_wide)."""
        vals = np.zeros((self.F, limit), np.uint32)
        cnt = np.zeros(self.F, np.uint32)
        mm = np.zeros((self.F, 2), np.uint32)
        self._ck(self.L.qr_bins_stats_wide(self.h, limit, _ptr(vals), _ptr(cnt), _ptr(mm)))
        return vals, cnt, mm

    def bins_stats(self, nthresholds):
        """Column statistics of this rank's documents (see qr_bins_stats)."""
        limit = nthresholds + 1 if nthresholds else 256
        vals = np.zeros((self.F, limit + 1), np.uint32)
        cnt = np.zeros(self.F, np.uint32)
        mm = np.zeros((self.F, 2), np.uint32)
        self._ck(self.L.qr_bins_stats(self.h, nthresholds, _ptr(vals), _ptr(cnt), _ptr(mm)))
        return vals, cnt, mm

    def build_bins_with(self, thr, ts):
        thr = np.ascontiguousarray(thr, np.float32)
        ts = np.ascontiguousarray(ts, np.uint32)
        assert thr.shape == (self.F, QR_MAX_BINS) and ts.shape == (self.F,)
        self._ck(self.L.qr_bins_build_with(self.h, _ptr(thr), _ptr(ts)))

    def read_bins(self):
        out = np.empty((self.N, self.F), np.uint8)
        self._ck(self.L.qr_bins_read(self.h, _ptr(out)))
        return out

    def verify_bins(self):
        """(cells of the block rows, cells of the feature-major copy) that do not hold what the binning
        computes from the raw rows: the verify-after-write qr_bins_build runs itself (round 6)."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self._ck(self.L.qr_bins_verify(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def debug_sample_key_mask(self, mask):
        """Test aid: the --subsample keys ANDed with `mask` (equal keys: ties go by ascending document)."""
        self._ck(self.L.qr_debug_sample_key_mask(self.h, mask))

    def debug_clobber_bins(self, which, first_doc, ndocs):
        self._ck(self.L.qr_debug_bins_clobber(self.h, which, first_doc, ndocs))

    def read_bins_fm(self):
        """The same from the feature-major copy the partition and the leaf walk read (debug)."""
        out = np.empty((self.N, self.F), np.uint8)
        self._ck(self.L.qr_bins_read_fm(self.h, _ptr(out)))
        return out

    # -- state --------------------------------------------------------------
    def reset_scores(self):
        self._ck(self.L.qr_scores_reset(self.h))

    def set_scores(self, s):
        s = np.ascontiguousarray(s, np.float64)
        self._ck(self.L.qr_scores_set(self.h, _ptr(s)))

    def get_scores(self):
        s = np.empty(self.N, np.float64)
        self._ck(self.L.qr_scores_get(self.h, _ptr(s)))
        return s

    def set_valid_scores(self, s):
        s = np.ascontiguousarray(s, np.float64)
        assert len(s) == self.vN
        self._ck(self.L.qr_valid_scores_set(self.h, _ptr(s)))

    def get_valid_scores(self):
        s = np.empty(self.vN, np.float64)
        self._ck(self.L.qr_valid_scores_get(self.h, _ptr(s)))
        return s

    def get_pseudo(self):
        l = np.empty(self.N, np.float64)
        w = np.empty(self.N, np.float64)
        self._ck(self.L.qr_pseudo_get(self.h, _ptr(l), _ptr(w)))
        return l, w

    def set_pseudo(self, lam, w=None):
        lam = np.ascontiguousarray(lam, np.float64)
        w = None if w is None else np.ascontiguousarray(w, np.float64)
        self._ck(self.L.qr_pseudo_set(self.h, _ptr(lam), _ptr(w)))

    # -- hot path -----------------------------------------------------------
    def compute_lambdas(self, metric="NDCG", cutoff=10):
        self._ck(self.L.qr_lambda_compute(self.h, METRICS[metric], cutoff))

    def compute_residuals(self):
        self._ck(self.L.qr_residual_compute(self.h))

    def metric_eval(self, which=0, metric="NDCG", cutoff=10):
        out = C.c_double()
        self._ck(self.L.qr_metric_eval(self.h, which, METRICS[metric], cutoff, C.byref(out)))
        return out.value

    def metric_last(self):
        out = C.c_double()
        self._ck(self.L.qr_metric_last(self.h, C.byref(out)))
        return out.value

    def fit_tree(self, nleaves=10, minls=1, newton=True, read=True):
        """read=False: enqueue only; fetch the records later with tree_nodes()."""
        if not read:
            self._ck(self.L.qr_tree_fit(self.h, nleaves, minls, int(newton), None, None))
            return None
        nodes = np.zeros(2 * nleaves + 1, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_fit(self.h, nleaves, minls, int(newton), _ptr(nodes), C.byref(n)))
        return nodes[:n.value].copy()

    def tree_nodes(self):
        """Records of the last fitted tree (waits for that tree only)."""
        nodes = np.zeros(1024, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_nodes(self.h, _ptr(nodes), C.byref(n)))
        return nodes[:n.value].copy()

    def set_subsample(self, subsample, seed=0, first_doc=None):
        """first_doc: document-sharded contexts -- the global index of the rank's first document"""
        if first_doc is None:
            self._ck(self.L.qr_subsample_set(self.h, float(subsample), int(seed)))
        else:
            self._ck(self.L.qr_subsample_set_doc(self.h, float(subsample), int(seed), int(first_doc)))

    def set_max_features(self, max_features, seed=0):
        self._ck(self.L.qr_tree_set_max_features(self.h, float(max_features), int(seed)))

    def fit_oblivious(self, depth=3, minls=1, newton=True, read=True):
        """read=False: enqueue only; fetch the records later with tree_nodes()."""
        if not read:
            self._ck(self.L.qr_oblivious_fit(self.h, depth, minls, int(newton), None, None))
            return None
        nodes = np.zeros((1 << (depth + 1)) - 1, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_oblivious_fit(self.h, depth, minls, int(newton), _ptr(nodes),
                                         C.byref(n)))
        return nodes[:n.value].copy()

    def update_scores(self, shrinkage):
        self._ck(self.L.qr_scores_update(self.h, float(shrinkage)))

    # -- multi-GPU phases -----------------------------------------------------
    def tree_begin(self, nleaves, minls=1):
        self._ck(self.L.qr_tree_begin(self.h, nleaves, minls))

    def tree_decide(self):
        self._ck(self.L.qr_tree_decide(self.h))

    def tree_apply(self):
        self._ck(self.L.qr_tree_apply(self.h))

    def tree_end(self, nleaves, newton=True):
        nodes = np.zeros(2 * nleaves + 1, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_end(self.h, int(newton), _ptr(nodes), C.byref(n)))
        return nodes[:n.value].copy()

    # feature-sharded oblivious trees, phase by phase
    def obl_begin(self, depth, minls=1):
        self._ck(self.L.qr_obl_begin(self.h, depth, minls))

    def obl_propose(self, level):
        self._ck(self.L.qr_obl_propose(self.h, level))

    def obl_mark(self, level):
        self._ck(self.L.qr_obl_mark(self.h, level))

    def obl_apply(self, level):
        self._ck(self.L.qr_obl_apply(self.h, level))

    def obl_end(self, depth, newton=True, read=True):
        if not read:
            self._ck(self.L.qr_tree_end(self.h, int(newton), None, None))
            return None
        nodes = np.zeros((1 << (depth + 1)) - 1, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_end(self.h, int(newton), _ptr(nodes), C.byref(n)))
        return nodes[:n.value].copy()

    def obl_exchange_buffers(self):
        a, b, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rb, mb = C.c_size_t(), C.c_size_t()
        self._ck(self.L.qr_obl_exchange_buffers(self.h, C.byref(a), C.byref(b), C.byref(rb),
                                                C.byref(m), C.byref(mb)))
        return dict(recs_local=a.value, recs_all=b.value, rec_bytes=rb.value,
                    mask=m.value, mask_bytes=mb.value)

    def obl_level_exchange(self, level):
        """document-sharded: (device pointer, int64 count) of the level's cells to sum"""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.L.qr_obl_level_exchange(self.h, level, C.byref(p), C.byref(n)))
        return p.value, n.value

    def lambda_finish(self):
        self._ck(self.L.qr_lambda_finish(self.h))

    # -- document-sharded trees, up to two splits per exchange (include/qr_hip.h)
    def tree_batch_supported(self, nleaves):
        return bool(self.L.qr_tree_batch_supported(self.h, nleaves))

    def tree_batch_begin(self, nleaves, minls):
        """root histogram enqueued; returns the number of steps to enqueue (a guess)"""
        n = C.c_size_t()
        self._ck(self.L.qr_tree_batch_begin(self.h, nleaves, minls, C.byref(n)))
        return n.value

    def tree_batch_root(self):
        self._ck(self.L.qr_tree_batch_root(self.h))

    def tree_batch_apply(self):
        self._ck(self.L.qr_tree_batch_apply(self.h))

    def tree_batch_decide(self, last):
        self._ck(self.L.qr_tree_batch_decide(self.h, int(bool(last))))

    def tree_batch_settle(self):
        """(incomplete, steps the tree has used): waits for the last control step"""
        inc, n = C.c_int(), C.c_size_t()
        self._ck(self.L.qr_tree_batch_settle(self.h, C.byref(inc), C.byref(n)))
        return bool(inc.value), n.value

    def tree_batch_exchange(self):
        """(device pointer, int64 count) of the batch's cells to sum"""
        p, n = C.c_void_p(), C.c_size_t()
        self._ck(self.L.qr_tree_batch_exchange(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def tree_end_local(self, newton=True):
        """document-sharded: local leaf sums only; all-reduce, then tree_leaves_finish"""
        self._ck(self.L.qr_tree_end(self.h, int(newton), None, None))

    def tree_leaves_finish(self, nleaves, newton=True, read=True):
        if not read:
            self._ck(self.L.qr_tree_leaves_finish(self.h, int(newton), None, None))
            return None
        nodes = np.zeros(2 * nleaves + 1, NODE_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_leaves_finish(self.h, int(newton), _ptr(nodes), C.byref(n)))
        return nodes[:n.value].copy()

    def doc_exchange_buffers(self):
        h, s, l = C.c_void_p(), C.c_void_p(), C.c_void_p()
        hn, sn, ln = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._ck(self.L.qr_doc_exchange_buffers(self.h, C.byref(h), C.byref(hn), C.byref(s),
                                                C.byref(sn), C.byref(l), C.byref(ln)))
        return dict(hist=h.value, hist_n=hn.value, scal=s.value, scal_n=sn.value,
                    leaf=l.value, leaf_n=ln.value)

    def exchange_buffers(self):
        a, b, m = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rb, mb = C.c_size_t(), C.c_size_t()
        self._ck(self.L.qr_exchange_buffers(self.h, C.byref(a), C.byref(b), C.byref(rb),
                                            C.byref(m), C.byref(mb)))
        return dict(recs_local=a.value, recs_all=b.value, rec_bytes=rb.value,
                    mask=m.value, mask_bytes=mb.value)

    def synchronize(self):
        """Drains the stream.  On a document-sharded context a tree that ended behind a guessed
        step count may still be incomplete afterwards: see tree_pending()."""
        self._ck(self.L.qr_synchronize(self.h))

    def tree_pending(self):
        p = C.c_int()
        self._ck(self.L.qr_tree_pending(self.h, C.byref(p)))
        return bool(p.value)

    # -- read-backs ---------------------------------------------------------
    def node_hist(self, node):
        s = np.zeros((self.F, QR_MAX_BINS), np.float64)
        c = np.zeros((self.F, QR_MAX_BINS), np.uint64)
        self._ck(self.L.qr_node_hist_read(self.h, node, _ptr(s), _ptr(c)))
        return s, c

    def node_samples(self, node):
        n = C.c_size_t()
        self._ck(self.L.qr_node_samples_read(self.h, node, None, C.byref(n)))
        ids = np.zeros(n.value, np.uint32)
        self._ck(self.L.qr_node_samples_read(self.h, node, _ptr(ids), C.byref(n)))
        return ids

    def split_log(self):
        out = np.zeros(1024, SPLIT_DTYPE)
        n = C.c_size_t()
        self._ck(self.L.qr_tree_split_log(self.h, _ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def metric_per_query(self):
        out = np.empty(self.Q, np.float64)
        self._ck(self.L.qr_metric_per_query(self.h, _ptr(out)))
        return out

    def ranks(self):
        out = np.empty(self.N, np.uint32)
        self._ck(self.L.qr_ranks_read(self.h, _ptr(out)))
        return out

    # -- inference ------------------------------------------------------------
    def upload_ensemble(self, nodes, weights, depth_order=False):
        """depth_order: walk and SUM the trees in ascending depth (faster on leaf-wise trees; the
        f64 sum in another order than the reference's: equal to rounding, not bit for bit)"""
        self._ck(self.L.qr_ensemble_set_depth_order(self.h, int(bool(depth_order))))
        nodes = np.ascontiguousarray(nodes)
        assert nodes.dtype == NODE_DTYPE and nodes.ndim == 2
        weights = np.ascontiguousarray(weights, np.float64)
        self._ck(self.L.qr_ensemble_upload(self.h, _ptr(nodes), nodes.shape[0], nodes.shape[1],
                                           _ptr(weights)))

    def score(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.shape[0], np.float64)
        ms = C.c_float()
        self._ck(self.L.qr_ensemble_score(self.h, _ptr(x), x.shape[0], x.shape[1], _ptr(out),
                                          C.byref(ms)))
        return out, ms.value

    def partial_scores(self, x, ntrees, ignore_weights=False):
        """Ensemble::partial_scores_instance for every row: f64 [N][ntrees]."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((x.shape[0], ntrees), np.float64)
        self._ck(self.L.qr_ensemble_partial_scores(self.h, _ptr(x), x.shape[0], x.shape[1],
                                                   int(ignore_weights), _ptr(out)))
        return out

    def upload_oblivious(self, feat, thr, leaves, weights, depths=None):
        feat = np.ascontiguousarray(feat, np.uint32)
        thr = np.ascontiguousarray(thr, np.float32)
        leaves = np.ascontiguousarray(leaves, np.float64)
        weights = np.ascontiguousarray(weights, np.float32)
        depths = None if depths is None else np.ascontiguousarray(depths, np.uint32)
        T, D = feat.shape
        assert leaves.shape == (T, 1 << D)
        self._ck(self.L.qr_oblivious_upload(self.h, _ptr(feat), _ptr(thr), _ptr(leaves),
                                            _ptr(weights), _ptr(depths), T, D))

    def score_oblivious(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(x.shape[0], np.float64)
        ms = C.c_float()
        self._ck(self.L.qr_oblivious_score(self.h, _ptr(x), x.shape[0], x.shape[1], _ptr(out),
                                           C.byref(ms)))
        return out, ms.value

    # -- instrumentation ------------------------------------------------------
    def prof_enable(self, on=True, children=False, every=1, lambdas=False):
        """HIP events on the root histogram launches (every `every`-th one), optionally on the
        child launches too -- or (`lambdas`, instead of the children) on the lambda pass's launch;
        both are read with prof_get_child()."""
        self._ck(self.L.qr_prof_enable(self.h, int(bool(on)) | (2 if children else 0) | (4 if lambdas else 0)
                                       | ((int(every) & 0xff) << 8)))

    def prof_get_child(self):
        n, ms = C.c_uint64(), C.c_double()
        self._ck(self.L.qr_prof_get_child(self.h, C.byref(n), C.byref(ms)))
        return dict(launches=n.value, total_ms=ms.value)

    def prof_reset(self):
        self._ck(self.L.qr_prof_reset(self.h))

    def prof_get(self):
        n, ms, b = C.c_uint64(), C.c_double(), C.c_double()
        self._ck(self.L.qr_prof_get(self.h, C.byref(n), C.byref(ms), C.byref(b)))
        return dict(launches=n.value, total_ms=ms.value, alg_bytes=b.value)

    def prof_lds_atomic(self):
        """qr_prof_lds_atomic: the bare ds_add_u64 rate of a CU, measured in this process."""
        cyc, ghz, ns, wi = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self._ck(self.L.qr_prof_lds_atomic(self.h, C.byref(cyc), C.byref(ghz), C.byref(ns), C.byref(wi)))
        return dict(cycles_per_instr=cyc.value, shader_ghz=ghz.value, ns_per_instr=ns.value,
                    root_wave_instr_per_cu=wi.value)
