"""ctypes bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker or the
timed CPU baseline.  The product package ``quickrank_amd`` never imports it.

``liboracle.so``   = oracle/qr_oracle.c (C restatement; header cites reference file:line)
``_ref/libqr_ref.so`` = the reference's own translation units (partial build,
                     no stand-in headers) behind oracle/ref_harness.cc.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
sz = C.c_size_t


class Split(C.Structure):
    _fields_ = [("score", C.c_double), ("feature", C.c_uint64),
                ("thr_id", C.c_uint64), ("lcount", C.c_uint64),
                ("rcount", C.c_uint64)]


NODE_DTYPE = np.dtype([("feature", np.int32), ("thr_id", np.int32),
                       ("threshold", np.float32), ("left", np.int32),
                       ("right", np.int32), ("value", np.float64),
                       ("deviance", np.float64), ("nsamples", np.uint64)],
                      align=True)
SPLIT_DTYPE = np.dtype([("score", np.float64), ("feature", np.uint64),
                        ("thr_id", np.uint64), ("lcount", np.uint64),
                        ("rcount", np.uint64)], align=True)


class TrainData(C.Structure):
    _fields_ = [("N", sz), ("F", sz), ("cap", sz), ("colmajor", C.c_void_p),
                ("stmap", C.c_void_p), ("thr", C.c_void_p),
                ("thr_size", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("algo", C.c_int), ("ntrees", sz), ("shrinkage", C.c_double),
                ("nthresholds", sz), ("nleaves", sz), ("depth", sz),
                ("minls", C.c_uint64), ("esr", sz), ("metric", C.c_int),
                ("cutoff", sz)]


class Model(C.Structure):
    _fields_ = [("ntrees", sz), ("ntrees_built", sz), ("max_nodes", sz),
                ("nodes", C.c_void_p), ("nnodes", C.POINTER(C.c_uint64)),
                ("train_metric", C.POINTER(C.c_double)),
                ("valid_metric", C.POINTER(C.c_double)),
                ("train_scores", C.POINTER(C.c_double)),
                ("thr", C.POINTER(C.c_float)),
                ("thr_size", C.POINTER(C.c_uint64)), ("cap", sz),
                ("best_model", sz), ("iter_seconds", C.POINTER(C.c_double))]


REFERENCE_PRESENT = os.path.isdir("/root/reference/src")


def build(ref=None):
    """Compile liboracle.so, and _ref/libqr_ref.so where /root/reference exists (the build
    container; `ref=None` = exactly there).  _ref never travels: .gpurunignore lists it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref is None:
        ref = REFERENCE_PRESENT
    if ref and REFERENCE_PRESENT:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def available_cores():
    """Cores this process may really use (affinity mask and cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # libgomp sizes its pool from the visible cores; on a many-core host with a
    # small cgroup quota that oversubscribes badly.  Cap before the first region.
    os.environ.setdefault("OMP_NUM_THREADS", str(min(8, available_cores())))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build(ref=False)
    L = C.CDLL(path)
    L.qro_rank_by_score.argtypes = [f64p, sz, u64p]
    L.qro_heapsort_by_score.argtypes = [f64p, sz, u64p]
    L.qro_sort_labels_desc_int.argtypes = [f32p, sz]
    L.qro_dcg.argtypes = [f32p, sz, sz]
    L.qro_dcg.restype = C.c_double
    L.qro_idcg.argtypes = [f32p, sz, sz]
    L.qro_idcg.restype = C.c_double
    for fn in (L.qro_dcg_query, L.qro_ndcg_query):
        fn.argtypes = [f32p, f64p, sz, sz]
        fn.restype = C.c_double
    L.qro_eval_dataset.argtypes = [C.c_int, f32p, f64p, u64p, sz, sz]
    L.qro_eval_dataset.restype = C.c_double
    L.qro_jacobian.argtypes = [C.c_int, f32p, sz, sz, f64p]
    L.qro_lambdas.argtypes = [C.c_int, f32p, f64p, u64p, sz, sz, f64p, f64p]
    L.qro_residuals.argtypes = [f32p, f64p, sz, f64p]
    L.qro_argsort_f32.argtypes = [f32p, sz, u64p]
    L.qro_thresholds.argtypes = [f32p, sz, sz, sz, f32p, u64p, sz]
    L.qro_binmap.argtypes = [f32p, sz, sz, f32p, u64p, sz, u32p, u64p]
    L.qro_hist_build.argtypes = [u32p, sz, sz, u64p, sz, f64p, C.c_void_p, sz,
                                 f64p, u64p]
    L.qro_hist_build.restype = C.c_double
    L.qro_hist_subtract.argtypes = [sz, u64p, sz, f64p, u64p, f64p, u64p, f64p,
                                    u64p]
    L.qro_split_find.argtypes = [sz, sz, u64p, sz, f64p, u64p, C.c_uint64,
                                 C.POINTER(Split)]
    for fn in (L.qro_tree_fit, L.qro_oblivious_fit):
        fn.argtypes = [C.POINTER(TrainData), f64p, sz, C.c_uint64, C.c_void_p,
                       i32p, i32p, C.POINTER(sz), C.c_void_p, C.POINTER(sz)]
        fn.restype = sz
    L.qro_update_output.argtypes = [C.c_void_p, i32p, sz, i32p, sz, f64p,
                                    C.c_void_p]
    L.qro_update_scores.argtypes = [C.c_void_p, f32p, sz, sz, C.c_int,
                                    C.c_double, f64p]
    L.qro_train.argtypes = [C.POINTER(Params), f32p, f32p, u64p, sz, sz, sz,
                            C.c_void_p, C.c_void_p, C.c_void_p, sz, sz,
                            C.POINTER(Model)]
    L.qro_train.restype = C.c_int
    L.qro_model_free.argtypes = [C.POINTER(Model)]
    L.qro_ensemble_score.argtypes = [C.c_void_p, u64p, sz, sz, f64p, f32p, sz,
                                     sz, f64p]
    L.qro_oblivious_score.argtypes = [u32p, f32p, f64p, f32p, sz, sz, f32p, sz,
                                      sz, f64p]
    L.qro_set_threads.argtypes = [C.c_int]
    L.qro_self_check.argtypes = [C.c_char_p, sz]
    L.qro_self_check.restype = C.c_int
    L.qro_heap_trace.argtypes = [f64p, i32p, sz, sz, i32p, u64p]
    L.qro_sym_index.argtypes = [sz, u64p]
    _LIB = L
    return L


def ref():
    """The partial reference build, or None when it is not present."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(_HERE, "_ref", "libqr_ref.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.ref_rank_by_score.argtypes = [f64p, sz, u64p]
    R.ref_heapsort_by_score.argtypes = [f64p, sz, u64p]
    R.ref_sort_labels_desc_int.argtypes = [f32p, sz]
    R.ref_eval_query.argtypes = [C.c_int, f32p, f64p, sz, sz]
    R.ref_eval_query.restype = C.c_double
    R.ref_eval_dataset.argtypes = [C.c_int, f32p, f64p, u64p, sz, sz, C.c_int]
    R.ref_eval_dataset.restype = C.c_double
    R.ref_jacobian.argtypes = [C.c_int, f32p, f64p, sz, sz, f64p, f32p, u64p]
    R.ref_argsort_f32.argtypes = [f32p, sz, u64p]
    R.ref_histograms.argtypes = [f32p, sz, sz, f32p, u64p, sz, f64p, u64p, sz,
                                 C.c_int, u32p, u64p, f64p, u64p,
                                 C.POINTER(C.c_double), f64p, u64p,
                                 C.POINTER(C.c_double), f64p, u64p,
                                 C.POINTER(C.c_double)]
    R.ref_svml_read.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.c_void_p,
                                C.c_void_p, C.c_void_p]
    R.ref_heap_trace.argtypes = [f64p, i32p, sz, sz, i32p, u64p]
    R.ref_sym_index.argtypes = [sz, u64p]
    R.ref_sorted_labels.argtypes = [f32p, f64p, sz, sz, f32p]
    R.ref_dataset_layout.argtypes = [f32p, f32p, u32p, sz, sz, u64p, C.c_void_p, C.c_void_p]
    R.ref_dataset_layout.restype = sz
    R.ref_svml_write.argtypes = [C.c_char_p, f32p, f32p, u64p, sz, sz]
    _REF = R
    return R


# ----------------------------------------------------------------------------
# numpy-level helpers
# ----------------------------------------------------------------------------
def rank_by_score(scores):
    s = np.ascontiguousarray(scores, np.float64)
    idx = np.empty(len(s), np.uint64)
    lib().qro_rank_by_score(s, len(s), idx)
    return idx


def lambdas(labels, scores, qoff, cutoff=10, metric=1):
    labels = np.ascontiguousarray(labels, np.float32)
    scores = np.ascontiguousarray(scores, np.float64)
    qoff = np.ascontiguousarray(qoff, np.uint64)
    lam = np.zeros(len(labels), np.float64)
    w = np.zeros(len(labels), np.float64)
    lib().qro_lambdas(metric, labels, scores, qoff, len(qoff) - 1, cutoff, lam, w)
    return lam, w


def eval_dataset(labels, scores, qoff, cutoff=10, metric=1):
    labels = np.ascontiguousarray(labels, np.float32)
    scores = np.ascontiguousarray(scores, np.float64)
    qoff = np.ascontiguousarray(qoff, np.uint64)
    return lib().qro_eval_dataset(metric, labels, scores, qoff, len(qoff) - 1, cutoff)


def thresholds(colmajor, nthresholds):
    """colmajor: [F][N] f32.  Returns (thr [F][cap], thr_size [F])."""
    x = np.ascontiguousarray(colmajor, np.float32)
    F, N = x.shape
    cap = nthresholds + 1 if nthresholds else N + 1
    thr = np.empty((F, cap), np.float32)
    ts = np.empty(F, np.uint64)
    lib().qro_thresholds(x, N, F, nthresholds, thr, ts, cap)
    return thr, ts


def binmap(colmajor, thr, thr_size):
    x = np.ascontiguousarray(colmajor, np.float32)
    F, N = x.shape
    cap = thr.shape[1]
    stmap = np.zeros((F, N), np.uint32)
    count0 = np.zeros((F, cap), np.uint64)
    lib().qro_binmap(x, N, F, thr, thr_size, cap, stmap, count0)
    return stmap, count0


def hist_build(stmap, thr_size, cap, labels, sampleids=None):
    F, N = stmap.shape
    s = np.zeros((F, cap), np.float64)
    c = np.zeros((F, cap), np.uint64)
    labels = np.ascontiguousarray(labels, np.float64)
    if sampleids is None:
        ss = lib().qro_hist_build(stmap, N, F, thr_size, cap, labels, None, N, s, c)
    else:
        ids = np.ascontiguousarray(sampleids, np.uint64)
        ss = lib().qro_hist_build(stmap, N, F, thr_size, cap, labels,
                                  ids.ctypes.data, len(ids), s, c)
    return s, c, ss


def split_find(sum_, count, thr_size, minls=1, f0=0, f1=None):
    F, cap = sum_.shape
    out = Split()
    lib().qro_split_find(f0, F if f1 is None else f1, thr_size, cap, sum_, count,
                         minls, C.byref(out))
    return out


class Trainer:
    """Holds column-major data + bins for tree-level oracle calls."""

    def __init__(self, rowmajor, nthresholds):
        x = np.ascontiguousarray(rowmajor, np.float32)
        self.N, self.F = x.shape
        self.col = np.ascontiguousarray(x.T)
        self.thr, self.thr_size = thresholds(self.col, nthresholds)
        self.cap = self.thr.shape[1]
        self.stmap, self.count0 = binmap(self.col, self.thr, self.thr_size)
        self._td = TrainData(self.N, self.F, self.cap, self.col.ctypes.data,
                             self.stmap.ctypes.data, self.thr.ctypes.data,
                             self.thr_size.ctypes.data)

    def fit_tree(self, pseudo, nleaves=10, minls=1, oblivious_depth=None):
        pseudo = np.ascontiguousarray(pseudo, np.float64)
        if oblivious_depth is None:
            maxn, maxl = 2 * nleaves + 1, nleaves + 1
        else:
            maxn, maxl = (1 << (oblivious_depth + 1)) - 1, (1 << oblivious_depth) + 1
        nodes = np.zeros(maxn, NODE_DTYPE)
        leaf_of_doc = np.zeros(self.N, np.int32)
        leaf_nodes = np.zeros(maxl, np.int32)
        log = np.zeros(maxn, SPLIT_DTYPE)
        nl, ns = sz(0), sz(0)
        if oblivious_depth is None:
            nn = lib().qro_tree_fit(C.byref(self._td), pseudo, nleaves, minls,
                                    nodes.ctypes.data, leaf_of_doc, leaf_nodes,
                                    C.byref(nl), log.ctypes.data, C.byref(ns))
        else:
            nn = lib().qro_oblivious_fit(C.byref(self._td), pseudo,
                                         oblivious_depth, minls,
                                         nodes.ctypes.data, leaf_of_doc,
                                         leaf_nodes, C.byref(nl),
                                         log.ctypes.data, C.byref(ns))
        self_check("fit_tree")
        return dict(nodes=nodes[:nn], leaf_of_doc=leaf_of_doc,
                    leaf_nodes=leaf_nodes[:nl.value], splits=log[:ns.value])

    def update_output(self, tree, pseudo, weights=None):
        pseudo = np.ascontiguousarray(pseudo, np.float64)
        w = None
        if weights is not None:
            weights = np.ascontiguousarray(weights, np.float64)
            w = weights.ctypes.data
        ln = np.ascontiguousarray(tree["leaf_nodes"], np.int32)
        lib().qro_update_output(tree["nodes"].ctypes.data, ln, len(ln),
                                tree["leaf_of_doc"], self.N, pseudo, w)
        self_check("update_output")

    def update_scores(self, tree, shrinkage, scores):
        lib().qro_update_scores(tree["nodes"].ctypes.data, self.col, self.N,
                                self.F, 1, shrinkage, scores)


class SelfCheckError(RuntimeError):
    """Two views the oracle holds of one fact disagreed during the call (qr_oracle.c, top):
    host memory changed under the run.  The result is not a verdict on anything."""


def self_check(what):
    buf = C.create_string_buffer(1024)
    n = lib().qro_self_check(buf, len(buf))
    if n:
        raise SelfCheckError(f"oracle self-check, {what}: {n} event(s); first: {buf.value.decode(errors='replace')}")


ALGOS = {"MART": 0, "LAMBDAMART": 1, "OBVMART": 2, "OBVLAMBDAMART": 3}


def train(rowmajor, labels, qoff, algo="LAMBDAMART", ntrees=10, shrinkage=0.1,
          nthresholds=0, nleaves=10, depth=3, minls=1, esr=100, metric=1,
          cutoff=10, valid=None, threads=None):
    """Mart::learn restatement.  valid = (rowmajor, labels, qoff) or None."""
    if threads:
        lib().qro_set_threads(threads)
    x = np.ascontiguousarray(rowmajor, np.float32)
    N, F = x.shape
    labels = np.ascontiguousarray(labels, np.float32)
    qoff = np.ascontiguousarray(qoff, np.uint64)
    p = Params(ALGOS[algo], ntrees, shrinkage, nthresholds, nleaves, depth,
               minls, esr, metric, cutoff)
    m = Model()
    if valid is not None:
        vx = np.ascontiguousarray(valid[0], np.float32)
        vl = np.ascontiguousarray(valid[1], np.float32)
        vq = np.ascontiguousarray(valid[2], np.uint64)
        lib().qro_train(C.byref(p), x, labels, qoff, len(qoff) - 1, N, F,
                        vx.ctypes.data, vl.ctypes.data, vq.ctypes.data,
                        len(vq) - 1, len(vl), C.byref(m))
    else:
        lib().qro_train(C.byref(p), x, labels, qoff, len(qoff) - 1, N, F, None,
                        None, None, 0, 0, C.byref(m))
    nb = m.ntrees_built
    nodes = np.ctypeslib.as_array(
        C.cast(m.nodes, C.POINTER(C.c_byte)),
        (max(nb, 1) * m.max_nodes * NODE_DTYPE.itemsize,)).view(NODE_DTYPE)
    res = dict(
        ntrees=m.ntrees, ntrees_built=nb, max_nodes=m.max_nodes,
        nodes=nodes[:nb * m.max_nodes].reshape(nb, m.max_nodes).copy(),
        nnodes=np.ctypeslib.as_array(m.nnodes, (max(nb, 1),))[:nb].copy(),
        train_metric=np.ctypeslib.as_array(m.train_metric, (max(nb, 1),))[:nb].copy(),
        valid_metric=np.ctypeslib.as_array(m.valid_metric, (max(nb, 1),))[:nb].copy(),
        train_scores=np.ctypeslib.as_array(m.train_scores, (N,)).copy(),
        iter_seconds=np.ctypeslib.as_array(m.iter_seconds, (max(nb, 1),))[:nb].copy(),
        thr=np.ctypeslib.as_array(m.thr, (F, m.cap)).copy(),
        thr_size=np.ctypeslib.as_array(m.thr_size, (F,)).copy(),
        best_model=m.best_model, shrinkage=shrinkage)
    lib().qro_model_free(C.byref(m))
    self_check("train")
    return res


def ensemble_score(model, rowmajor, ntrees=None):
    x = np.ascontiguousarray(rowmajor, np.float32)
    N, F = x.shape
    nt = model["ntrees"] if ntrees is None else ntrees
    nodes = np.ascontiguousarray(model["nodes"][:nt])
    w = np.full(nt, model["shrinkage"], np.float64)
    out = np.zeros(N, np.float64)
    lib().qro_ensemble_score(nodes.ctypes.data,
                             np.ascontiguousarray(model["nnodes"][:nt]), nt,
                             model["max_nodes"], w, x, N, F, out)
    return out
