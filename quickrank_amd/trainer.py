"""Host mirror of Mart / LambdaMart (mart.cc:208-416, lambdamart.cc:47-60).

The boosting loop keeps the reference's phase order -- compute_pseudoresponses
-> root histogram + fit_regressor_on_gradient -> ensemble push ->
update_modelscores -> evaluate_dataset (+ validation, best model, early stop,
rollback) -- with every phase running on the device through the C-ABI
(include/qr_hip.h).  Only tree nodes and metric scalars return to the host.
"""
import time

import numpy as np

from ._capi import Context, NODE_DTYPE

ALGOS = ("MART", "LAMBDAMART", "OBVMART", "OBVLAMBDAMART")


class Ensemble:
    """ensemble.h:82-98: (root, weight) pairs; nodes kept as flat records."""

    def __init__(self, max_nodes):
        self.max_nodes = max_nodes
        self.trees = []
        self.weights = []

    def push(self, nodes, weight):
        t = np.zeros(self.max_nodes, NODE_DTYPE)
        t["feature"] = -1
        t[:len(nodes)] = nodes
        self.trees.append(t)
        self.weights.append(weight)

    def pop(self):
        self.trees.pop()
        self.weights.pop()

    def __len__(self):
        return len(self.trees)

    def arrays(self):
        return np.stack(self.trees), np.asarray(self.weights, np.float64)


class Mart:
    """GBRT on the device; `algo` selects MART or LAMBDAMART pseudo-responses."""

    def __init__(self, algo="LAMBDAMART", ntrees=1000, shrinkage=0.1, nthresholds=0,
                 nleaves=10, minls=1, esr=100, metric="NDCG", cutoff=10, device=0,
                 ctx=None, dist=None, depth=3, subsample=1.0, max_features=1.0, seed=0):
        if algo not in ALGOS:
            raise ValueError(f"unsupported algorithm {algo}")
        self.algo, self.ntrees, self.shrinkage = algo, ntrees, shrinkage
        self.nthresholds, self.nleaves, self.minls, self.esr = nthresholds, nleaves, minls, esr
        self.metric, self.cutoff = metric, cutoff
        self.ctx = ctx if ctx is not None else Context(device)
        self.dist = dist
        self.depth = depth
        # --subsample / --max-features (mart.cc:287-329, rt.cc:222-243), reproducible streams
        self.subsample, self.max_features, self.seed = subsample, max_features, seed
        self.oblivious = algo.startswith("OBV")  # obliviousmart.cc / obliviouslambdamart.cc
        self.ensemble = Ensemble((1 << (depth + 1)) - 1 if self.oblivious else 2 * nleaves + 1)
        self.thr = self.thr_size = None
        self.train_metric, self.valid_metric, self.iter_seconds = [], [], []
        self.best_model = 0

    # Mart::init (mart.cc:117-176)
    def init(self, x, labels, qoff, valid=None):
        self.ctx.upload(x, labels, qoff)
        if valid is not None:
            self.ctx.upload_valid(*valid)
        self.thr, self.thr_size = self.ctx.build_bins(self.nthresholds)
        self.ctx.reset_scores()
        if self.subsample != 1.0:
            self.ctx.set_subsample(self.subsample, self.seed)
        if self.max_features != 1.0:
            self.ctx.set_max_features(self.max_features, self.seed)

    def _fit_tree(self, newton):
        if self.oblivious:
            if self.dist is not None:     # sharded ranks: level by level with the collectives in between
                return self.dist.fit_oblivious(self.ctx, self.depth, self.minls, newton)
            return self.ctx.fit_oblivious(self.depth, self.minls, newton)
        if self.dist is not None:
            return self.dist.fit_tree(self.ctx, self.nleaves, self.minls, newton)
        return self.ctx.fit_tree(self.nleaves, self.minls, newton)

    # Mart::learn main loop (mart.cc:307-383) + rollback (:390-395)
    def learn(self, x, labels, qoff, valid=None, verbose=False, eval_every=1, on_tree=None):
        """on_tree(self, m, nodes): called behind every tree fit, before its scores are added (test tools
        look at the device's state of THAT tree there: tests/tools/fuzz_parity.py FUZZ_LOCKSTEP=1)."""
        self.init(x, labels, qoff, valid)
        lam = self.algo.endswith("LAMBDAMART")
        best_valid = best_train = -np.inf
        self.best_model = 0
        # The lambda kernel ranks every query anyway, so the training metric of the
        # scores it ranks (= the previous iteration's, mart.cc:347) falls out of it:
        # without a validation set the bookkeeping of mart.cc:369-375 simply runs
        # one iteration late (same values, same best model at the end).
        fused = lam and valid is None and self.subsample == 1.0  # a sampled ranking is not the metric's
        for m in range(self.ntrees):
            if valid is not None and self.esr and m > self.best_model + self.esr:
                break
            t0 = time.perf_counter()
            if lam:
                self.ctx.compute_lambdas(self.metric, self.cutoff)
            else:
                self.ctx.compute_residuals()
            nodes = self._fit_tree(newton=lam)
            self.ensemble.push(nodes, self.shrinkage)
            if on_tree is not None:
                on_tree(self, m, nodes)
            self.ctx.update_scores(self.shrinkage)
            if fused:
                # (read AFTER the tree is enqueued: the scalars of the lambda pass are finished
                # by workgroups riding in the tree's root scan launch, csrc/qr_prep.h)
                if m > 0:
                    mt = self.ctx.metric_last()
                    self.train_metric.append(mt)
                    if mt > best_train:
                        best_train = mt
                        self.best_model = m - 1
                self.iter_seconds.append(time.perf_counter() - t0)
                continue
            mt = self.ctx.metric_eval(0, self.metric, self.cutoff) if eval_every else 0.0
            self.train_metric.append(mt)
            if valid is not None:
                mv = self.ctx.metric_eval(1, self.metric, self.cutoff)
                self.valid_metric.append(mv)
                if mv > best_valid:
                    best_train, best_valid = mt, mv
                    self.best_model = len(self.ensemble) - 1
            elif mt > best_train:
                best_train = mt
                self.best_model = len(self.ensemble) - 1
            self.iter_seconds.append(time.perf_counter() - t0)
            if verbose:
                print(f"{m + 1:7d} {mt:9.4f}" + (f" {self.valid_metric[-1]:9.4f}" if valid else ""))
        if fused and len(self.ensemble):
            mt = self.ctx.metric_eval(0, self.metric, self.cutoff)
            self.train_metric.append(mt)
            if mt > best_train:
                best_train = mt
                self.best_model = len(self.ensemble) - 1
        if valid is not None:
            while len(self.ensemble) > self.best_model + 1:
                self.ensemble.pop()
        return self

    # LTR_Algorithm::save (ltr_algorithm.cc:54-66): the XML model `quicklearn --model-out` writes
    def save(self, path):
        from . import io
        nodes, w = self.ensemble.arrays() if len(self.ensemble) else (np.zeros((0, self.ensemble.max_nodes), NODE_DTYPE),
                                                                      np.zeros(0))
        io.save_model(path, self.algo, nodes, w, self.ntrees, self.shrinkage, self.nthresholds, self.nleaves,
                      self.minls, self.esr, self.depth)

    # LTR_Algorithm::load_model_from_file (ltr_algorithm.cc:98-131): a model file -> a scorer
    @classmethod
    def load_model_from_file(cls, path, device=0, ctx=None):
        from . import io
        m = io.load_model(path)
        if m is None:
            return None
        self = cls(algo=m["algo"], ntrees=m["ntrees"], shrinkage=m["shrinkage"], nthresholds=m["nthresholds"],
                   nleaves=m["nleaves"], minls=m["minls"], esr=m["esr"], depth=m["depth"], device=device, ctx=ctx)
        self.ensemble = Ensemble(m["nodes"].shape[1] if len(m["nodes"]) else self.ensemble.max_nodes)
        for t, w in zip(m["nodes"], m["weights"]):
            self.ensemble.push(t, float(w))
        return self

    # LTR_Algorithm::score_dataset (ltr_algorithm.cc:44-52)
    def score_dataset(self, x):
        nodes, w = self.ensemble.arrays()
        self.ctx.upload_ensemble(nodes, w)
        return self.ctx.score(x)[0]
