"""Host-side stand-in for a DOCUMENT-sharded device context (TEST INFRASTRUCTURE).

Implements the document-sharded phase protocol of include/qr_hip.h (lambda
compute / finish, tree begin / decide / apply / end / leaves_finish and the three
int64 exchange buffers) on the CPU: lambdas and split search by the oracle, the
histogram in the device's exact fixed-point integers, so that the collective
sequence of quickrank_amd.dist.DocShardedTrainer runs under gloo with
world_size > 1 and must reproduce the unsharded oracle tree.
"""
import numpy as np

import oracle
from quickrank_amd._capi import NODE_DTYPE
from shard_standin import _Heap

QBITS = 33
NONE = 2 ** 64 - 1


def _bits(x):
    return np.float64(x).view(np.int64)


def _dbl(i):
    return np.int64(i).view(np.float64)


class DocStandinContext:
    def __init__(self, x, labels, qoff, thr, thr_size, rank, world, n_global, q_global):
        self.x = np.ascontiguousarray(x, np.float32)
        self.labels = np.ascontiguousarray(labels, np.float32)
        self.qoff = np.ascontiguousarray(qoff, np.uint64)
        self.N, self.F = self.x.shape
        self.rank, self.world = rank, world
        self.Ng, self.Qg = n_global, q_global
        self.thr, self.thr_size = thr, np.ascontiguousarray(thr_size, np.uint64)
        self.cap = thr.shape[1]
        col = np.ascontiguousarray(self.x.T)
        self.stmap, _ = oracle.binmap(col, self.thr, self.thr_size)   # [F][N] slot ids
        self.cells = self.F * self.cap
        self.hist = np.zeros(2 * self.cells + 2 * world, np.int64)
        self.scal = np.zeros(4 * world, np.int64)
        self.leaf = np.zeros(0, np.int64)
        self.scores = np.zeros(self.N)

    # -- exchange buffers ------------------------------------------------------
    def doc_exchange_buffers(self):
        return dict(hist=0, hist_n=len(self.hist), scal=0, scal_n=len(self.scal),
                    leaf=0, leaf_n=len(self.leaf))

    def host_buffers(self):
        return dict(hist=self.hist, scal=self.scal, leaf=self.leaf, batch=getattr(self, "batch", None))

    # -- pseudo-responses --------------------------------------------------------
    def compute_lambdas(self, metric="NDCG", cutoff=10):
        self.lam, self.w = oracle.lambdas(self.labels, self.scores, self.qoff, cutoff)
        msum = 0.0
        for q in range(len(self.qoff) - 1):
            a, b = int(self.qoff[q]), int(self.qoff[q + 1])
            msum += oracle.eval_dataset(self.labels[a:b], self.scores[a:b],
                                        np.array([0, b - a], np.uint64), cutoff)
        self.scal[:] = 0
        r = self.rank
        self.scal[4 * r] = _bits(np.abs(self.lam).max() if self.N else 0.0)
        self.scal[4 * r + 1] = _bits(float(np.sum(self.lam * self.lam)))
        self.scal[4 * r + 2] = _bits(float(np.sum(self.lam)))
        self.scal[4 * r + 3] = _bits(msum)

    def lambda_finish(self):
        v = self.scal.view(np.float64).reshape(self.world, 4)
        mx = v[:, 0].max()
        self.root_ss = sum(float(a) for a in v[:, 1])      # rank order
        self.root_sum = sum(float(a) for a in v[:, 2])
        self.metric = sum(float(a) for a in v[:, 3]) / self.Qg
        e = QBITS - (int(np.frexp(mx)[1]) if mx > 0 else 0)
        self.scale, self.inv_scale = np.ldexp(1.0, e), np.ldexp(1.0, -e)
        self.q = np.rint(self.lam * self.scale).astype(np.int64)

    # -- histogram in exact integers ----------------------------------------------
    def _local_hist(self, ids, tail):
        s = np.zeros((self.F, self.cap), np.int64)
        c = np.zeros((self.F, self.cap), np.int64)
        if len(ids):
            for f in range(self.F):
                b = self.stmap[f, ids]
                np.add.at(s[f], b, self.q[ids])
                np.add.at(c[f], b, 1)
        self.hist[:self.cells] = np.cumsum(s, axis=1).ravel()
        self.hist[self.cells:2 * self.cells] = np.cumsum(c, axis=1).ravel()
        self.hist[2 * self.cells:] = 0
        self.hist[2 * self.cells + 2 * self.rank] = _bits(tail[0])
        self.hist[2 * self.cells + 2 * self.rank + 1] = _bits(tail[1])

    def _global_hist(self):
        s = self.hist[:self.cells].reshape(self.F, self.cap).copy()
        c = self.hist[self.cells:2 * self.cells].reshape(self.F, self.cap).astype(np.uint64)
        t = self.hist[2 * self.cells:].view(np.float64).reshape(self.world, 2)
        return s, c, sum(float(a) for a in t[:, 0]), sum(float(a) for a in t[:, 1])

    def _best(self, s, c):
        sp = oracle.split_find(s.astype(np.float64) * self.inv_scale, c, self.thr_size, self.minls)
        return None if sp.feature == NONE else (sp.score, int(sp.feature), int(sp.thr_id),
                                                int(sp.lcount), int(sp.rcount))

    # -- protocol ----------------------------------------------------------------
    def tree_begin(self, nleaves, minls):
        self.nleaves, self.minls = nleaves, minls
        if len(self.leaf) != 2 * nleaves * self.world:
            self.leaf = np.zeros(2 * nleaves * self.world, np.int64)
        self.nodes, self.heap = [], _Heap()
        self.taken, self.done, self.step, self.desc = 0, False, 0, None
        self._local_hist(np.arange(self.N), (0.0, 0.0))

    @staticmethod
    def _node(ids, n, sm, ss, hist):
        return dict(ids=ids, n=n, sum=sm, ss=ss, dev=ss - sm * sm / n if n else float("nan"),
                    hist=hist, feature=-1, thr_id=-1, left=-1, right=-1, best=None)

    def _splittable(self, nd):
        return nd["dev"] > 0 and nd["best"] is not None

    def _make_desc(self, i):
        nd = self.nodes[i]
        _, f, t, lc, rc = nd["best"]
        nd["feature"], nd["thr_id"] = f, t
        self.desc = dict(node=i, f=f, t=t, small_is_left=lc <= rc, small_n=min(lc, rc))

    def tree_decide(self):
        s, c, ss_small, sum_small = self._global_hist()
        if self.step == 0:
            root = self._node(np.arange(self.N), self.Ng, self.root_sum, self.root_ss, (s, c))
            root["best"] = self._best(s, c)
            self.nodes = [root]
            if self._splittable(root):
                self._make_desc(0)
            else:
                self.done = True
            self.step = 1
            return
        if self.desc is not None:
            d = self.desc
            P = self.nodes[d["node"]]
            ps, pc = P["hist"]
            bs, bc = ps - s, pc - c                         # sibling by exact subtraction
            small = self._node(d["small_ids"], d["small_n"], sum_small, ss_small, (s, c))
            big = self._node(d["big_ids"], P["n"] - d["small_n"], P["sum"] - sum_small,
                             P["ss"] - ss_small, (bs, bc))
            small["best"], big["best"] = self._best(s, c), self._best(bs, bc)
            L, R = (small, big) if d["small_is_left"] else (big, small)
            P["left"], P["right"] = len(self.nodes), len(self.nodes) + 1
            self.nodes += [L, R]
            self.heap.push(L["dev"], P["left"])
            self.heap.push(R["dev"], P["right"])
            self.desc = None
        self.step += 1
        if self.done:
            return
        while len(self.heap) > 0 and self.taken + len(self.heap) < self.nleaves:
            i = self.heap.pop()
            if self._splittable(self.nodes[i]):
                self._make_desc(i)
                return
            self.taken += 1
        self.done = True

    def tree_apply(self):
        if self.desc is None:
            return                                           # stale buffer is summed and ignored
        d = self.desc
        ids = self.nodes[d["node"]]["ids"]
        go = self.stmap[d["f"], ids] <= d["t"]
        lids, rids = ids[go], ids[~go]
        d["small_ids"], d["big_ids"] = (lids, rids) if d["small_is_left"] else (rids, lids)
        l = self.lam[d["small_ids"]]
        self._local_hist(d["small_ids"], (float(np.sum(l * l)), float(np.sum(l))))

    # -- leaf-wise growth with up to two splits per exchange (include/qr_hip.h, qr_tree_batch_*) ---
    # A restatement of the device's control step (k_tree.hip batch_logic): rt.cc:58-90's loop, plus
    # the split of the most promising OTHER heap entry applied ahead of its turn -- its children
    # wait aside until the loop pops their parent and then take the ids the sequential order
    # assigns.  Exchange: [job][feature][slot][sum, count] per-slot cells of each job's directly
    # built child over the rank's own documents + [job][2 * world] f64 bits.
    BATCH = 2

    def tree_batch_supported(self, nleaves):
        # (as the device: ragged rows of more than 255 thresholds grow one split per exchange)
        return nleaves >= 2 and self.cap <= 256

    def tree_batch_exchange(self):
        return 0, len(self.batch)

    def tree_batch_begin(self, nleaves, minls):
        self.tree_begin(nleaves, minls)
        n = self.BATCH * self.F * self.cap * 2 + self.BATCH * 2 * self.world
        if getattr(self, "batch", None) is None or len(self.batch) != n:
            self.batch = np.zeros(n, np.int64)
        self.jobs, self.real_steps, self.incomplete = [], 0, False
        hint = getattr(self, "steps_force", None) or getattr(self, "steps_hint", 0)
        return max(1, min(nleaves - 1, hint) if hint else nleaves - 1)

    def _commit(self, i, L=None, R=None):
        """the split of node i becomes part of the tree; its children take the next two ids"""
        nd = self.nodes[i]
        _, f, t, _, _ = nd["best"]
        nd["feature"], nd["thr_id"] = f, t
        nd["left"], nd["right"] = len(self.nodes), len(self.nodes) + 1
        self.nodes += [L, R]

    def _next_batch(self, root_mode):
        jobs = []
        if root_mode:
            if self._splittable(self.nodes[0]):
                self._commit(0)
                jobs.append((0, True))
            else:
                self.done = True
        elif not self.done:
            found = False
            while len(self.heap) > 0 and self.taken + len(self.heap) < self.nleaves:
                i = self.heap.pop()
                nd = self.nodes[i]
                if not self._splittable(nd):
                    self.taken += 1
                    continue
                if nd.get("pre") is not None:        # applied ahead of its turn: the children take their ids
                    L, R = nd.pop("pre")
                    self._commit(i, L, R)
                    self.heap.push(L["dev"], nd["left"])
                    self.heap.push(R["dev"], nd["right"])
                    continue
                self._commit(i)
                jobs.append((i, True))
                found = True
                break
            if not found:
                self.done = True
        if len(jobs) == 1:
            while len(jobs) < self.BATCH:
                if self.nleaves - (self.taken + len(self.heap) + 2) < len(jobs):
                    break                              # the leaf budget cannot reach another candidate
                pick, key = -1, 0.0
                for k, v in self.heap.a[1:]:           # the largest deviance left, first in heap order
                    cand = self.nodes[v]
                    if cand.get("pre") is not None or not self._splittable(cand):
                        continue
                    if pick < 0 or k > key:
                        pick, key = v, k
                if pick < 0:
                    break
                jobs.append((pick, False))
        self.jobs = jobs
        if jobs:
            self.real_steps += 1

    def tree_batch_root(self):
        s, c, _, _ = self._global_hist()
        root = self._node(np.arange(self.N), self.Ng, self.root_sum, self.root_ss, (s, c))
        root["best"] = self._best(s, c)
        self.nodes = [root]
        self.step = 1
        self._next_batch(True)

    def tree_batch_apply(self):
        cells = self.batch[:self.BATCH * self.F * self.cap * 2].reshape(self.BATCH, self.F, self.cap, 2)
        tails = self.batch[self.BATCH * self.F * self.cap * 2:].reshape(self.BATCH, 2 * self.world)
        self.applied = []
        for j, (i, in_turn) in enumerate(self.jobs):      # (a job that does not exist: stale cells, ignored)
            nd = self.nodes[i]
            _, f, t, lc, rc = nd["best"]
            ids = nd["ids"]
            go = self.stmap[f, ids] <= t
            lids, rids = ids[go], ids[~go]
            small_is_left = lc <= rc
            sids, bids = (lids, rids) if small_is_left else (rids, lids)
            cells[j] = 0
            for ff in range(self.F):
                b = self.stmap[ff, sids]
                np.add.at(cells[j, ff, :, 0], b, self.q[sids])
                np.add.at(cells[j, ff, :, 1], b, 1)
            l = self.lam[sids]
            tails[j] = 0
            tails[j, 2 * self.rank] = _bits(float(np.sum(l * l)))
            tails[j, 2 * self.rank + 1] = _bits(float(np.sum(l)))
            self.applied.append(dict(node=i, in_turn=in_turn, small_is_left=small_is_left,
                                     small_n=min(lc, rc), small_ids=sids, big_ids=bids))

    def tree_batch_decide(self, last):
        cells = self.batch[:self.BATCH * self.F * self.cap * 2].reshape(self.BATCH, self.F, self.cap, 2)
        tails = self.batch[self.BATCH * self.F * self.cap * 2:].view(np.float64).reshape(self.BATCH, self.world, 2)
        for j, d in enumerate(self.applied):
            s = np.cumsum(cells[j, :, :, 0], axis=1)
            c = np.cumsum(cells[j, :, :, 1], axis=1).astype(np.uint64)
            ss_small = sum(float(a) for a in tails[j, :, 0])     # rank order
            sum_small = sum(float(a) for a in tails[j, :, 1])
            P = self.nodes[d["node"]]
            ps, pc = P["hist"]
            bs, bc = ps - s, pc - c
            small = self._node(d["small_ids"], d["small_n"], sum_small, ss_small, (s, c))
            big = self._node(d["big_ids"], P["n"] - d["small_n"], P["sum"] - sum_small, P["ss"] - ss_small, (bs, bc))
            small["best"], big["best"] = self._best(s, c), self._best(bs, bc)
            L, R = (small, big) if d["small_is_left"] else (big, small)
            if d["in_turn"]:
                self.nodes[P["left"]], self.nodes[P["right"]] = L, R
                self.heap.push(L["dev"], P["left"])          # rt.cc:76-77
                self.heap.push(R["dev"], P["right"])
            else:
                P["pre"] = (L, R)
        self.applied = []
        self.step += 1
        self._next_batch(False)
        self.incomplete = bool(last) and len(self.jobs) > 0

    def tree_batch_settle(self):
        if not self.incomplete:
            self.steps_hint = max(1, self.real_steps)
        return self.incomplete, self.real_steps

    # -- level-wise (oblivious) growth, ot.cc:32-201 over document shards -----------------
    # One int64 all-reduce per level: the per-slot (sum, count) of every directly built
    # child of the level over the rank's own documents (qr_obl_level_exchange).
    def obl_begin(self, depth, minls):
        self.depth, self.minls = depth, minls
        self.nleaves = 1 << depth
        if len(self.leaf) != 2 * self.nleaves * self.world:
            self.leaf = np.zeros(2 * self.nleaves * self.world, np.int64)
        self.nodes, self.obl_done, self.level_buf = [], False, np.zeros(0, np.int64)
        self.pending = None
        self._local_hist(np.arange(self.N), (0.0, 0.0))

    def host_buffers(self):
        return dict(hist=self.hist, scal=self.scal, leaf=self.leaf, level=getattr(self, "level_buf", None),
                    batch=getattr(self, "batch", None))

    def obl_level_exchange(self, level):
        return 0, len(self.level_buf)

    def _level_pick(self, level):
        """fill() + argmax (ot.cc:177-201, 67-92): gains summed over the level's nodes in node
        order, a slot where any node violates minls is out, only sums > 0 compete, first max."""
        lo, hi = (1 << level) - 1, (1 << (level + 1)) - 1
        best = None
        for f in range(self.F):
            tsz = int(self.thr_size[f])
            tot = np.zeros(tsz)
            bad = np.zeros(tsz, bool)
            for i in range(lo, hi):
                s, c = self.nodes[i]["hist"]
                cs = s[f, :tsz].astype(np.float64) * self.inv_scale
                S = float(s[f, tsz - 1]) * self.inv_scale
                lc = c[f, :tsz].astype(np.float64)
                rc = float(c[f, tsz - 1]) - lc
                ok = (lc >= self.minls) & (rc >= self.minls)
                bad |= ~ok
                with np.errstate(divide="ignore", invalid="ignore"):
                    g = cs * cs / lc + (S - cs) * (S - cs) / rc
                tot = tot + np.where(ok, g, 0.0)
            tot[bad] = -1.0
            t = int(np.argmax(tot))
            if tot[t] > 0.0 and (best is None or tot[t] > best[0]):
                best = (float(tot[t]), f, t)
        return best

    def obl_propose(self, level):
        if self.obl_done:
            return
        if level == 0:
            s, c, _, _ = self._global_hist()
            self.nodes = [self._node(np.arange(self.N), self.Ng, self.root_sum, self.root_ss, (s, c))]
        else:
            # the summed level buffer: per-slot values -> cumulative cells of the directly built
            # children, siblings by exact subtraction
            cells = self.level_buf.reshape(len(self.pending), self.F, self.cap, 2)
            for j, d in enumerate(self.pending):
                s = np.cumsum(cells[j, :, :, 0], axis=1)
                c = np.cumsum(cells[j, :, :, 1], axis=1).astype(np.uint64)
                ps, pc = self.nodes[d["node"]]["hist"]
                self.nodes[d["small"]]["hist"] = (s, c)
                self.nodes[d["big"]]["hist"] = (ps - s, pc - c)
        self.pick = self._level_pick(level)
        if self.pick is None:
            self.obl_done = True

    def obl_apply(self, level):
        if self.obl_done:
            return
        _, f, t = self.pick
        lo, hi = (1 << level) - 1, (1 << (level + 1)) - 1
        last = level + 1 == self.depth
        self.pending = []
        while len(self.nodes) < 2 * hi + 1:
            self.nodes.append(None)
        for i in range(lo, hi):
            nd = self.nodes[i]
            s, c = nd["hist"]
            tsz = int(self.thr_size[f])
            lc_all, n_all = int(c[f, t]), int(c[f, tsz - 1])
            rc_all = n_all - lc_all
            go = self.stmap[f, nd["ids"]] <= t
            lids, rids = nd["ids"][go], nd["ids"][~go]
            nd["feature"], nd["thr_id"], nd["left"], nd["right"] = f, t, 2 * i + 1, 2 * i + 2
            self.nodes[2 * i + 1] = self._node(lids, lc_all, 0.0, 0.0, None)
            self.nodes[2 * i + 2] = self._node(rids, rc_all, 0.0, 0.0, None)
            small_is_left = lc_all <= rc_all
            self.pending.append(dict(node=i, small=2 * i + 1 if small_is_left else 2 * i + 2,
                                     big=2 * i + 2 if small_is_left else 2 * i + 1))
        if last:
            return                                           # ot.cc:127: no histograms for the leaves
        buf = np.zeros((len(self.pending), self.F, self.cap, 2), np.int64)
        for j, d in enumerate(self.pending):
            ids = self.nodes[d["small"]]["ids"]
            for ff in range(self.F):
                b = self.stmap[ff, ids]
                np.add.at(buf[j, ff, :, 0], b, self.q[ids])
                np.add.at(buf[j, ff, :, 1], b, 1)
        self.level_buf = buf.ravel()

    def tree_end_local(self, newton=True):
        if getattr(self, "incomplete", False):
            # ended behind a guess that was too low (a tree read lazily): the device's leaf kernels and
            # score update leave at once; the caller settles, carries the tree on and ends it again
            self.redo = True
            return
        while self.nodes and self.nodes[-1] is None:         # (level-wise growth reserves a level ahead)
            self.nodes.pop()
        self.leaves = []
        stack = [0]
        while stack:                                         # DFS, left first (rtnode.cc:34-46)
            i = stack.pop()
            nd = self.nodes[i]
            if nd["feature"] < 0:
                self.leaves.append(i)
            else:
                stack += [nd["right"], nd["left"]]
        self.leaf[:] = 0
        base = self.rank * 2 * self.nleaves
        for l, i in enumerate(self.leaves):
            ids = self.nodes[i]["ids"]
            self.leaf[base + 2 * l] = _bits(float(np.sum(self.lam[ids])))
            self.leaf[base + 2 * l + 1] = _bits(float(np.sum(self.w[ids])) if newton else 0.0)

    def tree_leaves_finish(self, nleaves, newton=True, read=True):
        if getattr(self, "incomplete", False):
            return None
        out = self.last_tree = self._leaves_finish(nleaves, newton)
        if getattr(self, "redo", False):                  # a carried-on tree repeats the score update
            self.redo = False
            if getattr(self, "pending_shrinkage", None) is not None:
                self.update_scores(self.pending_shrinkage)
        self.pending_shrinkage = None
        return out

    def _leaves_finish(self, nleaves, newton=True):
        v = self.leaf.view(np.float64).reshape(self.world, nleaves, 2)
        out = np.zeros(len(self.nodes), NODE_DTYPE)
        for i, nd in enumerate(self.nodes):
            out[i]["feature"], out[i]["thr_id"] = nd["feature"], nd["thr_id"]
            out[i]["left"], out[i]["right"] = nd["left"], nd["right"]
            out[i]["nsamples"] = nd["n"]
            out[i]["deviance"] = nd["dev"]
            if nd["feature"] >= 0:
                out[i]["threshold"] = self.thr[nd["feature"], nd["thr_id"]]
        self.leaf_value = {}
        for l, i in enumerate(self.leaves):
            s1 = sum(float(a) for a in v[:, l, 0])
            s2 = sum(float(a) for a in v[:, l, 1])
            val = (s1 / s2 if s2 >= np.finfo(float).eps else 0.0) if newton else s1 / self.nodes[i]["n"]
            out[i]["value"] = val
            self.leaf_value[i] = val
        return out

    def update_scores(self, shrinkage):
        if getattr(self, "incomplete", False):
            self.pending_shrinkage = shrinkage
            return
        for i, val in self.leaf_value.items():
            ids = self.nodes[i]["ids"]
            self.scores[ids] = self.scores[ids] + shrinkage * val
