"""Host-side stand-in for a feature-sharded device context (TEST INFRASTRUCTURE).

Implements the phase protocol of include/qr_hip.h (qr_tree_begin / decide / apply /
end + the exchange buffers) with the CPU oracle doing the per-rank histogram work,
so the collective sequence of quickrank_amd.dist.ShardedTreeFitter and the
record-merge rule can run under gloo on CPU with world_size 2.
"""
import ctypes as C

import numpy as np

import oracle
from quickrank_amd._capi import NODE_DTYPE, SPLIT_DTYPE
from quickrank_amd.dist import owned_features

NONE = 0xFFFFFFFF


class _Heap:
    """maxheap.h:58-88"""

    def __init__(self):
        self.a = [(float("inf"), -1)]

    def __len__(self):
        return len(self.a) - 1

    def push(self, key, val):
        self.a.append(None)
        p = len(self.a) - 1
        while key > self.a[p >> 1][0]:
            self.a[p] = self.a[p >> 1]
            p >>= 1
        self.a[p] = (key, val)

    def pop(self):
        top = self.a[1][1]
        last = self.a.pop()
        size = len(self.a) - 1
        if size == 0:
            return top
        p = 1
        while (p << 1) <= size:
            child = p << 1
            if child < size and self.a[child + 1][0] > self.a[child][0]:
                child += 1
            if last[0] < self.a[child][0]:
                self.a[p] = self.a[child]
            else:
                break
            p = child
        self.a[p] = last
        return top


class StandinContext:
    def __init__(self, x, nthr, rank, world):
        self.tr = oracle.Trainer(x, nthr)
        self.N, self.F = self.tr.N, self.tr.F
        self.rank, self.world = rank, world
        self.own = owned_features(self.F, rank, world)
        self.recs_local = np.zeros(2, SPLIT_DTYPE)
        self.recs_all = np.zeros(2 * world, SPLIT_DTYPE)
        self.mask = np.zeros((self.N + 31) // 32, np.int32)
        self.obl_mask = np.zeros(len(self.mask) + 256, np.int32)   # go-left bits by document + the level's left counts
        self.lam = self.w = None

    # -- exchange buffers ------------------------------------------------------
    def exchange_buffers(self):
        return dict(rec_bytes=2 * SPLIT_DTYPE.itemsize, mask_bytes=self.mask.nbytes)

    def host_buffers(self):
        return dict(recs_local=self.recs_local.view(np.uint8), recs_all=self.recs_all.view(np.uint8),
                    mask=self.mask)

    def set_pseudo(self, lam, w):
        self.lam, self.w = np.ascontiguousarray(lam), np.ascontiguousarray(w)

    # -- helpers ---------------------------------------------------------------
    def _hist(self, ids):
        t = self.tr
        s = np.zeros((self.F, t.cap))
        c = np.zeros((self.F, t.cap), np.uint64)
        sub_s, sub_c, ss = oracle.hist_build(np.ascontiguousarray(t.stmap[self.own]),
                                             np.ascontiguousarray(t.thr_size[self.own]), t.cap,
                                             self.lam, ids)
        s[self.own], c[self.own] = sub_s, sub_c
        return s, c, ss

    def _local_best(self, s, c):
        sp = oracle.split_find(s, c, self.tr.thr_size, self.minls, int(self.own[0]), int(self.own[-1]) + 1)
        r = np.zeros(1, SPLIT_DTYPE)[0]
        r["score"] = sp.score
        r["feature"] = NONE if sp.feature == 2 ** 64 - 1 else sp.feature
        r["thr_id"] = NONE if sp.feature == 2 ** 64 - 1 else sp.thr_id
        r["lcount"], r["rcount"] = sp.lcount, sp.rcount
        return r

    def _merge(self, which):
        best = None
        for r in range(self.world):
            x = self.recs_all[2 * r + which]
            if x["feature"] == NONE:
                continue
            if best is None or x["score"] > best["score"] or \
                    (x["score"] == best["score"] and x["feature"] < best["feature"]):
                best = x.copy()
        return best

    def _stats(self, ids, ss):
        sm = float(np.sum(self.lam[ids])) if len(ids) else 0.0
        n = len(ids)
        dev = ss - sm * sm / n if n else float("nan")
        return dict(ids=ids, ss=ss, dev=dev, n=n)

    # -- protocol ----------------------------------------------------------------
    def tree_begin(self, nleaves, minls):
        self.nleaves, self.minls = nleaves, minls
        ids = np.arange(self.N, dtype=np.uint64)
        s, c, ss = self._hist(None)
        root = self._stats(ids, ss)
        root.update(hist=(s, c), feature=-1, thr_id=-1, left=-1, right=-1, best=None)
        self.nodes = [root]
        self.heap = _Heap()
        self.taken, self.done, self.step, self.desc = 0, False, 0, None
        self.recs_local[0] = self._local_best(s, c)
        self.recs_local[1]["feature"] = NONE
        self.recs_local[1]["score"] = -1

    def _splittable(self, nd):
        return nd["dev"] > 0 and nd["best"] is not None

    def _make_desc(self, i):
        nd = self.nodes[i]
        b = nd["best"]
        nd["feature"], nd["thr_id"] = int(b["feature"]), int(b["thr_id"])
        self.desc = dict(node=i, f=int(b["feature"]), t=int(b["thr_id"]))
        self.mask[:] = 0
        if nd["feature"] in self.own:
            go = self.tr.stmap[nd["feature"], nd["ids"]] <= nd["thr_id"]
            bits = np.zeros(len(self.mask) * 32, np.uint8)
            bits[:len(go)] = go
            self.mask[:] = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).view(np.int32).ravel()

    def tree_decide(self):
        if self.step == 0:
            self.nodes[0]["best"] = self._merge(0)
            if self._splittable(self.nodes[0]):
                self._make_desc(0)
            else:
                self.done = True
            self.step = 1
            return
        if self.desc is not None:
            d = self.desc
            nd = self.nodes[d["node"]]
            self.nodes[nd["left"]]["best"] = self._merge(0)
            self.nodes[nd["right"]]["best"] = self._merge(1)
            self.heap.push(self.nodes[nd["left"]]["dev"], nd["left"])
            self.heap.push(self.nodes[nd["right"]]["dev"], nd["right"])
            self.desc = None
        self.step += 1
        self.mask[:] = 0
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
            return
        nd = self.nodes[self.desc["node"]]
        n = nd["n"]
        w = self.mask.view(np.uint32)
        go = ((w[np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1).astype(bool)
        lids, rids = nd["ids"][go], nd["ids"][~go]
        ls, lc, lss = self._hist(lids)
        ps, pc = nd["hist"]
        rs, rc = ps - ls, pc - lc
        L = self._stats(lids, lss)
        R = self._stats(rids, nd["ss"] - lss)
        for ch, h in ((L, (ls, lc)), (R, (rs, rc))):
            ch.update(hist=h, feature=-1, thr_id=-1, left=-1, right=-1, best=None)
        nd["left"], nd["right"] = len(self.nodes), len(self.nodes) + 1
        self.nodes += [L, R]
        self.recs_local[0] = self._local_best(ls, lc)
        self.recs_local[1] = self._local_best(rs, rc)

    # -- level-wise (oblivious) growth, ot.cc:32-201, feature-sharded ----------------------
    # Per level: all-gather of the ranks' best (feature, slot) of the level (slot 0 of the
    # record pair), then a sum all-reduce of the owner's go-left bits by DOCUMENT followed by
    # the left count of every node of the level (QR_MAXLEVEL = 256 words).
    OBL_MAXLEVEL = 256

    def obl_exchange_buffers(self):
        return dict(mask=0, mask_bytes=(len(self.mask) + self.OBL_MAXLEVEL) * 4, recs_local=0, recs_all=0,
                    rec_bytes=2 * SPLIT_DTYPE.itemsize)

    def host_buffers(self):
        return dict(recs_local=self.recs_local.view(np.uint8), recs_all=self.recs_all.view(np.uint8),
                    mask=self.mask, obl_mask=getattr(self, "obl_mask", None))

    def obl_begin(self, depth, minls):
        self.depth, self.minls = depth, minls
        ids = np.arange(self.N, dtype=np.uint64)
        s, c, ss = self._hist(None)
        self.nodes = [dict(ids=ids, n=self.N, hist=(s, c), feature=-1, thr_id=-1, left=-1, right=-1)]
        self.obl_done = False

    def obl_propose(self, level):
        self.recs_local[0]["feature"], self.recs_local[0]["score"] = NONE, -1.0
        self.recs_local[1]["feature"], self.recs_local[1]["score"] = NONE, -1.0
        if self.obl_done:
            return
        lo, hi = (1 << level) - 1, (1 << (level + 1)) - 1
        best = None
        for f in self.own:
            tsz = int(self.tr.thr_size[f])
            tot = np.zeros(tsz)
            bad = np.zeros(tsz, bool)
            for i in range(lo, hi):
                s, c = self.nodes[i]["hist"]
                cs, S = s[f, :tsz], s[f, tsz - 1]
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
                best = (float(tot[t]), int(f), t)
        if best is not None:
            self.recs_local[0]["score"], self.recs_local[0]["feature"], self.recs_local[0]["thr_id"] = best

    def obl_mark(self, level):
        self.obl_mask[:] = 0
        if self.obl_done:
            return
        pick = self._merge(0)
        if pick is None:
            self.obl_done = True
            return
        self.pick = (int(pick["feature"]), int(pick["thr_id"]))
        f, t = self.pick
        if f in self.own:
            go = self.tr.stmap[f] <= t                       # by DOCUMENT: all nodes take the same split
            bits = np.zeros(len(self.mask) * 32, np.uint8)
            bits[:self.N] = go
            self.obl_mask[:len(self.mask)] = np.packbits(bits.reshape(-1, 32)[:, ::-1], axis=1).view(">u4") \
                .astype(np.uint32).view(np.int32).ravel()
            lo, hi = (1 << level) - 1, (1 << (level + 1)) - 1
            for j, i in enumerate(range(lo, hi)):
                self.obl_mask[len(self.mask) + j] = int(self.nodes[i]["hist"][1][f, t])

    def obl_apply(self, level):
        if self.obl_done:
            return
        f, t = self.pick
        w = self.obl_mask[:len(self.mask)].view(np.uint32)
        lo, hi = (1 << level) - 1, (1 << (level + 1)) - 1
        last = level + 1 == self.depth
        while len(self.nodes) < 2 * hi + 1:
            self.nodes.append(None)
        for j, i in enumerate(range(lo, hi)):
            nd = self.nodes[i]
            d = nd["ids"].astype(np.int64)
            go = ((w[d >> 5] >> (d & 31).astype(np.uint32)) & 1).astype(bool)
            lids, rids = nd["ids"][go], nd["ids"][~go]
            assert len(lids) == int(self.obl_mask[len(self.mask) + j])     # the counts came with the mask
            nd["feature"], nd["thr_id"], nd["left"], nd["right"] = f, t, 2 * i + 1, 2 * i + 2
            L = dict(ids=lids, n=len(lids), hist=None, feature=-1, thr_id=-1, left=-1, right=-1)
            R = dict(ids=rids, n=len(rids), hist=None, feature=-1, thr_id=-1, left=-1, right=-1)
            if not last:                                     # ot.cc:127: no histograms for the leaves
                ls, lc, _ = self._hist(lids)
                ps, pc = nd["hist"]
                L["hist"], R["hist"] = (ls, lc), (ps - ls, pc - lc)
            self.nodes[2 * i + 1], self.nodes[2 * i + 2] = L, R

    def obl_end(self, depth, newton, read=True):
        while self.nodes and self.nodes[-1] is None:
            self.nodes.pop()
        return self.tree_end(1 << depth, newton)

    def tree_end(self, nleaves, newton):
        out = np.zeros(len(self.nodes), NODE_DTYPE)
        for i, nd in enumerate(self.nodes):
            out[i]["feature"], out[i]["thr_id"] = nd["feature"], nd["thr_id"]
            out[i]["left"], out[i]["right"] = nd["left"], nd["right"]
            out[i]["nsamples"] = nd["n"]
            if nd["feature"] < 0:
                s1, s2 = self.lam[nd["ids"]].sum(), self.w[nd["ids"]].sum()
                out[i]["value"] = s1 / s2 if s2 >= np.finfo(float).eps else 0.0
        return out
