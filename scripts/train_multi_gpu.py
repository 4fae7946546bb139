#!/usr/bin/env python
"""LambdaMART / MART on several GPUs of one node, document-sharded.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 scripts/train_multi_gpu.py --train train.svml [--valid valid.svml] \
        --num-trees 500 --num-leaves 10 --num-thresholds 255 --model-out model.xml

Every rank parses the SVMLight files with the host library's parallel reader, keeps a
contiguous range of whole queries, and drives its device context through
quickrank_amd.dist.DocShardedTrainer; rank 0 writes the XML model in the reference's
format.  The loop is Mart::learn's (mart.cc:307-395): early stop on the validation
metric, rollback to the best model.  One GPU: same code, world size 1.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def read_svml(host, path):
    """one pass over the file (the parallel reader of quickrank_amd/host/svml.cc)"""
    sz = C.c_size_t
    N, F, Q = sz(), sz(), sz()
    h = host.qrh_svml_open(path.encode(), C.byref(N), C.byref(F), C.byref(Q))
    x = np.empty((N.value, F.value), np.float32)
    lab = np.empty(N.value, np.float32)
    qoff = np.empty(Q.value + 1, np.uint64)
    host.qrh_svml_copy(h, x.ctypes.data, lab.ctypes.data, qoff.ctypes.data)
    host.qrh_svml_close(h)
    return x, lab, qoff


def shard(x, lab, qoff, rank, world):
    """contiguous ranges of whole queries with about the same number of documents"""
    Q = len(qoff) - 1
    target = [int(qoff[-1]) * r // world for r in range(world + 1)]
    cuts = [int(np.searchsorted(qoff, t, side="left")) for t in target]
    cuts[0], cuts[-1] = 0, Q
    q0, q1 = cuts[rank], max(cuts[rank], cuts[rank + 1])
    d0, d1 = int(qoff[q0]), int(qoff[q1])
    return x[d0:d1], lab[d0:d1], qoff[q0:q1 + 1] - qoff[q0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", required=True)
    ap.add_argument("--valid", default="")
    ap.add_argument("--algo", default="LAMBDAMART", choices=["LAMBDAMART", "MART"])
    ap.add_argument("--num-trees", type=int, default=1000)
    ap.add_argument("--shrinkage", type=float, default=0.1)
    ap.add_argument("--num-thresholds", type=int, default=0)
    ap.add_argument("--num-leaves", type=int, default=10)
    ap.add_argument("--min-leaf-support", type=int, default=1)
    ap.add_argument("--end-after-rounds", type=int, default=100)
    ap.add_argument("--train-metric", default="NDCG", choices=["NDCG", "DCG"])
    ap.add_argument("--train-cutoff", type=int, default=10)
    ap.add_argument("--model-out", default="")
    ap.add_argument("--backend", default="nccl", help="gloo for tests (several ranks on one GPU)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if a.backend == "nccl" else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.init()
    torch.cuda.set_device(local)
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    from quickrank_amd import build
    if rank == 0:
        build.build()
        build.build_host()
    dist.barrier()
    from quickrank_amd._capi import Context, NODE_DTYPE, QrError, QR_ERR_UNSUPPORTED
    from quickrank_amd.dist import DocShardedTrainer, FeatureShardedTrainer, build_doc_bins
    host = C.CDLL(build.HOST_LIB)
    sz = C.c_size_t
    host.qrh_svml_open.argtypes = [C.c_char_p, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    host.qrh_svml_open.restype = C.c_void_p
    host.qrh_svml_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    host.qrh_svml_close.argtypes = [C.c_void_p]
    host.qrh_model_write.argtypes = [C.c_char_p, C.c_int, sz, C.c_double, sz, sz, sz, sz, sz, C.c_void_p, sz, sz]

    x, lab, qoff = read_svml(host, a.train)
    N, Q, F = len(lab), len(qoff) - 1, x.shape[1]
    xs, ls, qs = shard(x, lab, qoff, rank, world)
    ctx = Context(local, rank=rank, world=world, stream=torch.cuda.current_stream().cuda_stream,
                  doc_shard=(N, Q))
    ctx.upload(xs, ls, qs)
    valid = bool(a.valid)
    if valid:
        vx, vl, vq = read_svml(host, a.valid)
        if vx.shape[1] < F:                       # the reader sizes rows by the largest id it met
            vx = np.pad(vx, ((0, 0), (0, F - vx.shape[1])))
        ctx.upload_valid(*shard(vx[:, :F], vl, vq, rank, world))
    layout = "document-sharded"
    try:
        build_doc_bins(ctx, a.num_thresholds)    # (u8 bins up to 255 thresholds per feature, ragged rows beyond)
        tr = DocShardedTrainer(ctx)
    except QrError as e:
        # --num-thresholds 0 (or above 255) on columns whose rows a document-sharded node histogram
        # cannot hold (more than 65,536 distinct values in a column, 4M slots in all): the best split
        # over every distinct value needs prefix sums over ALL documents in slot order, which shards
        # by FEATURE.  Every rank meets the same merged statistics, so every rank lands here: the
        # feature layout -- every document on every rank, the pre-sorted lists of its own columns.
        if e.code != QR_ERR_UNSUPPORTED or world > F:
            raise
        ctx.close()
        layout = "feature-sharded (the lists of every distinct value shard by feature)"
        ctx = Context(local, rank=rank, world=world, stream=torch.cuda.current_stream().cuda_stream)
        ctx.upload(x, lab, qoff)
        if valid:
            ctx.upload_valid(vx[:, :F], vl, vq)
        ctx.build_bins(a.num_thresholds)
        tr = FeatureShardedTrainer(ctx)
    del x
    ctx.reset_scores()
    lam = a.algo == "LAMBDAMART"
    trees, best, best_train, best_valid = [], 0, -np.inf, -np.inf
    if rank == 0:
        print(f"# {a.algo} on {world} GPU(s): {N} docs x {F} features x {Q} queries, {layout}")
        print("# iter. training" + (" validation" if valid else ""))
    for m in range(a.num_trees):
        if valid and a.end_after_rounds and m > best + a.end_after_rounds:   # mart.cc:308-310
            break
        tr.compute_lambdas(a.train_metric, a.train_cutoff) if lam else tr.compute_residuals()
        trees.append(tr.fit_tree(a.num_leaves, a.min_leaf_support, lam))
        ctx.update_scores(a.shrinkage)
        mt = tr.metric_eval(0, a.train_metric, a.train_cutoff)
        star = False
        if valid:
            mv = tr.metric_eval(1, a.train_metric, a.train_cutoff)
            if mv > best_valid:
                best_train, best_valid, best, star = mt, mv, len(trees) - 1, True
        elif mt > best_train:
            best_train, best, star = mt, len(trees) - 1, True
        if rank == 0:
            print(f"{m + 1:7d} {mt:8.4f}" + (f" {mv:8.4f}" if valid else "") + (" *" if star else ""), flush=True)
    if valid:
        trees = trees[:best + 1]                                             # mart.cc:390-395
    if rank == 0:
        print(f"\n{a.train_metric}@{a.train_cutoff} on training data = {best_train:.4f}")
        if valid:
            print(f"{a.train_metric}@{a.train_cutoff} on validation data = {best_valid:.4f}")
        if a.model_out:
            mn = 2 * a.num_leaves + 1
            flat = np.zeros((len(trees), mn), NODE_DTYPE)
            flat["feature"] = -1
            flat["left"] = flat["right"] = -1
            for i, t in enumerate(trees):
                flat[i, :len(t)] = t
            rc = host.qrh_model_write(a.model_out.encode(), 1 if lam else 0, a.num_trees, a.shrinkage,
                                      a.num_thresholds, a.num_leaves, a.min_leaf_support, a.end_after_rounds, 0,
                                      flat.ctypes.data, len(trees), mn)
            print(f"# model written to {a.model_out}" if rc == 0 else "!!! could not write the model")
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
