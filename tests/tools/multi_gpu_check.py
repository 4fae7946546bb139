#!/usr/bin/env python
"""Parity of the two multi-GPU layouts on REAL GPUs (one rank per device, RCCL):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port 29500 tests/tools/multi_gpu_check.py

Every rank also trains on the whole set with a private single-GPU context and
compares: feature-sharded trees must be bit-identical, document-sharded trees
identical in structure with leaf values to rounding.  (The CI box has one GPU: there
the same drivers are exercised by tests/test_gpu_multiproc.py over gloo.)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.init()
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import quickrank_amd as qr
    from datagen import make_dataset
    from quickrank_amd.dist import DocShardedTrainer, ShardedTreeFitter, gather_thresholds
    x, labels, qoff = make_dataset(nq=400, docs_per_query=60, F=136, seed=77, adversarial=True)
    N, Q = len(labels), len(qoff) - 1
    stream = torch.cuda.current_stream().cuda_stream
    single = qr.Context(local)
    single.upload(x, labels, qoff)
    single.build_bins(255)
    single.reset_scores()
    # --- document-sharded
    cuts = [Q * r // world for r in range(world + 1)]
    q0, q1 = cuts[rank], cuts[rank + 1]
    d0, d1 = int(qoff[q0]), int(qoff[q1])
    c = qr.Context(local, rank=rank, world=world, stream=stream, doc_shard=(N, Q))
    c.upload(x[d0:d1], labels[d0:d1], qoff[q0:q1 + 1] - qoff[q0])
    c.build_bins_with(*gather_thresholds(c, 255))
    c.reset_scores()
    tr = DocShardedTrainer(c)
    ok_docs = True
    for it in range(6):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(10, 2, True)
        single.update_scores(0.1)
        tr.compute_lambdas("NDCG", 10)
        got = tr.fit_tree(10, 2, True)
        c.update_scores(0.1)
        for k in ("feature", "thr_id", "left", "right", "nsamples", "threshold"):
            ok_docs = ok_docs and np.array_equal(got[k], want[k])
        ok_docs = ok_docs and np.allclose(got["value"], want["value"], rtol=1e-10, atol=1e-13)
        ok_docs = ok_docs and abs(tr.metric_eval(0) - single.metric_eval(0)) < 1e-11
    c.close()
    # --- feature-sharded
    single.reset_scores()
    f = qr.Context(local, rank=rank, world=world, stream=stream)
    f.upload(x, labels, qoff)
    f.build_bins(255)
    f.reset_scores()
    fit = ShardedTreeFitter(f)
    ok_feat = True
    for it in range(6):
        single.compute_lambdas("NDCG", 10)
        want = single.fit_tree(10, 2, True)
        single.update_scores(0.1)
        f.compute_lambdas("NDCG", 10)
        got = fit.fit_tree(f, 10, 2, True)
        f.update_scores(0.1)
        for k in want.dtype.names:
            ok_feat = ok_feat and np.array_equal(got[k], want[k])
    ok_feat = ok_feat and np.array_equal(f.get_scores(), single.get_scores())
    f.close()
    single.close()
    t = torch.tensor([int(ok_docs), int(ok_feat)], device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"world {world}: document-sharded {'OK' if t[0].item() else 'MISMATCH'}, "
              f"feature-sharded {'OK' if t[1].item() else 'MISMATCH'}")
    dist.destroy_process_group()
    sys.exit(0 if t.min().item() else 1)


if __name__ == "__main__":
    main()
