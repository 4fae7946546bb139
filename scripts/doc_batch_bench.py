"""Document-sharded protocols on ONE GPU, world = 1, over RCCL on the context's stream (GPU box tool).

What a rank of a multi-GPU run enqueues per boosting iteration -- kernels AND collectives (a
one-rank all-reduce still is a launch on the stream) -- with nothing to wait for from other
ranks: the per-iteration cost of the protocol itself, next to the single-context iteration.

    python scripts/doc_batch_bench.py [docs_in_millions=1] [steps=40]
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402


def main():
    import torch
    import torch.distributed as dist
    import bench
    import quickrank_amd as qr
    from quickrank_amd.dist import DocShardedTrainer, gather_thresholds
    nm = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    Q, DPQ, F = 10000 * nm, 100, 136
    x, labels, qoff = bench.synth(Q, DPQ, F, seed=42)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1)
    out = {}
    for mode in ("single", "one_split", "batched"):
        if mode == "single":
            c = qr.Context(0)
            c.upload(x, labels, qoff)
            c.build_bins(255)
            tr = None
        else:
            c = qr.Context(0, stream=torch.cuda.current_stream().cuda_stream, doc_shard=(len(labels), Q))
            c.upload(x, labels, qoff)
            c.build_bins_with(*gather_thresholds(c, 255))
            tr = DocShardedTrainer(c)
        c.reset_scores()

        def step():
            if tr is None:
                c.compute_lambdas("NDCG", 10)
                c.fit_tree(10, 1, True, read=False)
            else:
                tr.compute_lambdas("NDCG", 10)
                tr.fit_tree(10, 1, True, read=False, batched=(mode == "batched"))
            c.update_scores(0.1)

        for _ in range(8):
            step()
        c.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        c.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        out[mode] = ms
        extra = ""
        if tr is not None and hasattr(tr, "collectives"):
            extra = f" (histogram all-reduces of the last tree: {tr.collectives})"
        print(f"{mode:10s} {ms:.4f} ms per iteration{extra}", flush=True)
        c.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
