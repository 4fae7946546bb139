#!/usr/bin/env python
"""bench.py -- docs/sec per LambdaMART boosting iteration on synthetic MSLR-shaped data.

Workload (BASELINE.json configs[1]): 1,000,000 docs x 136 features x 10,000
queries (100 docs each), LambdaMART, 255 thresholds (256 slots), 10 leaves,
shrinkage 0.1, min-leaf-support 1, NDCG@10, no validation set.

A step = one boosting iteration (mart.cc:307-383): per-query lambdas/weights,
root histogram, tree fit (split scans, partitions, child histograms), leaf
outputs, training-score update and training NDCG@10 -- all on the device,
inputs resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps 60 --warmup 5   (the defaults)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (quickrank_amd/dist.py), two layouts:
  --shard docs      (default) every rank holds its OWN 1M-document shard (whole
                    queries, all features; seed 42 + rank): the job is N x 1M documents,
                    one int64 all-reduce per node histogram, per-GPU work fixed
                    -> "scaling": "weak"; value = all ranks' documents / time.
                    The same run then also measures BASELINE.json's configs[2] as it
                    is written (object "config2_feature_sharded": the SAME 1M set on
                    every rank, feature blocks sharded, "strong"); --no-config2 skips it.
  --shard features  the 1M-document set replicated, feature blocks of the bin
                    matrix sharded -> "scaling": "strong" as the headline (DESIGN.md
                    section 6 on why this cannot beat one GPU at 1M documents).
Prints ONE JSON line on rank 0 (stdout carries nothing else: library banners are
routed to stderr).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth(nq, dpq, F, seed=42, sparse_cols=0, zero_frac=0.9):
    """MSLR-shaped synthetic data: U[0,1) f32 features, labels 0..4 driven by
    the first four features.  sparse_cols > 0 turns the LAST
    that many columns into MSLR-like count features: zero with probability
    zero_frac, else one of 32 integer levels (the "uniques <= nthresholds" threshold
    branch, and one very hot bin per column)."""
    rng = np.random.default_rng(seed)
    N = nq * dpq
    x = rng.random((N, F), dtype=np.float32)
    if sparse_cols:
        c0 = F - sparse_cols
        lv = np.floor(x[:, c0:] * 32).astype(np.float32) + 1
        lv[rng.random((N, sparse_cols), dtype=np.float32) < zero_frac] = 0
        x[:, c0:] = lv
    labels = np.minimum(4, np.floor(1.25 * x[:, :4].sum(axis=1, dtype=np.float64))).astype(np.float32)
    qoff = (np.arange(nq + 1, dtype=np.uint64) * dpq)
    return x, labels, qoff


def cpu_baseline(x, labels, qoff, args):
    """The oracle (C + OpenMP restatement of the reference's loop, bit-exact to
    it on the fixtures) timed on this host's cores on a bounded sample."""
    import oracle
    cores = oracle.available_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    nq = min(len(qoff) - 1, args.cpu_queries)
    n = int(qoff[nq])
    iters = args.cpu_iters
    m = oracle.train(x[:n], labels[:n], qoff[:nq + 1], algo="LAMBDAMART", ntrees=iters,
                     shrinkage=0.1, nthresholds=args.nthresholds, nleaves=args.nleaves, minls=1,
                     esr=0, threads=cores)
    sec = float(np.mean(m["iter_seconds"][1:])) if iters > 1 else float(m["iter_seconds"][0])
    return {"value": n / sec, "unit": "docs/s per boosting iteration", "cores": cores,
            "kind": "port",
            "sample": f"first {nq} queries ({n} docs x {x.shape[1]} features) of the same synthetic "
                      f"set, {iters} iterations, first discarded, {sec * 1e3:.1f} ms/iteration, "
                      f"OpenMP x{cores}"}


def scoring_metric(ctx, args, torch, rank=0, world=1, dist=None):
    """Second metric of BASELINE.json ("ensemble-score docs/sec") on config 5 at full
    size by default: 10,000 trees x 64 leaves over 10M docs x 200 f32 features.  The
    8 GB of features are generated on the device (seed 43) and handed over as a
    device pointer (qr_ensemble_score_device); timed with HIP events on the
    context's stream, after one warm-up pass.  N > 1: the documents are sharded over
    the ranks (every rank holds the whole model, no collective in the data path):
    "strong" scaling, value = all documents / the slowest rank's time."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from score_bench import make_model
    from quickrank_amd._capi import Context
    rng = np.random.default_rng(43)
    nodes, w = make_model(args.score_trees, 6, 200, rng)
    g = torch.Generator(device="cuda")
    g.manual_seed(43 + rank)
    per = (args.score_docs + world - 1) // world
    ndocs = max(0, min(per, args.score_docs - rank * per))      # this rank's shard
    xs = torch.rand((max(ndocs, 1), 200), generator=g, device="cuda", dtype=torch.float32)
    out = torch.empty(max(ndocs, 1), device="cuda", dtype=torch.float64)
    sc = Context(torch.cuda.current_device(), stream=torch.cuda.current_stream().cuda_stream)
    sc.upload_ensemble(nodes, w)

    def run():
        sc._ck(sc.L.qr_ensemble_score_device(sc.h, C.c_void_p(xs.data_ptr()), max(ndocs, 1), 200,
                                             C.c_void_p(out.data_ptr())))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    sc.close()
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    return {"metric": "ensemble-score docs/sec", "value": args.score_docs / ms * 1e3, "unit": "docs/s",
            "node_visits_per_s": args.score_docs * args.score_trees * 6 / ms * 1e3,
            "workload": f"{args.score_trees} trees x 64 leaves (depth 6) over {args.score_docs} docs x 200 "
                        "features, synthetic, features resident on the device"
                        + (f", documents sharded over {world} GPUs" if world > 1 else ""),
            "scaling": "strong", "n_gpus": world, "ms": ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--docs-per-query", type=int, default=100)
    ap.add_argument("--features", type=int, default=136)
    ap.add_argument("--nleaves", type=int, default=10)
    ap.add_argument("--nthresholds", type=int, default=255)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scoring", action="store_true")
    ap.add_argument("--score-trees", type=int, default=10000)
    ap.add_argument("--score-docs", type=int, default=10000000)
    ap.add_argument("--cpu-queries", type=int, default=2500)
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--sparse-cols", type=int, default=0,
                    help="make the last K feature columns MSLR-like sparse counts (robustness runs)")
    ap.add_argument("--zero-frac", type=float, default=0.9)
    ap.add_argument("--shard", choices=["docs", "features"], default="docs")
    ap.add_argument("--no-config2", action="store_true",
                    help="N > 1, --shard docs: skip the extra feature-sharded (config 2) measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, sharded driver) with one rank")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that print banners to
    # file descriptor 1 (RCCL does at communicator creation) go to stderr instead
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node == --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: quickrank_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from quickrank_amd import build as qbuild
    if rank == 0:
        qbuild.build()
    if dist is not None:
        dist.barrier()
    from quickrank_amd._capi import Context

    docs_mode = multi and args.shard == "docs"
    x, labels, qoff = synth(args.queries, args.docs_per_query, args.features,
                            seed=42 + rank if docs_mode else 42, sparse_cols=args.sparse_cols,
                            zero_frac=args.zero_frac)
    N, F = x.shape
    n_job = N * world if docs_mode else N        # documents one step processes
    stream = torch.cuda.current_stream().cuda_stream if multi else None
    fitter = trainer = None
    if docs_mode:
        from quickrank_amd.dist import DocShardedTrainer, gather_thresholds
        ctx = Context(local_rank, rank=rank, world=world, stream=stream,
                      doc_shard=(n_job, args.queries * world))
        ctx.upload(x, labels, qoff)
        ctx.build_bins_with(*gather_thresholds(ctx, args.nthresholds))
        ctx.reset_scores()
        trainer = DocShardedTrainer(ctx)
    else:
        ctx = Context(local_rank, rank=rank, world=world, stream=stream)
        ctx.upload(x, labels, qoff)
        ctx.build_bins(args.nthresholds)
        ctx.reset_scores()
        if multi:
            from quickrank_amd.dist import ShardedTreeFitter
            fitter = ShardedTreeFitter(ctx)

    ndcg, trees = [], []

    def step():
        # ranking + NDCG@10 of the current scores (= the training metric the
        # reference evaluates at the end of the previous iteration, mart.cc:347)
        # + lambdas/weights in one pass over the queries; tree; leaf outputs; score
        # update.  Everything is enqueued first; the metric and the tree records are
        # read last (pinned snapshots + events), so the host never drains the stream
        # in the middle of an iteration.
        if trainer is not None:
            trainer.compute_lambdas("NDCG", 10)
            trainer.fit_tree(args.nleaves, 1, True, read=False)
        else:
            ctx.compute_lambdas("NDCG", 10)
            if fitter is not None:
                fitter.fit_tree(ctx, args.nleaves, 1, True, read=False)
            else:
                ctx.fit_tree(args.nleaves, 1, True, read=False)
        ctx.update_scores(0.1)
        ndcg.append(ctx.metric_last())
        trees.append(ctx.tree_nodes())

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = ctx.prof_get()
    ctx.prof_enable(False)

    # BASELINE.json configs[2] literally: the SAME 1M x 136 set on every rank, feature
    # blocks of the bin matrix sharded, total work fixed ("strong").  Reported next
    # to the headline when the headline is the document-sharded layout.
    config2 = None
    if docs_mode and not args.no_config2:
        from quickrank_amd.dist import ShardedTreeFitter
        xf, lf, qf = synth(args.queries, args.docs_per_query, args.features, seed=42)
        c2 = Context(local_rank, rank=rank, world=world, stream=stream)
        c2.upload(xf, lf, qf)
        c2.build_bins(args.nthresholds)
        c2.reset_scores()
        f2 = ShardedTreeFitter(c2)

        def step2():
            c2.compute_lambdas("NDCG", 10)
            f2.fit_tree(c2, args.nleaves, 1, True, read=False)
            c2.update_scores(0.1)
            c2.metric_last()
            c2.tree_nodes()
        for _ in range(args.warmup):
            step2()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        sync()
        e2 = time.perf_counter() - t0
        t = torch.tensor([e2], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2 = float(t.item())
        config2 = {"workload": f"the same {len(lf)} docs x {xf.shape[1]} features on every rank",
                   "parallelism": f"feature-block sharding x{world}: best-split records all-gather + "
                                  "go-left mask all-reduce per split",
                   "scaling": "strong", "value": len(lf) * args.steps / e2, "unit": "docs/s",
                   "ms_per_step": e2 / args.steps * 1e3}
        c2.close()
        del xf, lf, qf

    def tree_shape(t):
        # sigma = documents whose histogram is built directly per
        # tree / N (we build the smaller child; the reference always the left one),
        # pi = documents partitioned per tree / N
        internal = np.nonzero(t["feature"] >= 0)[0]
        nl = t["nsamples"][t["left"][internal]].astype(np.float64)
        nr = t["nsamples"][t["right"][internal]].astype(np.float64)
        tot = float(t["nsamples"][0])
        return (np.minimum(nl, nr).sum() / tot, nl.sum() / tot,
                t["nsamples"][internal].astype(np.float64).sum() / tot)

    scoring = None
    if not args.no_scoring:   # every rank takes part (its shard of the documents)
        scoring = scoring_metric(ctx, args, torch, rank, world, dist)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = n_job * args.steps / elapsed
        roof = None
        if prof["launches"]:
            sec = prof["total_ms"] / prof["launches"] * 1e-3
            ach = prof["alg_bytes"] / sec / 1e9
            # HBM bytes per launch from the PMC passes committed under profiles/
            # (collected separately: counters cannot be read inside this process);
            # only quoted when it was measured on this very workload
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_hist.json")
            if (os.path.exists(pmc) and world == 1 and N == 1000000 and F == 136
                    and args.nthresholds == 255):
                traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
            roof = {"bound": "hbm", "kernel": "k_hist_root (root histogram build)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": prof["alg_bytes"],
                    "avg_launch_us": round(sec * 1e6, 2), "launches": prof["launches"]}
        out = {
            "metric": "docs/sec per LambdaMART boosting iter (1Mx136)",
            "value": value, "unit": "docs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak" if docs_mode or world == 1 else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic {n_job} docs x {F} features x "
                                   f"{args.queries * (world if docs_mode else 1)} queries"
                                   + (f" ({N} docs per GPU)" if docs_mode else "") + ", "
                                   f"LambdaMART {args.nleaves} leaves, {args.nthresholds} thresholds, "
                                   "NDCG@10, shrinkage 0.1, min-leaf-support 1"
                                   + (f", last {args.sparse_cols} columns sparse counts ({args.zero_frac:.0%} zeros)"
                                      if args.sparse_cols else ""),
                       "parallelism": "1 GPU" if world == 1 else
                                      (f"document sharding x{world}: one int64 all-reduce per node histogram"
                                       if docs_mode else f"feature-block sharding x{world}"),
                       "ndcg10_last": ndcg[-1] if ndcg else None,
                       "sigma_built": round(float(np.mean([tree_shape(t)[0] for t in trees[-args.steps:]])), 3),
                       "sigma_reference_left": round(float(np.mean([tree_shape(t)[1] for t in trees[-args.steps:]])), 3),
                       "pi": round(float(np.mean([tree_shape(t)[2] for t in trees[-args.steps:]])), 3)},
            "roofline": roof,
        }
        if config2 is not None:
            out["config2_feature_sharded"] = config2
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x, labels, qoff, args)
        if scoring is not None:
            out["ensemble_scoring"] = scoring
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
