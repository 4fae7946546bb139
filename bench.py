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
  python bench.py --gpus N ...   (no launcher: bench.py starts the N ranks itself)

N = 1 adds to the line: `roofline` (root histogram launch, HIP events on the launch),
`roofline_iteration` (algorithmic bytes of the whole iteration from the measured tree
shape / ms_per_step), `roofline_child_hist` (the child-histogram launches, a separate
short pass), `cpu_baseline` (the oracle on the whole set, all cores and 8 threads),
`ensemble_scoring` (config 5 at full size, with its own CPU baseline) and `strong_8M`
(the same training on 8M documents, the 1-GPU point of the larger scaling set).

N > 1 (quickrank_amd/dist.py): the headline is BASELINE.json configs[2] -- the SAME 1M
set, total work fixed ("scaling": "strong").  Both layouts are measured with K steps
each and the faster one is the headline (`config.parallelism` names it, both are under
`strong_layouts`):
  document_sharded  rank r holds queries [r Q/N, (r+1) Q/N) and every feature of them;
                    one int64 all-reduce per node histogram, everything else local
  feature_sharded   north_star's layout: every rank holds all documents and 1/N of the
                    feature columns of the bin matrix; best-split records all-gather +
                    go-left mask all-reduce per split (DESIGN.md section 6 on why it
                    cannot beat one GPU at 1M documents)
Extra keys: `weak_scaling` (every rank its own 1M shard, N x 1M documents in all) and
`strong_8M` (8M documents, the shards of the document-sharded layout), `--extra-steps`
timed steps each.  A run aborts if the ranks' trees differ.
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


def synth_mslr(nq=6000, mean_q=120, F=136, sparse_cols=40, seed=7):
    """Stand-in for MSLR-WEB10K fold 1 (BASELINE.json configs[0]; the files are not in the
    image): ~nq * mean_q documents in ragged queries (log-normal sizes, mean ~mean_q,
    clipped to [1, 1200]); the last `sparse_cols` columns are sparse count features (90 %
    zeros, else one of 32 integer levels: the "uniques <= nthresholds" threshold branch
    with one very hot bin per column), the others real-valued U[0,1); labels 0..4 with
    MSLR's skew P = .52/.32/.13/.02/.01, driven by four real columns, two count columns
    and noise."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.round(rng.lognormal(np.log(mean_q) - 0.18, 0.6, nq)), 1, 1200).astype(np.int64)
    qoff = np.zeros(nq + 1, np.uint64)
    qoff[1:] = np.cumsum(sizes)
    N = int(qoff[-1])
    x = rng.random((N, F), dtype=np.float32)
    if sparse_cols:
        c0 = F - sparse_cols
        lv = np.floor(x[:, c0:] * 32).astype(np.float32) + 1
        lv[rng.random((N, sparse_cols), dtype=np.float32) < 0.9] = 0
        x[:, c0:] = lv
    rel = (0.5 * x[:, 0] + 0.4 * x[:, 1] + 0.3 * x[:, 2] + 0.2 * x[:, 3]).astype(np.float64)
    if sparse_cols:
        rel += 0.15 * (x[:, F - 1] > 0) + 0.1 * (x[:, F - 2] > 0)
    rel += 0.3 * rng.standard_normal(N)
    cuts = np.quantile(rel, [0.52, 0.84, 0.97, 0.99])
    labels = np.searchsorted(cuts, rel).astype(np.float32)
    return x, labels, qoff


def synth_device(torch, blocks, nq, dpq, F, seed0=1142):
    """The same MSLR-shaped rows as synth(), generated ON THE DEVICE in blocks of nq * dpq
    documents (block b: torch generator seeded seed0 + b), so that a set too large to build and
    ship from the host in the bench's time -- 32M documents x 136 features = 17.4 GB -- is
    resident in HBM when the timed region starts.  Any rank that takes whole blocks gets the
    rows every other world size would give those blocks.  Returns (rows: CUDA tensor [n][F] f32,
    labels: host f32 [n], qoff: host u64)."""
    nb = len(blocks)
    n1 = nq * dpq
    x = torch.empty((nb * n1, F), device="cuda", dtype=torch.float32)
    lab = torch.empty(nb * n1, device="cuda", dtype=torch.float32)
    for i, b in enumerate(blocks):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed0 + int(b))
        xb = x[i * n1:(i + 1) * n1]
        xb.copy_(torch.rand((n1, F), generator=g, device="cuda", dtype=torch.float32))
        lab[i * n1:(i + 1) * n1] = torch.clamp(torch.floor(1.25 * xb[:, :4].to(torch.float64).sum(dim=1)), max=4.0).to(torch.float32)
    qoff = np.arange(nb * nq + 1, dtype=np.uint64) * dpq
    return x, lab.cpu().numpy(), qoff


def newest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (the counters of a launch cannot be
    read inside this process: they are quoted from the committed PMC passes, with their source)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return c[-1] if c else None


def hist_source_fingerprint():
    """sha1 of the histogram accumulation's SOURCE (k_tree.hip, `hist_accumulate` .. `k_hist_batch`): the
    PMC passes under profiles/ carry the fingerprint of the kernel they were collected on, and a line
    that quotes counters of another kernel says so (the figure must not age silently)."""
    import hashlib
    try:
        t = open(os.path.join(ROOT, "quickrank_amd", "csrc", "k_tree.hip")).read()
        a = t.index("__device__ __forceinline__ void hist_accumulate(")
        b = t.index("// k_reduce: sum the workgroup partials")
        return hashlib.sha1(t[a:b].encode()).hexdigest()[:16]
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


REFERENCE_8CORE_S_PER_ITER = 0.298   # the reference itself, 8 Xeon cores, this workload


def cpu_baseline(x, labels, qoff, args):
    """The oracle (C + OpenMP restatement of the reference's loop, bit-exact to
    it on the fixtures) timed on this host's cores on the FULL workload (SURVEY.md
    section 8d): all cores it may use, and again with 8 threads; a handful of
    iterations, the first discarded (about 10-20 s of CPU work in all, most of it
    the one-time argsort / threshold / bin-map initialisation)."""
    import oracle
    cores = oracle.available_cores()
    nq = min(len(qoff) - 1, args.cpu_queries) if args.cpu_queries else len(qoff) - 1
    n = int(qoff[nq])
    iters = args.cpu_iters

    def run(threads):
        os.environ["OMP_NUM_THREADS"] = str(threads)
        m = oracle.train(x[:n], labels[:n], qoff[:nq + 1], algo="LAMBDAMART", ntrees=iters,
                         shrinkage=0.1, nthresholds=args.nthresholds, nleaves=args.nleaves, minls=1,
                         esr=0, threads=threads)
        sec = float(np.mean(m["iter_seconds"][1:])) if iters > 1 else float(m["iter_seconds"][0])
        return sec, m
    sec_all, m = run(cores)
    out = {"value": n / sec_all, "unit": "docs/s per boosting iteration", "cores": cores,
           "cpu_model": cpu_model(), "kind": "port",
           "sample": ("the whole" if nq == len(qoff) - 1 else f"first {nq} queries of the") +
                     f" synthetic set ({n} docs x {x.shape[1]} features), {iters} iterations, first "
                     f"discarded, {sec_all * 1e3:.1f} ms/iteration, OpenMP x{cores}",
           "ms_per_iteration": sec_all * 1e3,
           "ndcg10_after": {str(iters): float(m["train_metric"][-1])},
           # how the port compares with the reference itself: BASELINE.md section 2 timed the
           # reference's own build at 0.298 s per iteration on 8 Xeon cores (another box);
           # the port needs 0.19-0.25 s there -- i.e. it is not a slower stand-in
           "reference_built_8core_other_box": {"s_per_iteration": REFERENCE_8CORE_S_PER_ITER,
                                               "docs_per_s": 1e6 / REFERENCE_8CORE_S_PER_ITER,
                                               "source": "BASELINE.md section 2"}}
    if cores != 8:
        sec8, _ = run(min(8, cores))
        out["threads8"] = {"value": n / sec8, "cores": min(8, cores), "ms_per_iteration": sec8 * 1e3,
                           "port_vs_reference_8threads_other_box": REFERENCE_8CORE_S_PER_ITER / sec8}
    else:
        out["port_vs_reference_8threads_other_box"] = REFERENCE_8CORE_S_PER_ITER / sec_all
    return out


def cpu_baseline_wide(x, labels, qoff, args):
    """The oracle with the reference's default `--num-thresholds 0` on the MSLR-shaped stand-in, all
    cores, two iterations (the second is the one quoted): the CPU figure beside
    `mslr_default_thresholds`."""
    import oracle
    cores = oracle.available_cores()
    m = oracle.train(x, labels, qoff, algo="LAMBDAMART", ntrees=2, shrinkage=0.1, nthresholds=0,
                     nleaves=args.nleaves, minls=1, esr=0, threads=cores)
    sec = float(m["iter_seconds"][-1])
    return {"value": len(labels) / sec, "unit": "docs/s per boosting iteration", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "ms_per_iteration": sec * 1e3,
            "sample": f"the whole stand-in, 2 iterations, the second quoted, OpenMP x{cores}"}


def cpu_scoring_baseline(args, make_model):
    """Scoring baseline (LTR_Algorithm::score_dataset, ltr_algorithm.cc:44-52: OpenMP over
    the documents, per-tree walk): the oracle's restatement on a bounded sample of
    config 5 -- the whole 10,000-tree model over the first `cpu_score_docs` documents."""
    import time as _t
    import oracle
    cores = oracle.available_cores()
    oracle.lib().qro_set_threads(cores)
    rng = np.random.default_rng(43)
    nodes, w = make_model(args.score_trees, 6, 200, rng)
    nd = args.cpu_score_docs
    x = np.random.default_rng(44).random((nd, 200), dtype=np.float32)
    model = dict(nodes=nodes, nnodes=np.full(len(nodes), nodes.shape[1], np.uint64), ntrees=len(nodes),
                 max_nodes=nodes.shape[1], shrinkage=0.1)
    oracle.ensemble_score(model, x[:256])
    t0 = _t.perf_counter()
    oracle.ensemble_score(model, x)
    sec = _t.perf_counter() - t0
    return {"value": nd / sec, "unit": "docs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "node_visits_per_s": nd * args.score_trees * 6 / sec,
            "sample": f"{args.score_trees} trees x 64 leaves over {nd} docs x 200 features, {sec:.1f} s, "
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
    # the same walk on leaf-wise SHAPED trees (what LambdaMART with 64 leaves leaves behind: a group
    # of trees runs for as many steps as its deepest member has levels), at config 5's FULL size
    # (config 5 does not state the trees' shape; both are quoted)
    from score_bench import make_leafwise_model
    lt, ld = max(16, args.score_trees), max(1, ndocs)
    ln_, lw_, lshape = make_leafwise_model(lt, 64, 200, np.random.default_rng(44))
    sc.upload_ensemble(ln_, lw_)
    def run2():
        sc._ck(sc.L.qr_ensemble_score_device(sc.h, C.c_void_p(xs.data_ptr()), ld, 200, C.c_void_p(out.data_ptr())))
    run2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run2()
    e1.record()
    torch.cuda.synchronize()
    lms = e0.elapsed_time(e1) / reps
    # ... and with the trees walked AND SUMMED in ascending depth (opt-in: qr_ensemble_set_depth_order;
    # the f64 sum in another order than ensemble.cc:111-118's: compared with the strict order here)
    strict = out[:ld].clone()
    sc.upload_ensemble(ln_, lw_, depth_order=True)
    run2()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run2()
    e1.record()
    torch.cuda.synchronize()
    dms = e0.elapsed_time(e1) / reps
    drel = float(((out[:ld] - strict).abs().max() / strict.abs().mean()).item())   # (against the scores' scale)
    sc.close()
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    visits = args.score_docs * args.score_trees * 6 / ms * 1e3
    # what bounds the walk: per node visit one 4-byte node record and one bin
    # byte, both at data-dependent LDS addresses = two dword-wide LDS accesses per lane;
    # peak = ds_read_b32 at 128 B/clk/CU (MI355X_MICROARCH.md, LDS table) x 256 CUs x 2.4 GHz
    lds_peak = 128.0 * 256 * 2.4   # GB/s
    lds_ach = visits * 8.0 / 1e9 / max(world, 1)
    return {"metric": "ensemble-score docs/sec", "value": args.score_docs / ms * 1e3, "unit": "docs/s",
            "node_visits_per_s": visits,
            "roofline": {"bound": "lds", "kernel": "k_score_p4 (binned tree walk out of LDS-staged tree tiles)",
                         "achieved": round(lds_ach, 1), "peak": round(lds_peak, 1), "unit": "GB/s",
                         "frac": round(lds_ach / lds_peak, 4), "lds_bytes_per_node_visit": 8,
                         "note": "ds_read_b32 (node record, level order: one bank per node of a level) + "
                                 "ds_read_u8 (bin, one dword per lane and feature quad: conflict-free) per "
                                 "(document, tree, level) and three vector instructions between them; the PMC "
                                 "passes (profiles/r03_score_pmc.md) put the LDS array at 52 % busy and the "
                                 "SIMDs' issue slots at 70 %: a visit's five instructions are the other bound; "
                                 "HBM traffic is the one 8 GB pass of k_doc_bins (~2 ms of the total)"},
            "workload": f"{args.score_trees} trees x 64 leaves (depth 6) over {args.score_docs} docs x 200 "
                        "features, synthetic, features resident on the device"
                        + (f", documents sharded over {world} GPUs" if world > 1 else ""),
            "scaling": "strong", "n_gpus": world, "ms": ms,
            "leafwise_shaped": {"workload": f"{lt} leaf-wise shaped trees x 64 leaves over {ld} docs x 200 features "
                                            "(this rank's)", "shape": lshape, "ms": lms,
                                "value": ld / lms * 1e3, "unit": "docs/s",
                                "docs_per_s_per_1000_trees": ld / lms * 1e3 * lt / 1000.0,
                                "note": "a batch's trees walk in lockstep for as many steps as the deepest has "
                                        "levels (DESIGN.md 3.6): cost follows the max depth, not the mean path",
                                "depth_order": {"ms": dms, "docs_per_s_per_1000_trees": ld / dms * 1e3 * lt / 1000.0,
                                                "max_abs_diff_vs_model_order_over_mean_abs_score": drel,
                                                "what": "opt-in (qr_ensemble_set_depth_order): the trees walked and "
                                                        "their f64 contributions added in ascending depth, so that a "
                                                        "batch's trees end together; equal to the model's order to f64 "
                                                        "rounding, not bit for bit"}}}


def tree_shape(t):
    """sigma = documents whose histogram is built directly per tree / N
    (we build the smaller child; the reference always the left one), pi = documents
    partitioned per tree / N."""
    internal = np.nonzero(t["feature"] >= 0)[0]
    nl = t["nsamples"][t["left"][internal]].astype(np.float64)
    nr = t["nsamples"][t["right"][internal]].astype(np.float64)
    tot = float(t["nsamples"][0])
    return (np.minimum(nl, nr).sum() / tot, nl.sum() / tot,
            t["nsamples"][internal].astype(np.float64).sum() / tot)


def obl_shape(t):
    """tree_shape for an oblivious tree (nodes in heap order 2i+1 / 2i+2, ot.cc): the children
    of the LAST internal level are leaves and get no histogram (ot.cc:127)."""
    internal = np.nonzero(t["feature"] >= 0)[0]
    tot = float(t["nsamples"][0])
    if len(internal) == 0:
        return 0.0, 0.0, 0.0
    last_level_first = (len(internal) + 1) // 2 - 1      # first internal node of the deepest internal level
    upper = internal[internal < last_level_first]
    nl = t["nsamples"][t["left"][upper]].astype(np.float64)
    nr = t["nsamples"][t["right"][upper]].astype(np.float64)
    return (np.minimum(nl, nr).sum() / tot, nl.sum() / tot,
            t["nsamples"][internal].astype(np.float64).sum() / tot)


def iteration_alg_bytes_obl(N, F, depth, sigma, pi):
    """iteration_alg_bytes for level-wise growth: one gain scan per node histogram
    (2^depth - 1 of them), no histograms for the leaves."""
    return (28.0 * N + (N * F + 8.0 * N) + sigma * N * (F + 12) + ((1 << depth) - 1) * F * 256 * 16
            + 12.0 * pi * N + 20.0 * N + 16.0 * N + 12.0 * N)


def iteration_alg_bytes(N, F, L, sigma, pi):
    """Algorithmic bytes of one boosting iteration (SURVEY.md 8d "whole iteration"):
    lambdas 28 N + root histogram (N F + 8 N) + child histograms sigma N (F + 12) + split
    scans (2 (L - 1) + 1) F 256 16 + partition 12 pi N + leaf outputs 20 N + score update
    16 N + NDCG 12 N."""
    return (28.0 * N + (N * F + 8.0 * N) + sigma * N * (F + 12) + (2 * (L - 1) + 1) * F * 256 * 16
            + 12.0 * pi * N + 20.0 * N + 16.0 * N + 12.0 * N)


class Run:
    """One timed configuration: a layout over this rank's part of a data set."""

    def __init__(self, torch, dist, layout, x, labels, qoff, n_global, q_global, args, rank, world,
                 local_rank):
        from quickrank_amd._capi import Context
        self.torch, self.dist, self.args, self.layout = torch, dist, args, layout
        self.n_global = n_global
        stream = torch.cuda.current_stream().cuda_stream if layout != "single" else None
        self.trainer = self.fitter = None
        if layout == "docs":
            from quickrank_amd.dist import DocShardedTrainer, gather_thresholds
            self.ctx = Context(local_rank, rank=rank, world=world, stream=stream,
                               doc_shard=(n_global, q_global))
            self._upload(x, labels, qoff)
            self.ctx.build_bins_with(*gather_thresholds(self.ctx, args.nthresholds))
            self.ctx.reset_scores()
            self.trainer = DocShardedTrainer(self.ctx)
            self.comm = self.trainer.direct
        else:
            self.ctx = Context(local_rank, rank=rank if layout == "features" else 0,
                               world=world if layout == "features" else 1, stream=stream)
            t0 = time.perf_counter()
            self._upload(x, labels, qoff)
            self.ctx.synchronize()
            self.h2d_ms = (time.perf_counter() - t0) * 1e3      # host rows -> HBM (PCIe), outside the timed region
            t0 = time.perf_counter()
            self.ctx.build_bins(args.nthresholds)
            self.ctx.synchronize()
            self.init_ms = (time.perf_counter() - t0) * 1e3     # Mart::init: thresholds + bin map
            self.ctx.reset_scores()
            self.comm = None
            if layout == "features":
                from quickrank_amd.dist import ShardedTreeFitter
                self.fitter = ShardedTreeFitter(self.ctx)
                self.comm = self.fitter.direct
        self.ndcg, self.trees = [], []

    def _upload(self, x, labels, qoff):
        """Host rows (numpy) or rows generated on the device (a torch CUDA tensor)."""
        if hasattr(x, "data_ptr"):
            self.ctx.upload_device(x.data_ptr(), x.shape[0], x.shape[1], labels, qoff)
        else:
            self.ctx.upload(x, labels, qoff)

    def step(self):
        # ranking + NDCG@10 of the current scores (= the training metric the
        # reference evaluates at the end of the previous iteration, mart.cc:347)
        # + lambdas/weights in one pass over the queries; tree; leaf outputs; score
        # update.  Everything is enqueued first; the metric and the tree records are read
        # from pinned snapshots the kernels publish, so the host never drains the stream in
        # the middle of an iteration -- and the records of tree i are read after iteration
        # i + 1's lambda pass is enqueued (qr_lambda_compute settles tree i on its last control
        # call, ~35 us before its leaf kernels and score update have run; nothing of tree i is
        # overwritten before the next fit_tree), so the GPU always has work queued.
        a, ctx = self.args, self.ctx
        if getattr(self, "obl_depth", 0):     # Oblivious-LambdaMART (BASELINE.json configs[3])
            ctx.compute_lambdas("NDCG", 10)
            self.flush()
            ctx.fit_oblivious(self.obl_depth, 1, True, read=False)
            ctx.update_scores(0.1)
            self.tree_pending = True
            self.ndcg.append(ctx.metric_last())
            return
        if self.trainer is not None:
            self.trainer.compute_lambdas("NDCG", 10)
            self.flush()
            self.trainer.fit_tree(a.nleaves, 1, True, read=False)
        else:
            ctx.compute_lambdas("NDCG", 10)
            self.flush()
            if self.fitter is not None:
                self.fitter.fit_tree(ctx, a.nleaves, 1, True, read=False)
            else:
                ctx.fit_tree(a.nleaves, 1, True, read=False)
        ctx.update_scores(0.1)
        self.tree_pending = True
        self.ndcg.append(ctx.metric_last())

    def flush(self):
        """The records of the last enqueued tree (every tree is read exactly once)."""
        if getattr(self, "tree_pending", False):
            if self.trainer is not None:
                self.trainer.settle()     # (a tree enqueued behind a guessed number of steps: carried on if needed)
            self.trees.append(self.ctx.tree_nodes())
            self.tree_pending = False

    def sync(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def timed(self, steps, warmup, prof=False):
        """W untimed steps, then exactly K steps between barrier + synchronize on both
        sides; returns the max over ranks of the elapsed seconds."""
        for _ in range(warmup):
            self.step()
        self.flush()
        if prof:
            # HIP events on every 4th root launch of the timed region: an evented launch costs
            # the stream ~7.5 us (scripts/ubench/launch_chain.hip), which is the timed region's
            self.ctx.prof_enable(True, every=4)
            self.ctx.prof_reset()
        self.sync()
        who = self.trainer if self.trainer is not None else self.fitter
        before = list(who.traffic) if who is not None else None
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.flush()
        self.sync()
        elapsed = time.perf_counter() - t0
        # collectives and payload bytes this rank handed over per tree of the timed region (counted by the
        # drivers in quickrank_amd/dist.py, not timed separately)
        self.traffic_per_tree = None if before is None else [(who.traffic[0] - before[0]) / steps,
                                                              (who.traffic[1] - before[1]) / steps]
        if self.dist is not None:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device="cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        self.steps, self.elapsed = steps, elapsed
        return elapsed

    def trees_identical_across_ranks(self):
        """Every rank must have built the same trees (structure bit for bit)."""
        if self.dist is None:
            return True
        import hashlib
        h = hashlib.sha256()
        for t in self.trees:
            for k in ("feature", "thr_id", "left", "right", "nsamples"):
                h.update(np.ascontiguousarray(t[k]).tobytes())
        parts = [None] * self.dist.get_world_size()
        self.dist.all_gather_object(parts, h.hexdigest())
        return len(set(parts)) == 1

    def summary(self, what, parallelism, shape=None):
        sh = [(shape or tree_shape)(t) for t in self.trees[-self.steps:]]
        return {"workload": what, "parallelism": parallelism,
                "value": self.n_global * self.steps / self.elapsed, "unit": "docs/s",
                "ms_per_step": self.elapsed / self.steps * 1e3, "steps": self.steps,
                "ndcg10_last": self.ndcg[-1] if self.ndcg else None,
                "sigma_built": round(float(np.mean([a[0] for a in sh])), 3),
                "sigma_reference_left": round(float(np.mean([a[1] for a in sh])), 3),
                "pi": round(float(np.mean([a[2] for a in sh])), 3),
                "collectives_per_tree": None if getattr(self, "traffic_per_tree", None) is None
                else round(self.traffic_per_tree[0], 2),
                "collective_bytes_per_tree": None if getattr(self, "traffic_per_tree", None) is None
                else int(self.traffic_per_tree[1]),
                "collectives": ("rccl-direct on the context's stream, nranks "
                                f"{self.comm.nranks}" if self.comm is not None else
                                ("torch.distributed" if self.dist is not None and self.layout != "single"
                                 else "none"))}

    def close(self):
        if self.comm is not None:
            self.comm.close()
        self.ctx.close()


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
    ap.add_argument("--cpu-queries", type=int, default=0, help="0 = the whole set")
    ap.add_argument("--cpu-score-docs", type=int, default=20000)
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--sparse-cols", type=int, default=0,
                    help="make the last K feature columns MSLR-like sparse counts (robustness runs)")
    ap.add_argument("--zero-frac", type=float, default=0.9)
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement (no weak-scaling / 8M / child-histogram passes)")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of the extra measurements")
    ap.add_argument("--big-blocks", type=int, default=8,
                    help="the larger strong-scaling set: this many 1M-document blocks (0 = skip)")
    ap.add_argument("--huge-blocks", type=int, default=64,
                    help="the strong-scaling set at which >= 6x on 8 GPUs is arithmetically possible (DESIGN.md 6): "
                         "this many 1M-document blocks generated on the device (0 = skip)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, sharded drivers) with one rank")
    ap.add_argument("--self-launch", action="store_true",
                    help="re-exec under torch.distributed.run even with --gpus 1 (what --gpus N > 1 does "
                         "when no launcher started this process; tests)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start
    # the N ranks ourselves, one process per GPU, exactly as the documented command line does
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_launch):
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        passed = [a for a in sys.argv[1:] if a != "--self-launch"]
        if args.self_launch and "--force-dist" not in passed:
            passed.append("--force-dist")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + passed
        print("bench.py: no launcher environment, starting " + " ".join(cmd), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)

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

    Q, DPQ, F = args.queries, args.docs_per_query, args.features
    x, labels, qoff = synth(Q, DPQ, F, seed=42, sparse_cols=args.sparse_cols, zero_frac=args.zero_frac)
    N = len(labels)
    desc = (f"LambdaMART {args.nleaves} leaves, {args.nthresholds} thresholds, NDCG@10, shrinkage 0.1, "
            "min-leaf-support 1"
            + (f", last {args.sparse_cols} columns sparse counts ({args.zero_frac:.0%} zeros)"
               if args.sparse_cols else ""))
    same_set = f"synthetic {N} docs x {F} features x {Q} queries"

    def doc_slice(nq_total, r, w):
        """Whole queries [q0, q1) of rank r of w."""
        return nq_total * r // w, nq_total * (r + 1) // w   # balanced: no rank is left empty while w <= nq_total

    def mk(layout, xx, ll, qq, n_global, q_global):
        return Run(torch, dist, layout, xx, ll, qq, n_global, q_global, args, rank, world, local_rank)

    extras, roof, roof_child, roof_iter = {}, None, None, None
    if not multi:
        # ---- N = 1: BASELINE.json configs[1] on one GPU ---------------------------
        head = mk("single", x, labels, qoff, N, Q)
        head.timed(args.steps, args.warmup, prof=not os.environ.get("QR_BENCH_NO_EVENTS"))
        prof = head.ctx.prof_get()
        in_region = int(prof["launches"])
        hs = head.summary(same_set + ", " + desc, "1 GPU")
        # (at least MIN_ROOT_EVENTS evented root launches whatever --steps is.  The
        # timed region carries events on every 4th launch -- an evented launch costs the stream
        # ~7.5 us, which would be the timed region's -- and a pass of its own BEHIND it, events on
        # every launch, tops the count up; an event pair reads the kernel's own begin / end
        # stamps, so a launch measures the same in both.)
        MIN_ROOT_EVENTS = 16
        if in_region and in_region < MIN_ROOT_EVENTS and not os.environ.get("QR_BENCH_NO_EVENTS"):
            head.ctx.prof_enable(True, every=1)
            n_keep = len(head.trees)
            for _ in range(MIN_ROOT_EVENTS - in_region):
                head.step()
            head.flush()
            del head.trees[n_keep:], head.ndcg[-(MIN_ROOT_EVENTS - in_region):]
            prof = head.ctx.prof_get()
        head.ctx.prof_enable(False)
        scaling = None
        if prof["launches"]:
            sec = prof["total_ms"] / prof["launches"] * 1e-3
            ach = prof["alg_bytes"] / sec / 1e9
            # HBM bytes per launch from the PMC passes committed under profiles/ (counters
            # cannot be read inside this process): quoted with its source, and only when it
            # was collected on this very workload with the kernel as it is now
            traffic, tsrc = None, None
            pmc = newest_profile("pmc_hist.json")
            if pmc and N == 1000000 and F == 136 and args.nthresholds == 255 \
                    and not args.sparse_cols:
                pj = json.load(open(pmc))
                traffic = pj["hbm_bytes_per_launch"]
                tsrc = ("profiles/" + os.path.basename(pmc) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                        "collected with scripts/r06_final.sh: counters cannot be read inside the process)")
                fp = hist_source_fingerprint()
                if pj.get("kernel_source_fingerprint") != fp:
                    tsrc += (f"; STALE: collected on a kernel whose source fingerprint was {pj.get('kernel_source_fingerprint')}, "
                             f"this one is {fp}")
            roof = {"bound": "hbm", "kernel": "k_hist_root (root histogram build)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": tsrc,
                    "alg_bytes_per_launch": prof["alg_bytes"],
                    "avg_launch_us": round(sec * 1e6, 2), "launches": prof["launches"],
                    "launches_in_timed_region": in_region}
            # The launch's SECOND bound, measured here and now (qr_prof_lds_atomic, k_ubench.hip): it
            # issues one ds_add_u64 per (document, accumulated column) and a CU retires them no
            # faster than the bare microbenchmark does.  wave instructions per CU = documents x
            # accumulated columns / active lanes per wave / CUs.
            try:
                lb = head.ctx.prof_lds_atomic()
                instr_cu = lb["root_wave_instr_per_cu"]
                bound_us = instr_cu * lb["ns_per_instr"] * 1e-3
                roof["lds_atomic_bound"] = {
                    "what": "one ds_add_u64 per (document, accumulated column): the CU's LDS pipeline",
                    "cycles_per_ds_add_u64": round(lb["cycles_per_instr"], 2),
                    "shader_ghz_under_load": round(lb["shader_ghz"], 3),
                    "wave_instr_per_cu": round(instr_cu, 0), "bound_us": round(bound_us, 2),
                    "launch_over_bound": round(sec * 1e6 / bound_us, 3) if bound_us else None,
                    "hbm_frac_at_this_bound": round(prof["alg_bytes"] / (bound_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    if bound_us else None,
                    "note": "measured in this process, outside the timed region; the launch adds the prologue, the "
                            "98 KB flush per workgroup and its write-back on top of this floor"}
                if roof["frac"] < 0.5 and bound_us:
                    need_us = prof["alg_bytes"] / (0.5 * HBM_PEAK_GBS * 1e9) * 1e6
                    roof["target_unreachable_because"] = (
                        f"north_star's 0.50 of the HBM peak is a launch of {need_us:.1f} us; the launch's ds_add_u64 "
                        f"instructions alone occupy every CU's LDS pipeline for {bound_us:.1f} us at the clock the bare "
                        f"microbenchmark holds (measured here, lds_atomic_bound) -- {bound_us / need_us:.2f} of that "
                        "budget -- and what is left for the prologue (descriptor + first tiles), zeroing 96 KB of LDS, "
                        "flushing it as 25 MB of partial slots and the dispatch of 256 workgroups of 1024 threads is "
                        f"{need_us - bound_us:.1f} us where the launch needs {sec * 1e6 - bound_us:.1f}.  The issue count "
                        "is within 7 % of its minimum (N x 136 / 64 lanes: 63 of 64 lanes busy, 8 padding columns in "
                        "144); an 8-column lane mapping (17 x 8 = 136) leaves 60 of 64 lanes busy and issues 2.267 "
                        "instructions per document against 2.286 now (DESIGN.md 3.1)")
            except Exception as e:   # the side measurement must not take the line down
                roof["lds_atomic_bound"] = {"error": str(e)}
        ib = iteration_alg_bytes(N, F, args.nleaves, hs["sigma_built"], hs["pi"])
        ia = ib / (hs["ms_per_step"] * 1e-3) / 1e9
        roof_iter = {"bound": "hbm", "what": "whole boosting iteration (every kernel, launch gaps included)",
                     "alg_bytes_per_step": ib, "achieved": round(ia, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ia / HBM_PEAK_GBS, 4)}
        if not args.no_extras:
            # child-histogram launches, timed with HIP events on the launches in a
            # separate short pass (the events would perturb the headline's host loop)
            head.ctx.prof_enable(True, children=True)
            head.ctx.prof_reset()
            n0 = len(head.trees)
            for _ in range(args.extra_steps):
                head.step()
            pc = head.ctx.prof_get_child()
            head.ctx.prof_enable(False)
            built = sum(tree_shape(t)[0] for t in head.trees[n0:]) * N   # documents built directly
            # ... and the lambda pass's launch the same way, in a third short pass (VERDICT r3 item 3:
            # "state its bound").  k_lambda moves 28 B per document -- it is not an HBM kernel; a
            # query is ~3 k vector instructions of f64 compares, fused multiply-adds and DPP moves on
            # ONE wave, and what bounds the launch is the rate at which the 1024 SIMDs issue them: a
            # wave64 vector instruction occupies its SIMD's issue port for 4 cycles (16 lanes per
            # cycle), f64 at the same rate as f32 on this chip.  The instruction count per query comes
            # from the SQ counters of the committed PMC pass of this command (profiles/).
            head.ctx.prof_enable(True, lambdas=True, every=255)
            head.ctx.prof_reset()
            for _ in range(args.extra_steps):
                head.step()
            pl = head.ctx.prof_get_child()
            head.ctx.prof_enable(False)
            roof_lambda = None
            if pl["launches"]:
                lus = pl["total_ms"] * 1e3 / pl["launches"]
                roof_lambda = {"bound": "valu_issue", "kernel": "k_lambda (one wave per query: ranking, NDCG@10, lambdas)",
                               "avg_launch_us": round(lus, 2), "launches": pl["launches"],
                               "hbm_frac": round(28.0 * N / (lus * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "peak": round(1024 * 2.4 / 4, 1), "unit": "G wave-instructions/s",
                               "peak_what": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 vector instruction"}
                lp = newest_profile("lambda_pmc.json")
                if lp and N == 1000000 and Q == 10000:
                    lj = json.load(open(lp))
                    vi = lj["valu_insts_per_query"]
                    ach = vi * Q / (lus * 1e-6) / 1e9
                    roof_lambda.update({"valu_insts_per_query": vi, "salu_insts_per_query": lj.get("salu_insts_per_query"),
                                        "lds_insts_per_query": lj.get("lds_insts_per_query"),
                                        "insts_source": "profiles/" + os.path.basename(lp) + " (rocprofv3 --pmc SQ_INSTS_VALU "
                                                        "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES pass of this command)",
                                        "issued": round(ach, 1), "issue_slot_occupancy": round(ach / (1024 * 2.4 / 4), 4),
                                        "issue_slot_occupancy_what": "instructions EXECUTED over issue slots: an occupancy "
                                        "figure (redundant instructions raise it), not a roofline fraction -- that is `frac`"})
                # The ALGORITHMIC count: per query of n documents at cutoff k --
                # n log2 n compares (a comparison sort's floor for the ranking), one exponential per
                # document (~20 f64 operations), k DCG terms, and per (rank r1 < k, rank r2 > r1) pair
                # term of lambdamart.cc:120-141 with ndcg.cc:76-88's closed form 22 f64 operations
                # (3 for the swap delta, 2 to normalise it, 7 for 1 / (e1 + e2) by Newton steps, 2 for
                # rho, 2 + 3 for lambda and weight, 3 to accumulate both sides); pairs of EQUAL labels
                # are skipped by the reference too but counted here (an upper bound on the useful work).
                # Lane operations / 64 = the wave instructions a perfect mapping would issue.
                nq_ = np.diff(qoff.astype(np.int64)).astype(np.float64)
                kk = np.minimum(10.0, nq_)
                pairs = nq_ * kk - kk * (kk + 1) / 2
                lane_ops = float((pairs * 22 + nq_ * np.log2(np.maximum(nq_, 2)) + nq_ * 20 + kk * 4).sum())
                alg = lane_ops / 64 / (lus * 1e-6) / 1e9
                roof_lambda.update({"alg_lane_ops_per_launch": lane_ops, "achieved": round(alg, 1),
                                    "frac": round(alg / (1024 * 2.4 / 4), 4),
                                    "frac_what": "algorithmic lane operations / 64 per second over the chip's vector issue "
                                                 "peak; the rest is what one wave per query pays around them: cross-lane "
                                                 "reductions, the counting rank's n^2 / 64 compares, LDS staging, the tie sort"})
                extras["roofline_lambda"] = roof_lambda
            # BASELINE.json configs[3]: Oblivious-LambdaMART depth 6 on the same set (level-batched
            # growth); configs[0]: the MSLR-WEB10K run, 100 trees x 10 leaves, on the
            # MSLR-shaped stand-in (ragged queries, 40 sparse count columns; the files are not in
            # the image).  Each with its own whole-iteration roofline.
            def side_run(xx, ll, qq, what, steps, obl_depth=0, warm=None):
                r = mk("single", xx, ll, qq, len(ll), len(qq) - 1)
                r.obl_depth = obl_depth
                r.timed(steps, min(args.warmup, 3) if warm is None else warm)
                sm = r.summary(what, "1 GPU", obl_shape if obl_depth else None)
                n_, f_ = len(ll), xx.shape[1]
                ib_ = (iteration_alg_bytes_obl(n_, f_, obl_depth, sm["sigma_built"], sm["pi"]) if obl_depth
                       else iteration_alg_bytes(n_, f_, args.nleaves, sm["sigma_built"], sm["pi"]))
                ia_ = ib_ / (sm["ms_per_step"] * 1e-3) / 1e9
                sm["roofline_iteration"] = {"bound": "hbm", "alg_bytes_per_step": ib_, "achieved": round(ia_, 1),
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ia_ / HBM_PEAK_GBS, 4)}
                sm["h2d_ms"], sm["init_ms"] = round(r.h2d_ms, 1), round(r.init_ms, 1)
                r.close()
                return sm
            extras["oblivious_d6"] = side_run(
                x, labels, qoff, same_set + f", Oblivious-LambdaMART depth 6 (64 leaves), {args.nthresholds} "
                "thresholds, NDCG@10, shrinkage 0.1, min-leaf-support 1", args.extra_steps, obl_depth=6)
            # ... and configs[3]'s scoring path: a 1000-tree depth-6 oblivious ensemble (random, of
            # the trained model's shape) over the same rows, generate_oblivious.cc's bit-interleaved
            # scorer (k_doc_bins + k_obl_score_s: HIP events around the two kernels)
            try:
                from quickrank_amd._capi import Context as _Ctx
                rg = np.random.default_rng(7)
                oc = _Ctx(torch.cuda.current_device())
                oc.upload_oblivious(rg.integers(0, F, (1000, 6)).astype(np.uint32),
                                    rg.random((1000, 6), dtype=np.float32),
                                    rg.standard_normal((1000, 64)), np.full(1000, 0.1, np.float32))
                oc.score_oblivious(x)
                oms = min(oc.score_oblivious(x)[1] for _ in range(3))
                oc.close()
                extras["oblivious_d6"]["scoring"] = {
                    "workload": f"1000 oblivious trees of depth 6 over the same {N} docs x {F} features",
                    "ms": round(oms, 3), "docs_per_s": N / oms * 1e3, "level_tests_per_s": N * 6000.0 / oms * 1e3,
                    "what": "kernel time (k_doc_bins + k_obl_score_s), rows already on the device"}
            except Exception as e:  # the side metric must not take the line down
                extras["oblivious_d6"]["scoring"] = {"error": str(e)}
            # SURVEY 8(d) metric 1 as written: the 500-tree run, iteration 0 discarded, the mean of
            # the rest (later trees are bushier than the headline's first dozens: pi and sigma are
            # the run's own averages)
            extras["steady_500"] = side_run(
                x, labels, qoff, same_set + f", {desc}; 500 trees, the first discarded (SURVEY 8d metric 1)", 499, warm=1)
            xm, lm, qm = synth_mslr(F=F)
            extras["mslr_shaped"] = side_run(
                xm, lm, qm, f"MSLR-WEB10K-shaped stand-in: {len(lm)} docs x {F} features in {len(qm) - 1} ragged "
                f"queries (1..{int(np.diff(qm.astype(np.int64)).max())} documents), 40 sparse count columns; "
                f"100 trees, {desc}", 100)
            # ... and the same stand-in with the reference's DEFAULT `--num-thresholds 0` (every distinct
            # value a threshold, mart.cc:155-158: up to ~700k slots per real-valued column): the
            # pre-sorted path of k_exact.hip, with the oracle's own time per iteration beside it
            # (the CPU leg is two iterations of ~2.5 s on 16 cores)
            try:
                t0 = time.perf_counter()
                from quickrank_amd._capi import Context as _Ctx
                wc = _Ctx(torch.cuda.current_device())
                wc.upload(xm, lm, qm)
                _, wts = wc.build_bins(0)
                wc.synchronize()
                winit = time.perf_counter() - t0
                wc.reset_scores()
                wsteps = 10
                for it in range(wsteps + 2):
                    if it == 2:
                        wc.synchronize()
                        t0 = time.perf_counter()
                    wc.compute_lambdas("NDCG", 10)
                    wc.fit_tree(args.nleaves, 1, True, read=False)
                    wc.update_scores(0.1)
                    wc.metric_last()
                    wc.tree_nodes()
                wc.synchronize()
                wms = (time.perf_counter() - t0) / wsteps * 1e3
                wnd = wc.metric_eval(0, "NDCG", 10)
                wc.close()
                extras["mslr_default_thresholds"] = {
                    "workload": f"the MSLR-shaped stand-in ({len(lm)} docs x {F} features) with --num-thresholds 0: "
                                f"{int(wts.max())} slots in the longest row, {int(wts.sum())} in all; LambdaMART "
                                f"{args.nleaves} leaves, NDCG@10", "path": "pre-sorted per-feature lists (k_exact.hip)",
                    "ms_per_step": wms, "value": len(lm) / wms * 1e3, "unit": "docs/s", "steps": wsteps,
                    "init_s": round(winit, 2), "ndcg10_last": wnd}
                if not args.no_cpu_baseline:
                    extras["mslr_default_thresholds"]["cpu_baseline"] = cpu_baseline_wide(xm, lm, qm, args)
            except Exception as e:  # the side metric must not take the line down
                extras["mslr_default_thresholds"] = {"error": str(e)}
            del xm, lm, qm
            if pc["launches"]:
                cb = built * (F + 12) + sum(int((t["feature"] >= 0).sum()) for t in head.trees[n0:]) * F * 256 * 16
                ca = cb / (pc["total_ms"] * 1e-3) / 1e9
                roof_child = {"bound": "hbm", "kernel": "k_hist_batch (child histograms, all launches of a tree)",
                              "alg_bytes_per_tree": cb / args.extra_steps, "achieved": round(ca, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ca / HBM_PEAK_GBS, 4),
                              "launches_per_tree": pc["launches"] / args.extra_steps,
                              "us_per_tree": round(pc["total_ms"] * 1e3 / args.extra_steps, 1)}
    else:
        # ---- N > 1: the SAME set, strong scaling (BASELINE.json configs[2]) ---------
        q0, q1 = doc_slice(Q, rank, world)
        a, b = int(qoff[q0]), int(qoff[q1])
        runs = {}
        r = mk("docs", x[a:b], labels[a:b], qoff[q0:q1 + 1] - qoff[q0], N, Q)
        r.timed(args.steps, args.warmup)
        runs["document_sharded"] = (r.summary(same_set + f" ({b - a} docs on rank {rank}), " + desc,
                                              f"document sharding x{world}: one int64 all-reduce per growth step of up "
                                              "to two splits (scalars + root + steps + leaves: "
                                              f"{2 + getattr(r.trainer, 'collectives', args.nleaves)} for the last tree; "
                                              f"{2 + args.nleaves} with one split per exchange, QR_DOC_BATCH=0)"),
                                    r.trees_identical_across_ranks())
        r.close()
        r = mk("features", x, labels, qoff, N, Q)
        r.timed(args.steps, args.warmup)
        runs["feature_sharded"] = (r.summary(same_set + " on every rank, " + desc,
                                             f"feature-block sharding x{world} (north_star's layout): best-split "
                                             "records all-gather + go-left mask all-reduce per split"),
                                   r.trees_identical_across_ranks())
        r.close()
        for k, (sm, same) in runs.items():
            if not same:
                raise SystemExit(f"bench.py: ranks built different trees in the {k} layout -- refusing to report")
            sm["trees_identical_across_ranks"] = True
        best = max(runs, key=lambda k: runs[k][0]["value"])
        hs = runs[best][0]
        scaling = "strong"
        extras["strong_layouts"] = {k: v[0] for k, v in runs.items()}
    if not args.no_extras and (multi or args.big_blocks or args.huge_blocks):
        es = args.extra_steps
        if multi:
            # weak scaling: every rank its OWN 1M-document shard (seed 42 + rank)
            xw, lw, qw = synth(Q, DPQ, F, seed=42 + rank)
            r = mk("docs", xw, lw, qw, N * world, Q * world)
            r.timed(es, args.warmup)
            extras["weak_scaling"] = dict(r.summary(f"synthetic {N * world} docs ({N} per GPU) x {F} features, " + desc,
                                                    f"document sharding x{world}"), scaling="weak")
            r.close()
            del xw, lw, qw
        if args.big_blocks:
            # a larger strong-scaling set (blocks of 1M documents, seeds 142 + block): at 1M
            # documents an iteration is a chain of ~40 launches of a few microseconds each and
            # no layout can divide that; here the per-document work dominates
            nb = args.big_blocks
            b0, b1 = doc_slice(nb, rank, world) if multi else (0, nb)
            parts = [synth(Q, DPQ, F, seed=142 + blk) for blk in range(b0, b1)]
            xb = np.concatenate([p_[0] for p_ in parts]) if parts else x[:0]
            lb = np.concatenate([p_[1] for p_ in parts]) if parts else labels[:0]
            qb = np.arange(len(lb) // DPQ + 1, dtype=np.uint64) * DPQ
            del parts
            r = mk("docs" if multi else "single", xb, lb, qb, N * nb, Q * nb)
            r.timed(es, min(args.warmup, 3), prof=not multi)
            extras[f"strong_{nb}M"] = dict(
                r.summary(f"synthetic {N * nb} docs x {F} features x {Q * nb} queries"
                          + (f" ({len(lb)} on rank {rank})" if multi else "") + ", " + desc,
                          f"document sharding x{world}" if multi else "1 GPU"), scaling="strong")
            if not multi:
                # the root histogram launch at this size (HIP events on the launch, as above):
                # nothing of it fits the 256 MB Infinity Cache, and the ~10 us of prologue,
                # flush and dispatch weigh an eighth of what they do at 1M documents
                pb = r.ctx.prof_get()
                r.ctx.prof_enable(False)
                if pb["launches"]:
                    secb = pb["total_ms"] / pb["launches"] * 1e-3
                    # (counters cannot be read inside this process: the PMC passes of this workload
                    # committed under profiles/, quoted only for the workload they were collected on)
                    tr8 = None
                    p8 = newest_profile(f"pmc_hist_{nb}M.json")
                    if p8 and F == 136 and args.nthresholds == 255 and not args.sparse_cols:
                        tr8 = json.load(open(p8)).get("hbm_bytes_per_launch")
                    extras[f"strong_{nb}M"]["roofline"] = {
                        "bound": "hbm", "kernel": "k_hist_root (root histogram build)",
                        "achieved": round(pb["alg_bytes"] / secb / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(pb["alg_bytes"] / secb / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr8,
                        "traffic_source": ("profiles/" + os.path.basename(p8 or "") + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                           "passes of this workload)") if tr8 else None,
                        "alg_bytes_per_launch": pb["alg_bytes"], "avg_launch_us": round(secb * 1e6, 2),
                        "launches": pb["launches"]}
            r.close()
            del xb, lb, qb
        if args.huge_blocks:
            # a strong-scaling set large enough for the per-document kernels to
            # carry the iteration (at 1M documents ~70 % of it is launch chain, which no layout
            # divides; DESIGN.md 6 has the predicted 1 / 2 / 4 / 8 table this falsifies).  Generated
            # on the device, whole 1M-document blocks per rank, document-sharded for N > 1.
            nh = args.huge_blocks
            try:
                h0, h1 = doc_slice(nh, rank, world) if multi else (0, nh)
                xh, lh, qh = synth_device(torch, list(range(h0, h1)), Q, DPQ, F)
                r = mk("docs" if multi else "single", xh, lh, qh, N * nh, Q * nh)
                del xh
                torch.cuda.empty_cache()
                hsteps = max(4, min(es, 8))
                r.timed(hsteps, 2)
                extras[f"strong_{nh}M"] = dict(
                    r.summary(f"synthetic {N * nh} docs x {F} features x {Q * nh} queries, generated on the device"
                              + (f" ({len(lh)} on rank {rank})" if multi else "") + ", " + desc,
                              f"document sharding x{world}" if multi else "1 GPU"), scaling="strong")
                r.close()
                del lh, qh
            except Exception as e:   # (a set this size must not take the line down: say what happened)
                if multi:
                    raise
                extras[f"strong_{nh}M"] = {"error": str(e)}

    scoring = None
    if not args.no_scoring:   # every rank takes part (its shard of the documents)
        scoring = scoring_metric(None, args, torch, rank, world, dist if world > 1 else None)

    if rank == 0:
        out = {
            "metric": "docs/sec per LambdaMART boosting iter (1Mx136)",
            "value": hs["value"], "unit": "docs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": hs["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {k: hs[k] for k in ("workload", "parallelism", "ndcg10_last", "sigma_built",
                                          "sigma_reference_left", "pi", "collectives", "collectives_per_tree",
                                          "collective_bytes_per_tree")},
            "roofline": roof,
        }
        if not multi:
            # the quality gate next to cpu_baseline.ndcg10_after: NDCG@10 of the training set after
            # `cpu_iters` trees on both sides (the device's list holds the metric BEFORE tree i)
            if len(head.ndcg) > args.cpu_iters:
                out["config"][f"ndcg10_after_{args.cpu_iters}"] = head.ndcg[args.cpu_iters]
            # host rows -> HBM and Mart::init, outside the timed region (reported apart)
            out["config"]["h2d_ms"] = round(head.h2d_ms, 1)
            out["config"]["init_ms"] = round(head.init_ms, 1)
        if roof_iter is not None:
            out["roofline_iteration"] = roof_iter
        if roof_child is not None:
            out["roofline_child_hist"] = roof_child
        out.update(extras)
        if world == 1 and not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(x, labels, qoff, args)
        if scoring is not None:
            if world == 1 and not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                from score_bench import make_model
                scoring["cpu_baseline"] = cpu_scoring_baseline(args, make_model)
            out["ensemble_scoring"] = scoring
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if not multi:
        head.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
