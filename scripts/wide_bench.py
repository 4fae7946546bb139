"""Time per boosting iteration with more than 255 thresholds (k_wide.hip) next to the u8
path, on the MSLR-shaped stand-in (TEST TOOL, GPU box):  python scripts/wide_bench.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
from datagen import make_mslr_like
from quickrank_amd._capi import Context
x, labels, qoff = make_mslr_like()
NIT = int(os.environ.get("WB_ITERS", "8"))      # iterations per setting (the first 3 are not timed)
for nthr in [int(v) for v in os.environ.get("WB_NTHR", "255,1024,4096,0").split(",")]:
    c = Context(0)
    c.upload(x, labels, qoff)
    t0 = time.perf_counter()
    thr, ts = c.build_bins(nthr)
    c.synchronize()
    t_init = time.perf_counter() - t0
    c.reset_scores()
    for it in range(NIT):
        if it == 3:
            c.synchronize(); t0 = time.perf_counter()
        c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True, read=False); c.update_scores(0.1)
        c.metric_last(); c.tree_nodes()
    c.synchronize()
    ms = (time.perf_counter() - t0) / (NIT - 3) * 1e3
    print(f"nthresholds {nthr}: wide={c.wide} slots per feature max {int(ts.max())} total {int(ts.sum())}, "
          f"init {t_init:.2f} s, {ms:.2f} ms per iteration, NDCG {c.metric_eval(0):.6f}", flush=True)
    c.close()
