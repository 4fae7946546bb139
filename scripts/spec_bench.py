import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
torch.cuda.init()
from bench import synth
from quickrank_amd._capi import Context
for nq in (10000, 20000, 40000, 80000):
    parts = [synth(10000, 100, 136, seed=142 + b) for b in range(nq // 10000)]
    x = np.concatenate([p[0] for p in parts]); l = np.concatenate([p[1] for p in parts])
    q = np.arange(nq + 1, dtype=np.uint64) * 100
    c = Context(0); c.upload(x, l, q); c.build_bins(255); c.reset_scores()
    for it in range(25):
        if it == 5:
            c.synchronize(); t0 = time.perf_counter()
        c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True, read=False); c.update_scores(0.1); c.metric_last(); c.tree_nodes()
    c.synchronize()
    print(nq * 100, "docs:", round((time.perf_counter() - t0) / 20 * 1e3, 3), "ms per iteration", flush=True)
    c.close()
