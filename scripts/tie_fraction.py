import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import synth
from quickrank_amd._capi import Context
x, labels, qoff = synth(10000, 100, 136)
c = Context(0); c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
for it in range(60):
    c.compute_lambdas("NDCG", 10); c.fit_tree(10, 1, True); c.update_scores(0.1)
    if it in (0, 1, 2, 4, 9, 19, 29, 39, 59):
        s = c.get_scores().reshape(10000, 100)
        ss = np.sort(s, axis=1)
        tied = (np.diff(ss, axis=1) == 0).any(axis=1).mean()
        print(f"after tree {it+1}: fraction of queries with a tied pair = {tied:.3f}")
