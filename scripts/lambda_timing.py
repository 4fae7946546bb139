import os, sys, shutil
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import quickrank_amd.build as b
b.LIB = "/root/repo/quickrank_amd/lib/libqr_timing.so"
import quickrank_amd._capi as capi
from bench import synth
x, labels, qoff = synth(10000, 100, 136)
c = capi.Context(0)
c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
for it in range(4):
    c.compute_lambdas("NDCG", 10); c.synchronize()
    c.fit_tree(10, 1, True); c.update_scores(0.1)
