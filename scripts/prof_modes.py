import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
from bench import synth, synth_mslr
from quickrank_amd._capi import Context
mode = sys.argv[1]
if mode == "obl":
    x, l, q = synth(10000, 100, 136)
else:
    x, l, q = synth_mslr()
c = Context(0); c.upload(x, l, q); c.build_bins(255); c.reset_scores()
pending = False
for it in range(12):
    c.compute_lambdas("NDCG", 10)
    if pending: c.tree_nodes()
    if mode == "obl": c.fit_oblivious(6, 1, True, read=False)
    else: c.fit_tree(10, 1, True, read=False)
    c.update_scores(0.1); pending = True
    c.metric_last()
c.tree_nodes(); c.synchronize()
