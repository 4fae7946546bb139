"""Leaf counts on both sides of the batched-growth limit (TEST TOOL): trees of 200 to
500 leaves against the oracle -- up to 255 leaves the device applies two splits per
step, beyond that one.  Run on a GPU box:
    python tests/tools/big_leaves_check.py"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import oracle
from datagen import make_dataset
from parity_util import assert_tree_parity
from quickrank_amd.trainer import Mart
oracle.build(ref=False)
for nl in (200, 255, 256, 300, 500):
    x, labels, qoff = make_dataset(nq=300, docs_per_query=40, F=20, seed=nl)
    kw = dict(ntrees=2, shrinkage=0.1, nthresholds=64, nleaves=nl, minls=1, esr=0)
    gm = Mart(algo="LAMBDAMART", **kw).learn(x, labels, qoff)
    om = oracle.train(x, labels, qoff, algo="LAMBDAMART", **kw)
    tr = oracle.Trainer(x, 64)
    for t in range(2):
        n = int(om["nnodes"][t])
        try:
            assert_tree_parity(tr.stmap, om["nodes"][t][:n], gm.ensemble.trees[t][:n], value_rtol=1e-6)
            print(nl, t, "ok nodes", n)
        except AssertionError as e:
            print(nl, t, "MISMATCH", str(e)[:200])
