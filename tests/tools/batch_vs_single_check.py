"""Batched growth against one split per step on a set large enough for several flushes
per histogram workgroup (TEST TOOL; 4M documents -> 3 flushes): the two code paths
share only the accumulation loop, so identical trees (modulo the nodes' f64 bookkeeping sums,
see parity_util.assert_same_tree_records) check the feature-major flush,
k_redscan's slot bookkeeping and the per-workgroup descriptors.  Run on a GPU box:
    python tests/tools/batch_vs_single_check.py [queries]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from bench import synth
from quickrank_amd._capi import Context

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
x, labels, qoff = synth(nq, 100, 136)
out = []
for no_batch in (False, True):
    if no_batch:
        os.environ["QR_NO_BATCH"] = "1"
    else:
        os.environ.pop("QR_NO_BATCH", None)
    c = Context(0)
    c.upload(x, labels, qoff)
    c.build_bins(255)
    c.reset_scores()
    trees = []
    for it in range(3):
        c.compute_lambdas("NDCG", 10)
        trees.append(c.fit_tree(10, 1, True))
        c.update_scores(0.1)
    out.append((trees, c.get_scores()))
    c.close()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_util import assert_same_tree_records
for t, (a, b) in enumerate(zip(out[0][0], out[1][0])):
    # (round 3: batched growth takes a node's f64 sums from its histogram pass, the one-split path
    # from its partition pass -- `deviance` of every node and `value` of internal nodes agree to
    # rounding, everything else bit for bit)
    assert_same_tree_records(a, b, node_sums_exact=False, where=t)
assert np.array_equal(out[0][1], out[1][1])
print(f"{len(labels)} docs: batched == one split per step, 3 trees: structure, leaf outputs, scores bit-identical")
