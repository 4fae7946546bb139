"""LambdaMART iteration time on the MSLR-shaped stand-in of BASELINE.json config 1 (bench.py's
`mslr_shaped` workload and call order), for A/B runs of the lambda pass's size-class launches
(TEST TOOL, GPU box):  [QR_HIP_LIB=...] python scripts/mslr_bench.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synth_mslr
from quickrank_amd._capi import Context
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
x, labels, qoff = synth_mslr(F=136)
c = Context(0); c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
pending = False
def step():  # (a tree's records are fetched under the next lambda pass, the metric after the tree is enqueued)
    global pending
    c.compute_lambdas("NDCG", 10)
    if pending: c.tree_nodes()
    c.fit_tree(10, 1, True, read=False)
    c.update_scores(0.1)
    pending = True
    c.metric_last()
for _ in range(3): step()
c.synchronize(); t0 = time.perf_counter()
for _ in range(K): step()
c.tree_nodes(); c.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"mslr-shaped ({len(labels)} docs, {len(qoff) - 1} queries): {dt * 1e3:.4f} ms/iter over {K} iterations, ndcg {c.metric_last():.6f}")
