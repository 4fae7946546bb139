"""Oblivious-LambdaMART iteration time (BASELINE.json config 4: depth 6) on the bench data."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bench import synth
from quickrank_amd._capi import Context
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x, labels, qoff = synth(10000, 100, 136)
c = Context(0); c.upload(x, labels, qoff); c.build_bins(255); c.reset_scores()
pending = False
def step():  # (bench.py's order: a tree's records are fetched under the next lambda pass)
    global pending, t
    c.compute_lambdas("NDCG", 10)
    if pending: t = c.tree_nodes()
    c.fit_oblivious(depth, 1, True, read=False)
    c.update_scores(0.1)
    pending = True
for _ in range(3): step()
c.synchronize(); t0 = time.perf_counter()
K = 40
for _ in range(K): step()
t = c.tree_nodes()
c.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"oblivious depth {depth}: {dt*1e3:.3f} ms/iter, {len(labels)/dt:.3e} docs/s, nodes {len(t)}, ndcg {c.metric_last():.6f}")
