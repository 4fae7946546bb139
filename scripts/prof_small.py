import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from datagen import make_dataset
import quickrank_amd as qr
x, labels, qoff = make_dataset(nq=30, docs_per_query=100, F=136, seed=1)
c = qr.Context(0)
c.upload(x, labels, qoff)
c.build_bins(255)
c.reset_scores()
for it in range(3):
    t0 = time.perf_counter(); c.compute_lambdas("NDCG", 10); c.synchronize(); t1 = time.perf_counter()
    nodes = c.fit_tree(16, 1, True); t2 = time.perf_counter()
    c.update_scores(0.1); m = c.metric_eval(0, "NDCG", 10); t3 = time.perf_counter()
    print(f"iter {it}: lambda {t1-t0:.4f}s tree {t2-t1:.4f}s update+eval {t3-t2:.4f}s ndcg {m:.4f} nodes {len(nodes)}")
