import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import oracle, quickrank_amd as qr
oracle.build(ref=False)
for n in (2000, 3300, 3400, 3600, 9000, 20000):
    rng = np.random.default_rng(n)
    N = n + 37
    x = rng.random((N, 4), dtype=np.float32)
    labels = rng.integers(0, 5, N).astype(np.float32)
    qoff = np.array([0, n, N], np.uint64)
    c = qr.Context(0); c.upload(x, labels, qoff); c.build_bins(16)
    scores = np.round(rng.standard_normal(N), 1)
    c.set_scores(scores)
    try:
        c.compute_lambdas("NDCG", 10)
        lam, w = c.get_pseudo()
        ol, ow = oracle.lambdas(labels, scores, qoff, 10, 1)
        print(n, "ok", np.allclose(lam, ol, rtol=1e-10, atol=1e-14), np.allclose(w, ow, rtol=1e-10, atol=1e-14),
              abs(c.metric_last() - oracle.eval_dataset(labels, scores, qoff, 10)) < 1e-13)
    except qr.QrError as e:
        print(n, "refused:", e)
    c.close()
